cd $GRAFT_REPO_ROOT
for v in 0 256 0 256; do echo "MMT_STRIP_PERSISTENT=$v"; MMT_STRIP_PERSISTENT=$v python mmt-psm_amd/tools/strip_k_scaling.py 2>&1 | grep "N="; done > gpurun_out/strip_persist.txt
