cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python mmt-psm_amd/tools/bench_f16x2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof/precision_strip.txt
tail -40 gpurun_out/prof/precision_strip.txt | cut -c1-250
