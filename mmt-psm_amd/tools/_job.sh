cd $GRAFT_REPO_ROOT
bash mmt-psm_amd/tools/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
