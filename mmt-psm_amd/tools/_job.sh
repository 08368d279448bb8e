cd $GRAFT_REPO_ROOT
bash mmt-psm_amd/tools/make_profiles.sh > gpurun_out/prof_log.txt 2>&1
tail -40 gpurun_out/prof_log.txt
