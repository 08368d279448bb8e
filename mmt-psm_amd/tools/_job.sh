cd $GRAFT_REPO_ROOT
MMT_BW_INLINE=1 python mmt-psm_amd/tools/host_profile.py 2>/dev/null | awk '/Ordered by: cumulative/{f=1} f' | head -75
