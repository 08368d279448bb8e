cd /root/repo
MMT_TEACHER_NO_READBACK=0 python mmt-psm_amd/tools/host_phases.py 2>/dev/null | tail -34 > gpurun_out/phases_readback.txt
MMT_TEACHER_NO_READBACK=1 python mmt-psm_amd/tools/host_phases.py 2>/dev/null | tail -34 > gpurun_out/phases_fixedcap.txt
paste -d'|' gpurun_out/phases_readback.txt gpurun_out/phases_fixedcap.txt | cut -c1-200
