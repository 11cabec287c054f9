bash scripts/profile_round.sh r05_m > gpurun_out/r05_m_profile.log 2>&1
tail -40 gpurun_out/r05_m_profile.log
bash scripts/lab/cfg_kt.sh sprint_joint 32 40 > gpurun_out/r05_m_cfg5_kernel_stats.txt 2>&1
head -5 gpurun_out/r05_m_cfg5_kernel_stats.txt
