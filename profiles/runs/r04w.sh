timeout 2400 bash profiles/collect.sh r02_final3 > gpurun_out/collect.log 2>&1; echo rc $?
tail -3 gpurun_out/collect.log | cut -c1-300
