mkdir -p gpurun_out/r04f; rm -f gpurun_out/r04f/*
for i in 1 2 3; do
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04f/pytest$i.txt 2>&1; echo rc $? >> gpurun_out/r04f/pytest$i.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r04f/pytest$i.txt | tail -2 | tr '\n' ' '; echo
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
