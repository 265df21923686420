mkdir -p gpurun_out/r02b
python -m pytest tests/test_dist_gpu.py tests/test_gridbuild_gpu.py tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r02b/tests.log
tail -5 gpurun_out/r02b/tests.log
for k in 4 2 1; do python profiles/micro/dist_overhead.py 256 2 $k >> gpurun_out/r02b/overhead.jsonl 2>> gpurun_out/r02b/err.log; done
python profiles/micro/dist_overhead.py 128 2 4 >> gpurun_out/r02b/overhead.jsonl 2>> gpurun_out/r02b/err.log
python profiles/micro/dist_overhead.py plume1024 8 4 --partition >> gpurun_out/r02b/overhead.jsonl 2>> gpurun_out/r02b/err.log
cat gpurun_out/r02b/overhead.jsonl
grep -v amdgpu.ids gpurun_out/r02b/err.log | tail -5
