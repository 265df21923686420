mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r02a/tests.log
python bench.py > gpurun_out/r02a/bench256.json 2> gpurun_out/r02a/bench256.err
python bench.py --config plume1024 --no-cpu-baseline > gpurun_out/r02a/bench_plume1024.json 2>> gpurun_out/r02a/bench256.err
python bench.py --config 128 --no-cpu-baseline > gpurun_out/r02a/bench_128.json 2>> gpurun_out/r02a/bench256.err
python bench.py --config 64 --no-cpu-baseline > gpurun_out/r02a/bench_64.json 2>> gpurun_out/r02a/bench256.err
python bench.py --config 512 --no-cpu-baseline --steps 5 > gpurun_out/r02a/bench_512.json 2>> gpurun_out/r02a/bench256.err
cat gpurun_out/r02a/tests.log | tail -8
