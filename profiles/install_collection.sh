#!/bin/bash
# profiles/install_collection.sh [tag] -- copy what profiles/collect_r06.sh wrote under gpurun_out/<tag>/ (merged back from the GPU box) into profiles/ under the names
# profiles/README.md lists (run here, in the build container, after the gpurun call)
set -eu
tag=${1:-r06}; src=gpurun_out/$tag; cd "$(dirname "$0")/.."
cp $src/pmc_latest.json profiles/pmc_latest.json
for c in 256 128 64 plume plume1024 512; do cp $src/bench_$c.json profiles/r06_final_bench_$c.json; done
for c in 256 128 512; do cp $src/kernel_stats_$c.csv profiles/r06_final_bench${c}_kernel_stats.csv; done
cp $src/kernel_stats_plume1024.csv profiles/r06_final_bench_plume1024_kernel_stats.csv
cp $src/kernel_stats_full256.csv profiles/r06_full256_kernel_stats.csv
for c in 256 512 128 plume1024 full256; do cp $src/pmc_$c.json profiles/r06_final_pmc_$c.json; done
for c in 256 512 128; do cp $src/full_$c.json profiles/r06_final_full_$c.json; done
for c in 256 128; do cp $src/cook_$c.json profiles/r06_final_cook_$c.json; done
python3 -c "
import json, bench
s = json.load(open('profiles/pmc_latest.json'))['kernel_source_sha16']
print('pmc_latest.json stamp', s, '| kernel sources here', bench.kernel_source_sha16(), '|', 'MATCH' if s == bench.kernel_source_sha16() else 'STALE: bench.py will print traffic = null')"
