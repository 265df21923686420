#!/bin/bash
# profiles/tools/adopt_collection.sh <tag>: copy what profiles/collect_r05.sh <tag> left in gpurun_out/<tag>/ (scratch) into profiles/ (tracked) under the round's names
set -e
cd "$(dirname "$0")/../.."
tag=${1:-r05_final}; src=gpurun_out/$tag
for c in 256 128 64 plume plume1024 512; do
	cp $src/bench_$c.json profiles/r05_final_bench_$c.json
	cp $src/pmc_$c.json profiles/r05_final_bench${c}_pmc.json
done
for c in 256 128 512; do cp $src/kernel_stats_$c.csv profiles/r05_final_bench${c}_kernel_stats.csv; done
cp $src/kernel_stats_plume1024.csv profiles/r05_final_bench_plume1024_kernel_stats.csv
cp $src/full_256.json profiles/r05_final_full256.json
cp $src/kernel_stats_full256.csv profiles/r05_final_full256_kernel_stats.csv
cp $src/cook_256.json profiles/r05_final_cook256.json
cp $src/cook_128.json profiles/r05_final_cook128.json
cp $src/pmc_latest.json profiles/pmc_latest.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench import kernel_source_sha16
print("pmc_latest stamp", json.load(open("profiles/pmc_latest.json"))["kernel_source_sha16"], "sources", kernel_source_sha16())
PY
