#!/bin/bash
# profiles/collect.sh <tag> -- run on the GPU box (via gpurun) from the repo root; writes everything under gpurun_out/<tag>/:
#   bench_<config>.json             python bench.py [--config c]   (default config with the CPU baseline)
#   kernel_stats.csv                rocprofv3 --kernel-trace --stats of the default bench command
#   pmc.json                        per-kernel PMC means, one counter group per pass (never combined with other tracing)
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
tag=${1:-final}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd $root
python3 bench.py > $out/bench_256.json 2> $out/bench_256.err
for c in 128 64 plume plume1024; do python3 bench.py --config $c --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_other.err; done
python3 bench.py --config 512 --steps 5 --no-cpu-baseline > $out/bench_512.json 2>> $out/bench_other.err
# one rank of a multi-GPU run before any wire time (local / loopback transports of hns_dist), and against emulated wire time
for a in "256 2 4" "256 2 2" "256 2 1" "128 2 4" "128 2 1" "plume1024 8 4 --partition" "plume1024 8 1 --partition"; do python3 profiles/micro/dist_overhead.py $a >> $out/dist_overhead.jsonl 2>> $out/bench_other.err; done
for w in 0 10 20 40; do python3 profiles/micro/dist_profile.py rank 4 $w 2>> $out/bench_other.err | grep "ms per substep" >> $out/dist_wire_sweep.txt; done
for k in 1 2 4; do python3 profiles/micro/dist_profile.py rank $k 0 rccl 2>> $out/bench_other.err | grep "ms per substep" >> $out/dist_rccl_loopback.txt; done
python3 profiles/micro/dist_profile.py rank 1 0 2>> $out/bench_other.err | grep "ms per substep" >> $out/dist_wire_sweep.txt
python3 profiles/micro/sor_sizes.py > $out/sor_sizes.txt 2>> $out/bench_other.err
# memory-side PMC of the SOR sweep outside the Infinity Cache (blocked kernel at 512^3, pair kernel on the 66k-leaf plume)
for c in 512 plume1024; do bash profiles/micro/pmc_sor.sh ${tag}_$c $c > $out/pmc_sor_$c.txt 2>&1; cp $root/gpurun_out/pmc_sor_${tag}_$c/pmc.json $out/pmc_sor_$c.json; done
python3 bench.py --cook > $out/cook_256.json 2>> $out/bench_other.err
python3 bench.py --cook --config 128 > $out/cook_128.json 2>> $out/bench_other.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
	d=$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs > $out/pmc_summary.txt
python3 - "$out" <<'PY'
import json, sys, os
out = sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bench import kernel_source_sha16
p = json.load(open(out + "/pmc.json"))
k = p.get("hns::k_rbgs_pair<false>") or p["hns::k_rbgs_pair"]
fetch_kb, write_kb = k["FETCH_SIZE"]["mean"], k["WRITE_SIZE"]["mean"]
j = {"config": "256", "kernel": "k_rbgs_pair", "kernel_source_sha16": kernel_source_sha16(), "hbm_bytes_per_launch": 1024.0 * (2.0 * fetch_kb + write_kb), "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
     "correction": "FETCH_SIZE x2 (gfx950: reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (= 4 B/voxel exactly)",
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 (profiles/collect.sh)",
     "algorithmic_bytes_per_launch": 201326592}
json.dump(j, open(out + "/pmc_latest.json", "w"), indent=1)
print(json.dumps(j))
PY
head -8 $out/kernel_stats.csv | cut -c1-150
cat $out/bench_256.json | cut -c1-300
