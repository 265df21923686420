#!/bin/bash
# profiles/collect_r03.sh <tag> -- round 3 collection; run on the GPU box (via gpurun) from the repo root; writes gpurun_out/<tag>/:
#   bench_<config>.json        python bench.py [--config c]            (256 with the CPU baseline)
#   kernel_stats_<config>.csv  rocprofv3 --kernel-trace --stats of bench.py at 256 / 128 / plume1024
#   pmc_256.json + pmc_latest.json   per-kernel PMC means at 256^3, one counter group per pass (counters only, never combined with tracing)
#   sor_forms.txt              SOR forms against grid size (profiles/micro/sor_block_check.py)
set -u
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd $root
python3 bench.py > $out/bench_256.json 2> $out/bench_256.err
for c in 128 64 plume plume1024; do python3 bench.py --config $c --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_other.err; done
python3 bench.py --config 512 --steps 5 --no-cpu-baseline > $out/bench_512.json 2>> $out/bench_other.err
python3 bench.py --cook > $out/cook_256.json 2>> $out/bench_other.err
python3 bench.py --cook --config 128 > $out/cook_128.json 2>> $out/bench_other.err
python3 profiles/micro/sor_block_check.py d32 d48 d64 d80 d96 d112 d128 d160 d192 d224 d256 d288 d320 d384 512 plume plume1024 2>> $out/bench_other.err | grep "us /" > $out/sor_forms.txt
cd /tmp && export TMPDIR=/tmp
for c in 256 128 plume1024; do
	rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 $root/bench.py --config $c --no-cpu-baseline > $out/stats_$c.log 2>&1
	cp $(find $out/stats_$c -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$c.csv
done
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
	d=$out/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc_256.json $dirs > $out/pmc_256_summary.txt
python3 - "$out" <<'PY'
import json, sys, os
out = sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bench import kernel_source_sha16
p = json.load(open(out + "/pmc_256.json"))
name = [k for k in p if k.startswith("hns::k_rbgs_block<2, 2, false")] or [k for k in p if k.startswith("hns::k_rbgs_pair<false>")]
k = p[name[0]]
fetch_kb, write_kb = k["FETCH_SIZE"]["mean"], k["WRITE_SIZE"]["mean"]
per_launch = 1024.0 * (2.0 * fetch_kb + write_kb)
ipl = 2 if "block" in name[0] else 1
j = {"config": "256", "kernel": "k_rbgs_block" if ipl == 2 else "k_rbgs_pair", "kernel_source_sha16": kernel_source_sha16(), "iterations_per_kernel_launch": ipl,
     "hbm_bytes_per_kernel_launch": per_launch, "hbm_bytes_per_launch": per_launch / ipl, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
     "correction": "FETCH_SIZE x2 (gfx950: reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; hbm_bytes_per_launch is per red+black ITERATION "
                   "(the unit of roofline.achieved): a kernel launch of the temporally blocked form holds iterations_per_kernel_launch of them",
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 (profiles/collect_r03.sh)",
     "algorithmic_bytes_per_launch": 201326592}
json.dump(j, open(out + "/pmc_latest.json", "w"), indent=1)
print(json.dumps(j))
PY
head -12 $out/kernel_stats_256.csv | cut -c1-150
cat $out/bench_256.json | cut -c1-400
