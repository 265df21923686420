#!/bin/bash
# profiles/collect_r05.sh <tag> [quick] -- round 6 collection; run on the GPU box (via gpurun) from the repo root; writes gpurun_out/<tag>/:
#   pmc_latest.json            HBM-side bytes per kernel launch (2 x FETCH_SIZE + WRITE_SIZE; separate --pmc passes, counters only) of every
#                              kernel of the substep at 256 / 512 / plume1024 / 128, stamped with the kernel-source hash bench.py checks
#   pmc_<config>.json          the raw per-kernel PMC means of those passes (256: plus the TA / SQ / LDS groups)
#   bench_<config>.json        python bench.py [--config c] run AFTER pmc_latest.json is in place (256 with the CPU baseline)
#   kernel_stats_<config>.csv  rocprofv3 --kernel-trace --stats of bench.py at 256 / 128 / 512 / plume1024
#   full_256.json / kernel_stats_full256.csv   python bench.py --full (the whole Compute_Sim substep, SURVEY 8d's last row) and its rocprofv3 kernel statistics
#   `quick`: the memory-side passes and bench lines only
set -u
tag=${1:-r06}; quick=${2:-}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in 256 512 plume1024 128 plume 64; do
	dirs=""
	groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
	if [ "$c" = 256 ] && [ -z "$quick" ]; then
		groups+=("TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
		         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
	fi
	if [ "$c" = 512 ] && [ -z "$quick" ]; then  # (the HBM regime: the unit counters too, for the floors table of DESIGN.md section 7)
		groups+=("TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU")
	fi
	steps=3; [ "$c" = 512 ] && steps=2
	for grp in "${groups[@]}"; do
		d=$out/pmc_${c}_$(echo $grp | tr ' ' '_' | cut -c1-40)
		rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --config $c --steps $steps --warmup 1 --no-cpu-baseline --no-strong > $d.log 2>&1
		dirs="$dirs $d"
	done
	python3 $root/profiles/summarize_pmc.py $out/pmc_$c.json $dirs > $out/pmc_${c}_summary.txt
done
# the full Compute_Sim substep at 256^3 (round 6: the fused divergence / combustion / buoyancy launch and the q4 advection): memory side and unit counters
if [ -z "$quick" ]; then
	dirs=""
	for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
		d=$out/pmc_full256_$(echo $grp | tr ' ' '_' | cut -c1-40)
		rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --full --steps 3 --warmup 1 > $d.log 2>&1
		dirs="$dirs $d"
	done
	python3 $root/profiles/summarize_pmc.py $out/pmc_full256.json $dirs > $out/pmc_full256_summary.txt
fi
python3 - "$out" <<'PY'
import json, sys, os
out = sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bench import kernel_source_sha16
j = {"kernel_source_sha16": kernel_source_sha16(),
     "correction": "hbm_bytes_per_kernel_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, counters only: bench.py --config <c> --steps 3 (512: 2) --warmup 1 --no-cpu-baseline --no-strong (profiles/collect_r05.sh)",
     "configs": {}}
for c in ("256", "512", "plume1024", "128", "plume", "64"):
    p = json.load(open(f"{out}/pmc_{c}.json"))
    ks = {}
    for name, v in p.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        short = name.replace("hns::", "").split("<")[0]
        # several instantiations of one kernel (the first launch of a solve knows p = 0): keep the one launched most often
        if short in ks and ks[short]["launches_profiled"] >= v["FETCH_SIZE"]["launches"]:
            continue
        hit, miss = v.get("TCC_HIT_sum", {}).get("mean"), v.get("TCC_MISS_sum", {}).get("mean")
        ks[short] = {"instantiation": name, "launches_profiled": v["FETCH_SIZE"]["launches"], "fetch_size_kb": v["FETCH_SIZE"]["mean"], "write_size_kb": v["WRITE_SIZE"]["mean"],
                     "hbm_bytes_per_kernel_launch": 1024.0 * (2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]),
                     "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss else None}
    j["configs"][c] = {"kernels": ks}
json.dump(j, open(out + "/pmc_latest.json", "w"), indent=1)
print(json.dumps({c: {k: round(v["hbm_bytes_per_kernel_launch"] / 1e6, 1) for k, v in j["configs"][c]["kernels"].items() if k.startswith(("k_rbgs", "k_adv", "k_div", "k_sub"))} for c in j["configs"]}))
PY
cp $out/pmc_latest.json $root/profiles/pmc_latest.json
cd $root
python3 bench.py > $out/bench_256.json 2> $out/bench_256.err   # (the default line: with cpu_baseline and, as strong_scaling, config 5 on the one GPU)
for c in 128 64 plume plume1024; do python3 bench.py --config $c --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench_other.err; done
python3 bench.py --config 512 --steps 5 --no-cpu-baseline > $out/bench_512.json 2>> $out/bench_other.err
if [ -z "$quick" ]; then
	python3 bench.py --full > $out/full_256.json 2>> $out/bench_other.err
	python3 bench.py --full --config 512 --steps 5 > $out/full_512.json 2>> $out/bench_other.err
	python3 bench.py --full --config 128 > $out/full_128.json 2>> $out/bench_other.err
	python3 bench.py --cook > $out/cook_256.json 2>> $out/bench_other.err
	python3 bench.py --cook --config 128 > $out/cook_128.json 2>> $out/bench_other.err
	cd /tmp
	for c in 256 128 512 plume1024; do
		rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 $root/bench.py --config $c --no-cpu-baseline --no-strong > $out/stats_$c.log 2>&1
		cp $(find $out/stats_$c -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$c.csv
		rm -rf $out/stats_$c
	done
	rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_full -- python3 $root/bench.py --full --steps 10 > $out/stats_full.log 2>&1
	cp $(find $out/stats_full -name "*kernel_stats.csv" | head -1) $out/kernel_stats_full256.csv
	rm -rf $out/stats_full
fi
find $out -name "*_counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
for c in ("256", "512", "plume1024", "128", "64", "plume"):
    try:
        j = json.loads(open(f"{out}/bench_{c}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(c, "substeps/s", round(j["value"], 1), "| SOR frac", round(r["frac"], 3), "compulsory", r.get("frac_compulsory") and round(r["frac_compulsory"], 3), "moved", r.get("frac_moved") and round(r["frac_moved"], 3),
              "| substep frac", round(r["substep"]["frac"], 3), "|", {k.split("<")[0]: (round(1e3 * v["ms_per_launch"], 1), v.get("frac_moved") and round(v["frac_moved"], 2)) for k, v in r["kernels"].items()})
    except Exception as e:
        print(c, "?", e)
PY
