#!/bin/bash
# profiles/pmc_stamp.sh <tag> -- the two memory-side PMC passes of collect.sh only (FETCH_SIZE, WRITE_SIZE; separate passes, never
# combined with tracing) and gpurun_out/<tag>/pmc_latest.json stamped with the hash of the current kernel sources: what
# bench.py's roofline.traffic quotes. Run on the GPU box from the repo root; copy the result to profiles/pmc_latest.json.
set -u
tag=${1:-stamp}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
	d=$out/pmc_$grp
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
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 (profiles/pmc_stamp.sh = those passes of profiles/collect.sh)",
     "algorithmic_bytes_per_launch": 201326592}
json.dump(j, open(out + "/pmc_latest.json", "w"), indent=1)
print(json.dumps(j))
PY
