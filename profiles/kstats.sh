#!/bin/bash
# usage: profiles/kstats.sh <tag> [env assignments...] -- prints per-kernel average durations of one bench.py run
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(sys.argv[2], r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
