#!/bin/bash
# usage: profiles/kstats_any.sh <tag> <python script and args...> -- per-kernel average durations of that run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out.log 2>&1
tail -1 $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$tag" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(sys.argv[2], r["Name"][:58], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "pct", r["Percentage"])
PY
