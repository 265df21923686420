#!/bin/bash
# usage: profiles/pmc_one.sh <tag> <config> -- FETCH/WRITE/L2-hit PMC passes of bench.py at one configuration; prints per-kernel means
tag=$1; cfg=$2
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
	d=$out/$(echo $grp | tr ' ' '_' | cut -c1-30)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --config $cfg > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs | grep -A9 "k_rbgs_pair<false>"
