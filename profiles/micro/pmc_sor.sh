#!/bin/bash
# usage: profiles/micro/pmc_sor.sh <tag> <config> [HNS_LIBRARY path] -- memory-side PMC passes of the SOR sweep alone (sor_one.py)
tag=$1; cfg=$2; lib=${3:-}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_sor_$tag
rm -rf $out; mkdir -p $out
[ -n "$lib" ] && export HNS_LIBRARY=$lib
cd /tmp && export TMPDIR=/tmp
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_TAG_STALL_sum TCC_BUBBLE_sum"; do
	d=$out/$(echo $grp | tr ' ' '_' | cut -c1-30)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/profiles/micro/sor_one.py $cfg > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs | grep -A14 "k_rbgs_pair<false>\|k_rbgs_tile<false>"
