#!/bin/bash
# usage: profiles/micro/pmc_advect.sh <tag> [HNS_LIBRARY path] -- texture-addresser / L1 counters of the two advection kernels at 256^3 (one PMC pass, counters only)
tag=$1; lib=${2:-}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_adv_$tag
rm -rf $out; mkdir -p $out
[ -n "$lib" ] && export HNS_LIBRARY=$lib
cd /tmp && export TMPDIR=/tmp
dirs=""
for grp in "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
	d=$out/$(echo $grp | tr ' ' '_' | cut -c1-30)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/profiles/micro/advect_stage_times.py 256 > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs | grep -A8 "k_advect_vector_n\|k_advect_scalars_n"
