#!/bin/bash
# usage: profiles/micro/pmc_sor_block.sh <tag> <config> [name=value options of sor_one.py ...]
# PMC passes (separate runs, counters only) of the SOR sweep alone under the given options; prints per-kernel means of the SOR kernels.
tag=$1; cfg=$2; shift 2
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_sb_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
	d=$out/$(echo $grp | tr ' ' '_' | cut -c1-30)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/profiles/micro/sor_one.py $cfg "$@" > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs | grep -A40 "k_rbgs_block<\|k_rbgs_pair<false>"
