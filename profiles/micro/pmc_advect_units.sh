#!/bin/bash
# usage: profiles/micro/pmc_advect_units.sh <tag> -- which unit the advection kernels keep busy at 256^3: per-wave busy time of the VALU, the
# vector memory path and the LDS (SQ_ACTIVE_INST_*, SQ_INST_CYCLES_*), the texture addresser, and the instruction counts (counters only)
tag=$1
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_advu_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*VALU[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*\|SQ_THREAD_CYCLES[A-Z_0-9]*" $out/counters.txt | sort -u > $out/sq_counters.txt
dirs=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum"; do
	d=$out/$(echo $grp | tr ' ' '_' | cut -c1-30)
	rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/profiles/micro/advect_stage_times.py 256 > $d.log 2>&1
	dirs="$dirs $d"
done
python3 $root/profiles/summarize_pmc.py $out/pmc.json $dirs | grep -A14 "k_advect_vector_n\|k_advect_scalars_n"
cat $out/sq_counters.txt | tr '\n' ' '
