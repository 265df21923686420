# round 6: shell cells of the 10^3 box out of a compile-time table (libhns_tab.so) against the arithmetic (libhns_prev.so = the committed sources): kernels and bench lines alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev tab; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/advect_two_libs.py 256 128 plume1024 --amp=96,400 >> $O/ab.txt 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong 2>>$O/err.txt | sed "s/^/$l /" >> $O/ab.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python bench.py --full > $O/full.json 2>>$O/err.txt; python -c "
import json; j=json.loads(open('$O/full.json').read().strip().splitlines()[-1]); print('$l --full', round(j['value'],1), {k:round(v['ms_per_substep']*1000) for k,v in j['roofline']['kernels'].items()})" >> $O/ab.txt
done; done
cat $O/ab.txt
