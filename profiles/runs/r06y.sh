# round 6: k_advect_vector_n with its box rows padded from 10 to 24 floats (conflict-free taps: pad) against the committed kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O; rm -f $O/*
for rep in 1 2; do for l in prev pad; do
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/advect_two_libs.py 256 128 plume1024 --amp=96,400 >> $O/ab.txt 2>>$O/err.txt
	HNS_LIBRARY=$PWD/profiles/micro/exp/libhns_$l.so timeout 300 python profiles/micro/bench_with_options.py - --no-cpu-baseline --no-strong 2>>$O/err.txt | sed "s/^/$l /" >> $O/ab.txt
done; done
cat $O/ab.txt
