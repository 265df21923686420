cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 120 profiles/micro/gather_width/gather_width > $O/gather_width.txt 2>&1; tail -7 $O/gather_width.txt
for cfg in 256 128; do
timeout 600 python3 profiles/micro/advect_lib_ab.py $PWD/hnanosolver_amd/lib/libhns.so $PWD/profiles/micro/exp/libhns_zdw.so $cfg 2>&1 | sed 's/\[{.*}, \({[^}]*}\), \({[^}]*}\)\]/\1 \2/' | tee -a $O/advect_zdw.txt
done
for l in $PWD/hnanosolver_amd/lib/libhns.so $PWD/profiles/micro/exp/libhns_zdw.so; do
HNS_LIBRARY=$l timeout 300 python3 bench.py --full --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l'.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_substep'],3) for k,v in d['roofline']['kernels'].items()})"
done
