cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05v; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
for od in 0 3; do
HNS_SB_ORDER=$od timeout 300 python3 profiles/micro/sb_ab.py 64 plume 128 256 2>&1 | grep -v amdgpu.ids | sed "s/^/order$od /" >> $O/ab.txt
done
done
for od in 0 3; do for rep in 1 2 3; do
HNS_SB_ORDER=$od timeout 300 python3 bench.py --no-cpu-baseline --no-strong 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('order$od bench256', round(d['value'],1), round(1e3*d['roofline']['ms_per_launch'],2))" >> $O/ab.txt
done; done
sort $O/ab.txt
