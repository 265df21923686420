cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05s; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for od in 0 3 4; do
HNS_SB_ORDER=$od timeout 300 python3 profiles/micro/sb_ab.py 128 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/order$od /" >> $O/ab.txt
done
done
cat $O/ab.txt
