cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for pad in 0 12000; do
HNS_SB_PAD=$pad timeout 300 python3 profiles/micro/sb_ab.py 256 plume1024 512 2>&1 | grep -v amdgpu.ids | sed "s/^/pad$pad /" >> $O/ab.txt
done
done
cat $O/ab.txt
