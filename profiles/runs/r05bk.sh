cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bk; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/profiles/micro/exp
for c in 256 512; do
  echo "base $c $(timeout 300 python3 profiles/micro/div_ab.py rev 1 1 $c 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)" >> $O/ab.txt
  for l in gnoz gnoxy gnoall; do
    echo "$l $c $(HNS_LIBRARY=$X/libhns_$l.so timeout 300 python3 profiles/micro/div_ab.py rev 1 1 $c 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)" >> $O/ab.txt
  done
done
cat $O/ab.txt
