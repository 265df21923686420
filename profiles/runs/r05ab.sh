cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ab; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for l in hnanosolver_amd/lib/libhns.so profiles/micro/exp/libhns_coal1.so profiles/micro/exp/libhns_coal0.so; do
HNS_LIBRARY=$PWD/$l timeout 300 python3 profiles/micro/advect_ab.py rev 1 1 256 2>&1 | grep -v amdgpu | sed "s|^|$(basename $l) |" | cut -c1-260 | tee -a $O/coal.txt
done
