cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ah; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for c in 128 plume; do HNS_LIBRARY=$GRAFT_REPO_ROOT/profiles/micro/exp/libhns_restrace.so timeout 300 python3 profiles/micro/res_trace.py $c 2>&1 | grep -v amdgpu.ids >> $O/trace.txt; done
cat $O/trace.txt
