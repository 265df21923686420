cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05av; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for r in 4 6; do timeout 600 python3 profiles/micro/dist_overhead.py plume1024 8 2 --partition --rank=$r 2>&1 | grep -v amdgpu.ids | grep config >> $O/overhead.jsonl; done
timeout 600 python3 profiles/micro/dist_overhead.py 256 2 2 2>&1 | grep -v amdgpu.ids | grep config >> $O/overhead.jsonl
cut -c1-900 $O/overhead.jsonl
