cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 200 python3 profiles/micro/dist_profile.py rank 4 0 2>/dev/null | tail -1
timeout 200 python3 profiles/micro/dist_profile.py rank 4 0 rccl 2>/dev/null | tail -1
timeout 200 python3 profiles/micro/dist_profile.py rank 2 0 rccl 2>/dev/null | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o rank -- python3 profiles/micro/dist_profile.py rank 4 0 rccl > $O/log.txt 2>&1
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
n=$(wc -l < $f); echo rows $n
python3 profiles/micro/trace_timeline.py $f $((n/2)) 60 > $O/timeline.txt
find $O/trace -name '*.csv' -size +20M -delete
