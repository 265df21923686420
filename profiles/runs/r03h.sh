cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o rank -- python3 profiles/micro/dist_profile.py rank 4 > $O/log.txt 2>&1
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 profiles/micro/trace_timeline.py $f "" 2>/dev/null | head -0
n=$(wc -l < $f); echo rows $n
python3 profiles/micro/trace_timeline.py $f $((n/2)) 80 > $O/timeline.txt
head -c 600 $f > $O/header.txt
tail -2 $O/log.txt
find $O/trace -name '*.csv' | head
# keep outputs small
find $O/trace -name '*.csv' -size +20M -delete
