# round 6: the EXCHANGED pressure loop (dist_mirror = 0: what RCCL ranks run) of rank 4 of 8 of BASELINE config 5, alone on the GPU, status quo:
# sweeps_per_exchange default (4) and 2, device-copy loopback and RCCL loopback, dist_pack 1 / 0; then a kernel + copy trace of the default
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
run() { # k pack extra...
k=$1; pk=$2; shift 2
timeout 600 python3 - $k $pk "$@" <<'PY' 2>&1 | grep -v amdgpu.ids | grep "loopback_substep\|plain_grid" | cut -c1-600 >> $O/t.txt
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
H.set_option("dist_pack", sys.argv[2])
print("k", sys.argv[1], "dist_pack", sys.argv[2], end=" ")
sys.argv = ["dist_overhead.py", "plume1024", "8", sys.argv[1], "--partition", "--rank=4", "--lone-only", "--three"] + sys.argv[3:]
runpy.run_path("profiles/micro/dist_overhead.py", run_name="__main__")
PY
}
run 0 1; run 2 1 --no-plain; run 0 0 --no-plain; run 2 0 --no-plain; run 0 1 --rccl --no-plain; run 2 1 --rccl --no-plain
cat $O/t.txt
cat > /tmp/tr.py <<'PY'
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
sys.argv = ["dist_overhead.py", "plume1024", "8", sys.argv[1], "--partition", "--rank=4", "--lone-only", "--no-plain"] + sys.argv[2:]
runpy.run_path(sys.argv[0] if False else "profiles/micro/dist_overhead.py", run_name="__main__")
PY
for k in 0 2; do
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k$k -- python3 /tmp/tr.py $k > $O/trace_k$k.log 2>&1
done
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k0_rccl -- python3 /tmp/tr.py 0 --rccl > $O/trace_k0_rccl.log 2>&1
ls -R $O | head -40
# keep the traces small enough to come back: the last 3000 rows of each
for f in $(find $O -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do (head -1 $f; tail -4000 $f) > $f.tail; rm $f; done
du -sh $O
