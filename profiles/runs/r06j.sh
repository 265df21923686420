# round 6: kernel + copy trace of a LARGE rank (one 256^3 slab of 8, 32,768 owned leaves) on the exchanged path, default sweeps_per_exchange
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O; rm -rf $O/*
cat > /tmp/big.py <<'PY'
import os, runpy, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
sys.argv = ["dist_overhead.py", "256", "8", sys.argv[1], "--rank=4", "--lone-only", "--no-plain"]
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", "micro", "dist_overhead.py"), run_name="__main__")
PY
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_k0 -- python3 /tmp/big.py 0 > $O/trace_k0.log 2>&1
for f in $(find $O -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do (head -1 $f; tail -3000 $f) > $f.tail; rm $f; done
find $O -name "*agent_info*" -delete
tail -2 $O/trace_k0.log
