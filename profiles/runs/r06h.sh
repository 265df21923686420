# round 6: a LARGE rank (one 256^3 slab of the weak-scaling bench: 32,768 owned leaves, two faces of 1,024 boundary leaves) on the exchanged path after the pipelining of the split loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O; rm -rf $O/*
for k in 0 2; do for tr in "" "--rccl"; do
timeout 600 python3 - $k $tr <<'PY' 2>&1 | grep "loopback_substep\|plain_grid" | cut -c1-600 >> $O/t.txt
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
print("k", sys.argv[1], end=" ")
sys.argv = ["dist_overhead.py", "256", "8", sys.argv[1], "--rank=4", "--lone-only", "--three"] + sys.argv[2:]
runpy.run_path("profiles/micro/dist_overhead.py", run_name="__main__")
PY
done; done
timeout 600 python3 profiles/micro/dist_overhead.py 256 8 2 --rank=4 --lone-only --three 2>&1 | grep "loopback_substep" | sed "s/^/chained (k 2, dist_mirror default) /" | cut -c1-600 >> $O/t.txt
cat $O/t.txt
