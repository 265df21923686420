# round 6: would a LARGE rank (256^3 slab) do better in line too? dist_unsplit = always against the default (split + pipelined beyond 16,384 leaves); and 128^3-slab ranks (4,096 leaves)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O; rm -rf $O/*
for cfg in 256 128; do for us in 1 always 0; do for k in 0 2; do
timeout 600 python3 - $k $us $cfg <<'PY' 2>&1 | grep "loopback_substep" | cut -c1-330 >> $O/t.txt
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
H.set_option("dist_unsplit", sys.argv[2])
print("config", sys.argv[3], "k", sys.argv[1], "dist_unsplit", sys.argv[2], end=" ")
sys.argv = ["dist_overhead.py", sys.argv[3], "8", sys.argv[1], "--rank=4", "--lone-only", "--three", "--no-plain"]
runpy.run_path("profiles/micro/dist_overhead.py", run_name="__main__")
PY
done; done; done
cat $O/t.txt
