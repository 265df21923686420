cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05az; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
for i in 1 2; do for pk in 1 0; do
timeout 600 python3 - 4 $pk <<'PY' 2>&1 | grep -v amdgpu.ids | grep "loopback_substep" | cut -c1-200 >> $O/t.txt
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_mirror", "0")
H.set_option("dist_pack", sys.argv[2])
print("dist_pack", sys.argv[2], end=" ")
sys.argv = ["dist_overhead.py", "plume1024", "8", "2", "--partition", "--rank=" + sys.argv[1], "--lone-only"]
runpy.run_path("profiles/micro/dist_overhead.py", run_name="__main__")
PY
done; done
cat $O/t.txt
