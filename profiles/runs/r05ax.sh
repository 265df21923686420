cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05ax; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
for pk in 1 0; do
for r in 4 6; do timeout 600 python3 - $r $pk <<'PY' 2>&1 | grep -v amdgpu.ids | grep config | cut -c1-40,330-700
import sys, runpy
import hnanosolver_amd as H
H.set_option("dist_pack", sys.argv[2])
sys.argv = ["dist_overhead.py", "plume1024", "8", "2", "--partition", "--rank=" + sys.argv[1]]
runpy.run_path("profiles/micro/dist_overhead.py", run_name="__main__")
PY
done; done
