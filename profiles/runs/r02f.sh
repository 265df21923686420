mkdir -p gpurun_out/r02f
python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -3
python profiles/micro/dist_profile.py single 2>&1 | tail -1
for w in 0 10 20 40 80; do for k in 4 2; do python profiles/micro/dist_profile.py rank $k $w 2>&1 | tail -1; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02f/prof_rank -- python3 $GRAFT_REPO_ROOT/profiles/micro/dist_profile.py rank 4 0 > $GRAFT_REPO_ROOT/gpurun_out/r02f/prof_rank.log 2>&1
