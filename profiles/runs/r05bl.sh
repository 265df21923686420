cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05bl; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -15 $O/pytest.txt
