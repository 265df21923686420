# round 6, last collection: GPU suite, smoke, then profiles/collect_r06.sh (PMC passes, bench lines, kernel statistics) on the final kernel sources
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
bash profiles/collect_r06.sh r06 > gpurun_out/r06_collect.log 2>&1; echo "collect rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest.log; tail -5 gpurun_out/r06_collect.log
