cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05t; mkdir -p $O; rm -rf $O/*
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_operators_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 bench.py --cook > $O/cook256.json 2> $O/cook256.err; python3 -c "
import json; d=json.load(open('$O/cook256.json')); [print(k, {a: round(b,2) for a,b in v.items()}) for k,v in d.items() if isinstance(v,dict)]"
timeout 600 python3 bench.py --cook --config 128 > $O/cook128.json 2> $O/cook128.err; python3 -c "
import json; d=json.load(open('$O/cook128.json')); [print(k, {a: round(b,2) for a,b in v.items()}) for k,v in d.items() if isinstance(v,dict)]"
