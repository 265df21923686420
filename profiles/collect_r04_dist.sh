#!/bin/bash
# profiles/collect_r04_dist.sh <tag> -- round 4, what one rank of a multi-GPU run costs on ONE GPU before any wire time (local / loopback transports):
# the exchanged pressure loop with and without the blocked range sweeps (option dist_block), the chained substep with one-iteration (k = 1) and
# two-iteration blocked (k = 2) launches; gpurun_out/<tag>/dist_overhead.jsonl, dist_ab.txt
tag=${1:-r04_dist}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd $root
for k in 1 2 4; do python3 profiles/micro/dist_overhead.py plume1024 8 $k --partition 2>/dev/null | tail -1 >> $out/dist_overhead.jsonl; done
for k in 1 2 4; do python3 profiles/micro/dist_overhead.py 256 2 $k 2>/dev/null | tail -1 >> $out/dist_overhead.jsonl; done
python3 profiles/micro/dist_overhead.py 128 2 2 2>/dev/null | tail -1 >> $out/dist_overhead.jsonl
for args in "2 256" "4 256" "2 128" "4 128"; do echo "k config = $args:" $(python3 profiles/micro/dist_ab.py dist_block 0 1 $args 2>/dev/null | tail -1) >> $out/dist_ab.txt; done
cat $out/dist_ab.txt
python3 - "$out/dist_overhead.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    j = json.loads(line)
    o = j["one_rank_loopback"]
    print(j["config"], "world", j["world"], "k", j["sweeps_per_exchange"], "| single-GPU substep", j["single_gpu_substep_ms"], "ms | one rank alone:", o["substep_ms"], "ms,", o.get("pressure_us_per_iteration"), "us per iteration, host enqueue", o["host_enqueue_ms"], "ms | all ranks in lockstep on the one device", j["all_ranks_lockstep_ms"], "ms")
PY
