for i in 1 2; do timeout 900 python -m pytest tests/test_dist_gpu.py -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2 | tr '\n' ' '; echo; done
timeout 300 python3 profiles/micro/dist_lone.py plume1024 8 4 1 --partition 2>/dev/null | cut -c1-60
timeout 300 python3 profiles/micro/dist_lone.py plume1024 8 0 1 --partition 2>/dev/null | cut -c1-60
timeout 300 python3 profiles/micro/dist_lone.py plume 4 1 1 --partition 2>/dev/null | cut -c1-60
timeout 200 python3 profiles/micro/dist_ab.py rbgs auto pair 1 128 2>/dev/null | tail -1
