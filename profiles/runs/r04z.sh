timeout 300 python3 profiles/micro/sor_one.py plume1024 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/sor_one.py plume1024 rbgs=wave 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/sor_one.py plume1024 rbgs=tile 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/sor_one.py plume1024 rbgs=wave schedule_segment=64 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/sor_one.py plume1024 schedule_segment=64 2>/dev/null | tail -1
timeout 300 python3 profiles/micro/sor_one.py plume1024 schedule_segment=256 2>/dev/null | tail -1
