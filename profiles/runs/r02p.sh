for seg in 0 1 64 128 256 512 1024 2048 4096 8192; do python profiles/micro/sor_one.py 256 512 plume1024 schedule_segment=$seg schedule=auto 2>&1 | grep sweep; done
