for seg in 0 128 512; do for f in pair tile; do python profiles/micro/sor_one.py 512 plume1024 schedule_segment=$seg rbgs=$f 2>&1 | grep sweep; done; done
