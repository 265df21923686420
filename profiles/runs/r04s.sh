timeout 200 python3 profiles/micro/advect_ab.py rev 1 1 2>/dev/null | tail -1 | cut -c1-260
HNS_CHAIN_PROBE=1 timeout 200 python3 profiles/micro/advect_ab.py rev 1 1 2>/dev/null | tail -1 | cut -c1-260
