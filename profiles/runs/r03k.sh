for a in "4 0" "4 0 rccl" "2 0 rccl" "1 0 rccl"; do timeout 200 python3 profiles/micro/dist_profile.py rank $a 2>/dev/null | grep "ms per substep"; done
