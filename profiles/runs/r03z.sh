for i in 1 2; do
timeout 200 python3 profiles/micro/dist_profile.py rank 1 0 2>/dev/null | grep "ms per"
HNS_ALT=2 timeout 200 python3 -c "
import sys; sys.argv=['x','rank','1','0']
import hnanosolver_amd as H; H.set_option('alternate','2')
exec(open('profiles/micro/dist_profile.py').read())" 2>/dev/null | grep "ms per"
done
