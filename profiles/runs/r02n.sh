python profiles/micro/sor_one.py 128 > /dev/null 2>&1
echo BASE; bash profiles/micro/pmc_sor.sh base 512
echo EXP16; bash profiles/micro/pmc_sor.sh e16 512 $PWD/profiles/micro/exp/libhns_exp16.so
echo EXP12; bash profiles/micro/pmc_sor.sh e12 512 $PWD/profiles/micro/exp/libhns_exp12.so
echo EXP31; bash profiles/micro/pmc_sor.sh e31 512 $PWD/profiles/micro/exp/libhns_exp31.so
