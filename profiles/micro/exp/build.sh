#!/bin/bash
# builds libhns_exp<mask>.so next to this script: the product sources with -DHNS_EXP=<mask> (see hns_pressure.hip: pair_load)
cd "$(dirname "$0")/../../../hnanosolver_amd/csrc"
# an argument of the form name:-Dflag[,-Dflag...] builds libhns_<name>.so with those flags instead
for m in "$@"; do
  flags="-DHNS_EXP=$m"; name="exp$m"
  case "$m" in *:*) name="${m%%:*}"; flags="$(echo "${m#*:}" | tr ',' ' ')";; esac
  out=../../profiles/micro/exp/libhns_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include $flags -x hip \
     hns_topology.cpp hns_nanovdb.cpp hns_leafio.cpp hns_gridbuild.hip hns_advect.hip hns_pressure.hip hns_pointwise.hip hns_api.hip hns_dist.hip -shared -pthread -ldl -o $out &
done
wait
