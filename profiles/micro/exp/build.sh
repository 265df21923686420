#!/bin/bash
# Experiment builds of libhns: the product sources plus ONE of the patches in this directory (timing switches and instrumentation live
# here, not in the product kernels), compiled with extra -D flags into profiles/micro/exp/libhns_<name>.so (load with HNS_LIBRARY=...).
# A patch is a record of an experiment: it applies to the product sources of the commit that added it (git log -- <patch>), not
# necessarily to today's.
#   build.sh <name> <patch file or -> [-Dflag ...]
#     build.sh halo4   sor_halo_exp.patch   -DHNS_EXP=4           # pair SOR kernel without its y-face halo (results wrong, timing only)
#     build.sh trace   sorblock_trace.patch -DHNS_SB_TRACE=2048   # s_memtime stamps of workgroups 2048..2111 (profiles/micro/sb_trace.py)
#     build.sh p4      advect_p4.patch      -DHNS_EXP_P4          # velocity gathered out of a float4-padded copy
set -e
here="$(cd "$(dirname "$0")" && pwd)"; src="$here/../../../hnanosolver_amd/csrc"
name=$1; patch=$2; shift 2
tmp=$(mktemp -d); cp "$src"/*.hip "$src"/*.hpp "$src"/*.cpp "$tmp"/
[ "$patch" != "-" ] && (cd "$tmp" && patch -s -p0 < "$here/$patch")
cd "$tmp"
objs=""
for f in hns_topology.cpp hns_nanovdb.cpp hns_leafio.cpp hns_gridbuild.hip hns_advect.hip hns_pressure.hip hns_sorblock.hip hns_pointwise.hip hns_api.hip hns_dist_plan.hip hns_dist_transport.hip hns_dist_substep.hip; do
  extra=""; [ "$f" = hns_sorblock.hip ] && extra="${HNS_SORBLOCK_FLAGS--fno-slp-vectorize}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I"$src/../../include" $extra "$@" -x hip -c $f -o $f.o &
  objs="$objs $f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$here/libhns_$name.so" $objs -ldl
rm -rf "$tmp"; echo "$here/libhns_$name.so"
#     r06_removed_sor_forms_and_tables.patch: NOT an experiment but the record of what round 6 took out of the product (VERDICT r5 task 4): the one-iteration SOR kernels of rounds
#       1-2 (one wave per leaf, per z-adjacent leaf pair, the 2 x 2 tile form, the mirroring pair form), the 16^3 rows-in-registers / parity-sorted "lean" / LDS-DMA forms of the blocked
#       kernel, their launch-start stagger, and the wave-record / tile-group builders of hns_gridbuild.hip. A reverse diff against commit 2e52202: `patch -p1` over that commit's
#       csrc/ restores them (hns_dist.hip, hns_internal.hpp, hns_topology.cpp of that commit are what call them).
