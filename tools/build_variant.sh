#!/bin/bash
# a tuning build of libhh_world.so beside the product one: bash tools/build_variant.sh NAME "-DHHX_ABL_NO_BARRIER ..."  -> hhmarl_2d_amd/lib/abl_NAME.so
# (git-ignored; load it with HH_WORLD_LIB=hhmarl_2d_amd/lib/abl_NAME.so; ablation switches give WRONG RESULTS on purpose: timing only)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-licm -fPIC -shared -Iinclude -Ihhmarl_2d_amd/csrc \
  $2 hhmarl_2d_amd/csrc/hh_world.hip -o hhmarl_2d_amd/lib/abl_$1.so
echo built abl_$1.so
