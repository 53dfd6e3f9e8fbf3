# tuning builds of hh_k_policy_w16 side by side on one box: VARIANTS="nobar noglds ..." (names of tools/build_variant.sh builds; "" = the product library)
for v in ${VARIANTS:-""}; do
  lib=""; [ "$v" != "product" ] && lib=hhmarl_2d_amd/lib/abl_$v.so
  for r in ${ROWS:-16384 32768}; do
    echo "== $v rows $r: $(HH_WORLD_LIB=$lib HH_POLICY_W=${W:-2} python tools/policy_bench.py $r 0 2>&1 | tail -1)"
    [ -n "$PROF" ] && W=${W:-2} HH_WORLD_LIB=$lib python tools/policy_w16_phase_profile.py $r
  done
done
