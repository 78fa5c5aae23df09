cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for v in orig NO_STORES U_AUX_1 U_AUX_2 U_AUX_3 U_AUX_17 U_AUX_18; do
  if [ $v = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$v.so ftk_amd/libftkx.so; fi
  echo "== $v"
  python tools/ab_mask.py c4 3 -- "V=6 TILE=3 PD=3" "V=5" 2>&1 | tail -2
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
