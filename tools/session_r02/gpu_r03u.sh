cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for v in orig NO_STORES NO_COMPUTE STREAM_ONLY; do
  if [ $v = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$v.so ftk_amd/libftkx.so; fi
  echo "== $v"
  python tools/ab_mask.py c4 3 -- "V=6 TILE=3 PD=3" "V=6 TILE=3 PD=2" "V=5"
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
