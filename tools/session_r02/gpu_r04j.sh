cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for lib in orig X_NO_STORES X_NO_U X_NO_M X_USMALL X_UQUARTER X_UROW0 orig; do
if [ $lib = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$lib.so ftk_amd/libftkx.so; fi
echo "=== $lib"
python3 tools/ab_mask.py c4 3 -- "V=6" "V=5" | tail -2
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
