cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for lib in orig AUX1 AUX2 AUX3 AUX16 AUX17 AUX18 AUX19 orig; do
if [ $lib = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$lib.so ftk_amd/libftkx.so; fi
echo "=== $lib"
python3 tools/ab_mask.py c4 4 -- "V=6" "V=5" | tail -2
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
