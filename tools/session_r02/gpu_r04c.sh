cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp ftk_amd/libftkx.so /tmp/libftkx_orig.so
for lib in orig NO_STORES NO_U NO_M; do
if [ $lib = orig ]; then cp /tmp/libftkx_orig.so ftk_amd/libftkx.so; else cp tools/probe/variants/libftkx_$lib.so ftk_amd/libftkx.so; fi
echo "=== $lib"
python3 tools/ab_mask.py c4 3 -- "V=6 TILE=3 PD=3 SWIZZLE=40" "V=6 TILE=3 PD=3" "V=5"
done
cp /tmp/libftkx_orig.so ftk_amd/libftkx.so
