#!/bin/bash
# Regenerates patches/ftk-xl-hip.patch from an edited copy of the reference tree.
#   tools/make_patch.sh <edited-tree>     (<edited-tree>/include/ftk/..., <edited-tree>/src/filters/critical_point_tracer_regular_hip.cpp)
# The text in front of the first `diff` line of the current patch (its description) is kept.  Typical round trip:
#   W=$(mktemp -d); cp -r /root/reference/include $W/include; (cd $W && git apply -p1 /root/repo/patches/ftk-xl-hip.patch)
#   ... edit $W ...; tools/make_patch.sh $W
set -e
REF=${REF:-/root/reference}
B=$(cd "$1" && pwd)
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$HERE/patches/ftk-xl-hip.patch
TMP=$(mktemp)
awk '/^diff -U2/ {exit} {print}' "$OUT" > "$TMP"
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
ln -s "$REF" "$W/a"; ln -s "$B" "$W/b"
FILES="include/ftk/config.hh.in include/ftk/filters/critical_point_tracker_2d_regular.hh include/ftk/filters/critical_point_tracker_3d_regular.hh include/ftk/filters/critical_point_tracker_regular.hh include/ftk/filters/filter.hh include/ftk/object.hh src/filters/critical_point_tracer_regular_hip.cpp"
for f in $FILES; do
  ( cd "$W" && { diff -U2 -N "a/$f" "b/$f" || true; } ) | sed -e "1s|^--- a/$f.*|--- a/$f|" -e "2s|^+++ b/$f.*|+++ b/$f|" > "$W/one.diff"
  if [ -s "$W/one.diff" ]; then
    echo "diff -U2 -r -N a/$f b/$f" >> "$TMP"
    if [ ! -e "$REF/$f" ]; then sed -i -e '1s|^--- .*|--- /dev/null|' "$W/one.diff"; fi
    cat "$W/one.diff" >> "$TMP"
  fi
done
mv "$TMP" "$OUT"
echo "wrote $OUT"
