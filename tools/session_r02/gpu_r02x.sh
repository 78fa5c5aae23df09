( time python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6
for c in c4 c3 c2 c5; do bash tools/collect_profiles.sh $c r02 2>&1 | tail -1 | cut -c1-200; done
