#!/bin/bash
for c in c2 c3 c5 c4; do bash tools/collect_profiles.sh $c r03 2>&1 | tail -1 | cut -c1-200; done
