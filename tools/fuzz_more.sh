#!/bin/bash
# An extended fuzz run: the seeded parity tests of tests/test_gpu_fuzz*.py on OTHER seeds (FTKX_FUZZ_OFFSET), one pytest process per offset.
#   tools/fuzz_more.sh FIRST COUNT [OUT]     e.g. tools/fuzz_more.sh 1000000 20 gpurun_out/fuzz_more.txt
first=${1:-1000000}; count=${2:-10}; out=${3:-gpurun_out/fuzz_more.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
for ((i = 0; i < count; i++)); do
  off=$((first + 1000 * i))
  if FTKX_FUZZ_OFFSET=$off timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fuzz_reference.py -q -m gpu -p no:cacheprovider > /tmp/fuzz_$off.log 2>&1; then
    echo "offset $off: $(tail -1 /tmp/fuzz_$off.log)" | tee -a "$out"
  else
    echo "offset $off: FAILED" | tee -a "$out"; grep -E "^(FAILED|E  )" /tmp/fuzz_$off.log | head -20 >> "$out"; cp /tmp/fuzz_$off.log "$(dirname "$out")/fuzz_failed_$off.log"; exit 1
  fi
done
