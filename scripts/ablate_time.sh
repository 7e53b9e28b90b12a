#!/bin/bash
# on the GPU box: match stage time of the bench batch with parts of match_kernel disabled (lib_dbg.so, -DLZSS_DEBUG_KNOBS)
cd "$(dirname "$0")/.."
for a in ${ABL:-0 1 4 16 20 21}; do
  MI355_LZSS_ABLATE=$a timeout -k 10 120 python bench.py --steps 2 --warmup 1 --stage-steps 2 --no-cpu --no-stream --no-latency --pipe-steps 0 --no-weak 2>/dev/null > /tmp/abl.json || { echo "run failed"; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/abl.json').readlines()[-1]); print('ablate=$a', d['stages_ms']['match'], d['stages_ms']['emit'])" || exit 1
done
