#!/bin/bash
# on the GPU box: ms per step against the number of blobs in the batch (what one rank of an N-GPU strong-scaling run sees)
cd "$(dirname "$0")/.."
for nb in ${BLOBS:-1250 2500 5000 10000}; do
  timeout -k 10 120 python bench.py --blobs $nb --steps 5 --warmup 1 --stage-steps 1 --no-cpu --no-stream --no-latency --pipe-steps 0 --no-weak 2>/tmp/sz.err > /tmp/sz.json || { echo "run failed"; tail -5 /tmp/sz.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/sz.json').readlines()[-1]); print('blobs=$nb', d['ms_per_step'], d['value'], d['stages_ms'])" || exit 1
done
