#!/bin/bash
# on the GPU box: ms per step of the bench batch against the number of overlapped sub-batches
cd "$(dirname "$0")/.."
for p in ${PARTS:-1 2 3 4 6}; do
  timeout -k 10 120 python bench.py --steps 4 --warmup 1 --stage-steps 1 --overlap $p --no-cpu --no-stream --no-latency --pipe-steps 0 --no-weak 2>/tmp/ov.err > /tmp/ov.json || { echo "run failed"; tail -5 /tmp/ov.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/ov.json').readlines()[-1]); print('parts=$p', d['ms_per_step'], d['value'])" || exit 1
done
