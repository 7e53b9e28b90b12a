#!/bin/bash
# on the GPU box: transfer-inclusive rate of the host-buffer pipeline against the chunk size
cd "$(dirname "$0")/.."
for c in ${CHUNKS:-1024 1536 2560 3072 4096}; do
  timeout -k 10 200 python bench.py --chunk-blobs $c --steps 4 --pipe-steps 4 --no-cpu --no-stream --no-latency --stage-steps 1 2>/tmp/ck.err > /tmp/ck.json || { echo "run failed"; tail -5 /tmp/ck.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/ck.json').readlines()[-1]); t=d['transfer_inclusive']; print('chunk=$c', d['ms_per_step'], t['ms_per_call'], t['value'], d['transfer_inclusive_frac_of_resident'])" || exit 1
done
