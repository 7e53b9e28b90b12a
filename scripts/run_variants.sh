#!/bin/bash
# on the GPU box: bench every library variant (stage times only); stops at the first failing run
cd "$(dirname "$0")/.."
cp compress_amd/libmi355_lzss.so /tmp/lib_orig.so
for v in compress_amd/variants/lib_*.so; do
  cp "$v" compress_amd/libmi355_lzss.so
  echo "== $v"
  timeout -k 10 120 python bench.py --steps 5 --warmup 1 --stage-steps 3 --no-cpu --no-stream --no-latency --pipe-steps 0 2>/dev/null > /tmp/var.json || { echo "run failed"; break; }
  python -c "import json; d=json.loads(open('/tmp/var.json').readlines()[-1]); print(d['ms_per_step'], d['stages_ms'])" || break
done
cp /tmp/lib_orig.so compress_amd/libmi355_lzss.so
