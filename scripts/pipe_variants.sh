#!/bin/bash
# on the GPU box: transfer-inclusive time of every library variant at several chunk sizes, all in one run on one host
cd "$(dirname "$0")/.."
cp compress_amd/libmi355_lzss.so /tmp/lib_orig.so
for rep in 1 2; do
for v in compress_amd/variants/lib_p*.so; do
  cp "$v" compress_amd/libmi355_lzss.so
  for c in ${CHUNKS:-1536 2560}; do
    timeout -k 10 200 python bench.py --chunk-blobs $c --steps 2 --pipe-steps 4 --no-cpu --no-stream --no-latency --stage-steps 1 2>/tmp/ck.err > /tmp/ck.json || { echo "run failed"; tail -5 /tmp/ck.err; cp /tmp/lib_orig.so compress_amd/libmi355_lzss.so; exit 1; }
    python -c "import json; d=json.loads(open('/tmp/ck.json').readlines()[-1]); t=d['transfer_inclusive']; print('$(basename $v) chunk=$c', t['ms_per_call'], d['transfer_inclusive_frac_of_resident'])"
  done
done
done
cp /tmp/lib_orig.so compress_amd/libmi355_lzss.so
