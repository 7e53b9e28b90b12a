# step time with the index / match stages run as 1..4 concurrent sub-batches (stops at the first failure)
for s in ${SPLITS:-1 2 3 4}; do
  echo "split=$s"
  MI355_LZSS_SPLIT=$s timeout -k 10 300 python bench.py --steps 4 --warmup 1 2>/dev/null > /tmp/split.json || { echo failed; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/split.json').readlines()[-1]); print(d['ms_per_step'], d['value'], d['bit_exact_vs_oracle'])" || exit 1
done
