# time of sa_build with the doubling rounds cut short (stops at the first failing run); needs a library built with
# -DLZSS_DEBUG_KNOBS (scripts/build_variants.sh "dbg:-DLZSS_DEBUG_KNOBS" for sa_build.hip and match.hip): the shipped
# library ignores these environment variables
for a in ${ROUNDS:-0 1 2 3 4 5 6 8 20}; do
  echo "rounds=$a"
  MI355_LZSS_SA_ROUNDS=$a MI355_LZSS_ABLATE=7 timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu --no-stream --no-latency --pipe-steps 0 2>/dev/null > /tmp/abl_sa.json || { echo "run failed"; exit 1; }
  python -c "import sys,json; d=json.loads(open('/tmp/abl_sa.json').readlines()[-1]); print(d['stages_ms']['sa_build'])" || exit 1
done
