#!/bin/bash
# on the GPU box: phase clocks of the run-compressed suffix sort (variant built with -DSA_META_PROFILE) on the bench batch
cd "$GRAFT_REPO_ROOT" || exit 1
cp compress_amd/variants/lib_${1:-metaprof}.so compress_amd/libmi355_lzss.so
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --stage-steps 1 --overlap 1 --no-stream --no-latency --pipe-steps 0 --no-cpu 2>&1 | grep -E "^blk|value" | sort | uniq -c | sort -rn | head -${2:-30}
