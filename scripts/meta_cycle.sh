#!/bin/bash
# on the GPU box: (1) the checked variant on small texts, (2) index tests + bench with the shipped library, (3) phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-cyc}
cp compress_amd/libmi355_lzss.so /tmp/lib_main.so
cp compress_amd/variants/lib_metachk.so compress_amd/libmi355_lzss.so
timeout -k 5 120 python scripts/meta_check.py > $out/${tag}_check.log 2>&1 || { echo "check run failed"; tail -20 $out/${tag}_check.log; exit 1; }
if grep -q "META CHECK\|DIFF" $out/${tag}_check.log; then echo "check found problems"; grep "META CHECK\|DIFF\|first diffs" $out/${tag}_check.log | head -30; exit 1; fi
tail -1 $out/${tag}_check.log
cp /tmp/lib_main.so compress_amd/libmi355_lzss.so
bash scripts/gpu_quick.sh $tag $2 || exit 1
bash scripts/meta_prof.sh metaprof 8
