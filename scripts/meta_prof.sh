#!/bin/bash
# on the GPU box: phase clocks of the run-compressed suffix sort (variant built with -DSA_META_PROFILE) on the bench batch:
# mean cycles per phase over the sampled workgroups of the timed launches
cd "$GRAFT_REPO_ROOT" || exit 1
cp compress_amd/variants/lib_${1:-metaprof}.so compress_amd/libmi355_lzss.so
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --stage-steps 1 --overlap 1 --no-stream --no-latency --pipe-steps 0 --no-cpu > /tmp/prof.log 2>&1
python - <<'PY'
import re, collections
acc = collections.OrderedDict(); n = 0
for line in open('/tmp/prof.log'):
    if not line.startswith('blk') or 'N 131073' not in line: continue
    n += 1
    for k, v in re.findall(r'([A-Za-z0-9]+) (\d+)', line.split('|', 1)[1]):
        acc[k] = acc.get(k, 0) + int(v)
tot = sum(acc.values())
print('workgroups sampled', n, 'mean cycles per blob', tot // max(n, 1))
print(' '.join('%s %dK (%.0f%%)' % (k, v // max(n, 1) // 1000, 100.0 * v / max(tot, 1)) for k, v in acc.items()))
import json
for line in open('/tmp/prof.log'):
    if line.startswith('{'):
        d = json.loads(line); print('sa_build ms', d['stages_ms']['sa_build'])
PY
