#!/bin/bash
# on the GPU box: GPU tests, then the default bench line; stops at the first failing step
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-check}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/${tag}_pytest.log 2>&1 || { echo "pytest failed"; tail -30 $out/${tag}_pytest.log; exit 1; }
tail -3 $out/${tag}_pytest.log
timeout -k 10 400 python bench.py > $out/${tag}_bench.log 2> $out/${tag}_bench.err || { echo "bench failed"; tail -20 $out/${tag}_bench.err; exit 1; }
tail -1 $out/${tag}_bench.log > $out/${tag}_bench.json
python -c "
import json; d=json.load(open('$out/${tag}_bench.json'))
print(d['value'], d['ms_per_step'], d['stages_ms']); print(d['transfer_inclusive']); print(d['roofline']['frac'], d['roofline']['step_frac']); print(d.get('stream')); print(d['decompress']); print(d.get('cpu_baseline')); print(d.get('bit_exact_vs_oracle'))"
