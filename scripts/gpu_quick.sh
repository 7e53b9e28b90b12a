#!/bin/bash
# on the GPU box: the index / compress parity tests first, then all GPU tests, then the stage times of the bench batch
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-quick}
timeout -k 10 600 python -u -m pytest tests/test_gpu_compress.py -m gpu -x -q --timeout 200 --timeout-method thread > $out/${tag}_pytest1.log 2>&1 || { echo "index tests failed"; tail -40 $out/${tag}_pytest1.log; exit 1; }
tail -2 $out/${tag}_pytest1.log
if [ "$2" != "notests" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/${tag}_pytest.log 2>&1 || { echo "pytest failed"; tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --stage-steps 2 --no-stream --no-latency --pipe-steps 0 --cpu-seconds 4 > $out/${tag}_bench.log 2> $out/${tag}_bench.err || { echo "bench failed"; tail -20 $out/${tag}_bench.err; exit 1; }
python -c "
import json; d=json.loads(open('$out/${tag}_bench.log').readlines()[-1])
print(d['value'], d['ms_per_step'], d['stages_ms']); print(d.get('bit_exact_vs_oracle'), d['decompress']['roundtrip_equal'])"
