#!/bin/bash
# on the GPU box: decompressor tests, then the bench's decompression leg
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_decompress.py tests/test_decode_errors.py tests/test_gpu_stream.py tests/test_info_cli.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --stage-steps 1 --no-stream --no-latency --pipe-steps 0 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['decompress']); print(d['ms_per_step'])"
