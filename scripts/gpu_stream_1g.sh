#!/bin/bash
# on the GPU box: config 4 at its stated size (1 GiB through the speculative packer, every blob decompressed again), then the
# 2-rank rehearsal of the multi-process bench path on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --stage-steps 1 --no-cpu --no-latency --pipe-steps 0 --stream-bytes 1073741824 > $out/r03_stream_1g.log 2> $out/r03_stream_1g.err || { echo "stream run failed"; tail -20 $out/r03_stream_1g.err; exit 1; }
tail -1 $out/r03_stream_1g.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); json.dump({'stream': d['stream'], 'from': 'bench.py --stream-bytes 1073741824', 'value_MBps_same_run': d['value']}, open('$out/r03_stream_1g.json','w'), indent=1); print(d['stream'])"
MI355_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --blobs 4000 --cpu-seconds 4 > $out/r03_rehearse2.log 2> $out/r03_rehearse2.err || { echo "rehearsal failed"; tail -20 $out/r03_rehearse2.err; exit 1; }
grep '^{' $out/r03_rehearse2.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['config'], d['weak_scaling'], d.get('cpu_baseline') is not None, d.get('bit_exact_vs_oracle'))"
