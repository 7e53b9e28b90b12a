"""N > 1 path on CPU: two gloo ranks shard the blob ids disjointly, agree on the max time and the
whole-job byte count exactly as bench.py computes them (no data-path collective exists to test)."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch.distributed as dist
from compress_amd import shard, synth
dist.init_process_group(backend="gloo")
rank, world, _ = shard.env_rank_world()
ids = shard.shard_ids(rank, world, 5)
assert all(shard.owner(b, world) == rank for b in ids)
strong = shard.batch_shard_ids(rank, world, 7)      # one batch of 7 blobs over the ranks (bench.py's timed leg)
assert strong == [b for b in shard.shard_ids(rank, world, 7) if b < 7]
n_all = shard.sum_over_ranks(float(len(strong)))
assert n_all == 7.0, n_all
arr = synth.blobs_array(rank, world, 5, 4096)       # the generator bench.py uses: rank, rank + world, ...
blobs = [synth.blob(b, 4096) for b in ids]
assert [arr[k].tobytes() for k in range(5)] == blobs
elapsed = 1.0 + rank          # pretend rank 1 is slower
t = shard.max_over_ranks(elapsed)
total = shard.sum_over_ranks(float(sum(map(len, blobs))))
gathered = [None] * world
dist.all_gather_object(gathered, ids)
dist.barrier()
if rank == 0:
    print(json.dumps({"t": t, "total": total, "ids": gathered}))
dist.destroy_process_group()
""" % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["t"] == 2.0
    assert res["total"] == 2 * 5 * 4096
    assert sorted(res["ids"][0] + res["ids"][1]) == list(range(10))
    assert set(res["ids"][0]).isdisjoint(res["ids"][1])
