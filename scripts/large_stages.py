#!/usr/bin/env python3
"""Stage times of one-shot compressions of single large / adversarial texts (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compress_amd import synth  # noqa: E402
from compress_amd.batch import Batch, new_ctx  # noqa: E402

rng = np.random.default_rng(3)
blk = bytes(rng.integers(0, 256, 1000, dtype=np.uint8))
cases = {
    "abc x 2 MiB": (b"abc" * 700000)[: (1 << 21) - 2],
    "ab x 400 KiB": b"ab" * 200000,
    "period 1000 x 4 MiB": (blk * 4200)[: 1 << 22],
    "synthetic blobs 4 MiB": b"".join(synth.blob(b) for b in range(32)),
    "random 4 MiB": bytes(rng.integers(0, 256, 1 << 22, dtype=np.uint8)),
}
ctx = new_ctx(synth.dictionary())
ctx.set_overlap(1)
for name, x in cases.items():
    bt = Batch(ctx, 1, len(x))
    for rep in range(2):
        bt.upload([x])
        bt.compress()
        bt.sizes()
    print(f"{name:24s} n={len(x):8d}", {k: round(v, 2) for k, v in bt.stage_times().items()}, flush=True)
    bt.close()
