#!/usr/bin/env python3
"""Timing check on adversarial inputs (GPU box): periodic texts, long runs and tiny alphabets at the sizes where the code
paths switch (deferred address bits below 2 MiB, the device-wide suffix sort above 512 KiB).  Checks the GPU round trip
(the oracle needs many minutes on these) and prints the time of every one-shot Compress."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compress_amd  # noqa: E402
from compress_amd import synth  # noqa: E402

rng = np.random.default_rng(3)
blk = bytes(rng.integers(0, 256, 1000, dtype=np.uint8))
cases = {
    "abc x 2 MiB": (b"abc" * 700000)[: (1 << 21) - 2],
    "zeros 2 MiB": bytes((1 << 21) - 2),
    "period 1000 x 2 MiB": (blk * 2100)[: (1 << 21) - 2],
    "two symbols 1 MiB": bytes(rng.integers(0, 2, 1 << 20, dtype=np.uint8)),
    "ab x 400 KiB": b"ab" * 200000,
    "0xFF x 300 KiB": b"\xff" * 300000,
    "period 1000 x 4 MiB": (blk * 4200)[: 1 << 22],
}
for d_name, d in (("no dict", b""), ("dict", synth.dictionary())):
    c = compress_amd.Compressor(d)
    for name, x in cases.items():
        t0 = time.time()
        out = c.Compress(x)
        t1 = time.time()
        back = compress_amd.Decompress(out, d)
        ok = back == x
        print(f"{d_name:8s} {name:22s} n={len(x):8d} -> {len(out):8d} B  compress {t1 - t0:7.3f} s  roundtrip {'ok' if ok else 'DIFFERS'}",
              flush=True)
