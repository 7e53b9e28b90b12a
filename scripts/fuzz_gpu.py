#!/usr/bin/env python3
"""Extended randomized parity run (GPU box): batches of generated inputs and random Write / Revert sequences, the
HIP path against the CPU oracle byte for byte.  Not part of the pytest suite (minutes of oracle time); run after
kernel changes:

    python scripts/fuzz_gpu.py [--seed S] [--batches B] [--streams K]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gen(rng, synth):
    """one input: mixtures of runs, small alphabets, reserved symbols, copies of earlier pieces and generator blobs"""
    n = int(rng.choice([0, 1, 2, 3, 5, 17, 64, 255, 256, 257, 1000, 4096, 20000, 70000]))
    n = int(rng.integers(0, n + 1)) if rng.random() < 0.5 else n
    kind = rng.integers(0, 6)
    if kind == 0:
        x = rng.integers(0, int(rng.choice([1, 2, 3, 4, 16, 256])), n, dtype=np.uint8)
    elif kind == 1:  # runs of random length, reserved symbols among the run bytes
        out = []
        while sum(map(len, out)) < n:
            b = int(rng.choice([0, 0, 0, 1, 7, 0xFD, 0xFE, 0xFF, int(rng.integers(0, 256))]))
            out.append(np.full(int(rng.choice([1, 2, 3, 5, 100, 159, 160, 161, 255, 256, 257, 600])), b, dtype=np.uint8))
        x = np.concatenate(out)[:n] if out else np.zeros(0, np.uint8)
    elif kind == 2:  # periodic
        p = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
        x = np.tile(p, n // len(p) + 1)[:n]
    elif kind == 3:  # generator blob slice
        b = synth.blob(int(rng.integers(0, 500)))
        s = int(rng.integers(0, max(1, len(b) - n)))
        x = np.frombuffer(b[s:s + n], dtype=np.uint8)
    elif kind == 4:  # random bytes with copies of earlier stretches at all distances
        x = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 20))):
            if n < 10:
                break
            ln = int(rng.integers(1, min(n // 2, 400) + 1))
            src = int(rng.integers(0, n - ln))
            dst = int(rng.integers(src, n - ln + 1))
            x[dst:dst + ln] = x[src:src + ln].copy()
    else:  # zero-heavy with sparse reserved symbols
        x = np.zeros(n, dtype=np.uint8)
        if n:
            idx = rng.integers(0, n, n // 20 + 1)
            x[idx] = rng.choice(np.array([1, 2, 0xFE, 0xFF, 0x80], dtype=np.uint8), len(idx))
    return bytes(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batches", type=int, default=30)
    ap.add_argument("--streams", type=int, default=40)
    args = ap.parse_args()
    import compress_amd
    from compress_amd import batch as B, synth
    import oracle_lib as O
    rng = np.random.default_rng(args.seed)
    dicts = [b"", synth.dictionary(), bytes(rng.integers(0, 4, 3000, dtype=np.uint8)), bytes(300) + b"abc" * 50]
    t0 = time.time()
    n_cases = 0
    for bi in range(args.batches):
        d = dicts[bi % len(dicts)]
        blobs = [gen(rng, synth) for _ in range(60)]
        ctx = B.new_ctx(d)
        ctx.set_overlap(1 + bi % 3)
        got = compress_amd.compress_batch(ctx, blobs)
        oc = O.Compressor(d)
        for k, x in enumerate(blobs):
            want = oc.compress(x)
            assert got[k] == want, f"batch {bi} blob {k} (n={len(x)}, dict {bi % len(dicts)})"
            assert compress_amd.Decompress(got[k], d) == O.decompress(want, d)
        n_cases += len(blobs)
        print(f"batch {bi}: {len(blobs)} blobs ok ({time.time() - t0:.0f} s)", flush=True)
    for si in range(args.streams):
        d = dicts[si % len(dicts)]
        gc, oc = compress_amd.Compressor(d), O.Compressor(d)
        wrote = False  # Revert undoes the last Write, once (compress.go:334-345)
        for step in range(25):
            r = rng.random()
            if r < 0.7 or (r < 0.85 and not wrote):
                x = gen(rng, synth)[:3000]
                gc.Write(x)
                oc.write(x)
                wrote = True
            elif r < 0.85:
                gc.Revert()
                oc.revert()
                wrote = False
            elif r < 0.93:
                wrote = False
                assert gc.ConsiderBypassing() == oc.consider_bypassing()
            else:
                gc.Reset()
                oc.reset()
                wrote = False
            assert gc.Bytes() == oc.bytes(), f"stream {si} step {step}"
            assert gc.Len() == oc.len()
        print(f"stream {si}: ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all equal: {n_cases} blobs, {args.streams} streams")


if __name__ == "__main__":
    main()
