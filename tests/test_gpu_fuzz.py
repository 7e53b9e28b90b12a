"""A short slice of scripts/fuzz_gpu.py in the GPU suite: generated inputs (runs around the RLE threshold, reserved
symbols, periodic and zero-heavy texts, copies at all distances) in ragged batches with every overlap setting, and
random Write / Revert / ConsiderBypassing / Reset sequences, byte for byte against the oracle."""
import importlib.util
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "scripts", "fuzz_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12])
def test_generated_batches_match_the_oracle(seed, dict_naive):
    import compress_amd
    from compress_amd import batch as B, synth
    gen = _fuzz().gen
    rng = np.random.default_rng(seed)
    for bi, d in enumerate([b"", dict_naive, bytes(rng.integers(0, 4, 3000, dtype=np.uint8))]):
        blobs = [gen(rng, synth) for _ in range(40)]
        ctx = B.new_ctx(d)
        ctx.set_overlap(1 + (bi + seed) % 3)
        got = compress_amd.compress_batch(ctx, blobs)
        oc = O.Compressor(d)
        for k, x in enumerate(blobs):
            want = oc.compress(x)
            assert got[k] == want, f"seed {seed} batch {bi} blob {k} (n={len(x)})"
            assert compress_amd.Decompress(got[k], d) == x


@pytest.mark.parametrize("seed", [21, 22])
def test_random_write_revert_sequences_match_the_oracle(seed, dict_naive):
    import compress_amd
    from compress_amd import synth
    gen = _fuzz().gen
    rng = np.random.default_rng(seed)
    for d in (b"", dict_naive):
        gc, oc = compress_amd.Compressor(d), O.Compressor(d)
        wrote = False
        for step in range(30):
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
                assert gc.ConsiderBypassing() == oc.consider_bypassing()
                wrote = False
            else:
                gc.Reset()
                oc.reset()
                wrote = False
            assert gc.Bytes() == oc.bytes(), f"seed {seed} step {step}"
            assert gc.Len() == oc.len()
