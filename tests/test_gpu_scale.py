"""GPU test at batch scale: rare, timing-dependent faults (a race, a miscompiled divergent loop) show up as a few
blobs in thousands, which the small parity cases cannot see.  A 2000-blob batch is compressed three times: the
streams must be identical from run to run, equal the oracle's on a sample, and decode back to the inputs on the GPU."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_large_batch_is_deterministic_and_exact():
    import compress_amd
    from compress_amd import batch, synth
    base = [synth.blob(500 + k) for k in range(100)]
    rng = np.random.default_rng(2)
    blobs = list(base)
    while len(blobs) < 2000:            # splices of two base blobs: same statistics, different texts
        a, b = rng.integers(0, len(base), 2)
        j = int(rng.integers(1, 131072))
        blobs.append(base[a][j:] + base[b][:j])
    dict_ = synth.dictionary()
    ctx = batch.new_ctx(dict_)
    bt = batch.Batch(ctx, len(blobs), sum(map(len, blobs)))
    bt.upload(blobs)
    runs = []
    for parts in (1, 2, 1):
        ctx.set_overlap(parts)
        bt.compress()
        runs.append(bt.download())
    for r in runs[1:]:
        bad = [k for k in range(len(blobs)) if r[k] != runs[0][k]]
        assert not bad, f"{len(bad)} blobs changed between two runs of the same batch, first {bad[:5]}"
    oc = O.Compressor(dict_)
    for k in rng.choice(len(blobs), 40, replace=False):
        assert runs[0][int(k)] == oc.compress(blobs[int(k)]), f"blob {int(k)} differs from the oracle"
    bt.upload_compressed(runs[0], 131072)
    bt.decompress()
    assert bt.download_decompressed() == blobs
    # the pipelined host-buffer path over the same batch, several chunks
    ctx.set_chunk_bytes(400 * 131072)
    assert compress_amd.compress_batch(ctx, blobs) == runs[0]
