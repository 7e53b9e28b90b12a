"""The C twin of the synthetic generator (compress_amd/synth_fast.c, used by bench.py) produces synth.py's bytes."""
import numpy as np

from compress_amd import synth


def test_c_generator_equals_python():
    arr = synth.blobs_array(5, 3, 9)
    assert arr.shape == (9, synth.BLOB_SIZE)
    for k in range(9):
        assert arr[k].tobytes() == synth.blob(5 + 3 * k)


def test_c_generator_other_sizes_and_single():
    for size in (1, 4097, 200000):
        assert synth.blobs_array(1234, 1, 1, size)[0].tobytes() == synth.blob(1234, size)
    assert synth.blobs_array(0, 1, 0).shape == (0, synth.BLOB_SIZE)
    big = synth.blobs_array(0, 8, 40, threads=3)
    assert all(big[k].tobytes() == synth.blob(8 * k) for k in (0, 1, 13, 14, 26, 27, 39))
