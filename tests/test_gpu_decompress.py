"""GPU parity tests (-m gpu) of the batched decompressor against the oracle's Decompress
(lzss/decompress.go:17-84), through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import fixture_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import compress_amd
    from compress_amd import batch, synth
    compress_amd.batch = batch
    compress_amd.synth = synth
    return compress_amd


def test_known_streams(amd):
    assert amd.Decompress(bytes.fromhex("00010001fe040000"), b"") == bytes([1] * 6)  # compress_test.go:379-392
    assert amd.Decompress(bytes.fromhex("000100fdff01000010"), b"") == bytes([0xFD, 0xFE, 0xFF])
    assert amd.Decompress(bytes.fromhex("00010000fe060000"), b"") == bytes(8)
    assert amd.Decompress(b"\x00\x01\x00", b"") == b""
    assert amd.Decompress(b"\x00\x01\x01abc", b"") == b"abc"     # NoCompression, decompress.go:29-31


def test_invalid_streams(amd):
    field = (1 - 255 - 1) & 0x3FFF                                   # compress_test.go:394-404
    bits = (0xFE << 22) | (4 << 14) | field
    bad = bytes.fromhex("000100") + (bits << 2).to_bytes(4, "big")
    with pytest.raises(amd.LzssError):
        amd.Decompress(bad, b"")
    with pytest.raises(amd.LzssError, match="unsupported compressor version"):
        amd.Decompress(b"\x00\x02\x00", b"")
    with pytest.raises(amd.LzssError, match="expected 0 or 1"):
        amd.Decompress(b"\x00\x01\x02", b"")
    with pytest.raises(amd.LzssError):
        amd.Decompress(b"\x00\x01", b"")
    with pytest.raises(amd.LzssError):                              # truncated reference -> the reader's EOF
        amd.Decompress(bytes.fromhex("00010001fe04"), b"")
    # a long reference that starts in the dictionary but would run past its end (decompress.go:66-69)
    ctx = amd.batch.new_ctx(b"abcdef")
    bits = (0xFF << 29) | (4 << 21) | (2 - 1)   # len 5, address 2 with empty output: dict[6:11) of 8
    res = amd.decompress_batch(ctx, [bytes.fromhex("000100") + (bits << 3).to_bytes(5, "big")], 64)
    assert isinstance(res[0], amd.LzssError)
    with pytest.raises(O.OracleError):
        O.decompress(bytes.fromhex("000100") + (bits << 3).to_bytes(5, "big"), b"abcdef")


def test_batch_roundtrip_matches_oracle(amd, dict_naive, average_block):
    rng = np.random.default_rng(9)
    inputs = [b"", b"a", bytes(8), bytes(300), b"hi" + bytes(300), bytes([0xFF]) * 500, bytes([0xFE]) * 170 + b"x",
              b"abc" * 700, average_block[:50000], fixture_bytes("3c2943_data.bin"), fixture_bytes("blobs_1-goerli-3690632")]
    for k in (2, 3, 256):
        inputs.append(bytes(rng.integers(0, k, 3000, dtype=np.uint8)))
    oc = O.Compressor(dict_naive)
    streams = [bytes(oc.compress(d)) for d in inputs]
    streams.append(b"\x00\x01\x01" + b"raw bytes")
    inputs.append(b"raw bytes")
    ctx = amd.batch.new_ctx(dict_naive)
    caps = [len(d) + 7 for d in inputs]
    got = amd.decompress_batch(ctx, streams, caps)
    for k, d in enumerate(inputs):
        assert got[k] == d, f"blob {k}"
        assert O.decompress(streams[k], dict_naive) == d
    # capacity one byte short -> per-blob status, the others are unaffected
    caps[8] = len(inputs[8]) - 1
    got = amd.decompress_batch(ctx, streams, caps)
    assert isinstance(got[8], amd.LzssError) and got[8].code == -5
    assert got[9] == inputs[9]


def test_fast_loop_boundaries(amd, dict_naive, average_block):
    """decompress_kernel's fast loop runs while 512 bits of stream and 320 bytes of capacity are left, the checked loop
    finishes: capacities from exact to generous, streams cut at every length around the switch, long literal runs (steps of
    63 literals), references shorter than a word (single reserved bytes), overlapping copies, dictionary sources"""
    rng = np.random.default_rng(77)
    inputs = [
        bytes(rng.integers(0, 254, 4000, dtype=np.uint8)),                                   # literals only
        bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),                                   # literals and one-byte references
        b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 9)) for _ in range(900)),   # short overlapping copies
        dict_naive[100:700] + bytes(rng.integers(0, 254, 200, dtype=np.uint8)) + dict_naive[50:400],   # dictionary sources
        average_block[:6000],
        bytes(rng.integers(0, 254, 70, dtype=np.uint8)),                                     # a stream of about 64 bytes
        b"ab" * 40 + bytes(rng.integers(0, 254, 30, dtype=np.uint8)),
    ]
    oc = O.Compressor(dict_naive)
    ctx = amd.batch.new_ctx(dict_naive)
    streams, caps, want = [], [], []
    for d in inputs:
        st = bytes(oc.compress(d))
        for extra in (0, 1, 63, 64, 319, 320, 321, 1000):
            streams.append(st); caps.append(len(d) + extra); want.append(d)
        streams.append(st); caps.append(len(d) - 1); want.append(None)                       # one byte short: E_CAPACITY
        for ln in sorted({len(st) - c for c in list(range(1, 12)) + [60, 64, 65, 70]} | {66, 67, 68, 69}):
            if not 3 <= ln < len(st):
                continue
            t = st[:ln]                                                                      # a stream cut short
            try:
                w = O.decompress(t, dict_naive)
            except O.OracleError:
                w = None
            streams.append(t); caps.append(len(d) + 400); want.append(w)
    got = amd.decompress_batch(ctx, streams, caps)
    for k in range(len(streams)):
        if want[k] is None:
            assert isinstance(got[k], amd.LzssError), k
        else:
            assert got[k] == want[k], k


def test_gpu_compress_then_gpu_decompress_config5(amd):
    """config 5 shape: compressed synthetic 128 KiB blobs (overlapping run-length references included)"""
    dict_ = amd.synth.dictionary()
    blobs = amd.synth.blobs(100, 12)
    ctx = amd.batch.new_ctx(dict_)
    comp = amd.compress_batch(ctx, blobs)
    oc = O.Compressor(dict_)
    assert comp[0] == oc.compress(blobs[0])
    assert oc.phrase_stats()["rle"] > 0
    b = amd.batch.Batch(ctx, len(blobs), sum(map(len, blobs)))
    b.upload_compressed(comp, 131072)
    b.decompress()
    back = b.download_decompressed()
    assert back == blobs
    assert b.stage_times()["decompress"] > 0


def test_streaming_blobs_decode_on_gpu(amd, dict_naive, average_block):
    """blobs produced by the Write/Revert flow (non-zero pad bits after Revert) decode to the exact spans"""
    data = average_block[:20000]
    gc = amd.Compressor(dict_naive)
    spans, streams = [], []
    i0 = 0
    while i0 < len(data):
        i = i0
        while i < len(data) and gc.Len() < 3000:
            gc.Write(data[i:i + 500])
            i += 500
        if gc.Len() > 3000:
            gc.Revert()
            i -= 500
        spans.append(data[i0:min(i, len(data))])
        streams.append(gc.Bytes())
        gc.Reset()
        i0 = i
    got = amd.decompress_batch(gc, streams, 20000)
    assert got == spans


@pytest.mark.gpu
def test_decompress_from_several_threads_with_one_dictionary(dict_naive):
    """Decompress is a pure function in the reference (decompress.go:17-84); here its decode-only context comes from a
    cache, so threads that share a dictionary share a context and must take turns on it, and more dictionaries than the
    cache holds make it evict (and close) contexts while other threads decompress"""
    import threading
    import compress_amd as amd
    from compress_amd import synth
    comp = amd.Compressor(dict_naive, device=0)
    datas = [synth.blob(k, 20000 + 977 * k) for k in range(6)]
    streams = [comp.Compress(d) for d in datas]
    dicts = [bytes([k]) * (k + 1) for k in range(7)]     # > the cache's 4 entries
    small = [(amd.Compressor(dk, device=0).Compress(b"abcabcabc" * 50 + dk), dk) for dk in dicts]
    errors = []

    def work(t):
        try:
            for rep in range(12):
                k = (t + rep) % len(datas)
                if amd.Decompress(streams[k], dict_naive) != datas[k]:
                    errors.append(("shared", t, rep))
                s, dk = small[(3 * t + rep) % len(small)]
                if amd.Decompress(s, dk) != b"abcabcabc" * 50 + dk:
                    errors.append(("evict", t, rep))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert errors == []
