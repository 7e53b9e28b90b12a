"""GPU tests of the host-buffer paths added in round 2: the chunked copy / compute / copy pipeline behind
mi355_lzss_compress_batch, the multi-context call, the thread-safe CompressedSize256k and the decode-only context.
Everything is byte-compared with the serial device-resident path and with the CPU oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import compress_amd
    from compress_amd import batch, synth
    compress_amd.batch = batch
    compress_amd.synth = synth
    return compress_amd


def _ragged(amd, count, seed):
    rng = np.random.default_rng(seed)
    blobs = [amd.synth.blob(100 + k)[: int(rng.integers(1, 30000))] for k in range(count)]
    blobs[3] = b""
    blobs[7] = bytes(5000)
    return blobs


def test_pipelined_chunks_equal_the_serial_path(amd, dict_naive):
    """mi355_lzss_compress_batch cuts the batch into chunks that overlap upload, kernels and download; whatever the
    chunk size the bytes equal the device-resident upload / compress / download path and the oracle"""
    blobs = _ragged(amd, 60, 5)
    ctx = amd.batch.new_ctx(dict_naive)
    b = amd.batch.Batch(ctx, len(blobs), sum(map(len, blobs)))
    b.upload(blobs)
    b.compress()
    serial = b.download()
    oc = O.Compressor(dict_naive)
    assert serial[:10] == [oc.compress(d) for d in blobs[:10]]
    for chunk in (1 << 30, 200000, 40000, 1):      # 1 chunk, a few, many, one blob per chunk
        ctx.set_chunk_bytes(chunk)
        for _ in range(2):                          # the second call reuses both pipeline slots
            assert amd.compress_batch(ctx, blobs) == serial, chunk
    # pinned caller buffers through the raw C ABI
    from compress_amd import _lib
    from compress_amd.lzss import _csr
    flat, off = _csr(blobs)
    pin_in = amd.PinnedBuffer(flat.size)
    pin_in.array[:flat.size] = flat
    caps = np.array([3 + (37 * len(x) + 7) // 8 + 1 for x in blobs], dtype=np.uint64)
    out_off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    pin_out = amd.PinnedBuffer(int(out_off[-1]))
    out_len = np.zeros(len(blobs), dtype=np.uint32)
    status = np.zeros(len(blobs), dtype=np.int32)
    ctx.set_chunk_bytes(100000)
    rc = _lib.lib().mi355_lzss_compress_batch(ctx.h, pin_in.ptr, off.ctypes.data_as(C.c_void_p), len(blobs), pin_out.ptr,
                                              out_off.ctypes.data_as(C.c_void_p), out_len.ctypes.data_as(C.c_void_p),
                                              status.ctypes.data_as(C.c_void_p))
    assert rc == 0 and not status.any()
    got = [pin_out.array[int(out_off[k]):int(out_off[k]) + int(out_len[k])].tobytes() for k in range(len(blobs))]
    assert got == serial


def test_expanding_batch_outgrows_the_output_guess(amd, dict_naive):
    """the output buffer is sized from the input (x 1.25) without a host round trip; the reference's own expanding
    input (compress_test.go:428-475: x 1.4) outgrows it and takes the re-emit path"""
    from test_oracle import craft_expanding_input
    blobs = [craft_expanding_input(dict_naive, n) for n in (60000, 40000, 60000)]
    ctx = amd.batch.new_ctx(dict_naive)
    got = amd.compress_batch(ctx, blobs)
    oc = O.Compressor(dict_naive)
    assert got == [oc.compress(d) for d in blobs]
    assert sum(map(len, got)) > 1.3 * sum(map(len, blobs))
    ctx.set_chunk_bytes(70000)      # and inside the pipeline
    assert amd.compress_batch(ctx, blobs) == got
    c = amd.Compressor(dict_naive)  # and for the single-buffer Write
    assert c.Compress(blobs[0]) == got[0]


def test_oversized_and_tiny_capacity_statuses(amd):
    from compress_amd import _lib
    from compress_amd.lzss import _csr
    ctx = amd.batch.new_ctx(b"")
    blobs = [b"abcabcabcabc" * 10, bytes(1000), b"xyz"]
    flat, off = _csr(blobs)
    out_off = np.array([0, 4, 200, 300], dtype=np.uint64)      # blob 0 does not fit 4 bytes
    out = np.zeros(300, dtype=np.uint8)
    out_len = np.full(3, 99, dtype=np.uint32)
    status = np.full(3, 99, dtype=np.int32)
    rc = _lib.lib().mi355_lzss_compress_batch(ctx.h, flat.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 3,
                                              out.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(C.c_void_p),
                                              out_len.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert status.tolist() == [_lib.E_CAPACITY, 0, 0]
    assert out_len[0] == 0 and out_len[1] > 3 and out_len[2] == 6


def test_multi_context_call(amd, dict_naive):
    """mi355_lzss_compress_batch_multi / decompress_batch_multi: contiguous ranges balanced by bytes, one host thread
    per context (here: two contexts on the one GPU of the box), same bytes as one context"""
    blobs = _ragged(amd, 41, 9)
    one = amd.batch.new_ctx(dict_naive)
    want = amd.compress_batch(one, blobs)
    ctxs = [amd.batch.new_ctx(dict_naive) for _ in range(2)]
    for c in ctxs:
        c.set_chunk_bytes(150000)
    assert amd.compress_batch_multi(ctxs, blobs) == want
    assert amd.compress_batch_multi(ctxs[:1], blobs) == want
    three = ctxs + [one]
    assert amd.compress_batch_multi(three, blobs[:2]) == want[:2]     # fewer blobs than contexts
    back = amd.decompress_batch_multi(ctxs, want, 30000)
    assert back == blobs
    first = amd.partition(np.cumsum([0] + [len(b) for b in blobs]), 2)
    assert first[0] == 0 and first[-1] == len(blobs) and 0 < first[1] < len(blobs)


def test_compressed_size_is_thread_safe(amd, dict_naive):
    """CompressedSize256k 'is state less and thread-safe' (compress.go:381-383): four threads on ONE compressor, each
    call in a pool workspace of its own; and the batch form"""
    gc, oc = amd.Compressor(dict_naive), O.Compressor(dict_naive)
    inputs = [amd.synth.blob(200 + k)[: 3000 + 2500 * k] for k in range(8)] + [b"", bytes(70000)]
    want = [oc.compressed_size_256k(d) for d in inputs]
    assert gc.CompressedSizes256k(inputs) == want
    errors, results = [], {}

    def worker(t):
        try:
            for rep in range(3):
                for k in range(t, len(inputs), 4):
                    results[(t, rep, k)] = gc.CompressedSize256k(inputs[k])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors
    assert all(v == want[k] for (_, _, k), v in results.items()) and len(results) == 3 * len(inputs)
    # the other entry points of the same compressor still work afterwards
    assert gc.Compress(inputs[2]) == oc.compress(inputs[2])
    with pytest.raises(amd.LzssError, match="input size must be <= 262144"):
        gc.CompressedSizes256k([b"a", bytes(262145)])


def test_decoder_context(amd, dict_naive):
    """Decompress only augments the dictionary (decompress.go:34): a decode-only context builds no index, is cached per
    dictionary, and refuses to compress"""
    from compress_amd import lzss
    oc = O.Compressor(dict_naive)
    data = amd.synth.blob(300)[:50000]
    stream = oc.compress(data)
    for _ in range(3):
        assert amd.Decompress(stream, dict_naive) == data
    assert len(lzss._DECODERS) >= 1
    dec = lzss._Ctx(dict_naive, 0, decoder=True)
    assert amd.decompress_batch(dec, [stream, stream], 60000) == [data, data]
    with pytest.raises(amd.LzssError, match="decompression only"):
        amd.compress_batch(dec, [data])
    # nil dictionary and a stream that refers to the augmented symbols
    s2 = O.Compressor(b"").compress(bytes([0xFD, 0xFE, 0xFF]) * 50)
    assert amd.Decompress(s2, b"") == bytes([0xFD, 0xFE, 0xFF]) * 50
