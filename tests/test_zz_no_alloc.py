"""GPU test of mi355_lzss_write_noalloc (include/mi355_lzss.h): the allocation-free variant of mi355_lzss_write.
Called through the C ABI with device buffers; its bits must equal mi355_lzss_write's (bit-exact, the bar of every
byte-level test here) on two calls in a row, the second with a smaller input - the static scratch array keeps the
first call's contents.  Collected last on purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write(lib, ctx, data, start, bit_off):
    a = np.frombuffer(data, dtype=np.uint8)
    cap = (bit_off + 37 * (len(data) - start) + 7) // 8 + 16
    out = np.zeros(cap, dtype=np.uint8)
    nbits = C.c_uint64()
    rc = lib.mi355_lzss_write(ctx.h, a.ctypes.data_as(C.c_void_p) if a.size else None, a.size, start, bit_off,
                              out.ctypes.data_as(C.c_void_p), cap, C.byref(nbits))
    assert rc == 0
    return out[:(bit_off + nbits.value + 7) // 8].tobytes(), nbits.value


def _write_noalloc(lib, ctx, torch, data, start, bit_off, out_cap=None):
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(data) or bytearray(1), dtype=torch.uint8).to(dev)
    cap = out_cap if out_cap is not None else (bit_off + 37 * (len(data) - start) + 7) // 8 + 16
    d_out = torch.full((cap,), 0xAA, dtype=torch.uint8, device=dev)   # the entry point must clear what it ORs into
    d_nbits = torch.zeros(1, dtype=torch.int64, device=dev)
    d_status = torch.full((1,), 77, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rc = lib.mi355_lzss_write_noalloc(ctx.h, d_in.data_ptr(), len(data), start, bit_off, d_out.data_ptr(), cap,
                                      d_nbits.data_ptr(), d_status.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "device memory changed hands during an allocation-free call"
    nbits = int(d_nbits.item())
    return bytes(d_out[:(bit_off + nbits + 7) // 8].cpu().numpy()), nbits, int(d_status.item())


def test_write_noalloc_equals_write():
    import torch
    from compress_amd import _lib, batch, synth
    lib = _lib.lib()
    dict_ = synth.dictionary()
    ctx = batch.new_ctx(dict_)
    big = synth.blob(0)                      # 131072 B = the entry point's maximum, every phrase kind
    small = synth.blob(1)[:20001]
    # two calls in a row on the shared scratch array, the second with a smaller input; then a streaming-style call
    for data, start, bit_off in [(big, 0, 0), (small, 0, 0), (big[:50000], 30000, 5), (b"", 0, 0), (bytes(300), 0, 3)]:
        want, want_bits = _write(lib, ctx, data, start, bit_off)
        got, got_bits, st = _write_noalloc(lib, ctx, torch, data, start, bit_off)
        assert st == 0
        assert got_bits == want_bits, (len(data), start, bit_off)
        assert got == want, (len(data), start, bit_off)
    # a second context on the same GPU shares the array and still gives the same bits
    ctx2 = batch.new_ctx(b"")
    want, want_bits = _write(lib, ctx2, small, 0, 0)
    got, got_bits, st = _write_noalloc(lib, ctx2, torch, small, 0, 0)
    assert (got, got_bits, st) == (want, want_bits, 0)


def test_write_noalloc_limits():
    import torch
    from compress_amd import _lib, batch, synth
    lib = _lib.lib()
    ctx = batch.new_ctx(synth.dictionary())
    dev = torch.device("cuda", 0)
    d_in = torch.zeros(131073, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(1024, dtype=torch.uint8, device=dev)
    d_nbits = torch.zeros(1, dtype=torch.int64, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.mi355_lzss_write_noalloc(ctx.h, d_in.data_ptr(), 131073, 0, 0, d_out.data_ptr(), 1024, d_nbits.data_ptr(),
                                      d_status.data_ptr(), None)
    assert rc == _lib.E_INPUT_SIZE
    assert "input size must be <= 131072" in _lib.last_error()
    # an output buffer that is too small is reported through the device status word, nothing is written past it
    data = synth.blob(2)[:30000]
    got, nbits, st = _write_noalloc(lib, ctx, torch, data, 0, 0, out_cap=100)
    assert st == _lib.E_CAPACITY and nbits > 800
