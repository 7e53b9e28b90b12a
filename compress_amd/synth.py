"""Synthetic "EVM-calldata-like" blobs and dictionary.

Deterministic (splitmix64, fixed seeds), numpy only.  Statistics are shaped after the reference's
fixtures (lzss/testdata): 55-72 % zero bytes, ~1 % 0xFF, ABI-word structure, a few zero runs >= 160.
"""
import numpy as np

SEED_POOL = 0xC0FFEE
SEED_BLOB = 0x5EED0000
SEED_DICT = 0xD1C7
BLOB_SIZE = 131072
DICT_SIZE = 65536

_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _splitmix_stream(seed: int, count: int) -> np.ndarray:
    """The first `count` outputs of splitmix64(seed), vectorised."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class _Rng:
    def __init__(self, seed: int, count: int = 1 << 16):
        self.seed = seed
        self.buf = _splitmix_stream(seed, count)
        self.i = 0

    def u64(self) -> int:
        if self.i >= self.buf.size:
            more = _splitmix_stream(self.seed + int(self.buf.size) * int(_GAMMA), self.buf.size)
            self.buf = np.concatenate([self.buf, more])
        v = int(self.buf[self.i])
        self.i += 1
        return v

    def below(self, n: int) -> int:
        return self.u64() % n

    def rbytes(self, n: int) -> bytes:
        out = bytearray()
        while len(out) < n:
            out += self.u64().to_bytes(8, "little")
        return bytes(out[:n])


_pools = None


def pools():
    global _pools
    if _pools is None:
        r = _Rng(SEED_POOL)
        sel = [r.rbytes(4) for _ in range(64)]
        addr = [r.rbytes(20) for _ in range(512)]
        words = [r.rbytes(32) for _ in range(128)]
        _pools = (sel, addr, words)
    return _pools


def _abi_word(r: _Rng, out: bytearray):
    sel, addr, words = pools()
    u = r.below(100)
    if u < 40:                      # small uint
        k = 1 + r.below(4)
        out += bytes(32 - k) + r.rbytes(k)
    elif u < 63:                    # address
        out += bytes(12) + (addr[r.below(512)] if r.below(10) < 8 else r.rbytes(20))
    elif u < 77:
        out += bytes(32)
    elif u < 87:
        out += words[r.below(128)]
    elif u < 92:
        out += r.rbytes(32)
    elif u < 93:
        out += b"\xff" * 32
    elif u < 94:                    # small negative int256
        out += b"\xff" * 30 + b"\xfe" + r.rbytes(1)
    else:                           # dynamic bytes: length word + payload padded to 32
        ln = 1 + r.below(100)
        out += bytes(28) + ln.to_bytes(4, "big") + r.rbytes(ln) + bytes((-ln) % 32)


def _transaction(r: _Rng, out: bytearray):
    sel, _, _ = pools()
    out += r.rbytes(3 + r.below(7))
    out += sel[min(r.below(64), r.below(64))]
    for _ in range(1 + r.below(12)):
        _abi_word(r, out)
    if r.below(64) == 0:
        out += bytes(160 + r.below(865))


def blob(b: int, size: int = BLOB_SIZE) -> bytes:
    """Synthetic blob number b (seed 0x5EED0000 + b)."""
    r = _Rng(SEED_BLOB + b)
    out = bytearray()
    while len(out) < size:
        _transaction(r, out)
    return bytes(out[:size])


def dictionary(size: int = DICT_SIZE) -> bytes:
    sel, addr, words = pools()
    r = _Rng(SEED_DICT)
    out = bytearray()
    for s in sel:
        out += s
    for a in addr:
        out += a
    for w in words:
        out += w
    while len(out) < size:
        _abi_word(r, out)
    return bytes(out[:size])


def blobs(first: int, count: int, size: int = BLOB_SIZE):
    return [blob(first + k, size) for k in range(count)]


def blobs_array(first: int, stride: int, count: int, size: int = BLOB_SIZE, threads: int = 8) -> np.ndarray:
    """Blobs first, first + stride, ... as ONE (count, size) uint8 array, from the C twin of this generator
    (compress_amd/synth_fast.c -> libsynth_fast.so, built by compress_amd/csrc/Makefile; ~0.6 ms per blob and no forked
    workers, which bench.py must not start under rocprofv3).  Byte for byte blob(): tests/test_synth_fast.py."""
    import ctypes
    import os
    import threading
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsynth_fast.so")
    if not os.path.exists(path):
        raise ImportError(path + " is missing: build it with `make -C compress_amd/csrc`")
    lib = ctypes.CDLL(path)
    lib.synth_fast_blobs.argtypes = [ctypes.c_uint64] * 4 + [ctypes.c_void_p]
    lib.synth_fast_blobs.restype = None
    out = np.empty((count, size), dtype=np.uint8)
    if count == 0:
        return out
    lib.synth_fast_blobs(first, stride, 1, size, out.ctypes.data)  # also fills the pools once, before the threads
    threads = max(1, min(threads, count - 1))
    per = (count - 1 + threads - 1) // threads if count > 1 else 0

    def work(t):
        lo, hi = 1 + t * per, min(count, 1 + (t + 1) * per)
        if lo < hi:
            lib.synth_fast_blobs(first + stride * lo, stride, hi - lo, size, out[lo].ctypes.data)
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)] if count > 1 else []
    for x in th:
        x.start()
    for x in th:
        x.join()
    return out
