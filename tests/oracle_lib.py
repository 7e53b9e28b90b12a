"""ctypes binding of the CPU oracle (oracle/liblzss_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  The product package (compress_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liblzss_oracle.so")

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "liblzss_oracle.so"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build_oracle()
    lib = C.CDLL(_LIB_PATH)
    lib.ora_suffix_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ora_suffix_array.restype = None
    lib.ora_lookup_longest.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, _i32p, _i32p]
    lib.ora_lookup_longest.restype = None
    lib.ora_augment_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ora_augment_dict.restype = C.c_size_t
    lib.ora_new_compressor.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.ora_new_compressor.restype = C.c_void_p
    lib.ora_free_compressor.argtypes = [C.c_void_p]
    lib.ora_free_compressor.restype = None
    lib.ora_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.ora_write.restype = C.c_long
    lib.ora_reset.argtypes = [C.c_void_p]
    lib.ora_reset.restype = None
    for name in ("ora_len", "ora_written"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = C.c_size_t
    for name in ("ora_bytes", "ora_written_bytes"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = C.c_void_p
    lib.ora_revert.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.ora_revert.restype = C.c_int
    lib.ora_consider_bypassing.argtypes = [C.c_void_p]
    lib.ora_consider_bypassing.restype = C.c_int
    lib.ora_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.ora_compress.restype = C.c_int
    lib.ora_compressed_size_256k.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.ora_compressed_size_256k.restype = C.c_long
    lib.ora_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    lib.ora_decompress.restype = C.c_int
    lib.ora_free.argtypes = [C.c_void_p]
    lib.ora_free.restype = None
    lib.ora_best_backrefs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    lib.ora_best_backrefs.restype = C.c_int
    lib.ora_phrase_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.ora_phrase_stats.restype = None
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _buf(b):
    """bytes-like -> (ctypes pointer-compatible object, length); keeps a reference alive."""
    b = bytes(b) if not isinstance(b, (bytes, bytearray)) else b
    arr = (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))
    return arr, len(b)


class OracleError(Exception):
    pass


def suffix_array(data: bytes) -> np.ndarray:
    arr, n = _buf(data)
    sa = np.zeros(max(n, 1), dtype=np.int32)
    lib().ora_suffix_array(arr, n, sa.ctypes.data)
    return sa[:n]


def lookup_longest(data: bytes, sa: np.ndarray, s: bytes, min_end, max_end, rs, re):
    darr, n = _buf(data)
    sarr, _ = _buf(s)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    idx = C.c_int32()
    ln = C.c_int32()
    lib().ora_lookup_longest(darr, n, sa.ctypes.data, sarr, min_end, max_end, rs, re, C.byref(idx), C.byref(ln))
    return idx.value, ln.value


def augment_dict(dict_: bytes) -> bytes:
    darr, n = _buf(dict_)
    out = (C.c_uint8 * (n + 2))()
    m = lib().ora_augment_dict(darr, n, out)
    return bytes(out[:m])


class Compressor:
    """Mirror of the reference's lzss.Compressor over the oracle (lzss/compress.go:11-379)."""

    def __init__(self, dict_: bytes = b""):
        dict_ = dict_ or b""
        darr, n = _buf(dict_)
        err = C.create_string_buffer(256)
        self._h = lib().ora_new_compressor(darr, n, err, 256)
        if not self._h:
            raise OracleError(err.value.decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().ora_free_compressor(h)
            self._h = None

    def write(self, d: bytes) -> int:
        arr, n = _buf(d)
        err = C.create_string_buffer(256)
        r = lib().ora_write(self._h, arr, n, err, 256)
        if r < 0:
            raise OracleError(err.value.decode())
        return r

    def reset(self):
        lib().ora_reset(self._h)

    def len(self) -> int:
        return lib().ora_len(self._h)

    def written(self) -> int:
        return lib().ora_written(self._h)

    def written_bytes(self) -> bytes:
        return C.string_at(lib().ora_written_bytes(self._h), self.written())

    def bytes(self) -> bytes:
        return C.string_at(lib().ora_bytes(self._h), self.len())

    def revert(self):
        err = C.create_string_buffer(256)
        if lib().ora_revert(self._h, err, 256) != 0:
            raise OracleError(err.value.decode())

    def consider_bypassing(self) -> bool:
        return bool(lib().ora_consider_bypassing(self._h))

    def compress(self, d: bytes) -> bytes:
        arr, n = _buf(d)
        err = C.create_string_buffer(256)
        if lib().ora_compress(self._h, arr, n, err, 256) != 0:
            raise OracleError(err.value.decode())
        return self.bytes()

    def compressed_size_256k(self, d: bytes) -> int:
        arr, n = _buf(d)
        err = C.create_string_buffer(256)
        r = lib().ora_compressed_size_256k(self._h, arr, n, err, 256)
        if r < 0:
            raise OracleError(err.value.decode())
        return r

    def best_backrefs(self, d: bytes, start: int = 0):
        """(type u8[n-start], addr i32, len i32) of bestBackref at every position >= start."""
        arr, n = _buf(d)
        m = max(n - start, 0)
        addr = np.zeros(max(m, 1), dtype=np.int32)
        ln = np.zeros(max(m, 1), dtype=np.int32)
        ty = np.zeros(max(m, 1), dtype=np.uint8)
        lib().ora_best_backrefs(self._h, arr, n, start, addr.ctypes.data, ln.ctypes.data, ty.ctypes.data)
        return ty[:m], addr[:m], ln[:m]

    def phrase_stats(self):
        st = np.zeros(6, dtype=np.uint64)
        lib().ora_phrase_stats(self._h, st.ctypes.data)
        return dict(zip(("literals", "short", "long_data", "long_dict", "reserved", "rle"), map(int, st)))


def decompress(data: bytes, dict_: bytes = b"") -> bytes:
    dict_ = dict_ or b""
    arr, n = _buf(data)
    darr, dn = _buf(dict_)
    out = C.c_void_p()
    out_len = C.c_size_t()
    err = C.create_string_buffer(256)
    rc = lib().ora_decompress(arr, n, darr, dn, C.byref(out), C.byref(out_len), err, 256)
    if rc != 0:
        raise OracleError(err.value.decode())
    try:
        return C.string_at(out.value, out_len.value)
    finally:
        lib().ora_free(out)


def compressed_stream_info(c: bytes, dict_: bytes = b""):
    """Literal transcription of CompressedStreamInfo (lzss/decompress.go:97-206) in plain Python (test
    infrastructure, small cases only): byte-at-a-time copying into `out` (which starts with the augmented
    dictionary), literals grouped, StartCompressed in bits after the header.  Returns a list of tuples
    (type, length, reference_address, start_decompressed, start_compressed, content)."""
    c = bytes(c)
    if len(c) < 3:
        raise OracleError("failed to read header")
    version, noc = (c[0] << 8) | c[1], c[2]
    if version != 1:
        raise OracleError("unsupported compressor version")
    if noc == 1:                                   # decompress.go:109-118
        return [(0, len(c) - 3, 0, 0, 0, c[3:])]
    out = bytearray(augment_dict(dict_))            # decompress.go:123-132
    res = []
    bits = "".join(format(b, "08b") for b in c[3:])
    pos = 0
    err = False

    def read(n):                                    # bitio TryReadBits: sets the error at EOF
        nonlocal pos, err
        if err or pos + n > len(bits):
            err = True
            return 0
        v = int(bits[pos:pos + n], 2)
        pos += n
        return v

    lit_start, in_i = -1, 0

    def emit_literal():                             # decompress.go:141-155
        nonlocal lit_start, in_i
        if lit_start == -1:
            return
        res.append((0, len(out) - lit_start, lit_start, lit_start, in_i, bytes(out[lit_start:])))
        in_i += (len(out) - lit_start) * 8
        lit_start = -1

    s = read(8)
    while not err:                                  # decompress.go:172-203
        if s == 0xFE or s == 0xFF:
            emit_literal()
            abits = 14 if s == 0xFE else 21
            length = read(8) + 1
            address = read(abits) + 1
            if err:
                raise OracleError("EOF")
            for _ in range(length):
                out.append(out[len(out) - address])
            res.append((s, length, len(out) - length - address, len(out) - length, in_i, bytes(out[len(out) - length:])))
            in_i += 8 + 8 + abits
        else:
            if lit_start == -1:
                lit_start = len(out)
            out.append(s)
        s = read(8)
    emit_literal()
    return res


def to_csv(phrases) -> bytes:
    """CompressionPhrases.ToCSV (lzss/decompress.go:208-236)"""
    b = "type,length,start_decompressed (bytes),start_compressed (bits),reference_address,content (hex)\n"
    for ty, length, ref, sd, sc, content in phrases:
        b += {0xFE: "short,", 0xFF: "long,", 0: "literal,"}[ty]
        b += f"{length},{sd},{sc},{ref},{content.hex()}\n"
    return b.encode()
