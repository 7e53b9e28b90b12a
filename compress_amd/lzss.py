"""Host-side mirror of the reference's Go package `lzss` over the C ABI (include/mi355_lzss.h).

Same names, argument meaning and error behaviour as lzss/compress.go, lzss/decompress.go and
lzss/header.go; the index build, the match search, the phrase walk and the bit emission of every
Write run on the GPU (mi355_lzss_write), the few bytes of stream state (header, re-opened last byte,
Revert bookkeeping) stay here exactly as in compress.go:91-129,289-367.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib

MaxInputSize = 1 << 22   # lzss/backref.go:10
MaxDictSize = 1 << 22    # lzss/backref.go:11
SymbolDynamic = 0xFF     # lzss/backref.go:15
SymbolShort = 0xFE       # lzss/backref.go:16
Version = 1              # lzss/header.go:11
HeaderSize = 3           # lzss/header.go:12


class LzssError(Exception):
    """The Go `error` of the reference API."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def _check(rc):
    if rc != 0:
        raise LzssError(_lib.last_error(), rc)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _as_u8(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def AugmentDict(dict_: bytes) -> bytes:
    """lzss/compress.go:70-88"""
    dict_ = bytes(dict_ or b"")
    found = 0
    for b in dict_:
        if b == SymbolShort:
            found |= 0b010
        elif b == SymbolDynamic:
            found |= 0b100
        else:
            continue
        if found == 0b110:
            return dict_
    return dict_ + bytes([SymbolShort, SymbolDynamic])


class Header:
    """lzss/header.go:17-44"""

    def __init__(self, version=Version, no_compression=False):
        self.Version = version
        self.NoCompression = no_compression

    def to_bytes(self) -> bytes:
        return self.Version.to_bytes(2, "big") + bytes([1 if self.NoCompression else 0])

    @classmethod
    def from_bytes(cls, b: bytes):
        if len(b) < HeaderSize:
            raise LzssError("unexpected EOF")
        if b[2] not in (0, 1):
            raise LzssError("expected 0 or 1")
        return cls(int.from_bytes(b[:2], "big"), b[2] == 1)


class _Ctx:
    """One dictionary on one device (mi355_lzss_ctx)."""

    def __init__(self, dict_: bytes, device: int = 0, decoder: bool = False):
        d = _as_u8(dict_ or b"")
        h = C.c_void_p()
        create = _lib.lib().mi355_lzss_ctx_create_decoder if decoder else _lib.lib().mi355_lzss_ctx_create
        _check(create(_ptr(d) if d.size else None, d.size, device, C.byref(h)))
        self.h = h
        self.device = device
        n = C.c_uint64()
        idx = (C.c_int32 * 2)()
        _check(_lib.lib().mi355_lzss_dict_info(self.h, C.byref(n), C.byref(idx)))
        self.dict_len = n.value
        self.reserved_idx = (idx[0], idx[1])

    def set_overlap(self, parts: int):
        """run the index / match stages of this context's batches as `parts` concurrent sub-batches (tuning knob)"""
        _check(_lib.lib().mi355_lzss_ctx_set_overlap(self.h, int(parts)))

    def set_chunk_bytes(self, nbytes: int):
        """chunk size of the copy / compute / copy pipeline behind compress_batch (tuning knob)"""
        _check(_lib.lib().mi355_lzss_ctx_set_chunk_bytes(self.h, int(nbytes)))

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().mi355_lzss_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Compressor:
    """lzss.Compressor (lzss/compress.go:11-32)."""

    def __init__(self, dict_: bytes = b"", device: int = 0):
        if len(AugmentDict(dict_ or b"")) > MaxDictSize:  # compress.go:39-41
            raise LzssError(f"dict size must be <= {MaxDictSize}", _lib.E_DICT_SIZE)
        self._ctx = _Ctx(dict_ or b"", device)
        self._out = bytearray()
        self._in = bytearray()
        self._pending = 0          # bits of the re-opened last byte (the bitio cache)
        self._npending = 0
        self.Reset()

    # -- compress.go:289-304
    def Reset(self):
        self._no_compression = False
        self._out = bytearray(Header(Version, False).to_bytes())
        self._in = bytearray()
        self._last_out_len = len(self._out)
        self._last_nb_skipped = 0
        self._nb_skipped = 0
        self._last_in_len = 0

    # -- compress.go:91-129
    def Write(self, d: bytes) -> int:
        d = bytes(d)
        # re-open the last output byte (compress.go:94-100)
        self._last_out_len = len(self._out)
        last = self._out.pop() >> self._nb_skipped
        nb = 8 - self._nb_skipped
        self._pending = ((self._pending << nb) | last) & ((1 << (self._npending + nb)) - 1)
        self._npending += nb
        while self._npending >= 8:       # a whole byte (nb_skipped == 0) goes straight back
            self._out.append((self._pending >> (self._npending - 8)) & 0xFF)
            self._npending -= 8
        self._pending &= (1 << self._npending) - 1
        self._last_nb_skipped = self._nb_skipped
        # appendInput, compress.go:472-479
        if len(self._in) + len(d) > MaxInputSize:
            raise LzssError(f"input size must be <= {MaxInputSize}", _lib.E_INPUT_SIZE)
        self._last_in_len = len(self._in)
        self._in += d
        if self._no_compression:
            self._out += d
            return len(d)
        # index + match + walk + emit of inBuf[lastInLen:] on the GPU
        inbuf = np.frombuffer(bytes(self._in), dtype=np.uint8)
        n, start = inbuf.size, self._last_in_len
        bit_off = self._npending
        cap = (bit_off + 37 * (n - start) + 7) // 8 + 16
        out = np.zeros(cap, dtype=np.uint8)
        nbits = C.c_uint64()
        _check(_lib.lib().mi355_lzss_write(self._ctx.h, _ptr(inbuf) if n else None, n, start, bit_off, _ptr(out), cap,
                                           C.byref(nbits)))
        total = bit_off + nbits.value
        nbytes = (total + 7) // 8
        if nbytes:
            if bit_off:
                out[0] |= (self._pending << (8 - bit_off)) & 0xFF
            self._out += out[:nbytes].tobytes()
        self._pending = 0
        self._npending = 0
        self._nb_skipped = (8 - total % 8) % 8   # bw.Align(), compress.go:127
        return n - start

    def Len(self) -> int:          # compress.go:307
        return len(self._out)

    def Written(self) -> int:      # compress.go:312
        return len(self._in)

    def WrittenBytes(self) -> bytes:   # compress.go:318
        return bytes(self._in)

    def Bytes(self) -> bytes:      # compress.go:370
        return bytes(self._out)

    # -- compress.go:324-345
    def Revert(self):
        if self._last_in_len == -1:
            raise LzssError("cannot revert twice in a row")
        del self._in[self._last_in_len:]
        self._last_in_len = -1
        if self._no_compression:
            data = bytes(self._in)
            self.Reset()
            self.Write(data)
            self.ConsiderBypassing()
            return
        del self._out[self._last_out_len:]
        self._nb_skipped = self._last_nb_skipped

    # -- compress.go:348-367
    def ConsiderBypassing(self) -> bool:
        if len(self._out) > len(self._in) + HeaderSize:
            self._no_compression = True
            self._nb_skipped = 0
            self._last_out_len = self._last_in_len + HeaderSize
            self._last_nb_skipped = 0
            self._out = bytearray(Header(Version, True).to_bytes()) + self._in
            return True
        return False

    # -- compress.go:375-379
    def Compress(self, d: bytes) -> bytes:
        self.Reset()
        self.Write(d)
        return self.Bytes()

    # -- compress.go:384-407
    def CompressedSize256k(self, d: bytes) -> int:
        if self._no_compression:
            return HeaderSize + len(d)
        if len(d) > (1 << 18):
            raise LzssError(f"input size must be <= {1 << 18}", _lib.E_INPUT_SIZE)
        a = _as_u8(d)
        size = C.c_uint64()
        _check(_lib.lib().mi355_lzss_compressed_size(self._ctx.h, _ptr(a) if a.size else None, a.size, C.byref(size)))
        return size.value

    def CompressedSizes256k(self, blobs):
        """CompressedSize256k of many independent inputs in one launch sequence (mi355_lzss_compressed_size_batch)"""
        if self._no_compression:
            return [HeaderSize + len(b) for b in blobs]
        flat, off = _csr(blobs)
        sizes = np.zeros(max(len(blobs), 1), dtype=np.uint64)
        _check(_lib.lib().mi355_lzss_compressed_size_batch(self._ctx.h, _ptr(flat) if flat.size else None, _ptr(off),
                                                           len(blobs), _ptr(sizes), None))
        return [int(x) for x in sizes[:len(blobs)]]

    def Close(self):
        """releases the device side (the Go shim's Close; the reference has no device to release)"""
        self._ctx.close()

    # -- diagnostics: the per-position hot path (compress.go:146-173)
    def best_backrefs(self, d: bytes, start: int = 0):
        a = _as_u8(d)
        m = max(a.size - start, 0)
        addr = np.zeros(max(m, 1), dtype=np.int32)
        ln = np.zeros(max(m, 1), dtype=np.int32)
        ty = np.zeros(max(m, 1), dtype=np.uint8)
        _check(_lib.lib().mi355_lzss_best_backrefs(self._ctx.h, _ptr(a) if a.size else None, a.size, start, _ptr(addr),
                                                   _ptr(ln), _ptr(ty)))
        return ty[:m], addr[:m], ln[:m]


def NewCompressor(dict_: bytes = b"", device: int = 0) -> Compressor:
    """lzss/compress.go:37-67"""
    return Compressor(dict_, device)


def _csr(blobs):
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in blobs], out=off[1:])
    flat = np.frombuffer(b"".join(bytes(b) for b in blobs), dtype=np.uint8) if len(blobs) else np.zeros(0, np.uint8)
    return flat, off


class PinnedBuffer:
    """A numpy uint8 view of pinned host memory (mi355_lzss_host_alloc): batch buffers placed here move by DMA."""

    def __init__(self, nbytes: int):
        self.nbytes = max(int(nbytes), 1)
        self.ptr = _lib.lib().mi355_lzss_host_alloc(self.nbytes)
        if not self.ptr:
            raise LzssError(_lib.last_error(), _lib.E_DEVICE)
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            _lib.lib().mi355_lzss_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition(in_off, n_parts: int):
    """contiguous blob ranges balanced by bytes, as the multi-device entry points cut a batch (mi355_lzss_partition)"""
    off = np.ascontiguousarray(in_off, dtype=np.uint64)
    first = np.zeros(n_parts + 1, dtype=np.uint32)
    _check(_lib.lib().mi355_lzss_partition(_ptr(off), off.size - 1, n_parts, _ptr(first)))
    return [int(x) for x in first]


def _ctx_array(ctxs):
    hs = [(c._ctx if isinstance(c, Compressor) else c).h for c in ctxs]
    return (C.c_void_p * len(hs))(*[h.value for h in hs])


def compress_batch_multi(ctxs, blobs):
    """compress_batch over several contexts (one per GPU of the node) in one call: mi355_lzss_compress_batch_multi"""
    flat, off = _csr(blobs)
    nb = len(blobs)
    caps = [HeaderSize + (37 * len(b) + 7) // 8 + 1 for b in blobs]
    out_off = np.zeros(nb + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
    out_len = np.zeros(max(nb, 1), dtype=np.uint32)
    status = np.zeros(max(nb, 1), dtype=np.int32)
    _check(_lib.lib().mi355_lzss_compress_batch_multi(_ctx_array(ctxs), len(ctxs), _ptr(flat) if flat.size else None, _ptr(off),
                                                      nb, _ptr(out), _ptr(out_off), _ptr(out_len), _ptr(status)))
    res = []
    for k in range(nb):
        if status[k] != 0:
            raise LzssError(f"blob {k}: status {status[k]}", int(status[k]))
        o = int(out_off[k])
        res.append(out[o:o + int(out_len[k])].tobytes())
    return res


def decompress_batch_multi(ctxs, blobs, max_out_len):
    """decompress_batch over several contexts in one call: mi355_lzss_decompress_batch_multi"""
    flat, off = _csr(blobs)
    nb = len(blobs)
    caps = [max_out_len] * nb if np.isscalar(max_out_len) else list(max_out_len)
    out_off = np.zeros(nb + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
    out_len = np.zeros(max(nb, 1), dtype=np.uint32)
    status = np.zeros(max(nb, 1), dtype=np.int32)
    _check(_lib.lib().mi355_lzss_decompress_batch_multi(_ctx_array(ctxs), len(ctxs), _ptr(flat) if flat.size else None,
                                                        _ptr(off), nb, _ptr(out), _ptr(out_off), _ptr(out_len), _ptr(status)))
    res = []
    for k in range(nb):
        if status[k] != 0:
            res.append(LzssError(f"blob {k}: status {status[k]}", int(status[k])))
        else:
            o = int(out_off[k])
            res.append(out[o:o + int(out_len[k])].tobytes())
    return res


def compress_batch(ctx_or_compressor, blobs):
    """One-shot Compress (compress.go:375-379) of independent blobs that share a dictionary."""
    ctx = ctx_or_compressor._ctx if isinstance(ctx_or_compressor, Compressor) else ctx_or_compressor
    flat, off = _csr(blobs)
    nb = len(blobs)
    caps = [HeaderSize + (37 * len(b) + 7) // 8 + 1 for b in blobs]
    out_off = np.zeros(nb + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
    out_len = np.zeros(max(nb, 1), dtype=np.uint32)
    status = np.zeros(max(nb, 1), dtype=np.int32)
    _check(_lib.lib().mi355_lzss_compress_batch(ctx.h, _ptr(flat) if flat.size else None, _ptr(off), nb, _ptr(out),
                                                _ptr(out_off), _ptr(out_len), _ptr(status)))
    res = []
    for k in range(nb):
        if status[k] != 0:
            raise LzssError(f"blob {k}: status {status[k]}", int(status[k]))
        o = int(out_off[k])
        res.append(out[o:o + int(out_len[k])].tobytes())
    return res


def decompress_batch(ctx_or_compressor, blobs, max_out_len):
    """Decompress (decompress.go:17-84) of independent streams that share a dictionary; returns a list of
    bytes (or LzssError instances for streams the device rejected)."""
    ctx = ctx_or_compressor._ctx if isinstance(ctx_or_compressor, Compressor) else ctx_or_compressor
    flat, off = _csr(blobs)
    nb = len(blobs)
    caps = [max_out_len] * nb if np.isscalar(max_out_len) else list(max_out_len)
    out_off = np.zeros(nb + 1, dtype=np.uint64)
    np.cumsum(caps, out=out_off[1:])
    out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
    out_len = np.zeros(max(nb, 1), dtype=np.uint32)
    status = np.zeros(max(nb, 1), dtype=np.int32)
    _check(_lib.lib().mi355_lzss_decompress_batch(ctx.h, _ptr(flat) if flat.size else None, _ptr(off), nb, _ptr(out),
                                                  _ptr(out_off), _ptr(out_len), _ptr(status)))
    res = []
    for k in range(nb):
        if status[k] != 0:
            res.append(LzssError(f"blob {k}: status {status[k]}", int(status[k])))
        else:
            o = int(out_off[k])
            res.append(out[o:o + int(out_len[k])].tobytes())
    return res


def Decompress(data: bytes, dict_: bytes = b"", device: int = 0) -> bytes:
    """lzss/decompress.go:17-84 (header handling on the host, phrase decoding on the GPU)."""
    data = bytes(data)
    try:
        h = Header.from_bytes(data)
    except LzssError as e:
        raise LzssError(f"failed to read header: {e}")
    if h.Version != Version:
        raise LzssError("unsupported compressor version")
    if h.NoCompression:
        return data[HeaderSize:]
    cap = max(16, (len(data) - HeaderSize) * 8 // 30 * 256 + 512)
    while True:
        ent = _decoder_entry(bytes(dict_ or b""), device)
        with ent.lock:
            if ent.ctx is None:   # evicted between the lookup and the lock: fetch (or rebuild) it again
                continue
            r = decompress_batch(ent.ctx, [data], min(cap, 1 << 28))[0]
        break
    if isinstance(r, LzssError):
        if r.code == _lib.E_FORMAT:
            msg = decode_error_text(data, len(AugmentDict(bytes(dict_ or b""))))
            if msg:
                raise LzssError(msg, _lib.E_FORMAT)
        raise r
    return r


def _backref_repr(address: int, length: int, delimiter: int, nb_addr: int, dict_len: int) -> str:
    """Go's %+v of the reference's `backref` (lzss/backref.go:21-30,55-59)"""
    nb = 8 + nb_addr + 8
    return (f"{{address:{address} length:{length} bType:{{Delimiter:{delimiter} NbBitsAddress:{nb_addr} NbBitsLength:8 "
            f"NbBitsBackRef:{nb} nbBytesBackRef:{(nb + 7) // 8} maxAddress:{1 << nb_addr} maxLength:256 DictLen:{dict_len}}}}}")


def decode_error_text(data: bytes, dict_len: int):
    """The reference decoder's message for a stream the device rejected as malformed (decompress.go:53-55,66-69).  The
    device reports MI355_LZSS_E_FORMAT per blob; the text needs the offending phrase, which a walk over the phrase
    boundaries finds on the host (no bytes are decoded).  A stream that ends inside a phrase gives "EOF": readFrom returns
    the bit reader's TryError (backref.go:77-79), and bitio hands on the io.EOF of the bytes.Reader under it.  None for a
    stream without an offending phrase.  dict_len: length of the AUGMENTED dictionary."""
    total = len(data) * 8
    b = HeaderSize * 8
    out_len = 0

    def bits(pos, nb):
        v = 0
        for k in range(nb):
            q = pos + k
            v = (v << 1) | ((data[q >> 3] >> (7 - (q & 7))) & 1)
        return v

    while b + 8 <= total:
        s = bits(b, 8)
        if s == 0xFE or s == 0xFF:
            nb_addr = 14 if s == 0xFE else 21
            if b + 16 + nb_addr > total:
                return "EOF"
            length = bits(b + 8, 8) + 1
            address = bits(b + 16, nb_addr) + 1
            if s == 0xFE:
                if address > out_len:
                    return (f"invalid short backref {_backref_repr(address, length, 0xFE, 14, 0)} - output buffer is only "
                            f"{out_len} bytes long")
            elif address > out_len:
                dict_start = dict_len - (address - out_len)
                if dict_start < 0 or dict_start > dict_len or dict_start + length > dict_len:
                    return (f"invalid dynamic backref {_backref_repr(address, length, 0xFF, 21, dict_len)} - dict is only "
                            f"{dict_len} bytes long; dictStart = {dict_start}")
            out_len += length
            b += 16 + nb_addr
        else:
            out_len += 1
            b += 8
    return None


# Decompress builds no index in the reference (decompress.go:34 only augments the dictionary): a decode-only
# context per (dictionary, device), kept in a small LRU so that a caller that decompresses in a loop pays the
# dictionary upload once.  A context is not thread-safe (its batch scratch is shared) and the reference's Decompress
# is a pure function, so every cache entry carries a lock that Decompress holds across its device call; the cache
# itself is guarded by _DECODERS_LOCK, and an evicted context is closed only once its lock is free.
_DECODERS = {}
_DECODERS_MAX = 4
_DECODERS_LOCK = threading.Lock()


class _Decoder:
    def __init__(self, ctx):
        self.ctx = ctx
        self.lock = threading.Lock()


def _decoder_entry(dict_: bytes, device: int) -> _Decoder:
    import hashlib
    key = (hashlib.sha256(dict_).digest(), len(dict_), device)
    evicted = []
    with _DECODERS_LOCK:
        ent = _DECODERS.pop(key, None)
        if ent is None:
            ent = _Decoder(_Ctx(dict_, device, decoder=True))
            while len(_DECODERS) >= _DECODERS_MAX:
                evicted.append(_DECODERS.pop(next(iter(_DECODERS))))
        _DECODERS[key] = ent   # most recently used last
    for old in evicted:        # outside the cache lock: waits for a Decompress that still uses the context
        with old.lock:
            old.ctx.close()
            old.ctx = None
    return ent
