"""Device-resident batches (mi355_lzss_batch_*): upload / run / download split so that callers such as
bench.py can time the device work alone with inputs and outputs resident in HBM."""
import ctypes as C

import numpy as np

from . import _lib
from .lzss import Compressor, LzssError, _Ctx, _check, _csr, _ptr, HeaderSize


class Batch:
    def __init__(self, ctx, max_blobs: int, max_total_bytes: int):
        self._ctx = ctx._ctx if isinstance(ctx, Compressor) else ctx
        h = C.c_void_p()
        _check(_lib.lib().mi355_lzss_batch_create(self._ctx.h, max_blobs, max_total_bytes, C.byref(h)))
        self.h = h
        self.n_blobs = 0
        self.lens = []

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().mi355_lzss_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- compression
    def upload(self, blobs, starts=None):
        flat, off = _csr(blobs)
        st = None
        if starts is not None:
            st = np.ascontiguousarray(starts, dtype=np.uint32)
        _check(_lib.lib().mi355_lzss_batch_upload(self.h, _ptr(flat) if flat.size else None, _ptr(off), len(blobs),
                                                  _ptr(st) if st is not None else None))
        self.n_blobs = len(blobs)
        self.lens = [len(b) for b in blobs]

    def upload_flat(self, flat: np.ndarray, off: np.ndarray):
        """upload() for blobs that already lie back to back: `flat` uint8, `off` uint64 CSR offsets (no host copies)"""
        n = off.size - 1
        _check(_lib.lib().mi355_lzss_batch_upload(self.h, _ptr(flat) if flat.size else None, _ptr(off), n, None))
        self.n_blobs = n
        self.lens = [int(x) for x in np.diff(off.astype(np.int64))]

    def compress(self, with_header: bool = True):
        _check(_lib.lib().mi355_lzss_batch_compress(self.h, 1 if with_header else 0))

    def sizes(self):
        n = max(self.n_blobs, 1)
        out_len = np.zeros(n, dtype=np.uint32)
        nbits = np.zeros(n, dtype=np.uint64)
        total = C.c_uint64()
        _check(_lib.lib().mi355_lzss_batch_sizes(self.h, _ptr(out_len), _ptr(nbits), C.byref(total)))
        return out_len[:self.n_blobs], nbits[:self.n_blobs], total.value

    def download(self):
        out_len, _, _ = self.sizes()
        out_off = np.zeros(self.n_blobs + 1, dtype=np.uint64)
        np.cumsum(out_len, out=out_off[1:])
        out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
        status = np.zeros(max(self.n_blobs, 1), dtype=np.int32)
        _check(_lib.lib().mi355_lzss_batch_download(self.h, _ptr(out), _ptr(out_off), _ptr(status)))
        res = []
        for k in range(self.n_blobs):
            if status[k] != 0:
                raise LzssError(f"blob {k}: status {status[k]}", int(status[k]))
            res.append(out[int(out_off[k]):int(out_off[k + 1])].tobytes())
        return res

    def results(self, blob: int):
        """RAW packed per-position results of blob: type<<30 | pending<<29 | (len-1)<<21 | field.  After
        compress() the field of a pending entry is a suffix rank and that of a type-3 entry a dictionary rank
        (include/mi355_lzss.h); Compressor.best_backrefs gives the reference's (type, address, length)."""
        n = self.lens[blob]
        packed = np.zeros(max(n, 1), dtype=np.uint32)
        _check(_lib.lib().mi355_lzss_batch_results(self.h, blob, _ptr(packed), n))
        return packed[:n]

    def index(self, blob: int):
        """(SA, ISA, LCP) of blob in the N-world (n+1 entries, rank 0 = end sentinel)"""
        N = self.lens[blob] + 1
        sa = np.zeros(N, dtype=np.int32)
        isa = np.zeros(N, dtype=np.int32)
        lcp = np.zeros(N, dtype=np.uint16)
        _check(_lib.lib().mi355_lzss_batch_index(self.h, blob, _ptr(sa), _ptr(isa), _ptr(lcp), N))
        return sa, isa, lcp

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = _lib.lib().mi355_lzss_batch_stage_times(self.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    # ---- decompression
    def upload_compressed(self, blobs, out_caps):
        flat, off = _csr(blobs)
        caps = [out_caps] * len(blobs) if np.isscalar(out_caps) else list(out_caps)
        cap_off = np.zeros(len(blobs) + 1, dtype=np.uint64)
        np.cumsum(caps, out=cap_off[1:])
        _check(_lib.lib().mi355_lzss_batch_upload_compressed(self.h, _ptr(flat) if flat.size else None, _ptr(off),
                                                             len(blobs), _ptr(cap_off)))
        self.n_dblobs = len(blobs)
        self._cap_off = cap_off

    def decompress(self):
        _check(_lib.lib().mi355_lzss_batch_decompress(self.h))

    def download_decompressed(self):
        nb = self.n_dblobs
        out = np.zeros(max(int(self._cap_off[-1]), 1), dtype=np.uint8)
        out_len = np.zeros(max(nb, 1), dtype=np.uint32)
        status = np.zeros(max(nb, 1), dtype=np.int32)
        _check(_lib.lib().mi355_lzss_batch_download_decompressed(self.h, _ptr(out), _ptr(self._cap_off), _ptr(out_len),
                                                                 _ptr(status)))
        res = []
        for k in range(nb):
            if status[k] != 0:
                res.append(LzssError(f"blob {k}: status {status[k]}", int(status[k])))
            else:
                o = int(self._cap_off[k])
                res.append(out[o:o + int(out_len[k])].tobytes())
        return res


def new_ctx(dict_: bytes = b"", device: int = 0) -> _Ctx:
    return _Ctx(dict_, device)
