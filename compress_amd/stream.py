"""Speculative packing of a stream into size-capped blobs.

The reference's callers fill a blob with the loop of lzss/compress_test.go:337-353: Write chunk after chunk while
Len() < outMax, Revert the chunk that overshot.  Every Write re-indexes the whole inBuf (compress.go:113-116), so the
loop is one short, latency-bound GPU job per chunk.  But the bits a chunk adds depend only on the blob's bytes before
and inside it, so the Writes of MANY candidate chunk counts are independent jobs: chunk j of a blob that starts at i0
is "inBuf = data[i0 : i0 + j*chunk], start = (j-1)*chunk" (the `starts` of mi355_lzss_batch_upload).  `pack_stream`
runs `lookahead` of them per batch - the same total work as the sequential loop, as batch work - and replays the
loop's decisions on the bit counts.  The blobs are byte-identical to the sequential loop's (tests/test_gpu_stream.py).
"""
import ctypes as C

import numpy as np

from . import _lib
from .batch import Batch
from .lzss import Compressor, Header, HeaderSize, LzssError, Version


def _sequential_one(compressor: Compressor, data: bytes, i0: int, chunk: int, out_max: int):
    """one turn of the reference loop (compress_test.go:337-361): returns (blob bytes, chunks consumed)"""
    compressor.Reset()
    i = i0
    while i < len(data) and compressor.Len() < out_max:
        compressor.Write(data[i:min(i + chunk, len(data))])
        uncompressed = i + chunk - i0 + 3
        if compressor.Len() >= out_max and uncompressed <= out_max and compressor.Len() > uncompressed:
            compressor.ConsiderBypassing()
        i += chunk
    if compressor.Len() > out_max:
        compressor.Revert()
        i -= chunk
    return compressor.Bytes(), (i - i0) // chunk


def pack_stream_sequential(compressor: Compressor, data: bytes, chunk: int, out_max: int):
    """The reference loop on any Compressor-like object: [(blob bytes, input bytes consumed)]."""
    data = bytes(data)
    res, i0 = [], 0
    while i0 < len(data):
        blob, k = _sequential_one(compressor, data, i0, chunk, out_max)
        if k <= 0:
            raise ValueError("a single chunk does not fit the output cap")
        res.append((blob, min(i0 + k * chunk, len(data)) - i0))
        i0 += k * chunk
    return res


class _Batches:
    """one growing device workspace for all lookahead batches of a stream"""

    def __init__(self, ctx):
        self.ctx, self.batch, self.cap_blobs, self.cap_bytes = ctx, None, 0, 0

    def get(self, n_blobs: int, n_bytes: int) -> Batch:
        if self.batch is None or n_blobs > self.cap_blobs or n_bytes > self.cap_bytes:
            self.close()
            self.cap_blobs, self.cap_bytes = max(n_blobs, self.cap_blobs), max(n_bytes, self.cap_bytes, 1)
            self.batch = Batch(self.ctx, self.cap_blobs, self.cap_bytes)
        return self.batch

    def close(self):
        if self.batch is not None:
            self.batch.close()
            self.batch = None


def _speculative_one(ws: _Batches, data: bytes, i0: int, chunk: int, out_max: int, lookahead: int):
    """(blob bytes, chunks consumed) of the blob that starts at i0, or None if ConsiderBypassing would fire"""
    bits, total, j = [], 0, 0   # accepted phrase bits / their count / accepted chunks

    def finish(k):
        allbits = np.concatenate(bits) if bits else np.zeros(0, np.uint8)
        return Header(Version, False).to_bytes() + np.packbits(allbits).tobytes(), k

    while True:
        # candidate Writes j+1 .. j+look (the loop starts a Write only while bytes are left).  After the first batch
        # the blob's own bits per chunk predict where the cap falls: the next batch reaches just past that point
        look = lookahead
        if j and total:
            left = (out_max - HeaderSize) * 8 - total
            look = max(4, min(4 * lookahead, int(left / (total / j) * 1.04) + 3))
        segs, starts = [], []
        for t in range(look):
            s = i0 + (j + t) * chunk
            if s >= len(data):
                break
            segs.append(data[i0:min(s + chunk, len(data))])
            starts.append(s - i0)
        if not segs:
            return finish(j)
        b = ws.get(len(segs), sum(map(len, segs)))
        b.upload(segs, starts)
        b.compress(with_header=False)
        _, nb, _ = b.sizes()
        streams = None
        for t in range(len(segs)):
            new_total = total + int(nb[t])
            new_len = HeaderSize + (new_total + 7) // 8
            uncompressed = (j + t + 1) * chunk + 3
            if new_len >= out_max and uncompressed <= out_max and new_len > uncompressed:
                return None             # the loop would call ConsiderBypassing here
            if streams is None:
                streams = b.download()
            if new_len > out_max:
                # Revert: the blob ends before this chunk.  outBuf.Truncate(lastOutLen) (compress.go:341) keeps the
                # re-opened last byte as the reverted Write left it, so the pad bits of the blob are the first bits
                # of the reverted chunk's phrases, not zeros
                pad = min((-total) % 8, int(nb[t]))
                if pad:
                    bits.append(np.unpackbits(np.frombuffer(streams[t][:1], dtype=np.uint8))[:pad])
                return finish(j + t)
            if nb[t]:
                bits.append(np.unpackbits(np.frombuffer(streams[t], dtype=np.uint8))[:int(nb[t])])
            total = new_total
            if new_len >= out_max:
                return finish(j + t + 1)
        j += len(segs)


def pack_stream(compressor: Compressor, data: bytes, chunk: int, out_max: int, lookahead: int = 64):
    """Same result as pack_stream_sequential, `lookahead` Writes per GPU batch.  `compressor` supplies the context
    (dictionary index) and runs the rare blobs that need ConsiderBypassing."""
    data = bytes(data)
    ws = _Batches(compressor._ctx)
    res, i0 = [], 0
    try:
        while i0 < len(data):
            r = _speculative_one(ws, data, i0, chunk, out_max, lookahead)
            if r is None:
                r = _sequential_one(compressor, data, i0, chunk, out_max)
            blob, k = r
            if k <= 0:
                raise ValueError("a single chunk does not fit the output cap")
            res.append((blob, min(i0 + k * chunk, len(data)) - i0))
            i0 += k * chunk
    finally:
        ws.close()
        compressor.Reset()
    return res


def pack_stream_native(compressor: Compressor, data: bytes, chunk: int, out_max: int, lookahead: int = 64):
    """pack_stream through the C entry point mi355_lzss_pack_blob (the path of a Go / C++ caller): one call per blob,
    the rare blobs in which ConsiderBypassing would fire run the loop on `compressor`."""
    data = bytes(data)
    arr = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
    out = np.zeros(out_max + 16, dtype=np.uint8)
    out_len, chunks, loop = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    lib = _lib.lib()
    res, i0 = [], 0
    try:
        while i0 < len(data):
            rc = lib.mi355_lzss_pack_blob(compressor._ctx.h, arr[i0:].ctypes.data, len(data) - i0, chunk, out_max, lookahead,
                                          out.ctypes.data, out.size, C.byref(out_len), C.byref(chunks), C.byref(loop))
            if rc != 0:
                raise LzssError(_lib.last_error(), rc)
            if loop.value:
                blob, k = _sequential_one(compressor, data, i0, chunk, out_max)
            else:
                blob, k = out[:out_len.value].tobytes(), int(chunks.value)
            if k <= 0:
                raise ValueError("a single chunk does not fit the output cap")
            res.append((blob, min(i0 + k * chunk, len(data)) - i0))
            i0 += k * chunk
    finally:
        compressor.Reset()
    return res
