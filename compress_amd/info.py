"""CompressedStreamInfo / ToCSV (lzss/decompress.go:86-236): the phrases of a compressed stream.

The bytes come from the GPU decompressor (mi355_lzss_decompress_batch); the phrase boundaries are a walk over the
bit stream on the host (three field widths, nothing to index), done with the reference's conventions: positions
count the augmented dictionary in front of the output, consecutive literals form one phrase, StartCompressed counts
bits after the header.
"""
from dataclasses import dataclass

from .lzss import AugmentDict, Decompress, Header, HeaderSize, LzssError, SymbolDynamic, SymbolShort, Version


@dataclass
class CompressionPhrase:          # decompress.go:86-93
    Type: int                     # 0 literal run, 0xFE short, 0xFF long
    Length: int
    ReferenceAddress: int
    StartDecompressed: int
    StartCompressed: int
    Content: bytes


class _Bits:
    """MSB-first reader with bitio's try-semantics: a read past the end sets `eof` and returns 0"""

    def __init__(self, data: bytes, pos_bits: int):
        self.v = int.from_bytes(data, "big")
        self.n = len(data) * 8
        self.pos = pos_bits
        self.eof = False

    def read(self, nb: int) -> int:
        if self.eof or self.pos + nb > self.n:
            self.eof = True
            return 0
        r = (self.v >> (self.n - self.pos - nb)) & ((1 << nb) - 1)
        self.pos += nb
        return r


def CompressedStreamInfo(c: bytes, dict_: bytes = b"", device: int = 0):
    """decompress.go:97-206"""
    c = bytes(c)
    try:
        h = Header.from_bytes(c)
    except LzssError as e:
        raise LzssError(f"failed to read header: {e}")
    if h.Version != Version:
        raise LzssError("unsupported compressor version")
    if h.NoCompression:           # decompress.go:109-118
        return [CompressionPhrase(0, len(c) - HeaderSize, 0, 0, 0, c[HeaderSize:])]
    aug = AugmentDict(dict_ or b"")
    out = aug + Decompress(c, dict_, device)   # the reference's `out` buffer starts with the dictionary
    res = []
    r = _Bits(c, HeaderSize * 8)
    pos = len(aug)                # out.Len() as the walk proceeds
    in_i = 0
    lit_start = -1

    def flush_literals():
        nonlocal lit_start, in_i
        if lit_start == -1:
            return
        res.append(CompressionPhrase(0, pos - lit_start, lit_start, lit_start, in_i, out[lit_start:pos]))
        in_i += (pos - lit_start) * 8
        lit_start = -1

    s = r.read(8)
    while not r.eof:
        if s in (SymbolShort, SymbolDynamic):
            flush_literals()
            abits = 14 if s == SymbolShort else 21   # backref.go:18,38
            length = r.read(8) + 1                   # backref.readFrom, backref.go:70-86
            address = r.read(abits) + 1
            if r.eof:
                raise LzssError("EOF")
            if address > pos:
                raise LzssError(f"invalid back reference: address {address} at {pos}")
            pos += length
            res.append(CompressionPhrase(s, length, pos - length - address, pos - length, in_i, out[pos - length:pos]))
            in_i += 8 + 8 + abits
        else:
            if lit_start == -1:
                lit_start = pos
            pos += 1
        s = r.read(8)
    flush_literals()
    if pos != len(out):
        raise LzssError("phrase walk and decompressor disagree on the output length")
    return res


def ToCSV(phrases) -> bytes:
    """decompress.go:208-236"""
    lines = ["type,length,start_decompressed (bytes),start_compressed (bits),reference_address,content (hex)\n"]
    for p in phrases:
        if p.Type == SymbolShort:
            t = "short"
        elif p.Type == SymbolDynamic:
            t = "long"
        elif p.Type == 0:
            t = "literal"
        else:
            raise ValueError("unknown phrase type")
        lines.append(f"{t},{p.Length},{p.StartDecompressed},{p.StartCompressed},{p.ReferenceAddress},{p.Content.hex()}\n")
    return "".join(lines).encode()
