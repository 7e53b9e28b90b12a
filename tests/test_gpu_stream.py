"""GPU parity of the speculative stream packer with the reference's Write / Revert loop
(lzss/compress_test.go:317-363), run on the GPU compressor AND on the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


class OracleCompressor:
    """the oracle behind the Compressor method names the loop uses"""

    def __init__(self, dict_):
        self.c = O.Compressor(dict_)

    def Reset(self): self.c.reset()
    def Write(self, d): self.c.write(d)
    def Len(self): return self.c.len()
    def Bytes(self): return self.c.bytes()
    def Revert(self): self.c.revert()
    def ConsiderBypassing(self): return self.c.consider_bypassing()


@pytest.mark.parametrize("chunk,out_max,lookahead", [(1000, 5000, 8), (1000, 5000, 64), (700, 3001, 5), (4096, 20000, 16)])
def test_pack_stream_equals_the_reference_loop(dict_naive, average_block, chunk, out_max, lookahead):
    import compress_amd
    from compress_amd import stream
    data = average_block[:150000]
    want = stream.pack_stream_sequential(OracleCompressor(dict_naive), data, chunk, out_max)
    comp = compress_amd.Compressor(dict_naive)
    got = stream.pack_stream(comp, data, chunk, out_max, lookahead)
    assert [n for _, n in got] == [n for _, n in want]
    assert [b for b, _ in got] == [b for b, _ in want]
    assert sum(n for _, n in got) == len(data)
    assert stream.pack_stream_native(comp, data, chunk, out_max, lookahead) == want  # the same through the C ABI
    off = 0
    for blob, n in got[:3]:  # the blobs decode to consecutive pieces of the stream (compress_test.go:358)
        assert compress_amd.Decompress(blob, dict_naive) == data[off:off + n]
        off += n


def test_pack_stream_with_incompressible_stretches(dict_naive):
    """random bytes make ConsiderBypassing fire: those blobs fall back to the sequential loop"""
    import compress_amd
    from compress_amd import stream, synth
    rng = np.random.default_rng(3)
    data = synth.blob(5)[:30000] + bytes(rng.integers(0, 254, 12000, dtype=np.uint8)) + synth.blob(6)[:20000]
    want = stream.pack_stream_sequential(OracleCompressor(dict_naive), data, 1000, 5000)
    got = stream.pack_stream(compress_amd.Compressor(dict_naive), data, 1000, 5000, 16)
    assert got == want
    assert stream.pack_stream_native(compress_amd.Compressor(dict_naive), data, 1000, 5000, 16) == want


def test_pack_stream_at_config4_cap(dict_naive):
    """BASELINE config 4's real cap: 1000-byte Writes into blobs of <= 126 976 B (4096 x 31, the payload of one
    EIP-4844 blob).  One such blob takes ~430 KB of calldata-like input, i.e. ~430 Writes that each re-index the whole
    inBuf - the oracle loop needs a while for it, so one full blob and the start of the next are checked."""
    import compress_amd
    from compress_amd import stream, synth
    data = b"".join(synth.blob(400 + k) for k in range(4))[:470000]
    want = stream.pack_stream_sequential(OracleCompressor(dict_naive), data, 1000, 126976)
    got = stream.pack_stream(compress_amd.Compressor(dict_naive), data, 1000, 126976)
    assert [n for _, n in got] == [n for _, n in want]
    assert got == want
    assert len(got) == 2 and 126976 - 1000 <= len(got[0][0]) <= 126976   # one more 1000-byte chunk would not have fitted
    assert stream.pack_stream_native(compress_amd.Compressor(dict_naive), data, 1000, 126976) == want
