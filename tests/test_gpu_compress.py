"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Bar: bit-exact (integer / byte / index work)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import fixture_bytes

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def amd():
    import compress_amd
    from compress_amd import batch, synth
    compress_amd.batch = batch
    compress_amd.synth = synth
    return compress_amd


def rand_cases():
    rng = np.random.default_rng(42)
    cases = [b"", b"a", b"hi", bytes(8), bytes(300), b"hihihi", bytes([0xFD, 0xFE, 0xFF]), bytes([1, 1, 1, 1, 2, 1, 1, 1, 1]),
             b"\x01" + bytes(8), b"hi" + bytes(300), bytes([0xFF]) * 400, bytes([0xFE]) * 170 + b"x" + bytes([0xFE]) * 200,
             b"abc" * 500, bytes(1000) + b"\x07" + bytes(700)]
    for k in (1, 2, 3, 4, 256):
        for n in (5, 63, 64, 65, 257, 1000, 5000):
            cases.append(bytes(rng.integers(0, k, n, dtype=np.uint8)))
    cases.append(bytes(rng.choice(np.array([0, 0, 0, 0xFE, 0xFF, 9], dtype=np.uint8), 3000)))
    return cases


def test_index_matches_oracle(amd):
    """suffix array / inverse / lcp of every blob of a ragged batch (suffixarray.go:32-45)"""
    cases = rand_cases() + [fixture_bytes("3c2943_data.bin"), amd.synth.blob(0)[:40000]]
    ctx = amd.batch.new_ctx(b"")
    b = amd.batch.Batch(ctx, len(cases), sum(map(len, cases)))
    b.upload(cases)
    b.compress()
    for k, d in enumerate(cases):
        n = len(d)
        sa, isa, lcp = b.index(k)
        ref = O.suffix_array(d)
        assert sa[0] == n
        assert np.array_equal(sa[1:], ref), f"case {k} n={n}"
        assert np.array_equal(isa[sa], np.arange(n + 1))
        # lcp against a direct computation on a sample of ranks
        rng = np.random.default_rng(k)
        for r in rng.integers(1, n + 1, min(n, 200)) if n else []:
            a, c = int(sa[r - 1]), int(sa[r])
            l = 0
            while l < 256 and a + l < n and c + l < n and d[a + l] == d[c + l]:
                l += 1
            assert lcp[r] == l


def run_cases():
    """inputs for the run-compressed suffix sort (sa_meta_kernel): runs whose lengths sit around the chunk length of 253
    and its multiples, long runs followed by smaller / larger bytes / the end of the text, more than 64 long runs of one
    byte (xp_long's list), sections with more than 1024 runs (expanded by all waves) and with many levels"""
    rng = np.random.default_rng(253)
    cases = [bytes(n) for n in (252, 253, 254, 255, 505, 506, 507, 508, 759, 760, 1012, 1013, 20000)]
    cases += [bytes(253) + b"\x01", bytes(254) + b"\x01" + bytes(506), b"\x09" * 300 + b"\x01" + b"\x09" * 301 + b"\xf0" + b"\x09" * 253,
              (bytes(254) + b"\x01") * 80, (b"\x05" * 300 + b"\x01" + b"\x05" * 280 + b"\x09") * 40,
              b"\xff" * 1000 + bytes(1000) + b"\xff" * 1001 + bytes(999), b"ab" * 3000, b"a" * 253 + b"b" * 253 + b"a" * 254 + b"b" * 252]
    t = bytearray()
    while len(t) < 60000:        # thousands of short runs of few bytes: big sections, many levels
        t += bytes([int(rng.integers(0, 3))]) * int(rng.integers(1, 40))
    cases.append(bytes(t))
    t = bytearray()
    while len(t) < 90000:        # long runs of every length between random bytes
        t += bytes([int(rng.integers(0, 2)) * 255]) * int(rng.integers(200, 330)) + bytes(rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8))
    cases.append(bytes(t))
    return cases


def test_run_compressed_sort_matches_oracle(amd):
    cases = run_cases()
    ctx = amd.batch.new_ctx(b"")
    b = amd.batch.Batch(ctx, len(cases), sum(map(len, cases)))
    b.upload(cases)
    b.compress()
    outs = b.download()
    oc = O.Compressor(b"")
    for k, d in enumerate(cases):
        sa, isa, lcp = b.index(k)
        ref = O.suffix_array(d)
        assert sa[0] == len(d)
        assert np.array_equal(sa[1:], ref), f"case {k} n={len(d)}"
        assert np.array_equal(isa[sa], np.arange(len(d) + 1))
        assert outs[k] == oc.compress(d), f"case {k}"


def sparse_window_cases():
    """short-window references whose smallest-rank occurrence lies behind many ranks outside the window: an 8-byte word
    followed by 0x00 at `far` places outside the window (the smallest ranks of the word's interval), by 0x01 inside the
    window, by 0x02 where it is compressed - the searches through the window index (window_search.h: leftmost), with
    intervals on both sides of its linear-scan threshold, and with windows that start at the text's start"""
    rng = np.random.default_rng(1614)

    def filler(n):
        return bytes(rng.integers(3, 250, n, dtype=np.uint8))

    def word():
        return bytes(rng.integers(3, 250, 8, dtype=np.uint8))

    cases = []
    # the window starts behind the far copies; 20 .. 100 of them: found by the scan, 130 .. 400: the index is needed
    t = bytearray(filler(100000))
    words = [(word(), far) for far in (20, 27, 28, 29, 56, 57, 100, 111, 112, 113, 130, 400)]
    pos = 100
    for w, far in words:
        for _ in range(far):
            t[pos:pos + 9] = w + b"\x00"
            pos += 80
    assert pos < 99000
    t += filler(30000)
    for k, (w, far) in enumerate(words):
        t[125000 + 40 * k:125000 + 40 * k + 9] = w + b"\x01"
    for k, (w, far) in enumerate(words):
        t += filler(30) + w + b"\x02"
    cases.append(bytes(t) + filler(50))
    # the window starts at the text's start: the copies outside it lie BEHIND the compressed position
    t = bytearray(filler(1000))
    words = [(word(), far) for far in (10, 40, 200)]
    for w, far in words:
        t += filler(20) + w + b"\x01"
    t += filler(1500)
    for w, far in words:
        t += filler(20) + w + b"\x02"
    t += filler(500)
    for w, far in words:
        for _ in range(far):
            t += filler(50) + w + b"\x00"
    cases.append(bytes(t))
    return cases


def test_short_window_searches_with_sparse_hits(amd):
    cases = sparse_window_cases()
    ctx = amd.batch.new_ctx(b"")
    b = amd.batch.Batch(ctx, len(cases), sum(map(len, cases)))
    oc = O.Compressor(b"")
    want = [oc.compress(d) for d in cases]
    for order in (cases, cases[::-1]):
        b.upload(order)
        b.compress()
        outs = b.download()
        exp = want if order is cases else want[::-1]
        for k in range(len(cases)):
            assert outs[k] == exp[k], f"case {k}"


def small_group_cases():
    """inputs that stress the suffix sort's small-group finisher (sa_finish) and the LDS / global LCP kernels"""
    rng = np.random.default_rng(11)
    x = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    words = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(400)]
    reps = b"".join(words[int(k)] for k in rng.integers(0, 400, 1500))          # 32-byte words, 2-6 copies each
    near = b"".join(w[:31] + bytes([int(rng.integers(0, 256))]) for w in words[:300] for _ in range(3))
    return [
        x + x,                              # every suffix has one twin, prefixes far longer than the compare span
        x[:5000] + x[:5000] + x[:5000],     # triples
        reps,                               # groups of 2-6 that part at depth <= 32 .. 64
        near,                               # groups that differ in the last byte of a word
        reps[:4000] + reps[:4000][:-1],     # a suffix that is a prefix of another one (ends the text)
        b"ab" * 3000 + b"a",                # one chain of nested suffixes
        b"abcabd" * 700,                    # period 6 with two-way splits
        bytes(200) + x[:300] + bytes(200) + x[:300] + bytes(150),  # runs of zeros around repeats
        amd_blob_sized(149990), amd_blob_sized(150000), amd_blob_sized(150001),  # the LCP kernels' size threshold
    ]


def amd_blob_sized(n):
    from compress_amd import synth
    return (synth.blob(90) + synth.blob(91))[:n]


def test_small_groups_and_lcp_variants_match_oracle(amd, dict_naive):
    cases = small_group_cases()
    oc = O.Compressor(dict_naive)
    for group in (cases[:8], cases[8:9], cases[9:10], cases[10:]):  # the size threshold is per batch (largest text)
        ctx = amd.batch.new_ctx(dict_naive)
        b = amd.batch.Batch(ctx, len(group), sum(map(len, group)))
        b.upload(group)
        b.compress()
        got = b.download()
        for k, d in enumerate(group):
            n = len(d)
            sa, isa, lcp = b.index(k)
            assert sa[0] == n
            assert np.array_equal(sa[1:], O.suffix_array(d)), f"case {k} n={n}"
            assert np.array_equal(isa[sa], np.arange(n + 1))
            rng = np.random.default_rng(k)
            for r in rng.integers(1, n + 1, 300):
                a, c = int(sa[r - 1]), int(sa[r])
                l = 0
                while l < 256 and a + l < n and c + l < n and d[a + l] == d[c + l]:
                    l += 1
                assert lcp[r] == l, f"case {k} rank {r}"
            assert got[k] == oc.compress(d), f"case {k} n={n}"


def test_text_sizes_across_the_suffix_sort_variants(amd, dict_naive):
    """the suffix sort has three code paths by text size: 9-bit digits (N <= 2^18), 10-bit digits (N <= 2^19) and the
    device-wide sort of sa_large.hip above; the one-workgroup kernels are picked by a batch's LARGEST text and the
    large texts go one by one, so each path is exercised with a ragged batch (small texts ride along) and checked on
    index and bytes"""
    big = b"".join(amd.synth.blob(k) for k in range(20, 25))  # 640 KiB of calldata-like bytes
    for n_big in (262143, 262144, 280000, 524287, 524288):
        cases = [big[:n_big], b"", b"x", amd.synth.blob(3)[:5000], bytes(70000)]
        ctx = amd.batch.new_ctx(dict_naive)
        b = amd.batch.Batch(ctx, len(cases), sum(map(len, cases)))
        b.upload(cases)
        b.compress()
        got = b.download()
        oc = O.Compressor(dict_naive)
        for k, d in enumerate(cases):
            sa, isa, lcp = b.index(k)
            assert sa[0] == len(d)
            assert np.array_equal(sa[1:], O.suffix_array(d)), f"n_big={n_big} case {k}"
            assert got[k] == oc.compress(d), f"n_big={n_big} case {k}"


def test_large_periodic_texts(amd):
    """sa_large.hip on texts whose suffixes share megabytes (many doubling rounds), mixed with small texts in one batch;
    the index against the oracle's suffix array, the bytes through the GPU round trip (the oracle's match search takes
    minutes on these)"""
    rng = np.random.default_rng(5)
    blk = bytes(rng.integers(0, 256, 1000, dtype=np.uint8))
    cases = [(b"abc" * 250000)[:700001], b"tiny", bytes(600000), (blk * 700)[:650000], amd.synth.blob(7),
             bytes(rng.integers(0, 2, 530000, dtype=np.uint8))]
    ctx = amd.batch.new_ctx(b"")
    b = amd.batch.Batch(ctx, len(cases), sum(map(len, cases)))
    b.upload(cases)
    b.compress()
    got = b.download()
    for k, d in enumerate(cases):
        sa, isa, lcp = b.index(k)
        assert sa[0] == len(d)
        assert np.array_equal(sa[1:], O.suffix_array(d)), f"case {k}"
        assert np.array_equal(isa[sa], np.arange(len(d) + 1))
        assert amd.Decompress(got[k], b"") == d, f"case {k}"


def test_large_dictionary(amd):
    """a dictionary of more than 2^19 bytes: its index is built by the device-wide sort too (compress.go:64)"""
    rng = np.random.default_rng(9)
    d = b"".join(amd.synth.blob(300 + k) for k in range(6))[:700000]  # MaxDictSize is 4 MiB (backref.go:11)
    x = amd.synth.blob(301)[1000:30000] + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + amd.synth.blob(305)[:20000]
    want = O.Compressor(d).compress(x)
    c = amd.Compressor(d)
    got = c.Compress(x)
    assert got == want
    assert amd.Decompress(got, d) == x


def test_overlapped_sub_batches_give_the_same_bytes(amd, dict_naive):
    """mi355_lzss_ctx_set_overlap: the stages of 2..4 sub-batches run on concurrent streams; every segment's scratch
    (sort records, then its window index) must stay inside the segment's own region"""
    rng = np.random.default_rng(7)
    blobs = [amd.synth.blob(40 + k)[: int(rng.integers(1, 40000))] for k in range(70)] + [b"", bytes(30000)]
    oc = O.Compressor(dict_naive)
    want = [oc.compress(d) for d in blobs]
    for parts in (1, 2, 3, 4):
        ctx = amd.batch.new_ctx(dict_naive)
        ctx.set_overlap(parts)
        for _ in range(2):  # twice: the second run starts with the first one's leftovers in every buffer
            got = amd.compress_batch(ctx, blobs)
            assert got == want, f"parts={parts}"


def _maxsize():
    import importlib.util
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_maxsize", os.path.join(here, "make_maxsize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(here, "maxsize.json")) as f:
        return mod.cases(), json.load(f)


def test_inputs_beyond_two_mib_match_the_oracle_digests(amd):
    """MaxInputSize (compress.go:473) and texts longer than the 21-bit window (backref.go:27-37), where the long
    window slides: the oracle takes minutes there, so its outputs are committed as digests
    (tests/golden/make_maxsize.py); the GPU streams must hash to them and decode back to the inputs"""
    import hashlib
    cases, want = _maxsize()
    for name, (d, x) in cases.items():
        assert hashlib.sha256(x).hexdigest() == want[name]["in_sha256"], name
        c = amd.Compressor(d)
        out = c.Compress(x)
        assert len(out) == want[name]["out_len"], name
        assert hashlib.sha256(out).hexdigest() == want[name]["out_sha256"], name
        # past 2^21 - len(dict) bytes a dictionary address no longer fits its 21 bits (backref.go:66-67): the stream
        # then decodes to something else than the input, the same something for the reference's decoder and ours
        back = amd.Decompress(out, d)
        assert hashlib.sha256(back).hexdigest() == want[name]["dec_sha256"], name
        assert (back == x) == want[name]["roundtrip"], name


def unpack(amd_ty, amd_addr, amd_len):
    return list(zip(amd_ty.tolist(), amd_addr.tolist(), amd_len.tolist()))


@pytest.mark.parametrize("dict_name", ["nil", "dict_naive"])
def test_best_backrefs_match_oracle(amd, dict_name, dict_naive):
    """every position's (type, address, length) equals the oracle's bestBackref (compress.go:146-173)"""
    dict_ = b"" if dict_name == "nil" else dict_naive
    gc = amd.Compressor(dict_)
    oc = O.Compressor(dict_)
    cases = rand_cases() + [fixture_bytes("3c2943_data.bin"), fixture_bytes("average_block.bin")[:30000]]
    for k, d in enumerate(cases):
        for start in sorted({0, len(d) // 3}):
            g = unpack(*gc.best_backrefs(d, start))
            o = unpack(*oc.best_backrefs(d, start))
            if g != o:
                bad = [i for i in range(len(o)) if g[i] != o[i]]
                raise AssertionError(f"case {k} n={len(d)} start={start}: {len(bad)} differ, first at {start + bad[0]}: "
                                     f"gpu={g[bad[0]]} oracle={o[bad[0]]}")


def test_known_answers(amd):
    c = amd.Compressor(b"")
    for d, expect in [(b"", "000100"), (b"hi", "0001006869"), (bytes(8), "00010000fe060000"),
                      (b"hihihi", "0001006869fe030004"), (bytes([0xFD, 0xFE, 0xFF]), "000100fdff01000010")]:
        assert c.Compress(d).hex() == expect


@pytest.mark.parametrize("dict_name", ["nil", "dict_naive"])
def test_compress_bytes_match_oracle(amd, dict_name, dict_naive):
    dict_ = b"" if dict_name == "nil" else dict_naive
    gc = amd.Compressor(dict_)
    oc = O.Compressor(dict_)
    for k, d in enumerate(rand_cases()):
        assert gc.Compress(d) == oc.compress(d), f"case {k} n={len(d)}"
        assert gc.CompressedSize256k(d) == oc.compressed_size_256k(d)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: c["name"] + "/" + c["dict"])
def test_golden_fixtures(amd, case, dict_naive, average_block):
    """one-shot Compress of every reference fixture equals the committed golden (oracle) stream"""
    name = case["name"]
    d = {"average_block[:102400]": average_block[:102400], "average_block": average_block, "zeros102400": bytes(102400),
         "zeros300": bytes(300), "hi+zeros300": b"hi" + bytes(300)}.get(name)
    if d is None:
        d = fixture_bytes(name)
    dict_ = b"" if case["dict"] == "nil" else dict_naive
    out = amd.Compressor(dict_).Compress(d)
    assert len(out) == case["out_len"]
    assert hashlib.sha256(out).hexdigest() == case["out_sha256"]


def test_batch_synthetic_blobs(amd):
    """config 2/3 shape: 128 KiB synthetic blobs with the shared 64 KiB dictionary, batched"""
    dict_ = amd.synth.dictionary()
    blobs = amd.synth.blobs(0, 6) + [b"", amd.synth.blob(7)[:1000], amd.synth.blob(8)[:77777]]
    ctx = amd.batch.new_ctx(dict_)
    got = amd.compress_batch(ctx, blobs)
    oc = O.Compressor(dict_)
    for k, d in enumerate(blobs):
        assert got[k] == oc.compress(d), f"blob {k}"
        if k == 0:  # blob #0 holds every phrase kind (config 2's "short+long backrefs")
            st = oc.phrase_stats()
            assert min(st["literals"], st["short"], st["long_data"], st["long_dict"], st["rle"]) > 0, st


def test_streaming_matches_oracle(amd, dict_naive, average_block):
    """Write / Len / Revert / ConsiderBypassing / Reset byte-for-byte (compress_test.go:317-363 flow)"""
    data = average_block[:40000]
    gc, oc = amd.Compressor(dict_naive), O.Compressor(dict_naive)
    in_chunk, out_max = 1000, 5000
    i0 = 0
    while i0 < len(data):
        i = i0
        while i < len(data) and gc.Len() < out_max:
            chunk = data[i:min(i + in_chunk, len(data))]
            gc.Write(chunk)
            oc.write(chunk)
            assert gc.Bytes() == oc.bytes()
            unc = i + in_chunk - i0 + 3
            if gc.Len() >= out_max and unc <= out_max and gc.Len() > unc:
                assert gc.ConsiderBypassing() and oc.consider_bypassing()
            i += in_chunk
        if gc.Len() > out_max:
            gc.Revert()
            oc.revert()
            i -= in_chunk
        assert gc.Bytes() == oc.bytes()
        assert O.decompress(gc.Bytes(), dict_naive) == data[i0:min(i, len(data))]
        gc.Reset()
        oc.reset()
        i0 = i


def test_fuzz_like_sequences(amd):
    rng = np.random.default_rng(3)
    alphabets = [bytes([0, 1]), bytes([0, 0xFE, 0xFF, 7]), bytes(range(256)), bytes([0xFE, 0xFF])]
    for t in range(8):
        alpha = alphabets[t % len(alphabets)]
        n = int(rng.integers(1, 60))
        inp = bytes(alpha[j] for j in rng.integers(0, len(alpha), n))
        dict_ = bytes(alpha[j] for j in rng.integers(0, len(alpha), int(rng.integers(0, 40))))
        gc, oc = amd.Compressor(dict_), O.Compressor(dict_)
        for b in inp:
            gc.Write(bytes([b])); oc.write(bytes([b]))
            gc.Revert(); oc.revert()
            gc.Write(bytes([b])); oc.write(bytes([b]))
            assert gc.Bytes() == oc.bytes()
        assert O.decompress(gc.Bytes(), dict_) == inp
        gc.Reset(); oc.reset()
        gc.Write(inp[:-1]); oc.write(inp[:-1])
        gc.Write(inp[-1:]); oc.write(inp[-1:])
        assert gc.Bytes() == oc.bytes()


def test_errors(amd):
    with pytest.raises(amd.LzssError, match="dict size must be <= 4194304"):
        amd.Compressor(bytes((1 << 22) - 1))
    c = amd.Compressor(b"")
    with pytest.raises(amd.LzssError, match="input size must be <= 4194304"):
        c.Write(bytes((1 << 22) + 1))
    with pytest.raises(amd.LzssError, match="input size must be <= 262144"):
        c.CompressedSize256k(bytes(262145))
    c.Reset()
    c.Write(b"abc")
    c.Revert()
    with pytest.raises(amd.LzssError, match="cannot revert twice in a row"):
        c.Revert()
