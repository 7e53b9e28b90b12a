"""CPU-only tests that pin the oracle (oracle/lzss_oracle.c) against everything the reference's
own tests hold for this path: known answers, round trips on lzss/testdata,
the five ratio windows of lzss/regress_test.go, the streaming flows of lzss/compress_test.go."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import FIX, fixture_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def roundtrip(d, dict_):
    c = O.Compressor(dict_)
    out = c.compress(d)
    assert O.decompress(out, dict_) == d
    return out


# ---- suffix array and index query -----------------------------------------------------------

def test_suffix_array_matches_naive_sort():
    rng = np.random.default_rng(7)
    for _ in range(400):
        n = int(rng.integers(1, 300))
        k = int(rng.integers(1, 6))
        d = bytes(rng.integers(0, k, n, dtype=np.uint8))
        assert list(O.suffix_array(d)) == sorted(range(n), key=lambda i: d[i:])


def closed_form(data, sa, s, min_end, max_end, rs, re):
    """longest L in [min_end,max_end] occurring at an offset in [rs,re); the offset
    is the in-range occurrence of smallest suffix rank."""
    rank = np.empty(len(sa), dtype=np.int64)
    rank[sa] = np.arange(len(sa))
    best_l, best_p = -1, -1
    for p in range(rs, re):
        l = 0
        while l < max_end and p + l < len(data) and data[p + l] == s[l]:
            l += 1
        if l < min_end:
            continue
        if l > best_l:
            best_l = l
    if best_l == -1:
        return -1, -1
    for p in range(rs, re):
        if data[p:p + best_l] == s[:best_l] and (best_p == -1 or rank[p] < rank[best_p]):
            best_p = p
    return best_p, best_l


def test_lookup_longest_literal_equals_closed_form():
    rng = np.random.default_rng(11)
    for _ in range(60):
        n = int(rng.integers(8, 400))
        k = int(rng.integers(1, 4))
        data = bytes(rng.integers(0, k, n, dtype=np.uint8))
        sa = O.suffix_array(data)
        for _ in range(20):
            i = int(rng.integers(0, n))
            max_end = min(256, n - i)
            min_end = int(rng.integers(1, 6))
            if min_end > max_end:
                continue
            rs = int(rng.integers(0, i + 1))
            s = data[i:i + max_end]
            got = O.lookup_longest(data, sa, s, min_end, max_end, rs, i)
            assert got == closed_form(data, sa, s, min_end, max_end, rs, i)
            # whole-text range as used for the dictionary (compress.go:463)
            got = O.lookup_longest(data, sa, s, min_end, max_end, 0, n)
            assert got == closed_form(data, sa, s, min_end, max_end, 0, n)


# ---- known answers ---------------------------------------------------------------------------

KATS = [  # SURVEY.md Appendix B, nil dictionary
    (b"", "000100"),                                  # lzss/compress_test.go:373-377
    (b"hi", "0001006869"),
    (bytes(8), "00010000fe060000"),
    (b"hihihi", "0001006869fe030004"),
    (bytes([0xFD, 0xFE, 0xFF]), "000100fdff01000010"),
]


@pytest.mark.parametrize("d,expect", KATS)
def test_known_answers(d, expect):
    assert O.Compressor(b"").compress(d).hex() == expect
    assert O.decompress(bytes.fromhex(expect), b"") == d


def test_invalid_backref_stream():  # lzss/compress_test.go:365-405
    assert O.decompress(bytes.fromhex("00010001fe040000"), b"") == bytes([1] * 6)
    # address 255 with 1 byte of output: offset field = 1-255-1 masked to 14 bits
    field = (1 - 255 - 1) & 0x3FFF
    bits = (0xFE << 22) | (4 << 14) | field
    stream = bytes.fromhex("000100") + (bits << 2).to_bytes(4, "big")
    with pytest.raises(O.OracleError):
        O.decompress(stream, b"")


def test_decompress_header_errors():  # lzss/decompress.go:22-31, header.go:55-63
    with pytest.raises(O.OracleError):
        O.decompress(b"\x00\x02\x00", b"")
    with pytest.raises(O.OracleError):
        O.decompress(b"\x00\x01\x02", b"")
    with pytest.raises(O.OracleError):
        O.decompress(b"\x00\x01", b"")
    assert O.decompress(b"\x00\x01\x01abc", b"") == b"abc"


@pytest.mark.parametrize("d", [bytes(8), bytes(300), b"hi", bytes([253, 254, 255]), bytes([1, 1, 1, 1, 2, 1, 1, 1, 1]),
                               b"\x01" + bytes(8), b"hi" + bytes(300), b"hihihi"])
def test_unit_roundtrips(d, dict_naive):  # lzss/compress_test.go:32-68,220-226
    roundtrip(d, dict_naive)
    roundtrip(d, b"")


def test_augment_dict():  # lzss/compress.go:70-88
    assert O.augment_dict(b"") == b"\xfe\xff"
    assert O.augment_dict(b"a\xfe") == b"a\xfe\xfe\xff"
    assert O.augment_dict(b"\xffa\xfe") == b"\xffa\xfe"


def test_dict_too_large():
    with pytest.raises(O.OracleError, match="dict size must be <= 4194304"):
        O.Compressor(bytes((1 << 22) - 1))


def test_input_too_large(dict_naive):
    c = O.Compressor(b"")
    with pytest.raises(O.OracleError, match="input size must be <= 4194304"):
        c.write(bytes((1 << 22) + 1))


# ---- reference fixtures ----------------------------------------------------------------------

@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: c["name"] + "/" + c["dict"])
def test_golden_cases(case, dict_naive, average_block):
    name = case["name"]
    if name == "average_block[:102400]":
        d = average_block[:102400]
    elif name == "average_block":
        d = average_block
    elif name == "zeros102400":
        d = bytes(102400)
    elif name == "zeros300":
        d = bytes(300)
    elif name == "hi+zeros300":
        d = b"hi" + bytes(300)
    else:
        d = fixture_bytes(name)
    dict_ = b"" if case["dict"] == "nil" else dict_naive
    assert hashlib.sha256(d).hexdigest() == case["in_sha256"]
    out = roundtrip(d, dict_)
    assert len(out) == case["out_len"]
    assert hashlib.sha256(out).hexdigest() == case["out_sha256"]


REF_RATIOS = {  # lzss/regress_test.go:16-32, tolerance :64
    "blobs_1-1865800": 4.19, "blobs_1-goerli-3690632": 23.81, "blobs_2-1865938": 3.73,
    "blobs_3-1866069": 3.55, "blobs_5-1128897": 7.17,
}


@pytest.mark.parametrize("name,ratio", sorted(REF_RATIOS.items()))
def test_reference_ratio_windows(name, ratio, dict_naive):
    d = fixture_bytes(name)
    out = roundtrip(d, dict_naive)
    assert abs(ratio - len(d) / len(out)) <= 0.05


# ---- streaming state machine -----------------------------------------------------------------

def craft_expanding_input(dict_, size):  # lzss/compress_test.go:428-475
    nb = 4
    covered = set()
    for i, b in enumerate(dict_):
        if b == 255:
            covered.add(int.from_bytes(dict_[i + 1:i + nb], "little"))
    res = bytearray(size)
    ctr = 0
    for i in range(0, size, nb):
        while ctr in covered:
            ctr += 1
        res[i] = 255
        chunk = ctr.to_bytes(8, "little")[:nb - 1]
        res[i + 1:i + nb] = chunk[:max(0, min(nb - 1, size - i - 1))]
        ctr += 1
    return bytes(res)


def test_craft_expanding_input(dict_naive):  # lzss/compress_test.go:407-426
    d = craft_expanding_input(dict_naive, 100000)
    c = O.Compressor(dict_naive)
    out = c.compress(d)
    len_c = len(out)
    assert 10 * len_c // len(d) > 12
    c.reset()
    c.write(d)
    assert c.consider_bypassing()
    assert c.len() < len_c
    assert O.decompress(c.bytes(), dict_naive) == d


def test_revert_flow(dict_naive, average_block):  # lzss/compress_test.go:317-363 on a 120 kB prefix
    data = average_block[:120000]
    c = O.Compressor(dict_naive)
    in_chunk, out_max = 1000, 5000
    i0 = 0
    nonzero_pad = 0
    while i0 < len(data):
        i = i0
        while i < len(data) and c.len() < out_max:
            c.write(data[i:min(i + in_chunk, len(data))])
            unc = i + in_chunk - i0 + 3
            if c.len() >= out_max and unc <= out_max and c.len() > unc:
                assert c.consider_bypassing()
            i += in_chunk
        if c.len() > out_max:
            c.revert()
            i -= in_chunk
        blob = c.bytes()
        assert len(blob) <= out_max
        assert O.decompress(blob, dict_naive) == data[i0:min(i, len(data))]
        c.reset()
        i0 = i
    assert i0 >= len(data)


def test_revert_after_bypass(dict_naive):  # lzss/compress_test.go:477-506 (feeds the hex TEXT)
    hex_text = fixture_bytes("average_block.bin")[:250].hex().encode()
    assert len(hex_text) == 500
    c = O.Compressor(dict_naive)
    c.write(hex_text)
    c.write(craft_expanding_input(dict_naive, 1000))
    assert c.consider_bypassing()
    c.revert()
    out = c.bytes()
    assert O.decompress(out, dict_naive) == hex_text
    assert len(out) < 500


def test_revert_twice_errors():
    c = O.Compressor(b"")
    c.write(b"abc")
    c.revert()
    with pytest.raises(O.OracleError, match="cannot revert twice in a row"):
        c.revert()


def fuzz_sequences(inp, dict_):  # lzss/compress_test.go:71-185
    def check(cb):
        assert O.decompress(cb, dict_) == inp
    c = O.Compressor(dict_)
    check(c.compress(inp))
    c = O.Compressor(dict_)
    for b in inp:
        c.write(bytes([b]))
    check(c.bytes())
    c = O.Compressor(dict_)
    for b in inp:
        c.write(bytes([b]))
        c.revert()
    c = O.Compressor(dict_)
    for b in inp:
        c.write(bytes([b]))
        c.revert()
        c.write(bytes([b]))
    check(c.bytes())
    c.reset()
    c.write(inp)
    check(c.bytes())
    if len(inp) > 1:
        c.reset()
        c.write(inp[:-1])
        c.write(inp[-1:])
        check(c.bytes())
        c.reset()
        c.write(inp[:1])
        c.write(inp[1:])
        check(c.bytes())


def test_fuzz_like_sequences():
    rng = np.random.default_rng(3)
    alphabets = [bytes([0, 1]), bytes([0, 0xFE, 0xFF, 7]), bytes(range(256)), bytes([0xFF]), bytes([0xFE, 0xFF])]
    for t in range(60):
        alpha = alphabets[t % len(alphabets)]
        n = int(rng.integers(0, 400))
        inp = bytes(alpha[j] for j in rng.integers(0, len(alpha), n))
        dn = int(rng.integers(0, 64))
        dict_ = bytes(alpha[j] for j in rng.integers(0, len(alpha), dn))
        fuzz_sequences(inp, dict_)


def test_compressed_size_equals_len(dict_naive, average_block):  # lzss/compress_test.go:187-218
    rng = np.random.default_rng(5)
    c = O.Compressor(dict_naive)
    for lo, n in [(0, 0), (0, 1), (1 << 17, 1 << 16), (5000, 3000), (100, 262144)]:
        d = average_block[lo:lo + n]
        assert c.compressed_size_256k(d) == len(c.compress(d))
    for _ in range(30):
        n = int(rng.integers(0, 600))
        d = bytes(rng.integers(0, 3, n, dtype=np.uint8) * 127)
        cc = O.Compressor(bytes(rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8)))
        assert cc.compressed_size_256k(d) == len(cc.compress(d))
    with pytest.raises(O.OracleError, match="input size must be <= 262144"):
        c.compressed_size_256k(bytes(262145))


def test_chunking_changes_bytes(dict_naive, average_block):
    """streaming Writes are not the one-shot stream (lookahead / ranks stop
    at the end of the current inBuf), yet both decode to the same input."""
    d = average_block[:30000]
    c = O.Compressor(dict_naive)
    one = bytes(c.compress(d))
    c.reset()
    for i in range(0, len(d), 1000):
        c.write(d[i:i + 1000])
    chunked = c.bytes()
    assert O.decompress(one, dict_naive) == d and O.decompress(chunked, dict_naive) == d
    assert (len(one), len(chunked)) == (6298, 6356)
