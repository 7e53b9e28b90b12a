#!/usr/bin/env python3
"""the oracle lands on the reference's recorded ratio to two decimals for two of the five
blobs of lzss/regress_test.go:16-32 and is ~0.03 low (0.8 % larger output) on the three literal-heavy ones, all
inside the test's +-0.05 window.  The README still documents a variable long-offset width NBBITS_DYN_OFS
(README.md:62) while the code fixes it at 21 bits (lzss/backref.go:37-40).  This script bounds the hypothesis
"the recorded ratios predate the fixed 21-bit offset": it takes the oracle's phrase list of each blob and recounts
the stream size with narrower long references (first-order: same phrases, the choice between phrases would shift
slightly too), for three candidate width rules.

Run in the build container:  python tests/golden/ratio_hypotheses.py   (CPU only, ~1 minute; test infrastructure)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

FIX = os.path.join(HERE, "fixtures")
RECORDED = {"blobs_1-1865800": 4.19, "blobs_1-goerli-3690632": 23.81, "blobs_2-1865938": 3.73, "blobs_3-1866069": 3.55,
            "blobs_5-1128897": 7.17}   # lzss/regress_test.go:16-32


def bitlen(x):
    return max(1, int(x).bit_length())


def main():
    dict_ = open(os.path.join(FIX, "dict_naive"), "rb").read()
    aug = O.augment_dict(dict_)
    rules = {
        "fixed 21 (the code)": lambda pos, n: 21,
        "bitlen(dict + position)": lambda pos, n: bitlen(len(aug) + pos),
        "bitlen(dict + input size)": lambda pos, n: bitlen(len(aug) + n),
        "fixed 19": lambda pos, n: 19,
        "fixed 18": lambda pos, n: 18,
    }
    print("%-26s %9s %9s | %s" % ("blob", "recorded", "oracle", "  ".join("%-26s" % r for r in list(rules)[1:])))
    hits = {r: 0 for r in rules}
    for name, want in RECORDED.items():
        data = open(os.path.join(FIX, name), "rb").read()
        out = O.Compressor(dict_).compress(data)
        phrases = O.compressed_stream_info(out, dict_)
        longs = [p[3] - len(aug) for p in phrases if p[0] == 0xFF]   # start_decompressed counts the dictionary
        bits_now = 8 * (len(out) - 3)
        row = []
        for r, fn in rules.items():
            saved = sum(21 - fn(pos, len(data)) for pos in longs)
            size = 3 + (bits_now - saved + 7) // 8
            ratio = len(data) / size
            ok = abs(round(ratio, 2) - want) < 0.005
            hits[r] += ok
            row.append("%.4f%s (%d B)" % (ratio, "*" if ok else " ", size))
        print("%-26s %9.2f %9s | %s" % (name, want, row[0].split()[0], "  ".join("%-26s" % x for x in row[1:])))
    print()
    for r, h in hits.items():
        print("%-28s reproduces %d of 5 recorded ratios to two decimals" % (r, h))


if __name__ == "__main__":
    main()
