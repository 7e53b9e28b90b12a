#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's test data and the CPU oracle.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py

* fixtures/*: DATA files the reference's own tests hold (lzss/testdata), copied byte for byte;
  average_block.hex is stored decoded (average_block.bin).
* golden.json: size + sha256 of the oracle's one-shot Compress output for every fixture.  These
  are ORACLE outputs (the Go reference cannot run here); the reference pins only sizes (ratio
  windows, lzss/regress_test.go:16-32) and round trips, which tests/test_oracle.py asserts.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF = "/root/reference/lzss/testdata"
FIX = os.path.join(HERE, "fixtures")

FILES = ["3c2943/data.bin", "705b24/data.bin", "777003/data.bin", "c9b5a2/data.bin", "e4207e/data.bin",
         "fa4a22/data.bin", "bug/data.bin", "large/data.bin", "blobs/1-1865800", "blobs/1-goerli-3690632",
         "blobs/2-1865938", "blobs/3-1866069", "blobs/5-1128897"]


def main():
    os.makedirs(FIX, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "dict_naive"), os.path.join(FIX, "dict_naive"))
    avg = bytes.fromhex(open(os.path.join(REF, "average_block.hex")).read().strip())
    open(os.path.join(FIX, "average_block.bin"), "wb").write(avg)
    names = {}
    for f in FILES:
        flat = f.replace("/", "_")
        shutil.copyfile(os.path.join(REF, f), os.path.join(FIX, flat))
        names[flat] = f
    dict_ = open(os.path.join(FIX, "dict_naive"), "rb").read()
    golden = {"_comment": "oracle one-shot Compress outputs; dict = dict_naive unless dict == 'nil'", "cases": []}

    def add(name, data, d, dname):
        c = O.Compressor(d)
        out = c.compress(data)
        assert O.decompress(out, d) == data
        golden["cases"].append({"name": name, "dict": dname, "in_len": len(data), "in_sha256": hashlib.sha256(data).hexdigest(),
                                "out_len": len(out), "out_sha256": hashlib.sha256(out).hexdigest(), "stats": c.phrase_stats()})

    add("average_block[:102400]", avg[:102400], b"", "nil")
    add("average_block[:102400]", avg[:102400], dict_, "dict_naive")
    add("average_block", avg, dict_, "dict_naive")
    add("zeros102400", bytes(102400), dict_, "dict_naive")
    add("zeros300", bytes(300), b"", "nil")
    add("hi+zeros300", b"hi" + bytes(300), b"", "nil")
    for flat in sorted(names):
        add(flat, open(os.path.join(FIX, flat), "rb").read(), dict_, "dict_naive")
    json.dump(golden, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote", len(golden["cases"]), "cases")


if __name__ == "__main__":
    main()
