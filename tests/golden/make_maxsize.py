#!/usr/bin/env python3
"""Golden digests for inputs beyond 2 MiB, where the 21-bit window slides (lzss/backref.go:27-37) and the input
size limit is reached (lzss/compress.go:473).  The oracle needs minutes on such inputs, so its outputs are kept
as digests in maxsize.json; the inputs are rebuilt by `cases()` from the package's deterministic generator.

Past 2^21 - len(dict) bytes a reference INTO THE DICTIONARY no longer fits the 21 address bits: backref.writeTo
(lzss/backref.go:66-67) hands i + DictLen - address - 1 to a 21-bit field, the high bits are lost and the decoder
copies from somewhere else.  The stream is what the reference writes and both decoders agree on what it decodes
to, but that is not the input ("roundtrip": false); the digests record both facts.

    python tests/golden/make_maxsize.py        # rewrites tests/golden/maxsize.json (oracle only, no GPU)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def cases():
    """name -> (dictionary, input).  Far repeats sit just inside and just outside the long window."""
    from compress_amd import synth
    base = b"".join(synth.blob(100 + k) for k in range(34))
    d = synth.dictionary()
    x = bytearray(base[: 1 << 22])                       # exactly MaxInputSize
    x[2300000:2303000] = x[0:3000]                       # > 2^21 back: out of reach of the long window
    x[4000000:4003000] = x[1903000:1906000]              # 2097000 back: just inside it (2^21 = 2097152)
    x[4100000:4101000] = x[2002847:2003847]              # 2097153 back: just outside
    y = bytearray(base[: 2621440])                       # 2.5 MiB, long zero stretch across the 2 MiB mark
    y[2090000:2110000] = bytes(20000)
    return {"max_input_dict": (d, bytes(x)), "sliding_window_nil": (b"", bytes(y))}


def main():
    import oracle_lib as O
    out = {}
    for name, (d, x) in cases().items():
        c = O.Compressor(d).compress(x)
        back = O.decompress(c, d)
        out[name] = {"n": len(x), "in_sha256": hashlib.sha256(x).hexdigest(), "out_len": len(c),
                     "out_sha256": hashlib.sha256(c).hexdigest(), "dec_sha256": hashlib.sha256(back).hexdigest(),
                     "roundtrip": back == x}
        print(name, out[name], flush=True)
    with open(os.path.join(HERE, "maxsize.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
