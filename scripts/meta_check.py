"""debug: suffix arrays of a few small texts from the library against a plain sort (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import compress_amd
from compress_amd import batch as B, synth

def true_sa(T):
    n = len(T)
    return [n] + sorted(range(n), key=lambda i: T[i:])

print("creating ctx", flush=True)
ctx = B.new_ctx(b"")
print("ctx ok", flush=True)
cases = [b"a", b"ab", b"aa", b"abab", b"\x00" * 10, b"\x00" * 300, b"ab" * 50 + b"\x00" * 260 + b"\x01", synth.blob(0)[:3000], synth.blob(1)[:40000]]
cases += [b"\x00" * 253 + b"\x01", b"\x00" * 506, b"\x00" * 507 + b"a", b"\x07" * 759 + b"\x01" + b"\x07" * 760 + b"\x09" + b"\x07" * 300,
          (b"\x00" * 254 + b"\x01") * 80, (b"\x05" * 300 + b"\x01" + b"\x05" * 280 + b"\x09") * 40, b"\x00" * 20000,
          b"ab" * 3000, b"\xff" * 1000 + b"\x00" * 1000 + b"\xff" * 1001 + b"\x00" * 999]
import random
random.seed(5)
for t in range(6):
    T = bytearray()
    n = random.randint(100, 5000)
    while len(T) < n:
        if random.random() < 0.3:
            T += bytes([random.randrange(4)]) * random.choice([1, 2, 3, 5, 30, 252, 253, 254, 255, 300])
        else:
            T += bytes(random.randrange(4) for _ in range(random.randint(1, 6)))
    cases.append(bytes(T[:n]))
bt = B.Batch(ctx, len(cases), sum(map(len, cases)))
bt.upload(cases)
print("uploaded", flush=True)
bt.compress()
print("compressed", flush=True)
bad = 0
for k, T in enumerate(cases):
    sa, isa, lcp = bt.index(k)
    ref = true_sa(T)
    ok = list(sa) == ref and all(isa[sa[j]] == j for j in range(len(ref)))
    print(k, len(T), "OK" if ok else "DIFF", flush=True)
    if not ok:
        bad += 1
        d = [j for j in range(len(ref)) if sa[j] != ref[j]]
        print("  first diffs at ranks", d[:10], "got", [int(sa[j]) for j in d[:10]], "want", [ref[j] for j in d[:10]])
print("bad", bad)
