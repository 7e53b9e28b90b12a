"""Debug: decompress_kernel time against the number of blobs in the batch (HIP-event stage time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from compress_amd import _lib
from compress_amd import synth
from compress_amd.batch import Batch, new_ctx

nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
size = 131072
arr = synth.blobs_array(0, 1, nmax, size)
ctx = new_ctx(synth.dictionary())
bt = Batch(ctx, nmax, nmax * size)
off = np.arange(nmax + 1, dtype=np.uint64) * size
bt.upload_flat(arr.reshape(-1), off)
bt.compress()
outs = bt.download()
blobs = [arr[k].tobytes() for k in range(nmax)]
for cnt in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1250,2500,5000,8192,10000".split(","))]:
    if cnt > nmax:
        continue
    bt.upload_compressed(outs[:cnt], size)
    bt.decompress()
    ms = []
    for _ in range(3):
        bt.decompress()
        ms.append(bt.stage_times()["decompress"])
    back = bt.download_decompressed()
    if hasattr(_lib.lib(), "mi355_lzss_debug_dec_profile"):
        buf = (C.c_ulonglong * 16)()
        _lib.lib().mi355_lzss_debug_dec_profile(buf, 1)
        w = max(buf[15], 1)
        print("   per wave-run: refill %d parse %d wait %d stores+loads %d slow %d cycles; batches %d steps %d slow %d (waves %d)" % (
            buf[0] // w, buf[1] // w, buf[2] // w, buf[3] // w, buf[4] // w, buf[8] // w, buf[9] // w, buf[10] // w, w))
    print("blobs %6d  decompress %.3f ms (min of 3)  equal %s" % (cnt, min(ms), back == blobs[:cnt]), flush=True)
