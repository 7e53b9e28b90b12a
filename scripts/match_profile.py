"""Debug: per-section clock totals of match_kernel (library built with -DMATCH_PROFILE)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from compress_amd import synth, _lib
from compress_amd.batch import Batch, new_ctx
lib = _lib.lib()
f = lib.mi355_lzss_debug_match_profile
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
blobs = [synth.blob(b) for b in range(nb)]
ctx = new_ctx(synth.dictionary())
bt = Batch(ctx, len(blobs), sum(len(b) for b in blobs))
bt.upload(blobs)
bt.compress()
buf = (C.c_ulonglong * 16)()
f(buf, 1)
bt.compress()
f(buf, 1)
names = ["setup", "up", "coopup", "dn", "coopdn", "lengths", "leftmost_wait", "dict", "prev_lower", "linear", "short_index", "long_index", "d_bucket_key", "d_search", "d_final_lcp"]
tot = sum(buf[:15])
for k, nm in enumerate(names):
    print("%-8s %6.1f%%  %10.0f cycles/wave" % (nm, 100.0 * buf[k] / tot, buf[k] / max(buf[15], 1)))
print("waves", buf[15], "stages", bt.stage_times() if hasattr(bt, "stage_times") else "")
