"""Debug: event counts of the dictionary lookups of match_kernel (library built with -DDICT_STATS)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compress_amd import synth, _lib
from compress_amd.batch import Batch, new_ctx
lib = _lib.lib()
f = lib.mi355_lzss_debug_dict_stats
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
blobs = [synth.blob(b) for b in range(nb)]
ctx = new_ctx(synth.dictionary())
bt = Batch(ctx, len(blobs), sum(len(b) for b in blobs))
bt.upload(blobs)
buf = (C.c_ulonglong * 16)()
f(buf, 1)
bt.compress()
f(buf, 1)
names = ["lookups", "bucket_miss", "probes", "probe_ties", "tie_bytes+8", "final_text_compares", "final_steps", "pd_searches",
         "bracket_sum", "Ld_sum", "bracket2_sum", "coop_up_lanes x64", "coop_up_steps x64", "coop_dn_lanes x64", "coop_dn_steps x64"]
n = max(buf[0], 1)
for k, nm in enumerate(names):
    print("%-20s %12d  %8.3f per lookup" % (nm, buf[k], buf[k] / n))
