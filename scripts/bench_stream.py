"""Config 4: a stream packed into EIP-4844-sized blobs, chunk 1000 B, outMax 126 976 B.
Sequential Write / Revert loop on the GPU compressor vs the speculative packer (compress_amd/stream.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compress_amd
from compress_amd import stream, synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
seq_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n_blobs = int(mb * 1e6 / synth.BLOB_SIZE) + 1
data = b"".join(synth.blob(b) for b in range(n_blobs))[: int(mb * 1e6)]
dict_ = synth.dictionary()
comp = compress_amd.Compressor(dict_)
comp.Compress(data[:100000])  # warm up
for la in (16, 32, 64, 128):
    t0 = time.perf_counter()
    got = stream.pack_stream(comp, data, 1000, 126976, la)
    dt = time.perf_counter() - t0
    print("speculative lookahead %3d: %6.2f MB/s  (%d blobs, %.1f s)" % (la, len(data) / dt / 1e6, len(got), dt), flush=True)
part = data[: int(seq_mb * 1e6)]
t0 = time.perf_counter()
want = stream.pack_stream_sequential(comp, part, 1000, 126976)
dt = time.perf_counter() - t0
print("sequential Write/Revert loop: %6.2f MB/s  (%d blobs, %.1f s)" % (len(part) / dt / 1e6, len(want), dt), flush=True)
k = len(want) - 1  # the last sequential blob ends with the part, not with the cap
print("first %d blobs equal: %s" % (k, got[:k] == want[:k]))
