import os, sys
sys.path.insert(0, "/root/repo")
from compress_amd import synth
from compress_amd.batch import Batch, new_ctx
ctx = new_ctx(synth.dictionary())
ctx.set_overlap(1)
for nb in (1, 8, 64):
    blobs = [synth.blob(b) for b in range(nb)]
    bt = Batch(ctx, nb, sum(map(len, blobs)))
    for rep in range(3):
        bt.upload(blobs); bt.compress(); bt.sizes()
    st = bt.stage_times()
    print(nb, round(sum(st.values()), 2), {k: round(v, 3) for k, v in st.items()}, flush=True)
    bt.close()
