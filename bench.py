#!/usr/bin/env python3
"""bench.py - compress MB/s of the MI355X lzss path on BASELINE.json's config 3 workload.

A step = one pass of the hot path (suffix-array index build, per-position back-reference search, phrase
walk, bit emission) over one batch of synthetic 128 KiB calldata-like blobs that share a 64 KiB
dictionary.  Config 3's batch is 10 000 blobs (1.31 GB; ~52 GB of index arrays) and fits one MI355X, so at
N = 1 a step is that whole batch; with N GPUs every rank takes a batch of its own (blob ids b with
b mod N == rank; weak scaling, no collective on the data path).  Round 1 ran 1250 blobs per GPU
(`--blobs 1250`): the same kernels, but their last wave of workgroups then fills less than half the chip.
`value`: inputs are resident in HBM before the timed region, outputs stay in HBM.  `transfer_inclusive`: the
same blobs from pinned host buffers through the library's chunk pipeline, host to host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blobs B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def _pmc_file():
    """the newest committed PMC summary (profiles/rNN_x_pmc_hbm_traffic.csv, named per round)"""
    d = os.path.join(ROOT, "profiles")
    names = sorted(f for f in os.listdir(d) if f.endswith("_pmc_hbm_traffic.csv")) if os.path.isdir(d) else []
    return names[-1] if names else None


PMC_FILE = _pmc_file()


def pmc_traffic(stage, n_blobs):
    """HBM bytes per launch of the stage's kernel from the COMMITTED rocprofv3 --pmc passes of this same command
    (scripts/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate runs with --overlap 1, KB -> bytes) - a
    profile of the committed kernels, not a measurement of the run that prints it (PMC passes cannot run inside
    the bench).  Returns (traffic, fetch_reported, write, source) with traffic = WRITE_SIZE + 2 x FETCH_SIZE: on gfx950
    FETCH_SIZE reports half the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM); scripts/calib_fetch.sh
    (profiles/r02_fetch_calibration.txt) measured that factor 0.500 for 1-, 4-, 8- and 16-byte-per-lane streams alike, 1.0
    per 64-byte line for scattered 4-byte gathers and 1.0-1.05 for WRITE_SIZE - the kernels here mix streams and gathers,
    so the doubled figure is an upper bound (by at most the gathers' share).  Launches of the whole batch in the profiled
    run: the file's "# batch_launches=" note (the one-blob and dictionary launches add < 0.1 % to the sums)."""
    none = (None, None, None, None)
    if PMC_FILE is None:
        return none
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    kernel = {"sa_build": "lzss::sa_build_kernel", "match": "lzss::match_kernel", "lcp": "lzss::lcp_kernel"}.get(stage)
    if kernel is None or not os.path.exists(path):
        return none
    lines = open(path).read().splitlines()
    profiled_blobs = 1250   # round-1 profiles carry no note; later ones start with "# blobs=N"
    batch_launches = None
    while lines and lines[0].startswith("#"):
        k_, v_ = lines[0][1:].strip().split("=")
        if k_ == "blobs":
            profiled_blobs = int(v_)
        elif k_ == "batch_launches":   # launches of the whole batch in the profiled run (the rest are one-blob launches)
            batch_launches = int(v_)
        lines = lines[1:]
    if n_blobs != profiled_blobs:
        return none
    fetch, write, launches = 0.0, 0.0, 1
    for line in lines[1:]:
        k, c, v, nd = line.rsplit(",", 3)
        k = k.replace("void ", "").split("<")[0]
        if k == "lzss::lcp_lds_kernel":
            k = "lzss::lcp_kernel"
        if k == kernel and c == "FETCH_SIZE":
            fetch += float(v) * 1024.0
        if k == kernel and c == "WRITE_SIZE":
            write += float(v) * 1024.0
        if k == "lzss::match_kernel" and c == "FETCH_SIZE":
            launches = max(1, int(nd))
    launches = batch_launches or launches
    return (int((2.0 * fetch + write) / launches), int(fetch / launches), int(write / launches),
            "committed profile profiles/%s (rocprofv3 --pmc, WRITE_SIZE + 2 x FETCH_SIZE per launch of the whole batch; "
            "not measured in this run)" % PMC_FILE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blobs", type=int, default=10000, help="blobs per GPU per step (config 3: a batch of 10 000 blobs; it "
                    "fits one MI355X, so every GPU takes a whole batch - weak scaling)")
    ap.add_argument("--blob-size", type=int, default=131072)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming (config 4) sample")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-blob (config 2) latency sample")
    ap.add_argument("--overlap", type=int, default=2, help="concurrent sub-batches of the index / match stages")
    ap.add_argument("--stage-steps", type=int, default=2, help="extra untimed steps (one part) for the per-kernel times")
    ap.add_argument("--pipe-repeats", type=int, default=1, help="the batch repeated this many times through the "
                    "host-buffer pipeline (transfer-inclusive figure); 0 = skip")
    ap.add_argument("--chunk-blobs", type=int, default=1536, help="blobs per chunk of that pipeline")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's parallel leg (0 = min(16, cores))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- workload: this rank's shard of config 3, generated by a pool of host processes BEFORE anything touches the
    #      GPU (the generator is pure Python, ~20 ms per blob)
    from compress_amd import shard, synth
    dict_ = synth.dictionary()
    ids = shard.shard_ids(rank, world, args.blobs)
    t_gen = time.perf_counter()
    nproc = max(1, min(16, os.cpu_count() or 1, len(ids) // 64))
    if nproc > 1:
        import multiprocessing as mp
        # close + join, not the context manager's terminate(): under rocprofv3 every worker carries the profiler's
        # SIGTERM handler, which can hang in its finalisation and stall the whole run
        pool = mp.get_context("fork").Pool(nproc)
        try:
            blobs = pool.starmap(synth.blob, [(b, args.blob_size) for b in ids], chunksize=32)
        finally:
            pool.close()
            pool.join()
    else:
        blobs = [synth.blob(b, args.blob_size) for b in ids]
    t_gen = time.perf_counter() - t_gen
    in_bytes = sum(map(len, blobs))

    import torch
    import torch.distributed as dist
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # MI355_BENCH_REHEARSE=1: the N > 1 launch path on a box with fewer GPUs than ranks (ranks share the devices, gloo
    # carries the barrier and the reductions) - for checking the multi-process flow, never for a number
    rehearse = os.environ.get("MI355_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearse else "cuda"
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import compress_amd
    from compress_amd import batch as B

    ctx = B.new_ctx(dict_, device=local_rank)
    bt = B.Batch(ctx, len(blobs), in_bytes)
    t0 = time.perf_counter()
    bt.upload(blobs)                       # host -> HBM, outside the timed region
    t_up = time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: the production setting, two sub-batches whose stages overlap on concurrent streams
    # (mi355_lzss_ctx_set_overlap; MI355_LZSS_SPLIT overrides).  Per-kernel stage times need the stages back to
    # back, so they come from --stage-steps extra steps with one part, outside the timed region.
    parts = int(os.environ.get("MI355_LZSS_SPLIT", args.overlap))
    ctx.set_overlap(parts)
    for _ in range(args.warmup):
        bt.compress()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bt.compress()                      # returns after the library's stream has drained
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, device=red_dev)
    stage_sum = {}
    ctx.set_overlap(1)
    for _ in range(args.stage_steps):
        bt.compress()
        for k, v in bt.stage_times().items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    ctx.set_overlap(parts)

    out_len, nbits, out_total = bt.sizes()
    t0 = time.perf_counter()
    outs = bt.download()
    t_down = time.perf_counter() - t0

    # ---- secondary kernel (config 5): batched decompression of the streams just produced, HBM resident
    bt.upload_compressed(outs, args.blob_size)
    bt.decompress()
    barrier()
    t0 = time.perf_counter()
    dec_ms = 0.0
    for _ in range(args.steps):
        bt.decompress()
        dec_ms += bt.stage_times()["decompress"]
    barrier()
    dec_elapsed = time.perf_counter() - t0
    dec_elapsed = shard.max_over_ranks(dec_elapsed, device=red_dev)
    back = bt.download_decompressed()
    dec_ok = back == blobs

    # ---- transfer-inclusive: the same blobs from pinned HOST buffers through mi355_lzss_compress_batch, which pipelines
    #      chunks of 1536 blobs (upload of chunk k+1 and download of chunk k-1 overlap the kernels of chunk k); timed
    #      around the whole call, outputs land in the caller's CSR layout on the host
    pipe = None
    if args.pipe_repeats > 0:
        import ctypes as C
        import numpy as np
        from compress_amd import _lib
        reps = args.pipe_repeats
        nb = len(blobs) * reps
        flat_one = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        pin_in = compress_amd.PinnedBuffer(flat_one.size * reps)
        for r_ in range(reps):
            pin_in.array[r_ * flat_one.size:(r_ + 1) * flat_one.size] = flat_one
        lens = np.array([len(b) for b in blobs] * reps, dtype=np.uint64)
        in_off = np.zeros(nb + 1, dtype=np.uint64)
        np.cumsum(lens, out=in_off[1:])
        caps = np.array([(int(x) + 15) // 16 * 16 + 16 for x in out_len] * reps, dtype=np.uint64)  # the known sizes + slack
        out_off = np.zeros(nb + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        pin_out = compress_amd.PinnedBuffer(int(out_off[-1]))
        p_len = np.zeros(nb, dtype=np.uint32)
        p_st = np.zeros(nb, dtype=np.int32)
        ctx.set_chunk_bytes(args.chunk_blobs * args.blob_size)

        def run_pipe():
            rc = _lib.lib().mi355_lzss_compress_batch(ctx.h, pin_in.ptr, in_off.ctypes.data_as(C.c_void_p), nb, pin_out.ptr,
                                                      out_off.ctypes.data_as(C.c_void_p), p_len.ctypes.data_as(C.c_void_p),
                                                      p_st.ctypes.data_as(C.c_void_p))
            if rc != 0 or p_st.any():
                raise SystemExit("pipelined batch failed: " + _lib.last_error())
        run_pipe()                          # warm-up: allocates the two pipeline workspaces
        barrier()
        t0 = time.perf_counter()
        run_pipe()
        barrier()
        t_pipe = shard.max_over_ranks(time.perf_counter() - t0, device=red_dev)
        same = all(pin_out.array[int(out_off[k]):int(out_off[k]) + int(p_len[k])].tobytes() == outs[k % len(blobs)]
                   for k in range(0, nb, max(1, nb // 97)))
        pipe = {"value": round(in_bytes * reps * world / t_pipe / 1e6, 2), "unit": "MB/s", "blobs_per_gpu": nb,
                "chunk_blobs": args.chunk_blobs, "seconds": round(t_pipe, 4), "host_buffers": "pinned (mi355_lzss_host_alloc)",
                "equal_to_resident_outputs_on_sample": bool(same),
                "note": "whole call: H2D + kernels + D2H + scatter into the caller's CSR layout; the batch is the step's "
                        "blobs repeated %d times" % reps}
        pin_in.close()
        pin_out.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = in_bytes * world * args.steps / elapsed / 1e6
        stages = {k: v / args.stage_steps for k, v in stage_sum.items() if v > 0}
        dom = max(stages, key=stages.get)
        # algorithmic bytes of one launch: sum(n_in + n_out) + n_dict
        alg_bytes = in_bytes + int(out_total) + len(compress_amd.AugmentDict(dict_))
        achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9
        traffic, fetch_rep, write_rep, traffic_src = pmc_traffic(dom, len(blobs))
        line = {
            "metric": "compress MB/s (whole node, bit-exact) + % HBM roofline at 1/2/4/8 GPUs",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "config3: batch of 10 000 independent 128 KiB calldata-like blobs, shared 64 KiB dictionary, "
                                   "one batch per GPU (blob ids b mod N == rank)", "blobs_per_gpu": len(blobs), "blob_bytes": args.blob_size,
                       "dict_bytes": len(dict_), "ratio": round(in_bytes / max(1, int(out_total)), 3),
                       "overlap_parts": parts},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_fetch_size_reported": fetch_rep,
                         "traffic_write_size": write_rep, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stages[dom], 3)},
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "stages_note": "HIP-event time of every stage with the stages back to back (overlap 1), measured on extra "
                           "steps outside the timed region; their sum exceeds ms_per_step by what the overlap hides",
            "decompress": {"value": round(in_bytes * world * args.steps / dec_elapsed / 1e6, 2), "unit": "MB/s (decompressed bytes)",
                           "kernel_ms": round(dec_ms / args.steps, 3), "roundtrip_equal": dec_ok,
                           "achieved_GBps": round((in_bytes + int(out_total)) / (dec_ms / args.steps * 1e-3) / 1e9, 3)},
            "kernel_only_MBps": round(value, 2),
            "transfer_inclusive": pipe,
            "transfer_inclusive_frac_of_resident": round(pipe["value"] / value, 3) if pipe else None,
            "serial_upload_compress_download_MBps": round(in_bytes / (t_up + elapsed / args.steps + t_down) / 1e6, 2),
        }
        # ---- config 2: latency of ONE 128 KiB blob with the 64 KiB dictionary through Compress (host buffers in and out)
        if world == 1 and not args.no_latency:
            comp1 = compress_amd.Compressor(dict_, device=local_rank)
            comp1.Compress(blobs[0])
            lat = []
            for _ in range(7):
                t0 = time.perf_counter()
                o1 = comp1.Compress(blobs[0])
                lat.append(time.perf_counter() - t0)
            line["single_blob_latency_ms"] = {"workload": "config2: synthetic blob #0 (131072 B), 64 KiB dictionary, Compress()",
                                              "median": round(sorted(lat)[len(lat) // 2] * 1e3, 3), "min": round(min(lat) * 1e3, 3),
                                              "equal_to_batch_output": o1 == outs[0]}
        # ---- config 4 (streaming): the Write / Revert loop vs the speculative packer on a sample
        if world == 1 and not args.no_stream:
            from compress_amd import stream as S
            sdata = b"".join(blobs[:16])                      # 2 MiB of the same synthetic stream
            comp = compress_amd.Compressor(dict_, device=local_rank)
            t0 = time.perf_counter()
            spec = S.pack_stream_native(comp, sdata, 1000, 126976)
            t_spec = time.perf_counter() - t0
            t0 = time.perf_counter()
            seq = S.pack_stream_sequential(comp, sdata[:300000], 1000, 126976)
            t_seq = time.perf_counter() - t0
            line["stream"] = {"workload": "config4 sample: chunks of 1000 B into blobs of <= 126976 B",
                              "speculative_MBps": round(len(sdata) / t_spec / 1e6, 2), "speculative_bytes": len(sdata),
                              "sequential_MBps": round(300000 / t_seq / 1e6, 3), "sequential_bytes": 300000,
                              "blobs": len(spec),
                              "equal_on_sequential_sample": S.pack_stream_native(comp, sdata[:300000], 1000, 126976) == seq}
        # ---- bit-exactness spot check + CPU baseline (oracle = checker / reported baseline only)
        if not args.no_cpu and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            import threading
            oc = O.Compressor(dict_)
            done = 0
            t0 = time.perf_counter()
            exact = True
            while done < len(blobs) and time.perf_counter() - t0 < args.cpu_seconds / 2:
                ref = oc.compress(blobs[done])
                exact = exact and (ref == outs[done])
                done += 1
            cpu_t = time.perf_counter() - t0
            one = done * args.blob_size / cpu_t / 1e6
            # the parallel leg: one blob per thread at a time, one oracle compressor per thread (ctypes drops the GIL)
            nthr = args.cpu_threads or min(16, os.cpu_count() or 1)
            counts = [0] * nthr
            flags = [True] * nthr
            t_end = time.perf_counter() + args.cpu_seconds / 2
            nxt = [done]
            lock = threading.Lock()

            def work(t):
                c = O.Compressor(dict_)
                while time.perf_counter() < t_end:
                    with lock:
                        k = nxt[0]
                        nxt[0] += 1
                    if k >= len(blobs):
                        break
                    flags[t] = flags[t] and (c.compress(blobs[k]) == outs[k])
                    counts[t] += 1
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            par_t = time.perf_counter() - t0
            par_done = sum(counts)
            line["cpu_baseline"] = {"value": round(par_done * args.blob_size / par_t / 1e6, 3), "unit": "MB/s", "cores": nthr,
                                    "kind": "port", "value_1thread": round(one, 3),
                                    "sample": f"blobs of the same batch, one-shot Compress, C restatement of the reference "
                                              f"algorithm: {done} blobs on 1 thread, then {par_done} blobs on {nthr} threads "
                                              f"(one blob per thread at a time; host has {os.cpu_count()} logical CPUs)"}
            line["bit_exact_vs_oracle"] = {"blobs_checked": done + par_done, "all_equal": bool(exact and all(flags))}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
