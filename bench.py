#!/usr/bin/env python3
"""bench.py - compress MB/s of the MI355X lzss path on BASELINE.json's config 3 workload.

A step = one pass of the hot path (suffix-array index build, per-position back-reference search, phrase
walk, bit emission) over config 3's batch: 10 000 synthetic 128 KiB calldata-like blobs that share a 64 KiB
dictionary (1.31 GB; ~52 GB of index arrays: it fits one MI355X).  At N = 1 a step is that whole batch on one GPU; at
N > 1 the SAME 10 000 blobs are sharded per blob (blob b -> rank b mod N, 10 000 / N per GPU, no collective on the data
path): `"scaling": "strong"`, which is config 3 as BASELINE.json states it.  The weak figure (every rank a batch of
10 000 blobs of its own, ids rank + N k) is measured after it and reported under `weak_scaling`.
`value`: inputs are resident in HBM before the timed region, outputs stay in HBM (the task contract's definition).
`transfer_inclusive`: the same blobs from pinned host buffers through the library's chunk pipeline, host to host, averaged
over --steps calls - SURVEY.md 8(d)'s "incl. H2D/D2H" figure.
The blobs come from the C twin of compress_amd/synth.py (no forked workers: safe under rocprofv3).

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
        if k == "lzss::sa_meta_kernel":      # the run-compressed suffix sort (round 3) is the sa_build stage's kernel
            k = "lzss::sa_build_kernel"
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
    ap.add_argument("--blobs", type=int, default=10000, help="blobs of the batch (config 3: 10 000), sharded b mod N over the ranks")
    ap.add_argument("--blob-size", type=int, default=131072)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming (config 4) leg")
    ap.add_argument("--stream-bytes", type=int, default=2 << 20, help="bytes of the streaming (config 4) leg; config 4 as "
                    "stated is 1 GiB (--stream-bytes 1073741824, ~1 min)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-blob (config 2) latency sample")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaling leg (a whole batch per rank)")
    ap.add_argument("--overlap", type=int, default=2, help="concurrent sub-batches of the index / match stages")
    ap.add_argument("--stage-steps", type=int, default=2, help="extra untimed steps (one part) for the per-kernel times")
    ap.add_argument("--pipe-steps", type=int, default=-1, help="calls of mi355_lzss_compress_batch (host buffers in and out) "
                    "the transfer-inclusive figure is averaged over; -1 = --steps, 0 = skip")
    ap.add_argument("--chunk-blobs", type=int, default=1536, help="blobs per chunk of that pipeline")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's parallel leg (0 = min(16, cores))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pipe_steps = args.steps if args.pipe_steps < 0 else args.pipe_steps

    # ---- workload: generated in this process by the C generator (threads, no fork) before anything touches the GPU.
    #      This rank's blobs are ids rank, rank + N, ...: the first `n_strong` of them are its shard of config 3's one
    #      batch (ids below --blobs), all `n_weak` of them its own whole batch for the weak-scaling leg.
    import numpy as np
    from compress_amd import shard, synth
    dict_ = synth.dictionary()
    n_strong = len(shard.batch_shard_ids(rank, world, args.blobs))
    do_weak = world > 1 and not args.no_weak
    n_weak = args.blobs if do_weak else n_strong
    t_gen = time.perf_counter()
    arr = synth.blobs_array(rank, world, n_weak, args.blob_size)       # (n_weak, blob_size) uint8
    t_gen = time.perf_counter() - t_gen
    flat_all = arr.reshape(-1)

    def csr(n):
        return flat_all[:n * args.blob_size], (np.arange(n + 1, dtype=np.uint64) * np.uint64(args.blob_size))

    in_bytes = n_strong * args.blob_size

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
    bt = B.Batch(ctx, max(n_weak, 1), max(n_weak * args.blob_size, 1))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    parts = int(os.environ.get("MI355_LZSS_SPLIT", args.overlap))

    def timed_steps(n_blobs):
        """host -> HBM outside the timed region, W untimed steps, then exactly K steps between barriers; the MAX over
        ranks of the elapsed time"""
        t0 = time.perf_counter()
        bt.upload_flat(*csr(n_blobs))
        t_up_ = time.perf_counter() - t0
        ctx.set_overlap(parts)      # the production setting: two sub-batches whose stages overlap on concurrent streams
        for _ in range(args.warmup):
            bt.compress()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            bt.compress()                  # returns after the library's stream has drained
        barrier()
        return shard.max_over_ranks(time.perf_counter() - t0, device=red_dev), t_up_

    # ---- weak-scaling leg first (N > 1 only), so that the batch left resident afterwards is the strong one
    weak = None
    if do_weak:
        el_w, _ = timed_steps(n_weak)
        weak = {"value": round(n_weak * args.blob_size * world * args.steps / el_w / 1e6, 2), "unit": "MB/s",
                "ms_per_step": round(el_w / args.steps * 1e3, 3), "blobs_per_gpu": n_weak,
                "workload": "every rank a batch of %d blobs of its own (ids rank + N k)" % n_weak}

    # ---- the timed region: config 3's one batch, this rank's shard
    elapsed, t_up = timed_steps(n_strong)
    total_in_bytes = int(round(shard.sum_over_ranks(float(in_bytes), device=red_dev)))
    # per-kernel stage times need the stages back to back: --stage-steps extra steps with one part, untimed
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
    blobs = [arr[k].tobytes() for k in range(n_strong)]

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

    # ---- transfer-inclusive (SURVEY.md 8d "incl. H2D/D2H"): the same blobs from pinned HOST buffers through
    #      mi355_lzss_compress_batch, which pipelines chunks of 1536 blobs (upload of chunk k+1 and download of chunk k-1
    #      overlap the kernels of chunk k); one warm-up call, then --pipe-steps calls between barriers, each timed around
    #      the whole call with the outputs landing in the caller's CSR layout on the host
    pipe = None
    if pipe_steps > 0 and n_strong > 0:
        import ctypes as C
        from compress_amd import _lib
        nb = n_strong
        pin_in = compress_amd.PinnedBuffer(in_bytes)
        pin_in.array[:in_bytes] = flat_all[:in_bytes]
        in_off = np.arange(nb + 1, dtype=np.uint64) * np.uint64(args.blob_size)
        caps = np.array([(int(x) + 15) // 16 * 16 + 16 for x in out_len], dtype=np.uint64)  # the known sizes + slack
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
        for _ in range(pipe_steps):
            run_pipe()
        barrier()
        t_pipe = shard.max_over_ranks(time.perf_counter() - t0, device=red_dev)
        same = all(pin_out.array[int(out_off[k]):int(out_off[k]) + int(p_len[k])].tobytes() == outs[k]
                   for k in range(0, nb, max(1, nb // 97)))
        pipe = {"value": round(total_in_bytes * pipe_steps / t_pipe / 1e6, 2), "unit": "MB/s", "calls": pipe_steps,
                "ms_per_call": round(t_pipe / pipe_steps * 1e3, 3), "blobs_per_gpu": nb,
                "chunk_blobs": args.chunk_blobs, "host_buffers": "pinned (mi355_lzss_host_alloc)",
                "equal_to_resident_outputs_on_sample": bool(same),
                "note": "whole call of mi355_lzss_compress_batch: H2D + kernels + D2H + scatter into the caller's CSR "
                        "layout, averaged over `calls` calls after one warm-up call"}
        pin_in.close()
        pin_out.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_in_bytes * args.steps / elapsed / 1e6
        stages = {k: v / args.stage_steps for k, v in stage_sum.items() if v > 0}
        dom = max(stages, key=stages.get)
        # algorithmic bytes of one launch on this GPU: sum(n_in + n_out) + n_dict
        alg_bytes = in_bytes + int(out_total) + len(compress_amd.AugmentDict(dict_))
        achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9
        step_rate = alg_bytes / (ms_per_step * 1e-3) / 1e9
        traffic, fetch_rep, write_rep, traffic_src = pmc_traffic(dom, n_strong)
        line = {
            "metric": "compress MB/s (whole node, bit-exact) + % HBM roofline at 1/2/4/8 GPUs",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "config3: ONE batch of %d independent 128 KiB calldata-like blobs, shared 64 KiB dictionary, "
                                   "sharded per blob over the GPUs (blob b -> rank b mod N)" % args.blobs,
                       "blobs_total": args.blobs, "blobs_per_gpu": n_strong, "blob_bytes": args.blob_size,
                       "dict_bytes": len(dict_), "ratio": round(in_bytes / max(1, int(out_total)), 3),
                       "overlap_parts": parts, "parallelism": "blob-sharded x%d, no collective" % world},
            "value_definition": "inputs resident in HBM when the timed region starts, outputs left in HBM (kernel sequence "
                                "only); transfer_inclusive.value is the same batch host to host (SURVEY.md 8d)",
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_fetch_size_reported": fetch_rep,
                         "traffic_write_size": write_rep, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stages[dom], 3),
                         "step_frac": round(step_rate / HBM_PEAK_GBPS, 6),
                         "step_frac_note": "algorithmic bytes of this GPU's launch / ms_per_step / peak: the whole kernel "
                                           "sequence against the roofline, not the dominant kernel alone"},
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "stages_note": "HIP-event time of every stage with the stages back to back (overlap 1), measured on extra "
                           "steps outside the timed region; their sum exceeds ms_per_step by what the overlap hides",
            "decompress": {"value": round(total_in_bytes * args.steps / dec_elapsed / 1e6, 2), "unit": "MB/s (decompressed bytes)",
                           "kernel_ms": round(dec_ms / args.steps, 3), "roundtrip_equal": dec_ok,
                           "achieved_GBps": round((in_bytes + int(out_total)) / (dec_ms / args.steps * 1e-3) / 1e9, 3)},
            "kernel_only_MBps": round(value, 2),
            "transfer_inclusive": pipe,
            "transfer_inclusive_frac_of_resident": round(pipe["value"] / value, 3) if pipe else None,
            "serial_upload_compress_download_MBps": round(in_bytes / (t_up + elapsed / args.steps + t_down) / 1e6, 2),
            "weak_scaling": weak,
            "generate_s": round(t_gen, 2),
        }
        # ---- config 2: latency of ONE 128 KiB blob with the 64 KiB dictionary through Compress (host buffers in and out)
        if not args.no_latency:
            comp1 = compress_amd.Compressor(dict_, device=local_rank)
            comp1.Compress(blobs[0])
            lat = []
            for _ in range(7):
                t0 = time.perf_counter()
                o1 = comp1.Compress(blobs[0])
                lat.append(time.perf_counter() - t0)
            line["single_blob_latency_ms"] = {"workload": "config2: synthetic blob #%d (131072 B), 64 KiB dictionary, Compress()" % rank,
                                              "median": round(sorted(lat)[len(lat) // 2] * 1e3, 3), "min": round(min(lat) * 1e3, 3),
                                              "equal_to_batch_output": o1 == outs[0]}
        # ---- config 4 (streaming): the Write / Revert loop of compress_test.go:317-363 (chunks of 1000 B into
        #      blobs of <= 126 976 B) through the speculative packer, every blob decompressed again on the GPU
        if not args.no_stream:
            from compress_amd import stream as S
            n_sb = (args.stream_bytes + args.blob_size - 1) // args.blob_size
            sdata = synth.blobs_array(0, 1, n_sb, args.blob_size).reshape(-1)[:args.stream_bytes].tobytes()  # blobs #0, #1, ...
            comp = compress_amd.Compressor(dict_, device=local_rank)
            S.pack_stream_native(comp, sdata[:600000], 1000, 126976)    # warm-up: the pooled workspace
            t0 = time.perf_counter()
            spec = S.pack_stream_native(comp, sdata, 1000, 126976)
            t_spec = time.perf_counter() - t0
            # round trip: every blob through the GPU decompressor, spans compared with the stream
            rt_ok, pos = True, 0
            for g0 in range(0, len(spec), 256):
                grp = spec[g0:g0 + 256]
                dec = compress_amd.decompress_batch(comp, [b_ for b_, _ in grp], max(c_ for _, c_ in grp) + 64)
                for (b_, c_), d_ in zip(grp, dec):
                    rt_ok = rt_ok and (not isinstance(d_, Exception)) and d_ == sdata[pos:pos + c_] and len(b_) <= 126976
                    pos += c_
            rt_ok = rt_ok and pos == len(sdata)
            seq_n = min(300000, len(sdata))
            t0 = time.perf_counter()
            seq = S.pack_stream_sequential(comp, sdata[:seq_n], 1000, 126976)
            t_seq = time.perf_counter() - t0
            line["stream"] = {"workload": "config4: %d B of the synthetic stream (blobs #0, #1, ... back to back), Writes of "
                                          "1000 B into blobs of <= 126976 B (the loop of compress_test.go:317-363)" % len(sdata),
                              "speculative_MBps": round(len(sdata) / t_spec / 1e6, 2), "speculative_bytes": len(sdata),
                              "seconds": round(t_spec, 2), "blobs": len(spec),
                              "compressed_bytes": sum(len(b_) for b_, _ in spec), "roundtrip_equal": bool(rt_ok),
                              "sequential_MBps": round(seq_n / t_seq / 1e6, 3), "sequential_bytes": seq_n,
                              "equal_on_sequential_sample": S.pack_stream_native(comp, sdata[:seq_n], 1000, 126976) == seq}
        # ---- bit-exactness spot check + CPU baseline (oracle = checker / reported baseline only), rank 0 at every N
        if not args.no_cpu:
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
                                    "sample": f"blobs of the same batch (rank 0's shard), one-shot Compress, C restatement of the "
                                              f"reference algorithm: {done} blobs on 1 thread, then {par_done} blobs on {nthr} "
                                              f"threads (one blob per thread at a time; host has {os.cpu_count()} logical CPUs)"}
            line["bit_exact_vs_oracle"] = {"blobs_checked": done + par_done, "all_equal": bool(exact and all(flags))}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
