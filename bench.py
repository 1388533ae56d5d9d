#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the flank-discovery hot path (findflanks + pairflanks) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]

A step is one pass of the hot path over one resident batch (``mfk_flankpairs``): ``mfk_findflanks`` (read scan -> sites -> filters -> site x read
join -> consensus) followed by ``mfk_pairflanks`` (window join, inverted repeats, assignment, spanning reads, direct
repeats) on the packed alignments that already sit in HBM.  The workload is synthetic simulated-insertion alignments
(mustache_amd/synth.py); the default is BASELINE.json configs[1], "S. aureus single isolate at 1000x" (2.8 Mb, 2x150 bp,
18.7 M reads, 20 insertions).  With N > 1 every rank holds its own isolate of that size (sample sharding, weak scaling)
and the per-rank flank tables are merged by one all-gather (RCCL) inside the step.

Prints ONE JSON line (rank 0): whole-job reads/s, the roofline of the dominant kernel (HIP-event timing inside the library,
on the library's stream) and, at N = 1, the CPU port (oracle/mfk_oracle.c) timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches


def algorithmic_bytes(kernel, meta):
    """Bytes one launch of `kernel` has to move for this workload (DESIGN.md "Kernels"); None = no streaming model."""
    n, listed, ev = meta["n_reads"], meta["clipped_reads"], meta["events"]
    if kernel.startswith("k_scan_clips"):
        return n * 1 + listed * 8              # flags of every read + one 8-byte list entry per lenient-clipped read
    if kernel.startswith("k_clip_events"):
        return listed * (8 + 7) + ev * 8       # list entry + mapq, contig_id, one position column (7 B) + one key per event
    if kernel == "k_classify":
        return meta["join_visits"] * 13        # ref_start, ref_end (8 B) + mapq, flags, link (5 B) per (site, read) visit
    if kernel.startswith("rocprim::radix_sort_keys") or kernel == "k_bucket_sort":
        return ev * 8 * 2                      # a sort has to read and write every key once
    if kernel == "k_direct_repeats":
        # (150 x 1.5 + 8) B per read over a pair's 50-bp window; a window of W bp at coverage c sees c (150 + W) / 150 reads
        return int(meta["pairs"] * meta["coverage"] * (150 + 50) / 150 * (150 * 1.5 + 8))
    if kernel == "k_event_info":
        return ev * (8 + 4 + 2 + 1 + 4)        # key, link, contig_id, info out, head out
    return None


def pmc_traffic(kernel, config):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json, made by
    scripts/profile_round.sh + scripts/pmc_table.py from this same command), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            tab = json.load(fh)
        return tab.get(config, {}).get(kernel, {}).get("traffic_bytes")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2_saureus_1000x")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-insertions", type=int, default=20)
    ap.add_argument("--dump-profile", default=None, help="write the per-kernel table to this file")
    ap.add_argument("--isolates", type=int, default=12, help="isolates per GPU of --config C4_ecoli_batch (96 over 8 GPUs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    from mustache_amd import lib, pairflanks as pfh
    from mustache_amd.dist import allgather_flanks
    from mustache_amd.findflanks import device_reads
    from mustache_amd.synth import CONFIGS, synth_packed

    t0 = time.time()
    batch = None
    if args.config == "C4_ecoli_batch":
        # BASELINE configs[3]: a batch of E. coli isolates per GPU (96 over the 8 GPUs of a node), run as ONE call per stage
        # (mustache_amd/cohort.py); the per-rank candidate-site tables are merged by the all-gather
        from mustache_amd import cohort
        cfg = dict(CONFIGS["C1_ecoli_100x"])
        packs, genomes, truths = [], [], []
        for i in range(args.isolates):
            p_, g_, t_ = synth_packed(**cfg, seed=1000 * (rank + 1) + i)
            packs.append(p_); genomes.append(g_); truths.append(t_)
        batch = cohort.concat_samples(packs, ["isolate%d" % (rank * args.isolates + i) for i in range(args.isolates)])
        packed = batch.packed
        truth = [t for ts in truths for t in ts]
        g, goff = cohort.batch_genome(batch, genomes)
        ranks = cohort.batch_contig_ranks(batch)
        cpu_args = (packs[0], genomes[0], truths[0])
    else:
        cfg = dict(CONFIGS[args.config])
        packed, genome, truth = synth_packed(**cfg, seed=100 + rank)
        g, goff = pfh.genome_arrays(genome, packed.contig_names)
        ranks = pfh.contig_ranks(packed.contig_names)
        cpu_args = (packed, genome, truth)
    gen_s = time.time() - t0
    hip = lib.load(local_rank)
    params = lib.make_params()
    reads = device_reads(hip, packed, device)
    g_dev = torch.from_numpy(g).to(device)

    def step():
        ff, pf = hip.run_flankpairs(reads, params, ranks, g_dev, goff)   # mfk_findflanks then mfk_pairflanks inside one library call
        if batch is not None:
            per_sample = cohort.split_flanks(batch, ff)          # per-isolate tables, as the per-sample runs give them
            if world > 1:
                cohort.cohort_table(per_sample, sample_numbers=[rank * args.isolates + i for i in range(args.isolates)], device=device)
        elif world > 1:
            allgather_flanks(ff["flanks"], ff["seqs"], device)
        return ff, pf

    for _ in range(args.warmup):
        ff, pf = step()
    # three more untimed steps with every kernel bracketed: the per-kernel table and the ranking of the kernels
    hip.profile(True)
    warm_calls = 3
    for _ in range(warm_calls):
        ff, pf = step()
    full = hip.profile_totals()
    scan_kernels = ["k_scan_clips", "k_clip_events_bkt", "k_clip_events"]   # _bkt: bucketed path; plain: list + rocPRIM sort fallback
    watched = list(scan_kernels)
    if full:
        watched.insert(0, max(full.items(), key=lambda kv: kv[1][0])[0])
    else:                                   # --warmup 0: no ranking yet, watch the usual suspects
        watched = ["k_direct_repeats"] + watched

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events only around the watched kernels (two event records per launch are not free)
    hip.profile_only(watched)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ff, pf = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = hip.profile_totals()
    hip.profile_only(None)
    hip.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nr = torch.tensor([packed.n_reads], dtype=torch.int64, device=device)
        dist.all_reduce(nr)
        total_reads = int(nr.item())
    else:
        total_reads = packed.n_reads

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed
        flags = packed.flags
        meta = {"n_reads": packed.n_reads, "clipped_reads": int(((flags & 15) != 0).sum()),
                "events": int(ff["stats"][10]),
                "join_visits": int(ff["stats"][3]) * 3 * int(cfg["coverage"]),
                "pairs": int(pf["stats"][1]), "coverage": int(cfg["coverage"])}
        # per-kernel rows (name, avg ms per launch, launches, total ms).  Shares and the device total come from the fully
        # bracketed warm-up steps; the durations of the watched kernels come from the timed region.
        full_table = sorted(((nm, ms / max(calls, 1), calls, ms) for nm, (ms, calls) in full.items()), key=lambda r: -r[3])
        full_tot = max(sum(r[3] for r in full_table), 1e-12)
        share = {r[0]: r[3] / full_tot for r in full_table}
        dev_ms = full_tot / warm_calls if full_table else None
        table = sorted(((nm, ms / max(calls, 1), calls, ms) for nm, (ms, calls) in prof.items()), key=lambda r: -share.get(r[0], 0.0))
        roofline = None
        if table:
            def entry(names):
                rows = [r for r in table if r[0] in names]
                if not rows:
                    return None
                ms = sum(r[1] for r in rows)                       # one launch of each per step
                ab = sum(algorithmic_bytes(r[0], meta) or 0 for r in rows)
                tr = [pmc_traffic(r[0], args.config) for r in rows]
                e = {"kernel": "+".join(r[0] for r in rows), "bound": "hbm", "avg_ms": round(ms, 5),
                     "share_of_device_time": round(sum(share.get(r[0], 0.0) for r in rows), 3), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                if ab:
                    ach = ab / (ms * 1e-3) / 1e9
                    e.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": ab})
                else:
                    e.update({"achieved": None, "frac": None})
                e["traffic"] = sum(tr) if all(t is not None for t in tr) else None
                return e

            # The roofline object is that of the read scan: the two kernels that process the metric's unit (every read) and
            # carry the path's algorithmic bytes.  By TIME no kernel dominates this path at C2: the largest is ~10 % of the
            # summed kernel time, the step is a chain of ~70 short launches on two streams, and the largest ones are
            # per-site / per-pair latency chains (bit-vector edit distance, direct-repeat pile-up, inverted-repeat
            # alignment) on a few dozen workgroups; `dominant_by_time` reports that kernel with a byte model where one exists.
            roofline = entry(scan_kernels)
            roofline["dominant_by_time"] = entry([table[0][0]])
            roofline["top5_by_time"] = [{"kernel": r[0], "avg_ms": round(r[1], 5), "launches_per_step": r[2] // warm_calls,
                                         "share": round(share[r[0]], 3)} for r in full_table[:5]]
        if args.dump_profile:
            with open(args.dump_profile, "w") as fh:
                fh.write("# per-kernel HIP-event timing, every launch bracketed, over %d untimed steps after the warm-up (config %s)\n" % (warm_calls, args.config))
                fh.write("%-40s %10s %8s %12s\n" % ("kernel", "avg_ms", "calls", "total_ms"))
                for nm, avg_ms, calls, tot in full_table:
                    fh.write("%-40s %10.5f %8d %12.4f\n" % (nm, avg_ms, calls, tot))
                fh.write("# watched kernels inside the %d timed steps\n" % args.steps)
                for nm, avg_ms, calls, tot in table:
                    fh.write("%-40s %10.5f %8d %12.4f\n" % (nm, avg_ms, calls, tot))
                fh.write("device ms per step (sum of kernels, bracketed steps) %.4f, wall ms per step (timed) %.4f\n" % (dev_ms or 0.0, ms_per_step))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cpu_args[0], cpu_args[1], cpu_args[2], params, args.cpu_sample_insertions)
        out = {"metric": "aligned reads/sec (findflanks+pairflanks, whole node)", "value": round(value, 1), "unit": "reads/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (+f64 ratio tests)",
               "data": "synthetic",
               "config": {"workload": args.config, "reads_per_gpu": packed.n_reads, "genome_bp": int(sum(cfg["contig_lens"])) * (args.isolates if batch is not None else 1),
                          "isolates_per_gpu": args.isolates if batch is not None else 1,
                          "coverage": cfg["coverage"], "read_length": 150, "insertions": len(truth),
                          "flank_rows": int(len(ff["flanks"])), "pairs": int(len(pf["pairs"])), "sharding": "sample per GPU",
                          "device_ms_per_step": round(dev_ms, 4) if dev_ms else None, "generate_s": round(gen_s, 1)},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(packed, genome, truth, params, n_insertions):
    """The CPU port (oracle) on a bounded sample: the reads of the first `n_insertions` insertion sites' stretch of the
    genome.  1 core, like the reference (single-threaded Python)."""
    import numpy as np
    from mustache_amd import pairflanks as pfh
    from oracle.oracle import Oracle
    ora = Oracle()
    pos = sorted(t[1] for t in truth if t[0] == packed.contig_names[0])
    cut = pos[min(n_insertions, len(pos)) - 1] + 2000 if pos else int(packed.contig_len[0])
    hi = int(np.searchsorted(packed.ref_start[:int(packed.contig_read_start[1])], cut))
    sub = packed.slice_reads(0, hi)
    R = ora.reads(sub)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    t0 = time.perf_counter()
    ff = ora.run_findflanks(R, params)
    ora.run_pairflanks(R, params, ff["flanks"], ff["seqs"], pfh.contig_ranks(packed.contig_names), g, goff)
    dt = time.perf_counter() - t0
    return {"value": round(sub.n_reads / dt, 1), "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads (%d insertion sites, %.1f%% of the workload), findflanks+pairflanks in %.2f s"
                      % (sub.n_reads, min(n_insertions, len(pos)), 100.0 * sub.n_reads / packed.n_reads, dt)}


if __name__ == "__main__":
    main()
