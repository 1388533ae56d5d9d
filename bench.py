#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the flank-discovery hot path (findflanks + pairflanks) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--sharding sample|coordinate|contig]

A step is one pass of the hot path over one resident batch: ``mfk_findflanks`` (read scan -> sites -> filters -> site x
read join -> consensus) followed by ``mfk_pairflanks`` (window join, inverted repeats, assignment, spanning reads, direct
repeats) on the packed alignments that already sit in HBM.  The workload is synthetic simulated-insertion alignments
(mustache_amd/synth.py, every read carrying its bases: nothing in the input uses the simulated truth); the default is
BASELINE.json configs[1], "S. aureus single isolate at 1000x" (2.8 Mb, 2x150 bp, 18.7 M reads, 20 insertions).

--sharding sample      (default) N > 1: every rank holds its own isolate (weak scaling; BASELINE configs[3] with
                       --config C4_ecoli_batch: --isolates per GPU in one call); one all-gather of the flank tables per step
--sharding coordinate  the ONE data set of --config cut into N read-balanced coordinate ranges with a halo (strong
                       scaling; BASELINE configs[4] with --config C5_meta_50mb); mustache_amd/dist.py
--sharding contig      whole contigs bin-packed over the ranks (strong scaling; BASELINE configs[2], C3_klebsiella)

Prints ONE JSON line (rank 0): whole-job reads/s; `roofline` of the kernel that takes most device time (durations from
HIP events inside the library on the stream of each launch, inside the timed region); and at N = 1 also
`cpu_baseline` (the CPU port oracle/mfk_oracle.c on one core and on all cores, on a bounded sample), `parity_in_run`
(sha256 of both TSVs, GPU vs CPU port, on that sample), `configs` (ms/step of the other BASELINE configs) and
`end_to_end` (BAM file + FASTA on disk -> both TSVs through the drop-in stage drivers, host decode and PCIe included).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches
# vector-ALU issue peak: 256 CUs x 4 SIMDs x 32 lanes per cycle x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32 units, a 64-lane
# wave issues over 2 cycles; = half the 157.3 TFLOP/s fp32 FMA figure) in 10^12 32-bit lane-operations per second
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# VALU instructions one wave issues per unit of work, from the rocprofv3 SQ_INSTS_VALU pass of the same command
# (profiles/r02_pmc_valu.json, scripts/pmc_table.py) divided by the work counter the kernel keeps (stats[12])
VALU_MODELS_FILE = os.path.join(ROOT, "profiles", "r03_valu_models.json")
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
# Both tables come from files (a PMC pass cannot run inside a timed run): they are used only when the library they were measured
# on is the library that runs now (sha256 recorded by scripts/pmc_table.py); otherwise the line says so and carries no number.


def lib_sha16():
    from mustache_amd import lib
    try:
        with open(lib.LIB_PATH, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def from_file(path, config):
    """(table of one config, provenance string) of a PMC-derived file; an empty table when it was measured on another build."""
    tab = _json_file(path).get(config, {})
    meta = tab.get("_meta", {}) if isinstance(tab, dict) else {}
    name = os.path.relpath(path, ROOT)
    if not tab:
        return {}, "no PMC table for %s in %s" % (config, name)
    if meta.get("lib_sha256_16") != lib_sha16():
        return {}, "STALE, not used: %s was measured on library %s, this run uses %s (run scripts/profile_round.sh)" % (name, meta.get("lib_sha256_16"), lib_sha16())
    return tab, "from_file %s, measured on this library build (sha256 %s): %s" % (name, meta.get("lib_sha256_16"), meta.get("what", ""))
SCAN_KERNELS = ["k_scan_clips", "k_clip_events_bkt", "k_clip_events"]   # _bkt: bucketed path; plain: list + rocPRIM sort fallback


def algorithmic_bytes(kernel, meta):
    """Bytes one launch of `kernel` has to move for this workload (DESIGN.md "Kernels"); None = no streaming model."""
    n, listed, ev = meta["n_reads"], meta["clipped_reads"], meta["events"]
    if kernel.startswith("k_scan_clips"):
        return n * 1 + listed * 8              # flags of every read + one 8-byte list entry per lenient-clipped read
    if kernel.startswith("k_clip_events"):
        return listed * (8 + 16) + ev * 8      # list entry + the read's dense 16-byte clip record (device_reads attaches them) + one key per event
    if kernel == "k_classify":
        return meta["join_visits"] * 13        # ref_start, ref_end (8 B) + mapq, flags, link (5 B) per (site, read) visit (estimated count)
    if kernel == "k_bucket_sort":
        return ev * (8 * 2 + 2 + 1 + 4)        # every key read and written once + CIGAR-end length in, info byte and head flag out (per-event information is written here)
    if kernel.startswith("rocprim::radix_sort_keys"):
        return ev * 8 * 2                      # a sort has to read and write every key once
    if kernel.startswith("k_direct_repeats"):
        # (150 x 1.5 + 8) B per read over a pair's 50-bp window; a window of W bp at coverage c sees c (150 + W) / 150 reads
        return int(meta["pairs"] * meta["coverage"] * (150 + 50) / 150 * (150 * 1.5 + 8))
    if kernel == "k_event_info":
        return ev * (8 + 4 + 2 + 1 + 4)        # key, link, contig_id, info out, head out
    return None


def _json_file(path):
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def kernel_entry(names, table, share, meta, config):
    """Roofline object of one kernel (or of kernels that run once each per step, summed): HBM model where the kernel
    streams, VALU model where it is an integer recurrence.  Work (bytes / operations) and time are both per step; a kernel
    launched several times per step reports the sum of its launches against the sum of their durations."""
    rows = [r for r in table if r[0] in names]
    if not rows:
        return None
    steps = max(meta["timed_steps"], 1)
    ms = sum(r[3] for r in rows) / steps                                    # device ms per step of these kernels
    launches = max(int(round(sum(r[2] for r in rows) / steps)), 1)
    e = {"kernel": "+".join(r[0] for r in rows), "avg_ms": round(ms / launches, 5), "launches_per_step": launches,
         "share_of_device_time": round(sum(share.get(r[0], 0.0) for r in rows), 3)}
    ab = [algorithmic_bytes(r[0], meta) for r in rows]
    traffic, traffic_src = from_file(PMC_TRAFFIC_FILE, config)
    def pmc(name):      # the PMC table spells every template argument ("k_direct_repeats<0, 384>"), the event table the first
        hit = traffic.get(name) or next((v for k, v in traffic.items() if name.endswith(">") and k.startswith(name[:-1] + ",")), {})
        return hit.get("traffic_bytes")
    tr = [pmc(r[0]) for r in rows]
    valu, valu_src = from_file(VALU_MODELS_FILE, config)
    m = valu.get(rows[0][0]) if len(rows) == 1 else None   # per config: the cost per unit depends on the data shape
    tr_sum = sum(tr) if all(t is not None for t in tr) else None
    if all(a is not None for a in ab):
        ach = sum(ab) / (ms * 1e-3) / 1e9
        e.update({"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                  "algorithmic_bytes_per_step": sum(ab),
                  "traffic": tr_sum,
                  "traffic_source": traffic_src})
        if any(r[0] == "k_classify" for r in rows):
            e["note"] = "algorithmic bytes from an ESTIMATED number of (site, read) visits: sites after the first neighbour filter x 3 positions x coverage"
    elif m and meta["work"].get(m["work_counter"]):
        units = meta["work"][m["work_counter"]]                               # per step, counted in this run
        ops = units * m["valu_insts_per_unit"] * 64.0                         # lane-operations: 64 lanes per wave instruction
        ach = ops / (ms * 1e-3) / 1e12
        e.update({"bound": "valu", "achieved": round(ach, 3), "peak": round(VALU_PEAK_TOPS, 1), "unit": "Tlane-op/s",
                  "frac": round(ach / VALU_PEAK_TOPS, 5), "work_units_per_step": units, "work_unit": m["unit"],
                  "valu_insts_per_unit": m["valu_insts_per_unit"], "valu_model_source": valu_src, "model_detail": m.get("source"), "traffic": tr_sum,
                  "traffic_source": traffic_src, "note": m.get("note")})
    else:
        e.update({"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_TOPS, 1), "unit": "Tlane-op/s", "frac": None, "traffic": tr_sum,
                  "traffic_source": traffic_src, "valu_model_source": valu_src})
    return e


def sha(text):
    return hashlib.sha256(text.encode() if isinstance(text, str) else bytes(text)).hexdigest()


def tsv_texts(ff, pf, contig_names):
    """Both result tables as the TSV text the stage drivers write (sample column = 'bench')."""
    from mustache_amd.findflanks import flank_frame
    from mustache_amd.pairflanks import pair_frame
    a = flank_frame(ff, contig_names)
    a.insert(0, "flank_id", list(a.index + 1)); a.insert(0, "sample", "bench")
    b = pair_frame(pf, ff["flanks"], ff["seqs"], contig_names)
    b.insert(0, "sample", "bench")
    return a.to_csv(sep="\t", index=False), b.to_csv(sep="\t", index=False)


# ---- CPU legs (before anything touches the GPU: the all-cores leg forks) ------------------------------------------------
_cpu = {}


def _cpu_stage1(k):
    from mustache_amd import dist as md
    plan = _cpu["plans"][k]
    rd = _cpu["ora"].reads(_cpu["sample"].take_ranges(plan.ranges))
    _cpu["rd"] = rd                      # a worker keeps its shard for stage 2 (imap with chunksize 1 and P workers: one shard each)
    _cpu["k"] = k
    return md.shard_findflanks(_cpu["ora"], rd, plan, _cpu["params"])


def _cpu_stage2(args):
    from mustache_amd import dist as md
    k, tab, seqs = args
    plan = _cpu["plans"][k]
    rd = _cpu["rd"] if _cpu.get("k") == k else _cpu["ora"].reads(_cpu["sample"].take_ranges(plan.ranges))
    return md.shard_pairflanks(_cpu["ora"], rd, plan, _cpu["params"], tab, seqs, _cpu["ranks"], _cpu["g"], _cpu["goff"])


def cpu_legs(packed, genome, truth, params, n_insertions, cores):
    """The CPU port (oracle) on a bounded sample: the reads of the first `n_insertions` insertion sites' stretch of the
    genome.  One core like the reference (single-threaded Python), then all cores over coordinate shards of the sample
    (the same shard plans and merges the GPUs use).  Returns (cpu_baseline object, sample, TSV hashes of the CPU result)."""
    import multiprocessing as mp
    import numpy as np
    from mustache_amd import dist as md, pairflanks as pfh
    from oracle.oracle import Oracle
    ora = Oracle()
    pos = sorted(t[1] for t in truth if t[0] == packed.contig_names[0])
    cut = pos[min(n_insertions, len(pos)) - 1] + 2000 if pos else int(packed.contig_len[0])
    hi = int(np.searchsorted(packed.ref_start[:int(packed.contig_read_start[1])], cut))
    sub = packed.slice_reads(0, hi)
    R = ora.reads(sub)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    t0 = time.perf_counter()
    ff = ora.run_findflanks(R, params)
    pf = ora.run_pairflanks(R, params, ff["flanks"], ff["seqs"], ranks, g, goff)
    dt1 = time.perf_counter() - t0
    texts = tsv_texts(ff, pf, packed.contig_names)
    out = {"value": round(sub.n_reads / dt1, 1), "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": "first %d reads (%d insertion sites, %.1f%% of the workload), findflanks+pairflanks in %.2f s"
                     % (sub.n_reads, min(n_insertions, len(pos)), 100.0 * sub.n_reads / packed.n_reads, dt1)}
    if cores > 1:
        h = md.halo(params.min_distance_to_mate, params.max_direct_repeat_length)
        md._tables()      # the table loops of the sharded path (host code of libmfk.so): opened once here, inherited by the workers
        _cpu.update(ora=ora, sample=sub, params=params, ranks=ranks, g=g, goff=goff, plans=md.plan_shards(sub, cores, "coordinate", h))
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            tab, seqs = md.merge_flanks(list(pool.imap(_cpu_stage1, range(cores), chunksize=1)))
            pairs, strs = md.merge_pairs(list(pool.imap(_cpu_stage2, [(k, tab, seqs) for k in range(cores)], chunksize=1)), ranks)
        dtn = time.perf_counter() - t0
        wp, ws = md._compact(pf["pairs"], pf["strs"], md._PAIR_STR)
        same = tab.tobytes() == ff["flanks"].tobytes() and seqs == ff["seqs"] and pairs.tobytes() == wp.tobytes() and strs == ws
        out["all_cores"] = {"value": round(sub.n_reads / dtn, 1), "unit": "reads/s", "cores": cores, "seconds": round(dtn, 2),
                            "sharding": "coordinate ranges of the sample, one process per core (fork), slicing and merges included",
                            "tables_equal_one_core": bool(same)}
        _cpu.clear()
    return out, sub, {"findflanks_tsv_sha256": sha(texts[0]), "pairflanks_tsv_sha256": sha(texts[1]),
                      "flank_rows": int(len(ff["flanks"])), "pairs": int(len(pf["pairs"]))}


# ---- other configs, end to end -----------------------------------------------------------------------------------------
def make_workload(name, seed, pool, isolates=12, rank=0):
    """(packed, genome bytes, genome offsets, contig ranks, cfg, truth, batch or None, per-isolate inputs for the CPU leg)."""
    from mustache_amd import cohort, pairflanks as pfh
    from mustache_amd.synth import CONFIGS, synth_packed
    if name == "C4_ecoli_batch":
        # BASELINE configs[3]: a batch of E. coli isolates per GPU (96 over the 8 GPUs of a node), run as ONE call per stage
        cfg = dict(CONFIGS["C1_ecoli_100x"])
        packs, genomes, truths = [], [], []
        for i in range(isolates):
            p_, g_, t_ = synth_packed(**cfg, seed=1000 * (rank + 1) + i, pool=pool)
            packs.append(p_); genomes.append(g_); truths.append(t_)
        batch = cohort.concat_samples(packs, ["isolate%d" % (rank * isolates + i) for i in range(isolates)])
        g, goff = cohort.batch_genome(batch, genomes)
        return (batch.packed, g, goff, cohort.batch_contig_ranks(batch), cfg, [t for ts in truths for t in ts], batch,
                (packs[0], genomes[0], truths[0]))
    cfg = dict(CONFIGS[name])
    packed, genome, truth = synth_packed(**cfg, seed=seed, pool=pool)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    return packed, g, goff, pfh.contig_ranks(packed.contig_names), cfg, truth, None, (packed, genome, truth)


def time_config(hip, name, pool, device, steps=20, warmup=3, isolates=12):
    """ms/step of one more BASELINE config on this GPU (resident inputs, the fused call), for the `configs` table."""
    import torch
    from mustache_amd import lib
    from mustache_amd.findflanks import device_reads
    t0 = time.time()
    packed, g, goff, ranks, cfg, truth, batch, _ = make_workload(name, 100, pool, isolates)
    gen_s = time.time() - t0
    reads = device_reads(hip, packed, device)
    g_dev = torch.from_numpy(g).to(device)
    params = lib.make_params()
    dev_ms = 0.0
    for _ in range(max(warmup, 1)):                 # (untimed: the device span of a step, first to last HIP event of each stage)
        ff, pf = hip.run_flankpairs(reads, params, ranks, g_dev, goff)
        dev_ms += hip.device_ms()
    dev_ms /= max(warmup, 1)
    hip.span_timing(False)                          # timed like the headline steps: without the bracketing events
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ff, pf = hip.run_flankpairs(reads, params, ranks, g_dev, goff)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hip.span_timing(True)
    row = {"workload": name if batch is None else "%s (%d isolates in one call)" % (name, isolates), "reads": packed.n_reads,
           "multi_gpu_mode": {"C3_klebsiella": "--sharding coordinate (one contig holds > 95 % of the reads: whole-contig shards cannot scale; coordinate cuts keep the plasmids whole)",
                              "C5_meta_50mb": "--sharding coordinate", "C4_ecoli_batch": "--sharding sample (--isolates per GPU)"}.get(name, "--sharding sample"),
           "ms_per_step": round(dt / steps * 1e3, 4), "reads_per_s": round(packed.n_reads * steps / dt, 1), "steps": steps,
           "device_span_ms": round(dev_ms, 4),   # first to last HIP event of findflanks + of pairflanks (mfk_last_device_ms of the fused call)
           "insertions": len(truth), "flank_rows": int(len(ff["flanks"])),
           "pairs": int(len(pf["pairs"])), "pool": pool, "generate_s": round(gen_s, 1)}
    del reads, g_dev, packed
    torch.cuda.empty_cache()
    return row


def end_to_end(n_records, tmp, device):
    """BAM + FASTA on disk -> findflanks TSV -> pairflanks TSV through the drop-in stage drivers (`_findflanks`,
    `_pairflanks`: multi-threaded BGZF decode + packing on the host, copies over PCIe, kernels, table assembly, to_csv).
    The file is a simulated read set repeated on its own contigs per copy (mustache_amd/bigbam.py), so the expected tables
    are the base set's tables repeated: checked."""
    import pandas as pd
    from mustache_amd import bamio, bigbam, lib
    from mustache_amd import findflanks as ff_mod
    from mustache_amd.findflanks import _findflanks
    from mustache_amd.pairflanks import _pairflanks
    from mustache_amd.simulate import simulate_readset
    genome, rs, truth = simulate_readset(seed=7, contig_lens=(60000, 8000), n_insertions=8, coverage=100)
    copies = max(1, n_records // len(rs.reads))
    bam, fa = os.path.join(tmp, "mfk_bench_e2e.bam"), os.path.join(tmp, "mfk_bench_e2e.fa")
    one, tsv1, tsv2 = os.path.join(tmp, "mfk_bench_e2e_one.bam"), os.path.join(tmp, "mfk_bench_e2e.ff.tsv"), os.path.join(tmp, "mfk_bench_e2e.pf.tsv")
    t0 = time.time()
    contigs, n_rec = bigbam.write_replicated_bam(bam, rs, copies, procs=8)
    bamio.write_fasta(fa, {n: genome[n.rsplit("_", 1)[0]] for n, _ in contigs})
    bigbam.write_replicated_bam(one, rs, 1, procs=1)
    bamio.write_fasta(one + ".fa", {n: genome[n.rsplit("_", 1)[0]] for n, _ in bigbam.replicated_contigs(rs.contigs, 1)})
    write_s = time.time() - t0
    d = lib.DEFAULT_PARAMS
    ff_args = (d["min_softclip_length"], d["min_softclip_count"], d["min_alignment_quality"], d["min_alignment_inner_length"],
               d["min_distance_to_mate"], d["min_softclip_ratio"], d["max_indel_ratio"], d["large_insertion_cutoff"], d["min_count_consensus"], "bench")
    pf_args = (d["max_direct_repeat_length"], d["min_alignment_quality"], d["min_alignment_inner_length"], d["max_junction_spanning_prop"], d["large_insertion_cutoff"])
    try:
        # base set first (one copy): the expectation, and the warm-up of the library
        _findflanks(one, *ff_args, output_file=tsv1, device=device, verbose=False, decoder="native")
        _pairflanks(tsv1, one, one + ".fa", *pf_args, output_file=tsv2, device=device, verbose=False)
        def base_table(path, id_col):      # the one-copy file names its contigs <name>_000000: back to <name>, then repeated per copy
            t = pd.read_csv(path, sep="\t").drop(columns=[id_col])
            t["contig"] = [str(c).rsplit("_", 1)[0] for c in t["contig"]]
            return bigbam.replicate_table(t, copies)
        want_ff, want_pf = base_table(tsv1, "flank_id"), base_table(tsv2, "pair_id")
        # the pairs table is sorted by contig NAME (pandas sort_values in the reference, pairflanks.py:248), the flank table keeps file order
        want_pf = want_pf.sort_values(["contig", "pos_5p", "pos_3p"], kind="stable").reset_index(drop=True)
        def run(decoder):
            t0 = time.perf_counter()
            _findflanks(bam, *ff_args, output_file=tsv1, device=device, verbose=False, decoder=decoder)
            t1 = time.perf_counter()
            _pairflanks(tsv1, bam, fa, *pf_args, output_file=tsv2, device=device, verbose=False)
            t2 = time.perf_counter()
            got_ff = pd.read_csv(tsv1, sep="\t").drop(columns=["flank_id"])
            got_pf = pd.read_csv(tsv2, sep="\t").drop(columns=["pair_id"])
            ok = got_ff.to_csv(sep="\t", index=False) == want_ff.to_csv(sep="\t", index=False) and \
                got_pf.to_csv(sep="\t", index=False) == want_pf.to_csv(sep="\t", index=False)
            return {"findflanks_s": round(t1 - t0, 3), "pairflanks_s": round(t2 - t1, 3), "value": round(n_rec / (t2 - t0), 1), "unit": "reads/s",
                    "flank_rows": int(len(got_ff)), "pairs": int(len(got_pf)), "stages": dict(ff_mod.LAST_TIMING), "tables_equal_base_repeated": bool(ok),
                    "record_index_bytes": os.path.getsize(tsv1 + ".mfki") if os.path.exists(tsv1 + ".mfki") else None}
        # The commands' default: findflanks decodes on the device (mfk_decode_bam: inflate, record boundaries, packing and name links
        # as kernels; only the compressed file crosses PCIe), pairflanks reads the blocks around its regions on the host.  The first
        # call sizes the ctx's decode buffers for this file, the second finds them in place: both are reported, `value` is the first.
        first = run("gpu")
        second = run("gpu")
        out = {"records": n_rec, "bam_bytes": os.path.getsize(bam), "contigs": len(contigs)}
        out.update(first)
        out["second_call"] = {"findflanks_s": second["findflanks_s"], "pairflanks_s": second["pairflanks_s"], "value": second["value"],
                              "tables_equal_base_repeated": second["tables_equal_base_repeated"]}
        out["what"] = ("BAM + FASTA files -> findflanks TSV -> pairflanks TSV, the two commands with their defaults (findflanks --decoder gpu: the whole decode on "
                       "the device, leaves a record index; pairflanks reads only the BGZF blocks around its candidate regions through it, as bam.fetch does "
                       "through the BAI, host decode threads = %s)" % os.environ.get("OMP_NUM_THREADS", "all"))
        out["write_inputs_s"] = round(write_s, 1)
        host = run("native")
        out["host_decode"] = dict(host, what="the same with findflanks --decoder native (libmfkio.so: libdeflate + record parse on %s host threads)" % os.environ.get("OMP_NUM_THREADS", "all"))
        return out
    finally:
        for p in (bam, fa, one, one + ".fa", tsv1, tsv2, tsv1 + ".mfki"):
            try:
                os.remove(p)
            except OSError:
                pass


def table_digest(ff, pf):
    h = hashlib.sha256()
    for a in (ff["flanks"].tobytes(), ff["seqs"], pf["pairs"].tobytes(), pf["strs"]):
        h.update(a)
    return h.hexdigest()[:16]


def concurrent_contexts(hip, reads, params, ranks, g_dev, goff, n_reads, contexts, steps, device_id):
    """The same fused call from `contexts` host threads at once, each with a library context of its own (own streams, arena and
    pinned buffers) on the same resident inputs: what a cohort run does with several samples per GPU (the reference: one process
    per sample under `snakemake --cores`, main.py:44).  One context's step is a chain of short launches and five read-backs that
    leaves most of the device idle; the other contexts' chains run in the gaps.  Reported beside `value`, which stays one call
    at a time."""
    import threading
    import torch
    from mustache_amd import lib
    ctxs = [hip]
    try:
        for _ in range(contexts - 1):
            ctxs.append(lib.Binding(lib.LIB_PATH, "mfk_", device_id))
        for c in ctxs:
            c.span_timing(False)
            for _ in range(3):
                ff, pf = c.run_flankpairs(reads, params, ranks, g_dev, goff)
        want = table_digest(ff, pf)
        got, errs = [None] * len(ctxs), []
        gate = threading.Barrier(len(ctxs) + 1)

        def work(i):
            try:
                gate.wait()
                for _ in range(steps):
                    ff_, pf_ = ctxs[i].run_flankpairs(reads, params, ranks, g_dev, goff)
                got[i] = table_digest(ff_, pf_)
            except BaseException as e:      # noqa: BLE001 -- reported in the JSON line
                errs.append(repr(e))
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
        torch.cuda.synchronize()
        for t in ths:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        hip.span_timing(True)
        for c in ctxs[1:]:
            c.close()
    return {"contexts": contexts, "steps": steps * contexts, "ms_per_step": round(dt / (steps * contexts) * 1e3, 4),
            "value": round(n_reads * steps * contexts / dt, 1), "unit": "reads/s", "tables_equal": bool(not errs and all(d == want for d in got)),
            "errors": errs,
            "what": "the fused call from %d host threads at once, one library context each, same resident inputs; every context's tables "
                    "compared with the single-context ones (sha256); `value` above stays one call at a time" % contexts}


def launch_ranks(n, argv, launch_only):
    """`python bench.py --gpus N` without an external launcher: this process starts N fresh child processes of the same
    script (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, as torch.distributed.run would),
    relays rank 0's JSON line and exits non-zero if any child does.  The parent never touches the GPU: it imports neither
    torch nor the library, and children are plain subprocess.Popen (no exec from a process that has initialised HIP).
    The reference's counterpart is Snakemake's --cores starting one process per sample (main.py:44, Snakefile:64-109)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MFK_BENCH_CHILD="1")
        # rank 0's stdout carries the one JSON line; in launch-only mode every child's line is collected
        out = subprocess.PIPE if (r == 0 or launch_only) else sys.stderr
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env, stdout=out))
    lines, failed = [], []
    try:
        # drain the pipes first (a child blocks on a full pipe), rank 0 first; then collect the exit codes
        for r, p in enumerate(procs):
            if p.stdout is not None:
                lines.append((r, p.stdout.read().decode()))
        for r, p in enumerate(procs):
            if p.wait() != 0:
                failed.append((r, p.returncode))
    finally:
        for p in procs:              # the exact children started here, by PID, if something above raised
            if p.poll() is None:
                p.kill()
    if failed:
        sys.stderr.write("bench.py: rank(s) %s exited non-zero\n" % ", ".join("%d (code %s)" % f for f in failed))
        for _, text in lines:
            sys.stderr.write(text)
        sys.exit(1)
    if launch_only:
        ranks = [json.loads(text.strip().splitlines()[-1]) for _, text in lines]
        print(json.dumps({"launch_only": True, "n_gpus": n, "ranks": ranks}), flush=True)
    else:
        sys.stdout.write(lines[0][1])
        sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU).  Without an external launcher (no WORLD_SIZE in the "
                    "environment) N > 1 makes this process start N child ranks itself; under torch.distributed.run it must equal WORLD_SIZE")
    ap.add_argument("--launch-only", action="store_true", help="every rank reports its rank / world size as a JSON line and exits before any GPU work (checks the launcher)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2_saureus_1000x")
    ap.add_argument("--sharding", default="sample", choices=["sample", "coordinate", "contig"])
    ap.add_argument("--pool", default="all", choices=["all", "near"], help="'near' picks the pooled reads with the simulated truth (quicker to generate; not for reported numbers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-insertions", type=int, default=20)
    ap.add_argument("--merge", choices=["library", "torch"], default="library", help="N > 1: all-gather of the tables inside libmfk.so (RCCL) or through torch.distributed")
    ap.add_argument("--dump-profile", default=None, help="write the per-kernel table to this file")
    ap.add_argument("--isolates", type=int, default=12, help="isolates per GPU of --config C4_ecoli_batch (96 over 8 GPUs)")
    ap.add_argument("--table-configs", default="C1_ecoli_100x,C3_klebsiella,C4_ecoli_batch,C5_meta_50mb",
                    help="other configs measured for the `configs` table at N = 1 ('' = none)")
    ap.add_argument("--end-to-end-reads", type=int, default=4_000_000, help="records of the BAM file of the end_to_end leg (0 = skip)")
    ap.add_argument("--contexts", type=int, default=3, help="N = 1: the `concurrent_contexts` leg runs the same steps from this many host threads at once, one library context each (1 = skip)")
    ap.add_argument("--quick", action="store_true", help="only the timed region and the roofline (no CPU legs, no configs table, no end-to-end leg, no concurrent contexts)")
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline, args.table_configs, args.end_to_end_reads, args.contexts = True, "", 0, 1

    # ---- ranks: an external launcher (WORLD_SIZE set) or our own (N children of this script) -----------------------------
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if (args.gpus or 1) > 1:
            return launch_ranks(args.gpus, sys.argv[1:], args.launch_only)
        args.gpus = 1
    else:
        if args.gpus is None:
            args.gpus = int(env_world)
        if args.gpus != int(env_world):
            sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: the numbers must agree "
                     "(a line that says n_gpus=%d measured on %s ranks would be wrong)" % (args.gpus, env_world, args.gpus, env_world))
        for k in ("RANK", "LOCAL_RANK"):
            if k not in os.environ:
                sys.exit("bench.py: WORLD_SIZE is set but %s is not: start the ranks with torch.distributed.run or with plain `--gpus N`" % k)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch_only:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(),
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}), flush=True)
        return

    import numpy as np
    from mustache_amd import pairflanks as pfh

    sharded = args.sharding != "sample"

    from mustache_amd import lib
    lib.tune_allocator()      # freed result tables stay in the heap: no page faults for the same megabytes on every step
    params = lib.make_params()
    t0 = time.time()
    # sample sharding: every rank its own data set; coordinate / contig sharding: every rank generates the SAME data set
    # (there is no file to read it from) and keeps only its share
    packed, g, goff, ranks, cfg, truth, batch, cpu_args = make_workload(args.config, 100 + (0 if sharded else rank), args.pool, args.isolates, rank)
    gen_s = time.time() - t0
    total_reads_here = packed.n_reads
    n_insertions, contig_names = len(truth), list(packed.contig_names)
    flags = packed.flags
    clipped_reads = int(((flags & 15) != 0).sum())

    # ---- CPU legs first: nothing has touched the GPU yet, so the all-cores leg may fork --------------------------------
    cpu = sample = cpu_hashes = None
    if world == 1 and not args.no_cpu_baseline:
        cores = int(os.environ.get("OMP_NUM_THREADS") or 0) or min(os.cpu_count() or 1, 16)
        cpu, sample, cpu_hashes = cpu_legs(cpu_args[0], cpu_args[1], cpu_args[2], params, args.cpu_sample_insertions, cores)
        sample_g = pfh.genome_arrays(cpu_args[1], cpu_args[0].contig_names)

    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    from mustache_amd import cohort, dist as mdist
    from mustache_amd.findflanks import device_reads
    hip = lib.load(local_rank)
    merge = "none (one rank)"
    lib_merge = None
    if world > 1:
        merge = "torch.distributed all_gather (host-staged)"
        if args.merge == "library":
            why = ""
            try:
                mdist.init_library_comm(hip)
                ok = 1
            except Exception as e:                # noqa: BLE001 -- reported in the JSON line, the run goes on with torch.distributed
                ok, why = 0, str(e)
            # every rank must take the same route: one rank on torch.distributed while the others sit in ncclAllGather is a deadlock
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                lib_merge = hip
                merge = "mfk_allgather_sites (RCCL inside libmfk.so, one ncclAllGather per table)"
            else:
                hip.free_comm()
                merge += "; library communicator failed on %s: %s" % ("this rank" if not ok else "another rank", why)
    plan = None
    if sharded:
        h = mdist.halo(params.min_distance_to_mate, params.max_direct_repeat_length)
        plan = mdist.plan_shards(packed, world, args.sharding, h)[rank]
        mine = packed.take_ranges(plan.ranges)
        reads = device_reads(hip, mine, device)
        reads_here = mine.n_reads
        del mine
    else:
        reads = device_reads(hip, packed, device)
        reads_here = packed.n_reads
    g_dev = torch.from_numpy(g).to(device)

    def step():
        if sharded:     # the rank's shard through both stages, each followed by the all-gather + merge of the owned rows
            tab, seqs = mdist.findflanks_sharded(hip, reads, plan, params, device)
            pairs, strs = mdist.pairflanks_sharded(hip, reads, plan, params, tab, seqs, ranks, g_dev, goff, device)
            return {"flanks": tab, "seqs": seqs, "stats": np.zeros(16, np.int64)}, {"pairs": pairs, "strs": strs, "stats": np.zeros(16, np.int64)}
        ff, pf = hip.run_flankpairs(reads, params, ranks, g_dev, goff)   # mfk_findflanks then mfk_pairflanks inside one library call
        if batch is not None:
            per_sample = cohort.split_flanks(batch, ff)          # per-isolate tables, as the per-sample runs give them
            if world > 1:
                cohort.cohort_table(per_sample, sample_numbers=[rank * args.isolates + i for i in range(args.isolates)], device=device, binding=lib_merge)
        elif world > 1:
            mdist.allgather_flanks(ff["flanks"], ff["seqs"], device, binding=lib_merge)
        return ff, pf

    for _ in range(args.warmup):
        ff, pf = step()
    # three more untimed steps with every kernel bracketed: the per-kernel table and the ranking of the kernels
    hip.profile(True)
    warm_calls = 3
    for _ in range(warm_calls):
        ff, pf = step()
    full = hip.profile_totals()
    # work counters of one step, for the VALU models: kept by the kernels (stats[12]) or known from the tables
    work = {"myers_limb_steps": int(hip.last_stats[12]), "candidate_pairs": int(hip.last_pair_stats[0]), "events": int(hip.last_stats[10])}
    watched = list(SCAN_KERNELS)
    if full:
        watched.insert(0, max(full.items(), key=lambda kv: kv[1][0])[0])
    else:                                   # --warmup 0: no ranking yet, watch the usual suspects
        watched = ["k_leaf_myers"] + watched

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events only around the watched kernels (two event records per launch are not free)
    hip.profile_only(watched)
    hip.span_timing(False)      # the timed steps do without the two bracketing events per stage (mfk_last_device_ms is not read here)
    mdist.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ff, pf = step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = hip.profile_totals()
    hip.profile_only(None)
    hip.profile(False)
    hip.span_timing(True)
    xstats = dict(mdist.STATS)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nr = torch.tensor([0 if sharded else packed.n_reads], dtype=torch.int64, device=device)
        dist.all_reduce(nr)
        total_reads = total_reads_here if sharded else int(nr.item())   # sharded: the ranks share ONE data set
    else:
        total_reads = total_reads_here

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed
        st_ff, st_pf = hip.last_stats, hip.last_pair_stats
        meta = {"n_reads": reads_here, "clipped_reads": clipped_reads if not sharded else clipped_reads // world, "events": int(st_ff[10]),
                "join_visits": int(st_ff[3]) * 3 * int(cfg["coverage"]),       # an estimate: sites after the first neighbour filter x 3 positions x coverage
                "pairs": int(st_pf[1]), "coverage": int(cfg["coverage"]), "timed_steps": args.steps, "work": work}
        # per-kernel rows (name, avg ms per launch, launches, total ms).  Shares and the device total come from the fully
        # bracketed warm-up steps; the durations of the watched kernels come from the timed region.
        full_table = sorted(((nm, ms / max(calls, 1), calls, ms) for nm, (ms, calls) in full.items()), key=lambda r: -r[3])
        full_tot = max(sum(r[3] for r in full_table), 1e-12)
        share = {r[0]: r[3] / full_tot for r in full_table}
        dev_ms = full_tot / warm_calls if full_table else None
        table = sorted(((nm, ms / max(calls, 1), calls, ms) for nm, (ms, calls) in prof.items()), key=lambda r: -share.get(r[0], 0.0))
        # `roofline` = the kernel with the largest share of device time (HBM model if it streams, VALU model if it is an
        # integer recurrence); `read_scan` beside it = the two kernels that touch every read (the metric's unit)
        roofline = None
        if table:
            roofline = kernel_entry([table[0][0]], table, share, meta, args.config)
            scan = kernel_entry(SCAN_KERNELS, table, share, meta, args.config)
            if roofline is None:
                roofline = scan
            if roofline is not None:
                roofline["read_scan"] = scan
                # the five largest kernels by device time, each with its own model (durations from the bracketed warm-up steps)
                meta_w = dict(meta, timed_steps=warm_calls)
                roofline["top5_by_time"] = [kernel_entry([r[0]], full_table, share, meta_w, args.config) for r in full_table[:5]]
                roofline["work_counters_per_step"] = work
                own = int(reads_here * 1 + meta["clipped_reads"] * 23 + meta["events"] * 24)
                # 24 B record + 1.5 B per clipped base + 8 B per extra block per read (= 25.6 B/read there), 13 B per
                # (site, overlapping read) of loop B, and per candidate pair 104 B + (13 + 150 x 1.5 + 8) B per read over it
                d8 = int(reads_here * 25.6 + meta["join_visits"] * 13 + int(st_pf[0]) * (104 + (13 + 233) * int(cfg["coverage"]) * 200 / 150))
                roofline["whole_step"] = {"algorithmic_bytes": own, "GBps": round(own / (ms_per_step * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(own / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                          "survey_8d_model": {"bytes": d8, "GBps": round(d8 / (ms_per_step * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(d8 / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                              "what": "SURVEY 8(d)'s per-read figure (25.6 B/read + 13 B per site x read visit + the pairing stage's per-pair bytes): what a 24-byte-record layout would move; the SoA layout of this build moves `algorithmic_bytes`"},
                                          "note": ("the step is a chain of dependent launches over a few dozen sites: launch- and latency-bound, not HBM-bound" if int(st_pf[0]) < 1000 else
                                                   "thousands of sites: the step is bound by the per-site-side consensus (k_job_small, vector-ALU issue at four waves per SIMD) and the per-pair kernels, not by HBM")}
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
        out = {"metric": "aligned reads/sec (findflanks+pairflanks, whole node)", "value": round(value, 1), "unit": "reads/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u8/int32 (+f64 ratio tests)",
               "data": "synthetic",
               # the exchange step, rank 0's view: tables of how many ranks the last all-gather returned (= n_gpus when the
               # collective really ran over all ranks), host ms per step inside the all-gathers and in the selection / merges around them
               "ranks_seen": int(xstats["ranks_seen"]), "gather_ms": round(xstats["gather_s"] / args.steps * 1e3, 4),
               "merge_ms": round(xstats["merge_s"] / args.steps * 1e3, 4), "gathers_per_step": round(xstats["gathers"] / args.steps, 2),
               "config": {"workload": args.config, "reads_per_gpu": reads_here, "reads_total": total_reads,
                          "genome_bp": int(sum(cfg["contig_lens"])) * (args.isolates if batch is not None else 1),
                          "isolates_per_gpu": args.isolates if batch is not None else 1,
                          "coverage": cfg["coverage"], "read_length": 150, "insertions": n_insertions,
                          "flank_rows": int(len(ff["flanks"])), "pairs": int(len(pf["pairs"])),
                          "sharding": {"sample": "sample per GPU", "coordinate": "coordinate ranges of one data set, halo 69", "contig": "whole contigs of one data set"}[args.sharding], "merge": merge,
                          "pool": "every read carries its bases (no use of the simulated truth)" if args.pool == "all" else "clipped reads + reads near the simulated insertions (uses the truth)",
                          "device_ms_per_step": round(dev_ms, 4) if dev_ms else None, "generate_s": round(gen_s, 1)},
               "roofline": roofline, "cpu_baseline": cpu}
        if world == 1:
            # the legs below are reported beside the headline: a failure in one of them is a field of the line, not the end of the run
            if args.contexts > 1 and not sharded:
                try:
                    out["concurrent_contexts"] = concurrent_contexts(hip, reads, params, ranks, g_dev, goff, reads_here, args.contexts, args.steps, local_rank)
                except Exception as e:      # noqa: BLE001
                    out["concurrent_contexts"] = {"error": repr(e)}
            del reads
            torch.cuda.empty_cache()
            if sample is not None:
                # the GPU on the CPU legs' sample: sha256 of both TSVs must agree with the CPU port's
                rd = device_reads(hip, sample, device)
                sg = torch.from_numpy(sample_g[0]).to(device)
                sff, spf = hip.run_flankpairs(rd, params, pfh.contig_ranks(sample.contig_names), sg, sample_g[1])
                texts = tsv_texts(sff, spf, sample.contig_names)
                mine_h = {"findflanks_tsv_sha256": sha(texts[0]), "pairflanks_tsv_sha256": sha(texts[1])}
                out["parity_in_run"] = {"sample": cpu["sample"].split(",")[0], "gpu": mine_h, "cpu_port": cpu_hashes,
                                        "equal": all(mine_h[k] == cpu_hashes[k] for k in mine_h)}
                del rd, sg
                torch.cuda.empty_cache()
            rows = []
            for name in [c for c in args.table_configs.split(",") if c and c != args.config]:
                try:
                    rows.append(time_config(hip, name, args.pool, device, isolates=args.isolates))
                except Exception as e:      # noqa: BLE001
                    rows.append({"workload": name, "error": repr(e)})
            if rows or args.table_configs:
                me = {"workload": args.config, "reads": reads_here, "ms_per_step": round(ms_per_step, 4), "reads_per_s": round(value, 1),
                      "steps": args.steps, "insertions": n_insertions, "flank_rows": int(len(ff["flanks"])), "pairs": int(len(pf["pairs"])), "pool": args.pool}
                out["configs"] = [me] + rows
            if args.end_to_end_reads > 0:
                try:
                    out["end_to_end"] = end_to_end(args.end_to_end_reads, os.environ.get("TMPDIR", "/tmp"), device)
                except Exception as e:      # noqa: BLE001
                    out["end_to_end"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
