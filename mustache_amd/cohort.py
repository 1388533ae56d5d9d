"""Multi-isolate batches (BASELINE config C4: 96 E. coli isolates, sample-sharded across the GPUs of a node).

The reference runs one ``mustache findflanks`` / ``pairflanks`` process per sample (Snakefile:64-109).  On a GPU a
100x isolate is launch-bound (a chain of ~70 short kernels for 3 M reads), so the isolates a rank owns are run as ONE
call: their packed reads are concatenated and every (sample, contig) becomes a contig of the batch.  Nothing in the
path crosses a contig -- site keys are (contig, pos) (findflanks.py:100-121), the min-distance neighbourhoods and the
pairing window stay inside a contig (findflanks.py:542-584, pairflanks.py:155-157), query-name chains are per sample
(``link`` is rebased per sample) -- so each sample's rows are exactly those of its own run; ``split_*`` cuts the
batch tables back into per-sample tables, row order unchanged.

``cohort_table`` is the "RCCL merge of candidate sites": one all-gather of the per-rank flank tables, with the global
sample number carried in the upper 16 bits of ``contig_id``.
"""
from __future__ import annotations

import numpy as np

from . import lib as _lib
from .pack import ALL_FIELDS, PackedReads, PackError, _HOT_FIELDS


class Batch:
    """Packed reads of several samples as one PackedReads + what is needed to cut results apart again."""

    def __init__(self, packed, contig_start, sample_ids, contig_names):
        self.packed = packed
        self.contig_start = np.asarray(contig_start, dtype=np.int64)   # first batch contig of sample s; [n_samples] = total
        self.sample_ids = list(sample_ids)
        self.contig_names = contig_names                                # per sample: its own contig names

    @property
    def n_samples(self):
        return len(self.sample_ids)

    def sample_of_contig(self, contig_ids):
        return np.searchsorted(self.contig_start, np.asarray(contig_ids, dtype=np.int64), side="right") - 1


def concat_samples(packs, sample_ids=None):
    """Concatenate per-sample PackedReads into one batch (read order: sample by sample, each in its BAM order)."""
    packs = list(packs)
    if not packs:
        raise ValueError("empty batch")
    if sample_ids is None:
        sample_ids = ["sample%d" % i for i in range(len(packs))]
    n_contigs = sum(len(p.contig_names) for p in packs)
    if n_contigs > 65535:
        raise PackError("more than 65535 (sample, contig) pairs in one batch")
    cols = {k: [] for k, _ in ALL_FIELDS}
    contig_start, names = [0], []
    r_off = c_off = blk_off = 0
    pool_off = 0
    for p in packs:
        a = p.a
        n = p.n_reads
        for k, _ in _HOT_FIELDS:
            if k not in ("link", "contig_id"):
                cols[k].append(a[k])
        link = a["link"].astype(np.int64)
        link[link >= 0] += r_off
        cols["link"].append(link.astype(np.int32))
        cols["contig_id"].append((a["contig_id"].astype(np.int64) + c_off).astype(np.uint16))
        cols["contig_len"].append(a["contig_len"])
        cols["contig_read_start"].append(a["contig_read_start"][:-1].astype(np.int64) + r_off)
        cols["cx_read"].append((a["cx_read"].astype(np.int64) + r_off).astype(np.uint32))
        cols["cx_off"].append((a["cx_off"][:-1].astype(np.int64) + blk_off).astype(np.uint32))
        for k in ("blk_start", "blk_end", "blk_ins"):
            cols[k].append(a[k])
        cols["pool_read"].append((a["pool_read"].astype(np.int64) + r_off).astype(np.uint32))
        cols["pool_off"].append(a["pool_off"][:-1].astype(np.uint64) + np.uint64(pool_off))
        nb = int(a["pool_off"][-1])            # bases of this sample's pool: even, every read starts on a byte boundary
        if nb & 1:
            raise PackError("pool of a sample does not end on a byte boundary")
        cols["pool_bases"].append(a["pool_bases"][:nb // 2])
        cols["pool_quals"].append(a["pool_quals"][:nb])
        r_off += n
        c_off += len(p.contig_names)
        blk_off += int(a["blk_start"].shape[0])
        pool_off += nb
        contig_start.append(c_off)
        names.append(list(p.contig_names))
    if r_off >= (1 << 31):
        raise PackError("more than 2^31-1 reads in one batch")
    cols["contig_read_start"].append(np.asarray([r_off], dtype=np.int64))
    cols["cx_off"].append(np.asarray([blk_off], dtype=np.uint32))
    cols["pool_off"].append(np.asarray([pool_off], dtype=np.uint64))
    arrays = {k: np.concatenate(cols[k]).astype(t, copy=False) for k, t in ALL_FIELDS}
    flat_names = ["%s|%s" % (sample_ids[s], c) for s, cn in enumerate(names) for c in cn]
    merged = PackedReads(flat_names, arrays, max(p.max_span for p in packs))
    return Batch(merged, contig_start, sample_ids, names)


def split_flanks(batch, result):
    """Result of ``run_findflanks`` on the batch -> per sample {"flanks", "seqs"} with the sample's own contig ids.
    Rows come sorted by batch contig and their sequences lie in row order, so a sample is one slice of each."""
    fl, seqs = result["flanks"], result["seqs"]
    cuts = np.searchsorted(fl["contig_id"], batch.contig_start) if len(fl) else np.zeros(batch.n_samples + 1, np.int64)
    ends = fl["seq_off"] + fl["seq_len"] if len(fl) else None
    out = []
    for s in range(batch.n_samples):
        i0, i1 = int(cuts[s]), int(cuts[s + 1])
        rows = fl[i0:i1].copy()
        if i1 > i0:
            b0, b1 = int(rows["seq_off"][0]), int(ends[i1 - 1])
            rows["contig_id"] -= int(batch.contig_start[s])
            rows["seq_off"] -= b0
            out.append({"flanks": rows, "seqs": seqs[b0:b1]})
        else:
            out.append({"flanks": rows, "seqs": b""})
    return out


def merge_flank_tables(batch, per_sample):
    """Per-sample flank tables (FLANK_DTYPE rows + seq bytes, own contig ids) -> the batch's table for pairflanks."""
    rows, blobs, off, first_row = [], [], 0, [0]
    for s, t in enumerate(per_sample):
        r = np.array(t["flanks"], dtype=_lib.FLANK_DTYPE, copy=True)
        r["contig_id"] += int(batch.contig_start[s])
        r["seq_off"] += off
        rows.append(r)
        blobs.append(bytes(t["seqs"]))
        off += len(t["seqs"])
        first_row.append(first_row[-1] + len(r))
    tab = np.concatenate(rows) if rows else np.zeros(0, dtype=_lib.FLANK_DTYPE)
    return tab, b"".join(blobs), np.asarray(first_row, dtype=np.int64)


def batch_contig_ranks(batch):
    """Sort rank of every batch contig for pairflanks' final sort_values(['contig', ...]) (pairflanks.py:248): samples
    in batch order, contigs of a sample by name like the per-sample run."""
    rank = np.zeros(int(batch.contig_start[-1]), dtype=np.int32)
    for s, cn in enumerate(batch.contig_names):
        base = int(batch.contig_start[s])
        for r, i in enumerate(sorted(range(len(cn)), key=lambda i: cn[i])):
            rank[base + i] = base + r
    return rank


def batch_genome(batch, genomes):
    """genomes: one dict name -> sequence per sample (or a single dict shared by all) -> (uint8 array, offsets)."""
    from .pairflanks import genome_arrays
    chunks, off = [], [0]
    for s, cn in enumerate(batch.contig_names):
        g = genomes if isinstance(genomes, dict) else genomes[s]
        arr, o = genome_arrays(g, cn)
        total = int(o[-1])
        chunks.append(arr[:total])
        base = off[-1]
        off.extend(base + int(x) for x in o[1:])
    g = np.concatenate(chunks) if chunks else np.zeros(1, np.uint8)
    if g.size == 0:
        g = np.zeros(1, np.uint8)
    return g, np.asarray(off, dtype=np.int64)


def split_pairs(batch, result, first_row):
    """Result of ``run_pairflanks`` on the batch -> per sample {"pairs", "strs"}: contig ids and flank indices of the
    sample's own tables (string offsets stay into the shared ``strs``)."""
    pr = result["pairs"]
    owner = batch.sample_of_contig(pr["contig_id"]) if len(pr) else np.zeros(0, np.int64)
    out = []
    for s in range(batch.n_samples):
        rows = pr[owner == s].copy()
        rows["contig_id"] -= int(batch.contig_start[s])
        rows["index_5p"] -= int(first_row[s])
        rows["index_3p"] -= int(first_row[s])
        out.append({"pairs": rows, "strs": result["strs"]})
    return out


def findflanks_batch(binding, batch, reads_struct, params):
    """One ``mfk_findflanks`` call over all samples of the batch -> per-sample tables."""
    return split_flanks(batch, binding.run_findflanks(reads_struct, params))


def pairflanks_batch(binding, batch, reads_struct, params, per_sample_flanks, genome_dev, genome_off):
    """One ``mfk_pairflanks`` call over all samples of the batch -> per-sample pair tables."""
    tab, seqs, first_row = merge_flank_tables(batch, per_sample_flanks)
    res = binding.run_pairflanks(reads_struct, params, tab, seqs, batch_contig_ranks(batch), genome_dev, genome_off)
    return split_pairs(batch, res, first_row)


def cohort_table(per_sample, sample_numbers=None, device=None, binding=None):
    """All-gather of the candidate sites of every rank's samples: rows carry ``sample << 16 | contig`` in contig_id
    (sample = global sample number, ``sample_numbers[k]`` for this rank's k-th table; default 0, 1, ...).  Returns
    (rows, seqs) in rank order; without torch.distributed it is this rank's table.  With `binding` (a library context whose
    communicator is up, dist.init_library_comm) the gather is mfk_allgather_sites inside libmfk.so."""
    from .dist import allgather_flanks
    rows, blobs, off = [], [], 0
    for s, t in enumerate(per_sample):
        r = np.array(t["flanks"], dtype=_lib.FLANK_DTYPE, copy=True)
        if len(r) and int(r["contig_id"].max()) >= (1 << 16):
            raise ValueError("contig id does not fit 16 bits")
        num = s if sample_numbers is None else int(sample_numbers[s])
        if not 0 <= num < (1 << 15):
            raise ValueError("sample number does not fit 15 bits")
        r["contig_id"] |= num << 16
        r["seq_off"] += off
        rows.append(r)
        blobs.append(bytes(t["seqs"]))
        off += len(t["seqs"])
    mine = np.concatenate(rows) if rows else np.zeros(0, dtype=_lib.FLANK_DTYPE)
    return allgather_flanks(mine, b"".join(blobs), device, binding=binding)


def flankpairs_concurrent(samples, params, device="cuda:0", contexts=3):
    """Samples that are NOT run as one batch (different parameters per call site, samples arriving one by one, a sample too
    large to concatenate): each is one fused findflanks + pairflanks call, and ``contexts`` host threads with a library
    context each take them from a queue, so the launch chains of several samples share the device (a deep isolate's step
    leaves most of it idle: three contexts run three samples in the time of 1.6).  The reference runs
    one process per sample (Snakefile:64-109); on a GPU that is the wrong unit, processes take turns.

    samples: list of (PackedReads or gpudecode.DevicePacked, genome bytes as uint8 array, genome offsets) as
    ``pairflanks.genome_arrays`` makes them.  Returns [(findflanks result, pairflanks result)] in the order of ``samples``;
    the first exception of any worker is raised after all have stopped."""
    import threading
    import torch
    from .findflanks import device_reads
    from .pairflanks import contig_ranks
    dev_id = int(str(device).split(":")[1]) if ":" in str(device) else 0
    n = len(samples)
    out, errs = [None] * n, []
    lock = threading.Lock()
    nxt = [0]

    def work(k):
        binding = None
        try:
            binding = _lib.load(dev_id) if k == 0 else _lib.Binding(_lib.LIB_PATH, "mfk_", dev_id)
            while not errs:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= n:
                    break
                packed, g, goff = samples[i]
                reads = device_reads(binding, packed, device)
                g_dev = torch.from_numpy(np.ascontiguousarray(g)).to(device)
                out[i] = binding.run_flankpairs(reads, params, contig_ranks(packed.contig_names), g_dev, goff)
                del reads, g_dev
        except BaseException as e:      # noqa: BLE001 -- raised by the caller's thread below
            errs.append(e)
        finally:
            if k and binding is not None:
                binding.close()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(max(1, min(int(contexts), n)))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return out
