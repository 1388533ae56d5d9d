"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The path shards without any data-path collective: samples are independent (C4), contigs are
independent (C3), and coordinate ranges of one contig are independent given a halo (C5).  The only exchanges are the
merges of the per-rank result tables -- an all-gather of a few KB..MB of fixed-width records after each stage,
latency-bound on xGMI.

Partitioning
* ``shard_samples``    round-robin assignment of isolates to ranks (``cohort.py`` runs a rank's isolates as one batch)
* ``plan_shards``      what every rank OWNS and which reads it needs for that:
    ``sharding="coordinate"``  cuts in BAM order balanced by read count; a rank owns the positions between two cuts
                               (a cut inside a small contig moves to the contig's border) and computes ``halo`` positions
                               beyond them: the three chained min-distance filters reach 3 x mindist + 3, not one window
    ``sharding="contig"``      whole contigs, largest first into the least loaded rank

Stages (the same functions run under RCCL, under gloo with the CPU oracle in tests/, and for K shards one after the other
on a single GPU in the GPU suite):
* ``shard_findflanks``   ``mfk_findflanks`` on the rank's reads -> the flank rows it owns
* ``merge_flanks``       parts of all ranks -> the table of the unsharded run, byte for byte
* ``shard_pairflanks``   ``mfk_pairflanks`` on the merged rows around the rank's range -> the pairs whose 5p flank it owns
* ``merge_pairs``        parts of all ranks -> the pair table of the unsharded run
* ``findflanks_sharded`` / ``pairflanks_sharded``  = stage + all-gather + merge for the calling rank
* ``run_sharded_sequential``  all ranks of a world one after the other in this process (no collective)
"""
from __future__ import annotations

import numpy as np

from . import lib as _lib


def halo(min_distance_to_mate=22, max_direct_repeat_length=20):
    """Positions a shard must compute beyond the range it owns.  filter_softclips_mindistance runs three times
    (reference findflanks.py:40, 43, 49) and each round's verdict on a site depends on its neighbours' survival in the
    previous round, so the reach is 3 x mindist; pairing looks back maxdr + 1."""
    # pairflanks on the owner of pos_5p: the spanning fetch starts at pos_3p - 1 >= pos_5p - maxdr - 2, the 50-bp
    # direct-repeat window reaches 25 beyond the middle of the pair (pairflanks.py:294-305, 377)
    return max(3 * min_distance_to_mate + 3, max_direct_repeat_length + 2, (max_direct_repeat_length + 1) // 2 + 27)


def shard_samples(n_samples, world_size, rank):
    return list(range(rank, n_samples, world_size))


def shard_contigs(reads_per_contig, world_size):
    """Largest-first bin packing; returns a list (per rank) of contig index lists."""
    bins = [[] for _ in range(world_size)]
    load = [0] * world_size
    for c in sorted(range(len(reads_per_contig)), key=lambda i: -reads_per_contig[i]):
        k = load.index(min(load))
        bins[k].append(c)
        load[k] += reads_per_contig[c]
    return [sorted(b) for b in bins]


def coordinate_shards(packed, contig, world_size):
    """Owned position ranges [a_k, b_k) of one contig, balanced by read count."""
    r0, r1 = int(packed.contig_read_start[contig]), int(packed.contig_read_start[contig + 1])
    clen = int(packed.contig_len[contig])
    cuts = [-1]
    for k in range(1, world_size):
        r = r0 + (r1 - r0) * k // world_size
        cuts.append(int(packed.ref_start[r]) if r < r1 else clen + 1)
    cuts.append(clen + 1)
    return [(cuts[k], cuts[k + 1]) for k in range(world_size)]


def slice_for_range(packed, contig, own, h, max_span=None):
    """Reads a shard needs to compute every site in [own[0]-h, own[1]+h): all reads that can overlap
    [a-h-1, b+h+2).  Returns a self-contained PackedReads (read indices renumbered)."""
    lo, hi = _read_range(packed, contig, own, h, max_span)
    return packed.slice_reads(lo, max(lo, hi))


def _read_range(packed, contig, own, h, max_span=None):
    r0, r1 = int(packed.contig_read_start[contig]), int(packed.contig_read_start[contig + 1])
    span = packed.max_span if max_span is None else max_span
    starts = packed.ref_start[r0:r1]
    lo = r0 + int(np.searchsorted(starts, own[0] - h - 1 - span, side="left"))
    hi = r0 + int(np.searchsorted(starts, own[1] + h + 2, side="left"))
    return lo, max(lo, hi)


def owned(rows, own, contig=None):
    """Filter flank / site rows to the positions a shard owns."""
    m = (rows["pos"] >= own[0]) & (rows["pos"] < own[1])
    if contig is not None:
        m &= rows["contig_id"] == contig
    return rows[m]


# ---------------------------------------------------------------------------------------------------------------------
# shard plans
# ---------------------------------------------------------------------------------------------------------------------
class ShardPlan:
    """What one rank owns: ``pieces`` = [(contig, a, b)] owned site positions a <= pos < b (a site position runs from -1
    to the contig length), ``ranges`` = the read index ranges it computes on (pieces + halo, merged, ascending)."""

    def __init__(self, pieces, ranges):
        self.pieces, self.ranges = list(pieces), list(ranges)
        self._lo = self._hi = None

    def n_reads(self):
        return sum(hi - lo for lo, hi in self.ranges)

    def owns(self, contig_id, pos):
        """Boolean mask over arrays of contig ids and positions.  A rank's pieces are one stretch of the (contig, position) order,
        so a contig appears in at most one of them: two tables indexed by contig id instead of a pass per piece (a metagenome
        rank owns hundreds of contigs)."""
        contig_id, pos = np.asarray(contig_id), np.asarray(pos)
        if contig_id.size == 0 or not self.pieces:
            return np.zeros(contig_id.shape, dtype=bool)
        if self._lo is None:
            top = max(c for c, _, _ in self.pieces) + 1
            lo, hi = np.zeros(top, dtype=np.int64), np.zeros(top, dtype=np.int64)      # empty interval: owns nothing
            seen = set()
            for c, a, b in self.pieces:
                if c in seen:                                                          # (not produced by plan_shards) fall back
                    lo = None
                    break
                seen.add(c)
                lo[c], hi[c] = a, b
            self._lo, self._hi = (lo, hi) if lo is not None else (False, False)
        if self._lo is False:
            m = np.zeros(contig_id.shape, dtype=bool)
            for c, a, b in self.pieces:
                m |= (contig_id == c) & (pos >= a) & (pos < b)
            return m
        cid = contig_id.astype(np.int64)
        inside = cid < len(self._lo)
        c2 = np.where(inside, cid, 0)
        p64 = pos.astype(np.int64)
        return inside & (p64 >= self._lo[c2]) & (p64 < self._hi[c2])

    def __repr__(self):
        return "ShardPlan(pieces=%r, ranges=%r)" % (self.pieces, self.ranges)


def _merge_ranges(ranges):
    out = []
    for lo, hi in sorted(r for r in ranges if r[1] > r[0]):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def plan_shards(packed, world_size, sharding="coordinate", h=None, small_contig_reads=None, cuts=None):
    """One ShardPlan per rank.  Every (contig, position) is owned by exactly one rank, whatever the world size.

    coordinate: cut k sits at read ``n k / world`` of the BAM order: (contig of that read, its reference start).  A cut
    inside a contig with fewer than ``small_contig_reads`` reads (default: a quarter of a rank's share) moves to the nearer
    border of that contig, so plasmids stay whole.  contig: whole contigs by ``shard_contigs``.  ``cuts``: the
    world_size - 1 inner cuts of a coordinate plan given outright as ascending (contig, first owned position) pairs."""
    if h is None:
        h = halo()
    n, nc = packed.n_reads, len(packed.contig_names)
    crs = np.asarray(packed.contig_read_start, dtype=np.int64)
    clen = np.asarray(packed.contig_len, dtype=np.int64)
    if sharding == "contig":
        bins = shard_contigs([int(crs[c + 1] - crs[c]) for c in range(nc)], world_size)
        return [ShardPlan([(c, -1, int(clen[c]) + 1) for c in b], _merge_ranges([(int(crs[c]), int(crs[c + 1])) for c in b]))
                for b in bins]
    if sharding != "coordinate":
        raise ValueError("sharding must be 'coordinate' or 'contig', not %r" % (sharding,))
    if small_contig_reads is None:
        small_contig_reads = max(n // (4 * world_size), 1)
    given = cuts
    cuts = [(0, -1)]                                       # (contig, first owned position)
    if given is not None:
        if len(given) != world_size - 1 or sorted(given) != list(given):
            raise ValueError("cuts must be world_size - 1 ascending (contig, position) pairs")
        cuts += [(int(c), int(p)) for c, p in given]
    for k in range(1, world_size if given is None else 0):
        r = n * k // world_size
        if r >= n:
            cuts.append((nc, -1))
            continue
        c = int(np.searchsorted(crs, r, side="right") - 1)
        if crs[c + 1] - crs[c] < small_contig_reads:
            cut = (c, -1) if r - crs[c] < crs[c + 1] - r else (c + 1, -1)
        else:
            cut = (c, int(packed.ref_start[r]))
        cuts.append(max(cut, cuts[-1]))                    # cuts never go backwards
    cuts.append((nc, -1))
    plans = []
    for k in range(world_size):
        (c0, p0), (c1, p1) = cuts[k], cuts[k + 1]
        pieces = []
        for c in range(c0, min(c1, nc - 1) + 1):
            a = p0 if c == c0 else -1
            b = p1 if c == c1 else int(clen[c]) + 1
            if c < nc and b > a:
                pieces.append((c, a, b))
        ranges = []
        for c, a, b in pieces:
            if a == -1 and b == int(clen[c]) + 1:
                ranges.append((int(crs[c]), int(crs[c + 1])))
            else:
                ranges.append(_read_range(packed, c, (a, b), h))
        plans.append(ShardPlan(pieces, _merge_ranges(ranges)))
    return plans


# ---------------------------------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------------------------------
# The selection / merge loops below exist twice: in numpy (this file: the readable statement, and what runs when libmfk.so is
# not built) and in C inside libmfk.so (mfk_tables.hip: mfk_flanks_owned, mfk_flank_runs_owned, mfk_pairs_owned, mfk_flanks_merge,
# mfk_pairs_merge), which is what the sharded stages call: at 20 000 flank rows and 10 000 pairs the numpy loops were 2.8 ms of
# a 7.4 ms step.  MFK_NUMPY_MERGE=1 forces the numpy statement; tests/test_dist.py compares the two.
_tables_so = None


def _tables():
    global _tables_so
    import os
    if os.environ.get("MFK_NUMPY_MERGE", "0") not in ("", "0"):
        return None
    if _tables_so is None:
        import ctypes as C
        try:
            # torch's own copy of the HIP runtime has to be in the process BEFORE libmfk.so is opened (lib.load does the same):
            # opened first, libmfk.so would pull in the system's libamdhip64 and a later lib.load finds "no usable HIP device"
            import torch  # noqa: F401
            so = C.CDLL(_lib.LIB_PATH)
            p, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
            so.mfk_flanks_owned.restype = i64
            so.mfk_flanks_owned.argtypes = [p, i64, p, p, p, i32, p, p, C.POINTER(i64)]
            so.mfk_flank_runs_owned.restype = i64
            so.mfk_flank_runs_owned.argtypes = [p, i64, p, p, i32, i32, p]
            so.mfk_pairs_owned.restype = i64
            so.mfk_pairs_owned.argtypes = [p, i64, p, p, p, i32, p, p, p, C.POINTER(i64)]
            so.mfk_flanks_merge.restype = C.c_int
            so.mfk_flanks_merge.argtypes = [p, p, i32]
            so.mfk_pairs_merge.restype = C.c_int
            so.mfk_pairs_merge.argtypes = [p, p, i32, p, i32]
            _tables_so = so
        except (OSError, AttributeError):
            _tables_so = False
    return _tables_so or None


def _own_bounds(plan, n_contigs):
    """(own_lo, own_hi) int32 per contig for the C loops, or None when a contig appears in two pieces (not made by plan_shards)."""
    key = ("_bounds", n_contigs)
    got = getattr(plan, "_cbounds", None)
    if got is not None and got[0] == key:
        return got[1]
    lo, hi = np.zeros(n_contigs, dtype=np.int32), np.zeros(n_contigs, dtype=np.int32)
    seen, out = set(), None
    for c, a, b in plan.pieces:
        if c in seen:
            break
        seen.add(c)
        if 0 <= c < n_contigs:
            lo[c], hi[c] = max(a, -2**31), min(b, 2**31 - 1)
    else:
        out = (lo, hi)
    plan._cbounds = (key, out)
    return out


def _n_contigs_of(plan, *id_arrays):
    top = max([c for c, _, _ in plan.pieces] + [int(a.max()) for a in id_arrays if len(a)] + [0])
    return top + 1


def _fast_copy(rows):
    """Copy of a structured array through its bytes (numpy copies structured elements one by one: 3 ms per 20 000 rows)."""
    rows = np.ascontiguousarray(rows)
    return rows.view(np.uint8).copy().view(rows.dtype) if len(rows) else rows.copy()


def _concat_rows(parts, dtype):
    """Concatenation of structured arrays through their bytes (np.concatenate copies structured elements one by one)."""
    views = [np.ascontiguousarray(np.asarray(r, dtype=dtype)).view(np.uint8) for r in parts if len(r)]
    if not views:
        return np.zeros(0, dtype=dtype)
    return np.concatenate(views).view(dtype)


def _fast_take(rows, order):
    """rows[order] for a structured array, as a 2-D take over its bytes."""
    rows = np.ascontiguousarray(rows)
    if len(rows) == 0:
        return rows[order]
    w = rows.dtype.itemsize
    return np.ascontiguousarray(rows.view(np.uint8).reshape(len(rows), w)[order]).reshape(-1).view(rows.dtype)


def _stable_order(*keys_major_first):
    """Stable order by the given int64 keys (most significant first), or None if the rows are in that order already (the
    tables of coordinate shards, concatenated in rank order, are)."""
    n = len(keys_major_first[0])
    if n < 2:
        return None
    in_order = np.ones(n - 1, dtype=bool)
    tie = np.ones(n - 1, dtype=bool)
    for k in keys_major_first:
        k = np.asarray(k, dtype=np.int64)
        in_order &= ~(tie & (k[1:] < k[:-1]))
        tie &= k[1:] == k[:-1]
    if in_order.all():
        return None
    return np.lexsort(tuple(np.asarray(k, dtype=np.int64) for k in reversed(keys_major_first)))


def _compact(rows, blob, off_len_fields):
    """Copy of ``rows`` whose strings lie in row order (fields in the given order inside a row) at the front of a fresh
    blob, offsets re-based; a negative length (None) takes no bytes."""
    rows = _fast_copy(rows)
    n = len(rows)
    if n == 0:
        return rows, b""
    lens = np.stack([np.maximum(rows[fl].astype(np.int64), 0) for _, fl in off_len_fields], 1).reshape(-1)
    offs = np.stack([rows[fo].astype(np.int64) for fo, _ in off_len_fields], 1).reshape(-1)
    new_off = np.concatenate([[0], np.cumsum(lens)])
    total = int(new_off[-1])
    src = np.frombuffer(bytes(blob), dtype=np.uint8)
    if not total:
        out = b""
    else:
        # the strings of consecutive rows are nearly always neighbours in the source blob too (a rank keeps stretches of a
        # table): copy stretch by stretch; only a table in scattered order needs an index per byte
        nz = lens > 0
        o, l = offs[nz], lens[nz]
        breaks = np.flatnonzero(o[1:] != o[:-1] + l[:-1]) + 1
        if len(breaks) < 512:
            first = np.concatenate([[0], breaks])
            last = np.concatenate([breaks, [len(o)]]) - 1
            out = b"".join(src[int(o[a]):int(o[z] + l[z])].tobytes() for a, z in zip(first, last))
        else:
            out = src[np.repeat(offs - new_off[:-1], lens) + np.arange(total, dtype=np.int64)].tobytes()
    placed = new_off[:-1].reshape(n, len(off_len_fields))
    for j, (fo, _) in enumerate(off_len_fields):
        rows[fo] = placed[:, j]
    return rows, out


_FLANK_STR = (("seq_off", "seq_len"),)
_PAIR_STR = (("ir3_off", "ir3_len"), ("ir5_off", "ir5_len"), ("dr_ref_off", "dr_ref_len"), ("dr_reads_off", "dr_len"))   # the order the library lays a pair's strings out in


def shard_findflanks(binding, reads_struct, plan, params):
    """Stage 1 of one rank: ``mfk_findflanks`` on the rank's reads, keep the flank rows the rank owns.
    Returns (rows, seqs) with the sequences packed in row order."""
    return _own_flanks(binding.run_findflanks(reads_struct, params), plan)


def _own_flanks(res, plan):
    fl = res["flanks"]
    so = _tables()
    if so is not None and len(fl):
        import ctypes as C
        fl = np.ascontiguousarray(fl, dtype=_lib.FLANK_DTYPE)
        bounds = _own_bounds(plan, _n_contigs_of(plan, fl["contig_id"]))
        if bounds is not None:
            seqs = res["seqs"]
            sb = np.frombuffer(seqs, dtype=np.uint8) if len(seqs) else np.zeros(1, np.uint8)
            out = np.empty(len(fl), dtype=_lib.FLANK_DTYPE)
            ob = np.empty(max(len(seqs), 1), dtype=np.uint8)
            nb = C.c_int64(0)
            k = so.mfk_flanks_owned(fl.ctypes.data, len(fl), sb.ctypes.data, bounds[0].ctypes.data, bounds[1].ctypes.data, len(bounds[0]),
                                    out.ctypes.data, ob.ctypes.data, C.byref(nb))
            return out[:k], ob[:nb.value].tobytes()
    mine = _fast_take(fl, np.flatnonzero(plan.owns(fl["contig_id"], fl["pos"]))) if len(fl) else fl
    return _compact(mine, res["seqs"], _FLANK_STR)


def merge_flanks(parts, compacted=False):
    """Per-rank (rows, seqs) -> the flank table of the unsharded run: contigs in BAM header order (first-seen order of a
    sorted BAM, findflanks.py:640-666), positions ascending, 5p before 3p (kept from the rank that owns the position)."""
    so = _tables()
    if so is not None and compacted and parts:
        tab = _concat_rows([r for r, _ in parts], _lib.FLANK_DTYPE)                              # (a copy: the offsets are re-based in place)
        counts = np.array([[len(r), len(b)] for r, b in parts], dtype=np.int64)
        blob = b"".join(bytes(b) for _, b in parts)
        if so.mfk_flanks_merge(tab.ctypes.data, counts.ctypes.data, len(parts)):
            return tab, blob
        order = _stable_order(tab["contig_id"], tab["pos"])                                      # whole-contig shards: sort, then compact
        return _compact(tab if order is None else _fast_take(tab, order), blob, _FLANK_STR)
    rows, blobs, off = [], [], 0
    for r, b in parts:
        r = _fast_copy(np.asarray(r, dtype=_lib.FLANK_DTYPE))
        r["seq_off"] += off
        off += len(b)
        rows.append(r)
        blobs.append(bytes(b))
    tab = np.concatenate(rows) if rows else np.zeros(0, dtype=_lib.FLANK_DTYPE)
    order = _stable_order(tab["contig_id"], tab["pos"])           # stable: the 5p / 3p order inside a position survives
    if order is None and compacted:
        return tab, b"".join(blobs)                               # parts in order, each with its strings in row order: done
    return _compact(tab if order is None else _fast_take(tab, order), b"".join(blobs), _FLANK_STR)


def _flank_rows_for(plan, tab, reach):
    """Rows of the merged flank table a rank hands to ``mfk_pairflanks``: on each contig it owns a piece of, every run of
    flanks with gaps <= reach (= maxdr + 1: candidate pairs, and through shared flanks the greedy assignment, never cross a
    wider gap, pairflanks.py:155-157, 229-245) that holds an owned position."""
    if len(tab) == 0:
        return np.zeros(0, dtype=np.int64)
    so = _tables()
    if so is not None:
        tab_c = np.ascontiguousarray(tab, dtype=_lib.FLANK_DTYPE)
        bounds = _own_bounds(plan, _n_contigs_of(plan, tab_c["contig_id"]))
        if bounds is not None:
            sel = np.empty(len(tab_c), dtype=np.int64)
            k = so.mfk_flank_runs_owned(tab_c.ctypes.data, len(tab_c), bounds[0].ctypes.data, bounds[1].ctypes.data, len(bounds[0]), int(reach), sel.ctypes.data)
            return sel[:k]
    cid, pos = tab["contig_id"], tab["pos"].astype(np.int64)
    new_run = np.ones(len(tab), dtype=bool)
    new_run[1:] = (cid[1:] != cid[:-1]) | (pos[1:] - pos[:-1] > reach)
    run = np.cumsum(new_run) - 1
    own = plan.owns(cid, pos)
    keep = np.isin(run, np.unique(run[own]))
    return np.flatnonzero(keep)


def shard_pairflanks(binding, reads_struct, plan, params, tab, seqs, contig_rank, genome, genome_off, _timing=False):
    """Stage 2 of one rank: ``mfk_pairflanks`` on the merged rows around the rank's range (its reads cover the spanning
    fetches and direct-repeat windows of every pair whose 5p flank it owns: halo >= maxdr + 2 + half a window), keep those
    pairs; flank indices are rows of the merged table (quirk P2).  Returns (rows, strs)."""
    import time
    t0 = time.perf_counter()
    sel = _flank_rows_for(plan, tab, int(params.max_direct_repeat_length) + 1)
    if len(sel) == 0:
        return np.zeros(0, dtype=_lib.PAIR_DTYPE), b""
    rows_in = _fast_take(tab, sel)
    t1 = time.perf_counter()
    res = binding.run_pairflanks(reads_struct, params, rows_in, seqs, contig_rank, genome, genome_off)
    t2 = time.perf_counter()
    pr = res["pairs"]
    so = _tables()
    bounds = _own_bounds(plan, _n_contigs_of(plan, pr["contig_id"])) if so is not None and len(pr) else None
    if bounds is not None:
        import ctypes as C
        pr = np.ascontiguousarray(pr, dtype=_lib.PAIR_DTYPE)
        strs = res["strs"]
        sb = np.frombuffer(strs, dtype=np.uint8) if len(strs) else np.zeros(1, np.uint8)
        sel64 = np.ascontiguousarray(sel, dtype=np.int64)
        rows_out = np.empty(len(pr), dtype=_lib.PAIR_DTYPE)
        ob = np.empty(max(len(strs), 1), dtype=np.uint8)
        nb = C.c_int64(0)
        k = so.mfk_pairs_owned(pr.ctypes.data, len(pr), sb.ctypes.data, bounds[0].ctypes.data, bounds[1].ctypes.data, len(bounds[0]),
                               sel64.ctypes.data, rows_out.ctypes.data, ob.ctypes.data, C.byref(nb))
        out = rows_out[:k], ob[:nb.value].tobytes()
    else:
        mine = _fast_take(pr, np.flatnonzero(plan.owns(pr["contig_id"], pr["pos_5p"]))) if len(pr) else pr
        mine["index_5p"] = sel[mine["index_5p"]]
        mine["index_3p"] = sel[mine["index_3p"]]
        out = _compact(mine, res["strs"], _PAIR_STR)
    if _timing:
        STATS["merge_s"] += (t1 - t0) + (time.perf_counter() - t2)
    return out


def merge_pairs(parts, contig_rank, compacted=False):
    """Per-rank (rows, strs) -> the pair table of the unsharded run: sorted by (contig name, pos_5p, pos_3p)
    (pairflanks.py:248)."""
    so = _tables()
    if so is not None and compacted and parts:
        tab = _concat_rows([r for r, _ in parts], _lib.PAIR_DTYPE)
        counts = np.array([[len(r), len(b)] for r, b in parts], dtype=np.int64)
        blob = b"".join(bytes(b) for _, b in parts)
        cr = np.ascontiguousarray(contig_rank, dtype=np.int32)
        if so.mfk_pairs_merge(tab.ctypes.data, counts.ctypes.data, len(parts), cr.ctypes.data, len(cr)):
            return tab, blob
        rank = cr[tab["contig_id"]] if len(tab) else np.zeros(0, np.int64)
        order = _stable_order(rank, tab["pos_5p"], tab["pos_3p"])
        return _compact(tab if order is None else _fast_take(tab, order), blob, _PAIR_STR)
    rows, blobs, off = [], [], 0
    for r, b in parts:
        r = _fast_copy(np.asarray(r, dtype=_lib.PAIR_DTYPE))
        for fo, _ in _PAIR_STR:
            r[fo] += off
        off += len(b)
        rows.append(r)
        blobs.append(bytes(b))
    tab = np.concatenate(rows) if rows else np.zeros(0, dtype=_lib.PAIR_DTYPE)
    rank = np.asarray(contig_rank)[tab["contig_id"]] if len(tab) else np.zeros(0, np.int64)
    order = _stable_order(rank, tab["pos_5p"], tab["pos_3p"])
    if order is None and compacted:
        return tab, b"".join(blobs)
    return _compact(tab if order is None else _fast_take(tab, order), b"".join(blobs), _PAIR_STR)


def run_sharded_sequential(binding, reads_of, packed, params, world_size, sharding, contig_rank, genome, genome_off,
                           plans=None):
    """Every rank of a ``world_size`` job one after the other in this process: the sharded path without a second GPU.
    ``reads_of(PackedReads)`` gives the reads struct of ``binding`` (device tensors for HIP).  Returns
    ((flank rows, seqs), (pair rows, strs), plans)."""
    h = halo(params.min_distance_to_mate, params.max_direct_repeat_length)
    if plans is None:
        plans = plan_shards(packed, world_size, sharding, h)
    structs = [reads_of(packed.take_ranges(p.ranges)) for p in plans]
    tab, seqs = merge_flanks([shard_findflanks(binding, s, p, params) for s, p in zip(structs, plans)], compacted=True)
    pairs = merge_pairs([shard_pairflanks(binding, s, p, params, tab, seqs, contig_rank, genome, genome_off)
                         for s, p in zip(structs, plans)], contig_rank, compacted=True)
    return (tab, seqs), pairs, plans


# ---------------------------------------------------------------------------------------------------------------------
# the merge collective
# ---------------------------------------------------------------------------------------------------------------------
GATHER_CAP = 64 * 1024     # bytes per rank that travel in the first (normally only) collective

# what the exchanges of this process have cost so far (bench.py reports them per step): seconds inside the all-gathers,
# seconds of host-side selection / merging around them, and the number of ranks the last gather returned tables of
STATS = {"gather_s": 0.0, "merge_s": 0.0, "gathers": 0, "ranks_seen": 1}


def reset_stats():
    STATS.update(gather_s=0.0, merge_s=0.0, gathers=0, ranks_seen=1)


def _all_gather_bytes(mine, ws):
    """all-gather of equally sized uint8 tensors -> (ws, n) tensor."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        out = torch.empty((ws, mine.numel()), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(out, mine)
        return out
    parts = [torch.zeros_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine)
    return torch.stack(parts)


def init_library_comm(binding):
    """One RCCL communicator inside libmfk.so for the ranks of the initialised torch.distributed group: rank 0 makes the
    128-byte id, torch.distributed broadcasts it (the library has no bootstrap of its own), every rank joins.  After it
    allgather_records(..., binding=binding) runs the merge inside the library (mfk_allgather_sites).  Collective."""
    import torch
    import torch.distributed as dist
    ws, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    cid = torch.from_numpy(binding.make_comm_id() if rank == 0 else np.zeros(128, dtype=np.uint8)).to(dev)
    dist.broadcast(cid, src=0)
    binding.init_comm(cid.cpu().numpy(), rank, ws)


def allgather_records(rows, blob, device=None, cap=GATHER_CAP, binding=None):
    import time
    t0 = time.perf_counter()
    parts = _allgather_records(rows, blob, device, cap, binding)
    STATS["gather_s"] += time.perf_counter() - t0
    STATS["gathers"] += 1
    STATS["ranks_seen"] = len(parts)
    return parts


def _allgather_records(rows, blob, device=None, cap=GATHER_CAP, binding=None):
    """All-gather of one table per rank (fixed-width rows + their string bytes) -> list over ranks of (rows, blob).
    Without torch.distributed initialised (or with one rank) it is [(rows, blob)].

    One collective in the common case: every rank sends a 16-byte header (row count, string bytes) followed by its
    rows and strings, padded to `cap` bytes.  Only if some rank's table is larger than `cap` does a second all-gather
    with the true maximum follow (all ranks see the same headers, so they agree on that without talking)."""
    import torch
    import torch.distributed as dist
    if binding is not None and getattr(binding, "comm_world", 0) > 0:
        # inside the library: one ncclAllGather on the ctx stream; the library sizes the exchange after the previous one unless
        # the caller asks for a size of its own
        return binding.run_allgather(rows, blob, 0 if cap == GATHER_CAP else cap)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(rows, blob)]
    ws = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    rec = rows.dtype.itemsize
    body = np.ascontiguousarray(rows).tobytes() + bytes(blob)
    head = np.array([rows.shape[0], len(blob)], dtype=np.int64).tobytes()

    def exchange(size):
        buf = np.zeros(16 + size, dtype=np.uint8)
        buf[:16] = np.frombuffer(head, dtype=np.uint8)
        n = min(len(body), size)
        buf[16:16 + n] = np.frombuffer(body[:n], dtype=np.uint8)
        return _all_gather_bytes(torch.from_numpy(buf).to(dev), ws).cpu().numpy()

    got = exchange(cap)
    counts = got[:, :16].copy().view(np.int64).reshape(ws, 2)
    need = int((counts[:, 0] * rec + counts[:, 1]).max())
    if need > cap:
        got = exchange(need)
    parts = []
    for r in range(ws):
        nr, nb = int(counts[r, 0]), int(counts[r, 1])
        g = got[r, 16:]
        parts.append((np.frombuffer(g[:nr * rec].tobytes(), dtype=rows.dtype).copy(), g[nr * rec:nr * rec + nb].tobytes()))
    return parts


def allgather_flanks(flanks, seqs, device=None, cap=GATHER_CAP, binding=None):
    """Merge per-rank flank tables (FLANK_DTYPE rows + their sequence bytes).  Returns (rows, seqs) with seq_off
    rebased, ranks concatenated in rank order.  Works without torch.distributed initialised (single process)."""
    parts = allgather_records(flanks, seqs, device, cap, binding)
    if len(parts) == 1:
        return parts[0]
    rows, blobs, off = [], [], 0
    for part, b in parts:
        part["seq_off"] += off
        rows.append(part)
        blobs.append(b)
        off += len(b)
    return np.concatenate(rows), b"".join(blobs)


def findflanks_sharded(binding, reads_struct, plan, params, device=None, cap=GATHER_CAP):
    """Stage 1 of the calling rank + the all-gather of the owned rows -> the merged flank table on every rank."""
    import time
    res = binding.run_findflanks(reads_struct, params)
    t0 = time.perf_counter()
    rows, seqs = _own_flanks(res, plan)
    t1 = time.perf_counter()
    parts = allgather_records(rows, seqs, device, cap, binding if getattr(binding, "comm_world", 0) else None)
    t2 = time.perf_counter()
    out = merge_flanks(parts, compacted=True)
    STATS["merge_s"] += (t1 - t0) + (time.perf_counter() - t2)
    return out


def pairflanks_sharded(binding, reads_struct, plan, params, tab, seqs, contig_rank, genome, genome_off, device=None,
                       cap=GATHER_CAP):
    """Stage 2 of the calling rank + the all-gather of its pairs -> the merged pair table on every rank."""
    import time
    t0 = time.perf_counter()
    rows, strs = shard_pairflanks(binding, reads_struct, plan, params, tab, seqs, contig_rank, genome, genome_off, _timing=True)
    parts = allgather_records(rows, strs, device, cap, binding if getattr(binding, "comm_world", 0) else None)
    t2 = time.perf_counter()
    out = merge_pairs(parts, contig_rank, compacted=True)
    STATS["merge_s"] += time.perf_counter() - t2
    return out
