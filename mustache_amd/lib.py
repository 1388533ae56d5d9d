"""ctypes binding of the C-ABI declared in ``include/mfk.h``.

``load()`` opens the in-tree HIP library ``mustache_amd/csrc/libmfk.so`` and fails loudly when it is missing or when no
GPU is present: there is no CPU fallback in the product path.  ``Binding`` itself is a thin, backend-agnostic wrapper
(symbol prefix + shared object path) so that the test-only CPU oracle, which exports the same signatures under the
``mfko_`` prefix, can be driven by the same code from ``tests/`` -- the product never loads it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

N_STATS = 16

STAGE_PARSE, STAGE_MINLENGTH, STAGE_MINCOUNT, STAGE_MINDIST1, STAGE_UNCLIPPED, STAGE_RATIOS, STAGE_MINDIST2, \
    STAGE_CONSENSUS, STAGE_ALL = range(1, 10)

_p = C.c_void_p


BGZF_BLOCK_DTYPE = np.dtype([("comp_off", np.uint64), ("out_off", np.uint64), ("comp_len", np.uint32), ("out_len", np.uint32)])   # mfk_bgzf_block


class MfkReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64),
                ("ref_start", _p), ("ref_end", _p), ("link", _p), ("contig_id", _p), ("qlen", _p),
                ("first_op_len", _p), ("last_op_len", _p), ("mapq", _p), ("flags", _p),
                ("n_contigs", C.c_int32), ("max_span", C.c_int32),
                ("h_contig_len", _p), ("h_contig_read_start", _p),
                ("n_cx", C.c_int64), ("cx_read", _p), ("cx_off", _p),
                ("n_blk", C.c_int64), ("blk_start", _p), ("blk_end", _p), ("blk_ins", _p),
                ("n_pool", C.c_int64), ("pool_read", _p), ("pool_off", _p), ("pool_bases", _p), ("pool_quals", _p),
                ("n_clip", C.c_int64), ("clip_rec", _p), ("clip_tile", _p)]


class MfkParams(C.Structure):
    _fields_ = [("min_softclip_length", C.c_int32), ("min_softclip_count", C.c_int32),
                ("min_alignment_quality", C.c_int32), ("min_alignment_inner_length", C.c_int32),
                ("min_distance_to_mate", C.c_int32), ("large_insertion_cutoff", C.c_int32),
                ("min_count_consensus", C.c_int32), ("max_direct_repeat_length", C.c_int32),
                ("min_softclip_ratio", C.c_double), ("max_indel_ratio", C.c_double),
                ("max_junction_spanning_prop", C.c_double)]


SITE_DTYPE = np.dtype([(n, np.int32) for n in (
    "contig_id", "pos", "softclip_count_5p", "softclip_count_3p", "runthrough_count", "small_insertion_count_5p",
    "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p", "deletion_count",
    "upstream_deletion_count", "downstream_deletion_count", "keep_5p", "keep_3p")])

FLANK_DTYPE = np.dtype([(n, np.int32) for n in (
    "contig_id", "pos", "orient", "softclip_count_5p", "softclip_count_3p", "runthrough_count",
    "small_insertion_count_5p", "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p",
    "deletion_count", "upstream_deletion_count", "downstream_deletion_count", "total_count",
    "consensus_softclip_count", "seq_len")] + [("seq_off", np.int64)])

PAIR_DTYPE = np.dtype([(n, np.int32) for n in (
    "contig_id", "pos_5p", "pos_3p", "index_5p", "index_3p", "softclip_count_5p", "softclip_count_3p",
    "total_count_5p", "total_count_3p", "spanning_count", "has_ir", "ir_length", "ir5_len", "ir3_len", "dr_len",
    "dr_ref_len")] + [(n, np.int64) for n in ("ir5_off", "ir3_off", "dr_ref_off", "dr_reads_off")])

ALN_DTYPE = np.dtype([(n, np.int32) for n in (
    "group", "name", "mate", "is_reverse", "contig_id", "ref_start", "ref_end", "contig_len", "qlen", "left_clip",
    "right_clip", "aln_len", "nm", "as")])

assert ALN_DTYPE.itemsize == 56 and SITE_DTYPE.itemsize == 56 and FLANK_DTYPE.itemsize == 72 and PAIR_DTYPE.itemsize == 96

# CLI defaults are authoritative (reference main.py:68-78, 102-107), not the class defaults
RECALL_COUNTS = ("softclip_count_5p", "softclip_count_3p", "runthrough_count", "small_insertion_count_5p",
                 "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p", "deletion_count",
                 "upstream_deletion_count", "downstream_deletion_count")

DEFAULT_PARAMS = dict(min_softclip_length=8, min_softclip_count=2, min_alignment_quality=20,
                      min_alignment_inner_length=21, min_distance_to_mate=22, large_insertion_cutoff=30,
                      min_count_consensus=2, max_direct_repeat_length=20, min_softclip_ratio=0.10,
                      max_indel_ratio=0.03, max_junction_spanning_prop=0.15)


def make_params(**kw):
    d = dict(DEFAULT_PARAMS)
    for k, v in kw.items():
        if k not in d:
            raise TypeError("unknown parameter %r" % k)
        d[k] = v
    return MfkParams(**d)


class MfkError(RuntimeError):
    pass


def _ptr(x):
    """Device or host address of a tensor / array (None -> NULL)."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data


class Binding:
    """One ctx of one shared object exporting the include/mfk.h signatures under ``prefix``."""

    def __init__(self, path, prefix, device_id=0):
        if not os.path.exists(path):
            raise MfkError("shared library %s is missing: build it first (python -c 'import __graft_entry__ as g; "
                           "g.build()')" % path)
        self.path, self.prefix = path, prefix
        self.so = C.CDLL(path)
        f = self._fn
        f("create", _p, [C.c_int])
        f("destroy", None, [_p])
        f("last_error", C.c_char_p, [_p])
        f("version", C.c_int, [])
        f("last_device_ms", C.c_double, [_p])
        f("findflanks", C.c_int, [_p, C.POINTER(MfkReads), C.POINTER(MfkParams), C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
        f("findflanks_fetch", C.c_int, [_p, _p, _p, _p, _p])
        f("pairflanks", C.c_int, [_p, C.POINTER(MfkReads), C.POINTER(MfkParams), _p, _p, C.c_int64, _p, _p, _p,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
        f("pairflanks_fetch", C.c_int, [_p, _p, _p, _p])
        if hasattr(self.so, prefix + "flankpairs"):
            f("flankpairs", C.c_int, [_p, C.POINTER(MfkReads), C.POINTER(MfkParams), _p, _p, _p] + [C.POINTER(C.c_int64)] * 5)
        f("ir_detect", C.c_int, [_p, _p, _p, C.c_int64, _p])
        f("recall", C.c_int, [_p, C.POINTER(MfkReads), C.POINTER(MfkParams), _p, _p, C.c_int64, _p])
        if hasattr(self.so, prefix + "recall_noalloc"):
            f("recall_noalloc", C.c_int, [_p, C.POINTER(MfkReads), C.POINTER(MfkParams), _p, _p, C.c_int64, _p])
        f("overlap", C.c_int, [_p, _p, _p, _p, _p, C.c_int64, _p])
        f("pairflanks_set_ir", C.c_int, [_p, C.c_int64, _p, _p, _p])
        f("inferseq_pairs", C.c_int, [_p, _p, C.c_int64, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(C.c_int64)])
        f("inferseq_pairs_fetch", C.c_int, [_p, _p, _p])
        f("inferseq_db_pairs", C.c_int, [_p, _p, C.c_int64, C.c_double, C.c_double, C.c_int64, C.POINTER(C.c_int64)])
        self.comm_world = 0
        if hasattr(self.so, prefix + "comm_init"):
            f("comm_unique_id", C.c_int, [_p, _p])
            f("comm_init", C.c_int, [_p, _p, C.c_int, C.c_int])
            f("comm_free", C.c_int, [_p])
            f("allgather_sites", C.c_int, [_p, _p, C.c_int64, C.c_int32, _p, C.c_int64, C.c_int64, _p])
            f("allgather_fetch", C.c_int, [_p, C.c_int, _p, _p])
            f("allgather_fetch_all", C.c_int, [_p, _p, _p])
            f("allgather_next_bytes", C.c_int64, [_p])
        if hasattr(self.so, prefix + "bgzf_inflate"):
            f("bgzf_inflate", C.c_int, [_p, _p, C.c_int64, _p, C.c_int64, _p, C.c_int64, _p, C.POINTER(C.c_int64)])
        if hasattr(self.so, prefix + "decode_bam"):
            f("bgzf_block_table", C.c_int, [_p, _p, C.c_int64, _p, _p, _p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
            f("bgzf_crc32", C.c_int, [_p, _p, C.c_int64, _p, C.c_int64, _p])
            f("decode_bam", C.c_int, [_p, _p, C.c_int64, _p, _p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, _p])
            f("decode_bam_fetch", C.c_int, [_p, C.POINTER(MfkReads), _p, C.POINTER(C.c_int32), C.c_int64, _p, _p])
        if hasattr(self.so, prefix + "span_timing"):
            fn = getattr(self.so, prefix + "span_timing")
            fn.restype, fn.argtypes = C.c_int, [_p, C.c_int]
            self._span_timing = fn
        if hasattr(self.so, prefix + "profile_enable"):
            f("profile_enable", C.c_int, [_p, C.c_int])
            f("profile_fetch", C.c_int, [_p, _p, C.c_int64, _p, _p, C.c_int64, C.POINTER(C.c_int64)])
            f("profile_select", C.c_int, [_p, C.c_char_p])
        self.last_stats = self.last_pair_stats = np.zeros(N_STATS, dtype=np.int64)   # counters of the last compute call
        self.ctx = self.create(device_id)
        if not self.ctx:
            raise MfkError("%screate(%d) failed: no usable HIP device for %s" % (prefix, device_id, path))

    def _fn(self, name, restype, argtypes):
        fn = getattr(self.so, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        setattr(self, name, fn)

    def close(self):
        if getattr(self, "ctx", None):
            self.destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.last_error(self.ctx)
            raise MfkError("%s error %d: %s" % (self.prefix, rc, msg.decode() if msg else "?"))

    # ------------------------------------------------------------------------------------------------------
    def reads_struct(self, arrays, n_reads, max_span, host_hdr):
        """arrays: dict name -> torch tensor (device) or numpy array (host oracle).  host_hdr: the numpy
        contig_len / contig_read_start arrays (always host).  Keeps references alive on the returned struct."""
        r = MfkReads()
        r.n_reads = n_reads
        for k in ("ref_start", "ref_end", "link", "contig_id", "qlen", "first_op_len", "last_op_len", "mapq", "flags",
                  "cx_read", "cx_off", "blk_start", "blk_end", "blk_ins", "pool_read", "pool_off", "pool_bases",
                  "pool_quals"):
            setattr(r, k, _ptr(arrays[k]))
        clen = np.ascontiguousarray(host_hdr["contig_len"], dtype=np.int32)
        crs = np.ascontiguousarray(host_hdr["contig_read_start"], dtype=np.int64)
        r.n_contigs = int(clen.shape[0])
        r.max_span = int(max_span)
        r.h_contig_len, r.h_contig_read_start = clen.ctypes.data, crs.ctypes.data
        r.n_cx = int(host_hdr["n_cx"])
        r.n_blk = int(host_hdr["n_blk"])
        r.n_pool = int(host_hdr["n_pool"])
        if arrays.get("clip_rec") is not None:     # optional dense records of the clipped reads (pack.PackedReads.clip_records)
            r.n_clip = int(arrays["n_clip"])
            r.clip_rec, r.clip_tile = _ptr(arrays["clip_rec"]), _ptr(arrays["clip_tile"])
        r._keep = (arrays, clen, crs)
        return r

    # ---- BGZF inflate on the device (include/mfk.h mfk_bgzf_inflate) ---------------------------------------------------------
    def run_bgzf_inflate(self, d_comp, blocks, d_out, want_status=False):
        """d_comp / d_out: uint8 device tensors; blocks: numpy array of BGZF_BLOCK_DTYPE.  Raises MfkError when a block does
        not inflate (after filling ``status`` when asked for).  -> (n_bad, status or None)."""
        blocks = np.ascontiguousarray(blocks, dtype=BGZF_BLOCK_DTYPE)
        status = np.zeros(blocks.shape[0], dtype=np.int32) if want_status else None
        bad = C.c_int64()
        rc = self.bgzf_inflate(self.ctx, _ptr(d_comp), int(d_comp.numel()), blocks.ctypes.data, blocks.shape[0], _ptr(d_out), int(d_out.numel()),
                               status.ctypes.data if want_status else None, C.byref(bad))
        if rc not in (0, 12) or (rc == 12 and not want_status):
            self._check(rc)
        return int(bad.value), status

    # ---- multi-GPU merge inside the library (RCCL; include/mfk.h mfk_comm_* / mfk_allgather_sites) ----------------------
    def make_comm_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        self._check(self.comm_unique_id(self.ctx, buf.ctypes.data))
        return buf

    def init_comm(self, comm_id, rank, world):
        """Collective over the `world` ranks that hold the same id (made by rank 0's make_comm_id)."""
        cid = np.ascontiguousarray(comm_id, dtype=np.uint8)
        assert cid.shape[0] == 128
        self._check(self.comm_init(self.ctx, cid.ctypes.data, int(rank), int(world)))
        self.comm_world = int(world)

    def free_comm(self):
        if self.comm_world:
            self._check(self.comm_free(self.ctx))
            self.comm_world = 0

    def run_allgather(self, rows, blob, pad_bytes=0):
        """All-gather of one table per rank: rows = structured numpy array, blob = bytes.  -> [(rows, blob)] per rank."""
        rows = np.ascontiguousarray(rows)
        b = np.frombuffer(bytes(blob) + b"\0", dtype=np.uint8)
        counts = np.zeros(2 * self.comm_world, dtype=np.int64)
        self._check(self.allgather_sites(self.ctx, rows.ctypes.data, rows.shape[0], rows.dtype.itemsize, b.ctypes.data, len(blob),
                                         int(pad_bytes), counts.ctypes.data))
        # one fetch for all ranks, then views per rank (a step of the scaling bench gathers eight tables of a few dozen rows)
        nr, nb = counts[0::2], counts[1::2]
        allr = np.empty(int(nr.sum()), dtype=rows.dtype)
        allb = np.empty(max(int(nb.sum()), 1), dtype=np.uint8)
        self._check(self.allgather_fetch_all(self.ctx, allr.ctypes.data, allb.ctypes.data))
        blob_all = allb.tobytes()
        r0 = np.concatenate([[0], np.cumsum(nr)])
        b0 = np.concatenate([[0], np.cumsum(nb)])
        return [(allr[int(r0[r]):int(r0[r + 1])], blob_all[int(b0[r]):int(b0[r + 1])]) for r in range(self.comm_world)]

    def run_findflanks(self, reads, params, stop_after=STAGE_ALL):
        ns, nf, sb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self.findflanks(self.ctx, C.byref(reads), C.byref(params), stop_after, C.byref(ns), C.byref(nf),
                                    C.byref(sb)))
        sites = np.empty(ns.value, dtype=SITE_DTYPE)       # (the fetch writes every row)
        flanks = np.empty(nf.value, dtype=FLANK_DTYPE)
        seqs = np.empty(max(sb.value, 1), dtype=np.uint8)
        stats = np.zeros(N_STATS, dtype=np.int64)
        self._check(self.findflanks_fetch(self.ctx, sites.ctypes.data, flanks.ctypes.data, seqs.ctypes.data,
                                          stats.ctypes.data))
        self.last_stats = stats
        return {"sites": sites, "flanks": flanks, "seqs": seqs[:sb.value].tobytes(), "stats": stats}

    def run_pairflanks(self, reads, params, flanks, flank_seqs, contig_rank, genome, genome_off):
        """flanks: FLANK_DTYPE array (host); flank_seqs: bytes; genome: device tensor / host array of bytes."""
        flanks = np.ascontiguousarray(flanks, dtype=FLANK_DTYPE)
        fs = np.frombuffer(flank_seqs + b"\0", dtype=np.uint8)
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        goff = np.ascontiguousarray(genome_off, dtype=np.int64)
        npairs, sb = C.c_int64(0), C.c_int64(0)
        self._check(self.pairflanks(self.ctx, C.byref(reads), C.byref(params), flanks.ctypes.data, fs.ctypes.data,
                                    flanks.shape[0], rank.ctypes.data, _ptr(genome), goff.ctypes.data,
                                    C.byref(npairs), C.byref(sb)))
        pairs = np.empty(npairs.value, dtype=PAIR_DTYPE)
        strs = np.zeros(max(sb.value, 1), dtype=np.uint8)
        stats = np.zeros(N_STATS, dtype=np.int64)
        self._check(self.pairflanks_fetch(self.ctx, pairs.ctypes.data, strs.ctypes.data, stats.ctypes.data))
        self.last_pair_stats = stats
        return {"pairs": pairs, "strs": strs[:sb.value].tobytes(), "stats": stats}

    def run_flankpairs(self, reads, params, contig_rank, genome, genome_off):
        """findflanks + pairflanks in one library call (mfk_flankpairs) -> (findflanks result, pairflanks result), the
        same dicts ``run_findflanks`` / ``run_pairflanks`` return."""
        rank = np.ascontiguousarray(contig_rank, dtype=np.int32)
        goff = np.ascontiguousarray(genome_off, dtype=np.int64)
        ns, nf, sb, npairs, pb = (C.c_int64(0) for _ in range(5))
        self._check(self.flankpairs(self.ctx, C.byref(reads), C.byref(params), rank.ctypes.data, _ptr(genome), goff.ctypes.data,
                                    C.byref(ns), C.byref(nf), C.byref(sb), C.byref(npairs), C.byref(pb)))
        sites = np.empty(ns.value, dtype=SITE_DTYPE)       # (the fetch writes every row)
        flanks = np.empty(nf.value, dtype=FLANK_DTYPE)
        seqs = np.empty(max(sb.value, 1), dtype=np.uint8)
        stats = np.zeros(N_STATS, dtype=np.int64)
        self._check(self.findflanks_fetch(self.ctx, sites.ctypes.data, flanks.ctypes.data, seqs.ctypes.data, stats.ctypes.data))
        pairs = np.empty(npairs.value, dtype=PAIR_DTYPE)
        strs = np.empty(max(pb.value, 1), dtype=np.uint8)
        pstats = np.zeros(N_STATS, dtype=np.int64)
        self._check(self.pairflanks_fetch(self.ctx, pairs.ctypes.data, strs.ctypes.data, pstats.ctypes.data))
        self.last_stats, self.last_pair_stats = stats, pstats
        return ({"sites": sites, "flanks": flanks, "seqs": seqs[:sb.value].tobytes(), "stats": stats},
                {"pairs": pairs, "strs": strs[:pb.value].tobytes(), "stats": pstats})

    def set_pair_ir(self, index_5p, index_3p, ir):
        """Inverted-repeat results of an external detector for the next run_pairflanks (ir: [n, 5] has, a1s, a1e, a2s, a2e)."""
        i5 = np.ascontiguousarray(index_5p, dtype=np.int32)
        i3 = np.ascontiguousarray(index_3p, dtype=np.int32)
        t = np.ascontiguousarray(ir, dtype=np.int32).reshape(-1, 5)
        assert i5.shape[0] == i3.shape[0] == t.shape[0]
        self._check(self.pairflanks_set_ir(self.ctx, i5.shape[0], i5.ctypes.data, i3.ctypes.data, t.ctypes.data))

    def run_ir_detect(self, seqs):
        """seqs: list of str -> int32 array [n, 6]: has_ir, arm1 start/end, arm2 start/end (1-based), n_reported."""
        blob = "".join(seqs).encode() + b"\0"
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        out = np.zeros((len(seqs), 6), dtype=np.int32)
        buf = np.frombuffer(blob, dtype=np.uint8)
        self._check(self.ir_detect(self.ctx, buf.ctypes.data, off.ctypes.data, len(seqs), out.ctypes.data))
        return out

    def run_recall(self, reads, params, contig_ids, positions):
        """Per-site tallies at given (contig id, position) pairs -> int32 array [n, 10] in RECALL_COUNTS order."""
        cid = np.ascontiguousarray(contig_ids, dtype=np.int32)
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros((cid.shape[0], 10), dtype=np.int32)
        self._check(self.recall(self.ctx, C.byref(reads), C.byref(params), cid.ctypes.data, pos.ctypes.data, cid.shape[0],
                                out.ctypes.data))
        return out

    def run_recall_noalloc(self, reads, params, contig_ids, positions):
        """``run_recall`` through the entry point that allocates nothing (mfk_recall_noalloc; <= 16384 sites per call)."""
        cid = np.ascontiguousarray(contig_ids, dtype=np.int32)
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros((cid.shape[0], 10), dtype=np.int32)
        self._check(self.recall_noalloc(self.ctx, C.byref(reads), C.byref(params), cid.ctypes.data, pos.ctypes.data, cid.shape[0],
                                        out.ctypes.data))
        return out

    def run_overlap(self, seqs5, seqs3):
        """Best sliding overlap per pair of strings -> int32 array [n, 6]: best_score, best_score_mismatches, r_start,
        r_end, q_start, q_end (q = seqs5[i], r = seqs3[i])."""
        assert len(seqs5) == len(seqs3)
        def blob(seqs):
            b = [s.encode() for s in seqs]
            off = np.zeros(len(b) + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(x) for x in b])
            return np.frombuffer(b"".join(b) + b"\0", dtype=np.uint8), off
        b5, o5 = blob(seqs5)
        b3, o3 = blob(seqs3)
        out = np.zeros((len(seqs5), 6), dtype=np.int32)
        self._check(self.overlap(self.ctx, b5.ctypes.data, o5.ctypes.data, b3.ctypes.data, o3.ctypes.data, len(seqs5),
                                 out.ctypes.data))
        return out

    def run_inferseq_pairs(self, alns, min_perc_identity, max_internal_softclip_prop, max_inferseq_size, min_inferseq_size):
        """alns: ALN_DTYPE array in BAM order -> (forward, reverse) int32 index arrays of the surviving pairs."""
        a = np.ascontiguousarray(alns, dtype=ALN_DTYPE)
        n = C.c_int64(0)
        self._check(self.inferseq_pairs(self.ctx, a.ctypes.data, a.shape[0], float(min_perc_identity),
                                        float(max_internal_softclip_prop), int(max_inferseq_size), int(min_inferseq_size),
                                        C.byref(n)))
        fwd, rev = np.zeros(max(n.value, 1), dtype=np.int32), np.zeros(max(n.value, 1), dtype=np.int32)
        self._check(self.inferseq_pairs_fetch(self.ctx, fwd.ctypes.data, rev.ctypes.data))
        return fwd[:n.value], rev[:n.value]

    def run_inferseq_db_pairs(self, alns, min_perc_identity, max_internal_softclip_prop, max_edge_distance):
        """inferseq-database pairing: alns (ALN_DTYPE, BAM order) -> (forward-strand, reverse-strand) index arrays."""
        a = np.ascontiguousarray(alns, dtype=ALN_DTYPE)
        n = C.c_int64(0)
        self._check(self.inferseq_db_pairs(self.ctx, a.ctypes.data, a.shape[0], float(min_perc_identity),
                                           float(max_internal_softclip_prop), int(max_edge_distance), C.byref(n)))
        fwd, rev = np.zeros(max(n.value, 1), dtype=np.int32), np.zeros(max(n.value, 1), dtype=np.int32)
        self._check(self.inferseq_pairs_fetch(self.ctx, fwd.ctypes.data, rev.ctypes.data))
        return fwd[:n.value], rev[:n.value]

    def profile(self, on=True):
        """Start (and clear) or stop per-kernel HIP-event timing of the compute calls."""
        self._check(self.profile_enable(self.ctx, 1 if on else 0))

    def profile_only(self, names):
        """Time only the named kernels (empty / None = all); clears the totals."""
        self._check(self.profile_select(self.ctx, "\n".join(names or []).encode()))

    def profile_totals(self):
        """dict kernel name -> (total ms, launches) since profile(True)."""
        cap = 256
        names = C.create_string_buffer(1 << 16)
        ms = np.zeros(cap, dtype=np.float64)
        calls = np.zeros(cap, dtype=np.int64)
        n = C.c_int64(0)
        self._check(self.profile_fetch(self.ctx, C.cast(names, _p), len(names), ms.ctypes.data, calls.ctypes.data, cap,
                                       C.byref(n)))
        lst = names.value.decode().split("\n")[:n.value]
        return {nm: (float(ms[i]), int(calls[i])) for i, nm in enumerate(lst)}

    def span_timing(self, on=True):
        """Record (default) or skip the two events per stage behind device_ms(); off saves their host time in tight loops."""
        if hasattr(self, "_span_timing"):
            self._check(self._span_timing(self.ctx, 1 if on else 0))

    def device_ms(self):
        return float(self.last_device_ms(self.ctx))


def tune_allocator():
    """For processes that call the library in a loop (bench.py, the CLI over a cohort): the result tables of a call are megabytes
    of fresh numpy memory, which glibc hands out as new anonymous mappings and returns to the kernel when they are freed, so
    every call pays their page faults again (0.15 ms per 4 MB here).  Raising the mmap / trim thresholds keeps freed blocks in the
    heap.  Process-wide policy, so only entry points call it, never an import.  MFK_NO_MALLOC_TUNE=1 leaves malloc alone."""
    if os.environ.get("MFK_NO_MALLOC_TUNE", "0") not in ("", "0"):
        return False
    try:
        libc = C.CDLL(None)
        return bool(libc.mallopt(-3, 1 << 28)) and bool(libc.mallopt(-1, 1 << 28))      # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD
    except (OSError, AttributeError):
        return False


def host_header(packed):
    a = packed.a
    return {"contig_len": a["contig_len"], "contig_read_start": a["contig_read_start"],
            "n_cx": a["cx_read"].shape[0], "n_blk": a["blk_start"].shape[0], "n_pool": a["pool_read"].shape[0]}


# MFK_LIB_PATH: another build of the same library (diagnostic builds, e.g. `make SJ_STAMP=1`)
LIB_PATH = os.environ.get("MFK_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmfk.so")
_loaded = {}


def load(device_id=0):
    """The HIP library on ``device_id``.  Raises MfkError when the extension is not built or no GPU is usable."""
    if device_id not in _loaded:
        # torch owns the device buffers and must bring up ITS HIP runtime first: libmfk.so then binds to the same
        # libamdhip64 instead of loading a second runtime (which leaves torch with "No HIP GPUs are available")
        import torch
        if not torch.cuda.is_available():
            raise MfkError("no HIP device visible to torch: the flank kernels need an MI355X (there is no CPU path)")
        torch.cuda.init()
        _loaded[device_id] = Binding(LIB_PATH, "mfk_", device_id)
    return _loaded[device_id]
