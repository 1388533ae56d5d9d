"""Host-side packer: alignment records -> structure-of-arrays the HIP kernels (and the CPU oracle) consume.

This is the only place where CIGAR strings and MD tags are looked at.  It reproduces, per read, the predicates of the
reference's ``sctools.py`` so the device never needs strings:

* ``lkind`` / ``rkind`` -- "lenient" soft-clip kind of the left / right end (``sctools.py:17-31``):
  0 = not clipped, 1 = CIGAR ``S`` at that end, 2 = no ``S`` but the MD tag says the terminal aligned base is a
  mismatch (left: ``MD[0] == '0'``; right: ``MD[-1] == '0' and not MD[-2].isdigit()``).
* ``first_op_len`` / ``last_op_len`` -- raw length of the first / last CIGAR op *whatever its type* (quirk Q1,
  ``sctools.py:5-9``): this is what ``meets_minlength`` compares.
* site coordinates follow ``sctools.py:99-109``: right S -> ``reference_end``; right mismatch -> ``reference_end-1``;
  left S -> ``reference_start-1``; left mismatch -> ``reference_start``.  (The packer verifies that the first / last
  reference-consuming op is an aligned op, so first/last aligned position == reference_start / reference_end-1.)
* clipped sequence (``sctools.py:111-157``): kind 1 -> the ``S`` bases; kind 2 -> the single terminal *query* base.

Layout (one entry per mapped read, BAM order = coordinate order), 22 bytes per read (SURVEY 8d budgets 24: its
lclip_len / rclip_len columns are derivable from the kinds and op lengths, qlen replaces them):

    ref_start i32 | ref_end i32 | link i32 | contig_id u16 | qlen u16 | first_op_len u16 | last_op_len u16 |
    mapq u8 | flags u8

``link`` is the index of the previous read with the same query name (-1 if none): every "count distinct query names"
in the reference (quirk Q3) becomes "count reads whose name-chain holds no earlier qualifying read", which needs no
hashing or sorting on the device.

Side tables: ``cx_*`` / ``blk_*`` hold the aligned blocks of reads whose CIGAR is more than ``[H][S]M[S][H]``
(12 B per block); ``pool_*`` holds full query sequences (4-bit BAM base codes, two per byte, BAM nibble order, every read starting
on a byte boundary so BAM seq bytes copy straight in) and qualities for the reads
that need them (clipped reads for findflanks; reads over direct-repeat windows for pairflanks).
"""
from __future__ import annotations

import os

import numpy as np

from .records import (CDEL, CDIFF, CEQUAL, CHARD_CLIP, CINS, CMATCH, CPAD, CREF_SKIP, CSOFT_CLIP, BASE_TO_CODE)

FLAG_LKIND_SHIFT = 0
FLAG_RKIND_SHIFT = 2
FLAG_COMPLEX = 1 << 4
FLAG_POOL = 1 << 5

_HOT_FIELDS = (("ref_start", np.int32), ("ref_end", np.int32), ("link", np.int32), ("contig_id", np.uint16),
               ("qlen", np.uint16), ("first_op_len", np.uint16), ("last_op_len", np.uint16), ("mapq", np.uint8),
               ("flags", np.uint8))
_SIDE_FIELDS = (("contig_len", np.int32), ("contig_read_start", np.int64), ("cx_read", np.uint32),
                ("cx_off", np.uint32), ("blk_start", np.int32), ("blk_end", np.int32), ("blk_ins", np.int32),
                ("pool_read", np.uint32), ("pool_off", np.uint64), ("pool_bases", np.uint8), ("pool_quals", np.uint8))
ALL_FIELDS = _HOT_FIELDS + _SIDE_FIELDS
BYTES_PER_READ = sum(np.dtype(t).itemsize for _, t in _HOT_FIELDS)  # 22
# include/mfk.h mfk_clip_rec / MFK_CLIP_TILE
CLIP_REC = np.dtype([("read", np.uint32), ("ref_start", np.int32), ("ref_end", np.int32), ("contig_id", np.uint16),
                     ("mapq", np.uint8), ("inner", np.uint8)])
CLIP_TILE = 4096


class PackError(ValueError):
    pass


def clip_kinds(cigartuples, md):
    """(lkind, rkind) per ``sctools.is_{left,right}_softclipped_lenient`` (reference sctools.py:17-31)."""
    if cigartuples[0][0] == CSOFT_CLIP:
        lkind = 1
    else:
        if md is None:
            raise PackError("read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:28)")
        lkind = 2 if md[0] == "0" else 0
    if cigartuples[-1][0] == CSOFT_CLIP:
        rkind = 1
    else:
        if md is None:
            raise PackError("read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:20)")
        rkind = 2 if (md[-1] == "0" and len(md) >= 2 and not md[-2].isdigit()) else 0
    return lkind, rkind


def _blocks_with_insertions(read):
    """Aligned blocks as pysam ``get_blocks`` gives them, plus the number of inserted query bases in front of each."""
    blocks, ins, pos, pending = [], [], read.reference_start, 0
    for k, (op, l) in enumerate(read.cigartuples):
        if op in (CMATCH, CEQUAL, CDIFF):
            blocks.append((pos, pos + l))
            ins.append(pending)
            pending = 0
            pos += l
        elif op == CINS or (op == CSOFT_CLIP and k > 0 and not blocks):
            pending += l  # a leading S that is not the first op (after H) still occupies query bases
        elif op in (CDEL, CREF_SKIP):
            pos += l
    return blocks, ins


class PackedReads:
    """The SoA arrays plus header information.  Arrays are numpy on the host; ``to_torch`` moves them to a device."""

    def __init__(self, contig_names, arrays, max_span, read_names=None):
        self.contig_names = list(contig_names)
        self.a = arrays
        self.max_span = int(max_span)
        self.read_names = read_names  # optional, host only (debugging)

    @property
    def n_reads(self):
        return int(self.a["ref_start"].shape[0])

    def __getattr__(self, name):
        a = self.__dict__.get("a")
        if a is not None and name in a:
            return a[name]
        raise AttributeError(name)

    def nbytes_hot(self):
        return self.n_reads * BYTES_PER_READ

    def clip_records(self):
        """(clip_rec, clip_tile) of include/mfk.h: the lenient-clipped reads (flags & 15, sctools.py:17-31) restated as
        dense 16-byte records in read order, and the number of them before every 4096-read tile.  findflanks' event
        kernel reads these instead of gathering nine columns for one read in seventy.  Derived from the columns alone,
        so every producer of a PackedReads (both packers, slices, cohorts, the generator) gets the same table."""
        got = self.__dict__.get("_clip")
        if got is None:
            a = self.a
            f = a["flags"]
            clipped = (f & 15) != 0
            idx = np.flatnonzero(clipped)
            rec = np.zeros(idx.shape[0], dtype=CLIP_REC)
            rec["read"] = idx
            for k in ("ref_start", "ref_end", "contig_id", "mapq"):
                rec[k] = a[k][idx]
            fl = f[idx]
            lk, rk = fl & 3, (fl >> 2) & 3
            llen = np.where(lk == 1, a["first_op_len"][idx], 1).astype(np.int64)
            rlen = np.where(rk == 1, a["last_op_len"][idx], 1).astype(np.int64)
            inner = np.clip(a["qlen"][idx].astype(np.int64) - llen - rlen, 0, 255)
            rec["inner"] = np.where((lk != 0) & (rk != 0), inner, 255)
            n = self.n_reads
            per_tile = np.add.reduceat(clipped.astype(np.uint32), np.arange(0, n, CLIP_TILE)) if n else np.zeros(0, np.uint32)
            tile = np.zeros((n + CLIP_TILE - 1) // CLIP_TILE + 1, dtype=np.uint32)
            np.cumsum(per_tile, out=tile[1:])
            got = self.__dict__["_clip"] = (rec, tile)
        return got

    def save(self, path):
        np.savez(path, contig_names=np.array(self.contig_names), max_span=np.int64(self.max_span), **self.a)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        arrays = {k: z[k] for k, _ in ALL_FIELDS}
        return PackedReads([str(s) for s in z["contig_names"]], arrays, int(z["max_span"]))

    def slice_reads(self, lo, hi):
        """Reads [lo, hi) as a self-contained PackedReads (links pointing outside the slice become -1)."""
        a = self.a
        out = {}
        for k, _ in _HOT_FIELDS:
            out[k] = a[k][lo:hi].copy()
        link = out["link"].astype(np.int64) - lo
        link[link < 0] = -1
        out["link"] = link.astype(np.int32)
        out["contig_len"] = a["contig_len"].copy()
        out["contig_read_start"] = np.clip(a["contig_read_start"] - lo, 0, hi - lo).astype(np.int64)
        c0, c1 = np.searchsorted(a["cx_read"], [lo, hi])
        out["cx_read"] = (a["cx_read"][c0:c1] - lo).astype(np.uint32)
        b0, b1 = int(a["cx_off"][c0]), int(a["cx_off"][c1])
        out["cx_off"] = (a["cx_off"][c0:c1 + 1] - b0).astype(np.uint32)
        for k in ("blk_start", "blk_end", "blk_ins"):
            out[k] = a[k][b0:b1].copy()
        p0, p1 = np.searchsorted(a["pool_read"], [lo, hi])
        out["pool_read"] = (a["pool_read"][p0:p1] - lo).astype(np.uint32)
        o0, o1 = int(a["pool_off"][p0]), int(a["pool_off"][p1])
        ob = o0  # pooled reads start on byte boundaries (even base offsets)
        out["pool_off"] = (a["pool_off"][p0:p1 + 1] - ob).astype(np.uint64)
        out["pool_bases"] = a["pool_bases"][ob // 2:(o1 + 1) // 2].copy()
        out["pool_quals"] = a["pool_quals"][ob:o1].copy()
        return PackedReads(self.contig_names, out, self.max_span)

    def take_ranges(self, ranges):
        """Reads of several disjoint ascending index ranges [(lo, hi), ...] as one self-contained PackedReads (what a
        rank holding several whole contigs, or the tail of one contig and the head of the next, computes on).  Links
        that point outside the kept reads become -1."""
        ranges = [(int(lo), int(hi)) for lo, hi in ranges if hi > lo]
        if len(ranges) == 1:
            return self.slice_reads(*ranges[0])
        if not ranges:
            return self.slice_reads(0, 0)
        for (_, h0), (l1, _) in zip(ranges, ranges[1:]):
            if l1 < h0:
                raise ValueError("read ranges must be disjoint and ascending")
        parts = [self.slice_reads(lo, hi) for lo, hi in ranges]
        base = np.concatenate([[0], np.cumsum([p.n_reads for p in parts])]).astype(np.int64)
        los = np.asarray([lo for lo, _ in ranges], dtype=np.int64)
        his = np.asarray([hi for _, hi in ranges], dtype=np.int64)
        a, out = self.a, {}
        for k, _ in _HOT_FIELDS:
            if k != "link":
                out[k] = np.concatenate([p.a[k] for p in parts])
        # a link may reach from one kept range into an earlier one: re-base it there instead of dropping it
        link = np.concatenate([a["link"][lo:hi] for lo, hi in ranges]).astype(np.int64)
        r = np.searchsorted(los, link, side="right") - 1
        ok = (link >= 0) & (r >= 0)
        ok &= link < his[np.maximum(r, 0)]
        out["link"] = np.where(ok, link - los[np.maximum(r, 0)] + base[np.maximum(r, 0)], -1).astype(np.int32)
        out["contig_len"] = a["contig_len"].copy()
        # reads kept in front of a contig's start = sum over the parts of the reads each holds in front of it
        out["contig_read_start"] = np.sum([p.a["contig_read_start"] for p in parts], axis=0).astype(np.int64)
        out["cx_read"] = np.concatenate([p.a["cx_read"].astype(np.int64) + b for p, b in zip(parts, base[:-1])]).astype(np.uint32)
        nblk = np.concatenate([[0], np.cumsum([p.a["blk_start"].shape[0] for p in parts])]).astype(np.int64)
        out["cx_off"] = np.concatenate([p.a["cx_off"][:-1].astype(np.int64) + o for p, o in zip(parts, nblk[:-1])]
                                       + [nblk[-1:]]).astype(np.uint32)
        for k in ("blk_start", "blk_end", "blk_ins"):
            out[k] = np.concatenate([p.a[k] for p in parts])
        out["pool_read"] = np.concatenate([p.a["pool_read"].astype(np.int64) + b for p, b in zip(parts, base[:-1])]).astype(np.uint32)
        nb = [int(p.a["pool_off"][-1]) for p in parts]          # even: every pooled read starts on a byte boundary
        pbase = np.concatenate([[0], np.cumsum(nb)]).astype(np.uint64)
        out["pool_off"] = np.concatenate([p.a["pool_off"][:-1] + o for p, o in zip(parts, pbase[:-1])] + [pbase[-1:]]).astype(np.uint64)
        out["pool_bases"] = np.concatenate([p.a["pool_bases"][:n // 2] for p, n in zip(parts, nb)])
        out["pool_quals"] = np.concatenate([p.a["pool_quals"][:n] for p, n in zip(parts, nb)])
        return PackedReads(self.contig_names, out, self.max_span)

    def pool_sequence(self, read_idx):
        """(sequence str, qualities list) of a pooled read; host-side helper for tests."""
        from .records import BASE_CODES
        k = int(np.searchsorted(self.a["pool_read"], read_idx))
        if k >= len(self.a["pool_read"]) or self.a["pool_read"][k] != read_idx:
            return None
        o0 = int(self.a["pool_off"][k])
        o1 = o0 + int(self.a["qlen"][read_idx])
        idx = np.arange(o0, o1)
        nib = (self.a["pool_bases"][idx >> 1] >> ((~idx & 1) * 4)) & 15
        return "".join(BASE_CODES[c] for c in nib), [int(q) for q in self.a["pool_quals"][o0:o1]]

    def to_torch(self, device, clip_rec=True):
        """Every array on ``device``.  clip_rec: also the dense clip records (``clip_records``); MFK_NO_CLIP_REC=1
        leaves them out whatever the caller says (the library then gathers the columns: same results)."""
        import torch
        out = {}
        if clip_rec and os.environ.get("MFK_NO_CLIP_REC", "0") in ("", "0"):
            rec, tile = self.clip_records()
            out["n_clip"] = int(rec.shape[0])
            raw = rec.view(np.uint8) if rec.shape[0] else np.zeros(16, np.uint8)
            out["clip_rec"] = torch.from_numpy(np.ascontiguousarray(raw)).to(device)
            out["clip_tile"] = torch.from_numpy(tile.view(np.int32)).to(device)
        for k, _ in ALL_FIELDS:
            arr = np.ascontiguousarray(self.a[k])
            # torch has no uint16/32/64 arithmetic but can hold the bits; reinterpret as signed of the same width
            view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32,
                    np.dtype(np.uint64): np.int64}.get(arr.dtype)
            if view is not None:
                arr = arr.view(view)
            t = torch.from_numpy(arr)
            if t.numel() == 0:
                t = torch.zeros(1, dtype=t.dtype)  # keep a valid pointer
            out[k] = t.to(device)
        return out


def pack_reads(contigs, reads, pool="clipped", keep_names=False, exclude_flags=0):
    """Pack mapped reads (an iterable of ``AlignedRead``-like objects in BAM order).

    contigs: list of (name, length) from the BAM header.
    pool:    'clipped' (findflanks), 'all', 'none', or a callable(read, contig_id) -> bool (pairflanks windows).
    exclude_flags: records with any of these FLAG bits are skipped (0x900 = what the reference's formatbam removes).
    """
    contig_index = {n: i for i, (n, _) in enumerate(contigs)}
    if len(contigs) > 65535:
        raise PackError("more than 65535 contigs")
    cols = {k: [] for k, _ in _HOT_FIELDS}
    cx_read, cx_off, blk_start, blk_end, blk_ins = [], [0], [], [], []
    pool_read, pool_off, pool_codes, pool_quals = [], [0], [], []
    last_by_name, names = {}, []
    max_span, prev_key, n = 0, (-1, -1), 0
    for read in reads:
        ct = read.cigartuples
        if ct is None or len(ct) == 0 or (getattr(read, "flag", 0) & 4):
            continue  # unmapped: mapping quality 0 in the reference, never reaches the CIGAR (quirk Q17)
        if getattr(read, "flag", 0) & exclude_flags:
            continue
        cid = contig_index[read.reference_name]
        key = (cid, read.reference_start)
        if key < prev_key:
            raise PackError("alignments are not coordinate-sorted (the reference needs a sorted, indexed BAM)")
        prev_key = key
        md = read.tags.get("MD") if hasattr(read, "tags") else read.get_tag("MD")
        lkind, rkind = clip_kinds(ct, md)
        # strip clips to find the alignment core
        lo, hi = 0, len(ct)
        while lo < hi and ct[lo][0] in (CHARD_CLIP, CSOFT_CLIP):
            lo += 1
        while hi > lo and ct[hi - 1][0] in (CHARD_CLIP, CSOFT_CLIP):
            hi -= 1
        core = [(op, l) for op, l in ct[lo:hi] if op != CPAD]
        if not core:
            raise PackError("alignment without aligned bases: %r" % (read,))
        for op, _ in core:
            if op in (CHARD_CLIP, CSOFT_CLIP):
                raise PackError("clip operation inside an alignment: %r" % (read,))
        ref_ops = [op for op, _ in core if op != CINS]
        if not ref_ops or ref_ops[0] in (CDEL, CREF_SKIP) or ref_ops[-1] in (CDEL, CREF_SKIP):
            raise PackError("alignment starts or ends with a deletion/skip: unsupported (%r)" % (read,))
        simple = len(core) == 1 and not (lo > 0 and ct[0][0] == CHARD_CLIP and ct[lo - 1][0] == CSOFT_CLIP)
        seq = read.query_sequence or ""
        qlen = len(seq)
        if qlen > 65535:
            raise PackError("query longer than 65535 bases")
        rend = read.reference_end
        span = rend - read.reference_start
        if span > max_span:
            max_span = span
        flags = (lkind << FLAG_LKIND_SHIFT) | (rkind << FLAG_RKIND_SHIFT)
        if not simple:
            flags |= FLAG_COMPLEX
            blocks, ins = _blocks_with_insertions(read)
            cx_read.append(n)
            for (s, e), i in zip(blocks, ins):
                blk_start.append(s)
                blk_end.append(e)
                blk_ins.append(i)
            cx_off.append(len(blk_start))
        if callable(pool):
            want = pool(read, cid)
        else:
            want = pool == "all" or (pool == "clipped" and (lkind or rkind))
        if want and qlen:
            flags |= FLAG_POOL
            pool_read.append(n)
            plen = qlen + (qlen & 1)  # every pooled read starts on a byte boundary of the nibble stream
            codes = np.zeros(plen, dtype=np.uint8)
            codes[:qlen] = np.fromiter((BASE_TO_CODE.get(c, 15) for c in seq), dtype=np.uint8, count=qlen)
            pool_codes.append(codes)
            q = read.query_qualities
            quals = np.zeros(plen, dtype=np.uint8)
            if q is not None:
                quals[:qlen] = np.asarray(q, dtype=np.uint8)
            pool_quals.append(quals)
            pool_off.append(pool_off[-1] + plen)
        cols["ref_start"].append(read.reference_start)
        cols["ref_end"].append(rend)
        cols["link"].append(last_by_name.get(read.query_name, -1))
        last_by_name[read.query_name] = n
        cols["contig_id"].append(cid)
        cols["qlen"].append(qlen)
        cols["first_op_len"].append(min(ct[0][1], 65535))
        cols["last_op_len"].append(min(ct[-1][1], 65535))
        cols["mapq"].append(read.mapping_quality)
        cols["flags"].append(flags)
        if keep_names:
            names.append(read.query_name)
        n += 1
    arrays = {k: np.asarray(cols[k], dtype=t) for k, t in _HOT_FIELDS}
    arrays["contig_len"] = np.asarray([l for _, l in contigs], dtype=np.int32)
    arrays["contig_read_start"] = np.searchsorted(arrays["contig_id"], np.arange(len(contigs) + 1)).astype(np.int64)
    arrays["cx_read"] = np.asarray(cx_read, dtype=np.uint32)
    arrays["cx_off"] = np.asarray(cx_off, dtype=np.uint32)
    arrays["blk_start"] = np.asarray(blk_start, dtype=np.int32)
    arrays["blk_end"] = np.asarray(blk_end, dtype=np.int32)
    arrays["blk_ins"] = np.asarray(blk_ins, dtype=np.int32)
    arrays["pool_read"] = np.asarray(pool_read, dtype=np.uint32)
    arrays["pool_off"] = np.asarray(pool_off, dtype=np.uint64)
    codes = np.concatenate(pool_codes) if pool_codes else np.zeros(0, np.uint8)
    arrays["pool_bases"] = pack_nibbles(codes)
    arrays["pool_quals"] = np.concatenate(pool_quals) if pool_quals else np.zeros(0, np.uint8)
    return PackedReads([n_ for n_, _ in contigs], arrays, max_span, names if keep_names else None)


def pack_nibbles(codes):
    """4-bit base codes -> bytes in BAM order: base k in byte k>>1, even k in the HIGH nibble."""
    codes = np.asarray(codes, dtype=np.uint8)
    if codes.size & 1:
        codes = np.concatenate([codes, np.zeros(1, np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------------
# native ingestion: libmfkio.so (include/mfk_bamio.h) does BGZF inflate + the rules above in C++ threads
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as _C
import os as _os

IO_LIB_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "csrc", "libmfkio.so")
POOL_MODES = {"none": 0, "clipped": 1, "all": 2, "regions": 3}


class _IoRegion(_C.Structure):
    _fields_ = [("contig_id", _C.c_int32), ("lo", _C.c_int32), ("hi", _C.c_int32), ("es", _C.c_int32),
                ("ee", _C.c_int32)]


_vp = _C.c_void_p


class _IoArrays(_C.Structure):
    _fields_ = ([("n_reads", _C.c_int64), ("n_contigs", _C.c_int32), ("max_span", _C.c_int32), ("n_cx", _C.c_int64),
                 ("n_blk", _C.c_int64), ("n_pool", _C.c_int64), ("pool_bases_bytes", _C.c_int64),
                 ("pool_quals_bytes", _C.c_int64)] +
                [(k, _vp) for k in ("ref_start", "ref_end", "link", "contig_id", "qlen", "first_op_len", "last_op_len",
                                    "mapq", "flags", "contig_len", "contig_read_start", "cx_read", "cx_off",
                                    "blk_start", "blk_end", "blk_ins", "pool_read", "pool_off", "pool_bases",
                                    "pool_quals", "contig_names")] +
                [("contig_names_bytes", _C.c_int64), ("n_records", _C.c_int64), ("compressed_bytes", _C.c_int64),
                 ("uncompressed_bytes", _C.c_int64), ("n_threads", _C.c_int32), ("inflate_backend", _C.c_int32),
                 ("seconds_inflate", _C.c_double), ("seconds_parse", _C.c_double), ("seconds_link", _C.c_double),
                 ("seconds_total", _C.c_double), ("n_redone_passes", _C.c_int64)])


_io_lib = None


def io_lib():
    global _io_lib
    if _io_lib is None:
        if not _os.path.exists(IO_LIB_PATH):
            raise PackError("shared library %s is missing: build it first (python -c 'import __graft_entry__ as g; "
                            "g.build()')" % IO_LIB_PATH)
        so = _C.CDLL(IO_LIB_PATH)
        so.mfkio_pack_bam.restype = _C.c_int
        so.mfkio_pack_bam.argtypes = [_C.c_char_p, _C.c_int, _vp, _C.c_int64, _C.c_int, _C.c_int64, _C.POINTER(_vp)]
        so.mfkio_pack_bam_filtered.restype = _C.c_int
        so.mfkio_pack_bam_filtered.argtypes = [_C.c_char_p, _C.c_int, _vp, _C.c_int64, _C.c_int, _C.c_int64, _C.c_uint32, _C.POINTER(_vp)]
        so.mfkio_pack_bam_index.restype = _C.c_int
        so.mfkio_pack_bam_index.argtypes = [_C.c_char_p, _C.c_int, _C.c_int, _C.c_int64, _C.c_uint32, _C.c_char_p, _C.POINTER(_vp)]
        so.mfkio_pack_bam_regions_indexed.restype = _C.c_int
        so.mfkio_pack_bam_regions_indexed.argtypes = [_C.c_char_p, _C.c_char_p, _vp, _C.c_int64, _C.c_int, _C.c_uint32, _C.POINTER(_vp)]
        so.mfkio_pack_arrays.restype = _C.POINTER(_IoArrays)
        so.mfkio_pack_arrays.argtypes = [_vp]
        so.mfkio_pack_error.restype = _C.c_char_p
        so.mfkio_pack_error.argtypes = [_vp]
        so.mfkio_pack_free.restype = None
        so.mfkio_pack_free.argtypes = [_vp]
        so.mfkio_version.restype = _C.c_int
        _io_lib = so
    return _io_lib


class _IoOwner:
    """Keeps a mfkio_pack alive for as long as any array that views its memory."""

    def __init__(self, so, handle):
        self.so, self.handle = so, handle

    def __del__(self):
        if self.handle:
            self.so.mfkio_pack_free(self.handle)
            self.handle = None


def pack_bam(path, pool="clipped", regions=None, threads=0, chunk_bytes=0, exclude_flags=0, index_out=None, index=None):
    """Native counterpart of ``pack_reads(BamReader(path))``: same arrays, bit for bit.

    pool:    'clipped' | 'all' | 'none' | 'regions'.
    regions: for pool='regions' (pairflanks), rows (contig_id, lo, hi, es, ee) as ``candidate_regions`` returns them:
             only reads over [lo, hi) are packed, those over [es, ee) carry their bases.
    exclude_flags: skip records with any of these FLAG bits (0x900: secondary / supplementary, as formatbam does).
    index_out: (full passes) also write the record index of the file there (include/mfk_bamio.h); best effort.
    index:   (pool='regions') a record index written by an earlier full pass over the same file: only the BGZF blocks
             around the regions are inflated, as the reference's bam.fetch does through the BAI.  A missing or stale
             index (other size / mtime / FLAG filter) falls back to the full pass; ``.io_stats['indexed']`` says which ran.
    The returned PackedReads views the library's memory (no copy); ``.io_stats`` says what the run did.
    """
    so = io_lib()
    mode = POOL_MODES[pool]
    reg_arr, n_reg = None, 0
    if mode == 3:
        regs = np.ascontiguousarray(np.asarray(list(regions) if regions is not None and not isinstance(regions, np.ndarray) else
                                               (regions if regions is not None else []), dtype=np.int32).reshape(-1, 5))
        n_reg = int(regs.shape[0])
        reg_arr = regs.ctypes.data_as(_vp) if n_reg else (_IoRegion * 1)()     # rows (contig_id, lo, hi, es, ee) = mfkio_region
    handle = _vp()
    indexed = False
    rc = -1
    if mode == 3 and index is not None and _os.path.exists(index):
        rc = so.mfkio_pack_bam_regions_indexed(_os.fsencode(path), _os.fsencode(index), _C.cast(reg_arr, _vp), n_reg, int(threads),
                                               int(exclude_flags), _C.byref(handle))
        if rc == 5:                       # MFKIO_ERR_INDEX: not an error of the call, the full pass below serves
            if handle.value:
                so.mfkio_pack_free(handle)
            handle = _vp()
            rc = -1
        else:
            indexed = True
    if rc == -1:
        if index_out is not None and mode != 3:
            rc = so.mfkio_pack_bam_index(_os.fsencode(path), mode, int(threads), int(chunk_bytes), int(exclude_flags),
                                         _os.fsencode(index_out), _C.byref(handle))
        else:
            rc = so.mfkio_pack_bam_filtered(_os.fsencode(path), mode, _C.cast(reg_arr, _vp) if reg_arr is not None else None, n_reg,
                                            int(threads), int(chunk_bytes), int(exclude_flags), _C.byref(handle))
    owner = _IoOwner(so, handle.value)
    if rc != 0:
        msg = so.mfkio_pack_error(handle).decode("utf-8", "replace") if handle.value else "out of memory"
        if rc == 3:
            raise PackError(msg)
        if rc == 1:
            raise OSError(msg)
        raise ValueError(msg)
    a = so.mfkio_pack_arrays(handle).contents

    def view(ptr, n, dtype):
        dt = np.dtype(dtype)
        if n == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        buf = (_C.c_ubyte * (n * dt.itemsize)).from_address(ptr)
        buf._owner = owner
        return np.frombuffer(buf, dtype=dt)
    n = int(a.n_reads)
    sizes = {"contig_len": a.n_contigs, "contig_read_start": a.n_contigs + 1, "cx_read": a.n_cx, "cx_off": a.n_cx + 1,
             "blk_start": a.n_blk, "blk_end": a.n_blk, "blk_ins": a.n_blk, "pool_read": a.n_pool,
             "pool_off": a.n_pool + 1, "pool_bases": a.pool_bases_bytes, "pool_quals": a.pool_quals_bytes}
    arrays = {}
    for k, t in ALL_FIELDS:
        arrays[k] = view(getattr(a, k), int(sizes.get(k, n)), t)
    names_blob = _C.string_at(a.contig_names, a.contig_names_bytes) if a.contig_names_bytes else b""
    contig_names = [s.decode() for s in names_blob.split(b"\0")[:a.n_contigs]]
    p = PackedReads(contig_names, arrays, int(a.max_span))
    p.io_stats = {k: getattr(a, k) for k in ("n_records", "compressed_bytes", "uncompressed_bytes", "n_threads",
                                             "inflate_backend", "seconds_inflate", "seconds_parse", "seconds_link",
                                             "seconds_total", "n_redone_passes")}
    p.io_stats["indexed"] = indexed
    p._owner = owner
    return p
