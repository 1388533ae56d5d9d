"""BAM file -> packed reads, decoded on the GPU (include/mfk.h mfk_decode_bam / mfk_decode_bam_fetch, mustache_amd/csrc/
mfk_decode.hip): the file's bytes go to the device as they are, BGZF blocks are inflated one wave per block, records are found,
parsed and packed by kernels, and the columns are written straight into device tensors.  Same rules and same arrays as the
host packers (``pack.pack_bam`` = libmfkio.so, ``pack.pack_reads`` = the Python statement; tests/test_gpu_decode.py compares
every array), which restate what the reference gets from pysam while it iterates the BAM (findflanks.py:100-121) and
evaluates sctools.py:5-31, 99-109 per read.

Only what needs no device stays on the host: the BGZF block table (18-byte headers) and the BAM header (contig names).
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import threading
import time
import warnings
import zlib

import numpy as np

from . import lib as _lib
from .lib import BGZF_BLOCK_DTYPE, MfkError, MfkReads
from .pack import PackedReads, PackError

POOL_MODES = {"none": 0, "clipped": 1, "all": 2}
INDEX_GRANULE = 32 * 1024          # mfk_bamio.cpp IDX_GRANULE
# why a record is rejected (mfk_decode.hip PE_*), worded like pack.pack_reads / mfk_bamio.cpp
_REJECT = {1: "negative sequence length in an alignment record", 2: "truncated alignment record",
           3: "alignment on a contig that is not in the header", 4: "alignment end beyond 2^31",
           5: "alignments are not coordinate-sorted (the reference needs a sorted, indexed BAM)",
           6: "malformed tags in an alignment record",
           7: "read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:28)",
           8: "read without MD tag: the reference's lenient soft-clip test needs it (sctools.py:20)",
           9: "clip operation inside an alignment", 10: "alignment without aligned bases",
           11: "alignment starts or ends with a deletion/skip: unsupported", 12: "query longer than 65535 bases"}

_COLS = (("ref_start", "int32"), ("ref_end", "int32"), ("link", "int32"), ("contig_id", "int16"), ("qlen", "int16"),
         ("first_op_len", "int16"), ("last_op_len", "int16"), ("mapq", "uint8"), ("flags", "uint8"))
_NP = {"ref_start": np.int32, "ref_end": np.int32, "link": np.int32, "contig_id": np.uint16, "qlen": np.uint16, "first_op_len": np.uint16,
       "last_op_len": np.uint16, "mapq": np.uint8, "flags": np.uint8, "cx_read": np.uint32, "cx_off": np.uint32, "blk_start": np.int32,
       "blk_end": np.int32, "blk_ins": np.int32, "pool_read": np.uint32, "pool_off": np.uint64, "pool_bases": np.uint8, "pool_quals": np.uint8}


class GpuDecodeUnverified(MfkError):
    """Decode this file with the host reader: a guessed record boundary did not verify on the device, or the file (its
    compressed bytes, its inflated stream and the per-record work arrays) does not fit the device's free memory."""


class DevicePacked:
    """Packed reads whose arrays are device tensors (what ``PackedReads.to_torch`` would give), plus the header on the host."""

    def __init__(self, contig_names, contig_len, contig_read_start, max_span, tensors, counts, io_stats):
        self.contig_names = list(contig_names)
        self.contig_len = np.asarray(contig_len, dtype=np.int32)
        self.contig_read_start = np.asarray(contig_read_start, dtype=np.int64)
        self.max_span = int(max_span)
        self.t = tensors
        self.counts = dict(counts)          # n_reads, n_cx, n_blk, n_pool, pool_bytes
        self.io_stats = io_stats

    @property
    def n_reads(self):
        return int(self.counts["n_reads"])

    def nbytes_hot(self):
        return self.n_reads * 22

    def device_struct(self, binding):
        hdr = {"contig_len": self.contig_len, "contig_read_start": self.contig_read_start, "n_cx": self.counts["n_cx"],
               "n_blk": self.counts["n_blk"], "n_pool": self.counts["n_pool"]}
        return binding.reads_struct(self.t, self.n_reads, self.max_span, hdr)

    def to_host(self):
        """A PackedReads with the same arrays in host memory (tests, slicing)."""
        c = self.counts
        n = {"cx_read": c["n_cx"], "cx_off": c["n_cx"] + 1, "blk_start": c["n_blk"], "blk_end": c["n_blk"], "blk_ins": c["n_blk"],
             "pool_read": c["n_pool"], "pool_off": c["n_pool"] + 1, "pool_bases": c["pool_bytes"] // 2, "pool_quals": c["pool_bytes"]}
        a = {}
        for k, dt in _NP.items():
            cnt = n.get(k, c["n_reads"])
            a[k] = self.t[k].cpu().numpy().view(dt)[:cnt].copy()
        a["contig_len"], a["contig_read_start"] = self.contig_len.copy(), self.contig_read_start.copy()
        return PackedReads(self.contig_names, a, self.max_span)


def _header(mm, blocks):
    """(contig names, contig lengths, offset of the first alignment record in the inflated stream) from the first blocks."""
    buf, k = b"", 0

    def need(nbytes):
        nonlocal buf, k
        while len(buf) < nbytes:
            if k >= blocks.shape[0]:
                raise MfkError("not a BAM file (truncated header)")
            b = blocks[k]
            o, n = int(b["comp_off"]), int(b["comp_len"])
            try:
                buf += zlib.decompress(bytes(mm[o:o + n]), -15)
            except zlib.error:
                raise MfkError("BGZF block does not inflate (corrupt file)")
            k += 1
    need(12)
    if buf[:4] != b"BAM\1":
        raise MfkError("not a BAM file")
    l_text, = struct.unpack_from("<i", buf, 4)
    need(12 + l_text)
    n_ref, = struct.unpack_from("<i", buf, 8 + l_text)
    o = 12 + l_text
    names, lens = [], []
    for _ in range(n_ref):
        need(o + 4)
        l_name, = struct.unpack_from("<i", buf, o)
        need(o + 8 + l_name)
        names.append(buf[o + 4:o + 4 + l_name - 1].decode())
        lens.append(struct.unpack_from("<i", buf, o + 4 + l_name)[0])
        o += 8 + l_name
    return names, lens, o


def _write_index(path, bam_path, total_u, first_off, max_span, exclude_flags, marks, mark_rp, file_off, blocks):
    """``<table>.mfki`` in the layout of mfk_bamio.cpp write_index (MFKI0002); best effort like there."""
    try:
        st = os.stat(bam_path)
        has = marks != np.uint64(0xFFFFFFFFFFFFFFFF)
        m = np.zeros(int(has.sum()), dtype=np.dtype([("u", "<u8"), ("ref", "<i4"), ("pos", "<i4")]))
        m["u"], m["ref"], m["pos"] = marks[has], mark_rp[has, 0], mark_rp[has, 1]
        blk = np.zeros(blocks.shape[0], dtype=np.dtype([("at", "<u8"), ("u", "<u8")]))
        blk["at"], blk["u"] = file_off, blocks["out_off"]
        tmp = path + ".tmp"
        with open(tmp, "wb") as fh:
            fh.write(b"MFKI0002" + struct.pack("<QqQQiIQQ", st.st_size, st.st_mtime_ns, total_u, first_off, int(max_span), int(exclude_flags),
                                               m.shape[0], blk.shape[0]))
            fh.write(m.tobytes())
            fh.write(blk.tobytes())
        os.replace(tmp, path)
    except OSError:
        pass


def decode_bam(binding, path, pool="clipped", exclude_flags=0, device="cuda:0", index_out=None, check_crc=True):
    """BAM file -> DevicePacked.  pool: 'none' | 'clipped' | 'all'.  index_out: also write the record index of the file there
    (``pack.pack_bam``'s ``index_out``).  Raises PackError for what the packers reject, GpuDecodeUnverified when the caller
    should use the host reader, MfkError for damaged files (a block that does not inflate, or whose bytes do not have the
    CRC-32 of its trailer: ``check_crc``, as the host reader checks it)."""
    import torch
    if pool not in POOL_MODES:
        raise ValueError("pool must be one of %s for the device decode" % (sorted(POOL_MODES),))
    t0 = time.perf_counter()
    size = os.path.getsize(path)
    if size == 0:
        raise MfkError("%s is not a BAM file (empty)" % path)
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    nb, total = C.c_int64(), C.c_int64()

    def table(cap):
        blocks = np.zeros(cap, dtype=BGZF_BLOCK_DTYPE)
        file_off = np.zeros(cap, dtype=np.uint64)
        crcs = np.zeros(max(cap, 1), dtype=np.uint32)
        binding._check(binding.bgzf_block_table(binding.ctx, mm.ctypes.data, size, blocks.ctypes.data, file_off.ctypes.data, crcs.ctypes.data, cap, C.byref(nb), C.byref(total)))
        return blocks, file_off, crcs
    if size < 28 or bytes(mm[:4]) != b"\x1f\x8b\x08\x04":
        table(0)                         # raises: not a BGZF file (before anything is copied anywhere)
    # The file's bytes go up as they are, on a thread of their own; the walk over the block headers and the header of the BAM,
    # which need nothing from the device, happen meanwhile.
    box = {}

    def upload():
        try:
            box["comp"] = torch.from_numpy(np.asarray(mm)).to(device)
            torch.cuda.synchronize(device)
        except BaseException as e:       # (handed to the calling thread below)
            box["err"] = e
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (the mapping is read-only; the tensor is only read)
        th = threading.Thread(target=upload)
        th.start()
        try:
            cap = size // 8192 + 1024    # one walk when the blocks average 8 KiB or more (they do), else a second
            blocks, file_off, crcs = table(cap)
            if nb.value > cap:
                blocks, file_off, crcs = table(nb.value)
            blocks, file_off, crcs = blocks[:nb.value], file_off[:nb.value], crcs[:max(nb.value, 1)]
            if total.value >= 1 << 34:   # (32-bit block offsets in the packed layout: the host reader takes such a file in passes)
                raise GpuDecodeUnverified("%d inflated bytes: more than the device decode takes in one piece" % total.value)
            names, lens, first_off = _header(mm, blocks)
            if not names:
                raise MfkError("%s has no reference sequences in its header" % path)
            if len(names) > 65535:
                raise PackError("more than 65535 contigs")
            t1 = time.perf_counter()
        finally:
            th.join()
    if isinstance(box.get("err"), torch.cuda.OutOfMemoryError):
        raise GpuDecodeUnverified("no room on the device for the file (%d bytes)" % size)
    if "err" in box:
        raise box["err"]
    comp = box.pop("comp")
    t2 = time.perf_counter()
    sizes = np.zeros(8, dtype=np.int64)
    rc = binding.decode_bam(binding.ctx, comp.data_ptr(), size, blocks.ctypes.data, crcs.ctypes.data if check_crc else None, blocks.shape[0], first_off,
                            len(names), POOL_MODES[pool], int(exclude_flags), sizes.ctypes.data)
    msg = binding.last_error(binding.ctx).decode() if rc else ""
    if rc == 13 or (rc == 1 and "out of device memory" in msg):     # (a file too large to decode in one piece: the host reader streams it)
        raise GpuDecodeUnverified(msg)
    if rc == 14:
        raise PackError("%s (alignment record %d of the file)" % (_REJECT.get(int(sizes[1]), "rejected record"), int(sizes[0])))
    binding._check(rc)
    del comp
    n, n_cx, n_blk, n_pool, pool_bytes = (int(x) for x in sizes[:5])
    t3 = time.perf_counter()

    def dev(count, dtype):
        return torch.empty(max(int(count), 1) + 16, dtype=getattr(torch, dtype), device=device)
    t = {k: dev(n, dt) for k, dt in _COLS}
    t.update({"cx_read": dev(n_cx, "int32"), "cx_off": dev(n_cx + 1, "int32"), "blk_start": dev(n_blk, "int32"), "blk_end": dev(n_blk, "int32"),
              "blk_ins": dev(n_blk, "int32"), "pool_read": dev(n_pool, "int32"), "pool_off": dev(n_pool + 1, "int64"),
              "pool_bases": dev(pool_bytes // 2, "uint8"), "pool_quals": dev(pool_bytes, "uint8")})
    dst = MfkReads()
    for k, v in t.items():
        setattr(dst, k, v.data_ptr())
    crs = np.zeros(len(names) + 1, dtype=np.int64)
    span = C.c_int32()
    n_marks = (int(total.value) + INDEX_GRANULE - 1) // INDEX_GRANULE if index_out else 0
    marks = np.zeros(max(n_marks, 1), dtype=np.uint64)
    mark_rp = np.zeros((max(n_marks, 1), 2), dtype=np.int32)
    rc = binding.decode_bam_fetch(binding.ctx, C.byref(dst), crs.ctypes.data, C.byref(span), INDEX_GRANULE if index_out else 0,
                                  marks.ctypes.data if index_out else None, mark_rp.ctypes.data if index_out else None)
    if rc == 14:
        raise PackError(_REJECT[5])
    binding._check(rc)
    t4 = time.perf_counter()
    if index_out:
        _write_index(index_out, path, int(total.value), first_off, span.value, exclude_flags, marks[:n_marks], mark_rp[:n_marks], file_off, blocks)
    stats = {"n_records": int(sizes[6]), "compressed_bytes": size, "uncompressed_bytes": int(total.value), "n_threads": 0, "inflate_backend": 2,
             "seconds_total": time.perf_counter() - t0, "seconds_inflate": 0.0, "seconds_parse": 0.0, "seconds_link": 0.0, "n_redone_passes": 0,
             "indexed": False, "decoder": "gpu", "seconds_block_table_and_header": t1 - t0, "seconds_upload_file": t2 - t1,
             "seconds_device_stage1": t3 - t2, "seconds_device_stage2": t4 - t3}
    return DevicePacked(names, lens, crs, span.value, t, {"n_reads": n, "n_cx": n_cx, "n_blk": n_blk, "n_pool": n_pool, "pool_bytes": pool_bytes}, stats)
