"""BGZF on the device: block table of a BGZF/BAM file on the host (18-byte headers, SAM specification 4.1), the deflate
streams inflated by ``mfk_bgzf_inflate`` (include/mfk.h; one wave per block, mustache_amd/csrc/mfk_inflate.hip).

The reference inflates inside pysam/htslib while it iterates the BAM (findflanks.py:22, 100-121).  This module is the first
stage of a decode that stays on the device; the stage drivers still read BAM files through the host
reader (libmfkio.so), whose record parser needs the inflated bytes in host memory.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

from .lib import BGZF_BLOCK_DTYPE, MfkError


def bgzf_blocks(data):
    """Block table of BGZF bytes: (blocks as BGZF_BLOCK_DTYPE with out_off = running sum of ISIZE, crc32 per block, total
    uncompressed bytes).  Empty blocks (the EOF marker) are left out."""
    mv = memoryview(data)
    n, o, u = len(mv), 0, 0
    rows, crcs = [], []
    while o < n:
        if o + 18 > n:
            raise MfkError("truncated BGZF block header")
        h = bytes(mv[o:o + 12])
        if h[0] != 31 or h[1] != 139 or h[2] != 8 or not (h[3] & 4):
            raise MfkError("not a BGZF file")
        xlen, = struct.unpack_from("<H", h, 10)
        if o + 12 + xlen > n:
            raise MfkError("truncated BGZF block header")
        extra, x, bsize = bytes(mv[o + 12:o + 12 + xlen]), 0, 0
        while x + 4 <= xlen:
            slen, = struct.unpack_from("<H", extra, x + 2)
            if extra[x] == 66 and extra[x + 1] == 67 and slen == 2 and x + 6 <= xlen:
                bsize = struct.unpack_from("<H", extra, x + 4)[0] + 1
            x += 4 + slen
        if not bsize or bsize < 12 + xlen + 8 or o + bsize > n:
            raise MfkError("bad BGZF block size")
        crc, isize = struct.unpack_from("<II", bytes(mv[o + bsize - 8:o + bsize]))
        if isize > 65536:
            raise MfkError("BGZF block larger than 64 KiB")
        if isize:
            rows.append((o + 12 + xlen, u, bsize - 12 - xlen - 8, isize))
            crcs.append(crc)
            u += isize
        o += bsize
    return np.array(rows, dtype=BGZF_BLOCK_DTYPE), np.array(crcs, dtype=np.uint32), u


def inflate_bytes(binding, data, device="cuda:0", check_crc=False):
    """BGZF bytes -> (uint8 device tensor with the inflated stream, seconds of the device call)."""
    import time

    import torch
    blocks, crcs, total = bgzf_blocks(data)
    comp = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device) if len(data) else torch.zeros(1, dtype=torch.uint8, device=device)
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    binding.run_bgzf_inflate(comp, blocks, out)
    dt = time.perf_counter() - t0
    if check_crc:
        host = out[:total].cpu().numpy()
        for b, c in zip(blocks, crcs):
            if zlib.crc32(host[int(b["out_off"]):int(b["out_off"]) + int(b["out_len"])].tobytes()) != int(c):
                raise MfkError("BGZF checksum mismatch")
    return out[:total], dt
