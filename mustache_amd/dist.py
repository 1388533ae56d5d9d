"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The path shards without any data-path collective: samples are independent (C4), contigs are
independent (C3), and coordinate ranges of one contig are independent given a halo (C5).  The only exchange is the
merge of the per-rank flank tables -- one all-gather of a few KB..MB of fixed-width records, latency-bound on xGMI.

* ``shard_samples``    round-robin assignment of isolates to ranks
* ``shard_contigs``    greedy bin-packing of whole contigs by read count
* ``coordinate_shards``/``slice_for_range``  read-balanced coordinate ranges of one contig with the halo that the
  three chained min-distance filters need (3 x mindist + 3, not just one pairing window)
* ``allgather_flanks`` the merge: every rank ends up with the concatenated table in rank order
"""
from __future__ import annotations

import numpy as np

from . import lib as _lib


def halo(min_distance_to_mate=22, max_direct_repeat_length=20):
    """Positions a shard must compute beyond the range it owns.  filter_softclips_mindistance runs three times
    (reference findflanks.py:40, 43, 49) and each round's verdict on a site depends on its neighbours' survival in the
    previous round, so the reach is 3 x mindist; pairing looks back maxdr + 1."""
    return max(3 * min_distance_to_mate + 3, max_direct_repeat_length + 2)


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
    r0, r1 = int(packed.contig_read_start[contig]), int(packed.contig_read_start[contig + 1])
    span = packed.max_span if max_span is None else max_span
    starts = packed.ref_start[r0:r1]
    lo = r0 + int(np.searchsorted(starts, own[0] - h - 1 - span, side="left"))
    hi = r0 + int(np.searchsorted(starts, own[1] + h + 2, side="left"))
    return packed.slice_reads(lo, max(lo, hi))


def owned(rows, own, contig=None):
    """Filter flank / site rows to the positions a shard owns."""
    m = (rows["pos"] >= own[0]) & (rows["pos"] < own[1])
    if contig is not None:
        m &= rows["contig_id"] == contig
    return rows[m]


GATHER_CAP = 64 * 1024     # bytes per rank that travel in the first (normally only) collective


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


def allgather_flanks(flanks, seqs, device=None, cap=GATHER_CAP):
    """Merge per-rank flank tables (FLANK_DTYPE rows + their sequence bytes).  Returns (rows, seqs) with seq_off
    rebased, ranks concatenated in rank order.  Works without torch.distributed initialised (single process).

    One collective in the common case: every rank sends a 16-byte header (row count, sequence bytes) followed by its
    rows and sequences, padded to `cap` bytes.  Only if some rank's table is larger than `cap` does a second all-gather
    with the true maximum follow (all ranks see the same headers, so they agree on that without talking)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flanks, seqs
    ws = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    rec = _lib.FLANK_DTYPE.itemsize
    body = np.ascontiguousarray(flanks).tobytes() + bytes(seqs)
    head = np.array([flanks.shape[0], len(seqs)], dtype=np.int64).tobytes()

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
    rows, blobs, off = [], [], 0
    for r in range(ws):
        nr, nb = int(counts[r, 0]), int(counts[r, 1])
        g = got[r, 16:]
        part = np.frombuffer(g[:nr * rec].tobytes(), dtype=_lib.FLANK_DTYPE).copy()
        part["seq_off"] += off
        rows.append(part)
        blobs.append(g[nr * rec:nr * rec + nb].tobytes())
        off += nb
    return np.concatenate(rows) if rows else flanks, b"".join(blobs)
