"""Synthetic simulated-insertion alignments (BASELINE.json configs).

Two generators:

* ``simulate_readset`` -- per-read Python simulator that emits ``AlignedRead`` objects with the CIGAR / MD / mapq a
  local aligner would give.  Small inputs only; it feeds the reference harness (golden fixtures), the packer tests and
  the BAM writer.
* ``synth_packed`` -- vectorised numpy generator that emits the packed SoA directly at BASELINE.json sizes (millions of
  reads) for ``bench.py`` and the full-size property tests.

Model: a random reference genome; IS elements with imperfect terminal inverted repeats inserted with a target-site
duplication (TSD) of length t at reference position p, so that the sample reads ``R[:p+t] + IS + R[p:]``.  Reads from
the left flank are right-clipped at ``pos_5p = p+t``; reads entering the right flank are left-clipped with
``pos_3p = p-1``; the direct repeat is ``R[p:p+t]``.
"""
from __future__ import annotations

import numpy as np

from .records import (AlignedRead, ReadSet, make_md, CMATCH, CINS, CDEL, CSOFT_CLIP)

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def make_is_library(rng, n=4, min_len=300, max_len=700):
    """IS elements: left arm IR, random body, right arm = reverse complement of the IR with a few changes."""
    lib = []
    for _ in range(n):
        ir_len = int(rng.integers(15, 36))
        ir = _rand_seq(rng, ir_len)
        body = _rand_seq(rng, int(rng.integers(min_len, max_len)))
        right = list(revcomp(ir))
        for _ in range(int(rng.integers(0, 3))):  # imperfect repeat
            k = int(rng.integers(3, ir_len - 3))
            right[k] = "ACGT"[("ACGT".index(right[k]) + 1) % 4]
        lib.append(ir + body + "".join(right))
    return lib


class _Seg:
    __slots__ = ("kind", "s0", "s1", "r0", "seq")

    def __init__(self, kind, s0, s1, r0, seq=None):
        self.kind, self.s0, self.s1, self.r0, self.seq = kind, s0, s1, r0, seq


def _build_sample(ref, events):
    """events: sorted list of (p, kind, arg).  kind: 'is' (arg=(tsd, seq)), 'sins' (arg=seq), 'del' (arg=length).
    Returns (sample sequence, segments).  Segment kinds: 'ref', 'is' (long insertion: clip boundary), 'sins'."""
    segs, out, cur_r, cur_s = [], [], 0, 0

    def add_ref(r0, r1):
        nonlocal cur_s
        if r1 > r0:
            segs.append(_Seg("ref", cur_s, cur_s + (r1 - r0), r0))
            out.append(ref[r0:r1])
            cur_s += r1 - r0

    for p, kind, arg in events:
        if kind == "is":
            tsd, seq = arg
            add_ref(cur_r, p + tsd)
            segs.append(_Seg("is", cur_s, cur_s + len(seq), -1, seq))
            out.append(seq)
            cur_s += len(seq)
            cur_r = p
        elif kind == "sins":
            add_ref(cur_r, p)
            segs.append(_Seg("sins", cur_s, cur_s + len(arg), -1, arg))
            out.append(arg)
            cur_s += len(arg)
            cur_r = p
        elif kind == "del":
            add_ref(cur_r, p)
            cur_r = p + arg
    add_ref(cur_r, len(ref))
    return "".join(out), segs


def _align_read(ref, sample, segs, a, b, min_anchor=8, tiny_overhang=2):
    """Turn sample interval [a,b) into (ref_start, cigartuples) or None (unaligned)."""
    pieces = []  # (kind, length, r0)
    for sg in segs:
        lo, hi = max(a, sg.s0), min(b, sg.s1)
        if hi > lo:
            pieces.append([sg.kind, hi - lo, sg.r0 + (lo - sg.s0) if sg.kind == "ref" else -1])
    # split at long insertions; keep the run with most reference bases
    runs, cur = [], []
    for i, pc in enumerate(pieces):
        if pc[0] == "is":
            runs.append(cur)
            cur = []
        else:
            cur.append(i)
    runs.append(cur)
    best = max(runs, key=lambda r: sum(pieces[i][1] for i in r if pieces[i][0] == "ref"))
    if not best or sum(pieces[i][1] for i in best if pieces[i][0] == "ref") < 20:
        return None
    lo_i, hi_i = best[0], best[-1]
    # trim short anchors / dangling short insertions at the ends of the run
    while True:
        changed = False
        if pieces[lo_i][0] != "ref" or (lo_i < hi_i and pieces[lo_i][1] < min_anchor):
            lo_i += 1
            changed = True
        if pieces[hi_i][0] != "ref" or (lo_i < hi_i and pieces[hi_i][1] < min_anchor):
            hi_i -= 1
            changed = True
        if lo_i > hi_i:
            return None
        if not changed:
            break
    left_clip = sum(pc[1] for pc in pieces[:lo_i])
    right_clip = sum(pc[1] for pc in pieces[hi_i + 1:])
    ref_start = pieces[lo_i][2]
    cig = []
    # tiny overhangs are aligned through (end mismatches) instead of clipped, as a local aligner does
    if 0 < left_clip <= tiny_overhang and ref_start - left_clip >= 0:
        ref_start -= left_clip
        cig.append([CMATCH, left_clip])
        left_clip = 0
    elif left_clip:
        cig.append([CSOFT_CLIP, left_clip])
    prev_end = None
    for i in range(lo_i, hi_i + 1):
        kind, ln, r0 = pieces[i]
        if kind == "ref":
            if prev_end is not None and r0 > prev_end:
                cig.append([CDEL, r0 - prev_end])
            if cig and cig[-1][0] == CMATCH:
                cig[-1][1] += ln
            else:
                cig.append([CMATCH, ln])
            prev_end = r0 + ln
        else:
            cig.append([CINS, ln])
    if 0 < right_clip <= tiny_overhang and prev_end + right_clip <= len(ref):
        cig[-1][1] += right_clip
        right_clip = 0
    elif right_clip:
        cig.append([CSOFT_CLIP, right_clip])
    return ref_start, [(o, l) for o, l in cig]


def simulate_readset(seed=1, contig_lens=(20000,), n_insertions=6, coverage=40, read_len=100, sub_rate=0.004,
                     frag_mean=300, frag_sd=30, n_small_ins=2, n_large_ins=1, n_dels=2, low_mapq_frac=0.03,
                     tsd_choices=(2, 3, 4, 5, 8, 9), is_library=None, wildtype_frac=0.0):
    """Returns (genome dict name->sequence, ReadSet, truth list)."""
    rng = np.random.default_rng(seed)
    lib = is_library or make_is_library(rng)
    genome, reads, truth = {}, [], []
    frag_id = 0
    for ci, clen in enumerate(contig_lens):
        name = "contig%d" % (ci + 1)
        ref = _rand_seq(rng, clen)
        genome[name] = ref
        n_ev = n_insertions + n_small_ins + n_large_ins + n_dels
        # well separated event positions
        slots = np.linspace(400, clen - 400, n_ev + 2)[1:-1].astype(int) if clen > 1500 else []
        kinds = (["is"] * n_insertions + ["sins"] * n_small_ins + ["lins"] * n_large_ins + ["del"] * n_dels)
        rng.shuffle(kinds)
        events = []
        for p, k in zip(slots, kinds):
            p = int(p + rng.integers(-50, 50))
            if k == "is":
                tsd = int(rng.choice(tsd_choices))
                seq = lib[int(rng.integers(len(lib)))]
                if rng.random() < 0.5:
                    seq = revcomp(seq)
                events.append((p, "is", (tsd, seq)))
                truth.append({"contig": name, "pos_5p": p + tsd, "pos_3p": p - 1, "tsd": tsd})
            elif k == "sins":
                events.append((p, "sins", _rand_seq(rng, int(rng.integers(3, 12)))))
            elif k == "lins":
                events.append((p, "sins", _rand_seq(rng, int(rng.integers(32, 45)))))
            else:
                events.append((p, "del", int(rng.integers(2, 15))))
        sample_mut, segs_mut = _build_sample(ref, events)
        # wild-type fragments come from the unmodified reference: they span insertion sites (pairflanks spanning_count)
        sample_wt, segs_wt = _build_sample(ref, [])
        n_frag = int(len(sample_mut) * coverage / (2 * read_len))
        for _ in range(n_frag):
            if rng.random() < wildtype_frac:
                sample, segs = sample_wt, segs_wt
            else:
                sample, segs = sample_mut, segs_mut
            flen = max(read_len, int(rng.normal(frag_mean, frag_sd)))
            fs = int(rng.integers(0, max(1, len(sample) - flen)))
            qname = "frag%07d" % frag_id
            frag_id += 1
            mapq = 60 if rng.random() >= low_mapq_frac else int(rng.integers(0, 20))
            for mate, a in enumerate((fs, fs + flen - read_len)):
                b = a + read_len
                if a < 0 or b > len(sample):
                    continue
                aln = _align_read(ref, sample, segs, a, b)
                if aln is None:
                    continue
                ref_start, cig = aln
                q = list(sample[a:b])
                errs = np.nonzero(rng.random(read_len) < sub_rate)[0]
                for e in errs:
                    q[e] = "ACGT"[("ACGT".index(q[e]) + 1 + int(rng.integers(0, 3))) % 4]
                q = "".join(q)
                quals = rng.integers(2, 41, read_len).tolist()
                rlen = sum(l for o, l in cig if o in (CMATCH, CDEL))
                md = make_md(ref[ref_start:ref_start + rlen], q, cig)
                reads.append(AlignedRead(qname, name, ref_start, mapq, cig, q, quals, md=md,
                                         flag=(0x41 if mate == 0 else 0x81)))
    contigs = [(n, len(s)) for n, s in genome.items()]
    return genome, ReadSet(contigs, reads), truth
