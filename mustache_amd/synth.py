"""Vectorised synthetic workloads at BASELINE.json sizes: emits the packed SoA directly.

``synth_packed`` builds, with numpy only, what a local aligner would report for 2x150 bp paired reads drawn from a
sample genome = random reference + IS-element insertions with target-site duplications:

* reads inside reference sequence: ``150M`` (a few percent carry a small insertion / deletion -> block table);
* reads crossing an insertion junction: soft-clipped (``mM cS`` / ``cS mM``); overhangs of 1-2 bases are aligned through
  and show up as terminal MD mismatches (the reference's "lenient" clips);
* sequencing errors on the first / last aligned base give the background of MD-mismatch "clips" every real BAM has;
* a fraction of fragments is wild type (no insertion): these span the insertion sites (pairflanks spanning_count);
* mates share a query name (``link``); a few percent of reads have mapq < 20.

Sequences and qualities are pooled for clipped reads and for every read near an insertion site (direct-repeat windows).
The result is deterministic in ``seed`` (numpy PCG64).
"""
from __future__ import annotations

import numpy as np

from .pack import (ALL_FIELDS, FLAG_COMPLEX, FLAG_POOL, PackedReads, pack_nibbles)

CONFIGS = {
    # name: genome length(s), coverage, insertions  (BASELINE.json configs)
    "tiny": dict(contig_lens=(200_000,), coverage=60, n_insertions=6),
    "C1_ecoli_100x": dict(contig_lens=(4_600_000,), coverage=100, n_insertions=20),
    "C2_saureus_1000x": dict(contig_lens=(2_800_000,), coverage=1000, n_insertions=20),
    "C3_klebsiella": dict(contig_lens=(5_300_000, 200_000, 120_000, 60_000, 9_000, 3_000), coverage=100, n_insertions=30),
    "C5_tenth": dict(contig_lens=(5_000_000,), coverage=200, n_insertions=1_000),   # C5 at a tenth: quick to generate
    "C5_meta_50mb": dict(contig_lens=(50_000_000,), coverage=200, n_insertions=10_000),
}

_CODE = np.array([1, 2, 4, 8], dtype=np.uint8)  # A C G T as BAM 4-bit codes


def _is_library(rng, n=8):
    lib = []
    for _ in range(n):
        ir_len = int(rng.integers(15, 41))
        ir = rng.integers(0, 4, ir_len)
        body = rng.integers(0, 4, int(rng.integers(700, 2500)))
        right = (3 - ir[::-1]).copy()  # reverse complement in 0..3 coding (A<->T, C<->G)
        for _ in range(int(rng.integers(0, 3))):
            k = int(rng.integers(3, ir_len - 3))
            right[k] = (right[k] + 1) % 4
        lib.append(np.concatenate([ir, body, right]).astype(np.uint8))
    return lib


def synth_packed(contig_lens=(200_000,), coverage=60, n_insertions=6, seed=1, read_len=150, frag_mean=400, frag_sd=50,
                 end_error=(0.002, 0.01), clip_error=0.004, indel_frac=0.02, low_mapq_frac=0.03, wildtype_frac=0.05,
                 tsd_choices=(2, 3, 4, 5, 8, 9), pool="near"):
    """Returns (PackedReads, genome dict name -> str, truth list of (contig, pos_5p, pos_3p)).

    pool: which reads carry their sequence + qualities.  "all": every read, which is what a ONE-pass run (findflanks and
    pairflanks on the same resident arrays, ``mfk_flankpairs``) needs when it cannot know beforehand where the
    direct-repeat windows of pairflanks will be; it uses no knowledge of the insertions (bench.py).  "near": the
    lenient-clipped reads plus the reads within 60 bp of a simulated insertion -- the same bytes reach the kernels, at
    a fraction of the generation time and memory, but the choice uses the ground truth (parity tests only)."""
    if pool not in ("near", "all"):
        raise ValueError("pool must be 'near' or 'all'")
    rng = np.random.Generator(np.random.PCG64(seed))
    lib = _is_library(rng)
    L = read_len
    cols = {k: [] for k, _ in ALL_FIELDS}
    genome, truth = {}, []
    total_len = float(sum(contig_lens))
    frag_base = 0
    cx_read, cx_off, blk_s, blk_e, blk_i = [], [np.zeros(1, np.int64)], [], [], []
    pool_read, pool_len, pool_codes, pool_quals = [], [], [], []
    read_base = 0
    for ci, G in enumerate(contig_lens):
        name = "contig%d" % (ci + 1)
        ref = rng.integers(0, 4, G).astype(np.uint8)
        genome[name] = ref
        n_ins = int(round(n_insertions * G / total_len)) if len(contig_lens) > 1 else n_insertions
        if G < 5000:
            n_ins = 0
        # ---- sample genome: ref[:p+t] + IS + ref[p:] at every insertion ------------------------------------------
        if n_ins:
            slots = np.linspace(1000, G - 1000, n_ins + 2)[1:-1]
            pos = np.sort((slots + rng.integers(-200, 200, n_ins)).astype(np.int64))
            tsd = rng.choice(np.asarray(tsd_choices), n_ins).astype(np.int64)
            which = rng.integers(0, len(lib), n_ins)
        else:
            pos = np.zeros(0, np.int64); tsd = np.zeros(0, np.int64); which = np.zeros(0, np.int64)
        pieces, seg_kind, seg_s0, seg_r0 = [], [], [], []
        cur_r, cur_s = 0, 0
        for k in range(n_ins):
            p, t = int(pos[k]), int(tsd[k])
            el = lib[int(which[k])]
            pieces.append(ref[cur_r:p + t]); seg_kind.append(0); seg_s0.append(cur_s); seg_r0.append(cur_r)
            cur_s += p + t - cur_r
            pieces.append(el); seg_kind.append(1); seg_s0.append(cur_s); seg_r0.append(-1)
            cur_s += len(el)
            cur_r = p
            truth.append((name, p + t, p - 1))
        pieces.append(ref[cur_r:]); seg_kind.append(0); seg_s0.append(cur_s); seg_r0.append(cur_r)
        cur_s += G - cur_r
        sample = np.concatenate(pieces)
        Gs = cur_s
        seg_kind = np.asarray(seg_kind); seg_s0 = np.asarray(seg_s0, np.int64); seg_r0 = np.asarray(seg_r0, np.int64)
        seg_s1 = np.concatenate([seg_s0[1:], [Gs]])
        # ---- fragments and mates ---------------------------------------------------------------------------------
        n_frag = int(Gs * coverage / (2 * L))
        flen = np.maximum(L, rng.normal(frag_mean, frag_sd, n_frag).astype(np.int64))
        wt = rng.random(n_frag) < wildtype_frac
        span = np.where(wt, G, Gs) - flen
        fs = (rng.random(n_frag) * np.maximum(span, 1)).astype(np.int64)
        frag = np.concatenate([np.arange(n_frag), np.arange(n_frag)]) + frag_base
        frag_base += n_frag
        a = np.concatenate([fs, fs + flen - L])            # sample (or reference, for wild type) start of the mate
        iswt = np.concatenate([wt, wt])
        lowq = np.concatenate([rng.random(n_frag) < low_mapq_frac] * 2)
        n = a.shape[0]
        ref_start = np.zeros(n, np.int64); ref_end = np.zeros(n, np.int64)
        lclip = np.zeros(n, np.int64); rclip = np.zeros(n, np.int64)
        keep = np.ones(n, bool)
        # wild-type mates: plain reference coordinates
        ref_start[iswt] = a[iswt]; ref_end[iswt] = a[iswt] + L
        # sample mates: locate segments of first and last base
        m = ~iswt
        sa = np.searchsorted(seg_s1, a[m], side="right")
        sb = np.searchsorted(seg_s1, a[m] + L - 1, side="right")
        ka, kb = seg_kind[sa], seg_kind[sb]
        idx = np.nonzero(m)[0]
        inside_ref = (sa == sb) & (ka == 0)
        ii = idx[inside_ref]
        ref_start[ii] = seg_r0[sa[inside_ref]] + (a[ii] - seg_s0[sa[inside_ref]]); ref_end[ii] = ref_start[ii] + L
        keep[idx[(sa == sb) & (ka == 1)]] = False            # inside the element: multi-mapper, not placed
        # left part in reference, right part in the element -> right clip (5p site at the junction)
        rc = (sa != sb) & (ka == 0) & (kb == 1)
        ir_ = idx[rc]
        aligned = seg_s1[sa[rc]] - a[ir_]
        ref_start[ir_] = seg_r0[sa[rc]] + (a[ir_] - seg_s0[sa[rc]]); ref_end[ir_] = ref_start[ir_] + aligned
        rclip[ir_] = L - aligned
        keep[ir_[aligned < 20]] = False
        # left part in the element, right part in reference -> left clip (3p site)
        lc = (sa != sb) & (ka == 1) & (kb == 0)
        il = idx[lc]
        clip = seg_s0[sb[lc]] - a[il]
        ref_start[il] = seg_r0[sb[lc]]; ref_end[il] = ref_start[il] + (L - clip)
        lclip[il] = clip
        keep[il[(L - clip) < 20]] = False
        keep[idx[(sa != sb) & ~(rc | lc)]] = False           # anything spanning more than one junction
        # tiny overhangs are aligned through: M extends, terminal base compared with the reference
        tiny_r = keep & (rclip > 0) & (rclip <= 2) & (ref_end + rclip <= G)
        tiny_l = keep & (lclip > 0) & (lclip <= 2) & (ref_start - lclip >= 0)
        # ---- per-read fields -------------------------------------------------------------------------------------
        lkind = np.zeros(n, np.uint8); rkind = np.zeros(n, np.uint8)
        first_op = np.full(n, L, np.int64); last_op = np.full(n, L, np.int64)
        # terminal query bases for the through-aligned overhangs
        tr = np.nonzero(tiny_r)[0]
        last_q = sample[a[tr] + L - 1]
        ref_end[tr] += rclip[tr]
        rkind[tr] = np.where(last_q != ref[ref_end[tr] - 1], 2, 0)
        rclip[tr] = 0
        tl = np.nonzero(tiny_l)[0]
        first_q = sample[a[tl]]
        ref_start[tl] -= lclip[tl]
        lkind[tl] = np.where(first_q != ref[ref_start[tl]], 2, 0)
        lclip[tl] = 0
        sclip_r = keep & (rclip > 0); sclip_l = keep & (lclip > 0)
        rkind[sclip_r] = 1; lkind[sclip_l] = 1
        # sequencing errors on the terminal aligned bases (only where that end is not S-clipped)
        e5 = keep & (lkind == 0) & (rng.random(n) < end_error[0])
        e3 = keep & (rkind == 0) & (rng.random(n) < end_error[1])
        lkind[e5] = 2; rkind[e3] = 2
        # CIGAR op lengths (sctools.py:5-9): first / last op raw length
        aligned_len = ref_end - ref_start
        first_op = np.where(lclip > 0, lclip, aligned_len)
        last_op = np.where(rclip > 0, rclip, aligned_len)
        # small indels in plain 150M reads
        plain = keep & (lclip == 0) & (rclip == 0) & (lkind == 0) & (rkind == 0) & (aligned_len == L)
        cx = plain & (rng.random(n) < indel_frac) & (ref_start + L + 8 < G)
        # ---- order like a sorted BAM -----------------------------------------------------------------------------
        sel = np.nonzero(keep)[0]
        order = sel[np.argsort(ref_start[sel], kind="stable")]
        nr = order.shape[0]
        rs_, re_ = ref_start[order], ref_end[order]
        cxo = cx[order]
        # complex reads: xM kD yM or xM kI yM
        cidx = np.nonzero(cxo)[0]
        ncx = cidx.shape[0]
        cut = rng.integers(20, L - 20, ncx)
        klen = rng.integers(1, 4, ncx)
        is_del = rng.random(ncx) < 0.5
        b0s = rs_[cidx]; b0e = b0s + cut
        b1s = np.where(is_del, b0e + klen, b0e)
        b1e = np.where(is_del, b1s + (L - cut), b1s + (L - cut - klen))
        re_[cidx] = b1e
        fo = first_op[order].copy(); lo = last_op[order].copy()
        fo[cidx] = cut; lo[cidx] = b1e - b1s
        flags = (lkind[order] | (rkind[order] << 2)).astype(np.uint8)
        flags[cidx] |= FLAG_COMPLEX
        cx_read.append(cidx + read_base)
        cx_off.append(cx_off[-1][-1] + 2 * (np.arange(ncx, dtype=np.int64) + 1))
        blk_s.append(np.stack([b0s, b1s], 1).reshape(-1)); blk_e.append(np.stack([b0e, b1e], 1).reshape(-1))
        blk_i.append(np.stack([np.zeros(ncx, np.int64), np.where(is_del, 0, klen)], 1).reshape(-1))
        # links: mates share a name
        fr = frag[order]
        by_frag = np.argsort(fr, kind="stable")
        same = fr[by_frag][1:] == fr[by_frag][:-1]
        link = np.full(nr, -1, np.int64)
        later, earlier = by_frag[1:][same], by_frag[:-1][same]
        link[later] = earlier + read_base
        # pool: clipped reads and reads near an insertion site, or every read
        near = np.zeros(nr, bool)
        if n_ins and pool == "near":
            lo_w, hi_w = pos - 60, pos + tsd + 60
            j = np.searchsorted(hi_w, rs_)                       # first window ending after the read start
            jj = np.minimum(j, n_ins - 1)
            near = (j < n_ins) & (re_ > lo_w[jj]) & (rs_ < hi_w[jj])
        pooled = np.ones(nr, bool) if pool == "all" else (((flags & 15) != 0) | near)
        flags[pooled] |= FLAG_POOL
        pidx_all = np.nonzero(pooled)[0]
        plen = L + (L & 1)
        sample_win = np.lib.stride_tricks.sliding_window_view(sample, L)
        ref_win = np.lib.stride_tricks.sliding_window_view(ref, L)
        a_ord, wt_ord = a[order], iswt[order]
        # in chunks: "all" at the 50 Mb / 200x size is 69 M reads x 150 bases
        CH = 1 << 21
        for c0 in range(0, pidx_all.shape[0], CH):
            pidx = pidx_all[c0:c0 + CH]
            src = a_ord[pidx]
            wtp = wt_ord[pidx]
            seqs = np.zeros((pidx.shape[0], plen), np.uint8)
            samp_rows = np.nonzero(~wtp)[0]; wt_rows = np.nonzero(wtp)[0]
            seqs[samp_rows, :L] = sample_win[src[samp_rows]]     # row copies: no n x L index array
            seqs[wt_rows, :L] = ref_win[src[wt_rows]]
            # sequencing errors inside pooled reads (clip bases get realistic disagreement -> several trie leaves)
            if pool == "near":
                err = rng.random((pidx.shape[0], L)) < clip_error
                seqs[:, :L] = np.where(err, (seqs[:, :L] + rng.integers(1, 4, (pidx.shape[0], L))) % 4, seqs[:, :L])
            else:   # the same error model drawn sparsely (number of errors, then their places): 150 x fewer random numbers
                n_err = int(rng.binomial(pidx.shape[0] * L, clip_error))
                er, ec = rng.integers(0, pidx.shape[0], n_err), rng.integers(0, L, n_err)
                seqs[er, ec] = (seqs[er, ec] + rng.integers(1, 4, n_err).astype(np.uint8)) % 4
            quals = np.zeros((pidx.shape[0], plen), np.uint8)
            if pool == "near":
                quals[:, :L] = rng.integers(2, 41, (pidx.shape[0], L), dtype=np.uint8)
            else:   # raw generator bytes scaled to 2..40: ten times faster than bounded integers, same range
                raw = rng.bit_generator.random_raw((pidx.shape[0] * L + 7) // 8).view(np.uint8)[:pidx.shape[0] * L]
                quals[:, :L] = (2 + ((raw.astype(np.uint16) * 39) >> 8)).astype(np.uint8).reshape(-1, L)
            codes = np.left_shift(np.uint8(1), seqs, dtype=np.uint8)   # A C G T -> 1 2 4 8 (_CODE)
            codes[:, L:] = 0
            pool_read.append(pidx + read_base)
            pool_len.append(np.full(pidx.shape[0], plen, np.int64))
            pool_codes.append(pack_nibbles(codes.reshape(-1))); pool_quals.append(quals.reshape(-1))
        # columns
        cols["ref_start"].append(rs_); cols["ref_end"].append(re_); cols["link"].append(link)
        cols["contig_id"].append(np.full(nr, ci)); cols["qlen"].append(np.full(nr, L))
        cols["first_op_len"].append(fo); cols["last_op_len"].append(lo)
        cols["mapq"].append(np.where(lowq[order], 7, 60)); cols["flags"].append(flags)
        read_base += nr
    arrays = {}
    for k, t in ALL_FIELDS:
        if cols[k]:
            arrays[k] = np.concatenate(cols[k]).astype(t)
    arrays["contig_len"] = np.asarray(contig_lens, dtype=np.int32)
    arrays["contig_read_start"] = np.searchsorted(arrays["contig_id"], np.arange(len(contig_lens) + 1)).astype(np.int64)
    arrays["cx_read"] = np.concatenate(cx_read).astype(np.uint32)
    arrays["cx_off"] = np.concatenate(cx_off).astype(np.uint32)
    arrays["blk_start"] = np.concatenate(blk_s).astype(np.int32)
    arrays["blk_end"] = np.concatenate(blk_e).astype(np.int32)
    arrays["blk_ins"] = np.concatenate(blk_i).astype(np.int32)
    arrays["pool_read"] = np.concatenate(pool_read).astype(np.uint32)
    plen_all = np.concatenate(pool_len)
    arrays["pool_off"] = np.concatenate([[0], np.cumsum(plen_all)]).astype(np.uint64)
    arrays["pool_bases"] = np.concatenate(pool_codes)   # per chunk an even number of bases: the packed pieces join on byte boundaries
    arrays["pool_quals"] = np.concatenate(pool_quals).astype(np.uint8)
    max_span = int((arrays["ref_end"].astype(np.int64) - arrays["ref_start"]).max()) if read_base else 1
    packed = PackedReads(list(genome.keys()), arrays, max_span)
    genome_str = {k: "ACGT".encode()[0:0].join([]) for k in ()}  # placeholder to keep the type obvious
    genome_str = {k: np.frombuffer(b"ACGT", np.uint8)[v].tobytes().decode("ascii") for k, v in genome.items()}
    return packed, genome_str, truth
