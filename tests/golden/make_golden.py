"""Generates the committed golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.json.gz

For every case the reference (imported from /root/reference through oracle/ref_harness.py) is run on a set of
alignments; inputs and the reference's outputs are stored together, so the tests need neither the reference nor this
script.  The fixtures hold data only: alignments (name, contig, start, mapq, CIGAR, MD, bases, qualities), the genome,
the parameters, and what the reference printed (findflanks TSV, the logged "N sites remain" counters, pairflanks TSV)
plus unit vectors from sctools / misc / flanktrie / SoftclipConsensus.

pairflanks: the reference cannot run EMBOSS einverted here; its inverted-repeat candidates come from this build's
restatement (oracle ir_report) and then go through the reference's own acceptance code, so has_IR / IR_* columns in the
fixtures are "parity unpinned" while every other column is the reference's own.
"""
import contextlib
import gzip
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from mustache_amd.records import AlignedRead, ReadSet, make_md, parse_cigar_string, cigar_to_string  # noqa: E402
from mustache_amd.simulate import simulate_readset  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

ORACLE = Oracle()


def read_to_json(r):
    return [r.query_name, r.reference_name, r.reference_start, r.mapping_quality, cigar_to_string(r.cigartuples),
            r.tags.get("MD"), r.query_sequence, "".join(chr(33 + int(q)) for q in r.query_qualities), r.flag]


def run_case(name, genome, rs, ff_params=None, pf_params=None, note=""):
    ff_params, pf_params = ff_params or {}, pf_params or {}
    tmp = tempfile.mkdtemp()
    ff, pf = os.path.join(tmp, "case.findflanks.tsv"), os.path.join(tmp, "case.pairflanks.tsv")
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        df, _, stats = rh.run_findflanks(rs, sample_id="sample1", output_file=ff, **ff_params)
        pdf, ptsv = rh.run_pairflanks(ff, rs, genome, ORACLE.ir_report, pf, **pf_params)
    case = {"name": name, "note": note, "contigs": rs.contigs, "genome": genome,
            "reads": [read_to_json(r) for r in rs.reads], "findflanks_params": ff_params, "pairflanks_params": pf_params,
            "findflanks_tsv": open(ff).read(), "findflanks_log_counts": stats, "pairflanks_tsv": ptsv}
    with gzip.open(os.path.join(HERE, name + ".json.gz"), "wt") as fh:
        json.dump(case, fh)
    print("%-14s reads=%-5d flanks=%-3d pairs=%-3d counts=%s" % (name, len(rs), len(df), len(pdf), stats))
    return case


# ---------------------------------------------------------------------------------------------------------------
class Builder:
    """Hand-made alignments against a fixed random genome."""

    def __init__(self, seed, contig_lens):
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.genome = {"c%d" % (i + 1): "".join("ACGT"[k] for k in rng.integers(0, 4, n))
                       for i, n in enumerate(contig_lens)}
        self.reads = []
        self.n = 0

    def other(self, base):
        return "ACGT"[("ACGT".index(base) + 1) % 4]

    def add(self, contig, start, cigar, name=None, mapq=60, mm=(), lclip=None, rclip=None, ins=None, quals=None,
            qual=30):
        """mm: query offsets (within the aligned part, 0 = first aligned base) to turn into mismatches."""
        ref = self.genome[contig]
        ct = parse_cigar_string(cigar)
        q, rpos, aligned_idx = [], start, []
        for k, (op, l) in enumerate(ct):
            if op in (0, 7, 8):
                for i in range(l):
                    aligned_idx.append(len(q))
                    q.append(ref[rpos + i])
                rpos += l
            elif op == 1:
                q.extend((ins or "GATTACAGATTACAGATTACAGATTACAGATTACAGATTACA")[:l])
            elif op == 4:
                src = lclip if k == 0 or (k == 1 and ct[0][0] == 5) else rclip
                if src is None:
                    src = "TGCATGCATGGGCCCAAATTTGGGCCCAAATTTCGCGATATCGCGAT" * 4
                q.extend(src[-l:] if (k == 0 or (k == 1 and ct[0][0] == 5)) else src[:l])
            elif op in (2, 3):
                rpos += l
        for m in mm:
            i = aligned_idx[m]
            q[i] = self.other(q[i])
        q = "".join(q)
        rlen = rpos - start
        md = make_md(ref[start:start + rlen], q, ct)
        if quals is None:
            quals = [qual] * len(q)
        name = name or "h%04d" % self.n
        self.n += 1
        self.reads.append(AlignedRead(name, contig, start, mapq, ct, q, list(quals), md=md))
        return self.reads[-1]

    def readset(self):
        return ReadSet([(n, len(s)) for n, s in self.genome.items()], self.reads)


IS_L = "GGTCTGAATCGCCCCATCAAAAGTCCGATTAGCTAGGATCCGATCGTTAGC"      # left end of an element (what 5p reads clip into)
IS_R = "GCTAACGATCGGATCCTAGCTAATCGGACTTTTGATGGGGCGATTCAGACC"      # right end (what 3p reads clip out of)


def case_worked_example():
    """SURVEY.md section 8c worked example + neighbours that exercise most quirks of findflanks."""
    b = Builder(11, (5000, 400))
    # insertion 1: TSD [100,105): right-clipped reads end at 105, left-clipped start at 100
    for n in (10, 14, 18, 23):
        b.add("c1", 45, "60M%dS" % n, rclip=IS_L)
    for n in (9, 12, 20, 23):
        b.add("c1", 100, "%dS60M" % n, lclip=IS_R)
    for s in (40, 43, 46, 49, 52, 55):
        b.add("c1", s, "100M")
    b.add("c1", 50, "100M", name="span1")
    b.add("c1", 51, "100M", name="span2")
    # insertion 2 at [1000,1004): mates overlapping the same site (quirk Q3/Q4), quality-driven merge, a third cluster
    alt = IS_L[:6] + "T" + IS_L[7:]
    b.add("c1", 940, "64M30S", name="m1", rclip=IS_L, quals=[20] * 94)
    b.add("c1", 950, "54M22S", name="m1", rclip=alt, quals=[35] * 76)        # mate wins positions 0..21
    b.add("c1", 930, "74M25S", name="m2", rclip=IS_L)
    b.add("c1", 945, "59M12S", name="m3", rclip=IS_L)
    b.add("c1", 948, "56M40S", name="m4", rclip="ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCA")   # unrelated flank: own cluster
    b.add("c1", 949, "55M40S", name="m5", rclip="ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCA")
    b.add("c1", 1000, "18S70M", name="m6", lclip=IS_R)
    b.add("c1", 1000, "28S60M", name="m7", lclip=IS_R)
    b.add("c1", 1000, "35S60M", name="m7", lclip=IS_R, quals=[12] * 95)      # mate pair on the 3p side
    b.add("c1", 1000, "9S80M", name="m8", lclip=IS_R)
    for s in range(905, 990, 9):
        b.add("c1", s, "100M")
    # quirk Q1/Q2: MD-mismatch "clips": 100M whose last base mismatches at 1004-1 ... and first base at 1000
    b.add("c1", 904, "100M", name="q1", mm=(99,))      # right mismatch-kind clip at site 1003
    b.add("c1", 1000, "90M", name="q2", mm=(0,))       # left mismatch-kind clip at site 1000
    # low mapq and double-clipped short-inner reads are ignored (Q17, a1)
    b.add("c1", 960, "44M30S", name="lowq", mapq=5, rclip=IS_L)
    b.add("c1", 990, "40S14M40S", name="dbl", lclip=IS_R, rclip=IS_L)
    b.add("c1", 980, "10H5S24M30S", name="hard", rclip=IS_L)    # hard clip first: left end falls to the MD rule
    # insertion 3 at [2000,2003) with indel evidence around it (a8 categories), kept by a permissive -maxir
    for n in (15, 22, 31, 12, 40):
        b.add("c1", 1950, "53M%dS" % n, rclip=IS_L)
    for n in (11, 19, 27, 33):
        b.add("c1", 2000, "%dS64M" % n, lclip=IS_R)
    b.add("c1", 1940, "60M4I36M", name="si3")          # insertion right after 1999: small_insertion_3p at 1999
    b.add("c1", 1941, "62M35I20M", name="li5")         # 35 bases before 2003: large_insertion_5p at 2003
    b.add("c1", 1960, "40M35I40M", name="li3")         # after 1999 -> large_insertion_3p at site 1999
    b.add("c1", 1935, "68M5I30M", name="si5")          # insertion before 2003 -> small_insertion_5p at site 2003
    b.add("c1", 1930, "68M6D30M", name="del1")         # deletion [1998,2004) covers both sites
    b.add("c1", 1931, "66M2D30M", name="up1")          # deletion [1997,1999): upstream of 1999
    b.add("c1", 1932, "72M3D30M", name="down1")        # deletion [2004,2007): downstream of 2003
    for s in range(1910, 1990, 6):
        b.add("c1", s, "100M")
    # site at the very start / end of a contig (pos -1 and pos == contig length)
    for n in (12, 17, 25):
        b.add("c2", 0, "%dS50M" % n, lclip=IS_R)
    for n in (10, 14, 21):
        b.add("c2", 3, "%dS50M" % n, lclip=IS_R[:-3] + "AAA")
    for n in (13, 16, 22):
        b.add("c2", 340, "60M%dS" % n, rclip=IS_L)
    for n in (9, 18):
        b.add("c2", 8 - 60 + 52, "8M%dS" % n, rclip=IS_L) if False else None
    b.add("c2", 0, "9M30S", name="e1", rclip=IS_L)
    b.add("c2", 0, "9M24S", name="e2", rclip=IS_L)
    b.add("c2", 0, "120M")
    b.add("c2", 2, "120M")
    return b.genome, b.readset()


def unit_vectors():
    """sctools / misc / flanktrie / SoftclipConsensus known answers computed by the reference."""
    sctools, misc, flanktrie, pysamtools = rh.reference_modules()
    from mustache.findflanks import SoftclipConsensus
    out = {}
    # misc.takeClosest* -- the table the reference prints in its __main__ (misc.py:80-116), extended
    tc = []
    for lst in ([], [1], [1, 2], [1, 2, 3], [5, 9, 30, 31, 77]):
        for x in range(-1, 80 if len(lst) > 3 else 7):
            try:
                s = misc.takeClosestSmaller(lst, x)
            except IndexError:
                s = "IndexError"
            try:
                l = misc.takeClosestLarger(lst, x)
            except IndexError:
                l = "IndexError"
            tc.append([lst, x, s, l])
    out["take_closest"] = tc
    # sctools on a zoo of CIGAR/MD shapes
    b = Builder(5, (600,))
    zoo = [("10S50M", {}), ("50M10S", {}), ("60M", {"mm": (0,)}), ("60M", {"mm": (59,)}), ("60M", {"mm": (58,)}),
           ("60M", {"mm": (0, 59)}), ("5S50M7S", {}), ("3H50M", {"mm": (0,)}), ("50M4H", {"mm": (49,)}),
           ("30M2D30M", {}), ("30M5I30M", {"mm": (59,)}), ("60M", {}), ("8S12M9S", {}), ("20M3D10M", {"mm": (20,)}),
           ("2H6S40M", {}), ("40M6S2H", {})]
    sc = []
    for k, (cig, kw) in enumerate(zoo):
        r = b.add("c1", 100 + k, cig, **kw)
        row = {"read": read_to_json(r),
               "is_left": sctools.is_left_softclipped_lenient(r), "is_right": sctools.is_right_softclipped_lenient(r),
               "left_len": sctools.left_softclip_length(r), "right_len": sctools.right_softclip_length(r),
               "left_seq": sctools.left_softclipped_sequence(r), "right_seq": sctools.right_softclipped_sequence(r),
               "left_qual": sctools.left_softclip_qualities(r), "right_qual": sctools.right_softclip_qualities(r),
               "inner21": sctools.read_meets_min_alignment_inner_length(r, 21),
               "inner45": sctools.read_meets_min_alignment_inner_length(r, 45),
               "blocks": r.get_blocks()}
        row["left_site"] = sctools.left_softclipped_site_lenient(r) if row["is_left"] else None
        row["right_site"] = sctools.right_softclipped_site_lenient(r) if row["is_right"] else None
        sc.append(row)
    out["sctools"] = sc
    # SoftclipConsensus micro-cases: words (already oriented) -> traverse_all, clusters, consensus, counts
    rng = np.random.default_rng(3)
    base = "GGTCTGAATCGCCCCATCAAAAGTCCGATTAGCTAGGATCC"
    cons = []
    for case in range(12):
        words = []
        n = int(rng.integers(2, 14))
        for _ in range(n):
            L = int(rng.integers(1, len(base)))
            w = list(base[:L])
            for p in range(L):
                if rng.random() < (0.03 if case % 3 else 0.12):
                    w[p] = "ACGTN"[int(rng.integers(0, 5))]
            if rng.random() < 0.15:
                w = list("TTTTGACCA"[:int(rng.integers(1, 9))])
            words.append(("".join(w), [int(x) for x in rng.integers(2, 41, len(w))]))
        t = flanktrie.Trie()
        for w, q in words:
            t.add(w.upper(), q)
        trav = t.traverse_all()
        sc_obj = SoftclipConsensus.__new__(SoftclipConsensus)
        sc_obj.min_count_consensus = int(rng.integers(1, 4))
        seq_dict = {s[0]: [s[1], s[2]] for s in trav}
        # make the reference's hash-ordered singleton set deterministic for the fixture: leaf (DFS) order
        clusters = sc_obj.get_sequence_clusters(seq_dict)
        singles = [c for c in clusters if len(c) == 1 and not _in_component(c, clusters)]
        order = {w: i for i, (w, _, _) in enumerate(trav)}
        n_single = _count_leading_singletons(seq_dict, clusters)
        head = sorted(clusters[:n_single], key=lambda c: order[next(iter(c))])
        clusters = head + clusters[n_single:]
        cseqs = sc_obj.get_cluster_consensus_seqs(clusters)
        counts = sc_obj.merge_cluster_counts(clusters, t)
        cons.append({"words": words, "mcc": sc_obj.min_count_consensus,
                     "leaves": [[w, q, c] for w, q, c in trav],
                     "clusters": [list(c.keys()) for c in clusters], "consensus": cseqs, "counts": counts})
        del singles
    out["consensus"] = cons
    with gzip.open(os.path.join(HERE, "unit_vectors.json.gz"), "wt") as fh:
        json.dump(out, fh)
    print("unit vectors: %d takeClosest rows, %d sctools reads, %d consensus cases" % (len(tc), len(sc), len(cons)))


def _in_component(c, clusters):
    return False


def _count_leading_singletons(seq_dict, clusters):
    """get_sequence_clusters returns [singletons..., components...] (findflanks.py:1069-1075); with one leaf it
    returns a single cluster.  Count the leading no-match entries by re-deriving the match relation."""
    if len(seq_dict) == 1:
        return 1
    seqs = list(seq_dict.keys())
    matched = set()
    for i in range(len(seqs)):
        for j in range(i + 1, len(seqs)):
            m = min(len(seqs[i]), len(seqs[j]))
            d = rh._levenshtein(seqs[i][:m], seqs[j][:m])
            if 1 - (d / m) >= 0.75:
                matched.add(seqs[i])
                matched.add(seqs[j])
    return len(seqs) - len(matched)


if __name__ == "__main__":
    g, rs, _ = simulate_readset(seed=1, contig_lens=(8000, 5000), n_insertions=3, coverage=30)
    run_case("sim_a", g, rs, note="two contigs, CLI defaults")
    g, rs, _ = simulate_readset(seed=2, contig_lens=(9000,), n_insertions=4, coverage=60, wildtype_frac=0.35)
    run_case("sim_b", g, rs, ff_params=dict(min_softclip_length=4, min_softclip_count=4, min_softclip_ratio=0.15),
             note="class defaults 4/4/0.15; wild-type fragments give spanning reads")
    g, rs, _ = simulate_readset(seed=3, contig_lens=(6000,), n_insertions=3, coverage=45, read_len=75, sub_rate=0.03)
    run_case("sim_c", g, rs, ff_params=dict(min_count_consensus=3, max_indel_ratio=0.2),
             pf_params=dict(max_direct_repeat_length=10), note="3% substitutions: many leaves and clusters")
    g, rs, _ = simulate_readset(seed=4, contig_lens=(2500, 2500, 2500), n_insertions=1, coverage=80, read_len=150,
                                wildtype_frac=0.1, frag_mean=400)
    run_case("sim_d", g, rs, note="2x150, deeper, three contigs")
    g, rs = case_worked_example()
    run_case("quirks", g, rs, ff_params=dict(max_indel_ratio=0.5, min_softclip_length=8),
             note="hand-made: worked example of SURVEY 8c, mate merges, MD-mismatch clips, hard clips, contig edges, "
                  "insertion/deletion categories")
    unit_vectors()
