"""Alignment record model for the flank-discovery path.

``AlignedRead`` is a small, pysam-compatible view of one BAM alignment.  It carries exactly the accessors the
reference's hot path touches: ``query_name, reference_name, reference_start, reference_end,
mapping_quality, cigartuples, query_sequence, query_qualities, get_tag('MD'), get_blocks(),
get_reference_positions(full_length=), get_reference_sequence()``.  The host-side packer (``pack.py``) and the
container-only reference harness (``oracle/ref_harness.py``) both consume it, so one synthetic read set can be fed to
the reference and to this framework.

The semantics follow pysam 0.13 / htslib (``setup.py:18`` of the reference pins pysam==0.13).
"""
from __future__ import annotations

# BAM CIGAR op codes (SAM spec section 4.2)
CMATCH, CINS, CDEL, CREF_SKIP, CSOFT_CLIP, CHARD_CLIP, CPAD, CEQUAL, CDIFF = range(9)
CIGAR_CHARS = "MIDNSHP=X"
_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)
_CONSUMES_QUERY = (True, True, False, False, True, False, False, True, True)
_ALIGNED = (True, False, False, False, False, False, False, True, True)

# BAM 4-bit base codes; this is also the device-side base alphabet (exact for N / IUPAC).
BASE_CODES = "=ACMGRSVTWYHKDBN"
BASE_TO_CODE = {c: i for i, c in enumerate(BASE_CODES)}
BASE_TO_CODE.update({c.lower(): i for i, c in enumerate(BASE_CODES)})


def parse_cigar_string(cigar: str):
    out, num = [], 0
    for ch in cigar:
        if ch.isdigit():
            num = num * 10 + ord(ch) - 48
        else:
            out.append((CIGAR_CHARS.index(ch), num))
            num = 0
    return out


def cigar_to_string(cigartuples) -> str:
    return "".join("%d%s" % (l, CIGAR_CHARS[op]) for op, l in cigartuples)


class AlignedRead:
    """One mapped alignment.  Hashable by identity (the reference stores reads in ``set``s)."""

    __slots__ = ("query_name", "flag", "reference_name", "reference_id", "reference_start", "mapping_quality",
                 "cigartuples", "query_sequence", "query_qualities", "tags", "_reference_end")

    def __init__(self, query_name, reference_name, reference_start, mapping_quality, cigartuples, query_sequence,
                 query_qualities, md=None, flag=0, reference_id=0, tags=None):
        self.query_name = query_name
        self.flag = flag
        self.reference_name = reference_name
        self.reference_id = reference_id
        self.reference_start = int(reference_start)
        self.mapping_quality = int(mapping_quality)
        self.cigartuples = [(int(o), int(l)) for o, l in cigartuples] if cigartuples is not None else None
        self.query_sequence = query_sequence
        self.query_qualities = query_qualities
        self.tags = dict(tags) if tags else {}
        if md is not None:
            self.tags["MD"] = md
        self._reference_end = None

    # -- pysam-compatible accessors ---------------------------------------------------------------------------
    @property
    def reference_end(self):
        if self._reference_end is None:
            pos = self.reference_start
            for op, l in self.cigartuples:
                if _CONSUMES_REF[op]:
                    pos += l
            self._reference_end = pos
        return self._reference_end

    @property
    def is_reverse(self):
        return bool(self.flag & 16)

    @property
    def is_read2(self):
        return bool(self.flag & 0x80)

    @property
    def is_unmapped(self):
        return bool(self.flag & 4)

    @property
    def query_alignment_length(self):
        """Query bases inside the alignment: the sequence without its soft-clipped ends (pysam semantics)."""
        n = len(self.query_sequence or "")
        ct = self.cigartuples or []
        lo, hi = 0, len(ct)
        while lo < hi and ct[lo][0] == CHARD_CLIP:
            lo += 1
        while hi > lo and ct[hi - 1][0] == CHARD_CLIP:
            hi -= 1
        if lo < hi and ct[lo][0] == CSOFT_CLIP:
            n -= ct[lo][1]
        if hi - lo > 1 and ct[hi - 1][0] == CSOFT_CLIP:
            n -= ct[hi - 1][1]
        return n

    @property
    def cigarstring(self):
        return cigar_to_string(self.cigartuples)

    def get_tag(self, name):
        return self.tags[name]

    def has_tag(self, name):
        return name in self.tags

    def get_blocks(self):
        blocks, pos = [], self.reference_start
        for op, l in self.cigartuples:
            if _ALIGNED[op]:
                blocks.append((pos, pos + l))
                pos += l
            elif op == CDEL or op == CREF_SKIP:
                pos += l
        return blocks

    def get_reference_positions(self, full_length=False):
        out, pos = [], self.reference_start
        for op, l in self.cigartuples:
            if _ALIGNED[op]:
                out.extend(range(pos, pos + l))
                pos += l
            elif op == CINS or op == CSOFT_CLIP:
                if full_length:
                    out.extend([None] * l)
            elif op == CDEL or op == CREF_SKIP:
                pos += l
        return out

    def get_reference_sequence(self):
        """Reference bases over the aligned region rebuilt from the MD tag: matches come from the query (upper
        case), mismatches are lower case, deleted bases are taken from the tag (pysam ``build_alignment_sequence``)."""
        md = self.tags["MD"]
        # aligned query bases, with '-' placeholders for deletions
        seq, qpos = [], 0
        for op, l in self.cigartuples:
            if _ALIGNED[op]:
                seq.extend(self.query_sequence[qpos:qpos + l].upper())
                qpos += l
            elif op == CDEL:
                seq.extend("-" * l)
            elif op == CINS or op == CSOFT_CLIP:
                qpos += l
        out, i, k, n = [], 0, 0, len(md)
        while k < n:
            ch = md[k]
            if ch.isdigit():
                num = 0
                while k < n and md[k].isdigit():
                    num = num * 10 + ord(md[k]) - 48
                    k += 1
                out.extend(seq[i:i + num])
                i += num
            elif ch == "^":
                k += 1
                while k < n and not md[k].isdigit():
                    out.append(md[k])
                    i += 1
                    k += 1
            else:
                out.append(ch.lower())
                i += 1
                k += 1
        return "".join(out)

    def __repr__(self):
        return "AlignedRead(%s %s:%d %s mapq=%d)" % (self.query_name, self.reference_name, self.reference_start,
                                                     self.cigarstring, self.mapping_quality)


def make_md(ref_aligned: str, query: str, cigartuples) -> str:
    """Build the MD tag for a read.  ``ref_aligned`` is the reference over [reference_start, reference_end)."""
    md, run, rpos, qpos = [], 0, 0, 0
    for op, l in cigartuples:
        if _ALIGNED[op]:
            for _ in range(l):
                if ref_aligned[rpos].upper() == query[qpos].upper():
                    run += 1
                else:
                    md.append(str(run))
                    md.append(ref_aligned[rpos].upper())
                    run = 0
                rpos += 1
                qpos += 1
        elif op == CDEL:
            md.append(str(run))
            md.append("^" + ref_aligned[rpos:rpos + l].upper())
            run = 0
            rpos += l
        elif op == CREF_SKIP:
            rpos += l
        elif op == CINS or op == CSOFT_CLIP:
            qpos += l
    md.append(str(run))
    return "".join(md)


class ReadSet:
    """A coordinate-sorted collection of reads with a header; doubles as the duck-typed BAM the reference
    harness feeds to the reference (iteration, ``fetch`` and ``header['SQ']``)."""

    def __init__(self, contigs, reads):
        # contigs: list of (name, length); reads sorted by (contig index, reference_start) like a sorted BAM
        self.contigs = list(contigs)
        self.contig_index = {n: i for i, (n, _) in enumerate(self.contigs)}
        self.reads = list(reads)
        for r in self.reads:
            r.reference_id = self.contig_index[r.reference_name]
        self.reads.sort(key=lambda r: (r.reference_id, r.reference_start))  # stable: keeps ties in input order
        self.header = {"SQ": [{"SN": n, "LN": l} for n, l in self.contigs]}

    def __iter__(self):
        return iter(self.reads)

    def __len__(self):
        return len(self.reads)

    def fetch(self, contig, start, end):
        """htslib region query: every read with reference_start < end and reference_end > start, file order."""
        for r in self.reads:
            if r.reference_name != contig:
                continue
            rend = r.reference_end if r.reference_end > r.reference_start else r.reference_start + 1
            if r.reference_start < end and rend > start:
                yield r
