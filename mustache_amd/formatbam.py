"""Host side of ``mustache formatbam``: same function name and arguments as the reference's ``_formatbam``
(reference mustache/formatbam.py:61-94), without samtools or pysam (neither is in this image).

What the reference does with three tools, here on the raw BAM records (every field and tag of a record is kept byte for
byte; nothing is decoded that does not have to be):

1. ``samtools view -b -h -F 0x900`` (samtools.py:9-18): drop secondary (0x100) and supplementary (0x800) alignments.
   The input may be BAM or SAM text (plain or gzip-compressed), as samtools detects it; SAM lines are encoded to BAM records
   the way htslib does (smallest integer type for ``i`` tags).
   Without this step a chimeric read gives three or more alignments of one query name at one site-side and findflanks
   stops with its READ NUMBER ERROR (findflanks.py:971-978, quirk Q3).
2. paired data (format_for_mustache, formatbam.py:19-59): consecutive records are mates; each mapped mate gets the
   other's sequence and quality string as tags ``MT`` / ``MQ``; unmapped reads are dropped; two consecutive records
   with different names are the reference's fatal error.  (The reference pairs with ``next()`` inside a generator,
   which under the Python 3.6 of its environment ends quietly at the end of the file: a trailing single record is
   dropped.  Same here.)
3. ``samtools sort`` (samtools.py:21-30): by (reference id, position, strand), unmapped reads last; ties keep their
   input order here (samtools does not promise an order for them).  The ``@HD`` line says ``SO:coordinate``.
4. ``samtools index`` (samtools.py:32-38): ``<out_bam>.bai`` with the binning index (and htslib's metadata pseudo-bin 37450
   per reference: the span of its records, its mapped / unmapped counts, what ``samtools idxstats`` reads), the 16 kb
   linear index and the count of reads without coordinates (SAM specification section 5.2).

The hot path does not need the index (the interval joins run on the GPU over the packed arrays); it is written so that
the file is a drop-in for every other consumer of the reference's formatbam output.
"""
from __future__ import annotations

import gzip
import multiprocessing as mp
import os
import struct
import sys
import zlib

import numpy as np

from .bamio import _BGZF_EOF, _SEQ_LUT, _reg2bin, bgzf_block

_META_BIN = 37450

_BLOCK = 0xFF00
_REF_CONSUMING = (0, 2, 3, 7, 8)     # M D N = X


def log(msg):
    print("[mustache_amd] " + msg, file=sys.stderr)


def read_raw_bam(path):
    """(header text, contigs [(name, length)], list of raw record bodies without their block_size field).

    The reference's first step, ``samtools view -b -h -F 0x900 {in_sam}`` (samtools.py:10), takes SAM text as well as BAM
    (the CLI argument is called in_sam): a file that does not start with the gzip magic is read as SAM text."""
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic != b"\x1f\x8b":
        return read_sam_text(path)
    with gzip.open(path, "rb") as fh:          # BGZF is a series of gzip members
        data = fh.read()
    if data[:4] != b"BAM\1":
        if data[:1] == b"@" or data.count(b"\t", 0, 4096) >= 10:      # gzip-compressed SAM text
            return read_sam_text(path, data)
        raise ValueError("%s is not a BAM file" % path)
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode("ascii", "replace").rstrip("\0")
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, o)
    o += 4
    contigs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, o)
        name = data[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", data, o + 4 + l_name)
        contigs.append((name, l_ref))
        o += 8 + l_name
    records, n = [], len(data)
    while o + 4 <= n:
        size, = struct.unpack_from("<i", data, o)
        if size < 32 or o + 4 + size > n:
            raise ValueError("truncated alignment record in %s" % path)
        records.append(data[o + 4:o + 4 + size])
        o += 4 + size
    return text, contigs, records


_CIGAR_OPS = {c: k for k, c in enumerate("MIDNSHP=X")}
_BASE_CODE = {c: k for k, c in enumerate("=ACMGRSVTWYHKDBN")}
_B_SUB = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}


def _sam_int_tag(v):
    """An integer tag as htslib stores it: the smallest type that holds the value (sam.c, aux 'i' parsing)."""
    if v < 0:
        if v >= -128:
            return b"c" + struct.pack("<b", v)
        if v >= -32768:
            return b"s" + struct.pack("<h", v)
        if v >= -(1 << 31):
            return b"i" + struct.pack("<i", v)
        raise ValueError("integer tag %d does not fit 32 bits" % v)
    if v <= 0xFF:
        return b"C" + struct.pack("<B", v)
    if v <= 0xFFFF:
        return b"S" + struct.pack("<H", v)
    if v <= 0xFFFFFFFF:
        return b"I" + struct.pack("<I", v)
    raise ValueError("integer tag %d does not fit 32 bits" % v)


def _sam_tag(field):
    try:
        tag, typ, val = field.split(":", 2)
    except ValueError:
        raise ValueError("bad SAM tag %r" % field)
    if len(tag) != 2:
        raise ValueError("bad SAM tag %r" % field)
    t = tag.encode()
    if typ == "i":
        return t + _sam_int_tag(int(val))
    if typ == "Z":
        return t + b"Z" + val.encode() + b"\0"
    if typ == "A":
        return t + b"A" + val.encode()[:1]
    if typ == "f":
        return t + b"f" + struct.pack("<f", float(val))
    if typ == "H":
        return t + b"H" + val.encode() + b"\0"
    if typ == "B":
        sub, *items = val.split(",")
        if sub not in _B_SUB:
            raise ValueError("bad SAM tag %r" % field)
        conv = float if sub == "f" else int
        return t + b"B" + sub.encode() + struct.pack("<i", len(items)) + b"".join(struct.pack(_B_SUB[sub], conv(x)) for x in items)
    raise ValueError("bad SAM tag type in %r" % field)


def sam_line_to_record(line, contig_index):
    """One SAM alignment line as a raw BAM record body (SAM specification 1.4 -> 4.2; what ``samtools view -b`` writes)."""
    f = line.split("\t")
    if len(f) < 11:
        raise ValueError("SAM line with %d fields (11 are mandatory): %r" % (len(f), line[:80]))
    qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
    flag, pos, mapq, pnext, tlen = int(flag), int(pos) - 1, int(mapq), int(pnext) - 1, int(tlen)

    def ref_id(name, same):
        if name == "*":
            return -1
        if name == "=":
            return same
        if name not in contig_index:
            raise ValueError("reference %r of read %s is not in the @SQ lines" % (name, qname))
        return contig_index[name]
    rid = ref_id(rname, -1)
    nid = ref_id(rnext, rid)
    cig, end = [], pos
    if cigar != "*":
        num = 0
        for ch in cigar:
            if ch.isdigit():
                num = num * 10 + ord(ch) - 48
            else:
                if ch not in _CIGAR_OPS:
                    raise ValueError("bad CIGAR %r of read %s" % (cigar, qname))
                op = _CIGAR_OPS[ch]
                cig.append((num << 4) | op)
                if op in _REF_CONSUMING:
                    end += num
                num = 0
    if end <= pos or (flag & 4):
        end = pos + 1
    if seq == "*":
        seq = ""
    codes = [_BASE_CODE.get(c, 15) for c in seq.upper()]
    if len(codes) & 1:
        codes.append(0)
    seqb = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    if qual == "*" or not seq:
        qualb = b"\xff" * len(seq)
    else:
        if len(qual) != len(seq):
            raise ValueError("SEQ and QUAL of read %s differ in length" % qname)
        qualb = bytes(ord(c) - 33 for c in qual)
    name = (qname if qname != "*" else "*").encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", rid, pos, len(name), mapq, _reg2bin(pos, end), len(cig), flag, len(seq), nid, pnext, tlen)
    return body + name + struct.pack("<%dI" % len(cig), *cig) + seqb + qualb + b"".join(_sam_tag(t) for t in f[11:])


def read_sam_text(path, data=None):
    """SAM text -> the same triple as ``read_raw_bam``."""
    if data is None:
        with open(path, "rb") as fh:
            data = fh.read()
    head, contigs, records, index = [], [], [], {}
    for ln in data.decode("ascii", "replace").split("\n"):
        ln = ln.rstrip("\r")
        if not ln:
            continue
        if ln.startswith("@") and not records:
            head.append(ln)
            if ln.startswith("@SQ"):
                kv = dict(x.split(":", 1) for x in ln.split("\t")[1:] if ":" in x)
                if "SN" not in kv or "LN" not in kv:
                    raise ValueError("@SQ line without SN / LN in %s" % path)
                index[kv["SN"]] = len(contigs)
                contigs.append((kv["SN"], int(kv["LN"])))
            continue
        records.append(sam_line_to_record(ln, index))
    if not head and not records:
        raise ValueError("%s is neither a BAM file nor SAM text" % path)
    return "\n".join(head) + ("\n" if head else ""), contigs, records


def _core(rec):
    ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
    return ref_id, pos, l_name, n_cig, flag, l_seq


def _name(rec):
    return rec[32:32 + rec[8] - 1]


def _seq_and_qual(rec):
    """(query_sequence, quality string as samtools prints it) of a raw record."""
    _, _, l_name, n_cig, _, l_seq = _core(rec)
    o = 32 + l_name + 4 * n_cig
    nb = (l_seq + 1) >> 1
    seq = "".join(_SEQ_LUT[np.frombuffer(rec, dtype=np.uint8, count=nb, offset=o)])[:l_seq] if l_seq else ""
    q = rec[o + nb:o + nb + l_seq]
    qual = "*" if (l_seq == 0 or q[0] == 0xFF) else bytes(b + 33 for b in q).decode("ascii")
    return seq, qual


_TAG_SIZE = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}


def _set_string_tags(rec, pairs):
    """pysam ``set_tag`` for string values: an existing tag of that name is removed, the new one goes to the end."""
    _, _, l_name, n_cig, _, l_seq = _core(rec)
    o = 32 + l_name + 4 * n_cig + ((l_seq + 1) >> 1) + l_seq
    head, kept, n = rec[:o], [], len(rec)
    drop = {k.encode() for k, _ in pairs}
    while o + 3 <= n:
        tag, typ, start = rec[o:o + 2], rec[o + 2:o + 3], o
        o += 3
        if typ in _TAG_SIZE:
            o += _TAG_SIZE[typ]
        elif typ in (b"Z", b"H"):
            o = rec.index(b"\0", o) + 1
        elif typ == b"B":
            cnt, = struct.unpack_from("<i", rec, o + 1)
            o += 5 + cnt * _TAG_SIZE[rec[o:o + 1]]
        else:
            raise ValueError("bad BAM tag type %r" % typ)
        if tag not in drop:
            kept.append(rec[start:o])
    new = b"".join(k.encode() + b"Z" + v.encode() + b"\0" for k, v in pairs)
    return head + b"".join(kept) + new


def remove_secondary_alignments(records):
    """``samtools view -F 0x900`` (reference samtools.py:9-18)."""
    return [r for r in records if not (struct.unpack_from("<H", r, 14)[0] & 0x900)]


def format_for_mustache(records):
    """Mate tags of paired data (reference formatbam.py:19-59) on raw records."""
    out = []
    for i in range(0, len(records) - 1, 2):
        p1, p2 = records[i], records[i + 1]
        if _name(p1) != _name(p2):
            print("Fatal Error: Mismatched reads in bwa_tools.format()")      # formatbam.py:31-34: message on stdout + sys.exit()
            raise SystemExit()
        u1, u2 = bool(_core(p1)[4] & 4), bool(_core(p2)[4] & 4)
        if u1 and u2:
            continue
        s1, q1 = _seq_and_qual(p1)
        s2, q2 = _seq_and_qual(p2)
        if not u1:
            out.append(_set_string_tags(p1, (("MT", s2), ("MQ", q2))))
        if not u2:
            out.append(_set_string_tags(p2, (("MT", s1), ("MQ", q1))))
    return out


def sort_coordinate(records):
    """``samtools sort``: (reference id, position, strand), reads without a reference last; stable."""
    def key(k):
        ref_id, pos, _, _, flag, _ = _core(records[k])
        return ((ref_id if ref_id >= 0 else 1 << 31), pos + 1, (flag >> 4) & 1, k)
    return [records[k] for k in sorted(range(len(records)), key=key)]


def _sorted_header(text, contigs):
    lines = [ln for ln in text.split("\n") if ln]
    if lines and lines[0].startswith("@HD"):
        fields = [f for f in lines[0].split("\t") if not f.startswith("SO:") and not f.startswith("GO:")]
        lines[0] = "\t".join(fields + ["SO:coordinate"])
    else:
        lines.insert(0, "@HD\tVN:1.6\tSO:coordinate")
    if not any(ln.startswith("@SQ") for ln in lines):
        lines[1:1] = ["@SQ\tSN:%s\tLN:%d" % c for c in contigs]
    return "\n".join(lines) + "\n"


def _compress(chunk):
    return bgzf_block(chunk)


def write_bam_and_index(path, text, contigs, records, procs=8):
    """Write a BAM of raw records and its ``.bai``.  Returns the number of records written."""
    t = text.encode()
    head = [b"BAM\1" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(contigs))]
    for n, l in contigs:
        nb = n.encode() + b"\0"
        head.append(struct.pack("<i", len(nb)) + nb + struct.pack("<i", l))
    head = b"".join(head)
    sizes = np.fromiter((len(r) + 4 for r in records), dtype=np.int64, count=len(records))
    starts = len(head) + np.concatenate([[0], np.cumsum(sizes)])                # stream offset of every record (+ end)
    stream = head + b"".join(struct.pack("<i", len(r)) + r for r in records)
    chunks = [stream[o:o + _BLOCK] for o in range(0, len(stream), _BLOCK)] or [b""]
    if procs > 1 and len(chunks) > 64:
        with mp.get_context("fork").Pool(procs) as pool:
            blocks = pool.map(_compress, chunks, chunksize=16)
    else:
        blocks = [_compress(c) for c in chunks]
    block_off = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int64)
    with open(path, "wb") as fh:
        for b in blocks:
            fh.write(b)
        fh.write(_BGZF_EOF)

    def voff(s):           # virtual file offset of stream offset s (the end of the stream sits at the start of the EOF block)
        s = np.asarray(s, dtype=np.int64)
        blk = s // _BLOCK
        return (block_off[np.minimum(blk, len(blocks))].astype(np.uint64) << np.uint64(16)) | (s % _BLOCK).astype(np.uint64)
    vbeg, vend = voff(starts[:-1]), voff(starts[1:])
    # ---- index (SAM specification 5.2) -----------------------------------------------------------------------------------
    n_ref = len(contigs)
    bins = [dict() for _ in range(n_ref)]
    linear = [dict() for _ in range(n_ref)]
    n_no_coor = 0
    meta = [None] * n_ref                                                        # per reference: [first offset, last end, mapped, unmapped]
    for k, r in enumerate(records):
        ref_id, pos, l_name, n_cig, flag, _ = _core(r)
        if ref_id < 0 or pos < 0:
            n_no_coor += 1
            continue
        if meta[ref_id] is None:
            meta[ref_id] = [int(vbeg[k]), int(vend[k]), 0, 0]
        meta[ref_id][1] = int(vend[k])
        meta[ref_id][3 if flag & 4 else 2] += 1
        end = pos
        for c in struct.unpack_from("<%dI" % n_cig, r, 32 + l_name):
            if (c & 15) in _REF_CONSUMING:
                end += c >> 4
        if end <= pos or (flag & 4):
            end = pos + 1
        b = _reg2bin(pos, end)
        lst = bins[ref_id].setdefault(b, [])
        if lst and lst[-1][1] == int(vbeg[k]):
            lst[-1][1] = int(vend[k])                                            # adjacent records of a bin: one chunk
        else:
            lst.append([int(vbeg[k]), int(vend[k])])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            if w not in linear[ref_id]:
                linear[ref_id][w] = int(vbeg[k])
    out = [b"BAI\1", struct.pack("<i", n_ref)]
    for c in range(n_ref):
        out.append(struct.pack("<i", len(bins[c]) + (1 if meta[c] else 0)))
        for b in sorted(bins[c]):
            out.append(struct.pack("<Ii", b, len(bins[c][b])))
            for beg, end in bins[c][b]:
                out.append(struct.pack("<QQ", beg, end))
        if meta[c]:   # the metadata pseudo-bin htslib writes (SAM specification 5.2: bin 37450 with two "chunks": the span of the
            out.append(struct.pack("<IiQQQQ", _META_BIN, 2, *meta[c]))   # reference's records, its mapped / unmapped counts; idxstats reads it)
        n_intv = max(linear[c]) + 1 if linear[c] else 0
        out.append(struct.pack("<i", n_intv))
        offs, nxt = [0] * n_intv, 0
        for w in range(n_intv - 1, -1, -1):                                       # a window without a read takes the next window's offset (htslib)
            nxt = linear[c].get(w, nxt)
            offs[w] = nxt
        out.append(struct.pack("<%dQ" % n_intv, *offs))
    out.append(struct.pack("<Q", n_no_coor))
    with open(path + ".bai", "wb") as fh:
        fh.write(b"".join(out))
    return len(records)


def _formatbam(in_sam, out_bam, single_end, keep_tmp_files, verbose=True):
    """Drop-in for reference formatbam.py:61-94 (same positional arguments).  ``keep_tmp_files`` is accepted and has
    nothing to keep: the stages pass their records in memory instead of through ``.cleaned.bam.tmp`` /
    ``.formatted.bam.tmp`` files."""
    if verbose:
        log("Removing secondary alignments...")
    text, contigs, records = read_raw_bam(in_sam)
    n_in = len(records)
    records = remove_secondary_alignments(records)
    if verbose:
        log("Successfully removed secondary alignments... (%d of %d records dropped)" % (n_in - len(records), n_in))
    if not single_end:
        if verbose:
            log("Formatting bam file for use by mustache...")
        records = format_for_mustache(records)
        if verbose:
            log("SAM file successfully reformatted...")
    if verbose:
        log("Sorting the BAM file by chromosomal location...")
    records = sort_coordinate(records)
    n = write_bam_and_index(out_bam, _sorted_header(text, contigs), contigs, records)
    if verbose:
        log("BAM file successfully sorted and indexed: %d records in %s (+ .bai)" % (n, out_bam))
    if not os.path.isfile(out_bam) or not os.path.isfile(out_bam + ".bai"):
        log("Fatal error: Failed to write the sorted BAM file and its index")
        raise SystemExit()
    return out_bam
