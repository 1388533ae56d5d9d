"""Self-contained BGZF/BAM reader and writer plus a FASTA reader (pysam/htslib are not in this image).

Only what the flank-discovery path needs: sequential iteration over alignment records of a coordinate-sorted BAM with the
accessors of ``AlignedRead``.  Region queries (``bam.fetch`` in the reference) are not needed on the host -- the interval
joins run on the GPU over the packed arrays.  The writer exists so tests and the simulator can produce real BAM files.
"""
from __future__ import annotations

import gzip
import struct
import zlib

import numpy as np

from .records import AlignedRead, BASE_CODES, BASE_TO_CODE

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_SEQ_LUT = np.array([BASE_CODES[i >> 4] + BASE_CODES[i & 15] for i in range(256)])


class BamReader:
    """Iterates the mapped and unmapped records of a BAM file as ``AlignedRead`` objects (unmapped ones have
    ``cigartuples is None``)."""

    def __init__(self, path):
        self.path = path
        self.fh = gzip.open(path, "rb")  # BGZF is a series of gzip members
        magic = self.fh.read(4)
        if magic != b"BAM\1":
            raise ValueError("%s is not a BAM file" % path)
        l_text, = struct.unpack("<i", self.fh.read(4))
        self.text = self.fh.read(l_text).decode("ascii", "replace")
        n_ref, = struct.unpack("<i", self.fh.read(4))
        self.contigs = []
        for _ in range(n_ref):
            l_name, = struct.unpack("<i", self.fh.read(4))
            name = self.fh.read(l_name)[:-1].decode()
            l_ref, = struct.unpack("<i", self.fh.read(4))
            self.contigs.append((name, l_ref))
        self.header = {"SQ": [{"SN": n, "LN": l} for n, l in self.contigs]}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        self.fh.close()

    def __iter__(self):
        read = self.fh.read
        names = [n for n, _ in self.contigs]
        while True:
            head = read(4)
            if len(head) < 4:
                return
            size, = struct.unpack("<i", head)
            buf = read(size)
            (ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen) = struct.unpack_from(
                "<iiBBHHHiiii", buf, 0)
            o = 32
            qname = buf[o:o + l_name - 1].decode()
            o += l_name
            cig = struct.unpack_from("<%dI" % n_cig, buf, o) if n_cig else ()
            o += 4 * n_cig
            nb = (l_seq + 1) >> 1
            seq = "".join(_SEQ_LUT[np.frombuffer(buf, dtype=np.uint8, count=nb, offset=o)])[:l_seq] if l_seq else ""
            o += nb
            quals = np.frombuffer(buf, dtype=np.uint8, count=l_seq, offset=o)
            if l_seq and quals[0] == 0xFF:
                quals = None
            o += l_seq
            tags = _parse_tags(buf, o)
            cigartuples = [(c & 15, c >> 4) for c in cig] if (n_cig and not flag & 4 and ref_id >= 0) else None
            yield AlignedRead(qname, names[ref_id] if ref_id >= 0 else None, pos, mapq, cigartuples, seq, quals,
                              flag=flag, reference_id=ref_id, tags=tags)


_TAG_FMT = {b"A": "<c", b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I", b"f": "<f"}
_TAG_SIZE = {k: struct.calcsize(v) for k, v in _TAG_FMT.items()}


def _parse_tags(buf, o):
    """Optional fields of a record -> dict (integers, floats, characters and strings; arrays are skipped)."""
    n, out = len(buf), {}
    while o + 3 <= n:
        tag, typ = buf[o:o + 2].decode("ascii", "replace"), buf[o + 2:o + 3]
        o += 3
        if typ in _TAG_FMT:
            v, = struct.unpack_from(_TAG_FMT[typ], buf, o)
            out[tag] = v.decode("ascii", "replace") if typ == b"A" else v
            o += _TAG_SIZE[typ]
        elif typ in (b"Z", b"H"):
            e = buf.index(b"\0", o)
            out[tag] = buf[o:e].decode()
            o = e + 1
        elif typ == b"B":
            sub = buf[o:o + 1]
            cnt, = struct.unpack_from("<i", buf, o + 1)
            o += 5 + cnt * _TAG_SIZE[sub]
        else:
            raise ValueError("bad BAM tag type %r" % typ)
    return out


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def encode_record(r, cid, name=None):
    """One alignment as BAM bytes (block_size included).  ``cid`` = reference id; ``name`` overrides r.query_name."""
    name = (r.query_name if name is None else name).encode() + b"\0"
    cig = r.cigartuples or []
    seq = r.query_sequence or ""
    codes = [BASE_TO_CODE.get(c, 15) for c in seq]
    if len(codes) & 1:
        codes.append(0)
    seqb = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    q = r.query_qualities
    qual = bytes(bytearray(int(x) for x in q)) if q is not None else b"\xff" * len(seq)
    tags = b""
    for k, v in r.tags.items():
        if isinstance(v, str):
            tags += k.encode() + b"Z" + v.encode() + b"\0"
        else:
            tags += k.encode() + b"i" + struct.pack("<i", int(v))
    end = r.reference_end if cig else r.reference_start + 1
    body = struct.pack("<iiBBHHHiiii", cid, r.reference_start, len(name), r.mapping_quality,
                       _reg2bin(r.reference_start, end), len(cig), r.flag, len(seq), -1, -1, 0)
    body += name + b"".join(struct.pack("<I", (l << 4) | op) for op, l in cig) + seqb + qual + tags
    return struct.pack("<i", len(body)) + body


def bgzf_block(data, level=6):
    """One BGZF block (SAM spec 4.1) holding ``data`` (at most 0xFF00 bytes, so that it always fits)."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    return (struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def bam_header_bytes(contigs, text=None):
    if text is None:
        text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)
    t = text.encode()
    out = [b"BAM\1" + struct.pack("<i", len(t)) + t + struct.pack("<i", len(contigs))]
    for n, l in contigs:
        nb = n.encode() + b"\0"
        out.append(struct.pack("<i", len(nb)) + nb + struct.pack("<i", l))
    return b"".join(out)


class BamWriter:
    def __init__(self, path, contigs, text=None):
        self.fh = open(path, "wb")
        self.buf = bytearray()
        self.contig_index = {n: i for i, (n, _) in enumerate(contigs)}
        self._put(bam_header_bytes(contigs, text))

    def _put(self, data):
        self.buf += data
        while len(self.buf) >= 0xFF00:
            self._flush_block(bytes(self.buf[:0xFF00]))
            del self.buf[:0xFF00]

    def _flush_block(self, data):
        self.fh.write(bgzf_block(data))

    def write(self, r):
        cid = self.contig_index[r.reference_name] if r.reference_name is not None else -1
        self._put(encode_record(r, cid))

    def close(self):
        if self.buf:
            self._flush_block(bytes(self.buf))
            self.buf = bytearray()
        self.fh.write(_BGZF_EOF)
        self.fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_bam(path, readset):
    with BamWriter(path, readset.contigs) as w:
        for r in readset:
            w.write(r)


def read_fasta(path):
    """FASTA -> dict id -> sequence (case preserved; id = first word of the header, like Bio.SeqIO rec.id)."""
    out, name, chunks = {}, None, []
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    out[name] = "".join(chunks)
                name, chunks = line[1:].split()[0] if line[1:].split() else "", []
            else:
                chunks.append(line.strip())
    if name is not None:
        out[name] = "".join(chunks)
    return out


def read_fasta_arrays(path):
    """FASTA -> dict id -> uint8 array of the sequence bytes: ``read_fasta`` for callers that want bytes (pairflanks' genome
    on the device), without a Python loop over the lines (a 60-column FASTA of a few Mb is a hundred thousand lines).
    Records whose lines carry blanks or tabs go through ``read_fasta``'s line rule (strip) one by one."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        raw = fh.read()
    data = np.frombuffer(raw, dtype=np.uint8)
    n = data.shape[0]
    if n == 0:
        return {}
    gt = np.flatnonzero(data == 62)
    starts = gt[(gt == 0) | (data[np.maximum(gt - 1, 0)] == 10)]                  # '>' at the start of a line
    nl = np.flatnonzero(data == 10)
    out = {}
    for k, s0 in enumerate(starts.tolist()):
        e = int(nl[np.searchsorted(nl, s0)]) if nl.size and nl[-1] > s0 else n   # end of the header line
        words = raw[s0 + 1:e].decode("ascii", "replace").split()
        name = words[0] if words else ""
        s1 = int(starts[k + 1]) if k + 1 < starts.shape[0] else n
        body = data[min(e + 1, s1):s1]
        if body.size and np.any((body == 32) | (body == 9) | (body == 11) | (body == 12)):
            seq = "".join(ln.strip() for ln in raw[min(e + 1, s1):s1].decode("ascii", "replace").split("\n"))
            out[name] = np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)
        else:
            out[name] = body[(body != 10) & (body != 13)]
    return out


def write_fasta(path, genome, width=60):
    with open(path, "w") as fh:
        for name, seq in genome.items():
            fh.write(">%s\n" % name)
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width] + "\n")
