"""``--ir-backend=einverted``: the escape hatch of SURVEY.md 8c.  The reference finds inverted repeats by shelling out to
EMBOSS ``einverted`` (reference mustache/embosstools.py:8-26, pairflanks.py:172-219); this build has its own detector
(parity unpinned because EMBOSS is not in the image).  Where the binary IS installed, this module
runs it exactly as the reference does -- one run per candidate pair on ``seq5[:40] + 'N'*20 + seq3[-40:]`` -- applies
the reference's acceptance loop, and hands the result to the library (``mfk_pairflanks_set_ir``), so the IR columns
come out reference-identical.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile

import numpy as np

TRUNCATED_FLANK_LENGTH = 40      # pairflanks.py:78
IR_DISTANCE_FROM_END = 15        # pairflanks.py:78


class EinvertedNotFound(RuntimeError):
    pass


def find_einverted(binary="einverted"):
    path = shutil.which(binary)
    if path is None:
        raise EinvertedNotFound("--ir-backend=einverted needs the EMBOSS 'einverted' executable on PATH (the reference "
                                "pins EMBOSS 6.6.0, envs/environment.yml:8); use --ir-backend=builtin otherwise")
    return path


def combined_sequence(seq5, seq3):
    """truncate_sequence + join, pairflanks.py:183-186, 268-275"""
    return seq5[:TRUNCATED_FLANK_LENGTH] + "N" * 20 + (seq3[-TRUNCATED_FLANK_LENGTH:] if len(seq3) > TRUNCATED_FLANK_LENGTH else seq3)


def parse_outseq(text):
    """read_emboss_seq_results, embosstools.py:17-26: FASTA ids ``<name>_<start>_<end>``, records come in pairs."""
    recs = []
    for line in text.splitlines():
        if line.startswith(">"):
            d = line[1:].split()[0].split("_")
            recs.append((int(d[1]), int(d[2])))
    return [(recs[i][0], recs[i][1], recs[i + 1][0], recs[i + 1][1]) for i in range(0, len(recs) - 1, 2)]


def run_einverted(seq, binary, tmp_dir):
    """run_einverted, embosstools.py:8-14 (same flags).  Returns [(arm1_start, arm1_end, arm2_start, arm2_end), ...]."""
    fa, out, outseq = (os.path.join(tmp_dir, n) for n in ("in.fasta", "einverted.out", "einverted.outseq"))
    with open(fa, "w") as fh:
        fh.write(">0 0\n%s\n" % seq)
    for p in (out, outseq):
        if os.path.exists(p):
            os.remove(p)
    subprocess.run([binary, "-sequence", fa, "-gap", "12", "-threshold", "15", "-match", "3", "-mismatch", "-4",
                    "-outfile", out, "-outseq", outseq, "-auto", "Y", "-warning", "N"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(outseq):
        return []
    with open(outseq) as fh:
        return parse_outseq(fh.read())


def accept(repeats, seqlen):
    """The acceptance loop of pairflanks.py:196-205 with pair_has_ir 253-266: ``ir_length`` is never updated there, so
    the LAST repeat that passes wins.  Returns (has_ir, a1s, a1e, a2s, a2e)."""
    res = (0, 0, 0, 0, 0)
    for a1s, a1e, a2s, a2e in repeats:
        if a1s <= IR_DISTANCE_FROM_END and a2e >= seqlen - IR_DISTANCE_FROM_END and a1e - a1s + 1 > 0:
            res = (1, a1s, a1e, a2s, a2e)
    return res


def candidate_pairs(tab, max_direct_repeat_length):
    """pair_all_nearby_flanks, pairflanks.py:138-169: (row of the 5p flank, row of the 3p flank) in the reference's order."""
    out = []
    three = np.nonzero(tab["orient"] == 3)[0]
    for a in np.nonzero(tab["orient"] == 5)[0]:
        p5, c = int(tab["pos"][a]), int(tab["contig_id"][a])
        for b in three:
            if int(tab["contig_id"][b]) == c and p5 - max_direct_repeat_length - 1 <= int(tab["pos"][b]) < p5:
                out.append((int(a), int(b)))
    return out


def ir_table(tab, seqs, max_direct_repeat_length, detector=None, binary="einverted"):
    """Rows for ``Binding.set_pair_ir``: (index_5p, index_3p, ir[n, 5]).  ``detector(seq) -> repeats`` replaces the
    executable in tests."""
    tmp = None
    if detector is None:
        exe = find_einverted(binary)
        tmp = tempfile.mkdtemp(prefix="mfk_einverted_")
        detector = lambda s: run_einverted(s, exe, tmp)   # noqa: E731
    try:
        i5, i3, ir = [], [], []
        for a, b in candidate_pairs(tab, max_direct_repeat_length):
            s5 = seqs[int(tab["seq_off"][a]):int(tab["seq_off"][a]) + int(tab["seq_len"][a])].decode("ascii")
            s3 = seqs[int(tab["seq_off"][b]):int(tab["seq_off"][b]) + int(tab["seq_len"][b])].decode("ascii")
            comb = combined_sequence(s5, s3)
            i5.append(a)
            i3.append(b)
            ir.append(accept(detector(comb), len(comb)))
        return np.array(i5, dtype=np.int32), np.array(i3, dtype=np.int32), np.array(ir, dtype=np.int32).reshape(-1, 5)
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
