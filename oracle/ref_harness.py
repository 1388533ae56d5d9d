"""BUILD-CONTAINER ONLY (never runs on the GPU box: /root/reference does not exist there).

Imports the real reference (durrantmm/mustache at /root/reference) with the third-party modules that are absent from
this image replaced by tiny stand-ins in ``sys.modules``, feeds it duck-typed alignments
(``mustache_amd.records.ReadSet``) and returns what it computes.  ``tests/golden/make_golden.py`` uses this to
generate the committed golden vectors; nothing else imports it.

Stand-ins (all for modules the reference imports but whose arithmetic is not under test here):
  pysam.AlignmentFile   -> returns a registered ReadSet (iteration, fetch, header['SQ'])
  jellyfish             -> textbook unit-cost Levenshtein distance (what jellyfish.levenshtein_distance computes)
  pygogo                -> logger that records messages (the "N sites remain" counters are parity probes)
  snakemake.shell       -> no-op (only used to run einverted / rm)
  Bio.SeqIO.parse       -> yields records of a registered genome dict
The one thing the reference cannot do here is run EMBOSS ``einverted`` (pairflanks.py:194): the harness replaces
``embosstools.run_einverted`` / ``read_emboss_seq_results`` by a caller-supplied repeat finder, so the reference's own
acceptance logic (pairflanks.py:196-213, 253-266) still runs on the candidates.  IR parity stays UNPINNED.
"""
import re
import sys
import types

sys.dont_write_bytecode = True  # /root/reference is read-only territory: never write __pycache__ into it
REF_ROOT = "/root/reference"

_registry = {"bam": {}, "genome": {}}
_log = []


def _levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


class _Logger:
    def info(self, msg, *a):
        _log.append(msg % a if a else msg)

    warning = error = debug = info


def install_stubs():
    if "mustache" in sys.modules:
        return
    m = types.ModuleType("pysam")
    m.AlignmentFile = lambda path, mode="rb": _registry["bam"][path]
    sys.modules["pysam"] = m
    m = types.ModuleType("jellyfish")
    m.levenshtein_distance = _levenshtein
    sys.modules["jellyfish"] = m
    m = types.ModuleType("pygogo")

    class Gogo:
        def __init__(self, name=None, verbose=False, **kw):
            self.logger = _Logger()
    m.Gogo = Gogo
    sys.modules["pygogo"] = m
    m = types.ModuleType("snakemake")
    m.shell = lambda *a, **k: None
    sys.modules["snakemake"] = m
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")

    class _Rec:
        def __init__(self, id_, seq):
            self.id, self.seq = id_, seq

    def parse(path, fmt):
        for name, seq in _registry["genome"][path].items():
            yield _Rec(name, seq)
    seqio.parse = parse
    seqio.write = lambda *a, **k: None
    bio.SeqIO = seqio
    for sub, attrs in (("Seq", {"Seq": lambda s, *a: s}), ("SeqRecord", {"SeqRecord": lambda *a, **k: None}),
                       ("Alphabet", {"IUPAC": types.SimpleNamespace(IUPACAmbiguousDNA=None)})):
        mod = types.ModuleType("Bio." + sub)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules["Bio." + sub] = mod
        setattr(bio, sub, mod)
    pw = types.ModuleType("Bio.pairwise2")            # imported by inferseqoverlap.py:9-10, used only in its verbose branch
    pw.format_alignment = lambda *a, **k: ""
    pw.align = types.SimpleNamespace(localms=lambda *a, **k: [])
    sys.modules["Bio.pairwise2"] = pw
    bio.pairwise2 = pw
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = seqio
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


FINDFLANKS_DEFAULTS = dict(min_softclip_length=8, min_softclip_count=2, min_alignment_quality=20,
                           min_alignment_inner_length=21, min_distance_to_mate=22, min_softclip_ratio=0.10,
                           max_indel_ratio=0.03, large_insertion_cutoff=30, min_count_consensus=2)
PAIRFLANKS_DEFAULTS = dict(max_direct_repeat_length=20, min_alignment_quality=20, min_alignment_inner_length=21,
                           max_junction_spanning_prop=0.15, large_insertion_cutoff=30)


def _stats_from_log():
    out = []
    for msg in _log:
        m = re.search(r"(\d+) (sites remain|flank sequences remain)", msg)
        if m:
            out.append(int(m.group(1)))
    return out


def run_findflanks(readset, sample_id="sample", output_file=None, **params):
    """Run the reference's _findflanks on a ReadSet.  Returns (DataFrame, TSV text, list of logged counters)."""
    install_stubs()
    from mustache.findflanks import _findflanks
    p = dict(FINDFLANKS_DEFAULTS)
    p.update(params)
    key = "__readset_%d.bam" % id(readset)
    _registry["bam"][key] = readset
    del _log[:]
    try:
        df = _findflanks(key, p["min_softclip_length"], p["min_softclip_count"], p["min_alignment_quality"],
                         p["min_alignment_inner_length"], p["min_distance_to_mate"], p["min_softclip_ratio"],
                         p["max_indel_ratio"], p["large_insertion_cutoff"], p["min_count_consensus"], sample_id,
                         output_file)
    finally:
        del _registry["bam"][key]
    return df, df.to_csv(sep="\t", index=False), _stats_from_log()


def run_pairflanks(flanks_tsv_path, readset, genome, ir_finder, output_file, **params):
    """Run the reference's _pairflanks.  ``ir_finder(seq) -> [(a1s, a1e, a2s, a2e), ...]`` (1-based inclusive arms in
    report order) stands in for the einverted binary.  Returns (DataFrame, TSV text)."""
    install_stubs()
    import numpy as np
    import pandas as pd
    from mustache import pairflanks as pf, embosstools, fastatools
    p = dict(PAIRFLANKS_DEFAULTS)
    p.update(params)
    bkey, gkey = "__readset_%d.bam" % id(readset), "__genome_%d.fa" % id(genome)
    _registry["bam"][bkey], _registry["genome"][gkey] = readset, genome
    state = {}

    def write_sequences_to_fasta(sequences, fasta, names=None):
        state["seq"] = sequences[0]

    def run_einverted(fasta, **kw):
        state["irs"] = ir_finder(state["seq"])

    def read_emboss_seq_results(path):
        s = state["seq"]
        for a1s, a1e, a2s, a2e in state["irs"]:
            yield [pd.Series({"ir_pos_5p": a1s, "ir_pos_3p": a1e, "seq": s[a1s - 1:a1e]}),
                   pd.Series({"ir_pos_5p": a2s, "ir_pos_3p": a2e, "seq": s[a2s - 1:a2e]})]

    orig = (fastatools.write_sequences_to_fasta, embosstools.run_einverted, embosstools.read_emboss_seq_results,
            pf.FlankPairer.assign_pairs)
    fastatools.write_sequences_to_fasta = write_sequences_to_fasta
    embosstools.run_einverted = run_einverted
    embosstools.read_emboss_seq_results = read_emboss_seq_results

    def assign_pairs(self, pairs):
        # pandas 0.23 (the reference's pin) reindexes on .loc[:, missing_labels]; newer pandas raises KeyError
        # (pairflanks.py:247-248).  Pre-seeding the four late columns reproduces the 0.23 behaviour.
        for col in ("pair_id", "spanning_count", "direct_repeat_reference", "direct_repeat_reads_consensus"):
            pairs[col] = np.nan
        return orig[3](self, pairs)
    pf.FlankPairer.assign_pairs = assign_pairs
    del _log[:]
    try:
        df = pf._pairflanks(flanks_tsv_path, bkey, gkey, p["max_direct_repeat_length"], p["min_alignment_quality"],
                            p["min_alignment_inner_length"], p["max_junction_spanning_prop"],
                            p["large_insertion_cutoff"], output_file)
    finally:
        (fastatools.write_sequences_to_fasta, embosstools.run_einverted, embosstools.read_emboss_seq_results,
         pf.FlankPairer.assign_pairs) = orig
        del _registry["bam"][bkey], _registry["genome"][gkey]
    return df, open(output_file).read()


def run_recall(pairs_tsv_path, readset, output_file, min_alignment_quality=20, min_alignment_inner_length=21,
               large_insertion_cutoff=30):
    """Run the reference's _recall (recall.py:15-28) on a ReadSet.  Returns (DataFrame, TSV text)."""
    install_stubs()
    from mustache.recall import _recall
    key = "__readset_%d.bam" % id(readset)
    _registry["bam"][key] = readset
    del _log[:]
    try:
        df = _recall(pairs_tsv_path, key, min_alignment_quality, min_alignment_inner_length, large_insertion_cutoff,
                     output_file)
    finally:
        del _registry["bam"][key]
    return df, open(output_file).read()


def run_inferseq_overlap(pairs_tsv_path, output_file, min_overlap_score=10, min_overlap_perc_identity=0.9,
                         min_inferseq_size=30):
    """Run the reference's _inferseq_overlap (inferseqoverlap.py:21-42).  Returns the TSV text it wrote."""
    install_stubs()
    from mustache.inferseqoverlap import _inferseq_overlap
    del _log[:]
    try:
        _inferseq_overlap(pairs_tsv_path, min_overlap_score, min_overlap_perc_identity, min_inferseq_size, output_file)
    except SystemExit:      # handle_empty_pairsfile writes the header-only table and exits
        pass
    return open(output_file).read()


def reference_overlap_functions():
    install_stubs()
    from mustache import inferseqoverlap as m
    return m.get_best_sliding_alignment, m.find_overlap


def reference_modules():
    install_stubs()
    from mustache import sctools, misc, flanktrie, pysamtools
    return sctools, misc, flanktrie, pysamtools


def run_inferseq(pairs, genome, reads, min_perc_identity=0.95, max_internal_softclip_prop=0.05, max_inferseq_size=200000,
                 min_inferseq_size=30, method_name="inferred_reference", shuffle_seed=None):
    """Run the reference's InferSequence (inferseq.py:16-183) on given flank alignments: everything infer_sequences
    (50-70) does AFTER the external aligner has run.  ``reads`` stand in for the aligner's BAM (bowtie2 is not in this
    image); the calls below are the body of infer_sequences from line 56 on, private names spelled out.
    Returns the TSV text of the frame make_dataframe builds.

    ``shuffle_seed``: the reference keeps its candidate pairs in ``set``s of objects that hash by address, so where two
    candidates tie (get_smallest_window_pair, inferseq.py:355-362) its result depends on memory layout.  With a seed the
    pair objects get random hashes for this run (a class attribute set at run time, nothing in the reference's files is
    touched): a result that changes with the seed is order dependent and no use as a golden vector."""
    install_stubs()
    import random
    from mustache import inferseq as _m
    from mustache.inferseq import InferSequence
    if shuffle_seed is not None:
        rnd, cache = random.Random(shuffle_seed), {}
        _m.AlignedPair.__hash__ = lambda self: cache.setdefault(id(self), rnd.getrandbits(60))
    gkey = "__genome_%d.fa" % id(genome)
    _registry["genome"][gkey] = genome
    try:
        inf = InferSequence(pairs, gkey, min_perc_identity, max_internal_softclip_prop, max_inferseq_size,
                            min_inferseq_size, True, method_name, "/tmp")
        inf.bam = list(reads)
        inf._InferSequence__prefilter_reads()
        for read in inf.bam:
            name = read.query_name.split("_")[0]
            inf.all_aligned_pairs[name].add_read(read)
        inf._InferSequence__match_forward_reverse_reads()
        inf._InferSequence__filter_smallest_overlapping()
        inf._InferSequence__filter_pairs_max_internal_softclip_prop()
        inf._InferSequence__filter_best_alignments()
        inf._InferSequence__filter_pairs_size()
        df = inf.make_dataframe()
    finally:
        del _registry["genome"][gkey]
        if shuffle_seed is not None:
            _m.AlignedPair.__hash__ = object.__hash__
    return df.to_csv(sep="\t", index=False)


def run_inferseq_database(database, reads, min_perc_identity=0.95, max_internal_softclip_prop=0.05, max_edge_distance=10):
    """Run the reference's infer_sequences_database (inferseqdatabase.py:116-139) on given flank alignments -- what
    _inferseq_database does AFTER the external aligner -- followed by the table steps of _inferseq_database lines 56-64
    (make_dataframe, sort_values, pair_id as text, inferred_seq_length > 0), spelled out here because the function itself
    starts with bowtie2.  ``reads`` stand in for the aligner's BAM.  Returns the TSV text (without the sample column)."""
    install_stubs()
    import contextlib
    import io
    from mustache import inferseqdatabase as m
    key = "__aln_%d.bam" % id(reads)
    _registry["bam"][key] = list(reads)
    try:
        with contextlib.redirect_stdout(io.StringIO()):            # "TOTAL PAIRS AFTER FILTER ..." prints
            seqs = m.infer_sequences_database(key, database, min_perc_identity, max_internal_softclip_prop, max_edge_distance)
    finally:
        del _registry["bam"][key]
    df = m.make_dataframe(seqs, method="inferred_database")
    df = df.sort_values(by=["pair_id", "method"])
    df.loc[:, "pair_id"] = list(map(str, map(int, list(df["pair_id"]))))
    df = df.query("inferred_seq_length > 0")
    return df.to_csv(sep="\t", index=False)
