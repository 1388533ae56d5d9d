"""Shared helpers for the tests: golden fixtures -> alignments, result comparison."""
import gzip
import io
import json
import os

import numpy as np
import pandas as pd

from mustache_amd import lib
from mustache_amd import pairflanks as pfh
from mustache_amd.findflanks import findflanks_packed
from mustache_amd.pack import pack_reads
from mustache_amd.records import AlignedRead, ReadSet, parse_cigar_string

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["quirks", "sim_a", "sim_b", "sim_c", "sim_d"]
# made by tests/golden/make_golden_deep.py (the reference itself, three hash seeds): hundreds of leaves and several clusters per
# site-side, N / IUPAC clip bases, 250-bp reads with clips past 200 bases, two surviving consensus sequences per side
DEEP_CASES = ["deep_clusters", "deep_iupac", "deep_long", "deep_two"]
ALL_CASES = CASES + DEEP_CASES


def load_json(name):
    with gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rt") as fh:
        return json.load(fh)


def read_from_json(row):
    name, contig, start, mapq, cigar, md, seq, qual, flag = row
    return AlignedRead(name, contig, start, mapq, parse_cigar_string(cigar), seq, [ord(c) - 33 for c in qual], md=md,
                       flag=flag)


def case_readset(case):
    return ReadSet([tuple(c) for c in case["contigs"]], [read_from_json(r) for r in case["reads"]])


def run_both_stages(binding, reads_struct, packed, case, genome_arr=None):
    """findflanks TSV text, stats, pairflanks TSV text through ``binding`` for a golden case."""
    fparams = lib.make_params(**case["findflanks_params"])
    buf = io.StringIO()
    res = binding.run_findflanks(reads_struct, fparams)
    df = findflanks_packed(binding, reads_struct, packed.contig_names, fparams, "sample1", None, verbose=False)
    ff_text = df.to_csv(sep="\t", index=False)
    flanks = pd.read_csv(io.StringIO(case["findflanks_tsv"]), sep="\t")
    pparams = lib.make_params(**case["pairflanks_params"])
    g, goff = pfh.genome_arrays(case["genome"], packed.contig_names)
    if genome_arr is not None:
        g = genome_arr(g)
    pdf = pfh.pairflanks_packed(binding, reads_struct, packed.contig_names, pparams, flanks, g, goff, None, False)
    del buf
    return ff_text, [int(x) for x in res["stats"]], pdf.to_csv(sep="\t", index=False)


def assert_same_sites(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    for name in a.dtype.names:
        assert np.array_equal(a[name], b[name]), (name, a[name][:20], b[name][:20])


def recall_text(binding, reads_struct, packed, entry):
    """TSV text of ``mustache recall`` through ``binding`` for one entry of recall_vectors.json.gz."""
    from mustache_amd.recall import recall_packed
    pairs = pd.read_csv(io.StringIO(entry["pairs_tsv"]), sep="\t")
    params = lib.make_params(**entry["params"])
    df = recall_packed(binding, reads_struct, packed.contig_names, params, pairs, None, verbose=False)
    return df.to_csv(sep="\t", index=False)


def overlap_text(binding, entry):
    """TSV text of ``mustache inferseq-overlap`` through ``binding`` for one table of overlap_vectors.json.gz."""
    from mustache_amd import inferseqoverlap as iso
    pairs = pd.read_csv(io.StringIO(entry["pairs_tsv"]), sep="\t", keep_default_na=False, na_values=iso._NA_VALUES)
    if pairs.shape[0] == 0:
        return pd.DataFrame(columns=iso.OVERLAP_COLUMNS).to_csv(sep="\t", index=False)
    p = dict(min_overlap_score=10, min_overlap_perc_identity=0.9, min_inferseq_size=30)
    p.update(entry["params"])
    return iso.inferseq_overlap_frame(binding, pairs, **p).to_csv(sep="\t", index=False)


# ---- full-size BASELINE configs whose oracle run is too long for the GPU suite: compared through digests ---------------
FULLSIZE_DIGESTS = os.path.join(GOLDEN, "fullsize_digests.json")


def fullsize_inputs(name):
    """(packed, genome bytes, genome offsets, contig ranks) of a digest case; deterministic in the numpy version's PCG64.
    C5_meta_50mb: BASELINE configs[4] (50 Mb, 200x, 10 000 insertions), seed 7.  C4_batch12: what one GPU holds of
    configs[3] (96 E. coli isolates over 8 GPUs): 12 x C1 in one batch, the seeds bench.py uses on rank 0."""
    from mustache_amd import cohort
    from mustache_amd.synth import CONFIGS, synth_packed
    if name == "C5_meta_50mb":
        packed, genome, _ = synth_packed(**CONFIGS["C5_meta_50mb"], seed=7)
        g, goff = pfh.genome_arrays(genome, packed.contig_names)
        return packed, g, goff, pfh.contig_ranks(packed.contig_names)
    if name == "C4_batch12":
        packs, genomes = [], []
        for i in range(12):
            p_, g_, _ = synth_packed(**CONFIGS["C1_ecoli_100x"], seed=1000 + i)
            packs.append(p_); genomes.append(g_)
        batch = cohort.concat_samples(packs, ["isolate%d" % i for i in range(12)])
        g, goff = cohort.batch_genome(batch, genomes)
        return batch.packed, g, goff, cohort.batch_contig_ranks(batch)
    raise KeyError(name)


def result_digests(ff, pf):
    """sha256 of every byte the two result tables hold (+ the logged counters), and the sizes for a readable failure."""
    import hashlib
    h = lambda b: hashlib.sha256(bytes(b)).hexdigest()
    return {"flanks": h(np.ascontiguousarray(ff["flanks"]).tobytes()), "seqs": h(ff["seqs"]),
            "stats": [int(x) for x in ff["stats"][:12]], "pairs": h(np.ascontiguousarray(pf["pairs"]).tobytes()),
            "strs": h(pf["strs"]), "pair_stats": [int(x) for x in pf["stats"][:5]],
            "n_flanks": int(len(ff["flanks"])), "n_pairs": int(len(pf["pairs"]))}
