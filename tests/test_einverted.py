"""--ir-backend=einverted (SURVEY 8c escape hatch).  EMBOSS is not in the image, so the executable is replaced by a
detector function; everything behind it -- candidate enumeration, the reference's acceptance loop, the table handed to
the library, the library using it instead of its own detector -- is exercised for real.  The golden pairflanks TSVs
were produced by the reference with the oracle's ir_report standing in for einverted, so feeding ir_report through this
path must reproduce them byte for byte."""
import io

import numpy as np
import pandas as pd
import pytest
from click.testing import CliRunner

from mustache_amd import einverted, lib
from mustache_amd import pairflanks as pfh
from mustache_amd.pack import pack_reads
from tests import util


def test_outseq_parser_and_acceptance():
    text = ">0_3_20 0\nACGTACGTACGTACGTAC\n>0_61_78 0\nGTACGTACGTACGTACGT\n>0_30_35 0\nAAAAAA\n>0_40_45 0\nTTTTTT\n"
    reps = einverted.parse_outseq(text)
    assert reps == [(3, 20, 61, 78), (30, 35, 40, 45)]
    assert einverted.accept(reps, 90) == (1, 3, 20, 61, 78)             # 78 >= 90 - 15, 3 <= 15
    assert einverted.accept(reps, 100) == (0, 0, 0, 0, 0)               # arm 2 ends too far from the end
    assert einverted.accept([(1, 9, 80, 88), (2, 8, 82, 89)], 100) == (1, 2, 8, 82, 89)   # the LAST accepted repeat wins
    assert einverted.combined_sequence("A" * 50, "C" * 45) == "A" * 40 + "N" * 20 + "C" * 40
    assert einverted.combined_sequence("AAA", "CC") == "AAA" + "N" * 20 + "CC"


def test_missing_executable_is_a_clear_error(tmp_path):
    with pytest.raises(einverted.EinvertedNotFound):
        einverted.find_einverted("einverted-is-not-installed-here")
    from mustache_amd.main import cli
    f = tmp_path / "x.tsv"
    f.write_text("a\n")
    if einverted.shutil.which("einverted") is None:
        r = CliRunner().invoke(cli, ["pairflanks", str(f), str(f), str(f), "--ir-backend", "einverted"])
        assert r.exit_code != 0 and isinstance(r.exception, einverted.EinvertedNotFound)


def _run(binding, reads_struct, packed, case, genome, **kw):
    flanks = pd.read_csv(io.StringIO(case["findflanks_tsv"]), sep="\t")
    g, goff = pfh.genome_arrays(case["genome"], packed.contig_names)
    params = lib.make_params(**case["pairflanks_params"])
    return pfh.pairflanks_packed(binding, reads_struct, packed.contig_names, params, flanks, genome(g), goff, None, False, **kw)


@pytest.mark.parametrize("name", util.CASES)
def test_external_ir_table_through_the_oracle(oracle, name):
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    R = oracle.reads(packed)
    df = _run(oracle, R, packed, case, lambda g: g, ir_backend="einverted", ir_detector=oracle.ir_report)
    assert df.to_csv(sep="\t", index=False) == case["pairflanks_tsv"]
    # a detector that never finds anything: no IR columns, pairing falls back to counts / lengths
    df0 = _run(oracle, R, packed, case, lambda g: g, ir_backend="einverted", ir_detector=lambda s: [])
    assert not df0["has_IR"].any() if len(df0) else True
    # an incomplete table is an error, and the table does not leak into the next call
    flanks = pd.read_csv(io.StringIO(case["findflanks_tsv"]), sep="\t")
    tab, seqs = pfh.flanks_to_table(flanks, packed.contig_names)
    cands = einverted.candidate_pairs(tab, 20)
    if len(cands) > 1:
        i5, i3, ir = einverted.ir_table(tab, seqs, 20, detector=oracle.ir_report)
        oracle.set_pair_ir(i5[1:], i3[1:], ir[1:])
        with pytest.raises(lib.MfkError):
            _run(oracle, R, packed, case, lambda g: g)
        assert _run(oracle, R, packed, case, lambda g: g).to_csv(sep="\t", index=False) == case["pairflanks_tsv"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", util.CASES)
def test_external_ir_table_through_the_hip_library(hip, oracle, name):
    import torch
    from mustache_amd.findflanks import device_reads
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    rsd = device_reads(hip, packed, "cuda:0")
    dev = lambda g: torch.from_numpy(g).to("cuda:0")   # noqa: E731
    df = _run(hip, rsd, packed, case, dev, ir_backend="einverted", ir_detector=oracle.ir_report)
    assert df.to_csv(sep="\t", index=False) == case["pairflanks_tsv"]
    none = lambda s: []   # noqa: E731
    got = _run(hip, rsd, packed, case, dev, ir_backend="einverted", ir_detector=none)
    want = _run(oracle, oracle.reads(packed), packed, case, lambda g: g, ir_backend="einverted", ir_detector=none)
    assert got.to_csv(sep="\t", index=False) == want.to_csv(sep="\t", index=False)
    flanks = pd.read_csv(io.StringIO(case["findflanks_tsv"]), sep="\t")
    tab, seqs = pfh.flanks_to_table(flanks, packed.contig_names)
    if len(einverted.candidate_pairs(tab, 20)) > 1:
        i5, i3, ir = einverted.ir_table(tab, seqs, 20, detector=oracle.ir_report)
        hip.set_pair_ir(i5[:-1], i3[:-1], ir[:-1])
        with pytest.raises(lib.MfkError):
            _run(hip, rsd, packed, case, dev)
        assert _run(hip, rsd, packed, case, dev).to_csv(sep="\t", index=False) == case["pairflanks_tsv"]
