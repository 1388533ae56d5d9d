"""mfk_recall_noalloc (the entry point that allocates nothing, include/mfk.h) against mfk_recall, through the C-ABI:
bit-exact counts on two calls in a row, the second with a smaller input; limits rejected with a clear error.
Named to be collected last: the round's end runs ``pytest -x``."""
import numpy as np
import pytest

from mustache_amd import lib
from mustache_amd.findflanks import device_reads
from mustache_amd.synth import CONFIGS, synth_packed

pytestmark = pytest.mark.gpu


def _sites(packed, truth, rng, n_random):
    cid, pos = [], []
    for name, p5, p3 in truth:
        c = packed.contig_names.index(name)
        for d in (-2, -1, 0, 1, 2):
            cid += [c, c]; pos += [p5 + d, p3 + d]
    for c, clen in enumerate(packed.contig_len):
        cid += [c, c, c]; pos += [-1, 0, int(clen)]
        r = rng.integers(0, int(clen), n_random)
        cid += [c] * n_random; pos += [int(x) for x in r]
    return np.asarray(cid, np.int32), np.asarray(pos, np.int32)


def test_noalloc_recall_equals_recall_twice(hip, oracle):
    packed, genome, truth = synth_packed(contig_lens=(300_000, 40_000, 9_000), coverage=60, n_insertions=12, seed=21)
    rd = device_reads(hip, packed, "cuda:0")
    params = lib.make_params()
    rng = np.random.default_rng(5)
    cid, pos = _sites(packed, truth, rng, 700)
    assert 2000 < cid.shape[0] <= 16384
    want = hip.run_recall(rd, params, cid, pos)
    got = hip.run_recall_noalloc(rd, params, cid, pos)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert int(want.sum()) > 0
    # the oracle agrees too (same fixture as the recall parity test, smaller)
    ro = oracle.reads(packed)
    assert np.array_equal(got[:300], oracle.run_recall(ro, params, cid[:300], pos[:300]))
    # second call on the same GPU: fewer sites, other sites, stricter filters; the array still holds the first call's bytes
    keep = rng.permutation(cid.shape[0])[:137]
    p2 = lib.make_params(min_alignment_quality=30, large_insertion_cutoff=2)
    want2 = hip.run_recall(rd, p2, cid[keep], pos[keep])
    got2 = hip.run_recall_noalloc(rd, p2, cid[keep], pos[keep])
    assert np.array_equal(got2, want2)
    # and the ordinary entry point is untouched by it
    assert np.array_equal(hip.run_recall(rd, params, cid, pos), want)


def test_noalloc_recall_rejects_what_does_not_fit(hip):
    packed, _, _ = synth_packed(**CONFIGS["tiny"], seed=3)
    rd = device_reads(hip, packed, "cuda:0")
    params = lib.make_params()
    n = 16385
    with pytest.raises(lib.MfkError, match="at most 16384 sites"):
        hip.run_recall_noalloc(rd, params, np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
    with pytest.raises(lib.MfkError, match="outside contig"):
        hip.run_recall_noalloc(rd, params, np.zeros(1, np.int32), np.asarray([int(packed.contig_len[0]) + 5], np.int32))
    out = hip.run_recall_noalloc(rd, params, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert out.shape == (0, 10)
