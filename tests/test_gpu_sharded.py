"""Coordinate- and contig-sharded runs through the HIP library on ONE GPU: the K ranks of a job run one after the other
(`dist.run_sharded_sequential`: the per-rank stages and merges bench.py --sharding uses under RCCL), and both merged
tables must be the bytes of the unsharded HIP run -- for balanced cuts and for cuts placed on, next to, and a halo away
from insertion sites."""
import numpy as np
import pytest

from mustache_amd import dist as mdist
from mustache_amd import lib, pairflanks as pfh
from mustache_amd.findflanks import device_reads
from mustache_amd.pack import pack_reads
from mustache_amd.synth import CONFIGS, synth_packed
from tests import util

pytestmark = pytest.mark.gpu


def _inputs(name):
    if name == "sim_b":
        case = util.load_json("sim_b")
        rs = util.case_readset(case)
        packed = pack_reads(rs.contigs, rs, pool="all")
        genome = case["genome"]
        params = lib.make_params(**{**case["findflanks_params"], **case["pairflanks_params"]})
    else:
        packed, genome, _ = synth_packed(**CONFIGS[name], seed=4)
        params = lib.make_params()
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    return packed, params, g, goff, pfh.contig_ranks(packed.contig_names)


@pytest.fixture(scope="module", params=["sim_b", "C3_klebsiella", "C5_tenth"])
def workload(request, hip):
    import torch
    packed, params, g, goff, ranks = _inputs(request.param)
    gdev = torch.from_numpy(g).to("cuda:0")
    rd = device_reads(hip, packed, "cuda:0")
    ff = hip.run_findflanks(rd, params)
    pf = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, gdev, goff)
    assert len(ff["flanks"]) > 0 and len(pf["pairs"]) > 0
    return request.param, packed, params, gdev, goff, ranks, ff, mdist._compact(pf["pairs"], pf["strs"], mdist._PAIR_STR)


def _check(hip, workload, world, sharding, plans=None):
    name, packed, params, gdev, goff, ranks, ff, (wp, ws) = workload
    (tab, seqs), (pairs, strs), plans = mdist.run_sharded_sequential(
        hip, lambda p: device_reads(hip, p, "cuda:0"), packed, params, world, sharding, ranks, gdev, goff, plans=plans)
    assert tab.tobytes() == ff["flanks"].tobytes() and seqs == ff["seqs"]
    assert pairs.tobytes() == wp.tobytes() and strs == ws
    return plans


@pytest.mark.parametrize("world", [2, 3, 8])
def test_coordinate_shards_through_hip(hip, workload, world):
    plans = _check(hip, workload, world, "coordinate")
    n = workload[1].n_reads
    if workload[0] != "sim_b":
        assert max(p.n_reads() for p in plans) < 1.3 * n / world + 2000        # balanced by reads


@pytest.mark.parametrize("world", [2, 8])
def test_contig_shards_through_hip(hip, workload, world):
    plans = _check(hip, workload, world, "contig")
    assert sum(p.n_reads() for p in plans) == workload[1].n_reads


def test_cuts_on_and_around_sites(hip, workload):
    """Cuts at a flank's position, one off, and a halo's width (69) to either side of it."""
    name, packed, params = workload[:3]
    fl = workload[6]["flanks"]
    h = mdist.halo(params.min_distance_to_mate, params.max_direct_repeat_length)
    first = fl[fl["contig_id"] == fl["contig_id"][0]]
    pos = np.unique(first["pos"])
    picks = pos[np.linspace(0, len(pos) - 1, 7).astype(int)[1:-1]]
    c = int(first["contig_id"][0])
    for deltas in ((0, 1, -1, h, -h), (-h + 1, h - 1, 2, -2, 10)):
        cuts = sorted({(c, int(p) + d) for p, d in zip(picks, deltas)})
        plans = mdist.plan_shards(packed, len(cuts) + 1, "coordinate", h, cuts=cuts)
        _check(hip, workload, len(cuts) + 1, "coordinate", plans=plans)
