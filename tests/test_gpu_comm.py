"""mfk_allgather_sites through the C-ABI: a communicator of one rank on the one GPU of the test box (RCCL does not take two
ranks on one device), so this covers loading RCCL at run time, the communicator, the staging and the one-collective /
second-exchange protocol; the merge logic on top of it for two ranks runs on gloo in tests/test_dist.py."""
import numpy as np
import pytest

from mustache_amd import lib, dist as mdist


@pytest.mark.gpu
def test_allgather_one_rank_roundtrip():
    hip = lib.load(0)
    hip.init_comm(hip.make_comm_id(), 0, 1)
    rng = np.random.default_rng(5)
    # the third does not fit its padding: second exchange; the last three let the library size the exchange after the previous one
    for n, nb, pad in [(0, 0, 0), (7, 300, 0), (1000, 70000, 4096), (3, 10, 4096), (5000, 100, 0), (5, 100, 0), (9000, 5, 0)]:
        rows = np.zeros(n, dtype=lib.FLANK_DTYPE)
        rows["pos"] = rng.integers(0, 1 << 30, n)
        rows["seq_off"] = np.arange(n)
        blob = rng.integers(0, 256, nb, dtype=np.uint8).tobytes()
        parts = hip.run_allgather(rows, blob, pad)
        assert len(parts) == 1
        assert parts[0][0].tobytes() == rows.tobytes() and parts[0][1] == blob
    # the helpers of the sharded path take the library route when the communicator is up
    rows = np.zeros(4, dtype=lib.FLANK_DTYPE)
    rows["seq_off"] = [0, 2, 4, 6]
    got_rows, got_seqs = mdist.allgather_flanks(rows, b"AACCGGTT", binding=hip)
    assert got_rows.tobytes() == rows.tobytes() and got_seqs == b"AACCGGTT"
    hip.free_comm()       # (lib.load hands out one binding per device: it stays open for the other tests)


@pytest.mark.gpu
def test_init_library_comm_under_torch_distributed():
    """The glue bench.py uses at N > 1: id from rank 0, broadcast by torch.distributed, communicator joined (world of one)."""
    import torch.distributed as dist
    hip = lib.load(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        mdist.init_library_comm(hip)
        assert hip.comm_world == 1
        rows = np.zeros(2, dtype=lib.FLANK_DTYPE)
        rows["seq_off"] = [0, 3]
        got_rows, got_seqs = mdist.allgather_flanks(rows, b"ACGTTT", binding=hip)
        assert got_rows.tobytes() == rows.tobytes() and got_seqs == b"ACGTTT"
    finally:
        dist.destroy_process_group()
        hip.free_comm()


@pytest.mark.gpu
def test_new_communicator_starts_from_the_first_size():
    """The size of a pad_bytes = 0 exchange follows the previous gather of the SAME communicator only: a ctx that joins
    another communicator must start from 16 KB again, or ranks with different histories call ncclAllGather with
    different counts."""
    hip = lib.load(0)
    hip.init_comm(hip.make_comm_id(), 0, 1)
    try:
        assert hip.allgather_next_bytes(hip.ctx) == 16 << 10
        rows = np.zeros(9000, dtype=lib.FLANK_DTYPE)
        rows["pos"] = np.arange(9000)
        parts = hip.run_allgather(rows, b"x" * 100, 0)
        assert parts[0][0].tobytes() == rows.tobytes()
        big = hip.allgather_next_bytes(hip.ctx)
        assert big >= 9000 * lib.FLANK_DTYPE.itemsize          # the hint follows the large table ...
        hip.init_comm(hip.make_comm_id(), 0, 1)                # ... and a new communicator forgets it
        assert hip.allgather_next_bytes(hip.ctx) == 16 << 10
        small = np.zeros(3, dtype=lib.FLANK_DTYPE)
        parts = hip.run_allgather(small, b"abc", 0)
        assert parts[0][0].tobytes() == small.tobytes() and parts[0][1] == b"abc"
        hip.free_comm()
        assert hip.allgather_next_bytes(hip.ctx) == 16 << 10
    finally:
        hip.free_comm()
