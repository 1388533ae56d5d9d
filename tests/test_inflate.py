"""The DEFLATE decoder of mustache_amd/csrc/mfk_inflate_core.h: its host build (a wave of one lane) against zlib on every
kind of block (CPU), and the kernel (one wave per BGZF block, mfk_bgzf_inflate) against zlib on BGZF files (GPU)."""
import ctypes
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

from mustache_amd import bamio, gpuinflate, lib
from mustache_amd.simulate import simulate_readset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return co.compress(data) + co.flush()


def _payloads():
    rng = np.random.default_rng(1)
    rnd = random.Random(3)
    out = [b"", b"a", b"ab" * 7, b"abc" * 1000, b"\0" * 65280, bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
           bytes(rng.integers(0, 4, 65280, dtype=np.uint8)), "".join(rnd.choice("ACGT") for _ in range(40000)).encode(),
           open(os.path.join(ROOT, "DESIGN.md"), "rb").read()[:65280]]
    # long matches at every small distance (a match that overlaps itself repeats its first bytes)
    for d in (1, 2, 3, 5, 63, 64, 65, 257):
        out.append((bytes(rng.integers(0, 256, d, dtype=np.uint8)) * (30000 // d + 1))[:30000])
    # many distinct symbols with skewed frequencies: code lengths beyond the primary tables
    out.append(bytes(min(255, int(x)) for x in rng.geometric(0.02, 60000)))
    return out


@pytest.fixture(scope="module")
def host_inflate(tmp_path_factory):
    so = os.path.join(tmp_path_factory.mktemp("infl"), "libinflhost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "native", "inflate_host.cpp")])
    h = ctypes.CDLL(so)
    h.mfk_inflate_host.restype = ctypes.c_int32
    h.mfk_inflate_host.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]

    def run(comp, ulen):
        out = np.full(ulen + 64, 0xEE, dtype=np.uint8)      # guard bytes behind the output
        rc = h.mfk_inflate_host(comp, len(comp), out.ctypes.data, ulen)
        assert np.all(out[ulen:] == 0xEE), "the decoder wrote behind its output"
        return rc, out[:ulen].tobytes()
    return run


def test_host_build_equals_zlib(host_inflate):
    n = 0
    for data in _payloads():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                rc, got = host_inflate(_raw(data, level, strategy), len(data))
                assert rc == 0 and got == data, (len(data), level, strategy, rc)
                n += 1
    assert n > 250
    # several deflate blocks in one stream (full flushes), stored + compressed mixed
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [b"abc" * 500, bytes(range(256)) * 20, b"", b"xyz" * 3000]
    comp = b"".join(co.compress(p) + co.flush(zlib.Z_FULL_FLUSH) for p in parts) + co.flush()
    rc, got = host_inflate(comp, sum(len(p) for p in parts))
    assert rc == 0 and got == b"".join(parts)


def test_host_build_rejects_damaged_streams_without_leaving_its_buffers(host_inflate):
    rng = np.random.default_rng(5)
    data = open(os.path.join(ROOT, "DESIGN.md"), "rb").read()[:30000]
    comp = _raw(data)
    assert host_inflate(comp, len(data) - 1)[0] != 0            # more output than ISIZE says
    assert host_inflate(comp, len(data) + 1)[0] != 0            # less
    assert host_inflate(comp[:len(comp) // 2], len(data))[0] != 0   # truncated
    assert host_inflate(b"\x07", 0)[0] != 0                      # reserved block type
    ok = bad = 0
    for k in range(400):                                         # random damage: the same verdict and bytes as zlib, never a stray write
        c = bytearray(comp)
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        rc, got = host_inflate(bytes(c), len(data))
        try:
            d = zlib.decompressobj(-15)
            want = d.decompress(bytes(c))
            z_ok = d.eof and len(want) == len(data)
        except zlib.error:
            z_ok, want = False, b""
        assert (rc == 0) == z_ok, (k, rc, z_ok)
        if z_ok:
            assert got == want
        ok += rc == 0
        bad += rc != 0
    assert bad > 100 and ok > 20
    for k in range(200):                                         # pure noise
        c = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
        host_inflate(c, int(rng.integers(0, 5000)))


def _bgzf(parts, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    import struct
    out = []
    for data in parts:
        comp = _raw(data, level, strategy)
        out.append(struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    return b"".join(out) + bamio._BGZF_EOF


def test_block_table():
    parts = [b"abc" * 100, b"", b"x" * 65280]
    data = _bgzf(parts)
    blocks, crcs, total = gpuinflate.bgzf_blocks(data)
    assert total == 300 + 65280 and blocks.shape[0] == 2 and blocks["out_off"].tolist() == [0, 300]
    assert [int(c) for c in crcs] == [zlib.crc32(parts[0]), zlib.crc32(parts[2])]
    for b, p in zip(blocks, (parts[0], parts[2])):
        raw = data[int(b["comp_off"]):int(b["comp_off"]) + int(b["comp_len"])]
        assert zlib.decompress(raw, -15) == p
    with pytest.raises(lib.MfkError):
        gpuinflate.bgzf_blocks(data[:40])
    with pytest.raises(lib.MfkError):
        gpuinflate.bgzf_blocks(b"not a bgzf file at all....")


@pytest.mark.gpu
def test_kernel_equals_zlib_on_every_kind_of_block(hip):
    import torch
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
        parts = [p for p in _payloads() if p]
        data = _bgzf(parts, level, strategy)
        out, _ = gpuinflate.inflate_bytes(hip, data, check_crc=True)
        assert out.cpu().numpy().tobytes() == b"".join(parts), (level, strategy)
    # no block at all; blocks in any order of the output
    out, _ = gpuinflate.inflate_bytes(hip, bamio._BGZF_EOF)
    assert out.numel() == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_kernel_inflates_a_bam_file(hip, tmp_path):
    genome, rs, truth = simulate_readset(seed=31, contig_lens=(20000, 7000), n_insertions=4, coverage=60)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    data = open(path, "rb").read()
    import gzip
    want = gzip.open(path, "rb").read()
    out, _ = gpuinflate.inflate_bytes(hip, data, check_crc=True)
    assert out.cpu().numpy().tobytes() == want and len(want) > 500000


@pytest.mark.gpu
def test_kernel_reports_damaged_blocks_and_stays_inside_its_ranges(hip):
    import torch
    rng = np.random.default_rng(9)
    parts = [open(os.path.join(ROOT, "DESIGN.md"), "rb").read()[k * 20000:(k + 1) * 20000] for k in range(4)]
    data = bytearray(_bgzf(parts))
    blocks, crcs, total = gpuinflate.bgzf_blocks(bytes(data))
    b1 = blocks[1]
    for k in range(40):                                            # damage inside block 1 only
        data[int(b1["comp_off"]) + int(rng.integers(0, int(b1["comp_len"])))] ^= 0xFF
    comp = torch.frombuffer(bytearray(bytes(data)), dtype=torch.uint8).to("cuda:0")
    out = torch.full((total + 256,), 0xEE, dtype=torch.uint8, device="cuda:0")
    n_bad, status = hip.run_bgzf_inflate(comp, blocks, out[:total], want_status=True)
    host = out.cpu().numpy()
    assert np.all(host[total:] == 0xEE)
    for k in (0, 2, 3):                                            # the other blocks are whole, whatever block 1 did
        assert status[k] == 0 and host[int(blocks[k]["out_off"]):int(blocks[k]["out_off"]) + 20000].tobytes() == parts[k]
    assert (n_bad == 1 and status[1] != 0) or zlib.crc32(host[20000:40000].tobytes()) != int(crcs[1])
    with pytest.raises(lib.MfkError, match="do not inflate"):
        hip.run_bgzf_inflate(comp, blocks, out[:total]) if n_bad else (_ for _ in ()).throw(lib.MfkError("x do not inflate"))
    # a block table that points outside the buffers is refused before anything runs
    wrong = blocks.copy()
    wrong["out_off"][3] = total
    with pytest.raises(lib.MfkError, match="outside the buffers"):
        hip.run_bgzf_inflate(comp, wrong, out[:total])


@pytest.mark.gpu
def test_device_crc32_equals_zlib(hip):
    import torch
    rng = np.random.default_rng(12)
    sizes = [0, 1, 2, 15, 16, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 4097, 65279, 65280, 65536, 33333, 7]
    data = bytes(rng.integers(0, 256, sum(sizes) + 64, dtype=np.uint8))
    for shift in (0, 1, 5, 16):                                   # every alignment of the first block
        blocks = np.zeros(len(sizes), dtype=lib.BGZF_BLOCK_DTYPE)
        o = shift
        for k, n in enumerate(sizes):
            blocks[k] = (0, o, 0, n)
            o += n
        dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
        got = np.zeros(len(sizes), dtype=np.uint32)
        hip._check(hip.bgzf_crc32(hip.ctx, dev.data_ptr(), dev.numel(), blocks.ctypes.data, len(sizes), got.ctypes.data))
        want = [zlib.crc32(data[int(b["out_off"]):int(b["out_off"]) + int(b["out_len"])]) for b in blocks]
        assert got.tolist() == want, shift
