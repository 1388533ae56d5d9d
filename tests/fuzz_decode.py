"""Parity fuzz of the device decoder (test infrastructure, not collected by pytest): simulated read sets written as BAM, the
inflated stream then cut into BGZF blocks of RANDOM sizes (so that blocks start anywhere: inside names, CIGARs, qualities, tags)
and at several deflate levels; every array of gpudecode.decode_bam against pack.pack_bam.

    python tests/fuzz_decode.py [--seeds 60] [--first 100] [--budget-s 240]
"""
import argparse
import gzip
import os
import struct
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reblock(stream, rng, level, lo, hi):
    from mustache_amd import bamio
    out, o = [], 0
    while o < len(stream):
        n = int(rng.integers(lo, hi + 1))
        data = stream[o:o + n]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        if len(comp) + 26 > 65536:          # (incompressible and large: store it in two blocks instead)
            n = n // 2
            continue
        out.append(struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        o += n
    return b"".join(out) + bamio._BGZF_EOF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--budget-s", type=float, default=240.0)
    a = ap.parse_args()
    import numpy as np
    from mustache_amd import bamio, gpudecode, lib
    from mustache_amd.pack import ALL_FIELDS, pack_bam
    from mustache_amd.simulate import simulate_readset
    hip = lib.load(0)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "f.bam")
    t0 = time.time()
    done = bad = fallback = 0
    for seed in range(a.first, a.first + a.seeds):
        if time.time() - t0 > a.budget_s:
            break
        rng = np.random.default_rng(seed)
        lens = tuple(int(x) for x in rng.integers(300, 40000, int(rng.integers(1, 5))))
        genome, rs, truth = simulate_readset(seed=seed, contig_lens=lens, n_insertions=int(rng.integers(0, 5)), coverage=int(rng.integers(5, 120)),
                                             read_len=int(rng.choice([50, 100, 150, 250])), sub_rate=float(rng.choice([0.0, 0.01, 0.05])),
                                             wildtype_frac=float(rng.choice([0.0, 0.3])))
        bamio.write_bam(path, rs)
        stream = gzip.open(path, "rb").read()
        lo, hi = [(40, 400), (200, 5000), (1000, 65280), (60000, 65280)][int(rng.integers(0, 4))]
        open(path, "wb").write(reblock(stream, rng, int(rng.choice([0, 1, 6, 9])), lo, hi))
        pool = str(rng.choice(["clipped", "all", "none"]))
        flags = int(rng.choice([0, 0x900, 0x10]))
        want = pack_bam(path, pool=pool, exclude_flags=flags)
        try:
            got = gpudecode.decode_bam(hip, path, pool=pool, exclude_flags=flags).to_host()
        except gpudecode.GpuDecodeUnverified:
            fallback += 1
            done += 1
            continue
        same = got.n_reads == want.n_reads and got.max_span == want.max_span and all(np.array_equal(got.a[k], want.a[k]) for k, _ in ALL_FIELDS)
        bad += not same
        done += 1
        if done % 10 == 0:
            print("%d cases, %d mismatches, %d unverified, %.0f s (last: %d reads, blocks of %d..%d bytes)" % (done, bad, fallback, time.time() - t0, want.n_reads, lo, hi), flush=True)
    print("fuzz decode: %d cases, %d mismatches, %d sent to the host reader" % (done, bad, fallback))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
