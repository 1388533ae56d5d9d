"""BAM -> packed arrays -> (optionally) GPU -> TSV: the end-to-end leg of the path, host decode included.

    python scripts/bench_decode.py --reads 4000000 --out gpurun_out/decode.json [--gpu]

Writes a replicated synthetic BAM (mustache_amd/bigbam.py) of about --reads records under --tmp, then times
  * libmfkio.so (pack_bam) at several thread counts,
  * the Python decoder (BamReader + pack_reads) on a bounded sample, for the ratio,
  * with --gpu: host->device copy of the packed arrays (the PCIe-inclusive leg) and mfk_findflanks on them, and checks
    the flank table against the base read set's table repeated per copy.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--base-coverage", type=int, default=100)
    ap.add_argument("--tmp", default="/tmp")
    ap.add_argument("--out", default=None)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args()

    from mustache_amd import bamio, bigbam, pack
    from mustache_amd.simulate import simulate_readset
    genome, rs, truth = simulate_readset(seed=7, contig_lens=(60000, 8000), n_insertions=8, coverage=a.base_coverage)
    copies = max(1, a.reads // len(rs.reads))
    path = os.path.join(a.tmp, "mfk_decode_bench.bam")
    t0 = time.time()
    contigs, n_rec = bigbam.write_replicated_bam(path, rs, copies, procs=a.procs)
    t_write = time.time() - t0
    res = {"records": n_rec, "copies": copies, "base_records": len(rs.reads), "bam_bytes": os.path.getsize(path),
           "write_s": round(t_write, 2), "native": [], "host_cpus_visible": os.cpu_count()}
    print("wrote %d records, %.1f MB in %.1f s" % (n_rec, res["bam_bytes"] / 1e6, t_write), flush=True)
    packed = None
    for T in [int(x) for x in a.threads.split(",")]:
        t0 = time.time()
        packed = pack.pack_bam(path, pool="clipped", threads=T)
        dt = time.time() - t0
        st = packed.io_stats
        res["native"].append({"threads": T, "seconds": round(dt, 3), "records_per_s": round(n_rec / dt),
                              "inflate_s": round(st["seconds_inflate"], 3), "parse_s": round(st["seconds_parse"], 3),
                              "link_s": round(st["seconds_link"], 3), "libdeflate": bool(st["inflate_backend"]),
                              "uncompressed_MBps": round(st["uncompressed_bytes"] / dt / 1e6)})
        print(res["native"][-1], flush=True)
    res["uncompressed_bytes"] = int(packed.io_stats["uncompressed_bytes"])
    res["packed_bytes"] = int(sum(v.nbytes for v in packed.a.values()))
    # Python decoder on the base set only (one copy), scaled
    one = os.path.join(a.tmp, "mfk_decode_bench_one.bam")
    bigbam.write_replicated_bam(one, rs, 1, procs=1)
    t0 = time.time()
    with bamio.BamReader(one) as bam:
        p1 = pack.pack_reads(bam.contigs, bam, pool="clipped")
    dt = time.time() - t0
    res["python_decoder"] = {"records": p1.n_reads, "seconds": round(dt, 2), "records_per_s": round(p1.n_reads / dt)}
    print(res["python_decoder"], flush=True)
    os.remove(one)

    if a.gpu:
        import torch
        from mustache_amd import lib
        from mustache_amd.findflanks import device_reads, flank_frame
        hip = lib.load(0)
        params = lib.make_params()
        torch.cuda.synchronize()
        t0 = time.time()
        rs_dev = device_reads(hip, packed, "cuda:0")
        torch.cuda.synchronize()
        t_h2d = time.time() - t0
        hip.run_findflanks(rs_dev, params)          # warm-up (workspace growth)
        t0 = time.time()
        out = hip.run_findflanks(rs_dev, params)
        t_ff = time.time() - t0
        have = flank_frame(out, packed.contig_names)
        base = pack.pack_reads(rs.contigs, rs, pool="clipped")
        b = flank_frame(hip.run_findflanks(device_reads(hip, base, "cuda:0"), params), base.contig_names)
        want = bigbam.replicate_table(b, copies)
        ok = want.to_csv(sep="\t") == have.to_csv(sep="\t")
        best = min(x["seconds"] for x in res["native"])
        res["gpu"] = {"h2d_s": round(t_h2d, 4), "h2d_GBps": round(res["packed_bytes"] / t_h2d / 1e9, 2),
                      "findflanks_s": round(t_ff, 4), "device_ms": round(hip.device_ms(), 3),
                      "flank_rows": int(len(have)), "table_equals_base_repeated": bool(ok),
                      "reads_per_s_resident": round(n_rec / t_ff),
                      "reads_per_s_pcie_inclusive": round(n_rec / (t_ff + t_h2d)),
                      "reads_per_s_from_bam_file": round(n_rec / (t_ff + t_h2d + best))}
        print(res["gpu"], flush=True)
        assert ok, "flank table of the big file differs from the base table repeated per copy"
    os.remove(path)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
