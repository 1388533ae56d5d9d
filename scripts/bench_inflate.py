"""GPU BGZF inflate (mfk_bgzf_inflate) on the BAM file of the bench's end-to-end leg: GB/s of inflated bytes, resident
compressed bytes -> resident inflated bytes, checked against the host inflate (crc32 of every block).

    python scripts/bench_inflate.py [--records 4000000] [--reps 5]
"""
import argparse, json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=4000000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import numpy as np, torch
    from mustache_amd import bigbam, gpuinflate, lib
    from mustache_amd.simulate import simulate_readset
    tmp = os.environ.get("TMPDIR", "/tmp")
    genome, rs, truth = simulate_readset(seed=11, contig_lens=(60000, 30000, 8000), n_insertions=9, coverage=60)
    copies = max(1, a.records // len(rs.reads))
    bam = os.path.join(tmp, "mfk_inflate_bench.bam")
    contigs, n_rec = bigbam.write_replicated_bam(bam, rs, copies, procs=8)
    data = open(bam, "rb").read()
    os.remove(bam)
    hip = lib.load(0)
    blocks, crcs, total = gpuinflate.bgzf_blocks(data)
    comp = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    out = torch.empty(total, dtype=torch.uint8, device="cuda:0")
    hip.run_bgzf_inflate(comp, blocks, out)          # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        hip.run_bgzf_inflate(comp, blocks, out)
        times.append(time.perf_counter() - t0)
    host = out.cpu().numpy()
    bad = sum(1 for b, c in zip(blocks, crcs) if zlib.crc32(host[int(b["out_off"]):int(b["out_off"]) + int(b["out_len"])].tobytes()) != int(c))
    t0 = time.perf_counter()
    n_chk = min(2000, blocks.shape[0])
    for b in blocks[:n_chk]:
        zlib.decompress(data[int(b["comp_off"]):int(b["comp_off"]) + int(b["comp_len"])], -15)
    z_s = (time.perf_counter() - t0) * blocks.shape[0] / n_chk
    best = min(times)
    print(json.dumps({"records": n_rec, "compressed_bytes": len(data), "inflated_bytes": int(total), "blocks": int(blocks.shape[0]),
                      "gpu_inflate_s": round(best, 5), "gpu_GBps_inflated": round(total / best / 1e9, 2), "all_reps_s": [round(t, 5) for t in times],
                      "crc_mismatches": bad, "zlib_one_core_s_extrapolated": round(z_s, 3)}))


if __name__ == "__main__":
    main()
