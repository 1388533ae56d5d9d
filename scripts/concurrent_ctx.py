"""Several library contexts on one GPU, one host thread each, running the fused findflanks + pairflanks call on the same
resident inputs at the same time (bench.py's `concurrent_contexts` leg for any config and context count).  One JSON line:
one context alone, the contexts together, and whether every context's tables equal the single-context ones.

    python scripts/concurrent_ctx.py [--config C2_saureus_1000x] [--contexts 2] [--steps 40] [--start-at EPOCH_SECONDS]

--start-at: wait for that wall-clock time before the concurrent leg (several PROCESSES of this script on one GPU started together,
scripts/concurrent_procs.sh: each process has its own HIP runtime, so their launch paths do not queue on one lock).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2_saureus_1000x")
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--start-at", type=float, default=0.0)
    a = ap.parse_args()
    import torch
    import bench
    from mustache_amd import lib
    from mustache_amd.findflanks import device_reads
    lib.tune_allocator()
    device = "cuda:0"
    packed, g, goff, ranks, cfg, truth, batch, _ = bench.make_workload(a.config, 100, "all")
    hip = lib.load(0)
    reads = device_reads(hip, packed, device)
    g_dev = torch.from_numpy(g).to(device)
    params = lib.make_params()
    hip.span_timing(False)
    for _ in range(3):
        hip.run_flankpairs(reads, params, ranks, g_dev, goff)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        hip.run_flankpairs(reads, params, ranks, g_dev, goff)
    torch.cuda.synchronize()
    alone = time.perf_counter() - t0
    n = packed.n_reads
    late = a.start_at > 0 and time.time() > a.start_at
    while a.start_at > 0 and time.time() < a.start_at:
        time.sleep(0.001)
    t_begin = time.time()
    together = bench.concurrent_contexts(hip, reads, params, ranks, g_dev, goff, n, a.contexts, a.steps, 0)
    print(json.dumps({"config": a.config, "reads": n,
                      "one_context": {"ms_per_step": round(alone / a.steps * 1e3, 4), "reads_per_s": round(n * a.steps / alone, 1)},
                      "together": together, "began": round(t_begin, 3), "ended": round(time.time(), 3), "started_late": late}))
    sys.exit(0 if together["tables_equal"] else 1)


if __name__ == "__main__":
    main()
