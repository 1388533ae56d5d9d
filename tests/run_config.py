"""Full-size parity run (test infrastructure, not collected by pytest): one synthetic BASELINE config through the HIP
library and the CPU oracle, with timings (run on the GPU box; C5 needs ~100 s of oracle time, too long for the suite).

    python tests/run_config.py --config C5_meta_50mb --seed 7 [--out gpurun_out/c5.json]
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
    ap.add_argument("--config", default="C5_meta_50mb")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import numpy as np
    import torch
    from mustache_amd import lib, pairflanks as pfh
    from mustache_amd.findflanks import device_reads
    from mustache_amd.synth import CONFIGS, synth_packed
    from oracle.oracle import Oracle
    t0 = time.time()
    packed, genome, truth = synth_packed(**CONFIGS[a.config], seed=a.seed)
    res = {"config": a.config, "reads": packed.n_reads, "insertions": len(truth), "generate_s": round(time.time() - t0, 1)}
    print(res, flush=True)
    hip = lib.load(0)
    params = lib.make_params()
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    rd = device_reads(hip, packed, "cuda:0")
    gd = torch.from_numpy(g).to("cuda:0")
    for k in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ff = hip.run_findflanks(rd, params)
        t1 = time.perf_counter()
        ms_ff = hip.device_ms()
        pf = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, gd, goff)
        t2 = time.perf_counter()
        print("step %d: findflanks %.2f ms (device %.2f), pairflanks %.2f ms (device %.2f)" % (
            k, (t1 - t0) * 1e3, ms_ff, (t2 - t1) * 1e3, hip.device_ms()), flush=True)
    res.update(findflanks_ms=round((t1 - t0) * 1e3, 3), findflanks_device_ms=round(ms_ff, 3),
               pairflanks_ms=round((t2 - t1) * 1e3, 3), pairflanks_device_ms=round(hip.device_ms(), 3),
               reads_per_s=round(packed.n_reads / (t2 - t0)), flank_rows=int(len(ff["flanks"])), pairs=int(len(pf["pairs"])),
               events=int(ff["stats"][10]))
    print(res, flush=True)
    ora = Oracle()
    ro = ora.reads(packed)
    t0 = time.time()
    off = ora.run_findflanks(ro, params)
    t1 = time.time()
    opf = ora.run_pairflanks(ro, params, off["flanks"], off["seqs"], ranks, g, goff)
    t2 = time.time()
    res.update(oracle_findflanks_s=round(t1 - t0, 2), oracle_pairflanks_s=round(t2 - t1, 2))
    same = (ff["flanks"].tobytes() == off["flanks"].tobytes() and ff["seqs"] == off["seqs"] and
            list(ff["stats"][:12]) == list(off["stats"][:12]) and pf["pairs"].tobytes() == opf["pairs"].tobytes() and
            pf["strs"] == opf["strs"])
    got = {(packed.contig_names[int(p["contig_id"])], int(p["pos_5p"]), int(p["pos_3p"])) for p in pf["pairs"]}
    res.update(identical_to_oracle=bool(same), truth_recovered=len(set(truth) & got), stats=[int(x) for x in ff["stats"][:12]],
               pair_stats=[int(x) for x in pf["stats"][:5]])
    print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)
    assert same, "HIP result differs from the oracle"


if __name__ == "__main__":
    main()
