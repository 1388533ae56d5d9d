"""Extended parity fuzz on the GPU box (test infrastructure, not collected by pytest): random simulated read sets through
mfk_flankpairs (HIP) and the CPU oracle, results compared byte for byte.

    python tests/fuzz_gpu.py [--seeds 120] [--first 1000] [--budget-s 240]

Two generators: records.simulate_readset (small genomes with indels, deletions, low mapq, wild-type reads, varied
parameters) and synth.synth_packed (deep pile-ups: a few sites at 300x .. 3000x, hundreds to thousands of words per job).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=120)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--budget-s", type=float, default=240.0)
    a = ap.parse_args()
    import numpy as np
    import torch
    from mustache_amd import lib, pairflanks as pfh
    from mustache_amd.findflanks import device_reads
    from mustache_amd.pack import pack_reads
    from mustache_amd.simulate import simulate_readset
    from mustache_amd.synth import synth_packed
    from oracle.oracle import Oracle
    hip, ora = lib.load(0), Oracle()
    t_start, bad, done = time.time(), [], 0
    for seed in range(a.first, a.first + a.seeds):
        if time.time() - t_start > a.budget_s:
            break
        rng = np.random.default_rng(seed)
        fp = dict(min_softclip_length=int(rng.choice([4, 8, 12])), min_softclip_count=int(rng.choice([2, 3, 4])),
                  min_softclip_ratio=float(rng.choice([0.05, 0.1, 0.15])), min_count_consensus=int(rng.choice([1, 2, 3])),
                  max_indel_ratio=float(rng.choice([0.03, 0.2])), min_distance_to_mate=int(rng.choice([12, 22, 30])),
                  max_direct_repeat_length=int(rng.choice([10, 20, 30])), min_alignment_quality=int(rng.choice([0, 20, 30])))
        if seed % 3 != 2:
            kw = dict(seed=seed, contig_lens=(int(rng.integers(3000, 12000)),) * int(rng.integers(1, 5)),
                      n_insertions=int(rng.integers(1, 5)), coverage=int(rng.integers(8, 200)),
                      read_len=int(rng.choice([75, 100, 150])), sub_rate=float(rng.choice([0.002, 0.01, 0.03, 0.06])),
                      wildtype_frac=float(rng.choice([0, 0.1, 0.4])))
            genome, rs, _ = simulate_readset(**kw)
            packed = pack_reads(rs.contigs, rs, pool="all")
        else:
            kw = dict(seed=seed, contig_lens=(int(rng.integers(20000, 60000)),) * int(rng.integers(1, 3)),
                      n_insertions=int(rng.integers(1, 4)), coverage=int(rng.choice([300, 800, 1500, 3000])))
            packed, genome, _ = synth_packed(**kw)
        params = lib.make_params(**fp)
        g, goff = pfh.genome_arrays(genome, packed.contig_names)
        ranks = pfh.contig_ranks(packed.contig_names)
        of, op = ora.run_flankpairs(ora.reads(packed), params, ranks, g, goff)
        hf, hp = hip.run_flankpairs(device_reads(hip, packed, "cuda:0"), params, ranks, torch.from_numpy(g).to("cuda:0"), goff)
        same = (hf["flanks"].tobytes() == of["flanks"].tobytes() and hf["seqs"] == of["seqs"] and hf["sites"].tobytes() == of["sites"].tobytes() and
                list(hf["stats"][:12]) == list(of["stats"][:12]) and hp["pairs"].tobytes() == op["pairs"].tobytes() and hp["strs"] == op["strs"] and
                list(hp["stats"][:5]) == list(op["stats"][:5]))
        done += 1
        if not same:
            bad.append((seed, kw, fp))
            print("MISMATCH seed %d %r %r" % (seed, kw, fp), flush=True)
        if done % 10 == 0:
            print("%d cases, %d mismatches, %.0f s (last: %d reads, %d flanks, %d pairs)" % (
                done, len(bad), time.time() - t_start, packed.n_reads, len(of["flanks"]), len(op["pairs"])), flush=True)
    print("fuzz: %d cases, %d mismatches" % (done, len(bad)), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
