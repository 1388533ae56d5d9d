import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mustache_amd import lib, pairflanks as pfh
from mustache_amd.findflanks import device_reads
from mustache_amd.synth import CONFIGS, synth_packed
cfg = sys.argv[1]
packed, genome, truth = synth_packed(**CONFIGS[cfg], seed=7)
g, goff = pfh.genome_arrays(genome, packed.contig_names); ranks = pfh.contig_ranks(packed.contig_names)
hip = lib.load(0); rd = device_reads(hip, packed, "cuda:0"); gd = torch.from_numpy(g).to("cuda:0"); params = lib.make_params()
for i in range(3): hip.run_flankpairs(rd, params, ranks, gd, goff)
os.environ["MFK_HOST_TIMING"] = "1"
t = time.perf_counter(); hip.run_flankpairs(rd, params, ranks, gd, goff); print("python wall %.1f us" % ((time.perf_counter() - t) * 1e6))
