"""Diagnostic (needs a library built with `make SJ_STAMP=1`): cycles of thread 0 between the phases of k_job_small,
summed over the jobs of a few C5 steps."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mustache_amd import lib, pairflanks as pfh
from mustache_amd.findflanks import device_reads
from mustache_amd.synth import CONFIGS, synth_packed
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5_tenth"
packed, genome, truth = synth_packed(**CONFIGS[cfg], seed=7)
hip = lib.load(0)
rd = device_reads(hip, packed, "cuda:0")
params = lib.make_params()
hip.run_findflanks(rd, params)
out = (ctypes.c_ulonglong * 56)()
hip.so.mfk_debug_sj(out)
for _ in range(3):
    hip.run_findflanks(rd, params)
hip.so.mfk_debug_sj(out)
names = ["load", "leaves", "LCP matrix + rep list", "summaries", "pivot round", "components", "LCP matrix: loop (+outsider rounds)", "components 2", "vote: sums, chars", "choice (+DFS on ties)", "vote: reach of the winner", "hasm + leaf sigs + barrier"]
jobs = max(out[12], 1)
tot = sum(out[i] for i in range(12))
for i, n in enumerate(names):
    print("%-18s %10.0f cycles/job  %5.1f %%" % (n, out[i] / jobs, 100.0 * out[i] / max(tot, 1)))
print("ties: %d" % out[15])
print("dynamic LDS need (KB bin: jobs):", {4 * i: int(out[16 + i]) for i in range(40) if out[16 + i]})
print("jobs %d (3 steps), leaves/job %.1f, total cycles/job %.0f" % (jobs, out[13] / jobs, tot / jobs))
