"""Digests of the CPU oracle's results on the two BASELINE configs whose oracle run is too long for the GPU suite
(test infrastructure; run in the build container, ~5 minutes):

    python tests/golden/make_fullsize_digests.py        # writes tests/golden/fullsize_digests.json

The GPU test (tests/test_gpu_fullsize.py::test_fullsize_digest) generates the same inputs from the same seeds, runs the
HIP library and compares digests: C5 (69 M reads, 10 000 insertions) and 12 x C1 in one batch (the per-GPU share of the
96-isolate config) are then checked at full size on every driver run without 100 s of oracle time on the GPU box.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from mustache_amd import lib
    from oracle.oracle import Oracle
    from tests.util import FULLSIZE_DIGESTS, fullsize_inputs, result_digests
    ora = Oracle()
    out = {"numpy": np.__version__}
    names = sys.argv[1:] or ["C4_batch12", "C5_meta_50mb"]
    if os.path.exists(FULLSIZE_DIGESTS):
        out.update(json.load(open(FULLSIZE_DIGESTS)))
    for name in names:
        t0 = time.time()
        packed, g, goff, ranks = fullsize_inputs(name)
        t1 = time.time()
        params = lib.make_params()
        ro = ora.reads(packed)
        ff = ora.run_findflanks(ro, params)
        pf = ora.run_pairflanks(ro, params, ff["flanks"], ff["seqs"], ranks, g, goff)
        d = result_digests(ff, pf)
        d.update(n_reads=int(packed.n_reads), generate_s=round(t1 - t0, 1), oracle_s=round(time.time() - t1, 1))
        out[name] = d
        print(name, d, flush=True)
    with open(FULLSIZE_DIGESTS, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
