"""Where the host time of the end-to-end leg goes (cProfile of one _findflanks + one _pairflanks call after a warm call).
    python scripts/profile_e2e.py [n_records] > gpurun_out/e2e_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n_records = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    from mustache_amd import bamio, bigbam, lib
    from mustache_amd import findflanks as ff_mod
    from mustache_amd.findflanks import _findflanks
    from mustache_amd.pairflanks import _pairflanks
    from mustache_amd.simulate import simulate_readset
    lib.tune_allocator()
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    genome, rs, truth = simulate_readset(seed=7, contig_lens=(60000, 8000), n_insertions=8, coverage=100)
    copies = max(1, n_records // len(rs.reads))
    bam, fa, tsv1, tsv2 = (os.path.join(tmp, x) for x in ("e.bam", "e.fa", "e.ff.tsv", "e.pf.tsv"))
    contigs, n_rec = bigbam.write_replicated_bam(bam, rs, copies, procs=8)
    bamio.write_fasta(fa, {n: genome[n.rsplit("_", 1)[0]] for n, _ in contigs})
    d = lib.DEFAULT_PARAMS
    ff_args = (d["min_softclip_length"], d["min_softclip_count"], d["min_alignment_quality"], d["min_alignment_inner_length"],
               d["min_distance_to_mate"], d["min_softclip_ratio"], d["max_indel_ratio"], d["large_insertion_cutoff"], d["min_count_consensus"], "bench")
    pf_args = (d["max_direct_repeat_length"], d["min_alignment_quality"], d["min_alignment_inner_length"], d["max_junction_spanning_prop"], d["large_insertion_cutoff"])
    for _ in range(2):
        t0 = time.perf_counter()
        _findflanks(bam, *ff_args, output_file=tsv1, device="cuda:0", verbose=False)
        t1 = time.perf_counter()
        _pairflanks(tsv1, bam, fa, *pf_args, output_file=tsv2, device="cuda:0", verbose=False)
        t2 = time.perf_counter()
        print("records %d  findflanks %.4f s  pairflanks %.4f s  stages %s" % (n_rec, t1 - t0, t2 - t1, dict(ff_mod.LAST_TIMING)))
    for name, fn in (("findflanks", lambda: _findflanks(bam, *ff_args, output_file=tsv1, device="cuda:0", verbose=False)),
                     ("pairflanks", lambda: _pairflanks(tsv1, bam, fa, *pf_args, output_file=tsv2, device="cuda:0", verbose=False))):
        pr = cProfile.Profile()
        pr.enable()
        fn()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print("==== %s ====" % name)
        print(s.getvalue())
    for p in (bam, fa, tsv1, tsv2, tsv1 + ".mfki"):
        try:
            os.remove(p)
        except OSError:
            pass


if __name__ == "__main__":
    main()
