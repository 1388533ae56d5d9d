"""The device decode of one BAM file, three times (for `rocprofv3 --kernel-trace --stats -- python3 scripts/profile_decode.py FILE`):
prints the stage clocks of every call.  `--make N`: write the end-to-end leg's file with about N records to FILE and exit (done in
a run of its own, without the profiler: the writer forks worker processes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    path = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == "--make":
        from mustache_amd import bigbam
        from mustache_amd.simulate import simulate_readset
        genome, rs, truth = simulate_readset(seed=7, contig_lens=(60000, 8000), n_insertions=8, coverage=100)
        contigs, n_rec = bigbam.write_replicated_bam(path, rs, max(1, int(sys.argv[3]) // len(rs.reads)), procs=8)
        print(json.dumps({"records": n_rec, "bytes": os.path.getsize(path)}))
        return
    import torch
    from mustache_amd import gpudecode, lib
    hip = lib.load(0)
    for k in range(3):
        dev = gpudecode.decode_bam(hip, path, pool="clipped", index_out=path + ".mfki")
        torch.cuda.synchronize()
        st = dev.io_stats
        print(json.dumps({"call": k, "reads": dev.n_reads, "records": st["n_records"], "inflated_bytes": st["uncompressed_bytes"],
                          "block_table_and_header_s": round(st["seconds_block_table_and_header"], 4), "upload_file_s": round(st["seconds_upload_file"], 4),
                          "device_stage1_s": round(st["seconds_device_stage1"], 4), "device_stage2_s": round(st["seconds_device_stage2"], 4),
                          "total_s": round(st["seconds_total"], 4)}), flush=True)
        del dev
    os.remove(path + ".mfki")


if __name__ == "__main__":
    main()
