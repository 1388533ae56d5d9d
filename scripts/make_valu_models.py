"""profiles/rNN_valu_models.json from the SQ_INSTS_VALU pass (profiles/rNN_pmc_valu.json, scripts/pmc_table.py) and the
work counters bench.py prints (roofline.work_counters_per_step of the bench line kept beside the pass):

    python scripts/make_valu_models.py r02 C2_saureus_1000x profiles/r02_bench_c2_quick.json [C5_meta_50mb profiles/r02_bench_c5_quick.json ...]

For every kernel without a streaming model: wave-level vector-ALU instructions per unit of work = SQ_INSTS_VALU per launch
x launches per step / units per step.  The unit is the counter the kernel's work scales with: columns x limbs of the
bit-vector edit distances (stats[12]) for k_leaf_myers, candidate pairs for the pair kernels, soft-clip events for the rest of
the consensus stage.  bench.py multiplies by the units counted in ITS run and by 64 lanes and divides by the kernel's measured
duration: the achieved vector-ALU issue rate, against 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNIT = {"k_leaf_myers": "myers_limb_steps", "k_inverted_repeat": "candidate_pairs", "k_spanning": "candidate_pairs", "k_assign": "candidate_pairs",
        "k_assign_small": "candidate_pairs"}
SKIP = ("k_scan_clips", "k_clip_events", "k_classify", "k_bucket_sort", "k_direct_repeats", "rocprim", "k_fill", "k_gather", "k_scan_")


def main():
    tag = sys.argv[1]
    valu = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_valu.json")))
    out = {}
    for config, bench_file in zip(sys.argv[2::2], sys.argv[3::2]):
        line = [x for x in open(bench_file) if x.startswith("{")][-1]
        j = json.loads(line)
        work = j["roofline"]["work_counters_per_step"]
        table = {}
        for l in open(bench_file.replace("_quick.json", "_hip_events.txt")):
            f = l.split()
            if len(f) == 4 and not l.startswith("#") and f[0] != "kernel" and f[0] not in table:
                table[f[0]] = int(f[2]) // 3           # launches per step (three bracketed steps)
        models = {}
        for kernel, v in valu[config].items():
            if kernel == "_meta":
                continue
            k = kernel.split("<")[0] if kernel.startswith("k_direct") else kernel
            if k.startswith(SKIP) or not v.get("valu_insts_per_launch"):
                continue
            counter = UNIT.get(kernel, "events")
            units = work.get(counter)
            if not units:
                continue
            launches = table.get(kernel, 1) or 1
            models[kernel] = {"work_counter": counter, "unit": counter.replace("_", " "),
                              "valu_insts_per_unit": round(v["valu_insts_per_launch"] * launches / units, 4),
                              "waves_per_launch": v.get("waves_per_launch"),
                              "source": "profiles/%s_pmc_valu.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES, %s): %d wave instructions per launch x %d launches per step / %d %s per step"
                                        % (tag, config, v["valu_insts_per_launch"], launches, units, counter.replace("_", " ")),
                              "note": "issue-rate roof; the kernel is a chain of dependent loads and LDS steps per job, so the distance to the roof is occupancy and latency, not arithmetic"}
        models["_meta"] = valu[config].get("_meta", {})
        out[config] = models
    path = os.path.join(ROOT, "profiles", tag + "_valu_models.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, {c: len(m) - 1 for c, m in out.items()})


if __name__ == "__main__":
    main()
