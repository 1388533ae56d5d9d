"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/rNN_pmc_traffic.json (bytes per launch and kernel).

    python scripts/pmc_table.py gpurun_out/round/pmc_fetch gpurun_out/round/pmc_write C2_saureus_1000x profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  As /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950,
FETCH_SIZE is doubled for kernels whose loads are 16 bytes per lane (a 128-byte request is counted as one 64-byte unit):
here that is the flags stream of k_scan_clips; the sparse gathers of the other kernels are taken as reported.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

WIDE_LOAD_KERNELS = ("k_scan_clips",)
MODELLED = ("k_scan_clips", "k_clip_events_bkt", "k_clip_events", "k_bucket_sort", "k_event_info", "k_classify", "k_direct_repeats")


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n


def per_launch(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    k = short(r["Kernel_Name"])
                    tot[k] += float(r["Counter_Value"])
                    cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch_dir, write_dir, config, out = sys.argv[1:5]
    f, w = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
    tab = {config: {}, "_all_kernels_kb_per_launch": {}}
    for k in sorted(set(f) | set(w)):
        tab["_all_kernels_kb_per_launch"][k] = {"fetch_size_kb": round(f.get(k, 0.0), 2), "write_size_kb": round(w.get(k, 0.0), 2)}
    for k in MODELLED:
        if k in f or k in w:
            mult = 2.0 if k in WIDE_LOAD_KERNELS else 1.0
            tab[config][k] = {"fetch_size_kb": round(f.get(k, 0.0), 2), "write_size_kb": round(w.get(k, 0.0), 2), "fetch_multiplier": mult,
                              "traffic_bytes": int(round((f.get(k, 0.0) * mult + w.get(k, 0.0)) * 1024))}
    with open(out, "w") as fh:
        json.dump(tab, fh, indent=1)
    print(json.dumps(tab[config], indent=1))


if __name__ == "__main__":
    main()
