"""rocprofv3 --pmc passes -> profiles/rNN_pmc_traffic.json (HBM bytes per launch and kernel) and, with a third pass,
profiles/rNN_pmc_valu.json (vector-ALU instructions per launch and kernel).

    python scripts/pmc_table.py FETCH_DIR WRITE_DIR CONFIG OUT_TRAFFIC.json [VALU_DIR OUT_VALU.json]

Existing output files are updated (one entry per config), so several configs share a file.
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  As /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950,
FETCH_SIZE is doubled for kernels whose loads are 16 bytes per lane (a 128-byte request is counted as one 64-byte unit):
here that is the flags stream of k_scan_clips; the sparse gathers of the other kernels are taken as reported.
SQ_INSTS_VALU counts wave instructions (x 64 lanes = lane-operations whatever the exec mask); SQ_WAVES the waves launched.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

WIDE_LOAD_KERNELS = ("k_scan_clips",)


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0] if n.startswith("k_scan_small") or n.startswith("k_scan_chain") else n


def per_launch(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    k = short(r["Kernel_Name"])
                    tot[k] += float(r["Counter_Value"])
                    cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def lib_sha():
    """sha256 of the library the passes ran on: bench.py uses a table only on the build it was measured on."""
    import hashlib
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mustache_amd", "csrc", "libmfk.so")
    with open(p, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def update(path, config, entry):
    try:
        with open(path) as fh:
            tab = json.load(fh)
    except (OSError, ValueError):
        tab = {}
    entry["_meta"] = {"lib_sha256_16": lib_sha(), "what": "rocprofv3 --pmc passes of `bench.py --quick --config %s` on this build (scripts/profile_round.sh)" % config}
    tab[config] = entry
    with open(path, "w") as fh:
        json.dump(tab, fh, indent=1, sort_keys=True)


def main():
    fetch_dir, write_dir, config, out = sys.argv[1:5]
    (f, _), (w, _) = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
    entry = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_") and not k.startswith("rocprim"):
            continue
        mult = 2.0 if k in WIDE_LOAD_KERNELS else 1.0
        entry[k] = {"fetch_size_kb": round(f.get(k, 0.0), 2), "write_size_kb": round(w.get(k, 0.0), 2), "fetch_multiplier": mult,
                    "traffic_bytes": int(round((f.get(k, 0.0) * mult + w.get(k, 0.0)) * 1024))}
    update(out, config, entry)
    print(json.dumps({k: v["traffic_bytes"] for k, v in entry.items() if k != "_meta"}, indent=1))
    if len(sys.argv) >= 7:
        valu_dir, out_valu = sys.argv[5:7]
        (v, calls), (wv, _) = per_launch(valu_dir, "SQ_INSTS_VALU"), per_launch(valu_dir, "SQ_WAVES")
        entry = {k: {"valu_insts_per_launch": int(round(v[k])), "waves_per_launch": int(round(wv.get(k, 0))), "launches_seen": calls[k]}
                 for k in sorted(v) if k.startswith("k_") or k.startswith("rocprim")}
        update(out_valu, config, entry)
        print(json.dumps({k: e["valu_insts_per_launch"] for k, e in entry.items() if k != "_meta"}, indent=1))


if __name__ == "__main__":
    main()
