#!/bin/bash
# GPU box: SQ counters of the inflate kernel (diagnostic)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_inflate
rm -rf $O && mkdir -p $O
timeout -s KILL 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/raw -- python3 scripts/bench_inflate.py --reps 1 > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_inflate/raw/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0]
        if 'inflate' not in k and 'crc' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items(): print("   %-20s %16.0f per launch" % (c, v / n[(k, c)]))
PY
