#!/bin/bash
# One PMC pass over tools/prof_forward.py for an arbitrary counter set; per-kernel sums of the LAST forward are printed.
# usage: tools/probe/pmc_pass.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ..."
export TMPDIR=/tmp
D=gpurun_out/pmc_probe; rm -rf $D; mkdir -p $D
ITERS=2 timeout 600 rocprofv3 --pmc $1 --output-format csv -d $D -o p -- python3 tools/prof_forward.py > $D.log 2>&1
python3 - "$D" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# segment forwards at k_preprocess; keep the last complete one
ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))
byid = collections.defaultdict(dict); name = {}
for r in rows:
    byid[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
starts = [i for i in ids if "k_preprocess" in name[i]]
lo, hi = starts[-2], starts[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for i in ids:
    if lo <= i < hi:
        k = name[i][:70]; cnt[k] += 1
        for c, v in byid[i].items(): agg[k][c] += v
for k in agg:
    print("%-72s n=%2d " % (k, cnt[k]) + "  ".join("%s=%.3g" % (c, v) for c, v in sorted(agg[k].items())))
PY
