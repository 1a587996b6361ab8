"""Kernel timeline of ONE bench step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: start offset, duration and the
idle gap in front of every kernel of the last complete step (a step = everything from one preprocess (or fused-stem) launch to the next)."""
import csv, glob, sys, os
src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
first = os.environ.get("STEP_FIRST", "k_preprocess")
idx = [i for i, r in enumerate(rows) if first in r[2]]
if len(idx) < 2:            # the fused stem reads the uint8 image itself: a step starts with the stem launch
    idx = [i for i, r in enumerate(rows) if "conv_stem_c32_c64" in r[2] or "conv_stem_pair_" in r[2]]       # (bf16 / fp16 stem; the split-fp16 stem)
# the last COMPLETE detect step (it ends with the NMS launch; bench.py's closing yolo_time_forward passes have none)
a, b = idx[-2], idx[-1]
for k in range(len(idx) - 1, 0, -1):
    if any("k_nms_image" in r[2] for r in rows[idx[k - 1]:idx[k]]):
        a, b = idx[k - 1], idx[k]
        break
t0 = rows[a][0]; prev_end = rows[a - 1][1] if a else t0
tot_gap = 0; tot_k = 0
for s, e, k in rows[a:b]:
    gap = s - prev_end; tot_gap += max(gap, 0); tot_k += e - s
    print("%9.2f us  dur %8.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, k[:100]))
    prev_end = max(prev_end, e)
print("step: %d launches, kernel time %.1f us, gaps %.1f us, span %.1f us" % (b - a, tot_k / 1e3, tot_gap / 1e3, (rows[b][0] - t0) / 1e3))
