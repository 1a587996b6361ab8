"""Condense tools/profile_round.sh output into the files kept under profiles/:
  <R>_bench.json, <R>_bench_under_rocprof.json, <R>_bench_kernel_stats.csv, <R>_hbm_traffic.json, <R>_mfma_util.json.
Counter conventions follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE is
doubled on gfx950 for 16-B-per-lane streams, SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over SIMDs."""
import sys, os, csv, json, glob, shutil, collections

src, R = sys.argv[1], sys.argv[2]
SIZE, BATCH = int(os.environ.get("SIZE", "416")), int(os.environ.get("B", "32"))
WORKLOAD = "YOLOv3 %dx%d batch %d bf16, conv kernels of one forward (the last one of tools/prof_forward.py, tuned plan)" % (SIZE, SIZE, BATCH)
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
os.makedirs(dst, exist_ok=True)


def last_json(path):
    line = None
    for l in open(path):
        if l.startswith("{"):
            line = l
    return json.loads(line) if line else None


for name in ("bench.json", "bench_fp8.json", "bench_mixed.json", "bench_fp16.json", "bench_fp32.json", "bench_fp16x2.json", "bench_mixed16.json", "bench_under_rocprof.json",
             "bench_fp8_under_rocprof.json", "bench_mixed_under_rocprof.json", "bench_fp16x2_under_rocprof.json", "bench_mixed16_under_rocprof.json"):
    j = last_json(os.path.join(src, name)) if os.path.exists(os.path.join(src, name)) else None
    if j:
        json.dump(j, open(os.path.join(dst, "%s_%s" % (R, name)), "w"), indent=1)

st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True) or glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, "%s_bench_kernel_stats.csv" % R))
for dt in ("fp8", "mixed", "fp16x2", "mixed16"):
    sd = glob.glob(os.path.join(src, "stats_%s" % dt, "**", "*kernel_stats.csv"), recursive=True)
    if sd:
        shutil.copy(sd[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (R, dt)))


def build_hash():
    """sha1 over the kernel sources and the committed bf16 plan: bench.py prints `roofline.traffic` only when the PMC
    passes were taken on exactly the code and plan it is running (a stale constant can never be printed)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from yolo_tensorflow_amd import buildinfo
    return buildinfo.source_hash(SIZE, BATCH, "bf16")


def counters(sub):
    """rows of one PMC pass in dispatch order: [(dispatch_id, kernel, {counter: value}, duration_ns)]; also copies the
    raw CSV next to the summaries (they are 70-300 KB)"""
    f = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return []
    shutil.copy(f[0], os.path.join(dst, "%s_%s_counter_collection.csv" % (R, sub)))
    rows = {}
    for r in csv.DictReader(open(f[0])):
        try:
            d = int(r["Dispatch_Id"])
            e = rows.setdefault(d, [d, r["Kernel_Name"], {}, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
            e[2][r["Counter_Name"]] = e[2].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        except (ValueError, KeyError, TypeError):
            pass
    return [tuple(rows[d]) for d in sorted(rows)]


def is_conv(k):
    return k.startswith("void conv_") or k.startswith("conv_")


def last_forward(rows):
    """the dispatches of the LAST forward: everything from the last k_preprocess (or, without one, fused-stem) dispatch on
    (tools/prof_forward.py runs one default-plan forward and then ITERS tuned ones; each forward starts with exactly one such launch)"""
    starts = [i for i, r in enumerate(rows) if "k_preprocess" in r[1]]
    if not starts:          # round 3: the fused stem reads the uint8 image itself -- a forward starts with the stem launch
        starts = [i for i, r in enumerate(rows) if "conv_stem_c32_c64" in r[1]]
    return rows[starts[-1]:] if starts else rows


def per_kernel(rows, counter):
    out = collections.OrderedDict()
    for _, k, c, ns in rows:
        if not is_conv(k) or counter not in c:
            continue
        e = out.setdefault(k, {"sum": 0.0, "launches": 0, "ns": 0})
        e["sum"] += c[counter]; e["launches"] += 1; e["ns"] += ns
    return out


fetch = last_forward(counters("pmc_FETCH_SIZE")); write = last_forward(counters("pmc_WRITE_SIZE"))
if fetch and write:
    f_det = per_kernel(fetch, "FETCH_SIZE"); w_det = per_kernel(write, "WRITE_SIZE")
    f_tot = sum(v["sum"] for v in f_det.values()); w_tot = sum(v["sum"] for v in w_det.values())
    json.dump({
        "workload": WORKLOAD,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; forwards segmented at their first dispatch (k_preprocess / the fused stem); KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md",
        "source_hash": build_hash(),
        "conv_launches": sum(v["launches"] for v in f_det.values()),
        "conv_fetch_size_kib_raw": f_tot, "conv_write_size_kib": w_tot,
        "conv_hbm_bytes_per_forward": int((2 * f_tot + w_tot) * 1024),
        "per_kernel_kib": {k: {"fetch_raw": f_det[k]["sum"], "write": w_det.get(k, {}).get("sum"), "launches": f_det[k]["launches"]} for k in f_det},
    }, open(os.path.join(dst, "%s_hbm_traffic.json" % R), "w"), indent=1)

mf = last_forward(counters("pmc_SQ_VALU_MFMA_BUSY_CYCLES"))
if mf:
    rows = {}
    for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES"):
        for k, v in per_kernel(mf, name).items():
            e = rows.setdefault(k, {"launches": v["launches"], "ns": v["ns"]})
            e[name] = v["sum"]
    for k, e in rows.items():
        busy, gui, ns = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("GRBM_GUI_ACTIVE"), e["ns"]
        # GRBM_GUI_ACTIVE is summed over 8 XCDs; MFMA busy is summed over 256 CUs x 4 SIMDs
        e["mfma_util"] = (busy / 1024.0) / (gui / 8.0) if busy and gui else None
        # GRBM-derived clock reads high on dispatches shorter than ~0.3 ms (guide, DVFS note), so also state utilisation
        # against wall time at the 2.4 GHz peak clock: this one x 2.5 PF = TFLOP/s
        e["mfma_util_wall_2p4ghz"] = (busy / 1024.0) / (ns * 2.4) if busy and ns else None
        e["eff_clock_ghz"] = (gui / 8.0) / ns if gui and ns else None
    tb = sum(r.get("SQ_VALU_MFMA_BUSY_CYCLES") or 0 for r in rows.values()); tg = sum(r.get("GRBM_GUI_ACTIVE") or 0 for r in rows.values())
    json.dump({"workload": WORKLOAD,
               "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; forwards segmented at their first dispatch (k_preprocess / the fused stem); util = busy/(256 CUs*4 SIMDs) / (GRBM_GUI_ACTIVE/8 XCDs)",
               "source_hash": build_hash(),
               "conv_launches": sum(r["launches"] for r in rows.values()),
               "all_conv_mfma_util": (tb / 1024.0) / (tg / 8.0) if tg else None,
               "all_conv_mfma_util_wall_2p4ghz": (tb / 1024.0) / (sum(r["ns"] for r in rows.values()) * 2.4), "per_kernel": rows},
              open(os.path.join(dst, "%s_mfma_util.json" % R), "w"), indent=1)
print("profiles written to", dst)
