"""Condense tools/profile_round.sh output into the files kept under profiles/:
  <R>_bench.json, <R>_bench_under_rocprof.json, <R>_bench_kernel_stats.csv, <R>_hbm_traffic.json, <R>_mfma_util.json.
Counter conventions follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE is
doubled on gfx950 for 16-B-per-lane streams, SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over SIMDs."""
import sys, os, csv, json, glob, shutil, collections

src, R = sys.argv[1], sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
os.makedirs(dst, exist_ok=True)


def last_json(path):
    line = None
    for l in open(path):
        if l.startswith("{"):
            line = l
    return json.loads(line) if line else None


for name in ("bench.json", "bench_fp8.json", "bench_under_rocprof.json"):
    j = last_json(os.path.join(src, name))
    if j:
        json.dump(j, open(os.path.join(dst, "%s_%s" % (R, name)), "w"), indent=1)

st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True) or glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, "%s_bench_kernel_stats.csv" % R))


def counters(sub):
    """{counter: {kernel: [values per dispatch in order]}} plus per-kernel durations of the same dispatches"""
    f = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
    if not f:
        return out, dur
    for r in csv.DictReader(open(f[0])):
        try:
            out[r["Counter_Name"]][r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            dur[r["Kernel_Name"]][int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except (ValueError, KeyError, TypeError):
            pass
    return out, dur


def last_forward(per_kernel, n_forwards):
    """sum over the dispatches of the LAST forward: each kernel's last (count / n_forwards) dispatches"""
    tot = 0.0; detail = {}
    for k, v in per_kernel.items():
        if not (k.startswith("void conv_") or k.startswith("conv_")):
            continue
        v = sorted(v); per = max(1, len(v) // n_forwards)
        s = sum(x for _, x in v[-per:])
        detail[k] = {"sum": s, "launches": per}; tot += s
    return tot, detail


NF = 3        # prof_forward.py: 1 planning forward + ITERS=2
fetch, _ = counters("pmc_FETCH_SIZE"); write, _ = counters("pmc_WRITE_SIZE")
if fetch and write:
    f_tot, f_det = last_forward(fetch["FETCH_SIZE"], NF); w_tot, w_det = last_forward(write["WRITE_SIZE"], NF)
    json.dump({
        "workload": "YOLOv3 416x416 batch 32 bf16, conv kernels of one forward",
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_forward.py (last forward); KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md",
        "conv_fetch_size_kib_raw": f_tot, "conv_write_size_kib": w_tot,
        "conv_hbm_bytes_per_forward": int((2 * f_tot + w_tot) * 1024),
        "per_kernel_kib": {k: {"fetch_raw": f_det[k]["sum"], "write": w_det.get(k, {}).get("sum"), "launches": f_det[k]["launches"]} for k in f_det},
    }, open(os.path.join(dst, "%s_hbm_traffic.json" % R), "w"), indent=1)

mf, dur = counters("pmc_SQ_VALU_MFMA_BUSY_CYCLES")
if mf:
    rows = {}
    for k in mf["SQ_VALU_MFMA_BUSY_CYCLES"]:
        if "conv_" not in k:
            continue
        n = max(1, len(mf["SQ_VALU_MFMA_BUSY_CYCLES"][k]) // NF)
        def tail(c):
            return sum(x for _, x in sorted(mf[c][k])[-n:]) if k in mf[c] else None
        ids = [d for d, _ in sorted(mf["SQ_VALU_MFMA_BUSY_CYCLES"][k])[-n:]]
        ns = sum(dur[k][d] for d in ids)
        busy, gui, cu = tail("SQ_VALU_MFMA_BUSY_CYCLES"), tail("GRBM_GUI_ACTIVE"), tail("SQ_BUSY_CU_CYCLES")
        rows[k] = {"launches": n, "ns": ns, "mfma_busy_cycles": busy, "grbm_gui_active": gui, "sq_busy_cu_cycles": cu,
                   # GRBM_GUI_ACTIVE is summed over 8 XCDs; MFMA busy is summed over 256 CUs x 4 SIMDs
                   "mfma_util": (busy / 1024.0) / (gui / 8.0) if busy and gui else None,
                   # GRBM-derived clock reads high on dispatches shorter than ~0.3 ms (guide, DVFS note), so also
                   # state utilisation against wall time at the 2.4 GHz peak clock: this one x 2.5 PF = TFLOP/s
                   "mfma_util_wall_2p4ghz": (busy / 1024.0) / (ns * 2.4) if busy and ns else None,
                   "eff_clock_ghz": (gui / 8.0) / ns if gui and ns else None}
    tb = sum(r["mfma_busy_cycles"] or 0 for r in rows.values()); tg = sum(r["grbm_gui_active"] or 0 for r in rows.values())
    json.dump({"workload": "YOLOv3 416x416 batch 32 bf16, conv kernels of one forward",
               "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE over tools/prof_forward.py (last forward); util = busy/(256 CUs*4 SIMDs) / (GRBM_GUI_ACTIVE/8 XCDs)",
               "all_conv_mfma_util": (tb / 1024.0) / (tg / 8.0) if tg else None,
               "all_conv_mfma_util_wall_2p4ghz": (tb / 1024.0) / (sum(r["ns"] for r in rows.values()) * 2.4), "per_kernel": rows},
              open(os.path.join(dst, "%s_mfma_util.json" % R), "w"), indent=1)
print("profiles written to", dst)
