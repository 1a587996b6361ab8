"""In-kernel clock and K-loop rate of the conv kernel per layer class (MI355X_MICROARCH.md 'DVFS give-back' item 6):
a diagnostic build stamps s_memtime / s_memrealtime ONCE around the K loop (YOLO_CONV_DIAG_LIGHT: nothing is stamped inside it), the same
launch is repeated back to back for >= 2 s on random operands of the real layer shape, and the stamps of the last launch are read.
  clock  = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the waves of the launch
  K loop = cycles per K-step (loop cycles / K-steps), and the PFLOP/s that gives for 256 CUs at that clock
Writes profiles/<round>_kloop_clock.json.   usage: python tools/kloop_clock.py r04"""
import json, os, re, sys, tempfile, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from yolo_tensorflow_amd import hip

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
SECONDS = float(os.environ.get("SECONDS_PER_SHAPE", "2.5"))
rng = np.random.default_rng(0)
# (form, YOLO_CONV_DIAG value, [(n, h, cin, cout)], waves x tile of the stamped build, workgroups per CU)
CASES = (("free-running halo f176c256 (the tuned plan's 52x52 / 26x26 kernel)", "free", ((32, 26, 256, 512), (32, 52, 128, 256)), (169, 256), 1),
         ("tiled p176c128_s2 (104x104 and stride-2 layers; 13x13 shape for reference)", "1", ((32, 104, 64, 128), (32, 26, 256, 512), (32, 13, 512, 1024)), (176, 128), 2))
out = {"method": "s_memtime / s_memrealtime around the K loop of a diagnostic build (no stamps inside the loop), >= %.1f s of back-to-back launches per shape, random operands, median over waves" % SECONDS,
       "peak_clock_mhz": 2400, "rows": []}


def run(mode, shape, reps, light=True):
    n, h, cin, cout = shape
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    r = rng.standard_normal((n, h, h, cout)).astype(np.float32)
    os.environ["YOLO_CONV_DIAG"] = mode; os.environ["YOLO_CONV_DIAG_REPS"] = str(reps)
    if light: os.environ["YOLO_CONV_DIAG_LIGHT"] = "1"
    else: os.environ.pop("YOLO_CONV_DIAG_LIGHT", None)
    with tempfile.TemporaryFile() as tf:
        sys.stderr.flush(); saved = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            t0 = time.time(); hip.op_conv2d(x, w, None, act=1, residual=r, dtype=hip.BF16); dt = time.time() - t0
        finally:
            os.dup2(saved, 2); os.close(saved)
        tf.seek(0); txt = tf.read().decode()
    m = re.search(r"KT (\d+) .*loop total/KT (\d+)\).*shader clock (\d+) MHz \(K loop alone (\d+) MHz, median over waves (\d+)\)", txt)
    if not m:
        raise RuntimeError("no diag line:\n" + txt)
    return dict(kt=int(m.group(1)), cyc_per_kstep=int(m.group(2)), clock_kernel=int(m.group(3)), clock_kloop_mean=int(m.group(4)), clock_kloop=int(m.group(5)), wall_s=dt, text=txt)


for name, mode, shapes, (tp, tc), wg_per_cu in CASES:
    for shape in shapes:
        probe = run(mode, shape, 200)
        per_launch = max(probe["wall_s"] / 200.0, 2e-5)          # (includes the upload: an upper bound, so reps is a lower bound on the time)
        reps = int(max(2000, min(200000, SECONDS / 5e-5)))         # launches of 30-80 us: >= SECONDS of device time
        r = run(mode, shape, reps)
        fine = run(mode, shape, 2000, light=False)                 # the per-phase stamps, for the issue / wait / MFMA split (costs ~11 % cycles)
        flop_step = 2.0 * tp * tc * 64                             # per workgroup and K-step (algorithmic pixels of the tile x channels x 64 of K)
        pf = 256 * wg_per_cu * flop_step / (r["cyc_per_kstep"] / (r["clock_kloop"] * 1e6)) / 1e15
        row = {"form": name, "shape": "N%d %dx%d %d->%d 3x3" % (shape[0], shape[1], shape[1], shape[2], shape[3]), "launches": reps,
               "k_steps": r["kt"], "cycles_per_k_step": r["cyc_per_kstep"], "in_kernel_clock_mhz_k_loop": r["clock_kloop"],
               "in_kernel_clock_mhz_whole_kernel": r["clock_kernel"], "k_loop_pflops_at_that_clock": round(pf, 3),
               "k_loop_frac_of_2p5_pflops": round(pf / 2.5, 3), "workgroups_per_cu": wg_per_cu,
               "fine_stamps_cycles_per_k_step": fine["cyc_per_kstep"], "fine_stamps_line": fine["text"].strip().split("\n")[0]}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_kloop_clock.json" % rnd), "w"), indent=1)
