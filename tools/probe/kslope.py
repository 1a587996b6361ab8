"""Where a 3x3 layer's time goes: device time of the same spatial size and tile configuration at K = 9 * Cin for several Cin.
The slope is the cost of a K-step (64 input channels of one tap), the intercept the per-launch fixed cost (dispatch, prologue,
epilogue, store drain).  A throw-away topology built here; per-layer HIP events (each adds ~5 us, the no-op layers show it)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from yolo_tensorflow_amd import hip, darknet_io as IO

B = int(os.environ.get("B", "32"))


def conv(f, k, s=1, act="leaky", bn=1):
    return "[convolutional]\n%sfilters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n" % ("batch_normalize=1\n" if bn else "", f, k, s, act)


def net(grid, cins, cout, cfg):
    pools = {104: 2, 52: 3, 26: 4, 13: 5}[grid]
    t = "[net]\nwidth=416\nheight=416\nchannels=3\n\n" + conv(32, 3)
    for _ in range(pools):
        t += "[maxpool]\nsize=2\nstride=2\n\n"
    idx = []
    n = 1 + pools
    for cin in cins:
        t += conv(cin, 1); n += 1
        t += conv(cout, 3); idx.append(n); n += 1
    t += conv(255, 1, act="linear", bn=0)
    t += "[yolo]\nmask=0,1,2\nanchors=10,13, 16,30, 33,23\nclasses=80\nnum=3\n\n"
    return t, idx


CASES = ((26, (256, 512, 1024), 512, (40, 36, 16)), (52, (128, 256, 512), 256, (40, 36, 16)),
         (13, (512, 1024, 2048), 1024, (41, 43, 42, 23)), (104, (64, 128, 256), 128, (16, 41)))
ONLY = [int(v) for v in os.environ.get("CFGS", "").split(",") if v]
for grid, cins, cout, cfgs in CASES:
    cfgs = [c for c in cfgs if not ONLY or c in ONLY]
    txt, idx = net(grid, cins, cout, cfgs)
    secs = IO.parse_cfg(txt)
    eng = hip.Engine(txt, max_batch=B); eng.set_weights(IO.synth_weights(secs, 0))
    img = np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)
    eng.forward(img, want_detections=False)
    for cfg in cfgs:
        plan = np.full(eng.num_layers, -1, np.int32)
        for i in idx:
            plan[i] = cfg
        eng.set_tile_configs(plan)
        ms = np.median([eng.time_layers(B, 10) for _ in range(3)], axis=0)
        t = [ms[i] * 1e3 for i in idx]; ks = [9 * c // 64 for c in cins]
        slope = (t[-1] - t[0]) / (ks[-1] - ks[0]); icpt = t[0] - slope * ks[0]
        flop = 2.0 * B * grid * grid * cout * 64 * 1.0          # per K-step
        print("grid %3d cout %4d cfg %2d: %s us at K-steps %s -> %.3f us per K-step (%.0f TFLOP/s in the K loop), fixed %.1f us (incl. ~5 us of event)" % (
            grid, cout, cfg, ["%.1f" % v for v in t], ks, slope, flop / slope / 1e6, icpt))
    eng.close()
