"""CPU study (oracle only): min IoU of the decoded boxes against the fp32 oracle as a function of the SIGNIFICAND WIDTH of the stored
activations and folded filters (8 = bf16, 11 = fp16), plain (`dev`) and mean-centred (`cen`, tools/study/study_centred.py), on the synthetic
weights with a trained file's batch-norm statistics.  Result (profiles/r04_precision_study.txt): 1 - IoU falls by 4x per 2 bits; 0.999
needs >= 17 bits -- no 16-bit storage type reaches it on this network, centred or not.
Usage: study_bits.py [log|real]   (real = the reference's real batch-norm vectors, tests/golden/yolov3_bn_real.npz; profiles/r05_precision_study.txt)"""
import glob, os, sys
import numpy as np
_H = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.join(_H, "..", ".."), os.path.join(_H, "..", "..", "tests"), _H]
from oracle import yolo_ref as R
from yolo_tensorflow_amd import darknet_io as IO
import study_centred as S
from test_gpu_tuned import box_deviation
from PIL import Image
def qbits(n):
    def q(x):
        x = np.asarray(x, np.float32); m, e = np.frexp(x)
        return np.ldexp(np.round(m * (1 << n)) / (1 << n), e).astype(np.float32)
    return q
txt = IO.cfg_text("yolov3"); secs = R.parse_cfg(txt)
paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "images", "*.jpg")))
imgs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
x_all = np.concatenate([R.input_process(im, 416) for im in imgs])
noise = np.random.default_rng(5).random((2, 416, 416, 3), dtype=np.float32)
STATS = sys.argv[1] if len(sys.argv) > 1 else "log"        # "log": drawn from the dump's ranges (round 4); "real": the dump's own vectors (round 5)
flat = IO.synth_weights(IO.parse_cfg(txt), seed=3, stats=STATS, obj_bias=-2.5)
params = R.unflatten_weights(flat, secs)
R.calibrate_bn_statistics(secs, params, np.concatenate([x_all, noise]), seed=3, keep_var=STATS == "real")
print("== study_bits, stats=%s" % STATS, flush=True)
x = x_all[:3]
h32 = R.forward(secs, params, x)[0]; ref = R.yolo_v3_detections(h32, 416, ratio=True)
oc = S.calib_offsets(secs, params, np.concatenate([x_all[3:5], noise[:1]]))
zero = [None if o is None else np.zeros_like(o) for o in oc]
for n in (8, 11, 13, 15, 17, 19):
    for nm, offs in (("dev", zero), ("cen", oc)):
        hs = S.forward_centred(secs, params, x, qbits(n), offs)
        det = R.yolo_v3_detections(hs, 416, ratio=True)
        rel = [float(np.sqrt(((a[1] - b[1]) ** 2).mean()) / np.sqrt((b[1] ** 2).mean())) for a, b in zip(hs, h32)]
        miou, mds, cnt, lost = box_deviation(ref, det, 1e-2, thr=0.4)
        print("significand %2d bits %s: head rel rms %.5f %.5f %.5f  min IoU %.4f  max|ds| %.4f lost %d" % (n, nm, rel[0], rel[1], rel[2], miou, mds, lost), flush=True)
