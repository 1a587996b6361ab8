"""Identity of the code + tile plan a measurement was taken on (used to tie profiles/*_hbm_traffic.json to bench.py)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_hash(size, batch, dtype):
    """sha1 over csrc/ (kernels + host planner) and the committed tile plan for (size, batch, dtype)."""
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(_HERE, "tuned", "yolov3_%d_b%d_%s.json" % (size, batch, dtype)))
    for f in files:
        h.update(os.path.basename(f).encode())
        if os.path.exists(f):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
