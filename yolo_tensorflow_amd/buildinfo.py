"""Identity of the code + tile plan a measurement was taken on (used to tie profiles/*_hbm_traffic.json to bench.py)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_hash(size, batch, dtype):
    """sha1 over everything that defines a measurement: csrc/ (kernels, host planner AND the Makefile with its arch / -O / -ffp-contract
    flags), the committed tile plan for (size, batch, dtype), and the two programs that define the measured workload (bench.py and the
    PMC driver tools/prof_forward.py)."""
    h = hashlib.sha1()
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(_HERE, "csrc", "Makefile"))
    files.append(os.path.join(_HERE, "tuned", "yolov3_%d_b%d_%s.json" % (size, batch, dtype)))
    files.append(os.path.join(root, "bench.py"))
    files.append(os.path.join(root, "tools", "prof_forward.py"))
    for f in files:
        h.update(os.path.basename(f).encode())
        if os.path.exists(f):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
