"""Aggregates tools/probe/layer_table.py output (stdin) by layer class: us per class."""
import re, sys, collections
cls = collections.OrderedDict(); last = ""
for l in sys.stdin:
    if l.startswith("sum of layers"): last = l.strip()
    m = re.match(r'\s*(\d+) convolutional\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d)\s+(\d+)\s+(-?\d+)\s+([\d.]+)\s+([\d.]+)', l)
    if not m: continue
    i, h, w, c, k, cin, cfg, ms, tf = m.groups(); i = int(i); h = int(h); k = int(k); ms = float(ms)
    key = "fused" if (ms < 0.006 and (k == 1 or i == 0)) else "208" if h == 208 else "%d 3x3" % h if k == 3 else "1x1 %d" % h
    cls[key] = cls.get(key, 0.0) + ms * 1e3
print("  ".join("%s %.0f" % kv for kv in cls.items()), "|", last)
