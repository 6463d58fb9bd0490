#!/usr/bin/env python3
"""Does JPEG decoding scale with threads on this box?  (PIL releases the GIL inside its decoder; voc_dets' decode threads measured SLOWER
than inline decoding on the GPU boxes.)  Pure host: no GPU call."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "VOC_test", "JPEGImages", "000005.jpg")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cgroup cpu.max: n/a", e)
def dec(_):
    return np.asarray(Image.open(path).convert("RGB")).shape
n = 200
for nt in (1, 2, 4, 8):
    t0 = time.perf_counter()
    if nt == 1:
        for i in range(n): dec(i)
    else:
        with ThreadPoolExecutor(nt) as ex: list(ex.map(dec, range(n)))
    dt = time.perf_counter() - t0
    print("threads %d: %.2f ms per frame, %.0f frames/s" % (nt, dt / n * 1e3, n / dt))
