#!/usr/bin/env python3
"""Board power and clocks (rocm-smi) while the conv engine runs one launch form in a loop.  Dev diagnostic:
is the fp32 MFMA kernel's ~120 TFLOP/s a power ceiling?"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.split(":", 1)[1].strip() if ":" in l else l for l in txt.splitlines() if "Power" in l or "sclk" in l or "junction" in l.lower()]
            out.append(" | ".join(k[-60:] for k in keep))
        except Exception as e:                                   # rocm-smi missing or not permitted
            out.append("rocm-smi: %r" % (e,))
            return
        time.sleep(0.3)


def run(tag, f, seconds=4.0):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            f()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    print("== %s: %.1f us per launch over %.1f s" % (tag, e0.elapsed_time(e1) / n * 1e3, seconds))
    for l in out[:: max(1, len(out) // 6)]:
        print("   ", l)


def main():
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(300, 7, 7, 512).astype(np.float32)).cuda()
    wt = (rs.randn(3, 3, 512, 512) * 0.02).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(512, np.float32), np.zeros(512, np.float32))
    pcb = ops.PackedConvBf16(wt, np.ones(512, np.float32), np.zeros(512, np.float32))
    xb = x.to(torch.bfloat16)
    y = ops.conv2d(x, pc, 1, "same", "relu")
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start(); time.sleep(1.0); stop.set(); th.join()
    print("== idle")
    for l in out[:2]:
        print("   ", l)
    run("fp32 128x128 (tile 21)", lambda: ops.conv2d(x, pc, 1, "same", "relu", out=y, tile=121))
    run("fp32 64x64 (tile 22)", lambda: ops.conv2d(x, pc, 1, "same", "relu", out=y, tile=122))
    run("bf16 (auto)", lambda: ops.conv2d_bf16(xb, pcb, 1, "same", "relu"))


if __name__ == "__main__":
    main()
