#!/usr/bin/env python3
"""Which shader clock does the chip hold under the conv engine's load?  (dev diagnostic)

Runs scripts/micro/clock_probe.hip on a second stream: idle, beside the 128x128 fp32 MFMA conv
kernel on the detector-head 3x3 shape, and beside the bf16 kernel.  s_memtime counts core-clock
cycles, s_memrealtime a constant 100 MHz."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import ops

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "micro", "clock_probe.so")
if not os.path.exists(SO):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-O3",
                           os.path.join(HERE, "micro", "clock_probe.hip"), "-o", SO])
lib = ctypes.CDLL(SO)
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]


def probe(load=None, spin_us=3000.0, blocks=8):
    out = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    if load is not None:
        for _ in range(5):
            load()                       # queue ~ms of work first so the probe runs in the middle of it
    with torch.cuda.stream(side):
        lib.clock_probe(out.data_ptr(), blocks, spin_us, ctypes.c_void_p(side.cuda_stream))
    if load is not None:
        for _ in range(30):
            load()
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(blocks, 2)
    return float(np.median(o[:, 0] / o[:, 1] * 100.0))       # MHz


def main():
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(300, 7, 7, 512).astype(np.float32)).cuda()
    wt = (rs.randn(3, 3, 512, 512) * 0.02).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(512, np.float32), np.zeros(512, np.float32))
    pcb = ops.PackedConvBf16(wt, np.ones(512, np.float32), np.zeros(512, np.float32))
    xb = x.to(torch.bfloat16)
    y = ops.conv2d(x, pc, 1, "same", "relu")
    three = lambda load=None: "  ".join("%.0f" % probe(load) for _ in range(3))
    print("idle                         : %s MHz" % three())
    for tile in (21, 22, 42, 41):
        print("beside f32 conv tile %-3d     : %s MHz" % (tile, three(lambda: ops.conv2d(x, pc, 1, "same", "relu", out=y, tile=tile))))
    print("beside bf16 conv             : %s MHz" % three(lambda: ops.conv2d_bf16(xb, pcb, 1, "same", "relu")))
    burn_out = torch.zeros(2040 * 256, dtype=torch.float32, device="cuda")       # 2040 blocks: leave the probe a slot
    lib.mfma_burn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    burn = lambda: lib.mfma_burn(burn_out.data_ptr(), 2040, 150, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    print("beside register-only MFMA    : %s MHz" % three(burn))
    print("idle again                   : %s MHz" % three())


if __name__ == "__main__":
    main()
