#!/usr/bin/env python3
"""Stress of the in-launch split-K combine under UNEVEN load (cdna guide G16: "test every hand-off under uneven
load, consumer L1-warm, checking every word"): four streams run differently-shaped split-K convs concurrently and
repeatedly, each with its own workspace, while a fifth stream streams a large copy through the L2s; every output
is compared bit for bit with the result the same launch gave alone on an idle chip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from faster_rcnn_amd import ops

rs = np.random.RandomState(0)
shapes = [(1, 38, 63, 256, 256, 3, 322), (1, 38, 63, 1024, 256, 1, 422), (1, 38, 63, 512, 36, 1, 822), (300, 1, 1, 2048, 101, 1, 1622),
          (1, 38, 94, 1024, 512, 3, 322), (2, 19, 23, 512, 512, 3, 622)]
jobs = []
for n, h, w, cin, cout, k, tile in shapes:
    x = torch.from_numpy(rs.randn(n, h, w, cin).astype(np.float32)).cuda()
    wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    pc = ops.PackedConv(wt, np.ones(cout, np.float32), np.zeros(cout, np.float32))
    pad = "same" if k == 3 else "valid"
    ws = ops.ConvWorkspace()
    with ops.conv_workspace(ws):
        ref = ops.conv2d(x, pc, 1, pad, "relu", tile=tile).clone()
    jobs.append((x, pc, pad, tile, ws, ref))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in jobs]
noise_s = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
big2 = torch.empty_like(big)
bad = 0
t0 = time.time()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for r in range(rounds):
    outs = []
    with torch.cuda.stream(noise_s):
        big2.copy_(big)
    order = rs.permutation(len(jobs))
    for j in order:
        x, pc, pad, tile, ws, ref = jobs[j]
        with torch.cuda.stream(streams[j]), ops.conv_workspace(ws):
            for _ in range(1 + (r + j) % 3):                    # uneven: a different number of back-to-back launches per stream
                y = ops.conv2d(x, pc, 1, pad, "relu", tile=tile)
            outs.append((j, y))
    torch.cuda.synchronize()
    for j, y in outs:
        if not torch.equal(y, jobs[j][5]):
            bad += 1
            print("round", r, "job", j, "MISMATCH max", float((y - jobs[j][5]).abs().max()))
for j, (x, pc, pad, tile, ws, ref) in enumerate(jobs):
    assert int(ws.buf[:16384].sum().item()) == 0, "tickets not left zero"
print("rounds %d, %d launches checked, mismatches %d, %.1f s" % (rounds, rounds * len(jobs), bad, time.time() - t0))
sys.exit(1 if bad else 0)
