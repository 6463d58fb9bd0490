#!/usr/bin/env python3
"""Does work on stream B that waits for an event recorded on stream A right after a hipGraph launch start when that graph ends -- or
later?  A: graph GA (20 kernels), [event], graph GB (20 kernels).  B: wait(event), 5 kernels.  Kernel trace shows where B's kernels land.
Variants: the event recorded after an eager kernel instead; B's work as a graph.   usage: graph_event_probe.py <variant>"""
import sys, time
import torch
variant = sys.argv[1] if len(sys.argv) > 1 else "graph"
dev = "cuda"
a = torch.zeros(8 << 20, device=dev)
b = torch.zeros(8 << 20, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cap = torch.cuda.Stream()


def chain(t, n):
    for _ in range(n):
        t.add_(1.0)


def capture(t, n):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="thread_local")
        chain(t, n)
        g.capture_end()
    return g


chain(a, 3); chain(b, 3)
GA, GB, GC = capture(a, 20), capture(a, 20), capture(b, 5)
torch.cuda.synchronize()
ev = torch.cuda.Event()
for it in range(6):
    with torch.cuda.stream(sA):
        if variant.startswith("eager"):
            chain(a, 20)
        else:
            GA.replay()
        if "tick" in variant:
            a[:1].add_(0.0)
        ev.record(sA)
    sB.wait_event(ev)
    with torch.cuda.stream(sB):
        if "bgraph" in variant:
            GC.replay()
        else:
            b.mul_(1.0); b.mul_(1.0); b.mul_(1.0); b.mul_(1.0); b.mul_(1.0)
    with torch.cuda.stream(sA):
        if variant.startswith("eager"):
            chain(a, 20)
        else:
            GB.replay()
    torch.cuda.synchronize()
    time.sleep(0.01)
print("done", variant)
