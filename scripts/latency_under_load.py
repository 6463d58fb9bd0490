#!/usr/bin/env python3
"""Dependent-load latency (pointer chase, one lane) idle and beside the inference pipeline's graphs: configs[1] fp32 with
8 images in flight, configs[3] bf16 with 4.  Two working sets: 2 MB (L2 / Infinity-Cache resident) and 512 MB (HBM).
Dev diagnostic for DESIGN 11 (why short dependent launches stretch when several images are in flight)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "micro", "lat_probe.so")
if not os.path.exists(SO):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-O3", os.path.join(HERE, "micro", "lat_probe.hip"), "-o", SO])
lib = ctypes.CDLL(SO)
lib.lat_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]


def chain(n_lines, seed=0):
    perm = np.random.RandomState(seed).permutation(n_lines).astype(np.uint32)
    nxt = np.empty(n_lines, np.uint32)
    nxt[perm] = np.roll(perm, -1)                            # one cycle through every line
    buf = np.zeros((n_lines, 32), np.uint32)
    buf[:, 0] = nxt
    return torch.from_numpy(buf).cuda()


def probe(buf, load=None, hops=4000, blocks=4):
    out = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    if load is not None:
        for _ in range(3):
            load()
    with torch.cuda.stream(side):
        lib.lat_probe(buf.data_ptr(), hops, out.data_ptr(), blocks, ctypes.c_void_p(side.cuda_stream))
    if load is not None:
        for _ in range(12):
            load()
    torch.cuda.synchronize()
    t = out.cpu().numpy().reshape(blocks, 2)[:, 0]
    return float(np.median(t)) * 10.0 / hops                # ns per hop (100 MHz ticks)


def main():
    import bench
    small, big = chain(2 * 1024 * 1024 // 128), chain(512 * 1024 * 1024 // 128)
    print("idle: %.0f ns per dependent load (2 MB set), %.0f ns (512 MB set)" % (probe(small), probe(big)))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    from faster_rcnn_amd.pipeline import InferencePipeline
    for cfg, S in (("c2", 8), ("c4", 4)):
        bench.HEIGHT, bench.WIDTH, bench.DEPTH, bench.DTYPE = 600, 1000, 50, "f32"
        bench.select_config(cfg)
        pipe, _, anchors = bench.build_pipeline()
        pipes = [pipe] + [InferencePipeline(pipe.rpn, pipe.det, anchors, max_proposals=bench.PROPOSALS) for _ in range(S - 1)]
        for i, pl in enumerate(pipes):
            pl.capture(bench.HEIGHT, bench.WIDTH, split_k=(cfg == "c2"), throughput=True)
            pl._static_in.copy_(torch.from_numpy(bench.synth_image(i)).cuda())
        streams = [torch.cuda.Stream() for _ in range(S)]
        torch.cuda.synchronize()

        def load():
            for pl, st in zip(pipes, streams):
                with torch.cuda.stream(st):
                    pl._graph.replay()
        print("%s, %d images in flight: %.0f ns (2 MB set), %.0f ns (512 MB set)" % (cfg, S, probe(small, load), probe(big, load)))
        del pipes

if __name__ == "__main__":
    main()
