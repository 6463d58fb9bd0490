#!/usr/bin/env python3
"""Step time of the two training configs at C2 shapes (ResNet-50, 600x1000, 9 anchors, 21 classes):
BASELINE configs[2] (RPN step 1, train_rpn_step1.py) and configs[4] (detector step 2, train_det_step2.py, 64 sampled
RoIs of <= 2000 proposals).  One image per GPU per step; with WORLD_SIZE > 1 every step all-reduces the flat
gradient buffer (RCCL through torch.distributed; FRCNN_BENCH_BACKEND=gloo lets two ranks share one GPU on a 1-GPU box).

    python scripts/bench_train.py [--bf16] [--steps K] [--warmup W] [--big-tiles]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train.py ...

Rank 0 prints ONE JSON line: per config the step time, images/s of the whole job, the gradient payload, the time of a
stand-alone all-reduce of that payload, and a `roofline` object (algorithmic FLOP of the step, SURVEY 8(d) / BASELINE.md 3,
over the measured step time, against the fp32 matrix peak -- or the bf16 one for the mixed-precision run)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from faster_rcnn_amd import dp, ops, resnet, train
from faster_rcnn_amd.weights import synthetic_resnet

H, W, A, C = 600, 1000, 9, 21
PEAK = {"f32": 157.3, "bf16": 2500.0}
# algorithmic GFLOP per step: forward of base + heads, backward (dgrad + wgrad ~ 2x forward) of the
# trainable part only (freeze_blocks=[1,2,3]: stage 4 (+ stage 5 + dense) and the heads)
GFLOP = {"rpn_step1": 97.9 + 2 * (33.9 + 22.7), "det_step2": (75.2 + 93.7) + 2 * (33.9 + 93.7)}


def timed(fn, steps, warmup):
    """fn() may return a train.PendingLosses (deferred step): it is read one step late, like train_util._LossLog does;
    every step's losses are read inside the timed region."""
    prev = None
    for _ in range(warmup):
        cur = fn()
        if hasattr(prev, "result"):
            prev.result()
        prev = cur
    if hasattr(prev, "result"):
        prev.result()
    prev = None
    torch.cuda.synchronize()
    if dp.world() > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        cur = fn()
        if hasattr(prev, "result"):
            prev.result()
        prev = cur
    if hasattr(prev, "result"):
        prev.result()
    train.finish_pending_updates()
    torch.cuda.synchronize()
    if dp.world() > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if dp.world() > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    return 1e3 * el / steps


def allreduce_ms(params, reps=10):
    """One stand-alone all-reduce of the flat gradient buffer (the step's only collective)."""
    if dp.world() == 1:
        return 0.0
    buf = torch.zeros_like(params.g)
    return timed(lambda: dp.allreduce_sum_(buf), reps, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true", help="mixed precision: bf16 activations / gradients / packed filters, f32 masters")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=40, help="untimed steps first: ~0.1 s of work, enough for the clocks to leave their idle state (with 3 the first config timed read 2.6-4.6 ms from box to box)")
    ap.add_argument("--big-tiles", action="store_true")
    ap.add_argument("--only", choices=("rpn", "det"), default=None)
    ap.add_argument("--no-split-k", action="store_true", help="dev: plain conv launches only (no split-K workspace) in the training steps")
    ap.add_argument("--through-loop", action="store_true", help="also time train_util.train_rpn / train_detector_step2 THEMSELVES over 32 distinct images "
                    "(image fetch, targets / proposals, host sampling, step, loss line): ms per iteration beside the bare step, with the managers' "
                    "device-resident feed and (fewer iterations) with the reference's host-numpy calls")
    ap.add_argument("--no-host-feed", action="store_true", help="with --through-loop: only the device-resident feed (the loops' default), not the reference's host-numpy calls")
    ap.add_argument("--sync-each-step", action="store_true", help="time only the plain Keras call (losses read back after every step); by default "
                    "the loop reads them one step late, the way train_util's loops do, and the per-step figure is reported beside it")
    args = ap.parse_args()
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")             # ONE JSON line on stdout: RCCL's version banner and anything else written
    os.dup2(2, 1)                                    # to fd 1 goes to stderr from here on
    backend = os.environ.get("FRCNN_BENCH_BACKEND")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > 1 and backend == "gloo":
        torch.cuda.set_device(0)                       # dev: several ranks on the one GPU of this box
    rank, world = dp.init_from_env(backend=backend)
    if world == 1:
        torch.cuda.set_device(0)
    if args.big_tiles:
        ops.AUTO_TILE = 50
    DT = "bf16" if args.bf16 else "f32"
    rs = np.random.RandomState(rank)
    x = (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]
    rows, cols = resnet.get_conv_rows_cols(H, W)
    out = {"world": world, "dtype": DT, "backend": (torch.distributed.get_backend() if world > 1 else None),
           "workload": "ResNet-50 600x1000, 1 image per GPU per step, SGD momentum 0.9, l2 1e-4, synthetic data",
           "losses_read": "after every step" if args.sync_each_step else "one step late (train_util's loops), all inside the timed region",
           "step_launch": ("hipGraph replay: prefix / forward / backward pieces / weight-gradient batches as linear graphs on three streams (train._StepGraph), "
                           "optimiser + re-pack eager" if train.STEP_GRAPHS else "eager launches (FRCNN_TRAIN_GRAPH=0)")}

    def report(tag, ms, params, ms_sync=None, step=None):
        ar = allreduce_ms(params)
        tf = GFLOP[tag] * world / ms                   # GFLOP / ms = TFLOP/s, whole job
        out[tag] = {"ms_per_step": round(ms, 3), "img_s": round(world * 1e3 / ms, 2), "grad_payload_MB": round(params.total * 4 / 1e6, 1),
                    "allreduce_ms": round(ar, 3), "allreduce_share": round(ar / ms, 4),
                    "roofline": {"bound": "mfma", "achieved": round(tf / world, 2), "peak": PEAK[DT], "unit": "TFLOP/s per GPU",
                                 "frac": round(tf / world / PEAK[DT], 4), "gflop_per_step": round(GFLOP[tag], 1)}}
        if ms_sync is not None:
            out[tag]["ms_per_step_losses_read_every_step"] = round(ms_sync, 3)
        if world > 1 and step is not None:
            # the same steps with the collective left out: what the all-reduce adds to a step once the next image's host
            # staging, upload and frozen stages run beside it (train._StepDriver._update / _finish_update)
            train.DP_SYNC = False
            try:
                ms_local = timed(step, args.steps, 1)
            finally:
                train.DP_SYNC = True
            out[tag]["ms_per_step_without_allreduce"] = round(ms_local, 3)
            out[tag]["exposed_allreduce_ms"] = round(max(ms - ms_local, 0.0), 3)

    if args.only in (None, "rpn"):
        w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)
        base = resnet.resnet50_base(weights=w, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
        rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
        can_use = rs.rand(1, rows, cols, A) < 0.012
        is_pos = rs.rand(1, rows, cols, A) < 0.01
        y_class = np.concatenate([can_use, is_pos], axis=3)
        y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32),
                                  (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
        rpn.compile(train.SGD(1e-3, 0.9))
        if args.no_split_k:
            rpn._trainer._conv_ws = rpn._trainer._conv_ws_prefix = ops.NO_SPLIT_K
        step = lambda: rpn.train_on_batch(x, [y_class, y_bbreg], defer=not args.sync_each_step)
        ms = timed(step, args.steps, args.warmup)
        ms_sync = None if args.sync_each_step else timed(lambda: rpn.train_on_batch(x, [y_class, y_bbreg]), args.steps, 2)
        report("rpn_step1", ms, rpn._trainer.params, ms_sync, step)
        del rpn, base
    if args.only in (None, "det"):
        dw = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2)
        dbase = resnet.resnet50_base(weights=dw, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype=DT)
        det = resnet.resnet50_classifier(64, C, dbase)
        n = 64
        x1 = rs.randint(0, cols - 8, n)
        y1 = rs.randint(0, rows - 8, n)
        rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, n), y1 + 1 + rs.randint(0, 7, n)], axis=1).astype(np.float32)[None]
        ci = rs.randint(0, C, n)
        yc = np.zeros((1, n, C), np.float32)
        yc[0, np.arange(n), ci] = 1
        lab = np.zeros((n, 4 * (C - 1)), np.float32)
        tg = np.zeros((n, 4 * (C - 1)), np.float32)
        for i, c in enumerate(ci):
            if c < C - 1:
                lab[i, 4 * c:4 * c + 4] = 1
                tg[i, 4 * c:4 * c + 4] = rs.randn(4)
        yb = np.concatenate([lab, tg], axis=1)[None]
        det.compile(train.SGD(1e-3, 0.9))
        if args.no_split_k:
            det._trainer._conv_ws = det._trainer._conv_ws_prefix = ops.NO_SPLIT_K
        step = lambda: det.train_on_batch([x, rois], [yc, yb], defer=not args.sync_each_step)
        ms = timed(step, args.steps, args.warmup)
        ms_sync = None if args.sync_each_step else timed(lambda: det.train_on_batch([x, rois], [yc, yb]), args.steps, 2)
        report("det_step2", ms, det._trainer.params, ms_sync, step)
    if args.through_loop and world == 1:
        import bench
        for tag in ("rpn_step1", "det_step2"):
            if tag in out:
                fast = bench.train_loop_leg(tag, DT, iterations=max(32, args.steps), fast=True, height=H, width=W)
                fast["bare_step_over_loop_iteration"] = round(out[tag]["ms_per_step"] / fast["ms_per_iteration"], 3)
                out[tag]["through_loop"] = {"fast_feed": fast}
                if not args.no_host_feed:
                    out[tag]["through_loop"]["host_feed"] = bench.train_loop_leg(tag, DT, iterations=12, warm=6, fast=False, height=H, width=W)
    # flat keys kept for the round-1 readers of this line
    for tag, short in (("rpn_step1", "rpn_step1"), ("det_step2", "det_step2")):
        if tag in out:
            out[short + "_ms"] = out[tag]["ms_per_step"]
            out[short + "_img_s"] = out[tag]["img_s"]
            out[short + "_params_MB"] = out[tag]["grad_payload_MB"]
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
