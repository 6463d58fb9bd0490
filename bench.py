#!/usr/bin/env python3
"""Headline benchmark: images/s, end-to-end (RPN + detector) Faster R-CNN inference,
ResNet-50, 600x1000 synthetic image, anchor scales 128/256/512, 300 proposals, 21 classes,
fp32 (BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ...` (RANK / WORLD_SIZE in the
environment) this process IS a rank; started plainly as `python bench.py --gpus N`, it starts the N ranks itself
(fresh child processes, before this process has touched the GPU), relays rank 0's JSON line and exits non-zero if any
rank fails.  Images shard by rank with NO data-path collective (inference has no exchange step), so the scaling is
weak: every rank processes its own image stream.  The N > 1 line also carries `train_dp`: a few data-parallel RPN
step-1 training steps (BASELINE configs[2], one image per GPU) whose flat gradient buffer goes through the path's
ONE collective -- an RCCL all-reduce -- so a scaling run shows what RCCL saw and what the collective costs.

A "step" = S images (--streams, default 8 on 8 hardware queues for fp32: one hipGraph + HIP stream per image in flight)
through the whole device-resident path (backbone convs, RPN heads,
decode, top-8000 ordering, NMS to 300, RoI crop-resize, stage-5 head, softmax, detection
post-process), replayed from a hipGraph.  The input is resident in HBM when timing starts.

The JSON line also carries
  roofline      the MFMA conv kernel: algorithmic FLOP of every conv launch of one image /
                their summed HIP-event durations, against the 157.3 TFLOP/s fp32 matrix peak.
  cpu_baseline  the oracle's torch-CPU restatement of the reference graph ("port"), timed on
                this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MATRIX_TFLOPS = 157.3        # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_TFLOPS = 2500.0             # dense bf16 MFMA peak (spec)
HEIGHT, WIDTH = 600, 1000
SCALES = [128, 256, 512]
NUM_CLASSES = 21
PROPOSALS = 300
DEPTH, DTYPE = 50, "f32"
HOIST = True
WORKLOAD = "configs[1]: ResNet-50, 600x1000, anchor_scales 128/256/512, RPN + detector inference, fp32"


def select_config(name):
    """c2 (default) = BASELINE configs[1], the headline.  c4 = configs[3] (ResNet-101, KITTI 600x1500,
    6 anchor scales, 10 classes, bf16 conv + fp32 NMS): a parity-test config, measurable on request."""
    global HEIGHT, WIDTH, SCALES, NUM_CLASSES, DEPTH, DTYPE, WORKLOAD
    if name == "c4":
        HEIGHT, WIDTH, SCALES, NUM_CLASSES, DEPTH, DTYPE = 600, 1500, [16, 32, 64, 128, 256, 512], 10, 101, "bf16"
        WORKLOAD = "configs[3]: ResNet-101, KITTI 600x1500, anchor_scales 16-512, RPN + detector inference, bf16 conv + fp32 NMS"
    if name == "c1":
        DEPTH = 16                                                   # VGG16
        WORKLOAD = "configs[0]: VGG16, 600x1000, RPN-only forward (vgg16_base + vgg16_rpn, the train_rpn_test.py path), fp32"


class RpnOnlyPipeline:
    """configs[0]: backbone + RPN heads only (what rpn_model.predict_on_batch runs, det_util.py:41); same capture / replay
    surface as InferencePipeline."""

    def __init__(self, rpn, batch=1):
        self.rpn, self._graph, self.batch = rpn, None, int(batch)

    def forward_dev(self, x, resize_ratio=1.0):
        from faster_rcnn_amd import ops
        ops.amax_begin()
        cls, reg, feat = self.rpn.forward_dev(x)
        return {"rpn_cls": cls, "rpn_reg": reg, "feat": feat}

    def capture(self, height, width, split_k=True, throughput=False, f32_engine="native"):
        from faster_rcnn_amd import ops
        self._static_in = torch.zeros((self.batch, height, width, 3), dtype=torch.float32, device="cuda")
        from faster_rcnn_amd.pipeline import no_gc
        self._conv_ws = ops.ConvWorkspace() if split_k else ops.NO_SPLIT_K
        self._amax = ops.AmaxArena() if f32_engine == "f16x3" else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), ops.f32_engine(f32_engine), ops.amax_arena(self._amax):
            for _ in range(2):
                self.forward_dev(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(self._graph, capture_error_mode="thread_local"), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), \
                ops.f32_engine(f32_engine), ops.amax_arena(self._amax):
            self._static_out = self.forward_dev(self._static_in)
        return self


def synth_image(seed):
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, (HEIGHT, WIDTH, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68])
    return img[None].astype(np.float32)


def build_pipeline(calibrate=True):
    from faster_rcnn_amd import resnet, util
    from faster_rcnn_amd.pipeline import InferencePipeline
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors(SCALES)
    if DEPTH == 16:
        from faster_rcnn_amd import vgg
        from faster_rcnn_amd.weights import synthetic_vgg16
        w = synthetic_vgg16(anchors_per_loc=len(anchors), seed=1, with_classifier=False)
        rpn = vgg.vgg16_rpn(vgg.vgg16_base(weights=w), include_conv=True, anchors_per_loc=len(anchors))
        return RpnOnlyPipeline(rpn), w, anchors
    w = synthetic_resnet(DEPTH, anchors_per_loc=len(anchors), num_classes=NUM_CLASSES, seed=1)
    base = (resnet.resnet50_base if DEPTH == 50 else resnet.resnet101_base)(weights=w, dtype=DTYPE)
    rpn = resnet.resnet50_rpn(base, include_conv=True, anchors_per_loc=len(anchors))
    det = (resnet.resnet50_classifier if DEPTH == 50 else resnet.resnet101_classifier)(PROPOSALS, NUM_CLASSES, weights=w, dtype=DTYPE)
    det.head.hoist = HOIST
    pipe = InferencePipeline(rpn, det, anchors, max_proposals=PROPOSALS)
    if calibrate and torch.cuda.is_available():
        calibrate_head(pipe, w)
    return pipe, w, anchors


def calibrate_head(pipe, w):
    """Random-init `dense_class` puts every RoI of every image in ONE class (the pooled features share a large common
    component), so detection-set comparisons, per-class NMS and the pair's mAP would exercise a single class.
    One eager pass over a calibration frame gives the class probabilities; weights.calibrate_classifier re-centres and
    re-scales the layer (a linear re-parametrisation) so that many classes fire.  The new arrays go into the SHARED weight
    dict: the oracle, every pipeline and every checker built from `w` see the same layer.  Same shapes, same launches."""
    from faster_rcnn_amd.weights import calibrate_classifier
    out = pipe.forward_dev(torch.from_numpy(synth_image(99)).cuda())
    n = int(out["n_rois"].item())
    probs = out["cls"][:n].float().cpu().numpy()
    name = "dense_class_%d" % NUM_CLASSES
    pipe.raw_dense_class = [np.array(a) for a in w[name]]           # the drawn layer, for frames unlike the calibration frame (e2e_parity)
    pipe.det.get_layer(name).set_weights(calibrate_classifier(w, NUM_CLASSES, probs))
    assert pipe.det.weights is w


def conv_roofline(pipe, x, reps=10, split_k=True, throughput=False, images=1, engine="native"):
    """Per-launch duration of every conv launch of one image, measured with HIP events on the launch
    stream.  An event pair around ONE short kernel also measures the event packets themselves
    (tens of microseconds on this stack), so each distinct launch (kernel instantiation x shape) is
    captured `reps` times back to back into a hipGraph -- the way the timed pipeline issues it -- and one
    event pair brackets a replay; its average is the launch duration.  Returns the roofline object for the
    DOMINANT kernel instantiation (largest summed duration per image) plus the aggregate over all conv launches."""
    from faster_rcnn_amd import ops
    from faster_rcnn_amd.pipeline import no_gc
    with ops.conv_workspace(None if split_k else ops.NO_SPLIT_K), ops.tile_policy(throughput), ops.f32_engine(engine):     # the launch forms the timed graphs hold
        pipe.forward_dev(x)
        torch.cuda.synchronize()
        ops.CONV_PROFILE = []
        pipe.rpn.base.net(x)                        # the base network on its own: which launches are "the backbone"
        torch.cuda.synchronize()
        base_prof, ops.CONV_PROFILE = ops.CONV_PROFILE, []
        pipe.forward_dev(x)
        torch.cuda.synchronize()
    prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    n_base = len(base_prof)
    # the pass starts with exactly those launches (same kernels, same shapes): anything else would shift FLOP and time
    # between "backbone" and "head" silently
    assert [(r["kernel"],) + r["shape"] for r in prof[:n_base]] == [(r["kernel"],) + r["shape"] for r in base_prof], "backbone launches are not the head of the pass"
    groups = {}
    for rec in prof:
        g = groups.setdefault((rec["kernel"],) + rec["shape"], {"count": 0, "rec": rec})
        g["count"] += 1
    per_kernel = {}
    tot_flops = tot_ms = 0.0
    for key, g in groups.items():
        rec = g["rec"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        graph = torch.cuda.CUDAGraph()          # the product path replays hipGraphs: time the launch the same way
        with no_gc(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(reps):
                rec["relaunch"]()
        graph.replay()                          # the shader clock settles over the first launches of a shape
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        g["ms"] = e0.elapsed_time(e1) / reps
        a = per_kernel.setdefault(rec["kernel"], [0.0, 0.0, 0])
        a[0] += rec["flops"] * g["count"]; a[1] += g["ms"] * g["count"]; a[2] += g["count"]
        tot_flops += rec["flops"] * g["count"] / images          # "per image" figures: a pass is `images` images (batched pipeline)
        tot_ms += g["ms"] * g["count"] / images
    # the backbone alone (conv1 .. res4f: the layers the north star's ">= 60 % of the MFMA roofline on the ResNet-50
    # backbone conv" speaks of): the first launches of the pass, one per ConvUnit of the base network
    base_flops = sum(rec["flops"] for rec in prof[:n_base]) / images
    base_ms = sum(groups[(rec["kernel"],) + rec["shape"]]["ms"] for rec in prof[:n_base]) / images
    dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1][1])
    achieved = dom[0] / (dom[1] * 1e-3) / 1e12
    heavy_key, heavy = max(((k, v) for k, v in groups.items() if k[0] == dom_name), key=lambda kv: kv[1]["ms"] * kv[1]["count"])
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")       # PMC passes (rocprofv3 --pmc), committed
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(dom_name, {}).get("hbm_bytes_per_launch")
    # a split engine's ceiling is the 16-bit matrix pipe: six v_mfma_f32_32x32x16_bf16 (bf16x6) or three v_mfma_f32_32x32x16_f16
    # (f16x3) matrix instructions per block of fp32 products
    x6_dom, h3_dom = "x6" in dom_name, "h3" in dom_name
    per_product = 6.0 if x6_dom else 3.0 if h3_dom else None
    peak = PEAK_BF16_TFLOPS / per_product if per_product else PEAK_F32_MATRIX_TFLOPS
    engine_peak = PEAK_BF16_TFLOPS / {"bf16x6": 6.0, "f16x3": 3.0}[engine] if engine in ("bf16x6", "f16x3") else None

    def both(tf):
        """A rate against BOTH denominators: the native fp32 matrix peak the north star was written against, and the
        ceiling of the engine the launches actually ran on (most of them: the stem and a few small grids stay native)."""
        d = {"frac": round(tf / PEAK_F32_MATRIX_TFLOPS, 4), "peak": PEAK_F32_MATRIX_TFLOPS, "peak_basis": "native fp32 matrix peak (v_mfma_f32_32x32x2_f32)"}
        if engine_peak:
            d["frac_of_engine_ceiling"] = round(tf / engine_peak, 4)
            d["engine_ceiling"] = round(engine_peak, 1)
            d["engine_ceiling_basis"] = "2.5 PFLOP/s dense 16-bit MFMA / %d matrix instructions per block of fp32 products (%s)" % (6 if engine == "bf16x6" else 3, engine)
        return d
    roof = {
        "bound": "mfma", "kernel": dom_name,
        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic,
        "traffic_source": "profiles/traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch of this kernel, from the committed rocprofv3 --pmc "
                          "passes of scripts/profile_round5.sh (counters cannot be read from inside this process; null if the kernel is not in the file)",
        "launches_per_image": dom[2], "images_per_launch": images, "avg_launch_us": round(1e3 * dom[1] / dom[2], 2),
        "gflop_per_launch_avg": round(dom[0] / dom[2] / 1e9, 3),
        "share_of_conv_time": round(dom[1] / images / tot_ms, 3),
        "heaviest_shape_MNK": list(heavy_key[1:4]),
        "heaviest_shape_tflops": round(heavy["rec"]["flops"] / (heavy["ms"] * 1e-3) / 1e12, 2),
        "all_conv_launches": {"launches_per_image": sum(v[2] for v in per_kernel.values()),
                              "gflop_per_image": round(tot_flops / 1e9, 2),
                              "ms_per_image": round(tot_ms, 3),
                              "achieved": round(tot_flops / (tot_ms * 1e-3) / 1e12, 2),
                              **both(tot_flops / (tot_ms * 1e-3) / 1e12)},
        "backbone_conv": {"launches_per_image": n_base, "layers_per_image": len(list(pipe.rpn.base.net.units())), "gflop_per_image": round(base_flops / 1e9, 2), "ms_per_image": round(base_ms, 3),
                          "achieved": round(base_flops / (base_ms * 1e-3) / 1e12, 2),
                          **both(base_flops / (base_ms * 1e-3) / 1e12),
                          "note": "conv1..res4f, each launch alone on the chip (single image, no other stream): the latency view; "
                                  "with several images in flight the same launches overlap"},
        "method": "HIP events on the launch stream around a hipGraph that holds each distinct launch %d x back to back" % reps,
    }
    if per_product:
        roof["peak_basis"] = ("fp32-equivalent FLOP/s: 2.5 PFLOP/s dense 16-bit MFMA / %d matrix instructions per block of fp32 products (%s); `achieved` counts each "
                              "fp32 multiply-add once (the matrix pipe executes %dx that: %.0f TFLOP/s of 16-bit MFMA work = %.3f of 2.5 PFLOP/s)"
                              % (per_product, "bf16x6: exact three-way split" if x6_dom else "f16x3: two-way split with a scaled low part",
                                 per_product, per_product * achieved, per_product * achieved / PEAK_BF16_TFLOPS))
        roof["frac_of_native_fp32_peak"] = round(achieved / PEAK_F32_MATRIX_TFLOPS, 4)
        roof["mfma_pipe_frac"] = round(per_product * achieved / PEAK_BF16_TFLOPS, 4)
    else:
        roof["peak_basis"] = "native fp32 matrix peak (v_mfma_f32_32x32x2_f32)"
    return roof, groups


def backbone_in_flight(pipe, n_images, base_gflop, steps=20, batch=1, engine="native"):
    """The backbone (conv1 .. last base stage, with its pools) of `n_images` images at once, one hipGraph per image on its
    own stream -- how the timed pipeline keeps the chip busy -- timed as a whole: ms per image and the conv TFLOP/s that
    is.  The per-launch sum in `backbone_conv` is the latency view of the same launches."""
    from faster_rcnn_amd import ops
    from faster_rcnn_amd.pipeline import no_gc
    net = pipe.rpn.base.net
    streams = [torch.cuda.Stream() for _ in range(n_images)]
    graphs = []

    def run(x):
        ops.amax_begin()
        return net(x)
    for i, st in enumerate(streams):
        x = torch.from_numpy(np.concatenate([synth_image(200 + i * batch + j) for j in range(batch)])).cuda()
        ws = ops.ConvWorkspace() if (batch == 1 or os.environ.get("FRCNN_BENCH_BATCH_SPLITK")) else ops.NO_SPLIT_K
        arena = ops.AmaxArena() if engine == "f16x3" else None
        shared = n_images > 1 or batch > 1
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st), ops.conv_workspace(ws), ops.tile_policy(shared), ops.f32_engine(engine), ops.amax_arena(arena):
            run(x); run(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"), ops.conv_workspace(ws), ops.tile_policy(shared), \
                ops.f32_engine(engine), ops.amax_arena(arena):
            y = run(x)
        graphs.append((g, x, y, ws, arena))

    def step():
        for (g, _, _, _, _), st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (steps * n_images * batch)
    tf = base_gflop / ms
    out = {"images_in_flight": n_images * batch, "graphs_in_flight": n_images, "images_per_graph": batch, "ms_per_image": round(ms, 3), "achieved": round(tf, 2),
           "frac": round(tf / (PEAK_BF16_TFLOPS if DTYPE == "bf16" else PEAK_F32_MATRIX_TFLOPS), 4),
           "peak": PEAK_BF16_TFLOPS if DTYPE == "bf16" else PEAK_F32_MATRIX_TFLOPS,
           "peak_basis": "dense bf16 MFMA peak" if DTYPE == "bf16" else "native fp32 matrix peak (v_mfma_f32_32x32x2_f32)"}
    if DTYPE != "bf16" and engine in ("bf16x6", "f16x3"):
        ceil = PEAK_BF16_TFLOPS / (6.0 if engine == "bf16x6" else 3.0)
        out.update({"frac_of_engine_ceiling": round(tf / ceil, 4), "engine_ceiling": round(ceil, 1)})
    return {**out,
            "what": "one hipGraph of the base network per image, %d replaying concurrently on their own streams, wall clock over %d rounds" % (n_images, steps)}


def real_voc_image():
    """tests/golden/VOC_test/000005 (375x500 JPEG, 5 annotated chairs) the way voc_dets.main feeds it: resized within
    600/1000 (util.resize_imgs), resnet.preprocess.  -> (x (1,600,800,3) f32, resize_ratio, (width, height) of the original)."""
    from faster_rcnn_amd import resnet, util
    from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
    img = extract_img_data(os.path.join(ROOT, "tests", "golden", "VOC_test"), "000005")
    (resized,), (ratio,) = util.resize_imgs([img], min_size=600, max_size=1000)
    return resnet.preprocess(resized.data)[None].astype(np.float32), ratio, (img.width, img.height)


def e2e_parity(pipe, weights, anchors, oracle_runs):
    """SURVEY 8(d) "box mAP delta": the oracle END TO END on its own (image -> detections) beside the device END TO END
    on its own, over the synthetic 600x1000 images the cpu_baseline leg has just run through the oracle plus the real
    VOC_test/000005 frame, compared directly and through voc_dets.write_dets + eval_dets.voc_eval (oracle/e2e.py)."""
    from oracle import e2e
    from oracle.keras_ref import KerasGraphs
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    items = []
    for seed, kept, dets in (r[:3] for r in oracle_runs):
        items.append({"name": "synth%03d" % seed, "size": (WIDTH, HEIGHT), "oracle": (kept, dets), "device": e2e.device_detect(pipe, synth_image(seed))})
    x, ratio, size = real_voc_image()
    # the photograph runs on the head AS DRAWN: the calibration (made on a uniform-noise frame) shifts all its RoIs' logits
    # together until the softmax saturates at exactly 1.0 for one class, and 140 exactly tied scores leave the order of the
    # reference's NMS undefined -- a comparison of two arbitrary picks.  Same kernels, a second lowering of the
    # detector on a weight dict that differs in that one layer; both sides of the pair use it.
    real_pipe, real_w = raw_head_pipeline(pipe, weights, anchors)
    real_pipe = real_pipe or pipe
    g = KerasGraphs(real_w, torch.float32)
    items.append({"name": "000005", "size": size, "oracle": e2e.oracle_detect(g, x, anchors, NUM_CLASSES, DEPTH, PROPOSALS, ratio),
                  "device": e2e.device_detect(real_pipe, x, ratio)})
    res = e2e.compare(items, VOC_CLASS_MAPPING)
    res["heads"] = "synthetic frames: dense_class calibrated so that many classes fire (weights.calibrate_classifier); 000005: the layer as drawn"
    res["bar"] = "map_pair_delta <= %g" % E2E_MAP_BAR
    res["ok"] = bool(res["map_pair_delta"] <= E2E_MAP_BAR)
    return res


def raw_head_pipeline(pipe, weights, anchors):
    """A second lowering of the detector whose dense_class layer is the one DRAWN (before calibrate_head), on the same RPN
    model: (pipeline, weight dict).  None when the head was not calibrated."""
    if getattr(pipe, "raw_dense_class", None) is None:
        return None, weights
    from faster_rcnn_amd import resnet
    from faster_rcnn_amd.pipeline import InferencePipeline
    raw_w = dict(weights)
    raw_w["dense_class_%d" % NUM_CLASSES] = pipe.raw_dense_class
    det = (resnet.resnet50_classifier if DEPTH == 50 else resnet.resnet101_classifier)(PROPOSALS, NUM_CLASSES, weights=raw_w, dtype=DTYPE)
    det.head.hoist = HOIST
    return InferencePipeline(pipe.rpn, det, anchors, max_proposals=PROPOSALS), raw_w


def e2e_drift_bf16(pipe, weights, anchors, oracle_runs):
    """The bf16 conv path END TO END against the FP32 reference graph END TO END (BASELINE's "box mAP delta vs ref" for the
    bf16 configs): `oracle_runs` are the cpu_baseline leg's images through the plain f32 restatement (no storage model --
    the comparand does NOT share the precision loss, unlike `parity`'s stage-wise bf16-storage oracle); the device runs the
    same images on its bf16 engine on its own.  Reported per head: proposals kept by both, detections paired by class +
    IoU >= 0.5, score differences of the pairs, and the pair's mAP delta through voc_dets.write_dets + eval_dets.voc_eval.

    Two heads, because random-init weights make the answer depend on it: `head_as_drawn` (one class wins every RoI: the
    figure isolates how far bf16 moves boxes, scores and the NMS outcome) carries the bars; `head_calibrated` (dense_class
    re-centred so that ~20 classes fire, their winning margins a few 1e-4 of probability -- far below one bf16 rounding of
    the pooled features) is the stress case: class flips dominate it by construction and a TRAINED head, whose margins are
    orders of magnitude wider, sits near the first figure."""
    from oracle import e2e
    from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    from faster_rcnn_amd.pipeline import InferencePipeline
    mapping = VOC_CLASS_MAPPING if NUM_CLASSES == len(VOC_CLASS_MAPPING) else KITTI_CLASS_MAPPING
    one = InferencePipeline(pipe.rpn, pipe.det, anchors, max_proposals=PROPOSALS)       # one image per pass (bit-identical to the batched pass per image)
    raw, _ = raw_head_pipeline(pipe, weights, anchors)
    res = {"what": "fp32 oracle end to end vs bf16 device end to end, %d synthetic %dx%d frames" % (len(oracle_runs), HEIGHT, WIDTH)}
    for tag, p, col in (("head_calibrated", one, 2), ("head_as_drawn", raw, 3)):
        if p is None or any(len(r) <= col for r in oracle_runs):
            continue
        items = [{"name": "synth%03d" % r[0], "size": (WIDTH, HEIGHT), "oracle": (r[1], r[col]), "device": e2e.device_detect(p, synth_image(r[0]))}
                 for r in oracle_runs]
        c = e2e.compare(items, mapping)
        m, t = (int(v) for v in c["detections_matched_iou50"].split("/"))
        c["matched_frac"] = round(m / max(t, 1), 4)
        res[tag] = c
    bars = {"head_as_drawn": DRIFT_BARS_AS_DRAWN, "head_calibrated": DRIFT_BARS_CALIBRATED}
    map_bar = drift_map_bar(len(oracle_runs))
    res["bars"] = {k: "detections matched (class, IoU >= 0.5) >= %g of the larger set, mean score difference of the pairs <= %g; "
                      "map_pair_delta <= %g for %d frames (bench.DRIFT_MAP_BARS)" % (v + (map_bar, len(oracle_runs))) for k, v in bars.items() if k in res}
    res["ok"] = bool(all(res[k]["matched_frac"] >= m and res[k]["matched_score_diff"]["mean"] <= d and res[k]["map_pair_delta"] <= map_bar
                         for k, (m, d) in bars.items() if k in res))
    return res


def drift_map_bar(frames):
    """The bound on `map_pair_delta` of a bf16 run against the fp32 oracle for a sample of `frames` frames (DRIFT_MAP_BARS)."""
    key = "configs[3]" if DEPTH == 101 else "configs[1] shapes"
    many, few = DRIFT_MAP_BARS[key]
    return many if frames >= 32 else few


# Bars (matched fraction, mean score difference of the matched pairs).  Measured on MI355X, round 4: head as drawn -- configs[3]
# 0.956 matched / 4.7e-6, configs[1] shapes 0.917-0.922 / 7.5e-4; calibrated head -- 0.886 / 5.3e-3 and 0.831-0.837 / 6.3e-3.
# VERDICT r3's example bar (>= 0.90 matched) holds on the head as drawn.  The pair's mAP delta is REPORTED but not barred: it
# uses one pseudo ground-truth box per class and image (the oracle's most confident detection of that class), and an
# untrained head scores its RoIs within a few 1e-3 of each other, so the ranking behind the AP is decided by differences
# smaller than one bf16 rounding -- 0.09-0.21 from run to run of 4-8 frames, the oracle itself reaching only 0.64-0.83
# against its own top detections.  (The f32 device against the same oracle measures exactly 0.0 on the same metric:
# `parity.e2e`.)  With trained weights -- absent offline -- the delta would be the number to bar.
DRIFT_BARS_AS_DRAWN = (0.90, 5e-3)
DRIFT_BARS_CALIBRATED = (0.80, 2e-2)
# Round 5: the pair's mAP delta over 32 frames with a bootstrap over frames (scripts/drift_bf16.py,
# profiles/round5_drift_bf16_*.json).  configs[3] (ResNet-101 600x1500): head as drawn 0.039 over all frames (bootstrap mean 0.056,
# 97.5 % 0.171), calibrated head 0.103 (0.094 +- 0.026, 97.5 % 0.142); configs[1] shapes on the bf16 engine: 0.142 / 0.152 (97.5 %
# 0.279 / 0.268).  A sample of n frames spreads like 1 / sqrt(n): the bar for >= 32 frames is the bootstrap's 97.5th percentile
# rounded up, the bar for the 4-8 frames of bench.py's own leg adds 2 sigma of an 8x smaller sample.  It bounds THIS metric on THESE
# weights (one pseudo ground-truth box per class and image, an untrained head whose winning margins are a few 1e-4 of probability);
# it is a regression bound for the bf16 path, not a statement about a trained model's mAP.
DRIFT_MAP_BARS = {"configs[3]": (0.20, 0.40), "configs[1] shapes": (0.32, 0.55)}
E2E_MAP_BAR = 1e-3        # measured on MI355X: 0.0 (2700/2700 proposals, 2382/2382 detections identical over 9 frames)


def full_size_parity(pipe, weights, anchors):
    """The oracle as CHECKER at the benchmark's own size (configs[1] only): one synthetic 600x1000 image through the
    HIP pipeline (eager) and, stage by stage, through the CPU restatement fed with the SAME stage inputs -- float
    stages within 1e-4 (|a-b| / max(|b|, 1)), discrete stages (proposal selection, detection emission) exact."""
    from oracle import np_ref
    from oracle.keras_ref import KerasGraphs
    from faster_rcnn_amd import ops
    g = KerasGraphs(weights, torch.float32)
    x = synth_image(100)
    with torch.no_grad():
        measured0 = ops.AMAX_MEASURED
        out = pipe.forward_dev(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        measured = ops.AMAX_MEASURED - measured0             # f16x3 engine: tensors without a producer's magnitude record (each costs a pass)
        host = {k: v.cpu() for k, v in out.items()}
        err = lambda a, b: float(((a.double() - b.double()).abs() / b.double().abs().clamp(min=1)).max())
        feat = g.resnet_base(x, 50)
        cls, reg = g.rpn(feat)
        res = {"feat": err(host["feat"].reshape(feat.shape), feat), "rpn_cls": err(host["rpn_cls"].reshape(cls.shape), cls),
               "rpn_reg": err(host["rpn_reg"].reshape(reg.shape), reg)}
        # discrete: proposals chosen on the device == the numpy selection from the device's own RPN outputs
        n = int(host["n_rois"])
        kept = np_ref.proposals(host["rpn_reg"].numpy().reshape(reg.shape), host["rpn_cls"].numpy().reshape(cls.shape), anchors, 16, 8000, PROPOSALS)[0]
        res["proposals_equal"] = bool(n == len(kept) and np.array_equal(host["rois"].numpy()[:n], kept.astype(np.float32)))
        # float: detector on the device's conv4 map and RoIs
        o_cls, o_reg = g.resnet_classifier(host["feat"].reshape(feat.shape), host["rois"].numpy()[:n], NUM_CLASSES, 50)
        res["det_cls"] = err(host["cls"][:n], o_cls.reshape(n, -1))
        res["det_reg"] = err(host["reg"][:n], o_reg.reshape(n, -1))
        # discrete: emitted detections == the numpy post-process of the device's own detector outputs
        want = np_ref.detections(host["rois"].numpy()[:n], host["cls"].numpy()[:n], host["reg"].numpy()[:n], NUM_CLASSES - 1, 1.0)
        nd = int(host["n_dets"])
        # (exactly tied scores inside a class: numpy's argsort order is implementation-defined, the device orders ties
        # by ascending index -- runs of equal (class, score) are compared as sets)
        got = [(int(host["det_cls"][i]), float(host["det_prob"][i]), tuple(int(v) for v in host["det_bbox"][i])) for i in range(nd)]
        exp = [(int(w[0]), float(w[1]), tuple(int(v) for v in w[2])) for w in want]
        runs = lambda seq: [sorted(b for c, p, b in seq if (c, p) == key) for key in dict.fromkeys((c, p) for c, p, _ in seq)]
        res["detections_equal"] = bool(nd == len(want) and [g[:2] for g in got] == [e[:2] for e in exp] and runs(got) == runs(exp))
    res = {k: (v if isinstance(v, bool) else float("%.3g" % v)) for k, v in res.items()}
    res["ok"] = bool(res["proposals_equal"] and res["detections_equal"] and all(v < 1e-4 for v in res.values() if not isinstance(v, bool)))
    res["n_rois"], res["n_detections"] = n, nd
    res["amax_measured"] = measured
    return res


def cpu_baseline(weights, anchors, budget_s=20.0, runs=None, min_images=1, alt_dense_class=None):
    """The oracle ("port" of the Keras CPU path) on this host: full path on whole images.  ``runs``: a list that receives
    (seed, kept proposals, detections) of every image, for the end-to-end pair comparison (e2e_parity); with
    ``alt_dense_class`` a fourth entry, the detections with that dense_class layer (one extra 2048 x C product per image)."""
    from oracle import np_ref
    from oracle.keras_ref import KerasGraphs
    g = KerasGraphs(weights, torch.float32)
    n, t_total = 0, 0.0
    with torch.no_grad():
        while n < min_images or (t_total < budget_s and n < 8):
            x = synth_image(100 + n)
            t0 = time.perf_counter()
            feat = g.resnet_base(x, DEPTH)
            cls, reg = g.rpn(feat)
            kept = np_ref.proposals(reg.numpy(), cls.numpy(), anchors, 16, 8000, PROPOSALS)[0]
            rois = np_ref.pad_rois(kept.astype(np.float32), 64)
            out_cls, out_reg, pooled = g.resnet_classifier(feat, rois, NUM_CLASSES, DEPTH, return_pooled=True)
            dets = np_ref.detections(kept, out_cls.numpy(), out_reg.numpy(), NUM_CLASSES - 1, 1.0)
            t_total += time.perf_counter() - t0
            if runs is not None:
                run = (100 + n, kept, dets)
                if alt_dense_class is not None:                         # (outside the timed path)
                    k, b = (torch.as_tensor(np.asarray(a), dtype=pooled.dtype) for a in alt_dense_class)
                    run += (np_ref.detections(kept, torch.softmax(pooled @ k + b, dim=1).numpy(), out_reg.numpy(), NUM_CLASSES - 1, 1.0),)
                runs.append(run)
            n += 1
    return {"value": round(n / t_total, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d synthetic %dx%d image(s), ResNet-%d, full RPN+detector path, torch-CPU fp32 restatement of the Keras graph "
                      "+ numpy proposal/NMS/post-process (%.1f s)" % (n, HEIGHT, WIDTH, DEPTH, t_total)}


def full_size_parity_bf16(pipe, weights, anchors):
    """configs[3] at its own size: the oracle under the bf16 STORAGE model (oracle/keras_ref.py, mixed=True: f64/f32 arithmetic
    with one bf16 rounding where the device stores bf16) as checker, stage by stage on the device's own stage inputs.
    Float stages: relative RMS <= 1e-2 and max <= 3e-2 of the tensor's scale (tests/test_configs_full_size_gpu.py);
    discrete stages given the device's RPN / detector outputs: same proposal count with equal scores position by
    position and equal box sets inside every run of tied scores, detections likewise (bf16 makes exact ties common;
    numpy's order inside a tie is implementation-defined, the device's is ascending index -- DESIGN 6)."""
    from oracle import np_ref
    from oracle.keras_ref import KerasGraphs
    g = KerasGraphs(weights, torch.float32, mixed=True)
    x = synth_image(100)

    def rms_max(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        scale = float(b.abs().max().clamp(min=1e-6))
        return float(((a - b) ** 2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12)), float((a - b).abs().max()) / scale

    with torch.no_grad():
        batch = getattr(pipe, "batch", 1)
        if batch > 1:
            # the batched pipeline (what the timed graphs replay): the checked image rides as image 0 of a batch of different
            # images; its slice of every output is what the oracle is compared with
            from faster_rcnn_amd import ops
            xb = np.concatenate([x] + [synth_image(300 + j) for j in range(batch - 1)])
            with ops.conv_workspace(ops.NO_SPLIT_K), ops.tile_policy(True):
                out = pipe.forward_dev(torch.from_numpy(xb).cuda())
            out = {k: (v[0] if isinstance(v, list) else v[0:1] if k in ("rpn_cls", "rpn_reg", "feat") else v[0]) for k, v in out.items()}
        else:
            out = pipe.forward_dev(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        host = {k: (v.float() if v.dtype == torch.bfloat16 else v).cpu() for k, v in out.items()}
        res = {}
        feat = g.resnet_base(x, DEPTH)
        res["feat"] = rms_max(host["feat"].reshape(feat.shape), feat)
        dev_feat = host["feat"].reshape(feat.shape)
        cls, reg = g.rpn(dev_feat)
        res["rpn_cls"] = rms_max(host["rpn_cls"].reshape(cls.shape), cls)
        res["rpn_reg"] = rms_max(host["rpn_reg"].reshape(reg.shape), reg)
        n = int(host["n_rois"])
        reg_np, cls_np = host["rpn_reg"].numpy().reshape(reg.shape), host["rpn_cls"].numpy().reshape(cls.shape)
        kept, kprobs = np_ref.proposals(reg_np, cls_np, anchors, 16, 8000, PROPOSALS)[:2]
        got = host["rois"].numpy()[:n]
        same = n == len(kept)
        if same:
            kp = np.asarray(kprobs)
            start = 0
            for end in list(np.nonzero(np.diff(kp))[0] + 1) + [n]:
                same = same and sorted(map(tuple, np.asarray(kept[start:end], np.float32).tolist())) == sorted(map(tuple, got[start:end].tolist()))
                start = end
        res["proposals_equal"] = bool(same)
        o_cls, o_reg = g.resnet_classifier(dev_feat, got, NUM_CLASSES, DEPTH)
        res["det_cls"] = rms_max(host["cls"][:n], o_cls.reshape(n, -1))
        res["det_reg"] = rms_max(host["reg"][:n], o_reg.reshape(n, -1))
        want = np_ref.detections(got, host["cls"].numpy()[:n], host["reg"].numpy()[:n], NUM_CLASSES - 1, 1.0)
        nd = int(host["n_dets"])
        gd = [(int(host["det_cls"][i]), float(host["det_prob"][i]), tuple(int(v) for v in host["det_bbox"][i])) for i in range(nd)]
        ed = [(int(w[0]), float(w[1]), tuple(int(v) for v in w[2])) for w in want]
        runs = lambda seq: [sorted(b for c, p, b in seq if (c, p) == key) for key in dict.fromkeys((c, p) for c, p, _ in seq)]
        res["detections_equal"] = bool(nd == len(want) and [t[:2] for t in gd] == [t[:2] for t in ed] and runs(gd) == runs(ed))
    floats = {k: v for k, v in res.items() if isinstance(v, tuple)}
    ok = bool(res["proposals_equal"] and res["detections_equal"] and all(r < 1e-2 and m < 3e-2 for r, m in floats.values()))
    res = {k: ([float("%.3g" % t) for t in v] if isinstance(v, tuple) else v) for k, v in res.items()}
    res.update({"ok": ok, "n_rois": n, "n_detections": nd, "bars": "float stages [relative RMS, max / scale] <= [1e-2, 3e-2] against the "
                "oracle under the bf16 storage model; proposals / detections exact given the device's own float outputs (ties as sets)"})
    return res


def vgg_rpn_cpu_and_parity(pipe, weights, budget_s=20.0):
    """configs[0] is the one config the reference itself runs on the CPU (train_rpn_test.py:21-38): the oracle's torch-CPU
    fp32 restatement of vgg16_base + vgg16_rpn timed on this host, and -- outside the timing --
    used as the checker of the HIP forward at this size (feature map and both RPN outputs within 1e-4)."""
    from oracle.keras_ref import KerasGraphs
    g = KerasGraphs(weights, torch.float32)
    n, t_total = 0, 0.0
    with torch.no_grad():
        while n < 1 or (t_total < budget_s and n < 12):
            x = synth_image(100 + n)
            t0 = time.perf_counter()
            feat = g.vgg_base(x)
            cls, reg = g.rpn(feat)
            t_total += time.perf_counter() - t0
            n += 1
        out = pipe.forward_dev(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        err = lambda a, b: float(((a.cpu().double() - b.double()).abs() / b.double().abs().clamp(min=1)).max())
        dev_feat = out["feat"].cpu().reshape(feat.shape)
        c2, r2 = g.rpn(dev_feat)                                          # heads on the device's own feature map
        par = {"feat": err(out["feat"].reshape(feat.shape), feat), "rpn_cls": err(out["rpn_cls"].reshape(cls.shape), c2),
               "rpn_reg": err(out["rpn_reg"].reshape(reg.shape), r2)}
    par = {k: float("%.3g" % v) for k, v in par.items()}
    par["ok"] = bool(all(v < 1e-4 for v in par.values()))
    base = {"value": round(n / t_total, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d synthetic 600x1000 image(s), VGG16 base + RPN heads, torch-CPU fp32 restatement of the Keras graph (%.1f s)" % (n, t_total)}
    return base, par


def spawn_ranks(n):
    """`python bench.py --gpus N` without a rank environment: start N ranks of this same command line, one per GPU, the
    way torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), wait for them, print
    rank 0's JSON line.  Runs BEFORE this process makes any HIP call: the devices are counted from the KFD topology in
    sysfs (dp.count_gpus), and a process that has initialised the runtime must never exec or fork GPU work.  Each rank pins
    itself to the cores next to its GPU when it starts (main: dp.pin_rank)."""
    import socket
    import subprocess
    backend = os.environ.get("FRCNN_BENCH_BACKEND", "nccl")
    from faster_rcnn_amd import dp
    have = dp.count_gpus()          # KFD topology + the *_VISIBLE_DEVICES variables, read from sysfs: no HIP call in this process
    if have < n and backend != "gloo":
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing (one rank per GPU over RCCL; "
                         "FRCNN_BENCH_BACKEND=gloo lets ranks share a GPU for a functional check)\n" % (n, have))
        sys.exit(2)
    if have < 1:
        sys.stderr.write("bench.py: no HIP device visible\n")
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained on a thread; the ranks are polled so that ONE rank dying (before the process group is up
    # the others would sit in the rendezvous until its timeout) takes the rest down at once -- by the PIDs started here
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    # an overall deadline: ranks stuck in a collective nobody else joins would otherwise wait for the RCCL watchdog
    deadline = time.monotonic() + float(os.environ.get("FRCNN_BENCH_DEADLINE_S", "3600"))
    while any(p.poll() is None for p in procs):
        late = time.monotonic() > deadline
        if late or any(p.poll() not in (None, 0) for p in procs):
            failed = True
            if late:
                sys.stderr.write("bench.py: ranks still running at the deadline (FRCNN_BENCH_DEADLINE_S): terminating them\n")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    reader.join(timeout=10)
    if not failed:
        for ln in b"".join(chunks).decode().splitlines():
            if ln.startswith("{"):                          # rank 0's line (anything else a library printed is dropped)
                sys.stdout.write(ln + "\n")
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or failed:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


TRAIN_H, TRAIN_W = 600, 1000          # configs[2] / configs[4] are ResNet-50 at 600x1000 whatever inference config the line is for


def train_loop_images(n_images, height, width, seed=900):
    """`n_images` distinct in-memory frames the way the reference's loaders hand them (shapes.Image over decoded uint8 BGR pixels with
    annotated boxes): uniform-noise pixels, 3-6 VOC-class boxes each, a few with anchor-like shapes so that positives exist."""
    from faster_rcnn_amd import shapes
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    names = [k for k in VOC_CLASS_MAPPING if k != "bg"]
    rs = np.random.RandomState(seed)
    imgs = []
    for k in range(n_images):
        px = rs.randint(0, 256, (height, width, 3)).astype(np.uint8)
        gts = []
        for _ in range(rs.randint(3, 7)):
            bw, bh = rs.choice([96, 128, 180, 256, 360]), rs.choice([96, 128, 180, 256, 360])
            x1, y1 = rs.randint(0, max(1, width - bw - 1)), rs.randint(0, max(1, height - bh - 1))
            gts.append(shapes.GroundTruthBox(names[rs.randint(len(names))], False, shapes.Box(int(x1), int(y1), int(min(width - 1, x1 + bw)), int(min(height - 1, y1 + bh)))))
        imgs.append(shapes.Image(shapes.Metadata("synth%03d" % k, width, height, gts, "none"), px))
    return imgs


def train_loop_leg(kind, dtype="f32", n_images=32, iterations=64, warm=64, fast=True, height=None, width=None):
    """ms per ITERATION of the reference's own training loops (train_util.train_rpn / train_detector_step2,
    train_util.py:37-54, 100-118) over `n_images` distinct images -- image fetch, targets / proposals, sampling, the step, the
    loss line -- beside the bare train_on_batch step bench_train.py times on one pre-staged input.  `fast`: the managers'
    device-resident feed (train_util.FAST_FEED); False: batched_image / rpn_y_true / get_training_input as host numpy.
    `warm` untimed iterations first (two walks over the image list by default: allocator pools of three streams, pinned staging
    areas and kernel images are first-use costs -- with 24 warm iterations the timed RPN loop read 2.7 ms per iteration, 2.03 once
    warm, scripts/dev/loop_profile.py)."""
    import contextlib
    import io
    import random
    from faster_rcnn_amd import det_util, resnet, rpn_util, train, train_util, util
    from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
    from faster_rcnn_amd.weights import synthetic_resnet
    height, width = height or TRAIN_H, width or TRAIN_W
    anchors = util.get_anchors([128, 256, 512])
    A, C = len(anchors), 21
    imgs = train_loop_images(n_images, height, width)
    random.seed(1); np.random.seed(1337)
    reg = dict(weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    if kind == "rpn_step1":
        model = resnet.resnet50_rpn(resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1), dtype=dtype, **reg), anchors_per_loc=A)
        mgr = rpn_util.RpnTrainingManager(resnet.get_conv_rows_cols, 16, resnet.preprocess, anchors)
        loop = train_util.train_rpn
    else:
        frozen = resnet.resnet50_rpn(resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1)), anchors_per_loc=A)
        model = resnet.resnet50_classifier(64, C, resnet.resnet50_base(weights=synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=2), dtype=dtype, **reg))
        mgr = det_util.DetTrainingManager(frozen, VOC_CLASS_MAPPING, resnet.preprocess, anchor_dims=anchors)
        loop = train_util.train_detector_step2
    opt = train.optimizer_from_str("sgd")
    was = train_util.FAST_FEED
    train_util.FAST_FEED = bool(fast)
    try:
        with contextlib.redirect_stdout(io.StringIO()):            # (the loops print one line per iteration)
            loop(model, imgs, mgr, opt, phases=[[warm, 1e-3]])
            train.finish_pending_updates()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(model, imgs, mgr, opt, phases=[[iterations, 1e-3]])
            train.finish_pending_updates()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    finally:
        train_util.FAST_FEED = was
    return {"ms_per_iteration": round(1e3 * el / iterations, 3), "iterations": iterations, "distinct_images": n_images, "dtype": dtype,
            "feed": "device-resident (train_util.FAST_FEED: uint8 frame up, resize / preprocess / targets on the device, next image prepared beside the step)"
                    if fast else "host numpy (batched_image / rpn_y_true / get_training_input, the reference's calls taken literally)"}


def train_dp_leg(anchors, rank, world, kind="rpn_step1", steps=10, warmup=10):
    """One training config inside the N > 1 line, one image per GPU per step, the flat gradient buffer through the path's
    ONE collective (dp.allreduce_sum_begin -> RCCL all-reduce over xGMI).  Every rank runs this; returns the object rank 0
    prints.
      kind "rpn_step1"  BASELINE configs[2]: ResNet-50 RPN step-1 steps (train_rpn_step1.py / train_util.py:38-54), fp32, 47.3 MB
      kind "det_step2"  BASELINE configs[4]: ResNet-50 detector step-2 steps (train_det_step2.py:52-115), 64 RoIs, mixed bf16
                        (bf16 activations / gradients / packed filters, f32 masters and an f32 89.0 MB gradient payload)
    `exposed_allreduce_ms` = step time with the collective - step time with the collective left out (same kernels)."""
    import torch.distributed as dist
    from faster_rcnn_amd import dp, resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    A, C = len(anchors), 21
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    seen = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(seen)                                           # what the collective library saw
    rs = np.random.RandomState(500 + rank)
    x = (rs.randint(0, 256, (TRAIN_H, TRAIN_W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]    # float64: what the managers hand (Keras casts on feed)
    rows, cols = resnet.get_conv_rows_cols(TRAIN_H, TRAIN_W)
    weights = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=1 if kind == "rpn_step1" else 2)    # identical on every rank
    if kind == "rpn_step1":
        base = resnet.resnet50_base(weights=weights, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
        model = resnet.resnet50_rpn(base, anchors_per_loc=A)
        can_use = rs.rand(1, rows, cols, A) < 0.012
        is_pos = rs.rand(1, rows, cols, A) < 0.01
        y_class = np.concatenate([can_use, is_pos], axis=3)
        y_bbreg = np.concatenate([np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32),
                                  (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)], axis=3)
        inputs, targets = x, [y_class, y_bbreg]
        workload = "configs[2]: ResNet-50 RPN step-1 training, 600x1000, 1 image per GPU per step, SGD momentum, l2 1e-4, fp32, synthetic"
    else:
        base = resnet.resnet50_base(weights=weights, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER, dtype="bf16")
        model = resnet.resnet50_classifier(64, C, base)
        n = 64
        x1, y1 = rs.randint(0, cols - 8, n), rs.randint(0, rows - 8, n)
        rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 7, n), y1 + 1 + rs.randint(0, 7, n)], axis=1).astype(np.float32)[None]
        ci = rs.randint(0, C, n)
        yc = np.zeros((1, n, C), np.float32)
        yc[0, np.arange(n), ci] = 1
        lab, tg = np.zeros((n, 4 * (C - 1)), np.float32), np.zeros((n, 4 * (C - 1)), np.float32)
        for i, c in enumerate(ci):
            if c < C - 1:
                lab[i, 4 * c:4 * c + 4] = 1
                tg[i, 4 * c:4 * c + 4] = rs.randn(4)
        inputs, targets = [x, rois], [yc, np.concatenate([lab, tg], axis=1)[None]]
        workload = ("configs[4]: ResNet-50 detector step-2 training, 600x1000, 64 RoIs, 1 image per GPU per step, SGD momentum, l2 1e-4, "
                    "mixed bf16 (f32 masters and gradient payload), synthetic")
    model.compile(train.SGD(1e-3, 0.9))
    params = model._trainer.params

    def timed(fn, k, w):
        prev = None
        for i in range(w + k):
            if i == w:
                if hasattr(prev, "result"):
                    prev.result()
                prev = None
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
            cur = fn()
            if hasattr(prev, "result"):
                prev.result()                                       # losses read one step late, like train_util's loops
            prev = cur
        if hasattr(prev, "result"):
            prev.result()
        train.finish_pending_updates()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return 1e3 * float(t.item()) / k

    step = lambda: model.train_on_batch(inputs, targets, defer=True)
    ms = timed(step, steps, warmup)
    train.DP_SYNC = False
    try:
        ms_local = timed(step, steps, 1)
    finally:
        train.DP_SYNC = True
    buf = torch.zeros_like(params.g)
    # (the begin / wait pair is the form a training step uses and the one dp.FORCE_COLLECTIVE applies to: a forced one-rank run
    #  times a real RCCL all-reduce here, not a no-op)
    def one_allreduce():
        handle, _ = dp.allreduce_sum_begin(buf)
        if handle is not None:
            handle.wait()
        torch.cuda.synchronize()
    ar = timed(one_allreduce, 10, 2)
    train.finish_pending_updates()
    return {"workload": workload,
            "ranks_seen": int(seen.item()), "backend": dist.get_backend(), "steps": steps,
            "ms_per_step": round(ms, 3), "img_s": round(world * 1e3 / ms, 2),
            "grad_payload_MB": round(params.total * 4 / 1e6, 1), "collectives_per_step": 1,
            "allreduce_ms": round(ar, 3), "ms_per_step_without_allreduce": round(ms_local, 3),
            "exposed_allreduce_ms": round(max(ms - ms_local, 0.0), 3),
            "overlap": "the all-reduce starts behind the last weight-gradient batch; optimiser + re-pack wait for it, the next image's "
                       "host staging, upload and frozen stages (stem..res3) run beside it"}


def reference_entry_leg(pipe, anchors, rank, n_images=32, passes=3, eager_images=8):
    """The reference's own inference entry point, timed: ``voc_dets.get_dets_by_cls`` (voc_dets.py:91-111) over a
    ``DetTrainingManager`` (det_util.py:136-158) on a list of ``n_images`` distinct synthetic uint8 frames of the
    benchmark's size -- host frame in, list of detection dicts out, everything the call does included (staging copy,
    H2D, captured pass, D2H, building the dicts).  The captured path (entry.DetectionEntry) is what a caller of that
    function gets; the eager path (`voc_dets.FAST_ENTRY = False`: the reference's own sequencing with its two host round
    trips) is timed beside it on ``eager_images`` frames so the gain is visible."""
    import contextlib
    import io as _io
    from faster_rcnn_amd import entry, resnet, shapes, voc_dets
    from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    from faster_rcnn_amd.det_util import DetTrainingManager
    mapping = VOC_CLASS_MAPPING if NUM_CLASSES == len(VOC_CLASS_MAPPING) else KITTI_CLASS_MAPPING
    mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=mapping, preprocess_func=resnet.preprocess, anchor_dims=anchors)
    rs = np.random.RandomState(2000 + rank)
    images = [shapes.Image(shapes.Metadata("synth%03d" % i, WIDTH, HEIGHT, [], "none"), rs.randint(0, 256, (HEIGHT, WIDTH, 3)).astype(np.uint8))
              for i in range(n_images)]
    ratios = [1.0] * n_images
    sink = _io.StringIO()

    def run(imgs):
        with contextlib.redirect_stdout(sink):                  # the reference's per-image progress lines
            t0 = time.perf_counter()
            dets = voc_dets.get_dets_by_cls(mgr, pipe.det, ratios[:len(imgs)], imgs)
            return time.perf_counter() - t0, dets
    t_first, _ = run(images)                                    # first sighting of the size: captures its passes
    run(images)
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(passes):
        dt, dets = run(images)
        t += dt
    n_dets = sum(len(v) for per_img in dets.values() for v in per_img.values())
    # the same call on a list eight times as long (the 32 frames repeated): a 32-frame call spends ~1/8 of its time filling and
    # draining the images in flight (the host waits on the device for 90 % of the call: scripts/dev/entry_profile.py)
    long_imgs = [shapes.Image(shapes.Metadata("synth%03d_%d" % (i, k), WIDTH, HEIGHT, [], "none"), im.raw) for k in range(8) for i, im in enumerate(images)]
    ratios = ratios * 8
    t_long, _ = run(long_imgs)
    ratios = ratios[:n_images]
    dtype = getattr(pipe.det.head, "dtype", "f32")
    eng = entry.for_models(mgr, pipe.det, 64, 16, entry.default_in_flight(dtype))
    voc_dets.FAST_ENTRY = False
    try:
        run(images[:2])
        t_eager, dets_eager = run(images[:eager_images])
    finally:
        voc_dets.FAST_ENTRY = True
    same = all(k in dets and all(len(v[i]) == len(dets[k][i]) for i in v) for k, v in dets_eager.items())
    # detection by detection (a captured four-image pass and an eager one-image launch sum some layers in another order: ~1e-6 in a
    # score, now and then a box edge on the other side of a .5 or two near-tied boxes swapped in an NMS)
    n_eager = n_same = 0
    worst = 0.0
    for k, v in dets_eager.items():
        for i, lst in v.items():
            mine = {tuple(int(c) for c in d["bbox"]): float(d["prob"]) for d in dets.get(k, {}).get(i, [])}
            n_eager += len(lst)
            for d in lst:
                key = tuple(int(c) for c in d["bbox"])
                if key in mine:
                    n_same += 1
                    worst = max(worst, abs(mine[key] - float(d["prob"])))
    # ---- the same call on FILES: the VOC frame as voc_dets.main feeds it (a 500x375 JPEG on disk, metadata resized to 800x600):
    # JPEG decode on a few host threads, the decoded frame uploaded, INTER_CUBIC resize + preprocess + network on the device;
    # the eager path beside it decodes and resizes on the host the way the reference does (cv2 there, integer numpy here)
    from_files = None
    try:
        from faster_rcnn_amd import util
        from faster_rcnn_amd.data.voc_data_helpers import extract_img_data
        base = extract_img_data(os.path.join(ROOT, "tests", "golden", "VOC_test"), "000005")
        frames = []
        for i in range(n_images):
            (r,), (ratio,) = util.resize_imgs([base], min_size=600, max_size=1000)
            r.metadata.name = "file%03d" % i
            frames.append(r)
        fr = [ratio] * n_images

        def run_files(imgs):
            with contextlib.redirect_stdout(sink):
                t0 = time.perf_counter()
                d = voc_dets.get_dets_by_cls(mgr, pipe.det, fr[:len(imgs)], imgs)
                return time.perf_counter() - t0, d
        run_files(frames)
        tf = sum(run_files(frames)[0] for _ in range(passes))
        voc_dets.FAST_ENTRY = False
        try:
            te, _ = run_files(frames[:4])
        finally:
            voc_dets.FAST_ENTRY = True
        from_files = {"value": round(n_images * passes / tf, 3), "unit": "img/s", "frame": "tests/golden/VOC_test 000005.jpg, 500x375 -> 800x600",
                      "decode": "inline (threads start when a fetch takes over %.1f ms)" % voc_dets.DECODE_INLINE_MS, "eager": {"value": round(4 / te, 3), "unit": "img/s", "images": 4},
                      "what": "JPEG decode (PIL) -> H2D of the decoded frame -> device INTER_CUBIC resize + preprocess + captured pass -> dicts; "
                              "eager: decode + integer-numpy resize + float64 preprocess on the host, then the eager device path"}
    except Exception as e:
        from_files = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"value": round(n_images * passes / t, 3), "from_files": from_files, "unit": "img/s", "images": n_images, "passes": passes,
            "detections_per_image": round(n_dets / n_images, 1), "first_pass_s": round(t_first, 3), "graph_cache": eng.stats(),
            "long_list": {"value": round(len(long_imgs) / t_long, 3), "unit": "img/s", "images": len(long_imgs)},
            "eager": {"value": round(eager_images / t_eager, 3), "unit": "img/s", "images": eager_images,
                      "what": "the same call with voc_dets.FAST_ENTRY = False: get_det_inputs returns the conv map and the RoIs as numpy "
                              "(det_util.py:136-158), the detector takes them back (voc_dets.py:49), one image at a time"},
            "same_detection_counts_as_eager": bool(same),
            "detections_identical_to_eager": "%d/%d" % (n_same, n_eager), "max_score_difference_from_eager": worst,
            "what": "voc_dets.get_dets_by_cls(DetTrainingManager, detector, ratios, images) over %d distinct %dx%d uint8 frames in host memory, "
                    "%d captured passes in flight x %d image(s) per pass, list of detection dicts out; wall clock of the whole call.  The captured passes score the %d kept proposals (the "
                    "reference's padded copies of a batch's first RoI, voc_dets.py:42-51, have their original's box, class and score: the per-class NMS "
                    "returns the same list without them; the eager path beside it scores the %d-row padded list)" % (n_images, HEIGHT, WIDTH, eng.in_flight, eng.batch, PROPOSALS, -(-PROPOSALS // 64) * 64)}


# a VOC07-like histogram of SOURCE sizes (width, height, share): the common camera formats, their +-1..2 pixel neighbours, and a tail
# of sizes that occur once or twice; after shapes.Image.resize_within_bounds(600, 1000) each is a geometry of its own
MIXED_SIZES = [((500, 375), 0.40), ((500, 333), 0.14), ((375, 500), 0.11), ((500, 334), 0.04), ((333, 500), 0.04), ((500, 374), 0.03),
               ((500, 332), 0.03), ((500, 400), 0.02), ((500, 357), 0.02), ((334, 500), 0.02), ((500, 376), 0.02), ((480, 360), 0.015),
               ((500, 281), 0.015)]


def mixed_sizes_leg(pipe, anchors, n_images=256, seed=77, canvas=True):
    """``voc_dets.get_dets_by_cls`` (voc_dets.py:91-111) over a SHUFFLED list of ``n_images`` frames whose source sizes follow
    MIXED_SIZES (the rest of the probability mass: sizes drawn once, 500 x 250..499): every source size resizes to its own geometry
    (shapes.py:106-123), a captured pass serves one geometry.  Timed: the first call (captures included) and the same call again
    (passes cached); reported with the number of geometries, captures, eager fallbacks and cache bytes."""
    import contextlib
    import io as _io
    from faster_rcnn_amd import entry, resnet, shapes, util, voc_dets
    from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    from faster_rcnn_amd.det_util import DetTrainingManager
    mapping = VOC_CLASS_MAPPING if NUM_CLASSES == len(VOC_CLASS_MAPPING) else KITTI_CLASS_MAPPING
    mgr = DetTrainingManager(rpn_model=pipe.rpn, class_mapping=mapping, preprocess_func=resnet.preprocess, anchor_dims=anchors)
    rs = np.random.RandomState(seed)
    sizes = []
    for (w, h), share in MIXED_SIZES:
        sizes += [(w, h)] * int(round(share * n_images))
    while len(sizes) < n_images:
        sizes.append((500, int(rs.randint(250, 500))))
    sizes = sizes[:n_images]
    rs.shuffle(sizes)
    pool = {}                                                    # one random frame per source size (the pixels do not matter to the timing)
    raw = []
    for i, (w, h) in enumerate(sizes):
        if (w, h) not in pool:
            pool[(w, h)] = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        raw.append(shapes.Image(shapes.Metadata("mixed%03d" % i, w, h, [], "none"), pool[(w, h)]))
    images, ratios = util.resize_imgs(raw, min_size=600, max_size=1000)
    geometries = len({(im.height, im.width) for im in images})
    dtype = getattr(pipe.det.head, "dtype", "f32")
    eng = entry.for_models(mgr, pipe.det, 64, 16, entry.default_in_flight(dtype))
    eng.cache.clear()                                            # (the leg measures a list nobody has seen: no pass of an earlier leg helps)
    was_capable = eng.canvas_capable
    eng.canvas_capable = bool(canvas) and was_capable
    before = eng.stats()
    sink = _io.StringIO()
    eager_calls = [0]
    real_eager = voc_dets._get_dets_eager

    def counting(*a, **k):
        eager_calls[0] += 1
        return real_eager(*a, **k)
    voc_dets._get_dets_eager = counting
    try:
        def run():
            with contextlib.redirect_stdout(sink):
                t0 = time.perf_counter()
                d = voc_dets.get_dets_by_cls(mgr, pipe.det, ratios, images)
                return time.perf_counter() - t0, d
        t1, d1 = run()
        e1 = eager_calls[0]
        mid = eng.stats()
        t2, d2 = run()
        e2 = eager_calls[0] - e1
    finally:
        voc_dets._get_dets_eager = real_eager
        eng.canvas_capable = was_capable
    after = eng.stats()
    same = d1.keys() == d2.keys() and all(d1[c].keys() == d2[c].keys() and all(len(d1[c][i]) == len(d2[c][i]) for i in d1[c]) for c in d1)
    return {"first_call": {"value": round(n_images / t1, 2), "unit": "img/s", "seconds": round(t1, 3), "captures": mid["captures"] - before["captures"],
                           "capture_seconds": round(mid["capture_seconds"] - before["capture_seconds"], 3), "eager_images": e1},
            "second_call": {"value": round(n_images / t2, 2), "unit": "img/s", "seconds": round(t2, 3), "captures": after["captures"] - mid["captures"], "eager_images": e2},
            "images": n_images, "geometries": geometries, "graph_cache_bytes": after["bytes"], "graphs": after["graphs"],
            "capture_breakdown_ms": {k: round(v - before.get("capture_breakdown_ms", {}).get(k, 0.0), 1) for k, v in after.get("capture_breakdown_ms", {}).items()},
            "passes": ("per canvas class (even sides, multiples of %d; an odd side sits at offset 1; true sizes as device values; the classes planned "
                       "from the list's histogram of sizes: entry.plan_canvas_classes; the default)" % entry.CANVAS_GRANULE)
                      if canvas else "per exact geometry (FRCNN_ENTRY_CANVAS=0: round 5's policy)",
            "canvas_classes": sorted({k[1:3] for k in eng.cache.keys() if k[:1] == ("canvas",)}),
            "reorder_window": voc_dets.REORDER_WINDOW, "capture_min": voc_dets.CAPTURE_MIN, "images_per_pass": eng.batch, "in_flight": eng.in_flight,
            "same_detection_counts_both_calls": bool(same),
            "what": "voc_dets.get_dets_by_cls over a shuffled list of %d frames of %d geometries (VOC07-like source sizes resized within 600 / 1000); images "
                    "of one pass shape (canvas class / geometry) share captured passes wherever they stand in the list (held back per shape, at most "
                    "REORDER_WINDOW); a shape seen fewer than CAPTURE_MIN times runs the eager sequence" % (n_images, geometries)}


def entry_legs(pipe, anchors, rank):
    """Everything measured through the reference's inference entry point: one geometry (32 / 256 frames, from files, eager beside it),
    and for the fp32 ResNet-50 the shuffled list of mixed image sizes (planned canvas classes / exact-geometry passes)."""
    try:
        via = reference_entry_leg(pipe, anchors, rank)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    if DEPTH == 50 and DTYPE == "f32":
        try:
            via["mixed_sizes"] = mixed_sizes_leg(pipe, anchors)
            via["mixed_sizes_exact_geometry_passes"] = mixed_sizes_leg(pipe, anchors, canvas=False)
        except Exception as e:
            via["mixed_sizes"] = {"error": "%s: %s" % (type(e).__name__, e)}
    via["process"] = {"hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                      "what": "GPU_MAX_HW_QUEUES as voc_dets.main sets it (the reference's entry point is a script: voc_dets.py:161-192); the captured passes in "
                              "flight follow it (entry.default_in_flight)"}
    return via


# The reference's entry point is a SCRIPT (voc_dets.py:161-192), and this package's voc_dets.main asks the runtime for more hardware queues
# than its default four before the first HIP call (GPU_MAX_HW_QUEUES: read once, when the runtime starts).  The headline loop above wants
# the four (four passes on four queues: 546 img/s; on eight queues 502), get_dets_by_cls the many: on four queues its passes -- staging
# copy, replay, read-back per pass, the host collecting the oldest -- measure 497 img/s, on eight or sixteen 542-547 (scripts/dev/r6_hw_queues.sh,
# r6_entry_child_sweep2.sh).
# So the entry-point legs run as a child of the default command, started before this process touches the GPU, under main's setting.
ENTRY_HW_QUEUES = "8"                            # = faster_rcnn_amd.voc_dets.ENTRY_HW_QUEUES (not imported here: this runs before anything touches torch)


def entry_legs_child(config, timeout_s=150.0):
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FRCNN_BENCH_NO_NATIVE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FRCNN_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.setdefault("GPU_MAX_HW_QUEUES", ENTRY_HW_QUEUES)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "bench.py", "--entry-only", "--config", config], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "exit code %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
        out = json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %.0f s" % timeout_s}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


_JSON_OUT = None


# ---- the other BASELINE configs in the driver's N=1 line (`extra`): each runs as a CHILD process of this command line, started and
# finished BEFORE this process makes its first HIP call (a process that has initialised the runtime starts no GPU children), one after
# the other, so each has the chip to itself exactly as when run by hand.
EXTRA_LEGS = (
    ("c0_vgg16_rpn", ["bench.py", "--config", "c1", "--no-extra", "--no-cpu-baseline", "--no-io", "--steps", "30", "--warmup", "5"]),
    ("c3_r101_bf16", ["bench.py", "--config", "c4", "--no-extra", "--no-cpu-baseline", "--no-io", "--steps", "12", "--warmup", "3"]),
    ("train_steps_f32", [os.path.join("scripts", "bench_train.py"), "--through-loop", "--no-host-feed", "--steps", "60", "--warmup", "40"]),
    ("train_steps_mixed_bf16", [os.path.join("scripts", "bench_train.py"), "--bf16", "--through-loop", "--no-host-feed", "--steps", "60", "--warmup", "40"]),
)


def compact_inference_leg(line):
    """What `extra` keeps of a child's bench line (configs[0] / configs[3]): the value with its config and its rooflines."""
    roof = line.get("roofline") or {}
    keep = {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
            "warmup": line["warmup"], "dtype": line["dtype"], "data": line["data"],
            "config": {k: line["config"].get(k) for k in ("workload", "images_per_step_per_gpu", "graphs_in_flight", "images_per_graph", "hw_queues", "launch", "split_k")},
            "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "error") if k in roof}}
    for part in ("all_conv_launches", "backbone_conv"):
        if isinstance(roof.get(part), dict):
            keep["roofline"][part] = {k: roof[part].get(k) for k in ("achieved", "frac", "gflop_per_image", "in_flight") if k in roof[part]}
    if "end_to_end_conv_tflops" in roof:
        keep["roofline"]["end_to_end_conv_tflops"] = roof["end_to_end_conv_tflops"]
    return keep


def compact_train_leg(line):
    """What `extra` keeps of scripts/bench_train.py's line: per step kind the bare step, the loop iteration and the roofline."""
    keep = {"dtype": line["dtype"], "workload": line["workload"], "losses_read": line["losses_read"], "step_launch": line.get("step_launch")}
    for tag in ("rpn_step1", "det_step2"):
        if tag in line:
            t = line[tag]
            keep[tag] = {"ms_per_step": t["ms_per_step"], "img_s": t["img_s"], "ms_per_step_losses_read_every_step": t.get("ms_per_step_losses_read_every_step"),
                         "roofline": t["roofline"]}
            loop = (t.get("through_loop") or {}).get("fast_feed")
            if loop:
                keep[tag]["through_train_util_loop"] = {k: loop.get(k) for k in ("ms_per_iteration", "iterations", "distinct_images", "bare_step_over_loop_iteration")}
    return keep


def _under_a_profiler():
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ)


def other_config_legs(timeout_s=170.0):
    """Run EXTRA_LEGS (see above) and return {name: compact result or {"error": ...}}; `seconds` per leg says what the line cost."""
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FRCNN_BENCH_NO_ENTRY="1", FRCNN_BENCH_NO_NATIVE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FRCNN_BENCH_FORCE_DIST", "GPU_MAX_HW_QUEUES"):
        env.pop(k, None)
    out = {}
    for name, argv in EXTRA_LEGS:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable] + argv, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": "exit code %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
            else:
                line = json.loads(lines[-1])
                out[name] = (compact_train_leg if name.startswith("train_") else compact_inference_leg)(line)
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out after %.0f s" % timeout_s}
        except Exception as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out[name]["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to fd 1 when
    its communicator goes away; eval_dets prints progress): from here on fd 1 IS stderr for everybody, and the JSON line goes
    to the original stdout through the handle kept here."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def default_pass_shape(config, bf16_run, batch=0, streams=0, no_graph=False):
    """(images per captured pass, passes in flight) for the flags given (<= 0: not given).
    fp32 detector configs (round 5): four images per pass, four passes in flight -- the trunk's launches are latency-bound at one image
    (matrix pipe 8 % busy on its 64x64 tiles), four times taller they fill the chip: trunk 0.59 -> 0.445 ms per image in flight, end to end
    502 -> 540 img/s (B x passes: 2x6 516, 3x4 522, 4x3 540, 4x4 541, 5x3 550, 6x3 542, 8x2 534, 8x3 547; `one_image_per_pass` in the line).
    A run that names `--streams` but not `--batch` keeps one image per pass (`--streams 1`: the latency form).  One-image fp32 passes:
    12 in flight on 12 hardware queues (8 / 12 / 16 streams 472 / 481 / 480 img/s with the f16x3 engine).  bf16: eight images per pass,
    four passes (configs[3] sweep, round 3: 4 graphs 982-994 img/s, 3 975-985, 5 933-943, 2 952; on 8 hardware queues 939-944).
    configs[0] (no detector): four images per pass too since round 6 (1 x 12 / 2 x 6 / 4 x 4 / 8 x 2 / 12 x 2: 710-715 / 769 / 767 / 762-786 /
    780 img/s: conv5's 2 294 rows per image leave the split-K 64x64 form for the 256x128 tile).  Eager runs: one image per pass."""
    if batch <= 0:
        batch = (8 if bf16_run else 4) if (not no_graph and (streams <= 0 or bf16_run)) else 1
    if streams <= 0:
        streams = 12 if (not bf16_run and batch == 1) else 4
    return batch, streams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--shared-tiles", action="store_true", help="with --no-graph: the launch forms of the multi-image default run")
    ap.add_argument("--config", choices=("c2", "c4", "c1"), default="c2",
                    help="c2 = BASELINE configs[1] (headline); c4 = configs[3]; c1 = configs[0] (VGG16 RPN-only forward)")
    ap.add_argument("--split-k", choices=("auto", "on", "off"), default="auto", help="split-K conv launches for small grids")
    ap.add_argument("--no-hoist", action="store_true",
                    help="detector head in the reference's order (resample, then res5a_branch2a / branch1 on every crop) instead of "
                         "applying those two 1x1 layers once to the conv4 map")
    ap.add_argument("--dtype", choices=("config", "f32", "bf16"), default="config",
                    help="override the config's arithmetic type (off-contract: e.g. configs[1] shapes on the bf16 conv path)")
    ap.add_argument("--no-io", action="store_true",
                    help="skip the second, I/O-inclusive timing (fresh uint8 images from pinned host memory in, detections out)")
    ap.add_argument("--streams", type=int, default=0,
                    help="captured passes in flight per GPU (one hipGraph + HIP stream each); default: 4 (12 for fp32 with --batch 1)")
    ap.add_argument("--batch", type=int, default=0,
                    help="images per hipGraph (BatchedInferencePipeline: trunk at batch B, one detector-head pass over B x 300 RoIs); "
                         "default: 8 for configs[3] (bf16), 4 for configs[1] (fp32; 1 when --streams is given), 1 for configs[0]")
    ap.add_argument("--unit-tiles", default="", help="dev: tile codes for named conv layers, e.g. res5a_branch2c=26,res5b_branch2c=26")
    ap.add_argument("--conv-table", action="store_true", help="print every distinct conv launch's duration alone on the chip to stderr")
    ap.add_argument("--f32-engine", choices=("native", "bf16x6", "f16x3"), default="f16x3",
                    help="matrix path of the fp32 convolutions: native f32 MFMA; bf16x6 = the large launches on the bf16 matrix cores by exact "
                         "three-way operand splitting (csrc/conv_x6.hip, six matrix instructions per block of products); f16x3 = the same launches on "
                         "the fp16 matrix cores by a two-way split with a scaled low part (csrc/conv_h3.hip, three).  fp32-grade results either way")
    ap.add_argument("--no-train-dp", action="store_true", help="N > 1: leave the data-parallel training steps (`train_dp`) out of the line")
    ap.add_argument("--no-extra", action="store_true", help="N = 1, --config c2: leave the other BASELINE configs (`extra`: configs[0], configs[3], the "
                    "training steps of configs[2] / [4]) out of the line")
    ap.add_argument("--entry-only", action="store_true", help="run only the legs through the reference's entry point (voc_dets.get_dets_by_cls) and "
                    "print their object: what the default command starts as a child process under voc_dets.main's environment")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                      # does not return
    json_out = claim_stdout()
    if args.entry_only:
        select_config(args.config)
        pipe, _, anchors = build_pipeline()
        json_out.write(json.dumps(entry_legs(pipe, anchors, 0)) + "\n")
        json_out.flush()
        return
    extra = None
    entry_child = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.config in ("c2", "c4") and args.dtype == "config" and not args.no_graph and not args.no_io
            and os.environ.get("FRCNN_BENCH_FORCE_DIST", "0") == "0" and "FRCNN_BENCH_NO_ENTRY" not in os.environ and not _under_a_profiler()):
        entry_child = entry_legs_child(args.config)  # (a child too, and first: this process has not touched the GPU yet)
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.config == "c2" and not args.no_extra and args.dtype == "config"
            and not args.no_graph and args.streams <= 0 and args.batch <= 0 and os.environ.get("FRCNN_BENCH_FORCE_DIST", "0") == "0"
            and "FRCNN_BENCH_BACKEND" not in os.environ and not _under_a_profiler()):
        # (the driver's command line; a run that names a pass shape or a dev backend is somebody's experiment.  Under rocprofv3 the tool's
        #  preloaded library may have initialised the GPU in THIS process already: no GPU children from here)
        extra = other_config_legs()                 # children first: this process has not touched the GPU yet
    global HOIST, DTYPE, WORKLOAD
    select_config(args.config)
    bf16_run = not (DTYPE == "f32" and args.dtype in ("config", "f32"))
    args.batch, args.streams = default_pass_shape(args.config, bf16_run, args.batch, args.streams, args.no_graph)
    # ROCm maps a process's streams onto 4 hardware queues unless told otherwise; more than four images in flight need
    # a queue each or they queue behind one another (measured on MI355X: 8 streams on 8 queues 245.6 img/s, 8 streams on
    # 4 queues 241.4, 4 on 4 240.7, 4 on 8 217.0; configs[3] (bf16) is fastest with 4 on 4).  Read when the HIP runtime
    # starts, i.e. at the first device call below; an explicit setting in the environment wins.
    if args.streams > 4:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(args.streams, 16)))
    if args.dtype != "config" and args.dtype != DTYPE:
        DTYPE = args.dtype
        WORKLOAD += " [OFF-CONTRACT: run with --dtype %s]" % args.dtype
    HOIST = not args.no_hoist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one rank per GPU; FRCNN_BENCH_BACKEND=gloo (dev) lets two ranks share one GPU to exercise this path on a 1-GPU box
    backend = os.environ.get("FRCNN_BENCH_BACKEND", "nccl")
    # a rank of a node job keeps to the cores next to its GPU's PCIe slot (host staging, the launch thread and RCCL's proxy
    # thread then stay off the other socket); sysfs + one syscall, before the HIP runtime starts
    affinity = None
    if world > 1 and backend == "nccl" and "FRCNN_BENCH_NO_PIN" not in os.environ:
        from faster_rcnn_amd import dp as _dp
        affinity = _dp.pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    torch.cuda.set_device(local % torch.cuda.device_count() if world > 1 else 0)

    # the engine policy of this PROCESS (captures, the per-launch roofline pass and the parity checkers below all see the same
    # launch forms); the training legs of an N > 1 line run on the native engine
    from faster_rcnn_amd import ops as _ops
    _ops.F32_ENGINE = args.f32_engine if DTYPE == "f32" else "native"      # (a bf16 run has a handful of small f32 layers: native)
    pipe, weights, anchors = build_pipeline()
    if args.unit_tiles:
        want = dict(kv.split("=") for kv in args.unit_tiles.split(","))
        for m in (pipe.rpn.base.net, pipe.rpn.head) + ((pipe.det.head,) if hasattr(pipe, "det") else ()):
            for u in m.units():
                if u.conv in want:
                    u.tile = int(want[u.conv])
    B = args.batch                                  # images per hipGraph (1 = InferencePipeline, one image per graph)
    synth_batch = lambda first: torch.from_numpy(np.concatenate([synth_image(first + j) for j in range(B)])).cuda()
    pipe1 = pipe                                    # the one-image pipeline: what the parity checkers walk stage by stage
    if B > 1 and DEPTH == 16:
        pipe = RpnOnlyPipeline(pipe.rpn, B)
    elif B > 1:
        from faster_rcnn_amd.pipeline import BatchedInferencePipeline
        raw_dense = getattr(pipe, "raw_dense_class", None)
        pipe = BatchedInferencePipeline(pipe.rpn, pipe.det, anchors, B, max_proposals=PROPOSALS)
        pipe.raw_dense_class = raw_dense
    x = synth_batch(rank)
    S = max(1, args.streams)
    # split-K is a latency tool: measured on MI355X it gains 6-9 % with one image in flight, and with eight (f32) it costs
    # 1.2-1.4 % (round 3: 266.2 / 267.2 img/s with it, 269.9 / 270.0 with `--split-k off`: concurrency already fills the small
    # grids and the plain launches do less total work); bf16 with four in flight loses 4 %.  The fp32 default keeps it on: the
    # graphs then hold the same launch forms the roofline section times one at a time (the small-grid layers alone on the chip
    # are what split-K is for; without it every 64x64 launch also falls under ONE instantiation name, which then outweighs the
    # 3x3 head kernel as "dominant kernel" at 0.53 of peak alone -- a bookkeeping change, not a slower pipeline).
    split_k = args.split_k == "on" or (args.split_k == "auto" and B == 1 and (S == 1 or DTYPE == "f32"))
    if not args.no_graph:
        from faster_rcnn_amd.pipeline import InferencePipeline
        more = lambda: (RpnOnlyPipeline(pipe.rpn, B) if DEPTH == 16 else
                        BatchedInferencePipeline(pipe.rpn, pipe.det, anchors, B, max_proposals=PROPOSALS) if B > 1 else
                        InferencePipeline(pipe.rpn, pipe.det, anchors, max_proposals=PROPOSALS))
        pipes = [pipe] + [more() for _ in range(S - 1)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        for i, (pl, st) in enumerate(zip(pipes, streams)):
            pl.capture(HEIGHT, WIDTH, split_k=split_k, throughput=S > 1 or B > 1, **({"f32_engine": args.f32_engine} if (B == 1 or not bf16_run) else {}))
            pl._static_in.copy_(synth_batch((rank * S + i) * B))
        torch.cuda.synchronize()

        def step():
            # one step = S images, each replayed from its own hipGraph on its own HIP stream
            for pl, st in zip(pipes, streams):
                with torch.cuda.stream(st):
                    pl._graph.replay()
    else:
        S = 1
        if args.shared_tiles:
            # eager, but with the launch forms the default multi-image graphs hold (plain launches, tiles chosen for a
            # shared chip): lets the PMC passes (profile_round.sh) see those kernel instantiations one at a time
            from faster_rcnn_amd import ops

            def step():
                with ops.conv_workspace(ops.NO_SPLIT_K), ops.tile_policy(True):
                    pipe.forward_dev(x)
        else:
            step = lambda: pipe.forward_dev(x)

    # FRCNN_BENCH_FORCE_DIST=1 (dev / tests): take the multi-rank code path with ONE rank -- process group over the real "nccl"
    # backend, barriers, max-over-ranks timing, the `train_dp` leg with its collective forced on -- on a one-GPU box
    force_dist = world == 1 and os.environ.get("FRCNN_BENCH_FORCE_DIST", "0") != "0"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        from faster_rcnn_amd import dp as _dp
        _dp.FORCE_COLLECTIVE = True
    if world > 1 or force_dist:
        # the process group comes up AFTER the hipGraph captures: its watchdog thread must not touch the HIP
        # runtime while a capture is open
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend=backend)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ranks_audit = None
    if dist is not None:
        # every rank's own clock over the SAME barrier-bracketed region, gathered: the line then shows that all ranks ran (ranks_seen),
        # what each one did by itself and how far apart they were -- `value` uses the slowest (the contract's max over ranks)
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank = [float(v.item()) for v in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks_audit = {"ranks_seen": len(per_rank), "per_rank_img_s": [round(S * B * args.steps / v, 2) for v in per_rank],
                       "slowest_over_fastest_seconds": round(max(per_rank) / min(per_rank), 4),
                       "value_is": "ranks x images per step x steps / the SLOWEST rank's seconds (all ranks between the same two barriers)"}

    # ---- the same K steps on the OTHER fp32 matrix paths, when the timed graphs above run their large launches on a split engine:
    # all numbers in one line, same process, same inputs (one rank only).  `native_f32_mfma`: every convolution on
    # v_mfma_f32_32x32x2_f32; `bf16x6_exact_split` (f16x3 runs): the exact three-way bf16 split of round 4.
    native = x6_alt = None

    def per_image_outputs(pls, batch):
        """{image id: (det_bbox, det_cls, det_prob, n_dets)} of captured pipelines; image id = pipeline index * batch + position in the pass"""
        res = {}
        for i, pl in enumerate(pls):
            o = pl._static_out
            for j in range(batch):
                pick = lambda k: o[k][j] if isinstance(o[k], list) else o[k]
                res[i * batch + j] = (pick("det_bbox"), pick("det_cls"), pick("det_prob"), pick("n_dets"))
        return res

    def time_engine(engine, what, batch=None, n_streams=None):
        """The same steps with other captured passes: another matrix path (``engine``) and / or another pass shape (``batch`` images
        per pass, ``n_streams`` passes in flight); image g of this run is image g of the value run."""
        from faster_rcnn_amd.pipeline import BatchedInferencePipeline, InferencePipeline
        b = B if batch is None else batch
        ns = S if n_streams is None else n_streams
        make = lambda: (RpnOnlyPipeline(pipe.rpn, b) if DEPTH == 16 else
                        BatchedInferencePipeline(pipe.rpn, pipe.det, anchors, b, max_proposals=PROPOSALS) if b > 1 else
                        InferencePipeline(pipe.rpn, pipe.det, anchors, max_proposals=PROPOSALS))
        npipes = [make() for _ in range(ns)]
        nstreams = streams[:ns] + [torch.cuda.Stream() for _ in range(max(0, ns - len(streams)))]
        sk = args.split_k == "on" or (args.split_k == "auto" and b == 1 and (ns == 1 or DTYPE == "f32"))
        for i, pl in enumerate(npipes):
            pl.capture(HEIGHT, WIDTH, split_k=sk, throughput=ns > 1 or b > 1, f32_engine=engine)
            pl._static_in.copy_(torch.from_numpy(np.concatenate([synth_image(rank * S * B + i * b + j) for j in range(b)])).cuda())
        torch.cuda.synchronize()

        def nstep():
            for pl, st in zip(npipes, nstreams):
                with torch.cuda.stream(st):
                    pl._graph.replay()
        for _ in range(args.warmup):
            nstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nstep()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res = {"value": round(ns * b * args.steps / el, 3), "unit": "img/s", "ms_per_step": round(1e3 * el / args.steps, 4), "what": what % args.steps}
        if (b, ns) != (B, S):
            res.update({"images_per_graph": b, "graphs_in_flight": ns})
        if "det_bbox" in pipes[0]._static_out:
            # detection by detection: a (class, box) pair of the value run that the other run has too is IDENTICAL; its scores are compared.
            # (Two fp32 summation orders differ by ~1e-6 in a score or a regression: over thousands of boxes one integer box edge lands
            # on the other side of a .5, or two near-tied scores swap places in an NMS -- counted here, not hidden.)
            mine, theirs = per_image_outputs(pipes, B), per_image_outputs(npipes, b)
            common = sorted(set(mine) & set(theirs))
            n_mine = n_theirs = n_same = 0
            worst = 0.0
            for g in common:
                (ba, ca, pa, na), (bb, cb, pb, nb) = ([t.cpu().numpy() for t in mine[g]], [t.cpu().numpy() for t in theirs[g]])
                na, nb = int(na.reshape(-1)[0]), int(nb.reshape(-1)[0])
                da = {(int(c),) + tuple(int(v) for v in bx): float(p) for bx, c, p in zip(ba.reshape(-1, 4)[:na], ca.reshape(-1)[:na], pa.reshape(-1)[:na])}
                db = {(int(c),) + tuple(int(v) for v in bx): float(p) for bx, c, p in zip(bb.reshape(-1, 4)[:nb], cb.reshape(-1)[:nb], pb.reshape(-1)[:nb])}
                n_mine += len(da); n_theirs += len(db)
                for k, p in da.items():
                    if k in db:
                        n_same += 1
                        worst = max(worst, abs(p - db[k]))
            res["same_boxes_and_classes_as_value_run"] = bool(n_same == n_mine == n_theirs)
            res["detections_identical_to_value_run"] = "%d/%d" % (n_same, max(n_mine, n_theirs))
            res["max_score_difference_from_value_run"] = worst
            res["images_compared"] = len(common)
        else:                                               # configs[0]: RPN outputs only
            res["max_rpn_cls_difference_from_value_run"] = float(max((a._static_out["rpn_cls"] - b_._static_out["rpn_cls"]).abs().max() for a, b_ in zip(pipes, npipes)))
        del npipes
        torch.cuda.empty_cache()
        return res
    single = None
    if not args.no_graph and args.f32_engine != "native" and DTYPE == "f32" and world == 1 and not force_dist and "FRCNN_BENCH_NO_NATIVE" not in os.environ:
        native = time_engine("native", "the same %d steps with every fp32 convolution on v_mfma_f32_32x32x2_f32 (--f32-engine native)")
        if args.f32_engine == "f16x3":
            x6_alt = time_engine("bf16x6", "the same %d steps with the split launches on the bf16 matrix cores by EXACT three-way operand splitting, six matrix "
                                           "instructions per block of products (--f32-engine bf16x6: round 4's headline path)")
        if B > 1 and DEPTH != 16:
            # the pass shape of rounds 1-5a: ONE image per captured pass, twelve in flight -- the same images, the same engine
            single = time_engine(args.f32_engine, "the same images, ONE per captured pass, 12 passes in flight, %d steps (the headline's pass shape until round 5)", batch=1, n_streams=12)

    # ---- the same K steps again with the host on both ends (voc_dets.get_dets' contract: image in, detections out,
    # voc_dets.py:20-88): per image a FRESH uint8 BGR frame leaves pinned host memory (1.8 MB over PCIe), resnet.preprocess
    # runs on the device into the graph's input, the graph replays, and the detection records come back to pinned host
    # memory.  4*S distinct frames rotate, so no step sees the frame of the step before.  Reported beside `value`, which
    # stays the HBM-resident rate the contract asks for.
    io = None
    if not args.no_graph and not args.no_io:
        from faster_rcnn_amd import ops
        rs = np.random.RandomState(1000 + rank)
        frames = [torch.from_numpy(rs.randint(0, 256, (HEIGHT, WIDTH, 3)).astype(np.uint8)).pin_memory() for _ in range(4 * S * B)]
        dev_u8 = [torch.empty((HEIGHT, WIDTH, 3), dtype=torch.uint8, device="cuda") for _ in range(S * B)]
        # (the detector's outputs are views into ONE buffer, `det_packed`: n_dets / det_bbox / det_cls / det_prob / det_roi
        #  reach the host in a single copy and ops.split_detections() carves them out of it)
        det_keys = [k for k in ("det_packed", "n_rois") if k in pipes[0]._static_out] or ["rpn_cls", "rpn_reg"]
        per_image = lambda v: v if isinstance(v, list) else [v]       # the batched pipeline returns per-image lists of the small outputs
        host_out = [{k: [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in per_image(pl._static_out[k])] for k in det_keys} for pl in pipes]
        mean = (103.939, 116.779, 123.68)

        def step_io(it):
            for i, (pl, st) in enumerate(zip(pipes, streams)):
                with torch.cuda.stream(st):
                    for j in range(B):
                        u8 = dev_u8[i * B + j]
                        u8.copy_(frames[((it * S + i) * B + j) % len(frames)], non_blocking=True)
                        ops.preprocess_u8(u8, mean, out=pl._static_in[j:j + 1])
                    pl._graph.replay()
                    for k in det_keys:
                        for h, d in zip(host_out[i][k], per_image(pl._static_out[k])):
                            h.copy_(d, non_blocking=True)

        for it in range(args.warmup):
            step_io(it)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(args.steps):
            step_io(it)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed_io = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed_io], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_io = float(t.item())
        io = {"value": round(world * S * B * args.steps / elapsed_io, 3), "unit": "img/s", "ms_per_step": round(1e3 * elapsed_io / args.steps, 4),
              "distinct_frames": len(frames), "n_detections_last": int(host_out[0]["det_packed"][0][0].item()) if "det_packed" in det_keys else None,
              "what": "per image: uint8 BGR frame from pinned host memory -> H2D -> device preprocess -> hipGraph replay -> D2H of "
                      "%s into pinned host memory" % " / ".join("n_dets + det_cls + det_prob + det_bbox + det_roi (one packed copy)" if k == "det_packed" else k for k in det_keys)}
        # leave the graphs' inputs as the resident-input run had them (the roofline / parity sections below use pipe._static_out)
        for i, pl in enumerate(pipes):
            pl._static_in.copy_(synth_batch((rank * S + i) * B))
            pl._graph.replay()
        torch.cuda.synchronize()

    via_entry = None
    if entry_child is not None and "error" not in entry_child:
        via_entry = entry_child                     # measured in a process of its own, under the environment voc_dets.main sets up
    elif world == 1 and not force_dist and not args.no_graph and not args.no_io and DEPTH != 16 and "FRCNN_BENCH_NO_ENTRY" not in os.environ:
        via_entry = entry_legs(pipe, anchors, rank)  # (under a profiler, or the child failed: in this process, on its four hardware queues)
        if entry_child is not None:
            via_entry["child_process_error"] = entry_child["error"]

    train_dp = None
    if dist is not None and args.config in ("c2", "c4") and not args.no_train_dp:
        try:
            from faster_rcnn_amd import util as _util
            train_anchors = _util.get_anchors([128, 256, 512])
            # configs[2] at the top level (the keys round 3's readers know), configs[4] under "det_step2"
            with _ops.f32_engine("native"):
                train_dp = train_dp_leg(train_anchors, rank, world, "rpn_step1")
                torch.cuda.empty_cache()
                train_dp["det_step2"] = train_dp_leg(train_anchors, rank, world, "det_step2")
        except Exception as e:
            # a rank that fails inside the leg has left its peers inside a collective it will never join: going on to the
            # final barrier would mismatch collectives (an RCCL hang).  Say why and leave with a non-zero code:
            # the launcher (spawn_ranks' poll loop, or torch.distributed.run) takes the other ranks down at once.
            import traceback
            traceback.print_exc()
            sys.stderr.write("bench.py: rank %d failed in the train_dp leg (%s: %s); exiting so the job fails instead of hanging\n" % (rank, type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)

    out = pipe._static_out if not args.no_graph else pipe.forward_dev(x)
    first = lambda v: v[0] if isinstance(v, list) else v                # (the batched pipeline returns per-image lists)
    n_rois = int(first(out["n_rois"]).item()) if "n_rois" in out else None
    n_dets = int(first(out["n_dets"]).item()) if "n_dets" in out else None

    if rank == 0:
        try:
            roof, groups = conv_roofline(pipe, x, split_k=split_k, throughput=S > 1 or B > 1, images=B, engine=args.f32_engine)
            if args.conv_table:                     # per distinct launch: shape, count, duration alone on the chip, rate (stderr)
                for key, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"] * kv[1]["count"]):
                    fl = g["rec"]["flops"]
                    sys.stderr.write("conv %-44s M=%-7d N=%-5d K=%-5d s=%d  x%-3d %8.1f us %7.1f TFLOP/s  (%.3f ms per image)\n" % (
                        key[0], key[1], key[2], key[3], key[4], g["count"], 1e3 * g["ms"], fl / (g["ms"] * 1e-3) / 1e12, g["ms"] * g["count"] / B))
        except Exception as e:                          # the throughput line must survive a failed per-kernel pass
            roof = None
            roof_error = "%s: %s" % (type(e).__name__, e)
        line = {
            "metric": ("images/sec RPN-only forward VGG16 %dx%d" % (HEIGHT, WIDTH)) if DEPTH == 16
                      else "images/sec end-to-end (RPN+det) ResNet-%d %dx%d" % (DEPTH, HEIGHT, WIDTH),
            "value": round(world * S * B * args.steps / elapsed, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": WORKLOAD,
                       "images_per_step_per_gpu": S * B, "graphs_in_flight": S, "images_per_graph": B, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "proposals": PROPOSALS, "classes": NUM_CLASSES,
                       "pre_nms_top_n": 8000, "launch": "eager" if args.no_graph else "hipGraph replay", "split_k": bool(split_k),
                       "f32_matrix_path": ("native: v_mfma_f32_32x32x2_f32" if (args.f32_engine == "native" or DTYPE != "f32") else
                                           ("bf16x6: launches of >= %d 64x64 output tiles and >= %d columns (and the long-k small grids, split-K) multiply on v_mfma_f32_32x32x16_bf16 with every f32 operand split EXACTLY "
                                            "into three bf16 values, six exact partial products per product, f32 accumulate (csrc/conv_x6.hip; error against fp64 at the native "
                                            "kernel's level); the rest on v_mfma_f32_32x32x2_f32" % (_ops.X6_MIN_TILES, _ops.X6_MIN_COUT)) if args.f32_engine == "bf16x6" else
                                           ("f16x3: launches of >= %d 64x64 output tiles and >= %d columns (and the long-k small grids, split-K) multiply on v_mfma_f32_32x32x16_f16: each operand tensor scaled by one "
                                            "power of two into fp16's range (from a device-resident bound of max|x| the producing launch leaves behind), every value split into f16(a) and f16((a - f16(a)) * 2^11) "
                                            "(23-24 of its 24 bits), ah*bh and ah*bl + al*bh accumulated in f32 in two accumulators: THREE matrix instructions per block of products instead of six "
                                            "(csrc/conv_h3.hip; error against fp64 UNDER the native kernel's on every operand class measured, tests/test_conv_h3_gpu.py; NOT an exact split: the al*bl term, "
                                            "<= 2^-22 of a product, is dropped); the rest on v_mfma_f32_32x32x2_f32" % (_ops.X6_MIN_TILES, _ops.X6_MIN_COUT))),
                       "head_order": "no detector head" if DEPTH == 16 else
                       "res5a 1x1 layers on the conv4 map, then RoI resampling (algebraically equal, see DESIGN 5)" if HOIST else "reference order",
                       "n_rois_kept": n_rois, "n_detections": n_dets, "parallelism": "replicas x%d (no collective)" % world,
                       "rank0_cpu_affinity": affinity},
            "roofline": roof,
        }
        if native is not None:
            line["native_f32_mfma"] = native
        if x6_alt is not None:
            line["bf16x6_exact_split"] = x6_alt
        if single is not None:
            line["one_image_per_pass"] = single
        if io is not None:
            line["with_host_io"] = io
        if via_entry is not None:
            line["via_reference_entry"] = via_entry
        if ranks_audit is not None:
            line["ranks"] = ranks_audit
        if train_dp is not None:
            line["train_dp"] = train_dp
        if extra is not None:
            line["extra"] = extra
        if roof is None:
            line["roofline"] = {"bound": "mfma", "error": roof_error}
        if roof is not None and HOIST and DEPTH != 16:   # what the same image costs in the reference's layer order (res5a_branch2a / branch1 on every crop)
            rows_cols = int(out["rpn_cls"].shape[-3] * out["rpn_cls"].shape[-2])
            saved = 2.0 * 1024 * (512 + 2048) * (PROPOSALS * 49 - rows_cols) / 1e9
            roof["all_conv_launches"]["gflop_per_image_reference_order"] = round(roof["all_conv_launches"]["gflop_per_image"] + saved, 2)
        # conv FLOP actually executed per second by the whole job (all images in flight)
        if roof is not None:
            line["roofline"]["end_to_end_conv_tflops"] = round(roof["all_conv_launches"]["gflop_per_image"] * line["value"] / 1e3, 2)
        if roof is not None and (S > 1 or B > 1) and world == 1:
            try:
                roof["backbone_conv"]["in_flight"] = backbone_in_flight(pipe, S, roof["backbone_conv"]["gflop_per_image"], batch=B, engine=args.f32_engine if DTYPE == "f32" else "native")
            except Exception as e:
                roof["backbone_conv"]["in_flight"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if roof is not None and DTYPE == "bf16":
            for k in ("peak",):
                roof[k] = PEAK_BF16_TFLOPS
            roof["frac"] = round(roof["achieved"] / PEAK_BF16_TFLOPS, 4)
            roof["all_conv_launches"]["frac"] = round(roof["all_conv_launches"]["achieved"] / PEAK_BF16_TFLOPS, 4)
            roof["backbone_conv"]["frac"] = round(roof["backbone_conv"]["achieved"] / PEAK_BF16_TFLOPS, 4)
        if world == 1 and not args.no_cpu_baseline and args.config == "c1":
            line["cpu_baseline"], line["parity"] = vgg_rpn_cpu_and_parity(pipe, weights)
        if world == 1 and not args.no_cpu_baseline and args.config == "c4":
            oracle_runs = []
            line["cpu_baseline"] = cpu_baseline(weights, anchors, budget_s=10.0, runs=oracle_runs, min_images=4 if DTYPE == "bf16" else 1,
                                                alt_dense_class=getattr(pipe, "raw_dense_class", None))
            try:
                line["parity"] = full_size_parity_bf16(pipe, weights, anchors)
            except Exception as e:
                line["parity"] = {"ok": False, "error": repr(e)[:200]}
            if DTYPE == "bf16":
                try:
                    line["parity"]["e2e_vs_fp32"] = e2e_drift_bf16(pipe, weights, anchors, oracle_runs)
                except Exception as e:
                    line["parity"]["e2e_vs_fp32"] = {"ok": False, "error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            oracle_runs = []
            line["cpu_baseline"] = cpu_baseline(weights, anchors, runs=oracle_runs, alt_dense_class=getattr(pipe, "raw_dense_class", None) if DTYPE != "f32" else None)
            if DTYPE == "f32":
                try:
                    line["parity"] = full_size_parity(pipe1, weights, anchors)
                except Exception as e:                              # the checker must not cost the bench line
                    line["parity"] = {"ok": False, "error": repr(e)[:200]}
                try:
                    line["parity"]["e2e"] = e2e_parity(pipe1, weights, anchors, oracle_runs)
                except Exception as e:
                    line["parity"]["e2e"] = {"ok": False, "error": repr(e)[:300]}
            else:                                                       # configs[1] shapes on the bf16 engine (off-contract run)
                try:
                    line["parity"] = {"e2e_vs_fp32": e2e_drift_bf16(pipe, weights, anchors, oracle_runs)}
                except Exception as e:
                    line["parity"] = {"e2e_vs_fp32": {"ok": False, "error": repr(e)[:300]}}
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist is not None:
        dist.barrier()                  # rank 0 is still measuring the roofline: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
