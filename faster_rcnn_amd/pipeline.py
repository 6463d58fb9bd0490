"""Device-resident end-to-end inference: image -> proposals -> detections on ONE HIP stream.

The reference round-trips through host numpy between the RPN and the detector for every image
(det_util.py:41, 49-55, 136-158; voc_dets.py:20-88).  Here the same stages run back to back on
the GPU with fixed shapes (counts live in device memory), so the whole pass can be captured
into a hipGraph and replayed with one host call per image.

Stages (reference line -> entry point):
    rpn_model.predict_on_batch   det_util.py:41      conv engine (nets.py)
    _get_rois + valid            det_util.py:370,196 frcnn_decode_proposals
    argsort()[::-1][:8000]       det_util.py:151-153 frcnn_topk_order + frcnn_gather_candidates
    nms(max 300, .7)             det_util.py:156     frcnn_nms_i16 + frcnn_gather_rois
    detector.predict             voc_dets.py:49      frcnn_roi_crop_resize_fwd + conv engine
    argmax / decode / per-class nms  voc_dets.py:51-86   frcnn_detections
"""
import numpy as np
import torch

from . import ops


class InferencePipeline:
    def __init__(self, rpn_model, det_model, anchor_dims, stride=16, pre_nms_top_n=8000, max_proposals=300,
                 roi_batch=64, pad_to_batch=False, bg_idx=None, det_threshold=0.0):
        self.rpn, self.det = rpn_model, det_model
        self.anchor_dims = np.asarray(anchor_dims)
        self.anchor_conv = self.anchor_dims // stride
        self.stride, self.pre, self.post = stride, pre_nms_top_n, max_proposals
        self.roi_batch = roi_batch
        # strict mode scores the reference's padded duplicates too (voc_dets.py:42-46)
        self.pad_to_batch = bool(pad_to_batch)
        self.n_rois = -(-max_proposals // roi_batch) * roi_batch if pad_to_batch else max_proposals
        self.bg_idx = det_model.num_classes - 1 if bg_idx is None else bg_idx
        self.det_threshold = det_threshold
        self._graph = None

    # ------------------------------------------------------------------ stages
    def proposals_dev(self, cls, reg, rois_out=None, true_rc=None):
        """RPN outputs (device) -> (rois (n_rois,4) f32, n_keep (1,) i32, cand, keep).  ``true_rc``: the outputs are canvas-shaped and
        the image's true map is true_rc = device int32 (rows, cols) (ops.decode_proposals_canvas)."""
        if true_rc is not None:
            rois_all, valid = ops.decode_proposals_canvas(reg, self.anchor_conv, true_rc)
        else:
            rois_all, valid = ops.decode_proposals(reg, self.anchor_conv)
        scores = cls.reshape(-1)
        order, n = ops.topk_order(scores, valid, self.pre)
        cand, cand_scores = ops.gather_candidates(rois_all, scores, order, n, self.pre)
        keep, n_keep = ops.nms_sorted(cand, n, 0.7, self.post)
        rois = ops.gather_rois(cand, keep, n_keep, self.roi_batch, self.n_rois, out=rois_out)
        return rois, n_keep, cand, keep

    def forward_dev(self, x, resize_ratio=1.0, dyn=None, extents=None):
        """x: (1,H,W,3) f32 device tensor (already preprocessed).  Returns a dict of device tensors.
        ``extents`` (nets.Extents): x is a canvas holding a smaller image in its top-left corner (see BatchedInferencePipeline).
        ``dyn``: device tensor [resize_ratio, det_threshold] (f64) read by the post-process INSTEAD of the two host scalars
        (entry.DetectionEntry: one captured pass serves every image of its size); the reference's padded RoI rows are then
        scored too when the pipeline was built with ``pad_to_batch`` (voc_dets.py:42-51)."""
        ops.amax_begin()                                    # f16x3 engine: this pass's magnitude records start from zero
        cls, reg, feat = self.rpn.forward_dev(x) if extents is None else self.rpn.forward_dev(x, extents)
        rois, n_keep, cand, keep = self.proposals_dev(cls, reg, true_rc=None if extents is None else extents.level(2)[0])
        out_cls, out_reg = self.det.forward_dev(feat, rois)
        res = {"rpn_cls": cls, "rpn_reg": reg, "feat": feat, "rois": rois, "n_rois": n_keep,
               "cls": out_cls, "reg": out_reg}
        if dyn is not None:
            res.update(ops.detections_dyn(rois, n_keep, out_cls, out_reg, self.roi_batch if self.pad_to_batch else 0, self.bg_idx,
                                          float(self.stride), dyn))
        else:
            res.update(ops.detections(rois, n_keep, out_cls, out_reg, self.roi_batch, self.bg_idx, self.det_threshold,
                                      float(self.stride), float(resize_ratio)))
        _pass_status(res, res["det_packed"])
        return res

    # ------------------------------------------------------------------ hipGraph
    def capture(self, height, width, resize_ratio=1.0, warmup=2, split_k=True, throughput=False, f32_engine="native"):
        """Capture one full pass for a fixed image size into a hipGraph.

        ``f32_engine``: "native" (v_mfma_f32_32x32x2_f32), "bf16x6" (ops.f32_engine: the large fp32 launches on the bf16
        matrix cores by exact operand splitting, fp32-grade results in a different summation order) or "f16x3" (the same
        launches on the fp16 matrix cores by a two-way split with a scaled low part: half the matrix instructions; the pass
        then owns an ``ops.AmaxArena`` with the magnitude records its tensors carry).

        ``split_k``: let small-grid convs cut K over several workgroups.  It shortens ONE image's pass (the
        stage-4 / RPN layers fill 60 % of the CUs otherwise); with several graphs replaying concurrently the
        chip is already full and the plain launches do less total work, so throughput set-ups pass False.
        ``throughput``: this graph will replay beside others (several images in flight): conv launches then pick
        their tiles for a shared chip (ops.tile_policy: larger tiles, whose MFMA efficiency is better, even where
        they leave CUs idle for a launch running alone)."""
        self._static_in = torch.zeros((1, height, width, 3), dtype=torch.float32, device="cuda")
        # the graph owns its split-K workspace: graphs of several pipelines replay concurrently.  The warm-up
        # passes size it, so the capture itself allocates (and re-zeroes) nothing.
        self._conv_ws = ops.ConvWorkspace() if split_k else ops.NO_SPLIT_K
        self._amax = ops.AmaxArena() if f32_engine == "f16x3" else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), ops.f32_engine(f32_engine), \
                ops.amax_arena(self._amax):
            for _ in range(warmup):
                self.forward_dev(self._static_in, resize_ratio)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: another thread of this process (e.g. an RCCL watchdog) may call into HIP during the capture
        with no_gc(), torch.cuda.graph(self._graph, capture_error_mode="thread_local"), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), \
                ops.f32_engine(f32_engine), ops.amax_arena(self._amax):
            self._static_out = self.forward_dev(self._static_in, resize_ratio)
        return self

    def close(self):
        """Destroy the captured graph and drop its tensors NOW, in the calling thread (after the last replay has finished): nothing
        is left for a finalizer to release from a collecting thread later.  The pipeline can be captured again."""
        g = getattr(self, "_graph", None)
        if g is not None:
            torch.cuda.synchronize()
            g.reset()
        self._graph = None
        self._static_in = self._static_out = self._conv_ws = self._amax = None

    def replay_u8(self, img_u8_bgr, mean_bgr=(103.939, 116.779, 123.68)):
        """Replay on a raw (H,W,3) uint8 BGR image: 3 bytes per pixel cross PCIe, resnet.preprocess runs on the device
        (bit-identical to the host path) straight into the graph's input tensor."""
        ops.preprocess_u8(img_u8_bgr, mean_bgr, out=self._static_in)
        self._graph.replay()
        return self._static_out

    def replay(self, x=None):
        if x is not None:
            self._static_in.copy_(x)
        self._graph.replay()
        return self._static_out


def _pass_status(res, packed):
    """f16x3 passes: the OR of the status words of every magnitude record the pass used (ops.AmaxArena.status; FRCNN_H3_*: a value above
    its record's bound, a clamped value, a non-finite record) lands in word 2 of ``packed`` (the first image's det_packed: it reaches
    the host with the detections) and in res["h3_status"]; 0 = clean."""
    arena = ops._AMAX_ARENA
    if arena is not None:
        res["h3_status"] = arena.status(out=packed[2:3])


import contextlib as _contextlib
import gc as _gc


@_contextlib.contextmanager
def no_gc():
    """No cyclic-garbage collection while a stream is capturing: a collection may run the finalizers of unrelated dead objects
    (other captured graphs, their memory pools) whose HIP calls are not legal in a capturing thread (entry.DetectionEntry._capture)."""
    was_on = _gc.isenabled()
    collect_before_capture()
    _gc.disable()
    try:
        yield
    finally:
        if was_on:
            _gc.enable()


_LAST_COLLECT = [0.0]


def collect_before_capture(min_interval_s=1.0):
    """A full collection in front of a capture -- at most one per ``min_interval_s``: what protects a capture is that the collector is
    OFF while it runs (no finalizer of a dead engine's captured passes can start inside it); collecting first merely keeps the backlog
    short, and a full collection of this process's heap is ~23 ms -- more than the rest of a DetectionEntry capture, which a list of
    mixed image sizes makes per geometry (scripts/dev/r6_capture_cost.py)."""
    import time as _time
    now = _time.monotonic()
    if now - _LAST_COLLECT[0] >= min_interval_s:
        _gc.collect()
        _LAST_COLLECT[0] = _time.monotonic()


import os as _os
# dev knob: False keeps the per-image stages of a batched pass on the pass's own stream
PARALLEL_BRANCHES = _os.environ.get("FRCNN_PAR_BRANCHES", "0") != "0"


class BatchedInferencePipeline(InferencePipeline):
    """B images per pass (bf16 conv path, configs[3]): the backbone and the RPN heads run at batch B (GEMMs B times taller),
    proposal selection and the detection post-process stay per image (the same kernels on each image's slice), and the
    detector head makes ONE pass over all B x n_rois RoIs (frcnn_roi_crop_resize_fwd_bf16_batch puts every image's crops in
    one position-major tensor).  One hipGraph replay = B images; several such graphs may replay concurrently.

    Why: one 600x1500 image gives the ResNet-101 trunk 94 launches of 28-56 row tiles (M = 3 572) for 256 CUs and the head
    GEMMs 115 row tiles; batched, every launch fills the chip and the big direct-to-LDS tiles apply everywhere
    (scripts/c4_stage_times.py: trunk 1.47 -> 0.63 ms per image at B = 8, head 0.66 -> 0.56).  Per output row the arithmetic is
    that of the per-image pipeline launched without split-K (a row's k order does not depend on the GEMM's height)."""

    def __init__(self, rpn_model, det_model, anchor_dims, batch, **kw):
        super().__init__(rpn_model, det_model, anchor_dims, **kw)
        self.batch = int(batch)
        assert hasattr(det_model.head, "forward_batched") and getattr(det_model.head, "hoist", False), "batched pipeline: a ResNet head in the hoisted order"
        # the per-image stages (proposal selection: ~10 short dependent launches; detection post-process: one workgroup) of the
        # B images are independent chains.  FRCNN_PAR_BRANCHES=1 runs each on its own stream, forked from and joined into the
        # pass's stream, so a captured graph holds them as B parallel branches: ONE graph in flight gains (B = 8: 663 -> 740
        # img/s on configs[3]) but several such graphs replaying concurrently lose (B = 4 x 4 graphs: 856 -> 627: the runtime
        # serialises branchy graphs against each other), and 3-4 plain graphs in flight hide the chains behind each other's
        # convolutions anyway (856-878 img/s).  Off by default.
        self._side = [torch.cuda.Stream() for _ in range(self.batch)] if self.batch > 1 and PARALLEL_BRANCHES else None

    def _fan_out(self, fn):
        """fn(i) for every image, image i on its own side stream between a fork from and a join into the current stream."""
        if self._side is None:
            return [fn(i) for i in range(self.batch)]
        main = torch.cuda.current_stream()
        outs = []
        for i, st in enumerate(self._side):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(fn(i))
        for st in self._side:
            main.wait_stream(st)
        return outs

    def forward_dev(self, x, resize_ratio=1.0, dyn=None, extents=None):
        """``extents`` (nets.Extents, round 6): x holds B CANVASES -- images of different true sizes, each in the top-left corner of its
        canvas, zeros elsewhere; the trunk zeroes what a 3x3 convolution would read beyond an image's extent, the proposals of image i
        come from its true map only.  Inside an image's extent every tensor is what a pass of that image's own size computes (up to the
        launch forms the policy picks for the canvas's row count).
        x: (B,H,W,3) f32 device tensor.  Returns a dict: batch tensors with a leading image axis (rpn_cls, rpn_reg, feat,
        rois (B,n_rois,4), cls (B,n_rois,C), reg) and per-image LISTS of the small outputs (n_rois, n_dets, det_packed, det_bbox,
        det_cls, det_prob, det_roi: entry i is image i's tensor, exactly what InferencePipeline returns for one image).
        ``dyn``: (B,2) f64 device tensor, row i = image i's [resize_ratio, det_threshold] (InferencePipeline.forward_dev)."""
        B = self.batch
        assert x.shape[0] == B
        ops.amax_begin()
        cls, reg, feat = self.rpn.forward_dev(x) if extents is None else self.rpn.forward_dev(x, extents)
        rois = torch.empty((B * self.n_rois, 4), dtype=torch.float32, device="cuda")
        n_keep = self._fan_out(lambda i: self.proposals_dev(cls[i], reg[i], rois_out=rois[i * self.n_rois:(i + 1) * self.n_rois],
                                                            true_rc=None if extents is None else extents.level(2)[i])[1])
        out_cls, out_reg = self.det.head.forward_batched(feat, rois, self.n_rois)
        res = {"rpn_cls": cls, "rpn_reg": reg, "feat": feat, "rois": rois.view(B, self.n_rois, 4), "n_rois": n_keep,
               "cls": out_cls.view(B, self.n_rois, -1), "reg": out_reg.view(B, self.n_rois, -1)}
        if dyn is not None:
            dets = self._fan_out(lambda i: ops.detections_dyn(rois[i * self.n_rois:(i + 1) * self.n_rois], n_keep[i], res["cls"][i], res["reg"][i],
                                                              self.roi_batch if self.pad_to_batch else 0, self.bg_idx, float(self.stride), dyn[i]))
        else:
            dets = self._fan_out(lambda i: ops.detections(rois[i * self.n_rois:(i + 1) * self.n_rois], n_keep[i], res["cls"][i], res["reg"][i], self.roi_batch,
                                                          self.bg_idx, self.det_threshold, float(self.stride), float(resize_ratio)))
        for k in dets[0]:
            res[k] = [d[k] for d in dets]
        _pass_status(res, res["det_packed"][0])
        return res

    def capture(self, height, width, resize_ratio=1.0, warmup=2, split_k=False, throughput=True, f32_engine=None):
        """``f32_engine``: None = the ambient ops.F32_ENGINE scope (bf16 models: their few f32 layers); an fp32 model names its engine
        like InferencePipeline.capture does ("f16x3": the pass owns an ops.AmaxArena sized for B images' records)."""
        engine = ops.F32_ENGINE if f32_engine is None else f32_engine
        self._static_in = torch.zeros((self.batch, height, width, 3), dtype=torch.float32, device="cuda")
        self._conv_ws = ops.ConvWorkspace() if split_k else ops.NO_SPLIT_K
        self._amax = ops.AmaxArena(192 + 16 * self.batch) if engine == "f16x3" else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), ops.f32_engine(engine), ops.amax_arena(self._amax):
            for _ in range(warmup):
                self.forward_dev(self._static_in, resize_ratio)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(self._graph, capture_error_mode="thread_local"), ops.conv_workspace(self._conv_ws), ops.tile_policy(throughput), \
                ops.f32_engine(engine), ops.amax_arena(self._amax):
            self._static_out = self.forward_dev(self._static_in, resize_ratio)
        return self
