"""fp32 passes over B images (pipeline.BatchedInferencePipeline with an f32 ResNet: trunk and RPN heads at batch B, one detector-head
pass over all B x n RoIs, frcnn_roi_crop_resize_fwd_batch) against the one-image pipeline on each image alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_batched_roi_resize_f32_equals_per_image():
    """frcnn_roi_crop_resize_fwd_batch, f32 output and plane output, both layouts, fill / ReLU, an empty and an outside RoI: RoI r crops
    image r // n_per_img's map exactly as the single-image calls do."""
    from faster_rcnn_amd import ops
    rs = np.random.RandomState(2)
    B, R, Cc, Cf, n = 3, 9, 13, 64, 7
    feat = torch.from_numpy(rs.randn(B, R, Cc, Cf).astype(np.float32)).cuda()
    x1, y1 = rs.randint(0, Cc - 2, B * n), rs.randint(0, R - 2, B * n)
    rois = np.stack([x1, y1, x1 + 1 + rs.randint(0, 6, B * n), y1 + 1 + rs.randint(0, 5, B * n)], 1).astype(np.float32)
    rois[4] = [3, 3, 3, 8]                                             # an empty RoI -> the fill vector
    rois[9] = [2, 1, Cc + 2, 4]                                        # out of the map -> fill
    rois_d = torch.from_numpy(rois).cuda()
    fill = torch.from_numpy(rs.randn(Cf).astype(np.float32)).cuda()
    for layout in (0, 1):
        for relu in (False, True):
            got = ops.roi_crop_resize(feat, rois_d, 7, fill=fill, relu=relu, layout=layout, n_per_img=n)
            for i in range(B):
                want = ops.roi_crop_resize(feat[i], rois_d[i * n:(i + 1) * n], 7, fill=fill, relu=relu, layout=layout)
                part = got[:, :, i * n:(i + 1) * n] if layout else got[i * n:(i + 1) * n]
                assert torch.equal(part, want)
    with ops.f32_engine("f16x3"):                                      # planes: one exponent for the whole batch (the bound over all maps)
        feat._amax = ops.amax_of(feat)
        got = ops.roi_crop_resize(feat, rois_d, 7, fill=fill, relu=True, layout=1, planes_out=True, n_per_img=n)
        assert isinstance(got, ops.PlaneTensor)
        f32 = ops.roi_crop_resize(feat, rois_d, 7, fill=fill, relu=True, layout=1, n_per_img=n)
        e = int(got.exponent.item())
        back = (got.planes[0].float() + got.planes[1].float() / 2048.0) * 2.0 ** -e
        assert (back - f32).abs().max().item() <= 2.0 ** -22 * float(max(feat.abs().max().item(), fill.abs().max().item()))


@pytest.mark.parametrize("engine", ["native", "f16x3"])
def test_batched_f32_pipeline_equals_per_image_pipeline(engine):
    """Both without split-K.  native: a row's k order does not depend on how tall the GEMM is -- every output bit-identical.  f16x3: a
    tensor's scale comes from the bound over the WHOLE batch, so values within 2^-29 of a tensor's largest may round differently: the
    continuous outputs to 1e-5, proposals, classes and boxes identical."""
    from faster_rcnn_amd import ops, resnet, util
    from faster_rcnn_amd.pipeline import BatchedInferencePipeline, InferencePipeline
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors([32, 64, 128])
    A, C, B, n = len(anchors), 10, 3, 40
    w = synthetic_resnet(50, anchors_per_loc=A, num_classes=C, seed=9)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w), include_conv=True, anchors_per_loc=A)
    det = resnet.resnet50_classifier(n, C, weights=w)
    rs = np.random.RandomState(5)
    x = torch.from_numpy((rs.randint(0, 256, (B, 176, 240, 3)) - 110.0).astype(np.float32)).cuda()
    with ops.conv_workspace(ops.NO_SPLIT_K), ops.f32_engine(engine):
        bp = BatchedInferencePipeline(rpn, det, anchors, B, max_proposals=n)
        out = bp.forward_dev(x)
        single = InferencePipeline(rpn, det, anchors, max_proposals=n)
        per = [single.forward_dev(x[i:i + 1].contiguous()) for i in range(B)]
    torch.cuda.synchronize()
    close = (lambda a, b: torch.equal(a, b)) if engine == "native" else (lambda a, b: bool((a - b).abs().max() <= 1e-5 * max(1.0, float(b.abs().max()))))
    for i, o in enumerate(per):
        assert close(out["rpn_cls"][i], o["rpn_cls"][0]) and close(out["rpn_reg"][i], o["rpn_reg"][0]) and close(out["feat"][i], o["feat"][0]), i
        assert int(out["n_rois"][i]) == int(o["n_rois"]) > 0 and torch.equal(out["rois"][i], o["rois"])
        assert close(out["cls"][i], o["cls"]) and close(out["reg"][i], o["reg"])
        assert int(out["n_dets"][i]) == int(o["n_dets"]) > 0
        for k in ("det_bbox", "det_cls", "det_roi"):
            assert torch.equal(out[k][i], o[k]), k
        assert close(out["det_prob"][i], o["det_prob"])
    # from a captured graph, fed another batch
    bp.capture(176, 240, f32_engine=engine)
    x2 = torch.from_numpy((rs.randint(0, 256, (B, 176, 240, 3)) - 110.0).astype(np.float32)).cuda()
    bp._static_in.copy_(x2)
    bp._graph.replay()
    torch.cuda.synchronize()
    with ops.conv_workspace(ops.NO_SPLIT_K), ops.f32_engine(engine), ops.tile_policy(True):      # (the captured pass's launch forms: its tiles are chosen
        want = bp.forward_dev(x2)                                                                #  for a shared chip, and which tensors travel as planes follows the tile)
    torch.cuda.synchronize()
    for i in range(B):
        assert torch.equal(bp._static_out["det_bbox"][i], want["det_bbox"][i]) and torch.equal(bp._static_out["det_prob"][i], want["det_prob"][i])
        assert torch.equal(bp._static_out["cls"][i], want["cls"][i])
    bp.close()


def test_batched_roi_resize_refuses_malformed_calls():
    """frcnn_roi_crop_resize_fwd_batch: exactly one of the two outputs, a positive RoI count per image, channels in fours."""
    import ctypes
    from faster_rcnn_amd import _lib, ops
    feat = torch.zeros((2, 5, 6, 8), device="cuda")
    rois = torch.tensor([[0, 0, 2, 2]] * 4, dtype=torch.float32, device="cuda")
    out = torch.zeros((4, 7, 7, 8), device="cuda")
    pt = ops.PlaneTensor((4, 7, 7, 8))
    yp = _lib.H3Planes(planes=pt.planes.data_ptr(), exponent=pt.exponent.data_ptr())
    call = lambda n_per, C, o, p: _lib.call("frcnn_roi_crop_resize_fwd_batch", ops._p(feat), 5, 6, C, ops._p(rois), 4, n_per, 7, None, 0, 0, o, p, None)
    for bad in ((2, 8, None, None), (2, 8, ops._p(out), ctypes.byref(yp)), (0, 8, ops._p(out), None), (2, 6, ops._p(out), None)):
        with pytest.raises(_lib.FrcnnError):
            call(*bad)
    call(2, 8, ops._p(out), None)
    call(2, 8, None, ctypes.byref(yp))
    torch.cuda.synchronize()
