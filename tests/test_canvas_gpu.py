"""Padded canvases (round 6, VERDICT r5 item 6: geometry as a VALUE): images of different true sizes in ONE pass of fixed shape.  Each
image sits at offset (h & 1, w & 1) of a canvas with EVEN sides (the offset stands for the extra zero row / column SAME padding at
stride 2 puts in front of an odd side under conv1's window, resnet.py:408); ops.zero_outside restores the zeros the reference's padding
reads wherever a 3x3 convolution follows; proposals come from the true map only (frcnn_decode_proposals_canvas).  Inside an image's
extent every tensor must be what a pass of the image's own size computes: compared here against exactly that pass, stage by stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MEAN = (103.939, 116.779, 123.68)


@pytest.fixture(scope="module")
def nets50():
    from faster_rcnn_amd import resnet, util
    from faster_rcnn_amd.weights import synthetic_resnet
    anchors = util.get_anchors([128, 256, 512])
    w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=11)
    rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w), include_conv=True, anchors_per_loc=9)
    det = resnet.resnet50_classifier(64, 21, weights=w)
    return rpn, det, anchors


def _u8(h, w, seed):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)).cuda()


def test_canvas_kernels_known_answers():
    """frcnn_preprocess_u8_canvas = frcnn_preprocess_u8 in the corner, zeros outside; frcnn_zero_outside zeroes exactly the cells at or
    beyond each image's extent (f32 and bf16); frcnn_decode_proposals_canvas = frcnn_decode_proposals of the true map, cell by cell."""
    from faster_rcnn_amd import ops
    img = _u8(37, 53, 1)
    canvas = torch.full((1, 48, 64, 3), 7.0, dtype=torch.float32, device="cuda")
    ops.preprocess_u8_canvas(img, MEAN, canvas, offset=(0, 0))
    want = ops.preprocess_u8(img, MEAN)
    assert torch.equal(canvas[0, :37, :53], want[0]) and float(canvas[0, 37:].abs().max()) == 0.0 and float(canvas[0, :, 53:].abs().max()) == 0.0
    canvas.fill_(7.0)
    ops.preprocess_u8_canvas(img, MEAN, canvas)                  # odd sides: offset (1, 1) by default
    assert torch.equal(canvas[0, 1:38, 1:54], want[0])
    canvas[0, 1:38, 1:54] = 0.0
    assert float(canvas.abs().max()) == 0.0
    from faster_rcnn_amd._lib import FrcnnError
    with pytest.raises(FrcnnError):
        ops.preprocess_u8_canvas(img, MEAN, canvas, offset=(12, 0))   # 37 + 12 > 48
    for dt in (torch.float32, torch.bfloat16):
        x = torch.ones((2, 10, 12, 32), dtype=dt, device="cuda")
        hw = torch.tensor([[7, 9], [10, 4]], dtype=torch.int32, device="cuda")
        ops.zero_outside(x, hw)
        m = torch.zeros((2, 10, 12), dtype=torch.bool, device="cuda")
        m[0, :7, :9] = True
        m[1, :10, :4] = True
        assert torch.equal(x[..., 0] != 0, m) and torch.equal(x[..., 31] != 0, m)
    rs = np.random.RandomState(2)
    anchors = np.array([[8, 8], [5, 11], [11, 5]])
    R, C, Rc, Cc = 9, 13, 12, 16
    reg_true = (rs.randn(R, C, 12) * 0.5).astype(np.float32)
    reg_canvas = (rs.randn(Rc, Cc, 12) * 0.5).astype(np.float32)
    reg_canvas[:R, :C] = reg_true
    r0, v0 = ops.decode_proposals(torch.from_numpy(reg_true).cuda(), anchors)
    r1, v1 = ops.decode_proposals_canvas(torch.from_numpy(reg_canvas).cuda(), anchors, torch.tensor([R, C], dtype=torch.int32, device="cuda"))
    r1, v1 = r1.reshape(Rc, Cc, 3, 4), v1.reshape(Rc, Cc, 3)
    assert torch.equal(r1[:R, :C].reshape(-1, 4), r0) and torch.equal(v1[:R, :C].reshape(-1), v0)
    assert int(v1[R:].sum()) == 0 and int(v1[:, C:].sum()) == 0


@pytest.mark.parametrize("engine", ["native", "f16x3"])
@pytest.mark.parametrize("sizes,canvas", [([(320, 480), (304, 450)], (320, 480)), ([(321, 479), (289, 451)], (322, 480)), ([(306, 451), (319, 417)], (320, 512))])
def test_canvas_pass_equals_the_passes_of_the_true_sizes(nets50, engine, sizes, canvas):
    """Two images of different sizes (even and odd sides, mixed in one pass: the stem's SAME padding differs) in one two-canvas pass against each image's
    own pass at its true size, same engine: conv4 map and RPN outputs inside the true extents to 1e-5 (bit-equal on the native
    engine when the launch forms coincide), identical proposals, identical detections, scores to 1e-5."""
    from faster_rcnn_amd import nets, ops
    from faster_rcnn_amd.pipeline import BatchedInferencePipeline, InferencePipeline
    rpn, det, anchors = nets50
    Hc, Wc = canvas
    imgs = [_u8(h, w, 40 + k) for k, (h, w) in enumerate(sizes)]
    x = torch.empty((2, Hc, Wc, 3), dtype=torch.float32, device="cuda")
    ext = nets.Extents(2)
    for i, (im, (h, w)) in enumerate(zip(imgs, sizes)):
        assert Hc % 2 == 0 and Wc % 2 == 0 and h + (h & 1) <= Hc and w + (w & 1) <= Wc
        ops.preprocess_u8_canvas(im, MEAN, x[i])
        ext.set(i, h, w)
    ext.upload()
    dyn = torch.tensor([[1.0, 0.0], [1.0, 0.0]], dtype=torch.float64, device="cuda")
    arena = ops.AmaxArena() if engine == "f16x3" else None
    with ops.f32_engine(engine), ops.conv_workspace(ops.NO_SPLIT_K), ops.amax_arena(arena):
        out = BatchedInferencePipeline(rpn, det, anchors, 2, max_proposals=300).forward_dev(x, dyn=dyn, extents=ext)
        if arena is not None:
            assert int(out["h3_status"].item()) == 0
        for i, (im, (h, w)) in enumerate(zip(imgs, sizes)):
            one = InferencePipeline(rpn, det, anchors, max_proposals=300).forward_dev(ops.preprocess_u8(im, MEAN), dyn=dyn[i])
            R, C = nets.Extents.levels_of(h, w)[2]
            assert tuple(one["feat"].shape[1:3]) == (R, C)
            for k in ("feat", "rpn_cls", "rpn_reg"):
                a, b = out[k][i][:R, :C].float(), one[k][0].float()
                assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), (k, i, float((a - b).abs().max()))
            outside = torch.cat([out["feat"][i][R:].reshape(-1), out["feat"][i][:, C:].reshape(-1)])
            assert outside.numel() == 0 or float(outside.abs().max()) == 0.0
            n = int(one["n_rois"].item())
            assert int(out["n_rois"][i].item()) == n > 0
            assert torch.equal(out["rois"][i][:n], one["rois"][:n])
            nd = int(one["n_dets"].item())
            assert int(out["n_dets"][i].item()) == nd
            assert torch.equal(out["det_bbox"][i][:nd], one["det_bbox"][:nd]) and torch.equal(out["det_cls"][i][:nd], one["det_cls"][:nd])
            assert float((out["det_prob"][i][:nd] - one["det_prob"][:nd]).abs().max()) <= 1e-5
