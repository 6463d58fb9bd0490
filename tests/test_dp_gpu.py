"""GPU, world_size 2 (both ranks on cuda:0, gloo transport: this box has one GPU; on a node the same code runs one rank
per GPU over RCCL): the data-parallel training step.  Properties that hold exactly:
  * fed the SAME image, the all-reduced gradient is (g + g) * 0.5 == g bit for bit, so both ranks finish with the
    single-process weights;
  * fed DIFFERENT images, both ranks finish with identical weights (one all-reduce of the flat gradient buffer, the
    same optimiser state), and those differ from either single-image step."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, A = 112, 144, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _image(seed):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 256, (H, W, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None]


def _targets(rows, cols, seed):
    rs = np.random.RandomState(seed)
    can_use = rs.rand(1, rows, cols, A) < 0.25
    is_pos = rs.rand(1, rows, cols, A) < 0.15
    y_class = np.concatenate([can_use, is_pos], axis=3)
    sel = np.repeat(can_use & is_pos, 4, axis=3).astype(np.float32)
    tg = (rs.randn(1, rows, cols, 4 * A) * is_pos.repeat(4, axis=3)).astype(np.float32)
    return y_class, np.concatenate([sel, tg], axis=3)


def _train(img_seeds, steps=2):
    """Two SGD-momentum steps of the ResNet-50 RPN on the given image seeds (one per step) -> trained weights."""
    from faster_rcnn_amd import resnet, train
    from faster_rcnn_amd.weights import synthetic_resnet
    w0 = synthetic_resnet(50, anchors_per_loc=A, seed=7)
    base = resnet.resnet50_base(weights=w0, weight_regularizer=resnet.WEIGHT_REGULARIZER, bias_regularizer=resnet.BIAS_REGULARIZER)
    rpn = resnet.resnet50_rpn(base, anchors_per_loc=A)
    rpn.compile(train.SGD(1e-3, 0.9))
    rows, cols = resnet.get_conv_rows_cols(H, W)
    for s in range(steps):
        y_class, y_bbreg = _targets(rows, cols, 100 + img_seeds[s])
        rpn.train_on_batch(_image(img_seeds[s]), [y_class, y_bbreg])
    rpn._flush_trainer()
    names = ("rpn_conv1", "rpn_out_cls", "rpn_out_bbreg", "res4f_branch2c", "res4a_branch1")
    return {n: [np.array(a) for a in rpn.get_layer(n).get_weights()] for n in names}


def _worker(rank, world, port, same_image, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from faster_rcnn_amd import dp
    torch.cuda.set_device(0)
    dp.init_from_env(backend="gloo")
    seeds = [3, 4] if same_image else [3 + 10 * rank, 4 + 10 * rank]
    out[rank] = _train(seeds)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _run_world2(same_image):
    mgr = mp.get_context("spawn").Manager()         # (a forked manager would inherit this process's initialised HIP runtime: it has crashed)
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), same_image, out), nprocs=2, join=True)
    return out[0], out[1]


def test_dp_same_image_equals_single_process():
    single = _train([3, 4])
    a, b = _run_world2(True)
    for n in single:
        for s, x, y in zip(single[n], a[n], b[n]):
            assert np.array_equal(x, y), n
            assert np.array_equal(s, x), n


def test_dp_different_images_agree_across_ranks():
    single = _train([3, 4])
    a, b = _run_world2(False)
    for n in single:
        for s, x, y in zip(single[n], a[n], b[n]):
            assert np.array_equal(x, y), n                     # one gradient, one update, everywhere
    assert any(not np.array_equal(single[n][0], a[n][0]) for n in single)       # and it is not rank 0's own step


def _rccl_world1_worker(port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FRCNN_DP_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from faster_rcnn_amd import dp
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    assert dp.FORCE_COLLECTIVE and dp.world() == 1
    out["w"] = _train([3, 4, 5], steps=3)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_step_through_real_rccl_in_a_one_rank_group():
    """The data-parallel step's asynchronous all-reduce (started behind the last weight-gradient batch; the optimiser enqueued
    behind it once the next step's upload + frozen stages are out) through the REAL RCCL backend: a one-rank "nccl" process
    group on this box's one GPU with the collective forced on (dp.FORCE_COLLECTIVE).  The sum over one rank is the identity,
    so three steps must end bit-identical to the plain single-process steps -- a mis-ordering between RCCL's stream and the
    step's streams (the gradient buffer reduced too early, the optimiser run too early) would show in the weights."""
    single = _train([3, 4, 5], steps=3)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), out))
    p.start()
    p.join(900)
    assert p.exitcode == 0, p.exitcode
    got = out["w"]
    for n in single:
        for s_, g_ in zip(single[n], got[n]):
            assert np.array_equal(s_, g_), n
