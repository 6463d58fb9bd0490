"""Dev: where do the device's proposals leave the numpy selection at configs[3] size (R101 600x1500 bf16, 18 anchors)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_amd import resnet, util, ops
from faster_rcnn_amd.pipeline import InferencePipeline
from faster_rcnn_amd.weights import synthetic_resnet
from oracle import np_ref

anchors = util.get_anchors([16, 32, 64, 128, 256, 512]); A, C = 18, 10
w = synthetic_resnet(101, anchors_per_loc=A, num_classes=C, seed=1)
base = resnet.resnet101_base(weights=w, dtype="bf16")
rpn = resnet.resnet101_rpn(base, include_conv=True, anchors_per_loc=A)
det = resnet.resnet101_classifier(300, C, weights=w, dtype="bf16")
pipe = InferencePipeline(rpn, det, anchors, max_proposals=300)
x = (np.random.RandomState(0).randint(0, 256, (600, 1500, 3)).astype(np.float64) - np.array([103.939, 116.779, 123.68]))[None].astype(np.float32)
cls, reg, feat = rpn.forward_dev(torch.from_numpy(x).cuda())
rois, n_keep, cand, keep = pipe.proposals_dev(cls, reg)
torch.cuda.synchronize()
cls_np, reg_np = cls.float().cpu().numpy().reshape(1, 38, 94, A), reg.float().cpu().numpy().reshape(1, 38, 94, 4 * A)
print("cls dtype", cls.dtype, "unique scores", len(np.unique(cls_np)), "of", cls_np.size, "max", cls_np.max(), "count==max", (cls_np == cls_np.max()).sum())
dev_boxes, dev_valid = ops.decode_proposals(reg, pipe.anchor_conv)
dev_boxes = dev_boxes.cpu().numpy()
v = np.nonzero(np_ref.valid_mask(dev_boxes))[0]
probs = cls_np.reshape(-1)[v]
order = np_ref.score_order(probs, 8000)
ref_cand = dev_boxes[v][order].astype("int16")
dcand = cand.cpu().numpy()
n = min(len(ref_cand), len(dcand))
neq = np.nonzero((ref_cand[:n] != dcand[:n]).any(axis=1))[0]
print("candidates: ref", len(ref_cand), "dev", len(dcand), "first mismatches", neq[:10], "count", len(neq))
if len(neq):
    i = neq[0]
    print("at", i, "ref", ref_cand[i], probs[order][i], "dev", dcand[i], "orig idx ref", v[order][i])
    print("scores around", probs[order][max(0, i - 3):i + 4])
kept, kprobs, pick = np_ref.nms(ref_cand, probs[order], 0.7, 300)
dk = keep.cpu().numpy()[:int(n_keep.item())]
print("n_keep dev", int(n_keep.item()), "ref", len(kept), "pick equal", np.array_equal(np.asarray(pick), dk))
if not np.array_equal(np.asarray(pick), dk):
    j = np.nonzero(np.asarray(pick)[:min(len(pick), len(dk))] != dk[:min(len(pick), len(dk))])[0]
    print("first pick mismatch at", j[:5], "ref", np.asarray(pick)[j[:5]], "dev", dk[j[:5]])
    if len(j):
        a, b = np.asarray(pick)[j[0]], dk[j[0]]
        print("ref box", ref_cand[a], "dev box", dcand[b])
