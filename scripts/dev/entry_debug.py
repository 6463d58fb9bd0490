import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_entry_gpu import *
from faster_rcnn_amd import resnet, util, entry, voc_dets, nets, ops
from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from faster_rcnn_amd.det_util import DetTrainingManager
from faster_rcnn_amd.pipeline import InferencePipeline
from faster_rcnn_amd.weights import calibrate_classifier, synthetic_resnet
anchors = util.get_anchors([128, 256, 512])
w = synthetic_resnet(50, anchors_per_loc=9, num_classes=21, seed=1)
rpn = resnet.resnet50_rpn(resnet.resnet50_base(weights=w), include_conv=True, anchors_per_loc=9)
det = resnet.resnet50_classifier(64, 21, weights=w)
if len(sys.argv) > 1:
    x = resnet.preprocess(synth_pixels(320, 480, 99))[None].astype(np.float32)
    out = InferencePipeline(rpn, det, anchors).forward_dev(torch.from_numpy(x).cuda())
    n = int(out["n_rois"].item())
    det.get_layer("dense_class_21").set_weights(calibrate_classifier(w, 21, out["cls"][:n].cpu().numpy()))
mgr = DetTrainingManager(rpn_model=rpn, class_mapping=VOC_CLASS_MAPPING, preprocess_func=resnet.preprocess, anchor_dims=anchors)
image = named_image("synth", synth_pixels(600, 1000, 7))
eng = entry.for_models(mgr, det, 64, 16, 1)
t = eng.submit(image, 1.0, 0.0)
s = t.slot
n, dets = eng.collect(t)
fo = {k: v.clone() for k, v in s.out.items()}
conv_out, rois = mgr.get_det_inputs(image)
print("n fast", n, "eager", len(rois), "fast dets", len(dets))
print("rois equal", np.array_equal(fo["rois"].cpu().numpy()[:n], rois.astype(np.float32)))
padded = voc_dets._pad_rois(np.asarray(rois, dtype=np.float32), 64)
print("padded equal", np.array_equal(fo["rois"].cpu().numpy(), padded))
print("feat maxdiff", float((fo["feat"].cpu() - torch.from_numpy(conv_out)).abs().max()))
oc, orr = det.forward_dev(nets.to_device_image(conv_out), torch.from_numpy(padded).cuda())
print("cls maxdiff", float((oc - fo["cls"]).abs().max()), "reg", float((orr - fo["reg"]).abs().max()))
print("argmax equal", bool((oc.argmax(1) == fo["cls"].argmax(1)).all()))
n_rows = torch.tensor([len(padded)], dtype=torch.int32, device="cuda")
res = ops.detections(torch.from_numpy(padded).cuda(), n_rows, oc, orr, 64, 20, 0.0, 16, 1.0)
print("eager n_dets", int(res["n_dets"]), "fast", int(fo["n_dets"]))
res2 = ops.detections(fo["rois"], n_rows, fo["cls"], fo["reg"], 64, 20, 0.0, 16, 1.0)
print("eager kernel on fast inputs", int(res2["n_dets"]))
dyn = torch.tensor([1.0, 0.0], dtype=torch.float64, device="cuda")
res3 = ops.detections_dyn(fo["rois"], fo["n_rois"], fo["cls"], fo["reg"], 64, 20, 16, dyn)
print("dyn kernel eager call", res3["det_packed"][:4].tolist())
res4 = ops.detections_dyn(fo["rois"], n_rows, fo["cls"], fo["reg"], 0, 20, 16, dyn)
print("dyn kernel batch 0 rows 320", res4["det_packed"][:4].tolist())
