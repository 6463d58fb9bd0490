"""dev: which detections differ between the f32 oracle end to end and the f32 device end to end (calibrated head)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from oracle import e2e, np_ref
from oracle.keras_ref import KerasGraphs
pipe, w, anchors = bench.build_pipeline()
k, b = w["dense_class_21"]
print("dense_class kernel rms", float(np.sqrt((k ** 2).mean())), "(drawn with std 0.03)")
g = KerasGraphs(w, torch.float32)
for seed in (100, 101):
    x = bench.synth_image(seed)
    ok, od = e2e.oracle_detect(g, x, anchors, 21)
    out = pipe.forward_dev(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    n = int(out["n_rois"])
    dk = out["rois"].cpu().numpy()[:n]
    print("seed", seed, "proposals equal", np.array_equal(dk, ok.astype(np.float32)))
    # oracle classifier on the device's feature map / rois -> compare logits row by row
    with torch.no_grad():
        feat = g.resnet_base(x, 50)
        o_cls, o_reg = g.resnet_classifier(feat, np_ref.pad_rois(ok.astype(np.float32), 64), 21, 50)
    o_cls, o_reg = o_cls.numpy().reshape(-1, 21)[:n], o_reg.numpy().reshape(-1, 80)[:n]
    d_cls, d_reg = out["cls"].cpu().numpy()[:n], out["reg"].cpu().numpy()[:n]
    am_o, am_d = o_cls.argmax(1), d_cls.argmax(1)
    print("  argmax flips", int((am_o != am_d).sum()), "of", n, " max |dcls|", float(np.abs(o_cls - d_cls).max()), "max |dreg|", float(np.abs(o_reg - d_reg).max()))
    top2 = np.sort(o_cls, axis=1)[:, -2:]
    print("  smallest top-2 prob gaps", np.sort(top2[:, 1] - top2[:, 0])[:5])
    dd = e2e.device_detect(pipe, x)[1]
    so = {(c,) + tuple(int(v) for v in bb): float(p) for c, p, bb in od}
    sd = {(c,) + tuple(int(v) for v in bb): float(p) for c, p, bb in dd}
    only_o = [k for k in so if k not in sd]; only_d = [k for k in sd if k not in so]
    print("  dets oracle", len(od), "device", len(dd), "only oracle", len(only_o), "only device", len(only_d))
    for k in only_o[:8]:
        near = [q for q in only_d if q[0] == k[0] and max(abs(a - b) for a, b in zip(q[1:], k[1:])) <= 2]
        print("    oracle-only", k, round(so[k], 5), "near device:", near[:2])
x, ratio, size = bench.real_voc_image()
ok, od = e2e.oracle_detect(g, x, anchors, 21, 50, 300, ratio)
dk, dd = e2e.device_detect(pipe, x, ratio)
print("real: proposals", len(ok), len(dk), "equal", np.array_equal(dk, ok.astype(np.float32)))
so = {(c,) + tuple(int(v) for v in bb): float(p) for c, p, bb in od}
sd = {(c,) + tuple(int(v) for v in bb): float(p) for c, p, bb in dd}
only_o = [k for k in so if k not in sd]; only_d = [k for k in sd if k not in so]
print("  dets oracle", len(od), "device", len(dd), "only oracle", len(only_o), "only device", len(only_d))
for k in only_o[:10]:
    near = [(q, round(sd[q], 5)) for q in only_d if q[0] == k[0] and max(abs(a - b) for a, b in zip(q[1:], k[1:])) <= 3]
    print("    oracle-only", k, round(so[k], 5), "near device:", near[:2])
out = pipe.forward_dev(torch.from_numpy(x).cuda(), ratio)
n = int(out["n_rois"])
with torch.no_grad():
    feat = g.resnet_base(x, 50)
    o_cls, o_reg = g.resnet_classifier(feat, np_ref.pad_rois(ok.astype(np.float32), 64), 21, 50)
o_cls, o_reg = o_cls.numpy().reshape(-1, 21)[:n], o_reg.numpy().reshape(-1, 80)[:n]
d_cls, d_reg = out["cls"].cpu().numpy()[:n], out["reg"].cpu().numpy()[:n]
print("  argmax flips", int((o_cls.argmax(1) != d_cls.argmax(1)).sum()), "max|dcls|", float(np.abs(o_cls - d_cls).max()), "max|dreg|", float(np.abs(o_reg - d_reg).max()),
      "feat max", float(feat.abs().max()), "reg max", float(np.abs(o_reg).max()))
top2 = np.sort(o_cls, axis=1)[:, -2:]
print("  smallest top-2 gaps", np.sort(top2[:, 1] - top2[:, 0])[:5], "max prob", float(o_cls.max()))
import collections
print("class histogram (device, last synthetic):", sorted(collections.Counter(int(c) for c, _, _ in e2e.device_detect(pipe, bench.synth_image(101))[1]).items()))
