"""oracle/e2e.py's pair comparison (the bench line's `parity.e2e` object) on hand-made detection sets: identical pairs
score a zero mAP delta with every proposal / detection matched; moving the device's most confident detection off its
pseudo ground truth shows up in `map_pair_delta`; the files go through voc_dets.write_dets and eval_dets.voc_eval."""
import numpy as np

from faster_rcnn_amd.data.voc_data_helpers import VOC_CLASS_MAPPING
from oracle import e2e


def _dets(seed, n=12):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        x1, y1 = int(rs.randint(0, 700)), int(rs.randint(0, 400))
        out.append((int(rs.randint(0, 20)), np.float32(0.95 - 0.05 * i), np.array([x1, y1, x1 + 60 + int(rs.randint(0, 100)), y1 + 50 + int(rs.randint(0, 90))], np.int64)))
    return out


def _item(name, seed, device_dets=None):
    kept = np.random.RandomState(seed).randint(0, 60, (20, 4)).astype(np.float32)
    od = _dets(seed)
    return {"name": name, "size": (1000, 600), "oracle": (kept, od), "device": (kept.copy(), od if device_dets is None else device_dets)}


def test_identical_pair_scores_zero_delta():
    res = e2e.compare([_item("a", 1), _item("b", 2)], VOC_CLASS_MAPPING)
    assert res["proposals_identical"] == "40/40" and res["detections_identical"] == "24/24"
    assert res["max_score_diff"] == 0.0 and res["map_pair_delta"] == 0.0
    assert res["map_pseudo_gt"]["oracle"] == res["map_pseudo_gt"]["device"] > 0.5          # its own top detections are the ground truth
    assert res["map_fixed_gt"]["oracle"] == res["map_fixed_gt"]["device"]


def test_moved_detection_shows_in_the_delta():
    it = _item("a", 1)
    dd = [(c, p, b.copy()) for c, p, b in it["oracle"][1]]
    c, p, b = dd[0]
    dd[0] = (c, p, b + np.array([400, 300, 400, 300]))         # the most confident detection lands elsewhere
    dd[3] = (dd[3][0], np.float32(dd[3][1] - 1e-3), dd[3][2])  # a score moves a little: still the same detection
    it["device"] = (it["device"][0][:-2], dd)                  # and two proposals are lost
    res = e2e.compare([it, _item("b", 2)], VOC_CLASS_MAPPING)
    assert res["proposals_identical"] == "38/40" and res["detections_identical"] == "23/24"
    assert abs(res["max_score_diff"] - 1e-3) < 1e-6
    assert res["map_pseudo_gt"]["device"] < res["map_pseudo_gt"]["oracle"] and res["map_pair_delta"] > 0.01
