#!/usr/bin/env python3
"""BASELINE's "box mAP delta vs ref" for the bf16 configs over MANY frames: the plain fp32 oracle end to end
against the bf16 device end to end on N synthetic frames (default 32; bench.py's own leg uses the 4-8 its cpu_baseline budget
allows and reads 0.09-0.21 from sample to sample), the pair's mAP delta with a bootstrap over frames (resampled with replacement):
mean, standard deviation and the 2.5 / 97.5 percentiles.  Rank 0 prints ONE JSON object; profiles/round5_drift_bf16_*.json are
its outputs, and bench.DRIFT_MAP_BARS / tests/test_configs_full_size_gpu.py take their bound from them.

    python scripts/drift_bf16.py [--config c4|c2] [--frames 32] [--boot 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("c4", "c2"), default="c4", help="c4: configs[3] (ResNet-101 600x1500 bf16); c2: configs[1] shapes on the bf16 engine")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--boot", type=int, default=200)
    args = ap.parse_args()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    bench.select_config(args.config)
    bench.DTYPE = "bf16"
    from oracle import e2e
    from faster_rcnn_amd.data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
    from faster_rcnn_amd.pipeline import InferencePipeline
    pipe, weights, anchors = bench.build_pipeline()
    mapping = VOC_CLASS_MAPPING if bench.NUM_CLASSES == len(VOC_CLASS_MAPPING) else KITTI_CLASS_MAPPING
    runs = []
    t0 = time.perf_counter()
    cpu = bench.cpu_baseline(weights, anchors, budget_s=0.0, runs=runs, min_images=args.frames, alt_dense_class=getattr(pipe, "raw_dense_class", None))
    oracle_s = time.perf_counter() - t0
    one = InferencePipeline(pipe.rpn, pipe.det, anchors, max_proposals=bench.PROPOSALS)
    raw, _ = bench.raw_head_pipeline(pipe, weights, anchors)
    out = {"what": "fp32 oracle end to end vs bf16 device end to end, %d synthetic %dx%d frames, ResNet-%d; bootstrap over frames (%d resamples)"
                   % (len(runs), bench.HEIGHT, bench.WIDTH, bench.DEPTH, args.boot),
           "oracle": cpu, "oracle_wall_s": round(oracle_s, 1)}
    rs = np.random.RandomState(0)
    for tag, p, col in (("head_calibrated", one, 2), ("head_as_drawn", raw, 3)):
        if p is None or any(len(r) <= col for r in runs):
            continue
        items = [{"name": "synth%03d" % r[0], "size": (bench.WIDTH, bench.HEIGHT), "oracle": (r[1], r[col]), "device": e2e.device_detect(p, bench.synth_image(r[0]))}
                 for r in runs]
        full = e2e.compare(items, mapping)
        m, t = (int(v) for v in full["detections_matched_iou50"].split("/"))
        deltas, per_frame = [], []
        for it in items:                                        # frame by frame: how far one frame's pair is apart on the same metric
            per_frame.append(e2e.compare([it], mapping)["map_pair_delta"])
        for _ in range(args.boot):
            pick = rs.randint(0, len(items), len(items))
            sub = [dict(items[i], name="%s_%02d" % (items[i]["name"], k)) for k, i in enumerate(pick)]       # (duplicates need distinct file names)
            deltas.append(e2e.compare(sub, mapping)["map_pair_delta"])
        d = np.asarray(deltas)
        out[tag] = {"map_pair_delta_all_frames": full["map_pair_delta"], "map_pseudo_gt": full.get("map_pseudo_gt"),
                    "matched_frac": round(m / max(t, 1), 4), "matched_score_diff": full["matched_score_diff"], "proposals_identical": full["proposals_identical"],
                    "bootstrap": {"mean": round(float(d.mean()), 4), "std": round(float(d.std()), 4),
                                  "p2.5": round(float(np.percentile(d, 2.5)), 4), "p97.5": round(float(np.percentile(d, 97.5)), 4), "max": round(float(d.max()), 4)},
                    "single_frame_deltas": {"mean": round(float(np.mean(per_frame)), 4), "max": round(float(np.max(per_frame)), 4)}}
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()


if __name__ == "__main__":
    main()
