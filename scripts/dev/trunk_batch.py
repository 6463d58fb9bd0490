#!/usr/bin/env python3
"""The fp32 trunk (conv1 .. res4f) at B images per captured pass, several passes in flight (dev tool: would batched fp32 passes pay?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from faster_rcnn_amd import ops

with ops.f32_engine("f16x3"):
    pipe, weights, anchors = bench.build_pipeline()
    for graphs, batch in ((12, 1), (6, 2), (8, 2), (4, 3), (3, 4), (4, 4), (12, 1)):
        r = bench.backbone_in_flight(pipe, graphs, 75.25, steps=20, batch=batch, engine="f16x3")
        print("graphs %2d x batch %d: %.3f ms per image" % (graphs, batch, r["ms_per_image"]), flush=True)
