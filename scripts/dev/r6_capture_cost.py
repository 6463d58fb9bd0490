#!/usr/bin/env python3
"""bench.py's mixed-sizes legs alone (exact-geometry passes, canvas passes) with the engine's capture breakdown: where a captured
pass's ~50 ms of host time goes, how many captures a list of 36 geometries costs.  Dev tool.
  usage: r6_capture_cost.py [exact|canvas|both] [n_images]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from faster_rcnn_amd import entry

which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bench.select_config("c2")
pipe, w, anchors = bench.build_pipeline()
for canvas in ([False, True] if which == "both" else [which == "canvas"]):
    leg = bench.mixed_sizes_leg(pipe, anchors, n_images=n, canvas=canvas)
    leg.pop("what", None)
    print(json.dumps(leg))
