#!/usr/bin/env python3
"""configs[3]'s batched pass (ResNet-101, 600x1500, bf16, B = 8 images, BatchedInferencePipeline) launched EAGERLY n times: the
program the rocprofv3 --pmc passes of scripts/profile_round3.sh profile (PMC needs one dispatch at a time; the timed bench
replays the same launches from hipGraphs).   usage: c4_batch_eager.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from faster_rcnn_amd import ops
from faster_rcnn_amd.pipeline import BatchedInferencePipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bench.select_config("c4")
pipe, weights, anchors = bench.build_pipeline()
B = 8
pipe = BatchedInferencePipeline(pipe.rpn, pipe.det, anchors, B, max_proposals=bench.PROPOSALS)
x = torch.from_numpy(np.concatenate([bench.synth_image(j) for j in range(B)])).cuda()
with ops.conv_workspace(ops.NO_SPLIT_K), ops.tile_policy(True):
    for _ in range(n):
        out = pipe.forward_dev(x)
torch.cuda.synchronize()
print("ran %d batched passes of %d images; detections of image 0: %d" % (n, B, int(out["n_dets"][0])))
