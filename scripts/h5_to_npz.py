#!/usr/bin/env python3
"""Offline converter: Keras 2.0.x weight/model .h5 -> the .npz layout faster_rcnn_amd loads
(`{layer_name}/{i}` arrays in get_weights() order).

h5py is not installed in this image's Python 3.10; run under an interpreter that has it, e.g.
    /opt/conda/bin/python3.9 scripts/h5_to_npz.py models/rpn_model_resnet50_step3.h5 step3.npz
Handles both `save_weights` files (root attrs layer_names) and full `save` files (group model_weights).
TimeDistributed wrappers store the inner layer's weights under the wrapper's name (resnet.py:282-313),
which is the name the lowered graphs use, so no renaming is needed.
"""
import sys

import h5py
import numpy as np


def convert(src, dst):
    out = {}
    with h5py.File(src, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]
        for lname in names:
            lg = g[lname]
            wnames = [n.decode() if isinstance(n, bytes) else n for n in lg.attrs.get("weight_names", [])]
            for i, wn in enumerate(wnames):
                out["%s/%d" % (lname, i)] = np.asarray(lg[wn])
    np.savez(dst, **out)
    return len(out)


if __name__ == "__main__":
    n = convert(sys.argv[1], sys.argv[2])
    print("wrote %d arrays to %s" % (n, sys.argv[2]))
