#!/usr/bin/env python3
"""Writes tests/golden/keras_weights_small.h5 and keras_model_small.h5: files in the layout Keras 2.0.8's
save_weights / model.save produce (keras/engine/topology.py save_weights_to_hdf5_group), with tiny tensors.
Needs h5py: run under /opt/conda/bin/python3.9.  The arrays are RandomState(0) draws in file order, so the
test regenerates the expected values without h5py."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LAYERS = [  # (layer name, [(weight name, shape)]) in model.layers order; weightless layers are listed like Keras does
    ("input_1", []),
    ("conv1", [("conv1/kernel:0", (7, 7, 3, 8)), ("conv1/bias:0", (8,))]),
    ("bn_conv1", [("bn_conv1/gamma:0", (8,)), ("bn_conv1/beta:0", (8,)), ("bn_conv1/moving_mean:0", (8,)), ("bn_conv1/moving_variance:0", (8,))]),
    ("activation_1", []),
    ("res2a_branch2a", [("res2a_branch2a/kernel:0", (1, 1, 8, 4)), ("res2a_branch2a/bias:0", (4,))]),
    ("scale2a_branch2a", [("scale2a_branch2a/scale2a_branch2a_gamma:0", (4,)), ("scale2a_branch2a/scale2a_branch2a_beta:0", (4,))]),
    ("res5a_branch2b", [("res5a_branch2b/kernel:0", (3, 3, 4, 4)), ("res5a_branch2b/bias:0", (4,))]),     # a TimeDistributed wrapper's name
    ("dense_class_21", [("dense_class_21/kernel:0", (16, 21)), ("dense_class_21/bias:0", (21,))]),
] + [("pad_%02d" % i, [("pad_%02d/kernel:0" % i, (2, 3))]) for i in range(40)]      # > 32 links: the group B-tree grows a level


def names(items, fixed):
    # h5py 2.7 (the reference's pin) turned a list of bytes into a fixed-length 'S' array; h5py 3 writes the same
    # list as variable-length strings.  Both flavours are generated.
    items = [n.encode("utf8") for n in items]
    return np.array(items, dtype="S") if fixed else items


def write(group, fixed=True):
    rs = np.random.RandomState(0)
    group.attrs["layer_names"] = names([n for n, _ in LAYERS], fixed)
    group.attrs["backend"] = b"tensorflow"
    group.attrs["keras_version"] = b"2.0.8"
    for name, ws in LAYERS:
        g = group.create_group(name)
        if fixed and not ws:
            g.attrs.create("weight_names", data=np.zeros((0,), dtype="S1"))
        else:
            g.attrs["weight_names"] = names([w for w, _ in ws], fixed)
        for w, shape in ws:
            g.create_dataset(w, data=rs.randn(*shape).astype("float32"))


with h5py.File(os.path.join(HERE, "keras_weights_small.h5"), "w") as f:
    write(f)
with h5py.File(os.path.join(HERE, "keras_model_small.h5"), "w") as f:
    f.attrs["keras_version"] = b"2.0.8"
    f.attrs["backend"] = b"tensorflow"
    f.attrs["model_config"] = b'{"class_name": "Model", "config": {}}'
    write(f.create_group("model_weights"))
    f.create_group("optimizer_weights")
with h5py.File(os.path.join(HERE, "keras_weights_small_vlen.h5"), "w") as f:
    write(f, fixed=False)
