"""In-process reader of Keras 2.0.x HDF5 weight files (faster_rcnn_amd/h5lite.py) against fixtures written by
h5py (tests/golden/make_keras_h5.py, run under an interpreter that has h5py).  The fixture values are
RandomState(0) draws in file order, so the expected arrays are regenerated here without h5py."""
import os

import numpy as np
import pytest

from faster_rcnn_amd import h5lite
from faster_rcnn_amd.weights import load_weights_file

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAYERS = [
    ("input_1", []),
    ("conv1", [(7, 7, 3, 8), (8,)]),
    ("bn_conv1", [(8,), (8,), (8,), (8,)]),
    ("activation_1", []),
    ("res2a_branch2a", [(1, 1, 8, 4), (4,)]),
    ("scale2a_branch2a", [(4,), (4,)]),
    ("res5a_branch2b", [(3, 3, 4, 4), (4,)]),
    ("dense_class_21", [(16, 21), (21,)]),
] + [("pad_%02d" % i, [(2, 3)]) for i in range(40)]


@pytest.mark.parametrize("name", ["keras_weights_small.h5", "keras_model_small.h5", "keras_weights_small_vlen.h5"])
def test_read_keras_weights(name):
    w = load_weights_file(os.path.join(GOLD, name))
    rs = np.random.RandomState(0)
    assert set(w) == {n for n, shapes in LAYERS if shapes}            # weightless layers are dropped
    for lname, shapes in LAYERS:
        for i, shape in enumerate(shapes):
            want = rs.randn(*shape).astype("float32")
            got = w[lname][i]
            assert got.dtype == np.float32 and got.shape == shape and got.flags.writeable
            assert np.array_equal(got, want), (lname, i)


def test_npz_still_loads(tmp_path):
    from faster_rcnn_amd.weights import save_npz
    p = str(tmp_path / "w.npz")
    save_npz(p, {"conv1": [np.ones((1, 1, 2, 3), np.float32), np.zeros(3, np.float32)]})
    w = load_weights_file(p)
    assert list(w) == ["conv1"] and w["conv1"][0].shape == (1, 1, 2, 3)


def test_rejects_what_it_does_not_parse(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    assert not h5lite.is_hdf5(str(p))
    with pytest.raises(h5lite.H5Error):
        h5lite.read_keras_weights(str(p))
    blob = bytearray(open(os.path.join(GOLD, "keras_weights_small.h5"), "rb").read())
    blob[8] = 2                                                       # superblock version 2 (libver='latest')
    p.write_bytes(bytes(blob))
    with pytest.raises(h5lite.H5Error, match="superblock version 2"):
        h5lite.read_keras_weights(str(p))
