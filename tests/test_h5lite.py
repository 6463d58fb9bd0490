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


def _sample_weights():
    rs = np.random.RandomState(3)
    w = {"conv1": [rs.randn(7, 7, 3, 8).astype("f4"), rs.randn(8).astype("f4")],
         "bn_conv1": [rs.randn(8).astype("f4") for _ in range(4)],
         "scale_conv1": [rs.randn(8).astype("f4") for _ in range(2)],
         "res5a_branch2b": [rs.randn(3, 3, 4, 4).astype("f4")],
         "dense_class_21": [rs.randn(16, 21).astype("f4"), rs.randn(21).astype("f4")]}
    for i in range(150):                                              # > 64 links: several symbol nodes under one B-tree node
        w["pad_%03d" % i] = [rs.randn(2, 3).astype("f4")]
    return w


@pytest.mark.parametrize("full_model", [False, True])
def test_writer_round_trip(tmp_path, full_model):
    from faster_rcnn_amd.weights import save_weights_file
    w = _sample_weights()
    p = str(tmp_path / "w.h5")
    save_weights_file(p, w, full_model=full_model)
    assert h5lite.is_hdf5(p)
    back = load_weights_file(p)
    assert list(back) == list(w)                                      # layer_names keeps the given order
    for k in w:
        assert len(back[k]) == len(w[k])
        for a, b in zip(w[k], back[k]):
            assert np.array_equal(a, b)


PY39 = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(PY39), reason="no h5py-capable interpreter")
def test_written_file_is_read_by_h5py(tmp_path):
    """The real HDF5 library (h5py under the image's second interpreter) reads what the writer lays out."""
    import subprocess
    from faster_rcnn_amd.weights import save_npz, save_weights_file
    w = _sample_weights()
    h5, npz = str(tmp_path / "w.h5"), str(tmp_path / "w.npz")
    save_weights_file(h5, w)
    save_npz(npz, w)
    check = (
        "import h5py, numpy as np\n"
        "ref = np.load(%r)\n"
        "f = h5py.File(%r, 'r')\n"
        "names = [n.decode() for n in f.attrs['layer_names']]\n"
        "assert f.attrs['keras_version'] == b'2.0.8'\n"
        "n = 0\n"
        "for ln in names:\n"
        "    for i, wn in enumerate(f[ln].attrs['weight_names']):\n"
        "        assert np.array_equal(np.asarray(f[ln][wn.decode()]), ref['%%s/%%d' %% (ln, i)]); n += 1\n"
        "assert [x.decode() for x in f['bn_conv1'].attrs['weight_names']][2] == 'bn_conv1/moving_mean:0'\n"
        "print(n)\n" % (npz, h5))
    out = subprocess.run([PY39, "-W", "ignore", "-c", check], check=True, capture_output=True, text=True).stdout
    assert int(out.strip()) == sum(len(v) for v in w.values())
