"""The offline Keras .h5 -> .npz converter (scripts/h5_to_npz.py) against a synthetic Keras-2.0.x style file.
Needs an interpreter with h5py (/opt/conda/bin/python3.9 in the build image); skipped elsewhere."""
import os
import subprocess

import numpy as np
import pytest

PY39 = "/opt/conda/bin/python3.9"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(PY39), reason="no h5py-capable interpreter")
def test_h5_to_npz_roundtrip(tmp_path):
    h5 = str(tmp_path / "w.h5")
    npz = str(tmp_path / "w.npz")
    make = (
        "import h5py, numpy as np\n"
        "rs = np.random.RandomState(0)\n"
        "f = h5py.File(%r, 'w')\n"
        "g = f.create_group('model_weights')\n"
        "layers = {'conv1': [('conv1/kernel:0', (7, 7, 3, 64)), ('conv1/bias:0', (64,))],\n"
        "          'bn_conv1': [('bn_conv1/gamma:0', (64,)), ('bn_conv1/beta:0', (64,)), ('bn_conv1/moving_mean:0', (64,)), ('bn_conv1/moving_variance:0', (64,))],\n"
        "          'activation_1': []}\n"
        "g.attrs['layer_names'] = [n.encode() for n in layers]\n"
        "for n, ws in layers.items():\n"
        "    lg = g.create_group(n)\n"
        "    lg.attrs['weight_names'] = [w.encode() for w, _ in ws]\n"
        "    for w, shp in ws:\n"
        "        lg.create_dataset(w, data=rs.randn(*shp).astype('float32'))\n"
        "f.close()\n" % h5)
    subprocess.run([PY39, "-c", make], check=True)
    subprocess.run([PY39, os.path.join(ROOT, "scripts", "h5_to_npz.py"), h5, npz], check=True)
    from faster_rcnn_amd.weights import load_npz
    w = load_npz(npz)
    assert set(w) == {"conv1", "bn_conv1"}
    assert w["conv1"][0].shape == (7, 7, 3, 64) and w["conv1"][1].shape == (64,)
    assert len(w["bn_conv1"]) == 4
    rs = np.random.RandomState(0)
    assert np.array_equal(w["conv1"][0], rs.randn(7, 7, 3, 64).astype("float32"))
