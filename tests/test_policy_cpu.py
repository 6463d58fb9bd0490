"""Launch-policy host functions of the C library and of ops.py that need no GPU: which tile the split-bf16 engine takes for a
descriptor (frcnn_conv2d_x6_config), when its split-K form applies and on which tile edge (frcnn_conv2d_x6_workspace_bytes), and
ops._use_x6's choice between the engines for the layer shapes of configs[1]."""
import ctypes

import pytest

torch = pytest.importorskip("torch")


def _desc(shape, k, cout, stride=1, padding="same", tile=0):
    from faster_rcnn_amd import ops
    return ops._conv_desc(shape, k, k, cout, stride, padding, 0, 0, tile)


def test_x6_tile_choice_on_the_host():
    from faster_rcnn_amd import _lib
    lib = _lib.load()
    cfg = lambda d, n1=0: lib.frcnn_conv2d_x6_config(ctypes.byref(d), n1)
    assert cfg(_desc((1, 149, 249, 64), 3, 64)) == 74                      # stage 2's 3x3: 64 columns, 290 row tiles of 128
    assert cfg(_desc((1, 600, 1000, 64), 3, 64)) == 77                     # VGG16 conv1_2: >= 1024 row tiles of 64 columns
    assert cfg(_desc((300, 7, 7, 512), 3, 512)) == 76                      # the head's 3x3: long k, >= 200 tiles of 256x128
    assert cfg(_desc((300, 7, 7, 512), 1, 2048, padding="valid")) == 71
    assert cfg(_desc((1, 38, 63, 1024), 1, 2560, padding="valid"), 512) == 76
    assert cfg(_desc((1, 149, 249, 64), 1, 320, padding="valid"), 64) == 74   # a pair whose boundary is not a multiple of 128
    assert cfg(_desc((300, 7, 7, 512), 3, 512, tile=73)) == 73
    assert lib.frcnn_conv2d_x6_config(None, 0) < 0


def test_x6_split_k_workspace_names_the_tile_edge():
    """tickets (16 KB) + slices x tiles x one f32 tile: 64x64 tiles on the smallest grids, the eight-wave 128x128 tile from 64 such
    tiles on, ONE round of two workgroups per CU (slices = 512 // tiles)."""
    from faster_rcnn_amd import _lib
    lib = _lib.load()
    need = lambda d: lib.frcnn_conv2d_x6_workspace_bytes(ctypes.byref(d))
    # stage 4's 3x3 (2 394 rows x 256): 38 tiles of 128x128 -> 64x64 tiles (152), three slices
    assert need(_desc((1, 38, 63, 256), 3, 256)) == 16384 + 152 * 3 * 64 * 64 * 4
    # rpn_conv1 (2 394 x 512, k 9 216): 76 tiles of 128x128, six slices
    assert need(_desc((1, 38, 63, 1024), 3, 512)) == 16384 + 76 * 6 * 128 * 128 * 4
    # the head's 3x3 over 64 RoIs (3 136 x 512): 100 tiles, five slices
    assert need(_desc((64, 7, 7, 512), 3, 512)) == 16384 + 100 * 5 * 128 * 128 * 4
    assert need(_desc((300, 7, 7, 512), 3, 512)) == 0                      # 1 840 tiles of 64x64: whole-tile launches
    assert need(_desc((1, 38, 63, 1024), 1, 256, padding="valid")) == 0    # k = 1 024: under 64 chunks
    assert need(_desc((1, 38, 63, 1024), 3, 512, tile=374)) == 16384 + 304 * 3 * 64 * 64 * 4      # forced: 64x64 tiles, three slices
    assert need(_desc((1, 8, 8, 48), 1, 64, padding="valid")) == 0         # cin % 32 != 0 has no split form


def test_engine_policy_for_the_layers_of_configs1():
    """ops._use_x6 under ops.f32_engine("bf16x6"): >= 256 tiles of 64x64 and >= 64 columns, or the engine's split-K form."""
    from faster_rcnn_amd import ops
    pc = lambda k, cin, cout: type("PC", (), {"kh": k, "kw": k, "cin": cin, "cout": cout})()
    cases = [  # (input shape, k, cin, cout, stride, padding, expected with a split-K workspace at hand, expected without)
        ((1, 600, 1000, 3), 7, 3, 64, 2, "same", False, False),            # stem: cin % 32 != 0
        ((1, 149, 249, 64), 3, 64, 64, 1, "same", True, True),             # stage 2 3x3
        ((1, 149, 249, 256), 1, 256, 64, 1, "valid", True, True),
        ((1, 75, 125, 128), 3, 128, 128, 1, "same", True, True),           # stage 3 3x3: 294 tiles
        ((1, 38, 63, 256), 1, 256, 1024, 1, "valid", True, True),          # stage 4 2c: 608 tiles
        ((1, 38, 63, 1024), 1, 1024, 256, 1, "valid", False, False),       # stage 4 2a: 152 tiles, k under 64 chunks
        ((1, 38, 63, 256), 3, 256, 256, 1, "same", True, False),           # stage 4 3x3: only as split-K
        ((1, 38, 63, 1024), 3, 1024, 512, 1, "same", True, True),          # rpn_conv1: 304 tiles
        ((1, 38, 63, 512), 1, 512, 36, 1, "valid", False, False),          # rpn_out_bbreg: under 64 columns
        ((300, 7, 7, 512), 3, 512, 512, 1, "same", True, True),
        ((300, 1, 1, 2048), 1, 2048, 101, 1, "valid", False, False),       # the dense pair: 10 tiles, and no split-K under 128 columns
    ]
    with ops.f32_engine("bf16x6"):
        for shape, k, cin, cout, stride, padding, with_ws, without_ws in cases:
            d = ops._conv_desc(shape, k, k, cout, stride, padding, 0, 0, 0)
            assert ops._use_x6(d, pc(k, cin, cout), 0) == with_ws, (shape, k, cout)
            with ops.conv_workspace(ops.NO_SPLIT_K):
                assert ops._use_x6(d, pc(k, cin, cout), 0) == without_ws, (shape, k, cout)
    d = ops._conv_desc((300, 7, 7, 512), 3, 3, 512, 1, "same", 0, 0, 0)
    assert not ops._use_x6(d, pc(3, 512, 512), 0)                          # the library default stays native
    assert ops._use_x6(d, pc(3, 512, 512), 76) and not ops._use_x6(d, pc(3, 48, 512), 76)      # an explicit tile code; never with cin % 32 != 0


def test_engine_policy_is_the_librarys_and_serves_both_split_engines():
    """frcnn_conv2d_engine: the policy lives behind the C ABI -- the Python scope only names the preferred engine.
    Same answers for the f16x3 engine as for bf16x6 on the layers of configs[1]; explicit tile codes pick their engine."""
    import ctypes
    from faster_rcnn_amd import _lib, ops
    lib = _lib.load()
    q = lambda shape, k, cout, prefer, ws, stride=1, padding="same", tile=0: lib.frcnn_conv2d_engine(
        ctypes.byref(ops._conv_desc(shape, k, k, cout, stride, padding, 0, 0, tile)), prefer, ws)
    for prefer in (1, 2):
        assert q((1, 149, 249, 64), 3, 64, prefer, 1) == prefer                    # stage 2 3x3
        # stage 4 2a (152 tiles, k = 32 chunks): the f16x3 engine's own split-K starts there on small grids, the bf16x6 one's at 64 chunks
        assert q((1, 38, 63, 1024), 1, 256, prefer, 1, padding="valid") == (2 if prefer == 2 else 0)
        assert q((1, 38, 63, 1024), 1, 256, prefer, 0, padding="valid") == 0
        assert q((1, 75, 125, 128), 3, 128, prefer, 1) == prefer                   # (stage 3's 3x3: 294 tiles, plain launches on either engine)
        assert q((1, 38, 63, 256), 3, 256, prefer, 1) == prefer and q((1, 38, 63, 256), 3, 256, prefer, 0) == 0     # only as split-K
        assert q((1, 38, 63, 512), 1, 36, prefer, 1, padding="valid") == 0         # under 64 columns
        assert q((1, 600, 1000, 3), 7, 64, prefer, 1, stride=2) == 0               # the stem
    assert q((300, 7, 7, 512), 3, 512, 0, 1) == 0                                  # prefer native: native
    assert q((300, 7, 7, 512), 3, 512, 0, 1, tile=76) == 1 and q((300, 7, 7, 512), 3, 512, 0, 1, tile=86) == 2
    assert q((300, 7, 7, 512), 3, 512, 2, 1, tile=21) == 0                         # an explicit native tile code stays native
    assert lib.frcnn_conv2d_engine(None, 1, 1) < 0 and q((300, 7, 7, 512), 3, 512, 7, 1) < 0
    pc = lambda k, cin, cout: type("PC", (), {"kh": k, "kw": k, "cin": cin, "cout": cout})()
    with ops.f32_engine("f16x3"):
        d = ops._conv_desc((300, 7, 7, 512), 3, 3, 512, 1, "same", 0, 0, 0)
        assert ops._split_engine(d, pc(3, 512, 512), 0) == "h3" and not ops._use_x6(d, pc(3, 512, 512), 0)
    assert (ops.X6_MIN_TILES, ops.X6_MIN_COUT) == (256, 64)                        # the constants bench.py prints are the library's
