"""Seeded synthetic inputs shared by the golden generator's consumers (tests, bench, smoke).
Must stay in lock-step with tests/golden/make_golden.py."""
import numpy as np

GT5 = [[100, 100, 300, 400], [400, 50, 900, 550], [10, 10, 60, 80], [500, 300, 620, 420], [700, 100, 990, 590]]
GT5_CLS_VOC = [7, 14, 8, 7, 1]          # cat, person, chair, cat, bicycle in VOC_CLASS_MAPPING order

SHAPES = {"c2": (38, 63, 9), "c4": (38, 94, 18), "tiny": (5, 7, 9)}


def rpn_outputs(tag):
    """(regr (1,R,C,4A) f32, cls (1,R,C,A) f32) exactly as make_golden.py draws them."""
    rows, cols, A = SHAPES[tag]
    rs = np.random.RandomState(0)
    regr = (rs.randn(1, rows, cols, 4 * A) * 0.5).astype(np.float32)
    n = rows * cols * A
    cls = ((rs.permutation(n).astype(np.float32) + 0.5) / n).reshape(1, rows, cols, A)
    return regr, cls
