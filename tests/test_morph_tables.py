"""CPU: the elliptical structuring elements of the mask clean-up (reference depth_transform.py:311-321:
cv2.getStructuringElement(MORPH_ELLIPSE, (res//50,)*2) and (res//250,)*2) pinned to published / hand-derived tables.

cv2 is absent, so the oracle (oracle/depth_ref.ellipse_kernel) and the product (csrc/geometry.hip ellipse_offsets,
read through the host-only hook dh_dbg_ellipse_offsets) restate OpenCV's row-span rule
    r = h//2, c = w//2;  row i: dy = i - r;  dx = cvRound(c * sqrt((r*r - dy*dy) / (r*r)));  ones in [c-dx, min(c+dx+1, w))
The 5x5 table is the one printed in OpenCV's own documentation (tutorial "Morphological Transformations",
`cv.getStructuringElement(cv.MORPH_ELLIPSE,(5,5))`); the others are derived by hand from the rule
(10x10: dx = round(sqrt(25 - dy^2)) = 0,3,4,5,5,5,5,5,4,3 for dy = -5..4; 2x2: r = c = 1, dx = 0,1;
15x15 / 3x3 are the 768^2 kernels)."""
import ctypes

import numpy as np
import pytest

from diffusionhandles_amd import _lib
from oracle import depth_ref as D


def rows(spans, w):
    k = np.zeros((len(spans), w), np.uint8)
    for i, (a, b) in enumerate(spans):
        k[i, a:b] = 1
    return k


TABLES = {
    5: np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8),   # OpenCV docs
    10: rows([(5, 6), (2, 9), (1, 10), (0, 10), (0, 10), (0, 10), (0, 10), (0, 10), (1, 10), (2, 9)], 10),
    2: np.array([[0, 1], [1, 1]], np.uint8),
    3: np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8),
    # r = c = 7: dx = round(sqrt(49 - dy^2)) = 0, 4(3.61), 5(4.90), 6(5.74), 6(6.32), 7(6.71), 7(6.93), 7, ...
    15: rows([(7, 8), (3, 12), (2, 13), (1, 14), (1, 14), (0, 15), (0, 15), (0, 15), (0, 15), (0, 15), (1, 14), (1, 14),
              (2, 13), (3, 12), (7, 8)], 15),
    1: np.ones((1, 1), np.uint8),
}


@pytest.mark.parametrize("k", sorted(TABLES))
def test_oracle_ellipse_matches_published_table(k):
    assert np.array_equal(D.ellipse_kernel(k, k), TABLES[k])


@pytest.mark.parametrize("k", sorted(TABLES))
def test_product_ellipse_offsets_match_published_table(k):
    L = _lib.lib()
    buf = (ctypes.c_int32 * (2 * k * k))()
    n = ctypes.c_int()
    _lib.check(L.dh_dbg_ellipse_offsets(k, buf, k * k, ctypes.byref(n)), "dh_dbg_ellipse_offsets")
    got = np.zeros((k, k), np.uint8)
    a = k // 2                                   # anchor = (w//2, h//2)
    for i in range(n.value):
        got[buf[2 * i + 1] + a, buf[2 * i] + a] = 1
    assert n.value == int(TABLES[k].sum())
    assert np.array_equal(got, TABLES[k])


def test_morphology_on_hand_worked_case():
    """CLOSE/OPEN semantics on a case small enough to do by hand: anchor (w//2, h//2), un-reflected kernel, pixels
    outside the image ignored.  With the 2x2 kernel [[0,1],[1,1]] (offsets (0,-1), (-1,0), (0,0)), erode keeps a pixel only
    if it, its upper and its left neighbour are set; dilate then sets a pixel if it or its LOWER or RIGHT neighbour ... is
    in the eroded set shifted back -- i.e. OPEN removes isolated pixels and 1-wide lines, keeps an L of three."""
    k2 = D.ellipse_kernel(2, 2)
    img = np.zeros((6, 6), np.uint8)
    img[1, 1] = 255                       # isolated pixel: removed
    img[3, 2:5] = 255                     # 1-wide horizontal line: removed
    img[4, 3] = 255                       # makes an L (3,3),(3,2)... with the line: (4,3) has upper (3,3) and left (4,2)=0 -> no
    out = D.morph_open(img, k2)
    assert out.sum() == 0
    img[4, 2] = 255                       # now (4,3): self, upper (3,3), left (4,2) all set -> survives erosion
    er = D.erode(img, k2)
    want = np.zeros_like(img)
    want[4, 3] = 255
    assert np.array_equal(er, want)
    op = D.dilate(er, k2)                 # dilate: out(y,x) = max src(y+dy, x+dx) over the same offsets (0,-1),(-1,0),(0,0)
    want2 = np.zeros_like(img)
    want2[4, 3] = want2[5, 3] = want2[4, 4] = 255
    assert np.array_equal(op, want2)
