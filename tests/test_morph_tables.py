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


@pytest.mark.parametrize("k", [1, 2, 3, 5, 10, 15])
def test_oracle_morphology_against_scipy(k):
    """The oracle's erode / dilate (its stand-in for cv2.erode / cv2.dilate: anchor (w//2, h//2), kernel not reflected,
    pixels outside the image ignored) against scipy.ndimage's binary morphology, an implementation this repo did not write.
    By definition  cv2.erode(A, B)(z)  = AND_b A(z + b - anchor) = scipy binary_erosion(A, B) with the pixels outside set,
                   cv2.dilate(A, B)(z) = OR_b  A(z + b - anchor) = scipy binary_dilation (the Minkowski sum, which reflects B)
                                         with B reflected; a reflected even-sized B has its anchor one to the left (origin -1)."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(k)
    ker = D.ellipse_kernel(k, k)
    org = -1 if k % 2 == 0 else 0
    for density in (0.2, 0.5, 0.8, 0.97):
        a = rng.random((48, 61)) < density
        a[10:30, 5:40] |= density > 0.4                 # a solid block: something survives the big kernels
        img = a.astype(np.uint8) * 255
        er = ndi.binary_erosion(a, structure=ker.astype(bool), border_value=1)
        di = ndi.binary_dilation(a, structure=ker[::-1, ::-1].astype(bool), origin=(org, org), border_value=0)
        assert np.array_equal(D.erode(img, ker) > 0, er), (k, density)
        assert np.array_equal(D.dilate(img, ker) > 0, di), (k, density)
        cl = ndi.binary_erosion(ndi.binary_dilation(a, structure=ker[::-1, ::-1].astype(bool), origin=(org, org), border_value=0),
                                structure=ker.astype(bool), border_value=1)
        assert np.array_equal(D.morph_close(img, ker) > 0, cl), (k, density)


def test_clean_mask_against_scipy_on_a_reprojected_mask():
    """The whole clean-up (CLOSE with ellipse(res//50), OPEN with ellipse(res//250)) of the raw mask of a real re-projection."""
    import scipy.ndimage as ndi
    from diffusionhandles_amd.synthetic import TRANSFORMS, make_scene
    res = 512
    depth, bg, mask = make_scene(res)
    raw = np.zeros((res, res), bool)
    ys, xs = np.nonzero(mask[0, 0].numpy() > 0.5)
    raw[np.clip(ys + 37, 0, res - 1), np.clip((xs * 1.07).astype(int) - 20, 0, res - 1)] = True      # a stretched copy: holes
    def sd(a, k):
        o = -1 if k.shape[0] % 2 == 0 else 0
        return ndi.binary_dilation(a, structure=k[::-1, ::-1].astype(bool), origin=(o, o), border_value=0)
    def se(a, k):
        return ndi.binary_erosion(a, structure=k.astype(bool), border_value=1)
    kc, ko = D.ellipse_kernel(res // 50, res // 50), D.ellipse_kernel(res // 250, res // 250)
    want = sd(se(se(sd(raw, kc), kc), ko), ko)
    got = D.clean_mask(raw, res) > 0
    assert raw.sum() > 1000 and (got != raw).sum() > 50          # the clean-up does something on this mask
    assert np.array_equal(got, want)
