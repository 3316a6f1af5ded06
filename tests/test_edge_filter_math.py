"""The fp32 tile kernel's early (pixel, face) test -- conservative affine edge functions, kaolin_amd/csrc/tile_lists.h
`edge_coefficients` + raster2.inc -- restated in numpy and checked against the reference's own float arithmetic
(kaolin/csrc/render/mesh/rasterization_cuda.cu:139-150) on millions of (pixel, face) pairs chosen where the test is closest
to wrong.  Property: whenever the early test drops a pair, the reference's `w_i / norm < 0` drops it too.  (No GPU: this is
the arithmetic of the argument in the header comment; the GPU parity tests check the kernel end to end.)"""
import numpy as np
import pytest

F32 = np.float32
U = 1.0 / 16777216.0


def coefficients(v, box):
    """numpy restatement of tl::edge_coefficients: v (n, 6) float32 vertices a.xy b.xy c.xy, box (n, 4) float32 x0 y0 x1 y1
    -> (n, 7) float32 {A0 B0 C0 A1 B1 C1 K}."""
    v = v.astype(np.float64)
    ax, ay, bx, by, cx, cy = (v[:, i] for i in range(6))
    xlo, ylo, xhi, yhi = (box[:, i].astype(np.float64) for i in range(4))
    al = np.stack([by - cy, cy - ay, ay - by], 1)
    be = np.stack([cx - bx, ax - cx, bx - ax], 1)
    ga = np.stack([bx * cy - by * cx, cx * ay - cy * ax, ax * by - ay * bx], 1)
    area2 = ga.sum(1)
    m = lambda p, lo, hi: np.maximum(np.abs(p - lo), np.abs(p - hi))
    axm, aym, bxm, bym, cxm, cym = m(ax, xlo, xhi), m(ay, ylo, yhi), m(bx, xlo, xhi), m(by, ylo, yhi), m(cx, xlo, xhi), m(cy, ylo, yhi)
    S = np.stack([bxm * cym + bym * cxm, cxm * aym + cym * axm, axm * bym + aym * bxm], 1)
    X, Y = np.maximum(np.abs(xlo), np.abs(xhi)), np.maximum(np.abs(ylo), np.abs(yhi))
    sgn = np.where(area2 < 0, -1.0, 1.0)
    with np.errstate(invalid='ignore', over='ignore'):
        ok = (np.abs(area2) > 32.0 * U * S.sum(1)) & (np.abs(area2) < 1e6) & (X < 1e30) & (Y < 1e30)
        Q = np.abs(al) * X[:, None] + np.abs(be) * Y[:, None] + np.abs(ga)
        M = 32.0 * U * (S + Q) + 1e-30
        K = (np.abs(area2) + M.sum(1) + 8.0 * U * (Q[:, 0] + Q[:, 1] + M[:, 0] + M[:, 1])) * (1.0 + 4.0 * U)
        ok &= K < 1e30
    out = np.zeros((v.shape[0], 7), F32)
    for i in range(2):
        out[:, i * 3 + 0] = np.where(ok, sgn * al[:, i], 0.0).astype(F32)
        out[:, i * 3 + 1] = np.where(ok, sgn * be[:, i], 0.0).astype(F32)
        out[:, i * 3 + 2] = np.where(ok, sgn * ga[:, i] + M[:, i], 1.0).astype(F32)
    out[:, 6] = np.where(ok, K, 3.0).astype(F32)
    return out


def fma32(a, b, c):
    """float32 fma: the product of two floats is exact in double; the sum is rounded once to double and once to float (a
    double rounding can differ from the hardware's single rounding by one ulp of the result -- far inside the margins)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def early_drop(co, px, py):
    e0 = fma32(co[:, 0], px, fma32(co[:, 1], py, co[:, 2]))
    e1 = fma32(co[:, 3], px, fma32(co[:, 4], py, co[:, 5]))
    return (e0 < 0) | (e1 < 0) | ((e0 + e1).astype(F32) > co[:, 6])


def reference_drop(v, px, py, eps):
    """The reference's test in its own float arithmetic (no contraction): some w_i / norm < 0."""
    ax, ay, bx, by, cx, cy = (v[:, i] for i in range(6))
    aex, aey, bex, bey, cex, cey = ax - px, ay - py, bx - px, by - py, cx - px, cy - py
    w0 = (bex * cey).astype(F32) - (bey * cex).astype(F32)
    w1 = (cex * aey).astype(F32) - (cey * aex).astype(F32)
    w2 = (aex * bey).astype(F32) - (aey * bex).astype(F32)
    norm = ((w0 + w1).astype(F32) + w2).astype(F32)
    norm = (norm.astype(np.float64) + np.copysign(np.float64(eps), norm.astype(np.float64))).astype(F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        q0, q1, q2 = (w0 / norm).astype(F32), (w1 / norm).astype(F32), (w2 / norm).astype(F32)
    return (q0 < 0) | (q1 < 0) | (q2 < 0)


def _faces(rng, n, scale):
    kind = rng.integers(0, 5, n)
    v = (rng.random((n, 6)) * 2 - 1) * scale
    # slivers: c almost on the line a-b
    t = rng.random(n)
    sl = kind == 1
    v[sl, 4] = v[sl, 0] + t[sl] * (v[sl, 2] - v[sl, 0]) + (rng.random(sl.sum()) - 0.5) * 1e-6 * scale
    v[sl, 5] = v[sl, 1] + t[sl] * (v[sl, 3] - v[sl, 1]) + (rng.random(sl.sum()) - 0.5) * 1e-6 * scale
    # pixel-sized faces far from the origin (cancellation in the constant term)
    sm = kind == 2
    c = (rng.random((sm.sum(), 2)) * 2 - 1) * scale
    v[sm] = np.tile(c, 3) + (rng.random((sm.sum(), 6)) - 0.5) * scale * 4e-3
    # exactly degenerate
    dg = kind == 3
    v[dg, 4:6] = v[dg, 2:4]
    return v.astype(F32)


@pytest.mark.parametrize('scale,eps', [(1000.0, 1e-8), (1.0, 1e-8), (1000.0, 1.0), (1e5, 1e-8), (1000.0, 0.0)])
def test_early_test_never_drops_what_the_reference_keeps(scale, eps):
    rng = np.random.default_rng(int(scale) % 97 + int(eps * 10))
    n = 400000
    v = _faces(rng, n, scale)
    lo = np.minimum(np.minimum(v[:, 0:2], v[:, 2:4]), v[:, 4:6])
    hi = np.maximum(np.maximum(v[:, 0:2], v[:, 2:4]), v[:, 4:6])
    box = np.concatenate([lo, hi], 1).astype(F32)
    co = coefficients(v, box)
    dropped = kept_by_reference = 0
    for mode in range(4):
        w = rng.random((n, 3))
        if mode == 0:      # anywhere in the box
            px = lo[:, 0] + rng.random(n).astype(F32) * (hi[:, 0] - lo[:, 0])
            py = lo[:, 1] + rng.random(n).astype(F32) * (hi[:, 1] - lo[:, 1])
        else:              # on (mode 1) or within a few ulps / 1e-6 of (modes 2, 3) an edge: where the decision is closest
            w[:, mode % 3] = 0.0 if mode == 1 else (rng.random(n) - 0.5) * (1e-7 if mode == 2 else 1e-5)
            w /= np.maximum(w.sum(1, keepdims=True), 1e-30)
            px = (w[:, 0] * v[:, 0] + w[:, 1] * v[:, 2] + w[:, 2] * v[:, 4])
            py = (w[:, 0] * v[:, 1] + w[:, 1] * v[:, 3] + w[:, 2] * v[:, 5])
        px, py = px.astype(F32), py.astype(F32)
        inside = (px >= box[:, 0]) & (px < box[:, 2]) & (py >= box[:, 1]) & (py < box[:, 3])   # the kernel only asks inside the box
        early = early_drop(co, px, py) & inside
        ref = reference_drop(v, px, py, eps)
        assert not np.any(early & ~ref), f'mode {mode}: the early test dropped {int(np.sum(early & ~ref))} pairs the reference keeps'
        dropped += int(early.sum())
        kept_by_reference += int((~ref & inside).sum())
    # the sample exercises both outcomes (at scale 1e5 most faces exceed the area bound of the early test: fewer drops)
    assert dropped > n // 10 and kept_by_reference > n // 4


def test_faces_the_bounds_do_not_cover_are_never_dropped():
    v = np.array([[0, 0, 1, 1, 2, 2],              # zero area
                  [0, 0, 1e-30, 0, 0, 1e-30],      # area far below the rounding bound
                  [0, 0, 3e38, 0, 0, 3e38],        # coefficients overflow float
                  [np.nan, 0, 1, 0, 0, 1]], F32)
    box = np.array([[0, 0, 2, 2], [0, 0, 1, 1], [0, 0, 3e38, 3e38], [0, 0, 1, 1]], F32)
    co = coefficients(v, box)
    assert np.array_equal(co, np.tile(np.array([0, 0, 1, 0, 0, 1, 3], F32), (4, 1)))
    px = np.array([0.5, 0.5, 1.0, 0.5], F32)
    assert not early_drop(co, px, px).any()
    # a box with a NaN limit rejects no pixel: its face is not dropped early either
    co = coefficients(np.array([[0, 0, 1, 0, 0, 1]], F32), np.array([[np.nan, 0, 1, 1]], F32))
    assert np.array_equal(co[0], np.array([0, 0, 1, 0, 0, 1, 3], F32))
