"""What running the reference's kernels on oracle/_ref leaves unpinned is the 29-function stand-in for the OpenCL
built-in library (oracle/ref_build/ocl_builtins.cpp).  These tests check that stand-in against values DERIVED from the
OpenCL 1.2 specification (s8.2 linear filtering, s8.3 repeat addressing, s6.12.2/4/5 math, common and geometric
functions) by construction -- an image that is an affine function of the texel index, texel centres, seams, ties of
round-to-nearest-even -- not by restating its formulas, and not against oracle.cpp (which is tuned to be bit-exact with
_ref and would inherit a shared misreading).  The product's texture fetch (csrc/pt_shade.h sampleLinearRepeat) is then
tied to the same facts on the GPU by test_gpu_render.py::test_texture_fetch_facts."""
import ctypes as C
import os

import numpy as np
import pytest

import orclib as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs the reference tree once; the built library travels)")


class RefImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("layers", C.c_int32), ("_pad", C.c_int32), ("rgba", C.c_void_p)]


@pytest.fixture(scope="module")
def ref():
    lib = O.ref_kernels()
    if not hasattr(lib, "ref_test_read_imagef"):
        pytest.skip("prebuilt oracle/_ref predates the test doors")
    lib.ref_test_scalar.restype = C.c_float
    lib.ref_test_scalar.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    lib.ref_test_atomic_inc.restype = C.c_uint
    return lib


def _image(texels):
    t = np.ascontiguousarray(texels, np.float32)
    layers, h, w, _ = t.shape
    return RefImage(w, h, layers, 0, t.ctypes.data), t


def _read(lib, img, s, t, layer=0.0):
    c = (C.c_float * 4)(s, t, layer, 0.0)
    out = (C.c_float * 4)()
    lib.ref_test_read_imagef(C.byref(img), c, out)
    return np.array(out[:], np.float32)


def test_read_imagef_linear_repeat_normalized(ref):
    W, H = 8, 4
    i, j = np.meshgrid(np.arange(W), np.arange(H))
    tex = np.zeros((3, H, W, 4), np.float32)
    for layer in range(3):  # an affine function of the texel index: bilinear filtering reproduces it between centres
        tex[layer, ..., 0] = 2.0 * i + 16.0 * j + 100.0 * layer
        tex[layer, ..., 1] = -1.0 * i + 0.5 * j
        tex[layer, ..., 2] = 7.0
        tex[layer, ..., 3] = (i + j) % 2
    img, keep = _image(tex)
    # (1) texel centres: s = (i + 1/2) / w addresses texel i exactly (s8.2: u = s*w, i0 = floor(u - 1/2), weight a = 0)
    for jj in range(H):
        for ii in range(W):
            got = _read(ref, img, (ii + 0.5) / W, (jj + 0.5) / H)
            assert np.array_equal(got, tex[0, jj, ii]), (ii, jj, got)
    # (2) between centres (no seam involved): the affine image is reproduced at (u - 1/2, v - 1/2)
    rng = np.random.default_rng(1)
    for _ in range(200):
        u, v = rng.uniform(0.5, W - 0.5), rng.uniform(0.5, H - 0.5)
        got = _read(ref, img, u / W, v / H)
        x, y = u - 0.5, v - 0.5
        assert np.allclose(got[:3], [2 * x + 16 * y, -x + 0.5 * y, 7.0], rtol=0, atol=2e-4), (u, v, got)
    # (3) the seam, CLK_ADDRESS_REPEAT (s8.3): s = 0 lies midway between the LAST and the FIRST texel of the row
    got = _read(ref, img, 0.0, 1.5 / H)
    assert np.allclose(got, 0.5 * (tex[0, 1, W - 1] + tex[0, 1, 0]), atol=1e-5)
    got = _read(ref, img, 2.5 / W, 0.0)
    assert np.allclose(got, 0.5 * (tex[0, H - 1, 2] + tex[0, 0, 2]), atol=1e-5)
    # just inside either edge the weight of the wrapped neighbour is 1/2 -+ eps*w
    eps = 1e-3
    got = _read(ref, img, 1.0 - eps, 1.5 / H)
    wgt = 0.5 + eps * W  # weight of texel W-1 ... (s*w - 1/2) = W - 1/2 - eps*W -> a = 1/2 - eps*W towards texel 0
    assert np.allclose(got[0], wgt * tex[0, 1, W - 1, 0] + (1 - wgt) * tex[0, 1, 0, 0], atol=2e-3)
    # (4) coordinates outside [0,1) repeat with period 1, negative ones included
    for s, t in ((0.3, 0.7), (0.05, 0.95)):
        base = _read(ref, img, s, t)
        for ds, dt in ((1, 0), (0, 1), (-1, 0), (-2, 3), (5, -4)):
            assert np.allclose(_read(ref, img, s + ds, t + dt), base, atol=2e-4), (s, t, ds, dt)
    # (5) array layer = clamp(rint(w), 0, layers - 1), rint = round half to EVEN; no filtering across layers
    centre = (3.5 / W, 2.5 / H)
    for w, layer in ((0.0, 0), (0.49, 0), (0.5, 0), (0.51, 1), (1.49, 1), (1.5, 2), (2.0, 2), (2.5, 2), (7.0, 2), (-3.0, 0)):
        assert np.array_equal(_read(ref, img, *centre, layer=w), tex[layer, 2, 3]), (w, layer)
    # (6) alpha is filtered like any channel (the cut-out test of shading.cl:590-595 compares the FILTERED alpha with 0)
    got = _read(ref, img, 1.0 / W, 0.5 / H)  # midway between texels 0 and 1 of row 0: alphas 0 and 1
    assert abs(got[3] - 0.5) < 1e-6
    del keep


def test_scalar_common_and_math_functions(ref):
    f = lambda op, x, y=0.0, z=0.0: float(ref.ref_test_scalar(op, x, y, z))
    # s6.12.4: min(x, y) = y < x ? y : x, max(x, y) = x < y ? y : x, clamp = min(max(x, lo), hi), mix = x + (y - x) a
    assert f(0, 2.0, -3.0) == -3.0 and f(0, -3.0, 2.0) == -3.0 and f(1, 2.0, -3.0) == 2.0 and f(1, -0.0, 0.0) == 0.0
    assert f(3, 5.0, 0.0, 1.0) == 1.0 and f(3, -5.0, 0.0, 1.0) == 0.0 and f(3, 0.25, 0.0, 1.0) == 0.25
    assert f(3, 7.0, 0.0, 6.2831855) == np.float32(6.2831855)  # the solid-angle clamp of shading.cl:431
    assert f(2, 1.0, 3.0, 0.25) == 1.5 and f(2, -2.0, 2.0, 0.0) == -2.0 and f(2, -2.0, 2.0, 1.0) == 2.0
    assert f(4, 1.0, 2.0) == 1.0 and f(5, 1.0, 2.0) == 2.0
    # s7.4: single-precision math within the ULP bounds of the full profile (cos/sin/exp/... <= 4 ulp, pow <= 16, sqrt <= 3)
    rng = np.random.default_rng(2)

    def ulps(got, want64):
        want = np.float32(want64)
        return abs(float(got) - float(want64)) / float(np.spacing(np.abs(want)))
    for _ in range(300):
        x = float(np.float32(rng.uniform(-6.0, 6.0)))
        y = float(np.float32(rng.uniform(0.1, 4.0)))
        u = float(np.float32(rng.uniform(-1.0, 1.0)))
        assert ulps(f(6, x), np.cos(x)) <= 4 and ulps(f(7, x), np.sin(x)) <= 4 and ulps(f(12, x), np.exp(x)) <= 4
        assert ulps(f(9, u), np.arccos(u)) <= 4 and ulps(f(10, x), np.arctan(x)) <= 5 and ulps(f(11, x, y), np.arctan2(x, y)) <= 6
        assert ulps(f(13, y, x), np.power(y, x)) <= 16 and ulps(f(14, y), np.sqrt(y)) <= 3 and f(15, x) == abs(x)
        assert ulps(f(16, u * 0.99), np.log1p(u * 0.99)) <= 4 and ulps(f(17, y), np.log2(y)) <= 4
        if abs(np.cos(x)) > 1e-2:
            assert ulps(f(8, x), np.tan(x)) <= 5
    # known points
    assert f(9, 1.0) == 0.0 and abs(f(9, -1.0) - np.pi) < 1e-6 and f(12, 0.0) == 1.0 and f(13, 2.0, 10.0) == 1024.0 and f(16, 0.0) == 0.0


def test_geometric_functions(ref):
    rng = np.random.default_rng(3)

    def v3(op, a, b=(0, 0, 0)):
        a3, b3, out = (C.c_float * 3)(*a), (C.c_float * 3)(*b), (C.c_float * 3)()
        ref.ref_test_vec3(op, a3, b3, out)
        return np.array(out[:], np.float64)
    for _ in range(200):
        a = rng.normal(size=3).astype(np.float32)
        b = rng.normal(size=3).astype(np.float32)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        scale = np.abs(a64).max() * np.abs(b64).max()
        assert abs(v3(0, a, b)[0] - a64 @ b64) <= 4e-7 * scale * 3  # s6.12.5 dot: sum of products
        assert np.allclose(v3(1, a, b), np.cross(a64, b64), rtol=0, atol=4e-7 * scale * 2)  # cross, right-handed
        n = v3(2, a)
        assert np.allclose(n, a64 / np.linalg.norm(a64), rtol=0, atol=3e-7) and abs(np.linalg.norm(n) - 1) < 3e-7  # normalize: unit length, same direction
        assert np.allclose(v3(3, a), np.exp(a64), rtol=3e-7) and np.array_equal(v3(5, a), np.abs(a64))
        assert np.allclose(v3(4, np.abs(a) + 0.1, b), np.power(np.abs(a64) + np.float32(0.1), b64), rtol=2e-6)
    assert np.array_equal(v3(1, (1, 0, 0), (0, 1, 0)), [0, 0, 1])


def test_atomic_inc_returns_the_old_value(ref):
    word = C.c_uint(41)
    assert ref.ref_test_atomic_inc(C.byref(word)) == 41 and word.value == 42  # s6.12.11: old value returned, *p = old + 1
