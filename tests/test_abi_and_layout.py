"""The C-ABI libraries load and export every symbol the headers declare; record layouts match the
reference's device structs."""
import ctypes
import os
import re

import numpy as np

from ptamd import device as D, host as H, layout as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, text)))


def test_device_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(D.DEVICE_LIB_PATH)
    names = _declared("ptamd.h", "pt_")
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(D.EXPORTS) == names


def test_host_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(H.HOST_LIB_PATH)
    names = _declared("ptamd_host.h", "pth_")
    assert len(names) >= 12
    assert not [n for n in names if not hasattr(lib, n)]


def test_no_gpu_means_loud_failure_not_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(D.PtError, match="no HIP device"):
        D.Context(16, 16)


def test_record_sizes_match_reference_device_structs():
    sizes = {L.VERTEX: 48, L.TRIANGLE: 16, L.MATERIAL: 48, L.EMISSIVE_TRIANGLE: 96, L.SUB_BVH_NODE: 48, L.TOP_BVH_NODE: 112,
             L.CAMERA: 128, L.RAY_DATA: 80, L.SHADING_DATA: 32, L.KERNEL_DATA: 176, L.LFSR113_STREAM: 48}
    for dt, sz in sizes.items():
        assert dt.itemsize == sz
    assert L.KERNEL_DATA.fields["numEmissiveTriangles"][1] == 128
    assert L.KERNEL_DATA.fields["newRays"][1] == 164
    assert L.RAY_DATA.fields["outputPixel"][1] == 48 and L.RAY_DATA.fields["flags"][1] == 56
    assert L.TOP_BVH_NODE.fields["invTransform"][1] == 32 and L.TOP_BVH_NODE.fields["isLeaf"][1] == 104
    assert L.MATERIAL.fields["type"][1] == 32 and L.MATERIAL.fields["metallic"][1] == 24


def test_material_factories_fill_the_union_like_the_reference():
    m = L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0)  # src/model/material.h:112-120
    assert m["type"] == L.MAT_REFRACTIVE and abs(m["smoothness"] - 0.9) < 1e-7 and abs(m["refractiveIndexRough"] - 1.5) < 1e-7
    assert np.allclose(m["colour"][:3], [0.0, 2.0, 2.0], rtol=1e-6)
    e = L.material_emissive((1, 0.5, 0.25), 500.0)
    assert list(e["colour"][:3]) == [500.0, 250.0, 125.0]
    d = L.material_diffuse((0.1, 0.2, 0.3))
    assert d["textureId"] == -1


def test_quantised_child_boxes_contain_the_exact_ones():
    """The 4-wide nodes hold their children's boxes as 8-bit planes relative to the node's origin and per-axis power-of-two scale
    (pt_bake.h, quantiseWideNode: shared by the host's collapse and the device's world-space copies).  The planes must be CONSERVATIVE
    when evaluated the way the traversal kernels do (origin + 2^exp * q in fp32): lower planes at or below the box, upper planes at or
    above -- for boxes of any size and position, flat ones, far-away ones, one child or four.  Runs without a GPU."""
    lib = ctypes.CDLL(D.DEVICE_LIB_PATH)
    lib.pt_debug_quantise_node.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_void_p]
    rng = np.random.default_rng(12)
    empty_ref = (1 << 27) | 12345
    for trial in range(3000):
        n = int(rng.integers(1, 5))
        centre = rng.normal(size=3) * 10.0 ** rng.uniform(-3, 4)
        size = 10.0 ** rng.uniform(-6, 3)
        lo = np.ones((4, 3), np.float32)
        hi = -np.ones((4, 3), np.float32)
        empty = np.ones(4, np.uint8)
        for k in range(n):
            a = centre + rng.uniform(-1, 1, 3) * size
            ext = rng.uniform(0, 1, 3) * size * (rng.random(3) > 0.15)  # some axes flat
            lo[k], hi[k], empty[k] = a.astype(np.float32), (a + ext).astype(np.float32), 0
            hi[k] = np.maximum(hi[k], lo[k])
        refs = np.arange(4, dtype=np.uint32) + 7
        out = np.zeros(16, np.uint32)
        assert lib.pt_debug_quantise_node(lo.ctypes.data, hi.ctypes.data, refs.ctypes.data, empty.ctypes.data, empty_ref, out.ctypes.data) == 0
        origin = out[:3].view(np.float32)
        scale = [out[3:4].view(np.float32)[0], out[10:11].view(np.float32)[0], out[11:12].view(np.float32)[0]]  # scaleX, scaleY, scaleZ (pt_device.h)
        for sc in scale:
            m, _ = np.frexp(sc)
            assert m == 0.5 and sc > 0, "the scale of an axis is a power of two"
        q = out[4:10]  # qlox, qhix, qloy, qhiy, qloz, qhiz: byte k = child k
        for k in range(4):
            assert out[12 + k] == (empty_ref if empty[k] else refs[k])
            for a in range(3):
                ql, qh = (int(q[2 * a]) >> (8 * k)) & 0xFF, (int(q[2 * a + 1]) >> (8 * k)) & 0xFF
                if empty[k]:
                    assert ql > qh, "an unused slot is an inverted box"
                    continue
                plane_lo = np.float32(origin[a] + scale[a] * np.float32(ql))
                plane_hi = np.float32(origin[a] + scale[a] * np.float32(qh))
                assert plane_lo <= lo[k, a] and plane_hi >= hi[k, a], (trial, k, a, plane_lo, lo[k, a], plane_hi, hi[k, a])
                # and tight to within one quantisation step
                assert lo[k, a] - plane_lo <= 1.001 * scale[a] and plane_hi - hi[k, a] <= 1.001 * scale[a]
