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
