"""Loader for tests/golden/golden_v1.npz (see tests/golden/make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "opencl-path-tracer_amd"))
from ptamd import layout as L  # noqa: E402
from ptamd.host import FlatScene  # noqa: E402

_cache = None


def load():
    global _cache
    if _cache is None:
        z = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))
        out = {}
        for k in z.files:
            if "@" in k:
                name, dt = k.split("@")
                dtype = getattr(L, dt)
                out[name] = np.frombuffer(z[k].tobytes(), dtype).copy()
            else:
                out[k] = z[k]
        _cache = out
    return _cache


def flat_scene(g, prefix):
    return FlatScene(g[prefix + "vertices"], g[prefix + "triangles"], g[prefix + "materials"], g[prefix + "sub_nodes"],
                     g[prefix + "lights"], g[prefix + "top_nodes"], int(g[prefix + "top_root"]),
                     int((g[prefix + "top_nodes"]["isLeaf"] != 0).sum()))


def scene_inputs(g, name):
    """(flat, camera, sky, material_textures) of golden scene `name` ('mixed' | 'inst')."""
    p = f"scene_{name}_"
    return flat_scene(g, p), g[p + "camera"][0], g[p + "sky"], g[p + "tex"]
