"""GPU box: PTAMD_TREE_HASH=1 makes the device library print a hash of everything a static conversion produces; this uploads a handful of scenes
(every builder, leaves of one to three triangles, an SBVH with duplicated references, parity mode) -- run before and after a change to the
conversion that is meant to keep its result, and compare the output."""
import os, sys
os.environ["PTAMD_TREE_HASH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import numpy as np  # noqa: E402
from ptamd import device as D, host as H, layout as L, scenes  # noqa: E402


def upload(name, bundle, flags=0, rng_mode=D.RNG_COUNTER):
    sys.stderr.write(f"== {name}\n")
    sys.stderr.flush()
    ctx = D.Context(64, 64, seed=1, device=0, samples_in_flight=1, flags=flags, rng_mode=rng_mode)
    try:
        ctx.upload_scene(bundle.flat, sky=None)
    finally:
        ctx.close()


upload("cornell", scenes.cornell_box(64, 64))
for b, n in ((H.BVH_BINNED_SAH, "binned"), (H.BVH_BINNED_FAST, "fast"), (H.BVH_SPATIAL_SPLIT, "spatial")):
    upload(f"blob_room_4_{n}", scenes.blob_room(64, 64, level=4, builder=b))
upload("grid_4x3_level5", scenes.instanced_grid(64, 64, level=5))
upload("grid_parity", scenes.instanced_grid(64, 64, level=3), rng_mode=D.RNG_LFSR113_PARITY)
upload("crowd", scenes.instanced_crowd(64, 64, nx=4, nz=3, level=3))
upload("mixed", scenes.mixed_material_room(64, 64, level=4))
