"""GPU box: how long do scene construction and the two uploads of the bench scene take?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import scenes, host as H, device as D
t0 = time.perf_counter()
b = scenes.instanced_grid(1920, 1080, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT)
flat = b.flat
t1 = time.perf_counter()
for flags, name in ((0, "instances baked"), (D.FLAG_TWO_LEVEL_ONLY, "two-level")):
    ctx = D.Context(1920, 1080, flags=flags)
    t2 = time.perf_counter()
    ctx.upload_scene(flat, sky=b.sky)
    t3 = time.perf_counter()
    print(f"{name}: upload_scene {t3 - t2:.2f} s")
    ctx.close()
print(f"host scene build (2 SBVH meshes of 82k triangles + flatten): {t1 - t0:.2f} s")
