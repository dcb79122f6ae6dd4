"""Diagnostic (round 5): of the OCCLUDED shadow rays of a batch, how many die in the leaf that stopped the previous
occluded ray of the same lane / of the same wave?  A first-test cache (test that leaf's triangles before walking the tree) pays only if the
share is large (the review's bar: > 25 %).  Needs the -DPT_TRACE_STATS build (tools/mkvariants.sh stats "-DPT_TRACE_STATS")."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTAMD_LIB", os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_stats.so"))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import scenes, device as D, host as H, layout as L


def read():
    out = (C.c_ulonglong * 64)()
    assert D.lib().pt_debug_trace_stats(out, 64) == 0
    return list(out)


W, Hh = 1920, 1080
IN_FLIGHT = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cases = [("config 4 (instances copied)", scenes.instanced_grid(W, Hh, level=6), 0),
         ("config 2 (diffuse mesh in the room)", scenes.blob_room(W, Hh, level=6, builder=H.BVH_BINNED_SAH), 0)]
for name, b, flags in cases:
    for mb in (1, 4):
        ctx = D.Context(W, Hh, seed=1, samples_in_flight=IN_FLIGHT, max_bounces=mb, flags=flags)
        ctx.upload_scene(b.flat, sky=b.sky, material_textures=b.material_textures)
        ctx.set_camera(b.camera)
        ctx.render(IN_FLIGHT)
        read()
        ctx.reset_stats()
        ctx.render(IN_FLIGHT)
        st, s = ctx.stats(), read()[24:48]
        occ = max(s[19], 1)
        print(f"{name}, {IN_FLIGHT} in flight, bounces < {mb}: {st['rays_shadow']} shadow rays, {s[19]} occluded ({s[19] / max(st['rays_shadow'], 1):.3f}); "
              f"of the occluded: same leaf as the lane's previous occluded ray {s[20] / occ:.3f}, same triangle {s[22] / occ:.3f}; "
              f"same leaf as the wave's latest occluder leaf {s[21] / occ:.3f}, one of the wave's last four {s[23] / occ:.3f}")
        ctx.close()
