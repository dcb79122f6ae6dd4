"""Old tree (commit f505f8e, the last one that held the 8-wide compressed BVH: pt_wide8.h / pt_trace8.h): the per-ray kernels on 4-wide
nodes (-DPT_BVH8=0) against 8-wide nodes (-DPT_BVH8=3), batch throughput and 1-spp 720p frames.  PTAMD_LIB selects the library."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "opencl-path-tracer_amd"))
import torch  # noqa
from ptamd import device as D, host as H, layout as L, scenes
what = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {"lib": os.path.basename(os.environ.get("PTAMD_LIB", "default"))}
if what in ("all", "batch"):
    W, Hh = 1920, 1080
    b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT)
    ctx = D.Context(W, Hh, seed=1, samples_in_flight=128)
    ctx.upload_scene(b.flat, sky=b.sky)
    ctx.set_camera(b.camera)
    ctx.render(128)
    ctx.synchronize(); ctx.clear(); ctx.reset_stats(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.render(128, sync=False)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    out["batch"] = {"mrays_per_s": round((st["rays_extension"] + st["rays_shadow"]) / dt / 1e6, 1), "rays_extension": st["rays_extension"], "rays_shadow": st["rays_shadow"],
                    "rays_generated": st["rays_generated"], "packet_launches": st.get("packet_launches", None), "renders": 3}
    ctx.close()
if what in ("all", "frame"):
    W, Hh = 1280, 720
    fr = {}
    for name, mat in (("glass", L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0)), ("diffuse", L.material_diffuse((0.8, 0.8, 0.8)))):
        b = scenes.blob_room(W, Hh, material=mat, builder=H.BVH_SPATIAL_SPLIT, level=6)
        ctx = D.Context(W, Hh, seed=1, samples_in_flight=1)
        ctx.upload_scene(b.flat, sky=b.sky)
        ctx.set_camera(b.camera)
        for _ in range(20):
            ctx.render(1)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.render(1, sync=False)
            ctx.synchronize()
        fr[name] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        ctx.close()
    out["frame_ms"] = fr
print(json.dumps(out), flush=True)
