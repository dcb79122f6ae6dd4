"""GPU box: the rebuilt-tree-per-frame tick of bench.py's `dynamic.refit.rebuild_20k`, stage by stage (host clock), without the rest of the bench."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import device as D, host as H, layout as L, scenes  # noqa: E402


OLD_ORDER = "--old-order" in sys.argv


def main():
    W, Hh = (1920, 1080) if "--1080p" in sys.argv else (1280, 720)
    v, f = scenes.icosphere(5)
    p0 = (v * 0.5).astype(np.float32)
    f = f.astype(np.uint32)
    mb = scenes._MeshBuilder()
    mats = scenes._room_materials()
    scenes._room(mb, mats)
    room = mb.build(mats, H.BVH_BINNED_SAH)
    cam = scenes.blob_room(W, Hh, level=2).camera
    mat = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    ctx = D.Context(W, Hh, seed=1, device=0, samples_in_flight=1)

    def new_scene(k):
        p = (p0 * (1.0 + 0.1 * np.sin(k + 1.0 + 5.0 * p0[:, :1]))).astype(np.float32)
        t0 = time.perf_counter()
        mesh = H.Mesh(p, f, [mat], builder=H.BVH_BINNED_FAST)
        t1 = time.perf_counter()
        scene = H.Scene()
        scene.add_node(room)
        scene.add_node(mesh, location=(0.0, 0.8, 0.1), scale=(1.2, 1.2, 1.2))
        flat = scene.flatten()
        return flat, t0, t1, time.perf_counter()

    flat, _, _, _ = new_scene(0)
    ctx.upload_scene(flat, sky=None)
    ctx.set_camera(cam)
    ctx.render(1)
    rows = []
    for k in range(1, 14):
        ctx.render(1, sync=False)
        flat, t0, t1, t2 = new_scene(k)
        if not OLD_ORDER:
            ctx.render(1, sync=False)
        t2b = time.perf_counter()
        ctx.upload_static_async(flat)
        t3 = time.perf_counter()
        ctx.upload_dynamic_async(flat)
        t4 = time.perf_counter()
        if OLD_ORDER:  # rounds 5-6a: the second frame of the old scene enqueued after the conversion
            ctx.render(1, sync=False)
        t5 = time.perf_counter()
        ctx.frame_tick()
        t6 = time.perf_counter()
        ctx.render(1, sync=False)
        t7 = time.perf_counter()
        ctx.synchronize()
        t8 = time.perf_counter()
        rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2b, t3), (t3, t4), (t2, t2b) if not OLD_ORDER else (t4, t5), (t5, t6), (t6, t7), (t7, t8), (t0, t8))])
    names = ["build", "flatten", "static_async", "dynamic_async", "render_enqueue", "tick", "render_enqueue2", "synchronize", "total"]
    med = np.median(np.array(rows[2:]), axis=0)
    print(" ".join(f"{n}={m:.3f}" for n, m in zip(names, med)))
    ctx.close()


main()
