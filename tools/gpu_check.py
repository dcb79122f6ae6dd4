"""Ad-hoc GPU bring-up check (not a test): parity of intersect / render against the oracle + a first timing."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orclib as O
from ptamd import scenes, layout as L, host as H, device as D

def rand_rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)

def check_intersect(name, bundle, lo, hi, n=20000):
    flat = bundle.flat
    ctx = D.Context(bundle.width, bundle.height)
    ctx.upload_scene(flat, sky=bundle.sky, material_textures=bundle.material_textures)
    sc = O.BoundScene(flat, sky=bundle.sky, material_textures=bundle.material_textures)
    o, d = rand_rays(n, 1, lo, hi)
    g = ctx.intersect(o, d)
    c = O.intersect_batch(sc, o, d, threads=8)
    hit_g, hit_c = g["prim"] >= 0, c["prim"] >= 0
    both = hit_g & hit_c
    print(f"[{name}] closest: hit frac {hit_c.mean():.3f} hit-mismatch {(hit_g != hit_c).sum()} prim-mismatch {(g['prim'][both] != c['prim'][both]).sum()} "
          f"inst-mismatch {(g['inst'][both] != c['inst'][both]).sum()} max|dt|/t {np.max(np.abs(g['t'][both]-c['t'][both])/c['t'][both]) if both.any() else 0:.2e} "
          f"max|du| {np.max(np.abs(g['u'][both]-c['u'][both])) if both.any() else 0:.2e}")
    tmax = np.random.default_rng(2).uniform(0.1, 3.0, n).astype(np.float32)
    g = ctx.intersect(o, d, tmax=tmax, any_hit=True)
    c = O.intersect_batch(sc, o, d, tmax=tmax, any_hit=True, threads=8)
    print(f"[{name}] any-hit: occluded frac {c['prim'].mean():.3f} mismatch {(g['prim'] != c['prim']).sum()}")
    ctx.close()

def check_render(name, bundle, spp=16, parity=False, max_active=0):
    W, H_ = bundle.width, bundle.height
    flat = bundle.flat
    sky = bundle.sky if bundle.sky is not None else np.full((1, 4, 8, 4), 0.6, np.float32)
    ctx = D.Context(W, H_, rng_mode=D.RNG_LFSR113_PARITY if parity else D.RNG_COUNTER, seed=1, max_active_rays=max_active)
    ctx.upload_scene(flat, sky=sky, material_textures=bundle.material_textures)
    ctx.set_camera(bundle.camera)
    t = time.time(); ctx.render(spp); tg = time.time() - t
    a = ctx.read_accum()[:, :3]
    st = ctx.stats()
    sc = O.BoundScene(flat, sky=sky, material_textures=bundle.material_textures)
    if parity:
        N = max_active or (W * H_ + 63) // 64 * 64
        state = O.QueueState(W, H_, N); streams = O.create_streams(W * H_)
        for s in range(spp):
            O.trace_rays("oracle", sc, bundle.camera, state, streams)
        b = state.accum[:, :3]
        cnt = {}
    else:
        b, cnt = O.render(sc, bundle.camera, W, H_, spp, seed=1, threads=8)
        b = b[:, :3]
    diff = np.abs(a - b)
    scale = max(b.max(), 1e-6)
    bad = (diff.max(axis=1) > 1e-3 * np.maximum(np.abs(b).max(axis=1), 1e-3 * scale))
    print(f"[{name}] parity={parity} spp={spp} mean gpu {a.mean():.5f} cpu {b.mean():.5f} rel-mean-diff {abs(a.mean()-b.mean())/b.mean():.2e} "
          f"max abs diff {diff.max():.3e} (max val {scale:.3g}) pixels off>1e-3rel: {bad.sum()}/{len(bad)} gpu {tg:.3f}s")
    if cnt:
        print("   rays gpu ext/shadow/gen/hits", st["rays_extension"], st["rays_shadow"], st["rays_generated"], st["shade_hits"],
              " cpu", cnt["raysExtension"], cnt["raysShadow"], cnt["raysGenerated"], cnt["shadeHits"])
    ctx.close()

def timing(bundle, spp=4):
    W, H_ = bundle.width, bundle.height
    ctx = D.Context(W, H_, seed=1)
    t = time.time(); flat = bundle.flat; print("flatten %.2fs, instanced tris %d" % (time.time() - t, flat.instanced_triangles))
    ctx.upload_scene(flat, sky=bundle.sky, material_textures=bundle.material_textures)
    ctx.set_camera(bundle.camera)
    ctx.render(1)
    ctx.reset_stats(); ctx.profile_kernels(True)
    t = time.time(); ctx.render(spp); wall = time.time() - t
    st = ctx.stats()
    rays = st["rays_extension"] + st["rays_shadow"]
    print(f"[timing {bundle.name} {W}x{H_}] {spp} spp: wall {wall*1e3:.1f} ms, device {st['ms_last_render']:.1f} ms, rays {rays/1e6:.1f} M -> {rays/st['ms_last_render']/1e3:.1f} Mrays/s "
          f"(ext {st['rays_extension']/1e6:.1f}M shadow {st['rays_shadow']/1e6:.1f}M) ms: gen {st['ms_gen']:.2f} isect {st['ms_intersect']:.2f} shade {st['ms_shade']:.2f} shadow {st['ms_shadow']:.2f}")
    ctx.close()

if __name__ == "__main__":
    print(D.lib().pt_version())
    check_intersect("cornell", scenes.cornell_box(64, 36), (-1, 0, -1), (1, 2, 1))
    check_intersect("grid L3", scenes.instanced_grid(64, 36, level=3, sky_size=(64, 32)), (-4, 0, -4), (4, 3, 4))
    check_render("cornell", scenes.cornell_box(64, 36))
    mats = [L.material_basic_refractive(1.5, (1, .6, .6), 5.0), L.material_refractive(0.9, 1.5, (.6, 1, .6), 5.0)]
    check_render("cornell glass", scenes.cornell_box(64, 36, box_materials=mats))
    check_render("grid L3", scenes.instanced_grid(64, 36, level=3, sky_size=(64, 32)))
    check_render("cornell", scenes.cornell_box(64, 36), parity=True)
    check_render("grid L3 refill", scenes.instanced_grid(64, 36, level=3, sky_size=(64, 32)), parity=True, max_active=256)
    check_render("grid L3 prod refill", scenes.instanced_grid(64, 36, level=3, sky_size=(64, 32)), max_active=512)
    timing(scenes.instanced_grid(1920, 1080, level=6))
