"""How long does the GPU parity mode (LFSR113 per slot, ordered compaction) track the serial oracle?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orclib as O, gpu_util as U
from ptamd import scenes, layout as L, device as D

cases = {
 "diffuse": scenes.cornell_box(64, 36),
 "pbr": scenes.cornell_box(64, 36, box_materials=[L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), L.material_pbr_dielectric((0.2, 0.5, 0.8), 0.6)]),
 "glass": scenes.cornell_box(64, 36, box_materials=[L.material_basic_refractive(1.5, (1, .6, .6), 5.0), L.material_refractive(0.9, 1.5, (.6, 1, .6), 5.0)]),
}
for name, b in cases.items():
    sky = np.full((1, 2, 2, 4), 0.5, np.float32)
    ctx = U.make_ctx(D, b, 64, 36, sky=sky, rng_mode=D.RNG_LFSR113_PARITY)
    sc = O.BoundScene(b.flat, sky=sky)
    st = O.QueueState(64, 36, 64 * 36); streams = O.create_streams(64 * 36)
    done = 0
    for spp in (1, 2, 4, 16, 64, 256):
        ctx.render(spp - done)
        for _ in range(spp - done):
            O.trace_rays("oracle", sc, b.camera, st, streams)
        done = spp
        a, g = ctx.read_accum()[:, :3], st.accum[:, :3]
        close = np.isclose(a, g, rtol=1e-3, atol=1e-3 * g.max()).all(axis=1)
        e = U.rmse(U.tonemap(a, spp, b.camera), U.tonemap(g, spp, b.camera))
        print(f"{name:8s} spp {spp:4d}: pixels within 1e-3: {close.mean():.4f}  tonemapped RMSE {e:.2e}  mean bias {abs(a.mean()-g.mean())/g.mean():.2e}")
    ctx.close()
