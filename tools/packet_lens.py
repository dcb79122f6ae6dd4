"""GPU box: does the packet kernel still pay for thin-lens (depth-of-field) primary rays?  python tools/packet_lens.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import scenes, host as H, device as D
W, Hh, SPP = 1920, 1080, 128
for lens in (False, True):
    b = scenes.instanced_grid(W, Hh, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT, thin_lens=lens)
    for flags, name in ((0, "packets"), (D.FLAG_NO_PACKETS, "per-ray")):
        ctx = D.Context(W, Hh, seed=1, samples_in_flight=SPP, flags=flags)
        ctx.upload_scene(b.flat, sky=b.sky)
        ctx.set_camera(b.camera)
        ctx.render(SPP)
        ctx.profile_kernels(True)
        ctx.reset_stats()
        ctx.render(SPP)
        st = ctx.stats()
        print(f"thin_lens={lens} {name:8s}: intersect {st['ms_intersect']:.2f} ms (packet part {st['ms_packet']:.2f}), shade {st['ms_shade']:.2f}, shadow {st['ms_shadow']:.2f}, "
              f"{(st['rays_extension'] + st['rays_shadow']) / (st['ms_intersect'] + st['ms_shade'] + st['ms_shadow'] + st['ms_gen']) / 1e3:.0f} Mrays/s over the kernels", flush=True)
        ctx.close()
