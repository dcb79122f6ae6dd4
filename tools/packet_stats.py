"""Diagnostic (GPU box, -DPT_TRACE_STATS build): what a closest-hit beam packet (or bundle of four, k_trace_multi) of the bench scene does -- nodes, leaves, instance entries."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTAMD_LIB", os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_stats.so"))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import scenes, device as D

def read():
    out = (C.c_ulonglong * 64)()
    assert D.lib().pt_debug_trace_stats(out, 64) == 0
    return list(out)

W, Hh = 1920, 1080
b = scenes.instanced_grid(W, Hh, level=6)
for name, flags in (("instances copied", 0), ("single leaves copied", D.FLAG_TWO_LEVEL_ONLY), ("every instance entered", D.FLAG_NO_BAKED_INSTANCES)):
    ctx = D.Context(W, Hh, seed=1, samples_in_flight=64, flags=flags)
    ctx.upload_scene(b.flat, sky=b.sky); ctx.set_camera(b.camera)
    ctx.render(64); read()
    ctx.render(64); ctx.synchronize()
    s = read()
    p = max(s[48], 1)
    print(f"{name:24s}: {s[48]} beam packets, per packet {s[49]/p:6.1f} nodes {s[50]/p:5.1f} leaves {s[51]/p:5.2f} instance entries ({s[52]/p:5.2f} found nothing below the root), "
          f"{s[53]/max(s[51],1):7.0f} cycles per entry, {s[54]} packets started over per lane")
    if s[56]:
        q = s[56]
        print(f"{'':24s}  {s[56]} bundles of 4 x 64 on the beam walk (k_trace_multi), per bundle {s[57]/q:6.1f} nodes {s[58]/q:5.1f} leaves {s[59]/q:6.1f} triangles, "
              f"{s[60]} started over sub-packet by sub-packet")
    ctx.close()
