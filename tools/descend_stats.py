"""Diagnostic: what the shared descent (pt_descend.h) does on the benchmark scene, and what it leaves to the per-ray kernel
(needs the -DPT_TRACE_STATS build: tools/mkvariants.sh stats -DPT_TRACE_STATS)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTAMD_LIB", os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_stats.so"))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ptamd import scenes, device as D


def read_trace():
    out = (C.c_ulonglong * 64)()
    assert D.lib().pt_debug_trace_stats(out, 64) == 0
    return list(out)


def read_descend():
    out = (C.c_ulonglong * 16)()
    assert D.lib().pt_debug_descend_stats(out, 16) == 0
    return list(out)


def lanes(tag, s, nrays):
    it, act, ki, kl, ks, li, ll, ls, ho, hr = s[:10]
    if it == 0:
        print(f"  {tag}: not traced by k_trace")
        return
    print(f"  {tag}: rays {nrays}  wave-iterations per 64 rays {it * 64 / max(nrays, 1):6.1f}  active lanes per iteration {act / it:5.1f}  inner / leaf / special iterations "
          f"{ki / it:.2f} / {kl / it:.2f} / {ks / it:.2f}  lanes served per inner / leaf step {li / max(ki, 1):5.1f} / {ll / max(kl, 1):5.1f}  inner lane-steps per ray {li / max(nrays, 1):5.2f}  "
          f"leaf lane-steps per ray {ll / max(nrays, 1):5.2f}  hand-outs of {hr / max(ho, 1):.1f} rays")


W, Hh = 1920, 1080
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = scenes.instanced_grid(W, Hh, level=6)
for use in (0, 3):
    os.environ["PTAMD_DESCENT"] = str(use)
    for mb in (1, 2):
        c = D.Context(W, Hh, seed=1, samples_in_flight=SPP, max_bounces=mb, flags=FLAGS)
        c.upload_scene(b.flat, sky=b.sky)
        c.set_camera(b.camera)
        c.render(SPP)
        read_trace(), read_descend()
        c.reset_stats()
        c.render(SPP)
        st = c.stats()
        s, d = read_trace(), read_descend()
        print(f"descent {use}, bounces < {mb}, flags {FLAGS}: descent launches {st['descent_launches']}")
        lanes("closest hit", s[:24], st["rays_extension"])
        lanes("any hit    ", s[24:48], st["rays_shadow"])
        if d[0]:
            print(f"  k_descend: packets {d[0]}  shared steps per packet {d[1] / d[0]:.2f}  followers per step {d[2] / max(d[1], 1):.1f}  stacked entries per ray {d[4] / max(d[3], 1):.2f}  "
                  f"rays that end on a leaf or instance {d[5] / max(d[3], 1):.3f}  drop-outs per ray: stack full {d[6] / max(d[3], 1):.3f}, origin outside the chosen child {d[7] / max(d[3], 1):.3f}  "
                  f"packets ended for lack of followers {d[8] / d[0]:.3f}")
        c.close()
