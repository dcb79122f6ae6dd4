"""Diagnostic: lane-level accounting of the k_trace loop (needs the -DPT_TRACE_STATS build)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTAMD_LIB", os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_stats.so"))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import numpy as np
from ptamd import scenes, device as D

def read():
    out = (C.c_ulonglong * 64)()
    assert D.lib().pt_debug_trace_stats(out, 64) == 0
    return list(out)

def report(tag, s, nrays):
    it, act, ki, kl, ks, li, ll, ls, ho, hr = s[:10]
    if it == 0:
        print(f"{tag:12s} rays {nrays:8d}: not traced by k_trace (packet kernel)")
        return
    print(f"{tag:12s} rays {nrays:8d} wave-iters/ray*64 {it*64/max(nrays,1):6.1f}  active/iter {act/max(it,1):5.1f}  iters inner/leaf/special {ki/it:.2f}/{kl/it:.2f}/{ks/it:.2f} "
          f" served/iter inner {li/max(ki,1):5.1f} leaf {ll/max(kl,1):5.1f} special {ls/max(ks,1):5.1f}  overall {(li+ll+ls)/it:5.1f}  handouts {ho} ({hr/max(ho,1):.1f} rays each)")
    if s[18]:
        print(f"             inner lane-steps at a node one of whose children contains the ray origin: {s[18] / max(li, 1):.3f} of all inner lane-steps")
    # (counter 16 is the hand-out's LDS-read CYCLES, pt_trace.h PT_TOC(16, tShfl) -- printed with the cycle shares below; round 2
    #  printed it here as "triangles tested per ray", which it is not: profiles/round2/r2v_trace_stats_*.txt carry that wrong line)
    tot, tin, tle, tsp, tha = s[10], s[11], s[12], s[13], s[14]
    if tot:
        print(f"             wave cycles: inner {100*tin/tot:4.1f}% ({tin/max(ki,1):6.0f}/step)  leaf {100*tle/tot:4.1f}% ({tle/max(kl,1):6.0f}/step)  special {100*tsp/tot:4.1f}% ({tsp/max(ks,1):6.0f}/pass)"
              f"  hand-out {100*tha/tot:4.1f}% ({tha/max(ho,1):6.0f}/hand-out)  (request {100*s[15]/tot:4.1f}% shuffles {100*s[16]/tot:4.1f}% assign {100*s[17]/tot:4.1f}%)  vote+rest {100*(tot-tin-tle-tsp-tha)/tot:4.1f}% ({(tot-tin-tle-tsp-tha)/it:6.0f}/iter)  total/iter {tot/it:6.0f}")

W, Hh = 1920, 1080
FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 2: every instance entered (PT_FLAG_NO_BAKED_INSTANCES)
print(f"context flags {FLAGS}")
b = scenes.instanced_grid(W, Hh, level=6)
ctx = D.Context(W, Hh, seed=1, flags=FLAGS)
ctx.upload_scene(b.flat, sky=b.sky); ctx.set_camera(b.camera)
o, d, _ = ctx.gen_rays(0, W * Hh)
REP = 8  # launch size comparable to one bench launch
o, d = np.tile(o, (REP, 1)), np.tile(d, (REP, 1))
read()
r = ctx.intersect(o, d)
report("primary", read(), len(o))
hit = r["prim"] >= 0
rng = np.random.default_rng(0)
p = o[hit] + d[hit] * r["t"][hit][:, None] * 0.999
nd = rng.normal(size=p.shape).astype(np.float32); nd /= np.linalg.norm(nd, axis=1, keepdims=True); nd[:, 1] = np.abs(nd[:, 1])
ctx.intersect(p, nd)
report("secondary", read(), len(p))
tm = np.full(len(p), 3.0, np.float32)
ctx.intersect(p, nd, tmax=tm, any_hit=True)
report("shadow-like", read()[24:], len(p))

# the real pipeline: one 32-sample batch, per bounce depth limit (difference between rows = that bounce)
for mb in (1, 2, 4):
    c2 = D.Context(W, Hh, seed=1, samples_in_flight=64, max_bounces=mb, flags=FLAGS)
    c2.upload_scene(b.flat, sky=b.sky); c2.set_camera(b.camera)
    c2.render(64); read(); c2.reset_stats()
    c2.render(64); st = c2.stats(); s = read()
    report(f"render b<{mb} ext", s[:24], st["rays_extension"])
    report(f"render b<{mb} shd", s[24:], st["rays_shadow"])
    c2.close()
