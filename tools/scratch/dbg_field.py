"""GPU box (diagnostic): the rays of the 1000-instance field whose hit differs between world-space copies and entered instances."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ptamd import scenes, device as D
import orclib as O, gpu_util as U
W, Hh = 256, 144
b = scenes.instance_field(W, Hh, n=1000, level=3)
flat = b.flat
baked = U.make_ctx(D, b, W, Hh)
entered = U.make_ctx(D, b, W, Hh, flags=D.FLAG_NO_BAKED_INSTANCES)
sc = U.oracle_scene(b)
side = 0.45 * 1000 ** 0.5
o, d = U.random_rays(40000, 11, (-side, 0.05, -side), (side, 4, side))
gb, ge, w = baked.intersect(o, d), entered.intersect(o, d), O.intersect_batch(sc, o, d, threads=8)
both = (gb["prim"] >= 0) & (w["prim"] >= 0)
dt = np.abs(gb["t"] - w["t"]) / np.maximum(np.abs(w["t"]), 1e-6)
bad = np.flatnonzero(both & (dt > 2e-4))
print("baked vs oracle: bad rays", len(bad))
# world-space triangles of every instance, float64 brute force for the bad rays
top = flat.top_nodes
leaves = np.flatnonzero(top["isLeaf"] != 0)
tri = flat.triangles["indices"]; V = flat.vertices["vertex"][:, :3].astype(np.float64)
def brute(ro, rd):
    best = (np.inf, -1, -1, 0, 0)
    for li in leaves:
        inv = top["invTransform"][li].reshape(4, 4).T.astype(np.float64)  # column-major -> matrix
        o2 = inv[:3, :3] @ ro + inv[:3, 3]; d2 = inv[:3, :3] @ rd
        # triangles of that mesh: all triangles reachable from sub-node a -- brute force over ALL triangles whose vertices lie in the mesh (here: test all, cheap enough)
        root = int(top["a"][li])
        lo, hi = mesh_range[root]
        p0, p1, p2 = V[tri[lo:hi, 0]], V[tri[lo:hi, 1]], V[tri[lo:hi, 2]]
        e1, e2 = p1 - p0, p2 - p0
        P = np.cross(d2, e2); det = (e1 * P).sum(1)
        ok = np.abs(det) > 1e-300
        T = o2 - p0; u = (T * P).sum(1) / det; Q = np.cross(T, e1); v = (Q @ d2) / det; t = (e2 * Q).sum(1) / det
        hit = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0)
        if hit.any():
            k = np.argmin(np.where(hit, t, np.inf))
            if t[k] < best[0]:
                best = (t[k], lo + k, li, u[k], v[k])
    return best
# triangle range of each mesh root: follow the sub-BVH
sub = flat.sub_nodes
mesh_range = {}
for root in {int(r) for r in top["a"][leaves]}:
    todo, lo, hi = [root], 1 << 30, 0
    while todo:
        i = todo.pop()
        if sub["count"][i]:
            lo, hi = min(lo, int(sub["left"][i])), max(hi, int(sub["left"][i]) + int(sub["count"][i]))
        else:
            todo += [int(sub["left"][i]), int(sub["left"][i]) + 1]
    mesh_range[root] = (lo, hi)
for k in bad[:12]:
    bt = brute(o[k].astype(np.float64), d[k].astype(np.float64))
    print(f"ray {k}: baked t={gb['t'][k]:.6f} prim={gb['prim'][k]} inst={gb['inst'][k]} uv=({gb['u'][k]:.4f},{gb['v'][k]:.4f}) | entered t={ge['t'][k]:.6f} prim={ge['prim'][k]} inst={ge['inst'][k]} | oracle t={w['t'][k]:.6f} prim={w['prim'][k]} inst={w['inst'][k]} uv=({w['u'][k]:.4f},{w['v'][k]:.4f}) | brute f64 t={bt[0]:.6f} prim={bt[1]} top={bt[2]} uv=({bt[3]:.4f},{bt[4]:.4f})")
