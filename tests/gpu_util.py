"""Helpers shared by the GPU parity tests (all calls go through the C-ABI via ptamd.device)."""
import numpy as np

import orclib as O


def make_ctx(D, bundle_or_flat, width, height, camera=None, sky=None, tex=None, **kw):
    flat = getattr(bundle_or_flat, "flat", bundle_or_flat)
    if hasattr(bundle_or_flat, "flat"):
        camera = bundle_or_flat.camera if camera is None else camera
        sky = bundle_or_flat.sky if sky is None else sky
        tex = bundle_or_flat.material_textures if tex is None else tex
    ctx = D.Context(width, height, **kw)
    ctx.upload_scene(flat, sky=sky, material_textures=tex)
    if camera is not None:
        ctx.set_camera(camera)
    return ctx


def oracle_scene(bundle_or_flat, sky=None, tex=None):
    flat = getattr(bundle_or_flat, "flat", bundle_or_flat)
    if hasattr(bundle_or_flat, "flat"):
        sky = bundle_or_flat.sky if sky is None else sky
        tex = bundle_or_flat.material_textures if tex is None else tex
    return O.BoundScene(flat, sky=sky, material_textures=tex)


def random_rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def tri_vertex_ids(flat, prim):
    """Geometric identity of a hit triangle: sorted global vertex indices (SBVH duplicates references)."""
    return np.sort(flat.triangles["indices"][prim], axis=1)


def compare_hits(flat, got, want, t_rtol=2e-4, max_tie_frac=1e-2, edge_flip_frac=0.0, t_outlier_frac=0.0, uv_atol=5e-3, t_atol=1e-6):
    """hit/miss identical; t/u/v within tolerance; (prim, inst) identical except where two primitives are
    hit at the same t within tolerance (shared edges, SBVH-duplicated references): those must agree on t.
    `t_outlier_frac` > 0 (deformed meshes: twisted triangles become slivers, and t of a hit that grazes a sliver is ill-conditioned)
    tolerates that fraction of rays beyond t_rtol, each within 1 %.
    `edge_flip_frac` > 0 (world-space copies of instances: triangle edges are rounded in another space than the
    reference's) tolerates that fraction of hit/miss disagreements, each of which must graze a triangle edge.
    `t_atol`: the absolute error of t scales with the magnitude of the coordinates, not with t (a ray that starts 5e-4 in front of a
    surface 15 units from the origin has t good to a few 1e-6): scenes far larger than the unit cube pass a few ulps of their extent.
    `uv_atol`: barycentrics are compared through the hit point they encode; the direct bound on u and v is loose (they are
    ill-conditioned for small or distant triangles: 5e-3 by default, more for scenes of tiny triangles seen from afar)."""
    gh, wh = got["prim"] >= 0, want["prim"] >= 0
    flips = gh != wh
    edge_flips = 0
    if edge_flip_frac > 0 and flips.any():
        for k in np.flatnonzero(flips):
            h = got if gh[k] else want
            u, v = float(h["u"][k]), float(h["v"][k])
            assert min(abs(u), abs(v), abs(1.0 - u - v)) < 2e-4, f"ray {k}: hit/miss differs away from any edge (u={u}, v={v})"
        edge_flips = int(flips.sum())
        assert flips.mean() <= edge_flip_frac, f"{edge_flips} edge-grazing rays flip"
        flips = np.zeros_like(flips)
    assert flips.mean() <= 1e-4, f"hit/miss differs for {flips.sum()} rays"
    both = gh & wh
    dt = np.abs(got["t"][both] - want["t"][both]) / np.maximum(np.abs(want["t"][both]), 1e-6)
    bad = ~np.isclose(got["t"][both], want["t"][both], rtol=t_rtol, atol=t_atol)
    assert bad.mean() <= t_outlier_frac, f"t differs for {bad.sum()} of {both.sum()} rays, worst {dt.max():.3e}"
    for k in np.flatnonzero(bad):  # a tolerated outlier is within 1 %, or its NEARER hit grazes a triangle edge (the other side missed that
        if dt[k] < 1e-2:           # triangle by round-off and reports what lies behind it)
            continue
        h = got if got["t"][both][k] < want["t"][both][k] else want
        u, v = float(h["u"][both][k]), float(h["v"][both][k])
        assert min(abs(u), abs(v), abs(1.0 - u - v)) < 2e-3, f"t differs by {dt[k]:.2e} away from any edge (u={u}, v={v})"
    same_geom = np.all(tri_vertex_ids(flat, got["prim"][both]) == tri_vertex_ids(flat, want["prim"][both]), axis=1)
    same = same_geom & (got["inst"][both] == want["inst"][both])
    assert (~same).mean() <= max_tie_frac, f"{(~same).sum()} of {both.sum()} rays hit a different primitive"
    # a different primitive is only legitimate as a tie: coincident / edge-sharing triangles hit at the same t
    # (which one is reported depends on the visit order, which is ours -- see pt_trace.h)
    same = same | bad  # (the tolerated t outliers hit what lies behind a gap: another primitive at another distance)
    tie_dt = np.abs(got["t"][both][~same] - want["t"][both][~same]) / np.maximum(want["t"][both][~same], 1e-6)
    assert tie_dt.size == 0 or tie_dt.max() < 2e-5, f"different primitive at a different distance: {tie_dt.max():.2e}"
    # barycentrics: compared through the hit point they encode (object space), v0 + u*e1 + v*e2, because
    # u and v of a grazing hit are ill-conditioned (errors scale with 1/det) while the point is not
    same = same & ~bad
    tri = flat.triangles["indices"][want["prim"][both][same]]
    p0, p1, p2 = (flat.vertices["vertex"][tri[:, k], :3].astype(np.float64) for k in range(3))

    def point(h):
        u, v = h["u"][both][same].astype(np.float64)[:, None], h["v"][both][same].astype(np.float64)[:, None]
        return p0 + u * (p1 - p0) + v * (p2 - p0)
    extent = float(np.abs(flat.vertices["vertex"][:, :3]).max())
    assert np.abs(point(got) - point(want)).max() < 2e-4 * max(extent, 1.0)
    assert np.allclose(got["u"][both][same], want["u"][both][same], atol=uv_atol)
    assert np.allclose(got["v"][both][same], want["v"][both][same], atol=uv_atol)
    return dict(flips=int(flips.sum()), ties=int((~same).sum()), n=int(both.sum()), edge_flips=edge_flips)


def tonemap(accum_rgb, spp, camera):
    """mean -> exposure -> Reinhard, pre-gamma, values in [0,1): the image-parity space of SURVEY 8(d)."""
    ev = np.log2(float(camera["relativeAperture"]) ** 2 / float(camera["shutterTime"]) * 100 / float(camera["ISO"]))
    lum = accum_rgb / spp / (1.2 * 2.0 ** ev)
    return lum / (1 + lum)


def rmse(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


# ---- parity margins on the record: what every image gate measured, next to the gate (tests/conftest.py writes them out at the end of
# the session: gpurun_out/parity_margins.json on the GPU box, copied to profiles/roundN/ and quoted in DESIGN.md section 2)
MARGINS = {}


def record_margin(name, **values):
    MARGINS[name] = {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in values.items()}


def image_margins(name, got, want, spp, camera, bias_gate, rmse_gate, **extra):
    """bias of the image mean and tone-mapped RMSE (SURVEY 8(d)'s two statistics) of `got` against the oracle's `want`, recorded and gated"""
    bias = abs(float(got.mean()) - float(want.mean())) / float(want.mean())
    e = rmse(tonemap(got, spp, camera), tonemap(want, spp, camera))
    record_margin(name, bias=bias, bias_gate=bias_gate, tonemapped_rmse=e, rmse_gate=rmse_gate, spp=spp, pixels=int(len(got)), **extra)
    assert bias < bias_gate, (name, bias)
    assert e < rmse_gate, (name, e)
    return bias, e


def fraction_gate(name, close, measured=None, legacy=0.0, slack=0.98):
    """The fraction of entries / pixels that agree with the oracle within a test's tolerances, on the record (parity_margins.json) and gated at what was
    MEASURED, not at a round number (`> 0.85` and `>= 0.97` would have hidden a regression of several percent of a stratum).
    `measured`: {name: fraction} of the test file (the values of the committed profiles/roundN/parity_margins.json); the gate is measured x 0.98 -- the
    kernels and the counter PRNG are deterministic, what moves a fraction between boxes and builds is a handful of fp32 decision flips -- and never
    below the `legacy` gate of rounds 1-5; a name without a measurement is held against `legacy` alone and shows up as such in the record."""
    close = np.asarray(close)
    n = int(close.size)
    frac = float(close.mean()) if n else 1.0
    m = None if measured is None else (measured.get(name) if isinstance(measured, dict) else measured)
    gate = legacy if m is None else max(legacy, m * slack)
    record_margin("fraction: " + name, fraction=frac, n=n, measured=m, gate=gate, legacy_gate=legacy)
    assert frac >= gate, (name, frac, gate)
    return frac
