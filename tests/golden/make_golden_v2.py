"""Generates tests/golden/golden_v2.npz -- fixtures added in round 2 -- by RUNNING THE REFERENCE'S OWN KERNELS
(oracle/_ref, see oracle/Makefile) on inputs already held by golden_v1.npz.  Only inputs and expected outputs are
stored.  Run in the build container (needs /root/reference for `make -C oracle ref`):

    python tests/golden/make_golden_v2.py

(1) Queue semantics with slot refill on the refraction-free `plain` scene (64x36 image, MAX_ACTIVE_RAYS = 512 < 2304
    pixels, raytracer.cpp:323-427 / kernel.cl:33-40): per-pass counters and the accumulator after 1 and 2 samples.
    fp32 round-off flips a decision so rarely in this scene that a GPU render in parity mode follows these sums pixel
    by pixel, which makes it a real test of the refill loop (the `inst` fixture of v1 contains glass).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
for p in (os.path.join(ROOT, "opencl-path-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import golden_io  # noqa: E402
import orclib as O  # noqa: E402


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle ref` in the build container"
    g = golden_io.load()
    flat, cam, sky, tex = golden_io.scene_inputs(g, "plain")
    sc = O.BoundScene(flat, sky=sky, material_textures=tex)
    G = {}
    W, H, CAP = 64, 36, 512
    st = O.QueueState(W, H, CAP)
    s = O.create_streams(W * H, use_ref=True)
    for spp in (1, 2):
        trace, _ = O.trace_rays("ref", sc, cam, st, s)
        G[f"refill_plain_trace_{spp}spp"] = trace
        G[f"refill_plain_accum_{spp}spp"] = st.accum[:, :3].copy()
    G["refill_plain_shape"] = np.array([W, H, CAP], np.uint32)
    out = os.path.join(HERE, "golden_v2.npz")
    np.savez_compressed(out, **G)
    print(out, {k: v.shape for k, v in G.items()}, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
