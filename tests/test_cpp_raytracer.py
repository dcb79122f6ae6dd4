"""The kept C++ API (Scene / Mesh / Material / Camera / RayTracer, opencl-path-tracer_amd/host/) driving the
HIP path through the C-ABI, as a maintainer of the reference would: examples/render_cornell.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "examples", "render_cornell")


def test_without_gpu_the_cpp_path_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "1", str(tmp_path / "x.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "x.ppm").exists()


@pytest.mark.gpu
def test_cpp_raytracer_renders(gpu, tmp_path):
    out = tmp_path / "cornell.ppm"
    r = subprocess.run([EXE, "32", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split() if "=" in kv)
    assert int(fields["spp"]) == 32 and int(fields["rays"]) > 32 * 256 * 256
    data = out.read_bytes()
    assert data.startswith(b"P6\n256 256\n255\n") and len(data) == len(b"P6\n256 256\n255\n") + 256 * 256 * 3
    assert 0.002 < float(fields["mean"]) < 0.9  # exposure 4.9e-3 (EV100 7.4) makes the tone-mapped room dark
    # same frame again: bit-identical image (counter PRNG, deterministic sums)
    out2 = tmp_path / "again.ppm"
    subprocess.run([EXE, "32", str(out2)], check=True, capture_output=True, timeout=120)
    assert out2.read_bytes() == data


@pytest.mark.gpu
def test_cpp_deforming_mesh_loop(gpu):
    """examples/deform_loop.cpp: the reference's MeshSequence use through the kept C++ classes -- Mesh::refit, RayTracer::updateGeometry,
    RayTracer::frameTick, RayTracer::rayTrace per frame, no binding in between.  The tick (refit + pt_update_geometry + upload + flip +
    synchronise) is TIMED by bench.py (`dynamic.refit`, and this example's own output); here only what it does is asserted -- wall-clock
    gates on a shared GPU box fail for reasons that are no regression -- with one bound an order of magnitude above the measurement."""
    exe = os.path.join(ROOT, "examples", "deform_loop")
    r = subprocess.run([exe, "135", "135", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split() if "=" in kv)
    assert int(fields["triangles"]) == 36450 and int(fields["spp"]) == 13
    assert 0 < float(fields["until_adopted_ms"]) < 40.0, fields  # measured 0.8 ms; the reference publishes 4.1-6.3 per frame
    assert 0 < float(fields["mean_first"]) and abs(float(fields["mean_last"]) - float(fields["mean_first"])) > 1e-6


@pytest.mark.gpu
def test_a_dynamic_mesh_that_does_not_count_generations_is_uploaded_every_time(gpu):
    """examples/untracked_mesh.cpp: an IMesh that is not raytracer::Mesh (the reference's MeshSequence is such a class: isDynamic(), no generation
    counter).  RayTracer::updateGeometry takes it for changed and hands the re-flattened arrays over (pt_update_geometry); the frames after the tick are
    bit for bit those of a RayTracer built on the deformed state (the mesh was skipped and the old geometry rendered)."""
    exe = os.path.join(ROOT, "examples", "untracked_mesh")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "stale_differs=1 same_as_fresh=1" in r.stdout


@pytest.mark.gpu
def test_cpp_rebuilt_tree_per_frame_loop(gpu):
    """examples/rebuild_loop.cpp: the OTHER branch of the reference's MeshSequence::buildBvh (src/model/mesh_sequence.cpp:89-96: a new tree per frame with the
    fast binned builder) through the kept C++ classes -- a sequence-like IMesh, RayTracer::rebuildGeometry (pt_upload_static_async), RayTracer::frameTick,
    frames rendered meanwhile.  After the loop the accumulator of four samples equals, bit for bit, that of a RayTracer that only ever saw the last frame
    (the mesh's bounds and root are re-read when the scene is flattened again).  Timed by the example's own output and bench.py's `rebuild_20k`, not here."""
    exe = os.path.join(ROOT, "examples", "rebuild_loop")
    r = subprocess.run([exe, "4", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    fields = dict(kv.split("=") for kv in r.stdout.split() if "=" in kv)
    assert int(fields["triangles"]) == 5120 and fields["same_as_fresh"] == "1"
    assert 0 < float(fields["until_first_new_frame_ms"]) < 100.0, fields
