"""In-tree builds: libptamd_host.so (g++) and libptamd.so (hipcc, gfx950 code objects, no JIT)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(_HERE, ".."))
HOST_DIR = os.path.join(ROOT, "host")
CSRC_DIR = os.path.join(ROOT, "csrc")
HOST_SOURCES = ["bvh_build.cpp", "mesh.cpp", "scene.cpp", "image.cpp", "capi.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_files(d, exts):
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts)]


def build_host(force=False):
    out = os.path.join(HOST_DIR, "libptamd_host.so")
    deps = _all_files(HOST_DIR, (".cpp", ".h")) + _all_files(os.path.join(ROOT, "..", "include"), (".h",))
    if force or _newer(out, deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-fPIC", "-shared", "-pthread"] + HOST_SOURCES + ["-o", out, "-lz"],  # zlib: PNG inflate
                       cwd=HOST_DIR, check=True)
    return out


# PTAMD_SANITIZE=1: the host library (the parsers of untrusted bytes: PNG / Radiance / OBJ / MTL / PLY / .bvh) and the oracle built with
# AddressSanitizer + UndefinedBehaviorSanitizer next to the regular files (lib*_san.so); ptamd.host and oracle/orclib.py load those when
# the variable is set.  CPU build only (the GPU pool refuses sanitizer runs).  Run python with the sanitizer runtime preloaded:
#   PTAMD_SANITIZE=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -m 'not gpu'
# (libstdc++ too: python does not link it, and the sanitizer's __cxa_throw interceptor aborts at the first C++ exception if it was not there at start-up)
SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def sanitize():
    return os.environ.get("PTAMD_SANITIZE", "") not in ("", "0")


def build_sanitized(force=False):
    out = os.path.join(HOST_DIR, "libptamd_host_san.so")
    deps = _all_files(HOST_DIR, (".cpp", ".h")) + _all_files(os.path.join(ROOT, "..", "include"), (".h",))
    if force or _newer(out, deps):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-fPIC", "-shared", "-pthread"] + SANITIZE_FLAGS + HOST_SOURCES + ["-o", out, "-lz"], cwd=HOST_DIR, check=True)
    odir = os.path.abspath(os.path.join(ROOT, "..", "oracle"))
    oout = os.path.join(odir, "liboracle_san.so")
    osrc = [f for f in os.listdir(odir) if f.endswith(".cpp")]
    if force or _newer(oout, _all_files(odir, (".cpp", ".h"))):
        subprocess.run(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"] + SANITIZE_FLAGS
                       + osrc + ["-o", oout], cwd=odir, check=True)
    return out, oout


# Division and square root as v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the correctly rounded expansions (~10
# instructions each): a quarter of k_shade's instructions were those expansions.  The reference builds its kernels
# with -cl-fast-relaxed-math (raytracer.cpp:819); every parity test passes either way.
# -fno-slp-vectorize (device code only): the vectoriser packs pairs of FP32 multiplies and multiply-adds into v_pk_* instructions, which issue at half
# rate and compete with the conversions, compares and selects of the traversal kernels; plain FP32 arithmetic next to one of those is nearly free
# (profiles/round5/r5r_valu_issue_pairs.md).  Benchmark scene: 11 040 -> 11 660 Mrays/s, k_shade 85 -> 75 VGPRs.
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
                "-Xarch_device", "-fno-slp-vectorize",
                "-Xarch_host", "-msse4.1"]  # (host side: floorf / ceilf of the node quantiser inline -- it is most of a conversion's packing pass)


def csrc_fingerprint(csrc_dir=None):
    """sha256 over the device library's sources (csrc/*.hip, *.h: names and bytes, sorted) and its compiler flags: what a set of hardware counters
    (profiles/roundN/traffic*.json) was taken on, and what bench.py holds them against before it quotes them (roofline.traffic_stale)."""
    import hashlib
    d = csrc_dir or CSRC_DIR
    h = hashlib.sha256(" ".join(DEVICE_FLAGS).encode())
    for name in sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h"))):
        h.update(name.encode() + b"\0")
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_device(force=False):
    out = os.path.join(CSRC_DIR, "libptamd.so")
    deps = _all_files(CSRC_DIR, (".hip", ".h")) + _all_files(os.path.join(ROOT, "..", "include"), (".h",))
    if force or _newer(out, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.run([hipcc] + DEVICE_FLAGS + ["ptamd.hip", "-o", out], cwd=CSRC_DIR, check=True)
    return out


def build_raytracer(force=False):
    """libptamd_raytracer.so: the C++ RayTracer class (host/raytracer.cpp) on top of both libraries,
    and the examples/render_cornell demo that drives it."""
    out = os.path.join(HOST_DIR, "libptamd_raytracer.so")
    deps = _all_files(HOST_DIR, (".cpp", ".h")) + _all_files(os.path.join(ROOT, "..", "include"), (".h",))  # pt_stats etc. are part of its ABI
    if force or _newer(out, deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-fPIC", "-shared", "raytracer.cpp", "-o", out, "-L.", "-lptamd_host",
                        "-L../csrc", "-lptamd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../csrc"], cwd=HOST_DIR, check=True)
    ex_dir = os.path.abspath(os.path.join(ROOT, "..", "examples"))
    exe = os.path.join(ex_dir, "render_cornell")
    if force or _newer(exe, [os.path.join(ex_dir, "render_cornell.cpp"), out]):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "render_cornell.cpp", "-o", "render_cornell", "-L" + HOST_DIR, "-lptamd_raytracer",
                        "-lptamd_host", "-L" + CSRC_DIR, "-lptamd", "-Wl,-rpath," + HOST_DIR, "-Wl,-rpath," + CSRC_DIR], cwd=ex_dir, check=True)
    exe2 = os.path.join(ex_dir, "deform_loop")
    if force or _newer(exe2, [os.path.join(ex_dir, "deform_loop.cpp"), out]):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "deform_loop.cpp", "-o", "deform_loop", "-L" + HOST_DIR, "-lptamd_raytracer",
                        "-lptamd_host", "-L" + CSRC_DIR, "-lptamd", "-Wl,-rpath," + HOST_DIR, "-Wl,-rpath," + CSRC_DIR], cwd=ex_dir, check=True)
    exe3 = os.path.join(ex_dir, "untracked_mesh")
    if force or _newer(exe3, [os.path.join(ex_dir, "untracked_mesh.cpp"), out]):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "untracked_mesh.cpp", "-o", "untracked_mesh", "-L" + HOST_DIR, "-lptamd_raytracer",
                        "-lptamd_host", "-L" + CSRC_DIR, "-lptamd", "-Wl,-rpath," + HOST_DIR, "-Wl,-rpath," + CSRC_DIR], cwd=ex_dir, check=True)
    exe4 = os.path.join(ex_dir, "rebuild_loop")
    if force or _newer(exe4, [os.path.join(ex_dir, "rebuild_loop.cpp"), out]):
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "rebuild_loop.cpp", "-o", "rebuild_loop", "-L" + HOST_DIR, "-lptamd_raytracer",
                        "-lptamd_host", "-L" + CSRC_DIR, "-lptamd", "-Wl,-rpath," + HOST_DIR, "-Wl,-rpath," + CSRC_DIR], cwd=ex_dir, check=True)
    return out, exe


def build_all(force=False):
    if sanitize():
        build_sanitized(force)
    host, dev = build_host(force), build_device(force)
    build_raytracer(force)
    return host, dev
