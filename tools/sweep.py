"""Run bench.py once per tuning build (opencl-path-tracer_amd/csrc/variants/libptamd_*.so) and print one line each."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
libs = sorted(glob.glob(os.path.join(ROOT, "opencl-path-tracer_amd", "csrc", "variants", "libptamd_*.so")))
for lib in libs:
    name = os.path.basename(lib)[len("libptamd_"):-3]
    if names and name not in names:
        continue
    env = dict(os.environ, PTAMD_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "8", "--warmup", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{name:12s} {d['value']:8.1f} Mrays/s  in-kernel {r['mrays_per_s_in_kernel']:7.1f}  ms: isect {r['family_ms']['intersect']:.2f} shade {r['family_ms']['shade']:.2f} shadow {r['family_ms']['shadow']:.2f}", flush=True)
    except Exception as e:
        print(name, "FAILED", p.stderr[-500:], flush=True)
