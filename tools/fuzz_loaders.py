"""Seeded mutation fuzz of the host library's parsers of untrusted bytes -- PNG (own decoder over zlib's inflate), Radiance .hdr, Wavefront
OBJ + MTL, PLY, the reference's .bvh cache file (the reference trusts it: src/model/mesh.cpp:227-263) -- through the C ABI (include/ptamd_host.h).
Every input must come back as a result or as an error return; a crash, a hang or a sanitizer report is a finding.  Meant to run on the
AddressSanitizer + UBSan build (CPU only, build container):

    PTAMD_SANITIZE=1 python -c "import sys; sys.path.insert(0, 'opencl-path-tracer_amd'); from ptamd import build; build.build_sanitized()"
    PTAMD_SANITIZE=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) /usr/lib/x86_64-linux-gnu/libstdc++.so.6" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
        UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python tools/fuzz_loaders.py --n 10000

The input being parsed is always on disk (<tmp>/cur.<ext>) when the parser runs: after a crash it is the reproducer.  tests/test_host_scene.py
runs a few hundred inputs per parser on the regular build."""
import argparse
import ctypes as C
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
from ptamd import host as H, layout as L, scenes  # noqa: E402


# ------------------------------------------------------------------ seeds
def _png_chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp):
    """every PNG filter type in turn, correctly applied (so that the decoder's unfiltering of all five is on the seeds' path)"""
    out, prev = bytearray(), bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = y % 5
        line = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) // 2, _paeth(a, b, c))[ft]
            line[i] = (x - pred) & 0xFF
        out += bytes([ft]) + line
        prev = row
    return bytes(out)


def make_png(w, h, colour_type, depth, interlace, rng):
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour_type]
    bits = channels * depth
    bpp = max(1, bits // 8)

    def image_rows(pw, ph):
        nbytes = (pw * bits + 7) // 8
        return [bytes(rng.integers(0, 256, nbytes, dtype=np.uint8) if colour_type != 3 else rng.integers(0, 4, nbytes, dtype=np.uint8)) for _ in range(ph)]
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            pw, ph = (w - x0 + dx - 1) // dx, (h - y0 + dy - 1) // dy
            if pw > 0 and ph > 0:
                raw += _filter_rows(image_rows(pw, ph), bpp)
    else:
        raw = _filter_rows(image_rows(w, h), bpp)
    out = b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour_type, 0, 0, interlace))
    if colour_type == 3:
        out += _png_chunk(b"PLTE", bytes(rng.integers(0, 256, 3 * 16, dtype=np.uint8)))
        out += _png_chunk(b"tRNS", bytes(rng.integers(0, 256, 7, dtype=np.uint8)))
    elif colour_type in (0, 2) and rng.random() < 0.5:
        out += _png_chunk(b"tRNS", bytes(rng.integers(0, 256, 2 if colour_type == 0 else 6, dtype=np.uint8)))
    out += _png_chunk(b"gAMA", struct.pack(">I", 45455))
    z = zlib.compress(raw, 6)
    cut = len(z) // 2
    out += _png_chunk(b"IDAT", z[:cut]) + _png_chunk(b"IDAT", z[cut:]) + _png_chunk(b"IEND", b"")
    return out


def make_hdr(w, h, rng, rle=True):
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {h} +X {w}\n".encode()
    body = bytearray()
    for _ in range(h):
        px = rng.integers(0, 256, (w, 4), dtype=np.uint8)
        px[: w // 2] = px[0]  # runs, so that the RLE seeds hold run codes as well as literal codes
        if rle and 8 <= w < 32768:
            body += bytes([2, 2, w >> 8, w & 255])
            for ch in range(4):
                col, x = px[:, ch], 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and col[x + run] == col[x]:
                        run += 1
                    if run >= 3:
                        body += bytes([128 + run, int(col[x])])
                        x += run
                    else:
                        n = min(w - x, 1 + int(rng.integers(0, 8)))
                        body += bytes([n]) + bytes(col[x:x + n])
                        x += n
        else:
            body += px.tobytes()
    return head + bytes(body)


def make_obj(rng, mtl_name="m.mtl"):
    v, f = scenes.icosphere(1)
    lines = [f"mtllib {mtl_name}", "o blob"]
    lines += [f"v {x:.6f} {y:.6f} {z:.6f}" for x, y, z in v]
    lines += [f"vn {x:.4f} {y:.4f} {z:.4f}" for x, y, z in v]
    lines += [f"vt {abs(x):.4f} {abs(y):.4f}" for x, y, _ in v]
    lines.append("usemtl red")
    for k, (a, b, c) in enumerate(f + 1):
        if k == len(f) // 2:
            lines.append("usemtl glass")
        form = k % 4
        lines.append({0: f"f {a} {b} {c}", 1: f"f {a}/{a} {b}/{b} {c}/{c}", 2: f"f {a}//{a} {b}//{b} {c}//{c}",
                      3: f"f {a - len(v) - 1}/{a}/{a} {b - len(v) - 1}/{b}/{b} {c - len(v) - 1}/{c}/{c}"}[form])  # (negative: relative to the end)
    lines.append("f 1 2 3 4 5")  # a polygon: fanned
    mtl = "\n".join(["newmtl red", "Kd 0.8 0.1 0.1", "Ks 0.0 0.0 0.0", "Ns 10", "d 1.0", "illum 2", "", "newmtl glass", "Kd 1 1 1", "Ks 1 1 1", "Ns 900",
                     "Ni 1.5", "d 0.2", "Tf 0.9 0.9 1.0", "illum 7", "map_Kd tex.png", "", "newmtl lamp", "Ke 10 10 8", "Kd 0 0 0"]) + "\n"
    return ("\n".join(lines) + "\n").encode(), mtl.encode()


def make_ply(rng, binary):
    v, f = scenes.icosphere(1)
    v = v.astype(np.float32)
    head = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0", "comment fuzz seed", f"element vertex {len(v)}", "property float x",
            "property float y", "property float z", "property float nx", "property float ny", "property float nz", f"element face {len(f)}",
            "property list uchar int vertex_indices", "end_header"]
    out = ("\n".join(head) + "\n").encode()
    if binary:
        out += np.concatenate([v, v], 1).astype("<f4").tobytes()
        for a, b, c in f:
            out += struct.pack("<Biii", 3, int(a), int(b), int(c))
    else:
        out += "".join(f"{x:.6f} {y:.6f} {z:.6f} {x:.4f} {y:.4f} {z:.4f}\n" for x, y, z in v).encode()
        out += "".join(f"3 {a} {b} {c}\n" for a, b, c in f).encode()
    return out


# ------------------------------------------------------------------ mutation
INTERESTING32 = [0, 1, 2, 0x7F, 0x80, 0xFF, 0x100, 0x7FFF, 0x8000, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, 1 << 24, 1 << 27, 1 << 30]
TOKENS = [b"-1", b"0", b"99999999999999999999", b"4294967296", b"-2147483649", b"nan", b"inf", b"1e999", b"", b"/", b"//", b"1/", b"/1/", b"1//", b"3 0 0", b"255",
          b"element vertex 4000000000", b"element face -1", b"property list uchar int vertex_indices", b"property list int int vertex_indices", b"format binary_big_endian 1.0",
          b"usemtl nothere", b"mtllib /nonexistent/x.mtl", b"newmtl", b"map_Kd /dev/null", b"-Y 1000000 +X 1000000", b"-Y -3 +X 7", b"+X 4 -Y 4", b"f", b"v", b"vt", b"\x00", b"\r\n"]


def mutate(data, rng, text):
    d = bytearray(data)
    for _ in range(int(rng.integers(1, 5))):
        if not d:
            d = bytearray(b"\x00")
        kind = int(rng.integers(0, 9 if text else 7))
        pos = int(rng.integers(0, len(d)))
        if kind == 0:
            d[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d[pos] = int(rng.choice([0, 1, 0x7F, 0x80, 0xFF, 0x0A, 0x20, 0x2F, 0x2D]))
        elif kind == 2:
            v = int(rng.choice(INTERESTING32))
            d[pos:pos + 4] = struct.pack("<I" if rng.random() < 0.5 else ">I", v)
        elif kind == 3:
            d = d[:pos]  # truncation
        elif kind == 4:
            n = int(rng.integers(1, 64))
            d[pos:pos] = d[pos:pos + n]  # duplicated run
        elif kind == 5:
            del d[pos:pos + int(rng.integers(1, 64))]
        elif kind == 6:
            d[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 16)), dtype=np.uint8))
        elif kind == 7:  # a token replaced (text formats)
            end = pos
            while end < len(d) and d[end] not in b" \n\r\t/":
                end += 1
            d[pos:end] = TOKENS[int(rng.integers(0, len(TOKENS)))]
        else:  # a whole line replaced or doubled
            a = d.rfind(b"\n", 0, pos) + 1
            b = d.find(b"\n", pos)
            b = len(d) if b < 0 else b + 1
            d[a:b] = TOKENS[int(rng.integers(0, len(TOKENS)))] + b"\n" if rng.random() < 0.5 else d[a:b] * 2
    return bytes(d)


def fix_png_crcs(data):
    """chunk CRCs recomputed (as far as the chunk structure can still be followed), so that a mutation gets past the CRC check and into the decoder"""
    if not data.startswith(b"\x89PNG\r\n\x1a\n"):
        return data
    out, pos = bytearray(data[:8]), 8
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        if pos + 12 + n > len(data):
            break
        body = data[pos + 4:pos + 8 + n]
        out += data[pos:pos + 4] + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
        pos += 12 + n
    return bytes(out) + data[pos:]


def mutate_png_scanlines(data, rng):
    """mutations BEHIND the compression: the filtered scanlines (filter-type bytes included) are inflated, mutated in place -- the length
    stays what the header implies -- and deflated again, so that the unfiltering, the bit-depth expansion, the palette / tRNS handling
    and the Adam7 scatter see hostile bytes, not only the inflater"""
    if not data.startswith(b"\x89PNG\r\n\x1a\n"):
        return data
    chunks, pos = [], 8
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        if pos + 12 + n > len(data):
            break
        chunks.append((data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]))
        pos += 12 + n
    try:
        raw = bytearray(zlib.decompress(b"".join(d for k, d in chunks if k == b"IDAT")))
    except zlib.error:
        return data
    for _ in range(int(rng.integers(1, 12))):
        if raw:
            raw[int(rng.integers(0, len(raw)))] = int(rng.choice([0, 1, 2, 3, 4, 5, 0x7F, 0x80, 0xFF, int(rng.integers(0, 256))]))
    out, done = bytearray(data[:8]), False
    for k, d in chunks:
        if k == b"IDAT":
            if done:
                continue
            d, done = zlib.compress(bytes(raw), 1), True
        elif k in (b"PLTE", b"tRNS") and rng.random() < 0.3:
            d = d[:int(rng.integers(0, len(d) + 1))] if rng.random() < 0.5 else d + bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        out += _png_chunk(k, d)
    return bytes(out)


# ------------------------------------------------------------------ parsers under test
class Targets:
    def __init__(self, tmp):
        self.tmp, self.lib = tmp, H.lib()
        self.mat = np.ascontiguousarray(np.asarray(L.material_diffuse((1, 1, 1)), dtype=L.MATERIAL).reshape(1))
        v, f = scenes.icosphere(1)
        self.v, self.f = v.astype(np.float32), f.astype(np.uint32)
        for p, b in (("tex.png", make_png(4, 4, 2, 8, 0, np.random.default_rng(0))),):
            open(os.path.join(tmp, p), "wb").write(b)

    def _write(self, name, data):
        path = os.path.join(self.tmp, name)
        with open(path, "wb") as fh:
            fh.write(data)
        return path

    def png(self, data):
        path = self._write("cur.png", data)
        w, h = C.c_uint32(0), C.c_uint32(0)
        if self.lib.pth_image_png_info(path.encode(), C.byref(w), C.byref(h)) != 0:
            return False
        if w.value == 0 or h.value == 0 or w.value * h.value > (1 << 22):
            return False  # (a valid header of an absurd size: the caller's buffer, not the parser, is the limit)
        out = np.zeros(w.value * h.value * 4, np.uint8)
        ok = self.lib.pth_image_load_png_rgba8(path.encode(), out.ctypes.data) == 0
        layer = np.zeros(8 * 8 * 4, np.float32)
        ok2 = self.lib.pth_image_load_material_png(path.encode(), 8, 8, 0, layer.ctypes.data) == 0
        return ok and ok2

    def hdr(self, data):
        path = self._write("cur.hdr", data)
        w, h = C.c_uint32(0), C.c_uint32(0)
        if self.lib.pth_image_hdr_info(path.encode(), C.byref(w), C.byref(h)) != 0:
            return False
        out = np.zeros(16 * 8 * 4, np.float32)
        return self.lib.pth_image_load_hdr(path.encode(), 16, 8, C.c_float(1.0), out.ctypes.data) == 0

    def _mesh_ok(self, handle):
        if not handle:
            return False
        m = H.Mesh(None, None, None, _handle=handle)
        st = m.stats()
        assert st["children_inside_parents"] and st["triangles_inside_leaves"] and st["all_triangles_referenced"], "a mesh was accepted with a broken BVH"
        return True

    def obj(self, data, mtl=None):
        path = self._write("cur.obj", data)
        if mtl is not None:
            self._write("m.mtl", mtl)
        reg = H.TextureFiles()
        return self._mesh_ok(self.lib.pth_mesh_from_obj_textured(path.encode(), None, None, None, None, H.BVH_BINNED_SAH, None, reg._h))

    def mtl(self, data):
        return self.obj(self.obj_seed, mtl=data)

    def ply(self, data):
        path = self._write("cur.ply", data)
        return self._mesh_ok(self.lib.pth_mesh_from_ply(path.encode(), self.mat.ctypes.data, H.BVH_BINNED_FAST))

    def bvh(self, data):
        path = self._write("cur.bvh", data)
        m = H.Mesh(self.v, self.f, [L.material_diffuse((1, 1, 1))], builder=H.BVH_BINNED_SAH, bvh_cache=path)  # a rejected file is rebuilt (and rewritten)
        st = m.stats()
        assert st["children_inside_parents"] and st["triangles_inside_leaves"] and st["all_triangles_referenced"], "a cache file was accepted with a broken BVH"
        return m.bvh_from_cache


def seeds(tmp, with_reference=True):
    rng = np.random.default_rng(1)
    s = {"png": [], "hdr": [], "obj": [], "mtl": [], "ply": [], "bvh": []}
    for ct, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for il in (0, 1):
                s["png"].append(make_png(int(rng.integers(1, 20)), int(rng.integers(1, 20)), ct, depth, il, rng))
    s["hdr"] = [make_hdr(16, 6, rng, True), make_hdr(9, 3, rng, True), make_hdr(5, 4, rng, False), make_hdr(40, 2, rng, True)]
    obj, mtl = make_obj(rng)
    s["obj"], s["mtl"] = [obj], [mtl]
    s["ply"] = [make_ply(rng, False), make_ply(rng, True)]
    v, f = scenes.icosphere(1)
    for builder in (H.BVH_BINNED_SAH, H.BVH_SPATIAL_SPLIT):
        p = os.path.join(tmp, "seed.bvh")
        if os.path.exists(p):
            os.remove(p)
        H.Mesh(v.astype(np.float32), f.astype(np.uint32), [L.material_diffuse((1, 1, 1))], builder=H.BVH_BINNED_SAH, bvh_cache=p)
        s["bvh"].append(open(p, "rb").read())
    ref = "/root/reference/assets"
    if with_reference and os.path.isdir(ref):  # build container only: a few small real files as extra seeds (read, not copied)
        small = []
        for dp, _, fs in os.walk(ref):
            for fn in fs:
                p = os.path.join(dp, fn)
                if fn.lower().endswith(".png") and os.path.getsize(p) < 40000:
                    small.append(p)
        for p in sorted(small)[:6]:
            s["png"].append(open(p, "rb").read())
    return s


def run(n, seed=1, only=None, tmp=None, verbose=True, with_reference=True):
    tmp = tmp or tempfile.mkdtemp(prefix="ptamd_fuzz_")
    t = Targets(tmp)
    sd = seeds(tmp, with_reference)
    t.obj_seed = sd["obj"][0]
    summary = {}
    for name in ("png", "hdr", "obj", "mtl", "ply", "bvh"):
        if only and name not in only:
            continue
        rng = np.random.default_rng(seed * 1000 + len(name) + ord(name[0]))
        fn = getattr(t, name)
        for sdata in sd[name]:  # the seeds themselves must parse
            if name == "mtl":
                assert t.obj(t.obj_seed, mtl=sdata), "seed mtl rejected"
            else:
                assert fn(sdata), f"seed {name} rejected"
        ok = err = 0
        t0 = time.perf_counter()
        for k in range(n):
            base = sd[name][int(rng.integers(0, len(sd[name])))]
            if name == "png" and rng.random() < 0.4:
                data = mutate_png_scanlines(base, rng)
            else:
                data = mutate(base, rng, text=name in ("obj", "mtl") or (name in ("ply", "hdr") and rng.random() < 0.5))
                if name == "png" and rng.random() < 0.6:
                    data = fix_png_crcs(data)
            if fn(data):
                ok += 1
            else:
                err += 1
        summary[name] = {"inputs": n, "accepted": ok, "rejected": err, "seconds": round(time.perf_counter() - t0, 1)}
        if verbose:
            print(f"{name:4s}: {n} mutated inputs from {len(sd[name])} seeds: {ok} parsed, {err} rejected with an error, 0 crashes ({summary[name]['seconds']} s)", flush=True)
    return summary


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    print(f"host library: {H.HOST_LIB_PATH}")
    run(a.n, a.seed, a.only)
