"""ctypes binding of the host-side scene library (include/ptamd_host.h): meshes + BVH builders,
scene graph with instancing, flattening into the device arrays, camera derivation."""
import ctypes as C
import os
import numpy as np
from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_SANITIZE=1: the AddressSanitizer / UBSan build of the same sources (ptamd/build.py; python needs the sanitizer runtime preloaded)
HOST_LIB_PATH = os.environ.get("PTAMD_HOST_LIB") or os.path.join(  # PTAMD_HOST_LIB: another build of the host library (A / B measurements)
    _HERE, "..", "host", "libptamd_host_san.so" if os.environ.get("PTAMD_SANITIZE", "") not in ("", "0") else "libptamd_host.so")

BVH_BINNED_SAH, BVH_BINNED_FAST, BVH_SPATIAL_SPLIT = 0, 1, 2


class MeshStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_vertices", "num_input_triangles", "num_triangle_refs", "num_nodes", "num_leaves", "max_depth",
        "max_leaf_size", "children_inside_parents", "triangles_inside_leaves", "all_triangles_referenced",
        "reachable_triangle_refs", "reachable_nodes")]


class SceneCounts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_vertices", "num_triangles", "num_materials", "num_sub_nodes", "num_lights", "num_top_nodes",
        "top_root", "num_instances")]


class CameraParams(C.Structure):
    _fields_ = [("location", C.c_float * 3), ("orientation_wxyz", C.c_float * 4), ("horizontal_fov_deg", C.c_float),
                ("aspect_ratio", C.c_float), ("focal_distance", C.c_float), ("focal_length_mm", C.c_float),
                ("aperture_fstops", C.c_float), ("shutter_time", C.c_float), ("iso", C.c_float), ("thin_lens", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} missing: run __graft_entry__.build() first")
        _lib = C.CDLL(HOST_LIB_PATH)
        _lib.pth_last_error.restype = C.c_char_p
        _lib.pth_mesh_create.restype = C.c_void_p
        _lib.pth_mesh_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        _lib.pth_mesh_create_cached.restype = C.c_void_p
        _lib.pth_mesh_create_cached.argtypes = _lib.pth_mesh_create.argtypes + [C.c_char_p]
        _lib.pth_mesh_store_bvh.argtypes = [C.c_void_p, C.c_char_p]
        _lib.pth_mesh_bvh_from_cache.argtypes = [C.c_void_p]
        _lib.pth_mesh_from_ply.restype = C.c_void_p
        _lib.pth_mesh_from_ply.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
        _lib.pth_mesh_from_obj.restype = C.c_void_p
        _lib.pth_mesh_from_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        _lib.pth_mesh_from_obj_textured.restype = C.c_void_p
        _lib.pth_mesh_from_obj_textured.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
        _lib.pth_texture_files_create.restype = C.c_void_p
        _lib.pth_texture_files_destroy.argtypes = [C.c_void_p]
        _lib.pth_texture_files_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float]
        _lib.pth_texture_files_count.argtypes = [C.c_void_p]
        _lib.pth_texture_files_path.restype = C.c_char_p
        _lib.pth_texture_files_path.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        _lib.pth_mesh_copy_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        _lib.pth_mesh_destroy.argtypes = [C.c_void_p]
        _lib.pth_mesh_info.argtypes = [C.c_void_p, C.POINTER(MeshStats)]
        _lib.pth_mesh_copy_bvh.argtypes = [C.c_void_p] * 4
        _lib.pth_scene_create.restype = C.c_void_p
        _lib.pth_scene_destroy.argtypes = [C.c_void_p]
        _lib.pth_scene_add_node.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.pth_scene_set_transform.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.pth_scene_flatten.argtypes = [C.c_void_p, C.POINTER(SceneCounts)]
        _lib.pth_scene_flatten_dynamic.argtypes = [C.c_void_p, C.POINTER(SceneCounts)]
        _lib.pth_scene_copy.argtypes = [C.c_void_p] * 7
        _lib.pth_scene_flatten_dynamic_only.argtypes = [C.c_void_p, C.POINTER(SceneCounts)]
        _lib.pth_scene_mesh_offsets.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib.pth_mesh_vertices.restype = C.c_void_p
        _lib.pth_mesh_vertices.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        _lib.pth_camera_data.argtypes = [C.POINTER(CameraParams), C.c_void_p]
        _lib.pth_image_hdr_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib.pth_image_load_hdr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        _lib.pth_image_png_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib.pth_image_load_png_rgba8.argtypes = [C.c_char_p, C.c_void_p]
        _lib.pth_image_load_material_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        _lib.pth_mesh_refit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.pth_image_load_material_png_bgra8.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    return _lib


def _err(what):
    raise RuntimeError(f"{what}: {lib().pth_last_error().decode()}")


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Mesh:
    def __init__(self, positions, indices, materials, material_index=None, normals=None, tex_coords=None,
                 builder=BVH_SPATIAL_SPLIT, _handle=None, bvh_cache=None):
        if _handle is not None:
            self._h = _handle
        else:
            pos = _f32(positions, (-1, 3))
            nrm = _f32(normals, (-1, 3))
            uv = _f32(tex_coords, (-1, 2))
            idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
            mi = None if material_index is None else np.ascontiguousarray(material_index, dtype=np.uint32)
            mats = np.ascontiguousarray(np.atleast_1d(np.asarray(materials, dtype=L.MATERIAL)))
            self._h = lib().pth_mesh_create_cached(_ptr(pos), _ptr(nrm), _ptr(uv), len(pos), _ptr(idx), _ptr(mi), len(idx),
                                                   _ptr(mats), len(mats), builder,
                                                   None if bvh_cache is None else str(bvh_cache).encode())
            if not self._h:
                _err("pth_mesh_create")
        self.builder = builder

    @staticmethod
    def from_ply(path, material, builder=BVH_SPATIAL_SPLIT):
        mat = np.ascontiguousarray(np.asarray(material, dtype=L.MATERIAL).reshape(1))
        h = lib().pth_mesh_from_ply(str(path).encode(), _ptr(mat), builder)
        if not h:
            _err("pth_mesh_from_ply")
        return Mesh(None, None, None, builder=builder, _handle=h)

    @staticmethod
    def from_obj(path, material=None, location=None, orientation_wxyz=None, scale=None, builder=BVH_SPATIAL_SPLIT, bvh_cache=None, textures=None):
        """Wavefront OBJ (+ MTL) import with the reference's material mapping (src/model/mesh.cpp:36-200); `textures`: a
        TextureFiles registry that receives the map_Kd files (their index becomes the material's tex_id)."""
        mat = None if material is None else np.ascontiguousarray(np.asarray(material, dtype=L.MATERIAL).reshape(1))
        loc, rot, scl = _f32(location, (3,)), _f32(orientation_wxyz, (4,)), _f32(scale, (3,))
        h = lib().pth_mesh_from_obj_textured(str(path).encode(), _ptr(mat), _ptr(loc), _ptr(rot), _ptr(scl), builder,
                                             None if bvh_cache is None else str(bvh_cache).encode(), None if textures is None else textures._h)
        if not h:
            _err("pth_mesh_from_obj")
        return Mesh(None, None, None, builder=builder, _handle=h)

    def refit(self, positions, normals=None):
        """A deformed frame of the same mesh: new positions (and normals; None = regenerated smooth) for the same vertices
        and triangles.  The BVH keeps topology and leaf order, its boxes are refitted (reference refitBVH,
        src/bvh/refit_bvh.cpp:6-34, as MeshSequence::buildBvh uses it, src/model/mesh_sequence.cpp:81-97)."""
        pos = _f32(positions, (-1, 3))
        if getattr(self, "_num_vertices", None) is None:  # (stats() walks the whole tree -- the BvhTester invariants --: 1.8 ms per call at 82 k triangles)
            self._num_vertices = self.stats()["num_vertices"]
        if len(pos) != self._num_vertices:
            raise RuntimeError("Mesh.refit: the vertex count must stay the same")
        nrm = _f32(normals, (-1, 3))
        if lib().pth_mesh_refit(self._h, _ptr(pos), _ptr(nrm)):
            _err("pth_mesh_refit")

    def vertices_view(self):
        """The mesh's vertex records IN PLACE (numpy view of the host library's array, no copy): what pt_refit_vertices takes after a refit."""
        n = C.c_uint32(0)
        ptr = lib().pth_mesh_vertices(self._h, C.byref(n))
        buf = (C.c_char * (n.value * L.VERTEX.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=L.VERTEX, count=n.value)

    def geometry(self):
        """(vertices, materials) in the reference's device layouts."""
        n = C.c_uint32(0)
        lib().pth_mesh_copy_geometry(self._h, None, None, C.byref(n))
        verts = np.zeros(self.stats()["num_vertices"], L.VERTEX)
        mats = np.zeros(n.value, L.MATERIAL)
        if lib().pth_mesh_copy_geometry(self._h, _ptr(verts), _ptr(mats), C.byref(n)):
            _err("pth_mesh_copy_geometry")
        return verts, mats

    def store_bvh(self, path):
        """Mesh::storeBvh (reference src/model/mesh.cpp:202-225): the .bvh cache file."""
        if lib().pth_mesh_store_bvh(self._h, str(path).encode()):
            _err("pth_mesh_store_bvh")

    @property
    def bvh_from_cache(self):
        return bool(lib().pth_mesh_bvh_from_cache(self._h))

    def stats(self):
        s = MeshStats()
        if lib().pth_mesh_info(self._h, C.byref(s)):
            _err("pth_mesh_info")
        return {n: getattr(s, n) for n, _ in s._fields_}

    def bvh(self):
        s = self.stats()
        nodes = np.zeros(s["num_nodes"], L.SUB_BVH_NODE)
        tris = np.zeros(s["num_triangle_refs"], L.TRIANGLE)
        orig = np.zeros(s["num_triangle_refs"], np.uint32)
        if lib().pth_mesh_copy_bvh(self._h, _ptr(nodes), _ptr(tris), _ptr(orig)):
            _err("pth_mesh_copy_bvh")
        return nodes, tris, orig

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pth_mesh_destroy(self._h)
            self._h = None


class FlatScene:
    """The flattened device arrays (reference layouts) + counts."""

    def __init__(self, vertices, triangles, materials, sub_nodes, lights, top_nodes, top_root, num_instances):
        self.vertices, self.triangles, self.materials = vertices, triangles, materials
        self.sub_nodes, self.lights, self.top_nodes = sub_nodes, lights, top_nodes
        self.top_root, self.num_instances = int(top_root), int(num_instances)

    @property
    def instanced_triangles(self):
        """Triangle references reachable through all instances (the '~1M-tri' figure)."""
        total = 0
        leaf = self.top_nodes[self.top_nodes["isLeaf"] != 0]
        cache = {}
        for root in leaf["a"]:
            root = int(root)
            if root not in cache:
                n, stack = 0, [root]
                while stack:
                    nd = self.sub_nodes[stack.pop()]
                    if nd["count"]:
                        n += int(nd["count"])
                    else:
                        stack += [int(nd["left"]), int(nd["left"]) + 1]
                cache[root] = n
            total += cache[root]
        return total


class Scene:
    def __init__(self):
        self._h = lib().pth_scene_create()
        self._meshes = []  # keep alive

    def add_node(self, mesh, location=(0, 0, 0), orientation_wxyz=(1, 0, 0, 0), scale=(1, 1, 1), parent=-1):
        loc, q, s = _f32(location), _f32(orientation_wxyz), _f32(scale)
        if s.size == 1:
            s = np.repeat(s, 3)
        nid = lib().pth_scene_add_node(self._h, mesh._h, _ptr(loc), _ptr(q), _ptr(s), parent)
        if nid < 0:
            _err("pth_scene_add_node")
        self._meshes.append(mesh)
        return nid

    def set_transform(self, node, location=None, orientation_wxyz=None, scale=None):
        loc, q, s = _f32(location), _f32(orientation_wxyz), _f32(scale)
        if lib().pth_scene_set_transform(self._h, node, _ptr(loc), _ptr(q), _ptr(s)):
            _err("pth_scene_set_transform")

    def flatten(self):
        c = SceneCounts()
        if lib().pth_scene_flatten(self._h, C.byref(c)):
            _err("pth_scene_flatten")
        v = np.zeros(c.num_vertices, L.VERTEX)
        t = np.zeros(c.num_triangles, L.TRIANGLE)
        m = np.zeros(c.num_materials, L.MATERIAL)
        n = np.zeros(c.num_sub_nodes, L.SUB_BVH_NODE)
        l = np.zeros(c.num_lights, L.EMISSIVE_TRIANGLE)
        top = np.zeros(c.num_top_nodes, L.TOP_BVH_NODE)
        if lib().pth_scene_copy(self._h, _ptr(v), _ptr(t), _ptr(m), _ptr(n), _ptr(l), _ptr(top)):
            _err("pth_scene_copy")
        return FlatScene(v, t, m, n, l, top, c.top_root, c.num_instances)

    def flatten_dynamic(self, flat):
        """The per-tick half of flatten(): lights and top-level BVH from the scene graph as it stands (what RayTracer::frameTick
        does on the host); the static arrays of `flat` are shared, not copied.  Returns (new FlatScene, seconds inside the host library)."""
        import time
        c = SceneCounts()
        t0 = time.perf_counter()
        if lib().pth_scene_flatten_dynamic(self._h, C.byref(c)):
            _err("pth_scene_flatten_dynamic")
        dt = time.perf_counter() - t0
        l = np.zeros(c.num_lights, L.EMISSIVE_TRIANGLE)
        top = np.zeros(c.num_top_nodes, L.TOP_BVH_NODE)
        if lib().pth_scene_copy(self._h, None, None, None, None, _ptr(l), _ptr(top)):
            _err("pth_scene_copy")
        return FlatScene(flat.vertices, flat.triangles, flat.materials, flat.sub_nodes, l, top, c.top_root, c.num_instances), dt

    def flatten_dynamic_only(self):
        """Lights and top-level BVH of the scene as it stands, nothing else (a FlatScene whose static arrays are None): for callers that hand
        deformed meshes to the device themselves (Context.refit_vertices).  Returns (FlatScene, seconds inside the host library)."""
        import time
        c = SceneCounts()
        t0 = time.perf_counter()
        if lib().pth_scene_flatten_dynamic_only(self._h, C.byref(c)):
            _err("pth_scene_flatten_dynamic_only")
        dt = time.perf_counter() - t0
        l = np.zeros(c.num_lights, L.EMISSIVE_TRIANGLE)
        top = np.zeros(c.num_top_nodes, L.TOP_BVH_NODE)
        if lib().pth_scene_copy(self._h, None, None, None, None, _ptr(l), _ptr(top)):
            _err("pth_scene_copy")
        return FlatScene(None, None, None, None, l, top, c.top_root, c.num_instances), dt

    def mesh_offsets(self, mesh):
        """(first vertex, first sub-BVH node) of `mesh` in the arrays of the last flatten()."""
        v, n = C.c_uint32(0), C.c_uint32(0)
        if lib().pth_scene_mesh_offsets(self._h, mesh._h, C.byref(v), C.byref(n)):
            raise RuntimeError("mesh_offsets: the mesh is not part of the flattened scene")
        return v.value, n.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.pth_scene_destroy(self._h)
            self._h = None


def camera_data(location, orientation_wxyz, horizontal_fov_deg, aspect_ratio, focal_distance=1.0, thin_lens=False,
                focal_length_mm=0.0, aperture_fstops=0.0, shutter_time=0.0, iso=0.0):
    p = CameraParams()
    p.location[:] = [float(x) for x in location]
    p.orientation_wxyz[:] = [float(x) for x in orientation_wxyz]
    p.horizontal_fov_deg, p.aspect_ratio, p.focal_distance = horizontal_fov_deg, aspect_ratio, focal_distance
    p.focal_length_mm, p.aperture_fstops, p.shutter_time, p.iso = focal_length_mm, aperture_fstops, shutter_time, iso
    p.thin_lens = int(bool(thin_lens))
    out = np.zeros((), L.CAMERA)
    if lib().pth_camera_data(C.byref(p), out.ctypes.data_as(C.c_void_p)):
        _err("pth_camera_data")
    return out


def look_at_quat(eye, target, up=(0, 1, 0)):
    """Orientation (w,x,y,z) whose local +z points from eye to target and local +y is 'up'-ish.
    The reference camera looks down +z with screen v pointing down (src/camera.cpp:52-53)."""
    eye, target, up = (np.asarray(a, np.float64) for a in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(up, f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    m = np.stack([r, u, f], axis=1)  # columns = local x, y, z in world
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    return np.asarray(q, np.float32)


def load_hdr(path, width=None, height=None, brightness=1.0):
    """Radiance .hdr -> [1][height][width][4] float32 layer for pt_upload_texture_array (reference
    CLTextureArray::loadImage, src/opencl/texture.cpp:72-120); size defaults to the file's own."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    if lib().pth_image_hdr_info(str(path).encode(), C.byref(w), C.byref(h)):
        _err("pth_image_hdr_info")
    width, height = width or w.value, height or h.value
    out = np.zeros((1, height, width, 4), np.float32)
    if lib().pth_image_load_hdr(str(path).encode(), width, height, float(brightness), _ptr(out)):
        _err("pth_image_load_hdr")
    return out


def load_png(path):
    """Decoded PNG as [height][width][4] uint8 RGBA, top row first (the file itself, no resampling)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    if lib().pth_image_png_info(str(path).encode(), C.byref(w), C.byref(h)):
        _err("pth_image_png_info")
    out = np.zeros((h.value, w.value, 4), np.uint8)
    if lib().pth_image_load_png_rgba8(str(path).encode(), _ptr(out)):
        _err("pth_image_load_png_rgba8")
    return out


def load_material_png(path, width=None, height=None, is_linear=False, as_bgra8=False):
    """PNG -> [1][height][width][4] layer of the material texture array (reference CLTextureArray::loadImage for the
    8-bit array, src/opencl/texture.cpp:72-92,112-131): Lanczos-3 rescale, FreeImage_AdjustGamma(1/2.2) unless is_linear,
    rows bottom-up; float32 r g b a = byte / 255 (what read_imagef returns), or with as_bgra8 the uint8 b g r a bitmap
    the reference uploads itself (device format PT_TEX_BGRA8_UNORM)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    if lib().pth_image_png_info(str(path).encode(), C.byref(w), C.byref(h)):
        _err("pth_image_png_info")
    width, height = width or w.value, height or h.value
    if as_bgra8:
        out8 = np.zeros((1, height, width, 4), np.uint8)
        if lib().pth_image_load_material_png_bgra8(str(path).encode(), width, height, int(is_linear), _ptr(out8)):
            _err("pth_image_load_material_png_bgra8")
        return out8
    out = np.zeros((1, height, width, 4), np.float32)
    if lib().pth_image_load_material_png(str(path).encode(), width, height, int(is_linear), _ptr(out)):
        _err("pth_image_load_material_png")
    return out


class TextureFiles:
    """UniqueTextureArray (reference src/opencl/texture.h:18-31): texture files, each once; index = material tex_id."""

    def __init__(self):
        self._h = lib().pth_texture_files_create()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pth_texture_files_destroy(self._h)
            self._h = None

    def add(self, path, is_linear=False, brightness=1.0):
        return lib().pth_texture_files_add(self._h, str(path).encode(), int(is_linear), float(brightness))

    def files(self):
        out = []
        for i in range(lib().pth_texture_files_count(self._h)):
            lin, br = C.c_int(0), C.c_float(0)
            out.append((lib().pth_texture_files_path(self._h, i, C.byref(lin), C.byref(br)).decode(), bool(lin.value), br.value))
        return out

    def load(self, width=1024, height=1024, as_bgra8=False):
        """[layers][height][width][4] for pt_upload_texture_array kind 0 (CLTextureArray ctor, 1024x1024 in the reference,
        src/raytracer.cpp:284): float32 r g b a, or with as_bgra8 the reference's own storage, uint8 b g r a; one zero layer
        when there is no file (std::max(1, arrayLength), texture.cpp:141)."""
        fs = self.files()
        if not fs:
            return np.zeros((1, height, width, 4), np.uint8 if as_bgra8 else np.float32)
        return np.concatenate([load_material_png(p, width, height, lin, as_bgra8) for p, lin, _ in fs])
