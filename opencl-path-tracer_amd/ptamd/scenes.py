"""Procedural scenes for the configurations of BASELINE.json / SURVEY.md section 8(d).

Everything is generated from code and seeds: /root/reference (and its bunny PLY) does not exist on
the GPU box, so the ~70k-triangle mesh is a displaced icosphere ("blob", 81 920 triangles at
subdivision level 6) as SURVEY 8(d) config 2 allows.  `Mesh.from_ply` still loads the real bunny
when a path is given."""
import numpy as np
from . import host as H
from . import layout as L


# ---------------------------------------------------------------- geometry helpers
def _quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2),(p0,p2,p3); geometric normal = (p1-p0)x(p2-p0)."""
    v = np.asarray([p0, p1, p2, p3], np.float32)
    n = np.cross(v[1] - v[0], v[2] - v[0])
    n = n / np.linalg.norm(n)
    uv = np.asarray([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    return v, np.tile(n.astype(np.float32), (4, 1)), uv, np.asarray([[0, 1, 2], [0, 2, 3]], np.uint32)


class _MeshBuilder:
    def __init__(self):
        self.v, self.n, self.uv, self.i, self.m = [], [], [], [], []
        self.count = 0

    def add_quad(self, p0, p1, p2, p3, mat):
        v, n, uv, i = _quad(p0, p1, p2, p3)
        self.v.append(v), self.n.append(n), self.uv.append(uv), self.i.append(i + self.count)
        self.m += [mat, mat]
        self.count += 4

    def add_box(self, centre, half, yaw_deg, mat, with_bottom=True):
        """Axis box rotated about +y; all faces wound so that normals point outwards."""
        c, h = np.asarray(centre, np.float64), np.asarray(half, np.float64)
        a = np.radians(yaw_deg)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])

        def P(sx, sy, sz):
            return c + rot @ (h * np.array([sx, sy, sz]))
        faces = [
            (P(-1, 1, -1), P(-1, 1, 1), P(1, 1, 1), P(1, 1, -1)),      # top    (+y)
            (P(-1, -1, 1), P(1, -1, 1), P(1, 1, 1), P(-1, 1, 1)),      # +z
            (P(1, -1, -1), P(-1, -1, -1), P(-1, 1, -1), P(1, 1, -1)),  # -z
            (P(1, -1, 1), P(1, -1, -1), P(1, 1, -1), P(1, 1, 1)),      # +x
            (P(-1, -1, -1), P(-1, -1, 1), P(-1, 1, 1), P(-1, 1, -1)),  # -x
        ]
        if with_bottom:
            faces.append((P(-1, -1, -1), P(1, -1, -1), P(1, -1, 1), P(-1, -1, 1)))  # bottom (-y)
        for f in faces:
            self.add_quad(*f, mat)

    def build(self, materials, builder):
        return H.Mesh(np.concatenate(self.v), np.concatenate(self.i), materials,
                      material_index=np.asarray(self.m, np.uint32), normals=np.concatenate(self.n),
                      tex_coords=np.concatenate(self.uv), builder=builder)


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        uniq, inv = np.unique(e, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        nf = len(f)
        a, b, c = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], a, c], 1), np.stack([f[:, 1], b, a], 1),
                            np.stack([f[:, 2], c, b], 1), np.stack([a, b, c], 1)])
    return v, f


def blob_mesh(material, level=6, seed=7, builder=H.BVH_SPATIAL_SPLIT, radius=0.5):
    """Displaced icosphere: low-frequency lobes + ripples, outward-wound, smooth normals generated
    by the host library.  level 6 -> 40 962 vertices / 81 920 triangles."""
    v, f = icosphere(level)
    rng = np.random.default_rng(seed)
    d = np.zeros(len(v))
    for k in range(6):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        freq = 1.5 + 1.7 * k
        d += (0.22 / (1 + k)) * np.sin(freq * (v @ axis) * np.pi + rng.uniform(0, 2 * np.pi))
    r = radius * (1.0 + 0.55 * d)
    p = (v * r[:, None]).astype(np.float32)
    return H.Mesh(p, f.astype(np.uint32), [material], builder=builder)


# ---------------------------------------------------------------- textures
def procedural_sky(width=2048, height=1024, seed=3, brightness=1.0):
    """Analytic equirect HDR sky: horizon-to-zenith gradient, ground, sun disc. float32 RGBA,
    layout [1][h][w][4] as pt_upload_texture_array expects (what read_imagef would return)."""
    rng = np.random.default_rng(seed)
    u = (np.arange(width) + 0.5) / width
    vv = (np.arange(height) + 0.5) / height
    # inverse of readSkydome (assets/cl/skydome.cl:12-26): u=(1+atan2(x,-z)/pi)/2, image v'=1-acos(y)/pi
    phi = (2 * u - 1) * np.pi
    theta = (1 - vv) * np.pi
    y = np.cos(theta)[:, None] * np.ones(width)[None, :]
    s = np.sin(theta)[:, None]
    x = s * np.sin(phi)[None, :]
    z = -s * np.cos(phi)[None, :]
    zen = np.clip(y, 0, 1)[..., None]
    sky = (1 - zen) * np.array([0.9, 0.95, 1.0]) + zen * np.array([0.25, 0.45, 0.9])
    ground = np.array([0.25, 0.22, 0.2]) * (0.6 + 0.4 * np.clip(-y, 0, 1))[..., None]
    img = np.where((y >= 0)[..., None], sky, ground)
    sun_dir = np.array([0.45, 0.75, -0.48])
    sun_dir /= np.linalg.norm(sun_dir)
    cosang = x * sun_dir[0] + y * sun_dir[1] + z * sun_dir[2]
    img = img + np.array([60.0, 52.0, 40.0]) * (cosang > np.cos(np.radians(2.5)))[..., None]
    img = img + np.array([1.2, 1.0, 0.7]) * (np.clip(cosang, 0, 1) ** 64)[..., None]
    img = img * brightness * (1.0 + 0.0 * rng.random())
    out = np.ones((1, height, width, 4), np.float32)
    out[0, ..., :3] = img
    return out


def checker_texture(size=64, a=(0.8, 0.8, 0.8, 1.0), b=(0.2, 0.3, 0.7, 1.0), holes=False):
    t = np.zeros((size, size, 4), np.float32)
    yy, xx = np.mgrid[0:size, 0:size]
    m = ((xx // (size // 8)) + (yy // (size // 8))) % 2 == 0
    t[m] = a
    t[~m] = b
    if holes:
        t[(xx % 16 < 4) & (yy % 16 < 4), 3] = 0.0  # alpha-0 cut-outs (shading.cl:590-595)
    return t


# ---------------------------------------------------------------- scenes
class SceneBundle:
    def __init__(self, scene, camera, width, height, sky=None, material_textures=None, name=""):
        self.scene, self.camera, self.width, self.height = scene, camera, width, height
        self.sky, self.material_textures, self.name = sky, material_textures, name
        self._flat = None

    @property
    def flat(self):
        if self._flat is None:
            self._flat = self.scene.flatten()
        return self._flat


WHITE, GREEN, RED = (0.73, 0.73, 0.73), (0.12, 0.45, 0.15), (0.65, 0.05, 0.05)


def _room(mb, mats, light_half=0.25, light_y=1.98):
    """Cornell-style room x in [-1,1], y in [0,2], z in [-1,1], open towards -z; normals point inwards.
    material ids: 0 white, 1 left, 2 right, 3 light."""
    mb.add_quad((-1, 0, -1), (-1, 0, 1), (1, 0, 1), (1, 0, -1), 0)      # floor, n=+y
    mb.add_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1), 0)      # ceiling, n=-y
    mb.add_quad((-1, 0, 1), (-1, 2, 1), (1, 2, 1), (1, 0, 1), 0)        # back, n=-z
    mb.add_quad((-1, 0, -1), (-1, 2, -1), (-1, 2, 1), (-1, 0, 1), 1)    # left wall x=-1, n=+x
    mb.add_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1), 2)        # right wall x=+1, n=-x
    h = light_half
    mb.add_quad((-h, light_y, -h), (h, light_y, -h), (h, light_y, h), (-h, light_y, h), 3)  # light, n=-y


def _room_materials(light_intensity=12.0):
    return [L.material_diffuse(WHITE), L.material_diffuse(GREEN), L.material_diffuse(RED),
            L.material_emissive((1.0, 0.92, 0.8), light_intensity)]


def _camera(width, height, eye, target, fov, thin_lens=False, focal_distance=None, **kw):
    q = H.look_at_quat(eye, target)
    if focal_distance is None:
        focal_distance = float(np.linalg.norm(np.asarray(target, np.float64) - np.asarray(eye, np.float64)))
    return H.camera_data(eye, q, fov, width / height, focal_distance=focal_distance, thin_lens=thin_lens, **kw)


def cornell_box(width=512, height=512, builder=H.BVH_BINNED_SAH, box_materials=None):
    """Config 1: 5 walls + ceiling light + two boxes = 36 triangles, one mesh, identity instance."""
    mats = _room_materials()
    if box_materials is None:
        box_materials = [L.material_diffuse(WHITE), L.material_diffuse(WHITE)]
    mats += list(box_materials)
    mb = _MeshBuilder()
    _room(mb, mats)
    # the boxes float 2 mm above the floor: coincident faces (box bottom == floor) would make the reported
    # hit depend on the BVH visit order -- in the reference too -- and with it the material of the path
    mb.add_box((0.35, 0.302, -0.3), (0.3, 0.3, 0.3), -18.0, 4)
    mb.add_box((-0.35, 0.602, 0.3), (0.3, 0.6, 0.3), 20.0, 5)
    scene = H.Scene()
    scene.add_node(mb.build(mats, builder))
    cam = _camera(width, height, (0.0, 1.0, -3.9), (0.0, 1.0, 0.0), 40.0)
    return SceneBundle(scene, cam, width, height, name="cornell")


def blob_room(width=1920, height=1080, material=None, builder=H.BVH_BINNED_SAH, level=6, textured_floor=False):
    """Configs 2/3: the ~70k-triangle mesh (scale as main.cpp:146 would give a ~0.6 m bunny) inside the
    5-wall room with one 2-triangle area light."""
    if material is None:
        material = L.material_diffuse((0.8, 0.8, 0.8))
    mats = _room_materials()
    textures = None
    if textured_floor:
        mats.append(L.material_diffuse((0, 0, 0), texture_id=0))
        textures = checker_texture(64, holes=False)[None]
    mb = _MeshBuilder()
    _room(mb, mats)
    if textured_floor:
        mb.m[0] = mb.m[1] = len(mats) - 1
    scene = H.Scene()
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    blob = blob_mesh(material, level=level, builder=builder)
    scene.add_node(blob, location=(0.0, 0.75, 0.1), scale=(1.3, 1.3, 1.3))
    aspect_fov = 40.0 * (width / height) ** 0.5
    cam = _camera(width, height, (0.0, 1.0, -3.9), (0.0, 1.0, 0.0), min(aspect_fov, 75.0))
    return SceneBundle(scene, cam, width, height, material_textures=textures, name="blob_room")


def mixed_material_room(width=1920, height=1080, level=6, pattern="patches", builder=H.BVH_SPATIAL_SPLIT, seed=13):
    """The room of configs 2/3 with a mesh whose triangles carry FIVE materials -- diffuse, PBR metal, PBR dielectric, rough glass,
    basic glass -- `patches`: by region (a low-frequency field over the surface: what a painted or assembled object looks like, the
    hits of a wave are mostly one type), `confetti`: per triangle at random (every wave sees every type: the worst case for a shading
    kernel that dispatches on the material)."""
    mats5 = [L.material_diffuse((0.7, 0.7, 0.2)), L.material_pbr_metal((0.955, 0.638, 0.538), 0.8), L.material_pbr_dielectric((0.2, 0.3, 0.75), 0.7),
             L.material_refractive(0.9, 1.5, (1.0, 0.6, 0.6), 5.0), L.material_basic_refractive(1.5, (0.5, 1.0, 0.5), 3.0)]
    v, f = icosphere(level)
    rng = np.random.default_rng(seed)
    d = np.zeros(len(v))
    for k in range(6):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        d += (0.22 / (1 + k)) * np.sin((1.5 + 1.7 * k) * (v @ axis) * np.pi + rng.uniform(0, 2 * np.pi))
    p = (v * (0.5 * (1.0 + 0.55 * d))[:, None]).astype(np.float32)
    if pattern == "confetti":
        mi = rng.integers(0, 5, len(f))
    else:
        c = v[f].mean(axis=1)
        field = sum(np.sin(3.1 * (c @ a) + ph) for a, ph in ((rng.normal(size=3), rng.uniform(0, 6.28)) for _ in range(4)))
        mi = np.clip(((field - field.min()) / (np.ptp(field) + 1e-9) * 5).astype(int), 0, 4)
    blob = H.Mesh(p, f.astype(np.uint32), mats5, material_index=mi.astype(np.uint32), builder=builder)
    mats = _room_materials()
    mb = _MeshBuilder()
    _room(mb, mats)
    scene = H.Scene()
    scene.add_node(mb.build(mats, H.BVH_BINNED_SAH))
    scene.add_node(blob, location=(0.0, 0.75, 0.1), scale=(1.3, 1.3, 1.3))
    cam = _camera(width, height, (0.0, 1.0, -3.9), (0.0, 1.0, 0.0), min(40.0 * (width / height) ** 0.5, 75.0))
    return SceneBundle(scene, cam, width, height, name=f"mixed_room_{pattern}")


def instanced_grid(width=1920, height=1080, nx=4, nz=3, level=6, builder=H.BVH_SPATIAL_SPLIT, thin_lens=False,
                   sky_size=(2048, 1024), rotate=False):
    """Configs 4/5: nx*nz instances (translate + uniform scale only) of two unique
    ~82k-triangle meshes (copper PBR metal, main.cpp:149-151, alternating with a PBR dielectric) on a
    ground quad, procedural HDR sky + one small emissive quad (quirk 4).  4x3 -> 983 040 instanced
    blob triangles + 4."""
    copper = L.material_pbr_metal((0.955, 0.638, 0.538), 0.8)
    ceramic = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    meshes = [blob_mesh(copper, level=level, seed=7, builder=builder),
              blob_mesh(ceramic, level=level, seed=11, builder=builder)]
    scene = H.Scene()
    mb = _MeshBuilder()
    ext = 1.4 * max(nx, nz)
    mb.add_quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), 0)
    scene.add_node(mb.build([L.material_pbr_dielectric((0.5, 0.5, 0.5), 0.3)], H.BVH_BINNED_SAH))
    lb = _MeshBuilder()
    lb.add_quad((-0.6, 0, -0.6), (0.6, 0, -0.6), (0.6, 0, 0.6), (-0.6, 0, 0.6), 0)  # n = -y (faces down)
    scene.add_node(lb.build([L.material_emissive((1.0, 0.9, 0.75), 30.0)], H.BVH_BINNED_SAH),
                   location=(0.0, 4.0, 0.0))
    rng = np.random.default_rng(5)
    k = 0
    for iz in range(nz):
        for ix in range(nx):
            s = float(1.0 + 0.35 * rng.random())
            x = (ix - (nx - 1) / 2) * 1.5
            z = (iz - (nz - 1) / 2) * 1.5
            q = (1, 0, 0, 0)
            if rotate:  # a random rotation per instance (tests: world-space boxes re-fitted, packets that change octant inside an instance)
                axis = rng.normal(size=3)
                axis /= np.linalg.norm(axis)
                ang = rng.uniform(0, 2 * np.pi)
                q = (float(np.cos(ang / 2)), *(float(c) for c in np.sin(ang / 2) * axis))
            scene.add_node(meshes[k % 2], location=(x, 0.62 * s, z), orientation_wxyz=q, scale=(s, s, s))
            k += 1
    eye, target = (0.0, 2.6, -5.2), (0.0, 0.5, 0.0)
    cam = _camera(width, height, eye, target, 60.0, thin_lens=thin_lens, focal_length_mm=50.0,
                  aperture_fstops=2.0 if thin_lens else 8.0)
    sky = procedural_sky(*sky_size)
    return SceneBundle(scene, cam, width, height, sky=sky, name=f"instanced_grid_{nx}x{nz}")


def instanced_crowd(width=1920, height=1080, nx=20, nz=16, level=6, builder=H.BVH_SPATIAL_SPLIT, transform="general", sky_size=(2048, 1024), seed=21):
    """Config 4's camera, materials, sky and light over nx * nz instances of its two ~82k-triangle meshes, rows running away from the camera
    (20 x 16 = 320 instances = 26 M instanced triangles: ~2.6 GB as world-space copies, more than the library's 2 GB copy budget -- instancing
    exists so that this geometry is NOT copied).  `transform`: "general" = every instance turned about the vertical axis by a random angle and
    scaled by three different factors (scene.cl:116-139 enters any 4 x 4 inverse transform); "uniform" = translation + one scale factor, as in
    configs 4 / 5 (what the fold table of the per-ray kernels serves for up to 95 instances); "mixed" = two of three instances general."""
    copper = L.material_pbr_metal((0.955, 0.638, 0.538), 0.8)
    ceramic = L.material_pbr_dielectric((0.75, 0.2, 0.15), 0.7)
    meshes = [blob_mesh(copper, level=level, seed=7, builder=builder), blob_mesh(ceramic, level=level, seed=11, builder=builder)]
    scene = H.Scene()
    mb = _MeshBuilder()
    ext = 1.5 * max(nx, 2 * nz)
    mb.add_quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), 0)
    scene.add_node(mb.build([L.material_pbr_dielectric((0.5, 0.5, 0.5), 0.3)], H.BVH_BINNED_SAH))
    lb = _MeshBuilder()
    lb.add_quad((-0.6, 0, -0.6), (0.6, 0, -0.6), (0.6, 0, 0.6), (-0.6, 0, 0.6), 0)  # n = -y (faces down)
    scene.add_node(lb.build([L.material_emissive((1.0, 0.9, 0.75), 30.0)], H.BVH_BINNED_SAH), location=(0.0, 4.0, 0.0))
    rng = np.random.default_rng(seed)
    k = 0
    for iz in range(nz):
        for ix in range(nx):
            x, z = (ix - (nx - 1) / 2) * 1.5, (iz - 1.0) * 1.5  # (the first three rows stand where config 4's do)
            if transform == "general" or (transform == "mixed" and k % 3 != 0):
                sx, sy, sz = (float(v) for v in rng.uniform(0.8, 1.35, 3))
                ang = float(rng.uniform(0, 2 * np.pi))
                q = (float(np.cos(ang / 2)), 0.0, float(np.sin(ang / 2)), 0.0)
            else:
                sx = sy = sz = float(1.0 + 0.35 * rng.random())
                q = (1, 0, 0, 0)
            scene.add_node(meshes[k % 2], location=(x, 0.62 * sy, z), orientation_wxyz=q, scale=(sx, sy, sz))
            k += 1
    cam = _camera(width, height, (0.0, 2.6, -5.2), (0.0, 0.5, 0.0), 60.0, focal_length_mm=50.0, aperture_fstops=8.0)
    bundle = SceneBundle(scene, cam, width, height, sky=procedural_sky(*sky_size), name=f"instanced_crowd_{nx}x{nz}_{transform}")
    bundle.crowd_extent = ((nx - 1) / 2 * 1.5 + 1.0, (nz - 2.0) * 1.5 + 1.0)  # half-width in x, far end in z: where tests aim their rays
    return bundle


def instance_field(width=1920, height=1080, n=1000, level=3, seed=9, builder=H.BVH_BINNED_SAH, sky_size=(256, 128)):
    """Many instances of a small mesh: n randomly placed, rotated and scaled copies of a 20 * 4^level-triangle blob
    (level 3: 1 280 triangles, n = 1000 -> 1.28 M instanced triangles) over a ground quad -- the top-level tree is the deep one here."""
    mesh = blob_mesh(L.material_pbr_dielectric((0.7, 0.3, 0.2), 0.6), level=level, seed=3, builder=builder)
    scene = H.Scene()
    mb = _MeshBuilder()
    ext = 0.5 * n ** 0.5 + 2
    mb.add_quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), 0)
    scene.add_node(mb.build([L.material_diffuse((0.6, 0.6, 0.6))], H.BVH_BINNED_SAH))
    lb = _MeshBuilder()
    lb.add_quad((-1.5, 0, -1.5), (1.5, 0, -1.5), (1.5, 0, 1.5), (-1.5, 0, 1.5), 0)
    scene.add_node(lb.build([L.material_emissive((1.0, 0.9, 0.75), 40.0)], H.BVH_BINNED_SAH), location=(0.0, 6.0, 0.0))
    rng = np.random.default_rng(seed)
    side = 0.45 * n ** 0.5
    for _ in range(n):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 2 * np.pi)
        q = (float(np.cos(ang / 2)), *(float(c) for c in np.sin(ang / 2) * axis))
        s = float(rng.uniform(0.5, 1.1))
        scene.add_node(mesh, location=(float(rng.uniform(-side, side)), float(rng.uniform(0.4, 2.5)), float(rng.uniform(-side, side))),
                       orientation_wxyz=q, scale=(s, s, s))
    cam = _camera(width, height, (0.0, 4.0, -1.1 * side - 4.0), (0.0, 1.0, 0.0), 55.0)
    return SceneBundle(scene, cam, width, height, sky=procedural_sky(*sky_size), name=f"instance_field_{n}")
