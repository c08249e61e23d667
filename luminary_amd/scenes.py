"""Deterministic synthetic scenes for tests and benchmarks (SURVEY.md §8d "Synthetic inputs").

The reference ships no usable assets (`Example.lum` points at a missing `Example.obj`, no Cornell box, no Sponza), so every
workload is generated here:
  cornell_box_files  C1: 5 quads + 2 boxes + 1 emissive quad as .obj/.mtl/.lum files (exercises the scene pipeline)
  example_scene      C2: "Example-class" scene, ~100 k triangles: ground, instanced spheres/boxes, emissive quads (seed 1)
  hall_scene         C3: 1 M-triangle "Sponza-class" hall: arcade of columns/arches/curtains, emissive panels (seed 2)
  scan_scene         C5: displaced icosphere "scan" on a ground plane (seed 3), triangle count by subdivision level
Benchmark settings: supersampling 0, adaptive sampling off, constant-colour sky, thin lens, RR threshold 0.1.
"""
import math
import os

import numpy as np

from . import RGBAF, RGBF, SKY_MODE_CONSTANT_COLOR, SUBSTRATE_TRANSLUCENT, Host, Vec3, default_material


def apply_benchmark_settings(host, width, height, max_ray_depth, sky=(1.0, 1.0, 1.0)):
    s = host.get_settings()
    s.width, s.height, s.max_ray_depth = width, height, max_ray_depth
    s.supersampling = 0
    s.undersampling = 0
    s.enable_adaptive_sampling = False
    s.shading_mode = 0
    host.set_settings(s)
    k = host.get_sky()
    k.mode = SKY_MODE_CONSTANT_COLOR
    k.constant_color = RGBF(*sky)
    host.set_sky(k)


def set_camera(host, pos, rotation, fov=1.0):
    c = host.get_camera()
    c.pos = Vec3(*pos)
    c.rotation = Vec3(*rotation)
    c.thin_lens.fov = fov
    c.thin_lens.aperture_size = 0.0
    c.russian_roulette_threshold = 0.1
    host.set_camera(c)


# ---------------------------------------------------------------------------------------------------------------------
# C1: Cornell box as files
# ---------------------------------------------------------------------------------------------------------------------

def _quad(a, b, c, d):
    return [a, b, c, d]


def cornell_box_files(directory, width=64, height=64, bounces=1):
    """Writes cornell.obj / cornell.mtl / cornell.lum into `directory` and returns the .lum path. 36 triangles."""
    os.makedirs(directory, exist_ok=True)
    verts = []
    faces = []  # (material name, [indices 1-based])

    def add_quad(mat, pts):
        base = len(verts)
        verts.extend(pts)
        faces.append((mat, [base + 1, base + 2, base + 3, base + 4]))

    def add_box(mat, lo, hi, angle):
        cx, cz = 0.5 * (lo[0] + hi[0]), 0.5 * (lo[2] + hi[2])
        ca, sa = math.cos(angle), math.sin(angle)

        def rot(p):
            x, z = p[0] - cx, p[2] - cz
            return (cx + ca * x - sa * z, p[1], cz + sa * x + ca * z)
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        c = [rot(p) for p in [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)]]
        for q in ([4, 7, 6, 5], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7], [0, 3, 2, 1]):
            add_quad(mat, [c[i] for i in q])

    # room: x in [-1,1], y in [0,2], z in [-1,1]; the camera looks down -z from z = 3.4
    add_quad("white", [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)])      # floor
    add_quad("white", [(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)])      # ceiling
    add_quad("white", [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)])    # back
    add_quad("red", [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)])      # left
    add_quad("green", [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)])        # right
    add_quad("light", [(-0.3, 1.995, -0.3), (0.3, 1.995, -0.3), (0.3, 1.995, 0.3), (-0.3, 1.995, 0.3)])
    add_box("white", (-0.7, 0.0, -0.65), (-0.1, 1.2, -0.05), 0.3)
    add_box("metal", (0.1, 0.0, 0.05), (0.7, 0.6, 0.65), -0.3)

    with open(os.path.join(directory, "cornell.mtl"), "w") as f:
        f.write("newmtl white\nKd 0.73 0.73 0.73\nNs 300\n\n")
        f.write("newmtl red\nKd 0.65 0.05 0.05\nNs 300\n\n")
        f.write("newmtl green\nKd 0.12 0.45 0.15\nNs 300\n\n")
        f.write("newmtl metal\nKd 0.9 0.8 0.5\nKs 1.0 1.0 1.0\nNs 700\n\n")
        f.write("newmtl light\nKd 0.78 0.78 0.78\nKe 17.0 12.0 4.0\nNs 300\n")
    with open(os.path.join(directory, "cornell.obj"), "w") as f:
        f.write("mtllib cornell.mtl\no cornell\n")
        for v in verts:
            f.write("v %.6f %.6f %.6f\n" % v)
        cur = None
        for mat, idx in faces:
            if mat != cur:
                f.write("usemtl %s\n" % mat)
                cur = mat
            f.write("f %d %d %d %d\n" % tuple(idx))
    with open(os.path.join(directory, "cornell.lum"), "w") as f:
        f.write("Luminary\nVERSION 4\n# Cornell box, generated\n")
        f.write("GENERAL WIDTH___ %d\nGENERAL HEIGHT__ %d\nGENERAL BOUNCES_ %d\nGENERAL MESHFILE cornell.obj\n" % (width, height, bounces))
        f.write("CAMERA POSITION 0.0 1.0 3.4\nCAMERA ROTATION 0.0 0.0 0.0\nCAMERA FOV_____ 0.4\nCAMERA RUSSIANR 0.1\n")
        f.write("SKY MODE____ 2\nSKY COLORCON 0.0 0.0 0.0\n")
    return os.path.join(directory, "cornell.lum")


def cornell_host(directory, width=64, height=64, bounces=1):
    """Loads the generated Cornell box through the .lum/.obj pipeline and applies the benchmark overrides the v4 format cannot express."""
    host = Host()
    host.load_lum_file(cornell_box_files(directory, width, height, bounces))
    s = host.get_settings()
    s.supersampling = 0
    s.undersampling = 0
    s.enable_adaptive_sampling = False
    host.set_settings(s)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# mesh helpers
# ---------------------------------------------------------------------------------------------------------------------

def _sphere(segments):
    """UV sphere of radius 1: (9 floats per triangle positions, smooth normals)."""
    tris = []
    for i in range(segments):
        t0, t1 = math.pi * i / segments, math.pi * (i + 1) / segments
        for j in range(2 * segments):
            p0, p1 = math.pi * j / segments, math.pi * (j + 1) / segments

            def pt(t, p):
                return (math.sin(t) * math.cos(p), math.cos(t), math.sin(t) * math.sin(p))
            a, b, c, d = pt(t0, p0), pt(t1, p0), pt(t1, p1), pt(t0, p1)
            if i != 0:
                tris.append((a, b, d))
            if i != segments - 1:
                tris.append((b, c, d))
    pos = np.array(tris, dtype=np.float32).reshape(-1, 9)
    return pos, pos.copy()


def _box():
    c = [(-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), (-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)]
    quads = [[4, 7, 6, 5], [0, 1, 2, 3], [0, 4, 5, 1], [1, 5, 6, 2], [2, 6, 7, 3], [3, 7, 4, 0]]
    tris = []
    for q in quads:
        tris.append((c[q[0]], c[q[1]], c[q[2]]))
        tris.append((c[q[0]], c[q[2]], c[q[3]]))
    return np.array(tris, dtype=np.float32).reshape(-1, 9), None


def _grid(nx, nz, x0, x1, z0, z1, height_fn=None):
    xs = np.linspace(x0, x1, nx + 1, dtype=np.float32)
    zs = np.linspace(z0, z1, nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = np.zeros_like(X) if height_fn is None else height_fn(X, Z).astype(np.float32)
    P = np.stack([X, Y, Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.concatenate([a, c, b], axis=-1).reshape(-1, 9)
    t2 = np.concatenate([a, d, c], axis=-1).reshape(-1, 9)
    return np.concatenate([t1, t2], axis=0).astype(np.float32)


def _material(albedo, roughness=0.7, metallic=False, emission=None, bidirectional=True, alpha=1.0):
    m = default_material()
    m.albedo = RGBAF(albedo[0], albedo[1], albedo[2], alpha)
    m.roughness = roughness
    m.metallic = metallic
    if emission is not None:
        m.emission = RGBF(*emission)
        m.emission_active = True
        m.bidirectional_emission = bidirectional
    return m


# ---------------------------------------------------------------------------------------------------------------------
# C2: Example-class scene
# ---------------------------------------------------------------------------------------------------------------------

def example_scene(width=1920, height=1080, bounces=8, seed=1, sphere_segments=20, ground_res=64, num_objects=64, num_lights=16):
    """~100 k triangles with the defaults: ground grid, instanced spheres and boxes, emissive quads."""
    rng = np.random.RandomState(seed)
    host = Host()
    apply_benchmark_settings(host, width, height, bounces)
    palette = []
    for _ in range(12):
        albedo = 0.2 + 0.7 * rng.rand(3)
        rough = [0.05, 0.3, 0.7][rng.randint(3)]
        metallic = rng.rand() < 0.05
        if os.environ.get("LUM_EXPERIMENT_UNIFORM_MATERIALS"):  # measurement only: one material class everywhere
            rough, metallic = 0.7, False
        palette.append(host.add_material(_material(albedo, rough, metallic=metallic)))
    ground_mat = host.add_material(_material((0.6, 0.6, 0.6), 0.7))
    light_mat = host.add_material(_material((0.8, 0.8, 0.8), 0.7, emission=(20.0, 18.0, 15.0)))

    ground = _grid(ground_res, ground_res, -40, 40, -40, 40, lambda X, Z: 0.4 * np.sin(0.3 * X) * np.cos(0.25 * Z))
    gid = host.add_mesh(ground, np.full(len(ground), ground_mat, dtype=np.uint16))
    host.new_instance(gid)

    sp_pos, sp_nrm = _sphere(sphere_segments)
    bx_pos, _ = _box()
    sphere_ids = [host.add_mesh(sp_pos, np.full(len(sp_pos), palette[k % len(palette)], dtype=np.uint16), normals=sp_nrm) for k in range(4)]
    box_ids = [host.add_mesh(bx_pos, np.full(len(bx_pos), palette[(k + 5) % len(palette)], dtype=np.uint16)) for k in range(4)]
    for k in range(num_objects):
        x, z = rng.uniform(-30, 30), rng.uniform(-35, 10)
        s = rng.uniform(0.8, 2.5)
        if k % 2 == 0:
            host.new_instance(sphere_ids[k % 4], (x, s + 0.5, z), (0, 0, 0), (s, s, s))
        else:
            host.new_instance(box_ids[k % 4], (x, s + 0.4, z), (0.0, rng.uniform(0, 3.14), 0.0), (s, s, 0.7 * s))
    quads = []
    for k in range(num_lights):
        x, z, y = rng.uniform(-30, 30), rng.uniform(-35, 10), rng.uniform(6, 12)
        w = rng.uniform(0.8, 2.0)
        a, b, c, d = (x - w, y, z - w), (x + w, y, z - w), (x + w, y, z + w), (x - w, y, z + w)
        quads.append(a + c + b)
        quads.append(a + d + c)
    lq = np.array(quads, dtype=np.float32)
    lid = host.add_mesh(lq, np.full(len(lq), light_mat, dtype=np.uint16))
    host.new_instance(lid)
    set_camera(host, (0.0, 6.0, 28.0), (-0.18, 0.0, 0.0), fov=0.9)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# C3: 1 M-triangle hall
# ---------------------------------------------------------------------------------------------------------------------

def hall_scene(width=1920, height=1080, bounces=8, seed=2, target_triangles=1_000_000):
    """Sponza-class hall as ONE mesh: floor/walls/vault grids, rows of tessellated columns, arches and wavy curtains, emissive panels.
    Mix of long thin, large and tiny triangles. `target_triangles` scales the tessellation."""
    rng = np.random.RandomState(seed)
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.6, 0.7, 1.0))
    mats = {
        "stone": host.add_material(_material((0.7, 0.65, 0.6), 0.7)),
        "floor": host.add_material(_material((0.5, 0.5, 0.52), 0.3)),
        "column": host.add_material(_material((0.8, 0.78, 0.7), 0.3)),
        "curtain_r": host.add_material(_material((0.7, 0.15, 0.12), 0.7)),
        "curtain_b": host.add_material(_material((0.15, 0.2, 0.7), 0.7)),
        "brass": host.add_material(_material((0.9, 0.7, 0.3), 0.05, metallic=True)),
        "light": host.add_material(_material((0.8, 0.8, 0.8), 0.7, emission=(30.0, 26.0, 20.0))),
    }
    scale = math.sqrt(target_triangles / 1_000_000.0)
    parts, ids = [], []

    def add(tris, mat):
        parts.append(tris.astype(np.float32))
        ids.append(np.full(len(tris), mats[mat], dtype=np.uint16))

    L, W, H = 48.0, 14.0, 12.0
    nf = max(4, int(180 * scale))
    add(_grid(nf, nf // 3, -L, L, -W, W), "floor")
    # side walls (grids rotated into the xy plane) and vault
    wall = _grid(max(4, int(140 * scale)), max(2, int(40 * scale)), -L, L, 0.0, H)

    def to_wall(t, z, flip):
        q = t.reshape(-1, 3, 3).copy()
        y = q[:, :, 2].copy()
        q[:, :, 2] = z
        q[:, :, 1] = y
        if flip:
            q = q[:, ::-1, :]
        return q.reshape(-1, 9)
    add(to_wall(wall, -W, False), "stone")
    add(to_wall(wall, W, True), "stone")
    nv = max(4, int(160 * scale))
    vault = _grid(nv, max(4, int(60 * scale)), -L, L, -W, W, lambda X, Z: H + 3.0 * np.cos(Z / W * math.pi / 2))
    add(vault.reshape(-1, 3, 3)[:, ::-1, :].reshape(-1, 9), "stone")
    # columns: tessellated cylinders with fluting
    ncol = 14
    seg_a, seg_h = max(8, int(96 * scale)), max(4, int(160 * scale))
    for side in (-1, 1):
        for k in range(ncol):
            cx, cz = -L + 4.0 + k * (2 * L - 8.0) / (ncol - 1), side * (W - 3.5)
            ang = np.linspace(0, 2 * math.pi, seg_a + 1, dtype=np.float32)
            hs = np.linspace(0, H - 2.5, seg_h + 1, dtype=np.float32)
            A, Hh = np.meshgrid(ang, hs, indexing="ij")
            R = 0.55 + 0.04 * np.cos(12 * A) + 0.08 * np.exp(-Hh * 2.0)
            P = np.stack([cx + R * np.cos(A), Hh, cz + R * np.sin(A)], axis=-1)
            a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
            t = np.concatenate([np.concatenate([a, c, b], -1).reshape(-1, 9), np.concatenate([a, d, c], -1).reshape(-1, 9)], 0)
            add(t, "column" if k % 5 else "brass")
    # curtains: wavy sheets between columns
    nc_u, nc_v = max(6, int(120 * scale)), max(6, int(150 * scale))
    for side in (-1, 1):
        for k in range(0, ncol - 1, 2):
            x0 = -L + 4.0 + k * (2 * L - 8.0) / (ncol - 1)
            x1 = x0 + (2 * L - 8.0) / (ncol - 1)
            us = np.linspace(x0 + 0.7, x1 - 0.7, nc_u + 1, dtype=np.float32)
            vs = np.linspace(2.5, H - 3.0, nc_v + 1, dtype=np.float32)
            U, V = np.meshgrid(us, vs, indexing="ij")
            phase = rng.uniform(0, 6.28)
            Z = side * (W - 3.5) + 0.25 * np.sin(6.0 * (U - x0) + phase) * (1.0 - (V - 2.5) / (H - 5.5) * 0.6)
            P = np.stack([U, V, Z], axis=-1)
            a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
            t = np.concatenate([np.concatenate([a, c, b], -1).reshape(-1, 9), np.concatenate([a, d, c], -1).reshape(-1, 9)], 0)
            add(t, "curtain_r" if (k // 2) % 2 == 0 else "curtain_b")
    # emissive panels under the vault
    quads = []
    for k in range(32):
        x = -L + 3.0 + (k % 16) * (2 * L - 6.0) / 15
        z = (-1 if k < 16 else 1) * 3.0
        w = 0.9
        y = H + 1.2
        a, b, c, d = (x - w, y, z - w), (x + w, y, z - w), (x + w, y, z + w), (x - w, y, z + w)
        quads.append(a + b + c)
        quads.append(a + c + d)
    add(np.array(quads, dtype=np.float32), "light")
    mesh = host.add_mesh(np.concatenate(parts, 0), np.concatenate(ids, 0))
    host.new_instance(mesh)
    set_camera(host, (-40.0, 4.0, 0.0), (0.0, -1.5707963, 0.0), fov=0.9)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# C5: displaced icosphere "scan"
# ---------------------------------------------------------------------------------------------------------------------

def _icosphere(level):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2),
                  (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    tri = v[f]
    for _ in range(level):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], 0)
    return tri


def _icosphere_frequency(n, indexed=False):
    """Icosphere by frequency subdivision: every face of the icosahedron becomes n*n triangles (20*n*n in all), vertices pushed to the
    unit sphere. Unlike the recursive 4-way split this reaches any size, e.g. n = 708 -> 10.03 M triangles.
    indexed=True returns (points [20*G, 3], triangles [20*n*n, 3] indices) with G = (n+1)(n+2)/2 grid points per face."""
    base = _icosphere(0)  # [20, 3, 3]
    i, j = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    inside = (i + j) <= n
    gid = np.full((n + 1, n + 1), -1, dtype=np.int64)
    gid[inside] = np.arange(int(inside.sum()))
    gi, gj = i[inside].astype(np.float64), j[inside].astype(np.float64)
    wb, wc = gi / n, gj / n
    wa = 1.0 - wb - wc
    G = gi.size
    pts = np.empty((20, G, 3), dtype=np.float64)
    for f in range(20):
        a, b, c = base[f]
        pts[f] = wa[:, None] * a + wb[:, None] * b + wc[:, None] * c
    pts /= np.sqrt((pts * pts).sum(axis=2, keepdims=True))
    up = (i + j) <= n - 1          # (i,j) (i+1,j) (i,j+1)
    down = (i + j) <= n - 2        # (i+1,j) (i+1,j+1) (i,j+1)
    iu, ju, idn, jdn = i[up], j[up], i[down], j[down]
    tri = np.concatenate([np.stack([gid[iu, ju], gid[iu + 1, ju], gid[iu, ju + 1]], 1),
                          np.stack([gid[idn + 1, jdn], gid[idn + 1, jdn + 1], gid[idn, jdn + 1]], 1)], 0)  # [n*n, 3]
    tris = (tri[None, :, :] + (np.arange(20) * G)[:, None, None]).reshape(-1, 3)
    pts = pts.reshape(-1, 3)
    return (pts, tris) if indexed else pts[tris]


def scan_scene(width=1920, height=1080, bounces=8, seed=3, level=9, triangles=None):
    """Displaced icosphere plus ground and 8 area lights. `level`: 20 * 4**level triangles by recursive splitting (level 9 = 5.2 M);
    `triangles`: at least that many by frequency subdivision (10_000_000 -> 20 * 708**2 = 10.03 M, BASELINE config 5's size)."""
    rng = np.random.RandomState(seed)
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.8, 0.85, 1.0))
    clay = host.add_material(_material((0.75, 0.6, 0.5), 0.3))
    ground_mat = host.add_material(_material((0.5, 0.5, 0.5), 0.7))
    light_mat = host.add_material(_material((0.8, 0.8, 0.8), 0.7, emission=(40.0, 36.0, 30.0)))
    index = None
    if triangles:
        p, index = _icosphere_frequency(int(math.ceil(math.sqrt(triangles / 20.0))), indexed=True)  # displace the grid points once, not per corner
    else:
        p = _icosphere(level).reshape(-1, 3)
    disp = np.zeros(len(p))
    freq, amp = 1.5, 0.25
    for _ in range(6):
        d = rng.randn(3, 3)
        ph = rng.uniform(0, 6.28, 3)
        q = p @ d.T * freq
        disp += amp * np.sin(q[:, 0] + ph[0]) * np.sin(q[:, 1] + ph[1]) * np.sin(q[:, 2] + ph[2])
        freq *= 2.0
        amp *= 0.5
    p = p * (10.0 * (1.0 + 0.15 * disp))[:, None]
    p[:, 1] += 11.0
    if index is not None:
        p = p.astype(np.float32)[index]
    pos = p.reshape(-1, 9).astype(np.float32)
    mid = host.add_mesh(pos, np.full(len(pos), clay, dtype=np.uint16))
    host.new_instance(mid)
    g = _grid(64, 64, -60, 60, -60, 60)
    gid = host.add_mesh(g, np.full(len(g), ground_mat, dtype=np.uint16))
    host.new_instance(gid)
    quads = []
    for k in range(8):
        ang = 2 * math.pi * k / 8
        x, z, y, w = 25 * math.cos(ang), 25 * math.sin(ang), 30.0, 3.0
        a, b, c, d = (x - w, y, z - w), (x + w, y, z - w), (x + w, y, z + w), (x - w, y, z + w)
        quads.append(a + b + c)
        quads.append(a + c + d)
    lq = np.array(quads, dtype=np.float32)
    lid = host.add_mesh(lq, np.full(len(lq), light_mat, dtype=np.uint16))
    host.new_instance(lid)
    set_camera(host, (0.0, 12.0, 38.0), (-0.05, 0.0, 0.0), fov=0.8)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# Parity scene: every material branch and a light tree deep enough to need node descent
# ---------------------------------------------------------------------------------------------------------------------

def zoo_scene(width=96, height=64, bounces=8, seed=7, light_triangles=320, sky_mode=SKY_MODE_CONSTANT_COLOR, aperture=0.0, blades=0):
    """Small scene that exercises what the benchmark scenes do not: translucent (rough and smooth, IOR above and below the medium),
    coloured and plain transparency, partially opaque surfaces, metals, roughness-as-smoothness, one-sided emitters, rotated and
    non-uniformly scaled instances, more light triangles than the light-tree root holds (so lanes descend through tree nodes), an
    optional lens aperture (round or bladed) and either sky mode."""
    rng = np.random.RandomState(seed)
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.6, 0.7, 0.9))
    if sky_mode != SKY_MODE_CONSTANT_COLOR:
        k = host.get_sky()
        k.mode = sky_mode
        host.set_sky(k)

    def mat(albedo, roughness, alpha=1.0, **kw):
        m = _material(albedo, roughness, alpha=alpha, metallic=kw.get("metallic", False), emission=kw.get("emission"),
                      bidirectional=kw.get("bidirectional", True))
        if kw.get("translucent"):
            m.base_substrate = SUBSTRATE_TRANSLUCENT
            m.refraction_index = kw.get("ior", 1.5)
        m.colored_transparency = kw.get("colored", False)
        m.roughness_as_smoothness = kw.get("smoothness", False)
        if "clamp" in kw:
            m.roughness_clamp = kw["clamp"]
        return host.add_material(m)

    mats = [
        mat((0.7, 0.7, 0.7), 0.8),                                         # 0 diffuse
        mat((0.9, 0.6, 0.2), 0.1, metallic=True),                          # 1 smooth metal
        mat((0.8, 0.8, 0.9), 0.5, metallic=True, smoothness=True),         # 2 metal, roughness stored as smoothness
        mat((0.9, 0.95, 1.0), 0.02, alpha=0.1, translucent=True, ior=1.5),  # 3 clear glass
        mat((0.6, 0.9, 0.7), 0.35, alpha=0.3, translucent=True, ior=1.33, colored=True),  # 4 rough tinted glass
        mat((0.9, 0.3, 0.3), 0.6, alpha=0.5, colored=True),                # 5 half-transparent coloured sheet
        mat((0.5, 0.5, 0.5), 0.6, alpha=0.0),                              # 6 fully transparent, uncoloured (invisible to shadow rays)
        mat((0.3, 0.4, 0.9), 0.25, alpha=0.75),                            # 7 partially opaque, plain
        mat((0.95, 0.95, 0.95), 0.04, clamp=0.0),                          # 8 glossy dielectric coat, no clamp
        mat((0.9, 0.9, 0.9), 0.3, alpha=1.0, translucent=True, ior=1.0),   # 9 translucent with IOR 1 (pass-through branch)
    ]
    light_one_sided = mat((0.8, 0.8, 0.8), 0.7, emission=(14.0, 12.0, 9.0), bidirectional=False)
    light_two_sided = mat((0.8, 0.8, 0.8), 0.7, emission=(3.0, 6.0, 12.0), bidirectional=True)

    ground = _grid(8, 8, -12, 12, -12, 12, lambda X, Z: 0.15 * np.sin(0.9 * X) * np.cos(0.7 * Z))
    host.new_instance(host.add_mesh(ground, np.full(len(ground), mats[0], dtype=np.uint16)))
    sp_pos, sp_nrm = _sphere(10)
    bx_pos, _ = _box()
    k = 0
    for mid in mats[1:]:
        x, z = -9.0 + 2.3 * k, -2.0 + 1.5 * ((k * 7) % 3)
        if k % 2 == 0:
            m = host.add_mesh(sp_pos, np.full(len(sp_pos), mid, dtype=np.uint16), normals=sp_nrm)
            host.new_instance(m, (x, 1.4, z), (0.3 * k, 0.2, 0.1 * k), (1.2, 1.0 + 0.1 * k, 0.9))
        else:
            m = host.add_mesh(bx_pos, np.full(len(bx_pos), mid, dtype=np.uint16))
            host.new_instance(m, (x, 1.2, z), (0.2, 0.5 * k, -0.3), (0.9, 1.3, 0.6 + 0.1 * k))
        k += 1
    # a sheet made of two materials hanging in front of the objects
    sheet = _grid(2, 2, -3, 3, 0, 2.5)
    sheet = sheet.reshape(-1, 3, 3)[:, :, [0, 2, 1]].reshape(-1, 9).copy()  # stand it up (swap y and z)
    sheet_m = np.array([mats[5], mats[6], mats[7], mats[5], mats[3], mats[4], mats[6], mats[7]][:len(sheet)], dtype=np.uint16)
    host.new_instance(host.add_mesh(sheet.astype(np.float32), sheet_m), (0.0, 0.2, 3.5), (0.0, 0.3, 0.0), (1.0, 1.0, 1.0))
    # many small emitters: a strip of one-sided triangles above, a ring of two-sided ones around
    tris = []
    n_strip = light_triangles // 2
    for i in range(n_strip):
        x = -10.0 + 20.0 * i / max(n_strip - 1, 1)
        z = 2.0 * np.sin(0.37 * i)
        y = 6.5 + 0.3 * np.cos(0.21 * i)
        s = 0.08 + 0.05 * rng.rand()
        tris.append((x - s, y, z - s, x + s, y, z - s, x, y, z + s))  # facing down (one-sided emitters emit along -n or +n? both orders used)
        if i % 3 == 0:
            tris[-1] = (x - s, y, z - s, x, y, z + s, x + s, y, z - s)
    strip = np.array(tris, dtype=np.float32)
    host.new_instance(host.add_mesh(strip, np.full(len(strip), light_one_sided, dtype=np.uint16)))
    tris = []
    n_ring = light_triangles - n_strip
    for i in range(n_ring):
        a = 2.0 * np.pi * i / n_ring
        r = 9.0 + 0.5 * np.sin(5 * a)
        x, z, y = r * np.cos(a), r * np.sin(a), 1.0 + 2.5 * rng.rand()
        s = 0.1 + 0.1 * rng.rand()
        tris.append((x, y - s, z, x + s * np.sin(a), y + s, z - s * np.cos(a), x - s * np.sin(a), y + s, z + s * np.cos(a)))
    ring = np.array(tris, dtype=np.float32)
    host.new_instance(host.add_mesh(ring, np.full(len(ring), light_two_sided, dtype=np.uint16)), (0.0, 0.0, 0.0), (0.0, 0.4, 0.0), (1.0, 1.2, 1.0))

    c = host.get_camera()
    c.pos = Vec3(0.5, 3.2, 13.0)
    c.rotation = Vec3(-0.16, 0.03, 0.0)
    c.thin_lens.fov = 0.8
    c.thin_lens.aperture_size = aperture
    c.aperture_shape = 1 if blades else 0
    c.aperture_blade_count = blades if blades else 7
    c.object_distance = 12.0
    c.russian_roulette_threshold = 0.1
    host.set_camera(c)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# Parity scene for textures: albedo + alpha cut-outs, roughness map, normal map, gamma
# ---------------------------------------------------------------------------------------------------------------------

def _quad_uv(a, b, c, d, rep=1.0):
    """Two triangles of the quad a-b-c-d with positions (2 x 9) and uvs (2 x 6); uv (0,0) at a, (rep, rep) at c."""
    pos = np.array([a + b + c, a + c + d], dtype=np.float32)
    uv = np.array([[0, 0, rep, 0, rep, rep], [0, 0, rep, rep, 0, rep]], dtype=np.float32)
    return pos, uv


def textured_scene(width=96, height=64, bounces=6, seed=11):
    """Textured ground (checker with gamma 2.2), a fence of alpha cut-outs (alpha 0 / 0.5 / 1 texels) in front of a light, a panel with
    roughness and normal maps, a tinted window whose alpha comes from a texture (coloured transparency), and emitters."""
    rng = np.random.RandomState(seed)
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.5, 0.6, 0.8))

    n = 16
    yy, xx = np.mgrid[0:n, 0:n]
    checker = np.zeros((n, n, 4), dtype=np.uint8)
    checker[..., :3] = np.where(((xx // 2 + yy // 2) % 2)[..., None] == 0, (220, 200, 160), (60, 90, 140))
    checker[..., 3] = 255
    t_checker = host.add_texture(checker, gamma=2.2)

    fence = np.zeros((8, 8, 4), dtype=np.uint8)
    fence[..., :3] = (180, 140, 90)
    fence[..., 3] = 0
    fence[:, ::3, 3] = 255          # opaque slats
    fence[4, :, 3] = 128            # one half-transparent rail
    fence[0, :, 3] = 255
    t_fence = host.add_texture(fence)

    rough = np.zeros((8, 8, 4), dtype=np.uint8)
    rough[..., 0] = (np.linspace(10, 230, 64).reshape(8, 8)).astype(np.uint8)
    rough[..., 3] = 255
    t_rough = host.add_texture(rough)

    m = 16
    yy, xx = np.mgrid[0:m, 0:m].astype(np.float32)
    nx, ny = 0.35 * np.sin(xx * 0.8), 0.35 * np.cos(yy * 0.7)
    nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
    nmap = np.zeros((m, m, 4), dtype=np.uint8)
    nmap[..., 0], nmap[..., 1], nmap[..., 2], nmap[..., 3] = (nx * 0.5 + 0.5) * 255, (ny * 0.5 + 0.5) * 255, (nz * 0.5 + 0.5) * 255, 255
    t_normal = host.add_texture(nmap)

    glass = rng.randint(0, 255, size=(4, 4, 4)).astype(np.uint8)
    glass[..., 3] = (rng.randint(0, 3, size=(4, 4)) * 100).astype(np.uint8)  # alpha 0, 100/255, 200/255
    t_glass = host.add_texture(glass)

    def mat(albedo, roughness, **kw):
        mt = _material(albedo, roughness, alpha=kw.get("alpha", 1.0), metallic=kw.get("metallic", False), emission=kw.get("emission"))
        mt.albedo_tex = kw.get("albedo_tex", 0xFFFF)
        mt.roughness_tex = kw.get("roughness_tex", 0xFFFF)
        mt.normal_tex = kw.get("normal_tex", 0xFFFF)
        mt.colored_transparency = kw.get("colored", False)
        mt.roughness_as_smoothness = kw.get("smoothness", False)
        return host.add_material(mt)

    m_ground = mat((0.5, 0.5, 0.5), 0.8, albedo_tex=t_checker)
    m_fence = mat((0.5, 0.5, 0.5), 0.6, albedo_tex=t_fence)
    m_panel = mat((0.8, 0.8, 0.85), 0.3, roughness_tex=t_rough, normal_tex=t_normal, metallic=True)
    m_window = mat((0.5, 0.5, 0.5), 0.2, albedo_tex=t_glass, colored=True)
    m_missing = mat((0.2, 0.9, 0.2), 0.5, albedo_tex=77)  # dangling handle: the default albedo applies
    m_light = mat((0.8, 0.8, 0.8), 0.7, emission=(16.0, 14.0, 11.0))

    def add(quads, material, rep=1.0):
        pos, uv = zip(*[_quad_uv(*q, rep=rep) for q in quads])
        pos, uv = np.concatenate(pos), np.concatenate(uv)
        return host.add_mesh(pos, np.full(len(pos), material, dtype=np.uint16), uvs=uv)

    host.new_instance(add([((-10, 0, -10), (10, 0, -10), (10, 0, 10), (-10, 0, 10))], m_ground, rep=5.0))
    host.new_instance(add([((-4, 0, 2), (4, 0, 2), (4, 3, 2), (-4, 3, 2))], m_fence, rep=2.0))
    host.new_instance(add([((-6, 0.2, -3), (-1, 0.2, -4), (-1, 3.5, -4), (-6, 3.5, -3))], m_panel, rep=1.5), (0, 0, 0), (0.0, 0.2, 0.0), (1.0, 1.0, 1.0))
    host.new_instance(add([((1, 0.3, -2), (5, 0.3, -2), (5, 3.0, -2), (1, 3.0, -2))], m_window))
    host.new_instance(add([((6, 0, -1), (8, 0, -1), (8, 2, -1), (6, 2, -1))], m_missing))
    host.new_instance(add([((-1.5, 2.0, -1.0), (1.5, 2.0, -1.0), (1.5, 2.0, 1.0), (-1.5, 2.0, 1.0)),     # behind the fence, facing down
                           ((-8, 5.5, -6), (-5, 5.5, -6), (-5, 5.5, -3), (-8, 5.5, -3)),
                           ((4, 6.0, -5), (7, 6.0, -5), (7, 6.0, -2), (4, 6.0, -2))], m_light))
    set_camera(host, (0.5, 2.6, 9.0), (-0.17, 0.02, 0.0), fov=0.75)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# Parity scene for emission textures (map_Ke): textured emitters in the surface context, in light sampling and in the light tree
# ---------------------------------------------------------------------------------------------------------------------

def emissive_texture_scene(width=72, height=48, bounces=4):
    """Floor and back wall lit only by textured emitters (black sky):
    * a screen whose texture is black over one of its two triangles (that triangle is no light: its integrated intensity is 0),
    * a coloured screen whose material also has an albedo texture with alpha 0.5 (the light's colour is scaled by the texture's alpha)
      and a non-zero constant emission colour (which enters the stored emission scale, device_structs.c:289-303),
    * a screen whose emission texture handle dangles (no light at all), and a small constant emitter."""
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.0, 0.0, 0.0))
    n = 16
    yy, xx = np.mgrid[0:n, 0:n]
    u, v = (xx + 0.5) / n, 1.0 - (yy + 0.5) / n       # texture_load flips v: image row 0 is v = 1
    half = np.zeros((n, n, 4), dtype=np.uint8)
    half[..., 3] = 255
    lit = ((u - v) > 0.3) & (u < 0.9) & (v > 0.1)       # well inside the triangle (0,0)-(1,0)-(1,1) of _quad_uv; a black border, because
                                                        # wrap addressing would carry the lit corner over to the other triangle's corner
    half[lit] = (255, 230, 180, 255)
    t_half = host.add_texture(half)
    bars = np.zeros((8, 8, 4), dtype=np.uint8)
    bars[..., 3] = 255
    bars[:, 0::2, :3] = (250, 60, 40)
    bars[:, 1::2, :3] = (40, 90, 250)
    t_bars = host.add_texture(bars, gamma=2.2)
    veil = np.full((4, 4, 4), 255, dtype=np.uint8)
    veil[..., 3] = 128
    t_veil = host.add_texture(veil)

    def emitter(tex, scale, constant=(0.0, 0.0, 0.0), albedo_tex=0xFFFF):
        mt = _material((0.8, 0.8, 0.8), 0.7, emission=constant)
        mt.luminance_tex, mt.emission_scale, mt.albedo_tex = tex, scale, albedo_tex
        return host.add_material(mt)

    m_grey = host.add_material(_material((0.7, 0.7, 0.7), 0.6))
    m_half = emitter(t_half, 30.0)
    m_bars = emitter(t_bars, 12.0, constant=(0.5, 0.25, 0.0), albedo_tex=t_veil)
    m_dangling = emitter(99, 50.0)
    m_const = host.add_material(_material((0.8, 0.8, 0.8), 0.7, emission=(4.0, 4.0, 4.0)))

    def add(quads, material):
        pos, uv = zip(*[_quad_uv(*q) for q in quads])
        pos, uv = np.concatenate(pos), np.concatenate(uv)
        return host.add_mesh(pos, np.full(len(pos), material, dtype=np.uint16), uvs=uv)

    host.new_instance(add([((-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8)), ((-8, 0, -4), (8, 0, -4), (8, 6, -4), (-8, 6, -4))], m_grey))
    host.new_instance(add([((-5, 0.5, -3.5), (-1, 0.5, -3.5), (-1, 3.5, -3.5), (-5, 3.5, -3.5))], m_half))
    host.new_instance(add([((1, 0.5, -3.5), (5, 0.5, -3.5), (5, 3.5, -3.5), (1, 3.5, -3.5))], m_bars))
    host.new_instance(add([((-1, 4.0, -3.5), (1, 4.0, -3.5), (1, 5.0, -3.5), (-1, 5.0, -3.5))], m_dangling))
    host.new_instance(add([((-0.5, 0.05, 1.0), (0.5, 0.05, 1.0), (0.5, 0.05, 2.0), (-0.5, 0.05, 2.0))], m_const))
    set_camera(host, (0.0, 2.2, 7.0), (-0.05, 0.0, 0.0), fov=0.9)
    return host


# ---------------------------------------------------------------------------------------------------------------------
# Edge cases for the parity tests
# ---------------------------------------------------------------------------------------------------------------------

def edge_scene(kind, width=48, height=32, bounces=4):
    """Small scenes at the borders of the input domain:
    "empty"        no geometry at all (every path leaves to the sky on its first ray)
    "no_lights"    geometry without any emissive triangle (no light tree: next-event estimation is skipped)
    "degenerate"   zero-area and duplicated coplanar triangles, a zero-scale instance, an instance of an empty mesh
    "one_triangle" a single emissive triangle seen from both sides
    """
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.4, 0.5, 0.7))
    grey = host.add_material(_material((0.6, 0.6, 0.6), 0.6))
    glow = host.add_material(_material((0.8, 0.8, 0.8), 0.7, emission=(9.0, 8.0, 7.0)))
    if kind == "empty":
        pass
    elif kind == "no_lights":
        g = _grid(4, 4, -8, 8, -8, 8)
        host.new_instance(host.add_mesh(g, np.full(len(g), grey, dtype=np.uint16)))
        bx, _ = _box()
        host.new_instance(host.add_mesh(bx, np.full(len(bx), grey, dtype=np.uint16)), (0.0, 1.0, -1.0), (0.3, 0.4, 0.0), (1.5, 1.0, 1.0))
    elif kind == "degenerate":
        tris = np.array([
            [-6, 0, -6, 6, 0, -6, 6, 0, 6], [-6, 0, -6, 6, 0, 6, -6, 0, 6],        # floor
            [-6, 0, -6, 6, 0, -6, 6, 0, 6], [-6, 0, -6, 6, 0, 6, -6, 0, 6],        # the same floor again (exact ties)
            [1, 1, 1, 1, 1, 1, 1, 1, 1], [0, 2, 0, 1, 2, 0, 2, 2, 0],              # a point and a line: zero area
            [-2, 0.5, -2, 2, 0.5, -2, 0, 3.0, -2],                                  # a regular triangle
        ], dtype=np.float32)
        m = host.add_mesh(tris, np.full(len(tris), grey, dtype=np.uint16))
        host.new_instance(m)
        host.new_instance(m, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 1.0, 1.0))   # collapsed along x
        lq = np.array([[-1, 4, -1, 1, 4, 1, 1, 4, -1], [-1, 4, -1, -1, 4, 1, 1, 4, 1]], dtype=np.float32)
        host.new_instance(host.add_mesh(lq, np.full(2, glow, dtype=np.uint16)))
    elif kind == "one_triangle":
        t = np.array([[-3, 0.5, -2, 3, 0.5, -2, 0, 4.0, -2]], dtype=np.float32)
        host.new_instance(host.add_mesh(t, np.full(1, glow, dtype=np.uint16)))
    else:
        raise ValueError(kind)
    set_camera(host, (0.0, 2.0, 9.0), (-0.1, 0.0, 0.0), fov=0.8)
    return host


def probe_light_scene(directory, triangle, emission, width=8, height=8, bounces=0):
    """One bidirectional emissive triangle under a black sky, the camera at (3, 0, 0) looking down -z past it: what tests/test_fog.py integrates
    the fog's single scattering over. `directory` is unused (kept for symmetry with the file-based scenes)."""
    host = Host()
    apply_benchmark_settings(host, width, height, bounces, sky=(0.0, 0.0, 0.0))
    glow = host.add_material(_material((0.0, 0.0, 0.0), 0.9, emission=emission, bidirectional=True))
    t = np.asarray(triangle, dtype=np.float32).reshape(1, 9)
    host.new_instance(host.add_mesh(t, np.full(1, glow, dtype=np.uint16)))
    set_camera(host, (3.0, 0.0, 0.0), (0.0, 0.0, 0.0), fov=0.05)
    return host
