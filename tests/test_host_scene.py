"""Host layer: .lum/.obj pipeline, device encoders and the light tree (reference: host/lum_v4.c, host/wavefront.c,
device/device_structs.c, device/device_packing.c, device/device_light.c). No GPU needed."""
import ctypes as C
import os

import numpy as np

import luminary_amd
from luminary_amd import scenes


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * np.dtype(dtype).itemsize,)).view(dtype).copy()


def test_cornell_through_lum_and_obj(tmp_path):
    host = scenes.cornell_host(str(tmp_path), 64, 48, 3)
    meshes, materials, instances = host.counts()
    assert (meshes, instances) == (1, 1)
    assert materials == 6  # default material 0 + 5 from the .mtl (wavefront.c:64-68)
    s = host.get_settings()
    assert (s.width, s.height, s.max_ray_depth) == (64, 48, 3)
    cam = host.get_camera()
    assert abs(cam.thin_lens.fov - 0.4) < 1e-7 and not cam.use_physical_camera and abs(cam.pos.z - 3.4) < 1e-6
    assert host.get_sky().mode == luminary_amd.SKY_MODE_CONSTANT_COLOR
    light = host.get_material(5)
    # Ke is scaled by emission_scale when read and again by the material's emission_scale (wavefront.c:385-396, :808); v4 forces
    # bidirectional emission (lum_v4.c:752); roughness = 1 - Ns/1000 (wavefront.c:806)
    assert light.emission_active and light.bidirectional_emission and abs(light.emission.r - 17.0) < 1e-6
    assert abs(host.get_material(4).roughness - 0.3) < 1e-6 and host.get_material(4).metallic
    v = host.device_scene()
    assert v.num_lights == 2 and v.width == 64 and v.height == 48
    tri_count = _arr(v.mesh_tri_offset, 2, np.uint32)[1]
    assert tri_count == 36


def test_obj_without_object_line_is_dropped(tmp_path):
    p = os.path.join(str(tmp_path), "noobj.obj")
    open(p, "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    host = luminary_amd.Host()
    host.load_obj_file(p)  # wavefront.c:843-848: warning, nothing added
    assert host.counts()[0] == 0


def test_obj_quads_negative_indices_and_degenerates(tmp_path):
    p = os.path.join(str(tmp_path), "q.obj")
    open(p, "w").write("o thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 0\nf 1 2 3 4\nf -5 -4 -3\nf 1 1 5\n")
    host = luminary_amd.Host()
    host.load_obj_file(p)
    host.new_instance(0)
    v = host.device_scene()
    assert _arr(v.mesh_tri_offset, 2, np.uint32)[1] == 3  # quad -> 2 triangles, negative indices -> 1, degenerate dropped
    verts = _arr(v.vertices, 3 * 3 * 4, np.float32).reshape(9, 4)
    assert np.allclose(verts[3:6, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])  # second half of the fan: v1 v3 v4


def test_material_encoding(tmp_path):
    host = luminary_amd.Host()
    m = luminary_amd.default_material()
    m.albedo = luminary_amd.RGBAF(0.25, 0.5, 0.75, 1.0)
    m.roughness, m.roughness_clamp, m.refraction_index = 0.3, 0.25, 1.5
    m.emission = luminary_amd.RGBF(4.0, 2.0, 1.0)
    m.emission_active = True
    m.metallic = True
    mid = host.add_material(m)
    host.add_mesh(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), np.array([mid], np.uint16))
    host.new_instance(0)
    w = _arr(host.device_scene().materials, 16, np.uint16)
    f = np.float32
    assert w[0] == (0x02 | 0x08 | 0x40) | ((int(f(0.25) * f(65535) + f(0.5)) >> 8) << 8)  # flags | clamp (device_structs.c:263-282)
    assert w[2] == int(f(0.3) * f(65535) + f(0.5)) and w[3] == int(f(0.5) * (f(1.5) - f(1.0)) * f(65535) + f(0.5))
    assert list(w[4:8]) == [int(f(x) * f(65535) + f(0.5)) for x in (0.25, 0.5, 0.75, 1.0)]
    norm = f(1.0) / f(5.0)  # 1 / (max emission + 1)
    assert list(w[8:11]) == [int(f(x) * norm * f(65535) + f(0.5)) for x in (4.0, 2.0, 1.0)]
    assert w[11] == (np.array([f(1.0) / norm], f).view(np.uint32)[0] >> 15) & 0xFFFF
    assert list(w[12:16]) == [0xFFFF] * 4


def test_light_tree_holds_every_emissive_triangle_once():
    host = scenes.example_scene(64, 36, 2, sphere_segments=6, ground_res=8, num_objects=10, num_lights=40)
    v = host.device_scene()
    assert v.num_lights == 80
    handles = _arr(v.light_tri_handles, 160, np.uint32).reshape(80, 2)
    assert len({(int(a), int(b)) for a, b in handles}) == 80
    inst = handles[0, 0]
    assert (handles[:, 0] == inst).all() and sorted(handles[:, 1]) == list(range(80))
    root = _arr(v.light_tree_root, 16, np.uint8)
    num_sections, num_root_lights = int(root[10]), int(root[6]) | (int(root[7]) << 8)
    sections = _arr(v.light_tree_root, 16 + 48 * num_sections, np.uint8)[16:].reshape(num_sections, 48)
    powers = sections[:, 32:].copy().view(np.uint16)
    children = int((powers > 0).sum())
    assert 2 <= children <= 128 and num_root_lights <= children
    assert powers.max() == 0xFFFF  # the strongest child normalises the 16-bit relative power (device_light.c:960-965)
    # light-BVH triangles are the world-space vertices of the listed triangles
    tris = _arr(v.light_bvh_tris, 80 * 12, np.float32).reshape(80, 3, 4)[:, :, :3]
    assert np.isfinite(tris).all() and (tris[:, :, 1] > 5.9).all()  # the emissive quads hang at y in [6, 12]
    # walking the nodes reaches every light exactly once
    nodes = _arr(v.light_tree_nodes, 64 * v.num_light_tree_nodes, np.uint8).reshape(-1, 64) if v.num_light_tree_nodes else np.zeros((0, 64), np.uint8)
    seen = list(range(num_root_lights))
    for n in nodes:
        nl = int(n[12])
        lp = int(n[20:24].copy().view(np.uint32)[0])
        seen += list(range(lp, lp + nl))
    assert sorted(seen) == list(range(80))


def test_defaults_match_reference_defaults():
    host = luminary_amd.Host()
    s = host.get_settings()
    assert (s.width, s.height, s.max_ray_depth, s.supersampling, s.undersampling, s.enable_adaptive_sampling) == (2560, 1440, 4, 1, 2, True)
    c = host.get_camera()
    assert c.aperture_blade_count == 7 and abs(c.russian_roulette_threshold - 0.1) < 1e-7 and abs(c.thin_lens.fov - 1.0) < 1e-7
    k = host.get_sky()
    assert k.mode == 0 and k.steps == 40 and abs(k.azimuth - 3.141) < 1e-6


def test_obj_loader_reads_texture_maps(tmp_path):
    """map_Kd / map_Ns / map_Bump of an .mtl become textures (wavefront.c:159-281); the PNG reader handles RGBA8, RGB8 with every
    scanline filter, 16-bit grey and palettes with tRNS, and takes the gamma from gAMA (png.c:541)."""
    import struct
    import zlib

    import numpy as np

    import luminary_amd
    import oracle_lib

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))

    def png(path, w, h, depth, colour, rows, extra=b""):
        raw = b"".join(rows)
        data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 0)) + extra + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
        open(path, "wb").write(data)

    rng = np.random.RandomState(3)
    rgba = rng.randint(0, 256, size=(5, 7, 4)).astype(np.uint8)
    png(str(tmp_path / "albedo.png"), 7, 5, 8, 6, [b"\x00" + rgba[y].tobytes() for y in range(5)], extra=chunk(b"gAMA", struct.pack(">I", 45455)))
    # RGB8 written with the Sub, Up, Average and Paeth filters (rows 1..4) to exercise the un-filtering
    rgb = rng.randint(0, 256, size=(5, 6, 3)).astype(np.int32)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    rows = []
    for y in range(5):
        cur, prev = rgb[y].reshape(-1), (rgb[y - 1].reshape(-1) if y else np.zeros(18, dtype=np.int32))
        out = []
        for i in range(18):
            a = cur[i - 3] if i >= 3 else 0
            b = prev[i]
            c = prev[i - 3] if i >= 3 else 0
            pred = [0, a, b, (a + b) // 2, paeth(int(a), int(b), int(c))][y]
            out.append((int(cur[i]) - int(pred)) & 0xFF)
        rows.append(bytes([y]) + bytes(out))
    png(str(tmp_path / "rough.png"), 6, 5, 8, 2, rows)
    grey16 = rng.randint(0, 65536, size=(3, 4)).astype(">u2")
    png(str(tmp_path / "bump.png"), 4, 3, 16, 0, [b"\x00" + grey16[y].tobytes() for y in range(3)])
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.5 0.5 0.5\nmap_Kd -s 1 1 1 albedo.png\nmap_Ns rough.png\nmap_Bump -bm 0.3 bump.png\nmap_Ke missing.png\n"
                                    "newmtl b\nKd 0.1 0.2 0.3\nmap_Kd albedo.png\n"
                                    "newmtl c\nKd 0 0 0\nmap_Ke albedo.png\nmap_refl rough.png\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\no quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\nusemtl b\nf 1/1 3/3 4/4\n")
    host = luminary_amd.Host()
    host.load_obj_file(str(tmp_path / "m.obj"))
    host.new_instance(0)
    ma, mb = host.get_material(1), host.get_material(2)  # material 0 is the loader's default material
    assert (ma.albedo_tex, ma.roughness_tex, ma.normal_tex, ma.luminance_tex, ma.metallic_tex) == (0, 1, 2, 0xFFFF, 0xFFFF)
    assert mb.albedo_tex == 0 and mb.roughness_tex == 0xFFFF  # the same file is one texture
    mc = host.get_material(3)  # an emission map makes the material an emitter (wavefront.c:809); a metallic map is stored (and never evaluated)
    assert (mc.luminance_tex, mc.metallic_tex, bool(mc.emission_active)) == (0, 1, True) and not bool(ma.emission_active)
    v = host.device_scene()
    assert v.num_textures == 3 + 2 and (v.sky_moon_albedo_tex, v.sky_moon_normal_tex) == (3, 4)  # the default sky mode brings the moon's two textures along
    import ctypes as C
    table = np.ctypeslib.as_array(C.cast(v.texture_table, C.POINTER(C.c_uint32)), shape=(3, 4)).copy()
    assert table[:, 1].tolist() == [7, 6, 4] and table[:, 2].tolist() == [5, 5, 3] and table[0, 0] == 0 and table[1, 0] == 35 and table[2, 0] == 65
    assert abs(table[0, 3:4].view(np.float32)[0] - 100000.0 / 45455.0) < 1e-6 and table[1, 3:4].view(np.float32)[0] == 1.0
    texels = np.ctypeslib.as_array(C.cast(v.texels, C.POINTER(C.c_uint32)), shape=(77,)).copy()
    assert np.array_equal(texels[:35].view(np.uint8).reshape(5, 7, 4), rgba)
    got_rgb = texels[35:65].view(np.uint8).reshape(5, 6, 4)
    assert np.array_equal(got_rgb[..., :3], rgb.astype(np.uint8)) and (got_rgb[..., 3] == 255).all()
    got_grey = texels[65:].view(np.uint8).reshape(3, 4, 4)
    assert np.array_equal(got_grey[..., 0], (grey16.astype(np.uint16) >> 8).astype(np.uint8)) and np.array_equal(got_grey[..., 0], got_grey[..., 2])
    # the packed uvs reach the device format and the oracle renders the textured quad
    host2 = oracle_lib.with_luts(v)
    assert host2.num_textures == 5


def test_png_reader_rejects_malformed_headers(tmp_path):
    """Textures named in a scene's .mtl are untrusted input: bit depths the format does not allow (0, 3, 5, 6, 7; 16 for palettes), absurd
    dimensions and a missing header end in 'texture ignored', never in a division by zero, a negative shift or an allocation failure."""
    import struct
    import zlib

    import luminary_amd

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))

    def png(path, w, h, depth, colour, raw, header=True):
        data = b"\x89PNG\r\n\x1a\n" + (chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 0)) if header else b"") + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
        open(path, "wb").write(data)

    cases = {"d0": (4, 4, 0, 0), "d3": (4, 4, 3, 0), "d5": (4, 4, 5, 3), "d7": (8, 2, 7, 0), "pal16": (2, 2, 16, 3), "huge": (0x7FFFFFFF, 0x7FFFFFFF, 8, 6), "wide": (20000, 1, 8, 0)}
    for name, (w, h, depth, colour) in cases.items():
        png(str(tmp_path / (name + ".png")), w, h, depth, colour, b"\x00" * 64)
    png(str(tmp_path / "nohdr.png"), 2, 2, 8, 0, b"\x00" * 6, header=False)
    names = list(cases) + ["nohdr"]
    (tmp_path / "m.mtl").write_text("".join("newmtl m_%s\nKd 0.5 0.5 0.5\nmap_Kd %s.png\n" % (n, n) for n in names))
    (tmp_path / "m.obj").write_text("mtllib m.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 0 1 0\n" + "".join("usemtl m_%s\nf 1 2 3\n" % n for n in names))
    host = luminary_amd.Host()
    host.load_obj_file(str(tmp_path / "m.obj"))  # must return: every bad texture is dropped with a warning
    assert host.get_num_meshes() == 1
    for i in range(host.get_num_materials()):
        assert host.get_material(i).albedo_tex == 0xFFFF, "no texture was created from a malformed file"
