"""Oracle pinned against the known answers available for this path and against closed-form properties.
The reference ships no tests, fixtures or golden vectors: beyond these values its device math is 'parity unpinned'."""
import ctypes as C

import numpy as np

import oracle_lib

L = oracle_lib.lib()


def test_squares_rng_known_answer():
    # random_uint32_t_base(0xfcbd6e15, 0..3) printed by the reference's own header
    want = [0xc4dd8039, 0x9a790012, 0x681b4e66, 0xa69b3786]
    got = [L.oracle_squares32(C.c_uint32(0xfcbd6e15), C.c_uint32(i)) for i in range(4)]
    assert got == want


def test_sobol_known_answer():
    # random_sobol(5, 17) = 76a64ec1 aefefe9d
    out = (C.c_uint32 * 2)()
    L.oracle_sobol(C.c_uint32(5), C.c_uint32(17), out)
    assert [out[0], out[1]] == [0x76a64ec1, 0xaefefe9d]


def test_random_target_table_follows_the_allocation_rule():
    # random.cuh:15-66: START_next = START + count*sets + 1
    rows = [("LENS_METHOD", 32, 1), ("LENS", 1, 1), ("LENS_BLADE", 1, 1), ("LENS_WAVELENGTH", 1, 1), ("BSDF_REFLECTION", 1, 3), ("BSDF_DIFFUSE", 1, 3),
            ("BSDF_REFRACTION", 1, 3), ("BSDF_RESAMPLING", 1, 3), ("BSDF_OPACITY", 1, 3), ("VOLUME_INTERSECTION", 1, 1), ("RUSSIAN_ROULETTE", 1, 1),
            ("CAMERA_JITTER", 1, 1), ("CAMERA_TIME", 1, 1), ("CLOUD_STEP_OFFSET", 3, 1), ("CLOUD_STEP_COUNT", 3, 1), ("CLOUD_DIR", 1, 1),
            ("SKY_STEP_OFFSET", 1, 1), ("SKY_INSCATTERING_STEP", 1, 1), ("CAUSTIC_INITIAL", 128, 2), ("CAUSTIC_RESAMPLING", 1, 2),
            ("CAUSTIC_SUN_RAY", 1, 2), ("LIGHT_SUN_INITIAL_VERTEX", 1, 1), ("LIGHT_SUN_BSDF", 1, 2), ("LIGHT_SUN_BSDF_METHOD", 1, 2),
            ("LIGHT_SUN_RAY", 1, 2), ("LIGHT_SUN_RESAMPLING", 1, 2), ("LIGHT_GEO_INITIAL_VERTEX", 8, 1), ("LIGHT_GEO_RAY", 8, 2),
            ("LIGHT_GEO_RESAMPLING", 1, 2), ("LIGHT_GEO_TREE_PREPASS", 8, 2), ("LIGHT_GEO_TREE_POSTPASS", 8, 2),
            ("LIGHT_GEO_BRIDGE_DISTANCE", 64, 1), ("LIGHT_GEO_BRIDGE_PHASE", 64, 1), ("LIGHT_GEO_BRIDGE_LIGHT_POINT", 8, 1),
            ("LIGHT_GEO_BRIDGE_VERTEX_COUNT", 8, 1), ("LIGHT_BSDF_CHOICE", 1, 1), ("LIGHT_BSDF_DIRECTION", 1, 1), ("LIGHT_BSDF_TRACE", 1, 1),
            ("LIGHT_BSDF_RR", 1, 1)]
    v, table = 0, {}
    for name, count, sets in rows:
        table[name] = v
        v += count * sets + 1
    assert v == 577
    expect = {"LENS": 33, "BSDF_REFLECTION": 39, "BSDF_RESAMPLING": 51, "BSDF_OPACITY": 55, "RUSSIAN_ROULETTE": 61, "CAMERA_JITTER": 63,
              "LIGHT_GEO_RAY": 367, "LIGHT_GEO_RESAMPLING": 384, "LIGHT_GEO_TREE_PREPASS": 387, "LIGHT_GEO_TREE_POSTPASS": 404,
              "LIGHT_BSDF_CHOICE": 569, "LIGHT_BSDF_DIRECTION": 571, "LIGHT_BSDF_TRACE": 573, "LIGHT_BSDF_RR": 575}
    for k, val in expect.items():
        assert table[k] == val, k


def test_random_2d_is_sobol_plus_bluenoise():
    bn = oracle_lib.bluenoise()
    out = (C.c_uint32 * 2)()
    sob = (C.c_uint32 * 2)()
    for target, px, py, sample, depth in [(39, 0, 0, 0, 0), (61, 17, 250, 12345, 3), (575, 1919, 1079, (1 << 20) - 1, 8)]:
        L.oracle_random_2d(bn.ctypes.data_as(C.c_void_p), C.c_uint32(target), C.c_uint32(px), C.c_uint32(py), C.c_uint32(sample), C.c_uint32(depth), out)
        dim = target + depth * 577
        L.oracle_sobol(C.c_uint32(sample), C.c_uint32(dim), sob)
        ox, oy = ((1 + dim) * 3242174889) & 0xFFFFFFFF, ((1 + dim) * 2447445413) & 0xFFFFFFFF
        texel = int(bn[((px + (ox >> 24)) & 255) + ((py + (oy >> 24)) & 255) * 256])
        assert out[0] == (sob[0] + (texel & 0xFFFF0000)) & 0xFFFFFFFF
        assert out[1] == (sob[1] + ((texel << 16) & 0xFFFFFFFF)) & 0xFFFFFFFF


def test_record_pack_keeps_21_bits():
    rng = np.random.RandomState(0)
    for _ in range(200):
        v = (rng.rand(3) * 10 ** rng.uniform(-6, 3)).astype(np.float32)
        packed = (C.c_uint32 * 2)()
        out = (C.c_float * 3)()
        L.oracle_record_roundtrip((C.c_float * 3)(*v), packed, out)
        want = (v.view(np.uint32) & 0xFFFFF800).view(np.float32)  # math.cuh:1547-1575: sign, exponent, 12 mantissa bits
        assert np.array_equal(np.array(out[:], dtype=np.float32), want)


def test_direction_and_normal_packing_round_trip():
    rng = np.random.RandomState(1)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for v in d:
        packed = (C.c_uint32 * 2)()
        out = (C.c_float * 3)()
        L.oracle_ray_roundtrip((C.c_float * 3)(*v), packed, out)
        assert np.abs(np.array(out[:]) - v).max() < 2e-6
        n = L.oracle_normal_pack((C.c_float * 3)(*v))
        L.oracle_normal_unpack(C.c_uint32(n), out)
        assert np.abs(np.array(out[:]) - v).max() < 1e-4  # 2 x 16 bit octahedral
    # saturation corner: +x maps to the last code, not to an overflowed 0 (CUDA float->uint conversion saturates)
    packed = (C.c_uint32 * 2)()
    out = (C.c_float * 3)()
    L.oracle_ray_roundtrip((C.c_float * 3)(1.0, 0.0, 0.0), packed, out)
    assert packed[0] == 0xFFFFFFFF


def test_deterministic_transcendentals_are_accurate():
    xs = np.concatenate([np.linspace(-7, 7, 20001), np.linspace(0, 2 * np.pi, 5001)]).astype(np.float32)
    out = (C.c_float * 2)()
    err = 0.0
    for x in xs[::7]:
        L.oracle_sincos(C.c_float(float(x)), out)
        err = max(err, abs(out[0] - np.sin(np.float64(x))), abs(out[1] - np.cos(np.float64(x))))
    assert err < 3e-7
    rng = np.random.RandomState(2)
    pts = rng.normal(size=(4000, 2)).astype(np.float32)
    e2 = max(abs(L.oracle_atan2(float(y), float(x)) - np.arctan2(np.float64(y), np.float64(x))) for y, x in pts)
    assert e2 < 5e-7
    assert L.oracle_atan2(0.0, -1.0) == np.float32(np.pi) and L.oracle_atan2(1.0, 0.0) == np.float32(np.pi / 2)
