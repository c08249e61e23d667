"""luminary_amd - Python host-side mirror of the Luminary C host API for the MI355X path-tracing core.

The product is libluminary_amd.so (HIP kernels + C ABI, see include/luminary_amd.h and include/lum_core.h). This package is a
thin ctypes binding with the same names and argument meaning as the reference's `luminary_host_*` functions
(/root/reference/include/luminary/host.h:29-129) plus the additive `luminary_ext_*` functions. There is no CPU rendering path:
rendering calls raise `LuminaryError` when no HIP device is usable.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
# LUM_LIB: an experiment variant built here with `python -m luminary_amd.build --variant NAME` (luminary_amd/lib/variants/NAME/), for A/B runs on the GPU box
LIB_PATH = os.environ.get("LUM_LIB") or os.path.join(_HERE, "lib", "libluminary_amd.so")


class LuminaryError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed: %s (code %d)" % (what, _lib().luminary_result_to_string(C.c_uint64(code)).decode(), code & 0xFFFF))


# ---- PODs (include/luminary_amd.h; reference include/luminary/structs.h:29-391) ----
class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class RGBF(C.Structure):
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float)]


class RGBAF(C.Structure):
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("a", C.c_float)]


class HostCreateInfo(C.Structure):
    _fields_ = [("device_mask", C.c_uint32)]


class RendererSettings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_ray_depth", C.c_uint32), ("bridge_max_num_vertices", C.c_uint32),
                ("undersampling", C.c_uint32), ("supersampling", C.c_uint32), ("enable_adaptive_sampling", C.c_bool),
                ("adaptive_sampling_max_sampling_rate", C.c_uint32), ("adaptive_sampling_avg_sampling_rate", C.c_uint32),
                ("adaptive_sampling_update_interval", C.c_uint32), ("adaptive_sampling_exposure_aware", C.c_bool),
                ("adaptive_sampling_output_mode", C.c_int), ("shading_mode", C.c_int), ("region_x", C.c_float), ("region_y", C.c_float),
                ("region_width", C.c_float), ("region_height", C.c_float)]


class _ThinLens(C.Structure):
    _fields_ = [("fov", C.c_float), ("aperture_size", C.c_float)]


class _Physical(C.Structure):
    _fields_ = [("allow_reflections", C.c_bool), ("use_spectral_rendering", C.c_bool)] + [(n, C.c_float) for n in (
        "focal_length", "front_focal_point", "back_focal_point", "front_principal_point", "back_principal_point", "aperture_point",
        "aperture_diameter", "exit_pupil_point", "exit_pupil_diameter", "image_plane_distance", "sensor_width")]


class Camera(C.Structure):
    _fields_ = [("pos", Vec3), ("rotation", Vec3), ("aperture_shape", C.c_int), ("aperture_blade_count", C.c_uint32), ("exposure", C.c_float),
                ("tonemap", C.c_int), ("agx_custom_slope", C.c_float), ("agx_custom_power", C.c_float), ("agx_custom_saturation", C.c_float),
                ("filter", C.c_int), ("use_local_error_minimization", C.c_bool), ("bloom_blend", C.c_float), ("dithering", C.c_bool),
                ("purkinje", C.c_bool), ("purkinje_kappa1", C.c_float), ("purkinje_kappa2", C.c_float), ("wasd_speed", C.c_float),
                ("mouse_speed", C.c_float), ("smooth_movement", C.c_bool), ("smoothing_factor", C.c_float),
                ("russian_roulette_threshold", C.c_float), ("use_color_correction", C.c_bool), ("color_correction", RGBF),
                ("film_grain", C.c_float), ("camera_scale", C.c_float), ("object_distance", C.c_float), ("use_physical_camera", C.c_bool),
                ("thin_lens", _ThinLens), ("physical", _Physical)]


class Sky(C.Structure):
    _fields_ = [("geometry_offset", Vec3)] + [(n, C.c_float) for n in ("azimuth", "altitude", "moon_azimuth", "moon_altitude", "moon_tex_offset",
                                                                       "sun_strength", "base_density")] + \
               [("ozone_absorption", C.c_bool), ("steps", C.c_uint32), ("stars_count", C.c_uint32), ("stars_seed", C.c_uint32)] + \
               [(n, C.c_float) for n in ("stars_intensity", "rayleigh_density", "mie_density", "ozone_density", "rayleigh_falloff", "mie_falloff",
                                         "mie_diameter", "ground_visibility", "ozone_layer_thickness", "multiscattering_factor")] + \
               [("hdri_dim", C.c_uint32), ("hdri_samples", C.c_uint32), ("aerial_perspective", C.c_bool), ("constant_color", RGBF), ("mode", C.c_int)]


class Fog(C.Structure):
    """LuminaryFog (include/luminary_amd.h): a homogeneous scattering volume around the camera (fog.c:6-16 for the defaults)."""
    _fields_ = [("active", C.c_bool), ("density", C.c_float), ("droplet_diameter", C.c_float), ("height", C.c_float), ("dist", C.c_float)]


class Ocean(C.Structure):
    """LuminaryOcean (include/luminary_amd.h; defaults ocean.c:6-22)."""
    _fields_ = [("active", C.c_bool), ("height", C.c_float), ("amplitude", C.c_float), ("frequency", C.c_float), ("refractive_index", C.c_float),
                ("water_type", C.c_int), ("caustics_active", C.c_bool), ("caustics_ris_sample_count", C.c_uint32), ("caustics_domain_scale", C.c_float),
                ("multiscattering", C.c_bool), ("triangle_light_contribution", C.c_bool)]


class CloudLayer(C.Structure):
    _fields_ = [("active", C.c_bool), ("height_max", C.c_float), ("height_min", C.c_float), ("coverage", C.c_float), ("coverage_min", C.c_float),
                ("type", C.c_float), ("type_min", C.c_float), ("wind_speed", C.c_float), ("wind_angle", C.c_float)]


class Cloud(C.Structure):
    """LuminaryCloud (include/luminary_amd.h; defaults cloud.c:6-52)."""
    _fields_ = [("active", C.c_bool), ("initialized", C.c_bool), ("atmosphere_scattering", C.c_bool), ("low", CloudLayer), ("mid", CloudLayer),
                ("top", CloudLayer), ("offset_x", C.c_float), ("offset_z", C.c_float), ("density", C.c_float), ("seed", C.c_uint32),
                ("droplet_diameter", C.c_float), ("steps", C.c_uint32), ("shadow_steps", C.c_uint32), ("noise_shape_scale", C.c_float),
                ("noise_detail_scale", C.c_float), ("noise_weather_scale", C.c_float), ("mipmap_bias", C.c_float), ("octaves", C.c_uint32)]


class Particles(C.Structure):
    """LuminaryParticles (include/luminary_amd.h; defaults particles.c:6-24)."""
    _fields_ = [("active", C.c_bool), ("seed", C.c_uint32), ("count", C.c_uint32), ("albedo", RGBF), ("speed", C.c_float), ("direction_altitude", C.c_float),
                ("direction_azimuth", C.c_float), ("phase_diameter", C.c_float), ("scale", C.c_float), ("size", C.c_float), ("size_variation", C.c_float)]


class Material(C.Structure):
    _fields_ = [("id", C.c_uint32), ("base_substrate", C.c_int), ("albedo", RGBAF), ("emission", RGBF), ("emission_scale", C.c_float),
                ("roughness", C.c_float), ("roughness_clamp", C.c_float), ("refraction_index", C.c_float), ("emission_active", C.c_bool),
                ("thin_walled", C.c_bool), ("metallic", C.c_bool), ("colored_transparency", C.c_bool), ("roughness_as_smoothness", C.c_bool),
                ("normal_map_is_compressed", C.c_bool), ("bidirectional_emission", C.c_bool), ("albedo_tex", C.c_uint16),
                ("luminance_tex", C.c_uint16), ("roughness_tex", C.c_uint16), ("metallic_tex", C.c_uint16), ("normal_tex", C.c_uint16)]


class PixelQueryResult(C.Structure):
    _fields_ = [("pixel_query_is_valid", C.c_bool), ("instance_id", C.c_uint32), ("material_id", C.c_uint16), ("depth", C.c_float), ("rel_hit_pos", Vec3)]


class DeviceInfo(C.Structure):
    _fields_ = [("is_main_device", C.c_bool), ("is_unavailable", C.c_bool), ("is_enabled", C.c_bool), ("name", C.c_char * 256), ("memory_size", C.c_size_t),
                ("allocated_memory_size", C.c_size_t)]


class OutputProperties(C.Structure):
    _fields_ = [("enabled", C.c_bool), ("width", C.c_uint32), ("height", C.c_uint32)]


class OutputRequestProperties(C.Structure):
    _fields_ = [("sample_count", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("ld", C.c_size_t), ("time", C.c_float), ("sample_count", C.c_uint32)]


OUTPUT_HANDLE_INVALID = 0xFFFFFFFF


class Instance(C.Structure):
    _fields_ = [("id", C.c_uint32), ("mesh_id", C.c_uint32), ("position", Vec3), ("rotation", Vec3), ("scale", Vec3)]


class DeviceSceneView(C.Structure):
    """include/lum_core.h LumDeviceSceneView == oracle/oracle.h OracleScene."""
    _fields_ = [("num_meshes", C.c_uint32), ("num_instances", C.c_uint32), ("num_materials", C.c_uint32), ("num_lights", C.c_uint32),
                ("mesh_tri_offset", C.c_void_p), ("vertices", C.c_void_p), ("tri_tex", C.c_void_p), ("instance_mesh_ids", C.c_void_p),
                ("instance_transforms", C.c_void_p), ("materials", C.c_void_p), ("light_tree_root", C.c_void_p), ("light_tree_nodes", C.c_void_p),
                ("light_tri_handles", C.c_void_p), ("light_bvh_tris", C.c_void_p), ("num_light_tree_nodes", C.c_uint32), ("num_textures", C.c_uint32),
                ("bluenoise_2d", C.c_void_p), ("lut_conductor", C.c_void_p), ("lut_glossy", C.c_void_p), ("lut_dielectric", C.c_void_p),
                ("lut_dielectric_inv", C.c_void_p), ("texture_table", C.c_void_p), ("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("max_ray_depth", C.c_uint32),
                ("shading_mode", C.c_uint32), ("cam_pos", C.c_float * 3), ("cam_rotation", C.c_float * 4), ("cam_fov", C.c_float),
                ("cam_aperture_size", C.c_float), ("cam_object_distance", C.c_float), ("cam_scale", C.c_float), ("cam_rr_threshold", C.c_float),
                ("cam_aperture_shape", C.c_uint32), ("cam_aperture_blade_count", C.c_uint32), ("sky_mode", C.c_uint32),
                ("sky_constant_color", C.c_float * 3),
                ("sky_steps", C.c_uint32), ("sky_ozone_absorption", C.c_uint32), ("sky_geometry_offset", C.c_float * 3), ("sky_sun_strength", C.c_float),
                ("sky_base_density", C.c_float), ("sky_rayleigh_density", C.c_float), ("sky_mie_density", C.c_float), ("sky_ozone_density", C.c_float),
                ("sky_rayleigh_falloff", C.c_float), ("sky_mie_falloff", C.c_float), ("sky_ground_visibility", C.c_float),
                ("sky_ozone_layer_thickness", C.c_float), ("sky_multiscattering_factor", C.c_float), ("sky_sun_pos", C.c_float * 3),
                ("sky_mie_phase", C.c_float * 4), ("sky_lut_transmittance", C.c_void_p), ("sky_lut_multiscattering", C.c_void_p),
                ("sky_moon_pos", C.c_float * 3), ("sky_moon_tex_offset", C.c_float), ("sky_moon_albedo_tex", C.c_uint32), ("sky_moon_normal_tex", C.c_uint32),
                ("sky_stars_intensity", C.c_float), ("sky_stars_count", C.c_uint32), ("sky_stars", C.c_void_p), ("sky_stars_offsets", C.c_void_p),
                ("sky_hdri", C.c_void_p), ("sky_hdri_dim", C.c_uint32), ("sky_hdri_samples", C.c_uint32), ("sky_hdri_origin", C.c_float * 3), ("sky_aerial_perspective", C.c_uint32),
                ("fog_active", C.c_uint32), ("fog_density", C.c_float), ("fog_dist", C.c_float), ("fog_height", C.c_float), ("fog_phase", C.c_float * 4),
                ("bridge_lut", C.c_void_p), ("bridge_max_num_vertices", C.c_uint32),
                ("particles_active", C.c_uint32), ("particles_count", C.c_uint32), ("particles_scale", C.c_float), ("particles_speed", C.c_float),
                ("particles_albedo", C.c_float * 3), ("particles_direction", C.c_float * 3), ("particles_phase", C.c_float * 4),
                ("particle_vertices", C.c_void_p), ("particle_normals", C.c_void_p),
                ("ocean_active", C.c_uint32), ("ocean_height", C.c_float), ("ocean_amplitude", C.c_float), ("ocean_frequency", C.c_float),
                ("ocean_refractive_index", C.c_float), ("ocean_scattering", C.c_float * 3), ("ocean_absorption", C.c_float * 3),
                ("ocean_molecular_weight", C.c_float), ("ocean_caustics_active", C.c_uint32), ("ocean_caustics_ris_sample_count", C.c_uint32),
                ("ocean_caustics_domain_scale", C.c_float), ("ocean_multiscattering", C.c_uint32), ("ocean_triangle_light_contribution", C.c_uint32),
                ("cloud_active", C.c_uint32), ("cloud_atmosphere_scattering", C.c_uint32), ("cloud_steps", C.c_uint32), ("cloud_shadow_steps", C.c_uint32),
                ("cloud_octaves", C.c_uint32), ("cloud_seed", C.c_uint32), ("cloud_offset_x", C.c_float), ("cloud_offset_z", C.c_float),
                ("cloud_density", C.c_float), ("cloud_noise_shape_scale", C.c_float), ("cloud_noise_detail_scale", C.c_float),
                ("cloud_noise_weather_scale", C.c_float), ("cloud_phase", C.c_float * 4), ("cloud_layers", (C.c_float * 10) * 3),
                ("cloud_noise_shape", C.c_void_p), ("cloud_noise_detail", C.c_void_p), ("cloud_noise_weather", C.c_void_p)]


SKY_MODE_DEFAULT, SKY_MODE_HDRI, SKY_MODE_CONSTANT_COLOR = 0, 1, 2
SUBSTRATE_OPAQUE, SUBSTRATE_TRANSLUCENT = 0, 1

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            _build.build()
        lib = C.CDLL(LIB_PATH)
        lib.luminary_result_to_string.restype = C.c_char_p
        lib.luminary_result_to_string.argtypes = [C.c_uint64]
        lib.luminary_ext_get_core_context.restype = C.c_void_p
        lib.lumc_last_error.restype = C.c_char_p
        lib.lumc_last_error.argtypes = [C.c_void_p]
        for name in dir(lib):
            pass
        _LIB = lib
    return _LIB


def _call(name, *args):
    fn = getattr(_lib(), name)
    fn.restype = C.c_uint64
    code = fn(*args)
    if code != 0:
        raise LuminaryError(code, name)


def _core_call(ctx, name, *args):
    fn = getattr(_lib(), name)
    fn.restype = C.c_int
    if fn(C.c_void_p(ctx), *args) != 0:
        raise RuntimeError("%s failed: %s" % (name, _lib().lumc_last_error(C.c_void_p(ctx)).decode()))


class Host:
    """Mirror of LuminaryHost (reference: src/luminary/host/host.c)."""

    def __init__(self, device_mask=0xFFFFFFFF):
        _lib().luminary_init()
        self._h = C.c_void_p()
        _call("luminary_host_create", C.byref(self._h), HostCreateInfo(device_mask))

    def close(self):
        if self._h:
            _call("luminary_host_destroy", C.byref(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _path(self, s):
        p = C.c_void_p()
        _call("luminary_path_create", C.byref(p))
        _call("luminary_path_set_from_string", p, s.encode())
        return p

    def load_lum_file(self, path):
        p = self._path(path)
        try:
            _call("luminary_host_load_lum_file", self._h, p)
        finally:
            _call("luminary_path_destroy", C.byref(p))

    def load_obj_file(self, path):
        p = self._path(path)
        try:
            _call("luminary_host_load_obj_file", self._h, p)
        finally:
            _call("luminary_path_destroy", C.byref(p))

    def _get(self, name, typ):
        v = typ()
        _call("luminary_host_get_" + name, self._h, C.byref(v))
        return v

    def get_settings(self):
        return self._get("settings", RendererSettings)

    def set_settings(self, s):
        _call("luminary_host_set_settings", self._h, C.byref(s))

    def get_camera(self):
        return self._get("camera", Camera)

    def set_camera(self, c):
        _call("luminary_host_set_camera", self._h, C.byref(c))

    def get_sky(self):
        return self._get("sky", Sky)

    def set_sky(self, s):
        _call("luminary_host_set_sky", self._h, C.byref(s))

    def get_fog(self):
        return self._get("fog", Fog)

    def set_fog(self, f):
        _call("luminary_host_set_fog", self._h, C.byref(f))

    def get_ocean(self):
        return self._get("ocean", Ocean)

    def set_ocean(self, o):
        _call("luminary_host_set_ocean", self._h, C.byref(o))

    def get_cloud(self):
        return self._get("cloud", Cloud)

    def set_cloud(self, c):
        _call("luminary_host_set_cloud", self._h, C.byref(c))

    def get_particles(self):
        return self._get("particles", Particles)

    def set_particles(self, p):
        _call("luminary_host_set_particles", self._h, C.byref(p))

    def get_material(self, i):
        m = Material()
        _call("luminary_host_get_material", self._h, C.c_uint16(i), C.byref(m))
        return m

    def set_material(self, i, m):
        _call("luminary_host_set_material", self._h, C.c_uint16(i), C.byref(m))

    def new_instance(self, mesh_id, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        inst = Instance()
        _call("luminary_host_new_instance", self._h, C.byref(inst))
        inst.mesh_id = mesh_id
        inst.position = Vec3(*position)
        inst.rotation = Vec3(*rotation)
        inst.scale = Vec3(*scale)
        _call("luminary_host_set_instance", self._h, C.byref(inst))
        return inst.id

    def get_instance(self, i):
        inst = Instance()
        _call("luminary_host_get_instance", self._h, C.c_uint32(i), C.byref(inst))
        return inst

    def set_instance(self, inst):
        _call("luminary_host_set_instance", self._h, C.byref(inst))

    def counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _call("luminary_host_get_num_meshes", self._h, C.byref(a))
        _call("luminary_host_get_num_materials", self._h, C.byref(b))
        _call("luminary_host_get_num_instances", self._h, C.byref(c))
        return a.value, b.value, c.value

    def get_pixel_info(self, x, y):
        r = PixelQueryResult()
        _call("luminary_host_get_pixel_info", self._h, C.c_uint16(x), C.c_uint16(y), C.byref(r))
        return r

    # ---- output chain ----
    def set_output_properties(self, width, height, enabled=True):
        _call("luminary_host_set_output_properties", self._h, OutputProperties(enabled, width, height))

    def request_output(self, sample_count, width, height):
        h = C.c_uint32()
        _call("luminary_host_request_output", self._h, OutputRequestProperties(sample_count, width, height), C.byref(h))
        return h.value

    def try_await_output(self, promise):
        h = C.c_uint32()
        _call("luminary_host_try_await_output", self._h, C.c_uint32(promise), C.byref(h))
        return None if h.value == OUTPUT_HANDLE_INVALID else h.value

    def acquire_output(self):
        h = C.c_uint32()
        _call("luminary_host_acquire_output", self._h, C.byref(h))
        return None if h.value == OUTPUT_HANDLE_INVALID else h.value

    def get_image(self, handle):
        """(ARGB8 words [height, width] as a copy, sample_count, time) of an acquired output."""
        import numpy as np
        img = Image()
        _call("luminary_host_get_image", self._h, C.c_uint32(handle), C.byref(img))
        words = np.ctypeslib.as_array(C.cast(img.buffer, C.POINTER(C.c_uint32)), shape=(img.height, img.ld))[:, :img.width].copy()
        return words, img.sample_count, img.time

    def release_output(self, handle):
        _call("luminary_host_release_output", self._h, C.c_uint32(handle))

    def save_png(self, handle, path):
        p = self._path(path)
        try:
            _call("luminary_host_save_png", self._h, C.c_uint32(handle), p)
        finally:
            _call("luminary_path_destroy", C.byref(p))

    def start_new_render(self):
        """Restarts the accumulation and starts the library's render thread (reference host.c:406-414); poll try_await_output /
        acquire_output for images."""
        _call("luminary_host_start_new_render", self._h)

    def get_num_meshes(self):
        n = C.c_uint32()
        _call("luminary_host_get_num_meshes", self._h, C.byref(n))
        return int(n.value)

    def get_num_materials(self):
        n = C.c_uint32()
        _call("luminary_host_get_num_materials", self._h, C.byref(n))
        return int(n.value)

    def get_num_instances(self):
        n = C.c_uint32()
        _call("luminary_host_get_num_instances", self._h, C.byref(n))
        return int(n.value)

    def get_device_count(self):
        n = C.c_uint32()
        _call("luminary_host_get_device_count", self._h, C.byref(n))
        return int(n.value)

    def get_device_info(self, device_id):
        info = DeviceInfo()
        _call("luminary_host_get_device_info", self._h, C.c_uint32(device_id), C.byref(info))
        return info

    def set_device_enable(self, device_id, enable):
        _call("luminary_host_set_device_enable", self._h, C.c_uint32(device_id), C.c_bool(enable))

    def stop_render(self):
        _call("luminary_ext_stop_render", self._h)

    def is_rendering(self):
        """(render thread active, samples accumulated so far)"""
        r, n = C.c_bool(), C.c_uint32()
        _call("luminary_ext_is_rendering", self._h, C.byref(r), C.byref(n))
        return bool(r.value), int(n.value)

    def queue_workers(self):
        """[(name, current task or None, seconds)] of the library's queue workers (reference host.c:615-703)."""
        n = C.c_uint32()
        _call("luminary_host_get_num_queue_workers", self._h, C.byref(n))
        out = []
        for i in range(n.value):
            name, string, t = C.c_char_p(), C.c_char_p(), C.c_double()
            _call("luminary_host_get_queue_worker_name", self._h, C.c_uint32(i), C.byref(name))
            _call("luminary_host_get_queue_worker_string", self._h, C.c_uint32(i), C.byref(string))
            _call("luminary_host_get_queue_worker_time", self._h, C.c_uint32(i), C.byref(t))
            out.append((name.value.decode() if name.value else None, string.value.decode() if string.value else None, float(t.value)))
        return out

    # ---- additive extension ----
    def add_mesh(self, positions, material_ids, normals=None, uvs=None):
        import numpy as np
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1)
        material_ids = np.ascontiguousarray(material_ids, dtype=np.uint16).reshape(-1)
        n = material_ids.size
        assert positions.size == 9 * n
        npt = C.c_void_p(0)
        upt = C.c_void_p(0)
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1)
            npt = normals.ctypes.data_as(C.c_void_p)
        if uvs is not None:
            uvs = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1)
            upt = uvs.ctypes.data_as(C.c_void_p)
        mid = C.c_uint32()
        _call("luminary_ext_add_mesh", self._h, positions.ctypes.data_as(C.c_void_p), npt, upt, material_ids.ctypes.data_as(C.c_void_p),
              C.c_uint32(n), C.byref(mid))
        return mid.value

    def get_mesh(self, mesh_id):
        """luminary_ext_get_mesh: the host-level mesh (copies): positions [n, 9], normals [n, 9], uvs [n, 6], material ids [n]."""
        import numpy as np
        p, nrm, uv, mat = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_uint16)()
        n = C.c_uint32()
        _call("luminary_ext_get_mesh", self._h, C.c_uint32(mesh_id), C.byref(p), C.byref(nrm), C.byref(uv), C.byref(mat), C.byref(n))
        n = n.value
        if n == 0:
            return np.zeros((0, 9), np.float32), np.zeros((0, 9), np.float32), np.zeros((0, 6), np.float32), np.zeros(0, np.uint16)
        return (np.ctypeslib.as_array(p, shape=(n, 9)).copy(), np.ctypeslib.as_array(nrm, shape=(n, 9)).copy(), np.ctypeslib.as_array(uv, shape=(n, 6)).copy(),
                np.ctypeslib.as_array(mat, shape=(n,)).copy())

    def add_texture(self, rgba8, gamma=1.0):
        """rgba8: uint8 array [height, width, 4]; returns the texture id for Material.albedo_tex / roughness_tex / normal_tex."""
        import numpy as np
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 4
        i = C.c_uint16()
        _call("luminary_ext_add_texture", self._h, img.ctypes.data_as(C.c_void_p), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0]), C.c_float(gamma),
              C.byref(i))
        return i.value

    def add_material(self, m):
        i = C.c_uint16()
        _call("luminary_ext_add_material", self._h, C.byref(m), C.byref(i))
        return i.value

    def device_scene(self):
        """Scene in the device format (valid until the next scene edit). Needs no GPU."""
        p = C.POINTER(DeviceSceneView)()
        _call("luminary_ext_build_device_scene", self._h, C.byref(p))
        view = p.contents
        view._owner = self  # the buffers behind the pointers live in the host
        return view

    def render_samples(self, first_sample, num_samples, pixels=None, samples_per_pass=1):
        import numpy as np
        if pixels is None:
            _call("luminary_ext_render_samples", self._h, C.c_void_p(0), C.c_uint32(0), C.c_uint32(first_sample), C.c_uint32(num_samples),
                  C.c_uint32(samples_per_pass))
        else:
            px = np.ascontiguousarray(pixels, dtype=np.uint32)
            _call("luminary_ext_render_samples", self._h, px.ctypes.data_as(C.c_void_p), C.c_uint32(px.size), C.c_uint32(first_sample),
                  C.c_uint32(num_samples), C.c_uint32(samples_per_pass))

    def render(self, num_samples):
        """The reference's render loop for `num_samples` more sample allocations (adaptive sampling per the renderer settings)."""
        _call("luminary_ext_render", self._h, C.c_uint32(num_samples))

    def accumulators(self):
        import numpy as np
        n = C.c_uint32()
        _call("luminary_ext_get_accumulators", self._h, C.c_void_p(0), C.c_void_p(0), C.byref(n))
        fm = np.zeros(3 * n.value, dtype=np.float32)
        sm = np.zeros(n.value, dtype=np.float32)
        _call("luminary_ext_get_accumulators", self._h, fm.ctypes.data_as(C.c_void_p), sm.ctypes.data_as(C.c_void_p), C.byref(n))
        return fm.reshape(3, -1), sm

    def ray_counters(self):
        out = (C.c_uint64 * 8)()
        _call("luminary_ext_get_ray_counters", self._h, out)
        return list(out)

    def request_sky_hdri_build(self):
        """luminary_host_request_sky_hdri_build: the sky panorama of HDRI mode is baked again, from the camera's current position."""
        _call("luminary_host_request_sky_hdri_build", self._h)

    def core_sky_hdri(self):
        """The panorama the GPU core last baked for this host, [dim, dim, 4] float32."""
        import numpy as np
        lib, ctx = _lib(), C.c_void_p(self.core_context())
        dim = C.c_uint32()
        lib.lumc_sky_hdri_download.restype = C.c_int
        if lib.lumc_sky_hdri_download(ctx, C.c_void_p(0), C.byref(dim)):
            raise RuntimeError("no baked sky")
        out = np.zeros((dim.value, dim.value, 4), dtype=np.float32)
        if lib.lumc_sky_hdri_download(ctx, out.ctypes.data_as(C.c_void_p), C.byref(dim)):
            raise RuntimeError("lumc_sky_hdri_download failed")
        return out

    def core_context(self):
        ctx = _lib().luminary_ext_get_core_context(self._h)
        if not ctx:
            raise RuntimeError("no usable HIP device (libluminary_amd has no CPU path)")
        return ctx


def default_material():
    """material.c:5-29"""
    m = Material()
    m.base_substrate = SUBSTRATE_OPAQUE
    m.albedo = RGBAF(0.9, 0.9, 0.9, 0.9)
    m.emission = RGBF(0, 0, 0)
    m.emission_scale = 1.0
    m.roughness = 0.7
    m.roughness_clamp = 0.25
    m.refraction_index = 1.0
    m.normal_map_is_compressed = True
    m.albedo_tex = m.luminance_tex = m.roughness_tex = m.metallic_tex = m.normal_tex = 0xFFFF
    return m
