"""ctypes binding of the low-level core C ABI (include/lum_core.h): context, scene upload, wavefront passes, counters."""
import ctypes as C

import numpy as np

from . import DeviceSceneView, _lib

DIRTY_CONSTANTS, DIRTY_MATERIALS, DIRTY_INSTANCES, DIRTY_LIGHTS, DIRTY_MESHES, DIRTY_TEXTURES, DIRTY_PARTICLES, DIRTY_ALL = 1, 2, 4, 8, 16, 32, 64, 127
CNT_TRACE, CNT_SHADOW, CNT_LIGHT_BVH, CNT_VERTICES, CNT_NODES, CNT_TRIS, CNT_NODES_SHADOW, CNT_TRIS_SHADOW, CNT_NODES_LIGHT, CNT_TRIS_LIGHT, CNT_NODES_LDS, CNT_NODES_LDS_SHADOW, CNT_AMBIENT_DEFERRED, CNT_AMBIENT_FALLBACK = range(14)
CNT_COUNT = 16  # LUMC_CNT_COUNT
KERNELS = ("generate", "trace", "shade", "shadow", "accumulate", "light_query", "resolve", "output", "sky", "sort", "volume")


class CoreError(RuntimeError):
    pass


class OutputParams(C.Structure):
    """include/lum_core.h LumOutputParams == oracle/oracle.h OracleOutputParamsAbi."""
    _fields_ = [("src_width", C.c_uint32), ("src_height", C.c_uint32), ("dst_width", C.c_uint32), ("dst_height", C.c_uint32),
                ("inv_sample_count", C.c_float), ("exposure", C.c_float), ("tonemap", C.c_uint32), ("filter", C.c_uint32), ("dithering", C.c_uint32),
                ("purkinje", C.c_uint32), ("use_color_correction", C.c_uint32), ("passthrough", C.c_uint32), ("purkinje_kappa1", C.c_float),
                ("purkinje_kappa2", C.c_float), ("cc_h", C.c_float), ("cc_s", C.c_float), ("cc_v", C.c_float), ("film_grain", C.c_float),
                ("agx_slope", C.c_float), ("agx_power", C.c_float), ("agx_saturation", C.c_float), ("supersampling", C.c_uint32),
                ("undersampling_stage", C.c_uint32)]

    def output_planes_shape(self):
        """Shape of the display-referred planes the chain keeps: the frame >> max(undersampling stage, supersampling)."""
        uo = max(self.undersampling_stage, self.supersampling)
        return (3, self.src_height >> uo, self.src_width >> uo)


def default_output_params(width, height, sample_count, dst=None, supersampling=0, undersampling_stage=0):
    """Camera defaults of the reference (camera.c:7-66): AgX, dithering and Purkinje shift on, exposure exp(0). `width`, `height`: the
    rendered frame (output size << supersampling); `dst` defaults to the nominal output size."""
    p = OutputParams()
    p.src_width, p.src_height = width, height
    p.supersampling, p.undersampling_stage = supersampling, undersampling_stage
    p.dst_width, p.dst_height = dst if dst else (width >> supersampling, height >> supersampling)
    p.inv_sample_count = np.float32(1.0) / np.float32(sample_count)
    p.exposure = 1.0
    p.tonemap, p.filter, p.dithering, p.purkinje = 4, 0, 1, 1
    p.purkinje_kappa1, p.purkinje_kappa2 = 0.2, 0.29
    p.agx_slope = p.agx_power = p.agx_saturation = 1.0
    return p


class AdaptiveParams(C.Structure):
    """include/lum_core.h LumAdaptiveParams."""
    _fields_ = [("max_sampling_rate", C.c_uint32), ("avg_sampling_rate", C.c_uint32), ("update_interval", C.c_uint32), ("exposure", C.c_float),
                ("tone", OutputParams)]


class AdaptiveInfo(C.Structure):
    """include/lum_core.h LumAdaptiveInfo."""
    _fields_ = [("stage_id", C.c_uint32), ("executions", C.c_uint32 * 5), ("num_blocks", C.c_uint32), ("blocks_x", C.c_uint32), ("blocks_y", C.c_uint32),
                ("tasks_per_execution", C.c_uint32), ("variance_total", C.c_float), ("build_pending", C.c_uint32)]


class Core:
    def __init__(self, device=0):
        self._lib = _lib()
        self._ctx = C.c_void_p()
        self._lib.lumc_context_create.restype = C.c_int
        rc = self._lib.lumc_context_create(C.c_int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.lumc_last_error(self._ctx).decode() if self._ctx else "context allocation failed"
            if self._ctx:
                self._lib.lumc_context_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise CoreError("lumc_context_create failed (no CPU fallback exists): " + msg)
        self.num_pixels = 0

    @property
    def flavour(self):
        """'fast' or 'exact': the arithmetic flavour of the wavefront kernels this context launches (lumc_get_flavour)."""
        fn = getattr(self._lib, "lumc_get_flavour", None)
        if fn is None:
            return "exact"
        fn.restype = C.c_int
        return "fast" if fn(self._ctx) == 1 else "exact"

    # ---- multi-GPU (lumc_comm_*, lumc_frame_*) ----
    @staticmethod
    def device_count():
        return int(_lib().lumc_device_count())

    @staticmethod
    def tile_pixels(width, height, rank, world, tile=32):
        """The C ABI's tile deal (== luminary_amd.distributed.tile_pixels)."""
        lib = _lib()
        n = C.c_uint32()
        if lib.lumc_tile_pixels(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(world), C.c_uint32(tile), C.c_void_p(0), C.byref(n)):
            raise CoreError("lumc_tile_pixels: bad arguments")
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        lib.lumc_tile_pixels(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(world), C.c_uint32(tile), out.ctypes.data_as(C.c_void_p), C.byref(n))
        return out[:n.value]

    @staticmethod
    def comm_unique_id():
        """128 bytes identifying a new RCCL communicator (made on one rank, handed to all)."""
        buf = (C.c_uint8 * 128)()
        if _lib().lumc_comm_unique_id(buf):
            raise CoreError("lumc_comm_unique_id failed")
        return bytes(buf)

    def comm_init_rank(self, world, rank, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._call("lumc_comm_init_rank", C.c_int(world), C.c_int(rank), buf)

    def frame_assemble(self, frame_pixels, root=0, stream=0):
        """This rank's accumulators scattered into its [4][frame_pixels] frame and reduced (RCCL SUM) to `root`; returns the device pointer of
        this rank's frame (complete on root)."""
        ptr = C.c_void_p()
        self._call("lumc_frame_assemble", C.c_uint32(frame_pixels), C.c_int(root), C.c_void_p(stream), C.byref(ptr))
        return ptr.value

    def frame_gather(self, width, height, root=0, stream=0):
        """The frame by a gather of the ranks' own pixels (this context holds its share of the 32x32 tile deal): pack, ONE ncclGather to `root`, scatter there.
        Returns the device pointer of the assembled frame on root, None elsewhere (lumc_frame_gather)."""
        ptr = C.c_void_p()
        self._call("lumc_frame_gather", C.c_uint32(width), C.c_uint32(height), C.c_int(root), C.c_void_p(stream), C.byref(ptr))
        return ptr.value

    def frame_download(self, frame_pixels):
        fm = np.zeros(3 * frame_pixels, dtype=np.float32)
        sm = np.zeros(frame_pixels, dtype=np.float32)
        self._call("lumc_frame_download", C.c_uint32(frame_pixels), fm.ctypes.data_as(C.c_void_p), sm.ctypes.data_as(C.c_void_p))
        return fm.reshape(3, -1), sm

    @property
    def ray_sorting(self):
        fn = self._lib.lumc_get_ray_sorting
        fn.restype = C.c_int
        return int(fn(self._ctx))

    def set_ray_sorting(self, mode):
        """0 queue order, 1 closest-hit rays of depth >= 1 sorted by (origin cell, direction octant), 2 visibility rays too."""
        self._call("lumc_set_ray_sorting", C.c_int(mode))

    def set_fused_resolve(self, on):
        """Fast flavour with the ambient reuse: the resolve of a depth rides in the next depth's shading kernel (lumc_set_fused_resolve; default on)."""
        self._call("lumc_set_fused_resolve", C.c_int(int(on)))  # (2: with k_resolve_ended as a kernel of its own)

    def set_sobol_table(self, on):
        """The shading kernel reads a pass's Sobol / Owen pairs from a table written once per pass instead of hashing (lumc_set_sobol_table; default on; bit-identical)."""
        self._call("lumc_set_sobol_table", C.c_int(1 if on else 0))

    def set_ambient_reuse(self, mode):
        """-1 by flavour (fast: on, exact: off), 0 off, 1 on - in the exact flavour the reuse then only takes what it can prove and stays bit-identical (lumc_set_ambient_reuse)"""
        self._call("lumc_set_ambient_reuse", C.c_int(mode))

    @property
    def ambient_reuse(self):
        """does the next render answer ambient samples from the closest-hit rays (lumc_get_ambient_reuse: by mode, flavour and scene)?"""
        fn = self._lib.lumc_get_ambient_reuse
        fn.restype = C.c_int
        return bool(fn(self._ctx))

    def set_flavour(self, name):
        self._call("lumc_set_flavour", C.c_int({"exact": 0, "fast": 1}[name]))

    def _call(self, name, *args):
        fn = getattr(self._lib, name)
        fn.restype = C.c_int
        if fn(self._ctx, *args) != 0:
            raise CoreError("%s failed: %s" % (name, self._lib.lumc_last_error(self._ctx).decode()))

    def close(self):
        if self._ctx:
            self._lib.lumc_context_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, view):
        self._view_keepalive = view
        self._call("lumc_scene_upload", C.byref(view))
        self.width, self.height = view.width, view.height

    def update(self, view, dirty):
        """lumc_scene_update: takes over only the parts of `view` named by `dirty` (DIRTY_* below); the accumulation is not touched."""
        self._view_keepalive = view
        self._call("lumc_scene_update", C.byref(view), C.c_uint(dirty))
        self.width, self.height = view.width, view.height

    def generate_output(self, params, first_moment=None, want_float=False):
        """ARGB8 image [dst_height, dst_width] (uint32 words b | g<<8 | r<<16 | a<<24). `first_moment`: None = the context's own
        accumulators, an int = device pointer to a planar [3*P] first moment, a numpy array = host data (uploaded). Optionally also
        returns the display-referred float planes [3, src_height, src_width]."""
        out = np.zeros((params.dst_height, params.dst_width), dtype=np.uint32)
        planes = np.zeros(params.output_planes_shape(), dtype=np.float32) if want_float else None
        pl = planes.ctypes.data_as(C.c_void_p) if want_float else C.c_void_p(0)
        if isinstance(first_moment, np.ndarray):
            fm = np.ascontiguousarray(first_moment, dtype=np.float32)
            self._call("lumc_generate_output_from_host", C.byref(params), fm.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), pl)
        else:
            ptr = C.c_void_p(0) if first_moment is None else C.c_void_p(int(first_moment))
            self._call("lumc_generate_output_host", C.byref(params), ptr, out.ctypes.data_as(C.c_void_p), pl)
        return (out, planes) if want_float else out

    def download_luts(self):
        out = {"conductor": np.zeros(1024, np.uint16), "glossy": np.zeros(1024, np.uint16), "dielectric": np.zeros(32768, np.uint16),
               "dielectric_inv": np.zeros(32768, np.uint16)}
        self._call("lumc_download_luts", *[out[k].ctypes.data_as(C.c_void_p) for k in ("conductor", "glossy", "dielectric", "dielectric_inv")])
        return out

    def download_sky_luts(self):
        """The procedural sky's transmittance and multiscattering tables as flat float32 arrays (2*64*256*4 and 2*32*32*4)."""
        tm = np.zeros(2 * 64 * 256 * 4, dtype=np.float32)
        ms = np.zeros(2 * 32 * 32 * 4, dtype=np.float32)
        self._call("lumc_download_sky_luts", tm.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p))
        return tm, ms

    def cloud_noise_generate(self, seed):
        """The clouds' noise textures as the core generates them: (shape [128^3], detail [32^3], weather [1024^2]) as uint32 RGBA8."""
        shape, detail, weather = np.zeros(128 ** 3, np.uint32), np.zeros(32 ** 3, np.uint32), np.zeros(1024 ** 2, np.uint32)
        self._call("lumc_cloud_noise_generate", C.c_uint32(seed), shape.ctypes.data_as(C.c_void_p), detail.ctypes.data_as(C.c_void_p),
                   weather.ctypes.data_as(C.c_void_p))
        return shape, detail, weather

    def sky_hdri_build(self, origin, dim, samples):
        """Bakes the procedural sky seen from `origin` (world space) and returns it as [dim, dim, 4] float32."""
        o = (C.c_float * 3)(*[float(x) for x in origin])
        self._call("lumc_sky_hdri_build", o, C.c_uint32(dim), C.c_uint32(samples))
        out = np.zeros((dim, dim, 4), dtype=np.float32)
        self._call("lumc_sky_hdri_download", out.ctypes.data_as(C.c_void_p), C.c_void_p(0))
        return out

    def sky_hdri_download(self):
        """The panorama the context holds (baked at upload in sky mode HDRI, or by sky_hdri_build), [dim, dim, 4] float32."""
        dim = C.c_uint32()
        self._call("lumc_sky_hdri_download", C.c_void_p(0), C.byref(dim))
        out = np.zeros((dim.value, dim.value, 4), dtype=np.float32)
        self._call("lumc_sky_hdri_download", out.ctypes.data_as(C.c_void_p), C.byref(dim))
        return out

    def set_pixels(self, pixels=None):
        if pixels is None:
            self._call("lumc_set_pixels", C.c_void_p(0), C.c_uint32(0))
            self.num_pixels = self.width * self.height
        else:
            px = np.ascontiguousarray(pixels, dtype=np.uint32)
            self._call("lumc_set_pixels", px.ctypes.data_as(C.c_void_p), C.c_uint32(px.size))
            self.num_pixels = int(px.size)

    def render(self, first_sample, num_samples, samples_per_pass=1, first_moment_ptr=0, second_moment_ptr=0, stream=0):
        self._call("lumc_render", C.c_uint32(first_sample), C.c_uint32(num_samples), C.c_uint32(samples_per_pass), C.c_void_p(first_moment_ptr),
                   C.c_void_p(second_moment_ptr), C.c_void_p(stream))

    def synchronize(self):
        self._call("lumc_synchronize")

    def clear(self):
        self._call("lumc_clear_accumulators")

    def accumulators(self):
        fm = np.zeros(3 * self.num_pixels, dtype=np.float32)
        sm = np.zeros(self.num_pixels, dtype=np.float32)
        self._call("lumc_download_accumulators", fm.ctypes.data_as(C.c_void_p), sm.ctypes.data_as(C.c_void_p))
        return fm.reshape(3, -1), sm

    def counters(self):
        out = (C.c_uint64 * CNT_COUNT)()
        self._call("lumc_counters", out)
        return [int(x) for x in out]

    def query_counters(self):
        """counters() with the visibility entry as QUERIES answered - rays traced plus the ambient samples the closest-hit rays answered (minus those traced
        after all): what a renderer that traces every visibility ray counts, e.g. the oracle."""
        c = self.counters()
        c[CNT_SHADOW] += c[CNT_AMBIENT_DEFERRED] - c[CNT_AMBIENT_FALLBACK]
        return c

    def reset_counters(self):
        self._call("lumc_reset_counters")

    def set_profiling(self, on):
        self._call("lumc_set_profiling", C.c_int(1 if on else 0))

    def kernel_times(self):
        ms = (C.c_double * len(KERNELS))()
        n = (C.c_uint32 * len(KERNELS))()
        self._call("lumc_kernel_times", ms, n)
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KERNELS)}

    # ---- adaptive sampling (lumc_adaptive_*) ----
    def adaptive_begin(self, max_rate, avg_rate, update_interval, exposure=0.0, tone=None):
        p = AdaptiveParams()
        p.max_sampling_rate, p.avg_sampling_rate, p.update_interval, p.exposure = max_rate, avg_rate, update_interval, exposure
        if tone is not None:
            p.tone = tone
        self._call("lumc_adaptive_begin", C.byref(p))

    def adaptive_render(self, executions, stream=0):
        self._call("lumc_adaptive_render", C.c_uint32(executions), C.c_void_p(stream))

    def adaptive_info(self):
        info = AdaptiveInfo()
        self._call("lumc_adaptive_info", C.byref(info))
        return {"stage_id": int(info.stage_id), "executions": [int(x) for x in info.executions], "num_blocks": int(info.num_blocks),
                "blocks": (int(info.blocks_x), int(info.blocks_y)), "tasks_per_execution": int(info.tasks_per_execution),
                "variance_total": float(info.variance_total), "build_pending": bool(info.build_pending)}

    @staticmethod
    def adaptive_info_of(ctx):
        """adaptive_info of a context owned by somebody else (Host.core_context())."""
        info = AdaptiveInfo()
        if _lib().lumc_adaptive_info(C.c_void_p(ctx), C.byref(info)):
            raise CoreError("lumc_adaptive_info failed")
        return {"stage_id": int(info.stage_id), "executions": [int(x) for x in info.executions]}

    def adaptive_download(self):
        n = self.adaptive_info()["num_blocks"]
        counts = np.zeros(n, dtype=np.uint32)
        variance = np.zeros(n, dtype=np.float32)
        self._call("lumc_adaptive_download", counts.ctypes.data_as(C.c_void_p), variance.ctypes.data_as(C.c_void_p))
        return counts, variance

    def adaptive_set_partition(self, block_mask):
        """Blocks (4x4 pixels, row-major) this context renders: uint8 array of num_blocks entries."""
        m = np.ascontiguousarray(block_mask, dtype=np.uint8)
        assert m.size == self.adaptive_info()["num_blocks"]
        self._call("lumc_adaptive_set_partition", m.ctypes.data_as(C.c_void_p))

    def adaptive_variance(self):
        v = np.zeros(self.adaptive_info()["num_blocks"], dtype=np.float32)
        self._call("lumc_adaptive_variance", v.ctypes.data_as(C.c_void_p))
        return v

    def adaptive_build_from(self, block_variance):
        v = np.ascontiguousarray(block_variance, dtype=np.float32)
        self._call("lumc_adaptive_build_from", v.ctypes.data_as(C.c_void_p))

    def adaptive_end(self):
        self._call("lumc_adaptive_end")

    def generate_result(self, mode=0, local_error_minimization=False, uniform_samples=0, exposure=1.0, tone=None):
        """Mean-radiance / diagnostic image [3, H, W] of the context's full-frame accumulators (lumc_generate_result_host)."""
        out = np.zeros((3, self.height, self.width), dtype=np.float32)
        tp = C.byref(tone) if tone is not None else C.c_void_p(0)
        self._call("lumc_generate_result_host", C.c_uint32(mode), C.c_uint32(1 if local_error_minimization else 0), C.c_uint32(uniform_samples),
                   C.c_float(exposure), tp, out.ctypes.data_as(C.c_void_p))
        return out

    def post_bloom(self, image, full_width, full_height, blend, stage=0):
        """Bloom of a planar result image [3, H >> stage, W >> stage] (host array); returns the processed copy."""
        img = np.ascontiguousarray(image, dtype=np.float32).copy()
        self._call("lumc_post_bloom_host", img.ctypes.data_as(C.c_void_p), C.c_uint32(full_width), C.c_uint32(full_height), C.c_uint32(stage), C.c_float(blend))
        return img

    def render_undersampled(self, stage, iteration, stream=0):
        """Adds sample 0 of the pixels of one undersampling iteration to the full-frame accumulators (lumc_render_undersampled)."""
        self._call("lumc_render_undersampled", C.c_uint32(stage), C.c_uint32(iteration), C.c_void_p(stream))

    def generate_result_undersampled(self, stage, iteration):
        """The compact preview image [3, H >> stage, W >> stage] of the pixels rendered so far."""
        out = np.zeros((3, self.height >> stage, self.width >> stage), dtype=np.float32)
        self._call("lumc_generate_result_undersampled_host", C.c_uint32(stage), C.c_uint32(iteration), out.ctypes.data_as(C.c_void_p))
        return out

    def adaptive_note_first_sample(self):
        self._call("lumc_adaptive_note_first_sample", C.c_void_p(0))

    def set_bvh_builder(self, name):
        """'sah' (host, default), 'lbvh', 'ploc' or 'sah_gpu' (the GPU builders) for the next upload."""
        self._call("lumc_set_bvh_builder", C.c_int({"sah": 0, "lbvh": 1, "ploc": 2, "sah_gpu": 3}[name]))

    def comm_count(self):
        """Ranks of the RCCL communicator this context belongs to (ncclCommCount; 1 without one)."""
        fn = self._lib.lumc_comm_count
        fn.restype = C.c_int
        return int(fn(self._ctx))

    def lds_stack_bytes(self):
        """Bytes of a ray workgroup's LDS that hold traversal stack entries (build-time constant of the library)."""
        fn = self._lib.lumc_lds_stack_bytes
        fn.restype = C.c_uint
        return int(fn())

    def bvh_build_seconds(self):
        fn = self._lib.lumc_bvh_build_seconds
        fn.restype = C.c_double
        return float(fn(self._ctx))

    def bvh_meshes_by_builder(self):
        out = (C.c_uint32 * 2)()
        self._call("lumc_bvh_meshes_by_builder", out)
        return {"sah": int(out[0]), "lbvh": int(out[1])}

    def bvh_stats(self):
        out = (C.c_uint64 * 4)()
        self._call("lumc_bvh_stats", out)
        return [int(x) for x in out]

    def trace_closest_host(self, origins, dirs, ignore=None):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = origins.shape[0]
        out = np.zeros((n, 3), dtype=np.uint32)
        ip = C.c_void_p(0)
        if ignore is not None:
            ignore = np.ascontiguousarray(ignore, dtype=np.uint32)
            ip = ignore.ctypes.data_as(C.c_void_p)
        self._call("lumc_trace_closest_host", C.c_uint32(n), origins.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p), ip,
                   out.ctypes.data_as(C.c_void_p))
        return out

    def trace_closest_device(self, n, origins_ptr, dirs_ptr, ignore_ptr, out_ptr, stream=0):
        self._call("lumc_trace_closest", C.c_uint32(n), C.c_void_p(origins_ptr), C.c_void_p(dirs_ptr), C.c_void_p(ignore_ptr), C.c_void_p(out_ptr),
                   C.c_void_p(stream))


def scene_view_sizeof():
    fn = _lib().lumc_scene_view_sizeof
    fn.restype = C.c_uint32
    return fn()


def undersampling_schedule(undersampling):
    """(stage, iteration) of every render iteration of the first sample, in order (device.c:392-420, :1298-1307): stage N four
    times (iterations 3, 2, 1, 0), every finer stage three times (2, 1, 0). Empty for undersampling 0."""
    out = []
    for stage in range(undersampling, 0, -1):
        out += [(stage, it) for it in ((3, 2, 1, 0) if stage == undersampling else (2, 1, 0))]
    return out
