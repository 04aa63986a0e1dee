"""ctypes binding of librustracer_hip.so (C ABI: include/rustracer_hip.h).

There is no CPU fallback: if the library is missing, importing a symbol from it raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RUSTRACER_HIP_LIB") or os.path.join(_HERE, "librustracer_hip.so")   # the override serves A/B builds (tools/build_variant.sh)

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_HIP, RT_ERR_INTERRUPTED, RT_ERR_LIMIT = range(6)
RT_BSDF_DIFFUSE, RT_BSDF_SPECULAR = 0, 1
RT_MODE_DEBUG, RT_MODE_FULL = 0, 1
RT_DIRECT_IMPORTANCE, RT_DIRECT_HEMISPHERE = 0, 1
RT_MISS = 0xFFFFFFFF


class RtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("rustracer_hip error %d: %s" % (code, text))
        self.code = code


class Material(C.Structure):
    _fields_ = [("bsdf", C.c_uint32), ("reflectance", C.c_float * 3), ("emittance", C.c_float * 3), ("pad", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("inv_w", C.c_float), ("inv_h", C.c_float), ("start", C.c_float), ("total", C.c_float),
        ("aspect_ratio", C.c_float), ("z", C.c_float),
        ("cam", C.c_float * 3),
        ("spp", C.c_uint32), ("light_samples", C.c_uint32), ("bounces", C.c_uint32),
        ("mode", C.c_uint32), ("direct_estimator", C.c_uint32),
        ("seed", C.c_uint64),
        ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
    ]

    def copy(self):
        q = RenderParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(RenderParams))
        return q


class SceneInfo(C.Structure):
    _fields_ = [("n_tri", C.c_uint32), ("n_sph", C.c_uint32), ("n_mat", C.c_uint32), ("n_lights", C.c_uint32),
                ("n_inner_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("device_bytes", C.c_uint64),
                ("n_wide_nodes", C.c_uint32), ("n_wide_prims", C.c_uint32), ("wide_depth", C.c_uint32), ("reserved", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("samples", C.c_uint64),
                ("closest_inner", C.c_uint64), ("closest_tris", C.c_uint64), ("closest_sphs", C.c_uint64),
                ("shadow_inner", C.c_uint64), ("shadow_tris", C.c_uint64), ("shadow_sphs", C.c_uint64),
                ("closest_hits_tri", C.c_uint64), ("closest_hits_sph", C.c_uint64),
                ("kernel_ms_closest", C.c_double), ("kernel_ms_shadow", C.c_double), ("kernel_ms_shade", C.c_double),
                ("launches_closest", C.c_uint64), ("launches_shadow", C.c_uint64), ("shadow_wide", C.c_uint64), ("kernel_ms_light", C.c_double), ("trace_steps", C.c_uint64 * 8), ("shaft", C.c_uint64 * 4)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n in ("trace_steps", "shaft") else getattr(self, n)) for n, _ in self._fields_}


# every symbol include/rustracer_hip.h declares
EXPORTS = [
    "rt_last_error", "rt_init", "rt_scene_create", "rt_scene_destroy", "rt_scene_info_get", "rt_set_global_option", "rt_scene_build_info", "rt_scene_lights", "rt_scene_bvh", "rt_scene_wide_bvh",
    "rt_render", "rt_render_device", "rt_cast_primary", "rt_cast_primary_device", "rt_cast_primary_noalloc", "rt_intersect_rays", "rt_occluded_rays",
    "rt_get_counters", "rt_set_option", "rt_get_option", "rt_tiles_per_shard", "rt_tiles_pack", "rt_tiles_unpack", "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_destroy", "rt_gather_tiles", "rt_render_comm", "rt_comm_describe", "rt_comm_last_gather_ms", "rt_screen_params",
    "rt_scene_new_named", "rt_scene_new_dragon_mesh", "rt_scene_export", "rt_make_dragon_standin", "rt_debug_math", "rt_debug_rng", "rt_debug_exact",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: build it with `make -C rustracer_amd/csrc` (or __graft_entry__.build()); "
                           "there is no CPU fallback" % LIB_PATH)
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so).  One process must use one runtime:
    # load torch's first so that this library binds to the same one and can share device memory,
    # streams and RCCL communicators with torch tensors.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.rt_last_error.restype = C.c_char_p
    L.rt_init.argtypes = [C.c_int]
    L.rt_scene_create.argtypes = [vp, vp, vp, u32, vp, vp, u32, vp, u32, C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_info_get.argtypes = [vp, C.POINTER(SceneInfo)]
    L.rt_set_global_option.argtypes = [C.c_char_p, u64]
    L.rt_scene_build_info.argtypes = [vp, vp, vp, C.c_char_p, u32]
    L.rt_scene_lights.argtypes = [vp, vp, u32]
    L.rt_scene_bvh.argtypes = [vp, vp, vp]
    L.rt_scene_wide_bvh.argtypes = [vp, vp, vp]
    L.rt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp]
    L.rt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    L.rt_cast_primary.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    L.rt_cast_primary_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp]
    L.rt_cast_primary_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp]
    L.rt_intersect_rays.argtypes = [vp, vp, vp, u32, vp, vp]
    L.rt_occluded_rays.argtypes = [vp, vp, vp, vp, u32, vp]
    L.rt_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.rt_set_option.argtypes = [vp, C.c_char_p, u64]
    L.rt_get_option.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    L.rt_tiles_per_shard.argtypes = [C.POINTER(RenderParams)]
    L.rt_tiles_per_shard.restype = u32
    L.rt_tiles_pack.argtypes = [C.POINTER(RenderParams), vp, vp, u32, vp]
    L.rt_tiles_unpack.argtypes = [C.POINTER(RenderParams), u32, vp, vp, u32, vp]
    L.rt_comm_unique_id.argtypes = [vp]
    L.rt_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.rt_comm_destroy.argtypes = [vp]
    L.rt_comm_destroy.restype = None
    L.rt_gather_tiles.argtypes = [vp, C.POINTER(RenderParams), vp, u32, vp]
    L.rt_render_comm.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp]
    L.rt_comm_describe.argtypes = [vp]
    L.rt_comm_describe.restype = C.c_char_p
    L.rt_comm_last_gather_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.rt_screen_params.argtypes = [u32, u32, f32, C.POINTER(RenderParams)]
    L.rt_screen_params.restype = None
    L.rt_scene_new_named.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.rt_scene_new_dragon_mesh.argtypes = [vp, vp, u32, vp, u32, C.POINTER(vp)]
    L.rt_scene_export.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.rt_make_dragon_standin.argtypes = [u32, u32, vp, vp, vp]
    L.rt_debug_math.argtypes = [u32, vp, vp, u32, vp]
    L.rt_debug_exact.argtypes = [u32, vp]
    L.rt_debug_rng.argtypes = [u64, u32, u32, u32, vp]
    _lib = L
    return L


def kernel_source_hash():
    """sha256 over the sources the device code is built from, comments and white space left out: stamps profiles (tools/profile.sh) so
    that bench.py can tell whether committed PMC figures were measured on the kernels it is running."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("rt_kernels.h", "rt_math.h", "rt_api.hip", "host_scene.h", "Makefile"):
        with open(os.path.join(_HERE, "csrc", name), "r", errors="replace") as f:
            text = f.read()
        if name != "Makefile":
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # block comments
            text = re.sub(r"//[^\n]*", " ", text)                   # line comments (no string literal in these sources holds "//")
        else:
            text = re.sub(r"#[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def ptr(a):
    """numpy array -> void* (None stays NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)
