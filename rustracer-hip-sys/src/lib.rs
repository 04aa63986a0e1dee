//! Raw bindings to `librustracer_hip.so` — one `extern "C"` item per declaration of `include/rustracer_hip.h`,
//! in the header's order.  `tests/test_sys_crate.py` parses this file and the header and fails on any difference
//! in names, argument counts, argument kinds or struct sizes (there is no Rust toolchain in the build image, so
//! the crate ships as source; the C++ host in `rustracer_amd/host/` is the tested caller of the same ABI).
//!
//! Where these are called from in the reference (INTEGRATION.md has the patch):
//! `Scene::new` -> `rt_scene_create`; `RaytracerInner::do_render` / `debug_render` -> `rt_render`;
//! the hit-id mode -> `rt_cast_primary`.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = 1;
pub const RT_ERR_UNSUPPORTED: c_int = 2;
pub const RT_ERR_HIP: c_int = 3;
pub const RT_ERR_INTERRUPTED: c_int = 4;
pub const RT_ERR_LIMIT: c_int = 5;

pub const RT_BSDF_DIFFUSE: u32 = 0;
pub const RT_BSDF_SPECULAR: u32 = 1;
pub const RT_MODE_DEBUG: u32 = 0;
pub const RT_MODE_FULL: u32 = 1;
pub const RT_DIRECT_IMPORTANCE: u32 = 0;
pub const RT_DIRECT_HEMISPHERE: u32 = 1;

/// `Material` (src/scene/objects.rs:16-20), 32 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_material {
    pub bsdf: u32,
    pub reflectance: [f32; 3],
    pub emittance: [f32; 3],
    pub pad: u32,
}

/// `RenderConfig` + `ScreenParams` + camera (src/raytracer.rs:88-134), the seed and the screen-tile shard; 88 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub inv_w: f32,
    pub inv_h: f32,
    pub start: f32,
    pub total: f32,
    pub aspect_ratio: f32,
    pub z: f32,
    pub cam: [f32; 3],
    pub spp: u32,
    pub light_samples: u32,
    pub bounces: u32,
    pub mode: u32,
    pub direct_estimator: u32,
    pub seed: u64,
    pub tile_w: u32,
    pub tile_h: u32,
    pub shard_index: u32,
    pub shard_count: u32,
}

/// 48 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_scene_info {
    pub n_tri: u32,
    pub n_sph: u32,
    pub n_mat: u32,
    pub n_lights: u32,
    pub n_inner_nodes: u32,
    pub bvh_depth: u32,
    pub device_bytes: u64,
    pub n_wide_nodes: u32,
    pub n_wide_prims: u32,
    pub wide_depth: u32,
    pub reserved: u32,
}

/// 208 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_counters {
    pub rays_closest: u64,
    pub rays_shadow: u64,
    pub samples: u64,
    pub closest_inner: u64,
    pub closest_tris: u64,
    pub closest_sphs: u64,
    pub shadow_inner: u64,
    pub shadow_tris: u64,
    pub shadow_sphs: u64,
    pub closest_hits_tri: u64,
    pub closest_hits_sph: u64,
    pub kernel_ms_closest: f64,
    pub kernel_ms_shadow: f64,
    pub kernel_ms_shade: f64,
    pub launches_closest: u64,
    pub launches_shadow: u64,
    pub shadow_wide: u64,
    pub kernel_ms_light: f64,
    pub trace_steps: [u64; 8],
    pub shaft: [u64; 4],
}

/// Opaque scene handle (`Scene`, src/scene/mod.rs:19-24, plus its device copy).
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

/// Opaque RCCL communicator of a multi-GPU job.
#[repr(C)]
pub struct rt_comm {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_init(device: c_int) -> c_int;

    pub fn rt_scene_create(tri_pos: *const f32, tri_nrm: *const f32, tri_mat: *const u32, n_tri: u32, sph: *const f32, sph_mat: *const u32, n_sph: u32, mats: *const rt_material, n_mat: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_info_get(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_set_global_option(name: *const c_char, value: u64) -> c_int;
    pub fn rt_scene_build_info(scene: *const rt_scene, ms4: *mut f64, on_device2: *mut u32, note: *mut c_char, note_capacity: u32) -> c_int;
    pub fn rt_scene_lights(scene: *const rt_scene, out: *mut u32, capacity: u32) -> c_int;
    pub fn rt_scene_bvh(scene: *const rt_scene, node_words: *mut u32, rank: *mut u32) -> c_int;
    pub fn rt_scene_wide_bvh(scene: *const rt_scene, node_words: *mut u32, prim_records: *mut f32) -> c_int;

    pub fn rt_render(scene: *mut rt_scene, params: *const rt_render_params, rgba: *mut u8, progress: *mut u32, interrupt: *const u32) -> c_int;
    pub fn rt_render_device(scene: *mut rt_scene, params: *const rt_render_params, rgba_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_cast_primary(scene: *mut rt_scene, params: *const rt_render_params, ids: *mut u32, t: *mut f32) -> c_int;
    pub fn rt_cast_primary_device(scene: *mut rt_scene, params: *const rt_render_params, ids_dev: *mut c_void, t_dev: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_cast_primary_noalloc(scene: *mut rt_scene, params: *const rt_render_params, ids_dev: *mut c_void, t_dev: *mut c_void, stream: *mut c_void) -> c_int;

    pub fn rt_intersect_rays(scene: *mut rt_scene, origins: *const f32, directions: *const f32, n: u32, ids: *mut u32, t: *mut f32) -> c_int;
    pub fn rt_occluded_rays(scene: *mut rt_scene, origins: *const f32, directions: *const f32, max_dist: *const f32, n: u32, out: *mut u8) -> c_int;

    pub fn rt_get_counters(scene: *const rt_scene, out: *mut rt_counters) -> c_int;
    pub fn rt_set_option(scene: *mut rt_scene, name: *const c_char, value: u64) -> c_int;
    pub fn rt_get_option(scene: *const rt_scene, name: *const c_char, value: *mut u64) -> c_int;

    pub fn rt_tiles_per_shard(params: *const rt_render_params) -> u32;
    pub fn rt_tiles_pack(params: *const rt_render_params, frame_dev: *const c_void, compact_dev: *mut c_void, bytes_per_pixel: u32, stream: *mut c_void) -> c_int;
    pub fn rt_tiles_unpack(params: *const rt_render_params, shard_index: u32, compact_dev: *const c_void, frame_dev: *mut c_void, bytes_per_pixel: u32, stream: *mut c_void) -> c_int;

    pub fn rt_comm_unique_id(id128: *mut u8) -> c_int;
    pub fn rt_comm_init_rank(id128: *const u8, world: c_int, rank: c_int, out: *mut *mut rt_comm) -> c_int;
    pub fn rt_comm_destroy(comm: *mut rt_comm);
    pub fn rt_gather_tiles(comm: *mut rt_comm, params: *const rt_render_params, frame_dev: *mut c_void, bytes_per_pixel: u32, stream: *mut c_void) -> c_int;
    pub fn rt_render_comm(scene: *mut rt_scene, params: *const rt_render_params, comm: *mut rt_comm, rgba: *mut u8, progress: *mut u32, interrupt: *const u32) -> c_int;
    pub fn rt_comm_describe(comm: *const rt_comm) -> *const c_char;
    pub fn rt_comm_last_gather_ms(comm: *mut rt_comm, ms: *mut f64) -> c_int;

    pub fn rt_screen_params(width: u32, height: u32, fov_radians: f32, params: *mut rt_render_params);
    pub fn rt_scene_new_named(name: *const c_char, obj_path: *const c_char, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_new_dragon_mesh(vpos: *const f32, vnrm: *const f32, n_vert: u32, faces: *const u32, n_face: u32, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_export(scene: *const rt_scene, tri_pos: *mut f32, tri_nrm: *mut f32, tri_mat: *mut u32, sph: *mut f32, sph_mat: *mut u32, mats: *mut rt_material) -> c_int;
    pub fn rt_make_dragon_standin(nu: u32, nv: u32, vpos: *mut f32, vnrm: *mut f32, faces: *mut u32) -> c_int;

    pub fn rt_debug_math(op: u32, x: *const f32, y: *const f32, n: u32, out: *mut f32) -> c_int;
    pub fn rt_debug_exact(op: u32, out2: *mut u64) -> c_int;
    pub fn rt_debug_rng(seed: u64, pixel_index: u32, sample: u32, n: u32, out: *mut f32) -> c_int;
}

#[cfg(test)]
mod layout {
    use super::*;
    use std::mem::size_of;

    #[test]
    fn struct_sizes_match_the_header() {
        assert_eq!(size_of::<rt_material>(), 32);
        assert_eq!(size_of::<rt_render_params>(), 88);
        assert_eq!(size_of::<rt_scene_info>(), 48);
        assert_eq!(size_of::<rt_counters>(), 240);
    }
}
