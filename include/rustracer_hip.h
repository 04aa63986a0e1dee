/* rustracer_hip.h — C ABI of the MI355X (gfx950) trace/shade library, librustracer_hip.so.
 *
 * This is the drop-in boundary for rustracer's per-pixel hot path.  The reference has no FFI of
 * its own (one binary crate, `mod` declarations only: src/main.rs:4-8); these entry points are
 * what a `-sys` crate would bind in order to replace the bodies of
 *   RaytracerInner::do_render      (src/raytracer.rs:167-221)
 *   RaytracerInner::debug_render   (src/raytracer.rs:353-391)
 *   Scene::new / BVH::build        (src/scene/mod.rs:82-106)
 * INTEGRATION.md shows the Rust side.  All functions return 0 on success or an RT_ERR_* code and
 * never unwind; rt_last_error() gives the text for the calling thread.
 * Plain pointers and sizes only.  Citations are path:line under the reference tree.
 */
#ifndef RUSTRACER_HIP_H
#define RUSTRACER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument (null pointer, zero size, index out of range) */
    RT_ERR_UNSUPPORTED = 2,  /* reference `unimplemented!()`: triangle lights (src/scene/objects.rs:98-100) */
    RT_ERR_HIP = 3,          /* a HIP runtime call failed (text in rt_last_error) */
    RT_ERR_INTERRUPTED = 4,  /* *interrupt became non-zero (src/raytracer.rs:184,199,204) */
    RT_ERR_LIMIT = 5         /* BVH deeper than the traversal stack, image larger than 2^31 pixels */
};

enum { RT_BSDF_DIFFUSE = 0, RT_BSDF_SPECULAR = 1 };          /* src/scene/objects.rs:10-13 */
enum { RT_MODE_DEBUG = 0, RT_MODE_FULL = 1 };                /* src/raytracer.rs:151-154 */
enum { RT_DIRECT_IMPORTANCE = 0, RT_DIRECT_HEMISPHERE = 1 }; /* src/raytracer.rs:251-286; hemisphere: README.md:9 */

/* Material (src/scene/objects.rs:16-20), 32 bytes. */
typedef struct rt_material {
    uint32_t bsdf;
    float reflectance[3];
    float emittance[3];
    uint32_t pad;
} rt_material;

/* Everything do_render reads from RenderConfig + ScreenParams + camera (src/raytracer.rs:88-134,
 * 174-177), plus the deterministic seed and the screen-tile shard this call renders. */
typedef struct rt_render_params {
    uint32_t width, height;
    float inv_w, inv_h, start, total, aspect_ratio, z;   /* ScreenParams, src/raytracer.rs:100-124 */
    float cam[3];                                        /* camera_position, src/raytracer.rs:73 */
    uint32_t spp, light_samples, bounces;                /* src/raytracer.rs:92-94 */
    uint32_t mode;                                       /* RT_MODE_* */
    uint32_t direct_estimator;                           /* RT_DIRECT_* */
    uint64_t seed;                                       /* per-(pixel,sample) stream = hash(seed, y*width+x, sample) */
    uint32_t tile_w, tile_h, shard_index, shard_count;   /* tile t = (y/tile_h)*tiles_x + x/tile_w belongs to shard t % shard_count */
} rt_render_params;

typedef struct rt_scene_info {
    uint32_t n_tri, n_sph, n_mat, n_lights;
    uint32_t n_inner_nodes;   /* device BVH: inner nodes only, 64 B each; leaves are child references */
    uint32_t bvh_depth;       /* inner levels on the longest root-to-leaf chain */
    uint64_t device_bytes;    /* flattened scene bytes resident in HBM */
    uint32_t n_wide_nodes;    /* occlusion BVH for is_occluded rays: 8-wide nodes, 80 B each (0 = not built, binary tree is used) */
    uint32_t n_wide_prims;    /* its primitive records (non-emissive objects), 48 B each */
    uint32_t wide_depth;
    uint32_t reserved;
} rt_scene_info;

/* Work counters of the last render / cast on this scene.  A ray is one
 * Scene::intersect (src/scene/mod.rs:406) or Scene::is_occluded (src/scene/mod.rs:452) call. */
typedef struct rt_counters {
    uint64_t rays_closest;    /* intersect calls */
    uint64_t rays_shadow;     /* is_occluded calls */
    uint64_t samples;         /* camera samples = owned pixels * spp */
    /* traversal work per ray class, filled only when the "stats" option is on: */
    uint64_t closest_inner, closest_tris, closest_sphs;   /* inner nodes fetched / triangles / spheres tested by intersect rays */
    uint64_t shadow_inner, shadow_tris, shadow_sphs;      /* ... by is_occluded rays */
    uint64_t closest_hits_tri, closest_hits_sph; /* shaded vertices by primitive type */
    double   kernel_ms_closest, kernel_ms_shadow, kernel_ms_shade;  /* HIP-event time per kernel class (timing enabled) */
    uint64_t launches_closest, launches_shadow;
    uint64_t shadow_wide;     /* 8-wide occlusion-BVH nodes (80 B) fetched by is_occluded rays ("stats" on) */
    double   kernel_ms_light; /* HIP-event time of k_light alone (Object::sample_l, own-object test, occluder cache), timing enabled */
    uint64_t trace_steps[8];  /* persistent traversal kernel, "stats" on: wave-wide node steps, lanes in them, primitive steps, lanes, refills, lanes,
                               * primitive records whose own box the ray misses (false candidates of the quantised tree); [7]: light-sample rays of clear beams whose
                               * term k_light added to the vertex's sum itself (no record written, "light_sum" option) */
    uint64_t shaft[4];        /* per-(vertex, light) beam test of k_light ("shaft" option), "stats" on: beams walked, beams found clear, nodes
                               * visited by the walks, light-sample rays settled as visible without a traversal */
} rt_counters;

typedef struct rt_scene rt_scene;

const char* rt_last_error(void);
/* Chooses the GPU of this process (one process per GPU).  hipSetDevice binds only the calling host thread, so the library
 * records the device here and in every scene created afterwards, and every entry point selects it again on the thread
 * that calls it: rt_render may be called from a thread other than the one that called rt_init (src/raytracer.rs:527). */
int rt_init(int device);

/* Scene::new (src/scene/mod.rs:82-106): objects = triangles then spheres, lights = objects with
 * non-black emittance, then the BVH (bvh 0.3.2 BVH::build restated) is built on the host, flattened
 * to device buffers and uploaded.  tri_pos/tri_nrm: 9 floats per triangle (p1,p2,p3 / vn1,vn2,vn3 as
 * Triangle stores them, src/scene/objects.rs:185-195); sph: (center, radius).  Inputs are copied. */
int rt_scene_create(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t n_tri,
                    const float* sph, const uint32_t* sph_mat, uint32_t n_sph,
                    const rt_material* mats, uint32_t n_mat, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_info_get(const rt_scene* scene, rt_scene_info* out);
/* Where scenes are built.  "device_build" 0/1 (default 1): once rt_init has chosen a GPU, the two trees of a scene with at least
 * "device_build_min" (default 4096) objects are built there (level-synchronous SAH build + 8-wide collapse, the host builders' trees word
 * for word), so that Raytracer::set_scene (src/raytracer.rs:514-518) does not wait for a host build; inputs the device builder declines
 * (a NaN, infinity or negative zero among the coordinates) and smaller scenes are built on the host.  Applies to scenes created afterwards.
 * "alloc_limit" bytes (default 0 = none): test hook, a pass buffer larger than this fails to allocate as if the device were full (the render
 * then halves its pass size until the buffers fit and remembers that size: rt_get_option "max_paths_fit"). */
int rt_set_global_option(const char* name, uint64_t value);
/* Build report of a scene: ms4 = wall milliseconds of the binary tree / the 8-wide tree build, then the part of each between HIP events
 * (uploads, kernels, downloads; 0 when built on the host), on_device2 = 1 where the GPU built it, note = why it did not when it was asked
 * to (any pointer may be NULL). */
int rt_scene_build_info(const rt_scene* scene, double* ms4, uint32_t* on_device2, char* note, uint32_t note_capacity);
/* Light object indices (Scene::light_indexes, src/scene/mod.rs:495). */
int rt_scene_lights(const rt_scene* scene, uint32_t* out, uint32_t capacity);
/* Host copy of the flattened BVH for inspection: per inner node 16 words
 * {l_min[3], l_max[3], r_min[3], r_max[3], left, right, 0, 0}; child bit 31 set = leaf(object index).
 * rank[object] = position of the object's leaf in the reference's DFS (right child first). */
int rt_scene_bvh(const rt_scene* scene, uint32_t* node_words, uint32_t* rank);
/* Host copy of the occlusion BVH that is_occluded rays traverse (no reference counterpart: the same
 * tree collapsed to 8 children per node with 8-bit quantised child boxes, non-emissive objects only;
 * candidates are re-tested exactly, so results equal Scene::is_occluded, src/scene/mod.rs:452-491).
 * node_words: 20 words per node; prim_records: 12 floats per record (sizes from rt_scene_info). */
int rt_scene_wide_bvh(const rt_scene* scene, uint32_t* node_words, float* prim_records);

/* do_render / debug_render (src/raytracer.rs:167, 353) for the pixels of one shard.
 * rgba: width*height*4 bytes, row-major, top row first, alpha 255 (SharedPixelBuffer,
 * src/raytracer.rs:22-51); only owned pixels are written.
 * rt_render: rgba is host memory; blocks; progress (0..100, src/raytracer.rs:79) and interrupt may be NULL.  With a progress word the
 * host buffer is refreshed after every round of passes (a pass = a chunk of pixels x a chunk of samples; a round = one pass per render
 * lane, see rt_set_option "lanes"): finished pixels appear chunk by chunk as the reference's columns do (src/raytracer.rs:203-214), and a
 * pixel whose samples are spread over several passes shows the mean of the samples done so far (= the reference's image at that many
 * spp), so a canvas polling the buffer (src/canvas.rs:179) sees the frame refine.  *interrupt is polled between rounds (the reference
 * polls per pixel, src/raytracer.rs:184-204): a render stops within one round of raising it (about 0.1 s at 1920x1080 with 871,200
 * triangles); with "lanes" = 1 it is also polled between the bounce levels of a pass, and stops within one level (a few ms).
 * rt_render_device: rgba is device memory; work is enqueued on `stream` (hipStream_t, may be NULL)
 * and the call returns without synchronising. */
int rt_render(rt_scene* scene, const rt_render_params* params, uint8_t* rgba,
              volatile uint32_t* progress, const volatile uint32_t* interrupt);
int rt_render_device(rt_scene* scene, const rt_render_params* params, void* rgba_dev, void* stream);

/* Primary-ray cast only (BASELINE config 3): object index of the winning leaf
 * (shape_index, src/scene/mod.rs:434-439; 0xFFFFFFFF = miss) and its distance, per owned pixel. */
int rt_cast_primary(rt_scene* scene, const rt_render_params* params, uint32_t* ids, float* t);
int rt_cast_primary_device(rt_scene* scene, const rt_render_params* params, void* ids_dev, void* t_dev, void* stream);

/* rt_cast_primary_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments and the same ids / t as
 * rt_cast_primary_device; the call itself makes no hipMalloc / hipFree (nor any allocating library call): the memory its
 * kernel needs beyond ids_dev and t_dev (the work counters that rt_get_counters reads afterwards, and the traversal-stack
 * overflow rows) lives in a static device array of the library, 16 MiB + 256 B per GPU, held until the process ends
 * (rt_init looks up its device address).  Limits, rejected with RT_ERR_LIMIT / RT_ERR_INVALID and a clear text: the scene
 * must already be resident on the GPU (uploading allocates: one earlier rt_cast_primary_device / rt_render_device call does
 * it), and its BVH may need at most 32 traversal-stack entries per ray (16 in LDS + 16 rows of the static array; the
 * 871,200-triangle stand-in needs 27).  The grid is capped at 1024 workgroups.  Every call on a GPU shares that GPU's
 * array: calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream do not.  The counters of the call
 * live in that array too: rt_get_counters on this scene returns them until another scene makes a no-alloc call on the same GPU, and
 * reports RT_ERR_INVALID afterwards instead of the other scene's numbers. */
int rt_cast_primary_noalloc(rt_scene* scene, const rt_render_params* params, void* ids_dev, void* t_dev, void* stream);

/* Arbitrary rays (host arrays, 3 floats each): Scene::intersect and Scene::is_occluded. */
int rt_intersect_rays(rt_scene* scene, const float* origins, const float* directions, uint32_t n, uint32_t* ids, float* t);
int rt_occluded_rays(rt_scene* scene, const float* origins, const float* directions, const float* max_dist, uint32_t n, uint8_t* out);

int rt_get_counters(const rt_scene* scene, rt_counters* out);
/* Options (none changes a result; all but the first three are tuning switches the parity tests flip):
 *   "stats" 0/1 traversal counters; "timing" 0/1 HIP-event kernel timing; "max_paths" camera samples in flight per pass and lane (default 64 Mi;
 *   halved until the pass buffers fit the device);
 *   "wide" 0/1 is_occluded rays use the 8-wide occlusion BVH (default 1); "persist" 0/1 light samples go through the
 *   k_light + persistent k_trace_p pair (default 1); "hot" 0/1 per-light occluder cache (default 1; only before the first
 *   render); "self_test" 0/1 test the object a shadow ray starts on first (default 1);
 *   "lanes" 1..4 (default 2) render lanes: rt_render / rt_render_device cut the owned pixels into
 *   that many ranges and render them concurrently on internal streams (joined to the caller's stream before the call returns its
 *   work to it), as long as each gets "lane_min_pixels" (65536) pixels; each lane has pass buffers of its own;
 *   "extend_persist" 0/1/2 Scene::intersect rays after the first bounce (2: camera rays too) go through the persistent
 *   k_extend_p, which refills finished lanes (default 1); "camera_once" 0/1 (default 1): the camera ray of a pixel is traced once per pass
 *   and its hit handed to the pixel's samples of that pass (0: one ray per sample, as the reference casts them);
 *   "shaft" 0..16 (default 6): k_light walks the occlusion BVH once per (vertex, light) with the convex hull of the vertex and
 *   the light's box, visiting at most that many nodes; a beam that can meet no primitive record settles its unblocked rays as visible
 *   and they skip k_trace_p (0 = off); "light_sum" 0/1 (default 1): k_light adds the terms of those rays to the vertex's sum itself, in sample
 *   order, and writes no record for them (0: every visible ray leaves a record for k_shade_b);
 *   "refill_min" (16), "phase_bias" (16), "shadow_waves" (6), "light_waves" (4), "inner_min" (12), "extend_refill" (24),
 *   "extend_inner" (20), "extend_grid" (8 workgroups per CU): scheduling constants;
 *   "light_grid" n (default 0 = no limit): at most n k_light workgroups per CU (8 = the resident size; one workgroup per 256 vertices otherwise);
 *   "light_token" 0/1 (default 0): the lanes' k_light launches wait for one another (measured slower); "trace_grid" (8): k_trace_p workgroups per CU
 *   in eighths of its resident capacity;
 *   "graze_guard" 0/1 (default 0): a Scene::intersect ray that tests a triangle lying within about 2^-10 rad of it stops skipping subtrees by
 *   distance for the rest of its search (measured: k_extend +14 %, no ray of the adversarial test changes: DESIGN.md section 4);
 *   "fail_after_passes" N (default 0 = never): test hook, the render reports RT_ERR_HIP once N passes are enqueued (the error exits join the
 *   render lanes like the normal one); "force_binary" m (a bit mask; default 0 = never): test hook, the light samples with
 *   (vertex index + sample index) & m == 0 go to k_binary (the exact binary-tree fallback for rays unfit for the 8-wide tree) whatever their
 *   ray looks like: 1 = every second one, 7 = every eighth, 0x80000000 = all. */
int rt_set_option(rt_scene* scene, const char* name, uint64_t value);
/* The current value of an option of rt_set_option (no reference counterpart). */
int rt_get_option(const rt_scene* scene, const char* name, uint64_t* value);

/* Screen tiles (no reference counterpart): pack the owned tiles of a frame into a
 * compact buffer of n_owned_tiles(padded) * tile_w*tile_h*bytes_per_pixel and back.  Device pointers. */
uint32_t rt_tiles_per_shard(const rt_render_params* params);   /* ceil(n_tiles / shard_count): equal for every shard */
int rt_tiles_pack(const rt_render_params* params, const void* frame_dev, void* compact_dev, uint32_t bytes_per_pixel, void* stream);
int rt_tiles_unpack(const rt_render_params* params, uint32_t shard_index, const void* compact_dev, void* frame_dev,
                    uint32_t bytes_per_pixel, void* stream);

/* Multi-GPU frame (no reference counterpart): one process per GPU, rank r renders the tiles t with t % world == r
 * (rt_render_device with shard_index = r, shard_count = world) and ONE collective completes the frame on every rank:
 * rt_gather_tiles = rt_tiles_pack -> ncclAllGather (RCCL over xGMI) -> rt_tiles_unpack of every rank's tiles, all on `stream`,
 * no host synchronisation, no allocation after the first frame of a geometry.  The communicator is made from a 128-byte RCCL
 * unique id that rank 0 creates (rt_comm_unique_id) and the caller distributes by whatever it has (a file, MPI, a torch
 * process group); rt_comm_init_rank is collective over the `world` ranks and uses the GPU chosen by rt_init.  The consumer of
 * the gathered frame is Canvas (src/canvas.rs:88-110). */
typedef struct rt_comm rt_comm;
int rt_comm_unique_id(uint8_t* id128);
int rt_comm_init_rank(const uint8_t* id128, int world, int rank, rt_comm** out);
void rt_comm_destroy(rt_comm* comm);
int rt_gather_tiles(rt_comm* comm, const rt_render_params* params, void* frame_dev, uint32_t bytes_per_pixel, void* stream);
/* rt_render for one rank of a multi-GPU job: renders this rank's shard (params->shard_index / shard_count must be the communicator's
 * rank / world), gathers every rank's tiles on the device and copies the complete frame to host memory on every rank.  Collective:
 * every rank must call it for every frame, interrupted or not.  comm == NULL is rt_render.
 * A rank whose local work fails (a HIP error, an interrupt) still enters the collective, with whatever its frame holds, and reports its
 * error afterwards, so its peers are never left waiting; a rank that cannot even get its buffers aborts the communicator
 * (ncclCommAbort: the peers' collective then fails instead of hanging) and the communicator is unusable from then on.
 * Verified on hardware at world = 1 only (the build pool has one GPU per box); the world = 2 path is exercised on CPU ranks with gloo.
 * Interrupted frames: with a progress word the caller's buffer is refreshed after every round with the running mean of the samples done so
 * far, so after RT_ERR_INTERRUPTED it holds, for the pixels being rendered when the flag was seen, the image at the samples finished by
 * then; the reference (src/raytracer.rs:184-204) leaves such pixels untouched and holds only fully sampled ones.  Without a progress word
 * only fully sampled pixels are written. */
int rt_render_comm(rt_scene* scene, const rt_render_params* params, rt_comm* comm, uint8_t* rgba,
                   volatile uint32_t* progress, const volatile uint32_t* interrupt);
/* Which RCCL the communicator runs on: "RCCL <version code> from <path of the shared object that provides ncclAllGather>" (a process that has
 * loaded PyTorch resolves to the copy PyTorch ships, a plain C++ host to /opt/rocm/lib/librccl.so: one RCCL per process either way). */
const char* rt_comm_describe(const rt_comm* comm);
/* Milliseconds between the HIP events around the last gather on this communicator (pack + ncclAllGather + unpack); waits for it. */
int rt_comm_last_gather_ms(rt_comm* comm, double* ms);

/* ---- host-side helpers mirroring the reference's scene constructors ---- */
/* ScreenParams::from_config (src/raytracer.rs:110-124); fills width..z of *params. */
void rt_screen_params(uint32_t width, uint32_t height, float fov_radians, rt_render_params* params);
/* Scene::new_dragon/new_teapot/new_specular/new_diffuse/new_triangle (src/scene/mod.rs:108-401),
 * plus "sphere_on_plane" and "specular_teapot" (BASELINE configs 1 and 5).  obj_path replaces the
 * reference's hard-coded "obj/teapot.obj" / "obj/dragon.obj" (src/scene/mod.rs:202, 240). */
int rt_scene_new_named(const char* name, const char* obj_path, rt_scene** out);
/* new_dragon with the mesh given as arrays (3 floats per vertex, 3 indices per face) instead of a file. */
int rt_scene_new_dragon_mesh(const float* vpos, const float* vnrm, uint32_t n_vert, const uint32_t* faces, uint32_t n_face, rt_scene** out);
/* The object list Scene::new received, in the rt_scene_create layout (arrays sized from rt_scene_info). */
int rt_scene_export(const rt_scene* scene, float* tri_pos, float* tri_nrm, uint32_t* tri_mat, float* sph, uint32_t* sph_mat, rt_material* mats);
/* Deterministic procedural dragon-class stand-in for the missing obj/dragon.obj: a displaced torus-knot
 * tube, nu*nv vertices, 2*nu*nv triangles.  vpos/vnrm: nu*nv*3 floats; faces: 2*nu*nv*3 indices. */
int rt_make_dragon_standin(uint32_t nu, uint32_t nv, float* vpos, float* vnrm, uint32_t* faces);

/* ---- debug probes used by the parity tests (device evaluation of the shared arithmetic) ---- */
/* op: 0 sin, 1 cos, 2 acos (double-precision definitions), 3 pow(x, y[i]), 4 to_color -> out as float,
 * 5 sin, 6 cos, 7 acos as the samplers evaluate them (single precision, rt_math.h sincos_s / acos_s) */
int rt_debug_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out);
/* The kernels' short forms of IEEE f32 1/x (op 0), sqrt (1), a/b (2) and Ray::new's normalise + reciprocals (3) against the compiler's
 * correctly rounded expansions, on the GPU: every float for 0 and 1, 2^35 / 2^32 operand sets for 2 and 3.  out2[0] = inputs whose results
 * differ in any bit (must be 0), out2[1] = one such input. */
int rt_debug_exact(uint32_t op, uint64_t* out2);
int rt_debug_rng(uint64_t seed, uint32_t pixel_index, uint32_t sample, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif
