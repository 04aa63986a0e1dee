// rt_kernels.h — gfx950 kernels of the wavefront path tracer (device side of the hot path).
//
// One pass renders `npaths` camera samples in lock step.  Per bounce level k (bounces_left =
// bounces - k, src/raytracer.rs:330-350) the host enqueues
//   k_camera   (level 0) the camera ray of every pixel of the pass, traced once and handed to the pixel's samples of the pass: ray queue,
//              RNG streams and hits in one kernel (render_helper, raytracer.rs:224-235; k_generate + k_extend with option "camera_once" 0)
//   k_extend_p closest-hit BVH traversal over the live ray queue after the first bounce (Scene::intersect, mod.rs:406-447), in persistent
//              waves that hand new rays to finished lanes (k_extend: one ray per thread, option "extend_persist" 0)
//   k_shade_a  hit point + shading normal, terminates misses / depth 0 (mod.rs:58-67, raytracer.rs:339)
//   k_light    one thread per vertex over its lights x light samples: Object::sample_l's point and ray (objects.rs:96-108) and the
//              is_occluded tests that need no traversal (the object the ray starts on, the light's occluder cache); the rays it
//              cannot settle are packed at the front of their row of 64
//   k_binary   those of them with a non-finite 1/d, on the binary tree     (mod.rs:452-491)
//   k_trace_p  the others, on the 8-wide occlusion BVH, in persistent waves with lane refill
//   k_shade_b  pdf, cosine and the ordered sum of the visible light samples, emission, Russian roulette, BSDF sample, and
//              wave64 ballot compaction of the surviving paths into the next ray queue
//                                                                      (objects.rs:109-115, raytracer.rs:251-326)
//   (k_shadow  one thread per (vertex, light sample), sampling and traversal in one kernel: the hemisphere estimator, and the
//              importance estimator when the occlusion BVH is switched off)
// and finally k_resolve adds the finished samples to their pixels in sample order
// (raytracer.rs:236-239) and tone-maps (common.rs:46-58, raytracer.rs:43-51).
//
// The integrator is evaluated in exactly the association order of the reference's recursion: each
// path keeps, per level, the direct term and the (reflected, cos, pdf) factors in HBM and folds
// them from the last vertex back to the camera when it terminates (fold_path).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_math.h"
#include "host_scene.h"

namespace rt {

#define RT_BLOCK 256
#define RT_STACK_LDS 16          // traversal stack entries per lane staged in LDS (16 KiB per workgroup)
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 8        // k_shadow: waves per SIMD the register budget is sized for
#endif
#define RT_MISS 0xFFFFFFFFu
#define RT_EPS 0.0000001f        // src/common.rs:6
#define RT_PI 3.14159265358979323846f
#define RT_ROULETTE 0.7f         // src/raytracer.rs:11
#define RT_INFO_BLOCKER 0x80000000u   // object info word (tris[3 i + 1].w, sph_aux.y): the object is not emissive, i.e. it blocks is_occluded rays
#define RT_INFO_FLAT 0x40000000u      // triangle whose box is flat on an axis (a wall): its vertices walk a beam in k_light
#define RT_INFO_MAT 0x3FFFFFFFu       // material index
#define RT_HOT 2                // occluders tested in k_light before any traversal, per light
#define RT_HOT_LIGHTS 4         // lights whose occluder cache travels in k_light's kernel arguments (further lights have none)

struct DevScene {
    const float4* nodes;     // 4 x float4 per inner node (64 B)
    const float4* tris;      // 3 x float4 per triangle (48 B): (p1, rank) (p2, info) (p3, 2^-10 |e1| |e2|: grazes_triangle)
    const float4* tri_nrm;   // 3 x float4 per triangle (48 B): vn1, vn2, vn3
    const float4* sph;       // (center, radius)
    const uint2* sph_aux;    // (rank, info)
    const float4* mats;      // 2 x float4 per material: (bsdf bits, reflectance) (emittance, 0)
    const uint32_t* lights;  // object indices
    uint32_t n_tri, n_sph, n_lights, root;
    uint32_t* ovf;           // traversal-stack overflow rows beyond RT_STACK_LDS: [row][global thread]
    uint32_t ovf_stride;
    uint32_t self_test;      // shadow rays test the object they start on first
    uint32_t inner_min;      // traverse(): leaf tests start when fewer lanes than this are still at inner nodes
    const float4* wnodes;    // occlusion BVH (host_scene.h WideBVH): 5 x float4 per 8-wide node (80 B)
    const float4* wprims;    // 3 x float4 per blocker primitive record (48 B), in node order
    const uint32_t* wrec;    // primitive record of each object in wprims (0xFFFFFFFF: emissive, not in the tree)
    const uint32_t* wsib;    // per primitive record: the record of a coplanar neighbour of an axis-flat triangle (bits 0..22, RT_W_NO_REC: none), its flat axis (bits 23..24)
    uint32_t n_wnodes;       // 0: shadow rays take the binary tree
    uint32_t refill_min;     // k_trace_p: idle lanes that trigger a refill of the wave
    uint32_t extend_refill_min;   // k_extend_p: the same for closest-hit rays
    uint32_t extend_inner_min;    // k_extend_p: lanes at an inner node below which the wave turns to the waiting leaf tests (sc.inner_min of traverse())
    uint32_t phase_bias;     // k_trace_p: node step when 4 * lanes_at_nodes >= phase_bias * lanes_at_primitives
    uint32_t graze_guard;    // Scene::intersect kernels: a ray that meets a triangle almost in its plane is searched unpruned from then on (1; 0 only for the tests)
};

// Occluder cache of k_light, passed by value as a kernel argument so that its fields arrive through scalar loads:
// per light the RT_HOT largest blockers next to it, with everything that does not depend on the ray worked out on the host by
// the same single-precision operations the kernels would use (edges, own box).
struct HotRec {
    uint32_t kind;            // 0 triangle, 1 sphere, 2 none
    uint32_t rec;             // its record in the occlusion BVH (a ray that starts on it is not tested against it twice)
    float a[3];               // p1 / centre
    float e1[3], e2[3];       // triangle edges / (radius, -, -), -
    float mn[3], mx[3];       // own box (Bounded, objects.rs:293-316)
};
struct HotTable {
    HotRec rec[RT_HOT_LIGHTS][RT_HOT];
};
static_assert(RT_HOT == 2, "k_light's shaft skips exactly the two cache records of a light");

struct TravStats {
    uint32_t inner, tris, sphs, wide;
    uint32_t box_miss;   // k_trace_p: primitive records whose own (exact) box the ray misses: false candidates of the quantised tree
};

struct PassCtx {
    rt_render_params P;
    uint64_t seed_mixed;
    uint32_t tiles_x, n_tiles;
    uint32_t pix0, npix, s0, spc;    // this pass: owned-pixel range [pix0, pix0+npix), samples [s0, s0+spc)
    uint32_t npaths;                 // npix * spc
    uint32_t n_groups, nlL;          // light-sample groups (lights, or 1 for the hemisphere estimator) and groups * light_samples
    uint32_t* counts;                // live rays per level, [bounces + 2]
    uint32_t* vcounts;               // path vertices to light per level (rays that hit something, depth left), [bounces + 2]
    uint32_t* bcounts;               // light-sample rays per level that must take the binary tree (k_binary), [bounces + 2]
    uint32_t* blist;                 // their slots
    float4* vS;                      // per vertex: (path slot bits, incoming ray direction wo)
    float4* ray_o[2];                // (origin, slot bits)   ping-pong by level parity
    float4* ray_d[2];                // (direction, 0)
    uint32_t* hit_id;
    float* hit_t;
    float4* vP;                      // per vertex, compacted by k_shade_a: (hit point, material index bits)
    float4* vN;                      // (shading normal, bits: hit object id + 1)
    uint64_t* qrng[2];               // per queued ray: the path's RNG state (it travels with the ray and the vertex, so that no kernel gathers it by path slot)
    uint64_t* vR;                    // per vertex: the same, copied by k_shade_a; k_light draws the light samples from it, k_shade_b jumps over them
    float4* contrib;                 // per (vertex, light sample): (rgb, visible); only when !in_wave
    float4* direct;                  // per vertex: light-sample sum (in_wave); persist: [light group][vertex, stride vstride] (what k_light summed itself, records left ? 1 : 0)
    uint32_t vstride;                // vertices the workspace holds (stride of `direct` per light group)
    uint32_t light_sum;              // k_light adds the terms of the samples it settles as visible itself (option "light_sum")
    uint32_t force_binary;           // test hook (option "force_binary", a bit mask m != 0): the light samples with (vertex + sample) & m == 0 are handed to k_binary as if their rays were unfit for the 8-wide tree
    uint32_t in_wave;                // nlL divides 64: k_shadow sums a vertex's light samples inside the wave
    unsigned long long* tmask;       // per row (64 vertices x one light sample), k_light: the vertices whose ray is not yet settled as blocked
    unsigned long long* tmask_t;     // per row: those of them k_trace_p still has to trace (tmask minus the rays a clear shaft settles as visible)
    uint32_t shaft_steps;            // k_light: node budget of the shaft walk per (vertex, light); 0 = no shafts
    float4* lray;                    // those rays, packed at the front of their row in lane order: (wi, distance to the light sample; its sign bit is set
                                     // by k_trace_p / k_binary when the ray is occluded: all k_shade_b reads of a ray)
    float4* linv;                    // same places, for the traversal: (1/d of the box test, word: RT_LR_* | occlusion-BVH record of the object the ray starts on)
    uint32_t persist;                // light samples go through k_light + k_trace_p; k_shade_b sums the rays left unblocked in pc.lray
    float4* stack;                   // [slot][level][2]: (direct rgb, cos), (reflected rgb, pdf): the 32 B of a level in one sector, the levels of a path
                                     // next to one another (fold_path reads them back in a row; indexed by slot because the queues reorder the paths)
    uint32_t stack_depth;            // levels per slot (= bounces)
    float4* radiance;                // per path slot: finished sample
    float4* accum;                   // per owned pixel, across sample chunks
    const ulonglong2* jump;          // LCG jump-ahead (mul, add) by 2*j draws, j in [0, nlL]
    unsigned long long* stats;       // device counters, see ST_*
    uint8_t* rgba;                   // full frame
    uint32_t preview;                // k_resolve also writes the running mean after every chunk of samples but the last
};

// traversal work is counted per ray class: [ST_C_*] closest-hit rays, [ST_S_*] shadow (any-hit) rays
enum { ST_RAYS_CLOSEST = 0, ST_RAYS_SHADOW, ST_C_INNER, ST_C_TRIS, ST_C_SPHS, ST_S_INNER, ST_S_TRIS, ST_S_SPHS, ST_HITS_TRI, ST_HITS_SPH, ST_S_WIDE,
       ST_P_NODE_STEPS, ST_P_NODE_LANES, ST_P_PRIM_STEPS, ST_P_PRIM_LANES, ST_P_REFILLS, ST_P_REFILL_LANES, ST_P_BOX_MISS, ST_L_SUMMED,
       ST_SHAFT_TRIED, ST_SHAFT_CLEAR, ST_SHAFT_STEPS, ST_SHAFT_RAYS, ST_COUNT };

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ vec3 xyz(float4 v) { return mk(v.x, v.y, v.z); }

// Owned pixel `op` (tile-major, row-major inside the tile) -> image coordinates.
__device__ __forceinline__ bool own_pixel_xy(const rt_render_params& P, uint32_t tiles_x, uint32_t op, uint32_t* x, uint32_t* y) {
    const uint32_t tp = P.tile_w * P.tile_h;
    const uint32_t k = op / tp, within = op - k * tp;
    const uint32_t t = P.shard_index + k * P.shard_count;
    const uint32_t ty = within / P.tile_w, tx = within - ty * P.tile_w;
    *x = (t % tiles_x) * P.tile_w + tx;
    *y = (t / tiles_x) * P.tile_h + ty;
    return *x < P.width && *y < P.height;
}

// ScreenParams::screen_to_world, src/raytracer.rs:127-133
__device__ __forceinline__ vec3 screen_to_world(const rt_render_params& P, uint32_t i, uint32_t j) {
    const float iw = ((float)i + 0.5f) * P.inv_w;
    const float jh = ((float)j + 0.5f) * P.inv_h;
    const float xi = (P.start + iw * P.total) * P.aspect_ratio;
    const float yi = -P.start - jh * P.total;
    return unit3(mk(xi, yi, -P.z));
}

// ---- IEEE f32 reciprocal, division and square root without the compiler's range handling ------------------------------
// hipcc expands `a / b` and `sqrtf(x)` (correctly rounded, the default) into 11 and 17 instructions: v_div_scale x2, v_rcp,
// five fma, v_div_fmas, v_div_fixup; and v_sqrt, two +-1 ulp candidates with their fma residuals, plus rescaling for tiny
// inputs.  Most of that only serves operands near the ends of the exponent range.  For operands in [2^-96, 2^96] the forms
// below give the SAME bits with 5, 3 (per quotient, sharing the reciprocal) and 9 instructions; anything else (zero,
// denormal, huge, inf, NaN) takes the compiler's expansion, so the result is the IEEE one for every input.
//   rcp_mid:  y0 = v_rcp_f32(x) (1 ulp), two Newton steps with exact fma residuals.  Equal to 1.0f / x on every float of the
//             range, checked exhaustively on the GPU (tests/test_gpu_parity.py::test_fast_ieee_forms_are_exact).
//   div_mid:  q = a y, r = a - b q (exact, fma), q' = q + r y with y = RN(1 / b): the correctly rounded quotient (Markstein's
//             theorem), for a and b in [2^-60, 2^60] so that q and r stay normal; checked on 2^35 operand pairs incl. every
//             mantissa of b.
//   sqrt_mid: the compiler's own algorithm minus the rescaling: s = v_sqrt_f32(x), then s - 1 ulp if x - (s - 1ulp) s <= 0,
//             s + 1 ulp if x - (s + 1ulp) s > 0.  Checked exhaustively.
// (mid_range, div_range, rcp_mid, div_mid, sqrt_mid: rt_math.h, where acos_s uses them too)
__device__ __forceinline__ bool in_2pm20(float x) { const float a = fabsf(x); return a >= 0x1p-20f && a <= 0x1p+20f; }
__device__ __forceinline__ bool in_2pm18_20(float x) { const float a = fabsf(x); return a >= 0x1p-18f && a <= 0x1p+20f; }
__device__ __forceinline__ float rcp_ieee(float x) {          // == 1.0f / x
    if (mid_range(x)) return rcp_mid(x);
    return 1.0f / x;
}
__device__ __forceinline__ float div_ieee(float a, float b) { // == a / b
    if (div_range(a) && div_range(b)) return div_mid(a, b, rcp_mid(b));
    return a / b;
}
__device__ __forceinline__ float sqrt_ieee(float x) {         // == sqrtf(x)
    if (x >= 0x1p-96f && x <= 0x1p+96f) return sqrt_mid(x);
    return sqrtf(x);
}
// unit3 (v / |v|, three divisions by one norm) and the three reciprocals of the result: Ray::new + bvh Ray::new (mod.rs:506-512)
__device__ __forceinline__ vec3 unit3_ieee(vec3 a) {
    const float n = sqrt_ieee(dot3(a, a));
    if (div_range(n) && div_range(a.x) && div_range(a.y) && div_range(a.z)) {
        const float y = rcp_mid(n);
        return mk(div_mid(a.x, n, y), div_mid(a.y, n, y), div_mid(a.z, n, y));
    }
    return mk(a.x / n, a.y / n, a.z / n);
}

// bvh::ray::Ray::new as called from ray_to_bvh_ray (src/scene/mod.rs:506-512): the direction is
// normalised again and the box tests use 1/d of that.
struct BoxRay {
    float ox, oy, oz, ix, iy, iz;
    bool sx, sy, sz;
};
__device__ __forceinline__ BoxRay make_box_ray(vec3 o, vec3 d) {
    BoxRay r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    if (in_2pm20(d.x) & in_2pm20(d.y) & in_2pm20(d.z)) {
        // every component in [2^-20, 2^20]: |d|^2, |d|, d / |d| and its reciprocals stay inside the ranges of the short forms (one test, no nesting)
        const float len = sqrt_mid(dot3(d, d)), y = rcp_mid(len);
        const vec3 u = mk(div_mid(d.x, len, y), div_mid(d.y, len, y), div_mid(d.z, len, y));
        r.ix = rcp_mid(u.x); r.iy = rcp_mid(u.y); r.iz = rcp_mid(u.z);
        r.sx = u.x < 0.0f; r.sy = u.y < 0.0f; r.sz = u.z < 0.0f;
        return r;
    }
    const vec3 n = unit3_ieee(d);
    if (mid_range(n.x) && mid_range(n.y) && mid_range(n.z)) {
        r.ix = rcp_mid(n.x); r.iy = rcp_mid(n.y); r.iz = rcp_mid(n.z);
    } else {
        r.ix = 1.0f / n.x; r.iy = 1.0f / n.y; r.iz = 1.0f / n.z;
    }
    r.sx = n.x < 0.0f; r.sy = n.y < 0.0f; r.sz = n.z < 0.0f;
    return r;
}
// bvh 0.3.2 Ray::intersects_aabb (Williams slab test, no far bound; result = ray_max > 0) with the
// comparisons in the crate's exact form so that NaN falls the same way.  *tnear (entry distance,
// with the z slab folded in) is extra: it only orders and prunes, it never decides a hit.
__device__ __forceinline__ bool slab(const BoxRay& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float* tnear) {
    float ray_min = ((r.sx ? mxx : mnx) - r.ox) * r.ix;
    float ray_max = ((r.sx ? mnx : mxx) - r.ox) * r.ix;
    const float y_min = ((r.sy ? mxy : mny) - r.oy) * r.iy;
    const float y_max = ((r.sy ? mny : mxy) - r.oy) * r.iy;
    bool miss = (ray_min > y_max) || (y_min > ray_max);
    if (y_min > ray_min) ray_min = y_min;
    if (y_max < ray_max) ray_max = y_max;
    const float z_min = ((r.sz ? mxz : mnz) - r.oz) * r.iz;
    const float z_max = ((r.sz ? mnz : mxz) - r.oz) * r.iz;
    miss = miss || (ray_min > z_max) || (z_min > ray_max);
    *tnear = z_min > ray_min ? z_min : ray_min;
    if (z_max < ray_max) ray_max = z_max;
    return !miss && (ray_max > 0.0f);
}

// The same test with the box given relative to the ray origin (b - o, the first operation of every term above): a path vertex sends many
// rays from one origin, so k_light subtracts once per vertex.  Same operations on the same values.
__device__ __forceinline__ bool slab_rel(const BoxRay& r, float dnx, float dny, float dnz, float dxx, float dxy, float dxz) {
    float ray_min = (r.sx ? dxx : dnx) * r.ix;
    float ray_max = (r.sx ? dnx : dxx) * r.ix;
    const float y_min = (r.sy ? dxy : dny) * r.iy;
    const float y_max = (r.sy ? dny : dxy) * r.iy;
    bool miss = (ray_min > y_max) || (y_min > ray_max);
    if (y_min > ray_min) ray_min = y_min;
    if (y_max < ray_max) ray_max = y_max;
    const float z_min = (r.sz ? dxz : dnz) * r.iz;
    const float z_max = (r.sz ? dnz : dxz) * r.iz;
    miss = miss || (ray_min > z_max) || (z_min > ray_max);
    if (z_max < ray_max) ray_max = z_max;
    return !miss && (ray_max > 0.0f);
}

// Object::intersect, triangle arm (src/scene/objects.rs:38-57).  The edges e1 = p2 - p1, e2 = p3 - p1 are the first two operations
// of the reference's function; hit_triangle_e takes them ready-made (the occluder cache computes them once, on the host).
__device__ __forceinline__ bool hit_triangle_e(vec3 p1, vec3 e1, vec3 e2, vec3 o, vec3 d, float* t_out);
__device__ __forceinline__ bool hit_triangle(vec3 p1, vec3 p2, vec3 p3, vec3 o, vec3 d, float* t_out) {
    return hit_triangle_e(p1, sub(p2, p1), sub(p3, p1), o, d, t_out);
}
// ... and hit_triangle_o the part that depends on the ray origin only: s = o - p1, s2 = s x e1, tnum = s2 . e2 (a path vertex
// sends many rays from one origin).  Same operations on the same values, in the reference's order.
__device__ __forceinline__ bool hit_triangle_o(vec3 e1, vec3 e2, vec3 s, vec3 s2, float tnum, vec3 d, float* t_out) {
    const vec3 s1 = cross3(d, e2);
    const float scalar = rcp_ieee(dot3(s1, e1));
    const float t = tnum * scalar;
    const float b1 = dot3(s1, s) * scalar;
    const float b2 = dot3(s2, d) * scalar;
    if (b1 < 0.0f || b2 < 0.0f || b1 > 1.0f || b2 > 1.0f || b1 + b2 > 1.0f + RT_EPS || t < RT_EPS) return false;
    *t_out = t;
    return true;
}
__device__ __forceinline__ bool hit_triangle_e(vec3 p1, vec3 e1, vec3 e2, vec3 o, vec3 d, float* t_out) {
    const vec3 s = sub(o, p1);
    const vec3 s2 = cross3(s, e1);
    return hit_triangle_o(e1, e2, s, s2, dot3(s2, e2), d, t_out);
}
// Is the ray within about 2^-10 rad of the triangle's plane?  s1 . e1 = (d x e2) . e1 is the divisor of the test above; below 2^-10 |e1| |e2|
// (`graze` = that product, stored with the triangle at upload: the third float4's w) its reciprocal has lost ten bits to cancellation and
// the distance the test returns can be off by more than the pruning slack.  Same s1 and dot product as hit_triangle_e (the compiler shares them).
__device__ __forceinline__ bool grazes_triangle(vec3 e1, vec3 e2, vec3 d, float graze) {
    return !(fabsf(dot3(cross3(d, e2), e1)) >= graze);      // NaN counts as grazing
}
// Object::intersect, sphere arm (src/scene/objects.rs:58-76): the nearer root, negative included.
__device__ __forceinline__ bool hit_sphere(float4 cr, vec3 o, vec3 d, float* t_out) {
    const vec3 l = sub(xyz(cr), o);
    const float adj = dot3(l, d);
    const float d2 = dot3(l, l) - (adj * adj);
    const float radius2 = cr.w * cr.w;
    if (d2 > radius2) return false;
    const float thc = sqrtf(radius2 - d2);
    const float t0 = adj - thc;
    const float t1 = adj + thc;
    if (t0 < 0.0f && t1 < 0.0f) return false;
    *t_out = t0 < t1 ? t0 : t1;
    return true;
}

// Per-lane traversal stack: RT_STACK_LDS entries in LDS ([entry][lane], conflict-free), deeper
// entries in a global overflow row.
struct Stack {
    uint32_t* lds;      // &s_stack[threadIdx.x]
    uint32_t* ovf;      // &sc.ovf[global thread]
    uint32_t ovf_stride;
    int sp;
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < RT_STACK_LDS) lds[sp * RT_BLOCK] = v;
        else ovf[(size_t)(sp - RT_STACK_LDS) * ovf_stride] = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        uint32_t v;
        if (sp < RT_STACK_LDS) {
            v = lds[sp * RT_BLOCK];
        } else {
            v = ovf[(size_t)(sp - RT_STACK_LDS) * ovf_stride];
            asm volatile("" : "+v"(v));   // keeps the two loads apart: merged, they become one flat_load through a generic pointer
        }
        return v;
    }
};

// Leaf box of a triangle exactly as the host builder forms it (Bounded for Triangle, objects.rs:302-316).
__device__ __forceinline__ float lo3(float a, float b, float c) { const float m = a < b ? a : b; return m < c ? m : c; }
__device__ __forceinline__ float hi3(float a, float b, float c) { const float m = a > b ? a : b; return m > c ? m : c; }

// Pruning slack: a subtree is skipped only when its entry distance exceeds the current limit by
// more than 2^-10 relative (absolute below 1).  The primitive's own t and the box entry are
// rounded differently; the slack keeps the winner identical to the reference's unpruned search
// as long as the triangle test returns a distance within 2^-11 of the true one.  It does not for
// ill-conditioned triangles (slivers: |e1 x e2| < 2^-5 |e1| |e2|, where 1 / (s1 . e1) loses its
// leading bits for every ray; the reference accepts whatever distance comes out, mod.rs:436-441):
// subtrees that hold one are flagged at upload (FRAGILE bits, rt_api.hip mark_fragile) and never
// skipped.  What remains is a ray that runs within about 1e-3 rad of the plane of a well-shaped
// triangle beyond its best hit and still passes the triangle's box.
__device__ __forceinline__ float prune_limit(float t) { return t + fmaxf(fabsf(t), 1.0f) * (1.0f / 1024.0f); }

// BVH traversal, one ray per lane, executed by the whole wave in step ("while-while"): lanes walk
// inner nodes until fewer than sc.inner_min of them are still at an inner node while others wait at a
// leaf, then the waiting lanes run their primitive test together.  Keeping the two code blocks apart
// keeps most of the 64 lanes active in each (the alternative, one if/else per step, ran at 45-49 %
// VALU lane utilisation: profiles/r01_*).
//
// KIND 0, Scene::intersect (src/scene/mod.rs:406-447): the reference visits every leaf whose
//   ancestors' boxes the ray passes, in DFS order with the right child first, and keeps the first
//   strictly smallest distance.  Here: nearer child first, farther child pushed, subtrees beyond the
//   best distance skipped, equal distances resolved by the leaf's rank in the reference's visiting
//   order — the same (object, distance).
// KIND 1, Scene::is_occluded (src/scene/mod.rs:452-491): true iff some visited leaf reports
//   0 < d < max_dist on a non-emissive object.  Order-independent, so subtrees entered beyond
//   max_dist are skipped, emissive leaves (which can never block, mod.rs:481) are not tested, and the
//   first blocker ends the ray.
// Lanes with have_ray == false take part in the wave votes only.
template <int KIND, bool STATS>
__device__ __forceinline__ void traverse(const DevScene& sc, bool have_ray, vec3 o, vec3 d, float max_dist, uint32_t self_obj, Stack& st,
                                         uint32_t* out_id, float* out_t, bool* out_occluded, TravStats* ts) {
    const BoxRay br = make_box_ray(o, d);
    float best = KIND == 0 ? __builtin_inff() : max_dist;
    const float lim_any = prune_limit(max_dist);
    uint32_t best_id = RT_MISS, best_rank = 0xFFFFFFFFu;
    bool occluded = false;
    bool noprune = false;      // KIND 0: the ray has met a triangle almost in its plane (grazes_triangle): the rest of its search is the reference's, unpruned
    st.sp = 0;
    uint32_t cur = sc.root;
    bool done = !have_ray || (sc.n_tri + sc.n_sph == 0);
    const uint32_t inner_min = sc.inner_min;
    if (KIND == 1 && !done && self_obj != RT_MISS) {
        // Shadow rays start on the surface the path just hit, and about half of them are blocked by that very
        // object at t ~ 1e-6 (the reference applies no surface offset: mod.rs:58-62).  Test it first.  This is
        // the reference's result, not an approximation: a leaf is reached iff the ray passes the boxes of all
        // its ancestors, and with finite 1/d every step of the slab test is monotone in the box bounds while
        // each ancestor's box contains the leaf's, so "reached" == "passes the leaf's own box" (same floats as
        // the host builder: min/max of the stored vertices, centre -+ radius).  Rays with a non-finite 1/d
        // (0 * inf = NaN in the slab test) skip this and take the full traversal.
        const float fin = br.ix - br.ix + (br.iy - br.iy) + (br.iz - br.iz);   // 0 if all finite, NaN otherwise
        if (fin == 0.0f) {
            float t, tn;
            bool ok = false;
            if (self_obj < sc.n_tri) {
                const float4* tp = sc.tris + 3 * (size_t)self_obj;
                const float4 A = tp[0], B = tp[1], C = tp[2];
                if (__float_as_uint(B.w) & RT_INFO_BLOCKER) {
                    if (STATS) ts->tris++;
                    const float mnx = lo3(A.x, B.x, C.x), mny = lo3(A.y, B.y, C.y), mnz = lo3(A.z, B.z, C.z);
                    const float mxx = hi3(A.x, B.x, C.x), mxy = hi3(A.y, B.y, C.y), mxz = hi3(A.z, B.z, C.z);
                    ok = slab(br, mnx, mny, mnz, mxx, mxy, mxz, &tn) && hit_triangle(xyz(A), xyz(B), xyz(C), o, d, &t);
                }
            } else {
                const float4 cr = sc.sph[self_obj - sc.n_tri];
                if (sc.sph_aux[self_obj - sc.n_tri].y & RT_INFO_BLOCKER) {
                    if (STATS) ts->sphs++;
                    ok = slab(br, cr.x - cr.w, cr.y - cr.w, cr.z - cr.w, cr.x + cr.w, cr.y + cr.w, cr.z + cr.w, &tn) && hit_sphere(cr, o, d, &t);
                }
            }
            if (ok && t > 0.0f && t < best) { occluded = true; done = true; }
        }
    }
    for (;;) {
        for (;;) {
            const bool at_inner = !done && !(cur & LEAF_BIT);
            const unsigned long long mi = __ballot(at_inner);
            if (!mi) break;
            if ((uint32_t)__popcll(mi) < inner_min && __ballot(!done && (cur & LEAF_BIT)) != 0ull) break;
            if (at_inner) {
                const float4* n = sc.nodes + 4 * (size_t)cur;
                const float4 a = n[0], b = n[1], c = n[2], k = n[3];
                if (STATS) ts->inner++;
                float tl, tr;
                bool hl = slab(br, a.x, a.y, a.z, a.w, b.x, b.y, &tl);
                bool hr = slab(br, b.z, b.w, c.x, c.y, c.z, c.w, &tr);
                const float lim = KIND == 0 ? (noprune ? __builtin_inff() : prune_limit(best)) : lim_any;
                const uint32_t left = __float_as_uint(k.x), right = __float_as_uint(k.y), fragile = __float_as_uint(k.z);
                hl = hl && (!(tl > lim) || (fragile & 1u));      // a subtree holding an ill-conditioned triangle is searched like the reference does: unpruned
                hr = hr && (!(tr > lim) || (fragile & 2u));
                if (hl && hr) {
                    const bool left_first = !(tr < tl);
                    st.push(left_first ? right : left);
                    cur = left_first ? left : right;
                } else if (hl) {
                    cur = left;
                } else if (hr) {
                    cur = right;
                } else if (st.sp == 0) {
                    done = true;
                } else {
                    cur = st.pop();
                }
            }
        }
        if (!done && (cur & LEAF_BIT)) {
            if (KIND == 0 || !(cur & NONBLOCK_BIT)) {
                const uint32_t obj = cur & OBJECT_MASK;
                float t;
                bool ok;
                uint32_t rank;
                if (obj < sc.n_tri) {
                    const float4* tp = sc.tris + 3 * (size_t)obj;
                    const float4 A = tp[0], B = tp[1], C = tp[2];
                    if (STATS) ts->tris++;
                    ok = hit_triangle(xyz(A), xyz(B), xyz(C), o, d, &t);
                    if (KIND == 0 && sc.graze_guard && grazes_triangle(sub(xyz(B), xyz(A)), sub(xyz(C), xyz(A)), d, C.w)) noprune = true;
                    rank = __float_as_uint(A.w);
                } else {
                    if (STATS) ts->sphs++;
                    ok = hit_sphere(sc.sph[obj - sc.n_tri], o, d, &t);
                    rank = KIND == 0 ? sc.sph_aux[obj - sc.n_tri].x : 0u;
                }
                if (KIND == 0) {
                    if (ok && (t < best || (t == best && best_id != RT_MISS && rank < best_rank))) {
                        best = t; best_id = obj; best_rank = rank;
                    }
                } else if (ok && t > 0.0f && t < best) {
                    occluded = true;
                    done = true;
                }
            }
            if (!done) {
                if (st.sp == 0) done = true;
                else cur = st.pop();
            }
        }
        if (__ballot(!done) == 0ull) break;
    }
    if (KIND == 0) { *out_id = best_id; *out_t = best; }
    else *out_occluded = occluded;
}

#define RT_W_PRIM 0x80000000u
#define RT_W_INV_MAX 1099511627776.0f   // 2^40: rays with a larger |1/d| component take the binary traversal

// Scene::is_occluded (src/scene/mod.rs:452-491) over the 8-wide occlusion BVH (host_scene.h).
//
// The result of is_occluded is order- and tree-independent: true iff some non-emissive object whose
// own box the ray passes (bvh 0.3.2 slab test, exact) reports 0 < d < max_dist.  So the wide nodes
// only have to be conservative: a child is entered whenever the exact test on any box inside it
// could pass.  Per node and axis, with step 2^e, grid origin p and s = (1/d) * 2^e (exact):
//   c    = ((p - o) * (1/d))                            two roundings
//   pad  = 2^-20 * (|c| + 256 |s|)
//   near = fma(q_near, s, c - pad),  far = fma(q_far, s, c + pad)
// The exact test computes fl(fl(b - o) * (1/d)) for a plane b inside [p, p + 255 * 2^e]; both differ
// from the real-number value by at most 2^-24 * (5.1 |c| + 768 |s|) together, under a third of the
// pad, so near <= exact near and far >= exact far on every axis, and
// "max(nears, 0) <= min(fars, limit)" holds whenever the exact slab test (max nears <= min fars,
// min fars > 0) does.  Candidates then take the exact box test on their own box and the exact
// primitive test, exactly what the reference evaluates at a leaf.
// Rays with a non-finite or huge 1/d component (0 * inf = NaN cases) or far-away origins take the
// binary tree (traverse<1>), whose slab test has the crate's NaN behaviour.
// Stack entries: bit 31 = primitive group, bits 30..8 = first node / record index, bits 7..0 = pending slots.
// Shadow rays start on the surface the path just hit, and about half of them are blocked by that very
// object at t ~ 1e-6 (the reference applies no surface offset: mod.rs:58-62).  Testing it first is the
// reference's result, not an approximation (same candidate rule: own box, then the primitive).
template <bool STATS>
__device__ __forceinline__ bool self_blocks(const DevScene& sc, const BoxRay& br, vec3 o, vec3 d, float max_dist, uint32_t self_obj, TravStats* ts) {
    float t = 0.0f, tn;
    bool ok = false;
    if (self_obj < sc.n_tri) {
        const float4* tp = sc.tris + 3 * (size_t)self_obj;
        const float4 A = tp[0], B = tp[1], C = tp[2];
        if (__float_as_uint(B.w) & RT_INFO_BLOCKER) {
            if (STATS) ts->tris++;
            ok = slab(br, lo3(A.x, B.x, C.x), lo3(A.y, B.y, C.y), lo3(A.z, B.z, C.z), hi3(A.x, B.x, C.x), hi3(A.y, B.y, C.y), hi3(A.z, B.z, C.z), &tn) &&
                 hit_triangle(xyz(A), xyz(B), xyz(C), o, d, &t);
        }
    } else {
        const float4 cr = sc.sph[self_obj - sc.n_tri];
        if (sc.sph_aux[self_obj - sc.n_tri].y & RT_INFO_BLOCKER) {
            if (STATS) ts->sphs++;
            ok = slab(br, cr.x - cr.w, cr.y - cr.w, cr.z - cr.w, cr.x + cr.w, cr.y + cr.w, cr.z + cr.w, &tn) && hit_sphere(cr, o, d, &t);
        }
    }
    return ok && t > 0.0f && t < max_dist;
}

// Rays the 8-wide tree may take: finite, moderate 1/d and origin (see occluded()).
__device__ __forceinline__ bool wide_ray_ok(const BoxRay& br, vec3 o) {
    return fabsf(br.ix) <= RT_W_INV_MAX && fabsf(br.iy) <= RT_W_INV_MAX && fabsf(br.iz) <= RT_W_INV_MAX &&
           fabsf(o.x) <= WIDE_COORD_MAX && fabsf(o.y) <= WIDE_COORD_MAX && fabsf(o.z) <= WIDE_COORD_MAX;
}

// One node of the 8-wide tree for this lane: `cur` is an inner group; on return it is the next group (0 = ray finished).
// The padded single-precision test of the 8 child slots of one wide node (derivation: the comment above): bit i set =
// the ray may pass the box of slot i.  n0 = (grid origin, meta), n2..n4 = the 48 plane bytes.
__device__ __forceinline__ uint32_t wide_child_mask(const BoxRay& br, float lim, float4 n0, float4 n2, float4 n3, float4 n4) {
    const uint32_t meta = __float_as_uint(n0.w);
    const float sx = ldexpf(br.ix, (int)(meta & 0xFFu) - WIDE_EXP_BIAS);
    const float sy = ldexpf(br.iy, (int)((meta >> 8) & 0xFFu) - WIDE_EXP_BIAS);
    const float sz = ldexpf(br.iz, (int)((meta >> 16) & 0xFFu) - WIDE_EXP_BIAS);
    const float cx = (n0.x - br.ox) * br.ix, cy = (n0.y - br.oy) * br.iy, cz = (n0.z - br.oz) * br.iz;
    const float px = fmaf(256.0f, fabsf(sx), fabsf(cx)) * 0x1p-20f;
    const float py = fmaf(256.0f, fabsf(sy), fabsf(cy)) * 0x1p-20f;
    const float pz = fmaf(256.0f, fabsf(sz), fabsf(cz)) * 0x1p-20f;
    const float cnx = cx - px, cfx = cx + px, cny = cy - py, cfy = cy + py, cnz = cz - pz, cfz = cz + pz;
    // plane bytes of the 8 slots: near = lo when the direction component is positive
    const uint32_t lox0 = __float_as_uint(n2.x), lox1 = __float_as_uint(n2.y), loy0 = __float_as_uint(n2.z), loy1 = __float_as_uint(n2.w);
    const uint32_t loz0 = __float_as_uint(n3.x), loz1 = __float_as_uint(n3.y), hix0 = __float_as_uint(n3.z), hix1 = __float_as_uint(n3.w);
    const uint32_t hiy0 = __float_as_uint(n4.x), hiy1 = __float_as_uint(n4.y), hiz0 = __float_as_uint(n4.z), hiz1 = __float_as_uint(n4.w);
    const uint32_t nx[2] = {br.sx ? hix0 : lox0, br.sx ? hix1 : lox1}, fx[2] = {br.sx ? lox0 : hix0, br.sx ? lox1 : hix1};
    const uint32_t ny[2] = {br.sy ? hiy0 : loy0, br.sy ? hiy1 : loy1}, fy[2] = {br.sy ? loy0 : hiy0, br.sy ? loy1 : hiy1};
    const uint32_t nz[2] = {br.sz ? hiz0 : loz0, br.sz ? hiz1 : loz1}, fz[2] = {br.sz ? loz0 : hiz0, br.sz ? loz1 : hiz1};
    uint32_t hm = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int w = i >> 2, sh = 8 * (i & 3);
        const float tnx = fmaf((float)((nx[w] >> sh) & 0xFFu), sx, cnx), tfx = fmaf((float)((fx[w] >> sh) & 0xFFu), sx, cfx);
        const float tny = fmaf((float)((ny[w] >> sh) & 0xFFu), sy, cny), tfy = fmaf((float)((fy[w] >> sh) & 0xFFu), sy, cfy);
        const float tnz = fmaf((float)((nz[w] >> sh) & 0xFFu), sz, cnz), tfz = fmaf((float)((fz[w] >> sh) & 0xFFu), sz, cfz);
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, lim));
        hm |= tn <= tf ? (1u << i) : 0u;
    }
    return hm & ((1u << (meta >> 28)) - 1u);                // slots in use
}

// Group entries for the hit slots `hm` of a node (n0.w = meta, n1 = first inner child, first record): `a` primitives
// (minus skip_rec, a record the caller has already tested for this ray, else RT_W_NO_REC), `b` inner nodes.
#define RT_W_NO_REC 0x7FFFFFu
__device__ __forceinline__ void wide_groups(uint32_t hm, uint32_t meta, uint32_t child_base, uint32_t prim_base, uint32_t skip_rec, uint32_t skip_rec2,
                                            uint32_t* a, uint32_t* b) {
    const uint32_t k = (meta >> 24) & 0xFu;
    const uint32_t ih = hm & ((1u << k) - 1u);
    uint32_t ph = hm >> k;
    // A ray that starts on a wall lies in the padded slab of that wall's own flat box, so the object a ray starts on is a
    // candidate of nearly every such ray; k_light has tested it.  skip_rec2: the wall's coplanar neighbour, when k_light has found
    // that the ray cannot pass its box (RT_LR_SKIP_SIB).
    const uint32_t rel = skip_rec - prim_base, rel2 = skip_rec2 - prim_base;
    if (rel < 8u) ph &= ~(1u << rel);
    if (rel2 < 8u) ph &= ~(1u << rel2);
    *a = ph ? (RT_W_PRIM | (prim_base << 8) | ph) : 0u;
    *b = ih ? ((child_base << 8) | ih) : 0u;
}

template <bool STATS>
__device__ __forceinline__ void wide_node_step(const DevScene& sc, const BoxRay& br, float lim, uint32_t skip_rec, uint32_t skip_rec2, uint32_t& cur, Stack& st, TravStats* ts) {
    const uint32_t node = (cur >> 8) + (uint32_t)__ffs((int)(cur & 0xFFu)) - 1u;
    uint32_t rest = cur & (cur - 1u);
    if (!(rest & 0xFFu)) rest = 0u;
    const float4* np = sc.wnodes + 5 * (size_t)node;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
    if (STATS) ts->wide++;
    const uint32_t hm = wide_child_mask(br, __float_as_uint(n1.z) ? __builtin_inff() : lim, n0, n2, n3, n4);   // n1.z: the node holds an ill-conditioned triangle (see prune_limit)
    uint32_t a, b;
    wide_groups(hm, __float_as_uint(n0.w), __float_as_uint(n1.x), __float_as_uint(n1.y), skip_rec, skip_rec2, &a, &b);   // primitives first: a blocker ends the ray
    uint32_t c = rest;
    if (!a) { a = b; b = 0u; }
    if (!a) { a = c; c = 0u; }
    if (c) st.push(c);
    if (b) st.push(b);
    cur = a;
    if (!cur && st.sp > 0) cur = st.pop();
}

// One primitive record for this lane: `cur` is a primitive group.  Returns true when it blocks the ray
// (cur = 0); otherwise cur is the next group (0 = ray finished unblocked).
template <bool STATS>
__device__ __forceinline__ bool wide_prim_step(const DevScene& sc, const BoxRay& br, vec3 o, vec3 d, float max_dist, uint32_t& cur, Stack& st, TravStats* ts) {
    const uint32_t prim = ((cur >> 8) & 0x7FFFFFu) + (uint32_t)__ffs((int)(cur & 0xFFu)) - 1u;
    uint32_t rest = cur & (cur - 1u);
    if (!(rest & 0xFFu)) rest = 0u;
    const float4* tp = sc.wprims + 3 * (size_t)prim;
    const float4 A = tp[0], B = tp[1];
    // blocked = own box passed (exact slab test) && primitive hit && 0 < t < max_dist, evaluated cheapest-first: most
    // candidates fail the primitive test, and only the few that pass need the box test that makes them a candidate at all
    float t = 0.0f, tn;
    bool ok;
    if (__float_as_uint(A.w) == 0u) {
        const float4 C = tp[2];
        if (STATS) {
            ts->tris++;
            if (!slab(br, lo3(A.x, B.x, C.x), lo3(A.y, B.y, C.y), lo3(A.z, B.z, C.z), hi3(A.x, B.x, C.x), hi3(A.y, B.y, C.y), hi3(A.z, B.z, C.z), &tn)) ts->box_miss++;
        }
        ok = hit_triangle(xyz(A), xyz(B), xyz(C), o, d, &t) && t > 0.0f && t < max_dist &&
             slab(br, lo3(A.x, B.x, C.x), lo3(A.y, B.y, C.y), lo3(A.z, B.z, C.z), hi3(A.x, B.x, C.x), hi3(A.y, B.y, C.y), hi3(A.z, B.z, C.z), &tn);
    } else {
        if (STATS) ts->sphs++;
        const float r = B.x;
        ok = hit_sphere(make_float4(A.x, A.y, A.z, r), o, d, &t) && t > 0.0f && t < max_dist &&
             slab(br, A.x - r, A.y - r, A.z - r, A.x + r, A.y + r, A.z + r, &tn);
    }
    if (ok) {
        cur = 0u;
        return true;
    }
    cur = rest;
    if (!cur && st.sp > 0) cur = st.pop();
    return false;
}

// ---- Shafts: the light samples of one path vertex as ONE beam (no reference counterpart) ------------------------------------
// The L rays a vertex p sends to a light all start at p and end on the light's sphere, and most beams hit nothing: each of the rays
// that k_light cannot settle then walks the occlusion BVH from the root only to find no candidate.  shaft_clear() makes that walk once
// per (vertex, light): it answers "could wide_child_mask() pass, for ANY ray from p to a point of the light's box, at a slot that
// holds a primitive record (other than the records in `skip`, which k_light tests itself)?".  If not, k_trace_p would hand no
// candidate to any of these rays and leave every one of them unblocked; they are settled as visible without being traced.  The
// answer is one-sided by construction: `false` ("not clear") only costs the traversal that would have run anyway.
//
// Geometry.  The rays lie in the convex hull of p and the light's box B_L, i.e. { p + tau w : w in [a, c] (per axis), tau >= 0 } with
// a = B_L.min - p, c = B_L.max - p.  A child box [lo, hi] meets it iff one tau satisfies, on every axis,
//     tau a <= hi - p     and     tau c >= lo - p
// (the cross-section of the hull at tau is the box p + tau [a, c]).  Each inequality bounds tau from one side: the first from below when
// a < 0 (tau >= (hi - p) / a) and from above when a > 0; the second from below when c > 0 and from above when c < 0.  So per axis
// VA = (hi - p) / a and VC = (lo - p) / c are evaluated like the near / far plane distances of a ray, and the child is entered iff
// max(lower bounds, 0) <= min(upper bounds, tau_max).
//
// Why it dominates the per-ray test (wide_child_mask above, rays with wide_ray_ok only; the others go to k_binary whatever the shaft
// says).  Let that test pass for a ray (p, wi) at a slot with planes lo, hi: some t* in [0, lim] has tn <= t* <= tf, so the point
// X = p + t* u (u the box ray's direction, real arithmetic on its float components) lies in the slot's box grown per axis by
// eps <= 1.4 * 2^-20 (256 * 2^e + |g - p|) (the pad of that test plus its rounding, times |u| |1/u| ~ 1).  u is ps / |ps| with each
// component off by under 2^-21 relative (the sample point, ps = sp - p, two normalisations, all correctly rounded), and every sample
// point lies in B_L grown by 2^-21 (|centre| + r) (|random_sphere| <= 1 + 2^-22 per component).  Hence X = p + tau w with w in [a, c]
// once a and c are moved outward by 2^-18 (|centre| + r + |p|), and tau = t* / |w| <= lim / (d_s (1 - 2^-20)) <= 1 + 2^-9 max(1, 1 / d_min),
// d_min a lower bound of the distance from p to B_L.  The box of the slot grown by eps therefore meets the hull below tau_max, and the
// bounds below — pad 2^-18 (|c| + 256 |s|), i.e. 4 * 2^-20 (...) in the units of tau, against eps' 1.4 and this function's own rounding of
// under 0.4 — pass.  Nodes that hold an ill-conditioned triangle have no far limit in the per-ray test (prune_limit) and none here.
// All operands are kept finite (coordinates up to 2^20, |a|, |c| >= 2^-20: shaft_make), so no bound is ever NaN.
struct Shaft {
    float px, py, pz;
    float iax, iay, iaz;   // 1 / a per axis (a < 0 or a > 0, never 0)
    float icx, icy, icz;   // 1 / c
    float tmax;
};
#define RT_SHAFT_COORD_MAX 1048576.0f   // 2^20
// The hull of p and the sphere light (centre, r)'s box.  False: no shaft for this pair (p inside the light's box, or coordinates beyond 2^20).
__device__ __forceinline__ bool shaft_make(vec3 p, float4 cr, Shaft* sh) {
    const float r = cr.w;
    const bool sane = fabsf(p.x) <= RT_SHAFT_COORD_MAX && fabsf(p.y) <= RT_SHAFT_COORD_MAX && fabsf(p.z) <= RT_SHAFT_COORD_MAX &&
                      fabsf(cr.x) <= RT_SHAFT_COORD_MAX && fabsf(cr.y) <= RT_SHAFT_COORD_MAX && fabsf(cr.z) <= RT_SHAFT_COORD_MAX && r >= 0.0f && r <= RT_SHAFT_COORD_MAX;
    float a[3], c[3];
    const float pc[3] = {p.x, p.y, p.z}, cc[3] = {cr.x, cr.y, cr.z};
    float dmin = 0.0f;
    bool all_straddle = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float slop = (fabsf(cc[k]) + r + fabsf(pc[k])) * 0x1p-18f;
        a[k] = ((cc[k] - r) - pc[k]) - slop;
        c[k] = ((cc[k] + r) - pc[k]) + slop;
        if (fabsf(a[k]) < 0x1p-20f) a[k] = -0x1p-20f;      // outward: a only ever decreases, c only ever increases
        if (fabsf(c[k]) < 0x1p-20f) c[k] = 0x1p-20f;
        const bool straddle = a[k] < 0.0f && c[k] > 0.0f;
        all_straddle = all_straddle && straddle;
        if (!straddle) dmin = fmaxf(dmin, a[k] > 0.0f ? a[k] : -c[k]);   // every ray is at least this long
    }
    sh->px = p.x; sh->py = p.y; sh->pz = p.z;
    sh->iax = 1.0f / a[0]; sh->iay = 1.0f / a[1]; sh->iaz = 1.0f / a[2];
    sh->icx = 1.0f / c[0]; sh->icy = 1.0f / c[1]; sh->icz = 1.0f / c[2];
    sh->tmax = 1.0f + 0x1p-9f * fmaxf(1.0f, 1.0f / (dmin * 0.99f));
    return sane && !all_straddle;
}
// One axis of the shaft test of a node: the constant terms of the four bounds (lower / upper from the a-inequality on the hi planes, lower /
// upper from the c-inequality on the lo planes; the two that do not apply are -inf / +inf) and the plane steps in units of tau.
struct ShaftAxis { float sA, sC, kAlo, kAup, kClo, kCup; };
__device__ __forceinline__ ShaftAxis shaft_axis(float g_minus_p, int e, float ia, float ic) {
    ShaftAxis x;
    x.sA = ldexpf(ia, e); x.sC = ldexpf(ic, e);
    const float cA = g_minus_p * ia, cC = g_minus_p * ic;
    const float pA = fmaf(256.0f, fabsf(x.sA), fabsf(cA)) * 0x1p-18f, pC = fmaf(256.0f, fabsf(x.sC), fabsf(cC)) * 0x1p-18f;
    const float inf = __builtin_inff();
    x.kAlo = ia < 0.0f ? cA - pA : -inf; x.kAup = ia < 0.0f ? inf : cA + pA;
    x.kClo = ic > 0.0f ? cC - pC : -inf; x.kCup = ic > 0.0f ? inf : cC + pC;
    return x;
}
// bit i set = the hull may meet the box of slot i (the shaft's counterpart of wide_child_mask)
__device__ __forceinline__ uint32_t shaft_child_mask(const Shaft& sh, float tmax, float4 n0, float4 n2, float4 n3, float4 n4) {
    const uint32_t meta = __float_as_uint(n0.w);
    const ShaftAxis X = shaft_axis(n0.x - sh.px, (int)(meta & 0xFFu) - WIDE_EXP_BIAS, sh.iax, sh.icx);
    const ShaftAxis Y = shaft_axis(n0.y - sh.py, (int)((meta >> 8) & 0xFFu) - WIDE_EXP_BIAS, sh.iay, sh.icy);
    const ShaftAxis Z = shaft_axis(n0.z - sh.pz, (int)((meta >> 16) & 0xFFu) - WIDE_EXP_BIAS, sh.iaz, sh.icz);
    const uint32_t lox[2] = {__float_as_uint(n2.x), __float_as_uint(n2.y)}, loy[2] = {__float_as_uint(n2.z), __float_as_uint(n2.w)};
    const uint32_t loz[2] = {__float_as_uint(n3.x), __float_as_uint(n3.y)}, hix[2] = {__float_as_uint(n3.z), __float_as_uint(n3.w)};
    const uint32_t hiy[2] = {__float_as_uint(n4.x), __float_as_uint(n4.y)}, hiz[2] = {__float_as_uint(n4.z), __float_as_uint(n4.w)};
    uint32_t hm = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int w = i >> 2, s = 8 * (i & 3);
        const float qhx = (float)((hix[w] >> s) & 0xFFu), qlx = (float)((lox[w] >> s) & 0xFFu);
        const float qhy = (float)((hiy[w] >> s) & 0xFFu), qly = (float)((loy[w] >> s) & 0xFFu);
        const float qhz = (float)((hiz[w] >> s) & 0xFFu), qlz = (float)((loz[w] >> s) & 0xFFu);
        float tn = fmaxf(fmaxf(fmaf(qhx, X.sA, X.kAlo), fmaf(qlx, X.sC, X.kClo)), 0.0f);
        tn = fmaxf(tn, fmaxf(fmaf(qhy, Y.sA, Y.kAlo), fmaf(qly, Y.sC, Y.kClo)));
        tn = fmaxf(tn, fmaxf(fmaf(qhz, Z.sA, Z.kAlo), fmaf(qlz, Z.sC, Z.kClo)));
        float tf = fminf(fminf(fmaf(qhx, X.sA, X.kAup), fmaf(qlx, X.sC, X.kCup)), tmax);
        tf = fminf(tf, fminf(fmaf(qhy, Y.sA, Y.kAup), fmaf(qly, Y.sC, Y.kCup)));
        tf = fminf(tf, fminf(fmaf(qhz, Z.sA, Z.kAup), fmaf(qlz, Z.sC, Z.kCup)));
        hm |= tn <= tf ? (1u << i) : 0u;
    }
    return hm & ((1u << (meta >> 28)) - 1u);                // slots in use
}
// The walk: depth-first over the nodes the hull may meet, at most max_steps (<= RT_STACK_LDS) of them; stops at the first primitive slot it
// may meet that is not one of `skip`.  lds = this lane's column of a [RT_STACK_LDS][RT_BLOCK] array (one push per step at most).
__device__ __forceinline__ bool shaft_clear(const DevScene& sc, const Shaft& sh, uint32_t skip0, uint32_t skip1, uint32_t skip2, uint32_t skip3, uint32_t* lds,
                                            uint32_t max_steps, uint32_t* steps) {
    uint32_t cur = 1u, n = 0u;   // inner group {node 0}
    int sp = 0;
    bool clear = true;
    while (cur != 0u) {
        if (n == max_steps) { clear = false; break; }
        ++n;
        const uint32_t node = (cur >> 8) + (uint32_t)__ffs((int)(cur & 0xFFu)) - 1u;
        uint32_t rest = cur & (cur - 1u);
        if (!(rest & 0xFFu)) rest = 0u;
        const float4* np = sc.wnodes + 5 * (size_t)node;
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        const uint32_t hm = shaft_child_mask(sh, __float_as_uint(n1.z) ? __builtin_inff() : sh.tmax, n0, n2, n3, n4);
        const uint32_t meta = __float_as_uint(n0.w), k = (meta >> 24) & 0xFu, prim_base = __float_as_uint(n1.y);
        const uint32_t ih = hm & ((1u << k) - 1u);
        uint32_t ph = hm >> k;
        const uint32_t r0 = skip0 - prim_base, r1 = skip1 - prim_base, r2 = skip2 - prim_base, r3 = skip3 - prim_base;
        if (r0 < 8u) ph &= ~(1u << r0);
        if (r1 < 8u) ph &= ~(1u << r1);
        if (r2 < 8u) ph &= ~(1u << r2);
        if (r3 < 8u) ph &= ~(1u << r3);
        if (ph) { clear = false; break; }
        const uint32_t b = ih ? ((__float_as_uint(n1.x) << 8) | ih) : 0u;
        if (b && rest) { lds[sp * RT_BLOCK] = rest; ++sp; }
        cur = b ? b : rest;
        if (!cur && sp > 0) { --sp; cur = lds[sp * RT_BLOCK]; }
    }
    *steps = n;
    return clear;
}

template <bool STATS>
__device__ __forceinline__ bool occluded(const DevScene& sc, bool have_ray, vec3 o, vec3 d, float max_dist, uint32_t self_obj, Stack& st, TravStats* ts) {
    const BoxRay br = make_box_ray(o, d);
    const bool wide = have_ray && sc.n_wnodes != 0u && wide_ray_ok(br, o);
    const bool binary = have_ray && !wide;
    bool occ = false;
    uint32_t cur = wide ? 1u : 0u;   // inner group {node 0}
    if (wide && self_obj != RT_MISS && self_blocks<STATS>(sc, br, o, d, max_dist, self_obj, ts)) { occ = true; cur = 0u; }
    const float lim = prune_limit(max_dist);
    const uint32_t inner_min = sc.inner_min;
    st.sp = 0;
    for (;;) {
        for (;;) {
            const bool at_inner = cur != 0u && !(cur & RT_W_PRIM);
            const unsigned long long mi = __ballot(at_inner);
            if (!mi) break;
            if ((uint32_t)__popcll(mi) < inner_min && __ballot((cur & RT_W_PRIM) != 0u) != 0ull) break;
            if (at_inner) wide_node_step<STATS>(sc, br, lim, RT_W_NO_REC, RT_W_NO_REC, cur, st, ts);
        }
        if (cur & RT_W_PRIM)
            if (wide_prim_step<STATS>(sc, br, o, d, max_dist, cur, st, ts)) occ = true;
        if (__ballot(cur != 0u) == 0ull) break;
    }
    if (__ballot(binary) != 0ull) {
        bool occ2 = false;
        traverse<1, STATS>(sc, binary, o, d, max_dist, self_obj, st, nullptr, nullptr, &occ2, ts);
        if (binary) occ = occ2;
    }
    return occ;
}

__device__ __forceinline__ uint32_t object_info(const DevScene& sc, uint32_t obj) {
    return obj < sc.n_tri ? __float_as_uint(sc.tris[3 * (size_t)obj + 1].w) : sc.sph_aux[obj - sc.n_tri].y;
}

// Object::surface_normal (src/scene/objects.rs:228-230, 270-290): sphere (p-c)/|p-c|; triangle =
// barycentric blend of the stored vertex normals, not re-normalised.
__device__ __forceinline__ vec3 surface_normal(const DevScene& sc, uint32_t obj, vec3 point) {
    if (obj >= sc.n_tri) return unit3(sub(point, xyz(sc.sph[obj - sc.n_tri])));
    const float4* tp = sc.tris + 3 * (size_t)obj;
    const vec3 p1 = xyz(tp[0]), v0 = sub(xyz(tp[1]), p1), v1 = sub(xyz(tp[2]), p1);
    const vec3 v2 = sub(point, p1);
    const float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    const float denom = d00 * d11 - d01 * d01;
    const float v = (d11 * d20 - d01 * d21) / denom;
    const float w = (d00 * d21 - d01 * d20) / denom;
    const float u = 1.0f - v - w;
    const float4* np = sc.tri_nrm + 3 * (size_t)obj;
    return add(add(scale(xyz(np[0]), u), scale(xyz(np[1]), v)), scale(xyz(np[2]), w));
}

// Vector::random_hemisphere / to_coord_space / random_sphere (src/scene/geo.rs:101-145)
__device__ __forceinline__ vec3 random_hemisphere(uint64_t* rng) {
    const float xi1 = next_f32(rng);
    const float xi2 = next_f32(rng);
    const float cos_theta = xi1;
    const float sin_theta = sqrtf(1.0f - xi1 * xi1);
    const float phi = 2.0f * RT_PI * xi2;
    float sin_phi, cos_phi;
    sincos_s(phi, &sin_phi, &cos_phi);
    return mk(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
__device__ __forceinline__ vec3 to_coord_space(vec3 v, vec3 n) {
    vec3 t, b;
    if (n.z < -0.9999999f) {
        t = mk(0.0f, -1.0f, 0.0f);
        b = mk(-1.0f, 0.0f, 0.0f);
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b_val = -n.x * n.y * a;
        t = mk(1.0f - n.x * n.x * a, b_val, -n.x);
        b = mk(b_val, 1.0f - n.y * n.y * a, -n.y);
    }
    return add(add(scale(t, v.x), scale(b, v.y)), scale(n, v.z));
}
__device__ __forceinline__ vec3 random_sphere(uint64_t* rng) {
    const float xi1 = next_f32(rng);
    const float xi2 = next_f32(rng);
    const float theta = 2.0f * RT_PI * xi1;
    const float phi = acos_s(1.0f - 2.0f * xi2);
    float sp, cp, st, ct;
    sincos_s(phi, &sp, &cp);
    sincos_s(theta, &st, &ct);
    return mk(sp * ct, sp * st, cp);
}

__device__ __forceinline__ bool spectrum_is_black(vec3 c) { return (c.x + c.y) + c.z <= RT_EPS; }   // src/common.rs:40-43
__device__ __forceinline__ vec3 mul3(vec3 a, vec3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }

// A path that ends at level k with radiance `color` unwinds the reference's recursion
// (global_illumination, src/raytracer.rs:318-325): for each earlier level, scale the sub-path
// radiance by reflected * cos * pdf unless it is black, then add that level's direct term.
__device__ __forceinline__ void fold_path(const PassCtx& pc, uint32_t slot, uint32_t k, vec3 color) {
    for (uint32_t j = k; j-- > 0;) {
        const float4* rec = pc.stack + 2 * ((size_t)slot * pc.stack_depth + j);
        const float4 A = rec[0], B = rec[1];
        if (!spectrum_is_black(color)) color = scale(scale(mul3(color, xyz(B)), A.w), B.w);
        color = add(xyz(A), color);
    }
    pc.radiance[slot] = make_float4(color.x, color.y, color.z, 0.0f);
}

// Queue compaction: wave64 ballot + prefix popcount inside each wave, the four wave totals
// combined through LDS, and ONE atomic per workgroup and loop iteration reserves the run in the
// queue (same-address atomics retire at roughly 10 ns each on MI355X, so one per wave made
// k_generate / k_shade_* atomic-bound: profiles/r01_v3).  Every thread of the workgroup must call
// it the same number of times.  s_tmp: RT_BLOCK / 64 + 1 words.
// `each`: queue entries per kept thread (k_camera reserves a run per pixel); the return value is the first of them.
__device__ __forceinline__ uint32_t block_append(bool keep, uint32_t* counter, uint32_t* s_tmp, uint32_t each = 1u) {
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) s_tmp[wave] = (uint32_t)__popcll(mask) * each;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < RT_BLOCK / 64; ++w) total += s_tmp[w];
        s_tmp[RT_BLOCK / 64] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_tmp[RT_BLOCK / 64];
    for (uint32_t w = 0; w < wave; ++w) base += s_tmp[w];
    __syncthreads();
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) * each;
}

// block_append for B entries per thread with one atomic (see block_append_classed_n).  Order inside the run: entry k, then wave, then lane, i.e.
// what B separate appends one after the other would have given.  s_tmp: B * RT_BLOCK / 64 + 1 words.
template <int B>
__device__ __forceinline__ void block_append_n(const bool (&keep)[B], uint32_t* counter, uint32_t* s_tmp, uint32_t (&pos)[B]) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    constexpr uint32_t W = RT_BLOCK / 64;
    unsigned long long m[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        m[k] = __ballot(keep[k]);
        if (lane == 0) s_tmp[(uint32_t)k * W + wave] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < B * W; ++i) total += s_tmp[i];
        s_tmp[B * W] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t run = s_tmp[B * W];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < B; ++k)
        for (uint32_t w = 0; w < W; ++w) {
            if (w == wave) pos[k] = run + (uint32_t)__popcll(m[k] & below);
            run += s_tmp[(uint32_t)k * W + w];
        }
    __syncthreads();
}

// block_append that also groups what it appends by a class 0..2: within the workgroup's run the entries of class 0 come first, then class 1, then
// class 2 (each in thread order).  k_shade_a uses it to put the vertices whose own-object test takes the same arm of k_light next to each other,
// so that fewer of k_light's waves run every arm.  s_tmp: 3 * RT_BLOCK / 64 + 1 words.
__device__ __forceinline__ uint32_t block_append_classed(bool keep, uint32_t cls, uint32_t* counter, uint32_t* s_tmp) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    unsigned long long m[3];
#pragma unroll
    for (uint32_t c = 0; c < 3u; ++c) {
        m[c] = __ballot(keep && cls == c);
        if (lane == 0) s_tmp[c * (RT_BLOCK / 64) + wave] = (uint32_t)__popcll(m[c]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < 3u * (RT_BLOCK / 64); ++i) total += s_tmp[i];
        s_tmp[3 * (RT_BLOCK / 64)] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_tmp[3 * (RT_BLOCK / 64)];
    for (uint32_t i = 0; i < cls * (RT_BLOCK / 64) + wave; ++i) base += s_tmp[i];          // the classes before, then the waves before in this class
    __syncthreads();
    const unsigned long long mine = cls == 0u ? m[0] : (cls == 1u ? m[1] : m[2]);
    return base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
}

// block_append_classed for B entries per thread and ONE atomic: the same-address atomic with a return value retires at about 10 ns, so one per 256
// entries bounds a compaction kernel at 25 G entries/s whatever else it does (k_shade_a's largest launch: 259 k atomics, 2.6 of its 3.1 ms).  Order
// inside the workgroup's run: entry k (i.e. groups of RT_BLOCK consecutive rays, as block_append_classed would have appended them one after the
// other), then class, then wave, then lane.  s_tmp: 3 * B * RT_BLOCK / 64 + 1 words.
template <int B>
__device__ __forceinline__ void block_append_classed_n(const bool (&keep)[B], const uint32_t (&cls)[B], uint32_t* counter, uint32_t* s_tmp, uint32_t (&pos)[B]) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    constexpr uint32_t W = RT_BLOCK / 64;
    unsigned long long m[B][3];
#pragma unroll
    for (int k = 0; k < B; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            m[k][c] = __ballot(keep[k] && cls[k] == c);
            if (lane == 0) s_tmp[((uint32_t)k * 3u + c) * W + wave] = (uint32_t)__popcll(m[k][c]);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t i = 0; i < 3u * B * W; ++i) total += s_tmp[i];
        s_tmp[3 * B * W] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t run = s_tmp[3 * B * W];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        uint32_t off = 0u;
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c)
            for (uint32_t w = 0; w < W; ++w) {
                if (w == wave && cls[k] == c) off = run + (uint32_t)__popcll(m[k][c] & below);
                run += s_tmp[((uint32_t)k * 3u + c) * W + w];
            }
        pos[k] = off;
    }
    __syncthreads();
}

// Sum of v over the workgroup, returned to thread 0 only.  s_tmp: RT_BLOCK / 64 words.
__device__ __forceinline__ uint32_t block_sum_to_thread0(uint32_t v, uint32_t* s_tmp) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (__lane_id() == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t total = 0;
    if (threadIdx.x == 0)
        for (uint32_t w = 0; w < RT_BLOCK / 64; ++w) total += s_tmp[w];
    __syncthreads();
    return total;
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(unsigned long long* stats, const TravStats& ts, int base) {
    if (!STATS) return;
    uint32_t a = ts.inner, b = ts.tris, c = ts.sphs, w = ts.wide;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        c += __shfl_xor(c, off);
        w += __shfl_xor(w, off);
    }
    const unsigned long long active = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)active) - 1) {
        atomicAdd(&stats[base], (unsigned long long)a);
        atomicAdd(&stats[base + 1], (unsigned long long)b);
        atomicAdd(&stats[base + 2], (unsigned long long)c);
        if (w) atomicAdd(&stats[ST_S_WIDE], (unsigned long long)w);
    }
}

#define RT_STACK_DECL                                                   \
    __shared__ uint32_t s_stack[RT_STACK_LDS * RT_BLOCK];               \
    Stack st;                                                           \
    st.lds = &s_stack[threadIdx.x];                                     \
    st.ovf = sc.ovf + (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;      \
    st.ovf_stride = sc.ovf_stride;                                      \
    st.sp = 0

// ---------------------------------------------------------------------------------------------
// Primary-ray cast (debug_render_helper, src/raytracer.rs:394-416; BASELINE config 3).
// ids/ts (optional) receive the winning object and distance; rgba (optional) the grey level.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) k_primary(DevScene sc, rt_render_params P, uint32_t tiles_x, uint32_t n_own_pixels,
                                                     uint32_t* ids, float* ts_out, uint8_t* rgba, unsigned long long* stats) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const vec3 cam = mk(P.cam[0], P.cam[1], P.cam[2]);
    for (uint32_t op = blockIdx.x * RT_BLOCK + threadIdx.x; op < n_own_pixels; op += gridDim.x * RT_BLOCK) {
        uint32_t x, y;
        const bool valid = own_pixel_xy(P, tiles_x, op, &x, &y);
        const vec3 d = screen_to_world(P, x, y);
        uint32_t id;
        float t;
        traverse<0, STATS>(sc, valid, cam, d, 0.0f, RT_MISS, st, &id, &t, nullptr, &tst);
        if (!valid) continue;
        const size_t pi = (size_t)y * P.width + x;
        if (ids) ids[pi] = id;
        if (ts_out) ts_out[pi] = t;
        if (rgba) {
            float g = 0.0f;
            if (id != RT_MISS) {
                const float max_distance = 100.0f;
                const float dm = (t != t) ? max_distance : (t < max_distance ? t : max_distance);
                g = 0.7f * (1.0f - (dm / max_distance));
            }
            const uint32_t c = to_color(g);
            reinterpret_cast<uint32_t*>(rgba)[pi] = c | (c << 8) | (c << 16) | 0xFF000000u;
        }
    }
    flush_stats<STATS>(stats, tst, ST_C_INNER);
}

// Camera rays for one pass (render_helper, src/raytracer.rs:224-241): every sample of a pixel
// starts with the same primary ray and its own RNG stream.
__global__ void __launch_bounds__(RT_BLOCK) k_generate(PassCtx pc) {
    const rt_render_params& P = pc.P;
    const vec3 cam = mk(P.cam[0], P.cam[1], P.cam[2]);
    __shared__ uint32_t s_tmp[RT_BLOCK / 64 + 1];
    for (uint32_t q0 = blockIdx.x * RT_BLOCK; q0 < pc.npaths; q0 += gridDim.x * RT_BLOCK) {
        const uint32_t q = q0 + threadIdx.x;
        const uint32_t pl = q / pc.spc, s = pc.s0 + (q - pl * pc.spc);
        uint32_t x = 0, y = 0;
        const bool valid = q < pc.npaths && own_pixel_xy(P, pc.tiles_x, pc.pix0 + pl, &x, &y);
        const uint32_t e = block_append(valid, &pc.counts[0], s_tmp);
        if (!valid) continue;
        pc.qrng[0][e] = stream_seed(pc.seed_mixed, y * P.width + x, s);
        const vec3 d = screen_to_world(P, x, y);
        pc.ray_o[0][e] = make_float4(cam.x, cam.y, cam.z, __uint_as_float(q));
        pc.ray_d[0][e] = make_float4(d.x, d.y, d.z, 0.0f);
    }
}

template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK, 8) k_extend(DevScene sc, PassCtx pc, uint32_t level) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const uint32_t n = pc.counts[level];
    const float4* ro = pc.ray_o[level & 1];
    const float4* rd = pc.ray_d[level & 1];
    for (uint32_t e = blockIdx.x * RT_BLOCK + threadIdx.x; e < n; e += gridDim.x * RT_BLOCK) {
        uint32_t id;
        float t;
        traverse<0, STATS>(sc, true, xyz(ro[e]), xyz(rd[e]), 0.0f, RT_MISS, st, &id, &t, nullptr, &tst);
        pc.hit_id[e] = id;
        pc.hit_t[e] = t;
    }
    flush_stats<STATS>(pc.stats, tst, ST_C_INNER);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pc.stats[ST_RAYS_CLOSEST], (unsigned long long)n);
}

// k_generate + k_extend for the camera rays of a pass in one kernel: every sample of a pixel starts with the same primary ray
// (render_helper casts screen_to_world(i, j) spp times, raytracer.rs:224-235), so the ray is traced once per pixel and its hit handed to the
// pixel's spc samples of the pass, which differ in their path slot and RNG stream only.  A wave's pixels reserve consecutive runs of spc queue
// entries, so the wave writes one contiguous stretch of the queue, 64 entries at a time.  The reference's spc intersect calls are counted as
// such (ST_RAYS_CLOSEST); the traversal counters ("stats") count the work done, i.e. one traversal per pixel.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK, 8) k_camera(DevScene sc, PassCtx pc) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const rt_render_params& P = pc.P;
    const vec3 cam = mk(P.cam[0], P.cam[1], P.cam[2]);
    __shared__ uint32_t s_tmp[RT_BLOCK / 64 + 1];
    __shared__ uint32_t s_lane[RT_BLOCK];              // per wave: the lane of its r-th valid pixel
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, spc = pc.spc;
    uint32_t n_entries = 0;
    for (uint32_t pl0 = blockIdx.x * RT_BLOCK; pl0 < pc.npix; pl0 += gridDim.x * RT_BLOCK) {
        const uint32_t pl = pl0 + threadIdx.x;
        uint32_t x = 0, y = 0, id = RT_MISS;
        float t = 0.0f;
        vec3 d = mk(0.0f, 0.0f, 1.0f);
        const bool valid = pl < pc.npix && own_pixel_xy(P, pc.tiles_x, pc.pix0 + pl, &x, &y);
        if (valid) d = screen_to_world(P, x, y);
        traverse<0, STATS>(sc, valid, cam, d, 0.0f, RT_MISS, st, &id, &t, nullptr, &tst);
        const unsigned long long m = __ballot(valid);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (valid) s_lane[wave * 64u + rank] = lane;
        const uint32_t q0 = block_append(valid, &pc.counts[0], s_tmp, spc);       // this pixel's run of spc entries (barriers inside: s_lane is visible)
        const uint32_t total = (uint32_t)__popcll(m) * spc;
        if (total == 0u) continue;
        const uint32_t wave_base = (uint32_t)__shfl((int)q0, __ffsll((long long)m) - 1);
        n_entries += lane == 0u ? total : 0u;
        for (uint32_t i0 = 0; i0 < total; i0 += 64u) {                                // every lane stays in the loop: the hits are handed over by __shfl
            const uint32_t i = i0 + lane;
            const bool mine = i < total;
            const uint32_t r = mine ? i / spc : 0u, k = i - r * spc;
            const int src = (int)s_lane[wave * 64u + r];
            const uint32_t pl_s = (uint32_t)__shfl((int)pl, src), x_s = (uint32_t)__shfl((int)x, src), y_s = (uint32_t)__shfl((int)y, src);
            const uint32_t id_s = (uint32_t)__shfl((int)id, src);
            const float t_s = __shfl(t, src), dx = __shfl(d.x, src), dy = __shfl(d.y, src), dz = __shfl(d.z, src);
            if (mine) {
                const uint32_t e = wave_base + i, slot = pl_s * spc + k;
                pc.qrng[0][e] = stream_seed(pc.seed_mixed, y_s * P.width + x_s, pc.s0 + k);
                pc.ray_o[0][e] = make_float4(cam.x, cam.y, cam.z, __uint_as_float(slot));
                pc.ray_d[0][e] = make_float4(dx, dy, dz, 0.0f);
                pc.hit_id[e] = id_s;
                pc.hit_t[e] = t_s;
            }
        }
    }
    flush_stats<STATS>(pc.stats, tst, ST_C_INNER);
    const uint32_t t_entries = block_sum_to_thread0(n_entries, s_tmp);
    if (threadIdx.x == 0 && t_entries) atomicAdd(&pc.stats[ST_RAYS_CLOSEST], (unsigned long long)t_entries);
}

// Scene::intersect for the ray queue as a persistent wave loop: the traversal of traverse<0> (same tree, same tests, same pruning and
// tie-break: the result of a ray does not depend on what the other lanes do), but a wave keeps going through rows of 64 rays and hands
// new rays to the lanes whose ray has finished, instead of waiting for the longest ray of every row.  After the first bounce rays of a row
// have little in common and their traversal lengths differ several-fold, which left k_extend at 0.23-0.29 of its lanes.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK, 8) k_extend_p(DevScene sc, PassCtx pc, uint32_t level) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const uint32_t n = (sc.n_tri + sc.n_sph == 0) ? 0u : pc.counts[level];
    const float4* ro = pc.ray_o[level & 1];
    const float4* rd = pc.ray_d[level & 1];
    const uint32_t n_waves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t inner_min = sc.extend_inner_min, refill_min = sc.extend_refill_min;
    uint32_t next_row = __builtin_amdgcn_readfirstlane((blockIdx.x * RT_BLOCK + threadIdx.x) >> 6);   // rows of 64 rays: this wave's, then + n_waves
    uint32_t pend_at = 0u, pend_end = 0u;                                     // rays of the current row not yet handed out
    // the ray this lane is tracing (have)
    bool have = false, noprune = false;
    uint32_t e = 0u, cur = 0u, best_id = RT_MISS, best_rank = 0xFFFFFFFFu;
    float best = __builtin_inff();
    vec3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f);
    BoxRay br = make_box_ray(o, d);
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        const bool more = pend_at < pend_end || (unsigned long long)next_row * 64ull < n;
        if (more && n_idle >= refill_min) {
            if (pend_at == pend_end) {
                pend_at = next_row * 64u;
                pend_end = n - pend_at < 64u ? n : pend_at + 64u;
                next_row += n_waves;
            }
            const uint32_t take = pend_end - pend_at < n_idle ? pend_end - pend_at : n_idle;
            const uint32_t irank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!have && irank < take) {
                e = pend_at + irank;
                o = xyz(ro[e]); d = xyz(rd[e]);
                br = make_box_ray(o, d);
                best = __builtin_inff(); best_id = RT_MISS; best_rank = 0xFFFFFFFFu;
                noprune = false;
                st.sp = 0;
                cur = sc.root;
                have = true;
            }
            pend_at += take;
            continue;
        }
        if (n_idle == 64u) break;
        bool done = !have;
        for (;;) {
            const bool at_inner = !done && !(cur & LEAF_BIT);
            const unsigned long long mi = __ballot(at_inner);
            if (!mi) break;
            if ((uint32_t)__popcll(mi) < inner_min && __ballot(!done && (cur & LEAF_BIT)) != 0ull) break;
            if (more && (uint32_t)__popcll(__ballot(done)) >= refill_min) break;      // enough lanes have run dry: hand out new rays first
            if (at_inner) {
                const float4* np = sc.nodes + 4 * (size_t)cur;
                const float4 a = np[0], b = np[1], c = np[2], k = np[3];
                if (STATS) tst.inner++;
                float tl, tr;
                bool hl = slab(br, a.x, a.y, a.z, a.w, b.x, b.y, &tl);
                bool hr = slab(br, b.z, b.w, c.x, c.y, c.z, c.w, &tr);
                const float lim = noprune ? __builtin_inff() : prune_limit(best);
                const uint32_t left = __float_as_uint(k.x), right = __float_as_uint(k.y), fragile = __float_as_uint(k.z);
                hl = hl && (!(tl > lim) || (fragile & 1u));
                hr = hr && (!(tr > lim) || (fragile & 2u));
                if (hl && hr) {
                    const bool left_first = !(tr < tl);
                    st.push(left_first ? right : left);
                    cur = left_first ? left : right;
                } else if (hl) {
                    cur = left;
                } else if (hr) {
                    cur = right;
                } else if (st.sp == 0) {
                    done = true;
                } else {
                    cur = st.pop();
                }
            }
        }
        if (!done && (cur & LEAF_BIT)) {
            const uint32_t obj = cur & OBJECT_MASK;
            float t;
            bool ok;
            uint32_t rank;
            if (obj < sc.n_tri) {
                const float4* tp = sc.tris + 3 * (size_t)obj;
                const float4 A = tp[0], B = tp[1], C = tp[2];
                if (STATS) tst.tris++;
                ok = hit_triangle(xyz(A), xyz(B), xyz(C), o, d, &t);
                if (sc.graze_guard && grazes_triangle(sub(xyz(B), xyz(A)), sub(xyz(C), xyz(A)), d, C.w)) noprune = true;
                rank = __float_as_uint(A.w);
            } else {
                if (STATS) tst.sphs++;
                ok = hit_sphere(sc.sph[obj - sc.n_tri], o, d, &t);
                rank = sc.sph_aux[obj - sc.n_tri].x;
            }
            if (ok && (t < best || (t == best && best_id != RT_MISS && rank < best_rank))) {
                best = t; best_id = obj; best_rank = rank;
            }
            if (st.sp == 0) done = true;
            else cur = st.pop();
        }
        if (have && done) {
            pc.hit_id[e] = best_id;
            pc.hit_t[e] = best;
            have = false;
        }
    }
    flush_stats<STATS>(pc.stats, tst, ST_C_INNER);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pc.stats[ST_RAYS_CLOSEST], (unsigned long long)pc.counts[level]);
}

// cast_ray dispatch (src/raytracer.rs:337-349) up to the point where light sampling starts.  RT_SA_BATCH rays per thread and loop iteration,
// appended to the vertex queue with one atomic (block_append_classed_n).
#ifndef RT_SA_BATCH
#define RT_SA_BATCH 4
#endif
__global__ void __launch_bounds__(RT_BLOCK) k_shade_a(DevScene sc, PassCtx pc, uint32_t level) {
    const uint32_t n = pc.counts[level];
    const uint32_t bounces_left = pc.P.bounces - level;
    const float4* ro = pc.ray_o[level & 1];
    const float4* rd = pc.ray_d[level & 1];
    __shared__ uint32_t s_tmp[3 * RT_SA_BATCH * (RT_BLOCK / 64) + 1];
    uint32_t n_alive = 0, n_alive_tri = 0;
    // rays that found a surface to light are compacted into the vertex queue (misses and depth-0 hits end here)
    for (uint32_t e0 = blockIdx.x * (RT_BLOCK * RT_SA_BATCH); e0 < n; e0 += gridDim.x * (RT_BLOCK * RT_SA_BATCH)) {
        bool alive[RT_SA_BATCH];
        uint32_t cls[RT_SA_BATCH], q[RT_SA_BATCH];
        float4 vp[RT_SA_BATCH], vn[RT_SA_BATCH], vs[RT_SA_BATCH];
        uint64_t vr[RT_SA_BATCH];
#pragma unroll
        for (int k = 0; k < RT_SA_BATCH; ++k) {
            const uint32_t e = e0 + (uint32_t)k * RT_BLOCK + threadIdx.x;
            alive[k] = false;
            cls[k] = 0u;
            vp[k] = vn[k] = vs[k] = make_float4(0, 0, 0, 0);
            vr[k] = 0ull;
            if (e < n) {
                const float4 o4 = ro[e];
                const uint32_t slot = __float_as_uint(o4.w);
                const uint32_t id = pc.hit_id[e];
                if (id == RT_MISS) {
                    fold_path(pc, slot, level, mk(0.0f, 0.0f, 0.0f));                      // raytracer.rs:348
                } else {
                    const uint32_t info = object_info(sc, id);
                    const uint32_t mat = info & RT_INFO_MAT;
                    if (bounces_left == 0) {
                        fold_path(pc, slot, level, xyz(sc.mats[2 * mat + 1]));             // zero_bounce_radiance, raytracer.rs:245-247
                    } else {
                        const vec3 o = xyz(o4), d = xyz(rd[e]);
                        const float min_dist = pc.hit_t[e] - RT_EPS;                       // RayIntersection::point, mod.rs:58-62
                        const vec3 p = add(o, scale(d, min_dist));
                        const vec3 nrm = surface_normal(sc, id, p);
                        vp[k] = make_float4(p.x, p.y, p.z, __uint_as_float(mat));
                        vn[k] = make_float4(nrm.x, nrm.y, nrm.z, __uint_as_float(id + 1u));
                        vs[k] = make_float4(o4.w, d.x, d.y, d.z);
                        vr[k] = pc.qrng[level & 1][e];
                        alive[k] = true;
                        cls[k] = id >= sc.n_tri ? 2u : ((info & RT_INFO_FLAT) ? 0u : 1u);
                        n_alive++;
                        n_alive_tri += id < sc.n_tri ? 1u : 0u;
                    }
                }
            }
        }
        // the order inside the workgroup's run follows the arm of k_light's own-object test: vertices on triangles with an axis-flat box (walls: they
        // walk a beam), on other triangles (mesh: no beam), on spheres (the arm with the square root)
        block_append_classed_n<RT_SA_BATCH>(alive, cls, &pc.vcounts[level], s_tmp, q);
#pragma unroll
        for (int k = 0; k < RT_SA_BATCH; ++k)
            if (alive[k]) { pc.vP[q[k]] = vp[k]; pc.vN[q[k]] = vn[k]; pc.vS[q[k]] = vs[k]; pc.vR[q[k]] = vr[k]; }
    }
    // ray and hit counters: one atomic each per workgroup
    const uint32_t t_alive = block_sum_to_thread0(n_alive, s_tmp);
    const uint32_t t_tri = block_sum_to_thread0(n_alive_tri, s_tmp);
    if (threadIdx.x == 0 && t_alive) {
        atomicAdd(&pc.stats[pc.P.direct_estimator == RT_DIRECT_HEMISPHERE ? ST_RAYS_CLOSEST : ST_RAYS_SHADOW], (unsigned long long)t_alive * pc.nlL);
        atomicAdd(&pc.stats[ST_HITS_TRI], (unsigned long long)t_tri);
        atomicAdd(&pc.stats[ST_HITS_SPH], (unsigned long long)(t_alive - t_tri));
    }
}

// One thread per (vertex, light, light sample) of one_bounce_radiance_importance
// (src/raytracer.rs:266-283): Object::sample_l (objects.rs:101-116), the shadow ray, and the
// contribution Le * bsdf * |wi.n| * pdf.  Ray r = e * nlL + j: the nlL rays of a vertex sit in
// neighbouring lanes (same origin, same light), so their traversals stay coherent at every level.
// When nlL divides 64 (pc.in_wave) the light samples of a vertex are summed right here, in sample
// order, by the vertex's first lane reading the wave's results back from LDS, and only the direct
// term (16 B per vertex) goes to HBM; otherwise each sample goes to pc.contrib for k_shade_b.
// HEMI = the hemisphere direct-lighting estimator (not in the reference source).
template <bool STATS, bool HEMI>
__global__ void __launch_bounds__(RT_BLOCK, RT_SHADOW_WAVES) k_shadow(DevScene sc, PassCtx pc, uint32_t level) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const uint32_t n = pc.vcounts[level] * pc.nlL;
    const uint32_t L = pc.P.light_samples;
    uint32_t* red = s_stack;   // rows 0..3 of this wave's own (then empty) stack columns: red[k * RT_BLOCK + thread] = component k
    for (uint32_t r = blockIdx.x * RT_BLOCK + threadIdx.x; r < n; r += gridDim.x * RT_BLOCK) {
        const uint32_t e = r / pc.nlL, j = r - e * pc.nlL;
        const float4 N4 = pc.vN[e];
        const bool alive = __float_as_uint(N4.w) != 0u;
        const float4 P4 = pc.vP[e];
        const vec3 p = xyz(P4), nrm = xyz(N4);
        float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        vec3 wi = mk(0.0f, 0.0f, 1.0f), c = mk(0.0f, 0.0f, 0.0f), reflected = mk(0.0f, 0.0f, 0.0f);
        float d_s = 0.0f, cos_theta = 0.0f;
        if (alive) {
            const ulonglong2 jm = pc.jump[j];
            uint64_t rng = jm.x * pc.vR[e] + jm.y;
            const float4 m0 = sc.mats[2 * __float_as_uint(P4.w)];
            // Object::bsdf, objects.rs:121-127
            reflected = __float_as_uint(m0.x) == RT_BSDF_DIFFUSE ? scale(mk(m0.y, m0.z, m0.w), 1.0f / RT_PI) : mk(0.0f, 0.0f, 0.0f);
            if (!HEMI) {
                const uint32_t light = sc.lights[j / L];
                const float4 cr = sc.sph[light - sc.n_tri];
                const vec3 le = xyz(sc.mats[2 * (sc.sph_aux[light - sc.n_tri].y & RT_INFO_MAT) + 1]);
                const vec3 s = add(xyz(cr), scale(random_sphere(&rng), cr.w));      // Sphere::random_point, objects.rs:223-226
                const vec3 ps = sub(s, p);
                d_s = length3(ps);
                wi = scale(ps, 1.0f / d_s);
                const float d_c = length3(sub(xyz(cr), p));
                const float cos_a = (d_c * d_c + cr.w * cr.w - d_s * d_s) / (2.0f * d_c * cr.w);
                const float pdf = 2.0f * RT_PI * (1.0f - cos_a);
                cos_theta = fabsf(dot3(wi, nrm));
                c = scale(scale(mul3(le, reflected), cos_theta), pdf);              // raytracer.rs:279
            } else {
                const vec3 w = to_coord_space(random_hemisphere(&rng), nrm);
                cos_theta = fabsf(dot3(w, nrm));
                wi = unit3(w);
            }
        }
        if (!HEMI) {
            const bool blocked = occluded<STATS>(sc, alive, p, wi, d_s, sc.self_test ? __float_as_uint(N4.w) - 1u : RT_MISS, st, &tst);
            if (alive && !blocked) out = make_float4(c.x, c.y, c.z, 1.0f);
        } else {
            uint32_t id = RT_MISS;
            float t;
            traverse<0, STATS>(sc, alive, p, wi, 0.0f, RT_MISS, st, &id, &t, nullptr, &tst);
            if (alive && id != RT_MISS) {
                const vec3 le = xyz(sc.mats[2 * (object_info(sc, id) & RT_INFO_MAT) + 1]);
                const vec3 ch = scale(scale(mul3(le, reflected), cos_theta), 2.0f * RT_PI);
                out = make_float4(ch.x, ch.y, ch.z, 1.0f);
            }
        }
        if (!pc.in_wave) {
            pc.contrib[r] = out;
            continue;
        }
        // ordered sum over the vertex's samples (raytracer.rs:269-283): color += ...; l += color / L
        red[threadIdx.x] = __float_as_uint(out.x);
        red[RT_BLOCK + threadIdx.x] = __float_as_uint(out.y);
        red[2 * RT_BLOCK + threadIdx.x] = __float_as_uint(out.z);
        red[3 * RT_BLOCK + threadIdx.x] = __float_as_uint(out.w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (j == 0 && alive) {
            const float inv_light_samples = 1.0f / (float)L;
            vec3 l = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t g = 0; g < pc.n_groups; ++g) {
                vec3 color = mk(0.0f, 0.0f, 0.0f);
                for (uint32_t s = 0; s < L; ++s) {
                    const uint32_t src = threadIdx.x + g * L + s;            // a lane of this wave: j == 0 and nlL | 64
                    if (red[3 * RT_BLOCK + src] != 0u)
                        color = add(color, mk(__uint_as_float(red[src]), __uint_as_float(red[RT_BLOCK + src]), __uint_as_float(red[2 * RT_BLOCK + src])));
                }
                l = add(l, scale(color, inv_light_samples));
            }
            pc.direct[e] = make_float4(l.x, l.y, l.z, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
    }
    flush_stats<STATS>(pc.stats, tst, HEMI ? ST_C_INNER : ST_S_INNER);
}

// ---------------------------------------------------------------------------------------------
// Light sampling with importance (one_bounce_radiance_importance, src/raytracer.rs:251-283) in two
// kernels, split where the divergence class changes:
//   k_light    one thread per path vertex, looping over its lights x light samples: Object::sample_l
//              (objects.rs:101-116; full-rate f64 polynomial trig, every lane busy) and the test of
//              the object the ray starts on, which ends about half of all shadow rays;
//   k_trace_p  Scene::is_occluded for the survivors, as a persistent wave loop over the 8-wide
//              occlusion BVH that refills finished lanes (per-ray cost varies 1..40 nodes).
// Slot of (vertex e, light sample j): tiles of 64 vertices, sample-major inside a tile, so that
// k_light (lane = vertex), k_trace_p (row of 64 slots) and k_shade_b (lane = vertex) all touch
// consecutive addresses.  Most rays are settled as blocked in k_light (the object the ray starts on, the
// occluder cache) and leave no record at all: they add nothing to the sum of raytracer.rs:279.  The others
// are packed at the front of their row (pc.tmask[row] says which vertices they belong to; a ray's place is
// the number of set bits below its lane), so that the kernels after k_light touch whole 64-byte sectors.
// What the reference computes for a visible sample only (pdf, |wi.n|) is left to k_shade_b.
#define RT_LR_BLOCKED 0x80000000u   // sign bit of the ray's distance in pc.lray, set by k_trace_p / k_binary: the ray is occluded (a distance is never negative)
#define RT_LR_LANE_SHIFT 23         // bits 23..28: the ray's vertex within its tile of 64 (bits 0..22: the record of the object it starts on)
#define RT_LR_SKIP_SIB 0x20000000u  // the coplanar neighbour of the triangle the ray starts on (sc.wsib) cannot be a candidate: k_trace_p skips its record too
#define RT_LR_BINARY 0x40000000u    // the ray has a non-finite or huge 1/d: k_binary settles it on the binary tree, k_trace_p drops it
// One visible light sample's term of one_bounce_radiance_importance's sum (raytracer.rs:276-279), shared by k_light (the samples it settles as
// visible and sums itself) and k_shade_b (the ones that went through k_trace_p / k_binary): what depends on (vertex, light) only, then the term.
struct LightPdf { float num0, den, rcp_den; bool den_mid; };
__device__ __forceinline__ LightPdf light_pdf_setup(vec3 p, float4 cr) {
    // Object::sample_l's cos_a = (d_c^2 + r^2 - d_s^2) / (2 d_c r), objects.rs:109-113: the divisor and its reciprocal are the vertex's
    LightPdf lp;
    const float d_c = length3(sub(xyz(cr), p));
    lp.num0 = d_c * d_c + cr.w * cr.w;
    lp.den = 2.0f * d_c * cr.w;
    lp.den_mid = div_range(lp.den);
    lp.rcp_den = rcp_mid(lp.den);
    return lp;
}
__device__ __forceinline__ vec3 light_term(const LightPdf& lp, vec3 base, vec3 nrm, vec3 wi, float d_s) {
    const float num = lp.num0 - d_s * d_s;
    const float cos_a = (lp.den_mid && div_range(num)) ? div_mid(num, lp.den, lp.rcp_den) : num / lp.den;
    const float pdf = 2.0f * RT_PI * (1.0f - cos_a);
    const float cos_theta = fabsf(dot3(wi, nrm));                                  // raytracer.rs:278
    return scale(scale(base, cos_theta), pdf);                                     // raytracer.rs:279
}
__device__ __forceinline__ size_t ls_slot(uint32_t e, uint32_t j, uint32_t nlL) {
    return (size_t)(e >> 6) * (64u * (size_t)nlL) + (size_t)j * 64u + (e & 63u);
}
template <bool STATS, int WAVES>   // WAVES: waves per SIMD the register budget is sized for
__global__ void __launch_bounds__(RT_BLOCK, WAVES) k_light(DevScene sc, PassCtx pc, uint32_t level, HotTable hot) {
    TravStats tst = {0, 0, 0, 0, 0};
    __shared__ uint32_t s_shaft[RT_STACK_LDS * RT_BLOCK];                      // shaft_clear()'s stack, [entry][lane]
    uint32_t sh_tried = 0u, sh_clear = 0u, sh_steps = 0u, sh_rays = 0u, sh_summed = 0u;        // STATS
    const uint32_t shaft_steps = pc.shaft_steps < RT_STACK_LDS ? pc.shaft_steps : RT_STACK_LDS;
    const uint32_t n = pc.vcounts[level], n_round = (n + 63u) & ~63u;
    const uint32_t nlL = pc.nlL, L = pc.P.light_samples;
    for (uint32_t e0 = blockIdx.x * RT_BLOCK + threadIdx.x; e0 < n_round; e0 += gridDim.x * RT_BLOCK) {
        // every queued vertex is live (k_shade_a compacts); the lanes beyond n only pad the last tile of 64: they run along on
        // vertex 0 so that the wave votes below see whole rows, and store nothing
        const bool live = e0 < n;
        const uint32_t e = live ? e0 : 0u;
        const float4 N4 = pc.vN[e];
        const float4 P4 = pc.vP[e];
        const vec3 p = xyz(P4);
                const uint64_t rng0 = pc.vR[e];
        // the object the rays start on, fetched once per vertex (see self_blocks())
        const uint32_t self_obj = sc.self_test ? __float_as_uint(N4.w) - 1u : RT_MISS;
        bool self_is_tri = false, self_is_sph = false;
        float4 A = make_float4(0, 0, 0, 0), B = A, C = A;
        if (self_obj != RT_MISS) {
            if (self_obj < sc.n_tri) {
                const float4* tp = sc.tris + 3 * (size_t)self_obj;
                A = tp[0]; B = tp[1]; C = tp[2];
                self_is_tri = (__float_as_uint(B.w) & RT_INFO_BLOCKER) != 0u;
            } else {
                A = sc.sph[self_obj - sc.n_tri];
                self_is_sph = (sc.sph_aux[self_obj - sc.n_tri].y & RT_INFO_BLOCKER) != 0u;
            }
        }
        // its record in the occlusion BVH: k_trace_p need not test it again for the rays that pass the test here
        const uint32_t self_rec = (self_is_tri || self_is_sph) ? sc.wrec[self_obj] : RT_W_NO_REC;
        // A wall is two coplanar triangles with the same flat box: a ray that starts on one lies in the padded slab of the other's box and
        // would take its primitive test in k_trace_p, nearly always for nothing.  The reference tests that neighbour only if the ray passes
        // its own box (bvh slab test); on the flat axis both planes are the coordinate c, so the test ends with ray_max <= (c - o) * (1/d)
        // and fails whenever that product is not positive.  (c - o) is a vertex value; the product and its sign are the slab test's own.
        uint32_t sib_axis = 0u;
        float sib_d = 0.0f;
        bool has_sib = false;
        if (self_is_tri && self_rec != RT_W_NO_REC) {
            const uint32_t w = sc.wsib[self_rec];
            has_sib = (w & RT_W_NO_REC) != RT_W_NO_REC;
            sib_axis = w >> 23;
            sib_d = (sib_axis == 0u ? A.x : (sib_axis == 1u ? A.y : A.z)) - (sib_axis == 0u ? p.x : (sib_axis == 1u ? p.y : p.z));
        }
        // Which vertices try a shaft (a choice of where to spend the walk, nothing more): those on a sphere or on a triangle with an axis-flat
        // box (a wall).  A vertex on a mesh triangle has its neighbours in every beam it sends.
        const bool shaft_vertex = shaft_steps != 0u && (self_is_sph || (self_is_tri && (__float_as_uint(B.w) & RT_INFO_FLAT) != 0u));
        const float own_dnx = (self_is_sph ? A.x - A.w : lo3(A.x, B.x, C.x)) - p.x, own_dny = (self_is_sph ? A.y - A.w : lo3(A.y, B.y, C.y)) - p.y,
                    own_dnz = (self_is_sph ? A.z - A.w : lo3(A.z, B.z, C.z)) - p.z;
        const float own_dxx = (self_is_sph ? A.x + A.w : hi3(A.x, B.x, C.x)) - p.x, own_dxy = (self_is_sph ? A.y + A.w : hi3(A.y, B.y, C.y)) - p.y,
                    own_dxz = (self_is_sph ? A.z + A.w : hi3(A.z, B.z, C.z)) - p.z;
        // the part of the own triangle's test that depends on the vertex only (hit_triangle_o), once for all its light samples
        const vec3 own_e1 = sub(xyz(B), xyz(A)), own_e2 = sub(xyz(C), xyz(A)), own_s = sub(p, xyz(A));
        const vec3 own_s2 = cross3(own_s, own_e1);
        const float own_tn = dot3(own_s2, own_e2);
        const bool p_wide_ok = fabsf(p.x) <= WIDE_COORD_MAX && fabsf(p.y) <= WIDE_COORD_MAX && fabsf(p.z) <= WIDE_COORD_MAX;
        uint64_t rng = rng0;                                                   // the vertex's stream: two draws per light sample, in sample order
        // Object::bsdf of the vertex's material (objects.rs:118-128): the factor of every light sample k_light sums itself (below)
        const float4 m0 = sc.mats[2 * __float_as_uint(P4.w)];
        const vec3 bsdf = __float_as_uint(m0.x) == RT_BSDF_DIFFUSE ? scale(mk(m0.y, m0.z, m0.w), 1.0f / RT_PI) : mk(0.0f, 0.0f, 0.0f);
        const vec3 nrm = xyz(N4);
        size_t row = ls_slot(e0 & ~63u, 0, nlL);                               // first slot of the row of (this tile, sample j): + 64 per sample
        unsigned long long* row_mask = pc.tmask + (size_t)(e0 >> 6) * nlL;    // mask of that row: + 1 per sample
        unsigned long long* row_mask_t = pc.tmask_t + (size_t)(e0 >> 6) * nlL;
        for (uint32_t g = 0, j = 0; g < pc.n_groups; ++g) {                    // for light in lights (raytracer.rs:266): per-light values stay in registers
          const uint32_t light = sc.lights[g];
          const float4 cr = sc.sph[light - sc.n_tri];
          // Occluder cache: is_occluded is "some non-emissive object blocks", whichever is found first, so the large occluders next
          // to this light (chosen at scene build: a ceiling the lamp pokes through blocks 4 of 5 samples of the Cornell lamp) are
          // tested here, by every lane and from registers, before any ray is queued for traversal.  Same exact tests as at a leaf.
          const uint32_t hg = g < RT_HOT_LIGHTS ? g : 0u;
          const bool has_hot = g < RT_HOT_LIGHTS;
          // the part of the cached triangles' test that depends on the vertex only (hit_triangle_o)
          vec3 hot_s[RT_HOT], hot_s2[RT_HOT], hot_dn[RT_HOT], hot_dx[RT_HOT];
          float hot_tn[RT_HOT];
#pragma unroll
          for (int h = 0; h < RT_HOT; ++h) {
              const HotRec& hr = hot.rec[hg][h];
              hot_dn[h] = sub(mk(hr.mn[0], hr.mn[1], hr.mn[2]), p);          // its own box relative to the vertex (slab_rel)
              hot_dx[h] = sub(mk(hr.mx[0], hr.mx[1], hr.mx[2]), p);
              hot_s[h] = sub(p, mk(hr.a[0], hr.a[1], hr.a[2]));
              hot_s2[h] = cross3(hot_s[h], mk(hr.e1[0], hr.e1[1], hr.e1[2]));
              hot_tn[h] = dot3(hot_s2[h], mk(hr.e2[0], hr.e2[1], hr.e2[2]));
          }
          // The beam of this vertex towards this light (shaft_clear above): when no primitive record other than the ones tested right here
          // (the object the rays start on, the light's occluder cache) and the coplanar neighbour — where every ray of the beam fails the
          // flat-axis rule below, i.e. would carry RT_LR_SKIP_SIB — can be a candidate of any of its rays, the rays not blocked here are visible.
          bool clear = false;
          if (shaft_vertex) {
              Shaft sh;
              if (shaft_make(p, cr, &sh)) {
                  const float ia_sib = sib_axis == 0u ? sh.iax : (sib_axis == 1u ? sh.iay : sh.iaz), ic_sib = sib_axis == 0u ? sh.icx : (sib_axis == 1u ? sh.icy : sh.icz);
                  // skip_sib of every ray: !(sib_d * (1/u) > 0) with u's sign that of the direction interval [a, c] on the flat axis
                  const bool sib_all = has_sib && (sib_d == 0.0f || (sib_d > 0.0f && ic_sib < 0.0f) || (sib_d < 0.0f && ia_sib > 0.0f));
                  const uint32_t sk1 = sib_all ? (sc.wsib[self_rec] & RT_W_NO_REC) : RT_W_NO_REC;
                  const uint32_t sk2 = (has_hot && hot.rec[hg][0].kind != 2u) ? hot.rec[hg][0].rec : RT_W_NO_REC;
                  const uint32_t sk3 = (has_hot && hot.rec[hg][RT_HOT - 1].kind != 2u) ? hot.rec[hg][RT_HOT - 1].rec : RT_W_NO_REC;
                  uint32_t steps = 0u;
                  clear = shaft_clear(sc, sh, self_rec, sk1, sk2, sk3, &s_shaft[threadIdx.x], shaft_steps, &steps);
                  if (STATS && live) { sh_tried++; sh_steps += steps; sh_clear += clear ? 1u : 0u; }
              }
          }
          // A ray of a clear beam that is not blocked here is visible, and everything its term of the sum of raytracer.rs:279 needs is in
          // registers: the vertex adds it right here, in sample order, and the ray leaves no record.  The sum is a chain of f32 additions, so
          // that holds for the samples BEFORE the first one of this (vertex, light) that does leave a record (a k_binary ray); from there on
          // the visible ones go to the row as before and k_shade_b carries on from the sum written below.  Same operations, same order as k_shade_b.
          const vec3 base = mul3(xyz(sc.mats[2 * (sc.sph_aux[light - sc.n_tri].y & RT_INFO_MAT) + 1]), bsdf);
          const LightPdf lp = light_pdf_setup(p, cr);
          vec3 color = mk(0.0f, 0.0f, 0.0f);
          bool sum_open = clear && pc.light_sum != 0u, has_rec = false;
          // the cache records' fields the sample loop uses, held in scalar registers across it (the compiler otherwise reloads them from the kernel
          // arguments inside the loop, ten loads each followed by its wait: k_light - 1.3 %)
          uint32_t hp_kind[RT_HOT], hp_rec[RT_HOT];
          float hp_a[RT_HOT][3], hp_e1[RT_HOT][3], hp_e2[RT_HOT][3];
#pragma unroll
          for (int h = 0; h < RT_HOT; ++h) {
              const HotRec& hr = hot.rec[hg][h];
              hp_kind[h] = hr.kind; hp_rec[h] = hr.rec;
              asm volatile("" : "+s"(hp_kind[h]), "+s"(hp_rec[h]));
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                  hp_a[h][c] = hr.a[c]; hp_e1[h][c] = hr.e1[c]; hp_e2[h][c] = hr.e2[c];
                  asm volatile("" : "+s"(hp_a[h][c]), "+s"(hp_e1[h][c]), "+s"(hp_e2[h][c]));
              }
          }
          for (uint32_t ls = 0; ls < L; ++ls, ++j, row += 64) {                // for _ in 0..light_samples (raytracer.rs:269)
            const vec3 sp = add(xyz(cr), scale(random_sphere(&rng), cr.w));      // Sphere::random_point, objects.rs:223-226
            const vec3 ps = sub(sp, p);
            // d_s = |ps|, wi = ps * (1 / d_s) (objects.rs:106-108), then the box ray of wi: direction normalised again, 1 / direction (mod.rs:506-512).
            // One range test serves the whole chain: with every component of ps in [2^-20, 2^20], |ps|^2, |ps|, wi, |wi| ~ 1, wi / |wi| and its
            // reciprocals all stay inside the ranges of the short exact forms (rt_math.h), which are then taken unguarded; any other sample point
            // (a zero or tiny component: a ray along an axis) takes the guarded forms.  Same bits either way.
            // pdf and |wi.n| (objects.rs:109-115, raytracer.rs:278): k_shade_b, for the visible samples only.
            float d_s;
            vec3 wi;
            BoxRay br;
            bool binary;
            if (in_2pm18_20(ps.x) & in_2pm18_20(ps.y) & in_2pm18_20(ps.z)) {
                d_s = sqrt_mid(dot3(ps, ps));
                wi = scale(ps, rcp_mid(d_s));
                const float n = sqrt_mid(dot3(wi, wi)), y = rcp_mid(n);
                const vec3 u = mk(div_mid(wi.x, n, y), div_mid(wi.y, n, y), div_mid(wi.z, n, y));
                br.ox = p.x; br.oy = p.y; br.oz = p.z;
                br.ix = rcp_mid(u.x); br.iy = rcp_mid(u.y); br.iz = rcp_mid(u.z);
                br.sx = u.x < 0.0f; br.sy = u.y < 0.0f; br.sz = u.z < 0.0f;
                // every |u| component is at least 2^-18 / (sqrt(3) 2^20) (1 - 2^-20) > 2^-39, so |1/u| < 2^40: the ray is fit for the 8-wide tree
                // (wide_ray_ok) as far as its direction goes; the origin's part of that test is the vertex's (p_wide_ok)
                binary = !p_wide_ok;
            } else {
                d_s = sqrt_ieee(dot3(ps, ps));
                wi = scale(ps, rcp_ieee(d_s));
                br = make_box_ray(p, wi);
                binary = !wide_ray_ok(br, p);
            }
            if (pc.force_binary != 0u && ((e0 + j) & pc.force_binary) == 0u) binary = true;
            bool blocked = false;
            if (!binary && (self_is_tri || self_is_sph) && slab_rel(br, own_dnx, own_dny, own_dnz, own_dxx, own_dxy, own_dxz)) {
                float t = 0.0f;
                bool ok;
                if (self_is_tri) {
                    if (STATS) tst.tris++;
                    ok = hit_triangle_o(own_e1, own_e2, own_s, own_s2, own_tn, wi, &t);
                } else {
                    if (STATS) tst.sphs++;
                    ok = hit_sphere(A, p, wi, &t);
                }
                blocked = ok && t > 0.0f && t < d_s;
            }
#pragma unroll
            for (int h = 0; h < RT_HOT; ++h) {
                HotRec hr;                                // (kernel argument: scalar operands straight from SGPRs)
                hr.kind = hp_kind[h]; hr.rec = hp_rec[h];
                for (int c = 0; c < 3; ++c) { hr.a[c] = hp_a[h][c]; hr.e1[c] = hp_e1[h][c]; hr.e2[c] = hp_e2[h][c]; }
                if (has_hot && hr.kind != 2u && hr.rec != self_rec && !binary && !blocked) {      // (uniform but for the last three)
                    float t = 0.0f;
                    bool ok;
                    if (hr.kind == 0u) {
                        if (STATS) tst.tris++;
                        ok = hit_triangle_o(mk(hr.e1[0], hr.e1[1], hr.e1[2]), mk(hr.e2[0], hr.e2[1], hr.e2[2]), hot_s[h], hot_s2[h], hot_tn[h], wi, &t) &&
                             t > 0.0f && t < d_s && slab_rel(br, hot_dn[h].x, hot_dn[h].y, hot_dn[h].z, hot_dx[h].x, hot_dx[h].y, hot_dx[h].z);
                    } else {
                        if (STATS) tst.sphs++;
                        ok = hit_sphere(make_float4(hr.a[0], hr.a[1], hr.a[2], hr.e1[0]), p, wi, &t) && t > 0.0f && t < d_s &&
                             slab_rel(br, hot_dn[h].x, hot_dn[h].y, hot_dn[h].z, hot_dx[h].x, hot_dx[h].y, hot_dx[h].z);
                    }
                    blocked = ok;
                }
            }
            // a ray not settled here leaves a record: k_trace_p searches the occlusion BVH for a blocker; the few with a 0 * inf = NaN
            // case in their slab test, or far-away coordinates, go to k_binary and the binary tree, which has the crate's NaN behaviour
            const bool unblocked = live && !blocked;
            // rays of a clear beam are settled: k_trace_p never sees them (k_binary's rays are k_binary's either way)
            const bool to_trace = unblocked && !(clear && !binary);
            const bool summed = unblocked && !to_trace && sum_open;
            sum_open = sum_open && !to_trace;
            const bool keep = unblocked && !summed;             // leaves a record
            const unsigned long long m = __ballot(keep);      // the whole wave is here: one row of 64 slots
            const unsigned long long mt = __ballot(to_trace);
            if (STATS) { sh_rays += (unblocked && !to_trace) ? 1u : 0u; sh_summed += summed ? 1u : 0u; }
            if (summed) color = add(color, light_term(lp, base, nrm, wi, d_s));
            if (keep) {
                has_rec = true;
                // place in the row: the rays still to trace first, in lane order, then the settled ones (k_trace_p reads a dense run of records)
                const unsigned long long mine = to_trace ? mt : (m & ~mt);
                const size_t at = row + (to_trace ? 0u : (uint32_t)__popcll(mt)) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
                pc.lray[at] = make_float4(wi.x, wi.y, wi.z, d_s);
                if (to_trace) {
                    const float sib_i = sib_axis == 0u ? br.ix : (sib_axis == 1u ? br.iy : br.iz);
                    const bool skip_sib = has_sib && !(sib_d * sib_i > 0.0f);
                    pc.linv[at] = make_float4(br.ix, br.iy, br.iz, __uint_as_float(self_rec | (__lane_id() << RT_LR_LANE_SHIFT) | (binary ? RT_LR_BINARY : 0u) |
                                                                                   (skip_sib ? RT_LR_SKIP_SIB : 0u)));
                    if (binary) pc.blist[atomicAdd(&pc.bcounts[level], 1u)] = (uint32_t)at;
                }
            }
            if (__lane_id() == 0u) { row_mask[j] = m; row_mask_t[j] = mt; }
          }
          if (live) pc.direct[(size_t)g * pc.vstride + e] = make_float4(color.x, color.y, color.z, has_rec ? 1.0f : 0.0f);
        }
    }
    flush_stats<STATS>(pc.stats, tst, ST_S_INNER);
    if (STATS) {
        for (int off = 32; off > 0; off >>= 1) {
            sh_tried += __shfl_xor(sh_tried, off); sh_clear += __shfl_xor(sh_clear, off);
            sh_steps += __shfl_xor(sh_steps, off); sh_rays += __shfl_xor(sh_rays, off); sh_summed += __shfl_xor(sh_summed, off);
        }
        if (__lane_id() == 0u && sh_summed) atomicAdd(&pc.stats[ST_L_SUMMED], (unsigned long long)sh_summed);
        if (__lane_id() == 0u && sh_tried) {
            atomicAdd(&pc.stats[ST_SHAFT_TRIED], (unsigned long long)sh_tried); atomicAdd(&pc.stats[ST_SHAFT_CLEAR], (unsigned long long)sh_clear);
            atomicAdd(&pc.stats[ST_SHAFT_STEPS], (unsigned long long)sh_steps); atomicAdd(&pc.stats[ST_SHAFT_RAYS], (unsigned long long)sh_rays);
        }
    }
}

// The light-sample rays k_light put on pc.blist (normally none): Scene::is_occluded on the binary tree, self test included.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) k_binary(DevScene sc, PassCtx pc, uint32_t level) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    const uint32_t n = pc.bcounts[level];
    uint32_t* words = reinterpret_cast<uint32_t*>(pc.lray);
    for (uint32_t i0 = blockIdx.x * RT_BLOCK + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * RT_BLOCK) {
        const uint32_t i = i0 + (threadIdx.x & 63u);
        const bool have = i < n;
        vec3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f);
        float max_dist = 0.0f;
        uint32_t self_obj = RT_MISS;
        size_t at = 0;
        if (have) {
            at = pc.blist[i];
            const float4 R4 = pc.lray[at];
            const uint32_t e = (uint32_t)(at / (64u * (size_t)pc.nlL)) * 64u + ((__float_as_uint(pc.linv[at].w) >> RT_LR_LANE_SHIFT) & 63u);
            o = xyz(pc.vP[e]); d = xyz(R4); max_dist = R4.w;
            if (sc.self_test) self_obj = __float_as_uint(pc.vN[e].w) - 1u;
        }
        bool occ = false;
        traverse<1, STATS>(sc, have, o, d, max_dist, self_obj, st, nullptr, nullptr, &occ, &tst);
        if (have && occ) words[4 * at + 3] |= RT_LR_BLOCKED;
    }
    flush_stats<STATS>(pc.stats, tst, ST_S_INNER);
}

// Position of the k-th set bit of m (k < popcount(m)).
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long m, uint32_t k) {
    uint32_t w = (uint32_t)m, pos = 0u;
    const uint32_t c0 = (uint32_t)__popc(w);
    if (k >= c0) { k -= c0; w = (uint32_t)(m >> 32); pos = 32u; }
#pragma unroll
    for (uint32_t s = 16u; s != 0u; s >>= 1) {
        const uint32_t c = (uint32_t)__popc(w & ((1u << s) - 1u));
        if (k >= c) { k -= c; w >>= s; pos += s; }
    }
    return pos;
}

// Wave state of k_trace_p: a wave owns windows of 64 consecutive rows (window = wave index, then + number of waves); a row is the 64
// slots of (vertex tile, light sample).  After k_light's beams most rows hold only a few rays to trace, so the wave takes the masks of a
// whole window at once (one row per lane), counts them with a prefix sum and hands ray number i of the window — found by a binary search
// over the prefix sums and a select of the i-th set bit — to its idle lanes as they become free.
template <bool STATS, int WAVES>   // WAVES: waves per SIMD the register budget is sized for
__global__ void __launch_bounds__(RT_BLOCK, WAVES) k_trace_p(DevScene sc, PassCtx pc, uint32_t level) {
    RT_STACK_DECL;
    // the current window of this wave: per row the rays still to trace (their records lead the row, in lane order) and the number of rays in the rows before
    __shared__ unsigned long long s_mt[RT_BLOCK];
    __shared__ uint32_t s_pre[RT_BLOCK];
    const uint32_t wbase = threadIdx.x & ~63u;
    TravStats tst = {0, 0, 0, 0, 0};
    const uint32_t nlL = pc.nlL;
    const uint32_t n_rows = ((pc.vcounts[level] + 63u) >> 6) * nlL;             // < 2^26 (render_impl caps paths * nlL at 2^31)
    const uint32_t n_win = (n_rows + 63u) >> 6;
    const uint32_t n_waves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t lane = __lane_id();
    uint32_t win = __builtin_amdgcn_readfirstlane((blockIdx.x * RT_BLOCK + threadIdx.x) >> 6);   // the window whose masks are in mt_n / mf_n
    uint32_t w_row0 = 0u, w_total = 0u, w_cursor = 0u;                         // the window in LDS: first row, rays to trace, rays handed out
    const uint32_t refill_min = sc.refill_min, phase_bias = sc.phase_bias;
    uint32_t* words = reinterpret_cast<uint32_t*>(pc.lray);
    // masks of this lane's row of the next window (k_light's votes), loaded one window ahead
    unsigned long long mt_n = (win < n_win && win * 64u + lane < n_rows) ? pc.tmask_t[win * 64u + lane] : 0ull;

    uint32_t w_steps[6] = {0, 0, 0, 0, 0, 0};   // STATS: wave steps and the lanes that took part (node, primitive, refill)

    // the ray this lane is tracing (cur != 0)
    uint32_t cur = 0u, slot = 0u, rec_word = 0u, skip_rec = RT_W_NO_REC, skip_rec2 = RT_W_NO_REC;
    vec3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f);
    BoxRay br = make_box_ray(o, d);
    float max_dist = 0.0f, lim = 0.0f;

    for (;;) {
        const unsigned long long idle = __ballot(cur == 0u);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if ((w_cursor < w_total || win < n_win) && n_idle >= refill_min) {
            if (w_cursor == w_total) {                           // next window: its masks to LDS, one row per lane, and the prefix sums of their counts
                const uint32_t c = (uint32_t)__popcll(mt_n);
                uint32_t inc = c;
#pragma unroll
                for (uint32_t off = 1u; off < 64u; off <<= 1) {
                    const uint32_t t = __shfl_up(inc, off);
                    if (lane >= off) inc += t;
                }
                __builtin_amdgcn_wave_barrier();                 // (the lanes that were reading the previous window are all here)
                s_mt[wbase + lane] = mt_n;
                s_pre[wbase + lane] = inc - c;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                w_row0 = win * 64u;
                w_total = __builtin_amdgcn_readlane(inc, 63);
                w_cursor = 0u;
                win += n_waves;
                mt_n = (win < n_win && win * 64u + lane < n_rows) ? pc.tmask_t[win * 64u + lane] : 0ull;
                if (w_total == 0u) continue;
            }
            const uint32_t left = w_total - w_cursor;
            const uint32_t take = left < n_idle ? left : n_idle;
            const uint32_t irank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (STATS) { w_steps[4]++; w_steps[5] += take; }
            if (cur == 0u && irank < take) {
                const uint32_t idx = w_cursor + irank;           // this lane's ray among the window's
                uint32_t r = 0u;                                 // its row: the last one with s_pre[row] <= idx
#pragma unroll
                for (uint32_t step = 32u; step != 0u; step >>= 1)
                    if (s_pre[wbase + r + step] <= idx) r += step;
                const uint32_t k = idx - s_pre[wbase + r];      // ... and its place among the row's rays to trace = the place of its record
                const uint32_t src = nth_set_bit(s_mt[wbase + r], k);             // its vertex within the tile
                slot = (w_row0 + r) * 64u + k;
                const uint32_t e = ((w_row0 + r) / nlL) * 64u + src;
                const float4 R4 = pc.lray[slot], I4 = pc.linv[slot];
                const float4 P4 = pc.vP[e];
                rec_word = __float_as_uint(I4.w);
                o = xyz(P4); d = xyz(R4); max_dist = R4.w;
                br.ox = o.x; br.oy = o.y; br.oz = o.z;
                br.ix = I4.x; br.iy = I4.y; br.iz = I4.z;                     // make_box_ray(o, d), computed by k_light
                skip_rec = rec_word & RT_W_NO_REC;                             // the object the ray starts on: tested by k_light
                skip_rec2 = (rec_word & RT_LR_SKIP_SIB) ? (sc.wsib[skip_rec] & RT_W_NO_REC) : RT_W_NO_REC;
                br.sx = I4.x < 0.0f; br.sy = I4.y < 0.0f; br.sz = I4.z < 0.0f; // sign of the normalised direction (finite 1/d only: wide_ray_ok)
                lim = prune_limit(max_dist);
                st.sp = 0;
                cur = (rec_word & RT_LR_BINARY) ? 0u : 1u;                     // inner group {node 0}; k_binary's rays are not traced here
            }
            w_cursor += take;
            continue;
        }
        if (n_idle == 64u) break;
        const bool at_node = cur != 0u && !(cur & RT_W_PRIM);
        const bool at_prim = (cur & RT_W_PRIM) != 0u;
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node)), n_prim = (uint32_t)__popcll(__ballot(at_prim));
        if (n_node != 0u && (n_prim == 0u || 4u * n_node >= phase_bias * n_prim)) {
            if (STATS) { w_steps[0]++; w_steps[1] += n_node; }
            if (at_node) wide_node_step<STATS>(sc, br, lim, skip_rec, skip_rec2, cur, st, &tst);
        } else {
            if (STATS) { w_steps[2]++; w_steps[3] += n_prim; }
            if (at_prim && wide_prim_step<STATS>(sc, br, o, d, max_dist, cur, st, &tst)) words[4 * (size_t)slot + 3] = __float_as_uint(max_dist) | RT_LR_BLOCKED;
        }
    }
    flush_stats<STATS>(pc.stats, tst, ST_S_INNER);
    if (STATS) {
        uint32_t bm = tst.box_miss;
        for (int off = 32; off > 0; off >>= 1) bm += __shfl_xor(bm, off);
        if (lane == 0u) {
            for (int k = 0; k < 6; ++k) atomicAdd(&pc.stats[ST_P_NODE_STEPS + k], (unsigned long long)w_steps[k]);
            atomicAdd(&pc.stats[ST_P_BOX_MISS], (unsigned long long)bm);
        }
    }
}

// The rest of one_bounce_radiance_importance and global_illumination (src/raytracer.rs:282-326).
// k_shade_b waits on memory (84 % of its wave cycles): 8 resident waves per SIMD (a 64-register budget, 60 B of spills) with 2 light-sample
// records in flight per lane (round 2, when it read 27 % of the light samples back: 1 in flight + 19 %, 3: + 2 %, 4: + 5 %, 6 waves + 15 %; now
// that it reads 4 %, 3 or 4 in flight or 6 waves change nothing).  What it waits for is the path bookkeeping, scattered by path slot.
#ifndef RT_SB_WAVES
#define RT_SB_WAVES 8
#endif
#ifndef RT_SB_BATCH
#define RT_SB_BATCH 2
#endif
#ifndef RT_SB_TILES
#define RT_SB_TILES 4          // vertices per thread and queue reservation
#endif
__global__ void __launch_bounds__(RT_BLOCK, RT_SB_WAVES) k_shade_b(DevScene sc, PassCtx pc, uint32_t level) {
    const uint32_t n = pc.vcounts[level];
    const uint32_t bounces_left = pc.P.bounces - level;
    float4* no = pc.ray_o[(level + 1) & 1];
    float4* nd = pc.ray_d[(level + 1) & 1];
    const uint32_t L = pc.P.light_samples;
    const float inv_light_samples = 1.0f / (float)L;
    // every thread of a workgroup runs the same number of iterations (block_append has barriers)
    __shared__ uint32_t s_tmp[RT_SB_TILES * (RT_BLOCK / 64) + 1];
    const ulonglong2 jm_path = pc.jump[pc.nlL];
    for (uint32_t eb = blockIdx.x * (RT_BLOCK * RT_SB_TILES); eb < n; eb += gridDim.x * (RT_BLOCK * RT_SB_TILES)) {
      // Which of this thread's RT_SB_TILES vertices send their path on is a coin flip on the path's RNG state (weighted_coin_flip, below): it is taken
      // for all of them first, so that ONE atomic reserves the workgroup's run in the next ray queue (block_append_n); then each vertex in turn.
      bool goes_on[RT_SB_TILES];
      uint32_t q_next[RT_SB_TILES];
#pragma unroll
      for (int t = 0; t < RT_SB_TILES; ++t) {
          const uint32_t e = eb + (uint32_t)t * RT_BLOCK + threadIdx.x;
          goes_on[t] = false;
          if (e < n && bounces_left != 1) {
              uint64_t rng = jm_path.x * pc.vR[e] + jm_path.y;
              goes_on[t] = next_f32(&rng) <= RT_ROULETTE;
          }
      }
      block_append_n<RT_SB_TILES>(goes_on, &pc.counts[level + 1], s_tmp, q_next);
#pragma unroll 1
      for (uint32_t t = 0; t < RT_SB_TILES; ++t) {
        const uint32_t e0 = eb + t * RT_BLOCK;
        if (e0 >= n) break;
        const uint32_t e = e0 + threadIdx.x;
        uint32_t q = q_next[0];
#pragma unroll
        for (int j = 1; j < RT_SB_TILES; ++j) q = t == (uint32_t)j ? q_next[j] : q;
        // whole tiles of 64: the idle lanes of a level's last tile run along on its last vertex (the row masks below are handed from lane to
        // lane, which takes every lane awake) and store nothing
        if ((e & ~63u) < n) {
            const bool live = e < n;
            const uint32_t ev = live ? e : n - 1u;
            const float4 N4 = pc.vN[ev];
            {
                const float4 P4 = pc.vP[ev];
                const uint32_t mat = __float_as_uint(P4.w);
                const float4 m0 = sc.mats[2 * mat];
                const vec3 p = xyz(P4), nrm = xyz(N4);
                vec3 l = mk(0.0f, 0.0f, 0.0f);
                if (pc.persist) {
                    // one_bounce_radiance_importance's sample loop (raytracer.rs:266-283) over the rays k_light / k_trace_p / k_binary left unblocked;
                    // a blocked sample adds nothing.  pdf (objects.rs:109-115) and |wi.n| (raytracer.rs:278) are worked out here, for those only.
                    const vec3 f = __float_as_uint(m0.x) == RT_BSDF_DIFFUSE ? scale(mk(m0.y, m0.z, m0.w), 1.0f / RT_PI) : mk(0.0f, 0.0f, 0.0f);   // Object::bsdf
                    const uint32_t lane = __lane_id();
                    for (uint32_t g = 0; g < pc.n_groups; ++g) {
                        const uint32_t light = sc.lights[g];
                        const float4 cr = sc.sph[light - sc.n_tri];
                        const vec3 le = xyz(sc.mats[2 * (sc.sph_aux[light - sc.n_tri].y & RT_INFO_MAT) + 1]);
                        const vec3 base = mul3(le, f);
                        const LightPdf lp = light_pdf_setup(p, cr);
                        // the samples k_light summed itself (the leading ones of a clear beam); the rows hold the rest, if any
                        const float4 pre = pc.direct[(size_t)g * pc.vstride + ev];
                        vec3 color = xyz(pre);
                        if (__ballot(live && pre.w != 0.0f) == 0ull) { l = add(l, scale(color, inv_light_samples)); continue; }     // no record in this wave's rows
                        const size_t row0 = ls_slot(e & ~63u, g * L, pc.nlL);                                  // first slot of the row of sample s: + 64 s
                        const unsigned long long* mrow = pc.tmask + (size_t)(e >> 6) * pc.nlL + (size_t)g * L;  // mask of that row: + s
                        const unsigned long long* mrow_t = pc.tmask_t + (size_t)(e >> 6) * pc.nlL + (size_t)g * L;   // ... and which of them were traced
                        // The row masks of 32 samples at a time, one per lane (lanes 0..31: tmask, 32..63: tmask_t) in ONE load, then handed round by
                        // v_readlane: the sample loop's only trips to memory are the records themselves.
                        for (uint32_t sb = 0; sb < L; sb += 32u) {
                            const uint32_t cnt = L - sb < 32u ? L - sb : 32u;
                            unsigned long long mword = 0ull;
                            if ((lane & 31u) < cnt) mword = (lane < 32u ? mrow : mrow_t)[sb + (lane & 31u)];
                            const uint32_t mw_lo = (uint32_t)mword, mw_hi = (uint32_t)(mword >> 32);
                            for (uint32_t s0 = 0; s0 < cnt; s0 += RT_SB_BATCH) {                   // RT_SB_BATCH samples in flight, then added in sample order
                                unsigned long long mk8[RT_SB_BATCH], mt8[RT_SB_BATCH];
                                uint32_t w[RT_SB_BATCH];
                                float4 r[RT_SB_BATCH];
#pragma unroll
                                for (uint32_t k = 0; k < RT_SB_BATCH; ++k) {
                                    const uint32_t si = s0 + k;       // (< 32 + RT_SB_BATCH; the lanes read beyond cnt hold 0 or are masked by the test)
                                    mk8[k] = si < cnt ? ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mw_hi, (int)(si & 31u)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)mw_lo, (int)(si & 31u)) : 0ull;
                                    mt8[k] = si < cnt ? ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mw_hi, (int)(32u + (si & 31u))) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)mw_lo, (int)(32u + (si & 31u))) : 0ull;
                                }
#pragma unroll
                                for (uint32_t k = 0; k < RT_SB_BATCH; ++k) {
                                    w[k] = RT_LR_BLOCKED;
                                    r[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                    if ((mk8[k] >> lane) & 1ull) {
                                        // the row holds the traced rays first, then the ones a clear beam settled, each group in lane order (k_light)
                                        const unsigned long long below = (1ull << lane) - 1ull;
                                        const uint32_t place = ((mt8[k] >> lane) & 1ull) ? (uint32_t)__popcll(mt8[k] & below) : (uint32_t)(__popcll(mt8[k]) + __popcll(mk8[k] & ~mt8[k] & below));
                                        const size_t at = row0 + 64 * (size_t)(sb + s0 + k) + place;
                                        r[k] = pc.lray[at];
                                        w[k] = __float_as_uint(r[k].w);
                                    }
                                }
#pragma unroll
                                for (uint32_t k = 0; k < RT_SB_BATCH; ++k)
                                    if (!(w[k] & RT_LR_BLOCKED)) color = add(color, light_term(lp, base, nrm, xyz(r[k]), r[k].w));
                            }
                        }
                        l = add(l, scale(color, inv_light_samples));                // raytracer.rs:282
                    }
                } else if (pc.in_wave) {
                    l = xyz(pc.direct[ev]);                                          // summed by k_shadow
                } else {
                    const float4* cp = pc.contrib + (size_t)ev * pc.nlL;
                    for (uint32_t g = 0; g < pc.n_groups; ++g) {
                        vec3 color = mk(0.0f, 0.0f, 0.0f);
                        for (uint32_t s = 0; s < L; ++s) {
                            const float4 c = cp[g * L + s];
                            if (c.w != 0.0f) color = add(color, xyz(c));
                        }
                        l = add(l, scale(color, inv_light_samples));                // raytracer.rs:282
                    }
                }
                // (what only the rest needs is fetched here, after the sample loop: k_shade_b waits on memory, and registers held across
                // the loop cost it resident waves)
                if (live) {
                    const float4 S4 = pc.vS[e], m1 = sc.mats[2 * mat + 1];
                    const uint32_t slot = __float_as_uint(S4.x);
                    l = add(l, xyz(m1));                                                // raytracer.rs:284
                    if (bounces_left == 1) {
                        fold_path(pc, slot, level, l);                                  // raytracer.rs:340-344
                    } else {
                        uint64_t rng = jm_path.x * pc.vR[e] + jm_path.y;
                        if (!(next_f32(&rng) <= RT_ROULETTE)) {                         // weighted_coin_flip, common.rs:136-138
                            fold_path(pc, slot, level, l);
                        } else {
                            const vec3 wo = mk(S4.y, S4.z, S4.w);
                            vec3 wi, reflected;
                            float pdf;
                            const vec3 refl = mk(m0.y, m0.z, m0.w);
                            if (__float_as_uint(m0.x) == RT_BSDF_DIFFUSE) {             // Object::sample_bsdf, objects.rs:130-148
                                wi = to_coord_space(random_hemisphere(&rng), nrm);
                                pdf = 2.0f * RT_PI;
                                reflected = scale(refl, 1.0f / RT_PI);
                            } else {
                                wi = sub(wo, scale(scale(nrm, 2.0f), dot3(wo, nrm)));
                                pdf = 1.0f;
                                reflected = scale(refl, 1.0f / fabsf(dot3(wi, nrm)));
                            }
                            const float cos_theta = fabsf(dot3(wi, nrm));               // raytracer.rs:321
                            float4* rec = pc.stack + 2 * ((size_t)slot * pc.stack_depth + level);
                            rec[0] = make_float4(l.x, l.y, l.z, cos_theta);
                            rec[1] = make_float4(reflected.x, reflected.y, reflected.z, pdf);
                            pc.qrng[(level + 1) & 1][q] = rng;
                            const vec3 dn = unit3(wi);                                  // Ray::new, geo.rs:19-24
                            no[q] = make_float4(p.x, p.y, p.z, __uint_as_float(slot));
                            nd[q] = make_float4(dn.x, dn.y, dn.z, 0.0f);
                        }
                    }
                }
            }
        }
      }
    }
}

// render_helper's sample loop and set_pixel (src/raytracer.rs:235-240, 43-51).
__global__ void __launch_bounds__(RT_BLOCK) k_resolve(PassCtx pc) {
    const rt_render_params& P = pc.P;
    for (uint32_t pl = blockIdx.x * RT_BLOCK + threadIdx.x; pl < pc.npix; pl += gridDim.x * RT_BLOCK) {
        uint32_t x, y;
        if (!own_pixel_xy(P, pc.tiles_x, pc.pix0 + pl, &x, &y)) continue;
        float4 a = pc.s0 == 0 ? make_float4(0, 0, 0, 0) : pc.accum[pc.pix0 + pl];
        for (uint32_t s = 0; s < pc.spc; ++s) {
            const float4 r = pc.radiance[(size_t)pl * pc.spc + s];
            a.x += r.x; a.y += r.y; a.z += r.z;
        }
        const bool last = pc.s0 + pc.spc >= P.spp;
        if (!last) pc.accum[pc.pix0 + pl] = a;
        if (last || pc.preview) {
            // the pixel after its last sample (raytracer.rs:238-240); with `preview`, also after every earlier chunk of samples: the mean of the
            // samples so far, i.e. the image the reference would give at that many spp, so that a canvas polling the buffer (canvas.rs:179) sees
            // the frame refine
            const float inv = 1.0f / (float)(last ? P.spp : pc.s0 + pc.spc);
            const uint32_t r8 = to_color(a.x * inv), g8 = to_color(a.y * inv), b8 = to_color(a.z * inv);
            reinterpret_cast<uint32_t*>(pc.rgba)[(size_t)y * P.width + x] = r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
        }
    }
}

// Arbitrary rays (rt_intersect_rays / rt_occluded_rays).
__global__ void __launch_bounds__(RT_BLOCK) k_rays_closest(DevScene sc, const float* org, const float* dir, uint32_t n, uint32_t* ids, float* ts_out) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        uint32_t id;
        float t;
        traverse<0, false>(sc, true, mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), 0.0f, RT_MISS, st, &id, &t, nullptr, &tst);
        ids[i] = id;
        ts_out[i] = t;
    }
}
__global__ void __launch_bounds__(RT_BLOCK) k_rays_occluded(DevScene sc, const float* org, const float* dir, const float* maxd, uint32_t n, uint8_t* out) {
    RT_STACK_DECL;
    TravStats tst = {0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        const bool occ = occluded<false>(sc, true, mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]), mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), maxd[i], RT_MISS, st, &tst);
        out[i] = occ ? 1 : 0;
    }
}

// Screen-tile pack / unpack for the multi-GPU gather.  words = bytes_per_pixel / 4.
__global__ void k_tiles_copy(rt_render_params P, uint32_t tiles_x, uint32_t shard, uint32_t n_slots, const uint32_t* src, uint32_t* dst,
                             uint32_t words, int unpack) {
    const uint32_t tp = P.tile_w * P.tile_h;
    const size_t total = (size_t)n_slots * tp;
    rt_render_params Q = P;
    Q.shard_index = shard;
    for (size_t op = (size_t)blockIdx.x * blockDim.x + threadIdx.x; op < total; op += (size_t)gridDim.x * blockDim.x) {
        uint32_t x, y;
        const uint32_t tiles_total = tiles_x * ((P.height + P.tile_h - 1) / P.tile_h);
        const uint32_t t = shard + (uint32_t)(op / tp) * P.shard_count;
        if (t >= tiles_total) continue;
        if (!own_pixel_xy(Q, tiles_x, (uint32_t)op, &x, &y)) continue;
        const size_t pi = (size_t)y * P.width + x;
        for (uint32_t w = 0; w < words; ++w) {
            if (unpack) dst[pi * words + w] = src[op * words + w];
            else dst[op * words + w] = src[pi * words + w];
        }
    }
}

__global__ void k_debug_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
        case 0: r = sin_f(x[i]); break;
        case 1: r = cos_f(x[i]); break;
        case 2: r = acos_f(x[i]); break;
        case 3: r = pow_f(x[i], y[i]); break;
        case 5: { float c; sincos_s(x[i], &r, &c); break; }
        case 6: { float sn; sincos_s(x[i], &sn, &r); break; }
        case 7: r = acos_s(x[i]); break;
        default: r = (float)to_color(x[i]); break;
    }
    out[i] = r;
}
// Exhaustive equality of the fast IEEE forms with the compiler's: counts inputs whose results differ in any bit (two NaNs count as equal).
// op 0: rcp_ieee(x) vs 1.0f / x, op 1: sqrt_ieee(x) vs sqrtf(x), every one of the 2^32 floats; op 2: div_ieee(a, b) vs a / b with b every mantissa at
// five exponents and 1024 numerators each (hashed, of every exponent class); op 3: unit3_ieee / make_box_ray vs the plain forms on 2^32 hashed vectors.
// out[0] = mismatches, out[1] = one offending input (bits; 64-bit for op 2).
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ __forceinline__ uint32_t hash32(uint64_t z) { return (uint32_t)(mix64(z) >> 16); }
__global__ void k_debug_exact(uint32_t op, unsigned long long* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, which = 0;
    if (op <= 1u) {
        for (uint64_t b = tid; b < (1ull << 32); b += nthreads) {
            const float x = __uint_as_float((uint32_t)b);
            const bool ok = op == 0u ? same_bits(rcp_ieee(x), 1.0f / x) : same_bits(sqrt_ieee(x), sqrtf(x));
            if (!ok) { ++bad; which = b; }
        }
    } else if (op == 2u) {
        const uint32_t exps[5] = {127u - 60u, 127u - 1u, 127u, 127u + 37u, 127u + 59u};
        for (uint64_t i = tid; i < 5ull * (1ull << 23); i += nthreads) {
            const uint32_t bb = (exps[i >> 23] << 23) | (uint32_t)(i & 0x7FFFFFu) | ((i & 1u) ? 0x80000000u : 0u);
            const float b = __uint_as_float(bb);
            for (uint32_t k = 0; k < 1024u; ++k) {
                uint32_t ab = hash32(i * 1024u + k);
                if ((k & 7u) == 0u) ab = (ab & 0x807FFFFFu) | (bb & 0x7F800000u);          // same exponent as b: quotients near 1
                if ((k & 63u) == 1u) ab = (k & 64u) ? 0x80000000u : 0u;                     // signed zeros
                const float a = __uint_as_float(ab);
                if (!same_bits(div_ieee(a, b), a / b)) { ++bad; which = ((unsigned long long)ab << 32) | bb; }
            }
        }
    } else {
        for (uint64_t i = tid; i < (1ull << 32); i += nthreads) {
            // directions as the path tracer makes them: near-unit vectors, some with tiny, zero or huge components
            const uint32_t h0 = hash32(3 * i), h1 = hash32(3 * i + 1), h2 = hash32(3 * i + 2);
            vec3 d = mk(__uint_as_float((h0 & 0x807FFFFFu) | 0x3F000000u), __uint_as_float((h1 & 0x807FFFFFu) | 0x3E800000u), __uint_as_float((h2 & 0x807FFFFFu) | 0x3F000000u));
            if ((i & 15u) == 0u) d.x = __uint_as_float(h0);                                  // any float at all
            if ((i & 63u) == 1u) d.y = (i & 64u) ? -0.0f : 0.0f;
            if ((i & 255u) == 2u) d.z = __uint_as_float(h2 & 0x80FFFFFFu);                   // denormal or tiny
            const vec3 u0 = unit3(d), u1 = unit3_ieee(d);
            const BoxRay br = make_box_ray(mk(0.0f, 0.0f, 0.0f), d);
            const vec3 n = unit3(d);
            const bool ok = same_bits(u0.x, u1.x) && same_bits(u0.y, u1.y) && same_bits(u0.z, u1.z) && same_bits(br.ix, 1.0f / n.x) &&
                            same_bits(br.iy, 1.0f / n.y) && same_bits(br.iz, 1.0f / n.z);
            if (!ok) { ++bad; which = i; }
        }
    }
    if (bad) { atomicAdd(&out[0], bad); out[1] = which; }
}
__global__ void k_debug_rng(uint64_t seed_mixed, uint32_t pixel_index, uint32_t sample, uint32_t n, float* out) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t s = stream_seed(seed_mixed, pixel_index, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = next_f32(&s);
}

}  // namespace rt
