// rt_api.hip — implementation of the C ABI in include/rustracer_hip.h: scene upload, pass
// scheduling of the wavefront kernels (rt_kernels.h) and the debug probes.  gfx950 only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/rustracer_hip.h"
#include "host_scene.h"
#include "rt_build.h"
#include "rt_kernels.h"

#include <chrono>

using namespace rt;

namespace {

thread_local std::string g_error;
// The GPU of this process, chosen by rt_init.  hipSetDevice binds only the calling host thread, and callers render from
// threads of their own (Raytracer::render(false) spawns one, src/raytracer.rs:527), so every entry point that touches HIP
// selects the device again: the scene's (recorded at creation) or, for calls without a scene, the process's.
int g_device = 0;
bool g_inited = false;          // rt_init has chosen a GPU: scenes may be built there
uint32_t g_device_build = 1;    // rt_set_global_option("device_build"): build the trees of scenes with >= g_device_build_min objects on the GPU
uint32_t g_device_build_min = 4096;
uint64_t g_alloc_limit = 0;      // rt_set_global_option("alloc_limit"), test hook: a pass buffer larger than this many bytes fails to allocate (0: no limit)

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

#define BIND(dev) HIP_TRY(hipSetDevice(dev))

// Scratch of rt_cast_primary_noalloc (include/rustracer_hip.h): the two pieces of device memory the primary-cast kernel uses
// beyond its inputs and outputs, which rt_cast_primary_device takes from hipMalloc'ed buffers of the scene (w_stats, d_ovf):
//   bytes [0, 256)   the work counters of the call (ST_COUNT x u64), cleared at the start of every call;
//   bytes [256, ...) the traversal-stack overflow rows, [row][global thread]: RT_NOALLOC_ROWS rows for a grid of at most
//                    RT_NOALLOC_GRID workgroups (entries below RT_STACK_LDS live in LDS); written before they are read.
// One copy per GPU, held until the process ends.  The array's name is only a host-side stand-in for the symbol: the device
// address comes from hipGetSymbolAddress, once per device with that device current (scratch_for), and is what kernels get.
#define RT_NOALLOC_ROWS 16u
#define RT_NOALLOC_GRID 1024u
#define RT_NOALLOC_BYTES (256u + RT_NOALLOC_ROWS * RT_NOALLOC_GRID * RT_BLOCK * 4u)   // 16 MiB + 256 B
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
static_assert(ST_COUNT * sizeof(unsigned long long) <= 256, "the counters must fit the first 256 bytes of the scratch");
void* g_scratch_ptr[RT_MAX_DEVICES];   // device address of `scratch` on each GPU this process has asked for (nullptr: not yet)
const void* g_scratch_owner[RT_MAX_DEVICES];   // the scene whose rt_cast_primary_noalloc call wrote the counters in that GPU's scratch last

// Device address of this GPU's copy of `scratch`; `device` must be the current device.
int scratch_for(int device, void** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) return fail(RT_ERR_LIMIT, "device index beyond the scratch table");
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return RT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (g_alloc_limit && (uint64_t)n * sizeof(T) > g_alloc_limit) return fail(RT_ERR_HIP, "buffer beyond the 'alloc_limit' test hook");
        HIP_TRY(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return RT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Pass buffers of one render lane (rt_scene::ws).  A frame is rendered by up to RT_MAX_LANES lanes, each on its own HIP stream over its
// own part of the owned pixels: the kernels that wait on memory in one lane (k_shade_*, k_extend_p) run under the VALU-bound ones of
// the other (k_light, k_trace_p).  Grow-only; no allocation in steady state.
#define RT_MAX_LANES 4
struct Workspace {
    DevBuf<uint32_t> counts, hit_id, blist;
    DevBuf<float> hit_t;
    DevBuf<float4> ray_o[2], ray_d[2], vP, vN, vS, contrib, direct, stack, radiance, lray, linv;
    DevBuf<uint64_t> rng[3];     // the paths' RNG states: per queued ray (two queues) and per vertex
    DevBuf<unsigned long long> tmask, tmask_t;
    hipStream_t stream = nullptr;   // lanes run on these when there is more than one
    hipEvent_t done = nullptr;
    hipEvent_t light = nullptr;     // after this lane's last k_light (light_token)
    void release() {
        counts.release(); hit_id.release(); blist.release(); hit_t.release();
        for (int q = 0; q < 2; ++q) { ray_o[q].release(); ray_d[q].release(); }
        vP.release(); vN.release(); vS.release(); contrib.release(); direct.release(); stack.release(); radiance.release();
        lray.release(); linv.release(); for (auto& r : rng) r.release(); tmask.release(); tmask_t.release();
    }
};

struct EventPair {
    hipEvent_t a, b;
    int cls;   // 0 closest, 1 shadow, 2 shade/other (k_binary too), 3 light sampling (k_light alone)
};

}  // namespace

struct rt_comm;

struct rt_scene {
    int device = 0;   // the GPU this scene lives on (g_device when it was created)
    HostScene host;
    FlatBVH bvh;
    WideBVH wide;
    std::vector<uint32_t> lights;
    bool has_triangle_light = false;
    bool uploaded = false;
    double build_dev_ms[2] = {0.0, 0.0};         // of which between HIP events (uploads, kernels, downloads; not the allocations)
    double build_ms[2] = {0.0, 0.0};             // wall time of the binary / 8-wide tree build
    uint32_t build_on_device[2] = {0u, 0u};      // ... and whether the GPU built it
    std::string build_note;                      // why the GPU did not (when it was asked to)
    uint32_t ovf_rows = 0;                       // traversal-stack rows beyond RT_STACK_LDS that the trees of this scene can need
    const unsigned long long* stats_src = nullptr;   // where the counters of the last call live (w_stats, or the no-alloc scratch)

    DevBuf<float4> d_nodes, d_tris, d_tri_nrm, d_sph, d_mats, d_wnodes, d_wprims;
    DevBuf<uint2> d_sph_aux;
    DevBuf<uint32_t> d_lights, d_ovf, d_wrec, d_wsib;
    HotTable hot_table;
    DevScene dev;
    uint64_t device_bytes = 0;

    Workspace ws[RT_MAX_LANES];                  // per-lane pass buffers
    hipEvent_t ev_fork = nullptr;                // the caller's stream at the start of a multi-lane render
    DevBuf<float4> w_accum;                      // per-pixel sums across sample chunks (lanes own disjoint pixels)
    DevBuf<ulonglong2> w_jump;
    uint32_t jump_nlL = 0xFFFFFFFFu;
    DevBuf<unsigned long long> w_stats;
    DevBuf<uint8_t> w_frame;
    DevBuf<uint32_t> w_ids;
    DevBuf<float> w_ts;
    DevBuf<float> w_rays;   // scratch for arbitrary-ray calls

    bool opt_stats = false, opt_timing = false;
    uint64_t opt_max_paths = 1ull << 26;
    uint64_t max_paths_fit = 0;                  // the pass size the device could hold when a larger one did not fit (0: none failed yet); reset by "max_paths"
    uint32_t opt_inner_min = 12, opt_self_test = 1, opt_wide = 1, opt_persist = 1, opt_refill_min = 16, opt_phase_bias = 16, opt_shadow_waves = 6, opt_light_waves = 4, opt_hot = 1, opt_extend_persist = 1, opt_extend_refill = 24, opt_extend_inner = 20, opt_extend_grid = 8, opt_lanes = 2, opt_shaft = 6, opt_graze_guard = 0, opt_trace_grid = 8, opt_light_token = 0, opt_light_sum = 1, opt_force_binary = 0, opt_camera_once = 1, opt_light_grid = 0;
    uint64_t opt_lane_min_pixels = 65536;
    uint64_t opt_fail_after_passes = 0;          // test hook: render_impl reports a failure once this many passes are enqueued (0 = never)
    uint32_t grid_max = 2048;

    std::vector<EventPair> events;
    size_t events_used = 0;
    uint64_t samples = 0;
    uint64_t launches[4] = {0, 0, 0, 0};
};

namespace {

uint32_t grid_for(const rt_scene* s, uint64_t n) {
    uint64_t g = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (g < 1) g = 1;
    if (g > s->grid_max) g = s->grid_max;
    return (uint32_t)g;
}

// Per primitive record of the occlusion BVH: a coplanar neighbour for k_light's RT_LR_SKIP_SIB (rt_kernels.h).  A triangle whose three
// vertices share one coordinate exactly has a flat box; among the other such triangles in the same plane (same axis, same coordinate) the
// one whose box overlaps its box most is its neighbour: for a wall made of two triangles, the other half.  Word: bits 0..22 the
// neighbour's record (RT_W_NO_REC: none), bits 23..24 the flat axis.  A choice of which record may be skipped on proof, nothing more.
std::vector<uint32_t> coplanar_neighbours(const WideBVH& w) {
    const uint32_t n = w.n_prims();
    std::vector<uint32_t> out(n, RT_W_NO_REC);
    struct Flat { uint32_t rec, axis; float lo[2], hi[2]; };
    std::map<std::pair<uint32_t, uint32_t>, std::vector<Flat>> planes;   // (axis, bits of the coordinate; -0 counted as +0)
    for (uint32_t r = 0; r < n; ++r) {
        const float* p = &w.prims[12 * (size_t)r];
        uint32_t kind;
        memcpy(&kind, &p[3], 4);
        if (kind != 0u) continue;   // a sphere
        for (uint32_t a = 0; a < 3; ++a) {
            if (!(p[a] == p[4 + a] && p[a] == p[8 + a])) continue;
            Flat f;
            f.rec = r; f.axis = a;
            for (uint32_t k = 0; k < 2; ++k) {
                const uint32_t b = (a + 1 + k) % 3;
                f.lo[k] = std::min(p[b], std::min(p[4 + b], p[8 + b]));
                f.hi[k] = std::max(p[b], std::max(p[4 + b], p[8 + b]));
            }
            const float c = p[a] + 0.0f;
            uint32_t bits;
            memcpy(&bits, &c, 4);
            planes[std::make_pair(a, bits)].push_back(f);
            break;
        }
    }
    for (auto& kv : planes) {
        std::vector<Flat>& g = kv.second;
        for (size_t i = 0; i < g.size(); ++i) {
            // the neighbours by record order are the neighbours in space (records follow the tree): look at 16 either side
            double best = 0.0;
            uint32_t pick = RT_W_NO_REC;
            const size_t j0 = i > 16 ? i - 16 : 0, j1 = std::min(g.size(), i + 17);
            for (size_t j = j0; j < j1; ++j) {
                if (j == i) continue;
                const double ox = (double)std::min(g[i].hi[0], g[j].hi[0]) - (double)std::max(g[i].lo[0], g[j].lo[0]);
                const double oy = (double)std::min(g[i].hi[1], g[j].hi[1]) - (double)std::max(g[i].lo[1], g[j].lo[1]);
                if (ox > 0.0 && oy > 0.0 && ox * oy > best) { best = ox * oy; pick = g[j].rec; }
            }
            out[g[i].rec] = pick | (g[i].axis << 23);
        }
    }
    return out;
}

// Host half of Scene::new: the BVH and the light list (src/scene/mod.rs:91-99).  No GPU needed.
void prepare_host(rt_scene* s) {
    const HostScene& h = s->host;
    const uint32_t T = h.n_tri(), S = h.n_sph();
    // Scene::new's BVH::build (src/scene/mod.rs:99) and the occlusion tree: on the GPU when one has been chosen and the scene is large
    // enough to pay for the launches (same trees either way: rt_build.hip), on the host otherwise.
    const bool try_device = g_inited && g_device_build && T + S >= g_device_build_min && hipSetDevice(g_device) == hipSuccess;
    typedef std::chrono::steady_clock clk;
    s->build_note.clear();
    clk::time_point t0 = clk::now();
    std::string why;
    s->build_on_device[0] = try_device && device_build_bvh(h, &s->bvh, &s->build_dev_ms[0], &why) ? 1u : 0u;
    if (!s->build_on_device[0]) {
        if (try_device) { s->build_note = "binary tree: " + why; (void)hipGetLastError(); }
        build_bvh(h, &s->bvh);
    }
    clk::time_point t1 = clk::now();
    s->build_on_device[1] = try_device && device_build_wide_bvh(h, s->bvh, &s->wide, &s->build_dev_ms[1], &why) ? 1u : 0u;
    if (!s->build_on_device[1]) {
        if (try_device) { s->build_note += (s->build_note.empty() ? "" : "; ") + std::string("8-wide tree: ") + why; (void)hipGetLastError(); }
        build_wide_bvh(h, s->bvh, &s->wide);
    }
    s->build_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    s->build_ms[1] = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
    s->lights.clear();
    s->has_triangle_light = false;
    for (uint32_t i = 0; i < T; ++i)
        if (!scene_is_black(h.mats[h.tri_mat[i]].emittance)) { s->lights.push_back(i); s->has_triangle_light = true; }
    for (uint32_t i = 0; i < S; ++i)
        if (!scene_is_black(h.mats[h.sph_mat[i]].emittance)) s->lights.push_back(T + i);
    s->device_bytes = (uint64_t)s->bvh.n_inner() * 64 + (uint64_t)T * 96 + (uint64_t)S * 24 + (uint64_t)h.mats.size() * 32 + s->lights.size() * 4 +
                      (uint64_t)s->wide.n_nodes() * 80 + (uint64_t)s->wide.n_prims() * 48 + (uint64_t)s->wide.rec_of_obj.size() * 4;
}

// Own box of a primitive record of the occlusion BVH: the same floats as Bounded for Triangle / Sphere (objects.rs:293-316) and as
// lo3 / hi3 in the kernels.
void record_box(const float* q, float* mn, float* mx) {
    if (float_to_bits(q[3]) == 0u) {
        for (int a = 0; a < 3; ++a) {
            const float l = q[a] < q[4 + a] ? q[a] : q[4 + a], g = q[a] > q[4 + a] ? q[a] : q[4 + a];
            mn[a] = l < q[8 + a] ? l : q[8 + a];
            mx[a] = g > q[8 + a] ? g : q[8 + a];
        }
    } else {
        for (int a = 0; a < 3; ++a) { mn[a] = q[a] - q[4]; mx[a] = q[a] + q[4]; }
    }
}

// Ill-conditioned triangles (rt_kernels.h prune_limit): the sine of the angle between the edges below 2^-5, or not a number.  The distance
// their test returns can be far from the true one for any ray, and the reference's unpruned search accepts it; the kernels therefore do not
// skip subtrees that hold one.  frag_obj[object]; *inner[node] bit 0 / 1 = left / right subtree holds one; *wide[node] = the 8-wide node does.
void mark_fragile(const rt_scene* s, std::vector<uint8_t>* frag_obj, std::vector<uint8_t>* inner, std::vector<uint8_t>* wide) {
    const HostScene& h = s->host;
    const uint32_t T = h.n_tri(), n_obj = T + h.n_sph();
    frag_obj->assign(n_obj, 0);
    for (uint32_t i = 0; i < T; ++i) {
        const float* p = &h.tri_pos[9 * (size_t)i];
        double e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = (double)p[3 + k] - (double)p[k]; e2[k] = (double)p[6 + k] - (double)p[k]; }
        const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
        const double c2 = cx * cx + cy * cy + cz * cz, l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
        if (!(c2 >= (1.0 / 1024.0) * l1 * l2) || !(l1 * l2 > 0.0) || !(l1 * l2 < 1e300)) (*frag_obj)[i] = 1;   // sin^2 < 2^-10, zero area, non-finite
    }
    const FlatBVH& b = s->bvh;
    const uint32_t ni = b.n_inner();
    inner->assign(ni, 0);
    auto ref_fragile = [&](uint32_t ref) -> bool { return (ref & LEAF_BIT) ? (*frag_obj)[ref & OBJECT_MASK] != 0 : (*inner)[ref] != 0; };
    for (uint32_t i = ni; i-- > 0;) {   // pre-order: children have larger indices
        const uint32_t* w = &b.words[16 * (size_t)i];
        (*inner)[i] = (uint8_t)((ref_fragile(w[12]) ? 1 : 0) | (ref_fragile(w[13]) ? 2 : 0));
    }
    const WideBVH& wd = s->wide;
    wide->assign(wd.ok ? wd.n_nodes() : 0, 0);
    for (uint32_t i = (uint32_t)wide->size(); i-- > 0;) {   // breadth-first: children have larger indices
        const uint32_t* q = &wd.nodes[20 * (size_t)i];
        const uint32_t n_inner = (q[3] >> 24) & 0xFu, n_child = q[3] >> 28;
        uint8_t f = 0;
        for (uint32_t k = 0; k < n_inner; ++k) f |= (*wide)[q[4] + k];
        for (uint32_t k = 0; k < n_child - n_inner; ++k) {
            const float* r = &wd.prims[12 * (size_t)(q[5] + k)];
            if (float_to_bits(r[3]) == 0u) f |= (*frag_obj)[float_to_bits(r[7])];
        }
        (*wide)[i] = f;
    }
}

// What the kernels assume about the two trees, checked on the host before anything is uploaded: every index a kernel follows
// stays inside its array, so a malformed tree is an error here and not a memory fault there.
bool trees_are_well_formed(const rt_scene* s, std::string* why) {
    const uint32_t n_obj = s->host.n_tri() + s->host.n_sph();
    const FlatBVH& b = s->bvh;
    const uint32_t ni = b.n_inner();
    if (b.rank.size() != n_obj) { *why = "rank table size"; return false; }
    if (n_obj && !(b.root & LEAF_BIT) && b.root >= ni) { *why = "root index"; return false; }
    for (uint32_t i = 0; i < ni; ++i)
        for (int c = 12; c <= 13; ++c) {
            const uint32_t ref = b.words[16 * (size_t)i + c];
            if ((ref & LEAF_BIT) ? (ref & OBJECT_MASK) >= n_obj : (ref >= ni || ref <= i)) { *why = "binary node child reference"; return false; }
        }
    const WideBVH& w = s->wide;
    if (!w.ok) return true;
    const uint32_t nn = w.n_nodes(), np = w.n_prims();
    if (!nn || w.rec_of_obj.size() != n_obj) { *why = "8-wide tree tables"; return false; }
    for (uint32_t i = 0; i < nn; ++i) {
        const uint32_t* q = &w.nodes[20 * (size_t)i];
        const uint32_t n_inner = (q[3] >> 24) & 0xFu, n_child = q[3] >> 28;
        if (n_child > 8 || n_inner > n_child) { *why = "8-wide node child counts"; return false; }
        if (n_inner && ((uint64_t)q[4] + n_inner > nn || q[4] <= i)) { *why = "8-wide node child index"; return false; }
        if (n_child > n_inner && (uint64_t)q[5] + (n_child - n_inner) > np) { *why = "8-wide node record index"; return false; }
    }
    for (uint32_t r = 0; r < np; ++r) {
        const uint32_t kind = float_to_bits(w.prims[12 * (size_t)r + 3]);
        const uint32_t obj = float_to_bits(w.prims[12 * (size_t)r + (kind == 0u ? 7 : 5)]);
        if (kind > 1u || obj >= n_obj || w.rec_of_obj[obj] != r) { *why = "8-wide primitive record"; return false; }
    }
    for (uint32_t o = 0; o < n_obj; ++o)
        if (w.rec_of_obj[o] != 0xFFFFFFFFu && w.rec_of_obj[o] >= np) { *why = "record-of-object table"; return false; }
    return true;
}

// Device half: flatten to the kernel layout and upload, on first use.
int upload_scene(rt_scene* s) {
    BIND(s->device);
    if (s->uploaded) return RT_OK;
    {
        std::string why;
        if (!trees_are_well_formed(s, &why)) return fail(RT_ERR_INVALID, "scene trees failed the pre-upload check: " + why);
    }
    const HostScene& h = s->host;
    const uint32_t T = h.n_tri(), S = h.n_sph(), M = (uint32_t)h.mats.size();
    const FlatBVH& b = s->bvh;
    if (M > RT_INFO_MAT) return fail(RT_ERR_LIMIT, "more than 2^30 materials");

    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev_id));
    s->grid_max = (uint32_t)prop.multiProcessorCount * 8u;
    if (s->grid_max == 0) s->grid_max = 2048;

    // deepest traversal stack: one entry per level of the binary tree, two per level of the 8-wide occlusion tree
    const uint32_t stack_need = b.depth > 2 * s->wide.depth ? b.depth : 2 * s->wide.depth;
    const uint32_t ovf_rows = stack_need > RT_STACK_LDS ? stack_need - RT_STACK_LDS : 0;
    if (ovf_rows > 224) return fail(RT_ERR_LIMIT, "BVH deeper than 256 levels");
    s->ovf_rows = ovf_rows;

    std::vector<float4> tris(3 * (size_t)T), nrm(3 * (size_t)T), sph(S ? S : 1), mats(2 * (size_t)(M ? M : 1));
    std::vector<uint2> aux(S ? S : 1);
    for (uint32_t i = 0; i < T; ++i) {
        const float* p = &h.tri_pos[9 * (size_t)i];
        const float* n = &h.tri_nrm[9 * (size_t)i];
        const vec3 p1 = mk(p[0], p[1], p[2]), p2 = mk(p[3], p[4], p[5]), p3 = mk(p[6], p[7], p[8]);
        const rt_material& m = h.mats[h.tri_mat[i]];
        const bool emissive = !scene_is_black(m.emittance);
        const bool flat = (fminf(fminf(p1.x, p2.x), p3.x) == fmaxf(fmaxf(p1.x, p2.x), p3.x)) || (fminf(fminf(p1.y, p2.y), p3.y) == fmaxf(fmaxf(p1.y, p2.y), p3.y)) ||
                          (fminf(fminf(p1.z, p2.z), p3.z) == fmaxf(fmaxf(p1.z, p2.z), p3.z));
        const uint32_t info = h.tri_mat[i] | (emissive ? 0u : RT_INFO_BLOCKER) | (flat ? RT_INFO_FLAT : 0u);
        tris[3 * (size_t)i] = make_float4(p1.x, p1.y, p1.z, bits_to_float(b.rank[i]));
        tris[3 * (size_t)i + 1] = make_float4(p2.x, p2.y, p2.z, bits_to_float(info));
        const double e1l = sqrt(((double)p2.x - p1.x) * ((double)p2.x - p1.x) + ((double)p2.y - p1.y) * ((double)p2.y - p1.y) + ((double)p2.z - p1.z) * ((double)p2.z - p1.z));
        const double e2l = sqrt(((double)p3.x - p1.x) * ((double)p3.x - p1.x) + ((double)p3.y - p1.y) * ((double)p3.y - p1.y) + ((double)p3.z - p1.z) * ((double)p3.z - p1.z));
        tris[3 * (size_t)i + 2] = make_float4(p3.x, p3.y, p3.z, (float)(e1l * e2l * (1.0 / 1024.0)));   // grazes_triangle's threshold
        for (int k = 0; k < 3; ++k) nrm[3 * (size_t)i + k] = make_float4(n[3 * k], n[3 * k + 1], n[3 * k + 2], 0.0f);
    }
    for (uint32_t i = 0; i < S; ++i) {
        const float* c = &h.sph[4 * (size_t)i];
        const rt_material& m = h.mats[h.sph_mat[i]];
        const bool emissive = !scene_is_black(m.emittance);
        sph[i] = make_float4(c[0], c[1], c[2], c[3]);
        aux[i] = make_uint2(b.rank[T + i], h.sph_mat[i] | (emissive ? 0u : RT_INFO_BLOCKER));
    }
    for (uint32_t i = 0; i < M; ++i) {
        const rt_material& m = h.mats[i];
        mats[2 * (size_t)i] = make_float4(bits_to_float(m.bsdf), m.reflectance[0], m.reflectance[1], m.reflectance[2]);
        mats[2 * (size_t)i + 1] = make_float4(m.emittance[0], m.emittance[1], m.emittance[2], 0.0f);
    }
    const size_t n_inner = b.n_inner();
    std::vector<uint8_t> frag_obj, frag_inner, frag_wide;
    mark_fragile(s, &frag_obj, &frag_inner, &frag_wide);
    int rc;
    if ((rc = s->d_nodes.ensure(4 * (n_inner ? n_inner : 1)))) return rc;
    if ((rc = s->d_tris.ensure(tris.size() ? tris.size() : 1))) return rc;
    if ((rc = s->d_tri_nrm.ensure(nrm.size() ? nrm.size() : 1))) return rc;
    if ((rc = s->d_sph.ensure(sph.size()))) return rc;
    if ((rc = s->d_sph_aux.ensure(aux.size()))) return rc;
    if ((rc = s->d_mats.ensure(mats.size()))) return rc;
    if ((rc = s->d_lights.ensure(s->lights.size() ? s->lights.size() : 1))) return rc;
    if (s->wide.ok) {
        if ((rc = s->d_wnodes.ensure(5 * (size_t)s->wide.n_nodes()))) return rc;
        if ((rc = s->d_wprims.ensure(3 * (size_t)s->wide.n_prims()))) return rc;
        {
            std::vector<uint32_t> wn(s->wide.nodes);   // device copy: word 6 = the node holds an ill-conditioned triangle
            for (size_t i = 0; i < frag_wide.size(); ++i) wn[20 * i + 6] = frag_wide[i];
            HIP_TRY(hipMemcpy(s->d_wnodes.p, wn.data(), wn.size() * 4, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMemcpy(s->d_wprims.p, s->wide.prims.data(), s->wide.prims.size() * 4, hipMemcpyHostToDevice));
        if ((rc = s->d_wrec.ensure(s->wide.rec_of_obj.size()))) return rc;
        HIP_TRY(hipMemcpy(s->d_wrec.p, s->wide.rec_of_obj.data(), s->wide.rec_of_obj.size() * 4, hipMemcpyHostToDevice));
        const std::vector<uint32_t> sib = coplanar_neighbours(s->wide);
        if ((rc = s->d_wsib.ensure(sib.size() ? sib.size() : 1))) return rc;
        if (!sib.empty()) HIP_TRY(hipMemcpy(s->d_wsib.p, sib.data(), sib.size() * 4, hipMemcpyHostToDevice));
    }
    if (ovf_rows && (rc = s->d_ovf.ensure((size_t)ovf_rows * s->grid_max * RT_BLOCK * RT_MAX_LANES))) return rc;   // one set of rows per render lane
    if (n_inner) {
        // device copy of the nodes: leaves of emissive objects carry NONBLOCK_BIT, word 14 the FRAGILE bits of the two subtrees
        std::vector<uint32_t> words(b.words);
        for (size_t i = 0; i < n_inner; ++i) words[16 * i + 14] = frag_inner[i];
        for (size_t i = 0; i < n_inner; ++i)
            for (int c = 12; c <= 13; ++c) {
                const uint32_t ref = words[16 * i + c];
                if (!(ref & LEAF_BIT)) continue;
                const uint32_t obj = ref & OBJECT_MASK;
                const rt_material& m = obj < T ? h.mats[h.tri_mat[obj]] : h.mats[h.sph_mat[obj - T]];
                if (!scene_is_black(m.emittance)) words[16 * i + c] = ref | NONBLOCK_BIT;
            }
        HIP_TRY(hipMemcpy(s->d_nodes.p, words.data(), n_inner * 64, hipMemcpyHostToDevice));
    }
    if (T) {
        HIP_TRY(hipMemcpy(s->d_tris.p, tris.data(), tris.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_tri_nrm.p, nrm.data(), nrm.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(s->d_sph.p, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_sph_aux.p, aux.data(), aux.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_mats.p, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (!s->lights.empty())
        HIP_TRY(hipMemcpy(s->d_lights.p, s->lights.data(), s->lights.size() * 4, hipMemcpyHostToDevice));
    DevScene& d = s->dev;
    d.nodes = s->d_nodes.p; d.tris = s->d_tris.p; d.tri_nrm = s->d_tri_nrm.p; d.sph = s->d_sph.p; d.sph_aux = s->d_sph_aux.p;
    d.mats = s->d_mats.p; d.lights = s->d_lights.p;
    d.n_tri = T; d.n_sph = S; d.n_lights = (uint32_t)s->lights.size(); d.root = b.root;
    if ((b.root & LEAF_BIT) && T + S == 1) {
        const rt_material& m = T ? h.mats[h.tri_mat[0]] : h.mats[h.sph_mat[0]];
        if (!scene_is_black(m.emittance)) d.root |= NONBLOCK_BIT;
    }
    d.ovf = s->d_ovf.p; d.ovf_stride = s->grid_max * RT_BLOCK;
    d.inner_min = s->opt_inner_min; d.self_test = s->opt_self_test;
    d.wnodes = s->d_wnodes.p; d.wprims = s->d_wprims.p; d.wrec = s->d_wrec.p; d.wsib = s->d_wsib.p;
    {
        // Occluder cache of k_light: per light, up to RT_HOT non-emissive objects whose box overlaps the light's box (a light recessed
        // into or poking through a wall has most of its samples blocked by that wall) and that block enough rays to earn their test.
        // A choice of test order only.
        for (int li = 0; li < RT_HOT_LIGHTS; ++li)
            for (int k = 0; k < RT_HOT; ++k) { memset(&s->hot_table.rec[li][k], 0, sizeof(HotRec)); s->hot_table.rec[li][k].kind = 2u; s->hot_table.rec[li][k].rec = RT_W_NO_REC; }
        if (s->wide.ok && s->opt_hot) {
            for (size_t li = 0; li < s->lights.size() && li < RT_HOT_LIGHTS; ++li) {
                if (s->lights[li] < T) continue;   // triangle lights are rejected before any sampling
                const float* c = &h.sph[4 * (size_t)(s->lights[li] - T)];
                // candidates: the 8 largest boxes that overlap the light's box
                const int NC = 8;
                float cand_area[NC];
                uint32_t cand_rec[NC];
                for (int k = 0; k < NC; ++k) { cand_area[k] = -1.0f; cand_rec[k] = RT_W_NO_REC; }
                for (uint32_t r = 0; r < s->wide.n_prims(); ++r) {
                    float mn[3], mx[3];
                    record_box(&s->wide.prims[12 * (size_t)r], mn, mx);
                    bool overlap = true;
                    for (int a = 0; a < 3; ++a) overlap = overlap && mn[a] <= c[a] + c[3] && mx[a] >= c[a] - c[3];
                    if (!overlap) continue;
                    const float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
                    float area = sx * sy + sx * sz + sy * sz;
                    uint32_t rec = r;
                    for (int k = 0; k < NC; ++k)   // insertion into the descending top list
                        if (area > cand_area[k]) {
                            const float ta = cand_area[k]; const uint32_t tr = cand_rec[k];
                            cand_area[k] = area; cand_rec[k] = rec;
                            area = ta; rec = tr;
                        }
                }
                // which of them earn their test: estimate, on a fixed pseudo-random set of (surface point, light sample) pairs, the share of
                // shadow rays each candidate blocks that the ones chosen before it do not; a test in k_light costs about what tracing 2-3 % of
                // the rays costs (measured on config 4), so a record below 2 % is left to the traversal.  Only the ORDER of exact tests depends on this.
                uint32_t best_rec[RT_HOT];
                for (int k = 0; k < RT_HOT; ++k) best_rec[k] = RT_W_NO_REC;
                {
                    const int NS = 4096;
                    std::vector<uint8_t> hit((size_t)NS * NC, 0);
                    uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)li;
                    auto rnd = [&st]() { st = st * RT_PCG_MUL + RT_PCG_INC; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
                    const uint32_t np = s->wide.n_prims();
                    // surface points in proportion to area, as path vertices fall
                    std::vector<double> cum(np);
                    double total = 0.0;
                    for (uint32_t r = 0; r < np; ++r) {
                        const float* q = &s->wide.prims[12 * (size_t)r];
                        double ar;
                        if (float_to_bits(q[3]) == 0u) {
                            double e1[3], e2[3];
                            for (int a = 0; a < 3; ++a) { e1[a] = (double)q[4 + a] - q[a]; e2[a] = (double)q[8 + a] - q[a]; }
                            const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                            ar = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
                        } else {
                            ar = 12.566370614359172 * (double)q[4] * q[4];
                        }
                        total += (ar == ar && ar < 1e300) ? ar : 0.0;
                        cum[r] = total;
                    }
                    for (int i = 0; i < NS && np && total > 0.0; ++i) {
                        const double pick_at = rnd() * total;
                        uint32_t lo_r = 0, hi_r = np - 1;
                        while (lo_r < hi_r) { const uint32_t md = (lo_r + hi_r) / 2; if (cum[md] < pick_at) lo_r = md + 1; else hi_r = md; }
                        const uint32_t r0 = lo_r;
                        const float* q = &s->wide.prims[12 * (size_t)r0];
                        double o[3];
                        if (float_to_bits(q[3]) == 0u) {
                            double u = rnd(), v = rnd();
                            if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
                            for (int a = 0; a < 3; ++a) o[a] = q[a] + u * ((double)q[4 + a] - q[a]) + v * ((double)q[8 + a] - q[a]);
                        } else {
                            const double z = 2.0 * rnd() - 1.0, ph = 6.283185307179586 * rnd(), rr = sqrt(1.0 - z * z);
                            o[0] = q[0] + q[4] * rr * cos(ph); o[1] = q[1] + q[4] * rr * sin(ph); o[2] = q[2] + q[4] * z;
                        }
                        const double z = 2.0 * rnd() - 1.0, ph = 6.283185307179586 * rnd(), rr = sqrt(1.0 - z * z);
                        const double d[3] = {c[0] + c[3] * rr * cos(ph) - o[0], c[1] + c[3] * rr * sin(ph) - o[1], c[2] + c[3] * z - o[2]};
                        for (int k = 0; k < NC; ++k) {
                            if (cand_rec[k] == RT_W_NO_REC || cand_rec[k] == r0) continue;
                            const float* t = &s->wide.prims[12 * (size_t)cand_rec[k]];
                            bool blocks = false;
                            if (float_to_bits(t[3]) == 0u) {   // Moeller-Trumbore, parameter along o -> light sample in (0, 1)
                                double e1[3], e2[3], sv[3];
                                for (int a = 0; a < 3; ++a) { e1[a] = (double)t[4 + a] - t[a]; e2[a] = (double)t[8 + a] - t[a]; sv[a] = o[a] - t[a]; }
                                const double s1[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                                const double det = s1[0] * e1[0] + s1[1] * e1[1] + s1[2] * e1[2];
                                if (det != 0.0) {
                                    const double s2[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
                                    const double tt = (s2[0] * e2[0] + s2[1] * e2[1] + s2[2] * e2[2]) / det, b1 = (s1[0] * sv[0] + s1[1] * sv[1] + s1[2] * sv[2]) / det,
                                                 b2 = (s2[0] * d[0] + s2[1] * d[1] + s2[2] * d[2]) / det;
                                    blocks = b1 >= 0.0 && b2 >= 0.0 && b1 + b2 <= 1.0 && tt > 1e-9 && tt < 1.0;
                                }
                            } else {
                                const double l[3] = {t[0] - o[0], t[1] - o[1], t[2] - o[2]};
                                const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], adj = (l[0] * d[0] + l[1] * d[1] + l[2] * d[2]) / dd;
                                const double d2 = (l[0] * l[0] + l[1] * l[1] + l[2] * l[2]) - adj * adj * dd;
                                if (d2 <= (double)t[4] * t[4]) {
                                    const double thc = sqrt(((double)t[4] * t[4] - d2) / dd);
                                    blocks = (adj - thc > 1e-9 && adj - thc < 1.0) || (adj + thc > 1e-9 && adj + thc < 1.0);
                                }
                            }
                            hit[(size_t)i * NC + k] = blocks ? 1 : 0;
                        }
                    }
                    std::vector<uint8_t> settled(NS, 0);
                    for (int pick = 0; pick < RT_HOT; ++pick) {
                        int best = -1, best_n = 0;
                        for (int k = 0; k < NC; ++k) {
                            if (cand_rec[k] == RT_W_NO_REC) continue;
                            int nblk = 0;
                            for (int i = 0; i < NS; ++i) nblk += (!settled[i] && hit[(size_t)i * NC + k]) ? 1 : 0;
                            if (nblk > best_n) { best = k; best_n = nblk; }
                        }
                        if (best < 0 || best_n * 50 < NS) break;   // under 2 % of the rays
                        best_rec[pick] = cand_rec[best];
                        for (int i = 0; i < NS; ++i) settled[i] |= hit[(size_t)i * NC + best];
                        cand_rec[best] = RT_W_NO_REC;
                    }
                }
                for (int k = 0; k < RT_HOT; ++k) {
                    if (best_rec[k] == RT_W_NO_REC) continue;
                    const float* q = &s->wide.prims[12 * (size_t)best_rec[k]];
                    HotRec& hr = s->hot_table.rec[li][k];
                    hr.rec = best_rec[k];
                    hr.kind = float_to_bits(q[3]) == 0u ? 0u : 1u;
                    for (int a = 0; a < 3; ++a) hr.a[a] = q[a];
                    if (hr.kind == 0u) {
                        for (int a = 0; a < 3; ++a) { hr.e1[a] = q[4 + a] - q[a]; hr.e2[a] = q[8 + a] - q[a]; }   // sub(p2, p1), sub(p3, p1)
                    } else {
                        hr.e1[0] = q[4];                                                                         // radius
                    }
                    record_box(q, hr.mn, hr.mx);
                }
            }
        }
    }
    d.n_wnodes = (s->wide.ok && s->opt_wide) ? s->wide.n_nodes() : 0u;
    d.graze_guard = s->opt_graze_guard;
    d.refill_min = s->opt_refill_min; d.phase_bias = s->opt_phase_bias; d.extend_refill_min = s->opt_extend_refill; d.extend_inner_min = s->opt_extend_inner;
    if ((rc = s->w_stats.ensure(ST_COUNT))) return rc;
    HIP_TRY(hipMemset(s->w_stats.p, 0, ST_COUNT * sizeof(unsigned long long)));
    s->uploaded = true;
    return RT_OK;
}

int validate_params(const rt_render_params* P) {
    if (!P) return fail(RT_ERR_INVALID, "params is NULL");
    if (P->width == 0 || P->height == 0) return fail(RT_ERR_INVALID, "width and height must be > 0");
    if ((uint64_t)P->width * P->height >= (1ull << 31)) return fail(RT_ERR_LIMIT, "image larger than 2^31 pixels");
    if (P->tile_w == 0 || P->tile_h == 0 || P->shard_count == 0 || P->shard_index >= P->shard_count)
        return fail(RT_ERR_INVALID, "bad tile/shard description");
    if (P->mode > RT_MODE_FULL || P->direct_estimator > RT_DIRECT_HEMISPHERE) return fail(RT_ERR_INVALID, "bad mode or estimator");
    return RT_OK;
}

struct Shard {
    uint32_t tiles_x, n_tiles, n_own_tiles;
    uint64_t n_own_pix;
};
// Owned pixels inside the image (edge tiles are clipped).
uint64_t valid_pixels(const rt_render_params& P, const struct Shard& sh);
Shard shard_of(const rt_render_params& P) {
    Shard s;
    s.tiles_x = (P.width + P.tile_w - 1) / P.tile_w;
    const uint32_t tiles_y = (P.height + P.tile_h - 1) / P.tile_h;
    s.n_tiles = s.tiles_x * tiles_y;
    s.n_own_tiles = s.n_tiles > P.shard_index ? (s.n_tiles - P.shard_index + P.shard_count - 1) / P.shard_count : 0;
    s.n_own_pix = (uint64_t)s.n_own_tiles * P.tile_w * P.tile_h;
    return s;
}

uint64_t valid_pixels(const rt_render_params& P, const Shard& sh) {
    uint64_t valid = 0;
    for (uint32_t k = 0; k < sh.n_own_tiles; ++k) {
        const uint32_t t = P.shard_index + k * P.shard_count;
        const uint32_t x0 = (t % sh.tiles_x) * P.tile_w, y0 = (t / sh.tiles_x) * P.tile_h;
        const uint32_t wv = x0 + P.tile_w <= P.width ? P.tile_w : (P.width > x0 ? P.width - x0 : 0);
        const uint32_t hv = y0 + P.tile_h <= P.height ? P.tile_h : (P.height > y0 ? P.height - y0 : 0);
        valid += (uint64_t)wv * hv;
    }
    return valid;
}

void begin_timing(rt_scene* s, hipStream_t stream, int cls, size_t* idx) {
    *idx = (size_t)-1;
    s->launches[cls]++;
    if (!s->opt_timing) return;
    if (s->events_used == s->events.size()) {
        EventPair ep;
        if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return;
        s->events.push_back(ep);
    }
    *idx = s->events_used++;
    s->events[*idx].cls = cls;
    (void)hipEventRecord(s->events[*idx].a, stream);
}
void end_timing(rt_scene* s, hipStream_t stream, size_t idx) {
    if (idx == (size_t)-1) return;
    (void)hipEventRecord(s->events[idx].b, stream);
}

void reset_counters(rt_scene* s, hipStream_t stream) {
    (void)hipMemsetAsync(s->w_stats.p, 0, ST_COUNT * sizeof(unsigned long long), stream);
    s->stats_src = s->w_stats.p;
    s->events_used = 0;
    s->samples = 0;
    s->launches[0] = s->launches[1] = s->launches[2] = s->launches[3] = 0;
}

int ensure_jump(rt_scene* s, uint32_t nlL, hipStream_t stream) {
    if (s->jump_nlL == nlL) return RT_OK;
    std::vector<ulonglong2> j((size_t)nlL + 1);
    // draw 2*i of the vertex: state_i = A_i * state + C_i for the LCG of fastrand 1.4.0
    unsigned long long A = 1, C = 0;
    for (uint32_t i = 0; i <= nlL; ++i) {
        j[i].x = A; j[i].y = C;
        for (int k = 0; k < 2; ++k) { A = A * RT_PCG_MUL; C = C * RT_PCG_MUL + RT_PCG_INC; }
    }
    int rc;
    if ((rc = s->w_jump.ensure(j.size()))) return rc;
    HIP_TRY(hipStreamSynchronize(stream));   // the previous table may still be in use
    HIP_TRY(hipMemcpy(s->w_jump.p, j.data(), j.size() * sizeof(ulonglong2), hipMemcpyHostToDevice));
    s->jump_nlL = nlL;
    return RT_OK;
}

// host_rgba (optional): the caller's host frame; with a progress word it receives the device frame after every pass, so that whoever polls
// the buffer while the render runs (the canvas, src/canvas.rs:179) sees it fill and refine as the reference's does (raytracer.rs:203-214).
int render_impl(rt_scene* s, const rt_render_params* params, uint8_t* rgba_dev, hipStream_t stream, volatile uint32_t* progress,
                const volatile uint32_t* interrupt, uint8_t* host_rgba = nullptr) {
    int rc;
    if ((rc = validate_params(params))) return rc;
    if (!rgba_dev) return fail(RT_ERR_INVALID, "rgba is NULL");
    if ((rc = upload_scene(s))) return rc;
    const rt_render_params& P = *params;
    const Shard sh = shard_of(P);
    reset_counters(s, stream);
    if (progress) *progress = 0;
    if (sh.n_own_pix == 0) { if (progress) *progress = 100; return RT_OK; }
    if (sh.n_own_pix >= (1ull << 31)) return fail(RT_ERR_LIMIT, "shard larger than 2^31 pixels");

    if (P.mode == RT_MODE_DEBUG) {
        size_t ev;
        begin_timing(s, stream, 0, &ev);
        const uint32_t g = grid_for(s, sh.n_own_pix);
        if (s->opt_stats)
            hipLaunchKernelGGL(k_primary<true>, dim3(g), dim3(RT_BLOCK), 0, stream, s->dev, P, sh.tiles_x, (uint32_t)sh.n_own_pix,
                               (uint32_t*)nullptr, (float*)nullptr, rgba_dev, s->w_stats.p);
        else
            hipLaunchKernelGGL(k_primary<false>, dim3(g), dim3(RT_BLOCK), 0, stream, s->dev, P, sh.tiles_x, (uint32_t)sh.n_own_pix,
                               (uint32_t*)nullptr, (float*)nullptr, rgba_dev, s->w_stats.p);
        end_timing(s, stream, ev);
        HIP_TRY(hipGetLastError());
        if (progress) { HIP_TRY(hipStreamSynchronize(stream)); *progress = 100; }
        return RT_OK;
    }

    if (P.spp == 0 || P.light_samples == 0) return fail(RT_ERR_INVALID, "spp and light_samples must be > 0");
    const bool hemi = P.direct_estimator == RT_DIRECT_HEMISPHERE;
    if (!hemi && s->has_triangle_light && P.bounces >= 1)
        return fail(RT_ERR_UNSUPPORTED, "triangle lights: Object::sample_l is unimplemented in the reference (objects.rs:98-100)");
    const uint32_t n_groups = hemi ? 1u : s->dev.n_lights;
    const uint64_t nlL64 = (uint64_t)n_groups * P.light_samples;
    if (nlL64 >= (1ull << 24)) return fail(RT_ERR_LIMIT, "lights * light_samples too large");
    const uint32_t nlL = (uint32_t)nlL64;

    const uint64_t cap_by_rays = (1ull << 31) / (nlL ? nlL : 1);   // light-sample slots of a pass are indexed with 32 bits
    const uint32_t levels = P.bounces + 1;
    const bool persist = !hemi && nlL > 0 && s->dev.n_wnodes != 0u && s->opt_persist;   // k_light + k_trace_p over the 8-wide occlusion BVH
    const bool in_wave = !persist && nlL > 0 && nlL <= 64 && (64 % nlL) == 0;

    // Render lanes: the owned pixels are cut into n_lanes contiguous ranges, each rendered on a stream of its own with pass buffers of its
    // own, so that the kernels of one lane that wait on memory overlap the VALU-bound ones of another.  Pixels are independent (RNG streams
    // are keyed by pixel and sample, accumulators are per pixel), so the image does not depend on the number of lanes.  For a caller that
    // polls progress or may interrupt, the host waits after every round (one pass per lane) to report and to look at the flag; with one lane
    // ("lanes" = 1) it also looks between bounce levels.
    uint32_t n_lanes = s->opt_lanes;
    if (n_lanes > RT_MAX_LANES) n_lanes = RT_MAX_LANES;
    while (n_lanes > 1 && sh.n_own_pix / n_lanes < s->opt_lane_min_pixels) --n_lanes;
    uint64_t lane_begin[RT_MAX_LANES + 1];
    for (uint32_t k = 0; k <= n_lanes; ++k) lane_begin[k] = k == n_lanes ? sh.n_own_pix : (sh.n_own_pix * k / n_lanes) & ~(uint64_t)63;
    uint64_t lane_pix = 0;   // the largest lane
    for (uint32_t k = 0; k < n_lanes; ++k) lane_pix = lane_begin[k + 1] - lane_begin[k] > lane_pix ? lane_begin[k + 1] - lane_begin[k] : lane_pix;
    if (n_lanes > 1) {
        if (!s->ev_fork) HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        for (uint32_t k = 0; k < n_lanes; ++k) {
            if (!s->ws[k].stream) HIP_TRY(hipStreamCreateWithFlags(&s->ws[k].stream, hipStreamNonBlocking));
            if (!s->ws[k].done) HIP_TRY(hipEventCreateWithFlags(&s->ws[k].done, hipEventDisableTiming));
            if (!s->ws[k].light) HIP_TRY(hipEventCreateWithFlags(&s->ws[k].light, hipEventDisableTiming));
        }
    }

    uint32_t npix_chunk = 0, spc = 0;
    // Workspace for `max_paths` camera samples in flight per lane (about 1.8 KB each at 32 light samples, depth 8).  Grow-only; when the
    // device cannot hold it the pass size is halved until it does (results do not depend on the pass size).
    // (a size that did not fit is not asked for again on the next frame: giving the workspace back and allocating it anew costs seconds)
    bool shrunk = false;
    for (uint64_t max_paths = (s->max_paths_fit && s->max_paths_fit < s->opt_max_paths) ? s->max_paths_fit : s->opt_max_paths;; max_paths /= 2, shrunk = true) {
        if (max_paths > cap_by_rays) max_paths = cap_by_rays;
        if (max_paths < 64) max_paths = 64;
        if (lane_pix <= max_paths) {
            npix_chunk = (uint32_t)lane_pix;
            uint64_t k = max_paths / lane_pix;
            spc = (uint32_t)(k < P.spp ? k : P.spp);
        } else {
            npix_chunk = (uint32_t)max_paths;
            spc = 1;
        }
        const uint64_t cap_paths = (uint64_t)npix_chunk * spc;
        const uint64_t cap_slots = ((cap_paths + 63) / 64) * 64 * nlL;
        const uint64_t per_level = cap_paths * (P.bounces ? P.bounces : 1);
        rc = s->w_accum.ensure(sh.n_own_pix);
        for (uint32_t k = 0; k < n_lanes && !rc; ++k) {
            Workspace& w = s->ws[k];
            rc = w.counts.ensure(3 * (levels + 1));
            for (int q = 0; q < 2 && !rc; ++q) {
                rc = w.ray_o[q].ensure(cap_paths);
                if (!rc) rc = w.ray_d[q].ensure(cap_paths);
            }
            if (!rc) rc = w.hit_id.ensure(cap_paths);
            if (!rc) rc = w.hit_t.ensure(cap_paths);
            if (!rc) rc = w.vP.ensure(cap_paths);
            if (!rc) rc = w.vN.ensure(cap_paths);
            if (!rc) rc = w.vS.ensure(cap_paths);
            for (int q = 0; q < 3 && !rc; ++q) rc = w.rng[q].ensure(cap_paths);
            if (!rc && persist) rc = w.lray.ensure(cap_slots);
            if (!rc && persist) rc = w.linv.ensure(cap_slots);
            if (!rc && persist) rc = w.blist.ensure(cap_slots);
            if (!rc && persist) rc = w.tmask.ensure(cap_slots / 64);
            if (!rc && persist) rc = w.tmask_t.ensure(cap_slots / 64);
            if (!rc && !persist && !in_wave) rc = w.contrib.ensure(cap_paths * (nlL ? nlL : 1));
            if (!rc) rc = w.direct.ensure(persist ? cap_paths * (n_groups ? n_groups : 1) : cap_paths);
            if (!rc) rc = w.stack.ensure(2 * per_level);
            if (!rc) rc = w.radiance.ensure(cap_paths);
        }
        if (!rc) { if (shrunk) s->max_paths_fit = max_paths; break; }
        (void)hipGetLastError();   // clear the out-of-memory error
        if (max_paths <= 65536) return rc;
        // the buffers that did fit at this size would compete with the smaller retry: give the whole workspace back first
        HIP_TRY(hipDeviceSynchronize());
        for (uint32_t k = 0; k < RT_MAX_LANES; ++k) s->ws[k].release();
        s->w_accum.release();
    }
    if ((rc = ensure_jump(s, nlL, stream))) return rc;

    PassCtx pcs[RT_MAX_LANES];
    DevScene devs[RT_MAX_LANES];
    hipStream_t streams[RT_MAX_LANES];
    for (uint32_t k = 0; k < n_lanes; ++k) {
        Workspace& w = s->ws[k];
        PassCtx& pc = pcs[k];
        memset(&pc, 0, sizeof pc);
        pc.P = P;
        pc.seed_mixed = mix64(P.seed);
        pc.tiles_x = sh.tiles_x; pc.n_tiles = sh.n_tiles;
        pc.n_groups = n_groups; pc.nlL = nlL; pc.vstride = (uint32_t)((uint64_t)npix_chunk * spc); pc.light_sum = s->opt_light_sum; pc.force_binary = s->opt_force_binary;
        pc.counts = w.counts.p; pc.vcounts = w.counts.p + (levels + 1); pc.bcounts = w.counts.p + 2 * (levels + 1); pc.vS = w.vS.p;
        for (int q = 0; q < 2; ++q) { pc.ray_o[q] = w.ray_o[q].p; pc.ray_d[q] = w.ray_d[q].p; }
        pc.hit_id = w.hit_id.p; pc.hit_t = w.hit_t.p;
        pc.vP = w.vP.p; pc.vN = w.vN.p; pc.qrng[0] = w.rng[0].p; pc.qrng[1] = w.rng[1].p; pc.vR = w.rng[2].p; pc.contrib = w.contrib.p; pc.direct = w.direct.p; pc.in_wave = in_wave ? 1u : 0u;
        pc.lray = w.lray.p; pc.linv = w.linv.p; pc.blist = w.blist.p; pc.tmask = w.tmask.p; pc.tmask_t = w.tmask_t.p; pc.shaft_steps = s->opt_shaft; pc.persist = persist ? 1u : 0u;
        pc.stack = w.stack.p; pc.stack_depth = P.bounces ? P.bounces : 1; pc.radiance = w.radiance.p; pc.accum = s->w_accum.p;
        pc.jump = s->w_jump.p; pc.stats = s->w_stats.p; pc.rgba = rgba_dev;
        pc.preview = (progress && host_rgba) ? 1u : 0u;
        devs[k] = s->dev;
        devs[k].ovf = s->d_ovf.p ? s->d_ovf.p + (size_t)k * s->ovf_rows * s->dev.ovf_stride : nullptr;   // concurrent kernels need rows of their own
        streams[k] = n_lanes > 1 ? w.stream : stream;
    }
    if (n_lanes > 1) {
        HIP_TRY(hipEventRecord(s->ev_fork, stream));
        for (uint32_t k = 0; k < n_lanes; ++k) HIP_TRY(hipStreamWaitEvent(streams[k], s->ev_fork, 0));
    }

    // One exit for every outcome: whatever happens while the passes are enqueued (an interrupt, a failed HIP call, the injected failure of the
    // tests), the lane streams are joined back into the caller's stream below before the call returns, so the caller never owns a frame that a
    // lane may still be writing.
    // passes of a lane: its pixels in chunks of npix_chunk, the samples of a chunk in chunks of spc; lanes take turns at enqueueing a pass
    uint64_t total_passes = 0, done = 0;
    const uint64_t fail_at = s->opt_fail_after_passes;   // test hook (rt_set_option "fail_after_passes"): report a failure once that many passes are enqueued
    uint64_t at_pix[RT_MAX_LANES];
    uint32_t at_s0[RT_MAX_LANES];
    for (uint32_t k = 0; k < n_lanes; ++k) {
        const uint64_t np = lane_begin[k + 1] - lane_begin[k];
        total_passes += ((np + npix_chunk - 1) / npix_chunk) * ((P.spp + spc - 1) / spc);
        at_pix[k] = lane_begin[k];
        at_s0[k] = 0;
    }
    const bool token = n_lanes > 1 && s->opt_light_token != 0;
    hipEvent_t last_light = nullptr;
    auto run_passes = [&]() -> int {
    for (bool any = true; any;) {
        any = false;
        if (fail_at && done >= fail_at) return fail(RT_ERR_HIP, "injected failure after " + std::to_string(done) + " passes (option fail_after_passes)");
        if (interrupt && *interrupt && done < total_passes) {   // between rounds: every lane has finished the same number of passes
            for (uint32_t j = 0; j < n_lanes; ++j) HIP_TRY(hipStreamSynchronize(streams[j]));
            return fail(RT_ERR_INTERRUPTED, "render interrupted");
        }
        bool active[RT_MAX_LANES] = {false, false, false, false};
        bool camera_once[RT_MAX_LANES] = {false, false, false, false};
        // a round: one pass per lane, enqueued level by level across the lanes (lane 0's level l, lane 1's level l, ..., then level l + 1), so that the
        // k_light launches of the lanes can be chained (light_token below)
        for (uint32_t k = 0; k < n_lanes; ++k) {
            if (at_pix[k] >= lane_begin[k + 1]) continue;
            any = true;
            PassCtx& pc = pcs[k];
            const hipStream_t st = streams[k];
            const uint64_t pix0 = at_pix[k];
            const uint32_t s0 = at_s0[k];
            const uint32_t npix = (uint32_t)((lane_begin[k + 1] - pix0) < npix_chunk ? (lane_begin[k + 1] - pix0) : npix_chunk);
            const uint32_t spc_now = (P.spp - s0) < spc ? (P.spp - s0) : spc;
            at_s0[k] += spc_now;
            if (at_s0[k] >= P.spp) { at_s0[k] = 0; at_pix[k] += npix; }
            pc.pix0 = (uint32_t)pix0; pc.npix = npix; pc.s0 = s0; pc.spc = spc_now; pc.npaths = npix * spc_now;
            HIP_TRY(hipMemsetAsync(pc.counts, 0, 3 * (levels + 1) * sizeof(uint32_t), st));
            size_t ev;
            active[k] = true;
            // the camera rays: one traversal per pixel serves the pass's samples of that pixel (k_camera), unless they are to go through the
            // persistent k_extend_p ("extend_persist" 2) or the scene is empty
            camera_once[k] = s->opt_camera_once && s->opt_extend_persist <= 1;
            if (camera_once[k]) {
                begin_timing(s, st, 0, &ev);
                if (s->opt_stats) hipLaunchKernelGGL(k_camera<true>, dim3(grid_for(s, npix)), dim3(RT_BLOCK), 0, st, devs[k], pc);
                else hipLaunchKernelGGL(k_camera<false>, dim3(grid_for(s, npix)), dim3(RT_BLOCK), 0, st, devs[k], pc);
                end_timing(s, st, ev);
            } else {
                begin_timing(s, st, 2, &ev);
                hipLaunchKernelGGL(k_generate, dim3(grid_for(s, pc.npaths)), dim3(RT_BLOCK), 0, st, pc);
                end_timing(s, st, ev);
            }
        }
        for (uint32_t level = 0; level < levels; ++level) {
          for (uint32_t k = 0; k < n_lanes; ++k) {
            if (!active[k]) continue;
            PassCtx& pc = pcs[k];
            const DevScene& dv = devs[k];
            const hipStream_t st = streams[k];
            size_t ev;
            {
                if (interrupt && level && n_lanes == 1) {
                    // interrupt_render() (raytracer.rs:496-498) is honoured between bounce levels: the host waits for the level it
                    // has enqueued, so at most one level of one pass runs on after the flag is raised
                    HIP_TRY(hipStreamSynchronize(st));
                    if (*interrupt) return fail(RT_ERR_INTERRUPTED, "render interrupted");
                }
                const uint32_t g = grid_for(s, pc.npaths);
                if (level == 0 && camera_once[k]) {
                    ev = (size_t)-1;           // traced by k_camera
                } else if (begin_timing(s, st, 0, &ev), s->opt_extend_persist && (level > 0 || s->opt_extend_persist > 1)) {
                    // after the first bounce: persistent waves that refill their idle lanes (k_extend_p); camera rays are coherent and stay with k_extend
                    uint32_t gp = s->grid_max / 8u * s->opt_extend_grid;
                    if (gp > g) gp = g;
                    if (s->opt_stats) hipLaunchKernelGGL(k_extend_p<true>, dim3(gp), dim3(RT_BLOCK), 0, st, dv, pc, level);
                    else hipLaunchKernelGGL(k_extend_p<false>, dim3(gp), dim3(RT_BLOCK), 0, st, dv, pc, level);
                } else if (s->opt_stats) hipLaunchKernelGGL(k_extend<true>, dim3(g), dim3(RT_BLOCK), 0, st, dv, pc, level);
                else hipLaunchKernelGGL(k_extend<false>, dim3(g), dim3(RT_BLOCK), 0, st, dv, pc, level);
                end_timing(s, st, ev);
                begin_timing(s, st, 2, &ev);
                hipLaunchKernelGGL(k_shade_a, dim3(g), dim3(RT_BLOCK), 0, st, dv, pc, level);
                end_timing(s, st, ev);
                if (level == P.bounces) continue;
                if (nlL) {
                    const uint32_t gs = grid_for(s, (uint64_t)pc.npaths * nlL);
                    begin_timing(s, st, hemi ? 0 : (persist ? 3 : 1), &ev);
                    if (persist) {
                        // "light_token" (off: measured slower, 134.1 vs 127.2 ms per 32-spp frame with three lanes, 140.2 vs 131.8 with two): chain the
                        // k_light launches of the lanes through events, lane after lane and level after level, so that no two of them share the GPU
                        if (token && last_light) HIP_TRY(hipStreamWaitEvent(st, last_light, 0));
                        // k_light: one workgroup per 256 vertices, however many that makes ("light_grid" 0; n: at most n per CU).  Its vertices differ
                        // several-fold in cost (beam walk or not, which arm of the own-object test), and with a grid of resident size and a fixed share
                        // each, the workgroups finished far apart: 73.6 vs 66.5 ms per 32-spp frame, one lane.  The hardware's dispatcher does the balancing.
                        uint32_t gl = (uint32_t)(((uint64_t)pc.npaths + RT_BLOCK - 1) / RT_BLOCK);
                        if (s->opt_light_grid && gl > s->grid_max / 8u * s->opt_light_grid) gl = s->grid_max / 8u * s->opt_light_grid;
                        if (gl < 1u) gl = 1u;
                        if (s->opt_stats) hipLaunchKernelGGL((k_light<true, 4>), dim3(gl), dim3(RT_BLOCK), 0, st, dv, pc, level, s->hot_table);
                        else if (s->opt_light_waves >= 8) hipLaunchKernelGGL((k_light<false, 8>), dim3(gl), dim3(RT_BLOCK), 0, st, dv, pc, level, s->hot_table);
                        else if (s->opt_light_waves >= 6) hipLaunchKernelGGL((k_light<false, 6>), dim3(gl), dim3(RT_BLOCK), 0, st, dv, pc, level, s->hot_table);
                        else if (s->opt_light_waves >= 5) hipLaunchKernelGGL((k_light<false, 5>), dim3(gl), dim3(RT_BLOCK), 0, st, dv, pc, level, s->hot_table);
                        else hipLaunchKernelGGL((k_light<false, 4>), dim3(gl), dim3(RT_BLOCK), 0, st, dv, pc, level, s->hot_table);
                        end_timing(s, st, ev);
                        if (token) { HIP_TRY(hipEventRecord(s->ws[k].light, st)); last_light = s->ws[k].light; }
                        begin_timing(s, st, 2, &ev);
                        if (s->opt_stats) hipLaunchKernelGGL(k_binary<true>, dim3(64), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else hipLaunchKernelGGL(k_binary<false>, dim3(64), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        end_timing(s, st, ev);
                        begin_timing(s, st, 1, &ev);
                        // a persistent kernel: no more workgroups than are resident at once (waves per SIMD of the variant = workgroups per CU),
                        // or the ones that start late run on a third of the chip (2.7 of 6 waves resident with grid_max: profiles/r03b)
                        const uint32_t tw = s->opt_stats ? 8u : (s->opt_shadow_waves >= 8 ? 8u : (s->opt_shadow_waves >= 6 ? 6u : 5u));
                        uint32_t gt = s->grid_max / 8u * tw * s->opt_trace_grid / 8u;
                        if (gt > gs) gt = gs;
                        if (gt < 1u) gt = 1u;
                        if (s->opt_stats) hipLaunchKernelGGL((k_trace_p<true, 8>), dim3(gt), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else if (tw == 8u) hipLaunchKernelGGL((k_trace_p<false, 8>), dim3(gt), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else if (tw == 6u) hipLaunchKernelGGL((k_trace_p<false, 6>), dim3(gt), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else hipLaunchKernelGGL((k_trace_p<false, 5>), dim3(gt), dim3(RT_BLOCK), 0, st, dv, pc, level);
                    } else if (hemi) {
                        if (s->opt_stats) hipLaunchKernelGGL((k_shadow<true, true>), dim3(gs), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else hipLaunchKernelGGL((k_shadow<false, true>), dim3(gs), dim3(RT_BLOCK), 0, st, dv, pc, level);
                    } else {
                        if (s->opt_stats) hipLaunchKernelGGL((k_shadow<true, false>), dim3(gs), dim3(RT_BLOCK), 0, st, dv, pc, level);
                        else hipLaunchKernelGGL((k_shadow<false, false>), dim3(gs), dim3(RT_BLOCK), 0, st, dv, pc, level);
                    }
                    end_timing(s, st, ev);
                }
                begin_timing(s, st, 2, &ev);
                hipLaunchKernelGGL(k_shade_b, dim3(g), dim3(RT_BLOCK), 0, st, dv, pc, level);
                end_timing(s, st, ev);
            }
          }
        }
        for (uint32_t k = 0; k < n_lanes; ++k) {
            if (!active[k]) continue;
            PassCtx& pc = pcs[k];
            const hipStream_t st = streams[k];
            size_t ev;
            begin_timing(s, st, 2, &ev);
            hipLaunchKernelGGL(k_resolve, dim3(grid_for(s, pc.npix)), dim3(RT_BLOCK), 0, st, pc);
            end_timing(s, st, ev);
            HIP_TRY(hipGetLastError());
            ++done;
        }
        if (any && (progress || interrupt)) {      // end of a round: one pass per lane is enqueued
            for (uint32_t j = 0; j < n_lanes; ++j) HIP_TRY(hipStreamSynchronize(streams[j]));
            if (progress && host_rgba && done < total_passes)
                HIP_TRY(hipMemcpy(host_rgba, rgba_dev, (size_t)P.width * P.height * 4, hipMemcpyDeviceToHost));
            if (progress) *progress = (uint32_t)((float)done / (float)total_passes * 100.0f);   // raytracer.rs:191, 213
        }
    }
    return RT_OK;
    };
    const int rc_run = run_passes();
    if (n_lanes > 1) {
        // join: the caller's stream waits for every lane, error or not; after an error or an interrupt the host waits as well
        int rc_join = RT_OK;
        for (uint32_t k = 0; k < n_lanes; ++k) {
            if (hipEventRecord(s->ws[k].done, streams[k]) != hipSuccess || hipStreamWaitEvent(stream, s->ws[k].done, 0) != hipSuccess) {
                (void)hipStreamSynchronize(streams[k]);
                if (!rc_join) rc_join = RT_ERR_HIP;
            }
        }
        if (rc_run)
            for (uint32_t k = 0; k < n_lanes; ++k) (void)hipStreamSynchronize(streams[k]);
        if (!rc_run && rc_join) return fail(RT_ERR_HIP, "joining the render lanes failed");
    }
    if (rc_run) return rc_run;
    // samples = owned valid pixels * spp (host arithmetic; edge tiles are clipped)
    {
        uint64_t valid = 0;
        for (uint32_t k = 0; k < sh.n_own_tiles; ++k) {
            const uint32_t t = P.shard_index + k * P.shard_count;
            const uint32_t x0 = (t % sh.tiles_x) * P.tile_w, y0 = (t / sh.tiles_x) * P.tile_h;
            const uint32_t wv = x0 + P.tile_w <= P.width ? P.tile_w : (P.width > x0 ? P.width - x0 : 0);
            const uint32_t hv = y0 + P.tile_h <= P.height ? P.tile_h : (P.height > y0 ? P.height - y0 : 0);
            valid += (uint64_t)wv * hv;
        }
        s->samples = valid * P.spp;
    }
    if (progress) *progress = 100;
    return RT_OK;
}

int cast_impl(rt_scene* s, const rt_render_params* params, uint32_t* ids_dev, float* t_dev, hipStream_t stream) {
    int rc;
    if ((rc = validate_params(params))) return rc;
    if ((rc = upload_scene(s))) return rc;
    const Shard sh = shard_of(*params);
    reset_counters(s, stream);
    if (sh.n_own_pix == 0) return RT_OK;
    if (sh.n_own_pix >= (1ull << 31)) return fail(RT_ERR_LIMIT, "shard larger than 2^31 pixels");
    const uint32_t g = grid_for(s, sh.n_own_pix);
    size_t ev;
    begin_timing(s, stream, 0, &ev);
    if (s->opt_stats)
        hipLaunchKernelGGL(k_primary<true>, dim3(g), dim3(RT_BLOCK), 0, stream, s->dev, *params, sh.tiles_x, (uint32_t)sh.n_own_pix, ids_dev,
                           t_dev, (uint8_t*)nullptr, s->w_stats.p);
    else
        hipLaunchKernelGGL(k_primary<false>, dim3(g), dim3(RT_BLOCK), 0, stream, s->dev, *params, sh.tiles_x, (uint32_t)sh.n_own_pix, ids_dev,
                           t_dev, (uint8_t*)nullptr, s->w_stats.p);
    end_timing(s, stream, ev);
    HIP_TRY(hipGetLastError());
    // one closest-hit ray per valid owned pixel
    uint64_t valid = 0;
    for (uint32_t k = 0; k < sh.n_own_tiles; ++k) {
        const uint32_t t = params->shard_index + k * params->shard_count;
        const uint32_t x0 = (t % sh.tiles_x) * params->tile_w, y0 = (t / sh.tiles_x) * params->tile_h;
        const uint32_t wv = x0 + params->tile_w <= params->width ? params->tile_w : (params->width > x0 ? params->width - x0 : 0);
        const uint32_t hv = y0 + params->tile_h <= params->height ? params->tile_h : (params->height > y0 ? params->height - y0 : 0);
        valid += (uint64_t)wv * hv;
    }
    s->samples = valid;
    return RT_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

int rt_init(int device) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(RT_ERR_HIP, "no HIP device visible");
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));
    g_device = device;
    g_inited = true;
    void* scratch_ptr = nullptr;
    return scratch_for(device, &scratch_ptr);   // the no-alloc entry point's scratch: its device address is looked up here, once per GPU
}

int rt_scene_create(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t n_tri, const float* sph,
                    const uint32_t* sph_mat, uint32_t n_sph, const rt_material* mats, uint32_t n_mat, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if ((n_tri && (!tri_pos || !tri_nrm || !tri_mat)) || (n_sph && (!sph || !sph_mat)) || ((n_tri || n_sph) && (!mats || !n_mat)))
        return fail(RT_ERR_INVALID, "NULL array with a non-zero count");
    if ((uint64_t)n_tri + n_sph >= (1ull << 30)) return fail(RT_ERR_LIMIT, "more than 2^30 objects");
    for (uint32_t i = 0; i < n_tri; ++i)
        if (tri_mat[i] >= n_mat) return fail(RT_ERR_INVALID, "triangle material index out of range");
    for (uint32_t i = 0; i < n_sph; ++i)
        if (sph_mat[i] >= n_mat) return fail(RT_ERR_INVALID, "sphere material index out of range");
    for (uint32_t i = 0; i < n_mat; ++i)
        if (mats[i].bsdf > RT_BSDF_SPECULAR) return fail(RT_ERR_INVALID, "unknown bsdf");
    rt_scene* s = new rt_scene();
    s->device = g_device;
    s->host.tri_pos.assign(tri_pos, tri_pos + 9 * (size_t)n_tri);
    s->host.tri_nrm.assign(tri_nrm, tri_nrm + 9 * (size_t)n_tri);
    s->host.tri_mat.assign(tri_mat, tri_mat + n_tri);
    s->host.sph.assign(sph, sph + 4 * (size_t)n_sph);
    s->host.sph_mat.assign(sph_mat, sph_mat + n_sph);
    s->host.mats.assign(mats, mats + n_mat);
    prepare_host(s);
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->device >= 0 && s->device < RT_MAX_DEVICES && g_scratch_owner[s->device] == s) g_scratch_owner[s->device] = nullptr;
    (void)hipSetDevice(s->device);
    if (s->uploaded) (void)hipDeviceSynchronize();
    s->d_nodes.release(); s->d_tris.release(); s->d_tri_nrm.release(); s->d_sph.release(); s->d_mats.release();
    s->d_sph_aux.release(); s->d_lights.release(); s->d_ovf.release(); s->d_wnodes.release(); s->d_wprims.release(); s->d_wrec.release(); s->d_wsib.release();
    for (int k = 0; k < RT_MAX_LANES; ++k) {
        s->ws[k].release();
        if (s->ws[k].stream) (void)hipStreamDestroy(s->ws[k].stream);
        if (s->ws[k].done) (void)hipEventDestroy(s->ws[k].done);
        if (s->ws[k].light) (void)hipEventDestroy(s->ws[k].light);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    s->w_accum.release(); s->w_jump.release(); s->w_stats.release();
    s->w_frame.release(); s->w_ids.release(); s->w_ts.release(); s->w_rays.release();
    for (size_t i = 0; i < s->events.size(); ++i) { (void)hipEventDestroy(s->events[i].a); (void)hipEventDestroy(s->events[i].b); }
    delete s;
}

int rt_scene_info_get(const rt_scene* s, rt_scene_info* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "NULL argument");
    out->n_tri = s->host.n_tri(); out->n_sph = s->host.n_sph(); out->n_mat = (uint32_t)s->host.mats.size();
    out->n_lights = (uint32_t)s->lights.size();
    out->n_inner_nodes = s->bvh.n_inner(); out->bvh_depth = s->bvh.depth; out->device_bytes = s->device_bytes;
    out->n_wide_nodes = s->wide.n_nodes(); out->n_wide_prims = s->wide.n_prims(); out->wide_depth = s->wide.depth; out->reserved = 0;
    return RT_OK;
}

int rt_set_global_option(const char* name, uint64_t value) {
    if (!name) return fail(RT_ERR_INVALID, "NULL argument");
    const std::string n(name);
    if (n == "device_build") g_device_build = value ? 1u : 0u;
    else if (n == "device_build_min") g_device_build_min = (uint32_t)value;
    else if (n == "alloc_limit") g_alloc_limit = value;
    else return fail(RT_ERR_INVALID, "unknown option '" + n + "'");
    return RT_OK;
}

int rt_scene_build_info(const rt_scene* s, double* ms4, uint32_t* on_device2, char* note, uint32_t note_capacity) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    if (ms4) { ms4[0] = s->build_ms[0]; ms4[1] = s->build_ms[1]; ms4[2] = s->build_dev_ms[0]; ms4[3] = s->build_dev_ms[1]; }
    if (on_device2) { on_device2[0] = s->build_on_device[0]; on_device2[1] = s->build_on_device[1]; }
    if (note && note_capacity) { strncpy(note, s->build_note.c_str(), note_capacity - 1); note[note_capacity - 1] = 0; }
    return RT_OK;
}

int rt_scene_lights(const rt_scene* s, uint32_t* out, uint32_t capacity) {
    if (!s || (!out && capacity)) return fail(RT_ERR_INVALID, "NULL argument");
    for (uint32_t i = 0; i < capacity && i < s->lights.size(); ++i) out[i] = s->lights[i];
    return RT_OK;
}

int rt_scene_bvh(const rt_scene* s, uint32_t* node_words, uint32_t* rank) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    if (node_words && !s->bvh.words.empty()) memcpy(node_words, s->bvh.words.data(), s->bvh.words.size() * 4);
    if (rank && !s->bvh.rank.empty()) memcpy(rank, s->bvh.rank.data(), s->bvh.rank.size() * 4);
    return RT_OK;
}

int rt_scene_wide_bvh(const rt_scene* s, uint32_t* node_words, float* prim_records) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    if (node_words && !s->wide.nodes.empty()) memcpy(node_words, s->wide.nodes.data(), s->wide.nodes.size() * 4);
    if (prim_records && !s->wide.prims.empty()) memcpy(prim_records, s->wide.prims.data(), s->wide.prims.size() * 4);
    return RT_OK;
}

int rt_render_device(rt_scene* s, const rt_render_params* params, void* rgba_dev, void* stream) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    return render_impl(s, params, (uint8_t*)rgba_dev, (hipStream_t)stream, nullptr, nullptr);
}

int rt_gather_tiles(rt_comm* c, const rt_render_params* P, void* frame_dev, uint32_t bytes_per_pixel, void* stream);
}  // extern "C"
namespace {
int gather_prepare(rt_comm* c, const rt_render_params* P, uint32_t bytes_per_pixel);
int gather_run(rt_comm* c, const rt_render_params* P, void* frame_dev, uint32_t bytes_per_pixel, hipStream_t stream);
void comm_abort(rt_comm* c);
}  // namespace
extern "C" {

// Two phases, so that a rank that fails does not leave its peers waiting in ncclAllGather for ever:
//   1. everything local that can fail (parameter checks, the frame and gather buffers, the upload, the render itself);
//   2. the collective, which this rank enters whenever its buffers exist — after a failed or interrupted render too, with whatever its
//      frame holds — and only then reports the local error.  If the buffers themselves could not be had, the communicator is aborted
//      (ncclCommAbort), which makes the peers' collective return an error instead of hanging; the communicator is unusable afterwards.
int rt_render_comm(rt_scene* s, const rt_render_params* params, rt_comm* comm, uint8_t* rgba, volatile uint32_t* progress, const volatile uint32_t* interrupt) {
    if (!s || !rgba) { comm_abort(comm); return fail(RT_ERR_INVALID, "NULL argument"); }
    int rc;
    if ((rc = validate_params(params))) { const std::string why = g_error; comm_abort(comm); return fail(rc, why); }
    BIND(s->device);
    const size_t bytes = (size_t)params->width * params->height * 4;
    rc = s->w_frame.ensure(bytes);
    if (!rc && comm) rc = gather_prepare(comm, params, 4);
    if (rc) { const std::string why = g_error; comm_abort(comm); return fail(rc, why + (comm ? " (communicator aborted: the peers' collective fails instead of waiting)" : "")); }
    std::string why;
    if (hipMemcpy(s->w_frame.p, rgba, bytes, hipMemcpyHostToDevice) != hipSuccess) {   // pixels of other shards / of an interrupted render stay as the caller has them
        rc = RT_ERR_HIP; why = "copying the caller's frame to the device failed";
    } else {
        rc = render_impl(s, params, s->w_frame.p, nullptr, progress, interrupt, rgba);
        if (rc) why = g_error;
    }
    if (comm) {
        const int rg = gather_run(comm, params, s->w_frame.p, 4, nullptr);
        if (rg && !rc) { rc = rg; why = g_error; }
    }
    if (hipDeviceSynchronize() != hipSuccess && !rc) { rc = RT_ERR_HIP; why = "hipDeviceSynchronize failed after the render"; }
    if ((rc == RT_OK || rc == RT_ERR_INTERRUPTED) && hipMemcpy(rgba, s->w_frame.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) {   // partial pixels stay in the buffer on interrupt
        rc = RT_ERR_HIP; why = "copying the frame back failed";
    }
    return rc ? fail(rc, why) : RT_OK;
}

int rt_render(rt_scene* s, const rt_render_params* params, uint8_t* rgba, volatile uint32_t* progress, const volatile uint32_t* interrupt) {
    return rt_render_comm(s, params, nullptr, rgba, progress, interrupt);
}

int rt_cast_primary_device(rt_scene* s, const rt_render_params* params, void* ids_dev, void* t_dev, void* stream) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    return cast_impl(s, params, (uint32_t*)ids_dev, (float*)t_dev, (hipStream_t)stream);
}

int rt_cast_primary_noalloc(rt_scene* s, const rt_render_params* params, void* ids_dev, void* t_dev, void* stream_) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    if (!ids_dev && !t_dev) return fail(RT_ERR_INVALID, "NULL argument");
    int rc;
    if ((rc = validate_params(params))) return rc;
    if (!s->uploaded)
        return fail(RT_ERR_INVALID, "rt_cast_primary_noalloc: the scene is not resident on the GPU yet (uploading allocates): make one call of "
                                    "rt_cast_primary_device or rt_render_device first");
    if (s->ovf_rows > RT_NOALLOC_ROWS)
        return fail(RT_ERR_LIMIT, "rt_cast_primary_noalloc: the scene's BVH is deeper than the static scratch holds (" + std::to_string(RT_STACK_LDS + RT_NOALLOC_ROWS) +
                                  " stack entries per ray); use rt_cast_primary_device");
    BIND(s->device);
    void* ptr = nullptr;
    if ((rc = scratch_for(s->device, &ptr))) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const Shard sh = shard_of(*params);
    s->events_used = 0;
    s->launches[0] = 1; s->launches[1] = s->launches[2] = s->launches[3] = 0;
    s->samples = 0;
    s->stats_src = (const unsigned long long*)ptr;
    if (s->device >= 0 && s->device < RT_MAX_DEVICES) g_scratch_owner[s->device] = s;
    // the counters start at zero; the overflow rows are written before they are read
    HIP_TRY(hipMemsetAsync(ptr, 0, 256, stream));
    if (sh.n_own_pix == 0) return RT_OK;
    if (sh.n_own_pix >= (1ull << 31)) return fail(RT_ERR_LIMIT, "shard larger than 2^31 pixels");
    uint32_t g = grid_for(s, sh.n_own_pix);
    if (g > RT_NOALLOC_GRID) g = RT_NOALLOC_GRID;
    DevScene d = s->dev;
    d.ovf = (uint32_t*)((unsigned char*)ptr + 256);
    d.ovf_stride = g * RT_BLOCK;
    if (s->opt_stats)
        hipLaunchKernelGGL(k_primary<true>, dim3(g), dim3(RT_BLOCK), 0, stream, d, *params, sh.tiles_x, (uint32_t)sh.n_own_pix, (uint32_t*)ids_dev,
                           (float*)t_dev, (uint8_t*)nullptr, (unsigned long long*)ptr);
    else
        hipLaunchKernelGGL(k_primary<false>, dim3(g), dim3(RT_BLOCK), 0, stream, d, *params, sh.tiles_x, (uint32_t)sh.n_own_pix, (uint32_t*)ids_dev,
                           (float*)t_dev, (uint8_t*)nullptr, (unsigned long long*)ptr);
    HIP_TRY(hipGetLastError());
    s->samples = valid_pixels(*params, sh);
    return RT_OK;
}

int rt_cast_primary(rt_scene* s, const rt_render_params* params, uint32_t* ids, float* t) {
    if (!s || (!ids && !t)) return fail(RT_ERR_INVALID, "NULL argument");
    int rc;
    if ((rc = validate_params(params))) return rc;
    BIND(s->device);
    const size_t n = (size_t)params->width * params->height;
    if (ids) { if ((rc = s->w_ids.ensure(n))) return rc; HIP_TRY(hipMemcpy(s->w_ids.p, ids, n * 4, hipMemcpyHostToDevice)); }
    if (t) { if ((rc = s->w_ts.ensure(n))) return rc; HIP_TRY(hipMemcpy(s->w_ts.p, t, n * 4, hipMemcpyHostToDevice)); }
    rc = cast_impl(s, params, ids ? s->w_ids.p : nullptr, t ? s->w_ts.p : nullptr, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    if (rc) return rc;
    if (ids) HIP_TRY(hipMemcpy(ids, s->w_ids.p, n * 4, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, s->w_ts.p, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_intersect_rays(rt_scene* s, const float* origins, const float* directions, uint32_t n, uint32_t* ids, float* t) {
    if (!s || !origins || !directions || !ids || !t) return fail(RT_ERR_INVALID, "NULL argument");
    if (!n) return RT_OK;
    int rc;
    if ((rc = upload_scene(s))) return rc;
    if ((rc = s->w_rays.ensure((size_t)n * 8))) return rc;
    float* d_o = s->w_rays.p; float* d_d = d_o + 3 * (size_t)n; float* d_t = d_d + 3 * (size_t)n; uint32_t* d_i = (uint32_t*)(d_t + n);
    HIP_TRY(hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rays_closest, dim3(grid_for(s, n)), dim3(RT_BLOCK), 0, nullptr, s->dev, d_o, d_d, n, d_i, d_t);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ids, d_i, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_occluded_rays(rt_scene* s, const float* origins, const float* directions, const float* max_dist, uint32_t n, uint8_t* out) {
    if (!s || !origins || !directions || !max_dist || !out) return fail(RT_ERR_INVALID, "NULL argument");
    if (!n) return RT_OK;
    int rc;
    if ((rc = upload_scene(s))) return rc;
    if ((rc = s->w_rays.ensure((size_t)n * 8))) return rc;
    float* d_o = s->w_rays.p; float* d_d = d_o + 3 * (size_t)n; float* d_m = d_d + 3 * (size_t)n; uint8_t* d_r = (uint8_t*)(d_m + n);
    HIP_TRY(hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_m, max_dist, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rays_occluded, dim3(grid_for(s, n)), dim3(RT_BLOCK), 0, nullptr, s->dev, d_o, d_d, d_m, n, d_r);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_r, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_counters(const rt_scene* cs, rt_counters* out) {
    if (!cs || !out) return fail(RT_ERR_INVALID, "NULL argument");
    rt_scene* s = const_cast<rt_scene*>(cs);
    memset(out, 0, sizeof *out);
    if (!s->uploaded) return RT_OK;
    BIND(s->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long st[ST_COUNT];
    if (s->stats_src && s->stats_src != s->w_stats.p && s->device >= 0 && s->device < RT_MAX_DEVICES && g_scratch_owner[s->device] != s)
        return fail(RT_ERR_INVALID, "the counters of this scene's last call (rt_cast_primary_noalloc) lived in the GPU's shared scratch, and another scene's call has overwritten them since");
    HIP_TRY(hipMemcpy(st, s->stats_src ? s->stats_src : s->w_stats.p, sizeof st, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->rays_closest = st[ST_RAYS_CLOSEST];
    out->rays_shadow = st[ST_RAYS_SHADOW];
    out->samples = s->samples;
    if (out->rays_closest == 0 && s->launches[0] && s->samples) out->rays_closest = s->samples;   // primary cast: one ray per owned pixel
    out->closest_inner = st[ST_C_INNER]; out->closest_tris = st[ST_C_TRIS]; out->closest_sphs = st[ST_C_SPHS];
    out->shadow_inner = st[ST_S_INNER]; out->shadow_tris = st[ST_S_TRIS]; out->shadow_sphs = st[ST_S_SPHS]; out->shadow_wide = st[ST_S_WIDE];
    for (int k = 0; k < 8; ++k) out->trace_steps[k] = st[ST_P_NODE_STEPS + k];
    for (int k = 0; k < 4; ++k) out->shaft[k] = st[ST_SHAFT_TRIED + k];
    out->closest_hits_tri = st[ST_HITS_TRI]; out->closest_hits_sph = st[ST_HITS_SPH];
    double ms[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < s->events_used; ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, s->events[i].a, s->events[i].b) == hipSuccess) ms[s->events[i].cls] += t;
    }
    out->kernel_ms_closest = ms[0]; out->kernel_ms_shadow = ms[1]; out->kernel_ms_shade = ms[2]; out->kernel_ms_light = ms[3];
    out->launches_closest = s->launches[0]; out->launches_shadow = s->launches[1];
    return RT_OK;
}

int rt_set_option(rt_scene* s, const char* name, uint64_t value) {
    if (!s || !name) return fail(RT_ERR_INVALID, "NULL argument");
    const std::string n(name);
    if (n == "stats") s->opt_stats = value != 0;
    else if (n == "timing") s->opt_timing = value != 0;
    else if (n == "max_paths") { s->opt_max_paths = value ? value : (1ull << 26); s->max_paths_fit = 0; }
    else if (n == "self_test") { s->opt_self_test = value ? 1 : 0; s->dev.self_test = s->opt_self_test; }
    else if (n == "wide") { s->opt_wide = value ? 1 : 0; s->dev.n_wnodes = (s->wide.ok && s->opt_wide) ? s->wide.n_nodes() : 0u; }
    else if (n == "persist") s->opt_persist = value ? 1 : 0;
    else if (n == "hot") { if (s->uploaded) return fail(RT_ERR_INVALID, "option 'hot' must be set before the first render"); s->opt_hot = value ? 1 : 0; }
    else if (n == "shadow_waves") s->opt_shadow_waves = (uint32_t)value;
    else if (n == "light_waves") s->opt_light_waves = (uint32_t)value;
    else if (n == "refill_min") { s->opt_refill_min = value > 64 ? 64 : (value < 1 ? 1 : (uint32_t)value); s->dev.refill_min = s->opt_refill_min; }
    else if (n == "phase_bias") { s->opt_phase_bias = (uint32_t)value; s->dev.phase_bias = s->opt_phase_bias; }
    else if (n == "lane_min_pixels") s->opt_lane_min_pixels = value < 64 ? 64 : value;
    else if (n == "lanes") s->opt_lanes = value < 1 ? 1 : (value > RT_MAX_LANES ? RT_MAX_LANES : (uint32_t)value);
    else if (n == "extend_persist") s->opt_extend_persist = (uint32_t)value;
    else if (n == "extend_refill") { s->opt_extend_refill = value > 64 ? 64 : (value < 1 ? 1 : (uint32_t)value); s->dev.extend_refill_min = s->opt_extend_refill; }
    else if (n == "extend_inner") { s->opt_extend_inner = value > 64 ? 64 : (uint32_t)value; s->dev.extend_inner_min = s->opt_extend_inner; }
    else if (n == "extend_grid") s->opt_extend_grid = value < 1 ? 1 : (uint32_t)value;
    else if (n == "light_token") s->opt_light_token = value ? 1 : 0;
    else if (n == "trace_grid") s->opt_trace_grid = value < 1 ? 1 : (uint32_t)value;
    else if (n == "graze_guard") { s->opt_graze_guard = value ? 1 : 0; s->dev.graze_guard = s->opt_graze_guard; }
    else if (n == "fail_after_passes") s->opt_fail_after_passes = value;
    else if (n == "light_sum") s->opt_light_sum = value ? 1 : 0;
    else if (n == "camera_once") s->opt_camera_once = value ? 1 : 0;
    else if (n == "light_grid") s->opt_light_grid = (uint32_t)value;
    else if (n == "force_binary") s->opt_force_binary = (uint32_t)value;
    else if (n == "shaft") s->opt_shaft = value > RT_STACK_LDS ? RT_STACK_LDS : (uint32_t)value;
    else if (n == "inner_min") { s->opt_inner_min = value > 64 ? 64 : (uint32_t)value; s->dev.inner_min = s->opt_inner_min; }
    else return fail(RT_ERR_INVALID, "unknown option '" + n + "'");
    return RT_OK;
}

int rt_get_option(const rt_scene* s, const char* name, uint64_t* value) {
    if (!s || !name || !value) return fail(RT_ERR_INVALID, "NULL argument");
    const std::string n(name);
    const struct { const char* name; uint64_t v; } table[] = {
        {"stats", s->opt_stats}, {"timing", s->opt_timing}, {"max_paths", s->opt_max_paths}, {"max_paths_fit", s->max_paths_fit}, {"self_test", s->opt_self_test}, {"wide", s->opt_wide},
        {"persist", s->opt_persist}, {"hot", s->opt_hot}, {"shadow_waves", s->opt_shadow_waves}, {"light_waves", s->opt_light_waves},
        {"refill_min", s->opt_refill_min}, {"phase_bias", s->opt_phase_bias}, {"inner_min", s->opt_inner_min}, {"lanes", s->opt_lanes},
        {"lane_min_pixels", s->opt_lane_min_pixels}, {"extend_persist", s->opt_extend_persist}, {"extend_refill", s->opt_extend_refill},
        {"extend_inner", s->opt_extend_inner}, {"extend_grid", s->opt_extend_grid}, {"shaft", s->opt_shaft}, {"trace_grid", s->opt_trace_grid}, {"light_token", s->opt_light_token}, {"graze_guard", s->opt_graze_guard}, {"light_sum", s->opt_light_sum}, {"camera_once", s->opt_camera_once}, {"light_grid", s->opt_light_grid}, {"force_binary", s->opt_force_binary}};
    for (const auto& e : table)
        if (n == e.name) { *value = e.v; return RT_OK; }
    return fail(RT_ERR_INVALID, "unknown option '" + n + "'");
}

uint32_t rt_tiles_per_shard(const rt_render_params* P) {
    if (!P || !P->tile_w || !P->tile_h || !P->shard_count) return 0;
    const uint32_t tiles = ((P->width + P->tile_w - 1) / P->tile_w) * ((P->height + P->tile_h - 1) / P->tile_h);
    return (tiles + P->shard_count - 1) / P->shard_count;
}

int rt_tiles_pack(const rt_render_params* P, const void* frame_dev, void* compact_dev, uint32_t bytes_per_pixel, void* stream) {
    int rc;
    if ((rc = validate_params(P))) return rc;
    if (!frame_dev || !compact_dev || bytes_per_pixel == 0 || bytes_per_pixel % 4) return fail(RT_ERR_INVALID, "bad tile copy arguments");
    BIND(g_device);
    const uint32_t slots = rt_tiles_per_shard(P);
    const uint32_t tiles_x = (P->width + P->tile_w - 1) / P->tile_w;
    const uint64_t total = (uint64_t)slots * P->tile_w * P->tile_h;
    uint64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_tiles_copy, dim3((uint32_t)g), dim3(256), 0, (hipStream_t)stream, *P, tiles_x, P->shard_index, slots,
                       (const uint32_t*)frame_dev, (uint32_t*)compact_dev, bytes_per_pixel / 4, 0);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_tiles_unpack(const rt_render_params* P, uint32_t shard_index, const void* compact_dev, void* frame_dev, uint32_t bytes_per_pixel,
                    void* stream) {
    int rc;
    if ((rc = validate_params(P))) return rc;
    if (!frame_dev || !compact_dev || bytes_per_pixel == 0 || bytes_per_pixel % 4 || shard_index >= P->shard_count)
        return fail(RT_ERR_INVALID, "bad tile copy arguments");
    BIND(g_device);
    const uint32_t slots = rt_tiles_per_shard(P);
    const uint32_t tiles_x = (P->width + P->tile_w - 1) / P->tile_w;
    const uint64_t total = (uint64_t)slots * P->tile_w * P->tile_h;
    uint64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_tiles_copy, dim3((uint32_t)g), dim3(256), 0, (hipStream_t)stream, *P, tiles_x, shard_index, slots,
                       (const uint32_t*)compact_dev, (uint32_t*)frame_dev, bytes_per_pixel / 4, 1);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---- multi-GPU: the frame's screen tiles over RCCL ----
}  // extern "C"

struct rt_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;   // around the last gather
    bool timed = false;
    std::string describe;         // which RCCL this process resolved to (path of the shared object, version), for the bench line
    DevBuf<uint8_t> send, recv;   // compact tiles of this rank / of every rank; grow-only: no allocation per frame once sized
};

namespace {
#define NCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_));  \
    } while (0)
}  // namespace

extern "C" {

int rt_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(RT_ERR_INVALID, "NULL argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id128, &id, 128);
    return RT_OK;
}

int rt_comm_init_rank(const uint8_t* id128, int world, int rank, rt_comm** out) {
    if (!id128 || !out) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad world size or rank");
    BIND(g_device);
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    rt_comm* c = new rt_comm();
    c->world = world; c->rank = rank; c->device = g_device;
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(RT_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    {
        // which librccl the dynamic loader bound this library's ncclAllGather to (a process that has torch loaded carries torch's copy too)
        int ver = 0;
        (void)ncclGetVersion(&ver);
        Dl_info info;
        const char* path = (dladdr((void*)&ncclAllGather, &info) && info.dli_fname) ? info.dli_fname : "?";
        c->describe = std::string("RCCL ") + std::to_string(ver) + " from " + path;
    }
    *out = c;
    return RT_OK;
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    c->send.release(); c->recv.release();
    delete c;
}

int rt_gather_tiles(rt_comm* c, const rt_render_params* P, void* frame_dev, uint32_t bytes_per_pixel, void* stream) {
    if (!c || !frame_dev) return fail(RT_ERR_INVALID, "NULL argument");
    int rc;
    if ((rc = gather_prepare(c, P, bytes_per_pixel))) return rc;   // local checks and buffers first: nothing below fails before the collective is entered
    return gather_run(c, P, frame_dev, bytes_per_pixel, (hipStream_t)stream);
}

const char* rt_comm_describe(const rt_comm* c) { return c ? c->describe.c_str() : ""; }

}  // extern "C"

namespace {
// The part of the gather that can fail on one rank alone: argument checks and the (grow-only) pack / receive buffers.
int gather_prepare(rt_comm* c, const rt_render_params* P, uint32_t bytes_per_pixel) {
    if (!c) return fail(RT_ERR_INVALID, "NULL communicator");
    if (!c->comm) return fail(RT_ERR_INVALID, "the communicator was aborted after an earlier failure");
    int rc;
    if ((rc = validate_params(P))) return rc;
    if (bytes_per_pixel == 0 || bytes_per_pixel % 4) return fail(RT_ERR_INVALID, "bytes_per_pixel must be a multiple of 4");
    if (P->shard_count != (uint32_t)c->world || P->shard_index != (uint32_t)c->rank)
        return fail(RT_ERR_INVALID, "params name shard " + std::to_string(P->shard_index) + " of " + std::to_string(P->shard_count) + " but the communicator is rank " +
                                    std::to_string(c->rank) + " of " + std::to_string(c->world));
    BIND(c->device);
    const size_t bytes = (size_t)rt_tiles_per_shard(P) * P->tile_w * P->tile_h * bytes_per_pixel;   // equal on every rank (padded)
    if ((rc = c->send.ensure(bytes)) || (rc = c->recv.ensure(bytes * (size_t)c->world))) return rc;   // first frame of a geometry only
    return RT_OK;
}
// pack -> ncclAllGather -> unpack of every rank's tiles on `stream`; HIP events around it give rt_comm_last_gather_ms.
int gather_run(rt_comm* c, const rt_render_params* P, void* frame_dev, uint32_t bytes_per_pixel, hipStream_t stream) {
    BIND(c->device);
    const size_t bytes = (size_t)rt_tiles_per_shard(P) * P->tile_w * P->tile_h * bytes_per_pixel;
    if (!c->ev_a) { (void)hipEventCreate(&c->ev_a); (void)hipEventCreate(&c->ev_b); }
    if (c->ev_a) (void)hipEventRecord(c->ev_a, stream);
    // a failed pack launch must not keep this rank out of the collective: remember it and go on
    const int rp = rt_tiles_pack(P, frame_dev, c->send.p, bytes_per_pixel, stream);
    const std::string why_pack = rp ? g_error : std::string();
    NCCL_TRY(ncclAllGather(c->send.p, c->recv.p, bytes, ncclUint8, c->comm, stream));
    int rc = rp;
    for (int r = 0; r < c->world && !rc; ++r) rc = rt_tiles_unpack(P, (uint32_t)r, c->recv.p + bytes * (size_t)r, frame_dev, bytes_per_pixel, stream);
    if (c->ev_b) { (void)hipEventRecord(c->ev_b, stream); c->timed = true; }
    return rp ? fail(rp, why_pack) : rc;
}
void comm_abort(rt_comm* c) {
    if (!c || !c->comm) return;
    (void)ncclCommAbort(c->comm);
    c->comm = nullptr;
}
}  // namespace

extern "C" {

// Milliseconds between the HIP events around the last rt_gather_tiles / rt_render_comm gather (pack + all-gather + unpack); waits for it.
int rt_comm_last_gather_ms(rt_comm* c, double* ms) {
    if (!c || !ms) return fail(RT_ERR_INVALID, "NULL argument");
    *ms = 0.0;
    if (!c->timed) return RT_OK;
    BIND(c->device);
    HIP_TRY(hipEventSynchronize(c->ev_b));
    float t = 0.0f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev_a, c->ev_b));
    *ms = t;
    return RT_OK;
}

// ---- host-side helpers ----
void rt_screen_params(uint32_t width, uint32_t height, float fov, rt_render_params* p) {
    const float w = (float)width, h = (float)height;
    const float half_fov = fov * 0.5f;
    const float start = sin_f(-half_fov);
    p->width = width; p->height = height;
    p->inv_w = 1.0f / w; p->inv_h = 1.0f / h;
    p->start = start;
    p->total = -2.0f * start;
    p->aspect_ratio = w / h;
    p->z = 1.7f;
}

int rt_scene_new_named(const char* name, const char* obj_path, rt_scene** out) {
    if (!out || !name) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    rt_scene* s = new rt_scene();
    s->device = g_device;
    std::string err;
    if (!make_named_scene(name, obj_path, nullptr, nullptr, 0, nullptr, 0, &s->host, &err)) { delete s; return fail(RT_ERR_INVALID, err); }
    prepare_host(s);
    *out = s;
    return RT_OK;
}

int rt_scene_new_dragon_mesh(const float* vpos, const float* vnrm, uint32_t n_vert, const uint32_t* faces, uint32_t n_face, rt_scene** out) {
    if (!out || !vpos || !faces) return fail(RT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    for (size_t i = 0; i < 3 * (size_t)n_face; ++i)
        if (faces[i] >= n_vert) return fail(RT_ERR_INVALID, "face index out of range");
    rt_scene* s = new rt_scene();
    s->device = g_device;
    std::string err;
    if (!make_named_scene("dragon", nullptr, vpos, vnrm, n_vert, faces, n_face, &s->host, &err)) { delete s; return fail(RT_ERR_INVALID, err); }
    prepare_host(s);
    *out = s;
    return RT_OK;
}

int rt_scene_export(const rt_scene* s, float* tri_pos, float* tri_nrm, uint32_t* tri_mat, float* sph, uint32_t* sph_mat, rt_material* mats) {
    if (!s) return fail(RT_ERR_INVALID, "NULL scene");
    const HostScene& h = s->host;
    if (tri_pos && !h.tri_pos.empty()) memcpy(tri_pos, h.tri_pos.data(), h.tri_pos.size() * 4);
    if (tri_nrm && !h.tri_nrm.empty()) memcpy(tri_nrm, h.tri_nrm.data(), h.tri_nrm.size() * 4);
    if (tri_mat && !h.tri_mat.empty()) memcpy(tri_mat, h.tri_mat.data(), h.tri_mat.size() * 4);
    if (sph && !h.sph.empty()) memcpy(sph, h.sph.data(), h.sph.size() * 4);
    if (sph_mat && !h.sph_mat.empty()) memcpy(sph_mat, h.sph_mat.data(), h.sph_mat.size() * 4);
    if (mats && !h.mats.empty()) memcpy(mats, h.mats.data(), h.mats.size() * sizeof(rt_material));
    return RT_OK;
}

int rt_make_dragon_standin(uint32_t nu, uint32_t nv, float* vpos, float* vnrm, uint32_t* faces) {
    if (nu < 3 || nv < 3 || !vpos || !vnrm || !faces) return fail(RT_ERR_INVALID, "bad stand-in mesh arguments");
    make_dragon_standin(nu, nv, vpos, vnrm, faces);
    return RT_OK;
}

// ---- debug probes ----
int rt_debug_math(uint32_t op, const float* x, const float* y, uint32_t n, float* out) {
    if (!x || !out || (op == 3 && !y)) return fail(RT_ERR_INVALID, "NULL argument");
    if (!n) return RT_OK;
    BIND(g_device);
    DevBuf<float> dx, dy, dout;   // released on every path out
    struct Guard { DevBuf<float>*a, *b, *c; ~Guard() { a->release(); b->release(); c->release(); } } guard{&dx, &dy, &dout};
    int rc;
    if ((rc = dx.ensure(n)) || (rc = dy.ensure(n)) || (rc = dout.ensure(n))) return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
    if (y) HIP_TRY(hipMemcpy(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, dx.p, dy.p, n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_exact(uint32_t op, uint64_t* out2) {
    if (!out2 || op > 3u) return fail(RT_ERR_INVALID, "bad argument");
    BIND(g_device);
    DevBuf<unsigned long long> dout;
    struct Guard { DevBuf<unsigned long long>* a; ~Guard() { a->release(); } } guard{&dout};
    int rc;
    if ((rc = dout.ensure(2))) return rc;
    HIP_TRY(hipMemset(dout.p, 0, 16));
    hipLaunchKernelGGL(k_debug_exact, dim3(4096), dim3(256), 0, nullptr, op, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out2, dout.p, 16, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_rng(uint64_t seed, uint32_t pixel_index, uint32_t sample, uint32_t n, float* out) {
    if (!out) return fail(RT_ERR_INVALID, "NULL argument");
    if (!n) return RT_OK;
    BIND(g_device);
    DevBuf<float> dout;
    struct Guard { DevBuf<float>* a; ~Guard() { a->release(); } } guard{&dout};
    int rc;
    if ((rc = dout.ensure(n))) return rc;
    hipLaunchKernelGGL(k_debug_rng, dim3(1), dim3(64), 0, nullptr, mix64(seed), pixel_index, sample, n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
