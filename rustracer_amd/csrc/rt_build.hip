// rt_build.hip — the scene's two trees built on the GPU: BVH::build as Scene::new calls it (src/scene/mod.rs:99) and
// the 8-wide occlusion tree, so that set_scene (src/raytracer.rs:514-518) does not wait for a host build.
//
// The result is the host builders' (host_scene.cpp), word for word: same splits, same node numbering, same ranks, same quantised planes
// (tests/test_gpu_parity.py::test_device_built_trees_equal_the_host_built).  That holds because every value that decides something is
// computed by the same single-precision operations in the same order (this file is compiled with -ffp-contract=off like the rest), and
// everything accumulated across objects is a minimum, a maximum or an integer count, which do not depend on the order of accumulation.
// Inputs for which that argument does not hold (a NaN or infinity among the coordinates, a negative zero, which `a < b ? a : b` keeps or
// drops depending on the order) are declined and built on the host.
//
// Binary tree, level by level (the host recursion is depth-first; with one shape per leaf the pre-order index of every node and the rank
// of every leaf follow from the sizes of the segments alone, so the order of construction is free):
//   a segment = a range [b, e) of the object permutation `idx` that becomes one inner node.  Per level, over all objects at once:
//   k_bounds   box of the segment's object boxes and of their centroids           (ordered-integer atomic min / max)
//   k_bucket   the crate's 6 SAH buckets along the widest centroid axis           (atomics; LDS when a workgroup lies inside one segment)
//   radix sort of (segment start, bucket) keys: a stable counting sort of every segment at once (rocPRIM)
//   k_plan     the 5 candidate splits, cost in f32 exactly as the host, fallback to halving
//   k_halfbox  child boxes of halved segments; k_emit writes the node, the ranks of its leaf children and the next level's segments
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "host_scene.h"
#include "rt_build.h"

namespace rt {
namespace {

#define BT 256
#define NONE 0xFFFFFFFFu
#define HIPB(expr)                                                                  \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

// monotone map float -> uint32: a < b  <=>  enc(a) < enc(b) (no NaN, no -0 among the inputs)
__host__ __device__ inline uint32_t enc(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float dec(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
#define ENC_PINF 0xFF800000u   // enc(+inf): start of a running minimum
#define ENC_NINF 0x007FFFFFu   // enc(-inf): start of a running maximum

__device__ inline float lo(float a, float b) { return a < b ? a : b; }
__device__ inline float hi(float a, float b) { return a > b ? a : b; }
struct DBox {
    float mn[3], mx[3];
};
__device__ inline DBox empty_box() {
    DBox b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = __builtin_inff(); b.mx[k] = -__builtin_inff(); }
    return b;
}
__device__ inline void join_into(DBox* a, const DBox& b) {
    for (int k = 0; k < 3; ++k) { a->mn[k] = lo(a->mn[k], b.mn[k]); a->mx[k] = hi(a->mx[k], b.mx[k]); }
}
__device__ inline float area(const DBox& b) {
    const float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}

// Bounded for Triangle / Sphere (objects.rs:293-316) and the centroid bvh 0.3.2 sorts by; same expressions as host_scene.cpp Builder.
__global__ void k_obj_boxes(const float* tri_pos, const float* sph, uint32_t T, uint32_t S, float* box, float* ctr, uint32_t* bad) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= T + S) return;
    float mn[3], mx[3];
    if (i < T) {
        const float* p = tri_pos + 9 * (size_t)i;
        for (int k = 0; k < 3; ++k) { mn[k] = lo(lo(p[k], p[3 + k]), p[6 + k]); mx[k] = hi(hi(p[k], p[3 + k]), p[6 + k]); }
    } else {
        const float* c = sph + 4 * (size_t)(i - T);
        for (int k = 0; k < 3; ++k) { mn[k] = c[k] - c[3]; mx[k] = c[k] + c[3]; }
    }
    bool b = false;
    for (int k = 0; k < 3; ++k) {
        box[6 * (size_t)i + k] = mn[k];
        box[6 * (size_t)i + 3 + k] = mx[k];
        const float c = mn[k] + (mx[k] - mn[k]) / 2.0f;
        ctr[3 * (size_t)i + k] = c;
        // finite, and no negative zero, in everything a minimum or maximum is taken of
        uint32_t u0, u1, u2;
        memcpy(&u0, &mn[k], 4); memcpy(&u1, &mx[k], 4); memcpy(&u2, &c, 4);
        b = b || !(fabsf(mn[k]) < __builtin_inff()) || !(fabsf(mx[k]) < __builtin_inff()) || !(fabsf(c) < __builtin_inff()) || u0 == 0x80000000u || u1 == 0x80000000u ||
            u2 == 0x80000000u;
    }
    if (b) *bad = 1u;
}
__global__ void k_init_perm(uint32_t n, uint32_t* idx, uint32_t* seg_of_pos) {
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    if (k < n) { idx[k] = k; seg_of_pos[k] = 0u; }
}

// Per-segment state of one level (structure of arrays, S <= n / 2 entries)
struct Level {
    uint32_t *b, *e, *me, *rank0;       // the segment
    uint32_t* acc;                      // [S][12] centroid bounds mn, mx; object-box bounds mn, mx (encoded)
    uint32_t* bk;                       // [S][6][6] bucket boxes (encoded)
    uint32_t* cnt;                      // [S][6]
    uint32_t* hb;                       // [S][12] child boxes of a halved segment (encoded)
    uint32_t *mid, *halved, *cc, *off;  // plan: split position, halved flag, number of child segments, their first index in the next level
    float* lr;                          // [S][12] child boxes (SAH split)
};

__global__ void k_seg_init(uint32_t S, Level L) {
    const uint32_t s = blockIdx.x * BT + threadIdx.x;
    if (s >= S) return;
    for (int k = 0; k < 12; ++k) L.acc[12 * (size_t)s + k] = (k % 6) < 3 ? ENC_PINF : ENC_NINF;
    for (int k = 0; k < 36; ++k) L.bk[36 * (size_t)s + k] = (k % 6) < 3 ? ENC_PINF : ENC_NINF;
    for (int k = 0; k < 6; ++k) L.cnt[6 * (size_t)s + k] = 0u;
    for (int k = 0; k < 12; ++k) L.hb[12 * (size_t)s + k] = (k % 6) < 3 ? ENC_PINF : ENC_NINF;
}

// min / max of `v` (12 encoded values per thread, v[k] with (k % 6) < 3 a minimum) into dst: through LDS when the whole workgroup adds to
// one destination, straight to memory otherwise.
__device__ inline void acc12(bool active, bool block_uniform, uint32_t* dst, const uint32_t* v, uint32_t* s_acc) {
    if (block_uniform) {
        if (threadIdx.x < 12) s_acc[threadIdx.x] = (threadIdx.x % 6) < 3 ? ENC_PINF : ENC_NINF;
        __syncthreads();
        if (active)
            for (int k = 0; k < 12; ++k) {
                if ((k % 6) < 3) atomicMin(&s_acc[k], v[k]);
                else atomicMax(&s_acc[k], v[k]);
            }
        __syncthreads();
        if (threadIdx.x < 12) {
            if ((threadIdx.x % 6) < 3) atomicMin(&dst[threadIdx.x], s_acc[threadIdx.x]);
            else atomicMax(&dst[threadIdx.x], s_acc[threadIdx.x]);
        }
    } else if (active) {
        for (int k = 0; k < 12; ++k) {
            if ((k % 6) < 3) atomicMin(&dst[k], v[k]);
            else atomicMax(&dst[k], v[k]);
        }
    }
}

// does the whole workgroup lie inside one segment?  (segments are ranges of positions, so: first and last position agree)
__device__ inline bool uniform_block(uint32_t n, const uint32_t* seg_of_pos, uint32_t* s_first) {
    const uint32_t k0 = blockIdx.x * BT, k1 = k0 + BT - 1;
    if (k1 >= n) return false;
    const uint32_t a = seg_of_pos[k0], b = seg_of_pos[k1];
    *s_first = a;
    return a == b && a != NONE;
}

__global__ void k_bounds(uint32_t n, const uint32_t* idx, const uint32_t* seg_of_pos, const float* box, const float* ctr, Level L) {
    __shared__ uint32_t s_acc[12];
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    uint32_t s0;
    const bool uni = uniform_block(n, seg_of_pos, &s0);
    const uint32_t s = k < n ? seg_of_pos[k] : NONE;
    const bool active = s != NONE;
    uint32_t v[12];
    if (active) {
        const uint32_t o = idx[k];
        for (int a = 0; a < 3; ++a) {
            const float c = ctr[3 * (size_t)o + a];
            v[a] = enc(c); v[3 + a] = enc(c);
            v[6 + a] = enc(box[6 * (size_t)o + a]); v[9 + a] = enc(box[6 * (size_t)o + 3 + a]);
        }
    }
    acc12(active, uni, L.acc + 12 * (size_t)(uni ? s0 : (active ? s : 0u)), v, s_acc);
}

// the split axis and extent of a segment from its centroid bounds (host_scene.cpp: `axis`, `extent`)
__device__ inline void seg_axis(const uint32_t* acc, int* axis, float* extent, float* cmin) {
    const float ex = dec(acc[3]) - dec(acc[0]), ey = dec(acc[4]) - dec(acc[1]), ez = dec(acc[5]) - dec(acc[2]);
    *axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
    *extent = dec(acc[3 + *axis]) - dec(acc[*axis]);
    *cmin = dec(acc[*axis]);
}

__global__ void k_bucket(uint32_t n, const uint32_t* idx, const uint32_t* seg_of_pos, const float* box, const float* ctr, Level L, uint32_t* keys) {
    __shared__ uint32_t s_bk[36], s_cnt[6];
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    uint32_t s0;
    const bool uni = uniform_block(n, seg_of_pos, &s0);
    const uint32_t s = k < n ? seg_of_pos[k] : NONE;
    const bool active = s != NONE;
    uint32_t bn = 0u, v[6];
    bool counted = false;
    if (active) {
        int axis;
        float extent, cmin;
        seg_axis(L.acc + 12 * (size_t)s, &axis, &extent, &cmin);
        const uint32_t o = idx[k];
        if (!(extent < 0.00001f)) {
            const int NB = 6;
            const float rel = (ctr[3 * (size_t)o + axis] - cmin) / extent;
            int b = (int)(rel * ((float)NB - 0.01f));
            b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
            bn = (uint32_t)b;
            counted = true;
            for (int a = 0; a < 6; ++a) v[a] = enc(box[6 * (size_t)o + a]);
        }
        keys[k] = (L.b[s] << 3) | bn;   // segment start, bucket: sorting these (stably) is the host's counting sort of every segment
    } else if (k < n) {
        keys[k] = k << 3;               // a finished position stays where it is
    }
    if (uni) {
        if (threadIdx.x < 36) s_bk[threadIdx.x] = (threadIdx.x % 6) < 3 ? ENC_PINF : ENC_NINF;
        if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
        if (counted) {
            for (int a = 0; a < 6; ++a) {
                if (a < 3) atomicMin(&s_bk[6 * bn + a], v[a]);
                else atomicMax(&s_bk[6 * bn + a], v[a]);
            }
            atomicAdd(&s_cnt[bn], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 36) {
            if ((threadIdx.x % 6) < 3) atomicMin(&L.bk[36 * (size_t)s0 + threadIdx.x], s_bk[threadIdx.x]);
            else atomicMax(&L.bk[36 * (size_t)s0 + threadIdx.x], s_bk[threadIdx.x]);
        }
        if (threadIdx.x < 6 && s_cnt[threadIdx.x]) atomicAdd(&L.cnt[6 * (size_t)s0 + threadIdx.x], s_cnt[threadIdx.x]);
    } else if (counted) {
        for (int a = 0; a < 6; ++a) {
            if (a < 3) atomicMin(&L.bk[36 * (size_t)s + 6 * bn + a], v[a]);
            else atomicMax(&L.bk[36 * (size_t)s + 6 * bn + a], v[a]);
        }
        atomicAdd(&L.cnt[6 * (size_t)s + bn], 1u);
    }
}

// The crate's choice among the 5 bucket boundaries (cost = (nl * SA(l) + nr * SA(r)) / SA(all), first strictly smallest), or halving.
__global__ void k_plan(uint32_t S, Level L) {
    const uint32_t s = blockIdx.x * BT + threadIdx.x;
    if (s >= S) return;
    const uint32_t b = L.b[s], e = L.e[s];
    const uint32_t* acc = L.acc + 12 * (size_t)s;
    int axis;
    float extent, cmin;
    seg_axis(acc, &axis, &extent, &cmin);
    DBox all;
    for (int a = 0; a < 3; ++a) { all.mn[a] = dec(acc[6 + a]); all.mx[a] = dec(acc[9 + a]); }
    bool halved = extent < 0.00001f;
    uint32_t mid = 0;
    DBox lbox = empty_box(), rbox = empty_box();
    if (!halved) {
        const int NB = 6;
        DBox bb[NB];
        uint32_t cnt[NB];
        for (int i = 0; i < NB; ++i) {
            for (int a = 0; a < 3; ++a) { bb[i].mn[a] = dec(L.bk[36 * (size_t)s + 6 * i + a]); bb[i].mx[a] = dec(L.bk[36 * (size_t)s + 6 * i + 3 + a]); }
            cnt[i] = L.cnt[6 * (size_t)s + i];
        }
        int best = -1;
        float best_cost = __builtin_inff();
        for (int i = 0; i < NB - 1; ++i) {
            DBox l = empty_box(), r = empty_box();
            uint32_t nl = 0, nr = 0;
            for (int j = 0; j <= i; ++j) { join_into(&l, bb[j]); nl += cnt[j]; }
            for (int j = i + 1; j < NB; ++j) { join_into(&r, bb[j]); nr += cnt[j]; }
            const float cost = ((float)nl * area(l) + (float)nr * area(r)) / area(all);
            if (cost < best_cost) { best = i; best_cost = cost; lbox = l; rbox = r; }
        }
        uint32_t start = 0;
        const int cut = best < 0 ? 0 : best;
        for (int i = 0; i <= cut; ++i) start += cnt[i];
        mid = b + start;
        if (best < 0 || mid == b || mid == e) halved = true;   // every cost NaN: degenerate bounds
    }
    if (halved) mid = b + (e - b) / 2;
    L.mid[s] = mid;
    L.halved[s] = halved ? 1u : 0u;
    L.cc[s] = (mid - b > 1u ? 1u : 0u) + (e - mid > 1u ? 1u : 0u);
    for (int a = 0; a < 3; ++a) {
        L.lr[12 * (size_t)s + a] = lbox.mn[a]; L.lr[12 * (size_t)s + 3 + a] = lbox.mx[a];
        L.lr[12 * (size_t)s + 6 + a] = rbox.mn[a]; L.lr[12 * (size_t)s + 9 + a] = rbox.mx[a];
    }
}

// child boxes of the halved segments: joins over the two halves of the (sorted) positions
__global__ void k_halfbox(uint32_t n, const uint32_t* idx, const uint32_t* seg_of_pos, const float* box, Level L) {
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = seg_of_pos[k];
    if (s == NONE || !L.halved[s]) return;
    const uint32_t o = idx[k];
    uint32_t* dst = L.hb + 12 * (size_t)s + (k < L.mid[s] ? 0 : 6);
    for (int a = 0; a < 3; ++a) {
        atomicMin(&dst[a], enc(box[6 * (size_t)o + a]));
        atomicMax(&dst[3 + a], enc(box[6 * (size_t)o + 3 + a]));
    }
}

// node words, ranks of leaf children, the next level's segments (in position order: L.off is the exclusive scan of L.cc)
__global__ void k_emit(uint32_t S, Level L, Level N, const uint32_t* idx, uint32_t* words, uint32_t* rank, uint32_t* left_id, uint32_t* right_id) {
    const uint32_t s = blockIdx.x * BT + threadIdx.x;
    if (s >= S) return;
    const uint32_t b = L.b[s], e = L.e[s], me = L.me[s], rank0 = L.rank0[s], mid = L.mid[s];
    uint32_t* w = words + 16 * (size_t)me;
    if (L.halved[s]) {
        for (int a = 0; a < 12; ++a) { const float f = dec(L.hb[12 * (size_t)s + a]); memcpy(&w[a], &f, 4); }
    } else {
        for (int a = 0; a < 12; ++a) memcpy(&w[a], &L.lr[12 * (size_t)s + a], 4);
    }
    // pre-order: the left subtree's mid - b - 1 inner nodes follow this one; right-first DFS: the right subtree's e - mid leaves rank first
    const uint32_t left_me = me + 1u, right_me = me + (mid - b), left_rank = rank0 + (e - mid);
    uint32_t next = L.off[s], lid = NONE, rid = NONE, left, right;
    if (mid - b == 1u) {
        const uint32_t o = idx[b];
        rank[o] = left_rank;
        left = LEAF_BIT | o;
    } else {
        left = left_me;
        lid = next++;
        N.b[lid] = b; N.e[lid] = mid; N.me[lid] = left_me; N.rank0[lid] = left_rank;
    }
    if (e - mid == 1u) {
        const uint32_t o = idx[mid];
        rank[o] = rank0;
        right = LEAF_BIT | o;
    } else {
        right = right_me;
        rid = next++;
        N.b[rid] = mid; N.e[rid] = e; N.me[rid] = right_me; N.rank0[rid] = rank0;
    }
    w[12] = left; w[13] = right; w[14] = 0u; w[15] = 0u;
    left_id[s] = lid; right_id[s] = rid;
}
__global__ void k_assign(uint32_t n, const uint32_t* seg_of_pos, uint32_t* seg_next, Level L, const uint32_t* left_id, const uint32_t* right_id) {
    const uint32_t k = blockIdx.x * BT + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = seg_of_pos[k];
    seg_next[k] = s == NONE ? NONE : (k < L.mid[s] ? left_id[s] : right_id[s]);
}

template <class T>
struct Buf {
    T* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};
uint32_t grid(uint64_t n) { return (uint32_t)((n + BT - 1) / BT); }

}  // namespace

bool device_build_bvh(const HostScene& sc, FlatBVH* out, double* ms, std::string* err) {
    const uint32_t T = sc.n_tri(), S = sc.n_sph(), n = T + S;
    if (n < 2) { *err = "fewer than two objects"; return false; }
    if (n >= (1u << 28)) { *err = "more than 2^28 objects"; return false; }
    hipEvent_t ev0, ev1;
    HIPB(hipEventCreate(&ev0));
    HIPB(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ev0, ev1};
    Buf<float> d_tri, d_sph, d_box, d_ctr, d_lr[2];
    Buf<uint32_t> d_bad, d_idx[2], d_seg[2], d_keys[2], d_words, d_rank, d_left, d_right, d_nseg;
    Buf<uint32_t> d_b[2], d_e[2], d_me[2], d_r0[2], d_acc, d_bk, d_cnt, d_hb, d_mid, d_halved, d_cc, d_off;
    Buf<unsigned char> d_tmp;
    const size_t maxS = n / 2 + 1;
    HIPB(d_tri.alloc(9 * (size_t)T)); HIPB(d_sph.alloc(4 * (size_t)S)); HIPB(d_box.alloc(6 * (size_t)n)); HIPB(d_ctr.alloc(3 * (size_t)n));
    HIPB(d_bad.alloc(1)); HIPB(d_words.alloc(16 * (size_t)(n - 1))); HIPB(d_rank.alloc(n)); HIPB(d_left.alloc(maxS)); HIPB(d_right.alloc(maxS));
    for (int q = 0; q < 2; ++q) {
        HIPB(d_idx[q].alloc(n)); HIPB(d_seg[q].alloc(n)); HIPB(d_keys[q].alloc(n));
        HIPB(d_b[q].alloc(maxS)); HIPB(d_e[q].alloc(maxS)); HIPB(d_me[q].alloc(maxS)); HIPB(d_r0[q].alloc(maxS));
    }
    HIPB(d_acc.alloc(12 * maxS)); HIPB(d_bk.alloc(36 * maxS)); HIPB(d_cnt.alloc(6 * maxS)); HIPB(d_hb.alloc(12 * maxS)); HIPB(d_lr[0].alloc(12 * maxS));
    HIPB(d_mid.alloc(maxS)); HIPB(d_halved.alloc(maxS)); HIPB(d_cc.alloc(maxS)); HIPB(d_off.alloc(maxS + 1));
    size_t tmp_sort = 0, tmp_scan = 0;
    HIPB(rocprim::radix_sort_pairs(nullptr, tmp_sort, d_keys[0].p, d_keys[1].p, d_idx[0].p, d_idx[1].p, n, 0, 32, 0));
    HIPB(rocprim::exclusive_scan(nullptr, tmp_scan, d_cc.p, d_off.p, 0u, maxS + 1, rocprim::plus<uint32_t>(), 0));
    const size_t tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    HIPB(d_tmp.alloc(tmp_bytes));

    HIPB(hipEventRecord(ev0, 0));
    if (T) HIPB(hipMemcpyAsync(d_tri.p, sc.tri_pos.data(), 9 * (size_t)T * 4, hipMemcpyHostToDevice, 0));
    if (S) HIPB(hipMemcpyAsync(d_sph.p, sc.sph.data(), 4 * (size_t)S * 4, hipMemcpyHostToDevice, 0));
    HIPB(hipMemsetAsync(d_bad.p, 0, 4, 0));
    hipLaunchKernelGGL(k_obj_boxes, dim3(grid(n)), dim3(BT), 0, 0, d_tri.p, d_sph.p, T, S, d_box.p, d_ctr.p, d_bad.p);
    hipLaunchKernelGGL(k_init_perm, dim3(grid(n)), dim3(BT), 0, 0, n, d_idx[0].p, d_seg[0].p);
    uint32_t bad = 0;
    HIPB(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) { *err = "a coordinate is not finite or is a negative zero: built on the host"; return false; }
    const uint32_t first[4] = {0u, n, 0u, 0u};
    HIPB(hipMemcpy(d_b[0].p, &first[0], 4, hipMemcpyHostToDevice)); HIPB(hipMemcpy(d_e[0].p, &first[1], 4, hipMemcpyHostToDevice));
    HIPB(hipMemcpy(d_me[0].p, &first[2], 4, hipMemcpyHostToDevice)); HIPB(hipMemcpy(d_r0[0].p, &first[3], 4, hipMemcpyHostToDevice));

    uint32_t nseg = 1, depth = 0;
    int cur = 0, pi = 0;   // segment arrays / permutation buffers in use
    // bits of the sort key: (position << 3 | bucket) < 8 n
    unsigned key_bits = 4;
    while (key_bits < 32 && (1ull << key_bits) < 8ull * n) ++key_bits;
    while (nseg) {
        ++depth;
        if (depth > 4096) { *err = "tree deeper than 4096 levels"; return false; }
        Level L, N;
        L.b = d_b[cur].p; L.e = d_e[cur].p; L.me = d_me[cur].p; L.rank0 = d_r0[cur].p;
        N.b = d_b[cur ^ 1].p; N.e = d_e[cur ^ 1].p; N.me = d_me[cur ^ 1].p; N.rank0 = d_r0[cur ^ 1].p;
        L.acc = d_acc.p; L.bk = d_bk.p; L.cnt = d_cnt.p; L.hb = d_hb.p; L.mid = d_mid.p; L.halved = d_halved.p; L.cc = d_cc.p; L.off = d_off.p; L.lr = d_lr[0].p;
        N.acc = N.bk = N.cnt = N.hb = N.mid = N.halved = N.cc = N.off = nullptr; N.lr = nullptr;
        hipLaunchKernelGGL(k_seg_init, dim3(grid(nseg)), dim3(BT), 0, 0, nseg, L);
        hipLaunchKernelGGL(k_bounds, dim3(grid(n)), dim3(BT), 0, 0, n, d_idx[pi].p, d_seg[pi].p, d_box.p, d_ctr.p, L);
        hipLaunchKernelGGL(k_bucket, dim3(grid(n)), dim3(BT), 0, 0, n, d_idx[pi].p, d_seg[pi].p, d_box.p, d_ctr.p, L, d_keys[0].p);
        size_t tb = tmp_bytes;
        HIPB(rocprim::radix_sort_pairs(d_tmp.p, tb, d_keys[0].p, d_keys[1].p, d_idx[pi].p, d_idx[pi ^ 1].p, n, 0, key_bits, 0));
        hipLaunchKernelGGL(k_plan, dim3(grid(nseg)), dim3(BT), 0, 0, nseg, L);
        hipLaunchKernelGGL(k_halfbox, dim3(grid(n)), dim3(BT), 0, 0, n, d_idx[pi ^ 1].p, d_seg[pi].p, d_box.p, L);
        tb = tmp_bytes;
        HIPB(rocprim::exclusive_scan(d_tmp.p, tb, d_cc.p, d_off.p, 0u, (size_t)nseg + 1, rocprim::plus<uint32_t>(), 0));   // off[nseg] = size of the next level
        hipLaunchKernelGGL(k_emit, dim3(grid(nseg)), dim3(BT), 0, 0, nseg, L, N, d_idx[pi ^ 1].p, d_words.p, d_rank.p, d_left.p, d_right.p);
        hipLaunchKernelGGL(k_assign, dim3(grid(n)), dim3(BT), 0, 0, n, d_seg[pi].p, d_seg[pi ^ 1].p, L, d_left.p, d_right.p);
        HIPB(hipGetLastError());
        uint32_t next = 0;
        HIPB(hipMemcpy(&next, d_off.p + nseg, 4, hipMemcpyDeviceToHost));
        if (next > maxS) { *err = "segment count out of range"; return false; }
        nseg = next;
        cur ^= 1;
        pi ^= 1;
    }
    out->words.resize(16 * (size_t)(n - 1));
    out->rank.resize(n);
    HIPB(hipMemcpyAsync(out->words.data(), d_words.p, out->words.size() * 4, hipMemcpyDeviceToHost, 0));
    HIPB(hipMemcpyAsync(out->rank.data(), d_rank.p, (size_t)n * 4, hipMemcpyDeviceToHost, 0));
    HIPB(hipEventRecord(ev1, 0));
    HIPB(hipEventSynchronize(ev1));
    float t = 0.0f;
    HIPB(hipEventElapsedTime(&t, ev0, ev1));
    if (ms) *ms = t;
    out->root = 0u;
    out->depth = depth;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 8-wide occlusion tree (host_scene.cpp WideBuilder), one breadth-first level per iteration:
//   k_wide_gather  per node: open the inner child with the largest box until 8 slots would overflow; inner children first, then primitives
//   exclusive scans of the inner-child and primitive counts: first child / first record of every node, in node order
//   k_wide_write   next level's references, the primitive records, and the node with its planes quantised by wide_axis_planes<double>
//                  (host_scene.h; the host runs the same function in long double): whenever double had to round, the build is declined
namespace {

struct WideIn {
    const uint32_t* words;        // binary tree, 16 words per inner node
    const uint8_t* inner_block;   // [inner node] the subtree holds a non-emissive object
    const uint8_t* obj_block;     // [object] non-emissive
    const float* tri_pos;
    const float* sph;
    uint32_t T;
};
__device__ inline bool ref_blocks(const WideIn& in, uint32_t ref) { return (ref & LEAF_BIT) ? in.obj_block[ref & OBJECT_MASK] != 0 : in.inner_block[ref] != 0; }
__device__ inline int children_of(const WideIn& in, uint32_t inner, uint32_t* ref, float* box) {   // box: 6 floats per child
    const uint32_t* w = in.words + 16 * (size_t)inner;
    int n = 0;
    for (int c = 0; c < 2; ++c) {
        const uint32_t r = w[12 + c];
        if (!ref_blocks(in, r)) continue;
        ref[n] = r;
        for (int a = 0; a < 6; ++a) memcpy(&box[6 * n + a], &w[6 * c + a], 4);
        ++n;
    }
    return n;
}
__device__ inline float area6(const float* b) {
    const float sx = b[3] - b[0], sy = b[4] - b[1], sz = b[5] - b[2];
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}

__global__ void k_wide_gather(uint32_t m, WideIn in, const uint32_t* refs, uint32_t* g_ref, float* g_box, uint32_t* g_inner, uint32_t* g_prims) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= m) return;
    uint32_t ref[8], sref[2];
    float box[48], sbox[12];
    int n = children_of(in, refs[i], ref, box);
    for (;;) {
        int best = -1;
        float best_area = -1.0f;
        for (int k = 0; k < n; ++k) {
            if (ref[k] & LEAF_BIT) continue;
            const float a = area6(box + 6 * k);
            if (best < 0 || a > best_area) { best = k; best_area = a; }
        }
        if (best < 0) break;
        const int ns = children_of(in, ref[best], sref, sbox);
        if (n - 1 + ns > 8) break;
        for (int k = best; k + 1 < n; ++k) {          // erase `best`, append its children at the end
            ref[k] = ref[k + 1];
            for (int a = 0; a < 6; ++a) box[6 * k + a] = box[6 * (k + 1) + a];
        }
        --n;
        for (int k = 0; k < ns; ++k) {
            ref[n] = sref[k];
            for (int a = 0; a < 6; ++a) box[6 * n + a] = sbox[6 * k + a];
            ++n;
        }
    }
    // inner children first, primitives after, each in the order gathered
    uint32_t n_inner = 0, at = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < n; ++k) {
            const bool leaf = (ref[k] & LEAF_BIT) != 0u;
            if (leaf != (pass == 1)) continue;
            g_ref[8 * (size_t)i + at] = ref[k];
            for (int a = 0; a < 6; ++a) g_box[48 * (size_t)i + 6 * at + a] = box[6 * k + a];
            ++at;
            if (!leaf) ++n_inner;
        }
    g_inner[i] = n_inner;
    g_prims[i] = (uint32_t)n - n_inner;
}

__global__ void k_wide_write(uint32_t m, WideIn in, uint32_t level_base, uint32_t next_base, uint32_t prims_before, const uint32_t* g_ref, const float* g_box,
                             const uint32_t* g_inner, const uint32_t* g_prims, const uint32_t* inner_off, const uint32_t* prim_off, uint32_t* next_refs,
                             uint32_t* nodes, float* prims, uint32_t* rec_of_obj, uint32_t* flags) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= m) return;
    const uint32_t n_inner = g_inner[i], n_ch = n_inner + g_prims[i];
    const uint32_t child_base = next_base + inner_off[i], prim_base = prims_before + prim_off[i];
    const uint32_t* ref = g_ref + 8 * (size_t)i;
    const float* box = g_box + 48 * (size_t)i;
    for (uint32_t k = 0; k < n_inner; ++k) next_refs[inner_off[i] + k] = ref[k];
    for (uint32_t k = n_inner; k < n_ch; ++k) {
        const uint32_t obj = ref[k] & OBJECT_MASK, rec = prim_base + (k - n_inner);
        rec_of_obj[obj] = rec;
        float* r = prims + 12 * (size_t)rec;
        for (int a = 0; a < 12; ++a) r[a] = 0.0f;
        uint32_t w;
        if (obj < in.T) {
            const float* p = in.tri_pos + 9 * (size_t)obj;
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
            r[4] = p[3]; r[5] = p[4]; r[6] = p[5]; w = obj; memcpy(&r[7], &w, 4);
            r[8] = p[6]; r[9] = p[7]; r[10] = p[8];
        } else {
            const float* c = in.sph + 4 * (size_t)(obj - in.T);
            r[0] = c[0]; r[1] = c[1]; r[2] = c[2]; w = 1u; memcpy(&r[3], &w, 4);
            r[4] = c[3]; w = obj; memcpy(&r[5], &w, 4);
        }
    }
    // the node: box of its children, per-axis grid, quantised planes
    float nmn[3], nmx[3];
    for (int a = 0; a < 3; ++a) { nmn[a] = __builtin_inff(); nmx[a] = -__builtin_inff(); }
    for (uint32_t k = 0; k < n_ch; ++k)
        for (int a = 0; a < 3; ++a) { nmn[a] = lo(nmn[a], box[6 * k + a]); nmx[a] = hi(nmx[a], box[6 * k + 3 + a]); }
    uint32_t* w = nodes + 20 * (size_t)(level_base + i);
    unsigned char* q = reinterpret_cast<unsigned char*>(w + 8);
    uint32_t meta = 0;
    bool ok = true, inexact = false;
    for (int a = 0; a < 3; ++a) {
        unsigned char ql[8], qh[8];
        int e = 0;
        ok = ok && wide_axis_planes<double>(nmn[a], nmx[a], (int)n_ch, box + a, box + 3 + a, 6, ql, qh, &e, &inexact);
        for (uint32_t k = 0; k < 8; ++k) { q[8 * a + k] = k < n_ch ? ql[k] : 255; q[24 + 8 * a + k] = k < n_ch ? qh[k] : 0; }
        meta |= (uint32_t)(e + WIDE_EXP_BIAS) << (8 * a);
    }
    meta |= n_inner << 24;
    meta |= n_ch << 28;
    for (int a = 0; a < 3; ++a) memcpy(&w[a], &nmn[a], 4);
    w[3] = meta;
    w[4] = child_base; w[5] = prim_base; w[6] = 0u; w[7] = 0u;
    if (!ok) flags[0] = 1u;        // outside the quantiser's range: the host decides what that means (no wide tree)
    if (inexact) flags[1] = 1u;    // double had to round where long double might not: declined
}

}  // namespace

bool device_build_wide_bvh(const HostScene& sc, const FlatBVH& bvh, WideBVH* out, double* ms, std::string* err) {
    const uint32_t T = sc.n_tri(), S = sc.n_sph(), n = T + S, ni = bvh.n_inner();
    if (n < 2 || ni != n - 1 || (bvh.root & LEAF_BIT)) { *err = "fewer than two objects"; return false; }
    // what the host builder decides before it starts (host_scene.cpp WideBuilder): which subtrees hold a blocker, and the range check
    std::vector<uint8_t> obj_block(n), inner_block(ni);
    for (uint32_t i = 0; i < T; ++i) obj_block[i] = scene_is_black(sc.mats[sc.tri_mat[i]].emittance);
    for (uint32_t i = 0; i < S; ++i) obj_block[T + i] = scene_is_black(sc.mats[sc.sph_mat[i]].emittance);
    for (uint32_t i = ni; i-- > 0;) {
        const uint32_t* w = &bvh.words[16 * (size_t)i];
        uint8_t f = 0;
        for (int c = 12; c <= 13; ++c) f |= (w[c] & LEAF_BIT) ? obj_block[w[c] & OBJECT_MASK] : inner_block[w[c]];
        inner_block[i] = f;
    }
    out->nodes.clear(); out->prims.clear(); out->depth = 0; out->ok = false;
    out->rec_of_obj.assign(n, 0xFFFFFFFFu);
    if (!inner_block[0]) return true;                       // nothing can block: no tree, as on the host
    for (size_t i = 0; i < sc.tri_pos.size(); ++i)
        if (!(fabsf(sc.tri_pos[i]) <= WIDE_COORD_MAX)) return true;
    for (size_t i = 0; i < sc.sph.size(); ++i)
        if (!(fabsf(sc.sph[i]) <= WIDE_COORD_MAX / 2.0f)) return true;

    hipEvent_t ev0, ev1;
    HIPB(hipEventCreate(&ev0));
    HIPB(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ev0, ev1};
    Buf<uint32_t> d_words, d_refs[2], d_gref, d_ginner, d_gprims, d_ioff, d_poff, d_nodes, d_rec, d_flags;
    Buf<uint8_t> d_iblock, d_oblock;
    Buf<float> d_tri, d_sph, d_gbox, d_prims;
    Buf<unsigned char> d_tmp;
    HIPB(d_words.alloc(16 * (size_t)ni)); HIPB(d_iblock.alloc(ni)); HIPB(d_oblock.alloc(n)); HIPB(d_tri.alloc(9 * (size_t)T)); HIPB(d_sph.alloc(4 * (size_t)S));
    for (int q = 0; q < 2; ++q) HIPB(d_refs[q].alloc(ni));
    HIPB(d_gref.alloc(8 * (size_t)ni)); HIPB(d_gbox.alloc(48 * (size_t)ni)); HIPB(d_ginner.alloc((size_t)ni + 1)); HIPB(d_gprims.alloc((size_t)ni + 1));
    HIPB(d_ioff.alloc((size_t)ni + 1)); HIPB(d_poff.alloc((size_t)ni + 1));
    HIPB(d_nodes.alloc(20 * (size_t)ni)); HIPB(d_prims.alloc(12 * (size_t)n)); HIPB(d_rec.alloc(n)); HIPB(d_flags.alloc(2));
    size_t tmp_bytes = 0;
    HIPB(rocprim::exclusive_scan(nullptr, tmp_bytes, d_ginner.p, d_ioff.p, 0u, (size_t)ni + 1, rocprim::plus<uint32_t>(), 0));
    HIPB(d_tmp.alloc(tmp_bytes));

    HIPB(hipEventRecord(ev0, 0));
    HIPB(hipMemcpyAsync(d_words.p, bvh.words.data(), 16 * (size_t)ni * 4, hipMemcpyHostToDevice, 0));
    HIPB(hipMemcpyAsync(d_iblock.p, inner_block.data(), ni, hipMemcpyHostToDevice, 0));
    HIPB(hipMemcpyAsync(d_oblock.p, obj_block.data(), n, hipMemcpyHostToDevice, 0));
    if (T) HIPB(hipMemcpyAsync(d_tri.p, sc.tri_pos.data(), 9 * (size_t)T * 4, hipMemcpyHostToDevice, 0));
    if (S) HIPB(hipMemcpyAsync(d_sph.p, sc.sph.data(), 4 * (size_t)S * 4, hipMemcpyHostToDevice, 0));
    HIPB(hipMemsetAsync(d_rec.p, 0xFF, (size_t)n * 4, 0));
    HIPB(hipMemsetAsync(d_flags.p, 0, 8, 0));
    const uint32_t root = 0u;
    HIPB(hipMemcpyAsync(d_refs[0].p, &root, 4, hipMemcpyHostToDevice, 0));
    WideIn in;
    in.words = d_words.p; in.inner_block = d_iblock.p; in.obj_block = d_oblock.p; in.tri_pos = d_tri.p; in.sph = d_sph.p; in.T = T;

    uint32_t m = 1, level_base = 0, n_prims = 0, depth = 0;
    int cur = 0;
    while (m) {
        ++depth;
        hipLaunchKernelGGL(k_wide_gather, dim3(grid(m)), dim3(BT), 0, 0, m, in, d_refs[cur].p, d_gref.p, d_gbox.p, d_ginner.p, d_gprims.p);
        size_t tb = tmp_bytes;
        HIPB(rocprim::exclusive_scan(d_tmp.p, tb, d_ginner.p, d_ioff.p, 0u, (size_t)m + 1, rocprim::plus<uint32_t>(), 0));
        tb = tmp_bytes;
        HIPB(rocprim::exclusive_scan(d_tmp.p, tb, d_gprims.p, d_poff.p, 0u, (size_t)m + 1, rocprim::plus<uint32_t>(), 0));
        uint32_t tot[2];
        HIPB(hipMemcpy(&tot[0], d_ioff.p + m, 4, hipMemcpyDeviceToHost));
        HIPB(hipMemcpy(&tot[1], d_poff.p + m, 4, hipMemcpyDeviceToHost));
        const uint64_t next_base = (uint64_t)level_base + m, nn = next_base + tot[0], np = (uint64_t)n_prims + tot[1];
        if (nn > ni || np > n) { *err = "8-wide tree larger than the binary one"; return false; }
        if (nn >= (1u << 23) || np + 8 >= (1u << 23)) {     // index fields of the stack entries (rt_kernels.h): no wide tree, as on the host
            out->rec_of_obj.assign(n, 0xFFFFFFFFu);
            return true;
        }
        hipLaunchKernelGGL(k_wide_write, dim3(grid(m)), dim3(BT), 0, 0, m, in, level_base, (uint32_t)next_base, n_prims, d_gref.p, d_gbox.p, d_ginner.p, d_gprims.p,
                           d_ioff.p, d_poff.p, d_refs[cur ^ 1].p, d_nodes.p, d_prims.p, d_rec.p, d_flags.p);
        HIPB(hipGetLastError());
        level_base = (uint32_t)next_base;
        n_prims = (uint32_t)np;
        m = tot[0];
        cur ^= 1;
    }
    uint32_t flags[2];
    HIPB(hipMemcpy(flags, d_flags.p, 8, hipMemcpyDeviceToHost));
    if (flags[1]) { *err = "a plane of the 8-wide tree would have been rounded in double precision: built on the host"; return false; }
    if (flags[0]) {                                         // outside the quantiser's range
        out->rec_of_obj.assign(n, 0xFFFFFFFFu);
        return true;
    }
    out->nodes.resize(20 * (size_t)level_base);
    out->prims.resize(12 * (size_t)n_prims);
    HIPB(hipMemcpyAsync(out->nodes.data(), d_nodes.p, out->nodes.size() * 4, hipMemcpyDeviceToHost, 0));
    HIPB(hipMemcpyAsync(out->prims.data(), d_prims.p, out->prims.size() * 4, hipMemcpyDeviceToHost, 0));
    HIPB(hipMemcpyAsync(out->rec_of_obj.data(), d_rec.p, (size_t)n * 4, hipMemcpyDeviceToHost, 0));
    HIPB(hipEventRecord(ev1, 0));
    HIPB(hipEventSynchronize(ev1));
    float t = 0.0f;
    HIPB(hipEventElapsedTime(&t, ev0, ev1));
    if (ms) *ms = t;
    out->depth = depth;
    out->ok = true;
    return true;
}

}  // namespace rt
