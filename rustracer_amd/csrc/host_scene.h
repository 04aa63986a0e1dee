// host_scene.h — host side of the scene boundary: the object list Scene::new receives
// (src/scene/mod.rs:82-106), the reference's scene constructors, and the BVH flattened to the
// device layout.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rustracer_hip.h"

namespace rt {

struct HostScene {
    std::vector<float> tri_pos, tri_nrm;   // 9 floats per triangle
    std::vector<uint32_t> tri_mat;
    std::vector<float> sph;                // 4 floats per sphere
    std::vector<uint32_t> sph_mat;
    std::vector<rt_material> mats;
    uint32_t n_tri() const { return (uint32_t)tri_mat.size(); }
    uint32_t n_sph() const { return (uint32_t)sph_mat.size(); }
};

const uint32_t LEAF_BIT = 0x80000000u;
const uint32_t NONBLOCK_BIT = 0x40000000u;   // device copy only: leaf of an emissive object (never blocks a shadow ray, mod.rs:481)
const uint32_t OBJECT_MASK = 0x3FFFFFFFu;

// Device BVH: inner nodes only, 16 words (64 B) each:
//   {l_min xyz, l_max xyz, r_min xyz, r_max xyz, left, right, 0, 0}
// A child word with LEAF_BIT set names an object (index into triangles-then-spheres); otherwise
// it is an inner-node index.  root is such a word too (a one-object scene is a bare leaf).
struct FlatBVH {
    std::vector<uint32_t> words;
    std::vector<uint32_t> rank;   // rank[object]: order of its leaf in the reference traversal (mod.rs:427-432: right child popped first)
    uint32_t root;
    uint32_t depth;               // inner nodes on the longest root-to-leaf chain
    uint32_t n_inner() const { return (uint32_t)(words.size() / 16); }
};

// Occlusion BVH for Scene::is_occluded rays (src/scene/mod.rs:452-491): the binary tree above collapsed
// to 8 children per node, non-emissive objects only (emissive ones never block, mod.rs:481), child
// boxes quantised to 8 bits per plane on a per-node grid (origin p, per-axis power-of-two step).
// A quantised box always contains the exact one, so the wide traversal only selects candidates; every
// candidate then takes the exact leaf-box test and the exact primitive test of the reference.
// Node, 20 words (80 B):
//   w0-2  p (grid origin = node box minimum)        w3  ex | ey<<8 | ez<<16 (step = 2^(e-128)) | n_inner<<24 | n_child<<28
//   w4    index of the first inner child (inner children are contiguous, slots 0..n_inner-1)
//   w5    index of the first primitive record (primitive children are contiguous, slots n_inner..n_child-1)
//   w8-19 lo_x[8] lo_y[8] lo_z[8] hi_x[8] hi_y[8] hi_z[8], one byte per child slot
// Primitive record, 12 floats (48 B): triangle (p1, 0) (p2, object) (p3, 0); sphere (centre, 1) (radius, object, 0, 0) (0).
struct WideBVH {
    std::vector<uint32_t> nodes;
    std::vector<float> prims;
    std::vector<uint32_t> rec_of_obj;   // primitive record of each object, 0xFFFFFFFF for the emissive ones (not in the tree)
    uint32_t depth = 0;     // nodes on the longest root-to-primitive chain
    bool ok = false;        // false: no blockers, or the scene is outside the quantiser's range -> binary traversal only
    uint32_t n_nodes() const { return (uint32_t)(nodes.size() / 20); }
    uint32_t n_prims() const { return (uint32_t)(prims.size() / 12); }
};
const int WIDE_EXP_BIAS = 128, WIDE_EXP_MIN = -64, WIDE_EXP_MAX = 60;
#define WIDE_EXP_MIN_ (-64)
#define WIDE_EXP_MAX_ 60
const float WIDE_COORD_MAX = 1.152921504606846976e18f;   // 2^60: larger scene or ray-origin coordinates take the binary traversal

// Quantised planes of one axis of an 8-wide node (shared by the host builder, in long double, and the device builder, in double with a
// check that nothing was rounded): the grid origin is the node box minimum p, the step 2^e is the smallest power of two with
// extent / 253 < 2^e (at most 253 steps, leaving room for the safety step), and every child plane moves outward to a grid plane that is
// <= the exact minimum (>= the exact maximum) in real arithmetic with 2^-20 of a step to spare.  Returns false when the node is outside
// the quantiser's range.  *inexact is set when an operation of `real` had to round (never in long double for scenes the range checks
// admit and coordinates that differ by less than 2^40 in magnitude).
#if defined(__HIP__)
#define RT_QHD __host__ __device__ inline
#else
#define RT_QHD inline
#endif
template <class real>
RT_QHD real q_sub(real a, real b, bool* inexact) {   // a - b, noting whether it was exact (TwoSum)
    const real s = a - b;
    const real bp = s - a;
    const real ap = s - bp;
    const real err = (a - ap) + (-b - bp);
    if (err != (real)0) *inexact = true;
    return s;
}
template <class real>
RT_QHD bool wide_axis_planes(float nb_mn, float nb_mx, int n_ch, const float* ch_mn, const float* ch_mx, int stride, unsigned char* qlo, unsigned char* qhi, int* e_out,
                             bool* inexact) {
    const real p = (real)nb_mn;
    const real ext = q_sub<real>((real)nb_mx, p, inexact);
    if (!(ext >= (real)0) || !(ext < (real)1e300)) return false;
    int e = WIDE_EXP_MIN_;
    if (ext > (real)0) {
        // ext = m 2^k with m in [1/2, 1): ext / 253 < 2^e first holds at e = k - 8 when m < 253/256, at e = k - 7 otherwise (exact comparisons)
        int k = 0;
        real m = ext;
        while (m >= (real)1) { m = m * (real)0.5; ++k; }
        while (m < (real)0.5) { m = m * (real)2; --k; }
        e = k - 8 + (m >= (real)(253.0 / 256.0) ? 1 : 0);
    }
    if (e < WIDE_EXP_MIN_) e = WIDE_EXP_MIN_;
    if (e > WIDE_EXP_MAX_) return false;
    real step = (real)1;
    for (int i = 0; i < (e < 0 ? -e : e); ++i) step = e < 0 ? step * (real)0.5 : step * (real)2;
    const real spare = step * (real)(1.0 / 1048576.0);
    for (int c = 0; c < n_ch; ++c) {
        const real dmn = q_sub<real>((real)ch_mn[c * stride], p, inexact), dmx = q_sub<real>((real)ch_mx[c * stride], p, inexact);
        real ql = dmn / step;   // exact: a power of two
        ql = ql - (ql - (real)(long long)ql < (real)0 ? (real)1 : (real)0);
        ql = (real)(long long)ql;                                   // floor for |ql| < 2^62
        if (ql > (real)255) return false;
        if (ql > (real)0 && dmn - ql * step < spare) ql -= (real)1;
        real qh = dmx / step;
        { const real t = (real)(long long)qh; qh = t < qh ? t + (real)1 : t; }   // ceil
        if (qh * step - dmx < spare) qh += (real)1;
        if (ql < (real)0) ql = (real)0;
        if (qh > (real)255 || !(ql * step <= dmn) || !(qh * step >= dmx)) return false;
        qlo[c] = (unsigned char)ql;
        qhi[c] = (unsigned char)qh;
    }
    *e_out = e;
    return true;
}

bool scene_is_black(const float* rgb);   // Spectrum::is_black, src/common.rs:40-43
void build_bvh(const HostScene& sc, FlatBVH* out);
void build_wide_bvh(const HostScene& sc, const FlatBVH& bvh, WideBVH* out);
bool make_named_scene(const char* name, const char* obj_path, const float* vpos, const float* vnrm, uint32_t n_vert,
                      const uint32_t* faces, uint32_t n_face, HostScene* out, std::string* err);
void make_dragon_standin(uint32_t nu, uint32_t nv, float* vpos, float* vnrm, uint32_t* faces);

}  // namespace rt
