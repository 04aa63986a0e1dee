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

bool scene_is_black(const float* rgb);   // Spectrum::is_black, src/common.rs:40-43
void build_bvh(const HostScene& sc, FlatBVH* out);
bool make_named_scene(const char* name, const char* obj_path, const float* vpos, const float* vnrm, uint32_t n_vert,
                      const uint32_t* faces, uint32_t n_face, HostScene* out, std::string* err);
void make_dragon_standin(uint32_t nu, uint32_t nv, float* vpos, float* vnrm, uint32_t* faces);

}  // namespace rt
