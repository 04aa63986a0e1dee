// host_scene.cpp — scene constructors, OBJ reader, dragon-class stand-in mesh and the host BVH
// builder that feeds the device flattener.  Citations are path:line under the reference tree.
#include "host_scene.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <atomic>
#include <limits>
#include <thread>

#include "rt_math.h"

namespace rt {

bool scene_is_black(const float* rgb) {
    const float c = rgb[0] + rgb[1] + rgb[2];
    return c <= 0.0000001f;
}

namespace {

// ---------------------------------------------------------------------------------------------
// Object list assembly (Triangle::new / new_without_vn / Sphere::new, src/scene/objects.rs:214-267).
class SceneBuilder {
  public:
    explicit SceneBuilder(HostScene* s) : s_(s) {}

    uint32_t material(uint32_t bsdf, float r, float g, float b, float er, float eg, float eb) {
        rt_material m;
        memset(&m, 0, sizeof m);
        m.bsdf = bsdf;
        m.reflectance[0] = r; m.reflectance[1] = g; m.reflectance[2] = b;
        m.emittance[0] = er; m.emittance[1] = eg; m.emittance[2] = eb;
        pending_.push_back(m);
        return (uint32_t)(pending_.size() - 1);   // builder-local id, resolved on first use
    }
    void triangle(vec3 p1, vec3 p2, vec3 p3, vec3 n1, vec3 n2, vec3 n3, uint32_t mat) {
        const vec3 p[3] = {p1, p2, p3};
        const vec3 n[3] = {unit3(n1), unit3(n2), unit3(n3)};   // objects.rs:249
        for (int k = 0; k < 3; ++k) {
            s_->tri_pos.push_back(p[k].x); s_->tri_pos.push_back(p[k].y); s_->tri_pos.push_back(p[k].z);
            s_->tri_nrm.push_back(n[k].x); s_->tri_nrm.push_back(n[k].y); s_->tri_nrm.push_back(n[k].z);
        }
        s_->tri_mat.push_back(resolve(mat));
    }
    void flat_triangle(vec3 p1, vec3 p2, vec3 p3, uint32_t mat) {   // objects.rs:264-267
        const vec3 n = cross3(sub(p2, p1), sub(p3, p1));
        triangle(p1, p2, p3, n, n, n, mat);
    }
    void sphere(vec3 c, float radius, uint32_t mat) {
        s_->sph.push_back(c.x); s_->sph.push_back(c.y); s_->sph.push_back(c.z); s_->sph.push_back(radius);
        s_->sph_mat.push_back(resolve(mat));
    }

  private:
    uint32_t resolve(uint32_t local) {
        const rt_material& m = pending_[local];
        for (size_t i = 0; i < s_->mats.size(); ++i)
            if (memcmp(&s_->mats[i], &m, sizeof m) == 0) return (uint32_t)i;
        s_->mats.push_back(m);
        return (uint32_t)(s_->mats.size() - 1);
    }
    HostScene* s_;
    std::vector<rt_material> pending_;
};

// Scene::load_obj (src/scene/mod.rs:135-189) over tobj::load_obj(path, triangulate = true):
// first model only, polygons fanned (0,k,k+1), each corner keeps its own v and vn.
struct ObjMesh {
    std::vector<vec3> v, vn;
    std::vector<long> corner_v, corner_n;   // 3 per triangle, 1-based, 0 = absent
};

bool read_obj(const char* path, ObjMesh* m, std::string* err) {
    FILE* f = fopen(path, "r");
    if (!f) { *err = std::string("cannot open OBJ file '") + path + "'"; return false; }
    char buf[8192];
    bool seen_face = false;
    while (fgets(buf, sizeof buf, f)) {
        const char* s = buf;
        while (*s == ' ' || *s == '\t') ++s;
        const bool ws1 = s[1] == ' ' || s[1] == '\t';
        if (s[0] == 'v' && ws1) {
            vec3 p;
            if (sscanf(s + 2, "%f %f %f", &p.x, &p.y, &p.z) == 3) m->v.push_back(p);
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            vec3 p;
            if (sscanf(s + 3, "%f %f %f", &p.x, &p.y, &p.z) == 3) m->vn.push_back(p);
        } else if (s[0] == 'f' && ws1) {
            std::vector<long> cv, cn;
            char* q = const_cast<char*>(s + 2);
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
                char* end = q;
                long vi = strtol(q, &end, 10), ni = 0;
                if (end == q) break;
                q = end;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') strtol(q, &q, 10);
                    if (*q == '/') { ++q; ni = strtol(q, &q, 10); }
                }
                if (vi < 0) vi += (long)m->v.size() + 1;
                if (ni < 0) ni += (long)m->vn.size() + 1;
                cv.push_back(vi); cn.push_back(ni);
            }
            for (size_t k = 1; k + 1 < cv.size(); ++k) {
                const size_t c3[3] = {0, k, k + 1};
                for (int j = 0; j < 3; ++j) { m->corner_v.push_back(cv[c3[j]]); m->corner_n.push_back(cn[c3[j]]); }
            }
            if (cv.size() >= 3) seen_face = true;
        } else if ((s[0] == 'o' || s[0] == 'g') && (ws1 || s[1] == '\n' || s[1] == '\r' || s[1] == '\0')) {
            if (seen_face) break;   // models[0] only (mod.rs:137)
        }
    }
    fclose(f);
    for (size_t i = 0; i < m->corner_v.size(); ++i) {
        if (m->corner_v[i] < 1 || (size_t)m->corner_v[i] > m->v.size()) { *err = "OBJ face references a missing vertex"; return false; }
        if (m->corner_n[i] < 0 || (size_t)m->corner_n[i] > m->vn.size()) { *err = "OBJ face references a missing normal"; return false; }
    }
    return true;
}

void place_obj(const ObjMesh& m, float scale_f, vec3 offset, uint32_t mat, SceneBuilder* b) {
    const bool with_normals = !m.corner_n.empty() && m.corner_n[0] > 0;   // mesh.normals non-empty (mod.rs:150)
    for (size_t t = 0; t + 2 < m.corner_v.size(); t += 3) {
        vec3 p[3], n[3];
        for (int j = 0; j < 3; ++j) {
            p[j] = add(offset, scale(m.v[m.corner_v[t + j] - 1], scale_f));   // mod.rs:147
            n[j] = with_normals && m.corner_n[t + j] > 0 ? m.vn[m.corner_n[t + j] - 1] : mk(0, 0, 0);
        }
        if (with_normals) b->triangle(p[0], p[1], p[2], n[0], n[1], n[2], mat);
        else b->flat_triangle(p[0], p[1], p[2], mat);
    }
}

void place_mesh(const float* vpos, const float* vnrm, const uint32_t* faces, uint32_t n_face, float scale_f, vec3 offset,
                uint32_t mat, SceneBuilder* b) {
    for (uint32_t f = 0; f < n_face; ++f) {
        vec3 p[3], n[3];
        for (int j = 0; j < 3; ++j) {
            const uint32_t i = faces[3 * f + j];
            p[j] = add(offset, scale(mk(vpos[3 * i], vpos[3 * i + 1], vpos[3 * i + 2]), scale_f));
            n[j] = vnrm ? mk(vnrm[3 * i], vnrm[3 * i + 1], vnrm[3 * i + 2]) : mk(0, 0, 0);
        }
        if (vnrm) b->triangle(p[0], p[1], p[2], n[0], n[1], n[2], mat);
        else b->flat_triangle(p[0], p[1], p[2], mat);
    }
}

// Scene::cornell_box (src/scene/mod.rs:266-331)
struct Cornell {
    float half, zoff;
    uint32_t red, green, blue, grey, light;
    vec3 light_center;
    float light_radius;
};
Cornell cornell_walls(SceneBuilder* b) {
    Cornell c;
    c.half = 20.0f; c.zoff = -48.0f;
    c.red = b->material(RT_BSDF_DIFFUSE, 1, 0, 0, 0, 0, 0);
    c.blue = b->material(RT_BSDF_DIFFUSE, 0, 0, 1, 0, 0, 0);
    c.green = b->material(RT_BSDF_DIFFUSE, 0, 1, 0, 0, 0, 0);
    c.grey = b->material(RT_BSDF_DIFFUSE, 0.78f, 0.78f, 0.78f, 0, 0, 0);
    c.light = b->material(RT_BSDF_DIFFUSE, 0, 0, 0, 1, 1, 1);
    c.light_radius = 7.0f;
    c.light_center = mk(0.0f, c.half + c.light_radius * 0.6f, c.zoff - c.half / 2.0f);
    const float h = c.half, z = c.zoff - c.half;
    const vec3 p0 = mk(-h, -h, 1.0f), p1 = mk(-h, -h, z), p2 = mk(h, -h, z), p3 = mk(h, -h, 1.0f), p4 = mk(-h, h, z), p5 = mk(h, h, z),
               p6 = mk(-h, h, 1.0f), p7 = mk(h, h, 1.0f), p8 = mk(-h, h, 1.0f), p9 = mk(-h, h, z), p10 = mk(h, h, z), p11 = mk(h, h, 1.0f);
    b->flat_triangle(p1, p0, p2, c.grey);   b->flat_triangle(p3, p2, p0, c.grey);     // bottom
    b->flat_triangle(p4, p5, p6, c.grey);   b->flat_triangle(p7, p6, p5, c.grey);     // top
    b->flat_triangle(p4, p1, p2, c.green);  b->flat_triangle(p2, p5, p4, c.green);    // back
    b->flat_triangle(p8, p0, p9, c.red);    b->flat_triangle(p1, p9, p0, c.red);      // left
    b->flat_triangle(p3, p11, p2, c.blue);  b->flat_triangle(p10, p2, p11, c.blue);   // right
    return c;
}

}  // namespace

bool make_named_scene(const char* name, const char* obj_path, const float* vpos, const float* vnrm, uint32_t n_vert,
                      const uint32_t* faces, uint32_t n_face, HostScene* out, std::string* err) {
    (void)n_vert;
    SceneBuilder b(out);
    const std::string n(name ? name : "");
    const float sphere_radius = 6.0f;
    if (n == "triangle") {   // Scene::new_triangle, mod.rs:108-133
        const uint32_t mirror = b.material(RT_BSDF_SPECULAR, 1, 1, 1, 0, 0, 0);
        const uint32_t light = b.material(RT_BSDF_DIFFUSE, 0, 0, 0, 1, 1, 1);
        b.triangle(mk(-5, -5, -20), mk(5, -5, -20), mk(5, 5, -20), unit3(mk(-0.4f, 0.0f, 1.0f)), unit3(mk(0.4f, 0.0f, 1.0f)),
                   unit3(mk(0.0f, 0.0f, 1.0f)), mirror);
        b.sphere(mk(0.0f, 0.0f, 10.0f), 8.0f, light);
        return true;
    }
    if (n == "sphere_on_plane") {   // BASELINE config 1; not a reference constructor (screenshots/sphere_on_top_of_plane.png)
        const uint32_t grey = b.material(RT_BSDF_DIFFUSE, 0.78f, 0.78f, 0.78f, 0, 0, 0);
        const uint32_t light = b.material(RT_BSDF_DIFFUSE, 0, 0, 0, 1, 1, 1);
        const vec3 a = mk(-60.0f, -20.0f, 10.0f), bb = mk(-60.0f, -20.0f, -140.0f), c = mk(60.0f, -20.0f, -140.0f), d = mk(60.0f, -20.0f, 10.0f);
        b.flat_triangle(bb, a, c, grey);
        b.flat_triangle(d, c, a, grey);
        b.sphere(mk(0.0f, 20.0f + 7.0f * 0.6f, -48.0f - 20.0f / 2.0f), 7.0f, light);   // the Cornell lamp (mod.rs:279-287)
        b.sphere(mk(0.0f, -20.0f + 8.0f, -48.0f), 8.0f, grey);
        return true;
    }
    Cornell cb = cornell_walls(&b);
    const float h = cb.half, zo = cb.zoff;
    if (n == "specular" || n == "diffuse") {   // mod.rs:333-401
        const uint32_t first = n == "specular" ? b.material(RT_BSDF_SPECULAR, 1, 1, 1, 0, 0, 0) : cb.grey;
        b.sphere(cb.light_center, cb.light_radius, cb.light);
        b.sphere(mk(-h / 3.0f, -h + sphere_radius, zo - 2.0f * h / 3.0f), sphere_radius, first);
        b.sphere(mk(h / 3.0f, -h + sphere_radius, zo - h / 3.0f), sphere_radius, cb.red);
        return true;
    }
    if (n == "teapot" || n == "specular_teapot") {   // mod.rs:229-264; specular_teapot = BASELINE config 5 (composed)
        ObjMesh m;
        if (!read_obj(obj_path ? obj_path : "obj/teapot.obj", &m, err)) return false;
        const uint32_t grey = b.material(RT_BSDF_DIFFUSE, 0.78f, 0.78f, 0.78f, 0, 0, 0);
        place_obj(m, 0.13f, mk(-h / 3.0f - 2.0f, -15.0f, zo - 2.5f * h / 3.0f), grey, &b);
        b.sphere(cb.light_center, cb.light_radius, cb.light);
        if (n == "specular_teapot") {
            b.sphere(mk(h / 3.0f + 2.0f, -h + sphere_radius, zo - 2.0f * h / 3.0f), sphere_radius, b.material(RT_BSDF_SPECULAR, 1, 1, 1, 0, 0, 0));
            b.sphere(mk(h / 3.0f, -h + 4.0f, zo + h / 3.0f), 4.0f, cb.red);
        } else {
            b.sphere(mk(h / 3.0f, -h + sphere_radius, zo - h / 3.0f), sphere_radius, cb.red);
        }
        return true;
    }
    if (n == "dragon") {   // mod.rs:191-227
        const uint32_t grey = b.material(RT_BSDF_DIFFUSE, 0.78f, 0.78f, 0.78f, 0, 0, 0);
        const vec3 off = mk(-h / 3.0f, -h, zo - 2.0f * h / 3.0f);
        if (faces) {
            place_mesh(vpos, vnrm, faces, n_face, 2.0f, off, grey, &b);
        } else {
            ObjMesh m;
            if (!read_obj(obj_path ? obj_path : "obj/dragon.obj", &m, err)) return false;
            place_obj(m, 2.0f, off, grey, &b);
        }
        b.sphere(cb.light_center, cb.light_radius, cb.light);
        b.sphere(mk(h / 3.0f + 2.0f, -h + sphere_radius, zo - h / 3.0f + 2.0f), sphere_radius, cb.red);
        return true;
    }
    *err = "unknown scene '" + n + "'";
    return false;
}

// ---------------------------------------------------------------------------------------------
// Procedural stand-in for the missing obj/dragon.obj (.MISSING_LARGE_BLOBS:1): a (2,3) torus-knot
// tube with a scaly displacement, in the model-space box the reference's dragon placement
// (scale 2, offset (-20/3, -20, -48-40/3), mod.rs:200-209) puts on the Cornell floor.
void make_dragon_standin(uint32_t nu, uint32_t nv, float* vpos, float* vnrm, uint32_t* faces) {
    const float TWO_PI = 6.2831855f;
    for (uint32_t iu = 0; iu < nu; ++iu) {
        const float u = TWO_PI * ((float)iu / (float)nu);
        float s2, c2, s3, c3;
        sincos_f(2.0f * u, &s2, &c2);
        sincos_f(3.0f * u, &s3, &c3);
        // centre curve C(u) = ((2 + cos 3u) cos 2u, (2 + cos 3u) sin 2u, sin 3u) and its derivative
        const float R = 2.0f + c3;
        const vec3 C = mk(R * c2, R * s2, s3);
        const vec3 dC = mk(-3.0f * s3 * c2 - 2.0f * R * s2, -3.0f * s3 * s2 + 2.0f * R * c2, 3.0f * c3);
        const vec3 T = unit3(dC);
        const vec3 B = unit3(cross3(T, mk(0.0f, 0.0f, 1.0f)));
        const vec3 N = cross3(B, T);
        float s5, c5;
        sincos_f(5.0f * u, &s5, &c5);
        for (uint32_t iv = 0; iv < nv; ++iv) {
            const float v = TWO_PI * ((float)iv / (float)nv);
            float sv, cv, sa, ca, sb, cb;
            sincos_f(v, &sv, &cv);
            sincos_f(60.0f * u + 3.0f * v, &sa, &ca);
            sincos_f(7.0f * v, &sb, &cb);
            const float r = 0.35f * (1.0f + 0.25f * s5) + 0.05f * sa * sb + 0.02f * ca;
            const vec3 radial = add(scale(N, cv), scale(B, sv));
            const vec3 P = add(C, scale(radial, r));
            // model space: knot plane upright (x, y), thickness along z; 1.4 x, resting just above y = 0
            const uint32_t i = iu * nv + iv;
            vpos[3 * i] = 1.4f * P.x;
            vpos[3 * i + 1] = 1.4f * P.y + 4.4f;
            vpos[3 * i + 2] = 1.4f * P.z;
            vnrm[3 * i] = radial.x; vnrm[3 * i + 1] = radial.y; vnrm[3 * i + 2] = radial.z;
        }
    }
    uint32_t k = 0;
    for (uint32_t iu = 0; iu < nu; ++iu) {
        const uint32_t ju = (iu + 1) % nu;
        for (uint32_t iv = 0; iv < nv; ++iv) {
            const uint32_t jv = (iv + 1) % nv;
            const uint32_t a = iu * nv + iv, b = ju * nv + iv, c = ju * nv + jv, d = iu * nv + jv;
            faces[k++] = a; faces[k++] = b; faces[k++] = c;
            faces[k++] = a; faces[k++] = c; faces[k++] = d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BVH::build of bvh 0.3.2 (Cargo.lock:108-111; called at src/scene/mod.rs:99), restated from the
// crate's published algorithm: top-down, split axis = largest axis of the centroid bounds, half
// split when that extent < 1e-5, otherwise 6 SAH buckets and the cheapest of the 5 partitions;
// one shape per leaf; each parent stores both child boxes.  Here the leaves are folded into the
// parent's child words, inner nodes are numbered in pre-order, and the working set is an index
// array partitioned in place (bucket by bucket, which is the order the crate concatenates in).
namespace {

// Worker threads of the host builders: RT_HOST_THREADS, else OMP_NUM_THREADS, else the hardware's count, at most 16 (a GPU
// box reports every CPU of the machine while a job owns a share of them).
unsigned host_threads() {
    for (const char* name : {"RT_HOST_THREADS", "OMP_NUM_THREADS"}) {
        const char* v = getenv(name);
        if (v && *v) {
            const long n = strtol(v, nullptr, 10);
            if (n >= 1) return n > 64 ? 64u : (unsigned)n;
        }
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1u : (hc > 16 ? 16u : hc);
}

// f(begin, end) over [0, n) in at most host_threads() contiguous pieces of at least `grain` items.
template <class F>
void parallel_for(size_t n, size_t grain, F f) {
    const size_t nt = host_threads();
    size_t pieces = grain ? n / grain : 1;
    if (pieces > nt) pieces = nt;
    if (pieces <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve(pieces - 1);
    for (size_t c = 1; c < pieces; ++c) th.emplace_back([=] { f(c * n / pieces, (c + 1) * n / pieces); });
    f((size_t)0, n / pieces);
    for (size_t c = 0; c < th.size(); ++c) th[c].join();
}

struct Box {
    float mn[3], mx[3];
};
inline float lo(float a, float b) { return a < b ? a : b; }
inline float hi(float a, float b) { return a > b ? a : b; }
inline Box empty_box() {
    const float inf = std::numeric_limits<float>::infinity();
    Box b = {{inf, inf, inf}, {-inf, -inf, -inf}};
    return b;
}
inline void join_into(Box* a, const Box& b) {
    for (int k = 0; k < 3; ++k) { a->mn[k] = lo(a->mn[k], b.mn[k]); a->mx[k] = hi(a->mx[k], b.mx[k]); }
}
inline float area(const Box& b) {
    const float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}

class Builder {
  public:
    Builder(const HostScene& sc, FlatBVH* out) : out_(out) {
        const uint32_t T = sc.n_tri(), S = sc.n_sph(), n = T + S;
        box_.resize(n);
        ctr_.resize(3 * (size_t)n);
        for (uint32_t i = 0; i < T; ++i) {   // Bounded for Triangle, objects.rs:302-316
            const float* p = &sc.tri_pos[9 * (size_t)i];
            for (int k = 0; k < 3; ++k) {
                box_[i].mn[k] = lo(lo(p[k], p[3 + k]), p[6 + k]);
                box_[i].mx[k] = hi(hi(p[k], p[3 + k]), p[6 + k]);
            }
        }
        for (uint32_t i = 0; i < S; ++i) {   // Bounded for Sphere, objects.rs:293-300
            const float* c = &sc.sph[4 * (size_t)i];
            for (int k = 0; k < 3; ++k) { box_[T + i].mn[k] = c[k] - c[3]; box_[T + i].mx[k] = c[k] + c[3]; }
        }
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) ctr_[3 * (size_t)i + k] = box_[i].mn[k] + (box_[i].mx[k] - box_[i].mn[k]) / 2.0f;
        idx_.resize(n);
        tmp_.resize(n);
        for (uint32_t i = 0; i < n; ++i) idx_[i] = i;
        // One shape per leaf: a subtree over m shapes has m - 1 inner nodes, so the pre-order index of every node and the rank
        // (position in the reference's right-first DFS, mod.rs:427-432) of every leaf are known before the subtree is built, and
        // disjoint subtrees can be built by different threads with the result of the sequential recursion.
        out_->words.assign(n > 1 ? 16 * (size_t)(n - 1) : 0, 0u);
        out_->rank.assign(n, 0u);
        depth_.store(0);
        grain_ = n / (4 * host_threads());
        if (grain_ < 4096) grain_ = 4096;
        out_->root = n ? split(0, n, 1, 0, 0) : LEAF_BIT;
        out_->depth = depth_.load();
    }

  private:
    // [b, e) of idx_, tree level, pre-order index of the node if it is an inner one, rank of the first leaf visited
    uint32_t split(uint32_t b, uint32_t e, uint32_t level, uint32_t me, uint32_t rank0) {
        if (e - b == 1) { out_->rank[idx_[b]] = rank0; return LEAF_BIT | idx_[b]; }
        uint32_t seen = depth_.load(std::memory_order_relaxed);
        while (level > seen && !depth_.compare_exchange_weak(seen, level, std::memory_order_relaxed)) {}

        Box all = empty_box(), cb = empty_box();
        for (uint32_t k = b; k < e; ++k) {
            const uint32_t o = idx_[k];
            join_into(&all, box_[o]);
            for (int a = 0; a < 3; ++a) { cb.mn[a] = lo(cb.mn[a], ctr_[3 * (size_t)o + a]); cb.mx[a] = hi(cb.mx[a], ctr_[3 * (size_t)o + a]); }
        }
        const float ex = cb.mx[0] - cb.mn[0], ey = cb.mx[1] - cb.mn[1], ez = cb.mx[2] - cb.mn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
        const float extent = cb.mx[axis] - cb.mn[axis];

        uint32_t mid;
        Box lbox = empty_box(), rbox = empty_box();
        bool halved = extent < 0.00001f;
        if (!halved) {
            const int NB = 6;
            Box bb[NB];
            uint32_t cnt[NB];
            for (int i = 0; i < NB; ++i) { bb[i] = empty_box(); cnt[i] = 0; }
            for (uint32_t k = b; k < e; ++k) {
                const uint32_t o = idx_[k];
                const float rel = (ctr_[3 * (size_t)o + axis] - cb.mn[axis]) / extent;
                int bn = (int)(rel * ((float)NB - 0.01f));
                bn = bn < 0 ? 0 : (bn > NB - 1 ? NB - 1 : bn);
                tmp_[k] = (uint32_t)bn;
                join_into(&bb[bn], box_[o]);
                cnt[bn]++;
            }
            int best = -1;
            float best_cost = std::numeric_limits<float>::infinity();
            for (int i = 0; i < NB - 1; ++i) {
                Box l = empty_box(), r = empty_box();
                uint32_t nl = 0, nr = 0;
                for (int j = 0; j <= i; ++j) { join_into(&l, bb[j]); nl += cnt[j]; }
                for (int j = i + 1; j < NB; ++j) { join_into(&r, bb[j]); nr += cnt[j]; }
                const float cost = ((float)nl * area(l) + (float)nr * area(r)) / area(all);
                if (cost < best_cost) { best = i; best_cost = cost; lbox = l; rbox = r; }
            }
            // counting sort of idx_[b,e) by bucket = the crate's concatenation of bucket vectors
            uint32_t start[NB + 1];
            start[0] = 0;
            for (int i = 0; i < NB; ++i) start[i + 1] = start[i] + cnt[i];
            uint32_t fill[NB];
            for (int i = 0; i < NB; ++i) fill[i] = start[i];
            static thread_local std::vector<uint32_t> scratch;
            if (scratch.size() < e - b) scratch.resize(e - b);
            for (uint32_t k = b; k < e; ++k) scratch[fill[tmp_[k]]++] = idx_[k];
            for (uint32_t k = b; k < e; ++k) idx_[k] = scratch[k - b];
            const int cut = best < 0 ? 0 : best;
            mid = b + start[cut + 1];
            if (best < 0 || mid == b || mid == e) halved = true;   // every cost NaN: degenerate bounds
        }
        if (halved) {
            mid = b + (e - b) / 2;
            lbox = empty_box(); rbox = empty_box();
            for (uint32_t k = b; k < mid; ++k) join_into(&lbox, box_[idx_[k]]);
            for (uint32_t k = mid; k < e; ++k) join_into(&rbox, box_[idx_[k]]);
        }
        // pre-order: the left subtree's mid - b - 1 inner nodes follow this one; right-first DFS: the right subtree's e - mid leaves rank first
        const uint32_t left_me = me + 1, right_me = me + (mid - b), left_rank = rank0 + (e - mid);
        uint32_t left, right;
        if (mid - b >= grain_ && e - mid >= grain_) {
            std::thread other([&] { left = split(b, mid, level + 1, left_me, left_rank); });
            right = split(mid, e, level + 1, right_me, rank0);
            other.join();
        } else {
            left = split(b, mid, level + 1, left_me, left_rank);
            right = split(mid, e, level + 1, right_me, rank0);
        }
        uint32_t* w = &out_->words[16 * (size_t)me];
        memcpy(w + 0, lbox.mn, 12); memcpy(w + 3, lbox.mx, 12);
        memcpy(w + 6, rbox.mn, 12); memcpy(w + 9, rbox.mx, 12);
        w[12] = left; w[13] = right; w[14] = 0; w[15] = 0;
        return me;
    }

    FlatBVH* out_;
    std::vector<Box> box_;
    std::vector<float> ctr_;
    std::vector<uint32_t> idx_, tmp_;
    std::atomic<uint32_t> depth_;
    uint32_t grain_ = 4096;
};

}  // namespace

void build_bvh(const HostScene& sc, FlatBVH* out) { Builder b(sc, out); }

// ---------------------------------------------------------------------------------------------
// Occlusion BVH (layout: host_scene.h).  No reference counterpart: the reference walks its binary
// tree for shadow rays too.  What makes this a drop-in is that is_occluded's result does not depend
// on the order or the shape of the search: it is "some non-emissive object whose own box the ray
// passes reports 0 < d < max_dist" (a leaf is reached iff the ray passes every ancestor box, each of
// which contains the leaf's own box, and the slab test is monotone in the box bounds).
namespace {

struct WChild {
    uint32_t ref;   // binary-tree child word
    Box box;
};

class WideBuilder {
  public:
    WideBuilder(const HostScene& sc, const FlatBVH& bvh, WideBVH* out) : sc_(sc), bvh_(bvh), out_(out) {
        out_->nodes.clear(); out_->prims.clear(); out_->depth = 0; out_->ok = false;
        const uint32_t T = sc.n_tri(), S = sc.n_sph(), n = T + S;
        out_->rec_of_obj.assign(n, 0xFFFFFFFFu);
        if (!n) return;
        blocker_.resize(n);
        for (uint32_t i = 0; i < T; ++i) blocker_[i] = scene_is_black(sc.mats[sc.tri_mat[i]].emittance);
        for (uint32_t i = 0; i < S; ++i) blocker_[T + i] = scene_is_black(sc.mats[sc.sph_mat[i]].emittance);
        // inner nodes are numbered in pre-order: children have larger indices than their parent
        const uint32_t ni = bvh.n_inner();
        has_blocker_.assign(ni, 0);
        for (uint32_t i = ni; i-- > 0;) {
            const uint32_t* w = &bvh.words[16 * (size_t)i];
            has_blocker_[i] = (ref_has_blocker(w[12]) || ref_has_blocker(w[13])) ? 1 : 0;
        }
        if (!ref_has_blocker(bvh.root)) return;
        for (size_t i = 0; i < sc.tri_pos.size(); ++i)
            if (!(fabsf(sc.tri_pos[i]) <= WIDE_COORD_MAX)) return;
        for (size_t i = 0; i < sc.sph.size(); ++i)
            if (!(fabsf(sc.sph[i]) <= WIDE_COORD_MAX / 2.0f)) return;

        // Breadth-first: the inner children of a node get consecutive node indices, its primitives consecutive records.  One tree
        // level at a time: the nodes of a level are collapsed and quantised by several threads; the child and record indices in
        // between are running sums in node order, which is the order a sequential queue would hand them out in.
        struct Gathered { std::vector<WChild> ch; uint32_t n_inner; };
        std::vector<uint32_t> refs(1, bvh.root), next_refs, child_base, prim_base;
        std::vector<Gathered> g;
        uint32_t level_base = 0, n_prims = 0;
        out_->nodes.assign(20, 0u);
        std::atomic<bool> ok(true);
        while (!refs.empty() && ok.load()) {
            ++out_->depth;
            const size_t m = refs.size();
            g.assign(m, Gathered());
            parallel_for(m, 512, [&](size_t b, size_t e) {
                std::vector<WChild> ch;
                for (size_t i = b; i < e; ++i) {
                    ch.clear();
                    gather_children(refs[i], &ch);
                    Gathered& gi = g[i];   // inner children first
                    for (size_t k = 0; k < ch.size(); ++k) if (!(ch[k].ref & LEAF_BIT)) gi.ch.push_back(ch[k]);
                    gi.n_inner = (uint32_t)gi.ch.size();
                    for (size_t k = 0; k < ch.size(); ++k) if (ch[k].ref & LEAF_BIT) gi.ch.push_back(ch[k]);
                }
            });
            child_base.resize(m); prim_base.resize(m);
            const uint64_t next_base = (uint64_t)level_base + m;
            uint64_t nn = next_base, np = n_prims;
            for (size_t i = 0; i < m; ++i) {
                child_base[i] = (uint32_t)nn; nn += g[i].n_inner;
                prim_base[i] = (uint32_t)np; np += g[i].ch.size() - g[i].n_inner;
            }
            if (nn >= (1u << 23) || np + 8 >= (1u << 23)) { ok.store(false); break; }   // index fields of the stack entries (rt_kernels.h)
            out_->nodes.resize(20 * (size_t)nn, 0u);
            out_->prims.resize(12 * (size_t)np, 0.0f);
            next_refs.assign((size_t)(nn - next_base), 0u);
            parallel_for(m, 512, [&](size_t b, size_t e) {
                for (size_t i = b; i < e; ++i) {
                    const Gathered& gi = g[i];
                    for (uint32_t k = 0; k < gi.n_inner; ++k) next_refs[child_base[i] - next_base + k] = gi.ch[k].ref;
                    for (size_t k = gi.n_inner; k < gi.ch.size(); ++k) emit_prim(gi.ch[k].ref & OBJECT_MASK, prim_base[i] + (uint32_t)(k - gi.n_inner));
                    if (!write_node(level_base + (uint32_t)i, gi.ch, gi.n_inner, child_base[i], prim_base[i])) ok.store(false);
                }
            });
            level_base = (uint32_t)next_base;
            n_prims = (uint32_t)np;
            refs.swap(next_refs);
        }
        if (!ok.load()) {
            out_->nodes.clear(); out_->prims.clear(); out_->depth = 0;
            return;
        }
        out_->ok = true;
    }

  private:
    bool ref_has_blocker(uint32_t ref) const { return (ref & LEAF_BIT) ? blocker_[ref & OBJECT_MASK] != 0 : has_blocker_[ref] != 0; }

    Box object_box(uint32_t obj) const {   // Bounded for Triangle / Sphere, objects.rs:293-316 (the same floats as Builder)
        Box b;
        const uint32_t T = sc_.n_tri();
        if (obj < T) {
            const float* p = &sc_.tri_pos[9 * (size_t)obj];
            for (int k = 0; k < 3; ++k) { b.mn[k] = lo(lo(p[k], p[3 + k]), p[6 + k]); b.mx[k] = hi(hi(p[k], p[3 + k]), p[6 + k]); }
        } else {
            const float* c = &sc_.sph[4 * (size_t)(obj - T)];
            for (int k = 0; k < 3; ++k) { b.mn[k] = c[k] - c[3]; b.mx[k] = c[k] + c[3]; }
        }
        return b;
    }

    void children_of(uint32_t inner_ref, std::vector<WChild>* out) const {
        const uint32_t* w = &bvh_.words[16 * (size_t)inner_ref];
        for (int c = 0; c < 2; ++c) {
            const uint32_t ref = w[12 + c];
            if (!ref_has_blocker(ref)) continue;
            WChild ch;
            ch.ref = ref;
            memcpy(ch.box.mn, w + 6 * c, 12);
            memcpy(ch.box.mx, w + 6 * c + 3, 12);
            out->push_back(ch);
        }
    }

    // Up to 8 children: open the inner child with the largest box until nothing is left to open.
    void gather_children(uint32_t ref, std::vector<WChild>* ch) const {
        if (ref & LEAF_BIT) {   // one-object scene
            WChild c;
            c.ref = ref;
            c.box = object_box(ref & OBJECT_MASK);
            ch->push_back(c);
            return;
        }
        children_of(ref, ch);
        for (;;) {
            int best = -1;
            float best_area = -1.0f;
            for (size_t k = 0; k < ch->size(); ++k) {
                if ((*ch)[k].ref & LEAF_BIT) continue;
                const float a = area((*ch)[k].box);
                if (best < 0 || a > best_area) { best = (int)k; best_area = a; }
            }
            if (best < 0) break;
            std::vector<WChild> sub;
            children_of((*ch)[best].ref, &sub);
            if (ch->size() - 1 + sub.size() > 8) break;
            ch->erase(ch->begin() + best);
            ch->insert(ch->end(), sub.begin(), sub.end());
        }
    }

    void emit_prim(uint32_t obj, uint32_t rec) {
        const uint32_t T = sc_.n_tri();
        out_->rec_of_obj[obj] = rec;
        float* r = &out_->prims[12 * (size_t)rec];   // zero-filled by the caller
        if (obj < T) {
            const float* p = &sc_.tri_pos[9 * (size_t)obj];
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = bits_to_float(0u);
            r[4] = p[3]; r[5] = p[4]; r[6] = p[5]; r[7] = bits_to_float(obj);
            r[8] = p[6]; r[9] = p[7]; r[10] = p[8];
        } else {
            const float* c = &sc_.sph[4 * (size_t)(obj - T)];
            r[0] = c[0]; r[1] = c[1]; r[2] = c[2]; r[3] = bits_to_float(1u);
            r[4] = c[3]; r[5] = bits_to_float(obj);
        }
    }

    bool write_node(uint32_t me, const std::vector<WChild>& ch, uint32_t n_inner, uint32_t child_base, uint32_t prim_base) {
        Box nb = empty_box();
        for (size_t k = 0; k < ch.size(); ++k) join_into(&nb, ch[k].box);
        uint32_t* w = &out_->nodes[20 * (size_t)me];
        uint8_t* q = reinterpret_cast<uint8_t*>(w + 8);
        memset(q, 0, 48);
        uint32_t meta = 0;
        for (int a = 0; a < 3; ++a) {
            // 64-bit mantissa: nothing below rounds for scenes the range checks admit
            float cmn[8], cmx[8];
            for (size_t k = 0; k < ch.size(); ++k) { cmn[k] = ch[k].box.mn[a]; cmx[k] = ch[k].box.mx[a]; }
            unsigned char ql[8], qh[8];
            int e = 0;
            bool inexact = false;
            if (!wide_axis_planes<long double>(nb.mn[a], nb.mx[a], (int)ch.size(), cmn, cmx, 1, ql, qh, &e, &inexact)) return false;
            for (size_t k = 0; k < ch.size(); ++k) { q[8 * a + k] = ql[k]; q[24 + 8 * a + k] = qh[k]; }
            for (size_t k = ch.size(); k < 8; ++k) { q[8 * a + k] = 255; q[24 + 8 * a + k] = 0; }
            meta |= (uint32_t)(e + WIDE_EXP_BIAS) << (8 * a);
        }
        meta |= n_inner << 24;
        meta |= (uint32_t)ch.size() << 28;
        memcpy(w, nb.mn, 12);
        w[3] = meta;
        w[4] = child_base; w[5] = prim_base; w[6] = 0; w[7] = 0;
        return true;
    }

    const HostScene& sc_;
    const FlatBVH& bvh_;
    WideBVH* out_;
    std::vector<uint8_t> blocker_, has_blocker_;
};

}  // namespace

void build_wide_bvh(const HostScene& sc, const FlatBVH& bvh, WideBVH* out) { WideBuilder b(sc, bvh, out); }

}  // namespace rt
