"""Scene — host-side mirror of the reference's `Scene` (src/scene/mod.rs:20-24, 82-401).

The object list, the BVH and every ray query live in librustracer_hip.so; this class keeps the
reference's constructor names (`Scene::new_dragon` ... -> `Scene.new_dragon()`) and hands the
opaque handle to `Raytracer`.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, lib, ptr

# standard stand-in sizes (nu, nv): 2*nu*nv triangles
DRAGON_STANDIN_FULL = (2178, 200)    # 871,200 triangles: the size class of the Stanford dragon
DRAGON_STANDIN_SMALL = (500, 100)    # 100,000 triangles


def dragon_standin_mesh(nu, nv):
    """Deterministic dragon-class stand-in mesh for the missing obj/dragon.obj (.MISSING_LARGE_BLOBS:1)."""
    vpos = np.zeros((nu * nv, 3), np.float32)
    vnrm = np.zeros((nu * nv, 3), np.float32)
    faces = np.zeros((2 * nu * nv, 3), np.uint32)
    check(lib().rt_make_dragon_standin(nu, nv, ptr(vpos), ptr(vnrm), ptr(faces)))
    return vpos, vnrm, faces


def write_obj(path, vpos, vnrm, faces):
    """Write a mesh in the `f a//a b//b c//c` form the reference's OBJ files use (obj/teapot.obj:1067)."""
    with open(path, "w") as f:
        f.write("# synthetic dragon-class mesh (rustracer_amd.scene.dragon_standin_mesh)\ng Object001\n")
        np.savetxt(f, vpos, fmt="v %.9g %.9g %.9g")
        np.savetxt(f, vnrm, fmt="vn %.9g %.9g %.9g")
        idx = (faces + 1).astype(np.int64)
        np.savetxt(f, np.stack([idx[:, 0], idx[:, 0], idx[:, 1], idx[:, 1], idx[:, 2], idx[:, 2]], axis=1), fmt="f %d//%d %d//%d %d//%d")


class Scene:
    """The Scene is static (src/scene/mod.rs:19): built once, then only queried."""

    def __init__(self, handle, name=""):
        self._h = C.c_void_p(handle)
        self.name = name
        info = _lib.SceneInfo()
        check(lib().rt_scene_info_get(self._h, C.byref(info)))
        self.info = info

    # ---- constructors (src/scene/mod.rs:108-401) ----
    @classmethod
    def _named(cls, name, obj_path=None):
        h = C.c_void_p()
        check(lib().rt_scene_new_named(name.encode(), obj_path.encode() if obj_path else None, C.byref(h)))
        return cls(h.value, name)

    @classmethod
    def new_triangle(cls):
        return cls._named("triangle")

    @classmethod
    def new_specular(cls):
        return cls._named("specular")

    @classmethod
    def new_diffuse(cls):
        return cls._named("diffuse")

    @classmethod
    def new_teapot(cls, obj_path="obj/teapot.obj"):
        return cls._named("teapot", obj_path)

    @classmethod
    def new_dragon(cls, obj_path="obj/dragon.obj", standin=None):
        """Scene::new_dragon.  With `standin=(nu, nv)`, or when obj_path does not exist, the procedural
        dragon-class mesh takes the file's place (same scale and offset, src/scene/mod.rs:200-209)."""
        if standin is None and os.path.exists(obj_path):
            return cls._named("dragon", obj_path)
        nu, nv = standin or DRAGON_STANDIN_FULL
        vpos, vnrm, faces = dragon_standin_mesh(nu, nv)
        return cls.from_dragon_mesh(vpos, vnrm, faces, name="dragon(synthetic %d triangles)" % faces.shape[0])

    @classmethod
    def from_dragon_mesh(cls, vpos, vnrm, faces, name="dragon"):
        vpos = np.ascontiguousarray(vpos, np.float32)
        vnrm = None if vnrm is None else np.ascontiguousarray(vnrm, np.float32)
        faces = np.ascontiguousarray(faces, np.uint32)
        h = C.c_void_p()
        check(lib().rt_scene_new_dragon_mesh(ptr(vpos), ptr(vnrm), vpos.shape[0], ptr(faces), faces.shape[0], C.byref(h)))
        return cls(h.value, name)

    @classmethod
    def new_sphere_on_plane(cls):
        """BASELINE config 1 (not a reference constructor)."""
        return cls._named("sphere_on_plane")

    @classmethod
    def new_specular_teapot(cls, obj_path="obj/teapot.obj"):
        """BASELINE config 5: new_specular's spheres composed with the teapot (not a reference constructor)."""
        return cls._named("specular_teapot", obj_path)

    @classmethod
    def by_name(cls, name, obj_dir="obj"):
        """SCENE environment variable dispatch (src/main.rs:112-124)."""
        if name == "dragon":
            return cls.new_dragon(os.path.join(obj_dir, "dragon.obj"))
        if name == "teapot":
            return cls.new_teapot(os.path.join(obj_dir, "teapot.obj"))
        if name == "specular_teapot":
            return cls.new_specular_teapot(os.path.join(obj_dir, "teapot.obj"))
        if name in ("specular", "diffuse", "triangle", "sphere_on_plane"):
            return cls._named(name)
        import sys
        print("Unknown scene '%s', using specular" % name, file=sys.stderr)
        return cls.new_specular()

    @classmethod
    def from_arrays(cls, tri_pos, tri_nrm, tri_mat, sph, sph_mat, mats, name="raw"):
        """Scene::new over a raw object list (rt_scene_create)."""
        tri_pos = np.ascontiguousarray(tri_pos, np.float32).reshape(-1, 9)
        tri_nrm = np.ascontiguousarray(tri_nrm, np.float32).reshape(-1, 9)
        tri_mat = np.ascontiguousarray(tri_mat, np.uint32).reshape(-1)
        sph = np.ascontiguousarray(sph, np.float32).reshape(-1, 4)
        sph_mat = np.ascontiguousarray(sph_mat, np.uint32).reshape(-1)
        mats = np.ascontiguousarray(mats).view(np.uint32).reshape(-1, 8)
        h = C.c_void_p()
        check(lib().rt_scene_create(ptr(tri_pos), ptr(tri_nrm), ptr(tri_mat), tri_pos.shape[0], ptr(sph), ptr(sph_mat), sph.shape[0],
                                    ptr(mats), mats.shape[0], C.byref(h)))
        return cls(h.value, name)

    # ---- queries ----
    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("scene was destroyed")
        return self._h

    def build_info(self):
        """(ms of the binary / 8-wide tree build, which of them the GPU built, why not)."""
        ms = (C.c_double * 4)()
        dev = (C.c_uint32 * 2)()
        note = C.create_string_buffer(512)
        check(lib().rt_scene_build_info(self.handle, ms, dev, note, 512))
        return {"binary_ms": ms[0], "wide_ms": ms[1], "binary_device_ms": ms[2], "wide_device_ms": ms[3], "binary_on_device": bool(dev[0]), "wide_on_device": bool(dev[1]), "note": note.value.decode()}

    def light_indexes(self):
        out = np.zeros(self.info.n_lights, np.uint32)
        check(lib().rt_scene_lights(self.handle, ptr(out), out.shape[0]))
        return out

    def export(self):
        i = self.info
        out = dict(tri_pos=np.zeros((i.n_tri, 9), np.float32), tri_nrm=np.zeros((i.n_tri, 9), np.float32),
                   tri_mat=np.zeros(i.n_tri, np.uint32), sph=np.zeros((i.n_sph, 4), np.float32), sph_mat=np.zeros(i.n_sph, np.uint32),
                   mats=np.zeros((i.n_mat, 8), np.uint32))
        check(lib().rt_scene_export(self.handle, ptr(out["tri_pos"]), ptr(out["tri_nrm"]), ptr(out["tri_mat"]), ptr(out["sph"]),
                                    ptr(out["sph_mat"]), ptr(out["mats"])))
        out["lights"] = self.light_indexes()
        return out

    def bvh(self):
        """(node words [n_inner,16], rank [n_objects]) of the flattened BVH."""
        words = np.zeros((self.info.n_inner_nodes, 16), np.uint32)
        rank = np.zeros(self.info.n_tri + self.info.n_sph, np.uint32)
        check(lib().rt_scene_bvh(self.handle, ptr(words), ptr(rank)))
        return words, rank

    def wide_bvh(self):
        """(node words [n_wide_nodes,20], primitive records [n_wide_prims,12]) of the 8-wide occlusion BVH."""
        nodes = np.zeros((self.info.n_wide_nodes, 20), np.uint32)
        prims = np.zeros((self.info.n_wide_prims, 12), np.float32)
        check(lib().rt_scene_wide_bvh(self.handle, ptr(nodes), ptr(prims)))
        return nodes, prims

    def intersect_rays(self, origins, directions):
        """Scene::intersect (src/scene/mod.rs:406) for a batch of rays -> (object ids, distances)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        ids = np.zeros(o.shape[0], np.uint32)
        ts = np.zeros(o.shape[0], np.float32)
        check(lib().rt_intersect_rays(self.handle, ptr(o), ptr(d), o.shape[0], ptr(ids), ptr(ts)))
        return ids, ts

    def occluded_rays(self, origins, directions, max_dist):
        """Scene::is_occluded (src/scene/mod.rs:452) for a batch of rays."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(max_dist, np.float32).reshape(-1)
        out = np.zeros(o.shape[0], np.uint8)
        check(lib().rt_occluded_rays(self.handle, ptr(o), ptr(d), ptr(m), o.shape[0], ptr(out)))
        return out

    def set_option(self, name, value):
        check(lib().rt_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_uint64(0)
        check(lib().rt_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def counters(self):
        c = _lib.Counters()
        check(lib().rt_get_counters(self.handle, C.byref(c)))
        return c.as_dict()

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
