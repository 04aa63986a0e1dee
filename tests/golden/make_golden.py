"""Generates the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no golden vectors and cannot be built here (no Rust toolchain), so these
fixtures pin the *oracle* (and through it the HIP path) against regressions; they are not outputs
of the reference binary ("parity unpinned", see oracle/oracle_rt.cpp and DESIGN.md).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402

TEAPOT = os.path.join(ROOT, "obj", "teapot.obj")

CASES = [
    dict(kind="ids", scene="specular", params=dict(width=96, height=96)),
    dict(kind="ids", scene="teapot", params=dict(width=96, height=64)),
    dict(kind="ids", scene="dragon_tiny", params=dict(width=64, height=64)),
    dict(kind="rgba", scene="teapot", params=dict(width=64, height=64, spp=4, light_samples=4, bounces=1)),
    dict(kind="rgba", scene="teapot", params=dict(width=48, height=48, spp=4, light_samples=2, bounces=8)),
    dict(kind="rgba", scene="specular", params=dict(width=64, height=64, spp=4, light_samples=4, bounces=8)),
    dict(kind="rgba", scene="sphere_on_plane", params=dict(width=64, height=64, spp=16, light_samples=8, bounces=1, direct_estimator=1)),
    dict(kind="rgba", scene="dragon_tiny", params=dict(width=48, height=48, spp=2, light_samples=4, bounces=8)),
    dict(kind="rgba", scene="teapot", params=dict(width=64, height=64, mode=0)),
]


def dragon_tiny_scene():
    # the mesh comes from the product's deterministic generator; it is input data, not arithmetic under test
    import rustracer_amd as ra
    vp, vn, f = ra.dragon_standin_mesh(120, 40)
    return ob.OracleScene.dragon_mesh(vp, vn, f)


def main():
    manifest = {"generator": "tests/golden/make_golden.py", "seed": 0x5EED, "cases": []}
    for i, case in enumerate(CASES):
        sc = dragon_tiny_scene() if case["scene"] == "dragon_tiny" else ob.OracleScene.named(case["scene"], TEAPOT)
        p = ob.make_params(**case["params"])
        name = "%02d_%s_%s.npz" % (i, case["scene"], case["kind"])
        if case["kind"] == "ids":
            ids, ts, _ = sc.cast_primary(p, nthreads=1)
            np.savez_compressed(os.path.join(HERE, name), ids=ids, t=ts)
        else:
            r = sc.render(p, nthreads=1)
            np.savez_compressed(os.path.join(HERE, name), rgba=r["rgba"],
                                counts=np.array([r["intersect_calls"], r["occluded_calls"]], np.uint64))
        manifest["cases"].append(dict(case, file=name))
        print(name, os.path.getsize(os.path.join(HERE, name)))
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
