"""The oracle against the committed golden vectors (tests/golden/, made by make_golden.py): pins the
oracle's results across compilers, hosts and later edits.  The GPU path is held to the same files in
tests/test_gpu_parity.py::test_golden_fixtures_on_gpu."""
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=[c["file"] for c in MANIFEST["cases"]])
def test_oracle_reproduces_golden(case, teapot_obj):
    data = np.load(os.path.join(GOLDEN, case["file"]))
    if case["scene"] == "dragon_tiny":
        import rustracer_amd as ra
        vp, vn, f = ra.dragon_standin_mesh(120, 40)
        sc = ob.OracleScene.dragon_mesh(vp, vn, f)
    else:
        sc = ob.OracleScene.named(case["scene"], teapot_obj)
    p = ob.make_params(**case["params"])
    if case["kind"] == "ids":
        ids, ts, _ = sc.cast_primary(p)
        assert np.array_equal(ids, data["ids"]) and np.array_equal(ts.view(np.uint32), data["t"].view(np.uint32))
    else:
        r = sc.render(p)        # multi-threaded: the result must not depend on the thread count
        assert np.array_equal(r["rgba"], data["rgba"])
        assert [r["intersect_calls"], r["occluded_calls"]] == data["counts"].tolist()
