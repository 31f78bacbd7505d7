import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built libraries are git-ignored: a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    libs = [os.path.join(ROOT, "morphablegraphs_amd", "csrc", "libmg_hip.so"), os.path.join(ROOT, "oracle", "libmg_oracle.so")]
    if not all(os.path.exists(l) for l in libs):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    """Golden vectors produced by oracle/gen_golden.py from the reference's own code."""
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def golden_model(name):
    """Model dict for a golden case: stored arrays, or the seeded synthetic model
    (checked against the digest recorded when the fixture was made)."""
    import hashlib
    from morphablegraphs_amd import synthetic
    g = load_golden(name)
    if "model_eigen_vectors_spatial" in g.files:
        data = {k[len("model_"):]: g[k] for k in g.files if k.startswith("model_")}
        data["n_basis_spatial"] = int(data.pop("n_basis"))
        data["n_dim_spatial"] = int(data.pop("n_dim"))
        if "n_basis_time" in data:
            data["n_basis_time"] = int(data["n_basis_time"])
        data["n_canonical_frames"] = int(g["n_canonical_frames"])
        data["name"] = name
    elif name == "walk_seed0":
        data = synthetic.make_walk_primitive(seed=0)
    elif name == "walk_seed7_tm":
        data = synthetic.make_walk_primitive(seed=7, translation_maxima=(1.5, 2.0, 0.5),
                                             dirichlet_weights=True, realistic=False)
    else:
        raise KeyError(name)
    h = hashlib.sha256()
    for key in ("eigen_vectors_spatial", "mean_spatial_vector", "b_spline_knots_spatial",
                "gmm_weights", "gmm_means", "gmm_covars", "translation_maxima"):
        h.update(np.ascontiguousarray(np.asarray(data[key], dtype=np.float64)).tobytes())
    assert h.hexdigest() == str(g["digest"]), "synthetic model drifted from the golden fixture: " + name
    return data, g


GOLDEN_CASES = ["tiny_tm", "walk_seed0", "walk_seed7_tm", "k1_near_singular", "odd_shape"]


@pytest.fixture(params=GOLDEN_CASES)
def golden_case(request):
    data, g = golden_model(request.param)
    return request.param, data, g
