"""Synthetic motion-primitive models in the reference's legacy JSON layout.

The reference ships no model data, so tests and bench.py run on
seeded synthetic primitives with the shapes of BASELINE.json's configs
(SURVEY.md §8(d)).  The dictionaries produced here have exactly the keys that
``MotionPrimitive._initialize_from_json`` consumes
(reference morphablegraphs/motion_model/motion_primitive.py:96-163), so the same
dictionary can be fed to the reference, to the oracle and to the HIP backend.

Layout conventions that are binding for the hot path:
  * knots: clamped cubic, ``[0,0,0, linspace(0, F-1, NB-2), F-1,F-1,F-1]``
    (reference morphablegraphs/construction/utils.py:187-198)
  * ``eigen_vectors_spatial`` is (L, NB*D); flat column index = coeff_idx*D + d
    (reference morphablegraphs/construction/fpca/pca_functional_data.py:132-142)
  * pose channel d<3 = root translation, 3+4j..3+4j+3 = joint j quaternion (w,x,y,z)
"""
import numpy as np

LEN_ROOT = 3
LEN_QUAT = 4


def cubic_b_spline_knots(n_basis, n_canonical_frames):
    """Clamped cubic knot vector of length n_basis + 4."""
    knots = np.zeros(n_basis + 4)
    knots[3:-3] = np.linspace(0, n_canonical_frames - 1, n_basis - 2)
    knots[-3:] = n_canonical_frames - 1
    return knots


def make_primitive(seed=0, n_components=40, n_frames=156, n_basis=None, n_dim=79, n_gmm=8,
                   translation_maxima=(1.0, 1.0, 1.0), root_scale=100.0, realistic=True,
                   dirichlet_weights=False, name=None, n_time_components=0, n_basis_time=None):
    """Seeded synthetic primitive ("walk" sized by default: L=40, F=156, NB=31, D=79, K=8).
    n_time_components > 0 adds the legacy time model (eigen_vectors_time (n_basis_time, n_t), mean_time_vector,
    b_spline_knots_time: reference motion_primitive.py:164-181) and the mixture then spans all L + n_t latents, as
    the reference's constructor fits it (construction/motion_model_constructor.py:424)."""
    rng = np.random.default_rng(seed)
    L, F, D, K = int(n_components), int(n_frames), int(n_dim), int(n_gmm)
    Lt = int(n_time_components)
    NB = int(0.2 * F) if n_basis is None else int(n_basis)
    eigen = 0.05 * rng.standard_normal((L, NB * D))
    mean = rng.standard_normal(NB * D)
    if realistic:
        m = mean.reshape(NB, D)
        e = eigen.reshape(L, NB, D)
        m[:, :LEN_ROOT] *= root_scale
        e[:, :, :LEN_ROOT] *= root_scale
        n_joints = (D - LEN_ROOT) // LEN_QUAT
        for j in range(n_joints):
            q = 0.1 * rng.standard_normal(4)
            q[0] = 1.0
            q /= np.linalg.norm(q)
            m[:, LEN_ROOT + LEN_QUAT * j:LEN_ROOT + LEN_QUAT * (j + 1)] = q
    if dirichlet_weights:
        weights = rng.dirichlet(np.ones(K))
    else:
        weights = np.full(K, 1.0 / K)
    Lg = L + Lt
    means = rng.standard_normal((K, Lg))
    covars = np.empty((K, Lg, Lg))
    for k in range(K):
        a = 0.3 * rng.standard_normal((Lg, Lg))
        covars[k] = a @ a.T + 0.5 * np.eye(Lg)
    n_joints = (D - LEN_ROOT) // LEN_QUAT
    time_model = {}
    if Lt > 0:
        trng = np.random.default_rng(seed + 10007)      # its own stream: models without a time part keep their values
        NBt = int(n_basis_time) if n_basis_time is not None else max(4, int(0.1 * F))
        # log of the time increments: around 0 (increments of one sample per canonical frame), harmonics of a few percent
        time_model = {"eigen_vectors_time": (0.04 * trng.standard_normal((NBt, Lt))).tolist(),
                      "mean_time_vector": (0.05 * trng.standard_normal(NBt)).tolist(),
                      "n_basis_time": NBt,
                      "b_spline_knots_time": cubic_b_spline_knots(NBt, F).tolist()}
    out = {
        "name": name or ("synthetic_%d" % seed),
        "n_canonical_frames": F,
        "translation_maxima": [float(v) for v in translation_maxima],
        "eigen_vectors_spatial": eigen.tolist(),
        "mean_spatial_vector": mean.tolist(),
        "n_basis_spatial": NB,
        "n_dim_spatial": D,
        "b_spline_knots_spatial": cubic_b_spline_knots(NB, F).tolist(),
        "gmm_weights": weights.tolist(),
        "gmm_means": means.tolist(),
        "gmm_covars": covars.tolist(),
        "animated_joints": ["joint_%d" % j for j in range(n_joints)],
    }
    out.update(time_model)
    return out


def make_walk_primitive(seed=0, **kw):
    """BASELINE.json configs 1/2/4/5: L=40, F=156, NB=31, D=79, K=8."""
    return make_primitive(seed=seed, n_components=40, n_frames=156, n_dim=79, n_gmm=8, name="walk", **kw)


def make_path_following_primitive(seed=0):
    """The 'walk' shape with a root path a path-following constraint meets in practice: a gentle curve the candidates
    vary around by a few units (the plain synthetic model's root coefficients are noise of amplitude 100)."""
    data = make_walk_primitive(seed=seed)
    NB, D = int(data["n_basis_spatial"]), int(data["n_dim_spatial"])
    mean = np.array(data["mean_spatial_vector"]).reshape(NB, D)
    eig = np.array(data["eigen_vectors_spatial"]).reshape(-1, NB, D)
    x = np.linspace(0.0, 1.0, NB)
    mean[:, 0], mean[:, 1], mean[:, 2] = 160.0 * x, 90.0 + 2.0 * np.sin(6.0 * x), 40.0 * np.sin(2.0 * x)
    eig[:, :, :3] *= 0.03
    data["mean_spatial_vector"] = mean.reshape(-1).tolist()
    data["eigen_vectors_spatial"] = eig.reshape(len(eig), -1).tolist()
    return data


def make_tiny_primitive(seed=1, **kw):
    """Tiny case the pure-Python oracle loops finish instantly: L=3, F=12, NB=7, D=7, K=2."""
    return make_primitive(seed=seed, n_components=3, n_frames=12, n_basis=7, n_dim=7, n_gmm=2,
                          name="tiny", **kw)


def make_graph_primitives(n_primitives=16, seed=100):
    """BASELINE.json config 3: ~16 primitives with (L, F, K) drawn per primitive, D=79."""
    out = []
    for p in range(n_primitives):
        rng = np.random.default_rng(seed + p)
        L = int(rng.integers(12, 41))
        F = int(rng.integers(40, 161))
        K = int(rng.integers(1, 9))
        out.append(make_primitive(seed=seed + p, n_components=L, n_frames=F, n_dim=79, n_gmm=K,
                                  name="prim_%02d" % p))
    return out


def to_mgrd_v3_json(data):
    """Re-express a legacy dict in the v3 ``sspm/tspm/gmm`` layout
    (reference morphablegraphs/motion_model/motion_primitive_wrapper.py:87-115)."""
    F = int(data["n_canonical_frames"])
    return {
        "name": data.get("name", ""),
        "sspm": {"eigen": data["eigen_vectors_spatial"], "mean": data["mean_spatial_vector"],
                 "n_coeffs": data["n_basis_spatial"], "n_dims": data["n_dim_spatial"],
                 "knots": data["b_spline_knots_spatial"],
                 "animated_joints": data.get("animated_joints", [])},
        "tspm": {"knots": [0.0, 0.0, 0.0, 0.0, float(F - 1), float(F - 1), float(F - 1), float(F - 1)]},
        "gmm": {"covars": data["gmm_covars"], "means": data["gmm_means"], "weights": data["gmm_weights"]},
    }


def write_graph_zip(path, actions, transitions=None, start_node=None, format_version=4.0, cluster_trees=None, stats=None,
                    node_stats=None, node_stats_prefixed=False):
    """A graph zip in the reference's layout (utilities/zip_io.py:37-233): ``actions`` = {action: {"primitives":
    {name: legacy or v3 dict}, "info": meta_information dict}}; ``cluster_trees`` = {(action, name): samples};
    ``node_stats`` = {(action, name): dict} written as '<stem>.stats' where ZipReader looks for it
    ('elementary_action_<action>/<stem>.stats', zip_io.py:196) or, with node_stats_prefixed, next to the model file;
    format_version < 2 puts the actions at the top level (zip_io.py:215-233)."""
    import json
    import zipfile
    graph_def = {"formatVersion": format_version, "transitions": transitions or {}}
    if start_node is not None:
        graph_def["startNode"] = list(start_node)
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("graph_definition.json", json.dumps(graph_def))
        z.writestr("skeleton.json", json.dumps({"name": "synthetic", "animated_joints": []}))
        for action, desc in actions.items():
            base = ("elementary_action_models/" if format_version >= 2.0 else "") + "elementary_action_%s/" % action
            info = dict(desc.get("info", {}))
            if stats is not None and action in stats:
                info["stats"] = stats[action]
            z.writestr(base + "meta_information.json", json.dumps(info))
            for name, mm in desc["primitives"].items():
                z.writestr(base + "%s_%s_quaternion_mm.json" % (action, name), json.dumps(mm))
                if node_stats and (action, name) in node_stats:
                    where = base if node_stats_prefixed else "elementary_action_%s/" % action
                    z.writestr(where + "%s_%s.stats" % (action, name), json.dumps(node_stats[(action, name)]))
                if cluster_trees and (action, name) in cluster_trees:
                    tree = {"data": np.asarray(cluster_trees[(action, name)]).tolist(), "features": [], "options": {}, "root": {}}
                    z.writestr(base + "%s_%s_quaternion_cluster_tree.json" % (action, name), json.dumps(tree))


def make_skeleton(n_animated=19):
    """A humanoid with `n_animated` animated joints (pose vector 3 + 4 n: 79 channels for 19 joints, the layout
    of the 'walk' primitive) plus non-animated end sites, in the shape anim_utils' skeleton.json gives:
    (name, parent, offset) with parents before children, and the animated joints in pose-vector order."""
    joints = [("Hips", None, (0.0, 0.0, 0.0)),
              ("Spine", "Hips", (0.0, 10.0, 0.5)), ("Spine1", "Spine", (0.0, 12.0, 0.0)), ("Neck", "Spine1", (0.0, 14.0, 0.5)),
              ("Head", "Neck", (0.0, 8.0, 1.0)), ("Head_EndSite", "Head", (0.0, 10.0, 0.0)),
              ("LeftShoulder", "Spine1", (6.0, 11.0, 0.0)), ("LeftArm", "LeftShoulder", (9.0, 0.0, 0.0)),
              ("LeftForeArm", "LeftArm", (27.0, 0.0, 0.0)), ("LeftHand", "LeftForeArm", (25.0, 0.0, 0.0)),
              ("LeftHand_EndSite", "LeftHand", (8.0, 0.0, 0.0)),
              ("RightShoulder", "Spine1", (-6.0, 11.0, 0.0)), ("RightArm", "RightShoulder", (-9.0, 0.0, 0.0)),
              ("RightForeArm", "RightArm", (-27.0, 0.0, 0.0)), ("RightHand", "RightForeArm", (-25.0, 0.0, 0.0)),
              ("RightHand_EndSite", "RightHand", (-8.0, 0.0, 0.0)),
              ("LeftUpLeg", "Hips", (9.0, -2.0, 0.0)), ("LeftLeg", "LeftUpLeg", (0.0, -42.0, 0.0)),
              ("LeftFoot", "LeftLeg", (0.0, -40.0, 0.0)), ("LeftToeBase", "LeftFoot", (0.0, -6.0, 12.0)),
              ("RightUpLeg", "Hips", (-9.0, -2.0, 0.0)), ("RightLeg", "RightUpLeg", (0.0, -42.0, 0.0)),
              ("RightFoot", "RightLeg", (0.0, -40.0, 0.0)), ("RightToeBase", "RightFoot", (0.0, -6.0, 12.0))]
    animated = [j[0] for j in joints if not j[0].endswith("EndSite")]   # 21 candidates
    if n_animated <= 19:                                                  # the toes are the first to go
        animated = [n for n in animated if not n.endswith("ToeBase")]
    return joints, animated[:n_animated]
