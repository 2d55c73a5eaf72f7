"""SMPL model parameter sets: a deterministic synthetic SMPL-shaped model, a loader for the
licensed ``SMPL_*.pkl`` files, and the three extra joint regressors.

The reference loads ``model_files/smpl/SMPL_{NEUTRAL,MALE,FEMALE}.pkl`` through smplx
(models/smpl_official.py:15-16, run_predict.py:61-64) and the regressors
``J_regressor_extra.npy`` / ``cocoplus_regressor.npy`` / ``J_regressor_h36m.npy``
(models/smpl_official.py:17-19).  The pkl files are licensed downloads that are not available to
this build, so every measured configuration runs on ``synthetic_smpl_model`` -- same array shapes,
same sparsity structure (<=4 skinning weights per vertex, sparse joint-regressor rows summing to 1),
seeded, float64 like the pkl contents.
"""
import os
import pickle

import numpy as np

from .configs import SMPL_PARENTS, NUM_VERTS, NUM_JOINTS

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# Approximate rest-pose joint centres of a T-posed body in metres (y up).  Only used to give the
# synthetic model a body-like spatial extent; not SMPL data.
_REST_JOINTS = np.array([
    [0.00, -0.22, 0.03], [0.06, -0.31, 0.01], [-0.06, -0.31, 0.01], [0.00, -0.11, -0.01],
    [0.10, -0.69, 0.01], [-0.10, -0.69, 0.01], [0.00, 0.02, 0.02], [0.09, -1.09, -0.03],
    [-0.09, -1.09, -0.03], [0.00, 0.07, 0.05], [0.11, -1.15, 0.09], [-0.11, -1.15, 0.09],
    [0.00, 0.28, -0.01], [0.08, 0.19, 0.00], [-0.08, 0.19, 0.00], [0.00, 0.35, 0.04],
    [0.17, 0.23, -0.01], [-0.17, 0.23, -0.01], [0.43, 0.21, -0.04], [-0.43, 0.21, -0.04],
    [0.68, 0.22, -0.04], [-0.68, 0.22, -0.04], [0.77, 0.21, -0.05], [-0.77, 0.21, -0.05],
], dtype=np.float64)


def synthetic_smpl_model(seed=0, num_betas=10, num_verts=NUM_VERTS, max_influences=4):
    """Seeded SMPL-shaped parameter set (dict of float64 numpy arrays, pkl field names).

    Keys: v_template (V,3), shapedirs (V,3,num_betas), posedirs (V,3,207), J_regressor (24,V),
    weights (V,24), kintree_table (2,24).  Uses numpy's MT19937 ``RandomState`` so the arrays are
    identical on every machine.
    """
    rs = np.random.RandomState(seed)
    parents = np.array(SMPL_PARENTS, dtype=np.int64)
    children = [[c for c in range(NUM_JOINTS) if parents[c] == j] for j in range(NUM_JOINTS)]

    # primary bone of every vertex; torso joints get more surface
    share = np.ones(NUM_JOINTS)
    share[[0, 3, 6, 9]] = 3.0
    share[[1, 2, 4, 5, 13, 14, 16, 17, 18, 19]] = 1.5
    primary = rs.choice(NUM_JOINTS, size=num_verts, p=share / share.sum())
    primary[:NUM_JOINTS] = np.arange(NUM_JOINTS)        # every joint owns at least one vertex

    par = np.where(parents[primary] < 0, primary, parents[primary])
    t = rs.uniform(0.0, 1.0, size=(num_verts, 1))
    v_template = (1 - t) * _REST_JOINTS[primary] + t * _REST_JOINTS[par]
    v_template = v_template + rs.normal(0.0, 0.035, size=(num_verts, 3))

    # skinning weights: 1..max_influences joints from {primary, parent, a child, grandparent}
    weights = np.zeros((num_verts, NUM_JOINTS))
    nnz = rs.randint(1, max_influences + 1, size=num_verts)
    for v in range(num_verts):
        p = int(primary[v])
        cand = [p]
        if parents[p] >= 0:
            cand.append(int(parents[p]))
        if children[p]:
            cand.append(int(children[p][rs.randint(len(children[p]))]))
        if parents[p] >= 0 and parents[parents[p]] >= 0:
            cand.append(int(parents[parents[p]]))
        cand = cand[:int(nnz[v])]
        w = rs.uniform(0.1, 1.0, size=len(cand))
        w[0] += 1.0                                      # primary bone dominates
        weights[v, cand] = w / w.sum()

    # joint regressor: convex combination of ~32 vertices of each bone
    J_regressor = np.zeros((NUM_JOINTS, num_verts))
    for j in range(NUM_JOINTS):
        own = np.nonzero(primary == j)[0]
        pick = rs.choice(own, size=min(32, len(own)), replace=False)
        w = rs.uniform(0.2, 1.0, size=len(pick))
        J_regressor[j, pick] = w / w.sum()

    shapedirs = rs.normal(0.0, 0.012, size=(num_verts, 3, num_betas))
    posedirs = rs.normal(0.0, 0.004, size=(num_verts, 3, (NUM_JOINTS - 1) * 9))
    kintree_table = np.stack([np.where(parents < 0, 2 ** 32 - 1, parents), np.arange(NUM_JOINTS)])
    return {
        "v_template": v_template, "shapedirs": shapedirs, "posedirs": posedirs,
        "J_regressor": J_regressor, "weights": weights, "kintree_table": kintree_table,
        "synthetic": True,
    }


def load_smpl_pkl(path, num_betas=10):
    """Read a (chumpy-free) ``SMPL_*.pkl`` into the same dict layout as ``synthetic_smpl_model``.

    Mirrors what smplx's SMPL.__init__ extracts from the file (reference call:
    models/smpl_official.py:15-16).  Raises FileNotFoundError if the licensed asset is absent.
    """
    if not os.path.isfile(path):
        raise FileNotFoundError(
            "SMPL model file %r not found: the SMPL pkl files are licensed downloads "
            "(reference README.md:45-63); use synthetic_smpl_model() for benchmarks" % path)
    with open(path, "rb") as f:
        raw = pickle.load(f, encoding="latin1")

    def arr(x):
        if hasattr(x, "toarray"):        # scipy sparse J_regressor
            x = x.toarray()
        if hasattr(x, "r"):              # chumpy array
            x = x.r
        return np.asarray(x, dtype=np.float64)

    shapedirs = arr(raw["shapedirs"])[:, :, :num_betas]
    kt = np.asarray(raw["kintree_table"]).astype(np.int64)
    return {
        "v_template": arr(raw["v_template"]), "shapedirs": shapedirs, "posedirs": arr(raw["posedirs"]),
        "J_regressor": arr(raw["J_regressor"]), "weights": arr(raw["weights"]),
        "kintree_table": kt, "synthetic": False,
    }


def resolve_smpl_model(model_path, gender="neutral", num_betas=10):
    """``model_path`` as the reference passes it (a directory holding SMPL_<GENDER>.pkl, a pkl file,
    or an already-loaded dict)."""
    if isinstance(model_path, dict):
        return model_path
    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, "SMPL_%s.pkl" % gender.upper())
    return load_smpl_pkl(model_path, num_betas=num_betas)


def parents_from_kintree(kintree_table):
    parents = np.asarray(kintree_table)[0].astype(np.int64).copy()
    parents[0] = -1
    return parents


def load_extra_joint_regressors(model_files_dir=None):
    """The three dense (K,6890) float64 regressors of models/smpl_official.py:17-19.

    If ``model_files_dir`` holds the reference's .npy files they are read from there; otherwise the
    sparse copy packaged under data/ (255 non-zeros in total) is expanded.
    Returns (extra (9,V), cocoplus (19,V), h36m (17,V)).
    """
    names = ("J_regressor_extra.npy", "cocoplus_regressor.npy", "J_regressor_h36m.npy")
    if model_files_dir is not None and all(os.path.isfile(os.path.join(model_files_dir, n)) for n in names):
        return tuple(np.load(os.path.join(model_files_dir, n)).astype(np.float64) for n in names)
    z = np.load(os.path.join(_DATA_DIR, "extra_joint_regressors.npz"))
    out = []
    for key in ("extra", "cocoplus", "h36m"):
        a = np.zeros(tuple(z[key + "_shape"]), dtype=np.float64)
        a[z[key + "_rows"], z[key + "_cols"]] = z[key + "_vals"]
        out.append(a)
    return tuple(out)
