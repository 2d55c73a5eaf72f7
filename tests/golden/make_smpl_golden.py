"""Pins the SMPL half of the oracle against the REAL thing -- the reference's
models/smpl_official.py:12-41 on top of smplx -- wherever smplx is installed.  ONE command:

    python tests/golden/make_smpl_golden.py                      # synthetic SMPL-shaped models: needs smplx only, no licensed asset
    HPS_SMPL_DIR=/path/to/model_files/smpl python tests/golden/make_smpl_golden.py      # ... and the licensed SMPL_{NEUTRAL,MALE,FEMALE}.pkl

and writes tests/golden/smpl_vectors.npz, which tests/test_oracle_smpl.py (oracle vs fixture, CPU) and tests/test_gpu_smpl.py (HIP
path vs fixture) pick up; without the file those tests skip ("smplx fixtures absent").

smplx is NOT installed in the build container (`pip download smplx==0.1.26`: no index), so there this script stops at a clear
message with exit status 3 and the SMPL arithmetic stays "parity unpinned" (DESIGN.md section 2).  smplx is not vendored and not
stubbed: what is written is data produced by the imported reference class running on the installed package.

How it can pin without the licensed model: smplx.SMPL reads a pickled dict (v_template, shapedirs, posedirs, J_regressor, weights,
kintree_table, f).  ``write_model_pkl`` writes this repository's seeded synthetic models (smpl_data.synthetic_smpl_model(seed):
seed 0 = "neutral", 1 = "male", 2 = "female", the models every SMPL test here uses) in that format into a temporary directory as
SMPL_<GENDER>.pkl, and the reference's SMPL class is constructed on it exactly as run_predict.py:61-65 / run_evaluate.py:40-49 do.
The synthetic fixtures are complete (every vertex); for the licensed models only 512 seeded vertex indices, all 90 joints and a
float64 checksum of all vertices are stored, so the file never contains the licensed template.

Cases per model (seeded inputs are stored with the outputs):
  rotmat   8 random poses as rotation matrices, pose2rot=False        (predict/...:112-115, sampling_utils.py:182-185)
  aa       the same poses as axis-angle, pose2rot=True                (evaluate/...:93-101)
  tpose    betas only, pose from the zero module parameters           (predict/...:136, evaluate/...:131)
  transl   the rotmat poses with a translation                        (data/pw3d_preprocess.py:161-169)
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("HPS_REFERENCE_DIR", "/root/reference")
OUT_PATH = os.path.join(HERE, "smpl_vectors.npz")
GENDER_SEEDS = {"neutral": 0, "male": 1, "female": 2}
N_POSES = 8
REAL_VERTEX_SUBSET = 512

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_model_pkl(model, path):
    """A model dict of smpl_data (synthetic_smpl_model / load_smpl_pkl layout) as the pickle smplx.SMPL.__init__ reads
    (smplx/body_models.py: ``Struct(**pickle.load(f, encoding='latin1'))`` -> .v_template (V,3), .shapedirs (V,3,B), .posedirs
    (V,3,207), .J_regressor (24,V), .weights (V,24), .kintree_table (2,24), .f faces).  Plain numpy arrays: no chumpy, no scipy."""
    V = np.asarray(model["v_template"]).shape[0]
    faces = np.stack([np.arange(V - 2), np.arange(1, V - 1), np.arange(2, V)], axis=1).astype(np.uint32)   # never used by forward()
    kt = np.asarray(model["kintree_table"]).astype(np.int64).copy()
    kt[0, 0] = 2 ** 32 - 1                               # the released files mark the root's parent this way; smplx sets parents[0] = -1
    data = {"v_template": np.asarray(model["v_template"], np.float64), "shapedirs": np.asarray(model["shapedirs"], np.float64),
            "posedirs": np.asarray(model["posedirs"], np.float64), "J_regressor": np.asarray(model["J_regressor"], np.float64),
            "weights": np.asarray(model["weights"], np.float64), "kintree_table": kt, "f": faces}
    with open(path, "wb") as f:
        pickle.dump(data, f, protocol=2)
    return path


def seeded_inputs(seed):
    """betas (8,10), axis-angle poses (8,24,3) (|angle| up to ~1.5 rad), translations (8,3): float32, torch-free recipe."""
    rs = np.random.RandomState(1000 + seed)
    betas = rs.normal(0.0, 1.0, size=(N_POSES, 10)).astype(np.float32)
    aa = (rs.normal(0.0, 0.5, size=(N_POSES, 24, 3))).astype(np.float32)
    aa[0] = 0.0                                         # one mesh in the rest pose
    transl = rs.normal(0.0, 0.5, size=(N_POSES, 3)).astype(np.float32)
    return betas, aa, transl


def _stop(msg, code=3):
    sys.stderr.write("make_smpl_golden.py: %s\n" % msg)
    sys.exit(code)


def main():
    try:
        import smplx  # noqa: F401
        from smplx.lbs import batch_rodrigues
    except ImportError:
        _stop("smplx not installed -- the SMPL arithmetic of the oracle stays 'parity unpinned' here.  Run this script where "
              "`pip install smplx==0.1.26` is possible (the reference's requirements.txt:10); nothing was written.")
    import torch
    if not os.path.isfile(os.path.join(REF, "models", "smpl_official.py")):
        _stop("the reference checkout is not at %s (set HPS_REFERENCE_DIR); nothing was written." % REF)
    from hierarchicalprobabilistic3dhuman_amd import smpl_data
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)                                       # configs/paths.py names the extra joint regressors relative to the checkout
    try:
        from models.smpl_official import SMPL as RefSMPL
        out = {}
        tmp = tempfile.mkdtemp(prefix="hps_smpl_pkl_")
        jobs = []
        for gender, seed in GENDER_SEEDS.items():
            write_model_pkl(smpl_data.synthetic_smpl_model(seed), os.path.join(tmp, "SMPL_%s.pkl" % gender.upper()))
            jobs.append(("syn", gender, tmp, seed, None))
        real_dir = os.environ.get("HPS_SMPL_DIR")
        if real_dir:
            for gender, seed in GENDER_SEEDS.items():
                if os.path.isfile(os.path.join(real_dir, "SMPL_%s.pkl" % gender.upper())):
                    jobs.append(("real", gender, real_dir, 10 + seed, REAL_VERTEX_SUBSET))
                else:
                    sys.stderr.write("make_smpl_golden.py: %s has no SMPL_%s.pkl -- skipped\n" % (real_dir, gender.upper()))
        else:
            sys.stderr.write("make_smpl_golden.py: HPS_SMPL_DIR not set -- synthetic models only (the licensed SMPL files pin "
                             "BASELINE configs[3]'s gendered models as well)\n")
        for tag, gender, model_dir, seed, subset in jobs:
            smpl = RefSMPL(model_dir, batch_size=1, gender=gender, num_betas=10)        # run_predict.py:61-65
            betas, aa, transl = (torch.from_numpy(a) for a in seeded_inputs(seed))
            R = batch_rodrigues(aa.reshape(-1, 3)).reshape(N_POSES, 24, 3, 3)
            key = "%s_%s_" % (tag, gender)
            sel = None
            if subset is not None:
                sel = np.sort(np.random.RandomState(77).choice(smpl.v_template.shape[0], subset, replace=False))
                out[key + "vertex_ids"] = sel.astype(np.int64)
            out[key + "betas"], out[key + "aa"], out[key + "transl"] = betas.numpy(), aa.numpy(), transl.numpy()
            out[key + "rotmats"] = R.numpy()                                            # smplx's own Rodrigues: pins the oracle's too
            out[key + "parents"] = smpl.parents.numpy().astype(np.int64)
            with torch.no_grad():
                cases = {
                    "rotmat": smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False),
                    "aa": smpl(body_pose=aa[:, 1:].reshape(N_POSES, 69), global_orient=aa[:, 0], betas=betas),
                    "tpose": smpl(betas=betas[:1]),
                    "transl": smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, transl=transl, pose2rot=False),
                }
            for name, o in cases.items():
                v = o.vertices.numpy()
                out[key + name + "_joints"] = o.joints.numpy()
                out[key + name + "_verts_sum64"] = np.abs(v.astype(np.float64)).sum(axis=(1, 2))
                out[key + name + "_verts"] = v if sel is None else v[:, sel]
        out["smplx_version"] = np.array(getattr(smplx, "__version__", "unknown"))
        np.savez_compressed(OUT_PATH, **out)
        print("wrote %s: %d arrays (%s)" % (OUT_PATH, len(out), ", ".join("%s/%s" % (t, g) for t, g, *_ in jobs)))
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
