"""TEST INFRASTRUCTURE -- float64 numpy twin of the SMPL forward, written independently of
oracle/ref_cpu.py (explicit per-joint world transforms and a per-vertex weighted sum of transformed
points rather than blended 4x4 matrices), following Loper et al. 2015 eqs. 2-10 as realised by
smplx 0.1.26 ``lbs`` (the third-party dependency the reference imports at models/smpl_official.py:3-8).
PARITY UNPINNED (no smplx source, no SMPL assets in the reference checkout) -- it exists so that the
fp32 restatement and the HIP kernels can be checked against an independent higher-precision
computation and against the analytic known-answer tests in tests/test_oracle_smpl.py.
"""
import numpy as np


def rodrigues64(r):
    """(K,3) axis-angle -> (K,3,3), smplx convention: angle = ||r + 1e-8||."""
    r = np.asarray(r, np.float64)
    angle = np.linalg.norm(r + 1e-8, axis=1)
    d = r / angle[:, None]
    K = np.zeros((r.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -d[:, 2], d[:, 1]
    K[:, 1, 0], K[:, 1, 2] = d[:, 2], -d[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -d[:, 1], d[:, 0]
    s, c = np.sin(angle)[:, None, None], np.cos(angle)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def smpl_forward64(model, extra_regressors, extra_vertex_ids, betas, rot_mats, transl=None):
    """betas (M,nb), rot_mats (M,24,3,3) float64 -> vertices (M,V,3), joints (M,90,3)."""
    v_template = np.asarray(model['v_template'], np.float64)
    nb = betas.shape[1]
    shapedirs = np.asarray(model['shapedirs'], np.float64)[:, :, :nb]
    posedirs = np.asarray(model['posedirs'], np.float64)            # (V,3,207)
    Jreg = np.asarray(model['J_regressor'], np.float64)
    W = np.asarray(model['weights'], np.float64)
    parents = np.asarray(model['kintree_table'])[0].astype(np.int64).copy()
    parents[0] = -1
    M, V = betas.shape[0], v_template.shape[0]
    verts = np.zeros((M, V, 3))
    joints45 = np.zeros((M, 45, 3))
    for m in range(M):
        v_shaped = v_template + shapedirs @ betas[m]
        J = Jreg @ v_shaped
        pf = (rot_mats[m, 1:] - np.eye(3)).reshape(-1)
        v_posed = v_shaped + posedirs @ pf
        Rw = np.zeros((24, 3, 3))
        tw = np.zeros((24, 3))
        Rw[0], tw[0] = rot_mats[m, 0], J[0]
        for i in range(1, 24):
            pa = parents[i]
            Rw[i] = Rw[pa] @ rot_mats[m, i]
            tw[i] = Rw[pa] @ (J[i] - J[pa]) + tw[pa]
        # vertex = sum_j w_j (Rw_j (v - J_j) + tw_j)
        out = np.zeros((V, 3))
        for j in range(24):
            wj = W[:, j:j + 1]
            if not np.any(wj):
                continue
            out += wj * ((v_posed - J[j]) @ Rw[j].T + tw[j])
        verts[m] = out
        joints45[m, :24] = tw
        joints45[m, 24:] = out[list(extra_vertex_ids)]
    if transl is not None:
        verts = verts + transl[:, None]
        joints45 = joints45 + transl[:, None]
    extra = [np.einsum('jv,mvk->mjk', np.asarray(R, np.float64), verts) for R in extra_regressors]
    return verts, np.concatenate([joints45] + extra, axis=1)
