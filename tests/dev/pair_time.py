"""A/B on one box: hps_smpl_pose_prep / hps_smpl_joints (round 5) against their first generations in the dev library
(hps_dev_smpl_pose_prep_v1 / hps_dev_smpl_joints_v1), alone, interleaved rounds, median of HIP-event times per launch.

    python tests/dev/pair_time.py [M]        # default 6528 meshes (BASELINE configs[1]: 64 x (100 + 2))
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import _capi, smpl_data  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 6528
    dev = torch.device("cuda:0")
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    g = torch.Generator().manual_seed(0)
    P = _capi.ptr
    J, V = 24, smpl.num_verts
    R = torch.linalg.qr(torch.randn(M, J, 3, 3, generator=g))[0].contiguous().to(dev)
    betas = torch.randn(M, 10, generator=g).to(dev)
    mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)
    xt = torch.empty(smpl._kp, mp, device=dev)
    a = torch.empty(M, J, 12, device=dev)
    jp = torch.empty(M, J, 3, device=dev)
    verts = torch.randn(M, V, 3, generator=g).to(dev)
    joints = torch.empty(M, J + smpl._n_joint_rows, 3, device=dev)
    glob, body = R[:, 0].contiguous(), R[:, 1:].contiguous()

    def prep(name):
        _capi.call(name, P(glob), P(body), 1, P(betas), 10, P(smpl._j_template), P(smpl._j_shapedirs), _capi.iptr(smpl._parents_i32),
                   _capi.iptr(smpl._depth_i32), J, P(xt), smpl._kp, mp, P(a), P(jp), None, M, _capi.stream())

    def jnt(name):
        _capi.call(name, P(verts), P(jp), _capi.iptr(smpl._csr_ptr), _capi.iptr(smpl._csr_col), P(smpl._csr_val), smpl._n_joint_rows, J, None,
                   P(joints), M, V, _capi.stream())

    picked = verts[:, (smpl._pick_slot >= 0).nonzero().flatten()].contiguous()       # what hps_smpl_mesh_fused_picks leaves beside the vertices

    def jnt_picked():
        _capi.call("hps_smpl_joints", P(picked), P(jp), _capi.iptr(smpl._csr_ptr), _capi.iptr(smpl._csr_slot), P(smpl._csr_val), smpl._n_joint_rows, J, None,
                   P(joints), M, smpl._n_picked, _capi.stream())

    arms = {"pose_prep v1": lambda: prep("hps_dev_smpl_pose_prep_v1"), "pose_prep r5": lambda: prep("hps_smpl_pose_prep"),
            "joints v1": lambda: jnt("hps_dev_smpl_joints_v1"), "joints r5": lambda: jnt("hps_smpl_joints"),
            "joints r5 on the mesh kernel's compact side output": jnt_picked}
    times = {k: [] for k in arms}
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # > the 256 MB Infinity Cache: the vertices come from HBM, as in the step
    with _capi.dev_library():
        for rnd in range(12):
            for k, fn in arms.items():
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[k].append(e0.elapsed_time(e1) * 1e3)
    for k, v in times.items():
        v.sort()
        print("%-52s M = %d: median %7.1f us  (min %7.1f, max %7.1f)" % (k, M, v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    main()
