"""Dev: the fused mesh kernel at 4 / 3 / 2 / 1 workgroups per CU (hps_dev_mesh_lds_floor), product and K-loop-only, interleaved medians."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi, smpl_data
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
dev = torch.device("cuda:0")
smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6528
g = torch.Generator().manual_seed(5)
betas = torch.randn(M, 10, generator=g).to(dev)
pose = (torch.randn(M, 72, generator=g) * 0.5).to(dev)
smpl.keep_intermediates = True
smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])
L = smpl._last
P = _capi.ptr
verts = torch.empty(M, 6890, 3, device=dev)
mp = L["xt"].shape[1]
flop = 2.0 * 224 * 3 * 6890 * M
with _capi.dev_library():
    cfgs = [(0, 0), (0, 5), (0, 3), (52 * 1024, 0), (52 * 1024, 5)]
    times = {c: [] for c in cfgs}
    def fn(ab):
        _capi.call("hps_dev_mesh_fused", P(L["xt"]), P(smpl._bmat_p), P(smpl._v_template_flat), P(L["a"]), _capi.iptr(smpl._w_idx),
                   P(smpl._w_val), 4, 24, None, P(verts), M, 6890, smpl._kp, mp, smpl._np_fused, ab, _capi.stream())
    for rnd in range(7):
        for (lds, ab) in cfgs:
            _capi.call("hps_dev_mesh_lds_floor", lds)
            for _ in range(2): fn(ab)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): fn(ab)
            e1.record(); torch.cuda.synchronize()
            times[(lds, ab)].append(e0.elapsed_time(e1) / 8)
    _capi.call("hps_dev_mesh_lds_floor", 0)
    for (lds, ab) in cfgs:
        t = sorted(times[(lds, ab)])[3]
        print("mesh M=%d lds floor %3d KiB (%s) %-12s median %.4f ms = %.1f TF/s" % (M, lds // 1024, {0: "4 WG/CU", 52: "3 WG/CU", 72: "2 WG/CU", 150: "1 WG/CU"}[lds // 1024], {0: "product", 3: "K loop only", 5: "DMA burst"}[ab], t, flop / t / 1e9))
