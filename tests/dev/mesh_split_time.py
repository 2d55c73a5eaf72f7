"""Dev: the bf16x3 form of the shared-shape mesh kernel against the fp32-MFMA form -- accuracy vs the float64 twin and kernel time
(B = 64, N = 100 = 6 528 meshes).  python tests/dev/mesh_split_time.py [--reps 30] [--ablations]
Ablations (dev library, timing only unless noted): abl1 K loop only, abl2 skinning only, abl3 operand stream without MFMAs, abl6 K loop
without the operand stream, abl8 ... and without barriers, abl5 operands one chunk ahead (valid), abl4 DMA pieces in a burst (valid),
stagN start delay of every CU's second workgroup."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data, _capi   # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL          # noqa: E402
from oracle import ref_cpu as O                                              # noqa: E402
from oracle.smpl_np64 import smpl_forward64                                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--N", type=int, default=100)
ap.add_argument("--only", default=None, help="run one arithmetic only (f32 | bf16x3), no accuracy table: for rocprofv3 --pmc passes")
ap.add_argument("--late", action="store_true", help="A/B: the stage-free barrier behind the first two product groups (ablate 13, valid results)")
ap.add_argument("--epilogue", action="store_true", help="the epilogue-only ablations")
ap.add_argument("--ablations", action="store_true", help="also time the dev library's ablations of the bf16x3 kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = smpl_data.synthetic_smpl_model(0)
extra = smpl_data.load_extra_joint_regressors(None)
smpl = SMPL(model).to(dev)
B, N = a.B, a.N
g = torch.Generator().manual_seed(1)
loc = torch.randn(B, 10, generator=g)
rows = list(range(B)) + list(range(B)) + [b for b in range(B) for _ in range(N)]
M = len(rows)
aa = torch.randn(M, 24, 3, generator=g) * 0.5
aa[B:2 * B] = 0
R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
betas = loc[torch.tensor(rows)]
mesh_row, group_rows = smpl.shared_shape_tables(rows)
args = dict(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False)
sh = (loc.to(dev), mesh_row, group_rows)
out = {}


def run(arith):
    for _ in range(3):
        o = smpl(_shared_shapes=sh, **args)
    torch.cuda.synchronize()
    smpl.lbs_events = []
    for _ in range(a.reps):
        o = smpl(_shared_shapes=sh, **args)
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for _, e0, e1 in smpl.lbs_events)
    smpl.lbs_events = None
    out[arith] = o
    print("%-12s mesh kernel: median %.4f ms  min %.4f  max %.4f" % (arith, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)


if a.only:
    smpl.mesh_arith = a.only
    run(a.only)
    sys.exit(0)
ARITHS = ("f32", "bf16x3", "f32", "bf16x3", "bf16x3-mg2")
ABLATIONS = ("bf16x3-abl1", "bf16x3-abl2", "bf16x3-abl3", "bf16x3-abl6", "bf16x3-abl8", "bf16x3-abl5", "bf16x3-abl4", "bf16x3-stag0", "bf16x3-stag8") if a.ablations else ()
if a.late:
    ARITHS, ABLATIONS = ("bf16x3",), ("bf16x3-abl13", "bf16x3-abl0", "bf16x3-abl5") * 3
if a.epilogue:
    ARITHS, ABLATIONS = ("bf16x3",), ("bf16x3-abl2", "bf16x3-abl10", "bf16x3-abl11", "bf16x3-abl12") * 2
for arith in ARITHS + ABLATIONS:
    smpl.mesh_arith = arith.split("-")[0]
    if "stag" in arith:
        with _capi.dev_library() as lib:
            lib.hps_dev_mesh_split_stagger(int(arith.split("stag")[1]))
            run(arith)
            lib.hps_dev_mesh_split_stagger(-1)
    elif "abl" in arith:
        with _capi.dev_library() as lib:
            lib.hps_dev_mesh_split_ablate(int(arith.split("abl")[1]))
            run(arith)
            lib.hps_dev_mesh_split_ablate(0)
    elif arith.endswith("mg2"):
        with _capi.dev_library() as lib:
            lib.hps_dev_mesh_split_groups(2)
            run(arith)
            lib.hps_dev_mesh_split_groups(0)
    else:
        run(arith)
pick = torch.cat([torch.arange(0, 3), torch.arange(B, B + 3), torch.arange(2 * B, 2 * B + 100), torch.arange(M - 60, M)])
pick = pick[pick < M].unique()
v64, j64 = smpl_forward64(model, extra, configs.SMPLX_EXTRA_VERTEX_IDS, betas[pick].double().numpy(), R[pick].double().numpy())
for arith in [a_ for a_ in dict.fromkeys(ARITHS) if a_ in out]:
    e = np.abs(out[arith].vertices[pick.to(dev)].cpu().numpy() - v64)
    print("%-10s |verts - float64 twin|: max %.3e  mean %.3e  rms %.3e" % (arith, e.max(), e.mean(), np.sqrt((e ** 2).mean())))
if "f32" in out and "bf16x3" in out:
    d = (out["f32"].vertices - out["bf16x3"].vertices).abs()
    print("bf16x3 vs f32: max %.3e mean %.3e; joints max %.3e" % (d.max().item(), d.mean().item(), (out["f32"].joints - out["bf16x3"].joints).abs().max().item()))
for name in ("bf16x3-abl13", "bf16x3-abl5", "bf16x3-abl4", "bf16x3-abl0", "bf16x3-mg2"):          # forms whose results are valid: the product's bits
    if name in out and "bf16x3" in out:
        print("%s == bf16x3 bit for bit: %s" % (name, torch.equal(out[name].vertices, out["bf16x3"].vertices)))
