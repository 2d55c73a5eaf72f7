"""Runs the SMPL forward (fused mesh kernel by default, `unfused` for blend + LBS) on M meshes a few times, for
rocprofv3 --pmc / --kernel-trace (tools/mesh_pmc.sh).  usage: mesh_one.py [M] [fused|unfused] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import smpl_data
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6528
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
smpl.fused_mesh = mode == "fused"
g = torch.Generator().manual_seed(5)
betas = torch.randn(M, 10, generator=g).to(dev)
pose = (torch.randn(M, 72, generator=g) * 0.5).to(dev)
for _ in range(iters):
    smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])
torch.cuda.synchronize()
