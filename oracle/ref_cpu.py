"""TEST INFRASTRUCTURE -- fp32 PyTorch-CPU restatement of the reference's per-image inference path.

This module is the parity oracle and the timed CPU baseline.  It is NOT part of the product: only
tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import it.  The product
(hierarchicalprobabilistic3dhuman_amd/) never falls back to it.

Every function follows the reference file:line it cites (paths relative to the reference repo).

Parity status
  * encoder / head / sampler / rotation conversions: PINNED -- checked against the imported
    reference and against golden vectors generated from it (tests/golden/make_golden.py,
    tests/test_oracle_golden.py).
  * SMPL arithmetic (smplx 0.1.26 ``SMPL.forward`` / ``lbs``; requirements.txt:10): the smplx source
    and the SMPL pkl assets are absent from the reference checkout, so this part restates smplx's
    published algorithm (Loper et al. 2015; op list in SURVEY.md section 8 row A11) and is anchored on the
    reference's call sites (models/smpl_official.py:27-41), the joint-layout contract
    (utils/label_conversions.py:17-20) and analytic known-answer tests plus a float64 twin
    (oracle/smpl_np64.py).  PARITY UNPINNED for that part.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# Rotation conversions -- utils/rigid_transform_utils.py
# ----------------------------------------------------------------------------------------------


def rot6d_to_rotmat(x):
    """utils/rigid_transform_utils.py:80-94 (Gram-Schmidt on the two columns of x.view(-1,3,2)).

    The reference calls ``torch.cross(b1, b2)`` without ``dim`` (line 93), which picks the first
    dimension of size 3 and is therefore wrong at exactly B == 3; the cross product meant is along
    dim 1, which is what is restated here (identical for every B != 3).
    """
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def rotmat_to_rot6d(R, stack_columns=False):
    """utils/rigid_transform_utils.py:97-110."""
    if stack_columns:
        return torch.cat([R[:, :, 0], R[:, :, 1]], dim=1)
    return R[:, :, :2].contiguous().view(-1, 6)


def quat_to_rotmat(quat):
    """utils/rigid_transform_utils.py:113-133; quat is (w, x, y, z), re-normalised first."""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def batch_rodrigues(rot_vecs):
    """smplx.lbs.batch_rodrigues (imported by the reference at predict/...:7, evaluate/...:6):
    angle = ||r + 1e-8||, R = I + sin(a) K + (1 - cos(a)) K^2 with K the cross-product matrix of r/angle."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle).unsqueeze(1)
    sin = torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


# ----------------------------------------------------------------------------------------------
# Encoder -- models/resnet.py (ResNet-18, final FC removed, eval-mode BatchNorm)
# ----------------------------------------------------------------------------------------------

def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], training=False, eps=1e-5)


def _basic_block(x, sd, prefix, stride, has_down):
    """models/resnet.py:62-78."""
    out = F.conv2d(x, sd[prefix + '.conv1.weight'], stride=stride, padding=1)
    out = F.relu(_bn(out, sd, prefix + '.bn1'))
    out = F.conv2d(out, sd[prefix + '.conv2.weight'], stride=1, padding=1)
    out = _bn(out, sd, prefix + '.bn2')
    if has_down:
        x = _bn(F.conv2d(x, sd[prefix + '.downsample.0.weight'], stride=stride), sd, prefix + '.downsample.1')
    return F.relu(out + x)


def resnet18_forward(sd, x, prefix='image_encoder'):
    """models/resnet.py:202-217 with layers [2,2,2,2] (:229-237). ``sd`` = state dict of the net."""
    p = prefix
    x = F.conv2d(x, sd[p + '.conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(x, sd, p + '.bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li in range(1, 5):
        stride = 1 if li == 1 else 2
        x = _basic_block(x, sd, '%s.layer%d.0' % (p, li), stride, li != 1)
        x = _basic_block(x, sd, '%s.layer%d.1' % (p, li), 1, False)
    return torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)


# ----------------------------------------------------------------------------------------------
# Head -- models/poseMF_shapeGaussian_net.py
# ----------------------------------------------------------------------------------------------

def all_ancestors(smpl_parents):
    """models/poseMF_shapeGaussian_net.py:14-21: per body joint (0..22) the ancestor list, nearest
    first, in body-joint numbering (SMPL joint index - 1; the root is not a body joint)."""
    anc = {}
    for i in range(1, len(smpl_parents)):
        joint, parent = i - 1, smpl_parents[i] - 1
        anc[joint] = ([parent] + anc[parent]) if parent >= 0 else []
    return [anc[j] for j in range(len(smpl_parents) - 1)]


def head_forward(sd, feats, smpl_parents, delta_i=True, delta_i_weight=1.0, num_betas=10):
    """models/poseMF_shapeGaussian_net.py:95-162 from image features (B,512).

    Returns pose_F, pose_U, pose_S, pose_V, pose_rotmats_mode, (shape_loc, shape_scale), glob, cam.
    """
    B = feats.shape[0]
    ancestors = all_ancestors(smpl_parents)
    nj = len(ancestors)
    x = F.elu(F.linear(feats, sd['fc1.weight'], sd['fc1.bias']))
    shape_params = F.linear(x, sd['fc_shape.weight'], sd['fc_shape.bias'])
    shape_loc = shape_params[:, :num_betas]
    shape_scale = torch.exp(shape_params[:, num_betas:])
    glob = F.linear(x, sd['fc_glob.weight'], sd['fc_glob.bias']) + sd['init_glob']
    cam = F.linear(x, sd['fc_cam.weight'], sd['fc_cam.bias']) + sd['init_cam']
    embed = F.elu(F.linear(torch.cat([feats, shape_params, glob, cam], dim=1),
                           sd['fc_embed.weight'], sd['fc_embed.bias']))

    pose_F = torch.zeros(B, nj, 3, 3)
    pose_U = torch.zeros(B, nj, 3, 3)
    pose_S = torch.zeros(B, nj, 3)
    pose_V = torch.zeros(B, nj, 3, 3)
    U_proper = torch.zeros(B, nj, 3, 3)
    S_proper = torch.zeros(B, nj, 3)
    mode = torch.zeros(B, nj, 3, 3)
    for j in range(nj):
        anc = ancestors[j]
        if anc:
            inp = torch.cat([embed, U_proper[:, anc].reshape(B, -1), S_proper[:, anc].reshape(B, -1),
                             mode[:, anc].reshape(B, -1)], dim=1)
        else:
            inp = embed
        h = F.elu(F.linear(inp, sd['fc_pose.%d.0.weight' % j], sd['fc_pose.%d.0.bias' % j]))
        Fj = F.linear(h, sd['fc_pose.%d.2.weight' % j], sd['fc_pose.%d.2.bias' % j]).view(-1, 3, 3)
        if delta_i:
            Fj = Fj + delta_i_weight * torch.eye(3)[None].expand_as(Fj)
        U, S, V = torch.svd(Fj)                                   # :137 (CPU LAPACK in the reference too)
        detU, detV = torch.det(U), torch.det(V)                   # :139-140
        Up, Sp, Vp = U.clone(), S.clone(), V.clone()
        Up[:, :, 2] *= detU.unsqueeze(-1)                         # :147-150
        Sp[:, 2] *= detU * detV
        Vp[:, :, 2] *= detV.unsqueeze(-1)
        pose_F[:, j], pose_U[:, j], pose_S[:, j], pose_V[:, j] = Fj, U, S, V
        U_proper[:, j], S_proper[:, j] = Up, Sp
        mode[:, j] = torch.matmul(Up, Vp.transpose(-1, -2))       # :152
    return pose_F, pose_U, pose_S, pose_V, mode, (shape_loc, shape_scale), glob, cam


def net_forward(sd, proxy_rep_input, smpl_parents, **kw):
    """models/poseMF_shapeGaussian_net.py:85-162 end to end."""
    return head_forward(sd, resnet18_forward(sd, proxy_rep_input), smpl_parents, **kw)


# ----------------------------------------------------------------------------------------------
# Matrix-Fisher sampling -- utils/sampling_utils.py
# ----------------------------------------------------------------------------------------------

def m_star(b):
    """utils/sampling_utils.py:46 / :125 (evaluated in float64 by numpy, then used as a Python float)."""
    return float(np.exp(-(4 - b) / 2) * ((4 / b) ** 2))


def bingham_sampling(A, num_samples, Omega, Gaussian_std, M_star, oversampling_ratio=8, log=None):
    """utils/sampling_utils.py:48-69. Draws from the global torch CPU generator in the reference's
    order: randn(8N,4) then rand(8N), once per rejection round.  ``log`` (a dict) receives the noise
    of the accepted round and the number of discarded rounds."""
    n_prop = num_samples * oversampling_ratio
    rounds = 0
    while True:
        eps = torch.randn(n_prop, 4).float()                                   # :51
        y = Gaussian_std * eps                                                  # :52
        samples = y / torch.norm(y, dim=1, keepdim=True)                        # :53
        p_bing = torch.exp(-torch.einsum('bn,n,bn->b', samples, A, samples))    # :56
        p_acg = torch.einsum('bn,n,bn->b', samples, Omega, samples) ** (-2)     # :57
        w = torch.rand(n_prop)                                                  # :60
        accept = w < p_bing / (M_star * p_acg)                                  # :61
        if int(accept.sum().item()) >= num_samples:                             # :62-63
            if log is not None:
                log['eps'], log['w'], log['discarded_rounds'] = eps, w, rounds
                log['accept'] = accept
            return samples[accept][:num_samples]                                # :64-65
        rounds += 1


def proper_svd(pose_U, pose_S, pose_V):
    """utils/sampling_utils.py:104-111."""
    detU, detV = torch.det(pose_U), torch.det(pose_V)
    Up, Sp, Vp = pose_U.clone(), pose_S.clone(), pose_V.clone()
    Sp[..., 2] *= detU * detV
    Up[..., :, 2] *= detU.unsqueeze(-1)
    Vp[..., :, 2] *= detV.unsqueeze(-1)
    return Up, Sp, Vp


def pose_matrix_fisher_sampling(pose_U, pose_S, pose_V, num_samples, b=1.5, oversampling_ratio=8,
                                return_noise=False):
    """utils/sampling_utils.py:74-143 on CPU tensors (the ``sample_on_cpu`` route).

    With ``return_noise`` also returns (eps (B,23,8N,4), w (B,23,8N), discarded_rounds (B,23)): the
    noise of the accepted round of every (image, joint) call, for replay through the HIP kernel.
    """
    B, nj = pose_U.shape[:2]
    Up, Sp, Vp = proper_svd(pose_U, pose_S, pose_V)
    A = torch.zeros(B, nj, 4)
    A[:, :, 1] = 2 * (Sp[:, :, 1] + Sp[:, :, 2])                                # :118-121
    A[:, :, 2] = 2 * (Sp[:, :, 0] + Sp[:, :, 2])
    A[:, :, 3] = 2 * (Sp[:, :, 0] + Sp[:, :, 1])
    Omega = torch.ones(B, nj, 4) + 2 * A / b                                    # :123
    std = Omega ** (-0.5)                                                       # :124
    Ms = m_star(b)
    quats = torch.zeros(B, num_samples, nj, 4)
    n_prop = num_samples * oversampling_ratio
    eps_all = torch.zeros(B, nj, n_prop, 4) if return_noise else None
    w_all = torch.zeros(B, nj, n_prop) if return_noise else None
    disc = torch.zeros(B, nj, dtype=torch.int64) if return_noise else None
    for i in range(B):                                                          # :128-137
        for j in range(nj):
            log = {} if return_noise else None
            quats[i, :, j] = bingham_sampling(A[i, j], num_samples, Omega[i, j], std[i, j], Ms,
                                              oversampling_ratio, log)
            if return_noise:
                eps_all[i, j], w_all[i, j], disc[i, j] = log['eps'], log['w'], log['discarded_rounds']
    R = quat_to_rotmat(quats.view(-1, 4)).view(B, num_samples, nj, 3, 3)        # :139
    R = torch.matmul(Up[:, None], torch.matmul(R, Vp[:, None].transpose(-1, -2)))  # :140-141
    if return_noise:
        return R, (eps_all, w_all, disc)
    return R


# ----------------------------------------------------------------------------------------------
# SMPL forward -- models/smpl_official.py:27-41 over smplx 0.1.26 SMPL.forward / lbs (restated)
# ----------------------------------------------------------------------------------------------

class SMPLParams:
    """fp32 CPU tensors of one SMPL model, laid out like smplx's buffers.

    model: dict as returned by smpl_data.synthetic_smpl_model / load_smpl_pkl;
    extra_regressors: (extra (9,V), cocoplus (19,V), h36m (17,V)) float64 arrays
    (models/smpl_official.py:17-25)."""

    def __init__(self, model, extra_regressors, extra_vertex_ids, num_betas=10, dtype=torch.float32):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        self.dtype = dtype
        self.v_template = t(model['v_template'])                                   # (V,3)
        self.shapedirs = t(np.asarray(model['shapedirs'])[:, :, :num_betas])       # (V,3,nb)
        pd = np.asarray(model['posedirs'])
        self.posedirs = t(pd.reshape(-1, pd.shape[-1]).T)                          # (207, 3V)
        self.J_regressor = t(model['J_regressor'])                                 # (24,V)
        self.lbs_weights = t(model['weights'])                                     # (V,24)
        parents = np.asarray(model['kintree_table'])[0].astype(np.int64).copy()
        parents[0] = -1
        self.parents = torch.tensor(parents)
        self.J_regressor_extra, self.J_regressor_cocoplus, self.J_regressor_h36m = (t(a) for a in extra_regressors)
        self.extra_vertex_ids = torch.tensor(list(extra_vertex_ids), dtype=torch.long)
        self.num_betas = num_betas


def _rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform: chain G_i = G_parent(i) . [R_i | J_i - J_parent(i)] in joint-index
    order; returns posed joints (M,24,3) and A = G with the rest pose removed (M,24,4,4)."""
    M, nj = joints.shape[:2]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                   F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1.0)], dim=2).view(M, nj, 4, 4)
    chain = [T[:, 0]]
    for i in range(1, nj):
        chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    jh = F.pad(joints, [0, 0, 0, 1])
    A = G - F.pad(torch.matmul(G, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, A


def smpl_forward(p, betas=None, body_pose=None, global_orient=None, pose2rot=True, transl=None,
                 return_intermediates=False):
    """models/smpl_official.py:27-41 -> smplx SMPL.forward -> lbs.

    Omitted arguments fall back to zero tensors with one row (the module parameters of a
    ``batch_size=1`` SMPL, run_predict.py:61-64).  Returns dict(vertices (M,6890,3), joints (M,90,3), ...).
    """
    dt = p.dtype
    if betas is None:
        betas = torch.zeros(1, p.num_betas, dtype=dt)
    if global_orient is None:
        global_orient = torch.zeros(1, 3, dtype=dt)
    if body_pose is None:
        body_pose = torch.zeros(1, 69, dtype=dt)
    full_pose = torch.cat([global_orient, body_pose], dim=1)
    M = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
    if betas.shape[0] != M:
        betas = betas.expand(int(M / betas.shape[0]), -1)

    V = p.v_template.shape[0]
    v_shaped = p.v_template + torch.einsum('bl,mkl->bmk', betas, p.shapedirs)       # (1)
    J = torch.einsum('bik,ji->bjk', v_shaped, p.J_regressor)                        # (2)
    ident = torch.eye(3, dtype=dt)
    if pose2rot:                                                                    # (3)
        rot_mats = batch_rodrigues(full_pose.reshape(-1, 3)).view(M, -1, 3, 3)
    else:
        rot_mats = full_pose.reshape(M, -1, 3, 3)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(M, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, p.posedirs).view(M, V, 3)
    J_posed, A = _rigid_transform(rot_mats, J, p.parents)                           # (4)
    nj = p.J_regressor.shape[0]
    T = torch.matmul(p.lbs_weights.unsqueeze(0).expand(M, -1, -1), A.view(M, nj, 16)).view(M, V, 4, 4)  # (5)
    v_h = torch.cat([v_posed, torch.ones(M, V, 1, dtype=dt)], dim=2)
    verts = torch.matmul(T, v_h.unsqueeze(-1))[:, :, :3, 0]
    joints = torch.cat([J_posed, verts[:, p.extra_vertex_ids]], dim=1)              # (6) 24 + 21 = 45
    if transl is not None:                                                          # (7)
        joints = joints + transl.unsqueeze(1)
        verts = verts + transl.unsqueeze(1)
    v2j = lambda R: torch.einsum('bik,ji->bjk', verts, R)                           # smpl_official.py:30-32
    all_joints = torch.cat([joints, v2j(p.J_regressor_extra), v2j(p.J_regressor_cocoplus),
                            v2j(p.J_regressor_h36m)], dim=1)                        # :33-34  -> 90
    out = dict(vertices=verts, joints=all_joints, betas=betas, body_pose=body_pose,
               global_orient=global_orient, full_pose=full_pose)
    if return_intermediates:
        out.update(v_shaped=v_shaped, v_posed=v_posed, J=J, A=A, rot_mats=rot_mats, J_posed=J_posed)
    return out


def vertex_uncertainty(verts_samples):
    """utils/sampling_utils.py:189-190 for one image: (N,V,3) -> (V,)."""
    mean_v = verts_samples.mean(dim=0)
    return torch.norm(verts_samples - mean_v, dim=-1).mean(dim=0)


def compute_vertex_uncertainties(p, pose_U, pose_S, pose_V, shape_loc, shape_scale, glob_rotmats, num_samples,
                                 use_mean_shape=False):
    """utils/sampling_utils.py:146-192 (compute_vertex_uncertainties_by_poseMF_shapeGaussian_sampling) for one image
    (:171 asserts B = 1), with this module's SMPL as ``smpl_model``.  Random draws in the reference's order: pose noise
    (:172-177), then -- only if not use_mean_shape -- Normal(loc, scale).sample([N]) (:181).  Pinned by the golden vectors
    a9_* that tests/golden/make_golden.py produced from the reference function with this SMPL injected.
    Returns (unc (V,), vertices (N,V,3), joints (N,90,3))."""
    assert pose_U.shape[0] == pose_S.shape[0] == pose_V.shape[0] == 1
    R = pose_matrix_fisher_sampling(pose_U, pose_S, pose_V, num_samples)               # :172-177
    if use_mean_shape:
        shape = shape_loc.expand(num_samples, -1)                                       # :178-179
    else:
        shape = torch.distributions.Normal(shape_loc, shape_scale).sample([num_samples])[:, 0, :]   # :180-181
    out = smpl_forward(p, body_pose=R[0], global_orient=glob_rotmats.unsqueeze(1).expand(num_samples, -1, -1, -1),
                       betas=shape, pose2rot=False)                                     # :182-185
    verts = out['vertices']
    return vertex_uncertainty(verts), verts, out['joints']                              # :189-192


# ----------------------------------------------------------------------------------------------
# Proxy-representation front end -- models/canny_edge_detector.py, utils/label_conversions.py
# ----------------------------------------------------------------------------------------------

def gaussian_taps(size=5, std=1.0):
    """models/canny_edge_detector.py:23-24: scipy.signal.windows.gaussian(size, std), normalised to sum 1 (float64
    then cast to fp32 exactly like the reference's torch.from_numpy(...).float())."""
    n = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-0.5 * (n / std) ** 2)
    return torch.from_numpy(g / g.sum()).float()


def canny_edge_detector(img, non_max_suppression=True, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=0.2):
    """models/canny_edge_detector.py:104-166 for img (B,C,H,W); returns the same dict of tensors."""
    B, C = img.shape[:2]
    g = gaussian_taps(gaussian_filter_size, gaussian_filter_std)
    pad = gaussian_filter_size // 2
    wh, wv = g.view(1, 1, 1, -1), g.view(1, 1, -1, 1)
    sob = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]])
    blurred_img = torch.zeros_like(img)
    grad_x = torch.zeros(B, 1, *img.shape[2:])
    grad_y = torch.zeros(B, 1, *img.shape[2:])
    for c in range(C):                                                                 # :113-123
        blurred = F.conv2d(F.conv2d(img[:, [c]], wh, padding=(0, pad)), wv, padding=(pad, 0))
        blurred_img[:, [c]] = blurred
        grad_x += F.conv2d(blurred, sob.view(1, 1, 3, 3), padding=1)
        grad_y += F.conv2d(blurred, sob.t().contiguous().view(1, 1, 3, 3), padding=1)
    grad_x, grad_y = grad_x / C, grad_y / C                                            # :126
    mag = (grad_x ** 2 + grad_y ** 2) ** 0.5
    ori = torch.atan2(grad_y, grad_x) * (180.0 / np.pi) + 180.0                        # :128
    ori = torch.round(ori / 45.0) * 45.0
    thr_mag = mag.clone()
    thr_mag[mag < threshold] = 0.0
    out = dict(blurred_img=blurred_img, grad_magnitude=mag, grad_orientation=ori, thresholded_grad_magnitude=thr_mag)
    if non_max_suppression:                                                            # :142-164
        offs = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]     # neighbour of filter_0 .. filter_315
        filt = torch.zeros(8, 1, 3, 3)
        for k, (dy, dx) in enumerate(offs):
            filt[k, 0, 1, 1] = 1.0
            filt[k, 0, 1 + dy, 1 + dx] = -1.0
        directional = F.conv2d(mag, filt, padding=1)
        positive_idx = (ori / 45) % 8
        thin = mag.clone()
        for pos_i in range(4):
            neg_i = pos_i + 4
            oriented = ((positive_idx == pos_i) * 1 + (positive_idx == neg_i) * 1)
            is_max = (torch.stack([directional[:, pos_i], directional[:, neg_i]]).min(dim=0)[0] > 0.0).unsqueeze(1)
            thin[(is_max == 0) * 1 * oriented > 0] = 0.0
        thr_thin = thin.clone()
        thr_thin[thin < threshold] = 0.0
        out.update(thin_edges=thin, thresholded_thin_edges=thr_thin)
    return out


def joints2d_to_gaussian_heatmaps(joints2D, img_wh, std=4):
    """utils/label_conversions.py:105-124: (B,N,2) (u = column, v = row) -> (B,N,img_wh,img_wh)."""
    ii, jj = torch.meshgrid(torch.arange(img_wh), torch.arange(img_wh), indexing="ij")
    ii, jj = ii[None, None].float(), jj[None, None].float()
    u = joints2D[:, :, 0, None, None]
    v = joints2D[:, :, 1, None, None]
    return torch.exp(-(((ii - v) / std) ** 2) / 2 - (((jj - u) / std) ** 2) / 2)


def proxy_representation(rgb, joints2D, joints2D_visib, edge_nms=True, edge_threshold=0.0, edge_std=1.0, edge_size=5,
                         img_wh=256, heatmap_std=4.0):
    """predict/predict_poseMF_shapeGaussian_net.py:88-100: RGB crop (B,3,D,D), joints (B,17,2), visibility (B,17)
    -> proxy representation (B,18,D,D)."""
    edges = canny_edge_detector(rgb, edge_nms, edge_std, edge_size, edge_threshold)
    edge = edges["thresholded_thin_edges"] if edge_nms else edges["thresholded_grad_magnitude"]
    heat = joints2d_to_gaussian_heatmaps(joints2D, img_wh, heatmap_std) * joints2D_visib[:, :, None, None]
    return torch.cat([edge, heat], dim=1).float()


def batch_crop_affine(input_wh, output_wh, rgb, bbox_centres, bbox_heights, bbox_widths, joints2D=None, scale_factor=1.2):
    """utils/image_utils.py:234-372 for given boxes (the inference use, predict/predict_hrnet.py:86-95 and
    predict/predict_poseMF_shapeGaussian_net.py:78-87), CPU tensors.  Pinned by the golden vectors crop*.
    Returns (rgb_cropped (B,3,oh,ow) or None, joints2D_cropped or None)."""
    B = bbox_centres.shape[0]
    iw, ih = float(input_wh[0]), float(input_wh[1])
    ow, oh = float(output_wh[0]), float(output_wh[1])
    h, w = bbox_heights.clone().float(), bbox_widths.clone().float()
    aspect = oh / ow                                                                # :310
    m = h > w * aspect
    w[m] = h[m] / aspect                                                            # :311
    m = h < w * aspect
    h[m] = w[m] * aspect                                                            # :312
    h, w = h * scale_factor, w * scale_factor                                       # :321-322
    A = torch.zeros(B, 2, 3)
    A[:, 0, 0], A[:, 1, 1] = ow / w, oh / h                                         # :331-332
    A[:, 0, 2] = ow * 0.5 - (ow / w) * bbox_centres[:, 1]                           # :334 (centres are (vertical, horizontal))
    A[:, 1, 2] = oh * 0.5 - (oh / h) * bbox_centres[:, 0]
    joints_out = None
    if joints2D is not None:                                                        # :365-369
        homo = torch.cat([joints2D, torch.ones(B, joints2D.shape[1], 1)], dim=-1)
        joints_out = torch.einsum('bij,bkj->bki', A, homo)
    rgb_out = None
    if rgb is not None:
        T = torch.zeros(B, 2, 3)                                                    # :342-346
        T[:, 0, 0], T[:, 1, 1] = w / iw, h / ih
        T[:, 0, 2] = (-A[:, 0, 2] / (ow / w)) / (iw * 0.5) + w / iw - 1
        T[:, 1, 2] = (-A[:, 1, 2] / (oh / h)) / (ih * 0.5) + h / ih - 1
        grid = F.affine_grid(T, size=[B, 1, int(oh), int(ow)], align_corners=False)
        rgb_out = F.grid_sample(rgb, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    return rgb_out, joints_out


def predict_front_end(image, hrnet_model, hrnet_image_size, hrnet_heatmap_size, proxy_size=256, heatmap_std=4.0,
                      edge_nms=True, edge_threshold=0.0, edge_std=1.0, edge_size=5, bbox_scale_factor=1.2,
                      visib_threshold=0.75):
    """predict/predict_poseMF_shapeGaussian_net.py:63-100 with predict/predict_hrnet.py:34-117 (no object detector: the
    whole image is the person box), CPU.  image: (3,H,W) RGB in [0,1]; hrnet_model: callable (1,3,h,w) -> (1,17,hh,hw).
    Only the crops are pinned by reference outputs (predict_hrnet.py needs torchvision and cannot be imported here)."""
    H, W = image.shape[1:]
    in_w, in_h = hrnet_image_size
    centre = torch.tensor([[H * 0.5, W * 0.5]])
    height, width = torch.tensor([float(H)]), torch.tensor([float(W)])
    aspect = float(in_h) / float(in_w)                                              # predict_hrnet.py:80-84
    if height[0] > width[0] * aspect:
        width = height / aspect
    elif height[0] < width[0] * aspect:
        height = width * aspect
    crop, _ = batch_crop_affine((W, H), (in_w, in_h), image[None], centre, height, width, scale_factor=bbox_scale_factor)
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    heat = hrnet_model(((crop[0] - mean) / std)[None])                              # predict_hrnet.py:98-100
    B, K, hh, hw = heat.shape
    confs, flat = torch.max(heat.reshape(B, K, -1), dim=2)                          # predict_hrnet.py:7-31
    kps = torch.zeros(B, K, 2)
    kps[:, :, 0] = flat % hw
    kps[:, :, 1] = torch.floor(flat / float(hw))
    kps = kps * (confs > 0.0)[:, :, None]
    kps = kps * (in_w / hrnet_heatmap_size[0])                                      # :104
    side = torch.tensor([float(in_h)])
    rgb2, j2 = batch_crop_affine((in_w, in_h), (proxy_size, proxy_size), crop, torch.tensor([[in_h * 0.5, in_w * 0.5]]), side,
                                 side.clone(), joints2D=kps, scale_factor=1.0)     # predict/...:74-87
    visib = confs[0] > visib_threshold                                              # :98-99
    visib[[0, 1, 2, 3, 4, 5, 6, 11, 12]] = True
    proxy = proxy_representation(rgb2, j2, visib[None], edge_nms=edge_nms, edge_threshold=edge_threshold, edge_std=edge_std,
                                 edge_size=edge_size, img_wh=proxy_size, heatmap_std=heatmap_std)
    return proxy, dict(hrnet_crop=crop, joints2D=kps, confs=confs, rgb=rgb2, joints2D_cropped=j2, visib=visib)


# ----------------------------------------------------------------------------------------------
# Evaluation metrics -- utils/eval_utils.py, metrics/eval_metrics_tracker.py (numpy, like the reference)
# ----------------------------------------------------------------------------------------------

def compute_similarity_transform(S1, S2):
    """utils/eval_utils.py:11-59 for (N,3) inputs (the transposed=True route)."""
    S1, S2 = S1.T, S2.T
    mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(U.shape[0])
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    return (scale * R.dot(S1) + t).T


def procrustes_analysis_batch(S1, S2):
    """utils/eval_utils.py:62-67."""
    return np.stack([compute_similarity_transform(S1[i], S2[i]) for i in range(S1.shape[0])])


def scale_and_translation_transform_batch(P, T):
    """utils/eval_utils.py:70-89."""
    P_mean = np.mean(P, axis=1, keepdims=True)
    P_trans = P - P_mean
    P_scale = np.sqrt(np.sum(P_trans ** 2, axis=(1, 2), keepdims=True) / P.shape[1])
    T_mean = np.mean(T, axis=1, keepdims=True)
    T_scale = np.sqrt(np.sum((T - T_mean) ** 2, axis=(1, 2), keepdims=True) / T.shape[1])
    return P_trans / P_scale * T_scale + T_mean


def metric_sums(pred, target, metrics):
    """Sums that metrics/eval_metrics_tracker.py:89-269 accumulates for one update_per_batch call (numpy arrays):
    3D metrics and their *_samples_min forms.  Returns {metric: sum of per-point errors}."""
    align = {"": lambda p, t: p, "-SC": scale_and_translation_transform_batch, "-PA": procrustes_analysis_batch}
    keys = {"PVE": "verts", "MPJPE": "joints3D", "PVE-T": "reposed_verts"}
    out = {}
    for m in metrics:
        smin = m.endswith("_samples_min")
        base = m[:-len("_samples_min")] if smin else m
        sfx = "-SC" if base.endswith("-SC") else ("-PA" if base.endswith("-PA") else "")
        key = keys[base[:len(base) - len(sfx)]]
        if smin:
            p = pred[key + "_samples"]
            t = np.tile(target[key], (p.shape[0], 1, 1))
            e = np.linalg.norm(align[sfx](p, t) - t, axis=-1)
            out[m] = float(np.sum(e[np.argmin(np.mean(e, axis=-1))]))
        else:
            e = np.linalg.norm(align[sfx](pred[key], target[key]) - target[key], axis=-1)
            out[m] = float(np.sum(e))
    return out


def heatmaps_to_joints2d(heat, eps=1e-6):
    """utils/label_conversions.py:127-155."""
    N, K, H, W = heat.shape
    mx, idx = torch.max(heat.reshape(N, K, -1), dim=-1)
    j = torch.zeros(N, K, 2)
    j[:, :, 0] = idx % W
    j[:, :, 1] = torch.floor(idx / float(W))
    vis = mx > eps
    j[torch.logical_not(vis)] = -1
    return j, vis


def joints2d_error_sorted(verts_samples, joints_samples, heat, cam_wp, coco_map):
    """utils/sampling_utils.py:195-233 with the pytorch3d flip written as diag(1,-1,-1); returns (sorted verts, order).
    Pinned by the reference function's own output (tests/golden: rank_order, produced with pytorch3d's Rodrigues closed form
    injected -- its pi rotation differs from diag(1,-1,-1) by sin(pi) = -8.7e-8 in two entries, i.e. 1e-5 pixels).  Equal
    errors keep their sample order (torch's CPU sort is stable; made explicit here)."""
    jc = joints_samples[:, coco_map, :] * torch.tensor([1.0, -1.0, -1.0])
    proj = cam_wp[:, None, [0]] * (jc[:, :, :2] + cam_wp[:, None, 1:])                 # utils/cam_utils.py:9-16
    proj = (proj + 1) * (heat.shape[-1] / 2.0)                                       # utils/joints2d_utils.py:5-10
    in_j, in_vis = heatmaps_to_joints2d(heat)
    l2 = torch.norm(proj[:, in_vis[0], :] - in_j[:, in_vis[0], :], dim=-1)
    order = torch.sort(l2.max(dim=-1)[0], descending=False, stable=True)[1]
    return verts_samples[order], order


def rotmat_to_axis_angle64(R):
    """SO(3) log map in float64: what cv2.Rodrigues(3x3) does for the target flip (utils/rigid_transform_utils.py:48-56).
    cv2 (opencv-python, unpinned in requirements.txt:1) is a third-party dependency absent from this image; this restates the
    published algorithm of OpenCV's calib3d `Rodrigues` for a matrix input step by step: project onto SO(3) by SVD (R = U V^T);
    r = (R21 - R12, R02 - R20, R10 - R01), s = |r| / 2 = sin(theta), c = (tr R - 1) / 2, theta = acos(c); for s >= 1e-5 the
    vector is r * theta / (2 s); below it theta is 0 (c > 0: zero vector) or pi, where the axis comes from the DIAGONAL of
    (R + I) / 2 = n n^T with the signs of n_y, n_z taken from R01, R02 (and R12 when n_x is the smallest component).
    Pinned against scipy.spatial.transform.Rotation (an independent implementation of the same map) in
    tests/test_oracle_golden.py, including R = I (whose flipped matrix is a rotation by exactly pi) and angles -> pi."""
    R = np.asarray(R, np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(float(r @ r) * 0.25)
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        x = np.sqrt(max((R[0, 0] + 1.0) * 0.5, 0.0))
        y = np.sqrt(max((R[1, 1] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        z = np.sqrt(max((R[2, 2] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and ((R[1, 2] > 0) != (y * z > 0)):
            z = -z
        n = np.array([x, y, z])
        return n * (theta / np.sqrt(float(n @ n)))
    return r * (theta / (2.0 * s))


def evaluate_frames(sd, smpl_neutral, smpl_by_gender, smpl_parents, frames, metrics, num_samples, edge_nms=True,
                    edge_threshold=0.0):
    """evaluate/evaluate_poseMF_shapeGaussian_net.py:64-234 for the 3D metrics, one frame at a time like the reference's
    batch-1 DataLoader; ``frames`` yields dicts with image (1,3,D,D), heatmaps (1,17,D,D), pose (1,72), shape (1,10),
    gender.  Returns the final metrics dict (metrics/eval_metrics_tracker.py:332-362)."""
    h36m_j14 = [73 + i for i in [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]]       # label_conversions.py:18-20
    sums = {m: 0.0 for m in metrics}
    total = 0
    flip = np.diag([1.0, -1.0, -1.0])
    # the reference iterates a torch DataLoader (:35-40, :64): creating its iterator draws the workers' base seed from
    # the global CPU generator (one int64 random_()), which shifts every later sample of the run
    torch.empty((), dtype=torch.int64).random_()
    for fr in frames:
        edges = canny_edge_detector(fr["image"], edge_nms, 1.0, 5, edge_threshold)
        edge = edges["thresholded_thin_edges"] if edge_nms else edges["thresholded_grad_magnitude"]
        proxy = torch.cat([edge, fr["heatmaps"]], dim=1)
        pose, shape = fr["pose"].clone(), fr["shape"]
        Rg = batch_rodrigues(pose[:, :3])[0].double().numpy()
        pose[:, :3] = torch.from_numpy(rotmat_to_axis_angle64(flip @ Rg)).float()                  # :84-92
        tp = smpl_by_gender[fr["gender"]]
        t_out = smpl_forward(tp, betas=shape, body_pose=pose[:, 3:], global_orient=pose[:, :3])
        t_rest = smpl_forward(tp, betas=shape)
        pose_F, U, S, V, mode, (loc, scale), glob, cam = net_forward(sd, proxy, smpl_parents)
        glob_R = rot6d_to_rotmat(glob)
        o_mode = smpl_forward(smpl_neutral, body_pose=mode, global_orient=glob_R.unsqueeze(1), betas=loc, pose2rot=False)
        o_rest = smpl_forward(smpl_neutral, betas=loc)
        pred = {"verts": o_mode["vertices"].numpy(), "reposed_verts": o_rest["vertices"].numpy(),
                "joints3D": o_mode["joints"][:, h36m_j14].numpy()}
        target = {"verts": t_out["vertices"].numpy(), "reposed_verts": t_rest["vertices"].numpy(),
                  "joints3D": t_out["joints"][:, h36m_j14].numpy()}
        if any("samples" in m for m in metrics):
            N = num_samples
            R_s = pose_matrix_fisher_sampling(U, S, V, N)                                            # :159-165
            shape_s = torch.distributions.Normal(loc, scale).rsample([N]).transpose(0, 1)            # :166
            o_s = smpl_forward(smpl_neutral, body_pose=R_s[0], global_orient=glob_R.unsqueeze(1).expand(N, -1, -1, -1),
                               betas=shape_s[0], pose2rot=False)
            vs, js = o_s["vertices"].clone(), o_s["joints"][:, h36m_j14].clone()
            vs[0], js[0] = o_mode["vertices"][0], o_mode["joints"][0, h36m_j14]                       # :172-174
            rs = smpl_forward(smpl_neutral, betas=shape_s[0], body_pose=torch.zeros(N, 69), global_orient=torch.zeros(N, 3))["vertices"].clone()
            rs[0] = o_rest["vertices"][0]
            pred.update(verts_samples=vs.numpy(), reposed_verts_samples=rs.numpy(), joints3D_samples=js.numpy())
        for k, v in metric_sums(pred, target, metrics).items():
            sums[k] += v
        total += 1
    return {m: sums[m] / (total * (6890 if "PVE" in m else 14)) for m in metrics}


# ----------------------------------------------------------------------------------------------
# The batched per-image path (predict/predict_poseMF_shapeGaussian_net.py:103-165, looped over B)
# ----------------------------------------------------------------------------------------------

def infer(sd, smpl_params, smpl_parents, proxy_rep_input, num_samples, use_mean_shape=True, feats=None,
          return_noise=False, timings=None):
    """What predict/predict_poseMF_shapeGaussian_net.py:103-165 computes, for a batch of B images,
    identical to looping the reference's B=1 calls (same RNG draw order: image outer, joint inner).
    """
    import time as _time
    _t = [_time.perf_counter()]

    def _mark(stage):           # bench.py's cpu_baseline leg: seconds per stage (``timings`` dict), otherwise a no-op
        if timings is not None:
            now = _time.perf_counter()
            timings[stage] = timings.get(stage, 0.0) + now - _t[0]
            _t[0] = now

    if feats is None:
        feats = resnet18_forward(sd, proxy_rep_input)
    _mark("encoder")
    pose_F, pose_U, pose_S, pose_V, mode, (loc, scale), glob, cam = head_forward(sd, feats, smpl_parents)
    _mark("head+svd")
    B = feats.shape[0]
    glob_R = rot6d_to_rotmat(glob)                                                 # predict:107-110
    out_mode = smpl_forward(smpl_params, body_pose=mode, global_orient=glob_R.unsqueeze(1), betas=loc,
                            pose2rot=False)                                        # predict:112-115
    out_tpose = smpl_forward(smpl_params, betas=loc, global_orient=torch.zeros(B, 3),
                             body_pose=torch.zeros(B, 69))                         # predict:136 per image
    _mark("smpl")
    R = pose_matrix_fisher_sampling(pose_U, pose_S, pose_V, num_samples, return_noise=return_noise)
    _mark("sampler")
    noise = None
    if return_noise:
        R, noise = R
    if use_mean_shape:
        betas_s = loc[:, None, :].expand(B, num_samples, -1)                       # sampling_utils:178-179
    else:
        betas_s = torch.stack([torch.distributions.Normal(loc[i:i + 1], scale[i:i + 1]).sample([num_samples])[:, 0]
                               for i in range(B)])
    out_s = smpl_forward(smpl_params, body_pose=R.reshape(B * num_samples, 23, 3, 3),
                         global_orient=glob_R[:, None, None].expand(B, num_samples, 1, 3, 3).reshape(-1, 1, 3, 3),
                         betas=betas_s.reshape(B * num_samples, -1), pose2rot=False)  # sampling_utils:182-185
    _mark("smpl")
    V = out_s['vertices'].shape[1]
    verts_s = out_s['vertices'].view(B, num_samples, V, 3)
    unc = torch.stack([vertex_uncertainty(verts_s[i]) for i in range(B)])
    _mark("uncertainty")
    res = dict(pose_F=pose_F, pose_U=pose_U, pose_S=pose_S, pose_V=pose_V, pose_rotmats_mode=mode,
               shape_loc=loc, shape_scale=scale, glob=glob, cam=cam, glob_rotmats=glob_R, feats=feats,
               verts_mode=out_mode['vertices'], joints_mode=out_mode['joints'],
               verts_tpose=out_tpose['vertices'], R_samples=R, verts_samples=verts_s,
               joints_samples=out_s['joints'].view(B, num_samples, -1, 3), betas_samples=betas_s, unc=unc)
    if return_noise:
        res['noise'] = noise
    return res
