"""``SMPL`` with the call surface of the reference's models/smpl_official.py (a thin subclass of
smplx.SMPL that appends 9 + 19 + 17 regressed joints -> 90), executed by hand-written gfx950 kernels.

    smpl = SMPL(model_path, batch_size=1, gender='neutral', num_betas=10).to(device)
    out = smpl(body_pose=(M,23,3,3), global_orient=(M,1,3,3), betas=(M,10), pose2rot=False)
    out.vertices (M,6890,3), out.joints (M,90,3)

Kernel sequence per call (all on the current HIP stream, no host synchronisation):
  hps_smpl_pose_prep  Rodrigues, rest joints, forward kinematics, blend-GEMM operand
  hps_smpl_mesh_fused v_template + [betas | pose feature] @ [shapedirs ; posedirs] (fp32 MFMA) skinned in the GEMM
                      epilogue: v_posed never exists in HBM.  ``fused_mesh = False`` selects the unfused pair
                      hps_smpl_blend + hps_smpl_lbs (bit-identical vertices; the definition SURVEY 8(d)'s LBS bytes use)
  hps_smpl_joints     24 kinematic joints + 21 vertex picks + 45 regressed joints
"""
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from . import _capi
from .configs import SMPLX_EXTRA_VERTEX_IDS
from .smpl_data import resolve_smpl_model, load_extra_joint_regressors, parents_from_kintree

SMPLOutput = namedtuple("SMPLOutput", ["vertices", "joints", "full_pose", "betas", "global_orient", "body_pose"])
SMPLOutput.__new__.__defaults__ = (None,) * 6

_LBS_K_CHOICES = (4, 8, 12, 24)


def _round_up(x, m):
    return (x + m - 1) // m * m


class SMPL(nn.Module):
    """models/smpl_official.py:12-41.  ``model_path``: directory holding SMPL_<GENDER>.pkl, a pkl path,
    or a dict of arrays (e.g. ``smpl_data.synthetic_smpl_model()``).  ``model_files_dir`` may point at the
    reference's model_files/ (extra joint regressors); the packaged copies are used otherwise."""

    NUM_JOINTS = 24
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path, batch_size=1, gender="neutral", num_betas=10, model_files_dir=None,
                 dtype=torch.float32, **kwargs):
        super().__init__()
        if dtype != torch.float32:
            raise ValueError("the gfx950 kernels compute in fp32 (the reference never changes dtype)")
        model = resolve_smpl_model(model_path, gender=gender, num_betas=num_betas)
        self.gender = gender
        self.batch_size = batch_size
        self.num_betas = num_betas
        self.dtype = dtype
        self.keep_intermediates = False
        self.lbs_events = None
        self.pad_v_posed = True
        self.fused_mesh = True
        # meshes that share their shape (infer(use_mean_shape=True): every mesh of an image has the image's betas) take the K = 207 form of
        # the fused kernel (hps_smpl_mesh_fused_shared_shape: the shape blend once per image, not once per mesh).  False: the K = 217 form
        # for every call (the bit-level partner of the unfused pair).
        self.shared_shape = True
        # arithmetic of the shared-shape form's pose blend GEMM.  "f32": v_mfma_f32_32x32x2_f32, the reference's own arithmetic type in every
        # instruction (default).  "bf16x3": every fp32 operand as three bf16 pieces, six exact piece products per product on the bf16 matrix
        # pipe, fp32 accumulation -- fp32 accuracy (as close to the float64 twin as "f32"), not the same bits (include/hps.h:
        # hps_smpl_mesh_fused_shared_shape_bf16x3).  Applies where the shared-shape form applies; every other call is "f32".
        self.mesh_arith = "f32"
        self._bsplit = None

        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        v_template = np.asarray(model["v_template"], np.float64)
        shapedirs = np.asarray(model["shapedirs"], np.float64)[:, :, :num_betas]
        posedirs_v3k = np.asarray(model["posedirs"], np.float64)                  # (V,3,207)
        J_regressor = np.asarray(model["J_regressor"], np.float64)
        weights = np.asarray(model["weights"], np.float64)
        parents = parents_from_kintree(model["kintree_table"])
        V, J = v_template.shape[0], J_regressor.shape[0]
        self.num_verts = V

        # ---- buffers with smplx's names / layouts (what a user of the reference can inspect) ----
        self.register_buffer("v_template", f32(v_template))
        self.register_buffer("shapedirs", f32(shapedirs))
        self.register_buffer("posedirs", f32(posedirs_v3k.reshape(-1, posedirs_v3k.shape[-1]).T))   # (207,3V)
        self.register_buffer("J_regressor", f32(J_regressor))
        self.register_buffer("lbs_weights", f32(weights))
        self.register_buffer("parents", torch.tensor(parents, dtype=torch.long))
        extra, cocoplus, h36m = load_extra_joint_regressors(model_files_dir)      # smpl_official.py:17-25
        self.register_buffer("J_regressor_extra", f32(extra))
        self.register_buffer("J_regressor_cocoplus", f32(cocoplus))
        self.register_buffer("J_regressor_h36m", f32(h36m))
        # zero-initialised module parameters of `batch_size` rows, the defaults smplx falls back to
        self.betas = nn.Parameter(torch.zeros(batch_size, num_betas), requires_grad=False)
        self.global_orient = nn.Parameter(torch.zeros(batch_size, 3), requires_grad=False)
        self.body_pose = nn.Parameter(torch.zeros(batch_size, 69), requires_grad=False)

        # ---- kernel-side constants (derived in float64, stored fp32; not part of the state dict) ----
        nb = num_betas
        n_pose = posedirs_v3k.shape[-1]
        self._N = 3 * V
        self._n_pose = n_pose
        self._kp = _round_up(nb + n_pose, 16)
        self._k_used = _round_up(nb + n_pose, 2)          # rows the fused kernel multiplies (the rest of the 16-row padding is zero)
        self._np = _round_up(self._N, 128)
        bmat = np.zeros((self._kp, self._np), np.float64)
        bmat[:nb, :self._N] = shapedirs.reshape(self._N, nb).T                    # row l: d v[n] / d beta_l
        bmat[nb:nb + n_pose, :self._N] = posedirs_v3k.reshape(self._N, n_pose).T
        self.register_buffer("_bmat", f32(bmat), persistent=False)
        # the same matrix with panel-permuted columns for the fused kernel (include/hps.h: hps_smpl_mesh_fused):
        # col(v, c) = (v // 64) * 192 + c * 64 + v % 64
        self._np_fused = -(-V // 64) * 192
        vi = np.arange(V)
        bmat_p = np.zeros((self._kp, self._np_fused), np.float64)
        for c in range(3):
            bmat_p[:, (vi // 64) * 192 + c * 64 + vi % 64] = bmat[:, 3 * vi + c]
        self.register_buffer("_bmat_p", f32(bmat_p), persistent=False)
        self.register_buffer("_v_template_flat", f32(v_template.reshape(-1)), persistent=False)
        # joint regression folded through the linear shape blend: J = J_reg (v_t + S beta)
        self.register_buffer("_j_template", f32(J_regressor @ v_template), persistent=False)                 # (J,3)
        self.register_buffer("_j_shapedirs", f32(np.einsum("jv,vcl->jcl", J_regressor, shapedirs)), persistent=False)
        depth = np.zeros(J, np.int32)
        for j in range(1, J):
            depth[j] = depth[parents[j]] + 1
        self.register_buffer("_parents_i32", torch.tensor(parents, dtype=torch.int32), persistent=False)
        self.register_buffer("_depth_i32", torch.tensor(depth, dtype=torch.int32), persistent=False)
        # compressed skinning weights: K entries per vertex, exact when K >= max non-zeros per row
        nnz = int((weights != 0).sum(1).max())
        K = next(k for k in _LBS_K_CHOICES if k >= nnz)
        order = np.argsort(-(weights != 0).astype(np.int8), axis=1, kind="stable")[:, :K]   # non-zeros first, index order
        w_val = np.take_along_axis(weights, order, axis=1)
        w_idx = np.where(w_val != 0, order, 0)
        self._lbs_k = K
        # the fused mesh kernel exists for K = 4 (any joint count) and K = 8, 12 (24 joints): include/hps.h hps_smpl_mesh_fused;
        # denser skinning weights take the unfused pair (same bits)
        self._fused_ok = K == 4 or (K in (8, 12) and J == 24)
        self.register_buffer("_w_idx", torch.tensor(w_idx, dtype=torch.int32).contiguous(), persistent=False)
        self.register_buffer("_w_val", f32(w_val).contiguous(), persistent=False)
        # joint rows after the 24 kinematic joints: 21 vertex picks (smplx VertexJointSelector), then the
        # extra / cocoplus / h36m regressors (smpl_official.py:30-34) -- one CSR matrix over the vertices
        picks = np.zeros((len(SMPLX_EXTRA_VERTEX_IDS), V))
        picks[np.arange(len(SMPLX_EXTRA_VERTEX_IDS)), SMPLX_EXTRA_VERTEX_IDS] = 1.0
        rows = [picks, extra, cocoplus, h36m]
        dense = np.concatenate(rows, axis=0)
        ptr_, col_, val_ = [0], [], []
        for r in range(dense.shape[0]):
            nz = np.nonzero(dense[r])[0]
            col_.extend(nz.tolist())
            val_.extend(dense[r, nz].tolist())
            ptr_.append(len(col_))
        self._n_joint_rows = dense.shape[0]
        # the distinct vertices those rows read (SMPL: 198 of 6 890) get a slot in a compact array the fused mesh kernel fills beside
        # the vertices (hps_smpl_mesh_fused_picks); the joint kernel then reads that array through the entries' slots
        uniq = sorted(set(col_))
        slot_of = {v: i for i, v in enumerate(uniq)}
        pick_slot = np.full(V, -1, np.int32)
        pick_slot[uniq] = np.arange(len(uniq), dtype=np.int32)
        self._n_picked = len(uniq)
        self.register_buffer("_pick_slot", torch.tensor(pick_slot, dtype=torch.int32), persistent=False)
        self.register_buffer("_csr_slot", torch.tensor([slot_of[v] for v in col_], dtype=torch.int32), persistent=False)
        self.picked_joints = True       # False: gather the regressor vertices from the mesh (the round-4 route; same bits)
        self.register_buffer("_csr_ptr", torch.tensor(ptr_, dtype=torch.int32), persistent=False)
        self.register_buffer("_csr_col", torch.tensor(col_, dtype=torch.int32), persistent=False)
        self.register_buffer("_csr_val", torch.tensor(val_, dtype=torch.float32), persistent=False)

    def shared_shape_tables(self, mesh_rows):
        """The (mesh_row, group_rows) tables of hps_smpl_mesh_fused_shared_shape for meshes whose shapes are rows of a small table:
        ``mesh_rows`` (M,) host integers, mesh m has shape row mesh_rows[m].  Device int32 tensors, padded to the kernel's mesh count."""
        rows = np.asarray(mesh_rows, np.int32).reshape(-1)
        M = rows.shape[0]
        mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)
        full = np.concatenate([rows, np.full(mp - M, rows[-1] if M else 0, np.int32)])
        groups = np.zeros((mp // 32, 3), np.int32)
        for g in range(mp // 32):
            r = full[32 * g:32 * g + 32]
            change = np.nonzero(r[1:] != r[:-1])[0]
            if change.size == 0:
                groups[g] = (r[0], r[0], 32)
            elif change.size == 1:
                groups[g] = (r[0], r[-1], change[0] + 1)
            else:
                groups[g] = (r[0], r[0], -1)
        dev = self.v_template.device
        return torch.from_numpy(full).to(dev), torch.from_numpy(groups.reshape(-1)).to(dev)

    def _blend_matrix_split(self, with_shape_rows=False):
        """The panel-permuted blend matrix as bf16 piece planes in MFMA fragment order (once per model and device): the 207 pose rows
        (shared shapes) or, ``with_shape_rows``, all num_betas + 207 rows (the shape blend inside the GEMM)."""
        dev = self._bmat_p.device
        if self._bsplit is None or self._bsplit[0] != dev:
            self._bsplit = (dev, {})
        cache = self._bsplit[1]
        if with_shape_rows not in cache:
            first, rows = (0, self.num_betas + self._n_pose) if with_shape_rows else (self.num_betas, self._n_pose)
            n = _capi.load(dev=_capi._use_dev).hps_smpl_split_bf16x3_bytes(rows, self._np_fused)
            buf = torch.empty(n, dtype=torch.uint8, device=dev)
            _capi.call("hps_smpl_split_bf16x3", _capi._P(self._bmat_p.data_ptr() + 4 * first * self._np_fused), rows, self._np_fused,
                       self._np_fused, 192, _capi._P(buf.data_ptr()), _capi.stream())
            cache[with_shape_rows] = buf
        return cache[with_shape_rows]

    def _one_shape_tables(self, mp):
        """(mesh_row, group_rows) of the bf16x3 kernel for a call whose meshes do NOT share shapes: one template row (v_template) for all."""
        dev = self.v_template.device
        key = (dev, mp)
        if getattr(self, "_one_shape", (None,))[0] != key:
            groups = torch.tensor([0, 0, 32], dtype=torch.int32).repeat(mp // 32)
            self._one_shape = (key, torch.zeros(mp, dtype=torch.int32, device=dev), groups.to(dev))
        return self._one_shape[1], self._one_shape[2]

    # ------------------------------------------------------------------------------------------
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, pose2rot=True,
                return_verts=True, return_full_pose=False, **kwargs):
        """smplx SMPL.forward semantics as used by the reference (models/smpl_official.py:27-41):
        omitted arguments default to the zero module parameters; betas with fewer rows than the pose
        are expanded; ``pose2rot=False`` takes rotation matrices, otherwise axis-angle."""
        dev = self.v_template.device
        _capi.require_device(self.v_template, "SMPL buffers (call .to('cuda'))")
        J = self.NUM_JOINTS
        global_orient = self.global_orient if global_orient is None else global_orient
        body_pose = self.body_pose if body_pose is None else body_pose
        betas = self.betas if betas is None else betas
        for name, t in (("betas", betas), ("body_pose", body_pose), ("global_orient", global_orient)):
            _capi.require_device(t, name)
        M = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != M:
            betas = betas.expand(int(M / betas.shape[0]), -1) if betas.shape[0] == 1 else betas.repeat(int(M / betas.shape[0]), 1)
        if global_orient.shape[0] != M or body_pose.shape[0] != M:
            raise ValueError("global_orient / body_pose must have the same number of rows")

        g = _capi.f32c(global_orient).reshape(M, -1)
        b = _capi.f32c(body_pose).reshape(M, -1)
        is_rotmat = 0 if pose2rot else 1
        exp_g, exp_b = (9, 9 * (J - 1)) if is_rotmat else (3, 3 * (J - 1))
        if g.shape[1] != exp_g or b.shape[1] != exp_b:
            raise ValueError("pose2rot=%s expects global_orient with %d and body_pose with %d values per mesh, got %d / %d"
                             % (pose2rot, exp_g, exp_b, g.shape[1], b.shape[1]))
        be = _capi.f32c(betas)
        if be.shape[1] != self.num_betas:
            raise ValueError("betas must have %d columns" % self.num_betas)
        tr = None if transl is None else _capi.f32c(transl).reshape(M, 3)

        V, N = self.num_verts, self._N
        mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)                  # padded mesh count of the blend operand
        f32 = dict(device=dev, dtype=torch.float32)
        xt = torch.empty(self._kp, mp, **f32)
        a = torch.empty(M, J, 12, **f32)
        j_posed = torch.empty(M, J, 3, **f32)
        s = _capi.stream()
        P = _capi.ptr
        verts = torch.empty(M, V, 3, **f32)
        joints = torch.empty(M, J + self._n_joint_rows, 3, **f32)
        # the SMPL configuration also gets the regressor vertices as a compact side output of the mesh kernel
        use_picks = (self.fused_mesh and self._fused_ok and self.picked_joints and self._lbs_k == 4 and J == 24 and self._k_used == 218
                     and self._n_picked > 0)
        picked = torch.empty(M, self._n_picked, 3, **f32) if use_picks else None
        # ``_shared_shapes = (shape_betas (R, num_betas), mesh_row, group_rows)``: betas[m] == shape_betas[mesh_row[m]] (the caller's
        # promise -- infer(use_mean_shape=True)): the shape blend once per distinct shape, the GEMM over the pose rows only
        shared = kwargs.get("_shared_shapes") if (self.shared_shape and use_picks and tr is None) else None
        v_shaped = None
        if self.mesh_arith not in ("f32", "bf16x3"):
            raise ValueError("mesh_arith must be 'f32' or 'bf16x3', got %r" % (self.mesh_arith,))
        # bf16x3: the shared-shape form (207 pose rows), or -- no shared shapes, SMPL configuration, no translation -- the same kernel over
        # all 217 rows with v_template as the one "shaped template" (the shape blend inside the GEMM, as in the fp32 K = 217 form)
        split = self.mesh_arith == "bf16x3" and use_picks and tr is None and self.fused_mesh
        split_rows = (self.num_betas, self._n_pose) if shared is not None else (0, self.num_betas + self._n_pose)      # (first row, rows)
        xsplit = bsplit = None
        if shared is not None:
            sb = _capi.f32c(shared[0])
            v_shaped = torch.empty(sb.shape[0], V, 3, **f32)
            _capi.call("hps_smpl_v_shaped", P(sb), self.num_betas, P(self._bmat), self._np, P(self._v_template_flat), P(v_shaped),
                       sb.shape[0], V, s)
        _capi.call("hps_smpl_pose_prep", P(g), P(b), is_rotmat, P(be), self.num_betas, P(self._j_template),
                   P(self._j_shapedirs), _capi.iptr(self._parents_i32), _capi.iptr(self._depth_i32), J, P(xt),
                   self._kp, mp, P(a), P(j_posed), None, M, s)
        if split:
            lib = _capi.load(dev=_capi._use_dev)
            bsplit = self._blend_matrix_split(with_shape_rows=shared is None)
            xsplit = torch.empty(lib.hps_smpl_split_bf16x3_bytes(split_rows[1], mp), dtype=torch.uint8, device=dev)
            _capi.call("hps_smpl_split_bf16x3", _capi._P(xt.data_ptr() + 4 * split_rows[0] * mp), split_rows[1], mp, mp,
                       lib.hps_smpl_split_bf16x3_mesh_tile(), _capi._P(xsplit.data_ptr()), s)
        # InferencePipeline: only the chip-filling mesh kernel(s) run alone; pose prep (before) and the joint regression (after)
        # may share the GPU with the neighbouring batches' encoders
        if kwargs.get("_before_mesh") is not None:
            entry_stream = torch.cuda.current_stream()
            ordered = kwargs["_before_mesh"]()
            now = torch.cuda.current_stream()
            if now != entry_stream:              # the hook moved the mesh kernel to another stream (InferencePipeline.inline_mesh)
                s = _capi.stream()
                # The operands were allocated on the entry stream.  A hook that returns "ordered" promises that its _after_mesh
                # partner makes the entry stream wait for the mesh kernel before anything else is queued there: the allocator hands
                # these blocks back to the entry stream's pool only, so every reuse is ordered behind the kernel.  Without the promise
                # the tensors are registered with the other stream -- which costs an event record ON THAT STREAM for each of them
                # when they are freed (seven markers of ~6 us in front of the next encoder, measured: profiles/r05_experiments.txt).
                if ordered != "ordered":
                    for t in (xt, a, verts, picked, be, g, b, v_shaped, xsplit):
                        if t is not None:
                            t.record_stream(now)
        ev = None
        if self.lbs_events is not None:      # bench.py: HIP events around the mesh kernel launch, on its own stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        trp = P(tr) if tr is not None else None
        v_posed = None
        if self.fused_mesh and self._fused_ok:
            if ev is not None:
                ev[0].record()
            if split:
                rows_t, groups_t = (shared[1], shared[2]) if shared is not None else self._one_shape_tables(mp)
                _capi.call("hps_smpl_mesh_fused_shared_shape_bf16x3", _capi._P(xsplit.data_ptr()), _capi._P(bsplit.data_ptr()),
                           P(v_shaped) if shared is not None else P(self._v_template_flat), _capi.iptr(rows_t), _capi.iptr(groups_t), P(a),
                           _capi.iptr(self._w_idx), P(self._w_val), self._lbs_k, J, P(verts), M, V, split_rows[1], mp,
                           _capi.iptr(self._pick_slot), P(picked), self._n_picked, s)
            elif shared is not None:
                nb = self.num_betas
                _capi.call("hps_smpl_mesh_fused_shared_shape", _capi._P(xt.data_ptr() + 4 * nb * mp), _capi._P(self._bmat_p.data_ptr() + 4 * nb * self._np_fused),
                           P(v_shaped), _capi.iptr(shared[1]), _capi.iptr(shared[2]), P(a), _capi.iptr(self._w_idx), P(self._w_val), self._lbs_k, J,
                           P(verts), M, V, self._k_used - nb, mp, self._np_fused, _capi.iptr(self._pick_slot), P(picked), self._n_picked, s)
            elif use_picks:
                _capi.call("hps_smpl_mesh_fused_picks", P(xt), P(self._bmat_p), P(self._v_template_flat), P(a),
                           _capi.iptr(self._w_idx), P(self._w_val), self._lbs_k, J, trp, P(verts), M, V, self._k_used, mp,
                           self._np_fused, _capi.iptr(self._pick_slot), P(picked), self._n_picked, s)
            else:
                _capi.call("hps_smpl_mesh_fused", P(xt), P(self._bmat_p), P(self._v_template_flat), P(a),
                           _capi.iptr(self._w_idx), P(self._w_val), self._lbs_k, J, trp, P(verts), M, V, self._k_used, mp,
                           self._np_fused, s)
        else:
            ldv = self._np if self.pad_v_posed else N          # row pitch of v_posed in floats (128-byte aligned rows)
            v_posed = torch.empty(M, ldv, **f32)
            _capi.call("hps_smpl_blend", P(xt), P(self._bmat), P(self._v_template_flat), P(v_posed), M, N, self._kp,
                       mp, self._np, ldv, s)
            if ev is not None:
                ev[0].record()
            _capi.call("hps_smpl_lbs", P(v_posed), ldv, P(a), _capi.iptr(self._w_idx), P(self._w_val), self._lbs_k, J,
                       trp, P(verts), M, V, s)
        if ev is not None:
            ev[1].record()
            self.lbs_events.append((M, ev[0], ev[1]))
        if kwargs.get("_after_mesh") is not None:
            # (an event already recorded right behind the mesh kernel -- bench.py's timing event -- is handed to the hook, which may wait
            # on it instead of recording one of its own: an event record costs the queue ~8 us)
            kwargs["_after_mesh"](ev[1] if ev is not None else None)
            s = _capi.stream()                   # (the hook may have switched back to the caller's stream)
        if picked is not None and kwargs.get("_defer_joints") is not None:
            # infer(): the joint regression rides in the launch of the uncertainty pass (hps_joints_and_uncertainty); the caller gets the
            # launch's joint arguments and fills ``joints`` itself
            kwargs["_defer_joints"](dict(picked=picked, j_posed=j_posed, csr_ptr=self._csr_ptr, csr_slot=self._csr_slot, csr_val=self._csr_val,
                                         n_rows=self._n_joint_rows, J=J, transl=tr, joints=joints, M=M, n_picked=self._n_picked))
        elif picked is not None:      # the regressor vertices lie side by side: same rows, same values, same sums
            _capi.call("hps_smpl_joints", P(picked), P(j_posed), _capi.iptr(self._csr_ptr), _capi.iptr(self._csr_slot),
                       P(self._csr_val), self._n_joint_rows, J, P(tr) if tr is not None else None, P(joints), M, self._n_picked, s)
        else:
            _capi.call("hps_smpl_joints", P(verts), P(j_posed), _capi.iptr(self._csr_ptr), _capi.iptr(self._csr_col),
                       P(self._csr_val), self._n_joint_rows, J, P(tr) if tr is not None else None, P(joints), M, V, s)
        full_pose = torch.cat([g, b], dim=1) if return_full_pose else None
        if self.keep_intermediates:                                         # tests / profiling only
            self._last = dict(xt=xt, a=a, j_posed=j_posed)
            if v_posed is not None:
                self._last.update(v_posed=v_posed[:, :N].reshape(M, V, 3), v_posed_raw=v_posed, ldv=ldv)
        return SMPLOutput(vertices=verts if return_verts else None, joints=joints, full_pose=full_pose,
                          betas=betas, global_orient=global_orient, body_pose=body_pose)
