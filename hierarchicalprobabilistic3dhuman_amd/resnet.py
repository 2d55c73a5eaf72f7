"""ResNet-18 image encoder (final FC removed) with the module tree and state-dict keys of the reference's
models/resnet.py, executed as implicit-GEMM convolutions on the gfx950 fp32 matrix cores.

The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children only hold parameters (so that
``load_state_dict(checkpoint['best_model_state_dict'])`` and default initialisation behave exactly as in
the reference); their own forward is never used.  Inference only: BatchNorm uses running statistics
(the reference runs the encoder under ``model.eval()``, predict/...:55) and is fused, together with the
residual add and ReLU, into the convolution epilogue.
"""
import torch
from torch import nn

from . import _capi


class _ConvBN:
    """Kernel-side form of conv + eval BatchNorm: k-major filter (ceil16(KH*KW*Cin_p), Cout), scale, shift."""

    def __init__(self, conv, bn, cin_pad=None):
        w = conv.weight.detach().float()                      # (Cout, Cin, KH, KW)
        cout, cin, kh, kw = w.shape
        cin_p = cin if cin_pad is None else cin_pad
        wk = torch.zeros(kh, kw, cin_p, cout, device=w.device, dtype=torch.float32)
        wk[:, :, :cin, :] = w.permute(2, 3, 1, 0)
        k_real = kh * kw * cin_p
        k_pad = (k_real + 15) // 16 * 16
        flat = torch.zeros(k_pad, cout, device=w.device, dtype=torch.float32)
        flat[:k_real] = wk.reshape(k_real, cout)
        self.wk = flat.contiguous()
        inv_std = torch.rsqrt(bn.running_var.detach().double() + bn.eps)
        scale = bn.weight.detach().double() * inv_std
        self.scale = scale.float().contiguous()
        self.shift = (bn.bias.detach().double() - bn.running_mean.detach().double() * scale).float().contiguous()
        self.cin_p, self.cout, self.kh, self.kw = cin_p, cout, kh, kw
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        # n-major filter (Cout, KH*KW*Cin) for the v2 kernel (Cin % 32 == 0: every conv after the stem)
        self.wn = wk.reshape(k_real, cout).t().contiguous() if cin_p % 32 == 0 else None
        # row-mode filter (Cout, KH * ceil32(KW*Cin)) for the halo-padded kernel when Cin % 32 != 0 (the stem): one
        # filter row of a window is KW*Cin contiguous NHWC floats, treated as one tap; the tail of each row is zero
        # Channels per pixel of a row-mode frame (row_c >= cin, the extra channels are zero = part of the zero padding): the
        # kernel's LDS-DMA windows start at multiples of stride * row_c floats and a window is ceil32(kw * row_c) floats long, so
        # row_c must be even (16-byte window starts on frames of even width) and the window's zero-weight tail must fit into
        # the one pixel of slack an even frame width leaves behind the last window.  18 (the proxy representation) qualifies as
        # it is; any other count rounds up to a multiple of 4, which always does.
        self.cin = cin
        self.wrow = None
        self.row_c = cin
        if cin % 32 != 0:
            fits = lambda c: c % 2 == 0 and (kw * c + 31) // 32 * 32 - kw * c <= c
            rc = cin if fits(cin) else (cin + 3) // 4 * 4
            self.row_c = rc
            ck = (kw * rc + 31) // 32 * 32
            wr = torch.zeros(cout, kh, kw, rc, device=w.device, dtype=torch.float32)
            wr[:, :, :, :cin] = w.permute(0, 2, 3, 1)
            wrow = torch.zeros(cout, kh, ck, device=w.device, dtype=torch.float32)
            wrow[:, :, :kw * rc] = wr.reshape(cout, kh, kw * rc)
            self.wrow = wrow.reshape(cout, kh * ck).contiguous()
        # Winograd F(2x2, 3x3) form of a stride-1 3x3 layer (csrc/conv_wino.hip): U = G g G^T in the kernel's layout
        #   [cin / 8][cout / 64][position 4a + b][(cin % 8) / 4][cout % 64][cin % 4]
        self.wino_u = None
        if kh == 3 and kw == 3 and self.stride == 1 and self.pad == 1 and cin % 8 == 0 and cout % 64 == 0:
            G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
            U = torch.einsum("ai,ocij,bj->abco", G, w.double(), G)                       # (4,4,cin,cout)
            U = U.reshape(16, cin // 8, 2, 4, cout // 64, 64)                            # p, chunk, kq, j, ct, n
            self.wino_u = U.permute(1, 4, 0, 2, 5, 3).contiguous().float()              # chunk, ct, p, kq, n, j
        # Winograd form of the stem (csrc/stem_wino.hip): the 7x7 / 2 correlation as four stride-1 phase correlations, each
        # F(2x2, r x s); U = G_y g G_x^T per phase in the kernel's layout (include/hps.h: hps_stem_winograd)
        self.stem_u = None
        if kh == 7 and kw == 7 and self.stride == 2 and self.pad == 3 and cin == 18 and cout == 64:
            self.stem_u = _stem_winograd_filters(w).to(w.device).contiguous()
        self.use_winograd = True
        self.latency = False      # ResNet.set_latency_mode: direct kernels with many K slices (see _auto_ksplit)
        self.variant = 0          # tile choice of the v2 / v3 kernels (0 = automatic)
        self.ksplit = 0           # split-K slices of the v3 kernel (0 = automatic, 1 = off)
        self.zeros = torch.zeros(64, device=w.device, dtype=torch.float32)

    def _auto_ksplit(self, pixels_per_image):
        """Split K (4 slices) for the late layers whose 128x128 output tiles cannot fill 256 CUs at the benchmark batch
        (8x8 outputs: layer4).  The rule depends on the layer only, never on the batch size: the summation order of a
        pixel must not change with B, otherwise per-image results (and, through accept decisions that sit on rounding
        ties, whole sample streams) would depend on how images are batched or sharded over GPUs."""
        chunks = self.kh * self.kw * self.cin_p // 32
        if self.latency:
            # latency mode (ResNet.set_latency_mode): a single image leaves the 3x3 layers of layer2-4 with 4-16 output tiles, each a
            # serial walk over 36-144 K chunks; 12-18 slices of 3-8 chunks put 50-300 workgroups on the chip instead (batch 1,
            # tools/latency_sweep.py: layer2 27 -> 18 us, layer3 50 -> 20, layer4 97 -> 27 including the slice-sum pass).  Still a
            # rule on the layer alone.
            if self.kh * self.kw == 1 or pixels_per_image > 1024 or chunks < 36:
                return 1
            for ks in (18, 12, 9, 8, 6, 4, 3, 2):
                if chunks % ks == 0 and chunks // ks >= 3:
                    return ks
            return 1
        if self.variant == 0 and self.cout % 128 == 0 and self.kh * self.kw > 1 and pixels_per_image <= 64 \
                and chunks % 4 == 0 and chunks // 4 >= 18:
            return 4
        return 1

    def _tile_variant(self, ksplit):
        """Tile choice handed to hps_conv2d_bn_act_pad.  Latency mode: 64 x 64 tiles with a four-stage K loop for every layer (variant
        5), split-K layers included -- a single image's maps leave 128-row tiles half empty and a quarter as many workgroups on the
        chip; an output's summation order does not depend on the tile shape, so the mode's bits are those of the 128-row tiles."""
        if self.latency and self.variant == 0:
            return 5
        return self.variant if ksplit <= 1 else 0

    def winograd_ok(self, H, W, ipad):
        """Winograd F(2x2, 3x3) applies: a 3x3 / 1 / 1 layer on a map that splits into 16 x 16-pixel blocks (8 x 8 tiles, one
        work item of csrc/conv_wino.hip) or on 8 x 8 maps (layer4: four images per item, K in slices) -- a rule on the layer
        and the image size only, never on the batch size (the summation order of a pixel, and with it the last bit of every
        feature, must not depend on how images are batched or sharded)."""
        return self.use_winograd and self.wino_u is not None and ipad >= 1 and \
            ((H % 16 == 0 and W % 16 == 0) or (H == 8 and W == 8))

    def stem_winograd_ok(self, C, H, W):
        """The Winograd form of the stem (csrc/stem_wino.hip) applies: the released model's conv1 (7x7 / 2 / 3, 18 -> 64) on maps
        that split into 32 x 32-pixel input blocks -- a rule on the layer and the image size only (see winograd_ok)."""
        return self.use_winograd and self.stem_u is not None and C == 18 and H % 32 == 0 and W % 32 == 0

    def wino_workspace_bytes(self, B, H, W, ipad=1):
        """Bytes of the K-slice buffer hps_conv3x3_winograd needs for this layer on (H, W) maps (0: none / not Winograd)."""
        if not self.winograd_ok(H, W, ipad):
            return 0
        return int(_capi.load(dev=_capi._use_dev).hps_conv3x3_winograd_workspace(B, H, W, self.cin_p, self.cout))

    def padded(self, xp, ipad, out, opad, residual=None, relu=True, ws=None):
        """Halo-padded generation (csrc/conv_pad.hip): xp (B, H+2*ipad, W+2*ipad, Cin) with a zero halo; writes the interior
        of ``out`` (B, Ho+2*opad, Wo+2*opad, Cout) -- the caller owns the halo (zeroed once); ``residual`` has out's frame."""
        B, Hp, Wp, C = xp.shape
        H, W = Hp - 2 * ipad, Wp - 2 * ipad
        if self.winograd_ok(H, W, ipad):
            assert C == self.cin_p and tuple(out.shape) == (B, H + 2 * opad, W + 2 * opad, self.cout)
            P = _capi.ptr
            need = self.wino_workspace_bytes(B, H, W, ipad)
            if need and (ws is None or ws.numel() * 4 < need):
                ws = torch.empty(need // 4, device=xp.device, dtype=torch.float32)
            _capi.call("hps_conv3x3_winograd", P(xp), P(self.wino_u), P(self.scale), P(self.shift),
                       P(residual) if residual is not None else None, P(out), B, H, W, ipad, C, self.cout, opad, 1 if relu else 0,
                       P(ws) if need else None, _capi.stream())
            return out
        row_mode = self.wn is None
        assert C == (self.row_c if row_mode else self.cin_p)
        Ho = (H + 2 * self.pad - self.kh) // self.stride + 1
        Wo = (W + 2 * self.pad - self.kw) // self.stride + 1
        assert tuple(out.shape) == (B, Ho + 2 * opad, Wo + 2 * opad, self.cout)
        ksplit = 1 if row_mode else (self.ksplit if self.ksplit > 0 else self._auto_ksplit(Ho * Wo))
        if ksplit > 1 and ws is None:
            ws = torch.empty(ksplit, B * Ho * Wo, self.cout, device=xp.device, dtype=torch.float32)
        P = _capi.ptr
        _capi.call("hps_conv2d_bn_act_pad", P(xp), P(self.wrow if row_mode else self.wn), P(self.scale), P(self.shift),
                   P(residual) if residual is not None else None, P(out), B, H, W, ipad, C, self.cout, self.kh, self.kw,
                   self.stride, self.pad, opad, 1 if relu else 0, 1 if row_mode else 0, self._tile_variant(ksplit),
                   ksplit, P(ws) if ksplit > 1 else None, _capi.stream())
        return out

    def folds_down(self, down, H, W, ipad):
        """This (a block's first convolution) can carry the block's 1x1 down-sample ``down`` in its own launch
        (hps_conv2d_bn_act_pad_down): direct kernel, odd square filter with pad = k // 2 (its centre tap is the 1x1 / stride / 0
        window), same stride and output channels -- a rule on the layer only."""
        return (down is not None and self.wn is not None and down.wn is not None and not self.winograd_ok(H, W, ipad)
                and self.kh == self.kw and self.kh % 2 == 1 and self.pad == self.kh // 2 and down.kh == down.kw == 1 and down.pad == 0
                and down.stride == self.stride and down.cout == self.cout and down.cin_p == self.cin_p and self.cout >= 128)

    def _down_variant(self, ksplit):
        """Tile choice of the fused launch: the main convolution's; 0 (automatic) resolves to 128 x 128 or 64 x 64 tiles for Cout >= 128."""
        return self._tile_variant(ksplit)

    def padded_with_down(self, xp, ipad, out, opad, down, out_down, ws=None):
        """padded(xp -> out, relu) and down.padded(xp -> out_down, no relu) in ONE launch (see folds_down); identical bits."""
        B, Hp, Wp, C = xp.shape
        H, W = Hp - 2 * ipad, Wp - 2 * ipad
        assert self.folds_down(down, H, W, ipad) and C == self.cin_p
        Ho, Wo = self.out_hw(H, W)
        assert tuple(out.shape) == tuple(out_down.shape) == (B, Ho + 2 * opad, Wo + 2 * opad, self.cout)
        ksplit = self.ksplit if self.ksplit > 0 else self._auto_ksplit(Ho * Wo)
        if ksplit > 1 and ws is None:
            ws = torch.empty(ksplit, B * Ho * Wo, self.cout, device=xp.device, dtype=torch.float32)
        P = _capi.ptr
        _capi.call("hps_conv2d_bn_act_pad_down", P(xp), P(self.wn), P(self.scale), P(self.shift), P(out), P(down.wn), P(down.scale),
                   P(down.shift), P(out_down), B, H, W, ipad, C, self.cout, self.kh, self.kw, self.stride, self.pad, opad, 1,
                   self._down_variant(ksplit), ksplit, P(ws) if ksplit > 1 else None, _capi.stream())
        return out, out_down

    def enc_op_with_down(self, xp, ipad, out, opad, down, out_down, ws=None):
        """The hps_enc_op of ``padded_with_down(...)`` for hps_encoder_run."""
        B, Hp, Wp, C = xp.shape
        H, W = Hp - 2 * ipad, Wp - 2 * ipad
        assert self.folds_down(down, H, W, ipad) and C == self.cin_p
        Ho, Wo = self.out_hw(H, W)
        assert tuple(out.shape) == tuple(out_down.shape) == (B, Ho + 2 * opad, Wo + 2 * opad, self.cout)
        ksplit = self.ksplit if self.ksplit > 0 else self._auto_ksplit(Ho * Wo)
        assert ksplit == 1 or ws is not None
        return _capi.EncOp(kind=_capi.ENC_CONV_DOWN, x=xp.data_ptr(), w=self.wn.data_ptr(), scale=self.scale.data_ptr(),
                           shift=self.shift.data_ptr(), residual=None, y=out.data_ptr(), splitk_ws=ws.data_ptr() if ksplit > 1 else None,
                           B=B, H=H, W=W, ipad=ipad, Cin=C, Cout=self.cout, KH=self.kh, KW=self.kw, stride=self.stride, pad=self.pad,
                           opad=opad, relu=1, row_mode=0, variant=self._down_variant(ksplit), ksplit=ksplit, w_down=down.wn.data_ptr(),
                           scale_down=down.scale.data_ptr(), shift_down=down.shift.data_ptr(), y_down=out_down.data_ptr())

    def enc_op(self, xp, ipad, out, opad, residual=None, relu=True, ws=None):
        """The hps_enc_op of ``padded(...)`` for hps_encoder_run (same arguments, nothing is launched)."""
        B, Hp, Wp, C = xp.shape
        H, W = Hp - 2 * ipad, Wp - 2 * ipad
        dp = lambda t: t.data_ptr() if t is not None else None
        if self.winograd_ok(H, W, ipad):
            assert tuple(out.shape) == (B, H + 2 * opad, W + 2 * opad, self.cout) and C == self.cin_p
            need = self.wino_workspace_bytes(B, H, W, ipad)
            assert need == 0 or (ws is not None and ws.numel() * 4 >= need)
            return _capi.EncOp(kind=_capi.ENC_CONV_WINOGRAD, x=dp(xp), w=dp(self.wino_u), scale=dp(self.scale), shift=dp(self.shift),
                               residual=dp(residual), y=dp(out), splitk_ws=dp(ws) if need else None, B=B, H=H, W=W, ipad=ipad,
                               Cin=C, Cout=self.cout, KH=3, KW=3, stride=1, pad=1, opad=opad, relu=1 if relu else 0, ksplit=1)
        row_mode = self.wn is None
        Ho, Wo = self.out_hw(H, W)
        assert tuple(out.shape) == (B, Ho + 2 * opad, Wo + 2 * opad, self.cout) and C == (self.row_c if row_mode else self.cin_p)
        ksplit = 1 if row_mode else (self.ksplit if self.ksplit > 0 else self._auto_ksplit(Ho * Wo))
        assert ksplit == 1 or ws is not None
        return _capi.EncOp(kind=_capi.ENC_CONV, x=dp(xp), w=dp(self.wrow if row_mode else self.wn), scale=dp(self.scale),
                           shift=dp(self.shift), residual=dp(residual), y=dp(out), splitk_ws=dp(ws) if ksplit > 1 else None,
                           B=B, H=H, W=W, ipad=ipad, Cin=C, Cout=self.cout, KH=self.kh, KW=self.kw, stride=self.stride,
                           pad=self.pad, opad=opad, relu=1 if relu else 0, row_mode=1 if row_mode else 0,
                           variant=self._tile_variant(ksplit), ksplit=ksplit)

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1


_STEM_G4 = [[0.5, 0, 0, 0], [-0.5, -0.5, -0.5, -0.5], [-1 / 6, 1 / 6, -1 / 6, 1 / 6], [1 / 6, 1 / 3, 2 / 3, 4 / 3], [0, 0, 0, 1]]
_STEM_G3 = [[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]]


def _stem_winograd_filters(w):
    """w (64, 18, 7, 7) -> the 81 transformed filter positions of csrc/stem_wino.hip, fp64 arithmetic rounded once to fp32.
    Phase (ry, rx) holds the taps w[:, :, ry::2, rx::2] (4 or 3 per axis); positions are ordered phase 2 ry + rx, row i, column j;
    one position is 1152 floats: [co / 32][k-block (c % 16) / 8][c % 2][co % 32][(c % 8) / 2] for c < 16, then c = 16, 17 as
    [c - 16][co % 32] behind them.  256 floats of slack follow (the last DMA piece of a row of positions over-reads)."""
    w = w.detach().double().cpu()
    pos = []
    for ry in (0, 1):
        for rx in (0, 1):
            gy = torch.tensor(_STEM_G4 if ry == 0 else _STEM_G3, dtype=torch.float64)
            gx = torch.tensor(_STEM_G4 if rx == 0 else _STEM_G3, dtype=torch.float64)
            u = torch.einsum("ia,ocab,jb->ijco", gy, w[:, :, ry::2, rx::2], gx)          # (ny, nx, 18, 64)
            pos.append(u.reshape(-1, 18, 64))
    u = torch.cat(pos, 0)                                                                # (81, c, co)
    lo = u[:, :16].reshape(81, 2, 4, 2, 2, 32)                                           # p, kblock, e, parity, co half, co % 32   (c = 8 kb + 2 e + parity)
    lo = lo.permute(0, 4, 1, 3, 5, 2).reshape(81, 2, 512)                                # p, half, [kb][parity][co % 32][e]
    hi = u[:, 16:].reshape(81, 2, 2, 32).permute(0, 2, 1, 3).reshape(81, 2, 64)          # p, half, [c - 16][co % 32]
    packed = torch.cat([lo, hi], 2).reshape(-1)
    return torch.cat([packed, torch.zeros(256, dtype=torch.float64)]).float()


class FilledStemFrames:
    """Marker returned by ResNet.stem_frames(): the encoder's own phase frames for a (B, C, H, W) input on the current stream have
    been (or are about to be, on that stream) filled by the caller -- hps_proxy_rep_phase_frames -- instead of by
    hps_stem_phase_split from an NCHW tensor.  ``ResNet.forward(filled)`` then starts at the stem convolution."""

    def __init__(self, frames, shape, device, fill=None, owner=None, generation=0):
        self.frames, self.shape, self.device = frames, tuple(shape), device
        self.is_cuda = True
        # the frame buffer is ONE per (shape, stream) and belongs to the encoder: this object is good for exactly one forward, the next
        # stem_frames() / phase split on the same buffer bumps the owner's generation and forward() refuses a stale object
        self.owner, self.generation = owner, generation
        # ``fill``: the launch that writes the frames, deferred to forward() -- issued right in front of the stem convolution, where
        # hps_stem_phase_split would have run: the stem then reads frames that were just written (L2 / Infinity Cache), not frames
        # written before the previous batch's mesh kernel streamed 540 MB through the caches (measured: filling them early made the
        # stem 0.2 ms slower and lost more than the skipped phase split gained)
        self.fill = fill

    def run_fill(self):
        if self.fill is not None:
            fill, self.fill = self.fill, None
            fill()

    def record_stream(self, stream):            # the frames belong to the encoder (allocated once per shape and stream)
        pass


class _FrameCache(dict):
    """Per-(batch shape, stream) activation frames and launch lists; device-bound scratch, never copied or pickled."""

    def __deepcopy__(self, memo):
        return _FrameCache()

    def __reduce__(self):
        return (_FrameCache, ())


class BasicBlock(nn.Module):
    """models/resnet.py:40-78 (parameter container; executed by ResNet._run_block)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


def _invalidate_after_load(module, incompatible_keys):
    """load_state_dict post hook (fires for sub-modules too); module level so that the module stays picklable."""
    module.invalidate()


class ResNet(nn.Module):
    """models/resnet.py:125-217 for BasicBlock stacks."""

    def __init__(self, layers, in_channels):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        # same initialisation, in the same module order, as models/resnet.py:161-166
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._prepared = None
        # kernel-selection modes are properties of the MODEL (set_winograd / set_latency_mode) and survive .to() / load_state_dict:
        # prepare() re-applies them to the rebuilt _ConvBN objects
        self._winograd = True
        self._latency = False
        self.composite = True     # padded layout: issue the launch list through hps_encoder_run (one call) instead of one by one
        # Winograd stem: form the max pool in the stem kernel's epilogue (hps_stem_winograd_pooled: the full-resolution stem output is never
        # written; identical values).  False: hps_stem_winograd + hps_maxpool3x3s2_pad (the cross-check of the tests)
        self.fused_pool = True
        # the 1x1 / 2 down-sample of layer2-4's first block rides in the launch of the block's 3x3 / 2 convolution (hps_conv2d_bn_act_pad_down:
        # three launches of 16-26 us at 0.24-0.32 MFMA-busy disappear; identical bits).  False: two launches per block (the cross-check of the tests)
        self.fold_downsample = True
        # ... and gather its phase windows from the NCHW input itself (hps_stem_winograd_pooled_nchw: no hps_stem_phase_split, no phase frames;
        # identical values; needs fused_pool): phase split 0.118 + frame-fed stem 0.61 -> 0.67 ms.  The default since round 6.  In round 5
        # it was off: the persistent stem as the encoder's first kernel raced the previous batch's joint / uncertainty kernels on the
        # caller's stream for the CUs, and lost or won 3-7 % of the step with the bench's event records; InferencePipeline.inline_side now
        # queues those kernels in stream order behind the mesh kernel (no race: the rate is the same with and without the markers,
        # profiles/r06_ab.txt).  False = phase split + frame-fed kernel (the cross-check of the tests; callers that fill the frames
        # themselves -- stem_frames -- use the frame-fed kernel either way).
        self.stem_reads_nchw = True
        self._frames = _FrameCache()
        self.register_load_state_dict_post_hook(_invalidate_after_load)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        blks = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            blks.append(BasicBlock(planes, planes))
        return nn.Sequential(*blks)

    # ---- weight preparation (BN folding, k-major filters); redone after .to() / load_state_dict ----
    def _apply(self, fn, *args, **kwargs):
        self._prepared = None
        self._frames = _FrameCache()
        return super()._apply(fn, *args, **kwargs)

    def set_winograd(self, on):
        """Product default True: the stride-1 3x3 layers of layer1-3 run as Winograd F(2x2, 3x3) (csrc/conv_wino.hip); False runs
        every layer on the direct implicit-GEMM kernel (csrc/conv_pad.hip) -- the cross-check of the tests."""
        self._winograd = bool(on)
        self._apply_modes(self._prepared or self.prepare())

    def _apply_modes(self, prep):
        for c in [prep["stem"]] + [c for blk in prep["blocks"] for c in blk if c is not None]:
            c.latency = self._latency
            c.use_winograd = self._winograd and not self._latency      # latency mode runs every layer on the direct kernel
        self._frames = _FrameCache()          # cached launch lists belong to the previous selection

    def set_latency_mode(self, on=True):
        """Per-model switch for latency-bound deployments (the reference's own operating point is ONE image per call,
        predict/predict_poseMF_shapeGaussian_net.py:58-104).  The default kernels are built for throughput: a Winograd work item is
        a serial walk of one workgroup over all input channels, and a single image has 4-64 of them for 256 CUs.  In latency mode
        every layer runs the direct implicit-GEMM kernel (the stem in row mode) and the 3x3 layers of layer2-4 split K over
        12-18 workgroups per output tile (+ the slice-sum pass): encoder 0.80 -> 0.42 ms at batch 1; at batch 64 it is ~1.5x
        SLOWER than the default.  Like set_winograd it is a property of the model, never of the batch: per-image results do not
        depend on the batch size within a mode; between the modes they differ in the last bits (another summation order, same
        1e-4 feature tolerance against the reference -- tests/test_gpu_net.py)."""
        self._latency = bool(on)              # switching it off restores whatever set_winograd selected before
        self._apply_modes(self._prepared or self.prepare())

    def invalidate(self):
        """Drop the folded BatchNorm / filter copies and the cached launch lists; the next forward rebuilds them from the
        current parameters.  Called automatically by .to() and by any load_state_dict that reaches this module (also through
        a parent: nn.Module.load_state_dict recurses with _load_from_state_dict and never calls a child's load_state_dict
        override, so the reset hangs on a post hook).  Call it by hand after editing parameters in place."""
        self._prepared = None
        self._frames = _FrameCache()

    def __getstate__(self):
        # copies / pickles never carry device-bound caches (raw pointers into the original's tensors)
        state = self.__dict__.copy()
        state["_prepared"] = None
        state["_frames"] = _FrameCache()
        return state

    def prepare(self):
        cin = self.conv1.in_channels
        self._cin_pad = (cin + 3) // 4 * 4
        prep = {"stem": _ConvBN(self.conv1, self.bn1, cin_pad=self._cin_pad), "blocks": []}
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                down = _ConvBN(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                prep["blocks"].append((_ConvBN(blk.conv1, blk.bn1), _ConvBN(blk.conv2, blk.bn2), down))
        self._prepared = prep
        self._apply_modes(prep)               # (also drops the launch lists: they hold pointers to the previous filters)
        return prep

    # ---- halo-padded activation frames: owned by the module, zeroed once, only interiors are ever written ----
    def _frame_set(self, prep, B, C, H, W, device, frames=False):
        """``frames``: the caller fills the stem's phase frames itself (stem_frames / FilledStemFrames): the frame-fed stem kernel."""
        stem = prep["stem"]
        stem_wino = stem.stem_winograd_ok(C, H, W)
        fused_pool = bool(stem_wino and self.fused_pool)
        from_nchw = bool(fused_pool and self.stem_reads_nchw and not frames)
        key = (B, C, H, W, str(device), _capi.stream().value, stem_wino, fused_pool, from_nchw, bool(self.fold_downsample))
        fs = self._frames.get(key)
        if fs is not None:
            return fs
        if len(self._frames) >= 6:                       # a handful of batch shapes / streams at most
            self._frames.pop(next(iter(self._frames)))
        z = lambda *shape: torch.zeros(*shape, device=device, dtype=torch.float32)
        # input frame of the direct stem: channels as the stem's filters expect them (row mode: row_c; Cin % 32 == 0: cin_p), an even
        # number of columns; zero channels / a zero column are part of the convolution's zero padding
        cf = stem.row_c if stem.wn is None else stem.cin_p
        wf = W + (W & 1)
        generic = cf != C or wf != W or C not in (4, 18, 64)
        if from_nchw:      # the stem reads the NCHW input itself
            fs = {"in": None}
        elif stem_wino:    # four phase frames per image (hps_stem_phase_split), out-of-image pixels zeroed here once
            fs = {"in": z(int(_capi.load(dev=_capi._use_dev).hps_stem_phase_frames_bytes(B, H, W)) // 4)}
        else:
            fs = {"in": z(B, H + 6, wf + 6, cf)}
        fs["stem_wino"], fs["generic_in"] = stem_wino, (C, cf, W, wf) if generic else None
        h, w = stem.out_hw(H, W)
        fs["fused_pool"], fs["from_nchw"] = fused_pool, from_nchw
        if fused_pool:       # the stem's full-resolution output is never materialised; scratch for the items' border contributions
            fs["stem"] = None
            fs["side"] = torch.empty(int(_capi.load(dev=_capi._use_dev).hps_stem_pool_side_bytes(B, H, W)) // 4, device=device, dtype=torch.float32)
        else:
            fs["stem"] = torch.empty(B, h, w, stem.cout, device=device, dtype=torch.float32)
        h_stem, w_stem = h, w
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        fs["pool"] = z(B, h + 2, w + 2, stem.cout)
        fs["blocks"] = []
        for c1, c2, down in prep["blocks"]:
            hin, win = h, w
            h, w = c1.out_hw(h, w)
            ent = {"c1": z(B, h + 2, w + 2, c1.cout), "c2": z(B, h + 2, w + 2, c2.cout),
                   "down": z(B, h + 2, w + 2, down.cout) if down is not None else None}
            ks = max(c._auto_ksplit(h * w) if c.ksplit == 0 else c.ksplit for c in (c1, c2))
            ws_bytes = max(_capi.query_workspace(_capi.WS_CONV_SPLITK, ks, B * h * w, c1.cout),
                           c1.wino_workspace_bytes(B, hin, win), c2.wino_workspace_bytes(B, h, w))
            ent["ws"] = torch.empty(ws_bytes // 4, device=device, dtype=torch.float32) if ws_bytes else None
            fs["blocks"].append(ent)
        fs["hw"] = (h, w)
        # the launch list of hps_encoder_run: every pointer but the input image and the feature output is fixed
        if from_nchw:
            first = [_capi.EncOp(kind=_capi.ENC_STEM_WINOGRAD_POOLED_NCHW, x=None, w=stem.stem_u.data_ptr(), scale=stem.scale.data_ptr(),
                                 shift=stem.shift.data_ptr(), y=fs["pool"].data_ptr(), splitk_ws=fs["side"].data_ptr(), B=B, H=H, W=W, Cin=C,
                                 Cout=stem.cout, KH=7, KW=7, stride=2, pad=3, opad=1, relu=1)]
        elif fused_pool:
            first = [_capi.EncOp(kind=_capi.ENC_STEM_SPLIT, x=None, y=fs["in"].data_ptr(), B=B, Cin=C, H=H, W=W),
                     _capi.EncOp(kind=_capi.ENC_STEM_WINOGRAD_POOLED, x=fs["in"].data_ptr(), w=stem.stem_u.data_ptr(), scale=stem.scale.data_ptr(),
                                 shift=stem.shift.data_ptr(), y=fs["pool"].data_ptr(), splitk_ws=fs["side"].data_ptr(), B=B, H=H, W=W, Cin=C,
                                 Cout=stem.cout, KH=7, KW=7, stride=2, pad=3, opad=1, relu=1)]
        elif stem_wino:
            first = [_capi.EncOp(kind=_capi.ENC_STEM_SPLIT, x=None, y=fs["in"].data_ptr(), B=B, Cin=C, H=H, W=W),
                     _capi.EncOp(kind=_capi.ENC_STEM_WINOGRAD, x=fs["in"].data_ptr(), w=stem.stem_u.data_ptr(), scale=stem.scale.data_ptr(),
                                 shift=stem.shift.data_ptr(), y=fs["stem"].data_ptr(), B=B, H=H, W=W, Cin=C, Cout=stem.cout, KH=7, KW=7,
                                 stride=2, pad=3, opad=0, relu=1)]
        else:
            relayout = (_capi.EncOp(kind=_capi.ENC_RELAYOUT_GENERIC, x=None, y=fs["in"].data_ptr(), B=B, Cin=C, Cout=cf, H=H, W=W, KW=wf, opad=3)
                        if generic else _capi.EncOp(kind=_capi.ENC_RELAYOUT, x=None, y=fs["in"].data_ptr(), B=B, Cin=C, H=H, W=W, opad=3))
            first = [relayout, stem.enc_op(fs["in"], 3, fs["stem"], 0, relu=True)]
        ops = first if fused_pool else first + [
               _capi.EncOp(kind=_capi.ENC_MAXPOOL, x=fs["stem"].data_ptr(), y=fs["pool"].data_ptr(), B=B, H=h_stem, W=w_stem, Cin=stem.cout, opad=1)]
        y = fs["pool"]
        for (c1, c2, down), ent in zip(prep["blocks"], fs["blocks"]):      # BasicBlock.forward :62-78
            identity = y
            hin, win = y.shape[1] - 2, y.shape[2] - 2
            if down is not None and self.fold_downsample and c1.folds_down(down, hin, win, 1):
                ops.append(c1.enc_op_with_down(y, 1, ent["c1"], 1, down, ent["down"], ws=ent["ws"]))
                identity = ent["down"]
            else:
                if down is not None:
                    ops.append(down.enc_op(y, 1, ent["down"], 1, relu=False))
                    identity = ent["down"]
                ops.append(c1.enc_op(y, 1, ent["c1"], 1, relu=True, ws=ent["ws"]))
            ops.append(c2.enc_op(ent["c1"], 1, ent["c2"], 1, residual=identity, relu=True, ws=ent["ws"]))
            y = ent["c2"]
        ops.append(_capi.EncOp(kind=_capi.ENC_AVGPOOL, x=y.data_ptr(), y=None, B=B, H=h, W=w, Cin=y.shape[3], ipad=1))
        fs["ops"] = (_capi.EncOp * len(ops))(*ops)
        fs["variants"] = self._variant_state(prep)
        self._frames[key] = fs
        return fs

    @staticmethod
    def _variant_state(prep):
        convs = [prep["stem"]] + [c for blk in prep["blocks"] for c in blk if c is not None]
        return tuple((c.variant, c.ksplit, c.use_winograd, c.latency) for c in convs)

    def stem_frames(self, B, C, H, W, device):
        """The phase-frame buffer the Winograd stem will read for a (B, C, H, W) input on the CURRENT stream, wrapped as a
        FilledStemFrames for the caller to fill (hps_proxy_rep_phase_frames) and hand to forward(); None when this shape does not
        take the Winograd stem (then build the NCHW tensor as usual)."""
        prep = self._prepared or self.prepare()
        if not prep["stem"].stem_winograd_ok(C, H, W):
            return None
        fs = self._frame_set(prep, B, C, H, W, device, frames=True)
        if not fs["stem_wino"]:
            return None
        fs["generation"] = fs.get("generation", 0) + 1
        return FilledStemFrames(fs["in"], (B, C, H, W), device, owner=fs, generation=fs["generation"])

    def _forward_padded(self, prep, x, gate=None):
        """``gate``: optional callable invoked after the input relayout has been enqueued and before the first convolution
        (InferencePipeline: the HBM-bound relayout may run beside the previous batch's MFMA-bound mesh kernel; the
        convolutions wait)."""
        filled = isinstance(x, FilledStemFrames)
        B, C, H, W = x.shape
        s = _capi.stream()
        P = _capi.ptr
        fs = self._frame_set(prep, B, C, H, W, x.device, frames=filled)
        if filled and (not fs["stem_wino"] or fs["in"].data_ptr() != x.frames.data_ptr()):
            raise _capi.HpsError("FilledStemFrames belong to another stream / shape / kernel selection than this forward (fill the "
                                 "frames stem_frames() returned on the stream the encoder runs on)")
        if filled:
            if x.owner is not fs or x.generation != fs.get("generation", 0):
                raise _capi.HpsError("stale FilledStemFrames: the encoder's frame buffer has been handed out again (or consumed by a forward) "
                                     "since stem_frames() returned this object; call stem_frames() once per forward")
            fs["generation"] += 1                        # consumed: a second forward with the same object would read whatever the frames hold then
        elif fs.get("in") is not None and fs["stem_wino"]:
            fs["generation"] = fs.get("generation", 0) + 1      # the phase split below overwrites the frames
        if self.composite and fs["variants"] == self._variant_state(prep):
            # one call across the C ABI for the whole encoder (csrc/composite.hip); two when the list is gated
            feats = torch.empty(B, fs["blocks"][-1]["c2"].shape[3], device=x.device, dtype=torch.float32)
            ops = fs["ops"]
            ops[len(ops) - 1].y = feats.data_ptr()
            if filled:                                   # the caller fills the phase frames: the list starts at the stem convolution
                import ctypes
                if gate is not None:
                    gate()
                x.run_fill()
                rest = ctypes.cast(ctypes.byref(ops, ctypes.sizeof(_capi.EncOp)), ctypes.POINTER(_capi.EncOp))
                _capi.call("hps_encoder_run", rest, len(ops) - 1, s)
                return feats
            ops[0].x = x.data_ptr()
            if fs["from_nchw"] and gate is not None:     # no relayout in front of the first convolution: the whole list waits
                gate()
                gate = None
            if gate is None:
                _capi.call("hps_encoder_run", ops, len(ops), s)
            else:
                import ctypes
                _capi.call("hps_encoder_run", ops, 1, s)
                gate()
                rest = ctypes.cast(ctypes.byref(ops, ctypes.sizeof(_capi.EncOp)), ctypes.POINTER(_capi.EncOp))
                _capi.call("hps_encoder_run", rest, len(ops) - 1, s)
            return feats
        stem = prep["stem"]
        if fs["from_nchw"]:
            if gate is not None:
                gate()
            y = None
            _capi.call("hps_stem_winograd_pooled_nchw", P(x), P(stem.stem_u), P(stem.scale), P(stem.shift), P(fs["pool"]), P(fs["side"]),
                       B, H, W, 1, 1, s)
        elif fs["stem_wino"]:
            if not filled:
                _capi.call("hps_stem_phase_split", P(x), P(fs["in"]), B, C, H, W, s)
            if gate is not None:
                gate()
            if filled:
                x.run_fill()
            y = fs["stem"]
            if fs["fused_pool"]:
                _capi.call("hps_stem_winograd_pooled", P(fs["in"]), P(stem.stem_u), P(stem.scale), P(stem.shift), P(fs["pool"]), P(fs["side"]),
                           B, H, W, 1, 1, s)
            else:
                _capi.call("hps_stem_winograd", P(fs["in"]), P(stem.stem_u), P(stem.scale), P(stem.shift), P(y), B, H, W, 0, 1, s)
        else:
            if fs["generic_in"] is not None:
                _, cf, _, wf = fs["generic_in"]
                _capi.call("hps_nchw_to_padded_nhwc_generic", P(x), P(fs["in"]), B, C, cf, H, W, wf, 3, s)
            else:
                _capi.call("hps_nchw_to_padded_nhwc", P(x), P(fs["in"]), B, C, H, W, 3, s)
            if gate is not None:
                gate()
            y = stem.padded(fs["in"], 3, fs["stem"], 0, relu=True)         # conv1 + bn1 + relu
        if not fs["fused_pool"]:
            _capi.call("hps_maxpool3x3s2_pad", P(y), P(fs["pool"]), B, y.shape[1], y.shape[2], y.shape[3], 1, s)
        y = fs["pool"]
        for (c1, c2, down), ent in zip(prep["blocks"], fs["blocks"]):      # BasicBlock.forward :62-78
            if down is not None and self.fold_downsample and c1.folds_down(down, y.shape[1] - 2, y.shape[2] - 2, 1):
                out, identity = c1.padded_with_down(y, 1, ent["c1"], 1, down, ent["down"], ws=ent["ws"])
            else:
                identity = down.padded(y, 1, ent["down"], 1, relu=False) if down is not None else y
                out = c1.padded(y, 1, ent["c1"], 1, relu=True, ws=ent["ws"])
            y = c2.padded(out, 1, ent["c2"], 1, residual=identity, relu=True, ws=ent["ws"])
        h, w = fs["hw"]
        feats = torch.empty(B, y.shape[3], device=x.device, dtype=torch.float32)
        _capi.call("hps_global_avgpool_pad", P(y), P(feats), B, h, w, y.shape[3], 1, s)
        return feats

    def forward(self, x, _gate=None):
        """models/resnet.py:202-217: (B,C,H,W) NCHW fp32 -> (B,512)."""
        if self.training:
            raise RuntimeError("the MI355X encoder path is inference-only (eval-mode BatchNorm); call .eval()")
        prep = self._prepared or self.prepare()
        if isinstance(x, FilledStemFrames):
            return self._forward_padded(prep, x, gate=_gate)
        _capi.require_device(x, "encoder input")
        # every (C, H, W) runs on the product kernels: shapes the stem's fast paths do not take get a channel-padded, even-width input
        # frame (hps_nchw_to_padded_nhwc_generic) in front of the row-mode / direct stem
        return self._forward_padded(prep, _capi.f32c(x), gate=_gate)


def resnet18(in_channels, pretrained=False, progress=True, **kwargs):
    """models/resnet.py:229-237 (pretrained ImageNet weights are never used by the reference's nets)."""
    if pretrained:
        raise NotImplementedError("no network access: pretrained weights are not available")
    return ResNet([2, 2, 2, 2], in_channels)
