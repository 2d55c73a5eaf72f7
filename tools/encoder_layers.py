"""Per-layer view of the encoder at the benchmark shape (B = 64, 18x256x256), on the GPU box.

    python tools/encoder_layers.py time   -> gpurun_out/enc_layers_time.json: every operation of hps_encoder_run's launch list run
                                             ALONE (20 launches between HIP events): time, algorithmic GFLOP, TF/s
    python tools/encoder_layers.py once   -> runs the whole list 3 times (target of the rocprofv3 --pmc passes of tools/encoder_pmc.sh)
tools/summarize_encoder_layers.py merges both into profiles/<tag>_encoder_layers.{json,md}."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import _capi, configs  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402


def op_names(enc, stem_wino=True, fused_pool=False, from_nchw=False, fold=False):
    names = (["conv1 7x7/2 18->64 + maxpool 3x3/2 (stem, winograd; windows gathered from the NCHW input, pool in the epilogue + border pass)"] if from_nchw else
             ["phase split NCHW->4 phase frames", "conv1 7x7/2 18->64 + maxpool 3x3/2 (stem, winograd F(2x2, 4x4|4x3|3x4|3x3); pool in the epilogue + border pass)"]
             if fused_pool else
             ["phase split NCHW->4 phase frames", "conv1 7x7/2 18->64 (stem, winograd F(2x2, 4x4|4x3|3x4|3x3))", "maxpool 3x3/2"] if stem_wino
             else ["relayout NCHW->padded NHWC", "conv1 7x7/2 18->64 (stem, row mode)", "maxpool 3x3/2"])
    for li, layer in enumerate((enc.layer1, enc.layer2, enc.layer3, enc.layer4), 1):
        for bi, blk in enumerate(layer):
            if blk.downsample is not None and fold:
                names.append("layer%d.%d.conv1 3x3/2 + downsample 1x1/2 (one launch)" % (li, bi))
            else:
                if blk.downsample is not None:
                    names.append("layer%d.%d.downsample 1x1/2" % (li, bi))
                names.append("layer%d.%d.conv1 3x3%s" % (li, bi, "/2" if blk.stride == 2 else ""))
            names.append("layer%d.%d.conv2 3x3" % (li, bi))
    names.append("global avgpool")
    return names


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
    enc = net.image_encoder
    if len(sys.argv) > 3 and sys.argv[3] == "direct":     # A/B: every layer on the direct kernel
        enc.set_winograd(False)
    if len(sys.argv) > 3 and sys.argv[3] == "unfused-pool":     # A/B: stem and max pool as two kernels
        enc.fused_pool = False
    if len(sys.argv) > 3 and sys.argv[3] == "frames":           # A/B: phase split + frame-fed stem (round 5's default)
        enc.stem_reads_nchw = False
    if len(sys.argv) > 3 and sys.argv[3] == "separate-downsample":     # A/B: the 1x1/2 down-samples as launches of their own (round 5)
        enc.fold_downsample = False
    x = torch.rand(B, 18, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        feats = enc(x)                                    # builds the frames and the launch list
        torch.cuda.synchronize()
        fs = next(iter(enc._frames.values()))
        ops, n = fs["ops"], len(fs["ops"])
        names = op_names(enc, fs["stem_wino"], fs.get("fused_pool", False), fs.get("from_nchw", False),
                         fold=any(o.kind == _capi.ENC_CONV_DOWN for o in ops))
        assert len(names) == n, (len(names), n)
        ops[0].x = x.data_ptr()
        ops[n - 1].y = feats.data_ptr()
        s = _capi.stream()
        if mode == "once":
            for _ in range(3):
                _capi.call("hps_encoder_run", ops, n, s)
            torch.cuda.synchronize()
            return
        rows = []
        for i in range(n):
            o = ops[i]
            one = (_capi.EncOp * 1)(o)
            for _ in range(3):
                _capi.call("hps_encoder_run", one, 1, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _capi.call("hps_encoder_run", one, 1, s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row = {"op": names[i], "ms_alone": ms}
            if o.kind in (_capi.ENC_CONV, _capi.ENC_CONV_WINOGRAD, _capi.ENC_CONV_DOWN):
                Ho = (o.H + 2 * o.pad - o.KH) // o.stride + 1
                Wo = (o.W + 2 * o.pad - o.KW) // o.stride + 1
                cin = 18 if o.row_mode else o.Cin
                gflop = 2.0 * o.B * Ho * Wo * o.Cout * (o.KH * o.KW + (1 if o.kind == _capi.ENC_CONV_DOWN else 0)) * cin / 1e9
                wino = o.kind == _capi.ENC_CONV_WINOGRAD
                row.update(gflop=gflop, tflops=gflop / ms, out_hw=[Ho, Wo], cin=cin, cout=o.Cout, ksplit=o.ksplit,
                           launches=2 if (o.ksplit > 1 or (wino and o.H == 8 and o.W == 8)) else 1,
                           algorithm=("winograd F(2x2,3x3), four images per item, K in 4 slices" if o.H == 8 else "winograd F(2x2,3x3)") if wino
                           else "direct implicit GEMM",
                           mfma_gflop=gflop / 2.25 if wino else gflop)
            if o.kind in (_capi.ENC_STEM_WINOGRAD, _capi.ENC_STEM_WINOGRAD_POOLED, _capi.ENC_STEM_WINOGRAD_POOLED_NCHW):
                gflop = 2.0 * o.B * (o.H // 2) * (o.W // 2) * 64 * 49 * 18 / 1e9
                row.update(gflop=gflop, tflops=gflop / ms, out_hw=[o.H // 2, o.W // 2], cin=18, cout=64, ksplit=1,
                           launches=2 if o.kind in (_capi.ENC_STEM_WINOGRAD_POOLED, _capi.ENC_STEM_WINOGRAD_POOLED_NCHW) else 1,
                           algorithm="winograd, four stride-1 phases F(2x2, r x s): 81 multiplications per tile instead of 196",
                           mfma_gflop=gflop * 81.0 / 196.0)
            rows.append(row)
            print("%-40s %.4f ms %s" % (names[i], ms, ("%.1f TF/s" % row["tflops"]) if "tflops" in row else ""), flush=True)
        # the same operations in the order of the list, one event between each (20 passes): the in-context time of every operation
        import ctypes
        passes = 20
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in range(passes)]
        for pi in range(passes):
            evs[pi][0].record()
            for i in range(n):
                one = ctypes.cast(ctypes.byref(ops, i * ctypes.sizeof(_capi.EncOp)), ctypes.POINTER(_capi.EncOp))
                _capi.call("hps_encoder_run", one, 1, s)
                evs[pi][i + 1].record()
        torch.cuda.synchronize()
        for i in range(n):
            rows[i]["ms_in_list"] = sum(evs[pi][i].elapsed_time(evs[pi][i + 1]) for pi in range(2, passes)) / (passes - 2)
            if "gflop" in rows[i]:
                rows[i]["tflops_in_list"] = rows[i]["gflop"] / rows[i]["ms_in_list"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            _capi.call("hps_encoder_run", ops, n, s)
        e0.record()
        for _ in range(20):
            _capi.call("hps_encoder_run", ops, n, s)
        e1.record()
        torch.cuda.synchronize()
        whole = e0.elapsed_time(e1) / 20
        out = {"batch": B, "whole_encoder_ms": whole, "sum_alone_ms": sum(r["ms_alone"] for r in rows),
               "encoder_tflops": 6.279 * B / whole, "ops": rows}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "enc_layers_time.json"), "w"), indent=1)
        print("whole encoder %.3f ms = %.1f TF/s; sum of the operations alone %.3f ms" % (whole, out["encoder_tflops"], out["sum_alone_ms"]))


if __name__ == "__main__":
    main()
