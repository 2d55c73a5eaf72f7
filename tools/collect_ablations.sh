#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the experiment outputs DESIGN.md section 4 quotes that are not part of the bench line --
# ablation tables of the Winograd and fused mesh kernels, head / SVD / uncertainty timings alone, the pipelined loop's device-side
# schedule from HIP events, the cost of the overlapped work, the small-batch A/B of the Winograd layers.  Plain text, one file:
# gpurun_out/ablations_$TAG.txt (copied to profiles/${TAG}_ablations.txt).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ablations_$TAG.txt
mkdir -p $R/gpurun_out; : > $OUT
sec() { echo; echo "==== $* ====" ; }
{
sec "tools/bin/mfma_peak (v_mfma_f32_32x32x2_f32, no memory traffic: constant vs random operands; 3 repetitions)"
for i in 1 2 3; do $R/tools/bin/mfma_peak 2>&1; done
sec "tools/bin/mfma_valu_overlap (does fp32 VALU overlap fp32 MFMA: same wave, and MFMA waves beside VALU waves; 2 repetitions)"
for i in 1 2; do $R/tools/bin/mfma_valu_overlap 2>&1; done
sec "tests/dev/gpu_bringup.py wino (B = 64; dev library: compile-time ablations of conv_wino_kernel)"
python $R/tests/dev/gpu_bringup.py wino 2>&1 | grep -E "^wino|^layer4|direct kernel|max \|"
sec "tests/dev/stem_ablate.py (stem convolution alone, modes interleaved over 9 rounds: 0 = product, 1 = no epilogue, 3 = DMA pieces in a burst (the earlier form), 5 / 13 = second CU slot de-phased by 14 / 41 us)"
python $R/tests/dev/stem_ablate.py 0 1 3 5 13 2>&1 | grep "^stem"
sec "tests/dev/stem_wino_check.py (Winograd stem against an fp64 convolution and the direct kernel; both alone, B = 64 and 16)"
PYTHONPATH=$R python $R/tests/dev/stem_wino_check.py 2>&1 | grep -E "^B=|median"
sec "tests/dev/stem_wino_ablate.py (Winograd stem, modes interleaved over 9 rounds: 0 = product, 1 = window pixels not read, 2 = no MFMAs, 3 = no output transform, 4 = no barriers, 5 = no filter fragment reads)"
PYTHONPATH=$R python $R/tests/dev/stem_wino_ablate.py 2>&1 | grep "^stem"
sec "tools/bin/mfma_valu_ops (which VALU instructions cost MFMA time, alone / same wave / other wave of the SIMD; cost of alternating MFMA and VALU runs)"
$R/tools/bin/mfma_valu_ops 2>&1 | grep -E "^v_"
sec "tools/stem_pmc.sh (SQ counters of stem_wino_kernel, B = 64, per dispatch)"
bash $R/tools/stem_pmc.sh 0 2>&1 | grep -E "^SQ_|^GRBM"
sec "tools/next_rows_time.py (kernels of the widened rows, SURVEY 8(f) items 1 and 2: time alone against their algorithmic HBM bytes)"
python $R/tools/next_rows_time.py 2>&1 | grep "^hps_"
sec "tools/evaluate_time.py 512 <batch> 10 (dataset-evaluation harness, synthetic 3DPW-like dataset: frames/s of the whole loop)"
for b in 32 8 1; do python $R/tools/evaluate_time.py 512 $b 10 0 2>&1 | grep "evaluate harness"; done
sec "tests/dev/gpu_bringup.py mesh_fused (ablations of mesh_fused_kernel)"
python $R/tests/dev/gpu_bringup.py mesh_fused 2>&1 | grep -E "mesh_fused M=|alone|ablate"
sec "tests/dev/gpu_bringup.py unc_modes"
python $R/tests/dev/gpu_bringup.py unc_modes 2>&1 | grep -E "^unc|registers =="
sec "tests/dev/unc_time.py (hps_vertex_uncertainty alone: one-sweep product kernel vs the two-sweep kernel, median of 10)"
python $R/tests/dev/unc_time.py 2>&1 | grep "^unc"
sec "tests/dev/svd_vs_mkl.py (host build of csrc/svd3_gesdd.h, both rounding flavours, against this host's torch.svd)"
python $R/tests/dev/svd_vs_mkl.py 2>&1 | grep -E "hps_host_svd_flavor|total"
sec "tools/bin/ldsdma_probe (LDS placement of global_load_lds of 4 / 12 / 16 bytes per lane: first 20 dwords)"
$R/tools/bin/ldsdma_probe 2>&1 | cut -d" " -f1-22
sec "tests/dev/head_time.py 64 (head alone, device SVD)"
python $R/tests/dev/head_time.py 64 2>&1 | grep "head alone"
sec "tests/dev/svd_time.py (hps_svd3_packed: one lane per matrix -- divergence)"
python $R/tests/dev/svd_time.py 2>&1 | grep "^svd3"
sec "bench.py --steps 20 --warmup 5 --trace-steps (device-side schedule from HIP events, ms)"
python $R/bench.py --steps 20 --warmup 5 --cpu-images 0 --trace-steps 2>&1 | grep -E "^batch|host ms" | cut -c1-400
sec "tests/dev/contention.py <drop> --steps 30 (images/s, ms/step, encoder ms inside the loop)"
for m in none unc sums unc,sums; do
  python $R/tests/dev/contention.py $m --steps 30 --warmup 5 --cpu-images 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('drop %-9s %6d images/s  %.3f ms/step  encoder %.3f ms' % ('$m', d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms']))"
done
sec "tools/encoder_layers.py time B [direct] (encoder alone, Winograd vs all-direct)"
for b in 64 16 1; do for m in wino direct; do echo "B=$b $m: $(python $R/tools/encoder_layers.py time $b $m 2>&1 | grep whole)"; done; done
sec "BASELINE configs[4] (B = 16, N = 1000): mesh kernel exclusive (off) or beside the encoder (on), shared CUs (0) or a CU partition of k CUs per XCD for the encoder"
for a in "off 0" "on 0" "on 8" "on 12"; do set -- $a
  python $R/bench.py --batch 16 --num-samples 1000 --steps 12 --warmup 3 --cpu-images 0 --lbs-unfused-reps 0 --mesh-overlap $1 --encoder-cus $2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mesh-overlap %-3s encoder-cus %-2s %6d images/s  %.3f ms/step  encoder %.3f ms  mesh kernel %.3f ms' % ('$1', '$2', d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms'], d['roofline']['avg_launch_ms']))"
done
sec "BASELINE configs[4] (B = 16, N = 1000), Winograd vs all-direct"
for t in bench.py tests/dev/bench_direct.py; do
  python $R/$t --batch 16 --num-samples 1000 --steps 10 --warmup 3 --cpu-images 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %6d images/s  %.3f ms/step  encoder %.3f ms' % ('$t', d['value'], d['ms_per_step'], d['secondary']['encoder']['avg_ms']))"
done
sec "bench.py --early-relayout A/B (30 steps)"
for f in "" "--early-relayout"; do
  python $R/bench.py --steps 30 --warmup 5 --cpu-images 0 $f 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-18s %6d images/s  %.3f ms/step  mesh kernel %.4f ms' % ('$f' or 'default', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
} >> $OUT 2>&1
python $R/tools/encoder_layers.py time > /dev/null 2>&1     # leave gpurun_out/enc_layers_time.json at the benchmark shape
echo "wrote $OUT"
