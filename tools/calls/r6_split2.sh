cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_split
python -m pytest tests/test_gpu_smpl.py tests/test_capi_symbols.py -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off > gpurun_out/r6_split/bench_default.json 2> gpurun_out/r6_split/bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off --mesh-arith bf16x3 > gpurun_out/r6_split/bench_split.json 2> gpurun_out/r6_split/bench_split.err
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off --split-steps 0 > gpurun_out/r6_split/bench_default2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --cpu-images 0 --from-rgb-steps 0 --latency-reps 0 --stress-steps 0 --lbs-unfused-reps 0 --live-traffic off --mesh-arith bf16x3 > gpurun_out/r6_split/bench_split2.json 2>/dev/null
tail -3 gpurun_out/r6_split/*.err
python - <<'PY'
import json
for n in ("bench_default", "bench_split", "bench_default2", "bench_split2"):
    d = json.load(open("gpurun_out/r6_split/%s.json" % n))
    print(n, round(d["value"]), "images/s", round(d["ms_per_step"], 3), "ms; mesh", round(d["roofline"]["avg_launch_ms"], 4), d["roofline"]["kernel"], "enc", round(d["secondary"]["encoder"]["avg_ms"], 3))
    if "mesh_bf16x3" in d["secondary"]:
        m = d["secondary"]["mesh_bf16x3"]
        print("   secondary.mesh_bf16x3:", round(m["images_per_s"]), "images/s", m["mesh_kernel_ms"], m["max_abs_diff_vs_f32_m"], "enc", m["encoder_avg_ms"])
PY
