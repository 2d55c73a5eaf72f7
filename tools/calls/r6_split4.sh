cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bf16x3 or bench_line" 2>&1 | tail -8
