cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_split
python -m pytest tests/test_gpu_smpl.py -x -q -m gpu -k "split" 2>&1 | tail -15
python tests/dev/mesh_split_time.py 2>&1 | tail -12
