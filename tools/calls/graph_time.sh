cd $GRAFT_REPO_ROOT
python tests/dev/graph_time.py 30 --pipeline --bisect 2>&1 | grep -v amdgpu.ids
