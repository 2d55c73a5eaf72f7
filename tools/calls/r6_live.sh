cd $GRAFT_REPO_ROOT
time python bench.py --steps 20 --warmup 5 2>/tmp/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), r['traffic'], r['traffic_imported'], r['traffic_detail'], r['traffic_source'][:80])"
tail -3 /tmp/err.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k bench_line 2>&1 | tail -3
