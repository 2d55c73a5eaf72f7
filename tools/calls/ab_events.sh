for r in 1 2 3; do
for e in 1 0 2 4; do
echo "event-every $e: $(python bench.py --steps 40 --warmup 10 --from-rgb-steps 0 --stress-steps 0 --event-every $e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['ms_per_step'], d['roofline'].get('launches'), d['roofline'].get('achieved'))")"
done; done
