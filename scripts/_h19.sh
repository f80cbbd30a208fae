P='import sys,json
d=json.loads(sys.stdin.read())
m=d["roofline"]["more"]
print([ (k, m[k]["us_per_launch"]) for k in ("gather_bwd","gather_bwd_bf16_dhist","gather_bwd_item_and_category_one_stream")], d["extra_workloads"][-1]["ms_per_step"] if d.get("extra_workloads") else None)'
for rep in 1 2; do
echo "hint   $(python bench.py --no-cpu-baseline --no-extra --steps 5 2>/dev/null | grep '^{"metric' | python -c "$P")"
echo "nohint $(CLSR_NO_BORDER_WCH=1 python bench.py --no-cpu-baseline --no-extra --steps 5 2>/dev/null | grep '^{"metric' | python -c "$P")"
done
