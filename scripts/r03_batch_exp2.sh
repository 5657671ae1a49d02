# Round 3: clustered route, device batch size beyond the default budget (how much is a launch worth?)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'], d['roofline']['launches_per_step'])"; }
for g in 96 140 200 240; do echo "== clustered, batch budget $g GiB: $(APPLES_BATCH_GIB=$g one --workload c3-clustered)"; done
