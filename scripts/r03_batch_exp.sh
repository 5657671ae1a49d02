# Round 3: device batch size with the lean sweep (C3 singleton route and clustered route)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'], d['roofline']['launches_per_step'])"; }
for g in 96 64 48 32 24; do echo "== c3, batch budget $g GiB: $(APPLES_BATCH_GIB=$g one)"; done
for g in 96 64 48 32; do echo "== clustered, batch budget $g GiB: $(APPLES_BATCH_GIB=$g one --workload c3-clustered)"; done
