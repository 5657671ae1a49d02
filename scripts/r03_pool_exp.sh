# Round 3: clustered route, size of the lean sweep's pool (queries beyond it overflow to the workgroup-sized teams)
one() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --timed resident "$@" 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['resident']['ms_per_step'],2), d['resident']['per_kernel_ms_per_step'], d['roofline']['launches_per_step'])"; }
for p in 12288 20000 32000; do echo "== clustered, pool $p MB: $(APPLES_LEAN_POOL_MB=$p one --workload c3-clustered)"; done
for p in 32000 48000; do echo "== clustered, pool $p MB, batch budget 140 GiB: $(APPLES_BATCH_GIB=140 APPLES_LEAN_POOL_MB=$p one --workload c3-clustered)"; done
for p in 48000 64000; do echo "== clustered, pool $p MB, batch budget 180 GiB: $(APPLES_BATCH_GIB=180 APPLES_LEAN_POOL_MB=$p one --workload c3-clustered)"; done
