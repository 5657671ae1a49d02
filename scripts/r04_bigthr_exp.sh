# the lean sweep's routing threshold at the full C3 set and at one 8-GPU shard (12 500 queries): host -> host ms and the phases
cd $GRAFT_REPO_ROOT
for T in 8192 4096 2048; do
  for Q in 100000 12500; do
    APPLES_BIG_THRESHOLD=$T python bench.py --no-cpu --no-extras --queries $Q --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{\"metric\"'):])
print('threshold $T queries $Q: %.2f ms' % d['ms_per_step'], {k: round(v,2) for k,v in d['roofline']['per_kernel_ms_per_step'].items()})"
  done
done
