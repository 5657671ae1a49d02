cd $GRAFT_REPO_ROOT
run() { python bench.py --workload c4 --no-cpu --no-extras --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{\"metric\"'):])
print('$1', 'c4', round(d['ms_per_step'],2), d['roofline']['per_kernel_ms_per_step'])"
python bench.py --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{\"metric\"'):])
print('$1', 'c3', round(d['ms_per_step'],2), d['roofline']['per_kernel_ms_per_step'])"; }
run head
python - <<'PY'
p='apples_amd/csrc/sweep_lean.hip'
t=open(p).read()
a=t.index("        // (a pool that has run dry is not asked again")
b=t.index("        const int64_t off = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)off0);")
t=t[:a]+"        unsigned int off0 = 0;\n        if (lane == 0) off0 = atomicAdd(a.pool_cursor, (unsigned int)qcap);\n"+t[b:]
open(p,'w').write(t)
PY
python -m apples_amd.build > /dev/null 2>&1
run oldcursor
