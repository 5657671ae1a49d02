# Kernels and copies of one host-buffer step of a bench run in time order (rocprofv3 kernel + memory-copy trace): where the gaps are.
# usage (on the GPU box): [APPLES_BENCH_SHARD=k] bash scripts/r04_step_timeline.sh --workload c2|c3|c4 [more bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_trace -- python3 $R/bench.py --no-cpu --no-extras --steps 3 --warmup 2 "$@" > /tmp/tl_bench.json 2>/dev/null
python3 - <<PY
import csv, glob, re, json
k = glob.glob("/tmp/tl_trace/*/*kernel_trace.csv")[0]
pat = r"(k_[a-zA-Z0-9_]+|__amd_rocclr_[a-zA-Z]+)"
rows = [("K", re.search(pat, r["Kernel_Name"]).group(1) if re.search(pat, r["Kernel_Name"]) else r["Kernel_Name"][:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k))]
for m in glob.glob("/tmp/tl_trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(m)):
        rows.append(("C", (r.get("Direction") or "copy").replace("MEMORY_COPY_", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[2])
d = json.load(open("/tmp/tl_bench.json"))
print("bench ms_per_step", round(d["ms_per_step"], 3), "resident", round(d["resident"]["ms_per_step"], 3))
packs = [i for i, r in enumerate(rows) if r[1] in ("k_pack_rows", "k_pack_aa")]
# the last host step = the packs before the resident loop's upload; take the step that starts at the 3rd-last group of packs
groups = []
for i in packs:
    if not groups or rows[i][2] - rows[groups[-1][-1]][2] > 1500000: groups.append([i])
    else: groups[-1].append(i)
g = groups[-3] if len(groups) >= 3 else groups[0]
s = g[0]
while s > 0 and rows[s][2] - rows[s - 1][3] < 150000 and rows[s - 1][0] == "C": s -= 1
t0 = rows[s][2]
end = rows[groups[groups.index(g) + 1][0]][2] if groups.index(g) + 1 < len(groups) else t0 + 60e6
for r in rows[s:]:
    if r[2] >= end: break
    print("%s %-30s start %8.3f ms  dur %7.3f ms" % (r[0], r[1][:30], (r[2] - t0) / 1e6, (r[3] - r[2]) / 1e6))
PY
