# Timeline of one pass of the clustered route at config 3's size (host buffers -> host): kernels and copies of the last step of a
# bench run (rocprofv3 kernel + memory-copy trace).  usage (on the GPU box): bash scripts/r05_clustered_timeline.sh [workload]
W=${1:-c3-clustered}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_trace -- python3 $R/bench.py --workload $W --no-cpu --no-extras --steps 2 --warmup 1 > /tmp/tl_bench.json 2>/dev/null
python3 - <<PY
import csv, glob, re, json
k = glob.glob("/tmp/tl_trace/*/*kernel_trace.csv")[0]
def nm(s):
    m = re.search(r"(k_[a-zA-Z0-9_]+(<[^>]*>)?|__amd_rocclr_[a-zA-Z]+)", s)
    return m.group(1) if m else s[:30]
rows = [("K", nm(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k))]
for m in glob.glob("/tmp/tl_trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(m)):
        size = next((r[c] for c in r if "ize" in c or "ytes" in c), "")
        rows.append(("C", (r.get("Direction") or r.get("Kind") or "copy")[-16:] + " " + str(size), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[2])
d = json.load(open("/tmp/tl_bench.json"))
print("bench ms_per_step", d["ms_per_step"], "resident", d["resident"]["ms_per_step"], d["resident"]["per_kernel_ms_per_step"])
# the last host->host pass: the resident passes follow it; take the window of ms_per_step before the last big D2H of the timed passes
# the last pass in the trace is a resident one (query block already uploaded and packed): its window
t_end = max(r[3] for r in rows)
t0 = t_end - int(d["resident"]["ms_per_step"] * 1e6) - 300000
last = None
for r in rows:
    if r[2] < t0 or r[2] > t_end: continue
    if r[3] - r[2] < 20000 and r[0] == "K" and "rocclr" in r[1]: continue
    print("%s %-44s start %8.3f ms  dur %7.3f ms" % (r[0], r[1][:44], (r[2] - t0) / 1e6, (r[3] - r[2]) / 1e6))
PY
