# Timeline of ONE 12 500-query shard of C3 (what one of eight ranks runs per step), host buffer -> host: kernels and copies
# of the last step of a bench run (rocprofv3 kernel + memory-copy trace).  usage (on the GPU box): bash scripts/r04_shard_timeline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_trace -- python3 $R/bench.py --no-cpu --no-extras --queries 12500 --steps 3 --warmup 2 --scaling weak "$@" > /tmp/tl_bench.json 2>/dev/null
python3 - <<PY
import csv, glob, re, json
k = glob.glob("/tmp/tl_trace/*/*kernel_trace.csv")[0]
rows = [("K", re.search(r"(k_[a-zA-Z0-9_]+|__amd_rocclr_[a-zA-Z]+)", r["Kernel_Name"]).group(1) if re.search(r"(k_[a-zA-Z0-9_]+|__amd_rocclr_[a-zA-Z]+)", r["Kernel_Name"]) else r["Kernel_Name"][:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k))]
for m in glob.glob("/tmp/tl_trace/*/*memory_copy_trace.csv"):
    rd = csv.DictReader(open(m))
    print("copy trace columns:", rd.fieldnames)
    for r in rd:
        size = next((r[c] for c in r if "ize" in c or "ytes" in c), "")
        rows.append(("C", (r.get("Direction") or r.get("Kind") or "copy") + " " + str(size), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[2])
# the steps: from every host-to-device copy of the query bytes (12.5 MB) on
big = [i for i, r in enumerate(rows) if r[0] == "C" and "12500000" in r[1]]
if not big:
    big = [i for i, r in enumerate(rows) if "k_pack_rows" in r[1]]
d = json.load(open("/tmp/tl_bench.json"))
print("bench ms_per_step", d["ms_per_step"], "resident", d["resident"]["ms_per_step"], d["resident"]["per_kernel_ms_per_step"])
for s in big[-2:-1]:
    t0 = rows[s][2]
    print("---- step starting at the copy of the query bytes")
    for r in rows[s:s + 40]:
        if r[2] - t0 > 12e6: break
        print("%s %-34s start %8.3f ms  dur %7.3f ms" % (r[0], r[1][:34], (r[2] - t0) / 1e6, (r[3] - r[2]) / 1e6))
PY
