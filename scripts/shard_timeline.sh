# Kernels AND copies of the last step of a bench run, in time order (rocprofv3 kernel + memory-copy trace): where a 12 500-query
# shard's fixed costs sit.  usage (on the GPU box): APPLES_BENCH_SHARD=1 bash scripts/shard_timeline.sh --workload c3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/st_trace
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/st_trace -- python3 $R/bench.py --no-cpu --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
ev = []
for f in glob.glob("/tmp/st_trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
for f in glob.glob("/tmp/st_trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s" % r.get("Direction", "")))
ev.sort()
gemm = [i for i, e in enumerate(ev) if e[2] == "k_jc69_gemm"]
lo = gemm[-1]
while lo > 0 and ev[gemm[-1]][0] - ev[lo - 1][0] < 9000000: lo -= 1  # from 9 ms ahead of the last distance pass (the step before)
t0 = ev[gemm[-1]][0]
hi = lo
while hi < len(ev) and ev[hi][0] - t0 < 9000000: hi += 1
for s, e, n in ev[lo:hi]:
    print("%-28s start %8.3f ms  dur %7.3f ms" % (n, (s - t0) / 1e6, (e - s) / 1e6))
PY
