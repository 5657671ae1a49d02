# Start / duration of the kernels of the last device batches of a bench run (rocprofv3 kernel trace): how the sweep's kernels
# overlap.  usage (on the GPU box): bash scripts/sweep_timeline.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_trace -- python3 $R/bench.py --no-cpu --no-extras --steps 2 --warmup 1 --timed resident "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("/tmp/tl_trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if re.search("lean|k_select|k_jc69_gemm|k_sweep", r["Kernel_Name"])]
t0 = int(sel[-22]["Start_Timestamp"])
for r in sel[-22:]:
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    print("%-22s start %8.3f ms  dur %7.3f ms" % (m.group(1) if m else "?", (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
