#!/bin/bash
# Per-bounce launch times of the traversal kernel from a rocprofv3 kernel trace (run on the GPU box from the repo root):
#   bash tools/bounce_times.sh <tag> [bench args]   -> gpurun_out/<tag>_bounce_times.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/bt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY' | tee $ROOT/gpurun_out/${TAG}_bounce_times.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if n.startswith("void k_traverse8_persistent") or n.startswith("void k_raygen"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n.split("(")[0]))
rows.sort()
seq, cur = [], None
for t, d, n in rows:
    if "k_raygen" in n:
        cur = []
        seq.append(cur)
    elif cur is not None:
        cur.append((d, n))
full = [s for s in seq if len(s) == max(len(x) for x in seq)]
print(f"{len(full)} batches of {len(full[0])} traversal launches (bounce 0..): mean launch time per bounce, us")
for b in range(len(full[0])):
    ds = [s[b][0] for s in full if "<8, 5, false" in s[b][1] or "false, false" in s[b][1]]
    ds = [s[b][0] for s in full]
    print(b, f"{sum(ds) / len(ds) / 1e3:.1f}", full[0][b][1][:70])
PY
