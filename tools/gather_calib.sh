#!/bin/bash
# FETCH_SIZE calibration for the traversal kernel's gathers (run on the GPU box from the repo root):
#   bash tools/gather_calib.sh      -> gpurun_out/gather_calib/r2_gather_calib.txt  (copy into profiles/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gather_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/gather_calib.hip -o $OUT/gather_calib || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $OUT/gather_calib > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/rdreq -- $OUT/gather_calib > $OUT/run2.log 2>&1 || echo "(RDREQ pass optional)"
python3 - "$OUT" <<'PY' | tee $OUT/r2_gather_calib.txt
import csv, glob, sys, collections
out = sys.argv[1]
v = collections.defaultdict(dict)
for d in ("fetch", "rdreq"):
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            v[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] = float(r["Counter_Value"])
N = 1 << 26
T = 4 << 30
print("FETCH_SIZE calibration on MI355X (tools/gather_calib.hip): 4 GiB zero-filled table, 2^26 lanes, one record per lane")
print("kernel | record B | 64-B lines touched per record x 64 | FETCH_SIZE x 1024 per record | factor to get the touched bytes")
for name, rec, touched in (("k_gather<5>", 80, 128.0), ("k_gather<3>", 48, 96.0), ("k_gather<1>", 16, 64.0)):
    c = v.get(name, {})
    if "FETCH_SIZE" in c:
        per = c["FETCH_SIZE"] * 1024.0 / N
        extra = "".join(f"  {k} per record {c[k] / N:.3f}" for k in sorted(c) if k != "FETCH_SIZE")
        print(f"{name} | {rec} | {touched:.0f} | {per:.1f} | {touched / per:.3f}{extra}")
c = v.get("k_stream", {})
if "FETCH_SIZE" in c:
    per = c["FETCH_SIZE"] * 1024.0
    print(f"k_stream (16 B/lane coalesced, whole table once) | bytes read {T} | FETCH_SIZE x 1024 = {per:.0f} | factor {T / per:.3f}")
PY
