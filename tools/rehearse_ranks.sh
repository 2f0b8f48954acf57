#!/bin/bash
# Rehearsal of the N-rank bench path on ONE GPU at the headline's full size (all ranks share GPU 0, gloo transport, bench.py's own launcher):
# what the first multi-GPU run will execute, minus RCCL.  Prints ranks_seen, rays and the value per N (the value itself means nothing here:
# the ranks time-share one GPU).
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for n in 1 2 4; do
  PRT_BENCH_SAME_DEVICE=1 python bench.py --gpus $n --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dump gpurun_out/rehearse_$n 2> gpurun_out/rehearse_$n.err > gpurun_out/rehearse_$n.json || { tail -5 gpurun_out/rehearse_$n.err; exit 1; }
  python - $n <<'PY'
import json, sys, hashlib
n = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/rehearse_{n}.json") if l.startswith("{")][-1])
h = hashlib.sha256(open(f"gpurun_out/rehearse_{n}.pfm", "rb").read()).hexdigest()[:16]
print(f"N={n}: ranks_seen {d['ranks_seen']}, n_gpus {d['n_gpus']}, rays_timed {d['config']['rays_timed']}, {d['value']:.0f} Mrays/s (one shared GPU), frame sha {h}", flush=True)
PY
  rm -f gpurun_out/rehearse_$n.pfm gpurun_out/rehearse_$n.ppm
done
