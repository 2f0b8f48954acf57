#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: 1, 2 and 4 ranks (gloo, all on GPU 0) render one C3 frame each
# and dump it; the three PFM files must be byte-identical (the image does not depend on how the tiles are dealt).
#   gpurun -- 'bash tools/rehearse_ranks.sh'
set -e
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --dump gpurun_out/reh_n1 > gpurun_out/reh_n1.log 2>&1
PRT_BENCH_SAME_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --backend gloo --dump gpurun_out/reh_n2 > gpurun_out/reh_n2.log 2>&1
PRT_BENCH_SAME_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --steps 1 --warmup 0 --no-cpu-baseline --backend gloo --dump gpurun_out/reh_n4 > gpurun_out/reh_n4.log 2>&1
python - <<'PY'
import numpy as np, hashlib
for n in (1,2,4):
    b=open(f'gpurun_out/reh_n{n}.pfm','rb').read()
    print(n, len(b), hashlib.sha256(b).hexdigest()[:16])
PY
tail -1 gpurun_out/reh_n2.log | cut -c1-160
tail -1 gpurun_out/reh_n4.log | cut -c1-160
rm -f gpurun_out/reh_n*.pfm gpurun_out/reh_n*.ppm
