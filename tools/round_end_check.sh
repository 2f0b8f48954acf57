#!/bin/bash
# What the driver runs at the end of a round, in one gpurun call: the GPU suite, smoke(), the default bench line (summary printed).
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t14.log 2>&1; tail -2 gpurun_out/r3_t14.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null > gpurun_out/r3_bench_end.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_bench_end.log") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stale"], d["cpu_baseline"]["value"],
      {k: (v.get("value") if isinstance(v, dict) else None) for k, v in d["secondary"].items()})
PY
