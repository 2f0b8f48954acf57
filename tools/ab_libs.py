#!/usr/bin/env python3
"""A/B timing of alternative BUILDS of libprt.so (tools/build_variant.sh): bench.py runs in a child process per build,
the builds alternate for --rounds rounds, and the per-build medians of the headline value and the stage split are printed.
  python tools/ab_libs.py --tags base,noslp --rounds 3 -- --steps 5 --warmup 2 [more bench.py args]
`base` = the product library csrc/libprt.so."""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parallelraytracing_amd", "csrc")


def main():
    argv = sys.argv[1:]
    bench_args = []
    if "--" in argv:
        i = argv.index("--"); bench_args = argv[i + 1:]; argv = argv[:i]
    ap = argparse.ArgumentParser()
    ap.add_argument("--tags", default="base")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args(argv)
    tags = args.tags.split(",")
    res = {t: [] for t in tags}
    for rd in range(args.rounds):
        for t in tags:
            env = dict(os.environ)
            if t != "base":
                env["PRT_LIB_PATH"] = os.path.join(CSRC, "ab", f"libprt_{t}.so")
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary"] + bench_args,
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                print(f"[{t}] FAILED rc {p.returncode}\n{p.stderr[-2000:]}", flush=True)
                continue
            j = json.loads(line[-1])
            st = j["roofline"].get("stage_ms", {})
            n = j["steps"]
            row = (j["value"], j["ms_per_step"], st.get("intersect", 0) / n, st.get("shade", 0) / n, st.get("raygen", 0) / n, st.get("accumulate", 0) / n)
            res[t].append(row)
            print(f"round {rd} [{t}] {row[0]:.0f} Mrays/s  step {row[1]:.3f} ms  trav {row[2]:.3f} shade {row[3]:.3f} raygen {row[4]:.3f} acc {row[5]:.3f}", flush=True)
    for t in tags:
        if res[t]:
            med = [statistics.median(c) for c in zip(*res[t])]
            print(f"== {t}: median {med[0]:.0f} Mrays/s  step {med[1]:.3f} ms  trav {med[2]:.3f} shade {med[3]:.3f} raygen {med[4]:.3f} acc {med[5]:.3f}")


if __name__ == "__main__":
    main()
