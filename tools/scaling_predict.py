#!/usr/bin/env python3
"""Strong-scaling prediction from ONE GPU: every rank's share of the fixed frame, rendered one after the other.

  python tools/scaling_predict.py [--config C3] [--spp 256] [--worlds 1,2,4,8]

bench.py --gpus N tiles the fixed frame over N ranks (8x8 tiles round-robin) and the step ends when the slowest rank
is done, so step(N) = max over ranks of that rank's time for its tiles (+ the film gather, a few 10 us of wire time).
Each rank's share runs here on the one GPU this box has, with the settings of the bench line; the predicted aggregate
rate is rays(all ranks) / max rank time.  What this cannot show: RCCL itself and contention for the host.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--params", default="")
    args = ap.parse_args()
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, _, depth = prt.scenes.config(args.config)
    base = None
    for world in [int(w) for w in args.worlds.split(",")]:
        times, rays = [], 0
        for rank in range(world):
            film = prt.Film(W, H)
            r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0, rank=rank, world_size=world)
            r.Init(film, scene, cam)
            for kv in filter(None, args.params.split(",")):
                k, v = kv.split("=")
                r.set_param(k, int(v))
            r.set_samples_in_flight(min(args.spp, 256))
            r.render_async(args.spp)
            r.synchronize()
            best = 1e30
            for _ in range(args.reps):
                r.reset_stats()
                t0 = time.perf_counter()
                r.render_async(args.spp)
                r.synchronize()
                best = min(best, time.perf_counter() - t0)
            times.append(best)
            rays += int(r.stats().rays_total)
            del r, film
        rate = rays / max(times) / 1e6
        if base is None:
            base = rate / world
        print(f"world {world}: rank times min {min(times) * 1e3:.2f} max {max(times) * 1e3:.2f} ms, rays {rays}, "
              f"predicted {rate:.0f} Mrays/s = {rate / (base * world):.3f} of linear", flush=True)


if __name__ == "__main__":
    main()
