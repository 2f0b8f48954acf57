#!/usr/bin/env python3
"""Does running two half-batches on two streams hide the tails of the persistent kernels?  Rank 0's share of an N-GPU
C3 step rendered (a) by one context, 256 samples per batch, (b) by two contexts on their own streams driven from two
host threads, 128 samples per batch each (same total work).
  python tools/sim_overlap.py [world ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402

torch.cuda.set_device(0)
scene, cam, W, H, spp, depth = prt.scenes.config("C3")
worlds = [int(a) for a in sys.argv[1:]] or [1, 8]


def make(world, sif):
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, rank=0, world_size=world)
    r.Init(film, scene, cam)
    r.set_samples_in_flight(sif)
    return r, film


def timed(fn, reps=4):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for world in worlds:
    one, _f = make(world, 256)

    def step_one():
        one.render_async(256)
        one.synchronize()

    t1 = timed(step_one)
    del one
    pair = [make(world, 128) for _ in range(2)]

    def half(r):
        r.render_async(128)
        r.synchronize()

    def step_two():
        th = [threading.Thread(target=half, args=(p[0],)) for p in pair]
        for t in th:
            t.start()
        for t in th:
            t.join()

    t2 = timed(step_two)
    print(f"world {world}: one context 256/batch {t1 * 1e3:.2f} ms   two contexts 128/batch each {t2 * 1e3:.2f} ms   ratio {t1 / t2:.3f}", flush=True)
    del pair
