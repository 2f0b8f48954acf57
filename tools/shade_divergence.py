#!/usr/bin/env python3
"""What the waves of k_shade hold, by material (SURVEY 8f-4 / wavefront.md:92-93: are material-coherent work queues worth
building?).  The reference's own mixed-material scenes (scene.cpp:62-170: 65 % Lambertian / 25 % Metal / 10 % Dielectric
spheres) and the mesh configs.  A wave executes the scatter code of every material type it holds (material_scatter is
select-based within a type, branched between types), so "distinct scattering types per wave" is the factor by which material
divergence multiplies the scatter part of the kernel; 1.0 = nothing to win by sorting rays by material.
  python tools/shade_divergence.py [--scenes RANDOM_BALLS_LARGE,RANDOM_BALLS_SMALL,MATERIAL_TEST,C3] [--spp 16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["miss", "Lambertian", "Metal", "Dielectric", "Emissive"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="RANDOM_BALLS_LARGE,RANDOM_BALLS_SMALL,MATERIAL_TEST,C3")
    ap.add_argument("--spp", type=int, default=16)
    args = ap.parse_args()
    import parallelraytracing_amd as prt
    for name in args.scenes.split(","):
        if name.startswith("C"):
            scene, cam, W, H, _, depth = prt.scenes.config(name)
        else:
            scene, W, H, depth = prt.Scene(name), 1920, 1080, 20
            cam = prt.Camera(width=W, height=H)
        film = prt.Film(W, H)
        r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
        r.Init(film, scene, cam)
        r.set_param("measure_spp", args.spp)
        d = r.measure_shade_divergence().astype(float)
        print(f"{name} ({W}x{H}, {depth} segments, {args.spp} samples per batch):")
        tot_sc = tot_w = 0.0
        for b in range(depth):
            waves, lanes = d[b, 0], d[b, 1]
            if waves == 0:
                break
            mix = "  ".join(f"{NAMES[t]} {100 * d[b, 2 + t] / lanes:4.1f} %" for t in range(5) if d[b, 2 + t])
            div = d[b, 14] / max(1.0, d[b, 15])
            tot_sc += d[b, 14]
            tot_w += d[b, 15]
            if b < 6 or b == depth - 1:
                print(f"  bounce {b:2d}: {int(waves):8d} waves, {lanes / waves:5.1f} lanes each;  {mix};  distinct scattering types per wave {div:.2f}")
        print(f"  all bounces: {tot_sc / max(1.0, tot_w):.3f} distinct scattering types per scattering wave "
              f"(sorting rays by material could remove at most {100 * (1 - tot_w / max(1.0, tot_sc)):.0f} % of the scatter branches' issue slots)", flush=True)
        film._renderer = None
        del r, film


if __name__ == "__main__":
    main()
