"""The error bound behind the primitive-BVH walk's culling pad, checked in EXACT arithmetic on the CPU.

The walk over the analytic primitives' world boxes (csrc/prt_kernels.hip scan_analytic<ABVH>) may only skip a primitive
the reference's arithmetic cannot hit.  Circle::Intersect (src/core/shape.h:157-203) decides with the SIGN of the fp32
value  disc = b*b - 4*a*c,  a = d.d, b = 2 o.d, c = o.o - r*r  (shape.h:160-163), which cancels for a far origin: a ray that
passes OUTSIDE the sphere can still be a hit of the reference.  DESIGN.md derives, with u = 2^-24,

    |fl(b*b) - fl(4*a*c)  -  disc_exact|  <=  E = u * |d|^2 * (60 |o|^2 + 24 r^2)          (first order in u)
    =>  a reported hit implies  rho - r  <=  E / (8 |d|^2 r)  =  u * (7.5 |o|^2 / r + 3 r)   (rho = distance centre-line)

and the host pads every ray by K/R * (|o|_1 + |c|_1)^2 with K = 1e-6 >= 7.5 u = 4.47e-7 (csrc/prt_api.cpp, DevScene::abvh_q;
the 3 u r term is inside the boxes' own relative slack of 1e-5).  These tests evaluate both inequalities with
fractions.Fraction on adversarial near-tangent rays from 0.5 to 10^4 radii away, replaying the reference's fp32 operation
order with numpy.float32 (every numpy float32 operation is correctly rounded, as on the device with contraction off).
Quad::Intersect's t = -o.y / d.y (shape.h:221-224) is covered by the second test: its error is LINEAR in the distance.
"""
from fractions import Fraction as Fr

import numpy as np
import pytest

f32 = np.float32
U = Fr(1, 2 ** 24)          # unit roundoff of fp32, round to nearest
K_PAD = Fr(1, 10 ** 6)      # csrc/prt_api.cpp: quad_pad = 1e-6 / R * (|o|_1 + |c|_1)^2


def fr(x):
    return Fr(float(x))     # exact: every float is a rational


def disc_fp32(o, d, r):
    """shape.h:160-163 in the reference's operation order (glm::dot = (x*x' + y*y') + z*z')."""
    def dot(p, q):
        return f32(f32(f32(p[0] * q[0]) + f32(p[1] * q[1])) + f32(p[2] * q[2]))
    a = dot(d, d)
    b = f32(f32(2.0) * dot(o, d))
    c = f32(dot(o, o) - f32(r * r))
    t1 = f32(b * b)
    t2 = f32(f32(f32(4.0) * a) * c)
    return f32(t1 - t2), t1, t2


def exact_terms(o, d, r):
    O = [fr(v) for v in o]
    D = [fr(v) for v in d]
    R = fr(r)
    od = sum(x * y for x, y in zip(O, D))
    dd = sum(y * y for y in D)
    oo = sum(x * x for x in O)
    disc = 4 * od * od - 4 * dd * (oo - R * R)
    rho2 = oo - od * od / dd   # squared distance of the centre (local origin) from the line
    return disc, dd, oo, rho2


def near_tangent_rays(rng, n, r, dist, spread):
    """Rays whose line passes the sphere of radius r (centre = local origin) at r * (1 + spread * N(0,1) * bound-ish) from a
    point `dist` away: unit directions, everything rounded to fp32 like the reference's inputs."""
    out = []
    for _ in range(n):
        # tangent point p on the sphere, tangent direction t, origin = p * (1 + eps) - t * sqrt(dist^2 - r^2)
        p = rng.normal(size=3)
        p /= np.linalg.norm(p)
        t = np.cross(p, rng.normal(size=3))
        t /= np.linalg.norm(t)
        eps = spread * rng.normal() * 3e-7 * (dist / r) ** 2      # around the band where the sign of disc is rounding noise
        s = np.sqrt(max(dist * dist - r * r, 0.0))
        o = (p * r * (1.0 + eps) - t * s).astype(np.float32)
        d = t.astype(np.float32)
        # the reference's ray directions are normalised in fp32 (Ray::Normalize / TransformNormal)
        inv = f32(1.0) / np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))
        d = (d * inv).astype(np.float32)
        out.append((o, d))
    return out


@pytest.mark.parametrize("r,dists", [(1.0, (0.5, 3.0, 30.0, 300.0, 3000.0)), (0.05, (0.2, 5.0, 60.0, 500.0)),
                                      (3.0, (4.0, 40.0, 1000.0, 30000.0))])
def test_discriminant_error_bound_and_phantom_hit_margin(r, dists):
    rng = np.random.default_rng(1234)
    r32 = f32(r)
    worst_E, worst_gap, phantoms = Fr(0), Fr(0), 0
    for dist in dists:
        for o, d in near_tangent_rays(rng, 400, float(r32), max(dist, 1.001 * r) if dist > r else dist, 1.0):
            disc32, t1, t2 = disc_fp32(o, d, r32)
            disc, dd, oo, rho2 = exact_terms(o, d, r32)
            R2 = fr(r32) ** 2
            # (1) forward error of the two rounded terms against the derived E (with the second-order factor)
            E = U * dd * (60 * oo + 24 * R2) * (1 + 64 * U)
            err = abs(fr(t1) - fr(t2) - disc)
            assert err <= E, (dist, float(err), float(E))
            if E > 0:
                worst_E = max(worst_E, err / E)
            # (2) the sign of the fp32 discriminant is the sign of t1 - t2 (a subtraction never flips a sign)
            assert (disc32 >= 0) == (fr(t1) - fr(t2) >= 0)
            # (3) a hit the exact geometry does not have (disc32 >= 0, line outside the sphere): how far outside?
            if disc32 >= 0 and rho2 > R2:
                phantoms += 1
                gap_bound = U * (Fr(15, 2) * oo / fr(r32) + 3 * fr(r32)) * (1 + 64 * U)
                # rho - r = (rho^2 - r^2) / (rho + r) <= (rho^2 - r^2) / (2 r)
                gap_upper = (rho2 - R2) / (2 * fr(r32))
                assert gap_upper <= gap_bound, (dist, float(gap_upper), float(gap_bound))
                # and the product's pad covers it: K / r * |o|_1^2 >= K / r * |o|_2^2 (here the centre is the local origin)
                assert gap_upper <= K_PAD / fr(r32) * oo + Fr(1, 10 ** 5) * fr(r32)
                worst_gap = max(worst_gap, gap_upper / (K_PAD / fr(r32) * oo + Fr(1, 10 ** 5) * fr(r32)))
    assert phantoms > 20, "the adversarial rays must actually produce phantom hits"
    assert worst_E <= 1 and worst_gap < Fr(3, 4), (float(worst_E), float(worst_gap))


def test_pad_constant_dominates_the_derived_coefficient():
    # K = 1e-6 against 7.5 u = 4.47e-7: a factor 2.2 of margin; the linear pad 2^-18 against the ~10 u of the transforms
    assert K_PAD >= 2 * Fr(15, 2) * U
    assert Fr(1, 2 ** 18) >= 6 * 10 * U


def test_quad_plane_distance_error_is_linear_in_the_distance():
    """Quad::Intersect: t = -o.y / d.y, p = o + d * t (shape.h:221-226).  For |d.y| >= 1e-8 (shape.h:218) the computed point
    is within 3 u (|o| + |t| |d|) of the exact point of the SAME fp32 ray at the exact parameter, i.e. linear in the distance:
    the walk's linear pad 2^-18 (|o|_1 + extent) = 64 u (...) covers it; no quadratic term is needed."""
    rng = np.random.default_rng(99)
    worst = Fr(0)
    for _ in range(4000):
        scale = 10.0 ** rng.uniform(-1, 3.5)
        o = (rng.normal(size=3) * scale).astype(np.float32)
        d = rng.normal(size=3)
        d[1] *= 10.0 ** rng.uniform(-6, 0)       # down to grazing: cos ~ 1e-6
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if abs(float(d[1])) < 1e-8:
            continue
        t = f32(f32(-o[1]) / d[1])
        p = [f32(o[i] + f32(d[i] * t)) for i in range(3)]
        O, D = [fr(v) for v in o], [fr(v) for v in d]
        T = -O[1] / D[1]
        P = [O[i] + D[i] * T for i in range(3)]
        err = max(abs(fr(p[i]) - P[i]) for i in range(3))
        mag = max(abs(x) for x in O) + abs(T) * max(abs(y) for y in D)
        worst = max(worst, err / (U * mag))
        assert err <= 3 * U * mag * (1 + 16 * U), (float(err), float(U * mag))
    assert worst > Fr(1, 10)  # the bound is not vacuous
