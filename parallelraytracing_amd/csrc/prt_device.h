// prt_device.h — device-side math of the hot path (gfx950 only; compiled with -ffp-contract=off).
//
// Arithmetic contract: IEEE fp32, no FMA contraction, correctly rounded div/sqrt, glm's operation
// order (the reference does all vector math through glm; src/core/core.h:22-25).  Explicit
// __builtin_fmaf appears ONLY in BVH box culling, which never changes a result (prt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prt_kernels.h"

#define PRT_DEV __device__ __forceinline__

// Streamed-once data (ray records, path results) is LOADED with the non-temporal hint, so that it does not push the tree out
// of the L2s / the Infinity Cache (-DPRT_NO_NT: plain, for A/B builds).  Measured (TUNING.md): loads C3 +-0, C5 +1.3 %;
// the same hint on the STORES made the first k_shade of a batch 39 % slower, on the vertex-normal gathers of k_shade 8 %
// slower, on the traversal's triangle gathers 24 % slower (they are re-used from L1): those stay plain.
typedef float prt_v4f __attribute__((ext_vector_type(4)));
#ifndef PRT_NO_NT
PRT_DEV float4 ld_stream(const float4* p) {
    const prt_v4f v = __builtin_nontemporal_load((const prt_v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
PRT_DEV uint32_t ld_stream(const uint32_t* p) { return __builtin_nontemporal_load(p); }
PRT_DEV float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }
#else
PRT_DEV float4 ld_stream(const float4* p) { return *p; }
PRT_DEV uint32_t ld_stream(const uint32_t* p) { return *p; }
PRT_DEV float ld_stream(const float* p) { return *p; }
#endif
PRT_DEV void st_stream(float4* p, float4 v) { *p = v; }
PRT_DEV void st_stream(uint32_t* p, uint32_t v) { *p = v; }
PRT_DEV void st_stream(float* p, float v) { *p = v; }

PRT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PRT_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PRT_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PRT_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PRT_DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PRT_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
PRT_DEV f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
PRT_DEV f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
// glm::dot(vec3) sums (x + y) + z
PRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PRT_DEV f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
// glm::normalize = v * (1 / sqrt(dot(v, v)))
PRT_DEV f3 normalize3(f3 v) { return v * (1.0f / __builtin_sqrtf(dot3(v, v))); }
// glm::reflect = I - N * dot(N, I) * 2
PRT_DEV f3 reflect3(f3 I, f3 N) { return I - N * dot3(N, I) * 2.0f; }
PRT_DEV float glm_min(float x, float y) { return (y < x) ? y : x; }

// ---- RNG contract: PCG hash chain (reference: src/backend/optix/device_types.h:109-114) --------------
PRT_DEV uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// seed 0 == pcg_hash(pixelIndex ^ (frameIndex * 719393u)), src/backend/optix/device_programs.cu:169
PRT_DEV uint32_t path_seed(uint32_t pixel, uint32_t sample, uint32_t seed) {
    return pcg_hash((pixel ^ (sample * 719393u)) + seed * 0x9E3779B9u);
}
// Random() (src/core/math.h:10-17): u = (pcg >> 8) * 2^-24 in [0,1)
PRT_DEV float rnd01(uint32_t& s) {
    s = pcg_hash(s);
    return (float)(s >> 8) * (1.0f / 16777216.0f);
}
// RandomUnitVector (src/core/math.h:26-36): rejection in [-1,1]^3, draw order x,y,z
PRT_DEV f3 random_unit_vector(uint32_t& s) {
    while (true) {
        float x = -1.0f + (1.0f - -1.0f) * rnd01(s);
        float y = -1.0f + (1.0f - -1.0f) * rnd01(s);
        float z = -1.0f + (1.0f - -1.0f) * rnd01(s);
        f3 p = mk3(x, y, z);
        float lensq = dot3(p, p);
        if (1e-8f < lensq && lensq <= 1.0f) return p / __builtin_sqrtf(lensq);
    }
}

// TransformPoint (src/core/geometry.h:145-148) with glm's mat4*vec4 grouping
PRT_DEV f3 transform_point(const float* m, f3 p) {
    f3 r;
    r.x = (m[0] * p.x + m[3] * p.y) + (m[6] * p.z + m[9] * 1.0f);
    r.y = (m[1] * p.x + m[4] * p.y) + (m[7] * p.z + m[10] * 1.0f);
    r.z = (m[2] * p.x + m[5] * p.y) + (m[8] * p.z + m[11] * 1.0f);
    return r;
}
// TransformNormal (src/core/geometry.h:139-142): normalize(mat3(transpose(M)) * n)
PRT_DEV f3 transform_normal(const float* m, f3 n) {
    f3 r;
    r.x = m[0] * n.x + m[1] * n.y + m[2] * n.z;
    r.y = m[3] * n.x + m[4] * n.y + m[5] * n.z;
    r.z = m[6] * n.x + m[7] * n.y + m[8] * n.z;
    return normalize3(r);
}

#define PRT_TMIN 0.001f  // kShapeRayTMin, src/core/shape.h:128

struct ShapeHit {
    f3 pos, normal;
    bool has, front;
};

// Circle::Intersect (src/core/shape.h:157-203)
PRT_DEV void circle_intersect(float radius, f3 o, f3 d, ShapeHit& h) {
    float a = dot3(d, d);
    float b = 2.0f * dot3(o, d);
    float c = dot3(o, o) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    h.has = false;
    h.front = false;
    if (disc >= 0.0f) {
        float sq = __builtin_sqrtf(disc);
        float t1 = (-b + sq) / (2.0f * a);
        float t2 = (-b - sq) / (2.0f * a);
        float t = 0.0f;
        h.has = true;
        if (t1 >= PRT_TMIN && t2 >= PRT_TMIN) {
            t = t1 < t2 ? t1 : t2;
            h.front = true;
        } else if (t1 >= PRT_TMIN) {
            t = t1;
        } else if (t2 >= PRT_TMIN) {
            t = t2;
        } else {
            h.has = false;
        }
        f3 p = o + d * t;
        f3 n = normalize3(p);
        if (!h.front) n = n * -1.0f;
        h.pos = p;
        h.normal = n;
    }
}

// Quad::Intersect (src/core/shape.h:213-239)
PRT_DEV void quad_intersect(float w, float hgt, f3 o, f3 d, ShapeHit& h) {
    h.has = false;
    h.front = false;
    if (__builtin_fabsf(d.y) < 1e-8f) return;
    float t = -o.y / d.y;
    f3 p = o + d * t;
    float hw = w / 2.0f;
    float hh = hgt / 2.0f;
    if (t > PRT_TMIN && (p.x * p.x < hw * hw) && (p.z * p.z < hh * hh)) {
        h.has = true;
        h.pos = p;
        h.front = o.y > 0.0f;
        h.normal = h.front ? mk3(0.0f, 1.0f, 0.0f) : mk3(-0.0f, -1.0f, -0.0f);
    }
}

// The distance metric of PrimitiveList::Intersect (src/core/primitive.cpp:42-43)
PRT_DEV float dist2(f3 o, f3 p) {
    f3 dv = o - p;
    return dot3(dv, dv);
}

// Triangle::Intersect (src/core/shape.h:262-303), position only.  Returns false when the reference
// returns without a hit.  `ld` is the primitive-local direction (identity Transform: normalize(d)).
PRT_DEV bool triangle_hit_pos(f3 P0, f3 P1, f3 P2, f3 o, f3 ld, f3& pos, float& b1o, float& b2o) {
    f3 S = o - P0;
    f3 E1 = P1 - P0;
    f3 E2 = P2 - P0;
    f3 S1 = cross3(ld, E2);
    f3 S2 = cross3(S, E1);
    float divisor = dot3(S1, E1);
    if (divisor == 0.0f) return false;
    float t = dot3(S2, E2) / divisor;
    float b1 = dot3(S1, S) / divisor;
    float b2 = dot3(S2, ld) / divisor;
    if (t < PRT_TMIN || b1 < 0.0f || b2 < 0.0f || b1 + b2 > 1.0f) return false;
    pos = (1.0f - b1 - b2) * P0 + b1 * P1 + b2 * P2;
    b1o = b1;
    b2o = b2;
    return true;
}

// Full surface interaction of one primitive hit in WORLD space: the body of the loop in
// PrimitiveList::Intersect (src/core/primitive.cpp:26-41).  `id` < n_prims: analytic primitive;
// otherwise leaf-order triangle slot id - n_prims.  Returns HasIntersection.
struct WorldHit {
    f3 pos, normal;
    float d2;
    uint32_t material;
    int32_t prim;  // global primitive index (analytic first, then triangles in mesh/face order)
    bool has, front;
};

PRT_DEV void analytic_hit(const DevPrim& p, f3 o, f3 d, WorldHit& w) {
    f3 lo = transform_point(p.inv, o);
    f3 ld = transform_normal(p.mat, d);
    ShapeHit h;
    if (p.shape_type == 0u)
        circle_intersect(p.p0, lo, ld, h);
    else
        quad_intersect(p.p0, p.p1, lo, ld, h);
    w.has = h.has;
    if (!h.has) return;
    w.pos = transform_point(p.mat, h.pos);
    w.normal = transform_normal(p.inv, h.normal);
    w.front = h.front;
    w.material = p.material;
    w.d2 = dist2(o, w.pos);
}

PRT_DEV void triangle_world_hit(const DevScene& sc, uint32_t slot, f3 o, f3 d, WorldHit& w) {
    const float4 a = sc.tris[3 * (size_t)slot + 0];
    const float4 b = sc.tris[3 * (size_t)slot + 1];
    const float4 c = sc.tris[3 * (size_t)slot + 2];
    f3 ld = normalize3(d);  // TransformNormal(identity, d)
    f3 pos;
    float b1, b2;
    w.has = triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, ld, pos, b1, b2);
    if (!w.has) return;
    const float4 n0 = sc.tri_normals[3 * (size_t)slot + 0];
    const float4 n1 = sc.tri_normals[3 * (size_t)slot + 1];
    const float4 n2 = sc.tri_normals[3 * (size_t)slot + 2];
    f3 n = (1.0f - b1 - b2) * mk3(n0.x, n0.y, n0.z) + b1 * mk3(n1.x, n1.y, n1.z) + b2 * mk3(n2.x, n2.y, n2.z);
    w.front = true;
    if (dot3(n, ld) > 0.0f) {
        n = n * -1.0f;
        w.front = false;
    }
    w.pos = pos;
    w.normal = normalize3(n);  // TransformNormal(identity, n)
    w.material = __float_as_uint(b.w);
    w.prim = (int32_t)__float_as_uint(a.w);
    w.d2 = dist2(o, pos);
}

// A triangle of a placed mesh copy: the loop body of PrimitiveList::Intersect with the copy's Transform
// (src/core/primitive.cpp:29-43): local ray, Triangle::Intersect in the mesh's space, position back through Mat, normal
// through Inv, distance in world space.  v = hit id - n_prims.
PRT_DEV void instance_world_hit(const DevScene& sc, uint32_t v, f3 o, f3 d, WorldHit& w) {
    uint32_t lo = 0, hi = sc.n_insts;  // the copy whose [virt_base, virt_base + n_tris) holds v
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sc.insts[mid].virt_base <= v) lo = mid; else hi = mid;
    }
    const DevInstance& I = sc.insts[lo];
    const uint32_t slot = I.slot_base + (v - I.virt_base);
    const float4 a = sc.tris[3 * (size_t)slot + 0];
    const float4 b = sc.tris[3 * (size_t)slot + 1];
    const float4 c = sc.tris[3 * (size_t)slot + 2];
    const f3 lo_ = transform_point(I.inv, o);
    const f3 ld = transform_normal(I.mat, d);
    f3 pos;
    float b1, b2;
    w.has = triangle_hit_pos(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), lo_, ld, pos, b1, b2);
    if (!w.has) return;
    const float4 n0 = sc.tri_normals[3 * (size_t)slot + 0];
    const float4 n1 = sc.tri_normals[3 * (size_t)slot + 1];
    const float4 n2 = sc.tri_normals[3 * (size_t)slot + 2];
    f3 n = (1.0f - b1 - b2) * mk3(n0.x, n0.y, n0.z) + b1 * mk3(n1.x, n1.y, n1.z) + b2 * mk3(n2.x, n2.y, n2.z);
    w.front = true;
    if (dot3(n, ld) > 0.0f) {
        n = n * -1.0f;
        w.front = false;
    }
    w.pos = transform_point(I.mat, pos);
    w.normal = transform_normal(I.inv, n);
    w.material = I.material == 0xFFFFFFFFu ? __float_as_uint(b.w) : I.material;  // world-space meshes: per triangle
    w.prim = (int32_t)(I.prim_base + __float_as_uint(a.w));
    w.d2 = dist2(o, w.pos);
}

template <bool INST = false>
PRT_DEV void world_hit_from_id(const DevScene& sc, uint32_t id, f3 o, f3 d, WorldHit& w) {
    if (id < sc.n_prims) {
        analytic_hit(sc.prims[id], o, d, w);
        w.prim = (int32_t)id;
    } else if (INST) {
        instance_world_hit(sc, id - sc.n_prims, o, d, w);
    } else {
        triangle_world_hit(sc, id - sc.n_prims, o, d, w);
    }
}

// ---- materials (src/core/material.h) -------------------------------------------------------------------
// fresnelReflectance (material.h:105-109): std::pow(float,int) promotes to double, i.e. the reference evaluates
// std::pow((double)(1 - cosine), 5.0) with the host's libm.  The oracle calls exactly that (oracle/prt_oracle.cpp);
// there is no glibc on the device, so this computes the CORRECTLY ROUNDED double x^5 instead: x has at most 24
// significant bits, so x*x is exact, x^4 = hi + lo exactly (one FMA error term), and x^5 = hi*x + (err + lo*x) is
// rounded once, off only when the exact value lies within ~2^-100 of a rounding midpoint.  glibc's pow is within
// 0.52 ulp of the exact value (it rounds correctly except within 0.02 ulp of a midpoint), so the two agree bit for bit
// except on measure-zero inputs; tests/test_gpu_parity.py::test_dielectric_scatter_sweep_bit_exact checks 10^6 cases.
// (The previous form x2*x2*x rounded three times and could be a double ulp off.)
PRT_DEV double pow5_rn(double x) {
    const double x2 = x * x;                           // exact: 48 significant bits
    const double hi = x2 * x2;                         // x^4 rounded
    const double lo = __builtin_fma(x2, x2, -hi);      // x^4 = hi + lo exactly
    const double p = hi * x;                           // leading part of x^5
    const double e = __builtin_fma(hi, x, -p);         // hi * x = p + e exactly
    return p + (e + lo * x);
}
PRT_DEV float fresnel_reflectance(float cosine, float ri) {
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    const double x5 = pow5_rn((double)(1.0f - cosine));
    return (float)((double)r0 + (double)(1.0f - r0) * x5);
}
// Reflect(), which is Snell refraction (src/core/math.h:45-50)
PRT_DEV f3 refract3(f3 uv, f3 n, float eta) {
    float cos_theta = glm_min(dot3(-uv, n), 1.0f);
    f3 perp = eta * (uv + cos_theta * n);
    f3 par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dot3(perp, perp))) * n;
    return perp + par;
}

// MaterialHandle::Emit + Scatter (material.h:139-161).  out_d is NOT normalised (the caller does,
// src/backend/cpu/renderer.cpp:84).  Written with selects instead of nested branches: the three
// scattering materials share the tail, and the dielectric evaluates both reflect and refract.
PRT_DEV bool material_scatter(uint32_t type, float4 rgbs, f3 in_d, f3 pos, f3 normal, bool front, uint32_t& rng,
                              f3& atten, f3& emitted, f3& out_o, f3& out_d) {
    const f3 rgb = mk3(rgbs.x, rgbs.y, rgbs.z);
    const f3 zero = mk3(0.0f, 0.0f, 0.0f);
    emitted = (type == 4u) ? rgb : zero;  // Emissive::Emit (material.h:124-127); everything else emits 0
    const bool scattering = (type >= 1u && type <= 3u);
    out_o = scattering ? pos : zero;
    atten = zero;
    out_d = mk3(0.0f, 0.0f, 1.0f);
    bool result = false;
    if (type == 1u) {  // Lambertian (material.h:16-31)
        f3 dir = normal + random_unit_vector(rng);
        const double s = 1e-8;
        if (((double)__builtin_fabsf(dir.x) < s) && ((double)__builtin_fabsf(dir.y) < s) &&
            ((double)__builtin_fabsf(dir.z) < s))
            dir = normal;
        out_d = normalize3(dir);
        atten = rgb;
        result = true;
    } else if (type == 2u) {  // Metal (material.h:48-57): the RNG is drawn even when roughness == 0
        f3 r = reflect3(in_d, normal);
        r = normalize3(r) + rgbs.w * random_unit_vector(rng);
        out_d = normalize3(r);
        atten = rgb;
        result = dot3(out_d, normal) > 0.0f;
    } else if (type == 3u) {  // Dielectric (material.h:76-95)
        atten = mk3(1.0f, 1.0f, 1.0f);
        const float ri = front ? (1.0f / rgbs.w) : rgbs.w;
        const float cos_theta = glm_min(dot3(-in_d, normal), 1.0f);
        const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
        const bool cannot = ri * sin_theta > 1.0f;
        bool do_reflect = cannot;
        if (!cannot) do_reflect = fresnel_reflectance(cos_theta, ri) > rnd01(rng);  // RNG drawn only if it can refract
        const f3 refl = reflect3(in_d, normal);
        const f3 refr = refract3(in_d, normal, ri);
        out_d = mk3(do_reflect ? refl.x : refr.x, do_reflect ? refl.y : refr.y, do_reflect ? refl.z : refr.z);
        result = true;
    }
    return result;
}

// Camera::GetCameraRay (src/core/camera.h:103-132)
PRT_DEV void camera_ray(const DevCamera& c, float px, float py, f3& o, f3& d) {
    float ndcX = (px / c.W) * 2.0f - 1.0f;
    float ndcY = 1.0f - (py / c.H) * 2.0f;
    float aspect = c.W / c.H;
    f3 dc = normalize3(mk3(ndcX * aspect * c.tan_fov_y, ndcY * c.tan_fov_y, -1.0f));
    f3 dw = dc.x * c.right + dc.y * c.up + dc.z * -c.front;
    d = normalize3(dw);
    o = c.pos;
}

// ---- wave64 helpers --------------------------------------------------------------------------------------
PRT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave-aggregated slot allocation: one atomic per wave (the wave64 form of the reference's
// warp-aggregated AllocateSlot, src/backend/cuda_wavefront/renderer.cu:43-67).  Must be called by all
// active lanes of the wave; lanes with want == false get an undefined slot.
PRT_DEV uint32_t wave_alloc(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    const uint32_t n = (uint32_t)__popcll(mask);
    if (n == 0) return 0;
    const uint32_t lane = lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, n);
    base = (uint32_t)__shfl((int)base, leader, 64);
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    return base + (uint32_t)__popcll(below);
}
