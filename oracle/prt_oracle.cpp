// prt_oracle.cpp — CPU restatement of the reference hot path (TEST INFRASTRUCTURE, NOT PRODUCT).
// See prt_oracle.h for the rules and the "PARITY UNPINNED" statement.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (no FMA contraction: the parity
// target is the reference's CPU backend built without contraction).  Scalar fp32 throughout,
// written in glm's operation order (restated from glm's public sources; glm is an un-vendored,
// un-pinned submodule of the reference: .gitmodules:7-9).
#include "prt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// glm-order vector helpers
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
inline V3 cross(V3 x, V3 y) {
    return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
// glm::reflect: I - N * dot(N, I) * 2
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
inline float gmin(float x, float y) { return (y < x) ? y : x; }  // glm::min
inline float gmax(float x, float y) { return (x < y) ? y : x; }  // glm::max

// TransformPoint: vec3(mat * vec4(p,1)) with glm's mat4*vec4 order (geometry.h:145-148)
inline V3 transform_point(const float* m, V3 p) {
    V3 r;
    r.x = (m[0] * p.x + m[4] * p.y) + (m[8] * p.z + m[12] * 1.0f);
    r.y = (m[1] * p.x + m[5] * p.y) + (m[9] * p.z + m[13] * 1.0f);
    r.z = (m[2] * p.x + m[6] * p.y) + (m[10] * p.z + m[14] * 1.0f);
    return r;
}
// TransformNormal: normalize(mat3(transpose(M)) * n)  (geometry.h:139-142)
inline V3 transform_normal(const float* m, V3 n) {
    V3 r;
    r.x = m[0] * n.x + m[1] * n.y + m[2] * n.z;
    r.y = m[4] * n.x + m[5] * n.y + m[6] * n.z;
    r.z = m[8] * n.x + m[9] * n.y + m[10] * n.z;
    return normalize(r);
}

// ---------------------------------------------------------------------------------------------
// RNG contract
// ---------------------------------------------------------------------------------------------
inline uint32_t pcg_hash(uint32_t input) {  // optix/device_types.h:109-114
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// Random() (math.h:10-17) with rand() replaced by the PCG chain: u = (pcg >> 8) * 2^-24 in [0,1)
inline float rnd(uint32_t* s) {
    *s = pcg_hash(*s);
    return (float)(*s >> 8) * (1.0f / 16777216.0f);
}
inline float rnd_range(float mn, float mx, uint32_t* s) { return mn + (mx - mn) * rnd(s); }  // math.h:19-23
inline V3 random_unit_vector(uint32_t* s) {  // math.h:26-36
    const float epsilon = 1e-8f;
    while (true) {
        float x = rnd_range(-1.0f, 1.0f, s);
        float y = rnd_range(-1.0f, 1.0f, s);
        float z = rnd_range(-1.0f, 1.0f, s);
        V3 p = v3(x, y, z);
        float lensq = dot(p, p);
        if (epsilon < lensq && lensq <= 1.0f) return p / std::sqrt(lensq);
    }
}

constexpr float kShapeRayTMin = 0.001f;  // shape.h:128

struct SI {  // surface_interaction.h:6-13
    V3 pos{0, 0, 0};
    V3 normal{0, 0, 0};
    bool has = false;
    bool front = false;
    uint32_t material = 0xFFFFFFFFu;
};

// Circle::Intersect (shape.h:157-203)
inline void circle_intersect(float radius, V3 o, V3 d, SI* si) {
    V3 l = o;
    float a = dot(d, d);
    float b = 2.0f * dot(l, d);
    float c = dot(l, l) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc >= 0.0f) {
        float t1 = (-b + sqrtf(disc)) / (2 * a);
        float t2 = (-b - sqrtf(disc)) / (2 * a);
        float t = 0.0f;
        si->has = true;
        if (t1 >= kShapeRayTMin && t2 >= kShapeRayTMin) {
            t = t1 < t2 ? t1 : t2;
            si->front = true;
        } else if (t1 >= kShapeRayTMin) {
            t = t1;
            si->front = false;
        } else if (t2 >= kShapeRayTMin) {
            t = t2;
            si->front = false;
        } else {
            si->has = false;
        }
        V3 p = o + d * t;
        V3 n = normalize(p);
        if (!si->front) n = n * -1.0f;
        si->pos = p;
        si->normal = n;
    } else {
        si->has = false;
    }
}

// Quad::Intersect (shape.h:213-239)
inline void quad_intersect(float w, float h, V3 o, V3 d, SI* si) {
    if (fabsf(d.y) < 1e-8f) {
        si->has = false;
        return;
    }
    float t = -o.y / d.y;
    V3 p = o + d * t;
    float hw = w / 2.0f;
    float hh = h / 2.0f;
    if (t > kShapeRayTMin && (p.x * p.x < hw * hw) && (p.z * p.z < hh * hh)) {
        si->has = true;
        si->pos = p;
        si->front = o.y > 0.0f;
        si->normal = si->front ? v3(0, 1, 0) : v3(-0.0f, -1.0f, -0.0f);
    } else {
        si->has = false;
    }
}

// Triangle::Intersect (shape.h:262-303)
inline void triangle_intersect(V3 P0, V3 P1, V3 P2, V3 N0, V3 N1, V3 N2, V3 o, V3 d, SI* si) {
    V3 S = o - P0;
    V3 E1 = P1 - P0;
    V3 E2 = P2 - P0;
    V3 S1 = cross(d, E2);
    V3 S2 = cross(S, E1);
    float divisor = dot(S1, E1);
    if (divisor == 0) return;
    float t = dot(S2, E2) / divisor;
    float b1 = dot(S1, S) / divisor;
    float b2 = dot(S2, d) / divisor;
    if (t < kShapeRayTMin || b1 < 0.0f || b2 < 0.0f || b1 + b2 > 1.0f) return;
    si->pos = (1 - b1 - b2) * P0 + b1 * P1 + b2 * P2;
    si->normal = (1 - b1 - b2) * N0 + b1 * N1 + b2 * N2;
    si->has = true;
    if (dot(si->normal, d) > 0.0f) {
        si->normal = si->normal * -1.0f;
        si->front = false;
    } else {
        si->front = true;
    }
}

// ---------------------------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------------------------
struct Tri {
    V3 p[3];
    V3 n[3];
    uint32_t material;
};
struct BNode {
    float bmin[3], bmax[3];
    int32_t left, right;   // children (internal) or -1
    uint32_t first, count; // leaf range in order[]
};
}  // namespace

struct OrcMesh {  // one instanced mesh in its own space, with the oracle's own BVH over it
    std::vector<Tri> tris;
    std::vector<BNode> nodes;
    std::vector<uint32_t> order;
    float pad = 0.0f;
};
struct OrcInstance {  // PrtInstance + the global index of its first triangle primitive
    uint32_t mesh, material;
    float mat[16], inv[16];
    uint32_t prim_base;
    float scale;  // uniform scale of mat (BVH culling only)
};

struct OrcScene {
    std::vector<PrtMaterial> materials;
    std::vector<PrtPrimitive> prims;
    std::vector<Tri> tris;  // global prim index = prims.size() + i
    float sky[3];
    // acceleration (oracle's own; result-equivalent to the linear scan)
    std::vector<BNode> nodes;
    std::vector<uint32_t> order;
    float pad = 0.0f;
    // placed mesh copies (PrtInstance): global prim index = instances[i].prim_base + face
    std::vector<OrcMesh> imeshes;
    std::vector<OrcInstance> instances;
};

namespace {

const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

struct Best {
    float d2 = FLT_MAX;
    int32_t prim = -1;
    SI si;
};

// One iteration of the loop body of PrimitiveList::Intersect (primitive.cpp:26-49) for primitive
// `index` with transform (mat, inv).  The tie rule "strict <, first primitive wins" is restated as
// "(d2 < best) or (d2 == best and index < best index)" so that the visiting order does not matter.
template <class F>
inline void test_primitive(const float* mat, const float* inv, V3 o, V3 d, int32_t index, uint32_t material, Best* best,
                           F&& shape) {
    V3 lo = transform_point(inv, o);
    V3 ld = transform_normal(mat, d);
    SI si;
    shape(lo, ld, &si);
    if (!si.has) return;
    si.pos = transform_point(mat, si.pos);
    si.normal = transform_normal(inv, si.normal);
    si.material = material;
    V3 dv = o - si.pos;
    float d2 = dot(dv, dv);
    if (d2 < best->d2 || (d2 == best->d2 && best->prim >= 0 && index < best->prim)) {
        best->d2 = d2;
        best->prim = index;
        best->si = si;
    }
}

inline void test_analytic(const OrcScene* s, uint32_t i, V3 o, V3 d, Best* best) {
    const PrtPrimitive& p = s->prims[i];
    if (p.shape_type == PRT_SHAPE_CIRCLE) {
        float r = p.shape_param[0];
        test_primitive(p.mat, p.inv, o, d, (int32_t)i, p.material_id, best,
                       [r](V3 lo, V3 ld, SI* si) { circle_intersect(r, lo, ld, si); });
    } else if (p.shape_type == PRT_SHAPE_QUAD) {
        float w = p.shape_param[0], h = p.shape_param[1];
        test_primitive(p.mat, p.inv, o, d, (int32_t)i, p.material_id, best,
                       [w, h](V3 lo, V3 ld, SI* si) { quad_intersect(w, h, lo, ld, si); });
    }
}

inline void test_triangle(const OrcScene* s, uint32_t ti, V3 o, V3 d, Best* best) {
    const Tri& t = s->tris[ti];
    test_primitive(kIdentity, kIdentity, o, d, (int32_t)(s->prims.size() + ti), t.material, best,
                   [&t](V3 lo, V3 ld, SI* si) {
                       triangle_intersect(t.p[0], t.p[1], t.p[2], t.n[0], t.n[1], t.n[2], lo, ld, si);
                   });
}

// ---- oracle BVH (median split; conservative culling) --------------------------------------------
template <class S>
void build_bvh(S* s, const std::vector<PrtPrimitive>& prims) {
    const uint32_t n = (uint32_t)s->tris.size();
    s->nodes.clear();
    s->order.resize(n);
    if (n == 0) return;
    std::vector<float> cx(n), cy(n), cz(n);
    float scale = 1.0f;
    for (uint32_t i = 0; i < n; ++i) {
        s->order[i] = i;
        const Tri& t = s->tris[i];
        cx[i] = (t.p[0].x + t.p[1].x + t.p[2].x);
        cy[i] = (t.p[0].y + t.p[1].y + t.p[2].y);
        cz[i] = (t.p[0].z + t.p[1].z + t.p[2].z);
        for (int k = 0; k < 3; ++k) {
            scale = std::max(scale, std::max(fabsf(t.p[k].x), std::max(fabsf(t.p[k].y), fabsf(t.p[k].z))));
        }
    }
    for (const PrtPrimitive& p : prims) {
        for (int k = 12; k < 15; ++k) scale = std::max(scale, fabsf(p.mat[k]) + 1.0f);
    }
    // Conservative padding: generous (the oracle favours safety over speed).
    s->pad = scale * 1e-4f;
    struct Item {
        uint32_t node, first, count;
    };
    std::vector<Item> stack;
    s->nodes.push_back(BNode{});
    stack.push_back({0, 0, n});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t k = it.first; k < it.first + it.count; ++k) {
            uint32_t ti = s->order[k];
            const Tri& t = s->tris[ti];
            for (int v = 0; v < 3; ++v) {
                const float c[3] = {t.p[v].x, t.p[v].y, t.p[v].z};
                for (int a = 0; a < 3; ++a) {
                    mn[a] = std::min(mn[a], c[a]);
                    mx[a] = std::max(mx[a], c[a]);
                }
            }
            const float cc[3] = {cx[ti], cy[ti], cz[ti]};
            for (int a = 0; a < 3; ++a) {
                cmn[a] = std::min(cmn[a], cc[a]);
                cmx[a] = std::max(cmx[a], cc[a]);
            }
        }
        BNode nd{};
        for (int a = 0; a < 3; ++a) {
            nd.bmin[a] = mn[a] - s->pad;
            nd.bmax[a] = mx[a] + s->pad;
        }
        nd.left = nd.right = -1;
        nd.first = it.first;
        nd.count = it.count;
        int axis = 0;
        float ext = cmx[0] - cmn[0];
        for (int a = 1; a < 3; ++a)
            if (cmx[a] - cmn[a] > ext) {
                ext = cmx[a] - cmn[a];
                axis = a;
            }
        if (it.count > 4 && ext > 0.0f) {
            const std::vector<float>& c = axis == 0 ? cx : (axis == 1 ? cy : cz);
            uint32_t mid = it.count / 2;
            std::nth_element(s->order.begin() + it.first, s->order.begin() + it.first + mid,
                             s->order.begin() + it.first + it.count,
                             [&c](uint32_t a, uint32_t b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
            nd.left = (int32_t)s->nodes.size();
            nd.right = nd.left + 1;
            nd.count = 0;
            s->nodes[it.node] = nd;
            s->nodes.push_back(BNode{});
            s->nodes.push_back(BNode{});
            stack.push_back({(uint32_t)nd.left, it.first, mid});
            stack.push_back({(uint32_t)nd.right, it.first + mid, it.count - mid});
        } else {
            s->nodes[it.node] = nd;
        }
    }
}

// Entry distance of the ray into a (padded) box, or +inf if missed.  Written to be conservative:
// any NaN (0 * inf) drops out of the min/max chain; the interval test allows equality.
inline float box_entry(const BNode& b, V3 o, V3 inv, float tlimit) {
    float tn = 0.0f, tf = tlimit;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
        float t0 = (b.bmin[a] - oo[a]) * ii[a];
        float t1 = (b.bmax[a] - oo[a]) * ii[a];
        float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
        tn = fmaxf(tn, lo);  // fmaxf ignores NaN
        tf = fminf(tf, hi);
    }
    return (tn <= tf * 1.0000005f) ? tn : INFINITY;
}

inline float limit_from_d2(float d2, float pad) {
    if (!(d2 < FLT_MAX)) return FLT_MAX;
    return std::sqrt(d2) * 1.0001f + 4.0f * pad + 1e-6f;
}

// One placed triangle = one iteration of PrimitiveList::Intersect with the instance's Transform.
inline void test_instance_triangle(const OrcScene* s, const OrcInstance& in, uint32_t face, V3 o, V3 d, Best* best) {
    const Tri& t = s->imeshes[in.mesh].tris[face];
    test_primitive(in.mat, in.inv, o, d, (int32_t)(in.prim_base + face), in.material, best,
                   [&t](V3 lo, V3 ld, SI* si) {
                       triangle_intersect(t.p[0], t.p[1], t.p[2], t.n[0], t.n[1], t.n[2], lo, ld, si);
                   });
}

// Walks the BVH of world-space triangles (in == nullptr) or of one instance's mesh in the mesh's own space.
// (o, d) is always the WORLD ray: the leaf tests transform it themselves, exactly as the reference does per primitive.
template <class S>
void traverse_bvh_of(const OrcScene* scene, const S* s, const OrcInstance* in, V3 o, V3 d, Best* best) {
    if (s->nodes.empty()) return;
    // The triangles' local ray (primitive.cpp:29-30): identity Transform: origin unchanged, direction re-normalised.
    // Culling uses that local ray; results never depend on it.
    const V3 lo = in ? transform_point(in->inv, o) : o;
    V3 ld = transform_normal(in ? in->mat : kIdentity, d);
    V3 inv = v3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    // world distance = scale * local ray parameter (rotation + uniform scale): a generous local bound
    auto limit = [&]() {
        const float w = limit_from_d2(best->d2, scene->pad);
        if (!in || !(w < FLT_MAX)) return w;
        return w / in->scale * 1.001f + 4.0f * s->pad + 1e-6f;
    };
    float tlimit = limit();
    while (sp > 0) {
        const BNode& nd = s->nodes[stack[--sp]];
        float te = box_entry(nd, lo, inv, tlimit);
        if (te == INFINITY) continue;
        if (nd.left < 0) {
            for (uint32_t k = nd.first; k < nd.first + nd.count; ++k) {
                if (in)
                    test_instance_triangle(scene, *in, s->order[k], o, d, best);
                else
                    test_triangle(scene, s->order[k], o, d, best);
            }
            tlimit = limit();
        } else {
            if (sp + 2 > 128) {  // cannot happen for median split of < 2^60 triangles
                sp = 0;
                break;
            }
            stack[sp++] = nd.left;
            stack[sp++] = nd.right;
        }
    }
}

void closest_hit(const OrcScene* s, V3 o, V3 d, int use_bvh, Best* best) {
    for (uint32_t i = 0; i < s->prims.size(); ++i) test_analytic(s, i, o, d, best);
    if (use_bvh) {
        traverse_bvh_of(s, s, (const OrcInstance*)nullptr, o, d, best);
        for (const OrcInstance& in : s->instances) traverse_bvh_of(s, &s->imeshes[in.mesh], &in, o, d, best);
    } else {
        for (uint32_t i = 0; i < s->tris.size(); ++i) test_triangle(s, i, o, d, best);
        for (const OrcInstance& in : s->instances)
            for (uint32_t f = 0; f < s->imeshes[in.mesh].tris.size(); ++f) test_instance_triangle(s, in, f, o, d, best);
    }
    if (!(best->d2 < FLT_MAX)) {  // primitive.cpp:51-58
        best->prim = -1;
        best->si.has = false;
    }
}

// fresnelReflectance (material.h:105-109): glm::pow(float,int) is std::pow -> double arithmetic,
// converted to float by the function's return type.  std::pow's last bit is the host libm's business (glibc 2.35 is
// within 0.52 ulp and differs from the correctly rounded x^5 on 0.09 % of float inputs by one double ulp; MSVC's differs
// again), so the CONTRACT both this oracle and the HIP path (csrc/prt_device.h, same text) follow is the correctly
// rounded value: x has <= 24 significant bits, x*x is exact, x^4 = hi + lo exactly, x^5 = hi*x + (err + lo*x) rounded
// once.  fresnel_reflectance_libm() is the literal std::pow form and is what THIS oracle's scatter / trace call (round 3:
// the checker follows the reference's text, the product the correctly rounded value); orc_fresnel() exposes the rounded form
// and tests/test_oracle_kat.py checks that the two agree after the conversion to float on 10^6 inputs (a double ulp
// survives that conversion with probability ~2^-29).
inline double pow5_rn(double x) {
    const double x2 = x * x;
    const double hi = x2 * x2;
    const double lo = std::fma(x2, x2, -hi);
    const double p = hi * x;
    const double e = std::fma(hi, x, -p);
    return p + (e + lo * x);
}
inline float fresnel_reflectance(float cosine, float ri) {
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return (float)((double)r0 + (double)(1 - r0) * pow5_rn((double)(1 - cosine)));
}
inline float fresnel_reflectance_libm(float cosine, float ri) {  // material.h:105-109 verbatim, host libm
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return (float)((double)r0 + (double)(1 - r0) * std::pow((double)(1 - cosine), 5));
}
// Reflect == refraction (math.h:45-50)
inline V3 refract_dir(V3 uv, V3 n, float eta) {
    float cos_theta = gmin(dot(-uv, n), 1.0f);
    V3 perp = eta * (uv + cos_theta * n);
    V3 par = -std::sqrt(std::fabs(1.0f - dot(perp, perp))) * n;
    return perp + par;
}

// MaterialHandle::Emit (material.h:139-148)
inline V3 emit(const PrtMaterial& m) {
    if (m.type == PRT_MAT_EMISSIVE) return ld3(m.rgb);
    return v3(0, 0, 0);
}

// MaterialHandle::Scatter (material.h:150-161) -> per-material Scatter
inline bool scatter(const PrtMaterial& m, V3 in_d, const SI& si, uint32_t* rng, V3* atten, V3* out_o, V3* out_d) {
    switch (m.type) {
        case PRT_MAT_LAMBERTIAN: {  // material.h:16-31
            V3 dir = si.normal + random_unit_vector(rng);
            double s = 1e-8;
            if (((double)fabsf(dir.x) < s) && ((double)fabsf(dir.y) < s) && ((double)fabsf(dir.z) < s)) dir = si.normal;
            *out_o = si.pos;
            *out_d = normalize(dir);
            *atten = ld3(m.rgb);
            return true;
        }
        case PRT_MAT_METAL: {  // material.h:48-57
            V3 r = reflect(in_d, si.normal);
            r = normalize(r) + m.scalar * random_unit_vector(rng);
            *out_o = si.pos;
            *out_d = normalize(r);
            *atten = ld3(m.rgb);
            return dot(*out_d, si.normal) > 0.0f;
        }
        case PRT_MAT_DIELECTRIC: {  // material.h:76-95
            *atten = v3(1.0f, 1.0f, 1.0f);
            float ri = si.front ? (1.0f / m.scalar) : m.scalar;
            V3 ud = in_d;
            float cos_theta = gmin(dot(-ud, si.normal), 1.0f);
            float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
            bool cannot = ri * sin_theta > 1.0f;
            V3 dir;
            // the reference's text verbatim, host libm (the oracle stays literal; the HIP path computes the correctly rounded
            // x^5 instead, csrc/prt_device.h: the two agree bit for bit except on inputs where glibc's pow is not correctly
            // rounded AND the double ulp survives the conversion to float, ~1e-12 per event; the device-vs-oracle sweep of
            // 10^6 dielectric events therefore compares the device with the reference's own form)
            if (cannot || fresnel_reflectance_libm(cos_theta, ri) > rnd(rng))
                dir = reflect(ud, si.normal);
            else
                dir = refract_dir(ud, si.normal, ri);
            *out_o = si.pos;
            *out_d = dir;
            return true;
        }
        default:  // Emissive (material.h:119-122) and invalid handles (material.h:152-153)
            return false;
    }
}

// Optional sampling upgrades (PrtSampling, include/prt.h).  All off = the reference CPU backend.
//  * jitter: the primary ray goes through (x + u1, y + u2), u1 and u2 being the path's first two RNG draws (the
//    reference's OptiX backend, backend/optix/device_programs.cu:172-173); off: pixel centres (cpu/renderer.cpp:45).
//  * Russian roulette (reference roadmap only, wavefront.md:98-100; the rule is this project's): when a scatter would
//    start segment index >= rr_depth, the path survives with p = clamp(max component of the new throughput, 0.05, 1),
//    decided by ONE draw u taken after the material's own draws (u >= p ends the path with nothing more added);
//    survivors carry throughput / p.
//  * clamp (wavefront.md:102-104): every component of the radiance a path delivers is limited to `clamp`.
static inline bool rr_active(const PrtSampling* sp, int next_index) { return sp && sp->rr_depth && (uint32_t)next_index >= sp->rr_depth; }
static inline float rr_prob(V3 thr) {
    float p = thr.x > thr.y ? thr.x : thr.y;
    p = p > thr.z ? p : thr.z;
    p = p > 1.0f ? 1.0f : p;
    return p < 0.05f ? 0.05f : p;
}
static inline V3 clamp_radiance(const PrtSampling* sp, V3 L) {
    if (!sp || !(sp->clamp > 0.0f)) return L;
    return v3(L.x > sp->clamp ? sp->clamp : L.x, L.y > sp->clamp ? sp->clamp : L.y, L.z > sp->clamp ? sp->clamp : L.z);
}

V3 trace_recursive_impl(const OrcScene* s, V3 o, V3 d, int depth, int index, V3 thr, uint32_t* rng, int use_bvh,
                        uint32_t* segs, const PrtSampling* sp) {
    // CPURenderer::TraceRay (backend/cpu/renderer.cpp:59-103); index / thr only feed the optional roulette
    if (depth <= 0) return v3(0, 0, 0);
    V3 L = v3(0, 0, 0);
    Best b;
    closest_hit(s, o, d, use_bvh, &b);
    ++*segs;
    if (b.prim >= 0) {
        const PrtMaterial& m = s->materials[b.si.material];
        L = L + emit(m);
        V3 atten = v3(0, 0, 0), so = v3(0, 0, 0), sd = v3(0, 0, 1);
        bool sc = scatter(m, d, b.si, rng, &atten, &so, &sd);
        if (sc) {
            sd = normalize(sd);  // scatteredRay.Normalize() (:84)
            V3 t2 = thr * atten;
            float p = 1.0f;
            bool alive = true;
            if (depth > 1 && rr_active(sp, index + 1)) {
                p = rr_prob(t2);
                alive = rnd(rng) < p;
                t2 = v3(t2.x / p, t2.y / p, t2.z / p);
            }
            if (alive) {
                V3 Li = trace_recursive_impl(s, so, sd, depth - 1, index + 1, t2, rng, use_bvh, segs, sp);
                L = L + atten * v3(Li.x / p, Li.y / p, Li.z / p);
            }
        }
    } else {
        L = L + ld3(s->sky);
    }
    return L;
}

V3 trace_recursive(const OrcScene* s, V3 o, V3 d, int depth, uint32_t* rng, int use_bvh, uint32_t* segs,
                   const PrtSampling* sp = nullptr) {
    return clamp_radiance(sp, trace_recursive_impl(s, o, d, depth, 0, v3(1, 1, 1), rng, use_bvh, segs, sp));
}

V3 trace_iterative(const OrcScene* s, V3 o, V3 d, int max_depth, uint32_t* rng, int use_bvh, uint32_t* segs,
                   const PrtSampling* sp = nullptr) {
    // TraceRayGPU (backend/cuda_megakernel/renderer.cu:81-119)
    V3 L = v3(0, 0, 0);
    V3 thr = v3(1, 1, 1);
    for (int depth = 0; depth < max_depth; ++depth) {
        Best b;
        closest_hit(s, o, d, use_bvh, &b);
        ++*segs;
        if (b.prim < 0) {
            L = L + thr * ld3(s->sky);
            break;
        }
        const PrtMaterial& m = s->materials[b.si.material];
        L = L + thr * emit(m);
        V3 atten = v3(0, 0, 0), so = v3(0, 0, 0), sd = v3(0, 0, 1);
        if (!scatter(m, d, b.si, rng, &atten, &so, &sd)) break;
        thr = thr * atten;
        o = so;
        d = normalize(sd);
        if (depth + 1 < max_depth && rr_active(sp, depth + 1)) {
            const float p = rr_prob(thr);
            if (!(rnd(rng) < p)) break;
            thr = v3(thr.x / p, thr.y / p, thr.z / p);
        }
    }
    return clamp_radiance(sp, L);
}

struct Cam {
    V3 pos, front, right, up;
    float W, H;
};
Cam make_cam(const PrtCameraDesc* c) {  // Camera::Camera (camera.h:10-16)
    Cam k;
    k.pos = ld3(c->position);
    k.W = c->width;
    k.H = c->height;
    k.front = normalize(ld3(c->front));
    k.right = normalize(cross(k.front, v3(0.0f, 1.0f, 0.0f)));
    k.up = normalize(cross(k.right, k.front));
    return k;
}
void camera_ray(const Cam& c, float px, float py, V3* o, V3* d) {  // Camera::GetCameraRay (camera.h:103-132)
    float ndcX = (px / c.W) * 2.0f - 1.0f;
    float ndcY = 1.0f - (py / c.H) * 2.0f;
    float tanFovY = tanf(0.5f);
    float aspect = c.W / c.H;
    V3 dc = normalize(v3(ndcX * aspect * tanFovY, ndcY * tanFovY, -1.0f));
    V3 dw = dc.x * c.right + dc.y * c.up + dc.z * -c.front;
    dw = normalize(dw);
    *d = dw;
    *o = c.pos;
}

// ---- glm matrix helpers for Scene::MakeTransform (scene.cpp:9-17; geometry.h:92-99) ---------------
struct M4 {
    float c[4][4];  // c[col][row]
};
M4 m4_identity() {
    M4 m{};
    for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f;
    return m;
}
M4 m4_mul(const M4& a, const M4& b) {  // glm mat4*mat4: Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3]
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}
M4 m4_translate(V3 v) {  // glm::translate(I, v): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
    M4 m = m4_identity();
    M4 r = m;
    for (int i = 0; i < 4; ++i) r.c[3][i] = ((m.c[0][i] * v.x + m.c[1][i] * v.y) + m.c[2][i] * v.z) + m.c[3][i];
    return r;
}
M4 m4_scale(V3 v) {  // glm::scale(I, v)
    M4 m = m4_identity();
    M4 r{};
    for (int i = 0; i < 4; ++i) {
        r.c[0][i] = m.c[0][i] * v.x;
        r.c[1][i] = m.c[1][i] * v.y;
        r.c[2][i] = m.c[2][i] * v.z;
        r.c[3][i] = m.c[3][i];
    }
    return r;
}
M4 m4_euler_xyz(float t1, float t2, float t3) {  // glm::eulerAngleXYZ (gtx/euler_angles.inl)
    float c1 = cosf(-t1), c2 = cosf(-t2), c3 = cosf(-t3);
    float s1 = sinf(-t1), s2 = sinf(-t2), s3 = sinf(-t3);
    M4 r{};
    r.c[0][0] = c2 * c3;
    r.c[0][1] = -c1 * s3 + s1 * s2 * c3;
    r.c[0][2] = s1 * s3 + c1 * s2 * c3;
    r.c[0][3] = 0.0f;
    r.c[1][0] = c2 * s3;
    r.c[1][1] = c1 * c3 + s1 * s2 * s3;
    r.c[1][2] = -s1 * c3 + c1 * s2 * s3;
    r.c[1][3] = 0.0f;
    r.c[2][0] = -s2;
    r.c[2][1] = s1 * c2;
    r.c[2][2] = c1 * c2;
    r.c[2][3] = 0.0f;
    r.c[3][0] = 0.0f;
    r.c[3][1] = 0.0f;
    r.c[3][2] = 0.0f;
    r.c[3][3] = 1.0f;
    return r;
}
M4 m4_inverse(const M4& mm) {  // glm::inverse(mat4) (detail/func_matrix.inl compute_inverse<4,4>)
    const float(*m)[4] = mm.c;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    const float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    const float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    const float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    const float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]};
    const float Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    const float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]};
    const float Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float SignA[4] = {+1, -1, +1, -1};
    const float SignB[4] = {-1, +1, -1, +1};
    M4 Inv{};
    for (int i = 0; i < 4; ++i) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        Inv.c[0][i] = Inv0 * SignA[i];
        Inv.c[1][i] = Inv1 * SignB[i];
        Inv.c[2][i] = Inv2 * SignA[i];
        Inv.c[3][i] = Inv3 * SignB[i];
    }
    const float Row0[4] = {Inv.c[0][0], Inv.c[1][0], Inv.c[2][0], Inv.c[3][0]};
    float D0 = m[0][0] * Row0[0], D1 = m[0][1] * Row0[1], D2 = m[0][2] * Row0[2], D3 = m[0][3] * Row0[3];
    float Dot1 = (D0 + D1) + (D2 + D3);
    float OneOverDet = 1.0f / Dot1;
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) r.c[j][i] = Inv.c[j][i] * OneOverDet;
    return r;
}
void make_transform(V3 scale, V3 euler_deg, V3 tr, float* mat, float* inv) {
    // glm::radians: degrees * 0.01745329251994329576923690768489
    const float k = 0.01745329251994329576923690768489f;
    V3 e = v3(euler_deg.x * k, euler_deg.y * k, euler_deg.z * k);
    M4 m = m4_mul(m4_mul(m4_translate(tr), m4_euler_xyz(e.x, e.y, e.z)), m4_scale(scale));
    M4 i = m4_inverse(m);
    memcpy(mat, m.c, 64);
    memcpy(inv, i.c, 64);
}

// ---- presets (scene.cpp:62-350) -------------------------------------------------------------------
struct Builder {
    std::vector<PrtMaterial> mats;
    std::vector<PrtPrimitive> prims;
    uint32_t mat(uint32_t type, V3 rgb, float scalar) {
        PrtMaterial m{};
        m.type = type;
        st3(m.rgb, rgb);
        m.scalar = scalar;
        mats.push_back(m);
        return (uint32_t)mats.size() - 1;
    }
    uint32_t lambert(V3 a) { return mat(PRT_MAT_LAMBERTIAN, a, 0.0f); }
    uint32_t metal(V3 a, float r) { return mat(PRT_MAT_METAL, a, r); }
    uint32_t dielectric(float ri) { return mat(PRT_MAT_DIELECTRIC, v3(0, 0, 0), ri); }
    uint32_t emissive(V3 e) { return mat(PRT_MAT_EMISSIVE, e, 0.0f); }
    void prim(uint32_t shape, float p0, float p1, uint32_t material, V3 scale, V3 euler, V3 tr) {
        PrtPrimitive p{};
        p.shape_type = shape;
        p.shape_param[0] = p0;
        p.shape_param[1] = p1;
        p.material_id = material;
        make_transform(scale, euler, tr, p.mat, p.inv);
        prims.push_back(p);
    }
    void circle(float r, uint32_t m, V3 s, V3 e, V3 t) { prim(PRT_SHAPE_CIRCLE, r, 0.0f, m, s, e, t); }
    void quad(float w, float h, uint32_t m, V3 s, V3 e, V3 t) { prim(PRT_SHAPE_QUAD, w, h, m, s, e, t); }
};

void preset_random_balls(Builder& b, int ballCount) {  // scene.cpp:62-170
    const V3 one = v3(1, 1, 1), zero = v3(0, 0, 0);
    uint32_t ground = b.lambert(v3(0.5f, 0.5f, 0.5f));
    b.quad(200.0f, 200.0f, ground, one, zero, zero);
    std::mt19937 rng(1337);
    std::uniform_real_distribution<float> dist01(0.0f, 1.0f);
    std::uniform_real_distribution<float> distPos(-40.0f, 40.0f);
    std::uniform_real_distribution<float> distRadius(0.2f, 1.0f);
    // NOTE: the reference draws several values inside one constructor-argument list
    // (scene.cpp:96-100,109-113), whose evaluation order C++ leaves unspecified; the contract here is
    // left-to-right as written.
    for (int i = 0; i < ballCount; ++i) {
        float radius = distRadius(rng);
        float px = distPos(rng);
        float pz = distPos(rng);
        V3 pos = v3(px, radius, pz);
        float m = dist01(rng);
        uint32_t mat;
        if (m < 0.65f) {
            float r = dist01(rng), g = dist01(rng), bl = dist01(rng);
            mat = b.lambert(v3(r, g, bl));
        } else if (m < 0.9f) {
            float a = 0.7f + 0.3f * dist01(rng);
            float rough = 0.05f * dist01(rng);
            mat = b.metal(v3(a, a, a), rough);
        } else {
            mat = b.dielectric(1.3f + 0.4f * dist01(rng));
        }
        b.circle(radius, mat, one, zero, pos);
    }
    for (int i = 0; i < 8; ++i) {
        float radius = 1.5f;
        float px = distPos(rng);
        float pz = distPos(rng);
        float e = 10.0f + 10.0f * dist01(rng);
        uint32_t lm = b.emissive(v3(e, e, e));
        b.circle(radius, lm, one, zero, v3(px, 8.0f, pz));
    }
}

void preset_default(Builder& b) {  // scene.cpp:188-278
    const V3 one = v3(1, 1, 1), zero = v3(0, 0, 0);
    b.circle(1.0f, b.emissive(v3(10, 5, 5)), v3(2, 2, 2), zero, v3(5, 6, 0));
    b.quad(8, 8, b.emissive(v3(3, 4, 2)), one, v3(50, 0, 0), v3(-4, 7, 7));
    b.quad(8, 8, b.emissive(v3(3, 2, 1)), one, v3(50, 0, 0), v3(4, 7, 7));
    b.circle(1.0f, b.lambert(v3(0.2f, 1.0f, 0.2f)), one, zero, v3(4, 1, 0));
    b.circle(1.0f, b.lambert(v3(1.0f, 0.2f, 0.2f)), one, zero, v3(-4, 1, 0));
    b.circle(1.0f, b.dielectric(0.9f), one, zero, v3(0, 1, 4));
    b.circle(1.0f, b.metal(v3(1, 0.7f, 0.8f), 0.01f), one, zero, v3(0, 1, -4));
    b.quad(20, 20, b.lambert(v3(0.7f, 0.7f, 0.4f)), one, zero, zero);
}

void preset_light_test(Builder& b) {  // scene.cpp:280-305
    const V3 one = v3(1, 1, 1), zero = v3(0, 0, 0);
    b.quad(30, 30, b.lambert(v3(0.6f, 0.6f, 0.6f)), one, zero, zero);
    for (int i = -5; i <= 5; ++i) {
        uint32_t lm = b.emissive(v3(4, 4, 4));
        b.circle(0.5f, lm, one, zero, v3(float(i * 2), 6, 0));
    }
}

void preset_material_test(Builder& b) {  // scene.cpp:307-330
    const V3 one = v3(1, 1, 1), zero = v3(0, 0, 0);
    b.quad(25, 25, b.lambert(v3(0.8f, 0.8f, 0.8f)), one, zero, zero);
    b.circle(1.0f, b.lambert(v3(1, 0, 0)), one, zero, v3(-4, 1, 0));
    b.circle(1.0f, b.metal(v3(0.9f, 0.9f, 0.9f), 0.0f), one, zero, v3(0, 1, 0));
    b.circle(1.0f, b.dielectric(1.5f), one, zero, v3(4, 1, 0));
}

void preset_cornell(Builder& b) {  // scene.cpp:332-350
    const V3 one = v3(1, 1, 1), zero = v3(0, 0, 0);
    uint32_t red = b.lambert(v3(0.75f, 0.1f, 0.1f));
    uint32_t green = b.lambert(v3(0.1f, 0.75f, 0.1f));
    uint32_t white = b.lambert(v3(0.8f, 0.8f, 0.8f));
    b.quad(10, 10, white, one, zero, zero);
    b.quad(10, 10, red, one, v3(90, 0, 0), v3(-5, 5, 0));
    b.quad(10, 10, green, one, v3(90, 0, 0), v3(5, 5, 0));
    uint32_t light = b.emissive(v3(15, 15, 15));
    b.quad(10, 10, light, one, v3(90, 0, 0), v3(0, 9, 0));
}

static void fill_hit(const Best& b, PrtHit* h) {
    memset(h, 0, sizeof(*h));
    h->prim = b.prim;
    if (b.prim < 0) {
        h->material_id = 0xFFFFFFFFu;
        h->d2 = FLT_MAX;
        return;
    }
    h->front_face = b.si.front ? 1u : 0u;
    h->material_id = b.si.material;
    h->d2 = b.d2;
    st3(h->position, b.si.pos);
    st3(h->normal, b.si.normal);
}

template <class F>
static void parallel_for(uint32_t n, int n_threads, F&& f) {
    if (n_threads <= 1 || n < 2) {
        for (uint32_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    const uint32_t chunk = std::max(1u, n / (uint32_t)(n_threads * 16));
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&]() {
            while (true) {
                uint32_t b = next.fetch_add(chunk);
                if (b >= n) break;
                uint32_t e = std::min(n, b + chunk);
                for (uint32_t i = b; i < e; ++i) f(i);
            }
        });
    for (auto& t : th) t.join();
}

}  // namespace

// ===============================================================================================
// C interface
// ===============================================================================================
extern "C" {

uint32_t orc_pcg_hash(uint32_t v) { return pcg_hash(v); }
uint32_t orc_path_seed(uint32_t pixel, uint32_t sample, uint32_t seed) {
    // seed == 0 reproduces the reference's OptiX seeding pcg_hash(pixelIndex ^ (frameIndex * 719393u))
    // (optix/device_programs.cu:169); a user seed decorrelates whole runs.
    return pcg_hash((pixel ^ (sample * 719393u)) + seed * 0x9E3779B9u);
}
float orc_random(uint32_t* state) { return rnd(state); }
void orc_random_unit_vector(uint32_t* state, float out[3]) { st3(out, random_unit_vector(state)); }

void orc_camera_basis(const PrtCameraDesc* cam, float front[3], float right[3], float up[3]) {
    Cam c = make_cam(cam);
    st3(front, c.front);
    st3(right, c.right);
    st3(up, c.up);
}
void orc_camera_rays(const PrtCameraDesc* cam, uint32_t n, const float* px, const float* py, float* origins,
                     float* dirs) {
    Cam c = make_cam(cam);
    for (uint32_t i = 0; i < n; ++i) {
        V3 o, d;
        camera_ray(c, px[i], py[i], &o, &d);
        st3(origins + 3 * i, o);
        st3(dirs + 3 * i, d);
    }
}

int orc_shape_intersect(int shape_type, const float* p, const float o[3], const float d[3], float pos[3],
                        float normal[3], int* front) {
    SI si;
    if (shape_type == PRT_SHAPE_CIRCLE)
        circle_intersect(p[0], ld3(o), ld3(d), &si);
    else if (shape_type == PRT_SHAPE_QUAD)
        quad_intersect(p[0], p[1], ld3(o), ld3(d), &si);
    else
        triangle_intersect(ld3(p), ld3(p + 3), ld3(p + 6), ld3(p + 9), ld3(p + 12), ld3(p + 15), ld3(o), ld3(d), &si);
    st3(pos, si.pos);
    st3(normal, si.normal);
    *front = si.front ? 1 : 0;
    return si.has ? 1 : 0;
}

void orc_transform_point(const float m[16], const float p[3], float out[3]) { st3(out, transform_point(m, ld3(p))); }
void orc_transform_normal(const float m[16], const float n[3], float out[3]) { st3(out, transform_normal(m, ld3(n))); }
void orc_make_transform(const float scale[3], const float euler_deg[3], const float translation[3], float mat[16],
                        float inv[16]) {
    make_transform(ld3(scale), ld3(euler_deg), ld3(translation), mat, inv);
}

int orc_scene_preset(int preset, PrtMaterial* materials, uint32_t* n_materials, PrtPrimitive* primitives,
                     uint32_t* n_primitives) {
    Builder b;
    switch (preset) {
        case PRT_PRESET_DEFAULT: preset_default(b); break;
        case PRT_PRESET_LIGHT_TEST: preset_light_test(b); break;
        case PRT_PRESET_MATERIAL_TEST: preset_material_test(b); break;
        case PRT_PRESET_CORNELL: preset_cornell(b); break;
        case PRT_PRESET_RANDOM_BALLS_SMALL: preset_random_balls(b, 100); break;
        case PRT_PRESET_RANDOM_BALLS_MEDIUM: preset_random_balls(b, 400); break;
        case PRT_PRESET_RANDOM_BALLS_LARGE: preset_random_balls(b, 800); break;
        default: return 1;
    }
    if (materials) memcpy(materials, b.mats.data(), b.mats.size() * sizeof(PrtMaterial));
    if (primitives) memcpy(primitives, b.prims.data(), b.prims.size() * sizeof(PrtPrimitive));
    if (n_materials) *n_materials = (uint32_t)b.mats.size();
    if (n_primitives) *n_primitives = (uint32_t)b.prims.size();
    return 0;
}

OrcScene* orc_scene_create(const PrtSceneDesc* desc) {
    OrcScene* s = new OrcScene();
    s->materials.assign(desc->materials, desc->materials + desc->n_materials);
    s->prims.assign(desc->primitives, desc->primitives + desc->n_primitives);
    memcpy(s->sky, desc->sky, sizeof(s->sky));
    for (uint32_t mi = 0; mi < desc->n_meshes; ++mi) {
        const PrtMesh& m = desc->meshes[mi];
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            Tri tri;
            for (int k = 0; k < 3; ++k) {
                uint32_t vi = m.indices[3 * t + k];
                tri.p[k] = ld3(m.positions + 3 * (size_t)vi);
                tri.n[k] = ld3(m.normals + 3 * (size_t)vi);
            }
            tri.material = m.material_id;
            s->tris.push_back(tri);
        }
    }
    build_bvh(s, s->prims);
    uint32_t prim_base = (uint32_t)(s->prims.size() + s->tris.size());
    s->imeshes.resize(desc->n_instanced_meshes);
    for (uint32_t mi = 0; mi < desc->n_instanced_meshes; ++mi) {
        const PrtMesh& m = desc->instanced_meshes[mi];
        OrcMesh& om = s->imeshes[mi];
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            Tri tri;
            for (int k = 0; k < 3; ++k) {
                uint32_t vi = m.indices[3 * t + k];
                tri.p[k] = ld3(m.positions + 3 * (size_t)vi);
                tri.n[k] = ld3(m.normals + 3 * (size_t)vi);
            }
            tri.material = 0;
            om.tris.push_back(tri);
        }
        build_bvh(&om, std::vector<PrtPrimitive>());
    }
    for (uint32_t i = 0; i < desc->n_instances; ++i) {
        const PrtInstance& pi = desc->instances[i];
        OrcInstance in;
        in.mesh = pi.mesh;
        in.material = pi.material_id;
        memcpy(in.mat, pi.mat, sizeof(in.mat));
        memcpy(in.inv, pi.inv, sizeof(in.inv));
        in.prim_base = prim_base;
        in.scale = std::sqrt(pi.mat[0] * pi.mat[0] + pi.mat[1] * pi.mat[1] + pi.mat[2] * pi.mat[2]);
        prim_base += (uint32_t)s->imeshes[pi.mesh].tris.size();
        s->instances.push_back(in);
        // the world-space culling pad must cover the placed copies as well
        for (int k = 12; k < 15; ++k) s->pad = std::max(s->pad, (fabsf(pi.mat[k]) + 1.0f) * 1e-4f);
    }
    return s;
}
void orc_scene_destroy(OrcScene* s) { delete s; }
uint32_t orc_scene_prim_count(const OrcScene* s) {
    uint32_t n = (uint32_t)(s->prims.size() + s->tris.size());
    for (const OrcInstance& in : s->instances) n += (uint32_t)s->imeshes[in.mesh].tris.size();
    return n;
}

void orc_closest_hit(const OrcScene* s, uint32_t n, const float* origins, const float* dirs, PrtHit* hits,
                     int use_bvh, int n_threads) {
    parallel_for(n, n_threads, [&](uint32_t i) {
        Best b;
        closest_hit(s, ld3(origins + 3 * (size_t)i), ld3(dirs + 3 * (size_t)i), use_bvh, &b);
        fill_hit(b, &hits[i]);
    });
}

int orc_scatter(const PrtMaterial* m, const float in_dir[3], const PrtHit* hit, uint32_t* rng_state,
                float attenuation[3], float emitted[3], float out_origin[3], float out_dir[3]) {
    SI si;
    si.has = true;
    si.front = hit->front_face != 0;
    si.pos = ld3(hit->position);
    si.normal = ld3(hit->normal);
    V3 atten = v3(0, 0, 0), so = v3(0, 0, 0), sd = v3(0, 0, 1);
    st3(emitted, emit(*m));
    bool sc = scatter(*m, ld3(in_dir), si, rng_state, &atten, &so, &sd);
    st3(attenuation, atten);
    st3(out_origin, so);
    st3(out_dir, sd);
    return sc ? 1 : 0;
}

void orc_scatter_batch(const PrtMaterial* materials, uint32_t n, const float* in_dirs, const PrtHit* hits,
                       uint32_t* rng_state, uint32_t* scattered, float* attenuation, float* emitted, float* out_origins,
                       float* out_dirs) {
    for (uint32_t i = 0; i < n; ++i)
        scattered[i] = (uint32_t)orc_scatter(&materials[hits[i].material_id], in_dirs + 3 * (size_t)i, &hits[i], &rng_state[i],
                                             attenuation + 3 * (size_t)i, emitted + 3 * (size_t)i, out_origins + 3 * (size_t)i,
                                             out_dirs + 3 * (size_t)i);
}

float orc_fresnel(float cosine, float ri) { return fresnel_reflectance(cosine, ri); }
float orc_fresnel_libm(float cosine, float ri) { return fresnel_reflectance_libm(cosine, ri); }
void orc_fresnel_batch(uint32_t n, const float* cosine, const float* ri, float* out, float* out_libm) {
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = fresnel_reflectance(cosine[i], ri[i]);
        out_libm[i] = fresnel_reflectance_libm(cosine[i], ri[i]);
    }
}

void orc_trace(const OrcScene* s, const float o[3], const float d[3], int max_depth, uint32_t* rng_state,
               int iterative, int use_bvh, float L[3], uint32_t* n_segments) {
    uint32_t segs = 0;
    V3 r = iterative ? trace_iterative(s, ld3(o), ld3(d), max_depth, rng_state, use_bvh, &segs)
                     : trace_recursive(s, ld3(o), ld3(d), max_depth, rng_state, use_bvh, &segs);
    st3(L, r);
    if (n_segments) *n_segments = segs;
}

void orc_render_sampling(const OrcScene* s, const PrtCameraDesc* cam, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0,
                         uint32_t x1, uint32_t y1, uint32_t spp, uint32_t first_sample, int max_depth, uint32_t seed,
                         int iterative, int use_bvh, int n_threads, const PrtSampling* sp, float* accum, float* weights,
                         uint64_t* rays);
void orc_render(const OrcScene* s, const PrtCameraDesc* cam, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0,
                uint32_t x1, uint32_t y1, uint32_t spp, uint32_t first_sample, int max_depth, uint32_t seed,
                int iterative, int use_bvh, int n_threads, float* accum, float* weights, uint64_t* rays) {
    orc_render_sampling(s, cam, W, H, x0, y0, x1, y1, spp, first_sample, max_depth, seed, iterative, use_bvh, n_threads,
                        nullptr, accum, weights, rays);
}

void orc_render_sampling(const OrcScene* s, const PrtCameraDesc* cam, uint32_t W, uint32_t H, uint32_t x0, uint32_t y0,
                         uint32_t x1, uint32_t y1, uint32_t spp, uint32_t first_sample, int max_depth, uint32_t seed,
                         int iterative, int use_bvh, int n_threads, const PrtSampling* sp, float* accum, float* weights,
                         uint64_t* rays) {
    Cam c = make_cam(cam);
    x1 = std::min(x1, W);
    y1 = std::min(y1, H);
    if (x1 <= x0 || y1 <= y0) {
        if (rays) *rays = 0;
        return;
    }
    std::atomic<uint64_t> total{0};
    const uint32_t rows = y1 - y0;
    parallel_for(rows, n_threads, [&](uint32_t r) {
        uint32_t j = y0 + r;
        uint64_t local = 0;
        for (uint32_t i = x0; i < x1; ++i) {
            const uint32_t idx = j * W + i;
            for (uint32_t sidx = first_sample; sidx < first_sample + spp; ++sidx) {
                V3 o, d;
                uint32_t rng = orc_path_seed(idx, sidx, seed);
                float fx = 0.5f, fy = 0.5f;  // pixel centre, cpu/renderer.cpp:45
                if (sp && sp->jitter) {      // backend/optix/device_programs.cu:172-173
                    fx = rnd(&rng);
                    fy = rnd(&rng);
                }
                camera_ray(c, (float)i + fx, (float)j + fy, &o, &d);
                uint32_t segs = 0;
                V3 L = iterative ? trace_iterative(s, o, d, max_depth, &rng, use_bvh, &segs, sp)
                                 : trace_recursive(s, o, d, max_depth, &rng, use_bvh, &segs, sp);
                local += segs;
                // Film::AddSample (film.cu:37-55), weight = 1
                const float weight = 1.0f;
                accum[3 * (size_t)idx + 0] += L.x * weight;
                accum[3 * (size_t)idx + 1] += L.y * weight;
                accum[3 * (size_t)idx + 2] += L.z * weight;
                weights[idx] += weight;
            }
        }
        total += local;
    });
    if (rays) *rays = total.load();
}

void orc_tonemap(const float* accum, const float* weights, uint32_t n_pixels, float exposure, float gamma,
                 uint8_t* rgba8) {
    // Film::UpdateDisplay (film.cu:134-194), Tonemap / ToByte (film.h:63-75)
    const float invGamma = 1.0f / gamma;
    auto tone = [exposure](float v) {
        float x = v * exposure;
        x = x / (1.0f + x);
        return x;
    };
    auto to_byte = [](float v) {
        v = std::fmax(0.0f, std::fmin(1.0f, v));
        return static_cast<uint8_t>(v * 255.0f + 0.5f);
    };
    for (uint32_t i = 0; i < n_pixels; ++i) {
        const float w = weights[i];
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (w > 0.0f) {
            const float invW = 1.0f / w;
            r = accum[3 * (size_t)i + 0] * invW;
            g = accum[3 * (size_t)i + 1] * invW;
            b = accum[3 * (size_t)i + 2] * invW;
            r = tone(r);
            g = tone(g);
            b = tone(b);
            r = std::pow(r, invGamma);
            g = std::pow(g, invGamma);
            b = std::pow(b, invGamma);
        }
        rgba8[4 * (size_t)i + 0] = to_byte(r);
        rgba8[4 * (size_t)i + 1] = to_byte(g);
        rgba8[4 * (size_t)i + 2] = to_byte(b);
        rgba8[4 * (size_t)i + 3] = 255;
    }
}

int orc_aabb_intersect_p(const float bmin[3], const float bmax[3], const float o[3], const float d[3]) {
    // AABB::IntersectP (geometry.h:170-192)
    float tNear = 0.0f;
    float tFar = FLT_MAX;
    for (int i = 0; i < 3; i++) {
        float invD = 1.0f / d[i];
        float t0 = (bmin[i] - o[i]) * invD;
        float t1 = (bmax[i] - o[i]) * invD;
        if (invD < 0.0f) std::swap(t0, t1);
        tNear = std::max(tNear, t0);
        tFar = std::min(tFar, t1);
        if (tFar < tNear) return 0;
    }
    return 1;
}

}  // extern "C"
