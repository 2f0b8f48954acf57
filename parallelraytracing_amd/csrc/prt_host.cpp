// prt_host.cpp — host-side data formats either side of the hot path (no HIP here):
//   * PLY ingest with the subset the reference's Mesh requests from tinyply (src/core/mesh.cpp:79-97,113-144)
//   * mesh utilities for the synthetic benchmark inputs (refinement, baking transforms, merging)
//   * the scene presets (src/core/scene.cpp:62-350) flattened to PrtMaterial / PrtPrimitive
//   * Scene::MakeTransform (src/core/scene.cpp:9-17) in glm's operation order
//   * PPM / PFM framebuffer dumps (stand-in for the GLFW/OpenGL viewer, src/main.cpp:504-527)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <memory>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/prt.h"
#include "prt_mesh.h"

// =================================================================================================
// glm-order matrix helpers
// =================================================================================================
namespace {

struct Mat4 {
    float m[16];  // column-major: m[4*c + r]
    float& at(int c, int r) { return m[4 * c + r]; }
    float at(int c, int r) const { return m[4 * c + r]; }
};

Mat4 identity() {
    Mat4 r{};
    r.at(0, 0) = r.at(1, 1) = r.at(2, 2) = r.at(3, 3) = 1.0f;
    return r;
}

// glm::operator*(mat4, mat4): column j of the product = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3]
Mat4 mul(const Mat4& A, const Mat4& B) {
    Mat4 R{};
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float acc = A.at(0, r) * B.at(j, 0) + A.at(1, r) * B.at(j, 1);
            acc = acc + A.at(2, r) * B.at(j, 2);
            acc = acc + A.at(3, r) * B.at(j, 3);
            R.at(j, r) = acc;
        }
    return R;
}

// glm::translate(mat4(1), v)
Mat4 translation(const float v[3]) {
    const Mat4 I = identity();
    Mat4 R = I;
    for (int r = 0; r < 4; ++r) {
        float acc = I.at(0, r) * v[0] + I.at(1, r) * v[1];
        acc = acc + I.at(2, r) * v[2];
        R.at(3, r) = acc + I.at(3, r);
    }
    return R;
}

// glm::scale(mat4(1), v)
Mat4 scaling(const float v[3]) {
    const Mat4 I = identity();
    Mat4 R{};
    for (int r = 0; r < 4; ++r) {
        R.at(0, r) = I.at(0, r) * v[0];
        R.at(1, r) = I.at(1, r) * v[1];
        R.at(2, r) = I.at(2, r) * v[2];
        R.at(3, r) = I.at(3, r);
    }
    return R;
}

// glm::eulerAngleXYZ(t1, t2, t3) (glm/gtx/euler_angles.inl)
Mat4 euler_xyz(float t1, float t2, float t3) {
    const float c1 = cosf(-t1), c2 = cosf(-t2), c3 = cosf(-t3);
    const float s1 = sinf(-t1), s2 = sinf(-t2), s3 = sinf(-t3);
    Mat4 R{};
    R.at(0, 0) = c2 * c3;
    R.at(0, 1) = -c1 * s3 + s1 * s2 * c3;
    R.at(0, 2) = s1 * s3 + c1 * s2 * c3;
    R.at(1, 0) = c2 * s3;
    R.at(1, 1) = c1 * c3 + s1 * s2 * s3;
    R.at(1, 2) = -s1 * c3 + c1 * s2 * s3;
    R.at(2, 0) = -s2;
    R.at(2, 1) = s1 * c2;
    R.at(2, 2) = c1 * c2;
    R.at(3, 3) = 1.0f;
    return R;
}

// glm::inverse(mat4) (glm/detail/func_matrix.inl): cofactors, then multiply by 1/determinant
Mat4 inverse(const Mat4& M) {
    auto m = [&M](int c, int r) { return M.at(c, r); };
    const float c00 = m(2, 2) * m(3, 3) - m(3, 2) * m(2, 3);
    const float c02 = m(1, 2) * m(3, 3) - m(3, 2) * m(1, 3);
    const float c03 = m(1, 2) * m(2, 3) - m(2, 2) * m(1, 3);
    const float c04 = m(2, 1) * m(3, 3) - m(3, 1) * m(2, 3);
    const float c06 = m(1, 1) * m(3, 3) - m(3, 1) * m(1, 3);
    const float c07 = m(1, 1) * m(2, 3) - m(2, 1) * m(1, 3);
    const float c08 = m(2, 1) * m(3, 2) - m(3, 1) * m(2, 2);
    const float c10 = m(1, 1) * m(3, 2) - m(3, 1) * m(1, 2);
    const float c11 = m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2);
    const float c12 = m(2, 0) * m(3, 3) - m(3, 0) * m(2, 3);
    const float c14 = m(1, 0) * m(3, 3) - m(3, 0) * m(1, 3);
    const float c15 = m(1, 0) * m(2, 3) - m(2, 0) * m(1, 3);
    const float c16 = m(2, 0) * m(3, 2) - m(3, 0) * m(2, 2);
    const float c18 = m(1, 0) * m(3, 2) - m(3, 0) * m(1, 2);
    const float c19 = m(1, 0) * m(2, 2) - m(2, 0) * m(1, 2);
    const float c20 = m(2, 0) * m(3, 1) - m(3, 0) * m(2, 1);
    const float c22 = m(1, 0) * m(3, 1) - m(3, 0) * m(1, 1);
    const float c23 = m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1);
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {m(1, 0), m(0, 0), m(0, 0), m(0, 0)}, V1[4] = {m(1, 1), m(0, 1), m(0, 1), m(0, 1)};
    const float V2[4] = {m(1, 2), m(0, 2), m(0, 2), m(0, 2)}, V3[4] = {m(1, 3), m(0, 3), m(0, 3), m(0, 3)};
    Mat4 adj{};
    for (int r = 0; r < 4; ++r) {
        const float sa = (r & 1) ? -1.0f : 1.0f, sb = -sa;
        adj.at(0, r) = ((V1[r] * F0[r] - V2[r] * F1[r]) + V3[r] * F2[r]) * sa;
        adj.at(1, r) = ((V0[r] * F0[r] - V2[r] * F3[r]) + V3[r] * F4[r]) * sb;
        adj.at(2, r) = ((V0[r] * F1[r] - V1[r] * F3[r]) + V3[r] * F5[r]) * sa;
        adj.at(3, r) = ((V0[r] * F2[r] - V1[r] * F4[r]) + V2[r] * F5[r]) * sb;
    }
    const float d0 = m(0, 0) * adj.at(0, 0), d1 = m(0, 1) * adj.at(1, 0), d2 = m(0, 2) * adj.at(2, 0),
                d3 = m(0, 3) * adj.at(3, 0);
    const float det = (d0 + d1) + (d2 + d3);
    const float inv_det = 1.0f / det;
    Mat4 R{};
    for (int i = 0; i < 16; ++i) R.m[i] = adj.m[i] * inv_det;
    return R;
}

}  // namespace

extern "C" void prt_make_transform(const float scale[3], const float euler_deg[3], const float translation_v[3],
                                   float mat[16], float inv[16]) {
    // glm::radians(v) = v * 0.0174532925...
    const float k = 0.01745329251994329576923690768489f;
    const Mat4 TR = mul(translation(translation_v), euler_xyz(euler_deg[0] * k, euler_deg[1] * k, euler_deg[2] * k));
    const Mat4 M = mul(TR, scaling(scale));
    const Mat4 I = inverse(M);
    memcpy(mat, M.m, sizeof(M.m));
    memcpy(inv, I.m, sizeof(I.m));
}

// =================================================================================================
// Scene presets (src/core/scene.cpp).  Table-driven; the random presets replay std::mt19937(1337)
// with libstdc++'s uniform_real_distribution<float>, draws taken left-to-right as written
// (scene.cpp:96-100,109-113 leave the order of draws inside one argument list unspecified).
// =================================================================================================
namespace {

struct PresetOut {
    std::vector<PrtMaterial> mats;
    std::vector<PrtPrimitive> prims;
    uint32_t add_mat(uint32_t type, float r, float g, float b, float s) {
        PrtMaterial m;
        m.type = type;
        m.rgb[0] = r;
        m.rgb[1] = g;
        m.rgb[2] = b;
        m.scalar = s;
        mats.push_back(m);
        return (uint32_t)(mats.size() - 1);
    }
    void add_prim(uint32_t shape, float p0, float p1, uint32_t mat, float sc, float ex, float tx, float ty, float tz) {
        PrtPrimitive p;
        memset(&p, 0, sizeof(p));
        p.shape_type = shape;
        p.shape_param[0] = p0;
        p.shape_param[1] = p1;
        p.material_id = mat;
        const float s[3] = {sc, sc, sc}, e[3] = {ex, 0.0f, 0.0f}, t[3] = {tx, ty, tz};
        prt_make_transform(s, e, t, p.mat, p.inv);
        prims.push_back(p);
    }
};

struct Row {  // one AddPrimitive call of a fixed preset
    uint32_t shape;
    float p0, p1;
    uint32_t mtype;
    float r, g, b, s;
    float scale, euler_x, tx, ty, tz;
};

const Row kDefault[] = {
    // scene.cpp:188-278
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_EMISSIVE, 10, 5, 5, 0, 2, 0, 5, 6, 0},
    {PRT_SHAPE_QUAD, 8, 8, PRT_MAT_EMISSIVE, 3, 4, 2, 0, 1, 50, -4, 7, 7},
    {PRT_SHAPE_QUAD, 8, 8, PRT_MAT_EMISSIVE, 3, 2, 1, 0, 1, 50, 4, 7, 7},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_LAMBERTIAN, 0.2f, 1.0f, 0.2f, 0, 1, 0, 4, 1, 0},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_LAMBERTIAN, 1.0f, 0.2f, 0.2f, 0, 1, 0, -4, 1, 0},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_DIELECTRIC, 0, 0, 0, 0.9f, 1, 0, 0, 1, 4},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_METAL, 1, 0.7f, 0.8f, 0.01f, 1, 0, 0, 1, -4},
    {PRT_SHAPE_QUAD, 20, 20, PRT_MAT_LAMBERTIAN, 0.7f, 0.7f, 0.4f, 0, 1, 0, 0, 0, 0},
};
const Row kMaterialTest[] = {
    // scene.cpp:307-330
    {PRT_SHAPE_QUAD, 25, 25, PRT_MAT_LAMBERTIAN, 0.8f, 0.8f, 0.8f, 0, 1, 0, 0, 0, 0},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_LAMBERTIAN, 1, 0, 0, 0, 1, 0, -4, 1, 0},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_METAL, 0.9f, 0.9f, 0.9f, 0.0f, 1, 0, 0, 1, 0},
    {PRT_SHAPE_CIRCLE, 1, 0, PRT_MAT_DIELECTRIC, 0, 0, 0, 1.5f, 1, 0, 4, 1, 0},
};

void emit_rows(PresetOut& o, const Row* rows, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const Row& r = rows[i];
        const uint32_t m = o.add_mat(r.mtype, r.r, r.g, r.b, r.s);
        o.add_prim(r.shape, r.p0, r.p1, m, r.scale, r.euler_x, r.tx, r.ty, r.tz);
    }
}

void preset_cornell(PresetOut& o) {  // scene.cpp:332-350 (materials first, one shared quad)
    const uint32_t red = o.add_mat(PRT_MAT_LAMBERTIAN, 0.75f, 0.1f, 0.1f, 0);
    const uint32_t green = o.add_mat(PRT_MAT_LAMBERTIAN, 0.1f, 0.75f, 0.1f, 0);
    const uint32_t white = o.add_mat(PRT_MAT_LAMBERTIAN, 0.8f, 0.8f, 0.8f, 0);
    o.add_prim(PRT_SHAPE_QUAD, 10, 10, white, 1, 0, 0, 0, 0);
    o.add_prim(PRT_SHAPE_QUAD, 10, 10, red, 1, 90, -5, 5, 0);
    o.add_prim(PRT_SHAPE_QUAD, 10, 10, green, 1, 90, 5, 5, 0);
    const uint32_t light = o.add_mat(PRT_MAT_EMISSIVE, 15, 15, 15, 0);
    o.add_prim(PRT_SHAPE_QUAD, 10, 10, light, 1, 90, 0, 9, 0);
}

void preset_light_test(PresetOut& o) {  // scene.cpp:280-305
    const uint32_t ground = o.add_mat(PRT_MAT_LAMBERTIAN, 0.6f, 0.6f, 0.6f, 0);
    o.add_prim(PRT_SHAPE_QUAD, 30, 30, ground, 1, 0, 0, 0, 0);
    for (int i = -5; i <= 5; ++i) {
        const uint32_t lm = o.add_mat(PRT_MAT_EMISSIVE, 4, 4, 4, 0);
        o.add_prim(PRT_SHAPE_CIRCLE, 0.5f, 0, lm, 1, 0, float(i * 2), 6, 0);
    }
}

void preset_random_balls(PresetOut& o, int balls) {  // scene.cpp:62-170
    const uint32_t ground = o.add_mat(PRT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0);
    o.add_prim(PRT_SHAPE_QUAD, 200.0f, 200.0f, ground, 1, 0, 0, 0, 0);
    std::mt19937 rng(1337);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f), upos(-40.0f, 40.0f), urad(0.2f, 1.0f);
    for (int i = 0; i < balls; ++i) {
        const float radius = urad(rng);
        const float x = upos(rng);
        const float z = upos(rng);
        const float pick = u01(rng);
        uint32_t m;
        if (pick < 0.65f) {
            const float r = u01(rng);
            const float g = u01(rng);
            const float b = u01(rng);
            m = o.add_mat(PRT_MAT_LAMBERTIAN, r, g, b, 0);
        } else if (pick < 0.9f) {
            const float a = 0.7f + 0.3f * u01(rng);
            const float rough = 0.05f * u01(rng);
            m = o.add_mat(PRT_MAT_METAL, a, a, a, rough);
        } else {
            m = o.add_mat(PRT_MAT_DIELECTRIC, 0, 0, 0, 1.3f + 0.4f * u01(rng));
        }
        o.add_prim(PRT_SHAPE_CIRCLE, radius, 0, m, 1, 0, x, radius, z);
    }
    for (int i = 0; i < 8; ++i) {
        const float x = upos(rng);
        const float z = upos(rng);
        const float e = 10.0f + 10.0f * u01(rng);
        const uint32_t lm = o.add_mat(PRT_MAT_EMISSIVE, e, e, e, 0);
        o.add_prim(PRT_SHAPE_CIRCLE, 1.5f, 0, lm, 1, 0, x, 8.0f, z);
    }
}

}  // namespace

extern "C" int prt_scene_preset(int preset, PrtMaterial* materials, uint32_t* n_materials, PrtPrimitive* primitives,
                                uint32_t* n_primitives) {
    PresetOut o;
    switch (preset) {
        case PRT_PRESET_DEFAULT: emit_rows(o, kDefault, sizeof(kDefault) / sizeof(Row)); break;
        case PRT_PRESET_LIGHT_TEST: preset_light_test(o); break;
        case PRT_PRESET_MATERIAL_TEST: emit_rows(o, kMaterialTest, sizeof(kMaterialTest) / sizeof(Row)); break;
        case PRT_PRESET_CORNELL: preset_cornell(o); break;
        case PRT_PRESET_RANDOM_BALLS_SMALL: preset_random_balls(o, 100); break;
        case PRT_PRESET_RANDOM_BALLS_MEDIUM: preset_random_balls(o, 400); break;
        case PRT_PRESET_RANDOM_BALLS_LARGE: preset_random_balls(o, 800); break;
        default: return PRT_ERR_INVALID;
    }
    if (materials) memcpy(materials, o.mats.data(), o.mats.size() * sizeof(PrtMaterial));
    if (primitives) memcpy(primitives, o.prims.data(), o.prims.size() * sizeof(PrtPrimitive));
    if (n_materials) *n_materials = (uint32_t)o.mats.size();
    if (n_primitives) *n_primitives = (uint32_t)o.prims.size();
    return PRT_OK;
}

// =================================================================================================
// Meshes
// =================================================================================================
namespace {

void compute_normals(PrtMeshData* m) {
    // Area-weighted vertex normals.  The reference leaves normals EMPTY when the file has none
    // (src/core/mesh.cpp:118-127), which Triangle::Intersect cannot shade; this is the documented
    // substitute.
    const size_t nv = m->pos.size() / 3;
    std::vector<double> acc(3 * nv, 0.0);
    for (size_t t = 0; t < m->idx.size() / 3; ++t) {
        const uint32_t a = m->idx[3 * t], b = m->idx[3 * t + 1], c = m->idx[3 * t + 2];
        const float* pa = &m->pos[3 * (size_t)a];
        const float* pb = &m->pos[3 * (size_t)b];
        const float* pc = &m->pos[3 * (size_t)c];
        const double e1[3] = {(double)pb[0] - pa[0], (double)pb[1] - pa[1], (double)pb[2] - pa[2]};
        const double e2[3] = {(double)pc[0] - pa[0], (double)pc[1] - pa[1], (double)pc[2] - pa[2]};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        for (uint32_t v : {a, b, c})
            for (int k = 0; k < 3; ++k) acc[3 * (size_t)v + k] += n[k];
    }
    m->nrm.resize(3 * nv);
    for (size_t v = 0; v < nv; ++v) {
        const double l = std::sqrt(acc[3 * v] * acc[3 * v] + acc[3 * v + 1] * acc[3 * v + 1] + acc[3 * v + 2] * acc[3 * v + 2]);
        if (l > 0.0) {
            for (int k = 0; k < 3; ++k) m->nrm[3 * v + k] = (float)(acc[3 * v + k] / l);
        } else {
            m->nrm[3 * v] = 0.0f;
            m->nrm[3 * v + 1] = 1.0f;
            m->nrm[3 * v + 2] = 0.0f;
        }
    }
}

enum PlyType { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
PlyType parse_type(const std::string& s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
size_t type_size(PlyType t) {
    switch (t) {
        case T_I8: case T_U8: return 1;
        case T_I16: case T_U16: return 2;
        case T_I32: case T_U32: case T_F32: return 4;
        case T_F64: return 8;
        default: return 0;
    }
}
struct PlyProp {
    std::string name;
    bool is_list = false;
    PlyType type = T_BAD, count_type = T_BAD;
};
struct PlyElem {
    std::string name;
    size_t count = 0;
    std::vector<PlyProp> props;
};

double read_bin(const uint8_t*& p, const uint8_t* end, PlyType t, bool& ok) {
    const size_t sz = type_size(t);
    if (p + sz > end) {
        ok = false;
        return 0;
    }
    double v = 0;
    switch (t) {
        case T_I8: { int8_t x; memcpy(&x, p, 1); v = x; break; }
        case T_U8: { uint8_t x; memcpy(&x, p, 1); v = x; break; }
        case T_I16: { int16_t x; memcpy(&x, p, 2); v = x; break; }
        case T_U16: { uint16_t x; memcpy(&x, p, 2); v = x; break; }
        case T_I32: { int32_t x; memcpy(&x, p, 4); v = x; break; }
        case T_U32: { uint32_t x; memcpy(&x, p, 4); v = x; break; }
        case T_F32: { float x; memcpy(&x, p, 4); v = x; break; }
        case T_F64: { double x; memcpy(&x, p, 8); v = x; break; }
        default: ok = false;
    }
    p += sz;
    return v;
}

}  // namespace

static int load_ply_impl(const char* path, PrtMeshData** out, char* err, size_t err_len);

// No exception crosses the C boundary: an allocation failure on a hostile header (or anything else that throws)
// becomes PRT_ERR_IO like every other malformed file.
extern "C" int prt_mesh_load_ply(const char* path, PrtMeshData** out, char* err, size_t err_len) {
    try {
        return load_ply_impl(path, out, err, err_len);
    } catch (const std::exception& e) {
        if (err && err_len) snprintf(err, err_len, "PLY load failed: %s", e.what());
        return PRT_ERR_IO;
    } catch (...) {
        if (err && err_len) snprintf(err, err_len, "PLY load failed");
        return PRT_ERR_IO;
    }
}

static int load_ply_impl(const char* path, PrtMeshData** out, char* err, size_t err_len) {
    auto fail = [&](const std::string& msg) {
        if (err && err_len) snprintf(err, err_len, "%s", msg.c_str());
        return PRT_ERR_IO;
    };
    if (!path || !out) return fail("null argument");
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(std::string("cannot open ") + path);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    // ---- header ----
    size_t pos = 0;
    auto next_line = [&](std::string& line) {
        if (pos >= buf.size()) return false;
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n') ++e;
        line.assign((const char*)&buf[pos], e - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") return fail("not a PLY file");
    bool binary = false, have_format = false;
    std::vector<PlyElem> elems;
    bool ended = false;
    while (next_line(line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") {
            std::string fmt;
            ss >> fmt;
            if (fmt == "ascii")
                binary = false;
            else if (fmt == "binary_little_endian")
                binary = true;
            else
                return fail("unsupported PLY format " + fmt);
            have_format = true;
        } else if (tok == "element") {
            PlyElem e;
            long long cnt = -1;
            ss >> e.name >> cnt;
            // the header's counts are not trusted: every instance of an element takes at least one byte of the body
            if (ss.fail() || cnt < 0 || (unsigned long long)cnt > buf.size()) return fail("bad element count in: " + line);
            e.count = (size_t)cnt;
            elems.push_back(e);
        } else if (tok == "property") {
            if (elems.empty()) return fail("property before element");
            PlyProp p;
            std::string t;
            ss >> t;
            if (t == "list") {
                std::string ct, it;
                ss >> ct >> it >> p.name;
                p.is_list = true;
                p.count_type = parse_type(ct);
                p.type = parse_type(it);
                if (p.count_type == T_BAD) return fail("bad list count type");
            } else {
                p.type = parse_type(t);
                ss >> p.name;
            }
            if (p.type == T_BAD) return fail("bad property type in: " + line);
            elems.back().props.push_back(p);
        } else if (tok == "end_header") {
            ended = true;
            break;
        }  // comment / obj_info: ignored
    }
    if (!ended || !have_format) return fail("truncated PLY header");

    std::unique_ptr<PrtMeshData> owner(new PrtMeshData());
    PrtMeshData* m = owner.get();
    const uint8_t* p = buf.data() + pos;
    const uint8_t* end = buf.data() + buf.size();
    // ascii tokenizer
    auto next_tok = [&](double& v) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        if (p >= end) return false;
        char tmp[64];
        size_t n = 0;
        while (p < end && !(*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t') && n < 63) tmp[n++] = (char)*p++;
        tmp[n] = 0;
        char* e = nullptr;
        v = strtod(tmp, &e);
        return e != tmp;
    };
    bool ok = true;
    bool have_normals = false;
    for (const PlyElem& e : elems) {
        const bool is_vertex = e.name == "vertex";
        const bool is_face = e.name == "face";
        int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, ilist = -1;
        for (size_t k = 0; k < e.props.size(); ++k) {
            const std::string& n = e.props[k].name;
            if (n == "x") ix = (int)k;
            if (n == "y") iy = (int)k;
            if (n == "z") iz = (int)k;
            if (n == "nx") inx = (int)k;
            if (n == "ny") iny = (int)k;
            if (n == "nz") inz = (int)k;
            if (e.props[k].is_list && (n == "vertex_indices" || n == "vertex_index")) ilist = (int)k;
        }
        if (is_vertex) {
            if (ix < 0 || iy < 0 || iz < 0) return fail("vertex element lacks x/y/z");
            have_normals = inx >= 0 && iny >= 0 && inz >= 0;
            m->pos.resize(3 * e.count);
            if (have_normals) m->nrm.resize(3 * e.count);
        }
        std::vector<double> vals(e.props.size());
        std::vector<uint32_t> poly;
        for (size_t i = 0; i < e.count && ok; ++i) {
            for (size_t k = 0; k < e.props.size() && ok; ++k) {
                const PlyProp& pr = e.props[k];
                if (!pr.is_list) {
                    double v = 0;
                    if (binary)
                        v = read_bin(p, end, pr.type, ok);
                    else
                        ok = next_tok(v);
                    vals[k] = v;
                } else {
                    double c = 0;
                    if (binary)
                        c = read_bin(p, end, pr.count_type, ok);
                    else
                        ok = next_tok(c);
                    // a list length is an integer in [0, bytes left] (one byte per entry at least); anything else is a
                    // malformed body (a negative or non-finite double must not reach the cast)
                    if (!ok || !(c >= 0.0) || !(c <= (double)(end - p))) {
                        ok = false;
                        break;
                    }
                    const size_t cnt = (size_t)c;
                    const bool keep = is_face && (int)k == ilist;
                    if (keep) poly.clear();
                    for (size_t j = 0; j < cnt && ok; ++j) {
                        double v = 0;
                        if (binary)
                            v = read_bin(p, end, pr.type, ok);
                        else
                            ok = next_tok(v);
                        if (keep) poly.push_back((uint32_t)(int64_t)v);
                    }
                }
            }
            if (!ok) break;
            if (is_vertex) {
                m->pos[3 * i + 0] = (float)vals[(size_t)ix];
                m->pos[3 * i + 1] = (float)vals[(size_t)iy];
                m->pos[3 * i + 2] = (float)vals[(size_t)iz];
                if (have_normals) {
                    m->nrm[3 * i + 0] = (float)vals[(size_t)inx];
                    m->nrm[3 * i + 1] = (float)vals[(size_t)iny];
                    m->nrm[3 * i + 2] = (float)vals[(size_t)inz];
                }
            } else if (is_face && ilist >= 0) {
                // The reference assumes triangles (mesh.cpp:96,132); polygons are fan-triangulated here.
                for (size_t j = 1; j + 1 < poly.size(); ++j) {
                    m->idx.push_back(poly[0]);
                    m->idx.push_back(poly[j]);
                    m->idx.push_back(poly[j + 1]);
                }
            }
        }
        if (!ok) break;
    }
    if (!ok) return fail("truncated or malformed PLY body");
    const size_t nv = m->pos.size() / 3;
    for (uint32_t i : m->idx)
        if (i >= nv) return fail("face index out of range");
    m->had_normals = have_normals;
    if (!have_normals) compute_normals(m);
    *out = owner.release();
    return PRT_OK;
}

extern "C" int prt_mesh_create(const float* positions, const float* normals, uint32_t n_vertices,
                               const uint32_t* indices, uint32_t n_triangles, PrtMeshData** out) {
    if (!positions || !indices || !out) return PRT_ERR_INVALID;
    for (size_t i = 0; i < 3 * (size_t)n_triangles; ++i)
        if (indices[i] >= n_vertices) return PRT_ERR_INVALID;
    PrtMeshData* m = new PrtMeshData();
    m->pos.assign(positions, positions + 3 * (size_t)n_vertices);
    m->idx.assign(indices, indices + 3 * (size_t)n_triangles);
    m->had_normals = normals != nullptr;
    if (normals)
        m->nrm.assign(normals, normals + 3 * (size_t)n_vertices);
    else
        compute_normals(m);
    *out = m;
    return PRT_OK;
}
extern "C" void prt_mesh_free(PrtMeshData* m) { delete m; }
extern "C" uint32_t prt_mesh_vertex_count(const PrtMeshData* m) { return m ? (uint32_t)(m->pos.size() / 3) : 0; }
extern "C" uint32_t prt_mesh_triangle_count(const PrtMeshData* m) { return m ? (uint32_t)(m->idx.size() / 3) : 0; }
extern "C" const float* prt_mesh_positions(const PrtMeshData* m) { return m ? m->pos.data() : nullptr; }
extern "C" const float* prt_mesh_normals(const PrtMeshData* m) { return m ? m->nrm.data() : nullptr; }
extern "C" const uint32_t* prt_mesh_indices(const PrtMeshData* m) { return m ? m->idx.data() : nullptr; }
extern "C" int prt_mesh_had_normals(const PrtMeshData* m) { return m && m->had_normals ? 1 : 0; }

// Deterministic longest-edge bisection: repeatedly split the globally longest edge (ties: smaller
// vertex pair) at its midpoint, splitting every triangle that shares it, until the triangle count
// reaches the target.  New vertex: position = (a+b)*0.5, normal = normalize(na+nb).
extern "C" int prt_mesh_refine(PrtMeshData* m, uint32_t target) {
    if (!m) return PRT_ERR_INVALID;
    struct EdgeRec {
        int32_t tri[2];
    };
    struct HeapItem {
        float len2;
        uint32_t a, b;
        bool operator<(const HeapItem& o) const {
            if (len2 != o.len2) return len2 < o.len2;
            if (a != o.a) return a > o.a;
            return b > o.b;
        }
    };
    auto key = [](uint32_t a, uint32_t b) { return ((uint64_t)std::min(a, b) << 32) | std::max(a, b); };
    auto len2 = [&](uint32_t a, uint32_t b) {
        const float dx = m->pos[3 * (size_t)a] - m->pos[3 * (size_t)b];
        const float dy = m->pos[3 * (size_t)a + 1] - m->pos[3 * (size_t)b + 1];
        const float dz = m->pos[3 * (size_t)a + 2] - m->pos[3 * (size_t)b + 2];
        return (dx * dx + dy * dy) + dz * dz;
    };
    std::unordered_map<uint64_t, EdgeRec> edges;
    edges.reserve((size_t)target * 2);
    std::priority_queue<HeapItem> heap;
    bool non_manifold = false;
    auto attach = [&](uint32_t a, uint32_t b, int32_t t) {
        auto it = edges.find(key(a, b));
        if (it == edges.end()) {
            edges.emplace(key(a, b), EdgeRec{{t, -1}});
            heap.push(HeapItem{len2(std::min(a, b), std::max(a, b)), std::min(a, b), std::max(a, b)});
        } else if (it->second.tri[0] < 0) {
            it->second.tri[0] = t;
        } else if (it->second.tri[1] < 0) {
            it->second.tri[1] = t;
        } else {
            non_manifold = true;  // a third triangle on one edge: it is simply not split with the others
        }
    };
    auto detach = [&](uint32_t a, uint32_t b, int32_t t) {
        auto it = edges.find(key(a, b));
        if (it == edges.end()) return;
        if (it->second.tri[0] == t) it->second.tri[0] = -1;
        else if (it->second.tri[1] == t) it->second.tri[1] = -1;
    };
    const size_t nt0 = m->idx.size() / 3;
    for (size_t t = 0; t < nt0; ++t)
        for (int k = 0; k < 3; ++k) attach(m->idx[3 * t + k], m->idx[3 * t + (k + 1) % 3], (int32_t)t);
    if (non_manifold) return PRT_ERR_INVALID;
    while (m->idx.size() / 3 < target && !heap.empty()) {
        const HeapItem top = heap.top();
        heap.pop();
        auto it = edges.find(key(top.a, top.b));
        if (it == edges.end()) continue;
        const EdgeRec rec = it->second;
        edges.erase(it);
        const uint32_t a = top.a, b = top.b;
        const uint32_t mid = (uint32_t)(m->pos.size() / 3);
        float nn[3];
        for (int k = 0; k < 3; ++k) {
            m->pos.push_back((m->pos[3 * (size_t)a + k] + m->pos[3 * (size_t)b + k]) * 0.5f);
            nn[k] = m->nrm[3 * (size_t)a + k] + m->nrm[3 * (size_t)b + k];
        }
        const float l = std::sqrt((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]);
        for (int k = 0; k < 3; ++k) m->nrm.push_back(l > 0.0f ? nn[k] / l : m->nrm[3 * (size_t)a + k]);
        for (int side = 0; side < 2; ++side) {
            const int32_t t = rec.tri[side];
            if (t < 0) continue;
            uint32_t v[3] = {m->idx[3 * (size_t)t], m->idx[3 * (size_t)t + 1], m->idx[3 * (size_t)t + 2]};
            int kc = 0;  // position of the vertex opposite the edge
            for (int k = 0; k < 3; ++k)
                if (v[k] != a && v[k] != b) kc = k;
            const uint32_t c = v[kc], p = v[(kc + 1) % 3], q = v[(kc + 2) % 3];
            // (c,p,q) -> (c,p,mid) in place + (c,mid,q) appended; winding preserved
            const int32_t t2 = (int32_t)(m->idx.size() / 3);
            m->idx[3 * (size_t)t] = c;
            m->idx[3 * (size_t)t + 1] = p;
            m->idx[3 * (size_t)t + 2] = mid;
            m->idx.push_back(c);
            m->idx.push_back(mid);
            m->idx.push_back(q);
            detach(q, c, t);       // edge (q,c) now belongs to the new triangle
            attach(q, c, t2);
            attach(p, mid, t);
            attach(mid, q, t2);
            attach(c, mid, t);
            attach(c, mid, t2);
        }
    }
    return PRT_OK;
}

extern "C" int prt_mesh_transform(PrtMeshData* m, const float mat[16], const float inv[16]) {
    if (!m || !mat || !inv) return PRT_ERR_INVALID;
    const size_t nv = m->pos.size() / 3;
    for (size_t v = 0; v < nv; ++v) {
        const float x = m->pos[3 * v], y = m->pos[3 * v + 1], z = m->pos[3 * v + 2];
        for (int r = 0; r < 3; ++r)  // TransformPoint, glm mat4*vec4 grouping
            m->pos[3 * v + r] = (mat[r] * x + mat[4 + r] * y) + (mat[8 + r] * z + mat[12 + r] * 1.0f);
        const float nx = m->nrm[3 * v], ny = m->nrm[3 * v + 1], nz = m->nrm[3 * v + 2];
        float o[3];
        for (int r = 0; r < 3; ++r)  // TransformNormal(inv, n) = normalize(mat3(transpose(inv)) * n)
            o[r] = inv[4 * r] * nx + inv[4 * r + 1] * ny + inv[4 * r + 2] * nz;
        const float l = 1.0f / std::sqrt((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]);
        for (int r = 0; r < 3; ++r) m->nrm[3 * v + r] = o[r] * l;
    }
    return PRT_OK;
}

extern "C" int prt_mesh_append(PrtMeshData* dst, const PrtMeshData* src) {
    if (!dst || !src) return PRT_ERR_INVALID;
    const uint32_t base = (uint32_t)(dst->pos.size() / 3);
    dst->pos.insert(dst->pos.end(), src->pos.begin(), src->pos.end());
    dst->nrm.insert(dst->nrm.end(), src->nrm.begin(), src->nrm.end());
    for (uint32_t i : src->idx) dst->idx.push_back(base + i);
    return PRT_OK;
}

// =================================================================================================
// Framebuffer dumps
// =================================================================================================
extern "C" int prt_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8) return PRT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return PRT_ERR_IO;
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row(3 * (size_t)width);
    for (uint32_t y = 0; y < height; ++y) {  // row 0 = top of the image, as PPM wants
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[3 * (size_t)x + c] = rgba8[4 * ((size_t)y * width + x) + c];
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return PRT_OK;
}

extern "C" int prt_write_pfm(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb) return PRT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return PRT_ERR_IO;
    fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    for (uint32_t y = 0; y < height; ++y)  // PFM stores the bottom row first
        fwrite(rgb + 3 * (size_t)(height - 1 - y) * width, sizeof(float), 3 * (size_t)width, f);
    fclose(f);
    return PRT_OK;
}
