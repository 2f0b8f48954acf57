// prt_renderer.hpp — C++ host-side mirror of the reference's Scene / Camera / Film / Renderer interfaces over
// the C-ABI of include/prt.h (header-only).  Same method names and call contract as the reference:
//   class Renderer { Init(Film&, const Scene&, const Camera&); ProgressiveRender(); SetCamera(const Camera&); }
//   (reference: src/core/renderer.h:8-16)
// Nothing here computes pixels; the bodies only marshal PODs into prt_* calls.  INTEGRATION.md shows the same
// adapter written against the reference's own headers.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/prt.h"

namespace prt {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Scene(preset) (reference: src/core/scene.h:17-62); materials/primitives flattened as PrtSceneDesc wants them.
class Scene {
public:
    explicit Scene(int preset = PRT_PRESET_RANDOM_BALLS_LARGE) {  // default preset: src/core/scene.h:20
        uint32_t nm = 0, np = 0;
        if (prt_scene_preset(preset, nullptr, &nm, nullptr, &np)) throw Error("unknown scene preset");
        materials.resize(nm);
        primitives.resize(np);
        prt_scene_preset(preset, materials.data(), &nm, primitives.data(), &np);
    }
    struct Empty {};
    explicit Scene(Empty) {}
    ~Scene() {
        for (PrtMeshData* m : owned_) prt_mesh_free(m);
    }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    uint32_t AddMaterial(uint32_t type, float r, float g, float b, float scalar) {
        materials.push_back(PrtMaterial{type, {r, g, b}, scalar});
        return (uint32_t)materials.size() - 1;
    }
    void AddPrimitive(uint32_t shape, float p0, float p1, uint32_t material, const float scale[3], const float euler_deg[3],
                      const float translation[3]) {  // Scene::AddPrimitive + MakeTransform, src/core/scene.cpp:9-36
        PrtPrimitive p{};
        p.shape_type = shape;
        p.shape_param[0] = p0;
        p.shape_param[1] = p1;
        p.material_id = material;
        prt_make_transform(scale, euler_deg, translation, p.mat, p.inv);
        primitives.push_back(p);
    }
    // Mesh(plyFilePath) (reference: src/core/mesh.h:8-21), appended as world-space triangles
    void AddMeshPly(const std::string& path, uint32_t material, uint32_t refine_to = 0) {
        PrtMeshData* m = nullptr;
        char err[256] = {0};
        if (prt_mesh_load_ply(path.c_str(), &m, err, sizeof(err))) throw Error(std::string("PLY: ") + err);
        if (refine_to > prt_mesh_triangle_count(m) && prt_mesh_refine(m, refine_to)) {
            prt_mesh_free(m);
            throw Error("mesh refinement failed (non-manifold edge)");
        }
        owned_.push_back(m);
        meshes.push_back(PrtMesh{prt_mesh_positions(m), prt_mesh_normals(m), prt_mesh_indices(m), prt_mesh_vertex_count(m),
                                 prt_mesh_triangle_count(m), material});
    }
    PrtSceneDesc desc() const {
        PrtSceneDesc d{};
        d.materials = materials.data();
        d.primitives = primitives.data();
        d.meshes = meshes.data();
        d.n_materials = (uint32_t)materials.size();
        d.n_primitives = (uint32_t)primitives.size();
        d.n_meshes = (uint32_t)meshes.size();
        d.sky[0] = sky[0];
        d.sky[1] = sky[1];
        d.sky[2] = sky[2];
        return d;
    }
    std::vector<PrtMaterial> materials;
    std::vector<PrtPrimitive> primitives;
    std::vector<PrtMesh> meshes;
    float sky[3] = {0.4f, 0.3f, 0.6f};  // src/backend/cpu/renderer.h:31

private:
    std::vector<PrtMeshData*> owned_;
};

// Camera(position, front, width, height) (reference: src/core/camera.h:10-16)
struct Camera {
    float position[3] = {5.0f, 5.0f, 8.0f};  // src/main.cpp:142
    float front[3] = {-5.0f, -5.0f, -8.0f};
    float width = 1920.0f, height = 1080.0f;
    PrtCameraDesc desc() const {
        PrtCameraDesc d{};
        for (int k = 0; k < 3; ++k) {
            d.position[k] = position[k];
            d.front[k] = front[k];
        }
        d.width = width;
        d.height = height;
        return d;
    }
};

// Film(width, height) (reference: src/core/film.h:10-76): host copies of the accumulation buffers.
class Film {
public:
    Film(uint32_t w, uint32_t h) : width(w), height(h), accum((size_t)w * h * 3), weights((size_t)w * h), display((size_t)w * h * 4) {}
    uint32_t GetWidth() const { return width; }
    uint32_t GetHeight() const { return height; }
    uint32_t width, height;
    std::vector<float> accum, weights;
    std::vector<uint8_t> display;
};

class Renderer {
public:
    virtual ~Renderer() = default;
    virtual void Init(Film& film, const Scene& scene, const Camera& camera) = 0;
    virtual void ProgressiveRender() = 0;
    virtual void SetCamera(const Camera& camera) = 0;
};

// One renderer, one or several GPUs of a node.  devices = {0} is the single-GPU backend; devices = {0, 1, ..., 7} tiles the
// image over eight GPUs (8x8-pixel tiles dealt round-robin, scene replicated, RNG keyed by global pixel and sample, so the
// image does not depend on the device count) and gathers the per-tile radiance to devices[0] once per Render call: RCCL
// (ncclSend / ncclRecv over xGMI) when every rank has its own GPU, peer copies otherwise (include/prt.h, prt_group_*).
class HipWavefrontRenderer : public Renderer {
public:
    explicit HipWavefrontRenderer(int device = 0, uint32_t max_depth = 20 /* src/backend/cpu/renderer.h:34 */, uint32_t seed = 0)
        : HipWavefrontRenderer(std::vector<int>{device}, max_depth, seed) {}
    explicit HipWavefrontRenderer(const std::vector<int>& devices, uint32_t max_depth = 20, uint32_t seed = 0) : max_depth_(max_depth), seed_(seed) {
        if (prt_group_create(devices.data(), (uint32_t)devices.size(), &grp_)) {
            std::string msg = prt_group_last_error(grp_);
            prt_group_destroy(grp_);
            grp_ = nullptr;
            throw Error("prt_group_create: " + msg);
        }
    }
    ~HipWavefrontRenderer() override { prt_group_destroy(grp_); }
    HipWavefrontRenderer(const HipWavefrontRenderer&) = delete;
    HipWavefrontRenderer& operator=(const HipWavefrontRenderer&) = delete;
    void Init(Film& film, const Scene& scene, const Camera& camera) override {
        PrtSceneDesc d = scene.desc();
        check(prt_group_set_scene(grp_, &d));  // flattened + BVH built once, cloned to the other GPUs
        check(prt_group_set_film(grp_, film.width, film.height));
        film_ = &film;
        frame_ = 0;
        SetCamera(camera);
    }
    void SetCamera(const Camera& camera) override {
        PrtCameraDesc d = camera.desc();
        check(prt_group_set_camera(grp_, &d));
    }
    void ProgressiveRender() override { Render(1); }  // exactly one sample per pixel
    void Render(uint32_t spp) {
        check(prt_group_render(grp_, spp, max_depth_, seed_, frame_));
        frame_ += spp;
    }
    void SetSamplesInFlight(uint32_t n) { check(prt_group_set_samples_in_flight(grp_, n)); }
    void SetParam(const char* name, int value) { check(prt_group_set_param(grp_, name, value)); }
    void Clear() {
        check(prt_group_film_clear(grp_));
        frame_ = 0;
    }
    void Download() { check(prt_group_film_read(grp_, film_->accum.data(), film_->weights.data())); }
    void UpdateDisplay(float exposure = 1.0f, float gamma = 2.2f) { check(prt_group_film_display(grp_, exposure, gamma, film_->display.data())); }
    PrtStats Stats() {
        PrtStats s{};
        check(prt_group_get_stats(grp_, &s));
        return s;
    }
    uint32_t DeviceCount() const { return prt_group_size(grp_); }
    const char* Transport() const { return prt_group_transport(grp_); }
    PrtContext* context(uint32_t rank = 0) { return prt_group_context(grp_, rank); }

private:
    void check(int rc) {
        if (rc) throw Error(prt_group_last_error(grp_));
    }
    PrtGroup* grp_ = nullptr;
    Film* film_ = nullptr;
    uint32_t max_depth_, seed_, frame_ = 0;
};

}  // namespace prt
