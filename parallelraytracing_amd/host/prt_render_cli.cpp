// prt_render — offline framebuffer dump: the stand-in for the reference's GLFW/ImGui/OpenGL viewer
// (src/main.cpp:138-170 builds Film/Scene/Camera and Inits the backends; :504-527 is the frame loop).
//   prt_render [--preset NAME | --ply FILE [--refine N]] [--width W --height H] [--spp N] [--depth D]
//              [--seed S] [--camera x y z] [--out PREFIX]
// Writes PREFIX.ppm (tonemapped RGBA8 as PPM) and PREFIX.pfm (mean radiance).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "prt_renderer.hpp"

static int preset_id(const std::string& n) {
    const char* names[] = {"DEFAULT", "LIGHT_TEST", "MATERIAL_TEST", "CORNELL", "RANDOM_BALLS_SMALL", "RANDOM_BALLS_MEDIUM", "RANDOM_BALLS_LARGE"};
    for (int i = 0; i < 7; ++i)
        if (n == names[i]) return i;
    return -1;
}

int main(int argc, char** argv) {
    std::string preset = "CORNELL", ply, out = "frame";
    uint32_t W = 256, H = 256, spp = 1, depth = 2, seed = 0, refine = 0;
    float cam[3] = {5.0f, 5.0f, 8.0f};
    bool cam_set = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--preset") preset = next();
        else if (a == "--ply") ply = next();
        else if (a == "--refine") refine = (uint32_t)atoi(next());
        else if (a == "--width") W = (uint32_t)atoi(next());
        else if (a == "--height") H = (uint32_t)atoi(next());
        else if (a == "--spp") spp = (uint32_t)atoi(next());
        else if (a == "--depth") depth = (uint32_t)atoi(next());
        else if (a == "--seed") seed = (uint32_t)atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--camera") { for (int k = 0; k < 3; ++k) cam[k] = (float)atof(next()); cam_set = true; }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        std::unique_ptr<prt::Scene> scene;
        if (!ply.empty()) {
            scene.reset(new prt::Scene(prt::Scene::Empty{}));
            const uint32_t ground = scene->AddMaterial(PRT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0);
            const uint32_t light = scene->AddMaterial(PRT_MAT_EMISSIVE, 15, 15, 15, 0);
            const uint32_t body = scene->AddMaterial(PRT_MAT_LAMBERTIAN, 0.8f, 0.8f, 0.8f, 0);
            const float one[3] = {1, 1, 1}, zero[3] = {0, 0, 0}, flip[3] = {180, 0, 0}, gt[3] = {0, -1, 0}, lt[3] = {0, 5, 0};
            scene->AddPrimitive(PRT_SHAPE_QUAD, 20, 20, ground, one, zero, gt);
            scene->AddPrimitive(PRT_SHAPE_QUAD, 4, 4, light, one, flip, lt);
            scene->AddMeshPly(ply, body, refine);
            if (!cam_set) { cam[0] = 1.2f; cam[1] = 0.4f; cam[2] = 1.9f; }
        } else {
            const int id = preset_id(preset);
            if (id < 0) { fprintf(stderr, "unknown preset %s\n", preset.c_str()); return 2; }
            scene.reset(new prt::Scene(id));
        }
        prt::Camera camera;
        for (int k = 0; k < 3; ++k) { camera.position[k] = cam[k]; camera.front[k] = -cam[k]; }
        camera.width = (float)W;
        camera.height = (float)H;
        prt::Film film(W, H);
        prt::HipWavefrontRenderer r(0, depth, seed);
        r.Init(film, *scene, camera);
        const auto t0 = std::chrono::steady_clock::now();
        r.Render(spp);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        r.Download();
        r.UpdateDisplay();
        const PrtStats st = r.Stats();
        std::vector<float> mean((size_t)W * H * 3);
        for (size_t i = 0; i < (size_t)W * H; ++i)
            for (int c = 0; c < 3; ++c) mean[3 * i + c] = film.weights[i] > 0 ? film.accum[3 * i + c] / film.weights[i] : 0.0f;
        if (prt_write_ppm((out + ".ppm").c_str(), film.display.data(), W, H) || prt_write_pfm((out + ".pfm").c_str(), mean.data(), W, H)) {
            fprintf(stderr, "cannot write %s.ppm/.pfm\n", out.c_str());
            return 1;
        }
        printf("%ux%u, %u spp, max_depth %u: %llu rays in %.3f s = %.1f Mrays/s -> %s.ppm, %s.pfm\n", W, H, spp, depth,
               (unsigned long long)st.rays_total, s, st.rays_total / s / 1e6, out.c_str(), out.c_str());
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
