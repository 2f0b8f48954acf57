// prt_render — offline framebuffer dump: the stand-in for the reference's GLFW/ImGui/OpenGL viewer
// (src/main.cpp:138-170 builds Film/Scene/Camera and Inits the backends; :504-527 is the frame loop).
//   prt_render [--preset NAME | --ply FILE [--refine N]] [--width W --height H] [--spp N] [--depth D]
//              [--seed S] [--camera x y z] [--out PREFIX] [--gpus N | --devices a,b,c] [--sif S] [--frames K]
// --gpus N tiles the image over devices 0..N-1 (--devices: any list; a device may repeat, which rehearses the multi-GPU
// path on one GPU); the frame is gathered to the first device once per frame (RCCL over xGMI, or peer copies).
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
    uint32_t W = 256, H = 256, spp = 1, depth = 2, seed = 0, refine = 0, sif = 0, frames = 1;
    std::vector<int> devices{0};
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
        else if (a == "--sif") sif = (uint32_t)atoi(next());
        else if (a == "--frames") frames = (uint32_t)atoi(next());
        else if (a == "--gpus") { const int n = atoi(next()); devices.clear(); for (int d = 0; d < n; ++d) devices.push_back(d); }
        else if (a == "--devices") { devices.clear(); std::string l = next(); for (size_t p = 0; p < l.size();) { size_t e = l.find(',', p); if (e == std::string::npos) e = l.size(); devices.push_back(atoi(l.substr(p, e - p).c_str())); p = e + 1; } }
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
        if (devices.empty()) { fprintf(stderr, "no devices\n"); return 2; }
        prt::HipWavefrontRenderer r(devices, depth, seed);
        r.Init(film, *scene, camera);
        if (sif) r.SetSamplesInFlight(sif);
        if (frames > 1) {  // warm-up frame (first-touch allocations, clocks), then the timed ones
            r.Render(spp);
            r.Clear();
        }
        const PrtStats st0 = r.Stats();
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t f = 0; f < frames; ++f) r.Render(spp);  // each call ends with the gather to the first device
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
        printf("%ux%u, %u spp, max_depth %u, %u GPU(s) [gather: %s]: %llu rays in %.3f s = %.1f Mrays/s -> %s.ppm, %s.pfm\n", W, H,
               spp * frames, depth, r.DeviceCount(), r.Transport(), (unsigned long long)(st.rays_total - st0.rays_total), s,
               (st.rays_total - st0.rays_total) / s / 1e6,
               out.c_str(), out.c_str());
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
