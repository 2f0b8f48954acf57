// prt_mesh.h — the mesh container behind the opaque PrtMeshData handle of include/prt.h
// (same three arrays the reference's Mesh exposes: src/core/mesh.h:12-14).
#pragma once
#include <stdint.h>

#include <vector>

struct PrtMeshData {
    std::vector<float> pos;     // 3 per vertex
    std::vector<float> nrm;     // 3 per vertex
    std::vector<uint32_t> idx;  // 3 per triangle
    bool had_normals = false;
};
