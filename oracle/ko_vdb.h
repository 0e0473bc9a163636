// TEST INFRASTRUCTURE ONLY (see ko_math.h). Plain-data mirror of the Depth-Box VDB structures
// (include/ray_caster/dpbx_vdb.h:39-93) shared by the CPU restatements of the build and of the marcher.
#pragma once
#include "ko_math.h"

#include <vector>

namespace ko {

constexpr int MaxLevelNum = 3;
constexpr uint32_t UndefRltIdx = 0xFFFFFFFFu;

// dpbx_vdb.h:39-54, same field order and alignment (128 bytes)
struct alignas(16) VDBParameter {
    uint8_t rootLev;
    uint8_t apronWid;
    uint8_t apronDepWid;
    uint8_t log2Dims[MaxLevelNum];
    int32_t dims[MaxLevelNum];
    int32_t voxPerAtlasBrick;
    int32_t minDepPosValInBrick;
    int32_t maxDepPosValInBrick;
    int32_t atlasBrickPerVDB[3];
    uint64_t nodeStarts[MaxLevelNum];
    uint64_t childStarts[MaxLevelNum];
    float voxsPerChild[MaxLevelNum];
    float voxPerVDB;
    float voxPerVol[3];
};
static_assert(sizeof(VDBParameter) == 128, "VDBParameter");

// dpbx_vdb.h:55-59 (32 bytes)
struct alignas(16) Node {
    int32_t pos[3];
    int32_t brickPosInAtlas[3];
    uint64_t childPoolOffsRelative;
};
static_assert(sizeof(Node) == 32, "Node");

// The built structure on the host. The atlas is kept brick by brick: brick k (the k-th leaf node, whose
// brickPosInAtlas is (k % lx, (k / lx) % ly, k / (lx ly))) occupies voxPerAtlasBrick^3 voxels, x
// fastest. The reference stores the same voxels in one 3-D CUDA array; only the addressing differs.
template <typename T> struct VDB {
    VDBParameter param;
    std::vector<Node> nodePools;
    std::vector<uint32_t> childPools;
    std::vector<uint32_t> atlasBrickToNodeIndices;
    std::vector<T> atlas;
    uint32_t leafNum = 0;

    const Node &GetNode(uint8_t lev, uint32_t rel) const {
        return nodePools[param.nodeStarts[lev] + rel];
    }
    Node &GetNode(uint8_t lev, uint32_t rel) { return nodePools[param.nodeStarts[lev] + rel]; }
    uint32_t PosRelativeToIdxRelative(uint8_t parLev, const int32_t p[3]) const {
        uint8_t l = param.log2Dims[parLev];
        return (static_cast<uint32_t>(p[2]) << (l << 1)) | (static_cast<uint32_t>(p[1]) << l) |
               static_cast<uint32_t>(p[0]);
    }
    uint32_t GetChild(uint8_t parLev, uint64_t offsRelative, const int32_t p[3]) const {
        return childPools[param.childStarts[parLev - 1] + offsRelative +
                          PosRelativeToIdxRelative(parLev, p)];
    }
};

} // namespace ko
