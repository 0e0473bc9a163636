// kouek::RayTracer::LBVH with the reference's public interface (reference
// include/ray_tracer/lbvh.h:20-78): BuildFrom(InputMesh, logPath), IsComplete(),
// GetDeivceDataDevicePtr() (sic), GetDeviceData(). The build runs in kk_lbvh_build.
#ifndef KOUEK_RAY_TRACER_LBVH_H
#define KOUEK_RAY_TRACER_LBVH_H

#include <filesystem>
#include <fstream>
#include <iostream>
#include <string_view>
#include <vector>

#include "../hip/device.h"

namespace kouek {

struct AABB {
    vec3 minPos, maxPos;
};

namespace RayTracer {

class LBVH : Noncopyable {
  public:
    using IndexTy = uint32_t;

    struct alignas(16) InternalNode {
        uint32_t lft;
        uint32_t rht;
        AABB aabb;
        static constexpr auto ChildIsLeafMask = uint32_t(0x80000000);
        static constexpr auto LeafIndexMask = uint32_t(0x7fffffff);
    };
    struct alignas(16) DeviceData {
        AABB rootAABB;
        IndexTy faceNum;
        IndexTy compactedFaceNum;
        IndexTy *faces;
        IndexTy *compactedFaces;
        InternalNode *internalNodes;
    };
    static_assert(sizeof(InternalNode) == sizeof(kk_lbvh_node) && sizeof(DeviceData) == sizeof(kk_lbvh_device_data),
                  "layouts must match the C ABI");

  private:
    bool isComplete = false;
    HIP::DeviceBuffer<IndexTy> d_faces;
    HIP::DeviceBuffer<IndexTy> d_compactedFaces;
    HIP::DeviceBuffer<InternalNode> d_internalNodes;
    HIP::DeviceBuffer<float> d_positions;
    HIP::DeviceBuffer<IndexTy> d_facePositionIndices;
    HIP::DeviceBuffer<uint8_t> d_workspace;
    size_t leafAABBsOffs = 0; // of the per-leaf boxes inside d_workspace
    HIP::DeviceBuffer<DeviceData> d_deviceDat;
    DeviceData deviceDat{};

    static constexpr std::string_view StatTag = "[LBVH Status]";
    static constexpr std::string_view ErrTag = "[LBVH Error]";

  public:
    struct InputMesh {
        const std::vector<vec3> &positions;
        const std::vector<uvec3> &facePositionIndices;
    };

    void BuildFrom(const InputMesh &inputMesh, const std::filesystem::path &logPath = "") {
        isComplete = false;
        clear();
        const uint32_t V = static_cast<uint32_t>(inputMesh.positions.size());
        const uint32_t N = static_cast<uint32_t>(inputMesh.facePositionIndices.size());
        kk_lbvh_ws_layout lay;
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_lbvh_workspace_layout(N, V, 0, &lay)))
            return;
        static_assert(sizeof(vec3) == 12 && sizeof(uvec3) == 12, "tightly packed vectors expected");
        bool ok = d_positions.resize(3ull * V) && d_facePositionIndices.resize(3ull * N) &&
                  d_faces.resize(N) && d_compactedFaces.resize(N) && d_internalNodes.resize(N) &&
                  d_workspace.resize(lay.total_bytes) && (d_deviceDat.size() || d_deviceDat.resize(1));
        ok = ok && kk_memcpy_h2d(d_positions.data(), inputMesh.positions.data(), 12ull * V, nullptr) == KK_OK;
        ok = ok && kk_memcpy_h2d(d_facePositionIndices.data(), inputMesh.facePositionIndices.data(), 12ull * N, nullptr) == KK_OK;
        if (!ok) {
            std::cerr << ErrTag << " " << kk_last_error() << "\n";
            return;
        }
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_lbvh_build(d_positions.data(), V, d_facePositionIndices.data(), N,
                                           d_faces.data(), d_compactedFaces.data(),
                                           reinterpret_cast<kk_lbvh_node *>(d_internalNodes.data()),
                                           reinterpret_cast<kk_lbvh_device_data *>(d_deviceDat.data()),
                                           d_workspace.data(), lay.total_bytes, 0, nullptr)))
            return;
        if (kk_memcpy_d2h(&deviceDat, d_deviceDat.data(), sizeof(deviceDat), nullptr) != KK_OK)
            return;
        uint32_t sortError = 0;
        kk_memcpy_d2h(&sortError, d_workspace.data() + lay.error_flag, 4, nullptr);
        if (sortError != 0) {
            std::cerr << ErrTag << " the sort reported a device-side failure.\n";
            return;
        }
        std::cout << StatTag << " Builded LBVH.\n\tcompact num: " << deviceDat.compactedFaceNum
                  << ", face num: " << deviceDat.faceNum << "\n";
        leafAABBsOffs = lay.leaf_aabbs;
        isComplete = true;
        if (!logPath.empty())
            writeLog(inputMesh, logPath);
    }

    bool IsComplete() const { return isComplete; }
    const DeviceData *GetDeivceDataDevicePtr() const { return d_deviceDat.data(); }
    const DeviceData &GetDeviceData() const { return deviceDat; }

    // Device copies of the mesh arrays the tree indexes; the renderer shares them.
    const float *GetDevicePositions() const { return d_positions.data(); }
    const IndexTy *GetDeviceFacePositionIndices() const { return d_facePositionIndices.data(); }
    // float[compactedFaceNum][6]: the boxes of the leaves, left in the workspace by the build.
    const float *GetDeviceLeafAABBs() const {
        return isComplete ? reinterpret_cast<const float *>(d_workspace.data() + leafAABBsOffs) : nullptr;
    }

  private:
    void clear() {
        d_faces.clear();
        d_compactedFaces.clear();
        d_internalNodes.clear();
        d_positions.clear();
        d_facePositionIndices.clear();
        d_workspace.clear();
    }

    // The text log of the reference (src/ray_tracer/lbvh.cu:330-384), same sections and line formats.
    void writeLog(const InputMesh &in, const std::filesystem::path &logPath) const {
        std::ofstream os(logPath, std::ios::out);
        if (!os.is_open()) {
            std::cerr << ErrTag << " Cannot open log file at " << logPath.string() << ".\n";
            return;
        }
        os << "[OBJ]\n";
        for (IndexTy i = 0; i < in.facePositionIndices.size(); ++i)
            os << "of:" << i << "-> v0:" << in.facePositionIndices[i][0] + 1
               << ", v1:" << in.facePositionIndices[i][1] + 1 << ", v2:" << in.facePositionIndices[i][2] + 1 << "\n";
        os << "\n[LBVH/Compacts]\n";
        auto compacts = d_compactedFaces.to_host(deviceDat.compactedFaceNum);
        auto faces = d_faces.to_host();
        for (IndexTy i = 0; i < compacts.size(); ++i) {
            auto fi = compacts[i];
            auto fiEnd = i == compacts.size() - 1 ? faces.size() - 1 : compacts[i + 1];
            os << "c:" << i << "-> ";
            while (fi < fiEnd) {
                os << "f:" << fi << "->of:" << faces[fi] << (fi == fiEnd - 1 ? "" : ", ");
                ++fi;
            }
            os << "\n";
        }
        os << "\n[LBVH/INodes]\n";
        auto inodes = d_internalNodes.to_host(deviceDat.compactedFaceNum > 1 ? deviceDat.compactedFaceNum - 1 : 0);
        for (IndexTy i = 0; i < inodes.size(); ++i) {
            bool lLeaf = (inodes[i].lft & InternalNode::ChildIsLeafMask) != 0;
            bool rLeaf = (inodes[i].rht & InternalNode::ChildIsLeafMask) != 0;
            os << "i:" << i << "-> lc:" << (inodes[i].lft & InternalNode::LeafIndexMask) << (lLeaf ? "L" : "I")
               << ", rc:" << (inodes[i].rht & InternalNode::LeafIndexMask) << (rLeaf ? "L" : "I")
               << ", aabb-> min:(" << inodes[i].aabb.minPos.x << "," << inodes[i].aabb.minPos.y << ","
               << inodes[i].aabb.minPos.z << "), max:(" << inodes[i].aabb.maxPos.x << ","
               << inodes[i].aabb.maxPos.y << "," << inodes[i].aabb.maxPos.z << ")\n";
        }
        std::cout << StatTag << " Log LBVH to " << logPath.string() << ".\n";
    }
};

} // namespace RayTracer
} // namespace kouek

#endif // KOUEK_RAY_TRACER_LBVH_H
