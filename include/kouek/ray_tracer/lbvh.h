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
    // The tree and the mesh it indexes live in one allocation (kk_lbvh_blob_layout_of: DeviceData,
    // positions, face indices, faces, compactedFaces, internalNodes), so that Broadcast() is one
    // collective; the reference's separate thrust vectors (lbvh.h:47-52) are pieces of it.
    HIP::DeviceBuffer<uint8_t> d_blob;
    kk_lbvh_blob_layout blob{};
    uint32_t faceNum = 0, vertNum = 0;
    HIP::DeviceBuffer<uint8_t> d_workspace;
    size_t leafAABBsOffs = 0; // of the per-leaf boxes inside d_workspace
    bool hasLeafAABBs = false;
    DeviceData deviceDat{};

    template <typename T> T *piece(uint64_t offs) const { return reinterpret_cast<T *>(d_blob.data() + offs); }

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
        // the kernels follow these indices into the positions: one past the array is a GPU memory fault
        for (uint32_t fi = 0; fi < N; ++fi)
            for (int i = 0; i < 3; ++i)
                if (inputMesh.facePositionIndices[fi][i] >= V) {
                    std::cerr << ErrTag << " Face " << fi << " refers to position "
                              << inputMesh.facePositionIndices[fi][i] << ", the mesh has " << V << ".\n";
                    return;
                }
        kk_lbvh_ws_layout lay;
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_lbvh_workspace_layout(N, V, 0, &lay)))
            return;
        static_assert(sizeof(vec3) == 12 && sizeof(uvec3) == 12, "tightly packed vectors expected");
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_lbvh_blob_layout_of(N, V, &blob)))
            return;
        faceNum = N, vertNum = V;
        bool ok = d_blob.resize(blob.total) && d_workspace.resize(lay.total_bytes);
        ok = ok && kk_memcpy_h2d(piece<float>(blob.positions), inputMesh.positions.data(), 12ull * V, nullptr) == KK_OK;
        ok = ok && kk_memcpy_h2d(piece<IndexTy>(blob.face_pos_idx), inputMesh.facePositionIndices.data(), 12ull * N, nullptr) == KK_OK;
        if (!ok) {
            std::cerr << ErrTag << " " << kk_last_error() << "\n";
            return;
        }
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_lbvh_build(piece<float>(blob.positions), V, piece<IndexTy>(blob.face_pos_idx), N,
                                           piece<IndexTy>(blob.faces), piece<IndexTy>(blob.compacted),
                                           piece<kk_lbvh_node>(blob.nodes),
                                           piece<kk_lbvh_device_data>(blob.device_data), d_workspace.data(),
                                           lay.total_bytes, 0, nullptr)))
            return;
        if (kk_memcpy_d2h(&deviceDat, piece<DeviceData>(blob.device_data), sizeof(deviceDat), nullptr) != KK_OK)
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
        hasLeafAABBs = true;
        isComplete = true;
        if (!logPath.empty())
            writeLog(inputMesh, logPath);
    }

    // Several GPUs: the tree built on rank `root` (BuildFrom there; the other ranks call nothing before)
    // reaches every rank of `comm` in two collectives - the sizes, then the tree's one allocation - and
    // every rank's DeviceData points at its own copy. Not part of the reference.
    void Broadcast(const HIP::Communicator &comm, int root = 0) {
        const bool sender = comm.GetRank() == root;
        if (sender && !isComplete) {
            std::cerr << ErrTag << " Broadcast of an incomplete LBVH.\n";
            return;
        }
        uint32_t sizes[2] = {faceNum, vertNum};
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_comm_broadcast_host(comm.Get(), sizes, sizeof(sizes), root, nullptr)))
            return;
        if (!sender) {
            isComplete = false;
            clear();
            faceNum = sizes[0], vertNum = sizes[1];
            if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_lbvh_blob_layout_of(faceNum, vertNum, &blob)) ||
                !d_blob.resize(blob.total)) {
                std::cerr << ErrTag << " " << kk_last_error() << "\n";
                return;
            }
        }
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_lbvh_broadcast(comm.Get(), d_blob.data(), faceNum, vertNum, root, nullptr)))
            return;
        if (kk_memcpy_d2h(&deviceDat, piece<DeviceData>(blob.device_data), sizeof(deviceDat), nullptr) != KK_OK)
            return;
        isComplete = true;
    }

    bool IsComplete() const { return isComplete; }
    const DeviceData *GetDeivceDataDevicePtr() const { return piece<DeviceData>(blob.device_data); }
    const DeviceData &GetDeviceData() const { return deviceDat; }

    // Device copies of the mesh arrays the tree indexes; the renderer shares them.
    const float *GetDevicePositions() const { return piece<float>(blob.positions); }
    const IndexTy *GetDeviceFacePositionIndices() const { return piece<IndexTy>(blob.face_pos_idx); }
    // float[compactedFaceNum][6]: the boxes of the leaves, left in the workspace by the build (a tree
    // that was received from another rank has none).
    const float *GetDeviceLeafAABBs() const {
        return isComplete && hasLeafAABBs ? reinterpret_cast<const float *>(d_workspace.data() + leafAABBsOffs)
                                          : nullptr;
    }

  private:
    void clear() {
        d_blob.clear();
        d_workspace.clear();
        hasLeafAABBs = false;
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
        std::vector<IndexTy> compacts(deviceDat.compactedFaceNum), faces(deviceDat.faceNum);
        kk_memcpy_d2h(compacts.data(), piece<IndexTy>(blob.compacted), 4ull * compacts.size(), nullptr);
        kk_memcpy_d2h(faces.data(), piece<IndexTy>(blob.faces), 4ull * faces.size(), nullptr);
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
        std::vector<InternalNode> inodes(deviceDat.compactedFaceNum > 1 ? deviceDat.compactedFaceNum - 1 : 0);
        if (!inodes.empty())
            kk_memcpy_d2h(inodes.data(), piece<InternalNode>(blob.nodes), sizeof(InternalNode) * inodes.size(), nullptr);
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
