// kouek::RayCaster::DepthBoxVDB with the reference's public interface (reference
// include/ray_caster/dpbx_vdb.h:27-143, src/ray_caster/dpbx_vdb.cu:275-637). The point-filtered volume
// texture of InputVolumeAndConfiguration becomes a HIP::Volume<T>.
#ifndef KOUEK_RAY_CASTER_DPBX_VDB_H
#define KOUEK_RAY_CASTER_DPBX_VDB_H

#include <array>
#include <limits>
#include <string_view>
#include <type_traits>
#include <vector>

#include "../hip/device.h"

namespace kouek {
namespace RayCaster {

class DepthBoxVDB : Noncopyable {
  public:
    static constexpr auto MaxLevelNum = uint8_t(3);
    static constexpr auto MaxLog2Dim = uint8_t(10);
    using IndexTy = uint64_t;
    using RelativeIndexTy = uint32_t;
    using CoordValTy = int32_t;
    static constexpr auto UndefRltIdx = std::numeric_limits<RelativeIndexTy>::max();

    using VDBParameter = kk_vdb_param;  // same layout as the reference's struct
    using Node = kk_vdb_node;
    using DeviceData = kk_vdb_device_data;

  private:
    bool isComplete = false;
    // Node pools, child pools, atlas map and atlas live in one allocation (kk_vdb_blob_layout_of), so
    // that Broadcast() is one collective; the reference's separate vectors and CUDA array
    // (dpbx_vdb.h:98-110) are pieces of it.
    HIP::DeviceBuffer<uint8_t> d_blob;
    kk_vdb_plan plan{};
    uint32_t voxelBytes = 0;
    DeviceData deviceDat{};
    HIP::DeviceBuffer<DeviceData> d_deviceDat;
    static constexpr std::string_view ErrTag = "[DepthBoxVDB Error]";

    template <typename T> struct Fn;

  public:
    template <typename T, typename Vec2T = vec2_of<T>> struct InputVolumeAndConfiguration {
        static_assert(std::is_same_v<T, uint8_t> || std::is_same_v<T, uint16_t>,
                      "T as voxel type is not implemented.");
        uint8_t apronWid = 1;
        std::array<uint8_t, MaxLevelNum> log2Dims = {4, 4, 3};
        bool useDepthBox = true;
        const HIP::Volume<T> &volTex;
        const std::vector<Vec2T> &emptyScalarRngs;
    };

    template <typename T, typename Vec2T = vec2_of<T>>
    void BuildFrom(const InputVolumeAndConfiguration<T, Vec2T> &cfg) {
        isComplete = false;
        clear();
        auto fail = [&]() { std::cerr << ErrTag << " " << kk_last_error() << "\n"; };
        const uint32_t *dim = cfg.volTex.GetDimension();
        std::vector<T> rngs;
        for (auto &r : cfg.emptyScalarRngs) {
            rngs.push_back(r[0]);
            rngs.push_back(r[1]);
        }
        const uint32_t nRng = static_cast<uint32_t>(cfg.emptyScalarRngs.size());
        for (auto l : cfg.log2Dims)
            if (!(l > 0 && l <= MaxLog2Dim)) {
                std::cerr << ErrTag << " log2Dim > 0 && log2Dim <= MaxLog2Dim should be true.\n";
                return;
            }
        const uint32_t B = 1u << cfg.log2Dims[0];
        const size_t slots = static_cast<size_t>((dim[0] + B - 1) / B) * ((dim[1] + B - 1) / B) * ((dim[2] + B - 1) / B);
        HIP::DeviceBuffer<uint32_t> d_valid(slots);
        if (Fn<T>::classify(cfg.volTex.Get(), dim[0], dim[1], dim[2], cfg.log2Dims[0], rngs.data(), nRng,
                            d_valid.data(), nullptr) != KK_OK)
            return fail();
        auto valid = d_valid.to_host();
        kk_vdb_plan plan;
        if (kk_vdb_plan_build(dim[0], dim[1], dim[2], cfg.log2Dims.data(), cfg.apronWid, cfg.useDepthBox,
                              valid.data(), &plan) != KK_OK)
            return fail();
        this->plan = plan;
        voxelBytes = sizeof(T);
        if (!allocBlob())
            return fail();
        // the atlas is filled from leaf positions made on the GPU (kk_vdb_leaf_nodes) while the host wires
        // the pools; these go over afterwards
        HIP::DeviceBuffer<Node> d_leafNodes(plan.leaf_num);
        if (kk_vdb_leaf_nodes(d_valid.data(), &plan, reinterpret_cast<kk_vdb_node *>(d_leafNodes.data()), nullptr) !=
                KK_OK ||
            Fn<T>::fill(cfg.volTex.Get(), dim[0], dim[1], dim[2], &plan,
                        reinterpret_cast<const kk_vdb_node *>(d_leafNodes.data()), rngs.data(), nRng,
                        static_cast<T *>(deviceDat.atlas), nullptr) != KK_OK)
            return fail();
        std::vector<Node> nodes(plan.node_num);
        std::vector<RelativeIndexTy> children(plan.child_num), amap(plan.atlas_map_num);
        if (kk_vdb_wire(&plan, valid.data(), nodes.data(), children.data(), amap.data()) != KK_OK)
            return fail();
        if (kk_memcpy_h2d(deviceDat.node_pools, nodes.data(), sizeof(Node) * nodes.size(), nullptr) != KK_OK ||
            kk_memcpy_h2d(deviceDat.child_pools, children.data(), 4 * children.size(), nullptr) != KK_OK ||
            kk_memcpy_h2d(deviceDat.atlas_brick_to_node_indices, amap.data(), 4 * amap.size(), nullptr) != KK_OK)
            return fail();
        kk_stream_synchronize(nullptr);
        // the reference's statistics (src/ray_caster/dpbx_vdb.cu:586-607): sizes of the pools, the atlas in
        // voxels as the reference's 3-D array would hold it (atlasBrickPerVDB bricks per axis), and the
        // memory in use - here the atlas holds the valid bricks only, one after the other
        const auto &pm = plan.param;
        const size_t atlasBytes = static_cast<size_t>(plan.atlas_voxels) * sizeof(T);
        const size_t memUsed = plan.node_num * sizeof(Node) + plan.child_num * sizeof(RelativeIndexTy) +
                               plan.atlas_map_num * sizeof(RelativeIndexTy) + atlasBytes;
        std::cout << "Build VDB with\n\tNode pools of size " << plan.node_num << "x" << sizeof(Node)
                  << "\n\tChild pools of size " << plan.child_num << "x" << sizeof(RelativeIndexTy)
                  << "\n\tMaps from atlas brick to node index of size " << plan.atlas_map_num << "x"
                  << sizeof(RelativeIndexTy) << "\n\tAtlas of size "
                  << pm.atlas_brick_per_vdb[0] * pm.vox_per_atlas_brick << "x"
                  << pm.atlas_brick_per_vdb[1] * pm.vox_per_atlas_brick << "x"
                  << pm.atlas_brick_per_vdb[2] * pm.vox_per_atlas_brick << "x" << sizeof(T) << " (" << plan.leaf_num
                  << " bricks of " << pm.vox_per_atlas_brick << "^3 stored)\n\tTotal use of memory " << memUsed
                  << "\n";
        isComplete = true;
    }

    // Several GPUs: the VDB built on rank `root` reaches every rank of `comm` in two collectives (the
    // 176-byte plan, then the one allocation). Not part of the reference.
    void Broadcast(const HIP::Communicator &comm, int root = 0) {
        const bool sender = comm.GetRank() == root;
        if (sender && !isComplete) {
            std::cerr << ErrTag << " Broadcast of an incomplete VDB.\n";
            return;
        }
        struct {
            kk_vdb_plan plan;
            uint64_t voxelBytes;
        } head{plan, voxelBytes};
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_comm_broadcast_host(comm.Get(), &head, sizeof(head), root, nullptr)))
            return;
        if (!sender) {
            isComplete = false;
            clear();
            plan = head.plan;
            voxelBytes = static_cast<uint32_t>(head.voxelBytes);
            if (!allocBlob()) {
                std::cerr << ErrTag << " " << kk_last_error() << "\n";
                return;
            }
        }
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_vdb_broadcast(comm.Get(), d_blob.data(), &plan, voxelBytes, root, nullptr)) ||
            kk_stream_synchronize(nullptr) != KK_OK)
            return;
        isComplete = true;
    }

    bool IsComplete() const { return isComplete; }
    const DeviceData *GetDeivceDataDevicePtr() const { return d_deviceDat.data(); }
    const DeviceData &GetDeviceData() const { return deviceDat; }
    size_t GetNodePoolsSize() const { return isComplete ? plan.node_num : 0; }
    size_t GetChildPoolsSize() const { return isComplete ? plan.child_num : 0; }
    size_t GetAtlasBrickToNodeIndicesSize() const { return isComplete ? plan.atlas_map_num : 0; }

  private:
    void clear() { d_blob.clear(); }
    // the blob for `plan` / `voxelBytes`, the host DeviceData pointing into it and its device mirror
    bool allocBlob() {
        kk_vdb_blob_layout lay;
        if (kk_vdb_blob_layout_of(&plan, voxelBytes, &lay) != KK_OK || !d_blob.resize(lay.total) ||
            kk_vdb_blob_device_data(d_blob.data(), &plan, voxelBytes, &deviceDat) != KK_OK)
            return false;
        if (!d_deviceDat.size() && !d_deviceDat.resize(1))
            return false;
        return kk_memcpy_h2d(d_deviceDat.data(), &deviceDat, sizeof(deviceDat), nullptr) == KK_OK;
    }
};

template <> struct DepthBoxVDB::Fn<uint8_t> {
    static constexpr auto classify = kk_vdb_classify_u8;
    static constexpr auto fill = kk_vdb_fill_atlas_u8;
    static constexpr auto render = kk_dvr_render_u8;
};
template <> struct DepthBoxVDB::Fn<uint16_t> {
    static constexpr auto classify = kk_vdb_classify_u16;
    static constexpr auto fill = kk_vdb_fill_atlas_u16;
    static constexpr auto render = kk_dvr_render_u16;
};

} // namespace RayCaster
} // namespace kouek

#endif // KOUEK_RAY_CASTER_DPBX_VDB_H
