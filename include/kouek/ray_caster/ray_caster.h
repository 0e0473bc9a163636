// kouek::RayCaster::RayCaster with the reference's public interface for the Depth-Box VDB path
// (reference include/ray_caster/ray_caster.h:16-89, src/ray_caster/ray_caster.cu,
// src/ray_caster/ray_caster_dpbx_vdb.cu:439-532). Surfaces and textures become HIP::FrameBuffer /
// HIP::TransferFunctionTable; the RAW volume is a HIP::Volume<T> (the reference binds an untyped
// texture, so its SetRAWVolume / RenderRAWVolume take no type; here they are templates on the voxel
// type).
#ifndef KOUEK_RAY_CASTER_RAY_CASTER_H
#define KOUEK_RAY_CASTER_RAY_CASTER_H

#include <memory>
#include <string_view>

#include "dpbx_vdb.h"

namespace kouek {
namespace RayCaster {

class RayCaster : Noncopyable {
  private:
    kk_dvr_params rndrParam{};
    kk_dvr_frame rndrParamPerFrame{};
    std::shared_ptr<HIP::TransferFunctionTable> tfTex;
    std::shared_ptr<DepthBoxVDB> vdb;
    std::shared_ptr<const void> volTex; // HIP::Volume<T>, type-erased
    const void *volDat = nullptr;
    uint32_t volDim[3] = {0, 0, 0};
    size_t volVoxelBytes = 0;
    uint8_t displayLev = 0, displayLeafLayer = 0; // DPBXVDBRenderParameter
    static constexpr std::string_view ErrTag = "[RayCaster Error]";

  public:
    RayCaster() {
        rndrParam.max_alpha = .9f;
        rndrParam.sampling_rate = 3.f;
    }
    template <typename T> void SetRAWVolume(std::shared_ptr<HIP::Volume<T>> vol) {
        static_assert(sizeof(T) == 1 || sizeof(T) == 2, "uint8_t or uint16_t voxels");
        volTex = vol;
        volDat = vol ? vol->Get() : nullptr;
        volVoxelBytes = sizeof(T);
        for (int i = 0; i < 3; ++i)
            volDim[i] = vol ? vol->GetDimension()[i] : 0;
    }
    void SetDepthBoxVDB(std::shared_ptr<DepthBoxVDB> vdb) { this->vdb = vdb; }
    void SetDepthBoxVDBDisplayLevel(uint8_t lev) { displayLev = lev; }
    void SetDepthBoxVDBDisplayLeafLayer(uint8_t layer) { displayLeafLayer = layer; }
    void SetMaximumAlpha(float maxAlpha) { rndrParam.max_alpha = maxAlpha; }
    void SetSamplingRate(float samplingRate) { rndrParam.sampling_rate = samplingRate; }
    void SetTransferFunctionTexture(std::shared_ptr<HIP::TransferFunctionTable> tfTex) { this->tfTex = tfTex; }
    void SetWorldToScene(const mat4 &w2s) { std::memcpy(rndrParam.w2s, &w2s[0][0], 64); }
    void SetCameraToWorld(const vec3 &eyePos2w, const mat3 &eyeRot2w) {
        std::memcpy(rndrParamPerFrame.eye_pos, &eyePos2w[0], 12);
        std::memcpy(rndrParamPerFrame.eye_rot, &eyeRot2w[0][0], 36);
    }
    void SetProjection(const mat4 &proj) {
        auto invProj = Math::InverseProjective(proj);
        std::memcpy(rndrParam.inv_proj, &invProj[0][0], 64);
    }
    struct InputShadingParameter {
        float ka, kd, ks, shiness;
        vec3 lightPosInVolumeSpace;
        vec3 lightCol;
    };
    void SetShading(const InputShadingParameter &in) {
        rndrParam.use_shading = 1;
        rndrParam.ka = in.ka, rndrParam.kd = in.kd, rndrParam.ks = in.ks, rndrParam.shiness = in.shiness;
        std::memcpy(rndrParam.light_pos, &in.lightPosInVolumeSpace[0], 12);
        std::memcpy(rndrParam.light_col, &in.lightCol[0], 12);
    }
    void UnsetShading() { rndrParam.use_shading = 0; }

    enum class RAWVolumeRenderTarget { Scene, AABB };
    // src/ray_caster/ray_caster_raw_volume.cu:147-246
    void RenderRAWVolume(HIP::FrameBuffer &rndrTo, const ivec2 &rndrSz,
                         RAWVolumeRenderTarget rndrTarget = RAWVolumeRenderTarget::Scene) {
        if (!volDat || !tfTex) {
            std::cerr << ErrTag << " at " << __FILE__ << ":" << __LINE__ << ". volTex or tfTex are not set.\n";
            return;
        }
        const int target = rndrTarget == RAWVolumeRenderTarget::Scene ? KK_RAW_TARGET_SCENE : KK_RAW_TARGET_AABB;
        if (volVoxelBytes == 1)
            KOUEK_HIP_CHECK(ErrTag.data(),
                            kk_raw_render_u8(static_cast<const uint8_t *>(volDat), volDim[0], volDim[1],
                                             volDim[2], tfTex->Get(), &rndrParam, &rndrParamPerFrame, target,
                                             rndrTo.Get(), rndrSz.x, rndrSz.y, 0, rndrSz.y, nullptr, nullptr));
        else
            KOUEK_HIP_CHECK(ErrTag.data(),
                            kk_raw_render_u16(static_cast<const uint16_t *>(volDat), volDim[0], volDim[1],
                                              volDim[2], tfTex->Get(), &rndrParam, &rndrParamPerFrame, target,
                                              rndrTo.Get(), rndrSz.x, rndrSz.y, 0, rndrSz.y, nullptr, nullptr));
    }

    enum class DPBXVDBRenderTarget { Scene, AABBs, Depths, PositionsAfterDepthSkip };
    template <typename T>
    void RenderDepthBoxVDB(HIP::FrameBuffer &rndrTo, const ivec2 &rndrSz,
                           DPBXVDBRenderTarget rndrTarget = DPBXVDBRenderTarget::Scene,
                           const HIP::Stream *stream = nullptr) {
        if (!vdb || !tfTex || !vdb->IsComplete()) {
            std::cerr << ErrTag << " at " << __FILE__ << ":" << __LINE__ << ". vdb or tfTex are not set.\n";
            return;
        }
        void *s = stream ? stream->Get() : nullptr;
        if (adaptiveTileOrder && rndrTarget == DPBXVDBRenderTarget::Scene && rndrSz.x > 0 && rndrSz.y > 0) {
            const uint32_t tiles = kk_dvr_tile_count(rndrSz.x, rndrSz.y);
            if (!prepareTileOrder({rndrSz.x, rndrSz.y, 0, 0}, tiles))
                return;
            auto ordered = sizeof(T) == 1 ? kk_dvr_render_ordered_u8 : kk_dvr_render_ordered_u16;
            if (KOUEK_HIP_CHECK(ErrTag.data(),
                                ordered(&vdb->GetDeviceData(), tfTex->Get(), &rndrParam, &rndrParamPerFrame,
                                        rndrTo.Get(), rndrSz.x, rndrSz.y, 0, rndrSz.y, tileCost.data(),
                                        haveTileOrder ? tileOrder.data() : nullptr, s)))
                sortTileOrder(tiles, s);
            return;
        }
        auto fn = sizeof(T) == 1 ? kk_dvr_render_target_u8 : kk_dvr_render_target_u16;
        KOUEK_HIP_CHECK(ErrTag.data(),
                        fn(&vdb->GetDeviceData(), tfTex->Get(), &rndrParam, &rndrParamPerFrame,
                           static_cast<int>(rndrTarget), displayLev, displayLeafLayer, rndrTo.Get(), rndrSz.x,
                           rndrSz.y, 0, rndrSz.y, s));
    }
    // Start each Scene frame's 16 x 16 tiles in the order of what they took in the frame before, the
    // costly ones first: the same image, a shorter frame, since the longest rays no longer start last.
    // Not part of the reference, whose launch order is the tiles' own
    // (src/ray_caster/ray_caster_dpbx_vdb.cu:1027-1034). The state is this object's: frames in flight on
    // several streams take a RayCaster each.
    void SetAdaptiveTileOrder(bool on) {
        adaptiveTileOrder = on;
        haveTileOrder = false;
    }
    // Waits for `stream` and tells whether the sort behind the tile order of the last frame did its work
    // (false, with the error printed: its look-back gave up; the library then keeps the launch order for
    // the next frame, so every tile is still rendered and only the gain in frame time is lost).
    bool CheckTileOrder(const HIP::Stream *stream = nullptr) {
        if (!haveTileOrder)
            return true;
        return KOUEK_HIP_CHECK(ErrTag.data(),
                               kk_dvr_tile_order_status(tileOrderWs.data(), static_cast<uint32_t>(tileCost.size()),
                                                        stream ? stream->Get() : nullptr));
    }

    // Several GPUs (not part of the reference): every rank of `comm` renders its round-robin 16-row
    // stripes of the Scene target straight into the block it sends, the blocks are gathered on rank
    // `root` and put back in image order in `rndrTo` there (other ranks pass any frame buffer of the
    // size; theirs is left alone). Blocking form of the exchange: one frame, waited for.
    template <typename T>
    void RenderDepthBoxVDBGathered(HIP::FrameBuffer &rndrTo, const ivec2 &rndrSz, const HIP::Communicator &comm,
                                   int root = 0) {
        if (!vdb || !tfTex || !vdb->IsComplete()) {
            std::cerr << ErrTag << " at " << __FILE__ << ":" << __LINE__ << ". vdb or tfTex are not set.\n";
            return;
        }
        const int n = comm.GetRankNum(), r = comm.GetRank();
        const size_t blockBytes = static_cast<size_t>(kk_stripes_per_rank(rndrSz.y, n)) * 16 * rndrSz.x * 4;
        if (stripeSend.size() != blockBytes && (!stripeSend.resize(blockBytes) || !stripeSend.zero()))
            return;
        if (r == root && stripeBlocks.size() != blockBytes * n && !stripeBlocks.resize(blockBytes * n))
            return;
        const uint32_t tiles = adaptiveTileOrder ? kk_dvr_stripe_tile_count(rndrSz.x, rndrSz.y, r, n) : 0u;
        if (tiles > 0) { // this rank's tiles in the order of what they took in its frame before
            if (!prepareTileOrder({rndrSz.x, rndrSz.y, r + 1, n}, tiles))
                return;
            auto ordered = sizeof(T) == 1 ? kk_dvr_render_stripes_compact_ordered_u8
                                          : kk_dvr_render_stripes_compact_ordered_u16;
            if (!KOUEK_HIP_CHECK(ErrTag.data(),
                                 ordered(&vdb->GetDeviceData(), tfTex->Get(), &rndrParam, &rndrParamPerFrame,
                                         stripeSend.data(), rndrSz.x, rndrSz.y, r, n, tileCost.data(),
                                         haveTileOrder ? tileOrder.data() : nullptr, nullptr)))
                return;
            sortTileOrder(tiles, nullptr);
        } else {
            auto fn = sizeof(T) == 1 ? kk_dvr_render_stripes_compact_u8 : kk_dvr_render_stripes_compact_u16;
            if (!KOUEK_HIP_CHECK(ErrTag.data(),
                                 fn(&vdb->GetDeviceData(), tfTex->Get(), &rndrParam, &rndrParamPerFrame,
                                    stripeSend.data(), rndrSz.x, rndrSz.y, r, n, nullptr, nullptr)))
                return;
        }
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_gather_stripes(comm.Get(), stripeSend.data(), stripeBlocks.data(),
                                                              rndrSz.x, rndrSz.y, root, nullptr)))
            return;
        if (r == root)
            KOUEK_HIP_CHECK(ErrTag.data(),
                            kk_unstripe_rgba(stripeBlocks.data(), n, rndrSz.x, rndrSz.y, rndrTo.Get(), nullptr));
        kk_stream_synchronize(nullptr);
    }

  private:
    HIP::DeviceBuffer<uint8_t> stripeSend, stripeBlocks, tileOrderWs;
    HIP::DeviceBuffer<uint32_t> tileCost, tileOrder;
    bool adaptiveTileOrder = false, haveTileOrder = false;
    struct TileLaunch { // which launch the tile times belong to: image size and the stripes rendered
        int w = 0, h = 0, first = 0, stride = 0;
        bool operator==(const TileLaunch &o) const {
            return w == o.w && h == o.h && first == o.first && stride == o.stride;
        }
    } tileLaunch;
    // buffers for `tiles` tiles of launch `now`; a launch other than the last one starts in launch order
    bool prepareTileOrder(const TileLaunch &now, uint32_t tiles) {
        if (now == tileLaunch && tiles == tileCost.size())
            return true;
        haveTileOrder = false;
        tileLaunch = now;
        return tileCost.resize(tiles) && tileOrder.resize(tiles) &&
               tileOrderWs.resize(kk_dvr_tile_order_workspace_bytes(tiles));
    }
    void sortTileOrder(uint32_t tiles, void *s) {
        haveTileOrder = KOUEK_HIP_CHECK(ErrTag.data(), kk_dvr_tile_order(tileCost.data(), tiles, tileOrder.data(),
                                                                         tileOrderWs.data(), tileOrderWs.size(), s));
    }
};

} // namespace RayCaster
} // namespace kouek

#endif // KOUEK_RAY_CASTER_RAY_CASTER_H
