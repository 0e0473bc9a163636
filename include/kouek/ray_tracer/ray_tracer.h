// kouek::RayTracer::RayTracer with the reference's public interface (reference
// include/ray_tracer/ray_tracer.h:23-136, src/ray_tracer/ray_tracer.cu, ray_tracer_render.cu:433-647).
// Changes forced by the platform: Render takes a HIP::FrameBuffer instead of a cudaSurfaceObject_t, and
// only RenderTarget::Scene is implemented (the other targets are debug views).
#ifndef KOUEK_RAY_TRACER_RAY_TRACER_H
#define KOUEK_RAY_TRACER_RAY_TRACER_H

#include <memory>
#include <string_view>
#include <vector>

#include "lbvh.h"
#include "material.h"

namespace kouek {
namespace RayTracer {

class RayTracer : Noncopyable {
  public:
    using IndexTy = uint32_t;
    static constexpr float Eps = .001f;

    struct alignas(16) Triangle {
        uvec3 normIdx;
        uvec3 texCoordIdx;
        IndexTy grpIdx;
    };
    static_assert(sizeof(Triangle) == sizeof(kk_pt_triangle), "layout must match the C ABI");

    enum class RenderTarget { Scene, AABBs, Triangles, Lights, Positions, Normals, TextureCoords, None };

  private:
    float prevMaxSamplePerPixel = 1024.f;
    float maxSamplePerPixel = 1024.f;
    float currAccuSamplePerPixel = 0.f;
    const HIP::FrameBuffer *prevRndrTo = nullptr;
    ivec2 prevRndrSz{0, 0};
    RenderTarget prevRndrTarget = RenderTarget::None;

    kk_pt_scene scene{};
    kk_pt_frame frame{};
    bool modified = true;
    std::shared_ptr<LBVH> lbvh;

    // The arrays SetMesh uploads (the reference's separate thrust vectors, ray_tracer.h:86-99) are pieces
    // of one allocation (kk_pt_scene_blob_layout_of), so that Broadcast() is one collective.
    HIP::DeviceBuffer<uint8_t> d_sceneBlob;
    kk_pt_scene_sizes sceneSizes{};
    kk_pt_scene_blob_layout sceneBlob{};
    HIP::DeviceBuffer<vec3> d_accuRndrTo;
    // several GPUs: this rank's stripes (RGBA8 block that is sent, compact accumulation), root's receive blocks
    HIP::DeviceBuffer<uint8_t> stripeSend, stripeBlocks;
    HIP::DeviceBuffer<vec3> d_accuStripes;
    struct StripeKey { // which shared frame the accumulation in d_accuStripes belongs to
        int w = 0, h = 0, rank = -1, ranks = 0;
        bool operator!=(const StripeKey &o) const { return w != o.w || h != o.h || rank != o.rank || ranks != o.ranks; }
    } prevStripeKey;
    HIP::DeviceBuffer<uint8_t> d_traceTree; // the 8-wide tree the path tracer walks

    static constexpr std::string_view ErrTag = "[RayTracer Error]";
    static constexpr uint32_t Seed = 1234; // curand_init(1234, pixel, 0) in the reference

    template <typename T> void setIfChanged(T &dst, const T &src) {
        if (std::memcmp(&dst, &src, sizeof(T)) != 0) {
            dst = src;
            modified = true;
        }
    }
    void setFloats(float *dst, const float *src, int n) {
        if (std::memcmp(dst, src, sizeof(float) * n) != 0) {
            std::memcpy(dst, src, sizeof(float) * n);
            modified = true;
        }
    }

  public:
    RayTracer() { scene.max_path_depth = 4; }

    struct InputMesh {
        const std::vector<vec3> &positions;
        const std::vector<vec3> &normals;
        const std::vector<vec2> &texCoords;
        const std::vector<IndexTy> &groupStartFaceIndices;
        const std::vector<uvec3> &facePositionIndices;
        const std::vector<uvec3> &faceNormalIndices;
        const std::vector<uvec3> &faceTexCoordIndices;
        const std::vector<Light> &lights;
        const std::vector<Material> &materials;
    };

    void SetMesh(const InputMesh &in) {
        // The kernels follow the mesh's indices into its arrays, and one past an array is a GPU memory fault:
        // a mesh whose indices do not fit its arrays is refused here. (The reference uploads it as it is.)
        if (in.faceNormalIndices.size() != in.facePositionIndices.size() ||
            in.faceTexCoordIndices.size() != in.facePositionIndices.size()) {
            std::cerr << ErrTag << " SetMesh: the three face index arrays differ in length.\n";
            return;
        }
        for (size_t fi = 0; fi < in.facePositionIndices.size(); ++fi)
            for (int i = 0; i < 3; ++i)
                if (in.facePositionIndices[fi][i] >= in.positions.size() ||
                    in.faceNormalIndices[fi][i] >= in.normals.size() ||
                    (!in.texCoords.empty() && in.faceTexCoordIndices[fi][i] >= in.texCoords.size())) {
                    std::cerr << ErrTag << " SetMesh: face " << fi
                              << " refers to a position, normal or texture coordinate the mesh does not have.\n";
                    return;
                }
        if (in.materials.size() < (in.groupStartFaceIndices.empty() ? 1u : in.groupStartFaceIndices.size())) {
            std::cerr << ErrTag << " SetMesh: fewer materials than groups.\n";
            return;
        }
        std::vector<Triangle> tris(in.facePositionIndices.size());
        for (size_t fi = 0; fi < tris.size(); ++fi) {
            tris[fi].normIdx = in.faceNormalIndices[fi];
            tris[fi].texCoordIdx = in.faceTexCoordIndices[fi];
            // group of a face: the last group that starts at or before it (ray_tracer.cu:43-57)
            IndexTy gi;
            for (gi = 1; gi < in.groupStartFaceIndices.size(); ++gi)
                if (in.groupStartFaceIndices[gi] > fi)
                    break;
            tris[fi].grpIdx = gi - 1;
        }
        float tot = 0.f;
        for (auto &lht : in.lights)
            tot += lht.area;
        kk_pt_scene_sizes z{};
        z.face_num = static_cast<uint32_t>(tris.size());
        z.vert_num = static_cast<uint32_t>(in.positions.size());
        z.normal_num = static_cast<uint32_t>(in.normals.size());
        z.tex_coord_num = static_cast<uint32_t>(in.texCoords.size());
        z.light_num = static_cast<uint32_t>(in.lights.size());
        z.material_num = static_cast<uint32_t>(in.materials.size());
        z.light_tot_area = tot;
        if (!allocScene(z))
            return;
        static_assert(sizeof(Light) == sizeof(kk_light) && sizeof(Material) == sizeof(kk_material), "layout");
        auto up = [&](uint64_t offs, const void *src, size_t bytes) {
            return bytes == 0 || kk_memcpy_h2d(d_sceneBlob.data() + offs, src, bytes, nullptr) == KK_OK;
        };
        if (!(up(sceneBlob.triangles, tris.data(), sizeof(Triangle) * tris.size()) &&
              up(sceneBlob.tri_to_position_indices, in.facePositionIndices.data(), 12ull * z.face_num) &&
              up(sceneBlob.positions, in.positions.data(), 12ull * z.vert_num) &&
              up(sceneBlob.normals, in.normals.data(), 12ull * z.normal_num) &&
              up(sceneBlob.tex_coords, in.texCoords.data(), 8ull * z.tex_coord_num) &&
              up(sceneBlob.lights, in.lights.data(), sizeof(Light) * in.lights.size()) &&
              up(sceneBlob.materials, in.materials.data(), sizeof(Material) * in.materials.size())))
            std::cerr << ErrTag << " " << kk_last_error() << "\n";
    }
    // Several GPUs (not part of the reference): what SetMesh uploaded on rank `root` reaches every rank of
    // `comm` in two collectives - the sizes, then the scene's one allocation. The other ranks call no
    // SetMesh; the tree travels by LBVH::Broadcast, and SetLBVH is called on every rank afterwards.
    void Broadcast(const HIP::Communicator &comm, int root = 0) {
        kk_pt_scene_sizes z = sceneSizes;
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_comm_broadcast_host(comm.Get(), &z, sizeof(z), root, nullptr)))
            return;
        if (comm.GetRank() != root && !allocScene(z))
            return;
        KOUEK_HIP_CHECK(ErrTag.data(), kk_pt_scene_broadcast(comm.Get(), d_sceneBlob.data(), &sceneSizes, root, nullptr));
        kk_stream_synchronize(nullptr);
    }
    // Also builds the trace-side tree of `lbvh` (kk_trace_accel_build_with_leaf_boxes); the image is the same with
    // and without it, it only makes the hits cheaper to find.
    void SetLBVH(std::shared_ptr<LBVH> lbvh) {
        this->lbvh = lbvh;
        scene.trace_accel = nullptr;
        if (!lbvh || !lbvh->IsComplete())
            return;
        const auto &dd = lbvh->GetDeviceData();
        const size_t bytes = kk_trace_accel_bytes(dd.faceNum, dd.compactedFaceNum);
        if (!d_traceTree.resize(bytes))
            return;
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_trace_accel_build_with_leaf_boxes(
                                 reinterpret_cast<const kk_lbvh_device_data *>(lbvh->GetDeivceDataDevicePtr()),
                                 lbvh->GetDevicePositions(), lbvh->GetDeviceFacePositionIndices(), dd.faceNum,
                                 dd.compactedFaceNum, lbvh->GetDeviceLeafAABBs(), d_traceTree.data(), bytes,
                                 nullptr)))
            return;
        scene.trace_accel = d_traceTree.data();
        scene.accel_face_num = dd.faceNum;
        scene.accel_compacted_face_num = dd.compactedFaceNum;
    }
    void SetWorldToScene(const mat4 &w2s) { setFloats(scene.w2s, &w2s[0][0], 16); }
    void SetCameraToWorld(const vec3 &eyePos2w, const mat3 &eyeRot2w) {
        setFloats(frame.eye_pos, &eyePos2w[0], 3);
        setFloats(frame.eye_rot, &eyeRot2w[0][0], 9);
    }
    void SetProjection(const mat4 &proj) {
        auto invProj = Math::InverseProjective(proj);
        setFloats(scene.inv_proj, &invProj[0][0], 16);
    }
    void SetDisplayTreeHeight(uint8_t h) { setIfChanged(scene.display_tree_height, h); }
    void SetMaxPathDepth(uint8_t d) { setIfChanged(scene.max_path_depth, d); }
    void SetMaxSamplePerPixel(const float &v) { maxSamplePerPixel = v; }
    const float GetMaxSamplePerPixel() const { return maxSamplePerPixel; }
    const float GetCurrentAccumulatedSamplePerPixel() const { return currAccuSamplePerPixel; }

    // One call adds one sample per pixel; accumulation restarts when a parameter, the size, the target
    // or the framebuffer changed, and the call is a no-op once the sample budget is used up
    // (ray_tracer_render.cu:561-577, 640-641).
    // `samples` > 1 (not in the reference) does the work of that many consecutive calls in one launch: the
    // same image bit for bit, never past the sample budget.
    void Render(HIP::FrameBuffer &rndrTo, const ivec2 &rndrSz, RenderTarget rndrTarget = RenderTarget::Scene,
                int samples = 1) {
        if (!lbvh || !lbvh->IsComplete()) {
            std::cerr << ErrTag << " at " << __FILE__ << ":" << __LINE__ << ". lbvh is not set.\n";
            return;
        }
        if (rndrTarget == RenderTarget::None)
            return;
        if (rndrTarget != RenderTarget::Scene) {
            // inspection targets: one frame, no accumulation (ray_tracer_render.cu:461-558)
            modified = false;
            static constexpr int Targets[] = {0,
                                              KK_RT_TARGET_AABBS,
                                              KK_RT_TARGET_TRIANGLES,
                                              KK_RT_TARGET_LIGHTS,
                                              KK_RT_TARGET_POSITIONS,
                                              KK_RT_TARGET_NORMALS,
                                              KK_RT_TARGET_TEXTURE_COORDS};
            KOUEK_HIP_CHECK(ErrTag.data(),
                            kk_rt_render_target(reinterpret_cast<const kk_lbvh_device_data *>(
                                                    lbvh->GetDeivceDataDevicePtr()),
                                                &scene, &frame, Targets[static_cast<int>(rndrTarget)],
                                                rndrTo.Get(), rndrSz.x, rndrSz.y, 0, rndrSz.y, nullptr));
            prevRndrTo = &rndrTo;
            prevRndrSz = rndrSz;
            prevRndrTarget = rndrTarget;
            return;
        }
        bool needRedraw = modified;
        modified = false;
        auto flatRes = static_cast<size_t>(rndrSz.x) * rndrSz.y;
        if (prevRndrSz != rndrSz || d_accuRndrTo.size() != flatRes)
            d_accuRndrTo.resize(flatRes);
        needRedraw |= prevMaxSamplePerPixel != maxSamplePerPixel || prevRndrTo != &rndrTo ||
                      prevRndrSz != rndrSz || prevRndrTarget != rndrTarget;
        if (needRedraw) {
            currAccuSamplePerPixel = 1.f;
            d_accuRndrTo.zero();
        }
        prevRndrTo = &rndrTo;
        prevRndrSz = rndrSz;
        prevRndrTarget = rndrTarget;
        if (currAccuSamplePerPixel > maxSamplePerPixel)
            return;
        int n = samples < 1 ? 1 : samples;
        const float left = maxSamplePerPixel - currAccuSamplePerPixel + 1.f;
        if (static_cast<float>(n) > left)
            n = static_cast<int>(left);
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_pt_render_samples(reinterpret_cast<const kk_lbvh_device_data *>(
                                                      lbvh->GetDeivceDataDevicePtr()),
                                                  &scene, &frame, reinterpret_cast<float *>(d_accuRndrTo.data()),
                                                  rndrTo.Get(), rndrSz.x, rndrSz.y, 0, rndrSz.y,
                                                  currAccuSamplePerPixel, n, Seed, nullptr)))
            return;
        prevMaxSamplePerPixel = maxSamplePerPixel;
        currAccuSamplePerPixel += static_cast<float>(n);
    }

    std::vector<vec3> ReadAccumulation() const { return d_accuRndrTo.to_host(); }

    // Several GPUs (not part of the reference): one Render(Scene) call of a frame shared between the ranks
    // of `comm`. Every rank adds the sample(s) to its round-robin 16-row stripes - straight into the block
    // it sends, accumulated in a compact buffer of the same rows - the blocks are gathered on `root` and
    // put back in image order in `rndrTo` there (other ranks pass any frame buffer of the size; theirs is
    // left alone). A pixel's random stream is keyed by its place in the image, so the frame is the one
    // Render() gives on one GPU, byte for byte. Blocking form of the exchange: one call, waited for.
    void RenderGathered(HIP::FrameBuffer &rndrTo, const ivec2 &rndrSz, const HIP::Communicator &comm, int root = 0,
                        int samples = 1) {
        if (!lbvh || !lbvh->IsComplete()) {
            std::cerr << ErrTag << " at " << __FILE__ << ":" << __LINE__ << ". lbvh is not set.\n";
            return;
        }
        const int n = comm.GetRankNum(), r = comm.GetRank();
        const size_t rows = static_cast<size_t>(kk_stripes_per_rank(rndrSz.y, n)) * 16;
        const size_t blockBytes = rows * rndrSz.x * 4;
        const StripeKey key{rndrSz.x, rndrSz.y, r, n};
        bool needRedraw = modified || prevMaxSamplePerPixel != maxSamplePerPixel || prevStripeKey != key ||
                          prevRndrTarget != RenderTarget::Scene || prevRndrTo != &rndrTo;
        modified = false;
        if (prevStripeKey != key || d_accuStripes.size() != rows * rndrSz.x) {
            if (!d_accuStripes.resize(rows * rndrSz.x) || !stripeSend.resize(blockBytes) || !stripeSend.zero())
                return;
            if (r == root && !stripeBlocks.resize(blockBytes * n))
                return;
        }
        if (needRedraw) {
            currAccuSamplePerPixel = 1.f;
            d_accuStripes.zero();
        }
        prevStripeKey = key;
        prevRndrTo = &rndrTo;
        prevRndrSz = ivec2{0, 0}; // a plain Render() afterwards starts over
        prevRndrTarget = RenderTarget::Scene;
        if (currAccuSamplePerPixel > maxSamplePerPixel)
            return;
        int cnt = samples < 1 ? 1 : samples;
        const float left = maxSamplePerPixel - currAccuSamplePerPixel + 1.f;
        if (static_cast<float>(cnt) > left)
            cnt = static_cast<int>(left);
        if (!KOUEK_HIP_CHECK(ErrTag.data(),
                             kk_pt_render_stripes_compact(
                                 reinterpret_cast<const kk_lbvh_device_data *>(lbvh->GetDeivceDataDevicePtr()), &scene,
                                 &frame, reinterpret_cast<float *>(d_accuStripes.data()), stripeSend.data(), rndrSz.x,
                                 rndrSz.y, r, n, currAccuSamplePerPixel, cnt, Seed, nullptr)))
            return;
        prevMaxSamplePerPixel = maxSamplePerPixel;
        currAccuSamplePerPixel += static_cast<float>(cnt);
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_gather_stripes(comm.Get(), stripeSend.data(), stripeBlocks.data(),
                                                              rndrSz.x, rndrSz.y, root, nullptr)))
            return;
        if (r == root)
            KOUEK_HIP_CHECK(ErrTag.data(),
                            kk_unstripe_rgba(stripeBlocks.data(), n, rndrSz.x, rndrSz.y, rndrTo.Get(), nullptr));
        kk_stream_synchronize(nullptr);
    }

  private:
    bool allocScene(const kk_pt_scene_sizes &z) {
        sceneSizes = z;
        if (!KOUEK_HIP_CHECK(ErrTag.data(), kk_pt_scene_blob_layout_of(&sceneSizes, &sceneBlob)) ||
            !d_sceneBlob.resize(sceneBlob.total) ||
            !KOUEK_HIP_CHECK(ErrTag.data(), kk_pt_scene_blob_bind(d_sceneBlob.data(), &sceneSizes, &scene))) {
            std::cerr << ErrTag << " " << kk_last_error() << "\n";
            return false;
        }
        modified = true;
        return true;
    }
};

} // namespace RayTracer
} // namespace kouek

#endif // KOUEK_RAY_TRACER_RAY_TRACER_H
