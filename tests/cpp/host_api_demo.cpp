// Exercises the C++ host classes the way the reference's two applications drive theirs
// (app/rt_lbvh.cpp:163-296, app/dvr_dpbx_vdb.cpp:271-381), headless, and dumps the results as raw
// binary files for tests/test_gpu_cpp_host.py to compare with the Python path and the CPU checker.
//   host_api_demo <in_dir> <out_dir>
// in_dir holds the scene as OBJ / MTL / XML text (scene.obj, ...), a RAW volume and a transfer-function
// file, read with the host loaders of include/kouek/data/ like the applications read theirs, plus the
// camera matrices as little-endian raw arrays written by the test.
#include <kouek/data/mesh.h>
#include <kouek/data/raw_volume.h>
#include <kouek/data/transfer_function.h>
#include <kouek/ray_caster/ray_caster.h>
#include <kouek/ray_tracer/ray_tracer.h>

#include <cstdio>
#include <fstream>
#include <string>

using namespace kouek;

template <typename T> static std::vector<T> load(const std::string &path) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in)
        return {};
    size_t bytes = static_cast<size_t>(in.tellg());
    std::vector<T> v(bytes / sizeof(T));
    in.seekg(0);
    in.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}
template <typename T> static void save(const std::string &path, const std::vector<T> &v) {
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char *>(v.data()), v.size() * sizeof(T));
}

int main(int argc, char **argv) {
    if (argc < 3)
        return 2;
    const std::string in = argv[1], out = argv[2];
    if (kk_set_device(0) != KK_OK) {
        std::fprintf(stderr, "%s\n", kk_last_error());
        return 1;
    }
    // ---------------- path 1: LBVH + path tracer ----------------
    // mesh, materials and lights as app/rt_lbvh.cpp:160-226 of the reference assembles them
    Data::SceneConfig scnCfg(in + "/scene.obj");
    if (!scnCfg.isComplete)
        return 1;
    Data::OBJMesh mesh(in + "/scene.obj", scnCfg.lights);
    if (!mesh.IsComplete())
        return 1;
    const auto &positions = mesh.GetPositions();
    const auto &normals = mesh.GetNormals();
    const auto &texCoords = mesh.GetTextureCoordinates();
    const auto &fpi = mesh.GetFacePositionIndices();
    const auto &fni = mesh.GetFaceNormalIndices();
    const auto &fti = mesh.GetFaceTextureCoordinateIndices();
    const auto &groups = mesh.GetGroupStartFaceIndices();
    std::vector<RayTracer::Light> lights;
    for (auto &l : mesh.GetLights())
        lights.push_back(l.type == Data::OBJMesh::Light::Type::Quad
                             ? RayTracer::Light::CreateQuad(l.quad.o, l.quad.u, l.quad.v, l.radiance)
                             : RayTracer::Light::CreateSphere(l.sphere.r, l.sphere.o, l.radiance));
    std::vector<RayTracer::Material> materials(mesh.GetGroupToMaterialNames().size());
    for (auto &[gi, name] : mesh.GetGroupToMaterialNames()) {
        const auto &m = mesh.GetNameToMaterials().at(name);
        auto itr = scnCfg.lights.find(name);
        materials[gi] = itr == scnCfg.lights.end()
                            ? RayTracer::Material::Create(m.kd, m.ks, m.tr, m.ni, m.ns)
                            : RayTracer::Material::Create(m.kd, m.ks, m.tr, m.ni, m.ns, itr->second.radiance);
    }
    auto cam = load<float>(in + "/camera.f32"); // proj 16, eyePos 3, eyeRot 9, W, H, spp
    auto lbvh = std::make_shared<RayTracer::LBVH>();
    lbvh->BuildFrom({positions, fpi}, out + "/lbvh.log");
    if (!lbvh->IsComplete())
        return 1;
    {
        auto dd = lbvh->GetDeviceData();
        std::vector<uint32_t> hdr = {dd.faceNum, dd.compactedFaceNum};
        save(out + "/lbvh_hdr.u32", hdr);
        std::vector<RayTracer::LBVH::InternalNode> nodes(dd.compactedFaceNum - 1);
        kk_memcpy_d2h(nodes.data(), dd.internalNodes, nodes.size() * 32, nullptr);
        save(out + "/lbvh_nodes.bin", nodes);
    }
    RayTracer::RayTracer tracer;
    tracer.SetWorldToScene(mat4::identity());
    tracer.SetMesh({positions, normals, texCoords, groups, fpi, fni, fti, lights, materials});
    tracer.SetLBVH(lbvh);
    tracer.SetMaxPathDepth(4);
    mat4 proj;
    std::memcpy(static_cast<void *>(&proj), cam.data(), 64);
    vec3 eye(cam[16], cam[17], cam[18]);
    mat3 rot;
    std::memcpy(static_cast<void *>(&rot), cam.data() + 19, 36);
    ivec2 sz{static_cast<int>(cam[28]), static_cast<int>(cam[29])};
    tracer.SetProjection(proj);
    tracer.SetCameraToWorld(eye, rot);
    HIP::FrameBuffer fb(sz);
    // the first sample as the reference's loop renders it, the others in one call
    tracer.Render(fb, sz);
    if (static_cast<int>(cam[30]) > 1)
        tracer.Render(fb, sz, RayTracer::RayTracer::RenderTarget::Scene, static_cast<int>(cam[30]) - 1);
    save(out + "/pt_accum.f32", tracer.ReadAccumulation());
    std::vector<HIP::uchar4> img;
    fb.ReadBack(img);
    save(out + "/pt_rgba.u8", img);
    std::vector<float> spp = {tracer.GetCurrentAccumulatedSamplePerPixel()};
    save(out + "/pt_spp.f32", spp);

    // ---------------- path 2: Depth-Box VDB + DVR ----------------
    auto vdim = load<uint32_t>(in + "/volume_dim.u32"); // dx dy dz, log2Dims[3]
    Data::RAWVolume<uint8_t> rawVolume(in + "/volume.raw", uvec3(vdim[0], vdim[1], vdim[2]));
    Data::TransferFunction<uint8_t, 0, 255> tfFile(in + "/tf.txt");
    if (!rawVolume.IsComplete() || !tfFile.IsComplete())
        return 1;
    const auto &vol = rawVolume.Get();
    const auto &rngs = tfFile.GetEmptyScalarRanges();
    const auto &tf = tfFile.Get();
    auto dcam = load<float>(in + "/dvr_camera.f32"); // w2s 16, proj 16, eyePos 3, eyeRot 9, W, H
    HIP::Volume<uint8_t> volume(vol, vdim[0], vdim[1], vdim[2]);
    auto vdb = std::make_shared<RayCaster::DepthBoxVDB>();
    vdb->BuildFrom<uint8_t>({.apronWid = 1,
                             .log2Dims = {uint8_t(vdim[3]), uint8_t(vdim[4]), uint8_t(vdim[5])},
                             .useDepthBox = true,
                             .volTex = volume,
                             .emptyScalarRngs = rngs});
    if (!vdb->IsComplete())
        return 1;
    RayCaster::RayCaster caster;
    mat4 w2s, dproj;
    std::memcpy(static_cast<void *>(&w2s), dcam.data(), 64);
    std::memcpy(static_cast<void *>(&dproj), dcam.data() + 16, 64);
    vec3 deye(dcam[32], dcam[33], dcam[34]);
    mat3 drot;
    std::memcpy(static_cast<void *>(&drot), dcam.data() + 35, 36);
    ivec2 dsz{static_cast<int>(dcam[44]), static_cast<int>(dcam[45])};
    caster.SetWorldToScene(w2s);
    caster.SetDepthBoxVDB(vdb);
    caster.SetTransferFunctionTexture(std::make_shared<HIP::TransferFunctionTable>(tf));
    caster.SetProjection(dproj);
    caster.SetCameraToWorld(deye, drot);
    HIP::FrameBuffer dfb(dsz);
    caster.RenderDepthBoxVDB<uint8_t>(dfb, dsz);
    kk_stream_synchronize(nullptr);
    dfb.ReadBack(img);
    save(out + "/dvr_rgba.u8", img);
    { // the same frame on a stream of the application's own (what two frames in flight use)
        HIP::Stream own;
        HIP::FrameBuffer dfb2(dsz);
        caster.RenderDepthBoxVDB<uint8_t>(dfb2, dsz, RayCaster::RayCaster::DPBXVDBRenderTarget::Scene, &own);
        own.Synchronize();
        dfb2.ReadBack(img);
        save(out + "/dvr_rgba_own_stream.u8", img);
    }
    { // and with the tiles started in the order of the frame before: the second such frame is ordered
        caster.SetAdaptiveTileOrder(true);
        HIP::FrameBuffer dfb3(dsz);
        caster.RenderDepthBoxVDB<uint8_t>(dfb3, dsz);
        caster.RenderDepthBoxVDB<uint8_t>(dfb3, dsz);
        kk_stream_synchronize(nullptr);
        dfb3.ReadBack(img);
        save(out + "/dvr_rgba_adaptive_order.u8", img);
        caster.SetAdaptiveTileOrder(false);
    }
    // the several-GPU calls with a one-rank communicator: both structures broadcast (each a single
    // allocation), the frame rendered as compact stripes, gathered and put back in image order
    {
        HIP::Communicator::Id id;
        if (HIP::Communicator::MakeId(id)) {
            HIP::Communicator comm(1, 0, id);
            if (!comm.IsComplete())
                return 1;
            lbvh->Broadcast(comm, 0);
            vdb->Broadcast(comm, 0);
            if (!lbvh->IsComplete() || !vdb->IsComplete())
                return 1;
            HIP::FrameBuffer gfb(dsz);
            caster.RenderDepthBoxVDBGathered<uint8_t>(gfb, dsz, comm, 0);
            gfb.ReadBack(img);
            save(out + "/dvr_gathered_rgba.u8", img);
            caster.SetAdaptiveTileOrder(true); // twice: the second frame's tiles start in the first's order
            caster.RenderDepthBoxVDBGathered<uint8_t>(gfb, dsz, comm, 0);
            caster.RenderDepthBoxVDBGathered<uint8_t>(gfb, dsz, comm, 0);
            caster.SetAdaptiveTileOrder(false);
            gfb.ReadBack(img);
            save(out + "/dvr_gathered_adaptive_rgba.u8", img);
            // the tree is still the tree after its broadcast: one more sample on top
            tracer.Render(fb, sz);
            save(out + "/pt_accum_after_broadcast.f32", tracer.ReadAccumulation());
            // the path-traced frame as a shared frame: the scene travels like the tree, every Render call
            // is rendered as this rank's compact stripes, gathered and put back in image order
            tracer.Broadcast(comm, 0);
            tracer.SetLBVH(lbvh);
            HIP::FrameBuffer pfb(sz);
            tracer.RenderGathered(pfb, sz, comm, 0);
            if (static_cast<int>(cam[30]) > 1)
                tracer.RenderGathered(pfb, sz, comm, 0, static_cast<int>(cam[30]) - 1);
            pfb.ReadBack(img);
            save(out + "/pt_gathered_rgba.u8", img);
            std::vector<float> gspp = {tracer.GetCurrentAccumulatedSamplePerPixel()};
            save(out + "/pt_gathered_spp.f32", gspp);
        } else
            std::fprintf(stderr, "no RCCL on this machine: %s\n", kk_last_error());
    }
    // the same view without empty-space skipping
    auto rawVol = std::make_shared<HIP::Volume<uint8_t>>(vol, vdim[0], vdim[1], vdim[2]);
    caster.SetRAWVolume(rawVol);
    caster.RenderRAWVolume(dfb, dsz);
    kk_stream_synchronize(nullptr);
    dfb.ReadBack(img);
    save(out + "/raw_rgba.u8", img);
    caster.RenderRAWVolume(dfb, dsz, RayCaster::RayCaster::RAWVolumeRenderTarget::AABB);
    kk_stream_synchronize(nullptr);
    dfb.ReadBack(img);
    save(out + "/raw_aabb_rgba.u8", img);
    // the inspection targets of the VDB marcher
    caster.SetDepthBoxVDBDisplayLevel(0);
    caster.RenderDepthBoxVDB<uint8_t>(dfb, dsz, RayCaster::RayCaster::DPBXVDBRenderTarget::AABBs);
    kk_stream_synchronize(nullptr);
    dfb.ReadBack(img);
    save(out + "/vdb_aabbs.u8", img);
    caster.SetDepthBoxVDBDisplayLeafLayer(1);
    caster.RenderDepthBoxVDB<uint8_t>(dfb, dsz, RayCaster::RayCaster::DPBXVDBRenderTarget::PositionsAfterDepthSkip);
    kk_stream_synchronize(nullptr);
    dfb.ReadBack(img);
    save(out + "/vdb_positions.u8", img);
    // and two inspection targets of the tracer
    tracer.SetDisplayTreeHeight(5);
    tracer.Render(fb, sz, RayTracer::RayTracer::RenderTarget::AABBs);
    fb.ReadBack(img);
    save(out + "/rt_aabbs.u8", img);
    tracer.Render(fb, sz, RayTracer::RayTracer::RenderTarget::Normals);
    fb.ReadBack(img);
    save(out + "/rt_normals.u8", img);
    std::vector<uint64_t> sizes = {vdb->GetNodePoolsSize(), vdb->GetChildPoolsSize(),
                                   vdb->GetAtlasBrickToNodeIndicesSize()};
    save(out + "/vdb_sizes.u64", sizes);
    std::printf("host_api_demo ok\n");
    return 0;
}
