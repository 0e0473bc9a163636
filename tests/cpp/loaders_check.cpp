// Runs the header-only host loaders (include/kouek/data/*.h) and writes what they read as raw arrays,
// for tests/test_cpp_loaders.py to compare with the Python loaders and the committed scene arrays.
// No GPU call is made.
//   loaders_check <scene.obj> <volume.raw> <dx> <dy> <dz> <tx> <ty> <tz> <tf.txt> <out_dir>
#include <kouek/data/mesh.h>
#include <kouek/data/raw_volume.h>
#include <kouek/data/transfer_function.h>
#include <kouek/ray_tracer/material.h>

#include <cstdio>
#include <fstream>
#include <string>

using namespace kouek;

template <typename T> static void save(const std::string &path, const T *p, size_t n) {
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char *>(p), static_cast<std::streamsize>(n * sizeof(T)));
}
template <typename T> static void save(const std::string &path, const std::vector<T> &v) { save(path, v.data(), v.size()); }

int main(int argc, char **argv) {
    if (argc != 11)
        return 2;
    const std::string out = argv[10];
    Data::SceneConfig cfg(argv[1]);
    if (!cfg.isComplete)
        return 3;
    Data::OBJMesh mesh(argv[1], cfg.lights);
    if (!mesh.IsComplete())
        return 4;
    save(out + "/positions.f32", mesh.GetPositions());
    save(out + "/normals.f32", mesh.GetNormals());
    save(out + "/tex_coords.f32", mesh.GetTextureCoordinates());
    save(out + "/fpi.u32", mesh.GetFacePositionIndices());
    save(out + "/fni.u32", mesh.GetFaceNormalIndices());
    save(out + "/fti.u32", mesh.GetFaceTextureCoordinateIndices());
    save(out + "/groups.u32", mesh.GetGroupStartFaceIndices());
    // materials and lights as the reference's application assembles them (app/rt_lbvh.cpp:186-226)
    std::vector<RayTracer::Light> lights;
    for (auto &l : mesh.GetLights())
        lights.push_back(l.type == Data::OBJMesh::Light::Type::Quad
                             ? RayTracer::Light::CreateQuad(l.quad.o, l.quad.u, l.quad.v, l.radiance)
                             : RayTracer::Light::CreateSphere(l.sphere.r, l.sphere.o, l.radiance));
    std::vector<RayTracer::Material> materials(mesh.GetGroupToMaterialNames().size());
    for (auto &[gi, name] : mesh.GetGroupToMaterialNames()) {
        if (mesh.GetNameToMaterials().find(name) == mesh.GetNameToMaterials().end() || gi >= materials.size())
            return 8; // a group's material that the library does not define
        const auto &m = mesh.GetNameToMaterials().at(name);
        auto itr = cfg.lights.find(name);
        materials[gi] = itr == cfg.lights.end()
                            ? RayTracer::Material::Create(m.kd, m.ks, m.tr, m.ni, m.ns)
                            : RayTracer::Material::Create(m.kd, m.ks, m.tr, m.ni, m.ns, itr->second.radiance);
    }
    save(out + "/lights.bin", lights);
    save(out + "/materials.bin", materials);
    const float cam[11] = {cfg.eye[0],  cfg.eye[1], cfg.eye[2], cfg.cntr[0], cfg.cntr[1], cfg.cntr[2],
                           cfg.up[0],   cfg.up[1],  cfg.up[2],  static_cast<float>(cfg.rndrSz.x),
                           static_cast<float>(cfg.rndrSz.y)};
    save(out + "/camera.f32", cam, 11);

    const uvec3 dim(std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
    const std::array<int8_t, 3> tr = {static_cast<int8_t>(std::atoi(argv[6])), static_cast<int8_t>(std::atoi(argv[7])),
                                      static_cast<int8_t>(std::atoi(argv[8]))};
    Data::RAWVolume<uint8_t> vol(argv[2], dim, tr);
    if (!vol.IsComplete())
        return 5;
    save(out + "/volume.u8", vol.Get());
    const uint32_t vd[3] = {vol.GetDimension()[0], vol.GetDimension()[1], vol.GetDimension()[2]};
    save(out + "/volume_dim.u32", vd, 3);

    Data::TransferFunction<uint8_t, 0, 255> tf(argv[9]);
    if (!tf.IsComplete())
        return 6;
    save(out + "/tf.f32", tf.Get());
    save(out + "/tf_empty.u8", tf.GetEmptyScalarRanges());

    auto grid = Data::OBJMesh::CreateGridTestMesh(uvec3(3, 5, 7), vec3(1.f, 1.f, 1.f));
    save(out + "/grid_positions.f32", grid.GetPositions());
    // a malformed file is refused like the reference refuses it
    Data::RAWVolume<uint8_t> bad(argv[2], dim, {1, 1, 3});
    return bad.IsComplete() ? 7 : 0;
}
