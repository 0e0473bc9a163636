// kouek::Data::OBJMesh with the reference's interface and rules (reference include/data/mesh.h:23-461):
// a strict OBJ + MTL reader (only the tags g, f, v, vn, vt, mtllib, usemtl and Kd, Ks, Tr, Ni, Ns,
// newmtl; any other line makes the mesh incomplete; faces are `f a/b/c d/e/f g/h/i`), the extraction of
// area lights from the groups whose material the scene names as a light (a quad from a two-triangle
// group, a sphere from the mean vertex and the mean distance to it; the last group is taken one face
// short, as in the reference), and CreateGridTestMesh. Written without <format>; numbers are read with
// the C library (strtof), which gives the bits the reference's sscanf("%f") gives.
// SceneConfig reads the camera and the lights of the scene's XML file (reference app/rt_lbvh.cpp:60-146)
// with a small attribute scanner instead of tinyxml2: the files are flat lists of elements.
#ifndef KOUEK_DATA_MESH_H
#define KOUEK_DATA_MESH_H

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <optional>
#include <sstream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../math_types.h"

namespace kouek {
namespace Data {

namespace detail {
// sscanf("%f" x n) on a string: the values read before the first failure
inline int scanFloats(const char *s, float *out, int n) {
    int got = 0;
    while (got < n) {
        char *end = nullptr;
        const float v = std::strtof(s, &end);
        if (end == s)
            break;
        out[got++] = v;
        s = end;
    }
    return got;
}
inline bool scanUInt(const char *&s, uint32_t &v) {
    while (*s == ' ' || *s == '\t')
        ++s;
    if (*s < '0' || *s > '9')
        return false;
    char *end = nullptr;
    const unsigned long long x = std::strtoull(s, &end, 10);
    s = end;
    v = static_cast<uint32_t>(x);
    return x <= 0xFFFFFFFFull; // (a larger number must not wrap into a valid index)
}
inline float dot3(const vec3 &a, const vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline vec3 sub3(const vec3 &a, const vec3 &b) { return vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
} // namespace detail

class OBJMesh {
  public:
    using IndexTy = uint32_t;
    struct Light {
        enum class Type : uint8_t { Quad, Sphere };
        Type type;
        struct {
            vec3 o, u, v;
        } quad;
        struct {
            vec3 o;
            float r;
        } sphere;
        vec3 radiance;
    };
    struct Material {
        vec3 kd, ks, tr;
        float ni = 0.f, ns = 0.f;
    };
    struct InputLight {
        Light::Type type;
        vec3 radiance;
    };

  private:
    bool isComplete = false;
    std::vector<vec3> positions, normals;
    std::vector<vec2> texCoords;
    std::vector<IndexTy> grpStartFaceIndices;
    std::vector<uvec3> facePositionIndices, faceTexCoordIndices, faceNormalIndices;
    std::unordered_map<IndexTy, std::string> grp2mtlNames;
    std::unordered_map<std::string, Material> name2mtls;
    std::vector<Light> lights;
    std::filesystem::path objPath, mtllibPath;

    static constexpr std::string_view ErrTag = "[OBJMesh Error]";
    OBJMesh() = default;

    bool fail(const std::string &msg) const {
        std::cerr << ErrTag << " " << msg << "\n";
        return false;
    }
    // index of the line's first word in `tags` (count when it is none of them), and where the word ends
    template <size_t N> static size_t parseTag(const std::string &line, const char *const (&tags)[N], size_t &offs) {
        offs = line.find(' ');
        if (offs == std::string::npos)
            offs = line.size();
        for (size_t i = 0; i < N; ++i)
            if (line.compare(0, offs, tags[i]) == 0 && std::strlen(tags[i]) == offs)
                return i;
        return N;
    }
    static void chomp(std::string &line) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n'))
            line.pop_back();
    }

  public:
    OBJMesh(const std::filesystem::path &objPath,
            std::optional<std::unordered_map<std::string, InputLight>> inputLights = {})
        : objPath(objPath) {
        std::ifstream is(objPath);
        if (!is.is_open()) {
            fail("Cannot open file at \"" + objPath.string() + "\".");
            return;
        }
        static const char *const Tags[] = {"g", "f", "v", "vn", "vt", "mtllib", "usemtl"};
        bool prevGrpIsLight = false;
        InputLight prevGrpLight{};
        std::string line;
        uint32_t ln = 0;
        while (std::getline(is, line)) {
            ++ln;
            chomp(line);
            if (line.empty())
                continue;
            size_t offs;
            const size_t tag = parseTag(line, Tags, offs);
            const char *rest = line.c_str() + (offs < line.size() ? offs + 1 : offs);
            auto bad = [&]() { return fail("Failed parsing line " + std::to_string(ln) + "."); };
            float f[3];
            switch (tag) {
            case 0:
                grpStartFaceIndices.push_back(static_cast<IndexTy>(facePositionIndices.size()));
                break;
            case 1: {
                uint32_t v[9];
                const char *s = rest;
                bool ok = true;
                for (int i = 0; i < 9 && ok; ++i) {
                    ok = detail::scanUInt(s, v[i]);
                    if (ok && i % 3 != 2)
                        ok = *s++ == '/';
                }
                for (int i = 0; i < 9 && ok; ++i)
                    ok = v[i] != 0; // indices count from one (the reference lets a zero wrap to 2^32 - 1)
                if (!ok) {
                    bad();
                    return;
                }
                facePositionIndices.emplace_back(v[0] - 1, v[3] - 1, v[6] - 1);
                faceTexCoordIndices.emplace_back(v[1] - 1, v[4] - 1, v[7] - 1);
                faceNormalIndices.emplace_back(v[2] - 1, v[5] - 1, v[8] - 1);
            } break;
            case 2:
            case 3:
                if (detail::scanFloats(rest, f, 3) != 3) {
                    bad();
                    return;
                }
                (tag == 2 ? positions : normals).emplace_back(f[0], f[1], f[2]);
                break;
            case 4: {
                if (detail::scanFloats(rest, f, 2) != 2) {
                    bad();
                    return;
                }
                vec2 t;
                t[0] = f[0], t[1] = f[1];
                texCoords.push_back(t);
            } break;
            case 5:
                mtllibPath = objPath.parent_path() / rest;
                if (!loadMTLLib())
                    return;
                break;
            case 6: {
                if (grpStartFaceIndices.empty()) { // a material belongs to a group (the reference indexes group -1 here)
                    bad();
                    return;
                }
                if (prevGrpIsLight &&
                    !appendLight(prevGrpLight, static_cast<IndexTy>(grpStartFaceIndices.size()) - 2))
                    return;
                const IndexTy gi = static_cast<IndexTy>(grpStartFaceIndices.size()) - 1;
                if (!grp2mtlNames.emplace(gi, rest).second) {
                    bad();
                    return;
                }
                prevGrpIsLight = false;
                if (inputLights.has_value()) {
                    auto itr = inputLights->find(rest);
                    if (itr != inputLights->end()) {
                        prevGrpIsLight = true;
                        prevGrpLight = itr->second;
                    }
                }
            } break;
            default:
                bad();
                return;
            }
        }
        if (prevGrpIsLight && !appendLight(prevGrpLight, static_cast<IndexTy>(grpStartFaceIndices.size()) - 1))
            return;
        // Every index must name an element that exists: the arrays go to the GPU as they are, and a kernel
        // that follows an index past its array faults. (The reference does not check.)
        for (size_t fi = 0; fi < facePositionIndices.size(); ++fi)
            for (int i = 0; i < 3; ++i)
                if (facePositionIndices[fi][i] >= positions.size() || faceNormalIndices[fi][i] >= normals.size() ||
                    faceTexCoordIndices[fi][i] >= texCoords.size()) {
                    fail("Face " + std::to_string(fi) + " refers to a position, normal or texture coordinate that "
                         "the file does not define.");
                    return;
                }
        isComplete = true;
    }

    bool IsComplete() const { return isComplete; }
    const std::vector<vec3> &GetPositions() const { return positions; }
    const std::vector<vec3> &GetNormals() const { return normals; }
    const std::vector<vec2> &GetTextureCoordinates() const { return texCoords; }
    const std::vector<IndexTy> &GetGroupStartFaceIndices() const { return grpStartFaceIndices; }
    const std::vector<uvec3> &GetFacePositionIndices() const { return facePositionIndices; }
    const std::vector<uvec3> &GetFaceTextureCoordinateIndices() const { return faceTexCoordIndices; }
    const std::vector<uvec3> &GetFaceNormalIndices() const { return faceNormalIndices; }
    const std::unordered_map<IndexTy, std::string> &GetGroupToMaterialNames() const { return grp2mtlNames; }
    const std::unordered_map<std::string, Material> &GetNameToMaterials() const { return name2mtls; }
    const std::vector<Light> &GetLights() const { return lights; }
    const std::filesystem::path &GetOBJPath() const { return objPath; }
    const std::filesystem::path &GetMaterialLibraryPath() const { return mtllibPath; }

    // One triangle per cell of a dim.x x dim.y x dim.z grid over a box of size sz; the cell origin
    // advances by repeated float addition (reference mesh.h:255-284).
    static OBJMesh CreateGridTestMesh(const uvec3 &dim, const vec3 &sz) {
        OBJMesh mesh;
        const vec3 d(sz[0] / dim[0], sz[1] / dim[1], sz[2] / dim[2]);
        vec3 g(0.f, 0.f, 0.f);
        IndexTy vi = 0;
        for (IndexTy z = 0; z < dim[2]; ++z) {
            g[1] = 0.f;
            for (IndexTy y = 0; y < dim[1]; ++y) {
                g[0] = 0.f;
                for (IndexTy x = 0; x < dim[0]; ++x) {
                    mesh.positions.emplace_back(g[0], g[1], g[2]);
                    mesh.positions.emplace_back(g[0] + d[0], g[1], g[2]);
                    mesh.positions.emplace_back(g[0], g[1] + d[1], g[2] + d[2]);
                    mesh.facePositionIndices.emplace_back(vi, vi + 1, vi + 2);
                    vi += 3;
                    g[0] += d[0];
                }
                g[1] += d[1];
            }
            g[2] += d[2];
        }
        mesh.isComplete = true;
        return mesh;
    }

  private:
    // reference mesh.h:322-381
    bool appendLight(const InputLight &in, IndexTy gi) {
        if (gi >= grpStartFaceIndices.size() || faceNormalIndices.empty())
            return fail("A light's material is used before a group with faces exists.");
        const IndexTy fiBeg = grpStartFaceIndices[gi];
        const IndexTy fiEnd = gi == grpStartFaceIndices.size() - 1
                                  ? static_cast<IndexTy>(faceNormalIndices.size()) - 1 // (sic) one face short
                                  : grpStartFaceIndices[gi + 1];
        // the group's faces are read here, while the file is still being parsed: they and their vertices
        // must have come before
        if (fiBeg > fiEnd || fiEnd > facePositionIndices.size())
            return fail("Group " + std::to_string(gi) + " of a light holds no faces.");
        for (IndexTy fi = fiBeg; fi < fiEnd; ++fi)
            for (int i = 0; i < 3; ++i)
                if (facePositionIndices[fi][i] >= positions.size())
                    return fail("Face " + std::to_string(fi) + " of a light refers to a position that is not defined.");
        Light lht{};
        lht.type = in.type;
        lht.radiance = in.radiance;
        if (in.type == Light::Type::Quad) {
            if (fiEnd - fiBeg != 2)
                return fail("Triangle number of group " + std::to_string(gi) + " is not 2.");
            const uvec3 &idx = facePositionIndices[fiBeg];
            const vec3 e[3] = {detail::sub3(positions[idx[1]], positions[idx[0]]),
                               detail::sub3(positions[idx[2]], positions[idx[1]]),
                               detail::sub3(positions[idx[0]], positions[idx[2]])};
            const float dots[3] = {detail::dot3(e[0], e[1]), detail::dot3(e[1], e[2]), detail::dot3(e[2], e[0])};
            int corner = -1;
            for (int i = 0; i < 3 && corner < 0; ++i)
                if (std::fabs(dots[i]) < 1.192092896e-07f)
                    corner = i;
            if (corner < 0)
                return fail("Triangle at " + std::to_string(fiBeg) + " is not right.");
            lht.quad.o = positions[idx[(corner + 1) % 3]];
            lht.quad.u = detail::sub3(positions[idx[(corner + 2) % 3]], lht.quad.o);
            lht.quad.v = detail::sub3(positions[idx[corner]], lht.quad.o);
        } else {
            if (fiEnd == fiBeg)
                return fail("Group " + std::to_string(gi) + " of a light holds no faces.");
            const float div = 3.f * static_cast<float>(fiEnd - fiBeg);
            vec3 o(0.f, 0.f, 0.f);
            for (IndexTy fi = fiBeg; fi < fiEnd; ++fi)
                for (int i = 0; i < 3; ++i) {
                    const vec3 &p = positions[facePositionIndices[fi][i]];
                    o = vec3(o[0] + p[0] / div, o[1] + p[1] / div, o[2] + p[2] / div);
                }
            float r = 0.f;
            for (IndexTy fi = fiBeg; fi < fiEnd; ++fi)
                for (int i = 0; i < 3; ++i) {
                    const vec3 d = detail::sub3(positions[facePositionIndices[fi][i]], o);
                    r += std::sqrt(detail::dot3(d, d)) / div;
                }
            lht.sphere.o = o;
            lht.sphere.r = r;
        }
        lights.push_back(lht);
        return true;
    }

    // reference mesh.h:383-461
    bool loadMTLLib() {
        std::ifstream is(mtllibPath);
        if (!is.is_open())
            return fail("Cannot open file at \"" + mtllibPath.string() + "\".");
        static const char *const Tags[] = {"Kd", "Ks", "Tr", "Ni", "Ns", "newmtl"};
        name2mtls.clear();
        Material *cur = nullptr;
        std::string line;
        uint32_t ln = 0;
        while (std::getline(is, line)) {
            ++ln;
            chomp(line);
            if (line.empty())
                continue;
            size_t offs;
            const size_t tag = parseTag(line, Tags, offs);
            const char *rest = line.c_str() + (offs < line.size() ? offs + 1 : offs);
            auto bad = [&]() { return fail("Failed parsing line " + std::to_string(ln) + "."); };
            float f[3];
            if (tag == 5) {
                auto res = name2mtls.emplace(rest, Material{});
                if (!res.second)
                    return bad();
                cur = &res.first->second;
            } else if (tag == 3 || tag == 4) {
                if (!cur || detail::scanFloats(rest, f, 1) != 1)
                    return bad();
                (tag == 3 ? cur->ni : cur->ns) = f[0];
            } else if (tag < 3) {
                if (!cur || detail::scanFloats(rest, f, 3) != 3)
                    return bad();
                (tag == 0 ? cur->kd : tag == 1 ? cur->ks : cur->tr) = vec3(f[0], f[1], f[2]);
            } else
                return bad();
        }
        return true;
    }
};

// The camera and the lights of the scene description next to the OBJ file (<name>.xml).
struct SceneConfig {
    bool isComplete = false;
    ivec2 rndrSz{0, 0};
    vec3 eye, cntr, up;
    std::unordered_map<std::string, OBJMesh::InputLight> lights;

    explicit SceneConfig(const std::filesystem::path &objPath) {
        const auto path = objPath.parent_path() / (objPath.stem().string() + ".xml");
        std::ifstream is(path);
        if (!is.is_open())
            return;
        std::stringstream ss;
        ss << is.rdbuf();
        const std::string xml = ss.str();
        std::string cam;
        if (!element(xml, "camera", 0, cam) || attribute(cam, "type") != "perspective")
            return;
        const std::string w = attribute(cam, "width"), h = attribute(cam, "height");
        if (w.empty() || h.empty())
            return;
        rndrSz.x = std::atoi(w.c_str());
        rndrSz.y = std::atoi(h.c_str());
        auto vec = [&](const char *name, vec3 &v) {
            std::string e;
            if (!element(xml, name, 0, e))
                return false;
            const char *keys[3] = {"x", "y", "z"};
            for (int i = 0; i < 3; ++i) {
                const std::string a = attribute(e, keys[i]);
                float f;
                if (a.empty() || detail::scanFloats(a.c_str(), &f, 1) != 1)
                    return false;
                v[i] = f;
            }
            return true;
        };
        if (!vec("eye", eye) || !vec("lookat", cntr) || !vec("up", up))
            return;
        size_t from = 0;
        std::string lht;
        while ((from = element(xml, "light", from, lht)) != 0) {
            OBJMesh::InputLight in{};
            const std::string type = attribute(lht, "type");
            if (type == "Quad")
                in.type = OBJMesh::Light::Type::Quad;
            else if (type == "Sphere")
                in.type = OBJMesh::Light::Type::Sphere;
            else
                return;
            std::string rad = attribute(lht, "radiance");
            float f[3];
            for (char &c : rad)
                if (c == ',')
                    c = ' ';
            if (detail::scanFloats(rad.c_str(), f, 3) != 3)
                return;
            in.radiance = vec3(f[0], f[1], f[2]);
            const std::string name = attribute(lht, "mtlname");
            if (name.empty())
                return;
            lights.emplace(name, in);
        }
        isComplete = true;
    }

  private:
    // the text of the first <name ...> start tag at or after `from`; returns the position behind it (0: none)
    static size_t element(const std::string &xml, const char *name, size_t from, std::string &out) {
        const std::string open = std::string("<") + name;
        size_t p = from;
        while ((p = xml.find(open, p)) != std::string::npos) {
            const char next = p + open.size() < xml.size() ? xml[p + open.size()] : '>';
            if (next == ' ' || next == '\t' || next == '\n' || next == '\r' || next == '>' || next == '/') {
                const size_t end = xml.find('>', p);
                if (end == std::string::npos)
                    return 0;
                out = xml.substr(p, end - p + 1);
                return end + 1;
            }
            p += open.size();
        }
        return 0;
    }
    static std::string attribute(const std::string &tag, const char *key) {
        const std::string k = std::string(key) + "=\"";
        size_t p = 0;
        while ((p = tag.find(k, p)) != std::string::npos) {
            if (p > 0 && (tag[p - 1] == ' ' || tag[p - 1] == '\t' || tag[p - 1] == '\n')) {
                const size_t b = p + k.size(), e = tag.find('"', b);
                return e == std::string::npos ? std::string() : tag.substr(b, e - b);
            }
            p += k.size();
        }
        return std::string();
    }
};

} // namespace Data
} // namespace kouek

#endif // KOUEK_DATA_MESH_H
