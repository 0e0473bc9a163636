// TEST INFRASTRUCTURE ONLY (see ko_math.h). CPU restatement of the reference's path-tracing frame:
// ray generation, renderScene, accumulation and RGBA8 output
// (src/ray_tracer/ray_tracer_render.cu:312-431, 579-638), with Material::BRDF and the light sampling of
// include/ray_tracer/material.h and the helpers of include/cg/math.h:75-129; and of the inspection
// render targets (AABBs, Triangles, Lights, Positions, Normals, TextureCoords; :8-100, 272-310, 461-558).
//
// Parity status: the traversal and intersection arithmetic is pinned (oracle_lbvh.cpp). The random
// stream is NOT the reference's: it draws from cuRAND XORWOW (curand_init(1234, pixel, 0)), which is
// not available here and which nothing in the reference pins. Both this checker and the HIP kernel use
// the counter-based generator defined below (Philox4x32-10, key = seed, counter = draw / sample /
// pixel), drawn in the reference's order. "RNG stream parity: unpinned."
#include "ko_math.h"

#include <cstring>
#include <vector>

using namespace ko;

namespace {

constexpr float Eps = .001f; // include/ray_tracer/ray_tracer.h:27
constexpr float Pi = 3.14159265358979323846264338327950288f;
constexpr float TwoPi = 6.28318530717958647692528676655900576f;
constexpr float OneOverPi = 0.318309886183790671537767526745028724f;
constexpr float OneOverTwoPi = 0.159154943091895335768883763372514362f;

struct InternalNode {
    uint32_t lft, rht;
    AABB aabb;
};
constexpr uint32_t ChildIsLeafMask = 0x80000000u;
constexpr uint32_t LeafIndexMask = 0x7fffffffu;

// include/ray_tracer/ray_tracer.h:29-33 (alignas(16): 28 -> 32 bytes)
struct Triangle {
    uint32_t normIdx[3];
    uint32_t texCoordIdx[3];
    uint32_t grpIdx;
    uint32_t pad;
};
static_assert(sizeof(Triangle) == 32, "Triangle");

// include/ray_tracer/material.h:12-22
struct Material {
    uint8_t emissive;
    uint8_t pad[3];
    float emission[3], kd[3], ks[3], tr[3];
    float ni, ns, kdOverKdAddKs;
};
static_assert(sizeof(Material) == 64, "Material");

// include/ray_tracer/material.h:51-66
struct Light {
    uint8_t type; // 0 quad, 1 sphere
    uint8_t pad[3];
    union {
        struct {
            float o[3], u[3], v[3], n[3];
            float invArea;
        } quad;
        struct {
            float o[3];
            float r, rSqr;
        } sphere;
    };
    float area;
    float radiance[3];
    float pad2[2];
};
static_assert(sizeof(Light) == 80, "Light");

struct Scene {
    const InternalNode *inodes;
    const uint32_t *faces, *compactedFaces;
    uint32_t faceNum, compactedFaceNum;
    const float *positions;
    const uint32_t *facePosIdx;
    const float *normals;
    const Triangle *triangles;
    const Material *materials;
    const Light *lights;
    uint32_t lightNum;
    float lightTotArea;
    uint8_t maxPathDepth;
};

inline vec3 V(const float *p) { return {p[0], p[1], p[2]}; }
inline vec3 LoadV3(const float *p, uint32_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// ---- random numbers ---------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
struct Rng {
    uint32_t seed, pixel, sample, draw;
    // four uniforms in (0, 1] (like curand_uniform): the words of one block; a sampling event of a path
    // (pixel jitter, light sample, bounce) takes one block
    void next4(float &u0, float &u1, float &u2, float &u3) {
        uint32_t c[4] = {draw++, sample, pixel, 0x4B4F5545u};
        uint32_t k[2] = {seed, 0xCAFEF00Du};
        for (int r = 0; r < 10; ++r) {
            uint64_t p0 = 0xD2511F53ull * c[0];
            uint64_t p1 = 0xCD9E8D57ull * c[2];
            uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k[0];
            uint32_t n1 = static_cast<uint32_t>(p1);
            uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k[1];
            uint32_t n3 = static_cast<uint32_t>(p0);
            c[0] = n0;
            c[1] = n1;
            c[2] = n2;
            c[3] = n3;
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        u0 = static_cast<float>((c[0] >> 8) + 1u) * 5.9604644775390625e-08f;
        u1 = static_cast<float>((c[1] >> 8) + 1u) * 5.9604644775390625e-08f;
        u2 = static_cast<float>((c[2] >> 8) + 1u) * 5.9604644775390625e-08f;
        u3 = static_cast<float>((c[3] >> 8) + 1u) * 5.9604644775390625e-08f;
    }
};

// ---- traversal (ray_tracer_render.cu:192-270), t only when simple ------------------------------------
struct Nearest {
    float t;
    vec2 bary;
    uint32_t fi;
};

Nearest rayCastLBVH(const Scene &s, const Ray &ray) {
    Nearest nearest{FloatMax, {0.f, 0.f}, 0xFFFFFFFFu};
    HitShell hit{0.f, 0.f};
    auto hitNode = [&](uint32_t idx) {
        if (idx & ChildIsLeafMask)
            return false;
        hit = HitAABB(ray, s.inodes[idx].aabb);
        return true;
    };
    hitNode(0);
    if (hit.tEnter > hit.tExit)
        return nearest;
    uint32_t stkNode[64];
    uint8_t stkCnt[64];
    int sz = 1;
    stkNode[0] = 0;
    stkCnt[0] = 0;
    while (sz != 0) {
        if (stkCnt[sz - 1] == 2) {
            --sz;
            continue;
        }
        uint32_t chIdx =
            stkCnt[sz - 1] == 0 ? s.inodes[stkNode[sz - 1]].lft : s.inodes[stkNode[sz - 1]].rht;
        ++stkCnt[sz - 1];
        bool childIsINode = hitNode(chIdx);
        if (childIsINode && hit.tEnter > hit.tExit)
            continue;
        else if (childIsINode) {
            stkNode[sz] = chIdx;
            stkCnt[sz] = 0;
            ++sz;
        } else {
            uint32_t c = chIdx & LeafIndexMask;
            uint32_t k = s.compactedFaces[c];
            uint32_t kEnd = (c == s.compactedFaceNum - 1) ? s.faceNum : s.compactedFaces[c + 1];
            for (; k < kEnd; ++k) {
                uint32_t fi = s.faces[k];
                HitTri h = HitTriangle(ray, LoadV3(s.positions, s.facePosIdx[3 * fi]),
                                       LoadV3(s.positions, s.facePosIdx[3 * fi + 1]),
                                       LoadV3(s.positions, s.facePosIdx[3 * fi + 2]));
                if (h.t >= 0.f && h.t < nearest.t) {
                    nearest.t = h.t;
                    nearest.bary = h.bary;
                    nearest.fi = fi;
                }
            }
        }
    }
    return nearest;
}

// ---- include/cg/math.h:75-129 ----------------------------------------------------------------------
vec3 GenerateTangent(vec3 norm) {
    if (std::fabs(norm.y) < .999f)
        return normalize(cross(vec3{0.f, 1.f, 0.f}, norm));
    return normalize(cross(norm, vec3{1.f, 0.f, 0.f}));
}
vec3 FrameToWorld(vec3 dir, vec3 norm, vec3 tngnt) {
    vec3 biTngnt = normalize(cross(norm, tngnt));
    mat3 m{{tngnt, biTngnt, norm}};
    return normalize(m * dir);
}
vec3 ThetaPhiToDirection(float theta, float phi, vec3 norm, vec3 tngnt) {
    float sinTheta = std::sin(theta);
    vec3 dir = {std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, std::cos(theta)};
    return FrameToWorld(dir, norm, tngnt);
}
vec3 CosineThetaPhiToDirection(float cosTheta, float phi, vec3 norm, vec3 tngnt) {
    float sinTheta = std::sqrt(gclamp(1.f - cosTheta * cosTheta, 0.f, 1.f));
    vec3 dir = {std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, cosTheta};
    return FrameToWorld(dir, norm, tngnt);
}

// include/ray_tracer/material.h:41-48
vec3 BRDF(const Material &m, vec3 dirIn, vec3 dirOut, vec3 norm) {
    vec3 diffuse = V(m.kd) * OneOverPi;
    vec3 reflect = normalize(2.f * dot(norm, dirOut) * norm - dirOut);
    vec3 specular =
        V(m.ks) * (m.ns + 2.f) * OneOverTwoPi * std::pow(gmax(dot(dirIn, reflect), 0.f), m.ns);
    return diffuse + specular;
}

struct LightSample {
    vec3 pos, norm;
    float pdf;
};
// include/ray_tracer/material.h:100-114
LightSample SampleLight(const Light &l, float u0, float u1, vec3 pos) {
    if (l.type == 0)
        return {V(l.quad.o) + u0 * V(l.quad.u) + u1 * V(l.quad.v), V(l.quad.n), l.quad.invArea};
    vec3 dirO2P = normalize(pos - V(l.sphere.o));
    vec3 tngnt = GenerateTangent(dirO2P);
    LightSample r;
    r.norm = ThetaPhiToDirection(std::acos(1 - u0), TwoPi * u1, dirO2P, tngnt);
    r.pos = V(l.sphere.o) + l.sphere.r * r.norm;
    r.pdf = OneOverTwoPi;
    return r;
}

// src/ray_tracer/ray_tracer_render.cu:312-431
vec3 renderScene(const Scene &s, Rng &rng, Ray ray) {
    constexpr float RR = .8f;
    vec3 throughout = V3(1.f);
    vec3 Lo = V3(0.f);
    for (uint8_t depth = 0;
         depth < s.maxPathDepth && gmax(throughout.x, gmax(throughout.y, throughout.z)) >= FloatEps;
         ++depth) {
        Nearest nearest = rayCastLBVH(s, ray);
        if (nearest.t == FloatMax)
            break;
        const Triangle &triangle = s.triangles[nearest.fi];
        const Material &matr = s.materials[triangle.grpIdx];
        if (matr.emissive) {
            Lo += throughout * V(matr.emission);
            break;
        }
        vec3 norm = normalize(nearest.bary.x * LoadV3(s.normals, triangle.normIdx[0]) +
                              nearest.bary.y * LoadV3(s.normals, triangle.normIdx[1]) +
                              (1.f - nearest.bary.x - nearest.bary.y) *
                                  LoadV3(s.normals, triangle.normIdx[2]));
        vec3 dirOut = -ray.dir;
        ray.pos = ray.pos + nearest.t * ray.dir + Eps * norm;

        vec3 Li = V3(0.f);
        for (uint32_t li = 0; li < s.lightNum; ++li) {
            const Light &lht = s.lights[li];
            float u0, u1, unused2, unused3;
            rng.next4(u0, u1, unused2, unused3);
            LightSample ls = SampleLight(lht, u0, u1, ray.pos);
            ls.pos += Eps * ls.norm;
            vec3 in = ls.pos - ray.pos;
            float sqrIn = dot(in, in);
            ray.dir = normalize(in);
            float factor = gmax(dot(ray.dir, norm), 0.f) * gmax(dot(-ray.dir, ls.norm), 0.f) /
                           sqrIn * lht.area / s.lightTotArea;
            if (factor < FloatEps)
                continue;
            Nearest shadow = rayCastLBVH(s, ray);
            if (shadow.t < std::sqrt(sqrIn) - Eps)
                continue;
            Li += factor / ls.pdf * V(lht.radiance) * BRDF(matr, ray.dir, dirOut, norm);
        }

        float pdfIndrct;
        float u, u0, u1, uRoulette;
        rng.next4(u, u0, u1, uRoulette);
        if (u <= matr.kdOverKdAddKs) {
            vec3 tngnt = GenerateTangent(norm);
            ray.dir = CosineThetaPhiToDirection(std::sqrt(1.f - u0), TwoPi * u1, norm, tngnt);
            pdfIndrct = dot(ray.dir, norm) * OneOverPi * matr.kdOverKdAddKs;
        } else {
            vec3 reflect = normalize(2.f * dot(norm, dirOut) * norm - dirOut);
            vec3 tngnt = GenerateTangent(reflect);
            ray.dir = ThetaPhiToDirection(std::acos(std::pow(u0, 1.f / (matr.ns + 1.f))), TwoPi * u1,
                                          reflect, tngnt);
            pdfIndrct = (matr.ns + 1.f) * OneOverTwoPi *
                        std::pow(gmax(dot(reflect, ray.dir), 0.f), matr.ns) *
                        (1.f - matr.kdOverKdAddKs);
        }
        Lo += throughout * Li;
        throughout *= dot(ray.dir, norm) / pdfIndrct / RR * BRDF(matr, ray.dir, dirOut, norm);

        if (uRoulette > RR && depth > 1)
            break;
    }
    // HDRToLDRCorrect + GammaCorrect, include/cg/math.h:75-93
    Lo = Lo / (Lo + 1.f);
    for (int k = 0; k < 3; ++k)
        Lo[k] = std::pow(Lo[k], 1.f / 2.2f);
    return Lo;
}

// ---- the inspection render targets (src/ray_tracer/ray_tracer_render.cu:8-100, 272-310, 503-556) ----

// static_cast<uint8_t>(float) as the device does it: toward zero into 32 unsigned bits with
// saturation (negative and NaN -> 0), then the low byte. glm::clamp(x, 0, 255) of a uint8_t is x.
inline uint8_t ToU8(float v) {
    uint32_t u = v > 0.f ? (v < 4294967040.f ? static_cast<uint32_t>(v) : 0xFFFFFFFFu) : 0u;
    return static_cast<uint8_t>(u);
}
struct RGBA8 {
    uint8_t r, g, b, a;
};
inline RGBA8 ToRGBA(vec3 c) { return {ToU8(255.f * c.x), ToU8(255.f * c.y), ToU8(255.f * c.z), 255}; }

// include/cg/ray.h:61-86
struct HitQuadResult {
    float t, u, v;
};
HitQuadResult HitQuad(const Ray &r, vec3 o, vec3 u, vec3 v) {
    HitQuadResult ret{-1.f, 0.f, 0.f};
    // glm::intersectRayPlane, 3rd/glm/gtx/intersect.inl:6-26
    vec3 n = normalize(cross(u, v));
    float d = dot(r.dir, n);
    bool hitPlane = false;
    if (std::fabs(d) > FloatEps) {
        float tmp = dot(o - r.pos, n) / d;
        if (tmp > 0.f) {
            ret.t = tmp;
            hitPlane = true;
        }
    }
    if (!hitPlane) {
        ret.t = -1.f;
        return ret;
    }
    vec3 xLocal = r.pos + ret.t * r.dir - o;
    float lenU = length(u), lenV = length(v);
    ret.u = dot(xLocal, u / lenU);
    ret.v = dot(xLocal, v / lenV);
    if (ret.u < 0.f || ret.u > lenU || ret.v < 0.f || ret.v > lenV)
        ret.t = -1.f;
    return ret;
}
// include/cg/ray.h:88-106
HitShell HitSphere(const Ray &r, float rSqr, vec3 o) {
    vec3 p2o = o - r.pos;
    float p2oDotDir = dot(p2o, r.dir);
    if (p2oDotDir < 0.f)
        return {0.f, -1.f};
    float qSqr = dot(p2o, p2o) - p2oDotDir * p2oDotDir;
    if (qSqr > rSqr)
        return {0.f, -2.f};
    float p = std::sqrt(rSqr - qSqr);
    return {gmax(p2oDotDir - p, 0.f), p2oDotDir + 2.f * p};
}

// ray_tracer_render.cu:8-100
RGBA8 renderAABB(const Scene &s, const AABB &rootAABB, uint8_t displayTreeHeight, const Ray &eyeRay) {
    vec3 extent = rootAABB.maxPos - rootAABB.minPos;
    for (int a = 0; a < 3; ++a)
        if (extent[a] < FloatEps)
            extent[a] = FloatEps;
    HitShell hit{0.f, 0.f};
    auto hitNode = [&](uint32_t idx) {
        if (idx & ChildIsLeafMask)
            return false;
        hit = HitAABB(eyeRay, s.inodes[idx].aabb);
        return true;
    };
    hitNode(0);
    if (hit.tEnter > hit.tExit)
        return {0, 0, 0, 0};
    constexpr float EnterAlpha = .7f;
    constexpr float ExitAlpha = 1.f - EnterAlpha;
    vec3 color = V3(-1.f);
    if (displayTreeHeight == 0) {
        vec3 enterPos = eyeRay.pos + hit.tEnter * eyeRay.dir - rootAABB.minPos;
        vec3 exitPos = eyeRay.pos + hit.tExit * eyeRay.dir - rootAABB.minPos;
        color = (EnterAlpha * enterPos + ExitAlpha * exitPos) / extent;
        return ToRGBA(color);
    }
    uint32_t stkNode[32];
    uint8_t stkCnt[32];
    uint8_t sz = 1;
    stkNode[0] = 0;
    stkCnt[0] = 0;
    float tNearest = FloatMax;
    while (sz != 0) {
        if (stkCnt[sz - 1] == 2) {
            --sz;
            continue;
        }
        uint32_t chIdx =
            stkCnt[sz - 1] == 0 ? s.inodes[stkNode[sz - 1]].lft : s.inodes[stkNode[sz - 1]].rht;
        ++stkCnt[sz - 1];
        bool childIsINode = hitNode(chIdx);
        if (childIsINode && hit.tEnter <= hit.tExit) {
            if (displayTreeHeight == sz - 1 && hit.tEnter < tNearest) {
                vec3 minPos = s.inodes[chIdx].aabb.minPos;
                vec3 ext = s.inodes[chIdx].aabb.maxPos - s.inodes[chIdx].aabb.minPos;
                for (int a = 0; a < 3; ++a)
                    if (ext[a] < FloatEps)
                        ext[a] = FloatEps;
                vec3 enterPos = eyeRay.pos + hit.tEnter * eyeRay.dir - minPos;
                vec3 exitPos = eyeRay.pos + hit.tExit * eyeRay.dir - minPos;
                color = (EnterAlpha * enterPos + ExitAlpha * exitPos) / ext;
                tNearest = hit.tEnter;
            } else if (sz <= displayTreeHeight) {
                stkNode[sz] = chIdx;
                stkCnt[sz] = 0;
                ++sz;
            }
        }
    }
    if (tNearest == FloatMax)
        return {0, 0, 0, 0};
    return ToRGBA(color);
}

// ray_tracer_render.cu:272-310
RGBA8 renderLights(const Scene &s, const Ray &eyeRay) {
    vec3 color = V3(0.f);
    float tNearest = FloatMax;
    for (uint32_t li = 0; li < s.lightNum; ++li) {
        const Light &l = s.lights[li];
        if (l.type == 0) {
            HitQuadResult h = HitQuad(eyeRay, V(l.quad.o), V(l.quad.u), V(l.quad.v));
            if (h.t >= 0.f && h.t < tNearest) {
                color.x = h.u / length(V(l.quad.u));
                color.y = h.v / length(V(l.quad.v));
                tNearest = h.t;
            }
        } else {
            HitShell h = HitSphere(eyeRay, l.sphere.rSqr, V(l.sphere.o));
            if (h.tEnter <= h.tExit && h.tEnter < tNearest) {
                vec3 dir = eyeRay.pos + h.tEnter * eyeRay.dir - V(l.sphere.o);
                dir = normalize(dir);
                color = vabs(dir);
                tNearest = h.tEnter;
            }
        }
    }
    if (tNearest == FloatMax)
        return {0, 0, 0, 0};
    return ToRGBA(color);
}

// the four closest-hit targets, ray_tracer_render.cu:508-551
RGBA8 renderHitTarget(const Scene &s, const AABB &rootAABB, int target, const Ray &eyeRay) {
    Nearest n = rayCastLBVH(s, eyeRay);
    if (n.t == FloatMax)
        return {0, 0, 0, 0};
    vec3 color = V3(0.f);
    const float bx = n.bary.x, by = n.bary.y;
    if (target == 2) { // Triangles
        color = {bx, by, 1.f - bx - by};
    } else if (target == 4) { // Positions
        vec3 pos = eyeRay.pos + n.t * eyeRay.dir;
        color = (pos - rootAABB.minPos) / (rootAABB.maxPos - rootAABB.minPos);
    } else if (target == 5) { // Normals: norms[0] is used for both of the first two weights (:537)
        const Triangle &f = s.triangles[n.fi];
        vec3 n0 = LoadV3(s.normals, f.normIdx[0]), n2 = LoadV3(s.normals, f.normIdx[2]);
        color = vabs(bx * n0 + by * n0 + (1.f - bx - by) * n2);
    } else { // TextureCoords: reads the NORMALS array with the texture-coordinate indices (:547-551)
        const Triangle &f = s.triangles[n.fi];
        vec3 a = LoadV3(s.normals, f.texCoordIdx[0]), c = LoadV3(s.normals, f.texCoordIdx[2]);
        vec2 n0 = {a.x, a.y}, n2 = {c.x, c.y};
        float w = 1.f - bx - by;
        color.x = std::fabs(bx * n0.x + by * n0.x + w * n2.x);
        color.y = std::fabs(bx * n0.y + by * n0.y + w * n2.y);
    }
    return ToRGBA(color);
}

} // namespace

extern "C" {

// Material::BRDF (include/ray_tracer/material.h:41-48) for one 64-byte Material.
void ko_material_brdf(const void *material, const float *dirIn, const float *dirOut, const float *norm,
                      float *out3) {
    const vec3 r = BRDF(*static_cast<const Material *>(material), vec3{dirIn[0], dirIn[1], dirIn[2]},
                        vec3{dirOut[0], dirOut[1], dirOut[2]}, vec3{norm[0], norm[1], norm[2]});
    out3[0] = r.x, out3[1] = r.y, out3[2] = r.z;
}
// Light::SampleQuad / SampleSphere (include/ray_tracer/material.h:95-114) for one 80-byte Light:
// out7 = pos, norm, pdf.
void ko_light_sample(const void *light, float u0, float u1, const float *pos, float *out7) {
    const LightSample r = SampleLight(*static_cast<const Light *>(light), u0, u1, vec3{pos[0], pos[1], pos[2]});
    out7[0] = r.pos.x, out7[1] = r.pos.y, out7[2] = r.pos.z;
    out7[3] = r.norm.x, out7[4] = r.norm.y, out7[5] = r.norm.z;
    out7[6] = r.pdf;
}

struct ko_pt_scene {
    const void *inodes;
    const uint32_t *faces, *compactedFaces;
    uint32_t faceNum, compactedFaceNum;
    const float *positions;
    const uint32_t *facePosIdx;
    const float *normals;
    const void *triangles;
    const void *materials;
    const void *lights;
    uint32_t lightNum;
    float lightTotArea;
    uint32_t maxPathDepth;
    float w2s[16], invProj[16];
    float eyePos[3], eyeRot[9];
};

// One Render(Scene) call of ray_tracer_render.cu:579-638 over rows [y0, y1): adds sample number
// `sampleIdx` (the reference's currAccuSamplePerPixel, 1-based) to accum (W*H*3 floats, running mean)
// and writes RGBA8. Pixel row 0 is the bottom row.
void ko_pt_render_sample(const ko_pt_scene *sc, float *accum, uint8_t *rgba, int W, int H, int y0,
                         int y1, float sampleIdx, uint32_t seed) {
    Scene s;
    s.inodes = static_cast<const InternalNode *>(sc->inodes);
    s.faces = sc->faces;
    s.compactedFaces = sc->compactedFaces;
    s.faceNum = sc->faceNum;
    s.compactedFaceNum = sc->compactedFaceNum;
    s.positions = sc->positions;
    s.facePosIdx = sc->facePosIdx;
    s.normals = sc->normals;
    s.triangles = static_cast<const Triangle *>(sc->triangles);
    s.materials = static_cast<const Material *>(sc->materials);
    s.lights = static_cast<const Light *>(sc->lights);
    s.lightNum = sc->lightNum;
    s.lightTotArea = sc->lightTotArea;
    s.maxPathDepth = static_cast<uint8_t>(sc->maxPathDepth);
    mat4 w2s, invProj;
    mat3 rot;
    std::memcpy(&w2s, sc->w2s, 64);
    std::memcpy(&invProj, sc->invProj, 64);
    std::memcpy(&rot, sc->eyeRot, 36);
    vec3 eyePos = V(sc->eyePos);

#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) {
            uint32_t idx = static_cast<uint32_t>(y) * W + x;
            Rng rng{seed, idx, static_cast<uint32_t>(sampleIdx), 0};
            Ray eyeRay;
            {
                vec4 tmp;
                float jx, jy, unused2, unused3;
                rng.next4(jx, jy, unused2, unused3);
                tmp.x = 1.f - jx + x;
                tmp.y = 1.f - jy + y;
                tmp.z = static_cast<float>(W);
                tmp.w = static_cast<float>(H);
                tmp.x = (2.f * tmp.x - tmp.z) / tmp.z;
                tmp.y = (2.f * tmp.y - tmp.w) / tmp.w;
                tmp.z = 1.f;
                tmp.w = 1.f;
                tmp = invProj * tmp;
                eyeRay.dir = {tmp.x, tmp.y, tmp.z};
                eyeRay.dir = rot * normalize(eyeRay.dir);
                // the reference transforms a temporary by w2s and discards it (:612-614)
                eyeRay.dir = normalize(eyeRay.dir);
                vec4 p = w2s * vec4{eyePos.x, eyePos.y, eyePos.z, 1.f};
                eyeRay.pos = {p.x, p.y, p.z};
            }
            vec3 color = renderScene(s, rng, eyeRay);
            vec3 prev = {accum[3 * idx], accum[3 * idx + 1], accum[3 * idx + 2]};
            color = prev + (color - prev) / sampleIdx;
            accum[3 * idx] = color.x;
            accum[3 * idx + 1] = color.y;
            accum[3 * idx + 2] = color.z;
            rgba[4 * idx] = static_cast<uint8_t>(255.f * color.x);
            rgba[4 * idx + 1] = static_cast<uint8_t>(255.f * color.y);
            rgba[4 * idx + 2] = static_cast<uint8_t>(255.f * color.z);
            rgba[4 * idx + 3] = 255;
        }
}

// One Render call of ray_tracer_render.cu:461-558 for a target other than Scene, rows [y0, y1).
// target: 1 AABBs, 2 Triangles, 3 Lights, 4 Positions, 5 Normals, 6 TextureCoords (the order of
// RayTracer::RenderTarget). rootAABB = LBVH::DeviceData::rootAABB (6 floats).
void ko_rt_render_target(const ko_pt_scene *sc, const float *rootAABB6, int displayTreeHeight,
                         int target, uint8_t *rgba, int W, int H, int y0, int y1) {
    Scene s;
    s.inodes = static_cast<const InternalNode *>(sc->inodes);
    s.faces = sc->faces;
    s.compactedFaces = sc->compactedFaces;
    s.faceNum = sc->faceNum;
    s.compactedFaceNum = sc->compactedFaceNum;
    s.positions = sc->positions;
    s.facePosIdx = sc->facePosIdx;
    s.normals = sc->normals;
    s.triangles = static_cast<const Triangle *>(sc->triangles);
    s.materials = static_cast<const Material *>(sc->materials);
    s.lights = static_cast<const Light *>(sc->lights);
    s.lightNum = sc->lightNum;
    s.lightTotArea = sc->lightTotArea;
    s.maxPathDepth = static_cast<uint8_t>(sc->maxPathDepth);
    AABB root{{rootAABB6[0], rootAABB6[1], rootAABB6[2]}, {rootAABB6[3], rootAABB6[4], rootAABB6[5]}};
    mat4 w2s, invProj;
    mat3 rot;
    std::memcpy(&w2s, sc->w2s, 64);
    std::memcpy(&invProj, sc->invProj, 64);
    std::memcpy(&rot, sc->eyeRot, 36);
    vec3 eyePos = V(sc->eyePos);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < W; ++x) {
            Ray eyeRay;
            vec4 tmp;
            tmp.z = static_cast<float>(W);
            tmp.w = static_cast<float>(H);
            tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
            tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
            tmp.z = 1.f;
            tmp.w = 1.f;
            tmp = invProj * tmp;
            eyeRay.dir = {tmp.x, tmp.y, tmp.z};
            eyeRay.dir = rot * normalize(eyeRay.dir);
            eyeRay.dir = normalize(eyeRay.dir); // the w2s-transformed temporary is discarded (:487-489)
            vec4 p = w2s * vec4{eyePos.x, eyePos.y, eyePos.z, 1.f};
            eyeRay.pos = {p.x, p.y, p.z};
            RGBA8 c;
            if (target == 1)
                c = renderAABB(s, root, static_cast<uint8_t>(displayTreeHeight), eyeRay);
            else if (target == 3)
                c = renderLights(s, eyeRay);
            else
                c = renderHitTarget(s, root, target, eyeRay);
            uint8_t *o = rgba + 4 * (static_cast<size_t>(y) * W + x);
            o[0] = c.r, o[1] = c.g, o[2] = c.b, o[3] = c.a;
        }
}

// The generator on its own, so that the device implementation can be checked value for value.
void ko_rng_uniforms(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t count, float *out) {
    Rng rng{seed, pixel, sample, 0};
    for (uint32_t i = 0; i < count; i += 4) {
        float u[4];
        rng.next4(u[0], u[1], u[2], u[3]);
        for (uint32_t j = 0; j < 4 && i + j < count; ++j)
            out[i + j] = u[j];
    }
}

} // extern "C"
