// TEST INFRASTRUCTURE ONLY (see ko_math.h). CPU restatement of the Depth-Box VDB ray marcher:
// RayCaster::RenderDepthBoxVDB<T> for the Scene target
// and its inspection targets AABBs / Depths / PositionsAfterDepthSkip
// (src/ray_caster/ray_caster_dpbx_vdb.cu:44-437, 465-531) with VDBStack, HDDA3D and DepthDDA2D of
// include/ray_caster/dda.h, and of the un-accelerated RayCaster::RenderRAWVolume
// (src/ray_caster/ray_caster_raw_volume.cu:27-246).
//
// Texture hardware is restated in software, with these stated conventions (CUDA's filtering weights
// are quantised to 8 fractional bits, which nothing here can reproduce or check: "sub-1/256
// interpolation-weight parity unpinned"):
//   atlasTex    linear, non-normalised coordinates, normalised-float read (include/cuda/texture.h:93-98):
//               xB = x - 0.5, i = floor(xB), a = xB - i with full fp32 weights, float(value) / max(T),
//               combined as ((1-a) v0 + a v1) along x, then y, then z;
//   atlasDepTex point, element type (src/ray_caster/dpbx_vdb.cu:158-167): voxel at floor(coordinate);
//   tfTex       1-D float4, normalised coordinates, linear, border 0 (app/dvr_dpbx_vdb.cpp:257-266):
//               x = u * 256 - 0.5, i = floor(x), a = x - i, (1-a) tf[i] + a tf[i+1], 0 outside [0,255].
#include "ko_vdb.h"

#include <cstring>

using namespace ko;

namespace {

constexpr float Eps = .01f; // include/ray_caster/dda.h:11

struct RenderParam {
    mat4 w2s, invProj;
    vec3 eyePos;
    mat3 eyeRot;
    float maxAlpha, samplingRate;
    bool useShading;
    float ka, kd, ks, shiness;
    vec3 lightPos, lightCol;
};

inline int sgn(float d) { return d > 0.f ? 1 : d < 0.f ? -1 : 0; }

// include/ray_caster/dda.h:35-82
struct HDDA3D {
    int sign[3], mask[3], chPos[3];
    float tCurr, tNext;
    vec3 tSide, tDlt;
    vec3 pos, dir;

    static HDDA3D Create(float tCurr, const Ray &ray) {
        HDDA3D h;
        h.pos = ray.pos;
        h.dir = ray.dir;
        for (int k = 0; k < 3; ++k) {
            h.sign[k] = sgn(ray.dir[k]);
            h.mask[k] = 0;
            h.chPos[k] = 0;
        }
        h.tCurr = tCurr;
        h.tNext = 0.f;
        return h;
    }
    void Prepare(vec3 min, float voxPerChild) {
        tDlt = vabs(voxPerChild / dir);
        vec3 pFlt = (pos + tCurr * dir - min) / voxPerChild;
        vec3 s = {static_cast<float>(sign[0]), static_cast<float>(sign[1]), static_cast<float>(sign[2])};
        tSide = ((vfloor(pFlt) - pFlt + .5f) * s + .5f) * tDlt + tCurr;
        vec3 f = vfloor(pFlt);
        chPos[0] = static_cast<int>(f.x);
        chPos[1] = static_cast<int>(f.y);
        chPos[2] = static_cast<int>(f.z);
    }
    void Next() {
        mask[0] = static_cast<int>((tSide.x < tSide.y) & (tSide.x <= tSide.z));
        mask[1] = static_cast<int>((tSide.y < tSide.z) & (tSide.y <= tSide.x));
        mask[2] = static_cast<int>((tSide.z < tSide.x) & (tSide.z <= tSide.y));
        tNext = mask[0] ? tSide.x : mask[1] ? tSide.y : mask[2] ? tSide.z : INFINITY;
    }
    void Step() {
        tCurr = tNext;
        tSide.x = std::isinf(tDlt.x) ? INFINITY : mask[0] ? tSide.x + tDlt.x : tSide.x;
        tSide.y = std::isinf(tDlt.y) ? INFINITY : mask[1] ? tSide.y + tDlt.y : tSide.y;
        tSide.z = std::isinf(tDlt.z) ? INFINITY : mask[2] ? tSide.z + tDlt.z : tSide.z;
        for (int k = 0; k < 3; ++k)
            chPos[k] += mask[k] * sign[k];
    }
};

// include/ray_caster/dda.h:84-172
struct DepthDDA2D {
    int sign[3], mask[3], posInBrick[3];
    float t, tStart, dep, tDlt2Dep;
    vec3 tSide, tDlt;

    bool Init(float t_, float maxPosValInBrick, int minDepPosValInBrick, int maxDepPosValInBrick,
              vec3 pib, const Ray &ray) {
        dep = 0.f;
        tDlt2Dep = 0.f;
        for (int k = 0; k < 3; ++k) {
            sign[k] = sgn(ray.dir[k]);
            mask[k] = 0;
        }
        t = tStart = t_;
        int depSign[3];
        {
            vec3 d;
            for (int k = 0; k < 3; ++k)
                d[k] = sign[k] == 0 ? INFINITY : sign[k] > 0 ? pib[k] : maxPosValInBrick - pib[k];
            depSign[0] = d.x < d.y && d.x <= d.z ? sign[0] : 0;
            depSign[1] = d.y < d.z && d.y <= d.x ? sign[1] : 0;
            depSign[2] = d.z < d.x && d.z <= d.y ? sign[2] : 0;
            if (depSign[0] != 0 && d.x >= .5f)
                return false;
            if (depSign[1] != 0 && d.y >= .5f)
                return false;
            if (depSign[2] != 0 && d.z >= .5f)
                return false;
        }
        tDlt = vabs(1.f / ray.dir);
        vec3 f = vfloor(pib);
        posInBrick[0] = static_cast<int>(f.x);
        posInBrick[1] = static_cast<int>(f.y);
        posInBrick[2] = static_cast<int>(f.z);
        vec3 s = {static_cast<float>(sign[0]), static_cast<float>(sign[1]), static_cast<float>(sign[2])};
        tSide = ((vfloor(pib) - pib + .5f) * s + .5f) * tDlt + t;
        for (int k = 0; k < 3; ++k)
            if (depSign[k] != 0) {
                posInBrick[k] = depSign[k] == 1 ? minDepPosValInBrick : maxDepPosValInBrick;
                sign[k] = 0;
                tSide[k] = INFINITY;
                tDlt2Dep = std::fabs(ray.dir[k]);
            }
        return (depSign[0] | depSign[1] | depSign[2]) != 0;
    }
    void StepNext() {
        mask[0] = static_cast<int>((tSide.x < tSide.y) & (tSide.x <= tSide.z));
        mask[1] = static_cast<int>((tSide.y < tSide.z) & (tSide.y <= tSide.x));
        mask[2] = static_cast<int>((tSide.z < tSide.x) & (tSide.z <= tSide.y));
        t = mask[0] ? tSide.x : mask[1] ? tSide.y : mask[2] ? tSide.z : INFINITY;
        dep = tDlt2Dep * (t - tStart);
        tSide.x = std::isinf(tDlt.x) ? INFINITY : mask[0] ? tSide.x + tDlt.x : tSide.x;
        tSide.y = std::isinf(tDlt.y) ? INFINITY : mask[1] ? tSide.y + tDlt.y : tSide.y;
        tSide.z = std::isinf(tDlt.z) ? INFINITY : mask[2] ? tSide.z + tDlt.z : tSide.z;
        for (int k = 0; k < 3; ++k)
            posInBrick[k] += mask[k] * sign[k];
    }
};

template <typename T> struct Marcher {
    const VDB<T> &vdb;
    const float *tf; // 256 x rgba
    const RenderParam &rp;
    uint64_t samples = 0;

    // voxel of the atlas at global atlas coordinate (gx, gy, gz); 0 outside (border)
    T atlasAt(int gx, int gy, int gz) const {
        const VDBParameter &p = vdb.param;
        const int A = p.voxPerAtlasBrick;
        if (gx < 0 || gy < 0 || gz < 0 || gx >= A * p.atlasBrickPerVDB[0] ||
            gy >= A * p.atlasBrickPerVDB[1] || gz >= A * p.atlasBrickPerVDB[2])
            return 0;
        int bx = gx / A, by = gy / A, bz = gz / A;
        uint32_t k = vdb.atlasBrickToNodeIndices[(static_cast<size_t>(bz) * p.atlasBrickPerVDB[1] + by) *
                                                     p.atlasBrickPerVDB[0] + bx];
        if (k == UndefRltIdx)
            return 0; // atlas brick that holds no leaf: never sampled by a ray inside a valid brick
        size_t off = static_cast<size_t>(k) * A * A * A +
                     (static_cast<size_t>(gz - bz * A) * A + (gy - by * A)) * A + (gx - bx * A);
        return vdb.atlas[off];
    }
    float atlasLinear(vec3 sp) const {
        constexpr float maxT = static_cast<float>(std::numeric_limits<T>::max());
        vec3 b = sp - .5f;
        vec3 f = vfloor(b);
        int i = static_cast<int>(f.x), j = static_cast<int>(f.y), k = static_cast<int>(f.z);
        float ax = b.x - f.x, ay = b.y - f.y, az = b.z - f.z;
        float v[2][2][2];
        for (int dz = 0; dz < 2; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx)
                    v[dz][dy][dx] = static_cast<float>(atlasAt(i + dx, j + dy, k + dz)) / maxT;
        float c00 = (1.f - ax) * v[0][0][0] + ax * v[0][0][1];
        float c01 = (1.f - ax) * v[0][1][0] + ax * v[0][1][1];
        float c10 = (1.f - ax) * v[1][0][0] + ax * v[1][0][1];
        float c11 = (1.f - ax) * v[1][1][0] + ax * v[1][1][1];
        float c0 = (1.f - ay) * c00 + ay * c01;
        float c1 = (1.f - ay) * c10 + ay * c11;
        return (1.f - az) * c0 + az * c1;
    }
    T atlasPoint(float x, float y, float z) const {
        return atlasAt(static_cast<int>(std::floor(x)), static_cast<int>(std::floor(y)),
                       static_cast<int>(std::floor(z)));
    }
    vec4 tfLinear(float u) const {
        float x = u * 256.f - .5f;
        float fl = std::floor(x);
        int i = static_cast<int>(fl);
        float a = x - fl;
        vec4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
        if (i >= 0 && i <= 255)
            t0 = {tf[4 * i], tf[4 * i + 1], tf[4 * i + 2], tf[4 * i + 3]};
        if (i + 1 >= 0 && i + 1 <= 255)
            t1 = {tf[4 * i + 4], tf[4 * i + 5], tf[4 * i + 6], tf[4 * i + 7]};
        return (1.f - a) * t0 + a * t1;
    }

    struct InLeaf {
        float tCurr, tExit;
        Node leaf;
    };

    // ray_caster_dpbx_vdb.cu:139-162
    bool depthSkip(InLeaf &param, vec3 posInBrick, vec3 atlasBrickMinVoxPos, const Ray &eyeRay) const {
        const VDBParameter &p = vdb.param;
        DepthDDA2D d;
        if (!d.Init(param.tCurr, p.voxsPerChild[1], p.minDepPosValInBrick, p.maxDepPosValInBrick,
                    posInBrick, eyeRay))
            return false;
        while (true) {
            T dep = atlasPoint(atlasBrickMinVoxPos.x + d.posInBrick[0],
                               atlasBrickMinVoxPos.y + d.posInBrick[1],
                               atlasBrickMinVoxPos.z + d.posInBrick[2]);
            if (dep <= d.dep)
                break;
            if (d.t >= param.tExit)
                return true;
            param.tCurr = d.t;
            d.StepNext();
        }
        return false;
    }

    // ray_caster_dpbx_vdb.cu:164-282 around rayCastVDB :44-137
    void renderScene(Ray eyeRay, uint8_t out[4]) {
        const VDBParameter &p = vdb.param;
        const bool useDepthBox = p.apronDepWid != p.apronWid;
        vec3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;
        float step = 0.f;
        uint32_t maxStepCnt = 0;
        vec3 dltPos = {0.f, 0.f, 0.f};
        const vec3 voxPerVol = {p.voxPerVol[0], p.voxPerVol[1], p.voxPerVol[2]};

        auto finish = [&]() {
            rgb = rgb / (rgb + 1.f);
            for (int k = 0; k < 3; ++k)
                rgb[k] = std::pow(rgb[k], 1.f / 2.2f);
            out[0] = static_cast<uint8_t>(255.f * rgb.x);
            out[1] = static_cast<uint8_t>(255.f * rgb.y);
            out[2] = static_cast<uint8_t>(255.f * rgb.z);
            out[3] = static_cast<uint8_t>(255.f * a);
        };

        auto inLeaf = [&](InLeaf &param) -> bool {
            param.tCurr = step * std::ceil(param.tCurr / step);
            vec3 brickMinVoxPos = {static_cast<float>(param.leaf.pos[0]),
                                   static_cast<float>(param.leaf.pos[1]),
                                   static_cast<float>(param.leaf.pos[2])};
            vec3 posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
            vec3 atlasBrickMinVoxPos = {
                static_cast<float>(param.leaf.brickPosInAtlas[0] * p.voxPerAtlasBrick + p.apronDepWid),
                static_cast<float>(param.leaf.brickPosInAtlas[1] * p.voxPerAtlasBrick + p.apronDepWid),
                static_cast<float>(param.leaf.brickPosInAtlas[2] * p.voxPerAtlasBrick + p.apronDepWid)};
            if (useDepthBox) {
                if (depthSkip(param, posInBrick, atlasBrickMinVoxPos, eyeRay))
                    return true;
                param.tCurr = step * std::ceil(param.tCurr / step);
                posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
            }
            while (param.tCurr < param.tExit && [&]() {
                for (int k = 0; k < 3; ++k)
                    if (posInBrick[k] < 0.f || posInBrick[k] >= p.voxsPerChild[1])
                        return false;
                return true;
            }()) {
                vec3 samplePos = atlasBrickMinVoxPos + posInBrick;
                float scalar = atlasLinear(samplePos);
                vec4 tfCol = tfLinear(scalar);
                vec3 tfColRGB = {tfCol.x, tfCol.y, tfCol.z};
                ++samples;
                if (rp.useShading) {
                    vec3 N;
                    for (int k = 0; k < 3; ++k) {
                        samplePos[k] += .5f;
                        float v0 = atlasLinear(samplePos);
                        samplePos[k] -= 1.f;
                        float v1 = atlasLinear(samplePos);
                        N[k] = v1 - v0;
                    }
                    N = normalize(N);
                    if (dot(eyeRay.dir, N) > 0.f)
                        N = -N;
                    vec3 p2lDir = normalize(rp.lightPos - brickMinVoxPos - posInBrick);
                    vec3 ambient = rp.ka * tfColRGB;
                    vec3 diffuse = rp.kd * gmax(0.f, dot(N, p2lDir)) * tfColRGB * rp.lightCol;
                    vec3 hfDir = normalize(p2lDir - eyeRay.dir);
                    vec3 specular = rp.ks * std::pow(gmax(0.f, dot(N, hfDir)), rp.shiness) * rp.lightCol;
                    for (int k = 0; k < 3; ++k)
                        tfColRGB[k] = ambient[k] + diffuse[k] + specular[k];
                }
                rgb = rgb + (1.f - a) * tfCol.w * tfColRGB;
                a = a + (1.f - a) * tfCol.w;
                if (a >= rp.maxAlpha)
                    return false;
                param.tCurr += step;
                posInBrick += dltPos;
            }
            return true;
        };

        // ---- rayCastVDB ----
        AABB unit{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
        HitShell hit = HitAABB(eyeRay, unit);
        if (hit.tEnter >= hit.tExit)
            return finish();
        {
            float tDlt = hit.tExit - hit.tEnter;
            vec3 d = tDlt * eyeRay.dir;
            maxStepCnt = static_cast<uint32_t>(
                std::ceil(rp.samplingRate * length(d) * gmax(voxPerVol.x, gmax(voxPerVol.y, voxPerVol.z))));
        }
        eyeRay.pos *= voxPerVol;
        eyeRay.dir = normalize(eyeRay.dir * voxPerVol);
        hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, voxPerVol});
        if (hit.tEnter >= hit.tExit)
            return finish();
        {
            float tDlt = hit.tExit - hit.tEnter;
            step = tDlt / (maxStepCnt - 1);
            dltPos = step * eyeRay.dir;
        }
        hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, {p.voxPerVDB, p.voxPerVDB, p.voxPerVDB}});
        if (hit.tEnter >= hit.tExit)
            return finish();

        // VDBStack (dda.h:13-33)
        int lev = p.rootLev + 1;
        float tExits[MaxLevelNum - 1] = {0.f, 0.f};
        Node nodes[MaxLevelNum - 1] = {};
        auto push = [&](uint32_t rel, float tExit) {
            --lev;
            nodes[lev - 1] = vdb.GetNode(static_cast<uint8_t>(lev), rel);
            tExits[lev - 1] = tExit;
        };
        auto empty = [&]() { return lev == p.rootLev + 1; };
        auto nodePos = [&]() {
            return vec3{static_cast<float>(nodes[lev - 1].pos[0]), static_cast<float>(nodes[lev - 1].pos[1]),
                        static_cast<float>(nodes[lev - 1].pos[2])};
        };

        push(0, hit.tExit - Eps);
        HDDA3D h = HDDA3D::Create(hit.tEnter + Eps, eyeRay);
        h.Prepare({0.f, 0.f, 0.f}, p.voxsPerChild[p.rootLev]);

        while (!empty() && [&]() {
            for (int k = 0; k < 3; ++k)
                if (h.chPos[k] < 0 || h.chPos[k] >= p.dims[lev])
                    return false;
            return true;
        }()) {
            h.Next();
            const Node &curr = nodes[lev - 1];
            uint32_t ch = vdb.GetChild(static_cast<uint8_t>(lev), curr.childPoolOffsRelative, h.chPos);
            if (ch != UndefRltIdx) {
                if (lev == 1) {
                    h.tCurr += Eps;
                    InLeaf param{h.tCurr, h.tNext - Eps, vdb.GetNode(0, ch)};
                    if (!inLeaf(param))
                        break;
                    h.Step();
                } else {
                    push(ch, h.tNext - Eps);
                    h.tCurr += Eps;
                    h.Prepare(nodePos(), p.voxsPerChild[lev]);
                }
            } else
                h.Step();

            while (h.tCurr >= tExits[lev - 1]) {
                ++lev;
                if (empty())
                    break;
                h.Prepare(nodePos(), p.voxsPerChild[lev]);
            }
        }
        finish();
    }

    // ---- the inspection targets, ray_caster_dpbx_vdb.cu:284-437 -------------------------------------
    // static_cast<uint8_t>(float) as the device does it: toward zero into 32 unsigned bits with
    // saturation (negative and NaN -> 0), then the low byte.
    static uint8_t ToU8(float v) {
        uint32_t u = v > 0.f ? (v < 4294967040.f ? static_cast<uint32_t>(v) : 0xFFFFFFFFu) : 0u;
        return static_cast<uint8_t>(u);
    }
    static vec3 toVec3(const int32_t p[3]) {
        return {static_cast<float>(p[0]), static_cast<float>(p[1]), static_cast<float>(p[2])};
    }

    // rayCastVDB :44-137 with its callbacks as std::function-free template parameters; a callback the
    // reference passes as nullptr is a no-op lambda here.
    template <bool Cull, typename ScnF, typename VolF, typename PushedF, typename LeafF>
    void rayCast(Ray &eyeRay, ScnF scn, VolF vol, PushedF pushed, LeafF inLeaf) const {
        const VDBParameter &p = vdb.param;
        const vec3 voxPerVol = {p.voxPerVol[0], p.voxPerVol[1], p.voxPerVol[2]};
        HitShell hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}});
        if (Cull && hit.tEnter >= hit.tExit)
            return;
        scn(hit.tEnter, hit.tExit);
        eyeRay.pos *= voxPerVol;
        eyeRay.dir = normalize(eyeRay.dir * voxPerVol);
        hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, voxPerVol});
        if (Cull && hit.tEnter >= hit.tExit)
            return;
        vol(hit.tEnter, hit.tExit);
        hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, {p.voxPerVDB, p.voxPerVDB, p.voxPerVDB}});
        if (hit.tEnter >= hit.tExit)
            return;

        int lev = p.rootLev + 1;
        float tExits[MaxLevelNum - 1] = {0.f, 0.f};
        Node nodes[MaxLevelNum - 1] = {};
        auto push = [&](uint32_t rel, float tExit) {
            --lev;
            nodes[lev - 1] = vdb.GetNode(static_cast<uint8_t>(lev), rel);
            tExits[lev - 1] = tExit;
        };
        push(0, hit.tExit - Eps);
        HDDA3D h = HDDA3D::Create(hit.tEnter + Eps, eyeRay);
        h.Prepare({0.f, 0.f, 0.f}, p.voxsPerChild[p.rootLev]);
        pushed(h.tCurr, lev, nodes[lev - 1]);

        while (lev != p.rootLev + 1 && [&]() {
            for (int k = 0; k < 3; ++k)
                if (h.chPos[k] < 0 || h.chPos[k] >= p.dims[lev])
                    return false;
            return true;
        }()) {
            h.Next();
            uint32_t ch = vdb.GetChild(static_cast<uint8_t>(lev), nodes[lev - 1].childPoolOffsRelative, h.chPos);
            if (ch != UndefRltIdx) {
                if (lev == 1) {
                    h.tCurr += Eps;
                    InLeaf param{h.tCurr, h.tNext - Eps, vdb.GetNode(0, ch)};
                    if (!inLeaf(param))
                        break;
                    h.Step();
                } else {
                    push(ch, h.tNext - Eps);
                    h.tCurr += Eps;
                    h.Prepare(toVec3(nodes[lev - 1].pos), p.voxsPerChild[lev]);
                    pushed(h.tCurr, lev, nodes[lev - 1]);
                }
            } else
                h.Step();
            while (h.tCurr >= tExits[lev - 1]) {
                ++lev;
                if (lev == p.rootLev + 1)
                    break;
                h.Prepare(toVec3(nodes[lev - 1].pos), p.voxsPerChild[lev]);
            }
        }
    }

    vec3 atlasBrickMin(const Node &leaf) const {
        const VDBParameter &p = vdb.param;
        return {static_cast<float>(leaf.brickPosInAtlas[0] * p.voxPerAtlasBrick + p.apronDepWid),
                static_cast<float>(leaf.brickPosInAtlas[1] * p.voxPerAtlasBrick + p.apronDepWid),
                static_cast<float>(leaf.brickPosInAtlas[2] * p.voxPerAtlasBrick + p.apronDepWid)};
    }
    static void store(vec3 rgb, float a, uint8_t out[4]) {
        out[0] = ToU8(255.f * rgb.x);
        out[1] = ToU8(255.f * rgb.y);
        out[2] = ToU8(255.f * rgb.z);
        out[3] = ToU8(255.f * a);
    }

    // renderAABBs :284-323: every node of level displayLev the ray enters adds its entry point,
    // normalised to the node, at half opacity; the box tests of scene and volume do not cull.
    void renderAABBs(int displayLev, Ray eyeRay, uint8_t out[4]) const {
        const VDBParameter &p = vdb.param;
        constexpr float AlphaPerLev = .5f;
        vec3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;
        auto nop = [](float, float) {};
        rayCast<false>(
            eyeRay, nop, nop,
            [&](float tCurr, int lev, const Node &node) {
                if (lev == displayLev) {
                    vec3 pos = (eyeRay.pos + tCurr * eyeRay.dir - toVec3(node.pos)) /
                               (lev == p.rootLev ? p.voxPerVDB : p.voxsPerChild[lev + 1]);
                    rgb = rgb + (1.f - a) * AlphaPerLev * pos;
                    a = a + (1.f - a) * AlphaPerLev;
                }
            },
            [&](InLeaf &param) {
                if (displayLev == 0) {
                    vec3 posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - toVec3(param.leaf.pos);
                    rgb = rgb + (1.f - a) * AlphaPerLev * (posInBrick / p.voxsPerChild[1]);
                    a = a + (1.f - a) * AlphaPerLev;
                }
                return true;
            });
        store(rgb, a, out);
    }

    // renderDepths :325-376: the displayLeafLayer-th leaf along the ray shows the depth value its
    // entry face holds (grey), or red where the ray does not enter through a face.
    void renderDepths(int displayLeafLayer, Ray eyeRay, uint8_t out[4]) const {
        const VDBParameter &p = vdb.param;
        vec3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;
        int layer = 0;
        auto nop = [](float, float) {};
        rayCast<true>(
            eyeRay, nop, nop, [](float, int, const Node &) {},
            [&](InLeaf &param) {
                if (layer != displayLeafLayer) {
                    layer = (layer + 1) & 0xFF; // a uint8_t counter in the reference
                    return true;
                }
                vec3 posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - toVec3(param.leaf.pos);
                vec3 abm = atlasBrickMin(param.leaf);
                DepthDDA2D d;
                if (d.Init(param.tCurr, p.voxsPerChild[1], p.minDepPosValInBrick, p.maxDepPosValInBrick,
                           posInBrick, eyeRay)) {
                    T dep = atlasPoint(abm.x + d.posInBrick[0], abm.y + d.posInBrick[1], abm.z + d.posInBrick[2]);
                    float depNorm = dep / p.voxsPerChild[1];
                    rgb = {depNorm, depNorm, depNorm};
                } else
                    rgb = {1.f, 0.f, 0.f};
                a = 1.f;
                return false;
            });
        store(rgb, a, out);
    }

    // renderPositionsAfterDepthSkip :378-437: where sampling would start in the displayLeafLayer-th
    // leaf after the depth skip, normalised to the brick; a leaf skipped whole passes the ray on
    // without counting as a layer.
    void renderPositionsAfterDepthSkip(int displayLeafLayer, Ray eyeRay, uint8_t out[4]) const {
        const VDBParameter &p = vdb.param;
        const vec3 voxPerVol = {p.voxPerVol[0], p.voxPerVol[1], p.voxPerVol[2]};
        vec3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;
        float step = 0.f;
        uint32_t maxStepCnt = 0;
        int layer = 0;
        rayCast<true>(
            eyeRay,
            [&](float tEnter, float tExit) {
                vec3 d = (tExit - tEnter) * eyeRay.dir;
                maxStepCnt = static_cast<uint32_t>(std::ceil(
                    rp.samplingRate * length(d) * gmax(voxPerVol.x, gmax(voxPerVol.y, voxPerVol.z))));
            },
            [&](float tEnter, float tExit) { step = (tExit - tEnter) / (maxStepCnt - 1); },
            [](float, int, const Node &) {},
            [&](InLeaf &param) {
                if (layer != displayLeafLayer) {
                    layer = (layer + 1) & 0xFF;
                    return true;
                }
                param.tCurr = step * std::ceil(param.tCurr / step);
                vec3 brickMinVoxPos = toVec3(param.leaf.pos);
                vec3 posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
                if (depthSkip(param, posInBrick, atlasBrickMin(param.leaf), eyeRay))
                    return true;
                param.tCurr = step * std::ceil(param.tCurr / step);
                posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
                rgb = posInBrick / p.voxsPerChild[1];
                a = 1.f;
                return false;
            });
        store(rgb, a, out);
    }
};

template <typename T>
void renderDVR(const VDB<T> &vdb, const float *tf, const RenderParam &rp, uint8_t *rgba, int W, int H,
               int y0, int y1, uint64_t *samplesOut) {
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int y = y0; y < y1; ++y) {
        Marcher<T> m{vdb, tf, rp};
        for (int x = 0; x < W; ++x) {
            // ray_caster_dpbx_vdb.cu:475-501
            Ray eyeRay;
            vec4 tmp;
            tmp.z = static_cast<float>(W);
            tmp.w = static_cast<float>(H);
            tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
            tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
            tmp.z = 1.f;
            tmp.w = 1.f;
            tmp = rp.invProj * tmp;
            eyeRay.dir = {tmp.x, tmp.y, tmp.z};
            eyeRay.dir = rp.eyeRot * normalize(eyeRay.dir);
            tmp = {eyeRay.dir.x, eyeRay.dir.y, eyeRay.dir.z, 0.f};
            tmp = rp.w2s * tmp;
            eyeRay.dir = {tmp.x, tmp.y, tmp.z};
            eyeRay.dir = normalize(eyeRay.dir);
            tmp = {rp.eyePos.x, rp.eyePos.y, rp.eyePos.z, 1.f};
            tmp = rp.w2s * tmp;
            eyeRay.pos = {tmp.x, tmp.y, tmp.z};
            m.renderScene(eyeRay, rgba + 4 * (static_cast<size_t>(y) * W + x));
        }
        total += m.samples;
    }
    if (samplesOut)
        *samplesOut = total;
}


// ray_caster_dpbx_vdb.cu:475-501
inline Ray makeEyeRay(const RenderParam &rp, int x, int y, int W, int H) {
    Ray eyeRay;
    vec4 tmp;
    tmp.z = static_cast<float>(W);
    tmp.w = static_cast<float>(H);
    tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
    tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
    tmp.z = 1.f;
    tmp.w = 1.f;
    tmp = rp.invProj * tmp;
    eyeRay.dir = {tmp.x, tmp.y, tmp.z};
    eyeRay.dir = rp.eyeRot * normalize(eyeRay.dir);
    tmp = {eyeRay.dir.x, eyeRay.dir.y, eyeRay.dir.z, 0.f};
    tmp = rp.w2s * tmp;
    eyeRay.dir = {tmp.x, tmp.y, tmp.z};
    eyeRay.dir = normalize(eyeRay.dir);
    tmp = {rp.eyePos.x, rp.eyePos.y, rp.eyePos.z, 1.f};
    tmp = rp.w2s * tmp;
    eyeRay.pos = {tmp.x, tmp.y, tmp.z};
    return eyeRay;
}

// targets 1 = AABBs, 2 = Depths, 3 = PositionsAfterDepthSkip (DPBXVDBRenderTarget, ray_caster.h:85)
template <typename T>
void renderVDBTarget(const VDB<T> &vdb, const RenderParam &rp, int target, int displayLev,
                     int displayLeafLayer, uint8_t *rgba, int W, int H, int y0, int y1) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = y0; y < y1; ++y) {
        Marcher<T> m{vdb, nullptr, rp};
        for (int x = 0; x < W; ++x) {
            Ray eyeRay = makeEyeRay(rp, x, y, W, H);
            uint8_t *out = rgba + 4 * (static_cast<size_t>(y) * W + x);
            if (target == 1)
                m.renderAABBs(displayLev, eyeRay, out);
            else if (target == 2)
                m.renderDepths(displayLeafLayer, eyeRay, out);
            else
                m.renderPositionsAfterDepthSkip(displayLeafLayer, eyeRay, out);
        }
    }
}

// ---- RayCaster::RenderRAWVolume (src/ray_caster/ray_caster_raw_volume.cu:27-246) ---------------------
// volTex: linear, non-normalised coordinates, border addressing, normalised-float read
// (include/cuda/texture.h:93-98), restated like atlasTex above with 0 outside the volume.
template <typename T> struct RawVolume {
    const T *vol;
    int dx, dy, dz;
    const float *tf;
    float maxT;
    uint64_t samples = 0;

    float linear(vec3 sp) const {
        vec3 b = sp - .5f;
        vec3 f = {std::floor(b.x), std::floor(b.y), std::floor(b.z)};
        int i = static_cast<int>(f.x), j = static_cast<int>(f.y), k = static_cast<int>(f.z);
        float ax = b.x - f.x, ay = b.y - f.y, az = b.z - f.z;
        float v[2][2][2];
        for (int zz = 0; zz < 2; ++zz)
            for (int yy = 0; yy < 2; ++yy)
                for (int xx = 0; xx < 2; ++xx) {
                    int x = i + xx, y = j + yy, z = k + zz;
                    T t = 0;
                    if (x >= 0 && y >= 0 && z >= 0 && x < dx && y < dy && z < dz)
                        t = vol[(static_cast<size_t>(z) * dy + y) * dx + x];
                    v[zz][yy][xx] = static_cast<float>(t) / maxT;
                }
        float c00 = (1.f - ax) * v[0][0][0] + ax * v[0][0][1];
        float c01 = (1.f - ax) * v[0][1][0] + ax * v[0][1][1];
        float c10 = (1.f - ax) * v[1][0][0] + ax * v[1][0][1];
        float c11 = (1.f - ax) * v[1][1][0] + ax * v[1][1][1];
        float c0 = (1.f - ay) * c00 + ay * c01;
        float c1 = (1.f - ay) * c10 + ay * c11;
        return (1.f - az) * c0 + az * c1;
    }
    vec4 tfLinear(float u) const {
        float x = u * 256.f - .5f;
        float fl = std::floor(x);
        int i = static_cast<int>(fl);
        float a = x - fl;
        vec4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
        if (i >= 0 && i <= 255)
            t0 = {tf[4 * i], tf[4 * i + 1], tf[4 * i + 2], tf[4 * i + 3]};
        if (i + 1 >= 0 && i + 1 <= 255)
            t1 = {tf[4 * i + 4], tf[4 * i + 5], tf[4 * i + 6], tf[4 * i + 7]};
        return (1.f - a) * t0 + a * t1;
    }

    // renderScene<UseShading>, :27-129, and renderAABB, :131-145
    void render(const RenderParam &rp, int target, Ray eyeRay, uint8_t out[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        HitShell hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}});
        if (hit.tEnter > hit.tExit)
            return;
        if (target == 1) {
            vec3 enterPos = eyeRay.pos + hit.tEnter * eyeRay.dir;
            vec3 exitPos = eyeRay.pos + hit.tExit * eyeRay.dir;
            vec3 color = .3f * enterPos + .7f * exitPos;
            out[0] = static_cast<uint8_t>(255.f * color.x);
            out[1] = static_cast<uint8_t>(255.f * color.y);
            out[2] = static_cast<uint8_t>(255.f * color.z);
            out[3] = 255;
            return;
        }
        const vec3 dim = {static_cast<float>(dx), static_cast<float>(dy), static_cast<float>(dz)};
        uint32_t stepCnt = 0;
        vec3 d0 = (hit.tExit - hit.tEnter) * eyeRay.dir;
        uint32_t maxStepCnt = static_cast<uint32_t>(
            std::ceil(rp.samplingRate * length(d0) * gmax(dim.x, gmax(dim.y, dim.z))));
        eyeRay.pos = eyeRay.pos * dim;
        eyeRay.dir = normalize(eyeRay.dir * dim);
        hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, dim});
        if (hit.tEnter > hit.tExit)
            return;
        float step = (hit.tExit - hit.tEnter) / (maxStepCnt - 1);
        auto inBox = [&]() {
            for (int k = 0; k < 3; ++k)
                if (eyeRay.pos[k] < 0.f || eyeRay.pos[k] >= dim[k])
                    return false;
            return true;
        };
        hit.tEnter = step * std::ceil(hit.tEnter / step);
        eyeRay.pos += hit.tEnter * eyeRay.dir;
        if (!inBox())
            eyeRay.pos += step * eyeRay.dir;
        vec3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;
        while (stepCnt <= maxStepCnt && inBox()) {
            float scalar = linear(eyeRay.pos);
            vec4 tfCol = tfLinear(scalar);
            vec3 tfColRGB = {tfCol.x, tfCol.y, tfCol.z};
            ++samples;
            if (rp.useShading) {
                vec3 N;
                vec3 sp = eyeRay.pos;
                for (int k = 0; k < 3; ++k) {
                    sp[k] += .5f;
                    float v0 = linear(sp);
                    sp[k] -= 1.f;
                    float v1 = linear(sp);
                    N[k] = v1 - v0;
                }
                N = normalize(N);
                if (dot(eyeRay.dir, N) > 0.f)
                    N = -N;
                vec3 p2lDir = normalize(rp.lightPos - eyeRay.pos);
                vec3 ambient = rp.ka * tfColRGB;
                vec3 diffuse = rp.kd * gmax(0.f, dot(N, p2lDir)) * tfColRGB * rp.lightCol;
                vec3 hfDir = normalize(p2lDir - eyeRay.dir);
                vec3 specular = rp.ks * std::pow(gmax(0.f, dot(N, hfDir)), rp.shiness) * rp.lightCol;
                for (int k = 0; k < 3; ++k)
                    tfColRGB[k] = ambient[k] + diffuse[k] + specular[k];
            }
            rgb = rgb + (1.f - a) * tfCol.w * tfColRGB;
            a = a + (1.f - a) * tfCol.w;
            if (a >= rp.maxAlpha)
                break;
            ++stepCnt;
            eyeRay.pos += step * eyeRay.dir;
        }
        rgb = rgb / (rgb + 1.f);
        for (int k = 0; k < 3; ++k)
            rgb[k] = std::pow(rgb[k], 1.f / 2.2f);
        out[0] = static_cast<uint8_t>(255.f * rgb.x);
        out[1] = static_cast<uint8_t>(255.f * rgb.y);
        out[2] = static_cast<uint8_t>(255.f * rgb.z);
        out[3] = static_cast<uint8_t>(255.f * a);
    }
};

template <typename T>
void renderRAW(const T *vol, int dx, int dy, int dz, const float *tf, const RenderParam &rp, int target,
               uint8_t *rgba, int W, int H, int y0, int y1, uint64_t *samplesOut) {
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int y = y0; y < y1; ++y) {
        RawVolume<T> rv{vol, dx, dy, dz, tf, sizeof(T) == 1 ? 255.f : 65535.f};
        for (int x = 0; x < W; ++x) {
            Ray eyeRay;
            vec4 tmp;
            tmp.z = static_cast<float>(W);
            tmp.w = static_cast<float>(H);
            tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
            tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
            tmp.z = 1.f;
            tmp.w = 1.f;
            tmp = rp.invProj * tmp;
            eyeRay.dir = {tmp.x, tmp.y, tmp.z};
            eyeRay.dir = rp.eyeRot * normalize(eyeRay.dir);
            tmp = {eyeRay.dir.x, eyeRay.dir.y, eyeRay.dir.z, 0.f};
            tmp = rp.w2s * tmp;
            eyeRay.dir = {tmp.x, tmp.y, tmp.z};
            eyeRay.dir = normalize(eyeRay.dir);
            tmp = {rp.eyePos.x, rp.eyePos.y, rp.eyePos.z, 1.f};
            tmp = rp.w2s * tmp;
            eyeRay.pos = {tmp.x, tmp.y, tmp.z};
            rv.render(rp, target, eyeRay, rgba + 4 * (static_cast<size_t>(y) * W + x));
        }
        total += rv.samples;
    }
    if (samplesOut)
        *samplesOut = total;
}

// The reference's own host-side self-check of a built VDB, test_rayCastVDB
// (include/ray_caster/test_options.h:88-242): six axis-parallel and four diagonal rays through the centre
// of the VDB's cube walk the pools with VDBStack / HDDA3D on the host; at every leaf met the brick's
// place in the atlas must map back (atlasBrickToNodeIndices) to a node at the same position.
// Returns per ray: leaves met, wrong reverse mappings, and the leaves' node indices (at most maxLeaves).
template <typename T>
static void selfTestRays(const VDB<T> &vdb, int32_t *leafCounts, int32_t *wrong, uint32_t *leafIdx,
                         int maxLeaves, float *raysOut) {
    const VDBParameter &p = vdb.param;
    Ray rays[10];
    for (int i = 0; i < 6; ++i) { // :112-121
        const int t0 = i / 2, t1 = i % 2;
        Ray &r = rays[i];
        r.dir.x = t0 != 0 ? 0.f : t1 == 0 ? 1.f : -1.f;
        r.dir.y = t0 != 1 ? 0.f : t1 == 0 ? 1.f : -1.f;
        r.dir.z = t0 != 2 ? 0.f : t1 == 0 ? 1.f : -1.f;
        r.pos = vec3{.5f * p.voxPerVDB, .5f * p.voxPerVDB, .5f * p.voxPerVDB} - r.dir * p.voxPerVDB;
    }
    for (int i = 0; i < 4; ++i) { // :122-132
        const int t0 = i & 1, t1 = i & 2;
        Ray &r = rays[6 + i];
        r.dir = normalize(vec3{t0 == 0 ? 1.f : -1.f, t1 == 0 ? 1.f : -1.f, 0.f});
        r.pos = vec3{.5f * p.voxPerVDB, .5f * p.voxPerVDB, .5f * p.voxPerVDB} - r.dir * p.voxPerVDB;
    }
    for (int ri = 0; ri < 10; ++ri) {
        const Ray &eyeRay = rays[ri];
        for (int k = 0; k < 3; ++k) {
            raysOut[6 * ri + k] = eyeRay.pos[k];
            raysOut[6 * ri + 3 + k] = eyeRay.dir[k];
        }
        leafCounts[ri] = wrong[ri] = 0;
        HitShell hit = HitAABB(eyeRay, AABB{{0.f, 0.f, 0.f}, {p.voxPerVDB, p.voxPerVDB, p.voxPerVDB}});
        if (hit.tEnter >= hit.tExit) { // :143-146 (the reference leaves the whole check here)
            leafCounts[ri] = -1;
            continue;
        }
        int lev = p.rootLev + 1; // VDBStack, dda.h:13-33
        float tExits[MaxLevelNum - 1] = {0.f, 0.f};
        Node nodes[MaxLevelNum - 1] = {};
        auto push = [&](uint32_t rel, float tExit) {
            --lev;
            nodes[lev - 1] = vdb.GetNode(static_cast<uint8_t>(lev), rel);
            tExits[lev - 1] = tExit;
        };
        auto nodePos = [&]() {
            return vec3{static_cast<float>(nodes[lev - 1].pos[0]), static_cast<float>(nodes[lev - 1].pos[1]),
                        static_cast<float>(nodes[lev - 1].pos[2])};
        };
        push(0, hit.tExit - Eps); // :166-170
        HDDA3D h = HDDA3D::Create(hit.tEnter + Eps, eyeRay);
        h.Prepare({0.f, 0.f, 0.f}, p.voxsPerChild[p.rootLev]);
        h.tCurr += Eps;
        while (lev != p.rootLev + 1 && [&]() {
            for (int k = 0; k < 3; ++k)
                if (h.chPos[k] < 0 || h.chPos[k] >= p.dims[lev])
                    return false;
            return true;
        }()) {
            h.Next();
            const Node &curr = nodes[lev - 1];
            const uint32_t ch = vdb.GetChild(static_cast<uint8_t>(lev), curr.childPoolOffsRelative, h.chPos);
            if (ch != UndefRltIdx) {
                if (lev == 1) { // :188-218
                    const Node leaf = vdb.GetNode(0, ch);
                    h.tCurr += Eps;
                    const size_t atlasBrickIdx =
                        (static_cast<size_t>(leaf.brickPosInAtlas[2]) * p.atlasBrickPerVDB[1] +
                         leaf.brickPosInAtlas[1]) * p.atlasBrickPerVDB[0] + leaf.brickPosInAtlas[0];
                    const uint32_t back = vdb.atlasBrickToNodeIndices[atlasBrickIdx];
                    const Node node = vdb.GetNode(0, back);
                    if (node.pos[0] != leaf.pos[0] || node.pos[1] != leaf.pos[1] || node.pos[2] != leaf.pos[2])
                        ++wrong[ri];
                    if (leafCounts[ri] < maxLeaves)
                        leafIdx[static_cast<size_t>(ri) * maxLeaves + leafCounts[ri]] = ch;
                    ++leafCounts[ri];
                    h.Step();
                } else { // :219-224
                    push(ch, h.tNext - Eps);
                    h.tCurr += Eps;
                    h.Prepare(nodePos(), p.voxsPerChild[lev]);
                }
            } else
                h.Step();
            while (h.tCurr >= tExits[lev - 1]) { // :232-239
                ++lev;
                if (lev == p.rootLev + 1)
                    break;
                h.Prepare(nodePos(), p.voxsPerChild[lev]);
            }
        }
    }
}

} // namespace

extern "C" {

void ko_vdb_self_test_rays(const void *vdbHandle, int is16, int32_t *leafCounts, int32_t *wrong,
                           uint32_t *leafIdx, int maxLeaves, float *raysOut) {
    if (is16)
        selfTestRays(*static_cast<const VDB<uint16_t> *>(vdbHandle), leafCounts, wrong, leafIdx, maxLeaves, raysOut);
    else
        selfTestRays(*static_cast<const VDB<uint8_t> *>(vdbHandle), leafCounts, wrong, leafIdx, maxLeaves, raysOut);
}

struct ko_dvr_params {
    float w2s[16], invProj[16];
    float eyePos[3], eyeRot[9];
    float maxAlpha, samplingRate;
    int32_t useShading;
    float ka, kd, ks, shiness;
    float lightPos[3], lightCol[3];
};

static RenderParam toRenderParam(const ko_dvr_params *pr) {
    RenderParam rp;
    std::memcpy(&rp.w2s, pr->w2s, 64);
    std::memcpy(&rp.invProj, pr->invProj, 64);
    rp.eyePos = {pr->eyePos[0], pr->eyePos[1], pr->eyePos[2]};
    std::memcpy(&rp.eyeRot, pr->eyeRot, 36);
    rp.maxAlpha = pr->maxAlpha;
    rp.samplingRate = pr->samplingRate;
    rp.useShading = pr->useShading != 0;
    rp.ka = pr->ka;
    rp.kd = pr->kd;
    rp.ks = pr->ks;
    rp.shiness = pr->shiness;
    rp.lightPos = {pr->lightPos[0], pr->lightPos[1], pr->lightPos[2]};
    rp.lightCol = {pr->lightCol[0], pr->lightCol[1], pr->lightCol[2]};
    return rp;
}

// The inspection targets of RenderDepthBoxVDB: 1 = AABBs, 2 = Depths, 3 = PositionsAfterDepthSkip.
void ko_dvr_render_target(const void *vdbHandle, int is16, const ko_dvr_params *pr, int target,
                          int displayLev, int displayLeafLayer, uint8_t *rgba, int W, int H, int y0,
                          int y1) {
    RenderParam rp = toRenderParam(pr);
    if (is16)
        renderVDBTarget(*static_cast<const VDB<uint16_t> *>(vdbHandle), rp, target, displayLev,
                        displayLeafLayer, rgba, W, H, y0, y1);
    else
        renderVDBTarget(*static_cast<const VDB<uint8_t> *>(vdbHandle), rp, target, displayLev,
                        displayLeafLayer, rgba, W, H, y0, y1);
}

void ko_dvr_render(const void *vdbHandle, int is16, const float *tf256x4, const ko_dvr_params *pr,
                   uint8_t *rgba, int W, int H, int y0, int y1, uint64_t *samplesOut) {
    RenderParam rp = toRenderParam(pr);
    if (is16)
        renderDVR(*static_cast<const VDB<uint16_t> *>(vdbHandle), tf256x4, rp, rgba, W, H, y0, y1,
                  samplesOut);
    else
        renderDVR(*static_cast<const VDB<uint8_t> *>(vdbHandle), tf256x4, rp, rgba, W, H, y0, y1,
                  samplesOut);
}

// RenderRAWVolume over the dense volume T[dz][dy][dx]; target 0 = Scene, 1 = AABB.
void ko_raw_render(const void *vol, int is16, int dx, int dy, int dz, const float *tf256x4,
                   const ko_dvr_params *pr, int target, uint8_t *rgba, int W, int H, int y0, int y1,
                   uint64_t *samplesOut) {
    RenderParam rp;
    std::memcpy(&rp.w2s, pr->w2s, 64);
    std::memcpy(&rp.invProj, pr->invProj, 64);
    rp.eyePos = {pr->eyePos[0], pr->eyePos[1], pr->eyePos[2]};
    std::memcpy(&rp.eyeRot, pr->eyeRot, 36);
    rp.maxAlpha = pr->maxAlpha;
    rp.samplingRate = pr->samplingRate;
    rp.useShading = pr->useShading != 0;
    rp.ka = pr->ka;
    rp.kd = pr->kd;
    rp.ks = pr->ks;
    rp.shiness = pr->shiness;
    rp.lightPos = {pr->lightPos[0], pr->lightPos[1], pr->lightPos[2]};
    rp.lightCol = {pr->lightCol[0], pr->lightCol[1], pr->lightCol[2]};
    if (is16)
        renderRAW(static_cast<const uint16_t *>(vol), dx, dy, dz, tf256x4, rp, target, rgba, W, H, y0, y1,
                  samplesOut);
    else
        renderRAW(static_cast<const uint8_t *>(vol), dx, dy, dz, tf256x4, rp, target, rgba, W, H, y0, y1,
                  samplesOut);
}

} // extern "C"
