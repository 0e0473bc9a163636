// TEST INFRASTRUCTURE ONLY (see ko_math.h). CPU restatement of the reference's LBVH build and of its
// closest-hit traversal. Every function names the reference lines it follows
// (paths relative to the reference checkout).
//
// Parity status: the arithmetic primitives used here (glm min/max/dot/cross/normalize, ray-triangle)
// are pinned against the reference's vendored glm through oracle/_ref; the reference ships no golden
// vectors for the build or the traversal, so the remaining logic is pinned by the structural
// invariants of include/ray_tracer/test_options.h:23-96 and by known-answer values obtained from the
// reference's own code for veach-mis and the grid test meshes (tests/test_oracle_lbvh.py).
#include "ko_math.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

using namespace ko;

namespace {

// include/ray_tracer/lbvh.h:24-34
struct InternalNode {
    uint32_t lft, rht;
    AABB aabb;
};
static_assert(sizeof(InternalNode) == 32, "InternalNode must be 32 bytes");
constexpr uint32_t ChildIsLeafMask = 0x80000000u;
constexpr uint32_t LeafIndexMask = 0x7fffffffu;

inline vec3 LoadV3(const float *p, uint32_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// src/ray_tracer/lbvh.cu:158-164
inline uint32_t expand10BitsTo32(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// src/ray_tracer/lbvh.cu:221-232 (host branch: std::countl_zero; clz(0) = 32 on both branches)
inline int lftCommZeroLen(const uint32_t *mortons, int64_t i0, int64_t i1, int64_t inodeNum) {
    if (i0 < 0 || i0 > inodeNum || i1 < 0 || i1 > inodeNum)
        return -1;
    uint32_t x = mortons[i0] ^ mortons[i1];
    return x == 0 ? 32 : __builtin_clz(x);
}

// src/ray_tracer/lbvh.cu:29-98
void computeLBVHLinks(InternalNode *inodes, const uint32_t *mortons, int64_t inodeNum,
                      int64_t inodeIdx) {
    auto lzl = [&](int64_t a, int64_t b) { return lftCommZeroLen(mortons, a, b, inodeNum); };

    int d;
    {
        int commLenLft = lzl(inodeIdx, inodeIdx - 1);
        int commLenRht = lzl(inodeIdx, inodeIdx + 1);
        d = (commLenLft <= commLenRht) ? 1 : -1;
    }

    int lzlCmp = lzl(inodeIdx, inodeIdx - d);
    int64_t iOffsMax = 2;
    while (true) {
        int64_t tmpIdx = inodeIdx + d * iOffsMax;
        if (lzl(inodeIdx, tmpIdx) <= lzlCmp)
            break;
        iOffsMax <<= 1;
    }

    int64_t iOffs = 0;
    while (true) {
        iOffsMax >>= 1;
        if (iOffsMax < 1)
            break;
        int64_t tmpIdx = inodeIdx + d * (iOffsMax + iOffs);
        if (lzl(inodeIdx, tmpIdx) > lzlCmp)
            iOffs += iOffsMax;
    }
    int64_t iMax = inodeIdx + d * iOffs;
    lzlCmp = lzl(inodeIdx, iMax);

    iOffsMax = iOffs;
    iOffs = 0;
    while (true) {
        iOffsMax = (iOffsMax + 1) >> 1;
        if (iOffsMax < 1)
            break;
        int64_t tmpIdx = inodeIdx + d * (iOffsMax + iOffs);
        if (lzl(inodeIdx, tmpIdx) > lzlCmp)
            iOffs += iOffsMax;
        if (iOffsMax == 1)
            break;
    }
    int64_t iSplit = inodeIdx + d * iOffs + (d < 0 ? -1 : 0);

    InternalNode inode;
    inode.aabb = AABB::CreateInitial();
    inode.lft = static_cast<uint32_t>(iSplit);
    if (std::min(inodeIdx, iMax) == static_cast<int64_t>(inode.lft))
        inode.lft |= ChildIsLeafMask;
    inode.rht = static_cast<uint32_t>(iSplit + 1);
    if (std::max(inodeIdx, iMax) == static_cast<int64_t>(inode.rht))
        inode.rht |= ChildIsLeafMask;
    inodes[inodeIdx] = inode;
}

} // namespace

extern "C" {

// src/ray_tracer/lbvh.cu:100-323. Outputs are caller-allocated with capacity N (faces) each.
//   refitMode 0: the reference's per-node subtree walk (lbvh.cu:265-297), O(sum of subtree sizes)
//   refitMode 1: one post-order pass (same min/max unions, identical bits), for large inputs
// Returns 0, or -1 when N == 0 (the reference does not guard it; size() - 1 wraps).
int ko_lbvh_build(const float *positions, uint32_t V, const uint32_t *facePosIdx, uint32_t N,
                  float *rootAABB6, uint32_t *mortonsUnsorted, uint32_t *mortonsSorted,
                  uint32_t *faces, uint32_t *compactedFaces, uint32_t *uniqueMortons,
                  float *leafAABBs6, void *internalNodes32, int refitMode, uint32_t *compactNumOut) {
    if (N == 0 || V == 0)
        return -1;

    // lbvh.cu:112-124: bounds over ALL positions
    AABB root;
    {
        vec3 mn = V3(FloatMax), mx = V3(FloatMin);
        for (uint32_t i = 0; i < V; ++i) {
            vec3 p = LoadV3(positions, i);
            mn = {gmin(mn.x, p.x), gmin(mn.y, p.y), gmin(mn.z, p.z)};
            mx = {gmax(mx.x, p.x), gmax(mx.y, p.y), gmax(mx.z, p.z)};
        }
        root.minPos = mn;
        root.maxPos = mx;
    }
    std::memcpy(rootAABB6, &root, 24);

    // lbvh.cu:130-134
    vec3 invDlt = root.maxPos - root.minPos;
    for (int k = 0; k < 3; ++k)
        if (std::fabs(invDlt[k]) < FloatEps)
            invDlt[k] = 1.f;
    invDlt = 1.f / invDlt;

    // lbvh.cu:136-170
    std::vector<uint32_t> mortons(N);
#pragma omp parallel for schedule(static)
    for (int64_t fi = 0; fi < static_cast<int64_t>(N); ++fi) {
        AABB aabb = AABB::CreateInitial();
        for (int i = 0; i < 3; ++i)
            aabb.Expand(LoadV3(positions, facePosIdx[3 * fi + i]));
        vec3 c = .5f * (aabb.minPos + aabb.maxPos);
        c = (c - root.minPos) * invDlt;
        c.x = gmin(gmax(c.x * 1024.f, 0.f), 1023.f);
        c.y = gmin(gmax(c.y * 1024.f, 0.f), 1023.f);
        c.z = gmin(gmax(c.z * 1024.f, 0.f), 1023.f);
        uint32_t cx = expand10BitsTo32(static_cast<uint32_t>(c.x));
        uint32_t cy = expand10BitsTo32(static_cast<uint32_t>(c.y));
        uint32_t cz = expand10BitsTo32(static_cast<uint32_t>(c.z));
        mortons[fi] = cx * 4 + cy * 2 + cz;
    }
    if (mortonsUnsorted)
        std::memcpy(mortonsUnsorted, mortons.data(), 4ull * N);

    // lbvh.cu:110,172: faces = 0..N-1, stable sort_by_key
    std::vector<uint32_t> order(N);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return mortons[a] < mortons[b]; });
    std::vector<uint32_t> sorted(N);
    for (uint32_t i = 0; i < N; ++i) {
        sorted[i] = mortons[order[i]];
        faces[i] = order[i];
    }
    if (mortonsSorted)
        std::memcpy(mortonsSorted, sorted.data(), 4ull * N);

    // include/cuda/algorithm.h:11-41 via lbvh.cu:181-184
    std::vector<uint32_t> ps(N);
    {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < N; ++i) {
            uint32_t diff = (i == 0) ? 1u : (sorted[i - 1] == sorted[i] ? 0u : 1u);
            acc += diff;
            ps[i] = acc;
        }
    }
    uint32_t M = ps[N - 1];
    compactedFaces[0] = 0; // zero-initialised device_vector entry the scatter never writes
    for (uint32_t i = 1; i < N; ++i)
        if (ps[i] != ps[i - 1])
            compactedFaces[ps[i] - 1] = i;
    *compactNumOut = M;

    // lbvh.cu:186-219
    std::vector<AABB> aabbs(M);
    std::vector<uint32_t> newMortons(M);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < static_cast<int64_t>(M); ++c) {
        newMortons[c] = sorted[compactedFaces[c]];
        AABB aabb = AABB::CreateInitial();
        uint32_t next = (c == static_cast<int64_t>(M) - 1) ? N : compactedFaces[c + 1];
        for (uint32_t fi = compactedFaces[c]; fi < next; ++fi) {
            uint32_t face = faces[fi];
            for (int i = 0; i < 3; ++i)
                aabb.Expand(LoadV3(positions, facePosIdx[3 * face + i]));
        }
        aabbs[c] = aabb;
    }
    if (uniqueMortons)
        std::memcpy(uniqueMortons, newMortons.data(), 4ull * M);
    if (leafAABBs6)
        std::memcpy(leafAABBs6, aabbs.data(), 24ull * M);

    // lbvh.cu:233, 253-260
    auto *inodes = static_cast<InternalNode *>(internalNodes32);
    int64_t inodeNum = static_cast<int64_t>(M) - 1;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < inodeNum; ++i)
        computeLBVHLinks(inodes, newMortons.data(), inodeNum, i);

    if (refitMode == 0) {
        // lbvh.cu:265-297
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < inodeNum; ++i) {
            uint32_t stk[64];
            int sz = 0;
            stk[sz++] = static_cast<uint32_t>(i);
            AABB aabb = AABB::CreateInitial();
            while (sz != 0) {
                const InternalNode &node = inodes[stk[--sz]];
                if (node.lft & ChildIsLeafMask)
                    aabb.Expand(aabbs[node.lft & LeafIndexMask]);
                else
                    stk[sz++] = node.lft;
                if (node.rht & ChildIsLeafMask)
                    aabb.Expand(aabbs[node.rht & LeafIndexMask]);
                else
                    stk[sz++] = node.rht;
            }
            inodes[i].aabb = aabb;
        }
    } else if (inodeNum > 0) {
        // Iterative post-order from the root. Unions are exact min/max, so the result has the same
        // bits as the walk above whatever the order.
        std::vector<uint32_t> stk;
        std::vector<uint8_t> state(inodeNum, 0);
        stk.push_back(0);
        while (!stk.empty()) {
            uint32_t n = stk.back();
            InternalNode &node = inodes[n];
            if (state[n] == 0) {
                state[n] = 1;
                if (!(node.rht & ChildIsLeafMask))
                    stk.push_back(node.rht);
                if (!(node.lft & ChildIsLeafMask))
                    stk.push_back(node.lft);
            } else {
                stk.pop_back();
                AABB aabb = AABB::CreateInitial();
                if (node.lft & ChildIsLeafMask)
                    aabb.Expand(aabbs[node.lft & LeafIndexMask]);
                else
                    aabb.Expand(inodes[node.lft].aabb);
                if (node.rht & ChildIsLeafMask)
                    aabb.Expand(aabbs[node.rht & LeafIndexMask]);
                else
                    aabb.Expand(inodes[node.rht].aabb);
                node.aabb = aabb;
            }
        }
    }
    return 0;
}

// src/ray_tracer/ray_tracer_render.cu:192-270 (rayCastLBVH<false>): fixed left-then-right order, a
// (node, visitCnt) stack, no distance culling, strict "t < nearest" so the first hit found wins ties.
// rays: R x 6 floats (pos, dir). Outputs per ray: t (FLT_MAX on miss), bary (2), prim (0xFFFFFFFF on
// miss). stats (optional, 2 x uint64): total internal-node boxes fetched, total triangles tested.
void ko_closest_hit(const float *rootAABB6, uint32_t faceNum, uint32_t compactedFaceNum,
                    const uint32_t *faces, const uint32_t *compactedFaces,
                    const void *internalNodes32, const float *positions, const uint32_t *facePosIdx,
                    const float *rays, uint32_t R, float *tOut, float *baryOut, uint32_t *primOut,
                    uint64_t *stats) {
    (void)rootAABB6;
    const auto *inodes = static_cast<const InternalNode *>(internalNodes32);
    uint64_t nodeFetches = 0, triTests = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nodeFetches, triTests)
    for (int64_t ri = 0; ri < static_cast<int64_t>(R); ++ri) {
        Ray ray{{rays[6 * ri], rays[6 * ri + 1], rays[6 * ri + 2]},
                {rays[6 * ri + 3], rays[6 * ri + 4], rays[6 * ri + 5]}};
        float nearestT = FloatMax;
        vec2 nearestBary{0.f, 0.f};
        uint32_t nearestFi = 0xFFFFFFFFu;

        HitShell hit{0.f, 0.f};
        auto hitNode = [&](uint32_t idx) {
            if (idx & ChildIsLeafMask)
                return false;
            hit = HitAABB(ray, inodes[idx].aabb);
            ++nodeFetches;
            return true;
        };
        auto finish = [&]() {
            tOut[ri] = nearestT;
            baryOut[2 * ri] = nearestBary.x;
            baryOut[2 * ri + 1] = nearestBary.y;
            primOut[ri] = nearestFi;
        };

        hitNode(0);
        if (hit.tEnter > hit.tExit) {
            finish();
            continue;
        }

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
                stkCnt[sz - 1] == 0 ? inodes[stkNode[sz - 1]].lft : inodes[stkNode[sz - 1]].rht;
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
                uint32_t k = compactedFaces[c];
                uint32_t kEnd = (c == compactedFaceNum - 1) ? faceNum : compactedFaces[c + 1];
                while (k < kEnd) {
                    uint32_t fi = faces[k];
                    HitTri h = HitTriangle(ray, LoadV3(positions, facePosIdx[3 * fi]),
                                           LoadV3(positions, facePosIdx[3 * fi + 1]),
                                           LoadV3(positions, facePosIdx[3 * fi + 2]));
                    ++triTests;
                    if (h.t >= 0.f && h.t < nearestT) {
                        nearestT = h.t;
                        nearestBary = h.bary;
                        nearestFi = fi;
                    }
                    ++k;
                }
            }
        }
        finish();
    }
    if (stats) {
        stats[0] = nodeFetches;
        stats[1] = triTests;
    }
}

// Single-call primitives so tests can compare them with the real glm in oracle/_ref.
void ko_hit_aabb(const float *ray6, const float *aabb6, float *out2) {
    Ray r{{ray6[0], ray6[1], ray6[2]}, {ray6[3], ray6[4], ray6[5]}};
    AABB b{{aabb6[0], aabb6[1], aabb6[2]}, {aabb6[3], aabb6[4], aabb6[5]}};
    HitShell h = HitAABB(r, b);
    out2[0] = h.tEnter;
    out2[1] = h.tExit;
}
void ko_hit_triangle(const float *ray6, const float *tri9, float *out3) {
    Ray r{{ray6[0], ray6[1], ray6[2]}, {ray6[3], ray6[4], ray6[5]}};
    HitTri h = HitTriangle(r, {tri9[0], tri9[1], tri9[2]}, {tri9[3], tri9[4], tri9[5]},
                           {tri9[6], tri9[7], tri9[8]});
    out3[0] = h.t;
    out3[1] = h.bary.x;
    out3[2] = h.bary.y;
}
void ko_normalize(const float *v3, float *out3) {
    vec3 n = normalize({v3[0], v3[1], v3[2]});
    out3[0] = n.x;
    out3[1] = n.y;
    out3[2] = n.z;
}
void ko_mat4_mul_vec4(const float *m16, const float *v4, float *out4) {
    mat4 m;
    std::memcpy(&m, m16, 64);
    vec4 r = m * vec4{v4[0], v4[1], v4[2], v4[3]};
    std::memcpy(out4, &r, 16);
}
void ko_mat3_mul_vec3(const float *m9, const float *v3, float *out3) {
    mat3 m;
    std::memcpy(&m, m9, 36);
    vec3 r = m * vec3{v3[0], v3[1], v3[2]};
    std::memcpy(out3, &r, 12);
}

} // extern "C"
