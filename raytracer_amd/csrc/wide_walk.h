// The trace-side 8-wide tree (built by kk_trace_accel_build) and what a lane needs to walk it: record
// layouts, the conservative slab test of a record, the reference's acceptance tests, the triangle
// test, and whole-ray walks for kernels that trace one ray per lane after another (the path tracer).
// The ray-buffer kernel in trace_accel.hip uses the same pieces inside its persistent wave loop.
#pragma once
#include "kk_math.h"
#include "traverse.h"

namespace kk {
namespace accel {

using trav::LEAF;
using trav::LEAF_MASK;

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

struct alignas(128) Wide8 {
    float origin[3];
    int8_t exp[3];
    uint8_t imask;      // slots that hold a binary subtree
    uint32_t childBase; // wide node of the first such slot; the others follow in slot order
    uint32_t triBase;   // first TriRec of this node's leaf slots (stored slot after slot)
    uint32_t leafMask;  // bits 0-7: slots that hold a leaf; LONG_RUN, HUGE_RUN: see counts
    uint32_t counts;    // 4 bits per slot: triangles of a leaf slot. LONG_RUN (some leaf has more than
                        // 15): they are in off16 / cnt16 instead. HUGE_RUN (they do not fit 16 bits
                        // either): the triangles of all leaf slots, which are then tested together.
    uint8_t q[6][8];    // minx[8] maxx[8] miny[8] maxy[8] minz[8] maxz[8]
    uint16_t off16[8];  // LONG_RUN only: first triangle of a leaf slot relative to triBase ...
    uint16_t cnt16[8];  // ... and its triangle count
    uint32_t spare[4];  // a visit reads the first 80 bytes (112 in a LONG_RUN node with a leaf hit)
};
static_assert(sizeof(Wide8) == 128, "one cache line");
constexpr uint32_t LONG_RUN = 0x100u, HUGE_RUN = 0x200u;

struct alignas(16) TriRec {
    float v0[3];
    uint32_t fi;
    float v1[3];
    uint32_t parent; // binary node whose child is this triangle's leaf
    float v2[3];
    uint32_t k; // position in the build's sorted face order
};
static_assert(sizeof(TriRec) == 48, "three 16-byte loads");

struct alignas(256) Header {
    float rootBox[6];
    uint32_t faceNum, nodeNum;
    const kk_lbvh_node *nodes; // the build's own nodes (parent boxes for the acceptance test)
    uint32_t wideNum;
    uint32_t tooDeep; // set by the build when wide nodes were left beyond its last level: the tree is
                      // deeper than the walks' stacks are made for and must not be traced
};

// A product (b - o) * (1 / d) is within 3 * 2^-24 of the reference's fl((b - o) / d) (one rounding
// each in 1 / d, the product, and the quotient it stands for); min / max keep the bound.
constexpr float SLAB_REL = 1e-6f, SLAB_ABS = 1e-30f;
constexpr float SLAB_SHRINK = 1.f - 1e-6f; // the same margin as a factor on the entry distance


typedef float v2f __attribute__((ext_vector_type(2)));



struct RayState {
    f3 pos, dir, inv;
};

// Does `r` pass the slab test of `box`, as the reference computes it? Decided by the reciprocal form
// when that is clear of its rounding margin, by the reference's divisions otherwise.
__device__ __forceinline__ bool passes_box(const AABB &box, const RayState &r) {
    const float h0 = (box.minPos.x - r.pos.x) * r.inv.x, h1 = (box.maxPos.x - r.pos.x) * r.inv.x;
    const float h2 = (box.minPos.y - r.pos.y) * r.inv.y, h3 = (box.maxPos.y - r.pos.y) * r.inv.y;
    const float h4 = (box.minPos.z - r.pos.z) * r.inv.z, h5 = (box.maxPos.z - r.pos.z) * r.inv.z;
    float en = fmaxf(fmaxf(fminf(h0, h1), fminf(h2, h3)), fminf(h4, h5));
    const float ex = fminf(fminf(fmaxf(h0, h1), fmaxf(h2, h3)), fmaxf(h4, h5));
    en = fmaxf(en, 0.f);
    const float m = SLAB_REL * fmaxf(en, fabsf(ex)) + SLAB_ABS;
    if (en <= ex - m)
        return true;
    if (en > ex + m)
        return false;
    Ray ray{r.pos, r.dir};
    float tEnter, tExit;
    HitAABB(ray, box, tEnter, tExit);
    return !(tEnter > tExit);
}

// Does `r` pass the slab test of the build's node `parent`, as the reference computes it?
__device__ __forceinline__ bool passes_parent(const kk_lbvh_node *__restrict__ nodes, uint32_t parent,
                                              const RayState &r) {
    return passes_box(trav::load_node(nodes, parent).box, r);
}

// Does `r` pass the reference's slab test of node `p` and of every node above it?
__device__ __forceinline__ bool passes_chain(const kk_lbvh_node *__restrict__ nodes,
                                             const uint32_t *__restrict__ parents, uint32_t p,
                                             const RayState &r) {
    const Ray ray{r.pos, r.dir};
    while (p != EMPTY) {
        const trav::NodeData nd = trav::load_node(nodes, p);
        float tEnter, tExit;
        HitAABB(ray, nd.box, tEnter, tExit);
        if (tEnter > tExit)
            return false;
        p = parents[p];
    }
    return true;
}

enum : uint32_t { MODE_FAST = 0, MODE_CAREFUL = 1, MODE_CHAIN = 2 };

// Triangles [k0, k0 + n) against the ray. Normally a better hit is taken at once and whether its
// leaf's parent box passes the reference's slab test is checked once, for the final hit, when the ray
// retires (MODE_FAST). A ray being re-traced after that check failed checks every hit it takes
// (MODE_CAREFUL); a ray with a zero direction component checks the whole chain of boxes (MODE_CHAIN).
//
// WATERTIGHT replaces the reference's Moeller-Trumbore test by the shear-and-scale test of Woop,
// Benthin and Wald (2013), two-sided, with its double-precision fallback on an edge: a ray through a
// shared edge or vertex cannot slip between two triangles. It is an option, not the default, because
// its hit / miss decisions differ from the reference's on grazing rays.
struct Shear {
    int kx, ky, kz;
    float sx, sy, sz;
};
__device__ __forceinline__ Shear make_shear(f3 d) {
    Shear s;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    s.kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    s.kx = s.kz + 1 == 3 ? 0 : s.kz + 1;
    s.ky = s.kx + 1 == 3 ? 0 : s.kx + 1;
    if (d[s.kz] < 0.f) {
        const int t = s.kx;
        s.kx = s.ky;
        s.ky = t;
    }
    s.sx = d[s.kx] / d[s.kz];
    s.sy = d[s.ky] / d[s.kz];
    s.sz = 1.f / d[s.kz];
    return s;
}
// returns t (< 0: miss); bx, by = weights of v1 and v2 like the reference's barycentrics.
// The edge functions decide hit or miss (that is what makes the test watertight); the distance and the
// barycentrics of a hit are then computed with the reference's Moeller-Trumbore arithmetic (the
// sheared-space t = (U Az + V Bz + W Cz) / det loses several digits on small triangles far from the
// origin), so that a ray that hits the same triangle under both tests reports the same bits. Only when
// that arithmetic cannot be used (its determinant is below its own threshold, or its t comes out
// negative) are the sheared-space values kept.
__device__ __forceinline__ float hit_watertight(const Shear &s, f3 org, f3 dir, f3 v0, f3 v1, f3 v2, float &bx,
                                                float &by) {
    const f3 A = v0 - org, B = v1 - org, C = v2 - org;
    const float Ax = A[s.kx] - s.sx * A[s.kz], Ay = A[s.ky] - s.sy * A[s.kz];
    const float Bx = B[s.kx] - s.sx * B[s.kz], By = B[s.ky] - s.sy * B[s.kz];
    const float Cx = C[s.kx] - s.sx * C[s.kz], Cy = C[s.ky] - s.sy * C[s.kz];
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.f || V == 0.f || W == 0.f) {
        U = static_cast<float>(static_cast<double>(Cx) * By - static_cast<double>(Cy) * Bx);
        V = static_cast<float>(static_cast<double>(Ax) * Cy - static_cast<double>(Ay) * Cx);
        W = static_cast<float>(static_cast<double>(Bx) * Ay - static_cast<double>(By) * Ax);
    }
    if ((U < 0.f || V < 0.f || W < 0.f) && (U > 0.f || V > 0.f || W > 0.f))
        return -1.f;
    const float det = U + V + W;
    if (det == 0.f)
        return -1.f;
    const float Az = s.sz * A[s.kz], Bz = s.sz * B[s.kz], Cz = s.sz * C[s.kz];
    const float T = U * Az + V * Bz + W * Cz;
    const float inv = 1.f / det;
    const float t = T * inv;
    if (!(t >= 0.f))
        return -1.f;
    bx = V * inv;
    by = W * inv;
    { // include/cg/ray.h:52-58 -> glm intersectRayTriangle, without its range tests
        const f3 e1 = v1 - v0, e2 = v2 - v0;
        const f3 p = cross(dir, e2);
        const float d = dot(e1, p);
        if (d > FloatEps || d < -FloatEps) {
            const f3 dist = org - v0;
            const float u = dot(dist, p);
            const f3 perp = cross(dist, e1);
            const float v = dot(dir, perp);
            const float invD = 1.f / d;
            const float tm = dot(e2, perp) * invD;
            if (tm >= 0.f) {
                bx = u * invD;
                by = v * invD;
                return tm;
            }
        }
    }
    return t;
}

struct Hit8 {
    float t, bx, by;
    uint32_t fi; // face index, 0xFFFFFFFF = miss
    uint32_t k;  // position in the sorted face order: the smaller one wins among equal t
    uint32_t w;  // its TriRec
    // The ray passes the reference's slab test of the triangle's own bounding box. The leaf's parent
    // box contains that box and the test is monotonic in the box, so the parent-box test (and with it
    // every test above) is known to pass without loading the parent.
    bool sure;
};

// The watertight test as it runs in the walk: the reference's Moeller-Trumbore quantities come first.
// A ray that passes clear of the triangle's edges - by more than a bound on their rounding error - is
// a hit or a miss under both tests, and a hit then IS the reference's hit (same t, same barycentrics).
// Only a ray inside that thin band around an edge (or with a determinant too small for the
// reference's test) is decided by the edge functions above, which two triangles sharing the edge
// evaluate identically: no ray slips between them.
__device__ __forceinline__ float hit_watertight_banded(const Shear &s, f3 org, f3 dir, f3 v0, f3 v1, f3 v2,
                                                       float &bx, float &by) {
    const f3 e1 = v1 - v0, e2 = v2 - v0;
    const f3 p = cross(dir, e2);
    const float det = dot(e1, p);
    const f3 dist = org - v0;
    const float u = dot(dist, p);
    const f3 perp = cross(dist, e1);
    const float v = dot(dir, perp);
    const float ad = fabsf(det);
    const float su = det < 0.f ? -u : u, sv = det < 0.f ? -v : v;
    const float l1 = (fabsf(dist.x) + fabsf(dist.y) + fabsf(dist.z)) * (fabsf(dir.x) + fabsf(dir.y) + fabsf(dir.z)) *
                     (fabsf(e1.x) + fabsf(e1.y) + fabsf(e1.z) + fabsf(e2.x) + fabsf(e2.y) + fabsf(e2.z));
    const float m = 64.f * FloatEps * l1;
    if (su < -m || sv < -m || su + sv > ad + m)
        return -1.f; // clear of the triangle
    if (ad > FloatEps && su > m && sv > m && su + sv < ad - m) { // clear of its edges: the reference's hit
        const float inv = 1.f / det;
        const float t = dot(e2, perp) * inv;
        bx = u * inv;
        by = v * inv;
        return t; // (a negative t is rejected by the caller, as in the reference)
    }
    return hit_watertight(s, org, dir, v0, v1, v2, bx, by);
}

template <bool COUNT, bool WATERTIGHT>
__device__ __forceinline__ void test_run(const TriRec *__restrict__ recs,
                                         const kk_lbvh_node *__restrict__ nodes,
                                         const uint32_t *__restrict__ parents, const RayState &r,
                                         uint32_t mode, uint32_t w0, uint32_t n, Hit8 &best,
                                         uint32_t &triTests, const Shear &shear = Shear{}) {
    const Ray ray{r.pos, r.dir};
    for (uint32_t w = w0; w < w0 + n; ++w) {
        const uint4 *p = reinterpret_cast<const uint4 *>(recs + w);
        const uint4 a = p[0], b = p[1], c = p[2];
        float bx = 0.f, by = 0.f;
        const f3 v0 = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        const f3 v1 = F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        const f3 v2 = F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
        const float t = WATERTIGHT ? hit_watertight_banded(shear, r.pos, r.dir, v0, v1, v2, bx, by)
                                   : HitTriangleFlat(ray, v0, v1, v2, bx, by);
        if (COUNT)
            ++triTests;
        if (t >= 0.f && (t < best.t || (t == best.t && c.w < best.k))) {
            if (mode == MODE_FAST || (mode == MODE_CAREFUL ? passes_parent(nodes, b.w, r)
                                                           : passes_chain(nodes, parents, b.w, r))) {
                best.t = t;
                best.bx = bx;
                best.by = by;
                best.fi = a.w;
                best.k = c.w;
                best.w = w;
                if (mode == MODE_FAST) {
                    AABB tb;
                    tb.minPos = {fminf(fminf(v0.x, v1.x), v2.x), fminf(fminf(v0.y, v1.y), v2.y), fminf(fminf(v0.z, v1.z), v2.z)};
                    tb.maxPos = {fmaxf(fmaxf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.z, v1.z), v2.z)};
                    best.sure = passes_box(tb, r);
                }
            }
        }
    }
}

// ---- the pending triangles of a wave's lanes, tested as one pool --------------------------------------

// Inclusive prefix sum / running maximum over the 64 lanes of a wave with data-parallel-primitive moves:
// four shifts inside each row of 16 lanes, then the last lane of a row into the rows after it.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    auto mx = [](uint32_t a, int b) { return a > static_cast<uint32_t>(b) ? a : static_cast<uint32_t>(b); };
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
    v = mx(v, __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
    return v;
}

// One round of the pooled triangle test; every lane of the wave calls it. A lane with `pendC` triangles
// pending (records pendW ...) and `joins` set owns that many (ray, triangle) pairs; the pairs of all lanes,
// in lane order, are numbered from 0 and pair i < 64 is tested by lane i with its owner's ray. An owner
// ends up with the hit its own one-by-one loop (test_run) would keep - the smallest (t, sorted position)
// among its pairs that are hit, not behind its best and accepted under its mode - as bestT (+0 for -0),
// bestK and bestW (the triangle's record; barycentrics and face are for the caller to recompute from
// it when the walk is over), and the function returns true for it. pendW / pendC move past the pairs
// that were served (the rest wait for the next round). closest == false (an any-hit ray): bestT is the
// ray's far end and any hit nearer than that is taken. waveKey / waveRec / waveMark: 64 entries of LDS
// each, this wave's own.
template <bool COUNT>
__device__ __forceinline__ bool pooled_round(const TriRec *__restrict__ recs, const kk_lbvh_node *__restrict__ nodes,
                                             const uint32_t *__restrict__ parents,
                                             unsigned long long *waveKey, uint32_t *waveRec, uint32_t *waveMark,
                                             int lane, const RayState &ray, uint32_t mode, bool closest, bool joins,
                                             uint32_t &pendW, uint32_t &pendC, float &bestT, uint32_t &bestK,
                                             uint32_t &bestW, uint32_t &triTests, uint32_t &pairsOut) {
    const uint32_t c = joins ? (pendC < 64u ? pendC : 64u) : 0u;
    const uint32_t incl = wave_inclusive_sum(c);
    const uint32_t start = incl - c;
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    pairsOut = total < 64u ? total : 64u;
    // owner of pair i = the last lane whose first pair is at or before i: every owner marks its first
    // pair, a running maximum over the pairs spreads the marks. An owner's word starts as its present
    // best, so a pair that ties in t with a later sorted position loses by itself.
    // (t >= 0: its bits order like the value; "+ 0" turns -0 into +0)
    const unsigned long long before = (static_cast<unsigned long long>(__float_as_uint(bestT + 0.f)) << 32) | bestK;
    waveMark[lane] = 0u;
    waveKey[lane] = before;
    __builtin_amdgcn_wave_barrier();
    if (c != 0u && start < 64u)
        waveMark[start] = static_cast<uint32_t>(lane);
    __builtin_amdgcn_wave_barrier();
    const bool pairOn = static_cast<uint32_t>(lane) < total;
    const int own = static_cast<int>(wave_inclusive_max(waveMark[lane]));
    auto fromF = [&](float v) { return __shfl(v, own, 64); };
    auto fromU = [&](uint32_t v) { return static_cast<uint32_t>(__shfl(static_cast<int>(v), own, 64)); };
    RayState oray;
    oray.pos = F3(fromF(ray.pos.x), fromF(ray.pos.y), fromF(ray.pos.z));
    oray.dir = F3(fromF(ray.dir.x), fromF(ray.dir.y), fromF(ray.dir.z));
    const float oT = fromF(bestT);
    const uint32_t rec = fromU(pendW - start) + static_cast<uint32_t>(lane);
    const uint32_t oFlags = fromU(mode | (closest ? 0x100u : 0u));
    const uint32_t oMode = oFlags & 0xffu;
    const bool oClosest = (oFlags & 0x100u) != 0u;
    unsigned long long key = ~0ull;
    if (pairOn) {
        const uint4 *tp = reinterpret_cast<const uint4 *>(recs + rec);
        const uint4 a = tp[0], b = tp[1], cc = tp[2];
        const f3 v0 = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        const f3 v1 = F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        const f3 v2 = F3(__uint_as_float(cc.x), __uint_as_float(cc.y), __uint_as_float(cc.z));
        float bx, by;
        const float t = HitTriangleFlat(Ray{oray.pos, oray.dir}, v0, v1, v2, bx, by);
        bool cand = t >= 0.f && (oClosest ? t <= oT : t < oT);
        if (cand && oMode != MODE_FAST) {
            oray.inv = 1.f / oray.dir; // (a ray with a zero component is in MODE_CHAIN, which divides)
            cand = oMode == MODE_CAREFUL ? passes_parent(nodes, b.w, oray) : passes_chain(nodes, parents, b.w, oray);
        }
        if (cand) {
            key = (static_cast<unsigned long long>(__float_as_uint(t + 0.f)) << 32) | cc.w;
            atomicMin(&waveKey[own], key);
        }
    }
    const uint32_t served = (incl < 64u ? incl : 64u) - (start < 64u ? start : 64u);
    if (COUNT)
        triTests += served;
    pendW += served;
    pendC -= served;
    if (__ballot(key != ~0ull) == 0ull)
        return false; // no pair of this round is a hit its owner might take
    __builtin_amdgcn_wave_barrier();
    if (key != ~0ull && waveKey[own] == key)
        waveRec[own] = rec; // (sorted positions are unique: one pair writes)
    __builtin_amdgcn_wave_barrier();
    const unsigned long long now = waveKey[lane];
    const bool won = c != 0u && now != before;
    if (won) {
        bestT = __uint_as_float(static_cast<uint32_t>(now >> 32));
        bestK = static_cast<uint32_t>(now);
        bestW = waveRec[lane];
    }
    __builtin_amdgcn_wave_barrier();
    return won;
}

// ---- one ray -----------------------------------------------------------------------------------------

// Reciprocal direction, near-plane signs, octant and acceptance mode of a ray. A zero / denormal / NaN
// component makes the ray irregular: its reciprocal is replaced by +-1e18 (finite stand-ins keep the
// walk a superset of the reference's visits: inside a slab -> (-huge, +huge), outside -> both the same
// sign) and every hit is checked against the whole chain of boxes (MODE_CHAIN).
struct RaySigns {
    bool negX, negY, negZ;
    uint32_t oct; // negX | negY << 1 | negZ << 2
};
__device__ __forceinline__ void prepare_ray(f3 pos, f3 dir, RayState &ray, RaySigns &sg, uint32_t &mode) {
    ray.pos = pos;
    ray.dir = dir;
    ray.inv = 1.f / dir;
    const bool regular = fabsf(ray.inv.x) < 1e18f && fabsf(ray.inv.y) < 1e18f && fabsf(ray.inv.z) < 1e18f;
    mode = MODE_FAST;
    if (!regular) {
        mode = MODE_CHAIN;
        if (!(fabsf(ray.inv.x) < 1e18f))
            ray.inv.x = copysignf(1e18f, ray.dir.x);
        if (!(fabsf(ray.inv.y) < 1e18f))
            ray.inv.y = copysignf(1e18f, ray.dir.y);
        if (!(fabsf(ray.inv.z) < 1e18f))
            ray.inv.z = copysignf(1e18f, ray.dir.z);
    }
    sg.negX = ray.inv.x < 0.f;
    sg.negY = ray.inv.y < 0.f;
    sg.negZ = ray.inv.z < 0.f;
    sg.oct = (sg.negX ? 1u : 0u) | (sg.negY ? 2u : 0u) | (sg.negZ ? 4u : 0u);
}

// bit i of the result = slot i ^ oct of `slots`: ascending bits run front to back for the ray's octant
__device__ __forceinline__ uint32_t octant_order(uint32_t slots, uint32_t oct) {
    slots = (oct & 1u) ? (((slots & 0xAAu) >> 1) | ((slots & 0x55u) << 1)) : slots;
    slots = (oct & 2u) ? (((slots & 0xCCu) >> 2) | ((slots & 0x33u) << 2)) : slots;
    slots = (oct & 4u) ? (((slots & 0xF0u) >> 4) | ((slots & 0x0Fu) << 4)) : slots;
    return slots;
}

// What a visit of one record found.
struct Visit {
    uint32_t inner;        // hit slots that hold subtrees, bit = slot
    uint32_t imask;        // all subtree slots of the record
    uint32_t childBase;    // wide node of the first subtree slot
    uint32_t leafW, leafC; // triangle records from the first to the last hit leaf slot (leafC == 0: none)
    uint32_t leafSlots;    // all leaf slots of the record
    uint32_t leafHits;     // the leaf slots that are hit
    uint32_t counts;       // the record's count word (4 bits per slot unless LONG_RUN / HUGE_RUN)
    uint32_t flags;        // the record's leaf word (leafSlots, LONG_RUN, HUGE_RUN)
    float restEntry;       // REST: smallest entry distance among the hit subtree slots other than the one
                           // that comes first in the ray's octant order (+inf when there is no other)
};

// Slab tests of the eight slots of record `cur` against the ray, conservative: a slot is dropped only
// when it is missed beyond the rounding margin or entered beyond `limit`. DIAG also reports whether
// any slot would be hit without the limit.
template <bool DIAG, bool REST = false>
__device__ __forceinline__ Visit visit_record(const char *__restrict__ wideBytes, uint32_t cur,
                                              const RayState &ray, const RaySigns &sg, float limit,
                                              bool &anyNoLimit) {
    const uint4 *rec = reinterpret_cast<const uint4 *>(wideBytes + (static_cast<size_t>(cur) << 7));
    const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
    // plane distance = q * (2^exp / d) + (origin - o) / d; the "+ 0" turns a -0 offset into
    // +0, so that no plane distance is -0 and a box that touches the origin is kept
    const float ax = ldexpf(ray.inv.x, static_cast<int>(static_cast<int8_t>(r0.w & 0xffu)));
    const float ay = ldexpf(ray.inv.y, static_cast<int>(static_cast<int8_t>((r0.w >> 8) & 0xffu)));
    const float az = ldexpf(ray.inv.z, static_cast<int>(static_cast<int8_t>((r0.w >> 16) & 0xffu)));
    const float bx = __builtin_fmaf(__uint_as_float(r0.x) - ray.pos.x, ray.inv.x, 0.f);
    const float by = __builtin_fmaf(__uint_as_float(r0.y) - ray.pos.y, ray.inv.y, 0.f);
    const float bz = __builtin_fmaf(__uint_as_float(r0.z) - ray.pos.z, ray.inv.z, 0.f);
    const v2f ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
    const v2f bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
    // near / far plane bytes by the sign of the direction: [0] slots 0-3, [1] slots 4-7
    const uint32_t qnx[2] = {sg.negX ? r2.z : r2.x, sg.negX ? r2.w : r2.y};
    const uint32_t qfx[2] = {sg.negX ? r2.x : r2.z, sg.negX ? r2.y : r2.w};
    const uint32_t qny[2] = {sg.negY ? r3.z : r3.x, sg.negY ? r3.w : r3.y};
    const uint32_t qfy[2] = {sg.negY ? r3.x : r3.z, sg.negY ? r3.y : r3.w};
    const uint32_t qnz[2] = {sg.negZ ? r4.z : r4.x, sg.negZ ? r4.w : r4.y};
    const uint32_t qfz[2] = {sg.negZ ? r4.x : r4.z, sg.negZ ? r4.y : r4.w};
    // A slot is kept unless it is missed beyond the rounding margin: the sign of
    // exit - entry * (1 - margin) is shifted into `missed`, slot 7 first.
    uint32_t missed = 0;
    float en[8];
#pragma unroll
    for (int pr = 3; pr >= 0; --pr) {
        const int h = pr >> 1, sh = 16 * (pr & 1); // slots 2 pr and 2 pr + 1
        auto two = [&](uint32_t w) {
            v2f q;
            q.x = static_cast<float>((w >> sh) & 0xffu);
            q.y = static_cast<float>((w >> (sh + 8)) & 0xffu);
            return q;
        };
        const v2f hnx = __builtin_elementwise_fma(two(qnx[h]), ax2, bx2);
        const v2f hny = __builtin_elementwise_fma(two(qny[h]), ay2, by2);
        const v2f hnz = __builtin_elementwise_fma(two(qnz[h]), az2, bz2);
        const v2f hfx = __builtin_elementwise_fma(two(qfx[h]), ax2, bx2);
        const v2f hfy = __builtin_elementwise_fma(two(qfy[h]), ay2, by2);
        const v2f hfz = __builtin_elementwise_fma(two(qfz[h]), az2, bz2);
        const float en1 = fmaxf(fmaxf(fmaxf(hnx.y, hny.y), hnz.y), 0.f);
        const float en0 = fmaxf(fmaxf(fmaxf(hnx.x, hny.x), hnz.x), 0.f);
        const float far1 = fminf(fminf(hfx.y, hfy.y), hfz.y), far0 = fminf(fminf(hfx.x, hfy.x), hfz.x);
        en[2 * pr] = en0;
        en[2 * pr + 1] = en1;
        if (DIAG)
            anyNoLimit |= !(en1 * SLAB_SHRINK > far1) || !(en0 * SLAB_SHRINK > far0);
        const float ex1 = fminf(far1, limit), ex0 = fminf(far0, limit);
        missed = __builtin_amdgcn_alignbit(missed, __float_as_uint(__builtin_fmaf(-SLAB_SHRINK, en1, ex1)), 31);
        missed = __builtin_amdgcn_alignbit(missed, __float_as_uint(__builtin_fmaf(-SLAB_SHRINK, en0, ex0)), 31);
    }
    const uint32_t hits = ~missed;
    Visit v;
    v.imask = r0.w >> 24;
    v.childBase = r1.x;
    v.inner = hits & v.imask;
    v.restEntry = __builtin_inff();
    if (REST) {
        // (entry distances are >= +0: their bits order like the values; a slot that is not wanted reads
        // as all ones)
        const uint32_t inner = v.inner & 0xffu;
        const uint32_t firstI = static_cast<uint32_t>(__ffs(static_cast<int>(octant_order(inner, sg.oct)))) - 1u;
        const uint32_t rest = inner & ~(1u << ((firstI ^ sg.oct) & 7u)); // (inner == 0: nothing to clear)
        const int unwanted = static_cast<int>(~rest);
        uint32_t key[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) // bit sl of `unwanted` spread over the word: 0 or all ones
            key[sl] = __float_as_uint(en[sl]) | static_cast<uint32_t>(__builtin_amdgcn_sbfe(unwanted, sl, 1));
        auto min3 = [](uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); };
        const uint32_t m = min3(min3(key[0], key[1], key[2]), min3(key[3], key[4], key[5]), min(key[6], key[7]));
        v.restEntry = m == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(m);
    }
    v.leafSlots = r1.z & 0xffu;
    v.leafW = v.leafC = 0u;
    const uint32_t leafHits = hits & r1.z & 0xffu;
    v.leafHits = leafHits;
    v.counts = r1.w;
    v.flags = r1.z;
    if (leafHits != 0u) {
        if (r1.z & HUGE_RUN) {
            v.leafW = r1.y;
            v.leafC = r1.w;
        } else if (r1.z & LONG_RUN) { // rare: 16-bit offsets and counts from the record's sixth and seventh piece
            const uint32_t first = static_cast<uint32_t>(__ffs(static_cast<int>(leafHits))) - 1u;
            const uint32_t last = 31u - static_cast<uint32_t>(__clz(static_cast<int>(leafHits)));
            const unsigned short *oc = reinterpret_cast<const unsigned short *>(rec + 5);
            const uint32_t b0 = oc[first], b1 = static_cast<uint32_t>(oc[last]) + oc[8 + last];
            v.leafW = r1.y + b0;
            v.leafC = b1 - b0;
        } else {
            // triangles before slot n = sum of the count nibbles below n
            const uint32_t first = static_cast<uint32_t>(__ffs(static_cast<int>(leafHits))) - 1u;
            const uint32_t last = 31u - static_cast<uint32_t>(__clz(static_cast<int>(leafHits)));
            auto below = [&](uint32_t n) {
                const uint32_t x = n >= 8u ? r1.w : (r1.w & ((1u << (4u * n)) - 1u));
                return (((x & 0x0F0F0F0Fu) + ((x >> 4) & 0x0F0F0F0Fu)) * 0x01010101u) >> 24;
            };
            const uint32_t b0 = below(first), b1 = below(last + 1u);
            v.leafW = r1.y + b0;
            v.leafC = b1 - b0;
        }
    }
    return v;
}

// ---- whole-ray walks for kernels that give each lane one ray after another ----------------------------

// Device pointers into an accel buffer (made on the host by make_view, trace_accel.hip).
struct View {
    const Header *hdr;
    const Wide8 *wide;
    const TriRec *recs;
    const uint32_t *parents;
};
View make_view(const void *d_accel, uint32_t faceNum, uint32_t compactedNum);

// A lane's stack of groups: LANE_LDS_DEPTH entries in LDS (entry e of the lane at base[e * stride],
// bits[e * stride]), the rest - a tree deeper than that many wide levels along one path - in private
// memory. A walk pushes at most one group per wide level and a wide path is at most 32 levels.
constexpr int LANE_LDS_DEPTH = 16, LANE_DEPTH = 34;
struct LaneStack {
    uint32_t *base;
    unsigned short *bits;
    int stride;
};

// The overflow entries are an array of their own (`deep`, LANE_DEPTH - LANE_LDS_DEPTH entries in the
// caller's private memory), so that the three words of the walk stay in registers.
struct LaneWalk {
    uint32_t grpBase, grpBits;
    int sp;
    __device__ __forceinline__ void start(uint32_t oct) {
        sp = 0;
        grpBase = 0u;
        grpBits = (1u << (0u ^ oct)) | (1u << 8); // the root: slot 0 of a node whose only subtree is slot 0
    }
    // next record to visit; false when the walk is over
    __device__ __forceinline__ bool next(const LaneStack &st, const uint2 *deep, uint32_t oct, uint32_t &cur) {
        while ((grpBits & 0xffu) == 0u) {
            if (sp == 0)
                return false;
            --sp;
            if (sp < LANE_LDS_DEPTH) {
                grpBase = st.base[sp * st.stride];
                grpBits = st.bits[sp * st.stride];
            } else if (sp < LANE_DEPTH) {
                grpBase = deep[sp - LANE_LDS_DEPTH].x;
                grpBits = deep[sp - LANE_LDS_DEPTH].y;
            }
            // else: an entry enter() could not keep. Unreachable for trees the builder accepts (at
            // most MAX_LEVELS = 32 wide levels, one push per level; deeper trees are flagged by the
            // build, see Header::tooDeep); popped as an empty group.
        }
        const uint32_t i = static_cast<uint32_t>(__ffs(static_cast<int>(grpBits))) - 1u;
        grpBits &= grpBits - 1u; // the lowest set bit is a pending slot
        cur = grpBase + __popc((grpBits >> 8) & ((1u << (i ^ oct)) - 1u));
        return true;
    }
    // the subtree hits of a visit become the current group; the rest of the old one waits
    __device__ __forceinline__ void enter(const LaneStack &st, uint2 *deep, const Visit &v, uint32_t oct) {
        if (v.inner == 0u)
            return;
        if ((grpBits & 0xffu) != 0u) {
            if (sp < LANE_LDS_DEPTH) {
                st.base[sp * st.stride] = grpBase;
                st.bits[sp * st.stride] = static_cast<unsigned short>(grpBits);
            } else if (sp < LANE_DEPTH)
                deep[sp - LANE_LDS_DEPTH] = make_uint2(grpBase, grpBits);
            ++sp;
        }
        grpBase = v.childBase;
        grpBits = octant_order(v.inner, oct) | (v.imask << 8);
    }
};

// Closest hit of one ray: the result of the reference's rayCastLBVH (see trace_accel.hip for why).
__device__ __forceinline__ Hit8 closest_hit_lane(const View &v, Ray r, const LaneStack &st) {
    Hit8 best;
    best.t = FloatMax;
    best.bx = best.by = 0.f;
    best.fi = best.k = best.w = 0xFFFFFFFFu;
    best.sure = false;
    AABB rootBox;
    rootBox.minPos = {v.hdr->rootBox[0], v.hdr->rootBox[1], v.hdr->rootBox[2]};
    rootBox.maxPos = {v.hdr->rootBox[3], v.hdr->rootBox[4], v.hdr->rootBox[5]};
    float tEnter, tExit;
    HitAABB(r, rootBox, tEnter, tExit);
    if (tEnter > tExit || !ray_is_finite(r.pos, r.dir))
        return best;
    const kk_lbvh_node *nodes = v.hdr->nodes;
    const char *wideBytes = reinterpret_cast<const char *>(v.wide);
    RayState ray;
    RaySigns sg;
    uint32_t mode;
    prepare_ray(r.pos, r.dir, ray, sg, mode);
    uint32_t unused = 0;
    bool unusedFlag = false;
    LaneWalk w;
    uint2 deep[LANE_DEPTH - LANE_LDS_DEPTH];
    while (true) { // a second walk only when the final hit of the first fails its parent-box test
        best.t = FloatMax;
        best.bx = best.by = 0.f;
        best.fi = best.k = best.w = 0xFFFFFFFFu;
        w.start(sg.oct);
        // Each pass of the loop a lane either visits a record or, while the last visit's triangles
        // last, tests one of them: the lanes of a wave that test triangles do not hold up the ones that
        // walk for the whole run. The order of a lane's own visits and tests is the plain one.
        uint32_t pendW = 0u, pendC = 0u;
        while (true) {
            if (pendC == 0u) {
                uint32_t cur;
                if (!w.next(st, deep, sg.oct, cur))
                    break;
                const float limit = best.t + trav::CULL_SLACK * best.t;
                const Visit vis = visit_record<false>(wideBytes, cur, ray, sg, limit, unusedFlag);
                pendW = vis.leafW;
                pendC = vis.leafC;
                w.enter(st, deep, vis, sg.oct);
            }
            if (pendC != 0u) {
                test_run<false, false>(v.recs, nodes, v.parents, ray, mode, pendW, 1u, best, unused);
                ++pendW;
                --pendC;
            }
        }
        if (best.fi != 0xFFFFFFFFu && mode == MODE_FAST && !best.sure &&
            !passes_parent(nodes, v.recs[best.w].parent, ray)) {
            mode = MODE_CAREFUL;
            continue;
        }
        return best;
    }
}

// True when the reference's closest hit of `r` is nearer than tMax: some triangle whose leaf's boxes
// the ray passes is hit at 0 <= t < tMax.
__device__ __forceinline__ bool occluded_lane(const View &v, Ray r, float tMax, const LaneStack &st) {
    AABB rootBox;
    rootBox.minPos = {v.hdr->rootBox[0], v.hdr->rootBox[1], v.hdr->rootBox[2]};
    rootBox.maxPos = {v.hdr->rootBox[3], v.hdr->rootBox[4], v.hdr->rootBox[5]};
    float tEnter, tExit;
    HitAABB(r, rootBox, tEnter, tExit);
    if (tEnter > tExit || !ray_is_finite(r.pos, r.dir))
        return false;
    const kk_lbvh_node *nodes = v.hdr->nodes;
    const char *wideBytes = reinterpret_cast<const char *>(v.wide);
    RayState ray;
    RaySigns sg;
    uint32_t mode;
    prepare_ray(r.pos, r.dir, ray, sg, mode);
    const float limit = tMax + trav::CULL_SLACK * tMax;
    bool unusedFlag = false;
    LaneWalk w;
    uint2 deep[LANE_DEPTH - LANE_LDS_DEPTH];
    w.start(sg.oct);
    // one record or one triangle per pass, as in closest_hit_lane
    uint32_t pendW = 0u, pendC = 0u;
    while (true) {
        if (pendC == 0u) {
            uint32_t cur;
            if (!w.next(st, deep, sg.oct, cur))
                return false;
            const Visit vis = visit_record<false>(wideBytes, cur, ray, sg, limit, unusedFlag);
            pendW = vis.leafW;
            pendC = vis.leafC;
            w.enter(st, deep, vis, sg.oct);
        }
        if (pendC != 0u) {
            const uint4 *p = reinterpret_cast<const uint4 *>(v.recs + pendW);
            const uint4 a = p[0], b = p[1], c = p[2];
            float bx, by;
            const float t = HitTriangle(r, F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)),
                                        F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)),
                                        F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)), bx, by);
            if (t >= 0.f && t < tMax &&
                (mode == MODE_CHAIN ? passes_chain(nodes, v.parents, b.w, ray) : passes_parent(nodes, b.w, ray)))
                return true;
            ++pendW;
            --pendC;
        }
    }
}

} // namespace accel
} // namespace kk
