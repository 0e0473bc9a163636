// Trace-side layout of a built LBVH and the closest-hit kernel that walks it.
//
// The reference's tree (32-byte nodes holding their OWN box, children found through faces ->
// triToPositionIndices -> positions, src/ray_tracer/lbvh.cu:317-321) costs a ray one scattered line
// per child box and four dependent gathers per triangle. For tracing, the same binary tree is
// repacked once after the build:
//   * WideNode (64 B, one aligned half line): both children's ids and both children's boxes. For an
//     internal child the box is that child's own box, bit for bit, so the slab test decides exactly
//     what rayCastLBVH (src/ray_tracer/ray_tracer_render.cu:192-270) decides when it visits the
//     child. For a leaf child (which the reference's layout gives no box) it is the box of the
//     leaf's triangles grown by a small pad; it only ever lets the walk skip triangles the ray
//     cannot reach.
//   * TriRec (48 B): the three vertices and the face index of the triangle at sorted position k, so a
//     triangle test is one contiguous fetch; the Moeller-Trumbore arithmetic is unchanged.
// Rays are traced in Morton order of their origin (ties: direction octant), found with the same radix
// sort the build uses; results are written back at the ray's own index, so callers see no reordering.
// The closest hit is defined as in traverse.h: smallest t, ties to the smaller sorted position k.
#include "kk_host.h"
#include "kk_math.h"
#include "traverse.h"

namespace kk {
namespace rsort {
int sort_pairs(uint32_t *keysA, uint32_t *keysB, uint32_t *valsA, uint32_t *valsB, bool iota,
               uint32_t n, void *ws, size_t wsBytes, hipStream_t stream, uint32_t **errorFlagOut);
}
extern "C" size_t kk_sort_workspace_bytes(uint32_t n);

namespace accel {

using trav::LEAF;
using trav::LEAF_MASK;

constexpr int THREADS = 256;
constexpr float LEAF_PAD = 1e-5f; // leaf boxes grow by this fraction of the scene's largest coordinate

struct alignas(64) WideNode {
    uint32_t child[2]; // internal: node index; leaf: LEAF | first sorted position k
    uint32_t cnt[2];   // leaf: number of triangles (0 for an internal child or a repeated leaf)
    float box[2][6];   // min xyz, max xyz of each child
};
static_assert(sizeof(WideNode) == 64, "one half cache line");

struct alignas(16) TriRec {
    float v0[3];
    uint32_t fi;
    float v1[3];
    uint32_t pad0;
    float v2[3];
    uint32_t pad1;
};
static_assert(sizeof(TriRec) == 48, "three 16-byte loads");

struct alignas(256) Header {
    float rootBox[6];
    uint32_t faceNum, nodeNum;
};

struct Layout {
    size_t header, nodes, tris, total;
};
inline Layout layout(uint32_t faceNum, uint32_t compactedNum) {
    Layout l;
    Carver c;
    const uint32_t nodeNum = compactedNum > 1 ? compactedNum - 1 : 1;
    l.header = c.take(sizeof(Header));
    l.nodes = c.take(sizeof(WideNode) * static_cast<size_t>(nodeNum));
    l.tris = c.take(sizeof(TriRec) * static_cast<size_t>(faceNum));
    l.total = c.off;
    return l;
}

__global__ __launch_bounds__(THREADS) void pack_tris_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, const float *__restrict__ positions,
    const uint32_t *__restrict__ tri, TriRec *__restrict__ recs, uint32_t faceNum) {
    const uint32_t k = blockIdx.x * THREADS + threadIdx.x;
    if (k >= faceNum)
        return;
    const uint32_t fi = lbvh->faces[k];
    const uint32_t i0 = tri[3ull * fi], i1 = tri[3ull * fi + 1], i2 = tri[3ull * fi + 2];
    const f3 a = trav::load_pos(positions, i0), b = trav::load_pos(positions, i1),
             c = trav::load_pos(positions, i2);
    uint4 *o = reinterpret_cast<uint4 *>(recs + k);
    o[0] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), fi);
    o[1] = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), 0u);
    o[2] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), 0u);
}

__global__ __launch_bounds__(THREADS) void make_wide_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, const TriRec *__restrict__ recs,
    WideNode *__restrict__ wide, Header *__restrict__ hdr, uint32_t nodeNum) {
    const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    if (i >= nodeNum)
        return;
    const kk_lbvh_node *nodes = lbvh->internal_nodes;
    const uint32_t faceNum = lbvh->face_num, leafNum = lbvh->compacted_face_num;
    const trav::NodeData root = trav::load_node(nodes, 0);
    float scale = 0.f;
    for (int a = 0; a < 3; ++a) {
        scale = fmaxf(scale, fabsf(root.box.minPos[a]));
        scale = fmaxf(scale, fabsf(root.box.maxPos[a]));
        scale = fmaxf(scale, root.box.maxPos[a] - root.box.minPos[a]);
    }
    const float pad = LEAF_PAD * scale;
    if (i == 0) {
        for (int a = 0; a < 3; ++a) {
            hdr->rootBox[a] = root.box.minPos[a];
            hdr->rootBox[3 + a] = root.box.maxPos[a];
        }
        hdr->faceNum = faceNum;
        hdr->nodeNum = nodeNum;
    }
    const trav::NodeData me = trav::load_node(nodes, i);
    WideNode w;
    const uint32_t ch[2] = {me.lft, me.rht};
    for (int s = 0; s < 2; ++s) {
        const uint32_t c = ch[s];
        AABB box;
        if (c & LEAF) {
            const uint32_t l = c & LEAF_MASK;
            const uint32_t k0 = lbvh->compacted_faces[l];
            const uint32_t k1 = (l == leafNum - 1) ? faceNum : lbvh->compacted_faces[l + 1];
            box = AABB::Initial();
            for (uint32_t k = k0; k < k1; ++k) {
                const TriRec &r = recs[k];
                box.Expand(F3(r.v0[0], r.v0[1], r.v0[2]));
                box.Expand(F3(r.v1[0], r.v1[1], r.v1[2]));
                box.Expand(F3(r.v2[0], r.v2[1], r.v2[2]));
            }
            box.minPos = box.minPos - pad;
            box.maxPos = box.maxPos + pad;
            w.child[s] = LEAF | k0;
            w.cnt[s] = (s == 1 && ch[0] == ch[1]) ? 0u : (k1 - k0); // single-leaf tree: test it once
        } else {
            box = trav::load_node(nodes, c).box;
            w.child[s] = c;
            w.cnt[s] = 0u;
        }
        for (int a = 0; a < 3; ++a) {
            w.box[s][a] = box.minPos[a];
            w.box[s][3 + a] = box.maxPos[a];
        }
    }
    uint4 *o = reinterpret_cast<uint4 *>(wide + i);
    const uint4 *src = reinterpret_cast<const uint4 *>(&w);
    o[0] = src[0];
    o[1] = src[1];
    o[2] = src[2];
    o[3] = src[3];
}

// ---- ray ordering ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t spread3(uint32_t v) { // 9 bits -> every third bit
    v &= 0x1ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = 27-bit Morton code of the origin inside the root box (9 bits per axis, clamped) << 3 | octant
__global__ __launch_bounds__(THREADS) void ray_keys_kernel(const Header *__restrict__ hdr,
                                                           const float *__restrict__ rays,
                                                           uint32_t rayNum,
                                                           uint32_t *__restrict__ keys) {
    const uint32_t r = blockIdx.x * THREADS + threadIdx.x;
    if (r >= rayNum)
        return;
    const float *p = rays + 6ull * r;
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = hdr->rootBox[a], hi = hdr->rootBox[3 + a];
        const float ext = hi - lo;
        float u = ext > 0.f ? (p[a] - lo) / ext : 0.f;
        u = fminf(fmaxf(u * 512.f, 0.f), 511.f); // NaN -> 0
        code |= spread3(static_cast<uint32_t>(u)) << (2 - a);
    }
    const uint32_t oct = (p[3] < 0.f ? 4u : 0u) | (p[4] < 0.f ? 2u : 0u) | (p[5] < 0.f ? 1u : 0u);
    keys[r] = (code << 3) | oct;
}

// ---- traversal ------------------------------------------------------------------------------------

constexpr int STACK_DEPTH = 32; // internal-node depth <= 30: every level fixes one more Morton bit
constexpr int LDS_DEPTH = 12;   // entries kept in LDS; deeper ones (rare) live in the workspace
constexpr int REFILL_BELOW = 40;
constexpr int PARK_MIN = 16; // run the triangle code once this many lanes wait for it ...
constexpr int STEP_MIN = 24; // ... or fewer than this many lanes can still step
constexpr uint32_t MAX_GRID = 1280; // 18 KB of LDS and < 96 VGPRs: five workgroups per CU


__device__ __forceinline__ void load_wide(const WideNode *__restrict__ wide, uint32_t i,
                                          uint32_t (&child)[2], uint32_t (&cnt)[2], AABB (&box)[2]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(wide + i);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    child[0] = a.x;
    child[1] = a.y;
    cnt[0] = a.z;
    cnt[1] = a.w;
    box[0].minPos = {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)};
    box[0].maxPos = {__uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.y)};
    box[1].minPos = {__uint_as_float(c.z), __uint_as_float(c.w), __uint_as_float(d.x)};
    box[1].maxPos = {__uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w)};
}

// Slab test with the ray's reciprocal direction. Each plane distance differs from the reference's
// (b - o) / d by at most 3 * 2^-24 relative (one rounding in 1/d, one in the product, one in the
// quotient it stands for), and min / max keep that bound, so the sign of enter - exit is only in
// doubt inside `SLAB_DOUBT`; there the caller repeats the test with the divisions.
constexpr float SLAB_DOUBT = 5e-7f;
struct FastRay {
    f3 pos, inv;
};
__device__ __forceinline__ void slab_fast(const FastRay &r, const AABB &b, float &en, float &ex,
                                          bool &doubt) {
    const float h0 = (b.minPos.x - r.pos.x) * r.inv.x, h1 = (b.maxPos.x - r.pos.x) * r.inv.x;
    const float h2 = (b.minPos.y - r.pos.y) * r.inv.y, h3 = (b.maxPos.y - r.pos.y) * r.inv.y;
    const float h4 = (b.minPos.z - r.pos.z) * r.inv.z, h5 = (b.maxPos.z - r.pos.z) * r.inv.z;
    en = fmaxf(fmaxf(fminf(h0, h1), fminf(h2, h3)), fminf(h4, h5));
    ex = fminf(fminf(fmaxf(h0, h1), fmaxf(h2, h3)), fmaxf(h4, h5));
    en = fmaxf(en, 0.f);
    doubt = fabsf(en - ex) <= SLAB_DOUBT * fmaxf(en, fabsf(ex)) + 1e-30f;
}

template <bool COUNT>
__device__ __forceinline__ void test_run(const TriRec *__restrict__ recs, const Ray &ray, uint32_t k0,
                                         uint32_t n, trav::Hit &best, uint32_t &triTests) {
    for (uint32_t k = k0; k < k0 + n; ++k) {
        const uint4 *p = reinterpret_cast<const uint4 *>(recs + k);
        const uint4 a = p[0], b = p[1], c = p[2];
        float bx = 0.f, by = 0.f;
        const float t = HitTriangle(
            ray, F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)),
            F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)),
            F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)), bx, by);
        if (COUNT)
            ++triTests;
        if (t >= 0.f && (t < best.t || (t == best.t && k < best.k))) {
            best.t = t;
            best.bx = bx;
            best.by = by;
            best.fi = a.w;
            best.k = k;
        }
    }
}

// Persistent wave64 workgroups pull rays (in sorted order) from a global counter; idle lanes are
// refilled with one ballot + one atomic per wave. A lane that reaches leaf triangles parks them
// (`pendCnt`) and waits; the wave runs the triangle code only when enough lanes are parked or too few
// can still step, so the two code paths are not both paid for on every iteration.
template <bool COUNT>
__global__ __launch_bounds__(THREADS) void closest_hit_accel_kernel(
    const Header *__restrict__ hdr, const WideNode *__restrict__ wide,
    const TriRec *__restrict__ recs, const float *__restrict__ rays,
    const uint32_t *__restrict__ order, uint32_t rayNum, float *__restrict__ tOut,
    float *__restrict__ baryOut, uint32_t *__restrict__ primOut, uint32_t *rayCounter,
    uint2 *__restrict__ spill, unsigned long long *counters) {
    // far children waiting to be visited: node index + entry distance cut to its upper 16 bits (a
    // lower bound of a non-negative float, so the test on pop can only keep more)
    __shared__ uint32_t s_node[LDS_DEPTH * THREADS];
    __shared__ unsigned short s_enter[LDS_DEPTH * THREADS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint32_t *stNode = s_node + tid;
    unsigned short *stEnter = s_enter + tid;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const size_t spillStride = static_cast<size_t>(gridDim.x) * THREADS;
    uint2 *mySpill = spill + static_cast<size_t>(blockIdx.x) * THREADS + tid;

    AABB rootBox;
    rootBox.minPos = {hdr->rootBox[0], hdr->rootBox[1], hdr->rootBox[2]};
    rootBox.maxPos = {hdr->rootBox[3], hdr->rootBox[4], hdr->rootBox[5]};

    bool active = false, poolEmpty = false;
    uint32_t rayIdx = 0;
    Ray ray;
    FastRay fray;
    bool exactOnly = false; // a direction component near zero: plane distances can be inf / NaN
    uint32_t cur = 0;       // node to visit next
    int sp = 0;
    bool walkDone = true;   // nothing left to visit (parked triangles may still be waiting)
    uint32_t pendK = 0, pendCnt = 0;
    trav::Hit best;
    uint32_t nodeVisits = 0, triTests = 0;
    unsigned long long totNodes = 0, totTris = 0, totRays = 0;
    unsigned long long dbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    while (true) {
        const unsigned long long idleMask = __ballot(!active);
        if (idleMask != 0ull && !poolEmpty) {
            const uint32_t want = __popcll(idleMask);
            const int leader = __ffsll(static_cast<long long>(idleMask)) - 1;
            uint32_t base = 0;
            if (lane == leader)
                base = atomicAdd(rayCounter, want);
            base = __shfl(base, leader, 64);
            if (base + want >= rayNum || base + want < base)
                poolEmpty = true;
            if (COUNT) { dbg[0] += 1; dbg[1] += want; }
            if (!active) {
                const uint32_t r = base + __popcll(idleMask & lanesBelow);
                if (r < rayNum && r >= base) {
                    rayIdx = order ? order[r] : r;
                    const float *p = rays + 6ull * rayIdx;
                    ray.pos = {p[0], p[1], p[2]};
                    ray.dir = {p[3], p[4], p[5]};
                    fray.pos = ray.pos;
                    fray.inv = 1.f / ray.dir;
                    exactOnly = !(fabsf(ray.dir.x) > 1e-18f && fabsf(ray.dir.y) > 1e-18f &&
                                  fabsf(ray.dir.z) > 1e-18f && fabsf(fray.inv.x) < 1e18f &&
                                  fabsf(fray.inv.y) < 1e18f && fabsf(fray.inv.z) < 1e18f);
                    nodeVisits = triTests = 0;
                    best.t = FloatMax;
                    best.bx = best.by = 0.f;
                    best.fi = best.k = 0xFFFFFFFFu;
                    float tEnter, tExit;
                    HitAABB(ray, rootBox, tEnter, tExit);
                    cur = 0;
                    sp = 0;
                    pendCnt = 0;
                    walkDone = tEnter > tExit;
                    active = true;
                }
            }
        }
        if (__ballot(active) == 0ull)
            break;

        while (true) {
            const bool canStep = active && !walkDone && pendCnt == 0u;
            if (COUNT) { dbg[2] += 1; const int c = __popcll(__ballot(canStep)); dbg[3] += (c != 0); dbg[4] += c; }
            if (canStep) {
                uint32_t child[2], cnt[2];
                AABB box[2];
                load_wide(wide, cur, child, cnt, box);
                if (COUNT)
                    ++nodeVisits;
                const float limit = best.t + trav::CULL_SLACK * best.t;
                float en[2], ex[2];
                bool hit[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bool doubt;
                    slab_fast(fray, box[s], en[s], ex[s], doubt);
                    hit[s] = !(en[s] > ex[s]);
                    if (cnt[s] != 0u)
                        hit[s] = hit[s] || doubt; // grown leaf box: only ever a filter
                    else if ((doubt || exactOnly) && !(child[s] & LEAF)) {
                        HitAABB(ray, box[s], en[s], ex[s]);
                        hit[s] = !(en[s] > ex[s]);
                    }
                    hit[s] = hit[s] && !(en[s] > limit);
                }
                // leaf children: park their triangles. Two leaf children are neighbouring runs of the
                // sorted order, so one (first, count) pair covers both.
                const bool leafL = cnt[0] != 0u && hit[0], leafR = cnt[1] != 0u && hit[1];
                if (leafL || leafR) {
                    pendK = (leafL ? child[0] : child[1]) & LEAF_MASK;
                    pendCnt = (leafL ? cnt[0] : 0u) + (leafR ? cnt[1] : 0u);
                }
                const bool goL = !(child[0] & LEAF) && hit[0];
                const bool goR = !(child[1] & LEAF) && hit[1];
                if (goL && goR && sp < STACK_DEPTH) {
                    const bool leftFirst = !(en[1] < en[0]);
                    const uint32_t farNode = leftFirst ? child[1] : child[0];
                    const uint32_t farEnter = __float_as_uint(leftFirst ? en[1] : en[0]) >> 16;
                    if (sp < LDS_DEPTH) {
                        stNode[sp * THREADS] = farNode;
                        stEnter[sp * THREADS] = static_cast<unsigned short>(farEnter);
                    } else
                        mySpill[(sp - LDS_DEPTH) * spillStride] = make_uint2(farNode, farEnter);
                    ++sp;
                    cur = leftFirst ? child[0] : child[1];
                } else if (goL) {
                    cur = child[0];
                } else if (goR) {
                    cur = child[1];
                } else {
                    walkDone = true;
                    while (sp > 0) {
                        if (COUNT) { const unsigned long long m = __ballot(1); if (lane == __ffsll((long long)m) - 1) { dbg[5] += 1; dbg[6] += __popcll(m); } }
                        --sp;
                        uint32_t node, enterBits;
                        if (sp < LDS_DEPTH) {
                            node = stNode[sp * THREADS];
                            enterBits = stEnter[sp * THREADS];
                        } else {
                            const uint2 e = mySpill[(sp - LDS_DEPTH) * spillStride];
                            node = e.x;
                            enterBits = e.y;
                        }
                        if (!(__uint_as_float(enterBits << 16) > limit)) {
                            cur = node;
                            walkDone = false;
                            break;
                        }
                    }
                }
            }
            // triangle phase, when it pays
            const int parked = __popcll(__ballot(pendCnt != 0u));
            const int stepping = __popcll(__ballot(active && !walkDone && pendCnt == 0u));
            if (parked != 0 && (parked >= PARK_MIN || stepping < STEP_MIN)) {
                if (COUNT) { dbg[7] += 1; dbg[8] += parked; }
                if (pendCnt != 0u) {
                    test_run<COUNT>(recs, ray, pendK, pendCnt, best, triTests);
                    pendCnt = 0u;
                }
            }
            if (active && walkDone && pendCnt == 0u) {
                tOut[rayIdx] = best.t;
                baryOut[2ull * rayIdx] = best.bx;
                baryOut[2ull * rayIdx + 1] = best.by;
                primOut[rayIdx] = best.fi;
                if (COUNT) {
                    totNodes += nodeVisits;
                    totTris += triTests;
                    ++totRays;
                }
                active = false;
            }
            const int alive = __popcll(__ballot(active));
            if (alive == 0 || (!poolEmpty && alive < REFILL_BELOW))
                break;
        }
    }

    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            totNodes += __shfl_xor(totNodes, o, 64);
            totTris += __shfl_xor(totTris, o, 64);
            totRays += __shfl_xor(totRays, o, 64);
        }
        for (int i = 0; i < 10; ++i) {
            unsigned long long v = dbg[i];
            if (i == 5 || i == 6) for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) atomicAdd(&counters[4 + i], v);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], totNodes);
            atomicAdd(&counters[1], totTris);
            atomicAdd(&counters[2], totRays);
        }
    }
}

struct TraceWs {
    size_t counter, spill, keysA, keysB, valsA, valsB, sortWs, sortWsBytes, total;
};
inline TraceWs trace_ws(uint32_t rayNum) {
    TraceWs w;
    Carver c;
    w.counter = c.take(256);
    w.spill = c.take(sizeof(uint2) * MAX_GRID * THREADS * (STACK_DEPTH - LDS_DEPTH));
    w.keysA = c.take(4ull * rayNum);
    w.keysB = c.take(4ull * rayNum);
    w.valsA = c.take(4ull * rayNum);
    w.valsB = c.take(4ull * rayNum);
    w.sortWsBytes = kk_sort_workspace_bytes(rayNum ? rayNum : 1);
    w.sortWs = c.take(w.sortWsBytes);
    w.total = c.off;
    return w;
}

} // namespace accel
} // namespace kk

extern "C" size_t kk_trace_accel_bytes(uint32_t face_num, uint32_t compacted_face_num) {
    return kk::accel::layout(face_num, compacted_face_num).total;
}

extern "C" int kk_trace_accel_build(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                                    const uint32_t *d_face_pos_idx, uint32_t face_num,
                                    uint32_t compacted_face_num, void *d_accel, size_t accel_bytes,
                                    void *stream_) {
    using namespace kk;
    using namespace kk::accel;
    KK_REQUIRE(d_lbvh && d_positions && d_face_pos_idx && d_accel, "null pointer");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES, "face_num out of range");
    KK_REQUIRE(compacted_face_num > 0 && compacted_face_num <= face_num, "leaf count out of range");
    KK_REQUIRE((reinterpret_cast<uintptr_t>(d_accel) & 255u) == 0, "d_accel must be 256-byte aligned");
    const Layout l = layout(face_num, compacted_face_num);
    KK_REQUIRE(accel_bytes >= l.total, "accel buffer too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *b = static_cast<char *>(d_accel);
    const uint32_t nodeNum = compacted_face_num > 1 ? compacted_face_num - 1 : 1;
    auto *recs = reinterpret_cast<TriRec *>(b + l.tris);
    pack_tris_kernel<<<div_up(face_num, THREADS), THREADS, 0, stream>>>(d_lbvh, d_positions,
                                                                        d_face_pos_idx, recs, face_num);
    KK_LAUNCH_CHECK();
    make_wide_kernel<<<div_up(nodeNum, THREADS), THREADS, 0, stream>>>(
        d_lbvh, recs, reinterpret_cast<WideNode *>(b + l.nodes), reinterpret_cast<Header *>(b + l.header),
        nodeNum);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" size_t kk_closest_hit_accel_workspace_bytes(uint32_t ray_num) {
    return kk::accel::trace_ws(ray_num).total;
}

extern "C" int kk_closest_hit_accel(const void *d_accel, uint32_t face_num,
                                    uint32_t compacted_face_num, const float *d_rays,
                                    uint32_t ray_num, float *d_t, float *d_bary, uint32_t *d_prim,
                                    uint32_t flags, unsigned long long *d_counters, void *d_workspace,
                                    size_t workspace_bytes, void *stream_) {
    using namespace kk;
    using namespace kk::accel;
    KK_REQUIRE(d_accel && d_rays && d_t && d_bary && d_prim && d_workspace, "null pointer");
    KK_REQUIRE(face_num > 0 && compacted_face_num > 0 && compacted_face_num <= face_num,
               "tree sizes out of range");
    KK_REQUIRE((flags & ~static_cast<uint32_t>(KK_TRACE_UNSORTED)) == 0, "unknown flag");
    if (ray_num == 0)
        return KK_OK;
    const TraceWs w = trace_ws(ray_num);
    KK_REQUIRE(workspace_bytes >= w.total, "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const Layout l = layout(face_num, compacted_face_num);
    const char *ab = static_cast<const char *>(d_accel);
    const auto *hdr = reinterpret_cast<const Header *>(ab + l.header);
    const auto *wide = reinterpret_cast<const WideNode *>(ab + l.nodes);
    const auto *recs = reinterpret_cast<const TriRec *>(ab + l.tris);
    char *b = static_cast<char *>(d_workspace);
    uint32_t *rayCounter = reinterpret_cast<uint32_t *>(b + w.counter);
    KK_HIP_TRY(hipMemsetAsync(rayCounter, 0, 256, stream));

    const uint32_t *order = nullptr;
    if (!(flags & KK_TRACE_UNSORTED)) {
        auto *keysA = reinterpret_cast<uint32_t *>(b + w.keysA);
        auto *valsA = reinterpret_cast<uint32_t *>(b + w.valsA);
        ray_keys_kernel<<<div_up(ray_num, THREADS), THREADS, 0, stream>>>(hdr, d_rays, ray_num, keysA);
        KK_LAUNCH_CHECK();
        int rc = rsort::sort_pairs(keysA, reinterpret_cast<uint32_t *>(b + w.keysB), valsA,
                                   reinterpret_cast<uint32_t *>(b + w.valsB), true, ray_num,
                                   b + w.sortWs, w.sortWsBytes, stream, nullptr);
        if (rc != KK_OK)
            return rc;
        order = valsA;
    }
    uint32_t grid = div_up(ray_num, THREADS);
    if (grid > MAX_GRID)
        grid = MAX_GRID;
    uint2 *spill = reinterpret_cast<uint2 *>(b + w.spill);
    if (d_counters)
        closest_hit_accel_kernel<true><<<grid, THREADS, 0, stream>>>(
            hdr, wide, recs, d_rays, order, ray_num, d_t, d_bary, d_prim, rayCounter, spill, d_counters);
    else
        closest_hit_accel_kernel<false><<<grid, THREADS, 0, stream>>>(
            hdr, wide, recs, d_rays, order, ray_num, d_t, d_bary, d_prim, rayCounter, spill, nullptr);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
