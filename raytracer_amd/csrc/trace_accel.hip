// Trace-side layout of a built LBVH and the closest-hit kernel that walks it.
//
// What decides a hit in the reference. rayCastLBVH (src/ray_tracer/ray_tracer_render.cu:192-270) tests
// a triangle when the ray passed the slab test (include/cg/ray.h:29-45) of every internal node above
// the triangle's leaf. A node's box is the exact union of its children's boxes (min / max of floats,
// src/ray_tracer/lbvh.cu:286-310), and b -> fl(fl(b - o) / d) is monotonic in b, so for a ray whose
// direction components are all non-zero the per-axis intervals of an ancestor contain those of a
// descendant and so do [tEnter, tExit]: passing the test of the leaf's PARENT node implies passing
// the test of every node above it. Hence
//     triangle k is a candidate of the reference  <=>  the ray passes the slab test of parent(leaf(k)),
// and the closest hit is the candidate with the smallest Moeller-Trumbore t (ties: smallest sorted
// position k, see traverse.h). How candidates are FOUND is free, as long as none is missed.
//
// So the binary tree is repacked once after the build, for finding candidates fast:
//   * Node4 (128 B, one cache line) per binary node n: the boxes and ids of n's four grandchildren
//     (a child that is a leaf takes one slot itself). Boxes are stored per axis (4 mins, 4 maxs) so a
//     lane fetches "near" and "far" planes directly by the sign of its ray direction. Leaf slots hold
//     the box of the leaf's triangles grown by a small pad.
//   * TriRec (48 B) per sorted triangle position k: three vertices, face index, index of the leaf's
//     parent node.
// The walk uses a reciprocal-multiply slab test made conservative by a relative margin (it may visit
// more than the reference, never less). A triangle whose Moeller-Trumbore test (same arithmetic as
// 3rd/glm/gtx/intersect.inl:28-94) beats the best hit is accepted only after the slab test of its
// leaf's parent box: decided by the fast test when that is outside its error margin, by the
// reference's division form otherwise. For a ray with a (near-)zero direction component plane
// distances become inf / NaN and the containment argument does not hold: such a ray walks with
// +-1e18 in place of the infinite reciprocal (still a superset of the reference's visits) and accepts
// a hit only after the reference's slab test of EVERY node from the leaf's parent up to the root.
// Rays are traced in Morton order of their origins (radix sort of (key, index)); outputs are written
// at the ray's own index.
#include "kk_host.h"
#include "kk_math.h"
#include "traverse.h"

namespace kk {
namespace rsort {
int sort_pairs(uint32_t *keysA, uint32_t *keysB, uint32_t *valsA, uint32_t *valsB, bool iota,
               uint32_t n, void *ws, size_t wsBytes, hipStream_t stream, uint32_t **errorFlagOut,
               int passes);
}
extern "C" size_t kk_sort_workspace_bytes(uint32_t n);

namespace accel {

using trav::LEAF;
using trav::LEAF_MASK;

constexpr int THREADS = 256;
constexpr float LEAF_PAD = 1e-5f; // leaf boxes grow by this fraction of the scene's largest coordinate
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

// 64 bytes: the four slot boxes are 8-bit offsets from `origin` in units of 2^exp per axis, rounded
// outwards by one extra unit, so a dequantised box always contains the float box it stands for.
struct alignas(64) Node4 {
    float origin[3];
    int8_t exp[3];
    uint8_t pad;
    uint8_t q[6][4];   // minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4]
    uint32_t child[4]; // binary node index | LEAF + first sorted position | EMPTY
    uint16_t cnt[4];   // triangles of a leaf slot (CNT_LONG: read the count from the first TriRec)
};
static_assert(sizeof(Node4) == 64, "half a cache line");
constexpr uint32_t CNT_LONG = 0xFFFFu;

struct alignas(16) TriRec {
    float v0[3];
    uint32_t fi;
    float v1[3];
    uint32_t parent; // binary node whose child is this triangle's leaf
    float v2[3];
    uint32_t run; // in the first record of a leaf's run: the number of triangles in the run
};
static_assert(sizeof(TriRec) == 48, "three 16-byte loads");

struct alignas(256) Header {
    float rootBox[6];
    uint32_t faceNum, nodeNum;
    const kk_lbvh_node *nodes; // the build's own nodes (parent boxes for the acceptance test)
};

struct Layout {
    size_t header, nodes, tris, parents, total;
};
inline Layout layout(uint32_t faceNum, uint32_t compactedNum) {
    Layout l;
    Carver c;
    const uint32_t nodeNum = compactedNum > 1 ? compactedNum - 1 : 1;
    l.header = c.take(sizeof(Header));
    l.nodes = c.take(sizeof(Node4) * static_cast<size_t>(nodeNum));
    l.tris = c.take(sizeof(TriRec) * static_cast<size_t>(faceNum));
    l.parents = c.take(sizeof(uint32_t) * static_cast<size_t>(nodeNum)); // parent of each binary node
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
    // .parent is written by make_node4_kernel
    float *o = reinterpret_cast<float *>(recs + k);
    o[0] = a.x, o[1] = a.y, o[2] = a.z;
    reinterpret_cast<uint32_t *>(o)[3] = fi;
    o[4] = b.x, o[5] = b.y, o[6] = b.z;
    o[8] = c.x, o[9] = c.y, o[10] = c.z;
    reinterpret_cast<uint32_t *>(o)[11] = 0u;
}

struct Slot {
    AABB box;
    uint32_t child, cnt;
};

__device__ __forceinline__ Slot empty_slot() {
    Slot s;
    s.box.minPos = F3(__builtin_inff());
    s.box.maxPos = F3(-__builtin_inff());
    s.child = EMPTY;
    s.cnt = 0u;
    return s;
}

__device__ __forceinline__ Slot leaf_slot(const kk_lbvh_device_data *__restrict__ lbvh,
                                          TriRec *__restrict__ recs, uint32_t leaf, uint32_t parent,
                                          float pad) {
    const uint32_t l = leaf & LEAF_MASK;
    const uint32_t k0 = lbvh->compacted_faces[l];
    const uint32_t k1 = (l == lbvh->compacted_face_num - 1) ? lbvh->face_num
                                                            : lbvh->compacted_faces[l + 1];
    Slot s;
    s.box = AABB::Initial();
    for (uint32_t k = k0; k < k1; ++k) {
        TriRec &r = recs[k];
        s.box.Expand(F3(r.v0[0], r.v0[1], r.v0[2]));
        s.box.Expand(F3(r.v1[0], r.v1[1], r.v1[2]));
        s.box.Expand(F3(r.v2[0], r.v2[1], r.v2[2]));
        r.parent = parent; // every writer of this word writes the same value
    }
    recs[k0].run = k1 - k0;
    s.box.minPos = s.box.minPos - pad;
    s.box.maxPos = s.box.maxPos + pad;
    s.child = LEAF | k0;
    s.cnt = k1 - k0;
    return s;
}

__global__ __launch_bounds__(THREADS) void make_node4_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, TriRec *__restrict__ recs,
    Node4 *__restrict__ out, uint32_t *__restrict__ parents, Header *__restrict__ hdr,
    uint32_t nodeNum) {
    const uint32_t n = blockIdx.x * THREADS + threadIdx.x;
    if (n >= nodeNum)
        return;
    const kk_lbvh_node *nodes = lbvh->internal_nodes;
    const trav::NodeData root = trav::load_node(nodes, 0);
    float scale = 0.f;
    for (int a = 0; a < 3; ++a) {
        scale = fmaxf(scale, fabsf(root.box.minPos[a]));
        scale = fmaxf(scale, fabsf(root.box.maxPos[a]));
        scale = fmaxf(scale, root.box.maxPos[a] - root.box.minPos[a]);
    }
    const float pad = LEAF_PAD * scale;
    if (n == 0) {
        for (int a = 0; a < 3; ++a) {
            hdr->rootBox[a] = root.box.minPos[a];
            hdr->rootBox[3 + a] = root.box.maxPos[a];
        }
        hdr->faceNum = lbvh->face_num;
        hdr->nodeNum = nodeNum;
        hdr->nodes = nodes;
    }
    const trav::NodeData me = trav::load_node(nodes, n);
    const uint32_t ch[2] = {me.lft, me.rht};
    if (n == 0)
        parents[0] = EMPTY;
    if (!(ch[0] & LEAF))
        parents[ch[0]] = n;
    if (!(ch[1] & LEAF))
        parents[ch[1]] = n;
    Slot slot[4];
    for (int s = 0; s < 2; ++s) {
        const uint32_t c = ch[s];
        if (s == 1 && ch[0] == ch[1] && (c & LEAF)) { // single-leaf tree: the leaf appears once
            slot[2] = slot[3] = empty_slot();
        } else if (c & LEAF) {
            slot[2 * s] = leaf_slot(lbvh, recs, c, n, pad);
            slot[2 * s + 1] = empty_slot();
        } else {
            const trav::NodeData g = trav::load_node(nodes, c);
            const uint32_t gc[2] = {g.lft, g.rht};
            for (int t = 0; t < 2; ++t) {
                if (gc[t] & LEAF)
                    slot[2 * s + t] = leaf_slot(lbvh, recs, gc[t], c, pad);
                else {
                    slot[2 * s + t].box = trav::load_node(nodes, gc[t]).box;
                    slot[2 * s + t].child = gc[t];
                    slot[2 * s + t].cnt = 0u;
                }
            }
        }
    }
    AABB u = AABB::Initial();
    for (int i = 0; i < 4; ++i)
        if (slot[i].child != EMPTY)
            u.Expand(slot[i].box);
    Node4 w;
    for (int a = 0; a < 3; ++a) {
        const float ext = u.maxPos[a] - u.minPos[a];
        const float r = ext / 253.f;
        int e = r > 0.f ? ilogbf(r) + 1 : -100; // 2^e > ext / 253: offsets stay below 254
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        const float inv = ldexpf(1.f, -e);
        w.origin[a] = u.minPos[a];
        w.exp[a] = static_cast<int8_t>(e);
        for (int i = 0; i < 4; ++i) {
            float lo = 255.f, hi = 0.f;
            if (slot[i].child != EMPTY) {
                lo = fmaxf(floorf((slot[i].box.minPos[a] - u.minPos[a]) * inv) - 1.f, 0.f);
                hi = fminf(ceilf((slot[i].box.maxPos[a] - u.minPos[a]) * inv) + 1.f, 255.f);
            }
            w.q[2 * a][i] = static_cast<uint8_t>(lo);
            w.q[2 * a + 1][i] = static_cast<uint8_t>(hi);
        }
    }
    w.pad = 0;
    for (int i = 0; i < 4; ++i) {
        w.child[i] = slot[i].child;
        w.cnt[i] = static_cast<uint16_t>(slot[i].cnt < CNT_LONG ? slot[i].cnt : CNT_LONG);
    }
    uint4 *o = reinterpret_cast<uint4 *>(out + n);
    const uint4 *src = reinterpret_cast<const uint4 *>(&w);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = src[i];
}

// ---- ray ordering ---------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t spread3(uint32_t v) { // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = Morton code of the origin inside the root box, KEY_BITS bits per axis (clamped); sorted with
// KEY_PASSES 8-bit passes
constexpr int KEY_BITS = 8;
constexpr int KEY_PASSES = (3 * KEY_BITS + 7) / 8;
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
        u = fminf(fmaxf(u * float(1 << KEY_BITS), 0.f), float((1 << KEY_BITS) - 1)); // NaN -> 0
        code |= spread3(static_cast<uint32_t>(u)) << (2 - a);
    }
    keys[r] = code;
}

// ---- traversal ------------------------------------------------------------------------------------

constexpr int STACK_DEPTH = 48; // <= 3 pushes per visited record, <= 15 records deep (tree depth <= 30)
constexpr int LDS_DEPTH = 20;   // entries kept in LDS; deeper ones (rare) live in the workspace
constexpr int REFILL_BELOW = 40;
constexpr int PARK_MIN = 16; // run the triangle code once this many lanes wait for it ...
constexpr int STEP_MIN = 24; // ... or fewer than this many lanes can still step
constexpr uint32_t MAX_GRID = 1280; // 30 KB of LDS and < 96 VGPRs: five workgroups per CU
// A product (b - o) * (1 / d) is within 3 * 2^-24 of the reference's fl((b - o) / d) (one rounding
// each in 1 / d, the product, and the quotient it stands for); min / max keep the bound.
constexpr float SLAB_REL = 1e-6f, SLAB_ABS = 1e-30f;

__device__ __forceinline__ float4 ld4(const char *p) { return *reinterpret_cast<const float4 *>(p); }

struct RayState {
    f3 pos, dir, inv;
};

// Does `r` pass the slab test of the build's node `parent`, as the reference computes it?
__device__ __forceinline__ bool passes_parent(const kk_lbvh_node *__restrict__ nodes, uint32_t parent,
                                              const RayState &r) {
    const trav::NodeData nd = trav::load_node(nodes, parent);
    const float h0 = (nd.box.minPos.x - r.pos.x) * r.inv.x, h1 = (nd.box.maxPos.x - r.pos.x) * r.inv.x;
    const float h2 = (nd.box.minPos.y - r.pos.y) * r.inv.y, h3 = (nd.box.maxPos.y - r.pos.y) * r.inv.y;
    const float h4 = (nd.box.minPos.z - r.pos.z) * r.inv.z, h5 = (nd.box.maxPos.z - r.pos.z) * r.inv.z;
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
    HitAABB(ray, nd.box, tEnter, tExit);
    return !(tEnter > tExit);
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
// returns t (< 0: miss); bx, by = weights of v1 and v2 like the reference's barycentrics
__device__ __forceinline__ float hit_watertight(const Shear &s, f3 org, f3 v0, f3 v1, f3 v2, float &bx,
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
    return t;
}

template <bool COUNT, bool WATERTIGHT>
__device__ __forceinline__ void test_run(const TriRec *__restrict__ recs,
                                         const kk_lbvh_node *__restrict__ nodes,
                                         const uint32_t *__restrict__ parents, const RayState &r,
                                         uint32_t mode, uint32_t k0, uint32_t n, trav::Hit &best,
                                         uint32_t &triTests) {
    const Ray ray{r.pos, r.dir};
    Shear shear{};
    if (WATERTIGHT && n != 0u)
        shear = make_shear(r.dir);
    for (uint32_t k = k0; k < k0 + n; ++k) {
        const uint4 *p = reinterpret_cast<const uint4 *>(recs + k);
        const uint4 a = p[0], b = p[1], c = p[2];
        float bx = 0.f, by = 0.f;
        const f3 v0 = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        const f3 v1 = F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
        const f3 v2 = F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
        const float t = WATERTIGHT ? hit_watertight(shear, r.pos, v0, v1, v2, bx, by)
                                   : HitTriangle(ray, v0, v1, v2, bx, by);
        if (COUNT)
            ++triTests;
        if (t >= 0.f && (t < best.t || (t == best.t && k < best.k))) {
            if (mode == MODE_FAST || (mode == MODE_CAREFUL ? passes_parent(nodes, b.w, r)
                                                           : passes_chain(nodes, parents, b.w, r))) {
                best.t = t;
                best.bx = bx;
                best.by = by;
                best.fi = a.w;
                best.k = k;
            }
        }
    }
}

// compare-exchange of (key, id) pairs, ascending by key
__device__ __forceinline__ void cswap(float &ka, uint32_t &ia, float &kb, uint32_t &ib) {
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const uint32_t i0 = sw ? ib : ia, i1 = sw ? ia : ib;
    ka = k0, kb = k1, ia = i0, ib = i1;
}

// Persistent wave64 workgroups pull rays (in sorted order) from a global counter. Per iteration every
// lane is at the same place in the code: (A) lanes that need a node pop one stack entry, (B) lanes
// with a node visit it, (C) when enough lanes have parked leaf triangles (or too few can still step)
// the parked lanes test them. A finished ray waits until the wave refills: then all finished lanes
// check their final hit against its leaf's parent box, write their result (or start the ray again
// in careful mode if the check fails) and take new rays with one ballot + one atomic per wave.
//
// Taking a better triangle hit without checking its parent box first is safe: the limit that prunes
// the walk is never below the final t, so nothing closer is skipped; if the final hit itself fails
// the check, the lane walks the ray again with every hit checked as it is taken.
template <bool COUNT, bool WATERTIGHT>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void closest_hit_accel_kernel(
    const Header *__restrict__ hdr, const Node4 *__restrict__ node4, const TriRec *__restrict__ recs,
    const float *__restrict__ rays, const uint32_t *__restrict__ order, uint32_t rayNum,
    float *__restrict__ tOut, float *__restrict__ baryOut, uint32_t *__restrict__ primOut,
    uint32_t *rayCounter, const uint32_t *__restrict__ parents,
    uint2 *__restrict__ spill, unsigned long long *counters) {
    // nodes waiting to be visited: node index + entry distance cut to its upper 16 bits (a lower
    // bound of a non-negative float, so the test on pop can only keep more)
    __shared__ uint32_t s_node[LDS_DEPTH * THREADS];
    __shared__ unsigned short s_enter[LDS_DEPTH * THREADS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint32_t *stNode = s_node + tid;
    unsigned short *stEnter = s_enter + tid;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const size_t spillStride = static_cast<size_t>(gridDim.x) * THREADS;
    uint2 *mySpill = spill + static_cast<size_t>(blockIdx.x) * THREADS + tid;
    const kk_lbvh_node *nodes = hdr->nodes;
    const char *node4Bytes = reinterpret_cast<const char *>(node4);

    AABB rootBox;
    rootBox.minPos = {hdr->rootBox[0], hdr->rootBox[1], hdr->rootBox[2]};
    rootBox.maxPos = {hdr->rootBox[3], hdr->rootBox[4], hdr->rootBox[5]};

    bool active = false;   // walking, or holding parked triangles
    bool finished = false; // walk over, result not yet written
    bool poolEmpty = false;
    uint32_t rayIdx = 0;
    RayState ray;
    bool negX = false, negY = false, negZ = false; // which of min / max is the near plane
    uint32_t cur = 0;                        // node to visit next
    bool needNode = false; // cur is used up: take the next node from the stack
    uint32_t mode = MODE_FAST; // how hits of this ray are checked (see test_run)
    int sp = 0;
    uint32_t pendK0 = 0, pendC0 = 0, pendK1 = 0, pendC1 = 0;
    trav::Hit best;
    uint32_t nodeVisits = 0, triTests = 0;
    unsigned long long totNodes = 0, totTris = 0, totRays = 0;

    auto push = [&](uint32_t node, float enter) {
        const uint32_t bits = __float_as_uint(enter) >> 16;
        if (sp < LDS_DEPTH) {
            stNode[sp * THREADS] = node;
            stEnter[sp * THREADS] = static_cast<unsigned short>(bits);
        } else if (sp < STACK_DEPTH)
            mySpill[(sp - LDS_DEPTH) * spillStride] = make_uint2(node, bits);
        ++sp;
    };

    while (true) {
        // ---- retire finished rays, refill idle lanes ------------------------------------------------
        if (finished) {
            bool ok = true;
            if (best.fi != 0xFFFFFFFFu && mode == MODE_FAST)
                ok = passes_parent(nodes, recs[best.k].parent, ray);
            finished = false;
            if (ok) {
                tOut[rayIdx] = best.t;
                baryOut[2ull * rayIdx] = best.bx;
                baryOut[2ull * rayIdx + 1] = best.by;
                primOut[rayIdx] = best.fi;
                if (COUNT) {
                    totNodes += nodeVisits;
                    totTris += triTests;
                    ++totRays;
                }
            } else { // walk this ray again, checking every hit when it is taken
                mode = MODE_CAREFUL;
                best.t = FloatMax;
                best.bx = best.by = 0.f;
                best.fi = best.k = 0xFFFFFFFFu;
                cur = 0;
                sp = 0;
                pendC0 = pendC1 = 0u;
                needNode = false;
                active = true;
            }
        }
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
            if (!active) {
                const uint32_t r = base + __popcll(idleMask & lanesBelow);
                if (r < rayNum && r >= base) {
                    rayIdx = order ? order[r] : r;
                    const float *p = rays + 6ull * rayIdx;
                    ray.pos = {p[0], p[1], p[2]};
                    ray.dir = {p[3], p[4], p[5]};
                    ray.inv = 1.f / ray.dir;
                    const bool regular = fabsf(ray.inv.x) < 1e18f && fabsf(ray.inv.y) < 1e18f &&
                                         fabsf(ray.inv.z) < 1e18f; // false for 0, denormal, NaN
                    mode = MODE_FAST;
                    if (!regular) {
                        // finite stand-ins keep the walk a superset of the reference's visits:
                        // inside a slab -> (-huge, +huge), outside -> both the same sign
                        mode = MODE_CHAIN;
                        if (!(fabsf(ray.inv.x) < 1e18f))
                            ray.inv.x = copysignf(1e18f, ray.dir.x);
                        if (!(fabsf(ray.inv.y) < 1e18f))
                            ray.inv.y = copysignf(1e18f, ray.dir.y);
                        if (!(fabsf(ray.inv.z) < 1e18f))
                            ray.inv.z = copysignf(1e18f, ray.dir.z);
                    }
                    {
                        negX = ray.inv.x < 0.f;
                        negY = ray.inv.y < 0.f;
                        negZ = ray.inv.z < 0.f;
                        nodeVisits = triTests = 0;
                        best.t = FloatMax;
                        best.bx = best.by = 0.f;
                        best.fi = best.k = 0xFFFFFFFFu;
                        float tEnter, tExit;
                        HitAABB(Ray{ray.pos, ray.dir}, rootBox, tEnter, tExit);
                        cur = 0;
                        sp = 0;
                        pendC0 = pendC1 = 0u;
                        needNode = false;
                        active = true;
                        if (tEnter > tExit) { // misses the scene: retire at the next refill
                            active = false;
                            finished = true;
                        }
                    }
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (__ballot(finished) != 0ull)
                continue; // only retirements left
            break;
        }

        while (true) {
            const float limit = best.t + trav::CULL_SLACK * best.t;
            // ---- (A) take the next node from the stack ---------------------------------------------
            if (active && needNode && (pendC0 | pendC1) == 0u) {
                if (sp > 0) {
                    --sp;
                    uint32_t node = 0, enterBits = 0x7f80u;
                    if (sp < LDS_DEPTH) {
                        node = stNode[sp * THREADS];
                        enterBits = stEnter[sp * THREADS];
                    } else if (sp < STACK_DEPTH) {
                        const uint2 e = mySpill[(sp - LDS_DEPTH) * spillStride];
                        node = e.x;
                        enterBits = e.y;
                    }
                    if (!(__uint_as_float(enterBits << 16) > limit)) {
                        cur = node;
                        needNode = false;
                    }
                } else { // walk over
                    active = false;
                    finished = true;
                }
            }
            // ---- (B) visit a node ------------------------------------------------------------------
            const bool canStep = active && !needNode && (pendC0 | pendC1) == 0u;
            if (canStep) {
                const uint4 *rec = reinterpret_cast<const uint4 *>(node4Bytes + (static_cast<size_t>(cur) << 6));
                const uint4 w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3];
                if (COUNT)
                    ++nodeVisits;
                // plane distance = q * (2^exp / d) + (origin - o) / d
                const float ax = ldexpf(ray.inv.x, static_cast<int>(static_cast<int8_t>(w0.w & 0xffu)));
                const float ay = ldexpf(ray.inv.y, static_cast<int>(static_cast<int8_t>((w0.w >> 8) & 0xffu)));
                const float az = ldexpf(ray.inv.z, static_cast<int>(static_cast<int8_t>((w0.w >> 16) & 0xffu)));
                const float bx = (__uint_as_float(w0.x) - ray.pos.x) * ray.inv.x;
                const float by = (__uint_as_float(w0.y) - ray.pos.y) * ray.inv.y;
                const float bz = (__uint_as_float(w0.z) - ray.pos.z) * ray.inv.z;
                // near / far plane bytes by the sign of the direction
                const uint32_t qnx = negX ? w1.y : w1.x, qfx = negX ? w1.x : w1.y;
                const uint32_t qny = negY ? w1.w : w1.z, qfy = negY ? w1.z : w1.w;
                const uint32_t qnz = negZ ? w2.y : w2.x, qfz = negZ ? w2.x : w2.y;
                const uint32_t child[4] = {w2.z, w2.w, w3.x, w3.y};
                uint32_t cnt[4] = {w3.z & 0xffffu, w3.z >> 16, w3.w & 0xffffu, w3.w >> 16};
                float key[4];
                uint32_t leafCnt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hnx = __builtin_fmaf(static_cast<float>((qnx >> (8 * i)) & 0xffu), ax, bx);
                    const float hny = __builtin_fmaf(static_cast<float>((qny >> (8 * i)) & 0xffu), ay, by);
                    const float hnz = __builtin_fmaf(static_cast<float>((qnz >> (8 * i)) & 0xffu), az, bz);
                    const float hfx = __builtin_fmaf(static_cast<float>((qfx >> (8 * i)) & 0xffu), ax, bx);
                    const float hfy = __builtin_fmaf(static_cast<float>((qfy >> (8 * i)) & 0xffu), ay, by);
                    const float hfz = __builtin_fmaf(static_cast<float>((qfz >> (8 * i)) & 0xffu), az, bz);
                    const float en = fmaxf(fmaxf(fmaxf(hnx, hny), hnz), 0.f);
                    const float ex = fminf(fminf(hfx, hfy), hfz);
                    // conservative: keep the slot unless the miss is beyond the rounding margin
                    // (a NaN, possible only with overflowing stand-ins, keeps the slot)
                    const bool hit = !(en > __builtin_fmaf(fabsf(ex), SLAB_REL, ex + SLAB_ABS)) &
                                     !(en > limit) & (child[i] != EMPTY);
                    const bool isLeaf = cnt[i] != 0u;
                    if (hit & (cnt[i] == CNT_LONG)) // a very long run: its length is in its first record
                        cnt[i] = recs[child[i] & LEAF_MASK].run;
                    leafCnt[i] = (hit & isLeaf) ? cnt[i] : 0u;
                    key[i] = (hit & !isLeaf) ? en : __builtin_inff();
                }
                // leaf slots: park their triangles. Slots 0,1 (and 2,3) that are both leaves are
                // neighbouring runs of the sorted order, so each pair parks one (first, count).
                pendK0 = (leafCnt[0] ? child[0] : child[1]) & LEAF_MASK;
                pendC0 = leafCnt[0] + leafCnt[1];
                pendK1 = (leafCnt[2] ? child[2] : child[3]) & LEAF_MASK;
                pendC1 = leafCnt[2] + leafCnt[3];
                // internal slots: nearest first, the others stacked farthest first
                uint32_t id[4] = {child[0], child[1], child[2], child[3]};
                cswap(key[0], id[0], key[1], id[1]);
                cswap(key[2], id[2], key[3], id[3]);
                cswap(key[0], id[0], key[2], id[2]);
                cswap(key[1], id[1], key[3], id[3]);
                cswap(key[1], id[1], key[2], id[2]);
                const float inf = __builtin_inff();
                if (sp + 3 <= LDS_DEPTH) {
                    // stores above the stack top are harmless: write all three, count the valid ones
                    stNode[sp * THREADS] = id[3];
                    stEnter[sp * THREADS] = static_cast<unsigned short>(__float_as_uint(key[3]) >> 16);
                    sp += key[3] < inf;
                    stNode[sp * THREADS] = id[2];
                    stEnter[sp * THREADS] = static_cast<unsigned short>(__float_as_uint(key[2]) >> 16);
                    sp += key[2] < inf;
                    stNode[sp * THREADS] = id[1];
                    stEnter[sp * THREADS] = static_cast<unsigned short>(__float_as_uint(key[1]) >> 16);
                    sp += key[1] < inf;
                } else {
                    if (key[3] < inf)
                        push(id[3], key[3]);
                    if (key[2] < inf)
                        push(id[2], key[2]);
                    if (key[1] < inf)
                        push(id[1], key[1]);
                }
                cur = id[0];
                needNode = !(key[0] < inf);
            }
            // ---- (C) triangle phase, when it pays --------------------------------------------------
            const int parked = __popcll(__ballot((pendC0 | pendC1) != 0u));
            const int stepping = __popcll(__ballot(active && (pendC0 | pendC1) == 0u));
            if (parked != 0 && (parked >= PARK_MIN || stepping < STEP_MIN)) {
                if ((pendC0 | pendC1) != 0u) {
                    test_run<COUNT, WATERTIGHT>(recs, nodes, parents, ray, mode, pendK0, pendC0, best, triTests);
                    test_run<COUNT, WATERTIGHT>(recs, nodes, parents, ray, mode, pendK1, pendC1, best, triTests);
                    pendC0 = pendC1 = 0u;
                }
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
    const size_t n = rayNum ? rayNum : 1;
    w.counter = c.take(256); // [0] ray queue head
    w.spill = c.take(sizeof(uint2) * MAX_GRID * THREADS * (STACK_DEPTH - LDS_DEPTH));
    w.keysA = c.take(4 * n);
    w.keysB = c.take(4 * n);
    w.valsA = c.take(4 * n);
    w.valsB = c.take(4 * n);
    w.sortWsBytes = kk_sort_workspace_bytes(static_cast<uint32_t>(n));
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
    make_node4_kernel<<<div_up(nodeNum, THREADS), THREADS, 0, stream>>>(
        d_lbvh, recs, reinterpret_cast<Node4 *>(b + l.nodes),
        reinterpret_cast<uint32_t *>(b + l.parents), reinterpret_cast<Header *>(b + l.header), nodeNum);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" size_t kk_closest_hit_accel_workspace_bytes(uint32_t ray_num) {
    return kk::accel::trace_ws(ray_num).total;
}

extern "C" int kk_closest_hit_accel(const void *d_accel, uint32_t face_num,
                                    uint32_t compacted_face_num,
                                    const float *d_rays, uint32_t ray_num, float *d_t, float *d_bary,
                                    uint32_t *d_prim, uint32_t flags,
                                    unsigned long long *d_counters, void *d_workspace,
                                    size_t workspace_bytes, void *stream_) {
    using namespace kk;
    using namespace kk::accel;
    KK_REQUIRE(d_accel && d_rays && d_t && d_bary && d_prim && d_workspace, "null pointer");
    KK_REQUIRE(face_num > 0 && compacted_face_num > 0 && compacted_face_num <= face_num,
               "tree sizes out of range");
    KK_REQUIRE((flags & ~static_cast<uint32_t>(KK_TRACE_UNSORTED | KK_TRACE_REUSE_ORDER |
                                               KK_TRACE_WATERTIGHT)) == 0,
               "unknown flag");
    if (ray_num == 0)
        return KK_OK;
    const TraceWs w = trace_ws(ray_num);
    KK_REQUIRE(workspace_bytes >= w.total, "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const Layout l = layout(face_num, compacted_face_num);
    const char *ab = static_cast<const char *>(d_accel);
    const auto *hdr = reinterpret_cast<const Header *>(ab + l.header);
    const auto *node4 = reinterpret_cast<const Node4 *>(ab + l.nodes);
    const auto *recs = reinterpret_cast<const TriRec *>(ab + l.tris);
    char *b = static_cast<char *>(d_workspace);
    uint32_t *rayCounter = reinterpret_cast<uint32_t *>(b + w.counter);
    KK_HIP_TRY(hipMemsetAsync(rayCounter, 0, 256, stream));

    const uint32_t *order = nullptr;
    if (flags & KK_TRACE_REUSE_ORDER) {
        order = reinterpret_cast<uint32_t *>(b + ((KEY_PASSES & 1) ? w.valsB : w.valsA));
    } else if (!(flags & KK_TRACE_UNSORTED)) {
        auto *keysA = reinterpret_cast<uint32_t *>(b + w.keysA);
        auto *valsA = reinterpret_cast<uint32_t *>(b + w.valsA);
        auto *valsB = reinterpret_cast<uint32_t *>(b + w.valsB);
        ray_keys_kernel<<<div_up(ray_num, THREADS), THREADS, 0, stream>>>(hdr, d_rays, ray_num, keysA);
        KK_LAUNCH_CHECK();
        int rc = rsort::sort_pairs(keysA, reinterpret_cast<uint32_t *>(b + w.keysB), valsA, valsB, true,
                                   ray_num, b + w.sortWs, w.sortWsBytes, stream, nullptr, KEY_PASSES);
        if (rc != KK_OK)
            return rc;
        order = (KEY_PASSES & 1) ? valsB : valsA;
    }
    uint32_t grid = div_up(ray_num, THREADS);
    if (grid > MAX_GRID)
        grid = MAX_GRID;
    uint2 *spill = reinterpret_cast<uint2 *>(b + w.spill);
    const auto *parents = reinterpret_cast<const uint32_t *>(ab + l.parents);
    const bool wt = (flags & KK_TRACE_WATERTIGHT) != 0;
#define KK_LAUNCH_TRACE(COUNT, WT)                                                                 \
    closest_hit_accel_kernel<COUNT, WT><<<grid, THREADS, 0, stream>>>(                              \
        hdr, node4, recs, d_rays, order, ray_num, d_t, d_bary, d_prim, rayCounter, parents, spill,  \
        d_counters)
    if (d_counters) {
        if (wt)
            KK_LAUNCH_TRACE(true, true);
        else
            KK_LAUNCH_TRACE(true, false);
    } else {
        if (wt)
            KK_LAUNCH_TRACE(false, true);
        else
            KK_LAUNCH_TRACE(false, false);
    }
#undef KK_LAUNCH_TRACE
    KK_LAUNCH_CHECK();
    return KK_OK;
}
