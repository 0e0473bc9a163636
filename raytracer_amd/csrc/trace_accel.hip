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
//   * Wide8 (128 B = one L2 line) per wide node: up to eight descendants of a binary node, chosen by
//     opening the child with the largest box until eight slots are filled. A slot is a binary subtree
//     (its wide node sits at childBase + its rank among the internal slots: children are contiguous)
//     or a leaf (a run of triangles). The slot boxes are 8-bit offsets from the record's origin in
//     units of 2^exp per axis, rounded outwards; leaf boxes are grown by a small pad first. Slots are
//     laid out by octant around the node's centre, so that "slot XOR ray octant" ascending is a
//     front-to-back order and no sorting is needed.
//   * TriRec (48 B) per triangle, stored in the order of the wide nodes (all leaf triangles of a wide
//     node are contiguous): three vertices, face index, the leaf's parent binary node, and the sorted
//     position k that resolves equal distances.
// Wide nodes are made level by level from the root: one launch per level, a thread per wide node,
// children and triangles allocated with one atomic per wave.
// The walk uses a reciprocal-multiply slab test made conservative by a relative margin (it may visit
// more than the reference, never less). A triangle hit is taken at once; when the ray retires its
// final hit is checked against the slab test of its leaf's parent box: decided by the fast test when
// that is outside its error margin, by the reference's division form otherwise (a failed check walks
// the ray again, checking every hit as it is taken). For a ray with a (near-)zero direction component
// plane distances become inf / NaN and the containment argument does not hold: such a ray walks with
// +-1e18 in place of the infinite reciprocal (still a superset of the reference's visits) and accepts
// a hit only after the reference's slab test of EVERY node from the leaf's parent up to the root.
// Rays are traced in Morton order of their origins (radix sort of (key, index)); outputs are written
// at the ray's own index.
#include "kk_host.h"
#include "kk_math.h"
#include "traverse.h"
#include "wide_walk.h"

#include "radix_hist.h"

#include <type_traits>

namespace kk {
extern "C" size_t kk_sort_workspace_bytes(uint32_t n);
extern "C" size_t kk_sort_workspace_error_offset(uint32_t n);

namespace accel {

constexpr int THREADS = 256;
constexpr float LEAF_PAD = 1e-5f; // leaf boxes grow by this fraction of the scene's largest coordinate

// A binary path fixes one more Morton bit per level, a wide level descends at least one binary level.
constexpr int MAX_LEVELS = 32;
struct alignas(256) BuildCtl {
    // reserved so far; one 64-bit word (low half: triangle records, high half: wide nodes) so that a
    // workgroup reserves both with a single atomic. Every reservation of a level goes to this one
    // address, and atomics on one address are served one after the other: a level of a million nodes is
    // bound by their number, not by the work.
    uint32_t triCount, wideCount;
    uint32_t bounds[MAX_LEVELS + 2]; // wide nodes of level l are [bounds[l], bounds[l + 1])
    uint32_t runCount;               // run nodes listed so far (Layout::runList)
};

// A leaf of the LBVH is a run of triangles with one Morton code; in a scene whose triangles are small
// against the 1/1024 grid such runs are long (dozens of faces), and a ray that touches the leaf's box
// would test them all. A run of SPLIT_MIN..SPLIT_MAX triangles therefore gets a wide node of its own
// (a "run node"): its triangles, ordered along the longest axis of their centres, in up to eight chunks
// with a box each. Which triangles are candidates is unchanged (every triangle still carries its
// leaf's parent for the acceptance test); fewer of them are met.
constexpr uint32_t SPLIT_MIN = 4, SPLIT_MAX = 120; // 8 chunks of at most 15 (the count nibbles)
constexpr uint32_t LINKS_IN_NODE = 0xFFFFFFFFu; // collapse: a subtree slot whose links are not packed
constexpr uint32_t RUN_FLAG = 0x40000000u;          // rootOf: low bits = leaf index instead of a binary node

struct Layout {
    size_t header, ctl, nodes, tris, parents, rootOf, leafParent, recSrc, runList, total;
    uint32_t wideMax;
};
inline Layout layout(uint32_t faceNum, uint32_t compactedNum) {
    Layout l;
    Carver c;
    const uint32_t nodeNum = compactedNum > 1 ? compactedNum - 1 : 1;
    // a wide node over binary nodes has >= 2 slots: at most nodeNum of them; a leaf with a run node has
    // >= SPLIT_MIN - 1 triangles beyond its first: at most (N - M) / (SPLIT_MIN - 1) of those
    l.wideMax = nodeNum + (faceNum - compactedNum) / (SPLIT_MIN - 1) + 1;
    l.header = c.take(sizeof(Header));
    l.ctl = c.take(sizeof(BuildCtl));
    l.nodes = c.take(sizeof(Wide8) * static_cast<size_t>(l.wideMax));
    l.tris = c.take(sizeof(TriRec) * static_cast<size_t>(faceNum));
    l.parents = c.take(sizeof(uint32_t) * static_cast<size_t>(nodeNum)); // parent of each binary node
    l.rootOf = c.take(sizeof(uint32_t) * static_cast<size_t>(l.wideMax)); // binary root (or run) of each wide node
    l.leafParent = c.take(sizeof(uint32_t) * static_cast<size_t>(compactedNum)); // of leaves with a run node
    // build scratch: (sorted position, leaf's parent) of every triangle record. The padded leaf boxes,
    // 24 B per leaf, borrow the triangle records' own space, which is written last.
    l.recSrc = c.take(sizeof(uint2) * static_cast<size_t>(faceNum));
    // the wide nodes that are run nodes, listed as they are reserved
    l.runList = c.take(sizeof(uint32_t) * (static_cast<size_t>(faceNum - compactedNum) / (SPLIT_MIN - 1) + 1));
    l.total = c.off;
    return l;
}

__global__ void init_build_kernel(const kk_lbvh_device_data *__restrict__ lbvh, Header *__restrict__ hdr,
                                  BuildCtl *__restrict__ ctl, uint32_t *__restrict__ rootOf,
                                  uint32_t nodeNum) {
    const kk_lbvh_node *nodes = lbvh->internal_nodes;
    const trav::NodeData root = trav::load_node(nodes, 0);
    for (int a = 0; a < 3; ++a) {
        hdr->rootBox[a] = root.box.minPos[a];
        hdr->rootBox[3 + a] = root.box.maxPos[a];
    }
    hdr->faceNum = lbvh->face_num;
    hdr->nodeNum = nodeNum;
    hdr->nodes = nodes;
    hdr->wideNum = 1;
    hdr->tooDeep = 0;
    ctl->wideCount = 1;
    ctl->triCount = 0;
    ctl->runCount = 0;
    for (int l = 0; l < MAX_LEVELS + 2; ++l)
        ctl->bounds[l] = l == 0 ? 0u : 1u;
    rootOf[0] = 0;
}

__global__ __launch_bounds__(THREADS) void parents_kernel(const kk_lbvh_device_data *__restrict__ lbvh,
                                                          uint32_t *__restrict__ parents,
                                                          uint32_t nodeNum) {
    const uint32_t n = blockIdx.x * THREADS + threadIdx.x;
    if (n >= nodeNum)
        return;
    const uint2 ch = *reinterpret_cast<const uint2 *>(lbvh->internal_nodes + n);
    if (n == 0)
        parents[0] = EMPTY;
    if (!(ch.x & LEAF))
        parents[ch.x] = n;
    if (!(ch.y & LEAF))
        parents[ch.y] = n;
}

// The box of every leaf (its triangles' vertices), 24 B per leaf, a thread per leaf in sorted order,
// for callers that do not hand over the LBVH build's own leaf boxes.
__device__ __forceinline__ float leaf_pad(const kk_lbvh_node *nodes) {
    const trav::NodeData root = trav::load_node(nodes, 0);
    float scale = 0.f;
    for (int a = 0; a < 3; ++a) {
        scale = fmaxf(scale, fabsf(root.box.minPos[a]));
        scale = fmaxf(scale, fabsf(root.box.maxPos[a]));
        scale = fmaxf(scale, root.box.maxPos[a] - root.box.minPos[a]);
    }
    return LEAF_PAD * scale;
}
__global__ __launch_bounds__(THREADS) void leaf_boxes_kernel(const kk_lbvh_device_data *__restrict__ lbvh,
                                                             const float *__restrict__ positions,
                                                             const uint32_t *__restrict__ tri,
                                                             float *__restrict__ leafBoxes) {
    const uint32_t faceNum = lbvh->face_num, leafNum = lbvh->compacted_face_num;
    const uint32_t l = blockIdx.x * THREADS + threadIdx.x;
    if (l >= leafNum)
        return;
    const uint32_t *__restrict__ faces = lbvh->faces;
    const uint32_t *__restrict__ compacted = lbvh->compacted_faces;
    const uint32_t k0 = compacted[l], k1 = (l == leafNum - 1) ? faceNum : compacted[l + 1];
    AABB b = AABB::Initial();
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t fi = faces[k];
        b.Expand(trav::load_pos(positions, tri[3ull * fi]));
        b.Expand(trav::load_pos(positions, tri[3ull * fi + 1]));
        b.Expand(trav::load_pos(positions, tri[3ull * fi + 2]));
    }
    float *o = leafBoxes + 6ull * l;
    o[0] = b.minPos.x, o[1] = b.minPos.y, o[2] = b.minPos.z;
    o[3] = b.maxPos.x, o[4] = b.maxPos.y, o[5] = b.maxPos.z;
}

// Triangle record w from its source (sorted position k, parent binary node of its leaf): a thread per
// record, after the last level (the records overwrite the leaf boxes).
__global__ __launch_bounds__(THREADS) void fill_records_kernel(const kk_lbvh_device_data *__restrict__ lbvh,
                                                               const float *__restrict__ positions,
                                                               const uint32_t *__restrict__ tri,
                                                               const uint2 *__restrict__ recSrc,
                                                               TriRec *__restrict__ recs) {
    const uint32_t w = blockIdx.x * THREADS + threadIdx.x;
    if (w >= lbvh->face_num)
        return;
    const uint2 src = recSrc[w];
    const uint32_t fi = lbvh->faces[src.x];
    const f3 a = trav::load_pos(positions, tri[3ull * fi]);
    const f3 b = trav::load_pos(positions, tri[3ull * fi + 1]);
    const f3 c = trav::load_pos(positions, tri[3ull * fi + 2]);
    uint4 *o = reinterpret_cast<uint4 *>(recs + w);
    o[0] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), fi);
    o[1] = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), src.y);
    o[2] = make_uint4(__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), src.x);
}

constexpr int BT = 256; // threads of a collapse workgroup: 36 KB of slot storage in LDS, four workgroups per CU

// Runs between two levels: the wide nodes reserved so far end the level that is built next.
__global__ void close_level_kernel(BuildCtl *ctl, Header *hdr, int level) {
    const uint32_t total = ctl->wideCount;
    ctl->bounds[level + 2] = total;
    hdr->wideNum = total;
}

// One level of wide nodes [start, end): thread w opens the binary subtree rootOf[w] into at most eight
// slots, reserves its children's wide nodes and its triangles' records, and writes the node and the
// records' sources. Slot boxes are quantised as they are met, against the subtree's own box (grown by
// the leaf pad), which contains them all. Workgroup `group` of `groups` takes every groups-th batch.
// The byte boxes of eight slots per thread: six bytes each, kept as a word and a half word so that a
// collapse workgroup of 256 threads stays within a quarter of a CU's LDS.
struct SlotBoxes {
    uint32_t lo[8][BT];       // minx miny minz maxx
    unsigned short hi[8][BT]; // maxy maxz
    __device__ __forceinline__ uint2 get(int j, int tid) const { return make_uint2(lo[j][tid], hi[j][tid]); }
    __device__ __forceinline__ void set(int j, int tid, uint2 v) {
        lo[j][tid] = v.x;
        hi[j][tid] = static_cast<unsigned short>(v.y);
    }
};
struct CollapseLds {
    SlotBoxes q;
    uint32_t ref[8][BT]; // binary node, or LEAF | leaf index
    uint32_t aux[8][BT]; // leaf: its parent binary node; subtree: half box area (float)
    uint2 waveTotal[BT / 64];    // x: triangle records, y: wide nodes asked for by a wave
    unsigned long long reserved; // what the workgroup's reservation returned
    uint32_t len[8][BT]; // leaf: its triangles; subtree: its two links in one word (or LINKS_IN_NODE)
};
// Quantisation frame of a wide node: its box grown by the pad, 2^ex per byte unit and axis.
struct QFrame {
    f3 org, inv, extQ; // origin, 2^-ex, the node's extent in byte units
    int ex[3];
};
__device__ __forceinline__ QFrame make_frame(const f3 &lo, const f3 &hi, float pad) {
    QFrame f;
    for (int a = 0; a < 3; ++a) {
        f.org[a] = lo[a] - pad;
        const float r = (hi[a] + pad - f.org[a]) / 253.f;
        int e = r > 0.f ? ilogbf(r) + 1 : -100; // 2^e > extent / 253: offsets stay below 254
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        f.ex[a] = e;
        f.inv[a] = ldexpf(1.f, -e);
        f.extQ[a] = (hi[a] + pad - f.org[a]) * f.inv[a];
    }
    return f;
}
// quantised outwards by one extra unit: the byte box contains the float box.
// x: minx miny minz maxx (bytes), y: maxy maxz
__device__ __forceinline__ uint2 quantise_box(const AABB &b, const QFrame &f) {
    uint32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = static_cast<uint32_t>(fminf(fmaxf(floorf((b.minPos[a] - f.org[a]) * f.inv[a]) - 1.f, 0.f), 255.f));
        hi[a] = static_cast<uint32_t>(fmaxf(fminf(ceilf((b.maxPos[a] - f.org[a]) * f.inv[a]) + 1.f, 255.f), 0.f));
    }
    return make_uint2(lo[0] | (lo[1] << 8) | (lo[2] << 16) | (hi[0] << 24), hi[1] | (hi[2] << 8));
}
// slot = octant of the child's centre around the node's centre, nearest free one; 4 bits per slot of the
// result: which of the cN children (byte boxes q[j * stride]) sits there (0xF: none)
__device__ __forceinline__ uint32_t assign_slots(int cN, const QFrame &f, const SlotBoxes &q, int tid) {
    uint32_t used = 0, childOf = 0xFFFFFFFFu;
    for (int j = 0; j < cN; ++j) {
        const uint2 b = q.get(j, tid);
        // (centre of the slot - centre of the node) * 2, in world units
        const float dx = ldexpf(static_cast<float>((b.x & 0xffu) + (b.x >> 24)) - f.extQ.x, f.ex[0]);
        const float dy = ldexpf(static_cast<float>(((b.x >> 8) & 0xffu) + (b.y & 0xffu)) - f.extQ.y, f.ex[1]);
        const float dz = ldexpf(static_cast<float>(((b.x >> 16) & 0xffu) + ((b.y >> 8) & 0xffu)) - f.extQ.z, f.ex[2]);
        int bestS = 0;
        float bestScore = -__builtin_inff();
        for (int sl = 0; sl < 8; ++sl) {
            if ((used >> sl) & 1u)
                continue;
            const float score = ((sl & 1) ? dx : -dx) + ((sl & 2) ? dy : -dy) + ((sl & 4) ? dz : -dz);
            if (!(score <= bestScore)) {
                bestScore = score;
                bestS = sl;
            }
        }
        used |= 1u << bestS;
        childOf = (childOf & ~(0xFu << (4 * bestS))) | (static_cast<uint32_t>(j) << (4 * bestS));
    }
    return childOf;
}
__device__ __forceinline__ void set_slot_box(Wide8 &rec, int sl, uint2 q) {
    rec.q[0][sl] = static_cast<uint8_t>(q.x), rec.q[1][sl] = static_cast<uint8_t>(q.x >> 24);
    rec.q[2][sl] = static_cast<uint8_t>(q.x >> 8), rec.q[3][sl] = static_cast<uint8_t>(q.y);
    rec.q[4][sl] = static_cast<uint8_t>(q.x >> 16), rec.q[5][sl] = static_cast<uint8_t>(q.y >> 8);
}
__device__ __forceinline__ void set_slot_empty(Wide8 &rec, int sl) { // cannot be hit
    rec.q[0][sl] = rec.q[2][sl] = rec.q[4][sl] = 255;
    rec.q[1][sl] = rec.q[3][sl] = rec.q[5][sl] = 0;
}
__device__ __forceinline__ void store_record(Wide8 *out, const Wide8 &rec) {
    uint4 *o = reinterpret_cast<uint4 *>(out);
    const uint4 *src = reinterpret_cast<const uint4 *>(&rec);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        o[i] = src[i];
}

struct CollapseArgs {
    const kk_lbvh_device_data *lbvh;
    const float *positions;  // the mesh, for the run nodes' chunk boxes
    const uint32_t *tri;
    const float *leafBoxes;
    Wide8 *out;
    uint2 *recSrc;
    uint32_t *rootOf;
    uint32_t *leafParent;
    BuildCtl *ctl;
    uint32_t *runList;
};
__device__ __forceinline__ void collapse_range(CollapseLds &lds, const CollapseArgs &args, uint32_t start,
                                               uint32_t end, uint32_t group, uint32_t groups) {
    const kk_lbvh_device_data *__restrict__ lbvh = args.lbvh;
    const float *__restrict__ leafBoxes = args.leafBoxes;
    Wide8 *__restrict__ out = args.out;
    uint2 *__restrict__ recSrc = args.recSrc;
    uint32_t *rootOf = args.rootOf;
    uint32_t *leafParent = args.leafParent;
    BuildCtl *ctl = args.ctl;
    auto &s_q = lds.q;
    auto &s_ref = lds.ref;
    auto &s_aux = lds.aux;
    const int tid = threadIdx.x, lane = tid & 63;
    const kk_lbvh_node *__restrict__ nodes = lbvh->internal_nodes;
    const uint32_t *__restrict__ compacted = lbvh->compacted_faces;
    const uint32_t faceNum = lbvh->face_num, leafNum = lbvh->compacted_face_num;

    const float pad = leaf_pad(nodes);
    auto runOf = [&](uint32_t leaf, uint32_t &k0, uint32_t &k1) {
        const uint32_t l = leaf & LEAF_MASK;
        k0 = compacted[l];
        k1 = (l == leafNum - 1) ? faceNum : compacted[l + 1];
    };
    auto splits = [](uint32_t len) { return len >= SPLIT_MIN && len <= SPLIT_MAX; }; // a run node of its own?
    for (uint32_t base = start + group * BT; base < end; base += groups * BT) {
        const uint32_t w = base + tid;
        const bool valid = w < end;
        int cN = 0;
        uint32_t childOf = 0xFFFFFFFFu;
        uint32_t imask = 0, leafMask = 0, ci = 0, ct = 0;
        QFrame fr{};
        const uint32_t root = valid ? rootOf[w] : 0u;
        const bool isRun = valid && (root & RUN_FLAG);
        if (isRun) { // made by run_nodes_kernel after the last level; here it reserves its triangles
            uint32_t k0, k1;
            runOf(root & ~RUN_FLAG, k0, k1);
            ct = k1 - k0;
        } else if (valid) {
            const uint32_t n = root;
            const trav::NodeData me = trav::load_node(nodes, n);
            fr = make_frame(me.box.minPos, me.box.maxPos, pad);
            // Everything a slot will be asked for later comes with the one round of loads that brings its
            // box: a leaf's triangle count, a subtree's two links (packed into one word when the right
            // child's index is the left one's plus one, as the builder numbers them; LINKS_IN_NODE: read
            // them from the node when it is opened). The loop below then waits for memory once per child
            // opened, not three times.
            auto put = [&](int j, uint32_t c, uint32_t parent) {
                AABB b;
                uint32_t aux, extra;
                if (c & LEAF) {
                    const uint32_t l = c & LEAF_MASK;
                    const float *lb = leafBoxes + 6ull * l;
                    const uint32_t k0 = compacted[l];
                    const uint32_t k1 = (l == leafNum - 1) ? faceNum : compacted[l + 1];
                    b.minPos = F3(lb[0], lb[1], lb[2]) - pad;
                    b.maxPos = F3(lb[3], lb[4], lb[5]) + pad;
                    aux = parent;
                    extra = k1 - k0;
                } else {
                    const trav::NodeData nd = trav::load_node(nodes, c);
                    b = nd.box;
                    const f3 e = b.maxPos - b.minPos;
                    aux = __float_as_uint(e.x * e.y + e.y * e.z + e.z * e.x);
                    const uint32_t li = nd.lft & LEAF_MASK, ri = nd.rht & LEAF_MASK;
                    extra = (ri == li + 1u && li < 0x3FFFFFFFu) ? (li | (nd.lft & LEAF) | ((nd.rht & LEAF) >> 1))
                                                              : LINKS_IN_NODE;
                }
                s_q.set(j, tid, quantise_box(b, fr));
                s_ref[j][tid] = c;
                s_aux[j][tid] = aux;
                lds.len[j][tid] = extra;
            };
            put(0, me.lft, n);
            cN = 1;
            if (!(me.lft == me.rht && (me.lft & LEAF))) { // a single-leaf tree names its leaf twice
                put(1, me.rht, n);
                cN = 2;
            }
            while (cN < 8) {
                int pick = -1;
                float bestArea = -1.f;
                for (int j = 0; j < cN; ++j)
                    if (!(s_ref[j][tid] & LEAF)) {
                        const float a = __uint_as_float(s_aux[j][tid]);
                        if (!(a <= bestArea)) { // also takes a NaN area: every subtree can be opened
                            bestArea = a;
                            pick = j;
                        }
                    }
                if (pick < 0)
                    break;
                const uint32_t c = s_ref[pick][tid];
                const uint32_t packed = lds.len[pick][tid];
                uint2 ch;
                if (packed != LINKS_IN_NODE) {
                    ch.x = (packed & 0x3FFFFFFFu) | (packed & LEAF);
                    ch.y = ((packed & 0x3FFFFFFFu) + 1u) | ((packed << 1) & LEAF);
                } else
                    ch = *reinterpret_cast<const uint2 *>(nodes + c);
                put(pick, ch.x, c);
                put(cN, ch.y, c);
                ++cN;
            }
            childOf = assign_slots(cN, fr, s_q, tid);
            for (int sl = 0; sl < 8; ++sl) {
                const uint32_t j = (childOf >> (4 * sl)) & 0xFu;
                if (j == 0xFu)
                    continue;
                const uint32_t ref = s_ref[j][tid];
                const uint32_t len = (ref & LEAF) ? lds.len[j][tid] : 0u;
                if ((ref & LEAF) && !splits(len)) {
                    leafMask |= 1u << sl;
                    ct += len;
                } else
                    imask |= 1u << sl;
            }
            ci = __popc(imask);
        }
        // one reservation per workgroup
        uint32_t ciIncl = ci, ctIncl = ct;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(ciIncl, o, 64), b = __shfl_up(ctIncl, o, 64);
            if (lane >= o) {
                ciIncl += a;
                ctIncl += b;
            }
        }
        const int wave = tid >> 6;
        if (lane == 63)
            lds.waveTotal[wave] = make_uint2(ctIncl, ciIncl);
        __syncthreads();
        if (tid == 0) {
            unsigned long long all = 0ull;
            for (int wv = 0; wv < BT / 64; ++wv)
                all += lds.waveTotal[wv].x | (static_cast<unsigned long long>(lds.waveTotal[wv].y) << 32);
            lds.reserved = all ? atomicAdd(reinterpret_cast<unsigned long long *>(&ctl->triCount), all) : 0ull;
        }
        __syncthreads();
        uint32_t triBase = static_cast<uint32_t>(lds.reserved) + (ctIncl - ct);
        uint32_t childBase = static_cast<uint32_t>(lds.reserved >> 32) + (ciIncl - ci);
        for (int wv = 0; wv < wave; ++wv) {
            triBase += lds.waveTotal[wv].x;
            childBase += lds.waveTotal[wv].y;
        }
        __syncthreads(); // the totals are written again in the next batch
        if (!valid)
            continue;

        if (isRun) {
            out[w].triBase = triBase;
            continue;
        }

        Wide8 rec;
        for (int a = 0; a < 3; ++a) {
            rec.origin[a] = fr.org[a];
            rec.exp[a] = static_cast<int8_t>(fr.ex[a]);
        }
        rec.imask = static_cast<uint8_t>(imask);
        rec.childBase = childBase;
        rec.triBase = triBase;
        uint32_t counts = 0, longest = 0;
        uint32_t rank = 0, run = 0;
        uint32_t firstOf[8]; // the leaf slots' first triangles, asked for together
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const uint32_t j = (childOf >> (4 * sl)) & 0xFu;
            const uint32_t ref = j == 0xFu ? 0u : s_ref[j][tid];
            firstOf[sl] = (ref & LEAF) ? compacted[ref & LEAF_MASK] : 0u;
        }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            rec.off16[sl] = rec.cnt16[sl] = 0;
            const uint32_t j = (childOf >> (4 * sl)) & 0xFu;
            if (j == 0xFu) {
                set_slot_empty(rec, sl);
                continue;
            }
            set_slot_box(rec, sl, s_q.get(j, tid));
            const uint32_t ref = s_ref[j][tid];
            if (!(ref & LEAF)) {
                rootOf[childBase + rank] = ref;
                ++rank;
                continue;
            }
            const uint32_t len = lds.len[j][tid];
            if (splits(len)) { // the slot keeps the leaf's box; its triangles hang one node further down
                rootOf[childBase + rank] = RUN_FLAG | (ref & LEAF_MASK);
                leafParent[ref & LEAF_MASK] = s_aux[j][tid];
                args.runList[atomicAdd(&ctl->runCount, 1u)] = childBase + rank;
                ++rank;
                continue;
            }
            const uint32_t k0 = firstOf[sl];
            const uint32_t parent = s_aux[j][tid];
            for (uint32_t k = k0; k < k0 + len; ++k)
                recSrc[triBase + run + (k - k0)] = make_uint2(k, parent);
            counts |= (len < 15u ? len : 15u) << (4 * sl);
            longest = len > longest ? len : longest;
            rec.off16[sl] = static_cast<uint16_t>(run < 65535u ? run : 65535u);
            rec.cnt16[sl] = static_cast<uint16_t>(len < 65535u ? len : 65535u);
            run += len;
        }
        const bool huge = ct > 65535u;
        rec.leafMask = leafMask | (longest > 15u ? LONG_RUN : 0u) | (huge ? HUGE_RUN : 0u);
        rec.counts = huge ? ct : counts;
        for (int i = 0; i < 4; ++i)
            rec.spare[i] = 0;
        store_record(out + w, rec);
    }
}

// The run nodes, after the last level: the leaf's triangles in the order of their centres along the
// longest axis of those (stable; -0 counts as +0, NaN keys last), cut into up to eight chunks of equal
// length, each with the box of its own triangles. The collapse listed the run nodes as it reserved them;
// a wave takes one run node at a time: two triangles per lane (a run has
// at most SPLIT_MAX = 120), corners and keys kept in LDS, a triangle's place found by counting the keys
// in front of it, the chunk boxes folded by one lane per chunk in the order of the chunk's triangles.
constexpr int RUN_WAVES = BT / 64;
constexpr int RUN_CAP = 128; // triangle slots of a wave (>= SPLIT_MAX, two per lane)
static_assert(SPLIT_MAX <= RUN_CAP, "two triangles per lane");
constexpr uint32_t RUN_GRID = 256; // 1024 waves share the list
__global__ __launch_bounds__(BT) void run_nodes_kernel(CollapseArgs args) {
    __shared__ SlotBoxes s_q; // column 64 * wave is the wave's
    __shared__ float s_corner[RUN_WAVES][9][RUN_CAP];
    __shared__ uint32_t s_key[RUN_WAVES][RUN_CAP];
    __shared__ uint32_t s_order[RUN_WAVES][RUN_CAP]; // triangle (within the run) at each place of the order
    __shared__ uint32_t s_chunkOut[RUN_WAVES][8];    // where a chunk's records start, relative to triBase
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t listed = args.ctl->runCount;
    if (blockIdx.x * RUN_WAVES >= listed)
        return;
    const kk_lbvh_device_data *__restrict__ lbvh = args.lbvh;
    const uint32_t *__restrict__ faces = lbvh->faces;
    const uint32_t leafNum = lbvh->compacted_face_num;
    const float pad = leaf_pad(lbvh->internal_nodes);
    const int col = wave * 64; // this wave's column of s_q
    float(*corner)[RUN_CAP] = s_corner[wave];
    uint32_t *keys = s_key[wave], *order = s_order[wave], *chunkOut = s_chunkOut[wave];
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (uint32_t e = blockIdx.x * RUN_WAVES + wave; e < listed; e += gridDim.x * RUN_WAVES) { // (uniform over the wave)
        const uint32_t w = args.runList[e];
        const uint32_t leaf = args.rootOf[w] & ~RUN_FLAG;
        const uint32_t k0 = lbvh->compacted_faces[leaf];
        const uint32_t k1 = leaf == leafNum - 1 ? lbvh->face_num : lbvh->compacted_faces[leaf + 1];
        const uint32_t n = k1 - k0; // SPLIT_MIN .. SPLIT_MAX
        const uint32_t parent = args.leafParent[leaf];
        const uint32_t triBase = args.out[w].triBase;
        const float *lb = args.leafBoxes + 6ull * leaf;
        const QFrame fr = make_frame(F3(lb[0], lb[1], lb[2]), F3(lb[3], lb[4], lb[5]), pad);
        // corners into LDS; the sums of a triangle's corners (three times its centre) stay in registers
        f3 sum[2];
        f3 cLo = F3(__builtin_inff()), cHi = F3(-__builtin_inff());
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t i = lane + 64u * h;
            sum[h] = F3(0.f);
            if (i < n) {
                const uint32_t fi = faces[k0 + i];
                const f3 a = trav::load_pos(args.positions, args.tri[3ull * fi]);
                const f3 b = trav::load_pos(args.positions, args.tri[3ull * fi + 1]);
                const f3 c = trav::load_pos(args.positions, args.tri[3ull * fi + 2]);
                corner[0][i] = a.x, corner[1][i] = a.y, corner[2][i] = a.z;
                corner[3][i] = b.x, corner[4][i] = b.y, corner[5][i] = b.z;
                corner[6][i] = c.x, corner[7][i] = c.y, corner[8][i] = c.z;
                sum[h] = a + b + c;
                for (int ax = 0; ax < 3; ++ax) {
                    cLo[ax] = fminf(cLo[ax], sum[h][ax]);
                    cHi[ax] = fmaxf(cHi[ax], sum[h][ax]);
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1)
            for (int ax = 0; ax < 3; ++ax) {
                cLo[ax] = fminf(cLo[ax], __shfl_xor(cLo[ax], o, 64));
                cHi[ax] = fmaxf(cHi[ax], __shfl_xor(cHi[ax], o, 64));
            }
        const f3 ce = cHi - cLo;
        const int axis = ce.x >= ce.y ? (ce.x >= ce.z ? 0 : 2) : (ce.y >= ce.z ? 1 : 2);
        // keys as unsigned words that order like the floats (-0 as +0; NaN beyond every number)
        uint32_t myKey[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t i = lane + 64u * h;
            const uint32_t bits = __float_as_uint(sum[h][axis] + 0.f);
            myKey[h] = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
            if (sum[h][axis] != sum[h][axis])
                myKey[h] = 0xFFFFFFFFu;
            if (i < n)
                keys[i] = myKey[h];
        }
        wave_sync();
        // place of triangle i = keys smaller than its own + equal keys in front of it
        uint32_t place[2] = {0u, 0u};
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t kj = keys[j];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = lane + 64u * h;
                place[h] += (kj < myKey[h] || (kj == myKey[h] && j < i)) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (lane + 64u * h < n)
                order[place[h]] = lane + 64u * h;
        wave_sync();
        const uint32_t cN = n < 8u ? n : 8u;
        const uint32_t each = n / cN, more = n % cN;
        auto chunk_at = [&](uint32_t j) { return j * each + (j < more ? j : more); };
        if (static_cast<uint32_t>(lane) < cN) { // the box of chunk `lane`, folded in the order of its triangles
            AABB b = AABB::Initial();
            for (uint32_t p = chunk_at(lane); p < chunk_at(lane + 1); ++p) {
                const uint32_t i = order[p];
                for (int c = 0; c < 3; ++c)
                    b.Expand(F3(corner[3 * c][i], corner[3 * c + 1][i], corner[3 * c + 2][i]));
            }
            b.minPos = b.minPos - pad;
            b.maxPos = b.maxPos + pad;
            s_q.set(lane, col, quantise_box(b, fr));
        }
        wave_sync();
        if (lane == 0) {
            const uint32_t childOf = assign_slots(static_cast<int>(cN), fr, s_q, col);
            Wide8 rec;
            for (int a = 0; a < 3; ++a) {
                rec.origin[a] = fr.org[a];
                rec.exp[a] = static_cast<int8_t>(fr.ex[a]);
            }
            rec.imask = 0;
            rec.childBase = 0;
            rec.triBase = triBase;
            uint32_t counts = 0, leafMask = 0, run = 0;
            for (int sl = 0; sl < 8; ++sl) {
                rec.off16[sl] = rec.cnt16[sl] = 0;
                const uint32_t j = (childOf >> (4 * sl)) & 0xFu;
                if (j == 0xFu) {
                    set_slot_empty(rec, sl);
                    continue;
                }
                set_slot_box(rec, sl, s_q.get(j, col));
                const uint32_t len = chunk_at(j + 1) - chunk_at(j); // <= 15
                chunkOut[j] = run;
                leafMask |= 1u << sl;
                counts |= len << (4 * sl);
                rec.off16[sl] = static_cast<uint16_t>(run);
                rec.cnt16[sl] = static_cast<uint16_t>(len);
                run += len;
            }
            rec.leafMask = leafMask;
            rec.counts = counts;
            for (int i = 0; i < 4; ++i)
                rec.spare[i] = 0;
            store_record(args.out + w, rec);
        }
        wave_sync();
        // the records' sources, chunk after chunk in slot order
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t i = lane + 64u * h;
            if (i < n) {
                const uint32_t p = place[h];
                const uint32_t big = more * (each + 1u); // places in the chunks that are one longer
                const uint32_t j = p < big ? p / (each + 1u) : more + (p - big) / each;
                args.recSrc[triBase + chunkOut[j] + (p - chunk_at(j))] = make_uint2(k0 + i, parent);
            }
        }
        wave_sync(); // the wave's LDS rows are written again for its next run node
    }
}

__global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(4))) void collapse_level_kernel(CollapseArgs args, int level) {
    __shared__ CollapseLds lds;
    const uint32_t start = args.ctl->bounds[level], end = args.ctl->bounds[level + 1];
    if (start >= end)
        return;
    collapse_range(lds, args, start, end, blockIdx.x, gridDim.x);
}

// Levels from `level` on, in one workgroup: past the first dozen levels a tree has few nodes left (or
// none), and a launch per level would cost more than the work. What one level wrote is read by the
// next after a device-scope fence and a barrier.
__global__ __launch_bounds__(BT) void collapse_tail_kernel(CollapseArgs args, Header *hdr, int level) {
    BuildCtl *ctl = args.ctl;
    __shared__ CollapseLds lds;
    __shared__ uint32_t s_range[2];
    for (; level < MAX_LEVELS; ++level) {
        if (threadIdx.x == 0) {
            s_range[0] = __atomic_load_n(&ctl->bounds[level], __ATOMIC_RELAXED);
            s_range[1] = __atomic_load_n(&ctl->bounds[level + 1], __ATOMIC_RELAXED);
        }
        __syncthreads();
        const uint32_t start = s_range[0], end = s_range[1];
        if (start >= end)
            return; // an empty level: nothing below it either
        collapse_range(lds, args, start, end, 0u, 1u);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) { // what close_level_kernel does between launches
            const uint32_t total = __atomic_load_n(&ctl->wideCount, __ATOMIC_RELAXED);
            __atomic_store_n(&ctl->bounds[level + 2], total, __ATOMIC_RELAXED);
            hdr->wideNum = total;
        }
        __threadfence();
        __syncthreads();
    }
    // nodes reserved for a level that is never built: a path of more than MAX_LEVELS wide levels
    if (threadIdx.x == 0 && __atomic_load_n(&ctl->bounds[MAX_LEVELS], __ATOMIC_RELAXED) <
                                __atomic_load_n(&ctl->bounds[MAX_LEVELS + 1], __ATOMIC_RELAXED))
        hdr->tooDeep = 1;
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
constexpr int KEY_BITS = 5;
constexpr int KEY_PASSES = (3 * KEY_BITS + 7) / 8;
constexpr uint32_t KEYS_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(THREADS) void ray_keys_kernel(const Header *__restrict__ hdr,
                                                           const float *__restrict__ rays,
                                                           uint32_t rayNum,
                                                           uint32_t *__restrict__ keys,
                                                           uint32_t *__restrict__ ghist) {
    // the digits of the keys are counted here for the sort that follows (radix_hist.h)
    __shared__ uint32_t s_hist[KEY_PASSES * rsort::HIST_RADIX];
    rsort::hist_clear<KEY_PASSES>(s_hist, threadIdx.x, THREADS);
    for (uint32_t r = blockIdx.x * THREADS + threadIdx.x; r < rayNum; r += gridDim.x * THREADS) {
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
        rsort::hist_add<KEY_PASSES>(s_hist, code);
    }
    rsort::hist_flush<KEY_PASSES>(s_hist, ghist, threadIdx.x, THREADS);
}

// ---- traversal ------------------------------------------------------------------------------------

constexpr int STACK_DEPTH = 48; // <= 1 push per visited record, <= MAX_LEVELS records deep
constexpr int LDS_DEPTH = 20;   // entries kept in LDS; deeper ones (rare) live in the workspace
constexpr int LDS_DEPTH_BOUND = 15; // ... when an entry carries its entry bound (8 bytes instead of 6)
constexpr int REFILL_BELOW = 40; // (24: 1.167 ms a launch, 32: 1.142, 40: 1.137, 48: 1.148, 56: 1.182)
constexpr int PARK_MIN = 16; // run the triangle code once this many lanes wait for it ...
constexpr int STEP_MIN = 24; // ... or fewer than this many lanes can still step
constexpr uint32_t MAX_GRID = 1280; // most workgroups the spill area is made for (five per CU on 256 CUs)
constexpr uint32_t GROUPS_PER_CU = 5; // 30 KB of LDS and < 96 VGPRs: five resident workgroups per CU
__device__ __forceinline__ float4 ld4(const char *p) { return *reinterpret_cast<const float4 *>(p); }
constexpr uint32_t QUEUES = 8;        // ray queues (one per XCD of an MI355X; a wave starts at queue
                                      // XCC_ID % QUEUES and moves on when it is used up, so any XCD count works)
constexpr uint32_t QUEUE_STRIDE = 32; // counters 128 bytes apart
// which XCD this wave runs on (hardware register XCC_ID, bits 3:0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }

// Persistent wave64 workgroups pull rays (in sorted order) from a global counter. Per iteration every
// lane is at the same place in the code: (A) lanes without a node take the next child of their current
// group (or pop a group), (B) lanes with a node visit it, (C) when enough lanes have parked leaf
// triangles (or too few can still step) the parked lanes test them. A group is what one visit found:
// the node's childBase, its internal-slot mask and the hit slots still to be entered, kept in
// front-to-back order of the ray's octant, so a visit pushes at most one stack entry. A finished ray
// waits until the wave refills: then all finished lanes check their final hit against its leaf's
// parent box, write their result (or start the ray again in careful mode if the check fails) and take
// new rays with one ballot + one atomic per wave.
//
// Taking a better triangle hit without checking its parent box first is safe: the limit that prunes
// the walk is never below the final t, so nothing closer is skipped; if the final hit itself fails
// the check, the lane walks the ray again with every hit checked as it is taken.
//
// BOUND (KK_TRACE_ENTRY_BOUND): a group also carries a lower bound of the entry distance of its pending
// slots - the smallest entry among the hit subtree slots of its visit other than the one entered first,
// 16 bits (the float's upper half, i.e. rounded down) in the stack entry - and what is left of a group is
// dropped without a visit once that bound lies behind the best hit. Same hits (a slot dropped this way
// would have been dropped by its record's visit had the limit been known then). On the 10 M-triangle
// soup it saves 4.05 of 29.7 visits per ray and costs as many instructions per visit as it saves
// visits: 1.225 ms against 1.178 ms per launch, so it is an option, not the default.
template <bool COUNT, bool WATERTIGHT, bool BOUND>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void closest_hit_accel_kernel(
    const Header *__restrict__ hdr, const Wide8 *__restrict__ wide, const TriRec *__restrict__ recs,
    const float *__restrict__ rays, const uint32_t *__restrict__ order, uint32_t rayNum,
    float *__restrict__ tOut, float *__restrict__ baryOut, uint32_t *__restrict__ primOut,
    uint32_t *rayCounter, const uint32_t *__restrict__ parents,
    uint2 *__restrict__ spill, unsigned long long *counters, const uint32_t *sortError, uint32_t *status) {
    // groups waiting to be entered: childBase + (pending slots | internal-slot mask << 8)
    // (BOUND: the second word is 32 bits, its upper half the entry bound)
    constexpr int LDS_DEPTH = BOUND ? LDS_DEPTH_BOUND : accel::LDS_DEPTH;
    using BitsWord = typename std::conditional<BOUND, uint32_t, unsigned short>::type;
    __shared__ uint32_t s_base[LDS_DEPTH * THREADS];
    __shared__ BitsWord s_bits[LDS_DEPTH * THREADS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint32_t *stBase = s_base + tid;
    BitsWord *stBits = s_bits + tid;
    const size_t spillStride = static_cast<size_t>(gridDim.x) * THREADS;
    uint2 *mySpill = spill + static_cast<size_t>(blockIdx.x) * THREADS + tid;
    // A ray sort whose look-back gave up leaves an order that is no permutation of the rays: the rays are
    // then traced in input order (the order only decides how fast, not what is found) and the call's
    // status word says so.
    const uint32_t sortFailed = (order != nullptr && sortError != nullptr) ? *sortError : 0u;
    if (sortFailed != 0u)
        order = nullptr;
    if (blockIdx.x == 0 && tid == 0)
        *status = sortFailed;
    const kk_lbvh_node *nodes = hdr->nodes;
    const char *wideBytes = reinterpret_cast<const char *>(wide);

    AABB rootBox;
    rootBox.minPos = {hdr->rootBox[0], hdr->rootBox[1], hdr->rootBox[2]};
    rootBox.maxPos = {hdr->rootBox[3], hdr->rootBox[4], hdr->rootBox[5]};

    // Rays are handed out from one queue per XCD (consecutive eighths of the sorted order, i.e. of
    // space): the workgroups of an XCD share its L2, so they should walk the same part of the tree.
    // A wave whose queue is used up moves on to the next one.
    const uint32_t queueLen = (rayNum + QUEUES - 1) / QUEUES;
    uint32_t queue = xcc_id() % QUEUES, queuesTried = 0;

    bool active = false;   // walking, or holding parked triangles
    bool finished = false; // walk over, result not yet written
    bool poolEmpty = false;
    uint32_t rayIdx = 0;
    RayState ray;
    RaySigns sg{false, false, false, 0u};          // which of min / max is the near plane; octant
    Shear shear{};                                 // WATERTIGHT: the ray's shear, made once per ray
    uint32_t cur = 0;                              // node to visit next
    bool have = false;                             // cur is valid
    float grpMin = 0.f;                            // lower bound of the entry distance of the group's pending slots,
    constexpr uint32_t FRESH = BOUND ? 0x10000u : 0u; // which does not cover the group's first slot while that is pending
                                                   // (bit 16 of grpBits, in the registers only)
    uint32_t grpBase = 0, grpBits = 0;             // current group (see above); bits 16-31: entry bound
    uint32_t mode = MODE_FAST;                     // how hits of this ray are checked (see test_run)
    int sp = 0;
    uint32_t pendW = 0, pendC = 0; // parked triangles
    Hit8 best;
    uint32_t nodeVisits = 0, triTests = 0;
    unsigned long long totNodes = 0, totTris = 0, totRays = 0;
    unsigned long long dbgIters = 0, dbgVisitLanes = 0, dbgTriRuns = 0, dbgTriLanes = 0, dbgEmpty = 0, dbgEmptyLimit = 0, dbgUsedSlots = 0, dbgLeafOnly = 0, dbgCulled = 0;

    // COUNT: when the wave started, when it first found every queue used up, when it left (wall clock, 100 MHz)
    const unsigned long long clkStart = COUNT ? wall_clock64() : 0ull;
    unsigned long long clkEmpty = 0ull;

    auto restart = [&]() {
        best.t = FloatMax;
        best.bx = best.by = 0.f;
        best.fi = best.k = best.w = 0xFFFFFFFFu;
        best.sure = false;
        sp = 0;
        pendC = 0u;
        // a group that holds just the root: slot 0 of a node whose only internal slot is slot 0
        grpBase = 0u;
        grpBits = (1u << (0u ^ sg.oct)) | (1u << 8) | FRESH;
        grpMin = 0.f;
        have = false;
        active = true;
    };

    while (true) {
        // ---- retire finished rays, refill idle lanes ------------------------------------------------
        if (finished) {
            bool ok = true;
            if (best.fi != 0xFFFFFFFFu && mode == MODE_FAST && !best.sure)
                ok = passes_parent(nodes, recs[best.w].parent, ray);
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
                restart();
            }
        }
        const unsigned long long idleMask = __ballot(!active);
        if (idleMask != 0ull && !poolEmpty) {
            const uint32_t want = __popcll(idleMask);
            const int leader = __ffsll(static_cast<long long>(idleMask)) - 1;
            const uint32_t qBegin = queue * queueLen;
            const uint32_t qLen = qBegin >= rayNum ? 0u : (rayNum - qBegin < queueLen ? rayNum - qBegin : queueLen);
            uint32_t base = 0;
            if (lane == leader)
                base = atomicAdd(rayCounter + queue * QUEUE_STRIDE, want);
            base = __shfl(base, leader, 64);
            if (base + want >= qLen || base + want < base) { // used up: the next refill asks the next queue
                queue = queue + 1 == QUEUES ? 0u : queue + 1;
                poolEmpty = ++queuesTried == QUEUES;
                if (COUNT && poolEmpty)
                    clkEmpty = wall_clock64();
            }
            if (!active) {
                // idle lanes below this one (mbcnt: no lane mask kept in registers)
                const uint32_t rel = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idleMask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idleMask), 0u));
                if (rel < qLen && rel >= base) {
                    const uint32_t r = qBegin + rel;
                    rayIdx = order ? order[r] : r;
                    rayIdx = rayIdx < rayNum ? rayIdx : rayNum - 1u; // (an order kept from other rays)
                    const float *p = rays + 6ull * rayIdx;
                    prepare_ray(F3(p[0], p[1], p[2]), F3(p[3], p[4], p[5]), ray, sg, mode);
                    if (WATERTIGHT)
                        shear = make_shear(ray.dir);
                    nodeVisits = triTests = 0;
                    restart();
                    float tEnter, tExit;
                    HitAABB(Ray{ray.pos, ray.dir}, rootBox, tEnter, tExit);
                    if (tEnter > tExit || !ray_is_finite(ray.pos, ray.dir)) { // misses the scene (or is no ray): retire at the next refill
                        active = false;
                        finished = true;
                    } else {
                        // Every ray's first record is the root: visited here, where its address is the
                        // same for the whole wave and the record comes through the scalar cache instead
                        // of five per-lane loads.
                        bool anyNoLimit = false;
                        const Visit vis = visit_record<false, BOUND>(wideBytes, 0u, ray, sg, FloatMax, anyNoLimit);
                        if (COUNT)
                            ++nodeVisits;
                        pendW = vis.leafW;
                        pendC = vis.leafC;
                        grpBase = vis.childBase;
                        grpBits = octant_order(vis.inner, sg.oct) | (vis.imask << 8) | FRESH;
                        grpMin = vis.restEntry;
                        if ((vis.inner | vis.leafC) == 0u) { // nothing of the root is hit
                            active = false;
                            finished = true;
                        }
                    }
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (__ballot(finished) != 0ull || !poolEmpty)
                continue; // retirements left, or another queue to ask
            break;
        }

        while (true) {
            const float limit = best.t + trav::CULL_SLACK * best.t;
            // ---- (A) next node: from the current group, else from the stack ------------------------
            if (active && !have && pendC == 0u) {
                // what is left of the group lies behind the best hit by now: dropped unvisited
                if (BOUND && !(grpBits & FRESH) && grpMin * SLAB_SHRINK > limit) {
                    if (COUNT)
                        dbgCulled += __popc(grpBits & 0xffu);
                    grpBits &= ~0xffu;
                }
                while ((grpBits & 0xffu) == 0u && sp > 0) {
                    --sp;
                    if (sp < LDS_DEPTH) {
                        grpBase = stBase[sp * THREADS];
                        grpBits = stBits[sp * THREADS];
                    } else if (sp < STACK_DEPTH) {
                        const uint2 e = mySpill[(sp - LDS_DEPTH) * spillStride];
                        grpBase = e.x;
                        grpBits = e.y;
                    }
                    if (!BOUND)
                        break; // one group per pass
                    grpMin = __uint_as_float(grpBits & 0xFFFF0000u);
                    grpBits &= 0xFFFFu;
                    if (grpMin * SLAB_SHRINK > limit) {
                        if (COUNT)
                            dbgCulled += __popc(grpBits & 0xffu);
                        grpBits &= ~0xffu;
                    }
                }
                if ((grpBits & 0xffu) == 0u) { // walk over
                    active = false;
                    finished = true;
                } else {
                    const uint32_t i = static_cast<uint32_t>(__ffs(static_cast<int>(grpBits))) - 1u;
                    grpBits &= (grpBits - 1u) & (BOUND ? 0xFFFFu : ~0u); // the lowest set bit is a pending slot
                    const uint32_t slot = i ^ sg.oct;
                    cur = grpBase + __popc((grpBits >> 8) & ((1u << slot) - 1u));
                    have = true;
                }
            }
            // ---- (B) visit a node ------------------------------------------------------------------
            if (COUNT) {
                ++dbgIters;
                dbgVisitLanes += __popcll(__ballot(have));
            }
            if (have) {
                have = false;
                if (COUNT)
                    ++nodeVisits;
                bool anyNoLimit = false;
                const Visit vis = visit_record<COUNT, BOUND>(wideBytes, cur, ray, sg, limit, anyNoLimit);
                if (COUNT) {
                    dbgUsedSlots += __popc(vis.imask | vis.leafSlots);
                    if (vis.imask == 0u)
                        ++dbgLeafOnly;
                }
                if (COUNT && (vis.leafC | vis.inner) == 0u) {
                    ++dbgEmpty; // visits that found nothing
                    if (anyNoLimit)
                        ++dbgEmptyLimit; // ... because of the distance limit
                }
                if (vis.leafC != 0u) { // park the triangles from the first to the last leaf slot that was hit
                    pendW = vis.leafW;
                    pendC = vis.leafC;
                }
                if (vis.inner != 0u) {
                    if ((grpBits & 0xffu) != 0u) {
                        const uint32_t word = (grpBits & 0xFFFFu) | (BOUND ? __float_as_uint(grpMin) & 0xFFFF0000u : 0u);
                        if (sp < LDS_DEPTH) {
                            stBase[sp * THREADS] = grpBase;
                            stBits[sp * THREADS] = static_cast<BitsWord>(word);
                        } else if (sp < STACK_DEPTH)
                            mySpill[(sp - LDS_DEPTH) * spillStride] = make_uint2(grpBase, word);
                        ++sp;
                    }
                    grpBase = vis.childBase;
                    grpBits = octant_order(vis.inner, sg.oct) | (vis.imask << 8) | FRESH;
                    grpMin = vis.restEntry;
                }
            }
            // ---- (C) triangle phase, when it pays --------------------------------------------------
            const int parked = __popcll(__ballot(pendC != 0u));
            const int stepping = __popcll(__ballot(active && pendC == 0u));
            if (parked != 0 && (parked >= PARK_MIN || stepping < STEP_MIN)) {
                if (COUNT) {
                    ++dbgTriRuns;
                    dbgTriLanes += parked;
                }
                if (pendC != 0u) { // one triangle per lane and phase: lanes with longer runs stay parked
                    test_run<COUNT, WATERTIGHT>(recs, nodes, parents, ray, mode, pendW, 1u, best, triTests, shear);
                    ++pendW;
                    --pendC;
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
            // per-wave figures: inner-loop passes, lanes that visited a node, triangle phases, lanes in
            // them, and their longest runs
            atomicAdd(&counters[3], dbgIters);
            atomicAdd(&counters[4], dbgVisitLanes);
            atomicAdd(&counters[5], dbgTriRuns);
            atomicAdd(&counters[6], dbgTriLanes);
        }
        for (int o = 32; o > 0; o >>= 1) {
            dbgEmpty += __shfl_xor(dbgEmpty, o, 64);
            dbgEmptyLimit += __shfl_xor(dbgEmptyLimit, o, 64);
            dbgUsedSlots += __shfl_xor(dbgUsedSlots, o, 64);
            dbgLeafOnly += __shfl_xor(dbgLeafOnly, o, 64);
            dbgCulled += __shfl_xor(dbgCulled, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters[7], dbgEmpty);
            atomicAdd(&counters[8], dbgEmptyLimit);
            atomicAdd(&counters[9], dbgUsedSlots);
            atomicAdd(&counters[10], dbgLeafOnly);
            atomicAdd(&counters[11], dbgCulled);
            // earliest start, earliest "all queues used up", latest exit, and the sum of the waves' lives
            constexpr unsigned long long TOP = 1ull << 62;
            const unsigned long long clkExit = wall_clock64();
            atomicMax(&counters[12], TOP - clkStart);
            if (clkEmpty != 0ull)
                atomicMax(&counters[13], TOP - clkEmpty);
            atomicMax(&counters[14], clkExit);
            atomicAdd(&counters[15], clkExit - clkStart);
        }
    }
}

struct TraceWs {
    size_t counter, spill, keysA, keysB, valsA, valsB, sortWs, sortWsBytes, status, total;
};
inline TraceWs trace_ws(uint32_t rayNum) {
    TraceWs w;
    Carver c;
    const size_t n = rayNum ? rayNum : 1;
    w.spill = c.take(sizeof(uint2) * MAX_GRID * THREADS * (STACK_DEPTH - LDS_DEPTH_BOUND));
    w.keysA = c.take(4 * n);
    w.keysB = c.take(4 * n);
    w.valsA = c.take(4 * n);
    w.valsB = c.take(4 * n);
    // the ray queue heads sit right before the sort's workspace: one clear covers both
    w.counter = c.take(4 * QUEUES * QUEUE_STRIDE);
    w.sortWsBytes = kk_sort_workspace_bytes(static_cast<uint32_t>(n));
    w.sortWs = c.take(w.sortWsBytes);
    w.status = c.take(4); // what the last call's kernel saw in the sort's error word
    w.total = c.off;
    return w;
}

} // namespace accel
} // namespace kk

kk::accel::View kk::accel::make_view(const void *d_accel, uint32_t faceNum, uint32_t compactedNum) {
    const Layout l = layout(faceNum, compactedNum);
    const char *ab = static_cast<const char *>(d_accel);
    View v;
    v.hdr = reinterpret_cast<const Header *>(ab + l.header);
    v.wide = reinterpret_cast<const Wide8 *>(ab + l.nodes);
    v.recs = reinterpret_cast<const TriRec *>(ab + l.tris);
    v.parents = reinterpret_cast<const uint32_t *>(ab + l.parents);
    return v;
}

extern "C" size_t kk_trace_accel_bytes(uint32_t face_num, uint32_t compacted_face_num) {
    return kk::accel::layout(face_num, compacted_face_num).total;
}

extern "C" int kk_trace_accel_layout(uint32_t face_num, uint32_t compacted_face_num,
                                     kk_trace_accel_layout_info *out) {
    KK_REQUIRE(out != nullptr, "null pointer");
    const kk::accel::Layout l = kk::accel::layout(face_num, compacted_face_num);
    out->header = l.header;
    out->nodes = l.nodes;
    out->tris = l.tris;
    out->parents = l.parents;
    out->total = l.total;
    return KK_OK;
}

extern "C" int kk_trace_accel_build_with_leaf_boxes(const kk_lbvh_device_data *d_lbvh,
                                                    const float *d_positions,
                                                    const uint32_t *d_face_pos_idx, uint32_t face_num,
                                                    uint32_t compacted_face_num,
                                                    const float *d_leaf_aabbs, void *d_accel,
                                                    size_t accel_bytes, void *stream_) {
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
    auto *hdr = reinterpret_cast<Header *>(b + l.header);
    auto *ctl = reinterpret_cast<BuildCtl *>(b + l.ctl);
    auto *rootOf = reinterpret_cast<uint32_t *>(b + l.rootOf);
    auto *wide = reinterpret_cast<Wide8 *>(b + l.nodes);
    init_build_kernel<<<1, 1, 0, stream>>>(d_lbvh, hdr, ctl, rootOf, nodeNum);
    KK_LAUNCH_CHECK();
    parents_kernel<<<div_up(nodeNum, THREADS), THREADS, 0, stream>>>(
        d_lbvh, reinterpret_cast<uint32_t *>(b + l.parents), nodeNum);
    KK_LAUNCH_CHECK();
    auto *recSrc = reinterpret_cast<uint2 *>(b + l.recSrc);
    const float *leafBoxes = d_leaf_aabbs;
    if (!leafBoxes) { // made here, where the triangle records go at the end
        float *own = reinterpret_cast<float *>(b + l.tris);
        leaf_boxes_kernel<<<div_up(compacted_face_num, THREADS), THREADS, 0, stream>>>(
            d_lbvh, d_positions, d_face_pos_idx, own);
        KK_LAUNCH_CHECK();
        leafBoxes = own;
    }
    const CollapseArgs args{d_lbvh, d_positions, d_face_pos_idx, leafBoxes, wide, recSrc, rootOf,
                            reinterpret_cast<uint32_t *>(b + l.leafParent), ctl,
                            reinterpret_cast<uint32_t *>(b + l.runList)};
    uint32_t grid = div_up(l.wideMax, BT);
    if (grid > 2048)
        grid = 2048;
    // a launch per level up to here, one workgroup for the rest; a small tree is one workgroup's work
    // from the root on (two launches per level would cost more than its levels take)
    const int TAIL_FROM = nodeNum <= 4096u ? 0 : 12;
    for (int level = 0; level < TAIL_FROM; ++level) {
        // level l has at most 8^l nodes
        const uint32_t g = level < 7 ? (grid < div_up(1u << (3 * level), BT) ? grid : div_up(1u << (3 * level), BT)) : grid;
        collapse_level_kernel<<<g, BT, 0, stream>>>(args, level);
        KK_LAUNCH_CHECK();
        close_level_kernel<<<1, 1, 0, stream>>>(ctl, hdr, level);
        KK_LAUNCH_CHECK();
    }
    collapse_tail_kernel<<<1, BT, 0, stream>>>(args, hdr, TAIL_FROM);
    KK_LAUNCH_CHECK();
    if (face_num - compacted_face_num >= SPLIT_MIN - 1) { // else no leaf is long enough for a run node
        run_nodes_kernel<<<RUN_GRID, BT, 0, stream>>>(args);
        KK_LAUNCH_CHECK();
    }
    fill_records_kernel<<<div_up(face_num, THREADS), THREADS, 0, stream>>>(
        d_lbvh, d_positions, d_face_pos_idx, recSrc, reinterpret_cast<TriRec *>(b + l.tris));
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" int kk_trace_accel_build(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                                    const uint32_t *d_face_pos_idx, uint32_t face_num,
                                    uint32_t compacted_face_num, void *d_accel, size_t accel_bytes,
                                    void *stream_) {
    return kk_trace_accel_build_with_leaf_boxes(d_lbvh, d_positions, d_face_pos_idx, face_num,
                                                compacted_face_num, nullptr, d_accel, accel_bytes, stream_);
}

extern "C" int kk_trace_accel_check(const void *d_accel, uint32_t face_num, uint32_t compacted_face_num,
                                    void *stream_) {
    using namespace kk;
    using namespace kk::accel;
    KK_REQUIRE(d_accel != nullptr, "null pointer");
    KK_REQUIRE(face_num > 0 && compacted_face_num > 0 && compacted_face_num <= face_num,
               "tree sizes out of range");
    const Layout l = layout(face_num, compacted_face_num);
    Header h;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KK_HIP_TRY(hipMemcpyAsync(&h, static_cast<const char *>(d_accel) + l.header, sizeof(Header),
                              hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    if (h.tooDeep) {
        set_error("the tree is deeper than %d wide levels: the trace-side layout cannot hold it", MAX_LEVELS);
        return KK_ERR_DEVICE;
    }
    if (h.wideNum == 0 || h.faceNum != face_num) {
        set_error("the accel buffer does not hold a tree of %u faces", face_num);
        return KK_ERR_DEVICE;
    }
    return KK_OK;
}

extern "C" size_t kk_closest_hit_accel_workspace_bytes(uint32_t ray_num) {
    return kk::accel::trace_ws(ray_num).total;
}

extern "C" int kk_closest_hit_accel_status(const void *d_workspace, uint32_t ray_num, void *stream_) {
    using namespace kk;
    KK_REQUIRE(d_workspace != nullptr, "null pointer");
    if (ray_num == 0)
        return KK_OK;
    uint32_t word = 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KK_HIP_TRY(hipMemcpyAsync(&word, static_cast<const char *>(d_workspace) + accel::trace_ws(ray_num).status,
                              sizeof(word), hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    if (word != 0) {
        set_error("kk_closest_hit_accel: the ray sort's look-back gave up (error word %u); the rays were traced "
                  "in input order instead (same hits, slower)", word);
        return KK_ERR_DEVICE;
    }
    return KK_OK;
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
                                               KK_TRACE_WATERTIGHT | KK_TRACE_ENTRY_BOUND)) == 0,
               "unknown flag");
    if (ray_num == 0)
        return KK_OK;
    const TraceWs w = trace_ws(ray_num);
    KK_REQUIRE(workspace_bytes >= w.total, "workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const Layout l = layout(face_num, compacted_face_num);
    const char *ab = static_cast<const char *>(d_accel);
    const auto *hdr = reinterpret_cast<const Header *>(ab + l.header);
    const auto *wide = reinterpret_cast<const Wide8 *>(ab + l.nodes);
    const auto *recs = reinterpret_cast<const TriRec *>(ab + l.tris);
    char *b = static_cast<char *>(d_workspace);
    uint32_t *rayCounter = reinterpret_cast<uint32_t *>(b + w.counter);
    const bool sorts = !(flags & (KK_TRACE_REUSE_ORDER | KK_TRACE_UNSORTED));
    if (!sorts)
        KK_HIP_TRY(hipMemsetAsync(rayCounter, 0, 4 * QUEUES * QUEUE_STRIDE, stream));

    const uint32_t *order = nullptr;
    if (flags & KK_TRACE_REUSE_ORDER) {
        order = reinterpret_cast<uint32_t *>(b + ((KEY_PASSES & 1) ? w.valsB : w.valsA));
    } else if (!(flags & KK_TRACE_UNSORTED)) {
        auto *keysA = reinterpret_cast<uint32_t *>(b + w.keysA);
        auto *valsA = reinterpret_cast<uint32_t *>(b + w.valsA);
        auto *valsB = reinterpret_cast<uint32_t *>(b + w.valsB);
        uint32_t *ghist = nullptr;
        int rc = rsort::sort_prepare(b + w.sortWs, w.sortWsBytes, ray_num, stream, &ghist,
                                     w.sortWs - w.counter); // ... and the queue heads before it
        if (rc != KK_OK)
            return rc;
        uint32_t kgrid = div_up(ray_num, THREADS);
        if (kgrid > KEYS_MAX_BLOCKS)
            kgrid = KEYS_MAX_BLOCKS;
        ray_keys_kernel<<<kgrid, THREADS, 0, stream>>>(hdr, d_rays, ray_num, keysA, ghist);
        KK_LAUNCH_CHECK();
        rc = rsort::sort_pairs(keysA, reinterpret_cast<uint32_t *>(b + w.keysB), valsA, valsB, true, ray_num,
                               b + w.sortWs, w.sortWsBytes, stream, nullptr, KEY_PASSES, true);
        if (rc != KK_OK)
            return rc;
        order = (KEY_PASSES & 1) ? valsB : valsA;
    }
    // (with KK_TRACE_REUSE_ORDER the word is still the one of the sort that made the order)
    const uint32_t *sortError =
        order ? reinterpret_cast<const uint32_t *>(b + w.sortWs + kk_sort_workspace_error_offset(ray_num)) : nullptr;
    uint32_t *status = reinterpret_cast<uint32_t *>(b + w.status);
    uint32_t grid = div_up(ray_num, THREADS);
    const uint32_t resident = GROUPS_PER_CU * static_cast<uint32_t>(device_cu_count());
    if (grid > resident)
        grid = resident;
    if (grid > MAX_GRID)
        grid = MAX_GRID;
    uint2 *spill = reinterpret_cast<uint2 *>(b + w.spill);
    const auto *parents = reinterpret_cast<const uint32_t *>(ab + l.parents);
    const bool wt = (flags & KK_TRACE_WATERTIGHT) != 0;
#define KK_LAUNCH_TRACE(COUNT, WT, BOUND)                                                          \
    closest_hit_accel_kernel<COUNT, WT, BOUND><<<grid, THREADS, 0, stream>>>(                       \
        hdr, wide, recs, d_rays, order, ray_num, d_t, d_bary, d_prim, rayCounter, parents, spill,  \
        d_counters, sortError, status)
    const bool bound = (flags & KK_TRACE_ENTRY_BOUND) != 0;
    KK_REQUIRE(!(bound && wt), "KK_TRACE_ENTRY_BOUND and KK_TRACE_WATERTIGHT do not combine");
    if (d_counters) {
        if (wt)
            KK_LAUNCH_TRACE(true, true, false);
        else if (bound)
            KK_LAUNCH_TRACE(true, false, true);
        else
            KK_LAUNCH_TRACE(true, false, false);
    } else {
        if (wt)
            KK_LAUNCH_TRACE(false, true, false);
        else if (bound)
            KK_LAUNCH_TRACE(false, false, true);
        else
            KK_LAUNCH_TRACE(false, false, false);
    }
#undef KK_LAUNCH_TRACE
    KK_LAUNCH_CHECK();
    return KK_OK;
}
