// LBVH build for gfx950: kouek::RayTracer::LBVH::BuildFrom (src/ray_tracer/lbvh.cu:100-323) as a chain
// of kernels on one stream with no host round trip. The number of leaves M is produced on the device
// and read there by the later kernels, which are launched for the upper bound N and exit early.
//
//   bounds_partial + build_setup : root box over ALL positions, 1/extent        (lbvh.cu:112-134)
//   morton                      : 30-bit code of each triangle's box centre    (lbvh.cu:136-170)
//   radix sort (radix_sort.hip) : stable sort of (code, face)                  (lbvh.cu:110,172)
//   count_heads/scan/compact    : run starts of equal codes -> leaves          (algorithm.h:11-41)
//   leaf_aabb                   : one box per run                              (lbvh.cu:186-219)
//   karras_links                : children of each internal node (+ parents)   (lbvh.cu:29-98,221-260)
//   refit_local + refit_top     : bottom-up box union with one arrival counter per internal node
//                                 (in LDS inside a workgroup's leaf window, global above it).
//                                 The reference walks each node's whole subtree (lbvh.cu:265-297);
//                                 min/max unions are exact, so both give the same bits.
#include "kk_host.h"
#include "kk_math.h"

namespace kk {
namespace rsort {
int sort_pairs(uint32_t *keysA, uint32_t *keysB, uint32_t *valsA, uint32_t *valsB, bool iota,
               uint32_t n, void *ws, size_t wsBytes, hipStream_t stream, uint32_t **errorFlagOut,
               int passes);
}
extern "C" size_t kk_sort_workspace_bytes(uint32_t n);

namespace lbvh {

constexpr int THREADS = 256;
constexpr uint32_t LEAF = 0x80000000u;
constexpr uint32_t LEAF_MASK = 0x7fffffffu;
constexpr int BOUNDS_MAX_BLOCKS = 2048;
constexpr int HEADS_PER_THREAD = 4;
constexpr int HEADS_PER_BLOCK = THREADS * HEADS_PER_THREAD;

// Device-resident parameters produced by build_setup and read by morton.
struct BuildParams {
    float minPos[3];
    float invDlt[3];
};

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = gmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = gmax(v, __shfl_xor(v, o, 64));
    return v;
}

// partials[block][6] = (min xyz, max xyz) over a grid-stride share of the vertices.
__global__ __launch_bounds__(THREADS) void bounds_partial_kernel(const float *__restrict__ pos,
                                                                 uint32_t vertNum,
                                                                 float *__restrict__ partials) {
    __shared__ float s_red[THREADS / 64][6];
    const int tid = threadIdx.x;
    float mn[3] = {FloatMax, FloatMax, FloatMax};
    float mx[3] = {FloatMin, FloatMin, FloatMin};
    for (uint32_t v = blockIdx.x * THREADS + tid; v < vertNum; v += gridDim.x * THREADS) {
        const float x = pos[3ull * v], y = pos[3ull * v + 1], z = pos[3ull * v + 2];
        mn[0] = gmin(mn[0], x);
        mn[1] = gmin(mn[1], y);
        mn[2] = gmin(mn[2], z);
        mx[0] = gmax(mx[0], x);
        mx[1] = gmax(mx[1], y);
        mx[2] = gmax(mx[2], z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mn[k] = wave_min(mn[k]);
        mx[k] = wave_max(mx[k]);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_red[tid >> 6][k] = mn[k];
            s_red[tid >> 6][3 + k] = mx[k];
        }
    }
    __syncthreads();
    if (tid < 6) {
        float r = s_red[0][tid];
        for (int w = 1; w < THREADS / 64; ++w)
            r = tid < 3 ? gmin(r, s_red[w][tid]) : gmax(r, s_red[w][tid]);
        partials[blockIdx.x * 6 + tid] = r;
    }
}

// One block: reduce the partials, write the root box, 1/extent (lbvh.cu:130-134) and the header.
__global__ __launch_bounds__(THREADS) void build_setup_kernel(
    const float *__restrict__ partials, uint32_t partialNum, uint32_t faceNum, uint32_t *faces,
    uint32_t *compacted, kk_lbvh_node *nodes, BuildParams *params, kk_lbvh_device_data *out) {
    __shared__ float s_red[THREADS / 64][6];
    const int tid = threadIdx.x;
    float mn[3] = {FloatMax, FloatMax, FloatMax};
    float mx[3] = {FloatMin, FloatMin, FloatMin};
    for (uint32_t b = tid; b < partialNum; b += THREADS) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = gmin(mn[k], partials[b * 6 + k]);
            mx[k] = gmax(mx[k], partials[b * 6 + 3 + k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mn[k] = wave_min(mn[k]);
        mx[k] = wave_max(mx[k]);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_red[tid >> 6][k] = mn[k];
            s_red[tid >> 6][3 + k] = mx[k];
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 3; ++k) {
            float lo = s_red[0][k], hi = s_red[0][3 + k];
            for (int w = 1; w < THREADS / 64; ++w) {
                lo = gmin(lo, s_red[w][k]);
                hi = gmax(hi, s_red[w][3 + k]);
            }
            float ext = hi - lo;
            if (fabsf(ext) < FloatEps)
                ext = 1.f;
            params->minPos[k] = lo;
            params->invDlt[k] = 1.f / ext;
            out->root_aabb_min[k] = lo;
            out->root_aabb_max[k] = hi;
        }
        out->face_num = faceNum;
        out->compacted_face_num = 0;
        out->faces = faces;
        out->compacted_faces = compacted;
        out->internal_nodes = nodes;
        out->_pad = 0;
    }
}

// lbvh.cu:158-164
__device__ __forceinline__ uint32_t expand10BitsTo32(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ f3 load3(const float *__restrict__ p, uint32_t i) {
    return {p[3ull * i], p[3ull * i + 1], p[3ull * i + 2]};
}

// lbvh.cu:136-170, one thread per face.
__global__ __launch_bounds__(THREADS) void morton_kernel(const float *__restrict__ pos,
                                                         const uint32_t *__restrict__ tri,
                                                         uint32_t faceNum,
                                                         const BuildParams *__restrict__ params,
                                                         uint32_t *__restrict__ mortons) {
    const uint32_t fi = blockIdx.x * THREADS + threadIdx.x;
    if (fi >= faceNum)
        return;
    const f3 minPos = {params->minPos[0], params->minPos[1], params->minPos[2]};
    const f3 invDlt = {params->invDlt[0], params->invDlt[1], params->invDlt[2]};
    AABB box = AABB::Initial();
    box.Expand(load3(pos, tri[3ull * fi]));
    box.Expand(load3(pos, tri[3ull * fi + 1]));
    box.Expand(load3(pos, tri[3ull * fi + 2]));
    f3 c = .5f * (box.minPos + box.maxPos);
    c = (c - minPos) * invDlt;
    c.x = gmin(gmax(c.x * 1024.f, 0.f), 1023.f);
    c.y = gmin(gmax(c.y * 1024.f, 0.f), 1023.f);
    c.z = gmin(gmax(c.z * 1024.f, 0.f), 1023.f);
    const uint32_t cx = expand10BitsTo32(static_cast<uint32_t>(c.x));
    const uint32_t cy = expand10BitsTo32(static_cast<uint32_t>(c.y));
    const uint32_t cz = expand10BitsTo32(static_cast<uint32_t>(c.z));
    mortons[fi] = cx * 4 + cy * 2 + cz;
}

// ---- run heads: flag[i] = (i == 0) || key[i-1] != key[i]  (include/cuda/algorithm.h:11-21) ----------

__device__ __forceinline__ uint32_t load_heads(const uint32_t *__restrict__ keys, uint32_t n,
                                               uint32_t first, uint32_t k[HEADS_PER_THREAD]) {
    // returns a bit mask of the heads among keys[first .. first+3]
    uint32_t prev = first == 0 ? 0u : (first < n ? keys[first - 1] : 0u);
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < HEADS_PER_THREAD; ++j) {
        const uint32_t i = first + j;
        if (i < n) {
            k[j] = keys[i];
            if (i == 0 || k[j] != prev)
                mask |= 1u << j;
            prev = k[j];
        } else
            k[j] = 0;
    }
    return mask;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o)
            v += t;
    }
    return v;
}

__global__ __launch_bounds__(THREADS) void count_heads_kernel(const uint32_t *__restrict__ keys,
                                                              uint32_t n,
                                                              uint32_t *__restrict__ blockCounts) {
    __shared__ uint32_t s_w[THREADS / 64];
    const int tid = threadIdx.x;
    uint32_t k[HEADS_PER_THREAD];
    const uint32_t first = (blockIdx.x * THREADS + tid) * HEADS_PER_THREAD;
    uint32_t cnt = __popc(load_heads(keys, n, first, k));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0)
        s_w[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0)
        blockCounts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// One block: exclusive scan of the block counts in place; the total is M (algorithm.h:27-28).
__global__ __launch_bounds__(1024) void scan_block_counts_kernel(uint32_t *blockCounts,
                                                                 uint32_t blockNum,
                                                                 kk_lbvh_device_data *out) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
        s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < blockNum; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < blockNum ? blockCounts[i] : 0u;
        uint32_t incl = wave_inclusive_scan_u32(v, lane);
        if (lane == 63)
            s_w[wave] = incl;
        __syncthreads();
        uint32_t prefix = s_carry;
        for (int w = 0; w < wave; ++w)
            prefix += s_w[w];
        if (i < blockNum)
            blockCounts[i] = prefix + incl - v;
        __syncthreads();
        if (tid == 1023)
            s_carry = prefix + incl;
        __syncthreads();
    }
    if (tid == 0)
        out->compacted_face_num = s_carry;
}

// compacted[c] = index of the c-th head (algorithm.h:31-39); uniq[c] = its code (lbvh.cu:202).
__global__ __launch_bounds__(THREADS) void compact_heads_kernel(
    const uint32_t *__restrict__ keys, uint32_t n, const uint32_t *__restrict__ blockOffsets,
    uint32_t *__restrict__ compacted, uint32_t *__restrict__ uniq) {
    __shared__ uint32_t s_w[THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t k[HEADS_PER_THREAD];
    const uint32_t first = (blockIdx.x * THREADS + tid) * HEADS_PER_THREAD;
    const uint32_t mask = load_heads(keys, n, first, k);
    const uint32_t cnt = __popc(mask);
    const uint32_t incl = wave_inclusive_scan_u32(cnt, lane);
    if (lane == 63)
        s_w[wave] = incl;
    __syncthreads();
    uint32_t c = blockOffsets[blockIdx.x] + incl - cnt;
    for (int w = 0; w < wave; ++w)
        c += s_w[w];
#pragma unroll
    for (int j = 0; j < HEADS_PER_THREAD; ++j)
        if (mask & (1u << j)) {
            compacted[c] = first + j;
            uniq[c] = k[j];
            ++c;
        }
}

// lbvh.cu:186-219, one thread per leaf (run of equal codes).
__global__ __launch_bounds__(THREADS) void leaf_aabb_kernel(
    const float *__restrict__ pos, const uint32_t *__restrict__ tri,
    const uint32_t *__restrict__ faces, const uint32_t *__restrict__ compacted,
    const kk_lbvh_device_data *__restrict__ hdr, float *__restrict__ leafAABBs) {
    const uint32_t M = hdr->compacted_face_num;
    const uint32_t N = hdr->face_num;
    const uint32_t c = blockIdx.x * THREADS + threadIdx.x;
    if (c >= M)
        return;
    const uint32_t beg = compacted[c];
    const uint32_t end = (c == M - 1) ? N : compacted[c + 1];
    AABB box = AABB::Initial();
    for (uint32_t k = beg; k < end; ++k) {
        const uint32_t face = faces[k];
        box.Expand(load3(pos, tri[3ull * face]));
        box.Expand(load3(pos, tri[3ull * face + 1]));
        box.Expand(load3(pos, tri[3ull * face + 2]));
    }
    float *o = leafAABBs + 6ull * c;
    o[0] = box.minPos.x;
    o[1] = box.minPos.y;
    o[2] = box.minPos.z;
    o[3] = box.maxPos.x;
    o[4] = box.maxPos.y;
    o[5] = box.maxPos.z;
}

// lbvh.cu:221-232: length of the common leading-zero run of code[i] ^ code[j], -1 out of range.
__device__ __forceinline__ int delta(const uint32_t *__restrict__ m, uint32_t mi, int64_t j,
                                     int64_t inodeNum) {
    if (j < 0 || j > inodeNum)
        return -1;
    const uint32_t x = mi ^ m[j];
    return x == 0 ? 32 : __clz(static_cast<int>(x));
}

// lbvh.cu:29-98, one thread per internal node; also records the parent of each child.
__global__ __launch_bounds__(THREADS) void karras_links_kernel(
    const uint32_t *__restrict__ mortons, const kk_lbvh_device_data *__restrict__ hdr,
    kk_lbvh_node *__restrict__ nodes, uint32_t *__restrict__ nodeParents,
    uint32_t *__restrict__ leafParents, uint32_t *__restrict__ rangeEnds) {
    const uint32_t M = hdr->compacted_face_num;
    const int64_t inodeNum = static_cast<int64_t>(M) - 1;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * THREADS + threadIdx.x;
    if (i >= inodeNum)
        return;
    const uint32_t mi = mortons[i];

    const int commLft = delta(mortons, mi, i - 1, inodeNum);
    const int commRht = delta(mortons, mi, i + 1, inodeNum);
    const int64_t d = (commLft <= commRht) ? 1 : -1;

    int cmp = delta(mortons, mi, i - d, inodeNum);
    int64_t offsMax = 2;
    while (delta(mortons, mi, i + d * offsMax, inodeNum) > cmp)
        offsMax <<= 1;

    int64_t offs = 0;
    while (true) {
        offsMax >>= 1;
        if (offsMax < 1)
            break;
        if (delta(mortons, mi, i + d * (offsMax + offs), inodeNum) > cmp)
            offs += offsMax;
    }
    const int64_t iMax = i + d * offs;
    cmp = delta(mortons, mi, iMax, inodeNum);

    offsMax = offs;
    offs = 0;
    while (true) {
        offsMax = (offsMax + 1) >> 1;
        if (offsMax < 1)
            break;
        if (delta(mortons, mi, i + d * (offsMax + offs), inodeNum) > cmp)
            offs += offsMax;
        if (offsMax == 1)
            break;
    }
    const int64_t split = i + d * offs + (d < 0 ? -1 : 0);

    uint32_t lft = static_cast<uint32_t>(split);
    uint32_t rht = static_cast<uint32_t>(split + 1);
    const int64_t lo = i < iMax ? i : iMax, hi = i < iMax ? iMax : i;
    if (lo == split) {
        leafParents[lft] = static_cast<uint32_t>(i);
        lft |= LEAF;
    } else
        nodeParents[lft] = static_cast<uint32_t>(i);
    if (hi == split + 1) {
        leafParents[rht] = static_cast<uint32_t>(i);
        rht |= LEAF;
    } else
        nodeParents[rht] = static_cast<uint32_t>(i);
    nodes[i].lft = lft;
    nodes[i].rht = rht;
    rangeEnds[i] = static_cast<uint32_t>(iMax); // the node covers leaves [min(i, iMax), max(i, iMax)]
    if (i == 0)
        nodeParents[0] = 0xFFFFFFFFu;
}

__device__ __forceinline__ AABB load_box(const float *p) {
    AABB b;
    b.minPos = {p[0], p[1], p[2]};
    b.maxPos = {p[3], p[4], p[5]};
    return b;
}

// Bottom-up box refit in two kernels. Every internal node has an arrival counter: the first child to
// arrive stops, the second one unions both children's boxes and climbs on.
//   refit_local : one workgroup owns REFIT_LEAVES consecutive leaves (consecutive in Morton order, so
//                 almost every subtree above them lies inside the group). While a parent's leaf range
//                 lies inside the group's leaves, its counter and the boxes of the group's internal
//                 nodes live in LDS (a node's index is an end of its range, so it falls inside the
//                 group's window too); only finished boxes go to memory, with plain stores and no
//                 fence. A thread whose parent straddles a group boundary appends its child to a work
//                 list and stops.
//   refit_top   : the listed children (about 1 % of the nodes, the top of the tree) climb on with
//                 global counters. A counter is an acquire-release agent-scope atomic: the arriving
//                 thread wrote its child's box itself (or an earlier kernel did), the release publishes
//                 it, and the second arrival's acquire makes the sibling's box visible.
constexpr int REFIT_LPT = 4;
constexpr int REFIT_LEAVES = THREADS * REFIT_LPT;
constexpr int REFIT_TOP_BLOCKS = 1024;

__device__ __forceinline__ void store_box(float *p, const AABB &b) {
    p[0] = b.minPos.x;
    p[1] = b.minPos.y;
    p[2] = b.minPos.z;
    p[3] = b.maxPos.x;
    p[4] = b.maxPos.y;
    p[5] = b.maxPos.z;
}

__global__ __launch_bounds__(THREADS) void refit_local_kernel(
    const float *__restrict__ leafAABBs, const kk_lbvh_device_data *__restrict__ hdr,
    kk_lbvh_node *nodes, const uint32_t *__restrict__ nodeParents,
    const uint32_t *__restrict__ leafParents, const uint32_t *__restrict__ rangeEnds,
    uint32_t *__restrict__ topList, uint32_t *topCount) {
    __shared__ uint32_t s_cnt[REFIT_LEAVES];
    __shared__ float s_box[REFIT_LEAVES][6];
    const uint32_t M = hdr->compacted_face_num;
    const uint32_t b0 = blockIdx.x * REFIT_LEAVES;
    if (b0 >= M)
        return;
    const uint32_t b1 = (b0 + REFIT_LEAVES < M) ? b0 + REFIT_LEAVES : M; // leaves [b0, b1)
    for (int i = threadIdx.x; i < REFIT_LEAVES; i += THREADS)
        s_cnt[i] = 0;
    __syncthreads();
    if (M == 1) {
        // No internal node exists; give traversal a root whose two children are the only leaf.
        if (threadIdx.x == 0) {
            nodes[0].lft = LEAF;
            nodes[0].rht = LEAF;
            for (int k = 0; k < 3; ++k) {
                nodes[0].aabb_min[k] = leafAABBs[k];
                nodes[0].aabb_max[k] = leafAABBs[3 + k];
            }
        }
        return;
    }
    for (int j = 0; j < REFIT_LPT; ++j) {
        const uint32_t c = b0 + j * THREADS + threadIdx.x;
        if (c >= b1)
            continue;
        AABB box = load_box(leafAABBs + 6ull * c);
        uint32_t child = c | LEAF;
        uint32_t node = leafParents[c];
        for (int guard = 0; guard < 64; ++guard) {
            const uint32_t other = rangeEnds[node];
            const uint32_t lo = node < other ? node : other, hi = node < other ? other : node;
            if (!(lo >= b0 && hi < b1)) {
                topList[atomicAdd(topCount, 1u)] = child;
                break;
            }
            const uint32_t lft = nodes[node].lft, rht = nodes[node].rht;
            const uint32_t up = nodeParents[node];
            const uint32_t sibling = (lft == child) ? rht : lft;
            // LDS only: the box written below and this counter are both LDS accesses of one wave and
            // are performed in order, so a workgroup-scope fence over LDS is all the ordering needed.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            const uint32_t old = __hip_atomic_fetch_add(&s_cnt[node - b0], 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (old == 0)
                break;
            box.Expand((sibling & LEAF) ? load_box(leafAABBs + 6ull * (sibling & LEAF_MASK))
                                        : load_box(s_box[sibling - b0]));
            store_box(s_box[node - b0], box);
            store_box(nodes[node].aabb_min, box);
            if (node == 0)
                break;
            child = node;
            node = up;
        }
    }
}

__global__ __launch_bounds__(THREADS) void refit_top_kernel(
    const float *__restrict__ leafAABBs, kk_lbvh_node *nodes,
    const uint32_t *__restrict__ nodeParents, const uint32_t *__restrict__ leafParents,
    const uint32_t *__restrict__ topList, const uint32_t *__restrict__ topCount,
    uint32_t *arrivals) {
    const uint32_t count = *topCount;
    for (uint32_t e = blockIdx.x * THREADS + threadIdx.x; e < count; e += gridDim.x * THREADS) {
        uint32_t child = topList[e];
        uint32_t node = (child & LEAF) ? leafParents[child & LEAF_MASK] : nodeParents[child];
        for (int guard = 0; guard < 64; ++guard) {
            const uint32_t old = __hip_atomic_fetch_add(&arrivals[node], 1u, __ATOMIC_ACQ_REL,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0)
                break;
            const uint32_t lft = nodes[node].lft, rht = nodes[node].rht;
            AABB box = (lft & LEAF) ? load_box(leafAABBs + 6ull * (lft & LEAF_MASK))
                                    : load_box(nodes[lft].aabb_min);
            box.Expand((rht & LEAF) ? load_box(leafAABBs + 6ull * (rht & LEAF_MASK))
                                    : load_box(nodes[rht].aabb_min));
            store_box(nodes[node].aabb_min, box);
            if (node == 0)
                break;
            child = node;
            node = nodeParents[node];
        }
    }
}

struct Ws {
    kk_lbvh_ws_layout pub;
    size_t params, partials, keys1, vals1, blockCounts, arrivals, rangeEnds, topList, sortWs;
    size_t sortWsBytes;
    uint32_t headBlocks;
};

inline Ws layout(uint32_t N, uint32_t flags) {
    Ws w;
    Carver c;
    const size_t n = N;
    w.headBlocks = div_up(N, HEADS_PER_BLOCK);
    w.params = c.take(sizeof(BuildParams));
    w.partials = c.take(sizeof(float) * 6 * BOUNDS_MAX_BLOCKS);
    w.pub.sorted_mortons = c.take(4 * n); // keys0: Morton codes, sorted in place by the 4 passes
    w.keys1 = c.take(4 * n);
    w.vals1 = c.take(4 * n);
    w.pub.unsorted_mortons = (flags & KK_LBVH_KEEP_STAGES) ? c.take(4 * n) : 0;
    w.pub.unique_mortons = c.take(4 * n);
    w.pub.leaf_aabbs = c.take(24 * n);
    w.pub.node_parents = c.take(4 * n);
    w.pub.leaf_parents = c.take(4 * n);
    w.blockCounts = c.take(4 * static_cast<size_t>(w.headBlocks));
    w.arrivals = c.take(4 * n + 4);
    w.rangeEnds = c.take(4 * n);
    w.topList = c.take(4 * n);
    w.sortWsBytes = kk_sort_workspace_bytes(N);
    w.sortWs = c.take(w.sortWsBytes);
    // the sort's error word sits right after its histogram and counters (radix_sort.hip: sort_ws)
    w.pub.error_flag = w.sortWs + align_up(4 * 4 * 256) + align_up(4 * 4);
    w.pub.total_bytes = c.off;
    return w;
}

} // namespace lbvh
} // namespace kk

extern "C" int kk_lbvh_workspace_layout(uint32_t face_num, uint32_t vert_num, uint32_t flags,
                                        kk_lbvh_ws_layout *out) {
    (void)vert_num;
    KK_REQUIRE(out != nullptr, "null pointer");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES, "face_num out of range");
    *out = kk::lbvh::layout(face_num, flags).pub;
    return KK_OK;
}

extern "C" int kk_lbvh_build(const float *d_positions, uint32_t vert_num,
                             const uint32_t *d_face_pos_idx, uint32_t face_num, uint32_t *d_faces,
                             uint32_t *d_compacted, kk_lbvh_node *d_nodes,
                             kk_lbvh_device_data *d_out, void *d_workspace, size_t workspace_bytes,
                             uint32_t flags, void *stream_) {
    using namespace kk;
    using namespace kk::lbvh;
    KK_REQUIRE(d_positions && d_face_pos_idx && d_faces && d_compacted && d_nodes && d_out &&
                   d_workspace,
               "null pointer");
    KK_REQUIRE(vert_num > 0, "no vertices");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES, "face_num out of range");
    const Ws w = layout(face_num, flags);
    if (workspace_bytes < w.pub.total_bytes) {
        set_error("LBVH workspace too small: %zu < %zu", workspace_bytes, w.pub.total_bytes);
        return KK_ERR_WORKSPACE;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *b = static_cast<char *>(d_workspace);
    auto *params = reinterpret_cast<BuildParams *>(b + w.params);
    auto *partials = reinterpret_cast<float *>(b + w.partials);
    auto *keys0 = reinterpret_cast<uint32_t *>(b + w.pub.sorted_mortons);
    auto *keys1 = reinterpret_cast<uint32_t *>(b + w.keys1);
    auto *vals1 = reinterpret_cast<uint32_t *>(b + w.vals1);
    auto *uniq = reinterpret_cast<uint32_t *>(b + w.pub.unique_mortons);
    auto *leafAABBs = reinterpret_cast<float *>(b + w.pub.leaf_aabbs);
    auto *nodeParents = reinterpret_cast<uint32_t *>(b + w.pub.node_parents);
    auto *leafParents = reinterpret_cast<uint32_t *>(b + w.pub.leaf_parents);
    auto *blockCounts = reinterpret_cast<uint32_t *>(b + w.blockCounts);
    auto *arrivals = reinterpret_cast<uint32_t *>(b + w.arrivals);
    auto *rangeEnds = reinterpret_cast<uint32_t *>(b + w.rangeEnds);
    auto *topList = reinterpret_cast<uint32_t *>(b + w.topList);
    const uint32_t N = face_num;

    uint32_t bgrid = div_up(vert_num, THREADS);
    if (bgrid > BOUNDS_MAX_BLOCKS)
        bgrid = BOUNDS_MAX_BLOCKS;
    bounds_partial_kernel<<<bgrid, THREADS, 0, stream>>>(d_positions, vert_num, partials);
    KK_LAUNCH_CHECK();
    build_setup_kernel<<<1, THREADS, 0, stream>>>(partials, bgrid, N, d_faces, d_compacted, d_nodes,
                                                  params, d_out);
    KK_LAUNCH_CHECK();
    morton_kernel<<<div_up(N, THREADS), THREADS, 0, stream>>>(d_positions, d_face_pos_idx, N, params,
                                                              keys0);
    KK_LAUNCH_CHECK();
    if (flags & KK_LBVH_KEEP_STAGES)
        KK_HIP_TRY(hipMemcpyAsync(b + w.pub.unsorted_mortons, keys0, 4ull * N,
                                  hipMemcpyDeviceToDevice, stream));

    // payload = face index; four passes leave keys in keys0 and faces in d_faces
    int rc = rsort::sort_pairs(keys0, keys1, d_faces, vals1, true, N, b + w.sortWs, w.sortWsBytes,
                               stream, nullptr, 4);
    if (rc != KK_OK)
        return rc;

    count_heads_kernel<<<w.headBlocks, THREADS, 0, stream>>>(keys0, N, blockCounts);
    KK_LAUNCH_CHECK();
    scan_block_counts_kernel<<<1, 1024, 0, stream>>>(blockCounts, w.headBlocks, d_out);
    KK_LAUNCH_CHECK();
    compact_heads_kernel<<<w.headBlocks, THREADS, 0, stream>>>(keys0, N, blockCounts, d_compacted,
                                                               uniq);
    KK_LAUNCH_CHECK();

    const uint32_t ngrid = div_up(N, THREADS);
    leaf_aabb_kernel<<<ngrid, THREADS, 0, stream>>>(d_positions, d_face_pos_idx, d_faces,
                                                    d_compacted, d_out, leafAABBs);
    KK_LAUNCH_CHECK();
    karras_links_kernel<<<ngrid, THREADS, 0, stream>>>(uniq, d_out, d_nodes, nodeParents,
                                                       leafParents, rangeEnds);
    KK_LAUNCH_CHECK();
    // arrivals[0 .. N) are the top-phase counters; the word after them counts the work list
    KK_HIP_TRY(hipMemsetAsync(arrivals, 0, 4ull * N + 4, stream));
    uint32_t *topCount = arrivals + N;
    refit_local_kernel<<<div_up(N, REFIT_LEAVES), THREADS, 0, stream>>>(
        leafAABBs, d_out, d_nodes, nodeParents, leafParents, rangeEnds, topList, topCount);
    KK_LAUNCH_CHECK();
    refit_top_kernel<<<REFIT_TOP_BLOCKS, THREADS, 0, stream>>>(leafAABBs, d_nodes, nodeParents,
                                                               leafParents, topList, topCount,
                                                               arrivals);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
