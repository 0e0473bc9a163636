// LBVH build for gfx950: kouek::RayTracer::LBVH::BuildFrom (src/ray_tracer/lbvh.cu:100-323) as a chain
// of kernels on one stream with no host round trip. The number of leaves M is produced on the device
// and read there by the later kernels, which are launched for the upper bound N and exit early.
//
//   bounds_partial + build_setup : root box over ALL positions, 1/extent        (lbvh.cu:112-134)
//   morton                      : 30-bit code of each triangle's box centre    (lbvh.cu:136-170)
//   radix sort (radix_sort.hip) : stable sort of (code, face)                  (lbvh.cu:110,172)
//   heads                       : run starts of equal codes -> leaves          (algorithm.h:11-41)
//   karras_links                : children of each internal node (+ parents)   (lbvh.cu:29-98,221-260)
//                                 and, in the same threads, one box per run      (lbvh.cu:186-219)
//   refit_window/block_totals/  : node boxes as in-order folds over each node's leaf range (no
//   refit_span                    arrival counters). The reference walks each node's whole subtree
//                                 (lbvh.cu:265-297); min/max unions are exact, so both give the same bits.
#include "kk_host.h"
#include "kk_math.h"

#include "radix_hist.h"

namespace kk {
extern "C" size_t kk_sort_workspace_bytes(uint32_t n);
extern "C" size_t kk_sort_workspace_error_offset(uint32_t n);

namespace lbvh {

constexpr int THREADS = 256;
constexpr uint32_t LEAF = 0x80000000u;
constexpr int BOUNDS_MAX_BLOCKS = 2048;

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
    auto take = [&](float x, float y, float z) {
        mn[0] = gmin(mn[0], x);
        mn[1] = gmin(mn[1], y);
        mn[2] = gmin(mn[2], z);
        mx[0] = gmax(mx[0], x);
        mx[1] = gmax(mx[1], y);
        mx[2] = gmax(mx[2], z);
    };
    // four vertices = 48 contiguous bytes = three 16-byte loads per thread, when the array allows it
    const bool aligned = (reinterpret_cast<uintptr_t>(pos) & 15u) == 0;
    const uint32_t quads = aligned ? vertNum / 4 : 0;
    const float4 *pos4 = reinterpret_cast<const float4 *>(pos);
    for (uint32_t q = blockIdx.x * THREADS + tid; q < quads; q += gridDim.x * THREADS) {
        const float4 a = pos4[3ull * q], b = pos4[3ull * q + 1], c = pos4[3ull * q + 2];
        take(a.x, a.y, a.z);
        take(a.w, b.x, b.y);
        take(b.z, b.w, c.x);
        take(c.y, c.z, c.w);
    }
    for (uint32_t v = quads * 4 + blockIdx.x * THREADS + tid; v < vertNum; v += gridDim.x * THREADS)
        take(pos[3ull * v], pos[3ull * v + 1], pos[3ull * v + 2]);
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
    uint32_t *compacted, kk_lbvh_node *nodes, BuildParams *params, kk_lbvh_device_data *out,
    uint32_t *headStatus, uint32_t headTiles, uint32_t *topCount) {
    __shared__ float s_red[THREADS / 64][6];
    const int tid = threadIdx.x;
    // words that later kernels of the build count in or poll start at zero
    for (uint32_t i = tid; i < headTiles; i += THREADS)
        headStatus[i] = 0u;
    if (tid == 0) {
        topCount[0] = 0u;
        topCount[1] = 0u; // the run-head pass's ticket counter (ticketed mode)
        topCount[2] = 0u; // workgroups of block_totals that are through with level 3
    }
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
__device__ __forceinline__ AABB load_box(const float *p) {
    AABB b;
    b.minPos = {p[0], p[1], p[2]};
    b.maxPos = {p[3], p[4], p[5]};
    return b;
}
__device__ __forceinline__ void store_box(float *p, const AABB &b) {
    p[0] = b.minPos.x;
    p[1] = b.minPos.y;
    p[2] = b.minPos.z;
    p[3] = b.maxPos.x;
    p[4] = b.maxPos.y;
    p[5] = b.maxPos.z;
}

// lbvh.cu:136-170, a thread per face; a workgroup takes every gridDim-th block of faces and counts the
// digits of the codes it writes for the sort that follows (radix_hist.h).
constexpr uint32_t MORTON_MAX_BLOCKS = 2048;
__global__ __launch_bounds__(THREADS) void morton_kernel(const float *__restrict__ pos,
                                                         const uint32_t *__restrict__ tri,
                                                         uint32_t faceNum,
                                                         const BuildParams *__restrict__ params,
                                                         uint32_t *__restrict__ mortons,
                                                         float *__restrict__ faceBoxes,
                                                         uint32_t *__restrict__ ghist) {
    __shared__ uint32_t s_hist[4 * rsort::HIST_RADIX];
    rsort::hist_clear<4>(s_hist, threadIdx.x, THREADS);
    const f3 minPos = {params->minPos[0], params->minPos[1], params->minPos[2]};
    const f3 invDlt = {params->invDlt[0], params->invDlt[1], params->invDlt[2]};
    // From the last face to the first: bounds_partial has just streamed the positions front to back and the
    // Infinity Cache holds what it read last.
    for (uint32_t k = blockIdx.x * THREADS + threadIdx.x; k < faceNum; k += gridDim.x * THREADS) {
        const uint32_t fi = faceNum - 1u - k;
        AABB box = AABB::Initial();
        box.Expand(load3(pos, tri[3ull * fi]));
        box.Expand(load3(pos, tri[3ull * fi + 1]));
        box.Expand(load3(pos, tri[3ull * fi + 2]));
        // kept for the leaf boxes: gathering one box per sorted face later costs a third of the lines that
        // gathering its indices and vertices again would. 32 bytes per face (two 16-byte pieces), so that a
        // gathered box never straddles a 128-byte line
        float4 *fb = reinterpret_cast<float4 *>(faceBoxes + 8ull * fi);
        fb[0] = make_float4(box.minPos.x, box.minPos.y, box.minPos.z, box.maxPos.x);
        fb[1] = make_float4(box.maxPos.y, box.maxPos.z, 0.f, 0.f);
        f3 c = .5f * (box.minPos + box.maxPos);
        c = (c - minPos) * invDlt;
        c.x = gmin(gmax(c.x * 1024.f, 0.f), 1023.f);
        c.y = gmin(gmax(c.y * 1024.f, 0.f), 1023.f);
        c.z = gmin(gmax(c.z * 1024.f, 0.f), 1023.f);
        const uint32_t cx = expand10BitsTo32(static_cast<uint32_t>(c.x));
        const uint32_t cy = expand10BitsTo32(static_cast<uint32_t>(c.y));
        const uint32_t cz = expand10BitsTo32(static_cast<uint32_t>(c.z));
        const uint32_t code = cx * 4 + cy * 2 + cz;
        mortons[fi] = code;
        rsort::hist_add<4>(s_hist, code);
    }
    rsort::hist_flush<4>(s_hist, ghist, threadIdx.x, THREADS);
}

// ---- run heads: flag[i] = (i == 0) || key[i-1] != key[i]  (include/cuda/algorithm.h:11-41) ----------
// compacted[c] = index of the c-th head, uniq[c] = its code (lbvh.cu:202), M = their number: flags, scan
// and compaction in one pass over the keys. A tile of HEADS_TILE keys counts its heads, publishes the
// count, adds up the counts of the tiles in front (a wave reads 64 of their status words per round trip
// and stops at the first that already holds an inclusive count) and writes its heads. A tile is its
// workgroup's index, as in the sort (radix_sort.hip): workgroups start in index order. TICKET: the tile
// is a ticket from an atomic counter instead (kk_sort_set_ticketed).
constexpr int HEADS_ITEMS = 16;
constexpr int HEADS_TILE = THREADS * HEADS_ITEMS;
constexpr uint32_t HEADS_AGG = 1u << 30, HEADS_INC = 2u << 30, HEADS_FLAGS = 3u << 30, HEADS_VALUE = ~HEADS_FLAGS;
constexpr uint32_t HEADS_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o)
            v += t;
    }
    return v;
}

template <bool TICKET>
__global__ __launch_bounds__(THREADS) void heads_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                        uint32_t *tileStatus, uint32_t *__restrict__ compacted,
                                                        uint32_t *__restrict__ uniq, kk_lbvh_device_data *out,
                                                        uint32_t *errorFlag, uint32_t *ticket) {
    __shared__ uint32_t s_w[THREADS / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_index[HEADS_TILE], s_code[HEADS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (TICKET) {
        if (tid == 0)
            s_base = atomicAdd(ticket, 1u);
        __syncthreads();
    }
    const uint32_t tile = TICKET ? s_base : blockIdx.x; // (s_base is written again after the next barrier)
    // thread t owns the keys [first, first + HEADS_ITEMS)
    const uint32_t first = tile * HEADS_TILE + tid * HEADS_ITEMS;
    uint32_t k[HEADS_ITEMS];
    if (first + HEADS_ITEMS <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(keys + first);
#pragma unroll
        for (int j = 0; j < HEADS_ITEMS / 4; ++j) {
            const uint4 v = p[j];
            k[4 * j] = v.x, k[4 * j + 1] = v.y, k[4 * j + 2] = v.z, k[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < HEADS_ITEMS; ++j)
            k[j] = first + j < n ? keys[first + j] : 0u;
    }
    uint32_t prev = (first != 0 && first < n) ? keys[first - 1] : 0u;
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < HEADS_ITEMS; ++j) {
        const uint32_t i = first + j;
        if (i < n && (i == 0 || k[j] != prev))
            mask |= 1u << j;
        prev = k[j];
    }
    const uint32_t cnt = __popc(mask);
    const uint32_t incl = wave_inclusive_scan_u32(cnt, lane);
    if (lane == 63)
        s_w[wave] = incl;
    __syncthreads();
    uint32_t within = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        if (w < wave)
            within += s_w[w];
        total += s_w[w];
    }
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store(tileStatus + tile, total | (tile == 0 ? HEADS_INC : HEADS_AGG), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t before = 0, t = tile, spins = 0; // the tiles t - 1, t - 2, ... 0 are still to be added
        bool done = tile == 0;
        while (!done) {
            const bool there = static_cast<uint32_t>(lane) < t; // lane l looks at tile t - 1 - l
            const uint32_t s = there ? __hip_atomic_load(tileStatus + (t - 1u - lane), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)
                                     : 0u;
            const uint32_t f = s & HEADS_FLAGS;
            const unsigned long long notYet = __ballot(there && f == 0u);
            const unsigned long long inclusive = __ballot(there && f == HEADS_INC);
            const int firstNot = notYet ? __ffsll(static_cast<long long>(notYet)) - 1 : 64;
            const int firstInc = inclusive ? __ffsll(static_cast<long long>(inclusive)) - 1 : 64;
            const int left = t < 64u ? static_cast<int>(t) : 64;
            int take = firstNot < firstInc + 1 ? firstNot : firstInc + 1; // words usable this round: lanes [0, take)
            take = take < left ? take : left;
            uint32_t v = lane < take ? (s & HEADS_VALUE) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                v += __shfl_xor(v, o, 64);
            before += v;
            if ((firstInc < 64 && take == firstInc + 1) || static_cast<uint32_t>(take) == t)
                done = true;
            else {
                if (take == 0) {
                    if (++spins > HEADS_SPIN_LIMIT) {
                        if (lane == 0)
                            atomicOr(errorFlag, 1u);
                        done = true;
                    } else
                        __builtin_amdgcn_s_sleep(2);
                }
                t -= static_cast<uint32_t>(take);
            }
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store(tileStatus + tile, ((before + total) & HEADS_VALUE) | HEADS_INC, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_base = before;
            if (tile == gridDim.x - 1)
                out->compacted_face_num = before + total; // M (algorithm.h:27-28)
        }
    }
    // the tile's heads, in order, through LDS: a thread's own heads are up to HEADS_ITEMS words apart from
    // its neighbour's, and stores of single words scattered like that cost a sector each
    uint32_t c = within;
#pragma unroll
    for (int j = 0; j < HEADS_ITEMS; ++j)
        if (mask & (1u << j)) {
            s_index[c] = first + j;
            s_code[c] = k[j];
            ++c;
        }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t e = tid; e < total; e += THREADS) {
        compacted[base + e] = s_index[e];
        uniq[base + e] = s_code[e];
    }
}

// lbvh.cu:221-232: length of the common leading-zero run of code[i] ^ code[j], -1 out of range.
// Codes near the workgroup's own nodes come from LDS (most searches stay within a few hundred leaves).
constexpr int KARRAS_THREADS = 512; // the halo is loaded once per workgroup
constexpr int KARRAS_HALO = 1024;
constexpr int KARRAS_WIN = KARRAS_THREADS + 2 * KARRAS_HALO;
static_assert(KARRAS_WIN % KARRAS_THREADS == 0, "the window is loaded in whole rounds of the workgroup");
// Two sources of codes for the search. The window in LDS serves nearly every node; a search that
// leaves it is flagged and run again from global memory afterwards. The common path thus waits for
// LDS only, and loads from global memory that are in flight beside it (the leaf's box) stay in flight.
struct WindowCodes {
    const uint32_t *lds;
    int winStart;
    bool escaped;
    __device__ __forceinline__ bool fetch(int j, uint32_t &code) {
        const int r = j - winStart;
        if (r < 0 || r >= KARRAS_WIN) {
            escaped = true;
            return false;
        }
        code = lds[r];
        return true;
    }
};
struct GlobalCodes {
    const uint32_t *__restrict__ global;
    __device__ __forceinline__ bool fetch(int j, uint32_t &code) {
        code = global[j];
        return true;
    }
};
template <class C> __device__ __forceinline__ int delta(C &m, uint32_t mi, int j, int inodeNum) {
    uint32_t other;
    if (j < 0 || j > inodeNum || !m.fetch(j, other))
        return -1;
    const uint32_t x = mi ^ other;
    return x == 0 ? 32 : __clz(static_cast<int>(x));
}

// lbvh.cu:29-98: the other end of node i's leaf range and its split position. `afterFirstStage` runs
// between the first and the second stage of the search (the caller's place to ask for memory).
struct RangeAndSplit {
    int iMax, split;
};
template <class C, class F>
__device__ __forceinline__ RangeAndSplit karras_search(C &mortons, uint32_t mi, int i, int inodeNum, bool hasNode,
                                                       F &&afterFirstStage) {
    // a thread without a node (the last leaf's) skips the search: cmp = 33 ends every loop at once
    const int commLft = hasNode ? delta(mortons, mi, i - 1, inodeNum) : 0;
    const int commRht = hasNode ? delta(mortons, mi, i + 1, inodeNum) : 0;
    const int d = (commLft <= commRht) ? 1 : -1;

    int cmp = hasNode ? delta(mortons, mi, i - d, inodeNum) : 33;
    int offsMax = 2;
    while (delta(mortons, mi, i + d * offsMax, inodeNum) > cmp)
        offsMax <<= 1;
    afterFirstStage();

    int offs = 0;
    while (true) {
        offsMax >>= 1;
        if (offsMax < 1)
            break;
        if (delta(mortons, mi, i + d * (offsMax + offs), inodeNum) > cmp)
            offs += offsMax;
    }
    RangeAndSplit r;
    r.iMax = i + d * offs;
    cmp = delta(mortons, mi, r.iMax, inodeNum);

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
    r.split = i + d * offs + (d < 0 ? -1 : 0);
    return r;
}

// lbvh.cu:29-98, one thread per internal node. PARENTS also records the parent of each child (an
// inspection stage; nothing in the build reads it).
//
// Thread i also makes the box of leaf i (lbvh.cu:186-219: the in-order fold of the boxes of its faces,
// which is the in-order fold of their vertices). Its three dependent loads - where the run starts, the
// run's first face, that face's box - are random gathers bound by DRAM; the search is bound by issue and
// LDS. Each load is asked for before a stage of the search and used after it, so one hides the other.
template <bool PARENTS>
__global__ __launch_bounds__(KARRAS_THREADS) void karras_links_kernel(
    const uint32_t *__restrict__ mortonsGlobal, const kk_lbvh_device_data *__restrict__ hdr,
    kk_lbvh_node *__restrict__ nodes, uint32_t *__restrict__ nodeParents,
    uint32_t *__restrict__ leafParents, uint2 *__restrict__ nodeInfo,
    const float *__restrict__ faceBoxes, const uint32_t *__restrict__ faces,
    const uint32_t *__restrict__ compacted, float *__restrict__ leafAABBs) {
    __shared__ uint32_t s_codes[KARRAS_WIN];
    const uint32_t M = hdr->compacted_face_num;
    // 32-bit positions: the leaf count is below 2^30 and a search looks at most one power of two past
    // the range, so i +- offset stays inside an int
    const int inodeNum = static_cast<int>(M) - 1;
    const int blockFirst = static_cast<int>(blockIdx.x) * KARRAS_THREADS;
    if (blockFirst > inodeNum)
        return;
    const int i = blockFirst + static_cast<int>(threadIdx.x);
    const bool hasLeaf = i <= inodeNum, hasNode = i < inodeNum;
    uint32_t faceBeg = 0, faceEnd = 0;
    if (hasLeaf) {
        faceBeg = compacted[i];
        faceEnd = (i == inodeNum) ? hdr->face_num : compacted[i + 1];
    }
    WindowCodes window{s_codes, blockFirst - KARRAS_HALO, false};
    {
        // every load of the window is asked for before the first is waited for: positions outside the
        // array are clamped into it (delta() never looks at those entries), so no load hides in a branch
        uint32_t code[KARRAS_WIN / KARRAS_THREADS];
#pragma unroll
        for (int q = 0; q < KARRAS_WIN / KARRAS_THREADS; ++q) {
            int j = window.winStart + static_cast<int>(threadIdx.x) + q * KARRAS_THREADS;
            j = j < 0 ? 0 : (j > inodeNum ? inodeNum : j);
            code[q] = mortonsGlobal[j];
        }
#pragma unroll
        for (int q = 0; q < KARRAS_WIN / KARRAS_THREADS; ++q)
            s_codes[threadIdx.x + q * KARRAS_THREADS] = code[q];
    }
    __syncthreads();
    if (!hasLeaf)
        return;
    const uint32_t firstFace = faces[faceBeg];
    const uint32_t mi = s_codes[static_cast<int>(threadIdx.x) + KARRAS_HALO];
    float4 boxLo = make_float4(0.f, 0.f, 0.f, 0.f), boxHi = boxLo;
    RangeAndSplit found = karras_search(window, mi, i, inodeNum, hasNode, [&]() {
        const float4 *fb = reinterpret_cast<const float4 *>(faceBoxes + 8ull * firstFace);
        boxLo = fb[0];
        boxHi = fb[1];
    });
    if (window.escaped) {
        GlobalCodes all{mortonsGlobal};
        found = karras_search(all, mi, i, inodeNum, hasNode, []() {});
    }
    const int iMax = found.iMax, split = found.split;

    uint32_t lft = static_cast<uint32_t>(split);
    uint32_t rht = static_cast<uint32_t>(split + 1);
    const int lo = i < iMax ? i : iMax, hi = i < iMax ? iMax : i;
    if (PARENTS && hasNode) {
        (lo == split ? leafParents : nodeParents)[lft] = static_cast<uint32_t>(i);
        (hi == split + 1 ? leafParents : nodeParents)[rht] = static_cast<uint32_t>(i);
        if (i == 0)
            nodeParents[0] = 0xFFFFFFFFu;
    }
    if (lo == split)
        lft |= LEAF;
    if (hi == split + 1)
        rht |= LEAF;
    if (hasNode) {
        // Links and range go to a dense array, eight bytes a node: x = the split position with the two
        // leaf flags (the right child's index is the left one's plus one), y = the other end of the node's
        // leaf range [min(i, iMax), max(i, iMax)]. The refit writes each node whole, links and box together
        // (a node's 8 bytes here and its 24 bytes there would both be partial writes of every 128-byte line).
        nodeInfo[i] = make_uint2(static_cast<uint32_t>(split) | (lft & LEAF) | ((rht & LEAF) >> 1),
                                 static_cast<uint32_t>(iMax));
    }

    AABB box;
    box.minPos = {boxLo.x, boxLo.y, boxLo.z};
    box.maxPos = {boxLo.w, boxHi.x, boxHi.y};
    for (uint32_t k = faceBeg + 1; k < faceEnd; ++k) { // a leaf of several faces with one code
        const float4 *fb = reinterpret_cast<const float4 *>(faceBoxes + 8ull * faces[k]);
        const float4 l4 = fb[0], h4 = fb[1];
        AABB one;
        one.minPos = {l4.x, l4.y, l4.z};
        one.maxPos = {l4.w, h4.x, h4.y};
        box.Expand(one);
    }
    store_box(leafAABBs + 6ull * i, box);
}


// Internal-node boxes without a bottom-up dependency chain.
//
// The reference gives node i the union of everything in its leaf range [lo, hi], folded in leaf order
// with "replace only if strictly smaller / larger" (bounding_box.h:24-41 through lbvh.cu:286-310), so a
// coordinate of the result is the LEFTMOST extreme element of the range (this matters for -0 / +0
// only). Any way of combining in-order pieces that keeps the left operand on a tie gives the same
// bits, so the box of a node is a range query over the leaf boxes:
//   refit_window : one workgroup per REFIT_LEAVES consecutive leaves builds, component by component, a
//                  sparse table in LDS (level k holds the fold of 2^k leaves) and answers every node
//                  whose range lies inside the window with two overlapping table entries. It also
//                  writes the fold of every aligned 32-leaf block and of the whole window.
//   block_totals : folds of aligned blocks of 32^3, 32^4, ... leaves (a few thousand values).
//   refit_span   : a node whose range crosses a window boundary walks its range left to right in the
//                  largest aligned blocks that fit (at most 31 per level and side).
constexpr int REFIT_LEAVES = 1024; // leaves of a window = 32^2
constexpr int REFIT_LOG = 10;
constexpr int RADIX_LOG = 5; // aligned block totals exist for 32^l leaves, l = 1, 2, ...
constexpr int MAX_TOTAL_LEVELS = 7;
constexpr int REFIT_SPAN_BLOCKS = 4096; // sixteen lanes per crossing node (about 1 % of the nodes): 16 nodes per block and pass
static_assert(REFIT_LEAVES == (1 << REFIT_LOG) && REFIT_LOG == 2 * RADIX_LOG, "window = 32^2 leaves");


// Where the totals of each level start (in boxes) inside the totals array; level l has
// ceil(capacity / 32^l) entries. Level 0 is the leaf boxes themselves and is not stored here.
struct TotalsLayout {
    uint32_t levels;                      // highest level stored
    uint64_t start[MAX_TOTAL_LEVELS + 1]; // start[l], l = 1 .. levels
    uint64_t boxes;
};
inline TotalsLayout totals_layout(uint32_t capacity) {
    TotalsLayout t{};
    uint64_t off = 0;
    uint32_t l = 1;
    for (;; ++l) {
        const uint64_t unit = 1ull << (RADIX_LOG * l);
        t.start[l] = off;
        off += (capacity + unit - 1) / unit;
        if (unit >= capacity || l == MAX_TOTAL_LEVELS)
            break;
    }
    t.levels = l;
    t.boxes = off;
    return t;
}

// keeps the left operand on a tie, like AABB::Expand
__device__ __forceinline__ float fold_min(float a, float b) { return (a > b) ? b : a; }
__device__ __forceinline__ float fold_max(float a, float b) { return (a < b) ? b : a; }

// The table of a window: level k (fold of 2^k leaves from every start j) in full for k = 1 .. FINE_LOG,
// and for the levels above only at the starts that are multiples of 32 (folds of 2^(k-5) aligned blocks
// of 32 leaves). A range of fewer than 64 leaves is two overlapping entries of a full level; a longer one
// is its head up to a multiple of 32, its run of aligned blocks (two overlapping entries of a coarse
// level) and its tail, folded left to right. (All ten levels in full are 74 KB of LDS, two workgroups
// per CU; this is 50 KB, three.)
constexpr int FINE_LOG = RADIX_LOG;
// start of full level k (1 .. FINE_LOG): level j holds REFIT_LEAVES + 1 - 2^j entries
__host__ __device__ constexpr int tab_off(int k) { return (k - 1) * (REFIT_LEAVES + 1) - ((1 << k) - 2); }
constexpr int TAB_ENTRIES = tab_off(FINE_LOG + 1);
constexpr int COARSE_PER_LEVEL = REFIT_LEAVES >> RADIX_LOG; // 32 block starts
// coarse level k (FINE_LOG .. REFIT_LOG), entry b = fold of the leaves [32 b, 32 b + 2^k)
__host__ __device__ constexpr int coarse_off(int k) { return (k - FINE_LOG) * COARSE_PER_LEVEL; }
constexpr int COARSE_ENTRIES = coarse_off(REFIT_LOG + 1);

// A workgroup barrier for data exchanged through LDS only: waits for the workgroup's LDS operations, not
// for its global loads, stores and atomics in flight (__syncthreads() waits for those too).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

constexpr int REFIT_THREADS = 512;                      // threads of a window's workgroup ...
constexpr int REFIT_PER = REFIT_LEAVES / REFIT_THREADS; // ... each with this many leaves / nodes (tid, tid + threads, ...)
__global__ __launch_bounds__(REFIT_THREADS) void refit_window_kernel(
    const float *__restrict__ leafAABBs, const kk_lbvh_device_data *__restrict__ hdr,
    kk_lbvh_node *nodes, const uint2 *__restrict__ nodeInfo, float *__restrict__ totals, TotalsLayout tl, uint32_t *__restrict__ topList, uint32_t *topCount) {
    // the table is built for one axis at a time, (min, max) together, two levels per barrier.
    // (Measured at 10 M leaves: 512 threads 183 us, 1024 threads 192 us, 256 threads 248 us.)
    __shared__ float2 s_l0[REFIT_LEAVES];        // level 0 of the current axis
    __shared__ float2 s_tab[TAB_ENTRIES];        // s_tab[tab_off(k) + j] = fold of leaves [j, j + 2^k)
    __shared__ float2 s_coarse[COARSE_ENTRIES];  // s_coarse[coarse_off(k) + b] = fold of leaves [32 b, 32 b + 2^k)
    const uint32_t M = hdr->compacted_face_num;
    // From the last window to the first: karras_links has just written the leaf boxes and links front to
    // back and the Infinity Cache holds what it wrote last.
    const uint32_t windows = (M + REFIT_LEAVES - 1u) / REFIT_LEAVES;
    if (blockIdx.x >= windows)
        return;
    const uint32_t b0 = (windows - 1u - blockIdx.x) * REFIT_LEAVES;
    const uint32_t b1 = (b0 + REFIT_LEAVES < M) ? b0 + REFIT_LEAVES : M; // leaves [b0, b1)
    const int tid = threadIdx.x;
    if (M == 1) {
        // No internal node exists; give traversal a root whose two children are the only leaf.
        if (tid == 0) {
            nodes[0].lft = LEAF;
            nodes[0].rht = LEAF;
            for (int k = 0; k < 3; ++k) {
                nodes[0].aabb_min[k] = leafAABBs[k];
                nodes[0].aabb_max[k] = leafAABBs[3 + k];
            }
        }
        return;
    }
    // this thread's leaves (the fold's identity past the last one) and nodes
    AABB mine[REFIT_PER];
    uint32_t qlo[REFIT_PER], qlen[REFIT_PER];
    bool exists[REFIT_PER], inside[REFIT_PER];
    uint2 link[REFIT_PER];
#pragma unroll
    for (int e = 0; e < REFIT_PER; ++e) {
        const uint32_t i = b0 + tid + e * REFIT_THREADS;
        mine[e] = i < b1 ? load_box(leafAABBs + 6ull * i) : AABB::Initial();
        exists[e] = i < b1 && i < M - 1;
        const uint2 info = exists[e] ? nodeInfo[i] : make_uint2(0u, 0u);
        const uint32_t other = info.y;
        link[e] = make_uint2((info.x & 0x3FFFFFFFu) | (info.x & LEAF),
                             ((info.x & 0x3FFFFFFFu) + 1u) | ((info.x << 1) & LEAF));
        const uint32_t lo = i < other ? i : other, hi = i < other ? other : i;
        inside[e] = exists[e] && lo >= b0 && hi < b1;
        qlo[e] = lo - b0;
        qlen[e] = inside[e] ? hi - lo + 1u : 2u; // range length >= 2
    }
    // Nodes that cross the window's boundary go to the work list: one global atomic per workgroup, asked
    // for now and used after the table work, so that its round trip is not waited for.
    __shared__ uint32_t s_listed, s_listBase;
    if (tid == 0)
        s_listed = 0;
    lds_barrier();
    uint32_t slot[REFIT_PER];
#pragma unroll
    for (int e = 0; e < REFIT_PER; ++e)
        slot[e] = (exists[e] && !inside[e]) ? atomicAdd(&s_listed, 1u) : 0u;
    lds_barrier();
    uint32_t listBase = 0;
    if (tid == 0 && s_listed != 0)
        listBase = atomicAdd(topCount, s_listed);
    float2 res[REFIT_PER][3];
    const bool fullWindow = b1 - b0 == REFIT_LEAVES; // totals of a partial block are never used
    auto comb = [](float2 a, float2 b) { return make_float2(fold_min(a.x, b.x), fold_max(a.y, b.y)); };
    // fold of the leaves [from, from + len), 1 <= len < 64, from the full levels
    auto fine = [&](uint32_t from, uint32_t len) {
        const int k = 31 - __clz(static_cast<int>(len));
        const float2 *lev = k == 0 ? s_l0 : s_tab + tab_off(k);
        return comb(lev[from], lev[from + len - (1u << k)]);
    };
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int e = 0; e < REFIT_PER; ++e)
            s_l0[tid + e * REFIT_THREADS] = make_float2(mine[e].minPos[ax], mine[e].maxPos[ax]);
        lds_barrier();
#pragma unroll
        for (int k = 1; k <= FINE_LOG; k += 2) { // levels k and k + 1 from level k - 1
            const float2 *prev = k == 1 ? s_l0 : s_tab + tab_off(k - 1);
            const int h = 1 << (k - 1);
#pragma unroll
            for (int e = 0; e < REFIT_PER; ++e) {
                const int j = tid + e * REFIT_THREADS;
                if (j + 2 * h <= REFIT_LEAVES) {
                    const float2 lo2 = comb(prev[j], prev[j + h]);
                    s_tab[tab_off(k) + j] = lo2;
                    if (k + 1 <= FINE_LOG && j + 4 * h <= REFIT_LEAVES)
                        s_tab[tab_off(k + 1) + j] = comb(lo2, comb(prev[j + 2 * h], prev[j + 3 * h]));
                }
            }
            lds_barrier();
        }
        // the coarse levels, by the first 32 threads (one wave: a level's entries are written before the
        // next level reads them)
        if (tid < COARSE_PER_LEVEL) {
            float2 v = s_tab[tab_off(FINE_LOG) + (tid << RADIX_LOG)];
            s_coarse[coarse_off(FINE_LOG) + tid] = v;
#pragma unroll
            for (int k = FINE_LOG + 1; k <= REFIT_LOG; ++k) {
                const int hb = 1 << (k - 1 - RADIX_LOG); // half of the level's span, in blocks
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (tid + 2 * hb <= COARSE_PER_LEVEL) {
                    v = comb(v, s_coarse[coarse_off(k - 1) + tid + hb]);
                    s_coarse[coarse_off(k) + tid] = v;
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int e = 0; e < REFIT_PER; ++e) {
            if (!inside[e])
                continue;
            const uint32_t lo = qlo[e], end = qlo[e] + qlen[e]; // leaves [lo, end)
            if (qlen[e] < 2u * COARSE_PER_LEVEL)
                res[e][ax] = fine(lo, qlen[e]);
            else {
                const uint32_t a = (lo + 31u) & ~31u, b = end & ~31u; // at least one block between
                const uint32_t blocks = (b - a) >> RADIX_LOG;
                const int km = 31 - __clz(static_cast<int>(blocks));
                const float2 *lev = s_coarse + coarse_off(FINE_LOG + km);
                float2 r = comb(lev[a >> RADIX_LOG], lev[(b >> RADIX_LOG) - (1u << km)]);
                if (a != lo)
                    r = comb(fine(lo, a - lo), r);
                if (b != end)
                    r = comb(r, fine(b, end - b));
                res[e][ax] = r;
            }
        }
        // aligned block totals: 32 blocks of 32 leaves and the window itself
        if (tid < 32 && (b0 + (tid + 1) * 32 <= M)) {
            const float2 v = s_coarse[coarse_off(FINE_LOG) + tid];
            float *o = totals + 6 * (tl.start[1] + (b0 >> RADIX_LOG) + tid);
            o[ax] = v.x;
            o[3 + ax] = v.y;
        }
        if (tid == 32 && fullWindow) {
            const float2 v = s_coarse[coarse_off(REFIT_LOG)];
            float *o = totals + 6 * (tl.start[2] + (b0 >> REFIT_LOG));
            o[ax] = v.x;
            o[3 + ax] = v.y;
        }
        lds_barrier();
    }
    if (tid == 0)
        s_listBase = listBase; // (thread 0 waits for its atomic here, after the table work)
    lds_barrier();
#pragma unroll
    for (int e = 0; e < REFIT_PER; ++e) {
        if (!exists[e])
            continue;
        const uint32_t i = b0 + tid + e * REFIT_THREADS;
        // the whole node in two 16-byte stores; a node that crosses the window gets its box from refit_span
        uint4 *o = reinterpret_cast<uint4 *>(nodes + i);
        if (inside[e]) {
            o[0] = make_uint4(link[e].x, link[e].y, __float_as_uint(res[e][0].x), __float_as_uint(res[e][1].x));
            o[1] = make_uint4(__float_as_uint(res[e][2].x), __float_as_uint(res[e][0].y),
                              __float_as_uint(res[e][1].y), __float_as_uint(res[e][2].y));
        } else {
            o[0] = make_uint4(link[e].x, link[e].y, 0u, 0u);
            o[1] = make_uint4(0u, 0u, 0u, 0u);
            topList[s_listBase + slot[e]] = i;
        }
    }
}

// Totals of aligned blocks of 32^l leaves, l >= 3, each the in-order fold of its 32 children of
// level l - 1. Thirty-two lanes per block: lane j loads child j and the lanes' boxes are folded in lane
// order (the left operand is kept on ties, so this is the left-to-right fold). Level 3 is spread over the
// grid; the levels above it (a few entries in all) are made by the workgroup that finishes level 3 last.
constexpr int TOTALS_THREADS = 256, TOTALS_PER_BLOCK = TOTALS_THREADS / 32;
__device__ __forceinline__ void fold_children_32(const float *__restrict__ children, float *__restrict__ out,
                                                 int lane32) {
    const float *c = children + 6 * lane32;
    float v[6] = {c[0], c[1], c[2], c[3], c[4], c[5]};
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        float r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            r[k] = __shfl_down(v[k], o, 32);
        if (lane32 + o < 32) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k] = fold_min(v[k], r[k]);
                v[3 + k] = fold_max(v[3 + k], r[3 + k]);
            }
        }
    }
    if (lane32 == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            out[k] = v[k];
    }
}
__global__ __launch_bounds__(TOTALS_THREADS) void block_totals_kernel(
    const kk_lbvh_device_data *__restrict__ hdr, float *totals, TotalsLayout tl, uint32_t *doneCount) {
    __shared__ uint32_t s_last;
    const uint32_t M = hdr->compacted_face_num;
    const int lane32 = threadIdx.x & 31, group = threadIdx.x >> 5;
    const uint32_t full3 = static_cast<uint32_t>(M >> (RADIX_LOG * 3)); // only whole blocks are ever used
    for (uint32_t b = blockIdx.x * TOTALS_PER_BLOCK + group; b < full3; b += gridDim.x * TOTALS_PER_BLOCK)
        fold_children_32(totals + 6 * (tl.start[2] + 32ull * b), totals + 6 * (tl.start[3] + b), lane32);
    if (tl.levels < 4)
        return;
    // the last workgroup to get here sees every level-3 total (release by each, acquire by the last)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = atomicAdd(doneCount, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last == 0u)
        return;
    __threadfence();
    for (uint32_t l = 4; l <= tl.levels; ++l) {
        const uint32_t full = static_cast<uint32_t>(M >> (RADIX_LOG * l));
        for (uint32_t b = group; b < full; b += TOTALS_PER_BLOCK)
            fold_children_32(totals + 6 * (tl.start[l - 1] + 32ull * b), totals + 6 * (tl.start[l] + b), lane32);
        __threadfence_block();
        __syncthreads();
    }
}

// Sixteen lanes per window-crossing node. Its leaf range is covered, left to right, by runs of aligned
// blocks: single leaves up to a multiple of 32, blocks of 32 up to a multiple of 1024, ... and back down
// to single leaves at the right end. The pieces are numbered in that order, each of the sixteen lanes
// folds a contiguous share of them and the lanes' boxes are folded in lane order; Expand keeps its
// left operand on ties, so this is the left-to-right fold. (A thread per node walks some eighty
// dependent loads; a whole wave per node spends more on finding the runs than on the fold.)
__global__ __launch_bounds__(THREADS) void refit_span_kernel(
    const float *__restrict__ leafAABBs, const kk_lbvh_device_data *__restrict__ hdr,
    kk_lbvh_node *nodes, const uint2 *__restrict__ nodeInfo, const float *__restrict__ totals,
    TotalsLayout tl, const uint32_t *__restrict__ topList, const uint32_t *__restrict__ topCount) {
    constexpr int SPAN_LANES = 16, GROUPS = THREADS / SPAN_LANES;
    constexpr int MAX_RUNS = 2 * (MAX_TOTAL_LEVELS + 1);
    __shared__ uint64_t s_start[MAX_TOTAL_LEVELS + 1];
    __shared__ uint32_t s_runLev[GROUPS][MAX_RUNS], s_runNum[GROUPS][MAX_RUNS];
    __shared__ uint64_t s_runFirst[GROUPS][MAX_RUNS];
    if (threadIdx.x <= MAX_TOTAL_LEVELS)
        s_start[threadIdx.x] = tl.start[threadIdx.x];
    __syncthreads();
    const uint32_t count = *topCount;
    const int lane = threadIdx.x & (SPAN_LANES - 1), group = threadIdx.x / SPAN_LANES;
    // the lanes of a group share `e`, so they stay or leave together (the shuffles below stay
    // inside a group)
    for (uint32_t e = blockIdx.x * GROUPS + group; e < count; e += gridDim.x * GROUPS) {
        const uint32_t i = topList[e];
        const uint32_t other = nodeInfo[i].y;
        uint64_t p = i < other ? i : other;                  // next leaf to take in
        const uint64_t end = (i < other ? other : i) + 1ull; // one past the range
        // the runs: level, first block of that level, number of blocks (the lanes of a group compute
        // and write the same values into the group's rows)
        uint32_t *runLev = s_runLev[group], *runNum = s_runNum[group];
        uint64_t *runFirst = s_runFirst[group];
        int runs = 0;
        uint32_t total = 0, l = 0;
        while (true) { // climb
            const uint64_t unit = 1ull << (RADIX_LOG * l), next = unit << RADIX_LOG;
            const uint64_t toAlign = (p & (next - 1)) ? (next - (p & (next - 1))) >> (RADIX_LOG * l) : 0ull;
            const uint64_t fit = (end - p) >> (RADIX_LOG * l);
            const uint32_t n = static_cast<uint32_t>(toAlign < fit ? toAlign : fit);
            runLev[runs] = l, runFirst[runs] = p >> (RADIX_LOG * l), runNum[runs] = n;
            ++runs;
            total += n;
            p += static_cast<uint64_t>(n) * unit;
            if (l < tl.levels && (p & (next - 1)) == 0 && p + next <= end)
                ++l;
            else
                break;
        }
        while (true) { // descend
            const uint32_t n = static_cast<uint32_t>((end - p) >> (RADIX_LOG * l));
            runLev[runs] = l, runFirst[runs] = p >> (RADIX_LOG * l), runNum[runs] = n;
            ++runs;
            total += n;
            p += static_cast<uint64_t>(n) << (RADIX_LOG * l);
            if (l == 0)
                break;
            --l;
        }
        const uint32_t per = (total + SPAN_LANES - 1u) / SPAN_LANES;
        const uint32_t mine0 = lane * per, mine1 = mine0 + per < total ? mine0 + per : total;
        AABB acc = AABB::Initial();
        uint32_t before = 0;
        for (int r = 0; r < runs; ++r) {
            const uint32_t n = runNum[r];
            const uint32_t q0 = mine0 > before ? mine0 - before : 0u;
            const uint32_t q1 = mine1 > before ? (mine1 - before < n ? mine1 - before : n) : 0u;
            const float *src = runLev[r] == 0 ? leafAABBs : totals + 6 * s_start[runLev[r]];
            for (uint32_t q = q0; q < q1; ++q)
                acc.Expand(load_box(src + 6 * (runFirst[r] + q)));
            before += n;
        }
        for (int o = 1; o < SPAN_LANES; o <<= 1) {
            AABB right;
            right.minPos = {__shfl_down(acc.minPos.x, o, SPAN_LANES), __shfl_down(acc.minPos.y, o, SPAN_LANES),
                            __shfl_down(acc.minPos.z, o, SPAN_LANES)};
            right.maxPos = {__shfl_down(acc.maxPos.x, o, SPAN_LANES), __shfl_down(acc.maxPos.y, o, SPAN_LANES),
                            __shfl_down(acc.maxPos.z, o, SPAN_LANES)};
            if (lane + o < SPAN_LANES)
                acc.Expand(right);
        }
        if (lane == 0)
            store_box(nodes[i].aabb_min, acc);
    }
}

struct Ws {
    kk_lbvh_ws_layout pub;
    size_t params, partials, keys1, vals1, faceBoxes, headStatus, topCount, nodeInfo, topList, totals, sortWs;
    size_t sortWsBytes;
    uint32_t headTiles;
};

inline Ws layout(uint32_t N, uint32_t flags) {
    Ws w;
    Carver c;
    const size_t n = N;
    w.headTiles = div_up(N, HEADS_TILE);
    w.params = c.take(sizeof(BuildParams));
    w.partials = c.take(sizeof(float) * 6 * BOUNDS_MAX_BLOCKS);
    w.pub.sorted_mortons = c.take(4 * n); // keys0: Morton codes, sorted in place by the 4 passes
    w.keys1 = c.take(4 * n);
    w.vals1 = c.take(4 * n);
    w.faceBoxes = c.take(32 * n);
    w.pub.unsorted_mortons = (flags & KK_LBVH_KEEP_STAGES) ? c.take(4 * n) : 0;
    w.pub.unique_mortons = c.take(4 * n);
    w.pub.leaf_aabbs = c.take(24 * n);
    w.pub.node_parents = c.take(4 * n);
    w.pub.leaf_parents = c.take(4 * n);
    w.headStatus = c.take(4 * static_cast<size_t>(w.headTiles));
    w.topCount = c.take(12); // the refit's list length; the run-head pass's ticket counter; block_totals' arrivals
    w.nodeInfo = c.take(8 * n);
    w.topList = c.take(4 * n);
    w.totals = c.take(24 * totals_layout(N).boxes);
    w.sortWsBytes = kk_sort_workspace_bytes(N);
    w.sortWs = c.take(w.sortWsBytes);
    // the sort's error word sits right after its histogram and counters (radix_sort.hip: sort_ws)
    w.pub.error_flag = w.sortWs + kk_sort_workspace_error_offset(N);
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
    KK_REQUIRE((reinterpret_cast<uintptr_t>(d_nodes) & 15u) == 0, "d_nodes must be 16-byte aligned");
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
    auto *faceBoxes = reinterpret_cast<float *>(b + w.faceBoxes);
    auto *uniq = reinterpret_cast<uint32_t *>(b + w.pub.unique_mortons);
    auto *leafAABBs = reinterpret_cast<float *>(b + w.pub.leaf_aabbs);
    auto *nodeParents = reinterpret_cast<uint32_t *>(b + w.pub.node_parents);
    auto *leafParents = reinterpret_cast<uint32_t *>(b + w.pub.leaf_parents);
    auto *headStatus = reinterpret_cast<uint32_t *>(b + w.headStatus);
    auto *topCount = reinterpret_cast<uint32_t *>(b + w.topCount);
    auto *totals = reinterpret_cast<float *>(b + w.totals);
    auto *nodeInfo = reinterpret_cast<uint2 *>(b + w.nodeInfo);
    auto *topList = reinterpret_cast<uint32_t *>(b + w.topList);
    const uint32_t N = face_num;

    uint32_t bgrid = div_up(vert_num, THREADS);
    if (bgrid > BOUNDS_MAX_BLOCKS)
        bgrid = BOUNDS_MAX_BLOCKS;
    bounds_partial_kernel<<<bgrid, THREADS, 0, stream>>>(d_positions, vert_num, partials);
    KK_LAUNCH_CHECK();
    build_setup_kernel<<<1, THREADS, 0, stream>>>(partials, bgrid, N, d_faces, d_compacted, d_nodes,
                                                  params, d_out, headStatus, w.headTiles, topCount);
    KK_LAUNCH_CHECK();
    uint32_t *ghist = nullptr;
    int rc = rsort::sort_prepare(b + w.sortWs, w.sortWsBytes, N, stream, &ghist);
    if (rc != KK_OK)
        return rc;
    uint32_t mgrid = div_up(N, THREADS);
    if (mgrid > MORTON_MAX_BLOCKS)
        mgrid = MORTON_MAX_BLOCKS;
    morton_kernel<<<mgrid, THREADS, 0, stream>>>(d_positions, d_face_pos_idx, N, params, keys0, faceBoxes,
                                                 ghist);
    KK_LAUNCH_CHECK();
    if (flags & KK_LBVH_KEEP_STAGES)
        KK_HIP_TRY(hipMemcpyAsync(b + w.pub.unsorted_mortons, keys0, 4ull * N,
                                  hipMemcpyDeviceToDevice, stream));

    // payload = face index; four passes leave keys in keys0 and faces in d_faces
    rc = rsort::sort_pairs(keys0, keys1, d_faces, vals1, true, N, b + w.sortWs, w.sortWsBytes, stream,
                           nullptr, 4, true);
    if (rc != KK_OK)
        return rc;

    if (rsort::ticketed())
        heads_kernel<true><<<w.headTiles, THREADS, 0, stream>>>(keys0, N, headStatus, d_compacted, uniq, d_out,
                                                                 reinterpret_cast<uint32_t *>(b + w.pub.error_flag),
                                                                 topCount + 1);
    else
        heads_kernel<false><<<w.headTiles, THREADS, 0, stream>>>(keys0, N, headStatus, d_compacted, uniq, d_out,
                                                                  reinterpret_cast<uint32_t *>(b + w.pub.error_flag),
                                                                  topCount + 1);
    KK_LAUNCH_CHECK();

    const uint32_t ngrid = div_up(N, KARRAS_THREADS);
    if (flags & KK_LBVH_KEEP_STAGES)
        karras_links_kernel<true><<<ngrid, KARRAS_THREADS, 0, stream>>>(uniq, d_out, d_nodes, nodeParents, leafParents,
                                                                 nodeInfo, faceBoxes, d_faces, d_compacted,
                                                                 leafAABBs);
    else
        karras_links_kernel<false><<<ngrid, KARRAS_THREADS, 0, stream>>>(uniq, d_out, d_nodes, nodeParents, leafParents,
                                                                  nodeInfo, faceBoxes, d_faces, d_compacted,
                                                                  leafAABBs);
    KK_LAUNCH_CHECK();
    const TotalsLayout tl = totals_layout(N);
    refit_window_kernel<<<div_up(N, REFIT_LEAVES), REFIT_THREADS, 0, stream>>>(
        leafAABBs, d_out, d_nodes, nodeInfo, totals, tl, topList, topCount);
    KK_LAUNCH_CHECK();
    if (tl.levels >= 3) {
        // level 3 has at most N / 32^3 entries, eight to a workgroup
        const uint32_t tgrid = div_up((static_cast<uint64_t>(N) >> (3 * RADIX_LOG)) + 1, TOTALS_PER_BLOCK);
        block_totals_kernel<<<tgrid, TOTALS_THREADS, 0, stream>>>(d_out, totals, tl, topCount + 2);
        KK_LAUNCH_CHECK();
    }
    refit_span_kernel<<<REFIT_SPAN_BLOCKS, THREADS, 0, stream>>>(leafAABBs, d_out, d_nodes, nodeInfo,
                                                                 totals, tl, topList, topCount);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
