// Stable LSD radix sort of (uint32 key, uint32 payload) pairs for gfx950: one histogram pass over the
// keys, then four 8-bit "onesweep" passes. Replaces thrust::sequence + thrust::sort_by_key of
// src/ray_tracer/lbvh.cu:110,172 (stable: equal Morton codes keep ascending face index).
//
// One pass, per 4096-key tile (256 threads = 4 waves, 16 keys per lane):
//   1. the tile is the workgroup's index: workgroups start in index order, so every lower-numbered
//      tile has already started and the look-back below cannot wait on a tile that is not running
//      (or, in ticketed mode - kk_sort_set_ticketed - a ticket from an atomic counter, which does not
//      lean on the start order and is what sorts that run beside other grids should use);
//   2. each wave ranks its 1024 keys among equal digits with eight 64-lane ballots per key
//      (match-any) and a per-wave digit counter in LDS;
//   3. thread d owns digit d: sums the four wave counters, publishes the tile's count for the digit
//      (AGGREGATE) in the tile's status row, walks back over earlier tiles' rows until it meets an
//      INCLUSIVE word, then publishes its own inclusive count. A status word is flag(2 bits) +
//      count(30 bits) in one 32-bit agent-scope atomic, so it needs no fence;
//   4. keys and payloads are first scattered into LDS in tile-sorted order, then streamed out so that
//      each run of equal digits is written to consecutive addresses (coalesced stores).
// HBM traffic per key: 4 B (histogram) + 4 x (8 B read + 8 B write) = 68 B.
#include "kk_host.h"
#include "radix_hist.h"

#include <atomic>
#include <mutex>

namespace kk {
namespace rsort {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int ITEMS = 16;
constexpr int TILE = THREADS * ITEMS;
constexpr int RADIX = 256;
constexpr int PASSES = 4;
constexpr uint32_t FLAG_AGG = 0x40000000u;
constexpr uint32_t FLAG_INC = 0x80000000u;
constexpr uint32_t FLAG_MASK = 0xC0000000u;
constexpr uint32_t VALUE_MASK = 0x3FFFFFFFu;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

static_assert(THREADS == RADIX, "thread d owns digit d");
static_assert(RADIX == HIST_RADIX && HIST_COPY_STRIDE == PASSES * RADIX, "radix_hist.h describes these rows");

__device__ __forceinline__ uint32_t ld_status(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o)
            v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over the 256-thread block. s_tmp holds WAVES words.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_tmp, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = wave_inclusive_scan(v, lane);
    if (lane == 63)
        s_tmp[wave] = incl;
    __syncthreads();
    uint32_t prefix = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
        if (w < wave)
            prefix += s_tmp[w];
    __syncthreads();
    return prefix + incl - v;
}

// ghist[p][d] += number of keys whose digit p equals d, for all four digits in one read of the keys.
__global__ __launch_bounds__(THREADS) void histogram_kernel(const uint32_t *__restrict__ keys,
                                                            uint32_t n,
                                                            uint32_t *__restrict__ ghist) {
    __shared__ uint32_t s_h[PASSES * RADIX];
    const int tid = threadIdx.x;
    for (int i = tid; i < PASSES * RADIX; i += THREADS)
        s_h[i] = 0;
    __syncthreads();
    const uint32_t n4 = n / 4;
    const uint4 *keys4 = reinterpret_cast<const uint4 *>(keys);
    for (uint32_t i = blockIdx.x * THREADS + tid; i < n4; i += gridDim.x * THREADS) {
        uint4 k = keys4[i];
        uint32_t kk4[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&s_h[kk4[j] & 255u], 1u);
            atomicAdd(&s_h[RADIX + ((kk4[j] >> 8) & 255u)], 1u);
            atomicAdd(&s_h[2 * RADIX + ((kk4[j] >> 16) & 255u)], 1u);
            atomicAdd(&s_h[3 * RADIX + (kk4[j] >> 24)], 1u);
        }
    }
    if (blockIdx.x == 0 && tid < static_cast<int>(n - n4 * 4)) {
        uint32_t k = keys[n4 * 4 + tid];
        atomicAdd(&s_h[k & 255u], 1u);
        atomicAdd(&s_h[RADIX + ((k >> 8) & 255u)], 1u);
        atomicAdd(&s_h[2 * RADIX + ((k >> 16) & 255u)], 1u);
        atomicAdd(&s_h[3 * RADIX + (k >> 24)], 1u);
    }
    __syncthreads();
    for (int i = tid; i < PASSES * RADIX; i += THREADS) {
        uint32_t c = s_h[i];
        if (c)
            atomicAdd(&ghist[(blockIdx.x % HIST_COPIES) * HIST_COPY_STRIDE + i], c);
    }
}

template <bool IOTA, bool TICKET>
__global__ __launch_bounds__(THREADS) void onesweep_pass_kernel(
    const uint32_t *__restrict__ keysIn, const uint32_t *__restrict__ valsIn,
    uint32_t *__restrict__ keysOut, uint32_t *__restrict__ valsOut, uint32_t n, uint32_t shift,
    const uint32_t *__restrict__ ghist, uint32_t *tileStatus, uint32_t *errorFlag, uint32_t *ticket) {
    __shared__ uint32_t s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];
    __shared__ uint32_t s_whist[WAVES][RADIX];
    __shared__ uint32_t s_digitStart[RADIX];
    __shared__ uint32_t s_globalBase[RADIX];
    __shared__ uint32_t s_tmp[WAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    if (TICKET && tid == 0)
        s_tmp[0] = atomicAdd(ticket, 1u);
    for (int i = tid; i < WAVES * RADIX; i += THREADS)
        (&s_whist[0][0])[i] = 0;
    __syncthreads();
    // The tile is the workgroup's index. The look-back below waits for the tiles in front, so each of
    // them must be running or done by then: workgroups of a grid are started in index order (every XCD
    // takes its share in order) and all ask for the same resources, so none is passed over. (A ticket
    // from an atomic counter would not lean on that order, but 2 442 returning atomics on one word cost
    // 12 us per pass at 10 M keys.) Should the order ever not hold, the spin limit ends the wait and
    // raises the error word. TICKET: the tile is the order in which the workgroups got here instead; a
    // workgroup that holds a ticket runs, so every tile in front of it runs or is done, whatever else
    // shares the GPU.
    const uint32_t tile = TICKET ? s_tmp[0] : blockIdx.x;
    const uint32_t tileBase = tile * TILE; // (s_tmp is written again only after the next barrier)

    // 1. load: wave w owns keys [tileBase + 1024 w, +1024), item i of lane l is key 64 i + l of those
    uint32_t key[ITEMS], val[ITEMS];
    const uint32_t base = tileBase + wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = base + i * 64;
        const bool valid = idx < n;
        key[i] = valid ? keysIn[idx] : 0xFFFFFFFFu;
        if (IOTA)
            val[i] = idx;
        else
            val[i] = valid ? valsIn[idx] : 0u;
    }

    // 2. rank among equal digits inside the wave, in key order
    uint32_t rank[ITEMS];
    uint32_t *whist = s_whist[wave];
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t d = (key[i] >> shift) & 255u;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const int leader = __ffsll(static_cast<long long>(peers)) - 1;
        const uint32_t cnt = __popcll(peers);
        const uint32_t below = __popcll(peers & lanesBelow);
        uint32_t prev = 0;
        if (lane == leader) {
            prev = whist[d];
            whist[d] = prev + cnt;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        prev = __shfl(prev, leader, 64);
        rank[i] = prev + below;
    }
    __syncthreads();

    // 3. thread d owns digit d: its count in this tile goes out at once (the tiles behind wait for it)
    const int dg = tid;
    uint32_t total = 0, dstart, gstart;
    uint32_t *row = tileStatus + static_cast<size_t>(tile) * RADIX + dg;
    {
        uint32_t c[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            c[w] = s_whist[w][dg];
            s_whist[w][dg] = total; // exclusive over waves
            total += c[w];
        }
        st_status(row, total | (tile == 0 ? FLAG_INC : FLAG_AGG));
        dstart = block_exclusive_scan(total, s_tmp, tid);
        uint32_t digitTotal = 0; // over the copies of the row (radix_hist.h)
#pragma unroll
        for (int c = 0; c < HIST_COPIES; ++c)
            digitTotal += ghist[c * HIST_COPY_STRIDE + dg];
        gstart = block_exclusive_scan(digitTotal, s_tmp, tid);
        s_digitStart[dg] = dstart;
    }
    __syncthreads();

    // 4. scatter into LDS in tile-sorted order; the tiles in front publish meanwhile
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t d = (key[i] >> shift) & 255u;
        const uint32_t p = s_digitStart[d] + s_whist[wave][d] + rank[i];
        s_keys[p] = key[i];
        s_vals[p] = val[i];
    }

    // 5. where the digit starts in the output: the counts of all tiles in front
    {
        uint32_t exclusive = 0;
        if (tile != 0) {
            // look back over the predecessors, LOOK of them per round trip to memory: their words are
            // read together and used in order, up to the first that is not published yet
            constexpr uint32_t LOOK = 4; // wider (8, 16, 32) was measured: no faster, the wait is not here
            uint32_t t = tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                uint32_t s[LOOK];
#pragma unroll
                for (uint32_t u = 0; u < LOOK; ++u)
                    s[u] = u <= t ? ld_status(tileStatus + static_cast<size_t>(t - u) * RADIX + dg) : 0u;
                bool open = true; // the words are used in order up to the first that is not there
                uint32_t used = 0;
#pragma unroll
                for (uint32_t u = 0; u < LOOK; ++u) {
                    const uint32_t f = s[u] & FLAG_MASK;
                    const bool ready = open && u <= t && f != 0u;
                    exclusive += ready ? (s[u] & VALUE_MASK) : 0u;
                    used += ready ? 1u : 0u;
                    const bool last = ready && (f == FLAG_INC || u == t);
                    done = done || last;
                    open = ready && !last;
                }
                if (!done) {
                    if (used == 0u) { // tile t not published yet: read again from it
                        if (++spins > SPIN_LIMIT) {
                            atomicOr(errorFlag, 1u);
                            done = true;
                        } else
                            __builtin_amdgcn_s_sleep(2);
                    }
                    t -= used;
                }
            }
            st_status(row, ((exclusive + total) & VALUE_MASK) | FLAG_INC);
        }
        s_globalBase[dg] = gstart + exclusive - dstart;
    }
    __syncthreads();

    // 6. stream out
    const uint32_t remaining = n - tileBase;
    const uint32_t validCount = remaining < static_cast<uint32_t>(TILE) ? remaining : TILE;
    for (uint32_t j = tid; j < validCount; j += THREADS) {
        const uint32_t k = s_keys[j];
        const uint32_t d = (k >> shift) & 255u;
        const uint32_t dst = s_globalBase[d] + j;
        keysOut[dst] = k;
        valsOut[dst] = s_vals[j];
    }
}

// Test hook: raises the error word as a look-back that gave up would.
__global__ void raise_error_kernel(uint32_t *errorFlag) { atomicOr(errorFlag, 2u); }

static std::atomic<int> g_ticketed{0};
static std::atomic<int> g_forceError{0};

bool ticketed() { return g_ticketed.load(std::memory_order_relaxed) != 0; }
bool force_error() { return g_forceError.load(std::memory_order_relaxed) != 0; }

struct SortWs {
    size_t ghist, counters, error, status, total;
    uint32_t tiles;
};

inline SortWs sort_ws(uint32_t n) {
    SortWs w;
    Carver c;
    w.tiles = div_up(n, TILE);
    w.ghist = c.take(sizeof(uint32_t) * HIST_COPIES * HIST_COPY_STRIDE);
    w.counters = c.take(sizeof(uint32_t) * PASSES); // the passes' ticket counters (ticketed mode)
    w.error = c.take(sizeof(uint32_t));
    w.status = c.take(sizeof(uint32_t) * PASSES * static_cast<size_t>(w.tiles) * RADIX);
    w.total = c.off;
    return w;
}

// Enqueues the whole sort. errorFlagOut (optional) receives the device address of the error word.
// `passes` (1..4) sorts on the low 8 * passes key bits only; after an odd number of passes the result
// is in the B arrays.
// Clears the sort's workspace and names its histogram rows (row p = digit p, 256 words each), for a
// caller whose own kernel counts the digits (radix_hist.h).
int sort_prepare(void *ws, size_t wsBytes, uint32_t n, hipStream_t stream, uint32_t **ghistOut,
                 size_t alsoClearBefore) {
    SortWs w = sort_ws(n);
    if (wsBytes < w.total) {
        set_error("sort workspace too small: %zu < %zu", wsBytes, w.total);
        return KK_ERR_WORKSPACE;
    }
    KK_HIP_TRY(hipMemsetAsync(static_cast<char *>(ws) - alsoClearBefore, 0, w.total + alsoClearBefore, stream));
    *ghistOut = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + w.ghist);
    return KK_OK;
}

int sort_pairs(uint32_t *keysA, uint32_t *keysB, uint32_t *valsA, uint32_t *valsB, bool iota,
               uint32_t n, void *ws, size_t wsBytes, hipStream_t stream, uint32_t **errorFlagOut,
               int passes, bool prepared) {
    SortWs w = sort_ws(n);
    if (wsBytes < w.total) {
        set_error("sort workspace too small: %zu < %zu", wsBytes, w.total);
        return KK_ERR_WORKSPACE;
    }
    char *b = static_cast<char *>(ws);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(b + w.ghist);
    uint32_t *error = reinterpret_cast<uint32_t *>(b + w.error);
    uint32_t *status = reinterpret_cast<uint32_t *>(b + w.status);
    uint32_t *tickets = reinterpret_cast<uint32_t *>(b + w.counters);
    const bool ticketed = rsort::ticketed();
    if (errorFlagOut)
        *errorFlagOut = error;
    if (!prepared) {
        // ghist, counters, error and the status rows are contiguous pieces of one allocation
        KK_HIP_TRY(hipMemsetAsync(b, 0, w.total, stream));

        // every workgroup ends with up to 1024 atomics on the same 1024 words: few, long-running workgroups
        const uint32_t hgrid = w.tiles < 512u ? (w.tiles ? w.tiles : 1u) : 512u;
        histogram_kernel<<<hgrid, THREADS, 0, stream>>>(keysA, n, ghist);
        KK_LAUNCH_CHECK();
    }

    uint32_t *kin = keysA, *kout = keysB, *vin = valsA, *vout = valsB;
    for (int p = 0; p < passes; ++p) {
        uint32_t *st = status + static_cast<size_t>(p) * w.tiles * RADIX;
#define KK_LAUNCH_PASS(IOTA, TICKET)                                                               \
    onesweep_pass_kernel<IOTA, TICKET><<<w.tiles, THREADS, 0, stream>>>(                           \
        kin, (IOTA) ? nullptr : vin, kout, vout, n, 8u * p, ghist + p * RADIX, st, error, tickets + p)
        if (p == 0 && iota) {
            if (ticketed)
                KK_LAUNCH_PASS(true, true);
            else
                KK_LAUNCH_PASS(true, false);
        } else {
            if (ticketed)
                KK_LAUNCH_PASS(false, true);
            else
                KK_LAUNCH_PASS(false, false);
        }
#undef KK_LAUNCH_PASS
        KK_LAUNCH_CHECK();
        uint32_t *t = kin;
        kin = kout;
        kout = t;
        t = vin;
        vin = vout;
        vout = t;
    }
    if (g_forceError.load(std::memory_order_relaxed) != 0) {
        raise_error_kernel<<<1, 1, 0, stream>>>(error);
        KK_LAUNCH_CHECK();
    }
    return KK_OK;
}

} // namespace rsort
} // namespace kk

extern "C" int kk_sort_set_ticketed(int on) { return kk::rsort::g_ticketed.exchange(on ? 1 : 0); }
extern "C" int kk_debug_sort_force_error(int on) { return kk::rsort::g_forceError.exchange(on ? 1 : 0); }

// The word a look-back raises when it gives up, read back after the work queued on `stream`:
// KK_ERR_DEVICE when it is set (the arrays the sort wrote are then no permutation of its input).
static int sort_error_status(const uint32_t *d_error, void *stream_, const char *what) {
    uint32_t word = 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KK_HIP_TRY(hipMemcpyAsync(&word, d_error, sizeof(word), hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    if (word != 0) {
        kk::set_error("%s: the radix sort's look-back gave up waiting for a tile in front (error word %u); "
                      "its output is not a permutation of its input", what, word);
        return KK_ERR_DEVICE;
    }
    return KK_OK;
}
extern "C" int kk_sort_workspace_status(const void *d_workspace, uint32_t n, void *stream) {
    KK_REQUIRE(d_workspace != nullptr, "null pointer");
    KK_REQUIRE(n > 0, "n out of range");
    return sort_error_status(reinterpret_cast<const uint32_t *>(static_cast<const char *>(d_workspace) +
                                                                kk::rsort::sort_ws(n).error),
                             stream, "kk_sort_pairs_u32");
}

extern "C" size_t kk_sort_workspace_bytes(uint32_t n) { return kk::rsort::sort_ws(n).total; }
extern "C" size_t kk_sort_workspace_error_offset(uint32_t n) { return kk::rsort::sort_ws(n).error; }

// ---- the sort for callers that cannot allocate --------------------------------------------------------
// The sort's scratch (digit histograms, tile counters, error word, look-back status rows) in a
// file-scope device array instead of a caller's workspace. Each GPU holds its own copy at its own
// address: the address is asked for once per device and kept; host code only ever uses that pointer.
namespace kk {
namespace rsort {

constexpr size_t static_scratch_bytes(uint32_t n) {
    // what sort_ws() carves: ghist, counters and error word (one 256-byte aligned piece each), then
    // the status rows of all passes
    const size_t tiles = (static_cast<size_t>(n) + TILE - 1) / TILE;
    return sizeof(uint32_t) * HIST_COPIES * HIST_COPY_STRIDE + 256 + 256 + sizeof(uint32_t) * PASSES * tiles * RADIX;
}
constexpr size_t STATIC_SCRATCH_BYTES = static_scratch_bytes(KK_SORT_STATIC_MAX_N);
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratchMutex;
static void *g_scratchPtr[MAX_DEVICES] = {};

// Device address of this GPU's copy of `scratch`, looked up on the first call per device.
static int scratch_ptr(void **out) {
    int dev = -1;
    KK_HIP_TRY(hipGetDevice(&dev));
    KK_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[dev]) {
        void *p = nullptr;
        KK_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        KK_REQUIRE(p != nullptr, "no device address for the static scratch");
        g_scratchPtr[dev] = p;
    }
    *out = g_scratchPtr[dev];
    return KK_OK;
}

} // namespace rsort
} // namespace kk

extern "C" int kk_sort_pairs_u32_static(uint32_t *d_keys_a, uint32_t *d_keys_b, uint32_t *d_vals_a,
                                        uint32_t *d_vals_b, int vals_are_iota, uint32_t n,
                                        void *stream) {
    using namespace kk::rsort;
    KK_REQUIRE(d_keys_a && d_keys_b && d_vals_a && d_vals_b, "null pointer");
    KK_REQUIRE(n > 0, "n out of range");
    KK_REQUIRE(n <= KK_SORT_STATIC_MAX_N,
               "n exceeds KK_SORT_STATIC_MAX_N, the size the static scratch is made for");
    KK_REQUIRE((reinterpret_cast<uintptr_t>(d_keys_a) & 15u) == 0, "d_keys_a must be 16-byte aligned");
    KK_REQUIRE(sort_ws(n).total <= STATIC_SCRATCH_BYTES, "static scratch too small");
    void *ws = nullptr;
    const int rc = scratch_ptr(&ws);
    if (rc != KK_OK)
        return rc;
    // sort_pairs clears the part of the scratch it uses (histograms, counters, status rows) first
    return sort_pairs(d_keys_a, d_keys_b, d_vals_a, d_vals_b, vals_are_iota != 0, n, ws,
                      STATIC_SCRATCH_BYTES, static_cast<hipStream_t>(stream), nullptr, PASSES);
}

extern "C" int kk_sort_static_init(void) {
    void *ws = nullptr;
    return kk::rsort::scratch_ptr(&ws);
}

extern "C" int kk_sort_static_status(void *stream) {
    void *ws = nullptr;
    const int rc = kk::rsort::scratch_ptr(&ws);
    if (rc != KK_OK)
        return rc;
    // the error word's place does not depend on n (sort_ws: histograms, counters, error, status rows)
    return sort_error_status(reinterpret_cast<const uint32_t *>(static_cast<const char *>(ws) +
                                                                kk::rsort::sort_ws(1).error),
                             stream, "kk_sort_pairs_u32_static");
}

extern "C" int kk_sort_pairs_u32(uint32_t *d_keys_a, uint32_t *d_keys_b, uint32_t *d_vals_a,
                                 uint32_t *d_vals_b, int vals_are_iota, uint32_t n,
                                 void *d_workspace, size_t workspace_bytes, void *stream) {
    KK_REQUIRE(d_keys_a && d_keys_b && d_vals_a && d_vals_b && d_workspace, "null pointer");
    KK_REQUIRE(n > 0 && n <= KK_LBVH_MAX_FACES, "n out of range");
    KK_REQUIRE((reinterpret_cast<uintptr_t>(d_keys_a) & 15u) == 0, "d_keys_a must be 16-byte aligned");
    return kk::rsort::sort_pairs(d_keys_a, d_keys_b, d_vals_a, d_vals_b, vals_are_iota != 0, n,
                                 d_workspace, workspace_bytes, static_cast<hipStream_t>(stream),
                                 nullptr, kk::rsort::PASSES);
}
