// The radix sort's digit histograms, for kernels that produce the keys: a kernel that writes the keys
// anyway counts their digits on the way (LDS counters per workgroup, one flush of the non-zero ones at
// the end), and the sort then starts at its first pass instead of reading the keys once more.
// Usage: sort_prepare() clears the sort's workspace and names its histogram rows (copy 0); the producing kernel
// calls hist_clear / hist_add per key / hist_flush (a few hundred long-running workgroups: every flush
// is up to 256 atomics per digit on the same words); sort_pairs(..., prepared = true) follows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kk {
namespace rsort {

constexpr int HIST_RADIX = 256;
// The histogram rows exist HIST_COPIES times (copy c at word c * HIST_COPY_STRIDE); a workgroup adds to
// copy blockIdx.x % HIST_COPIES and the sort sums the copies: an eighth of the atomics meet on a word.
constexpr int HIST_COPIES = 8;
constexpr int HIST_COPY_STRIDE = 4 * HIST_RADIX; // four digit rows per copy

// (alsoClearBefore: bytes of the caller's own, directly before ws, to be cleared by the same memset)
int sort_prepare(void *ws, size_t wsBytes, uint32_t n, hipStream_t stream, uint32_t **ghistOut,
                 size_t alsoClearBefore = 0);
int sort_pairs(uint32_t *keysA, uint32_t *keysB, uint32_t *valsA, uint32_t *valsB, bool iota, uint32_t n,
               void *ws, size_t wsBytes, hipStream_t stream, uint32_t **errorFlagOut, int passes,
               bool prepared = false);

// Ticketed mode (kk_sort_set_ticketed): tiles take tickets instead of being their workgroup's index.
bool ticketed();
// The test hook that raises the error word after every sort (kk_debug_sort_force_error) is on.
bool force_error();

// lds: DIGITS * 256 words of the workgroup's LDS
template <int DIGITS> __device__ __forceinline__ void hist_clear(uint32_t *lds, int tid, int threads) {
    for (int i = tid; i < DIGITS * HIST_RADIX; i += threads)
        lds[i] = 0;
    __syncthreads();
}
template <int DIGITS> __device__ __forceinline__ void hist_add(uint32_t *lds, uint32_t key) {
#pragma unroll
    for (int d = 0; d < DIGITS; ++d)
        atomicAdd(&lds[d * HIST_RADIX + ((key >> (8 * d)) & 255u)], 1u);
}
template <int DIGITS>
__device__ __forceinline__ void hist_flush(const uint32_t *lds, uint32_t *ghist, int tid, int threads) {
    __syncthreads();
    uint32_t *copy = ghist + (blockIdx.x % HIST_COPIES) * HIST_COPY_STRIDE;
    for (int i = tid; i < DIGITS * HIST_RADIX; i += threads) {
        const uint32_t c = lds[i];
        if (c)
            atomicAdd(&copy[i], c);
    }
}

} // namespace rsort
} // namespace kk
