// Closest-hit kernel over an explicit ray buffer: persistent wave64 workgroups that pull rays from a
// global counter. A wave keeps walking while most of its lanes have a ray; when enough lanes have
// finished, the idle lanes are found with one ballot, numbered with a prefix popcount, and refilled
// from the counter with a single atomic per wave, so lanes do not sit idle behind the longest ray.
// The per-lane stack of far children lives in LDS, one word per level, interleaved across lanes.
#include "kk_host.h"
#include "traverse.h"

namespace kk {
namespace trav {

constexpr int THREADS = 256;
constexpr int REFILL_BELOW = 40; // refill once fewer than this many of the 64 lanes hold a ray

template <bool COUNT>
__global__ __launch_bounds__(THREADS) void closest_hit_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, const float *__restrict__ positions,
    const uint32_t *__restrict__ tri, const float *__restrict__ rays, uint32_t rayNum,
    float *__restrict__ tOut, float *__restrict__ baryOut, uint32_t *__restrict__ primOut,
    uint32_t *rayCounter, unsigned long long *counters) {
    __shared__ uint32_t s_stack[STACK_DEPTH * THREADS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint32_t *stack = s_stack + tid;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;

    Scene s;
    s.nodes = lbvh->internal_nodes;
    s.faces = lbvh->faces;
    s.compacted = lbvh->compacted_faces;
    s.faceNum = lbvh->face_num;
    s.compactedNum = lbvh->compacted_face_num;
    s.positions = positions;
    s.tri = tri;

    bool active = false;
    bool poolEmpty = false;
    uint32_t rayIdx = 0;
    Ray ray;
    Walk w;
    Hit best;
    uint32_t nodeFetches = 0, triTests = 0;
    unsigned long long totNodes = 0, totTris = 0, totRays = 0;

    while (true) {
        // refill idle lanes
        const unsigned long long idleMask = __ballot(!active);
        if (idleMask != 0ull && !poolEmpty) {
            const uint32_t want = __popcll(idleMask);
            uint32_t base = 0;
            if (lane == (__ffsll(static_cast<long long>(idleMask)) - 1))
                base = atomicAdd(rayCounter, want);
            base = __shfl(base, __ffsll(static_cast<long long>(idleMask)) - 1, 64);
            if (base + want >= rayNum || base + want < base)
                poolEmpty = true; // wave-uniform: every lane sees the same base and want
            if (!active) {
                const uint32_t r = base + __popcll(idleMask & lanesBelow);
                if (r < rayNum && r >= base) {
                    rayIdx = r;
                    const float *p = rays + 6ull * r;
                    ray.pos = {p[0], p[1], p[2]};
                    ray.dir = {p[3], p[4], p[5]};
                    nodeFetches = triTests = 0;
                    walk_begin<COUNT>(s, ray, w, best, nodeFetches);
                    active = true;
                }
            }
        }
        if (__ballot(active) == 0ull)
            break;

        // walk until too few lanes are left (or, once the pool is dry, until all are done)
        while (true) {
            if (active) {
                if (!w.done)
                    walk_step<COUNT>(s, ray, w, best, stack, THREADS, nodeFetches, triTests);
                if (w.done) {
                    tOut[rayIdx] = best.t;
                    baryOut[2ull * rayIdx] = best.bx;
                    baryOut[2ull * rayIdx + 1] = best.by;
                    primOut[rayIdx] = best.fi;
                    if (COUNT) {
                        totNodes += nodeFetches;
                        totTris += triTests;
                        ++totRays;
                    }
                    active = false;
                }
            }
            const int alive = __popcll(__ballot(active));
            if (alive == 0 || (!poolEmpty && alive < REFILL_BELOW))
                break;
        }
    }

    if (COUNT) {
        // one atomic per wave
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

} // namespace trav
} // namespace kk

extern "C" size_t kk_closest_hit_workspace_bytes(void) { return 256; }

extern "C" int kk_closest_hit(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                              const uint32_t *d_face_pos_idx, const float *d_rays, uint32_t ray_num,
                              float *d_t, float *d_bary, uint32_t *d_prim, int mode,
                              unsigned long long *d_counters, void *d_workspace,
                              size_t workspace_bytes, void *stream_) {
    using namespace kk;
    using namespace kk::trav;
    KK_REQUIRE(d_lbvh && d_positions && d_face_pos_idx && d_rays && d_t && d_bary && d_prim &&
                   d_workspace,
               "null pointer");
    KK_REQUIRE(workspace_bytes >= 256, "workspace too small");
    KK_REQUIRE(mode == KK_TRAVERSE_ORDERED, "unknown traversal mode");
    if (ray_num == 0)
        return KK_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    uint32_t *rayCounter = static_cast<uint32_t *>(d_workspace);
    KK_HIP_TRY(hipMemsetAsync(rayCounter, 0, 256, stream));
    // 4 resident workgroups per CU; fewer when there are not enough rays to feed them
    uint32_t grid = div_up(ray_num, THREADS);
    const uint32_t resident = 4u * static_cast<uint32_t>(device_cu_count());
    if (grid > resident)
        grid = resident;
    if (d_counters)
        closest_hit_kernel<true><<<grid, THREADS, 0, stream>>>(d_lbvh, d_positions, d_face_pos_idx,
                                                               d_rays, ray_num, d_t, d_bary, d_prim,
                                                               rayCounter, d_counters);
    else
        closest_hit_kernel<false><<<grid, THREADS, 0, stream>>>(d_lbvh, d_positions, d_face_pos_idx,
                                                                d_rays, ray_num, d_t, d_bary, d_prim,
                                                                rayCounter, nullptr);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
