// Measures what a SIMD of this GPU issues: wave64 VALU instructions per second with 1..8 waves per SIMD, for
// streams of independent v_fma_f32, v_max3_f32, v_cvt_f32_ubyte1, v_mul_lo_u32, v_pk_fma_f32 and a mix of
// them in the proportions of a wide-node visit (2 conversions : 1 packed fma : 1 max3). The rate from the
// launch's wall time is the figure to use (the shader clock follows the load); cycles are shown beside it.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_rate.hip -o scripts/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 16384;

template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float *out, unsigned long long *cycles, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        a[i] = seed + threadIdx.x * 0.001f + i;
    const float m = 1.0001f, c = 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 1)
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 2)
                asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
            else if (KIND == 3)
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 5) { // the mix
                if ((i & 3) < 2)
                    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
                else if ((i & 3) == 2)
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                else
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = static_cast<unsigned long long>(t1 - t0);
}

// packed fma: two floats per lane and instruction
__global__ __launch_bounds__(64) void rate_pk_kernel(float *out, unsigned long long *cycles, float seed) {
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = float2v{seed + threadIdx.x * 0.001f + i, seed + i};
    const float2v m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = static_cast<unsigned long long>(t1 - t0);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float *out;
    unsigned long long *cyc;
    const int maxBlocks = cus * 4 * 8;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * maxBlocks));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * maxBlocks));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[6] = {"v_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_mul_lo_u32", "v_pk_fma_f32",
                            "mix cvt:cvt:max3:fma"};
    printf("4 cycles per instruction at 2.4 GHz would be 0.600 G wave-instructions/s per SIMD, 2 cycles 1.200\n");
    for (int kind = 0; kind < 6; ++kind) {
        for (int wavesPerSimd : {1, 2, 3, 4, 5, 8}) {
            // one-wave workgroups: the dispatcher spreads them over the SIMDs of all CUs
            const int blocks = cus * 4 * wavesPerSimd;
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (kind) {
                case 0: rate_kernel<0><<<blocks, 64>>>(out, cyc, 1.f); break;
                case 1: rate_kernel<1><<<blocks, 64>>>(out, cyc, 1.f); break;
                case 2: rate_kernel<2><<<blocks, 64>>>(out, cyc, 1.f); break;
                case 3: rate_kernel<3><<<blocks, 64>>>(out, cyc, 1.f); break;
                case 5: rate_kernel<5><<<blocks, 64>>>(out, cyc, 1.f); break;
                default: rate_pk_kernel<<<blocks, 64>>>(out, cyc, 1.f); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> h(blocks);
            CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto v : h)
                mean += double(v);
            mean /= blocks;
            const double instrPerWave = double(ITER) * 16;
            // cycles a SIMD spends per wave instruction when `wavesPerSimd` waves share it
            printf("%-22s %d waves/SIMD: %.3f G wave-instructions/s per SIMD (launch %.3f ms; a wave: %.2f cycles "
                   "per own instruction by the shader clock counter)\n",
                   names[kind], wavesPerSimd, instrPerWave * wavesPerSimd / (ms * 1e-3) / 1e9, ms,
                   mean / instrPerWave);
        }
    }
    return 0;
}
