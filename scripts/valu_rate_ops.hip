// Measures, per instruction, what a SIMD of this GPU issues: wave64 instructions per second with 3 and 5 waves
// per SIMD, for streams of 16 independent instructions of one kind. Two groups show: additions, multiplications,
// fused multiply-adds and the two-operand bit operations issue at about 0.8-0.9 G/s per SIMD, everything else
// (minimum / maximum, conversions, three-operand bit operations, compares, selects, packed arithmetic, the
// mixed-precision fma) at about 0.5-0.58 G/s, the reciprocal at 0.29.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_rate_ops.hip -o scripts/bin/valu_rate_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 8192;
template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float *out, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    const float m = 1.0001f, c = 0.5f; const unsigned long long mask = __ballot(threadIdx.x & 1), mask2 = __ballot(threadIdx.x & 2); const float sm = __builtin_amdgcn_readfirstlane(seed);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            else if (KIND == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(m));
            else if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 9) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            else if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 13) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            else if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 15) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 16) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 17) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 18) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            else if (KIND == 19) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 20) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&a[(i&~1)]));
            else if (KIND == 21) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i]));
            else if (KIND == 22) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 30) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mask));
            else if (KIND == 31) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(m), "v"(c), "s"(mask));
            else if (KIND == 32) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 33) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(i & 1 ? mask : mask2));
            else if (KIND == 34) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 35) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 36) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 37) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 38) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if (KIND == 39) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            else if (KIND == 40) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
            else if (KIND == 41) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            else if (KIND == 42) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            else if (KIND == 43) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            else if (KIND == 50) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            else if (KIND == 51) asm volatile("v_cmp_lt_f32_e64 %3, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(a[i]) : "v"(m), "v"(c), "s"(mask));
            else if (KIND == 52) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_add_f32 %0, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            else if (KIND == 53) asm volatile("v_cmp_lt_f32_e64 %3, %1, %2\n v_add_f32 %0, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(a[i]) : "v"(m), "v"(c), "s"(mask));
            else if (KIND == 54) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            else if (KIND == 55) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            else if (KIND == 23) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int K> void run(const char *name, float *out, int cus) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w : {3, 5}) {
        const int blocks = cus * 4 * w; float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0)); rate_kernel<K><<<blocks, 64>>>(out, 1.f); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-28s %d waves/SIMD: %.3f G wave-instr/s per SIMD\n", name, w, double(ITER) * 16 * w / (ms * 1e-3) / 1e9);
    }
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; CHECK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8));
    printf("fast group\n");
    run<0>("v_fma_f32", out, cus); run<40>("v_fma_f32 (sgpr operand)", out, cus); run<12>("v_fmac_f32", out, cus); run<7>("v_add_f32", out, cus);
    run<19>("v_sub_f32", out, cus); run<8>("v_mul_f32", out, cus); run<2>("v_and_b32", out, cus); run<38>("v_xor_b32", out, cus);
    run<21>("v_lshrrev_b32", out, cus); run<10>("v_add_u32", out, cus); run<14>("v_mov_b32", out, cus);
    printf("slow group\n");
    run<4>("v_max_f32", out, cus); run<35>("v_min_f32", out, cus); run<36>("v_max_u32", out, cus); run<15>("v_min3_f32", out, cus);
    run<37>("v_med3_f32", out, cus); run<18>("v_cvt_f32_ubyte0", out, cus); run<42>("v_cvt_f32_u32", out, cus); run<16>("v_ldexp_f32", out, cus);
    run<3>("v_perm_b32", out, cus); run<9>("v_bfe_u32", out, cus); run<39>("v_bfi_b32", out, cus); run<6>("v_alignbit_b32", out, cus);
    run<11>("v_lshl_add_u32", out, cus); run<22>("v_and_or_b32", out, cus); run<17>("v_mad_u32_u24", out, cus); run<20>("v_pk_mul_f32", out, cus);
    run<1>("v_fma_mix_f32 (f16 low half)", out, cus); run<23>("v_fma_mix_f32 (f16 high half)", out, cus); run<13>("v_cmp_lt_f32 (vcc)", out, cus);
    run<30>("v_cndmask_b32 (sgpr pair)", out, cus); run<55>("v_cndmask_b32_e64 (vcc)", out, cus); run<54>("v_addc_co_u32 (vcc)", out, cus); run<41>("v_rcp_f32", out, cus);
    printf("pairs and triples: the rate is per group of instructions\n");
    run<50>("v_cmp vcc + v_cndmask vcc", out, cus); run<51>("v_cmp sgpr + v_cndmask sgpr", out, cus);
    run<52>("v_cmp vcc + v_add + v_cndmask vcc", out, cus); run<53>("v_cmp sgpr + v_add + v_cndmask sgpr", out, cus);
    return 0;
}
