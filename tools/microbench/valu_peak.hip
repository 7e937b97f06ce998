// valu_peak.hip — measured VALU issue rates of one MI355X, the yardstick for "instruction-issue-bound" in DESIGN.md §5.
// Each wave runs `iters` trips of a block of 64 hand-written VALU instructions (inline asm, so the count is exact):
//   dep     one dependent v_fma_f32 chain            (what a lone wave with no ILP can issue)
//   ind4    four independent v_fma_f32 chains
//   mix     v_mul_f32 / v_add_f32 / v_cndmask / v_max3 mix on 4 chains (the render kernel's flavour: no FMA contraction)
//   pk      v_pk_mul_f32 on 4 chains
//   rcp     v_rcp_f32 (quarter-rate transcendental unit)
//   int     xorshift32 steps (v_lshlrev / v_lshrrev / v_xor) on 2 chains — the RNG of the render kernels
//   cmp     v_cmp_lt_f32 vcc + v_cndmask pairs on 4 chains (box_exact / select chains)
//   salu    the mix block with one s_add_u32 after every VALU instruction (scalar work interleaved, as in the render loops)
// for 1, 2, 4, 5 and 8 waves per SIMD.  Output: G wave-instructions/s over the whole chip and cycles per instruction per SIMD at the
// clock implied by the fastest case.   Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip ;  run: ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float b, float c)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float p0 = a0, p1 = a1;      // second halves of 64-bit pairs for the packed case
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(R16(R4("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(b), "v"(c));
        if (MODE == 1) asm volatile(R16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (MODE == 2) asm volatile(R16("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_cndmask_b32 %2, %2, %0, vcc\n v_max3_f32 %3, %3, %1, %2\n")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (MODE == 3) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x0 = {a0, p0}, x1 = {a1, p1}, x2 = {a2, p0}, x3 = {a3, p1}, bb = {b, b};
            asm volatile(R16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(bb));
            a0 = x0.x; p0 = x0.y; a1 = x1.x; p1 = x1.y; a2 = x2.x; a3 = x3.x;
        }
        if (MODE == 4) asm volatile(R16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (MODE == 5) asm volatile(R16("v_lshlrev_b32 %2, 13, %0\n v_xor_b32 %0, %0, %2\n v_lshrrev_b32 %3, 17, %1\n v_xor_b32 %1, %1, %3\n")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (MODE == 6) asm volatile(R16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %0, vcc\n v_cmp_lt_f32 vcc, %1, %3\n v_cndmask_b32 %3, %3, %1, vcc\n")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
        if (MODE == 7) asm volatile(R16("v_mul_f32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %5\n s_add_u32 s21, s21, 1\n v_cndmask_b32 %2, %2, %0, vcc\n s_add_u32 s22, s22, 1\n v_max3_f32 %3, %3, %1, %2\n s_add_u32 s23, s23, 1\n")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + p0 + p1;
}

template <int MODE>
double run(float* d, int wavesPerSimd, int simds, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = wavesPerSimd * simds;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 100, 1.0001f, 0.5f);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * iters * 64.0 / (ms * 1e-3);          // wave-instructions per second
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4;
    printf("%s: %d CUs, %d SIMDs, clock %d MHz\n", p.name, p.multiProcessorCount, simds, p.clockRate / 1000);
    float* d; hipMalloc(&d, (size_t)simds * 8 * 64 * 4);
    const char* names[8] = {"dep", "ind4", "mix", "pk", "rcp", "int", "cmp", "salu(VALU only counted)"};
    const int iters = 20000;
    for (int w : {1, 2, 4, 5, 8}) {
        double r[8] = {run<0>(d, w, simds, iters), run<1>(d, w, simds, iters), run<2>(d, w, simds, iters), run<3>(d, w, simds, iters), run<4>(d, w, simds, iters / 4),
                       run<5>(d, w, simds, iters), run<6>(d, w, simds, iters), run<7>(d, w, simds, iters)};
        printf("%d wave(s)/SIMD:", w);
        for (int m = 0; m < 8; m++) printf("  %s %.0f G/s (%.2f cyc/instr/SIMD @2.4GHz)", names[m], r[m] / 1e9, 2.4e9 * simds / r[m]);
        printf("\n");
    }
    return 0;
}
