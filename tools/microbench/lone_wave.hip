// lone_wave.hip — how fast does ONE wavefront issue a dependent instruction chain, and does the number of active lanes matter?
// (latency mode of render_tiles_kernel: a launch of one window ends on the serial chain of its most expensive wavefront)
// One wavefront per workgroup, `blocks` workgroups; lanes >= `active` leave at once; the rest run `n` rounds of (a) 16 dependent v_fma_f32,
// (b) 8 dependent v_fma_f32 interleaved with 8 dependent s_add_u32, (c) 16 independent v_fma_f32 (4 chains).  Prints cycles per instruction
// from s_memtime of lane 0 of block 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(64) void chain(float* out, unsigned long long* clk, uint32_t active, uint32_t n)
{
    if (threadIdx.x >= active) return;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0000001f, c = 1e-7f, a1 = a + 1, a2 = a + 2, a3 = a + 3;
    uint32_t s = n;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (uint32_t i = 0; i < n; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s)); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * 64 + threadIdx.x] = a + a1 + a2 + a3 + (float)s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
int main()
{
    float* out; unsigned long long* clk; unsigned long long h;
    (void)hipMalloc(&out, 4096 * 64 * 4); (void)hipMalloc(&clk, 8);
    const uint32_t n = 20000;
    const char* names[3] = {"16 dependent v_fma", "8 v_fma + 8 s_add interleaved", "16 v_fma in 4 independent chains"};
    for (int mode = 0; mode < 3; mode++)
        for (uint32_t blocks : {1u, 1024u, 4096u})
            for (uint32_t active : {64u, 32u, 16u, 8u}) {
                for (int rep = 0; rep < 2; rep++) {
                    if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(64), 0, 0, out, clk, active, n);
                    else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(64), 0, 0, out, clk, active, n);
                    else hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(64), 0, 0, out, clk, active, n);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
                printf("%-36s %5u waves, %2u active lanes: %.2f s_memtime ticks per instruction\n", names[mode], blocks, active, (double)h / ((double)n * 16.0));
            }
    return 0;
}
