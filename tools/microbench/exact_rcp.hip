// exact_rcp.hip — exhaustive check (all 2^32 float bit patterns) of candidate short sequences for the correctly rounded reciprocal against the compiler's
// IEEE division (v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup), inside the guard range the kernels would use.  Prints mismatch counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rcp_nr1(float x) { float r = hw_rcp(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float rcp_nr2(float x) { float r = rcp_nr1(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float div_fast(float a, float b) { const float r = rcp_nr1(b); const float q = a * r; const float rem = __builtin_fmaf(-b, q, a); return __builtin_fmaf(rem, r, q); }
// quotients a / b of pseudo-random operands with 2^-60 <= |a|, |b| <= 2^60 (the guard range of the kernels: no intermediate can over- / underflow), plus operands one ulp
// around each other and around powers of two (the hard cases of division)
__global__ void check_div(unsigned long long* out, uint32_t rounds)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long bad = 0, n = 0;
    for (uint32_t i = 0; i < rounds; i++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5; uint32_t ua = s;
        s ^= s << 13; s ^= s >> 17; s ^= s << 5; uint32_t ub = s;
        // exponent into [67, 187] (2^-60 .. 2^60), any mantissa, any sign
        ua = (ua & 0x807fffffu) | ((67u + ((ua >> 23) & 0xffu) % 121u) << 23);
        ub = (ub & 0x807fffffu) | ((67u + ((ub >> 23) & 0xffu) % 121u) << 23);
        if ((i & 7u) == 7u) ub = (ub & 0xff800000u) | ((ua + (i >> 3) % 5u - 2u) & 0x007fffffu);     // nearly equal mantissas
        if ((i & 15u) == 3u) ub &= 0xff80000fu;                                                            // divisors next to a power of two
        const float a = __uint_as_float(ua), b = __uint_as_float(ub);
        n++;
        if (__float_as_uint(div_fast(a, b)) != __float_as_uint(a / b)) bad++;
    }
    atomicAdd(&out[3], n); atomicAdd(&out[4], bad);
}
__global__ void check(unsigned long long* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    unsigned long long bad1 = 0, bad2 = 0, inrange = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += nthreads) {
        const float x = __uint_as_float((uint32_t)b);
        const float ax = __builtin_fabsf(x);
        if (!(ax >= 1.1754943508e-38f && ax <= 8.5070591730e37f)) continue;      // guard: 2^-126 <= |x| <= 2^126 (normal input, normal result)
        inrange++;
        const float want = 1.0f / x;
        if (__float_as_uint(rcp_nr1(x)) != __float_as_uint(want)) bad1++;
        if (__float_as_uint(rcp_nr2(x)) != __float_as_uint(want)) bad2++;
    }
    atomicAdd(&out[0], inrange); atomicAdd(&out[1], bad1); atomicAdd(&out[2], bad2);
}
int main()
{
    unsigned long long* d; unsigned long long h[5] = {0, 0, 0, 0, 0};
    (void)hipMalloc(&d, 40); (void)hipMemset(d, 0, 40);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(check_div, dim3(8192), dim3(256), 0, 0, d, 32768u);
    (void)hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("inputs in guard range: %llu; mismatches vs IEEE 1/x: rcp + 1 Newton step %llu, rcp + 2 Newton steps %llu\n", h[0], h[1], h[2]);
    printf("a / b via rcp + Newton + one fma correction: %llu of %llu pseudo-random quotients differ from IEEE a / b\n", h[4], h[3]);
    return 0;
}
