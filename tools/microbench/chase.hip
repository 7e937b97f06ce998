// chase.hip — dependent 64-byte record fetches (4 x global_load_dwordx4, the next record's index comes out of the record) by ONE lane per wavefront:
// the memory round trip a lone traversal step pays on MI355X, for a working set that lives in L2 (190 KB, the bunny's node pairs), in MALL (64 MB) and
// in HBM (1 GB), with 1 .. 8192 such wavefronts in flight.   Build: hipcc --offload-arch=gfx950 -O3 -o chase chase.hip ; run: ./chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
typedef float rec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void chase(const char* __restrict__ base, uint32_t nRec, uint32_t steps, unsigned long long* out, int useLds)
{
    extern __shared__ rec4 sm[];
    if (useLds) { for (uint32_t i = threadIdx.x; i < nRec * 4u; i += 64u) sm[i] = reinterpret_cast<const rec4*>(base)[i]; __syncthreads(); }
    if (threadIdx.x != 0) return;
    uint32_t idx = (blockIdx.x * 977u) % nRec; float acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t s = 0; s < steps; s++) {
        rec4 a, b, c, d;
        if (useLds) { a = sm[idx * 4u]; b = sm[idx * 4u + 1]; c = sm[idx * 4u + 2]; d = sm[idx * 4u + 3]; }
        else { const rec4* p = reinterpret_cast<const rec4*>(base + (size_t)idx * 64u); a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        acc += a.y + b.x + c.x + d.w;
        idx = __float_as_uint(a.x);
    }
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x] = t1 - t0; if (acc == 123.456f) out[0] = 0;
}
int main()
{
    for (size_t bytes : {(size_t)190 << 10, (size_t)32 << 10, (size_t)64 << 20, (size_t)1 << 30}) {
        const uint32_t n = (uint32_t)(bytes / 64);
        std::vector<uint32_t> perm(n); for (uint32_t i = 0; i < n; i++) perm[i] = i;
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<float> h((size_t)n * 16, 0.0f);
        for (uint32_t i = 0; i < n; i++) { const uint32_t nx = perm[(i + 1) % n]; memcpy(&h[(size_t)perm[i] * 16], &nx, 4); }      // one cycle through all records, in random order
        char* d; hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        unsigned long long* out; hipMalloc(&out, 8192 * 8);
        for (int useLds = 0; useLds < 2; useLds++) {
            if (useLds && bytes > (64u << 10)) continue;
            for (int waves : {1, 1024, 4096, 8192}) {
                const uint32_t steps = 20000;
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), useLds ? bytes : 0, 0, d, n, 100u, out, useLds);
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), useLds ? bytes : 0, 0, d, n, steps, out, useLds);
                std::vector<unsigned long long> t(waves); hipMemcpy(t.data(), out, waves * 8, hipMemcpyDeviceToHost);
                double s = 0; for (auto v : t) s += v;
                printf("%8zu KB %s  %5d waves of one lane: %.1f ns per dependent 64-B fetch\n", bytes >> 10, useLds ? "LDS copy" : "global  ", waves, s / waves / steps * 10.0);
            }
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}
