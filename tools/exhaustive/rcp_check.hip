// Exhaustive check (all 2^32 f32 bit patterns) of candidate short reciprocal sequences against the correctly rounded
// IEEE division hipcc emits for 1.0f / x.  Run on the GPU box: prints, per candidate, the number of mismatching inputs
// and the exponent range they fall in.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off rcp_check.hip -o rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float cand1(float x) { float r = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float cand2(float x) { float r = cand1(x); float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ void k(unsigned long long* bad, unsigned* emin, unsigned* emax, unsigned* sample) {
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float ref = 1.0f / x;
        const unsigned ex = ((unsigned)i >> 23) & 0xffu;
        const float c[2] = { cand1(x), cand2(x) };
        for (int j = 0; j < 2; j++) if (!same(c[j], ref)) {
            unsigned long long old = atomicAdd(&bad[j], 1ull);
            atomicMin(&emin[j], ex); atomicMax(&emax[j], ex);
            if (old < 8) sample[j * 8 + old] = (unsigned)i;
            // mismatches with the exponent strictly inside [2, 252]
            if (ex >= 2 && ex <= 252) atomicAdd(&bad[2 + j], 1ull);
        }
    }
}
int main() {
    unsigned long long* bad; unsigned *emin, *emax, *sample;
    hipMalloc(&bad, 32); hipMalloc(&emin, 8); hipMalloc(&emax, 8); hipMalloc(&sample, 64);
    hipMemset(bad, 0, 32); hipMemset(emin, 0xff, 8); hipMemset(emax, 0, 8); hipMemset(sample, 0, 64);
    k<<<4096, 256>>>(bad, emin, emax, sample);
    hipDeviceSynchronize();
    unsigned long long hb[4]; unsigned hmin[2], hmax[2], hs[16];
    hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(hmin, emin, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hmax, emax, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 64, hipMemcpyDeviceToHost);
    for (int j = 0; j < 2; j++) {
        printf("candidate %d (%d Newton steps): %llu mismatches of 2^32, biased exponents %u..%u, %llu with exponent in [2,252]; samples:", j + 1, j + 1, hb[j], hmin[j], hmax[j], hb[2 + j]);
        for (int q = 0; q < 8 && q < (int)hb[j]; q++) printf(" %08x", hs[j * 8 + q]);
        printf("\n");
    }
    return 0;
}
