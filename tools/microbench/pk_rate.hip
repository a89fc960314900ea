// Issue rate of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950 (is packed FP32 two results per lane and issue slot?)
//   mkdir -p _build && hipcc --offload-arch=gfx950 -O2 -o _build/pk_rate pk_rate.hip && _build/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f; const f2 pm = {m, m};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define S(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        } else if (MODE == 1) {
#define S(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pm));
            S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7) S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7)
#undef S
        } else if (MODE == 2) {
#define S(x) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x) : "v"(pm));
            S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7) S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7)
#undef S
        } else if (MODE == 3) {      // broadcast through op_sel: both halves of src1 from its low element
#define S(x) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(x) : "v"(pm));
            S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7) S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7)
#undef S
        } else {
#define S(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(m));
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE> float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int blocks = 256 * 8, iters = 20000; float* d; hipMalloc(&d, blocks * 256 * 4);
    const double insts = (double)blocks * 4 * iters * 16;      // wave instructions
    const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32 (neg)", "v_pk_mul_f32 op_sel broadcast", "v_max_f32"};
    float ms[5] = { run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters) };
    for (int i = 0; i < 5; i++) printf("%-32s %8.3f ms  %.2f cycles per wave instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", names[i], ms[i], ms[i] * 1e-3 * 2.4e9 * 1024 / insts);
    return 0;
}
