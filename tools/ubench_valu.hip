// Micro-benchmark: issue rate of scalar vs packed f32 FMA, f64 FMA, min/max/med3 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, da = a, db = b;
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                         x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
        if (MODE == 1) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb); }
        if (MODE == 2) { d0 = __builtin_fma(d0, da, db); d1 = __builtin_fma(d1, da, db); d2 = __builtin_fma(d2, da, db); d3 = __builtin_fma(d3, da, db); }
        if (MODE == 3) { x0 = fminf(x0, x1); x1 = fmaxf(x1, x2); x2 = fminf(x2, x3); x3 = fmaxf(x3, x4); x4 = fminf(x4, x5); x5 = fmaxf(x5, x6); x6 = fminf(x6, x7); x7 = fmaxf(x7, x0 + a); }
        if (MODE == 4) { x0 = __builtin_amdgcn_fmed3f(x0, x1, x2); x1 = __builtin_amdgcn_fmed3f(x1, x2, x3); x2 = __builtin_amdgcn_fmed3f(x2, x3, x4); x3 = __builtin_amdgcn_fmed3f(x3, x4, x5);
                         x4 = __builtin_amdgcn_fmed3f(x4, x5, x6); x5 = __builtin_amdgcn_fmed3f(x5, x6, x7); x6 = __builtin_amdgcn_fmed3f(x6, x7, x0); x7 = __builtin_amdgcn_fmed3f(x7, x0, x1) + a; }
        if (MODE == 5) { x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b; }
        if (MODE == 6) { p0 = p0 * pa; p1 = p1 + pb; p2 = p2 * pa; p3 = p3 + pb; }
        if (MODE == 7) { d0 = d0 * da; d1 = d1 + db; d2 = d2 * da; d3 = d3 + db; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
}
template <int MODE> void run(const char* name, int lane_ops_per_iter, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS * lane_ops_per_iter;
    printf("%-28s %8.3f ms  %7.2f T lane-results/s\n", name, ms, ops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32 x8", 8, d); run<1>("v_pk_fma_f32 x4 (8 results)", 8, d); run<2>("v_fma_f64 x4", 4, d);
    run<3>("v_min/max_f32 x8", 8, d); run<4>("v_med3_f32 x8", 8, d); run<5>("v_mul/add_f32 x8", 8, d);
    run<6>("v_pk_mul/add_f32 x4 (8 res)", 8, d); run<7>("v_mul/add_f64 x4", 4, d);
    return 0;
}
