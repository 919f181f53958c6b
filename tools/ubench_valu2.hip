// Micro-benchmark 2: per-opcode VALU issue rate on gfx950 with 8 fully independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int n) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    double d[4] = {x[0], x[1], x[2], x[3]};
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) x[i] = fminf(x[i], a) ;
            if (MODE == 2) x[i] = fmaxf(fminf(x[i], a), b);                 // 2 ops (or 1 med3 if clamped)
            if (MODE == 3) x[i] = __builtin_amdgcn_fmed3f(x[i], a, b);
            if (MODE == 4) x[i] = fminf(fminf(x[i], a), b);                 // v_min3
            if (MODE == 5) x[i] = x[i] > a ? x[i] : b;                      // cmp + cndmask
            if (MODE == 6) x[i] = x[i] * a;
            if (MODE == 7) x[i] = x[i] + a;
            if (MODE == 8) x[i] = (float)(int)(x[i] * a);                   // cvt round trip + mul
            if (MODE == 9) x[i] = __builtin_amdgcn_exp2f(x[i]);
        }
        if (MODE == 10) {
#pragma unroll
            for (int i = 0; i < 4; i++) d[i] = __builtin_fma(d[i], (double)a, (double)b);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(d[0] + d[1] + d[2] + d[3]);
}
template <int MODE> void run(const char* name, int ops, float* dbuf) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(dbuf, 0.999f, 0.5f, ITERS);
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(dbuf, 0.999f, 0.5f, ITERS); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double waveinstr = (double)blocks * 4 * ITERS * ops;   // 4 waves per block
    double cyc = ms * 1e-3 * 2.4e9 * 1024;                 // SIMD-cycles available at 2.4 GHz
    printf("%-34s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, ms, cyc / waveinstr);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 8, d); run<1>("v_min_f32", 8, d); run<2>("min+max (2 instr)", 16, d); run<3>("v_med3_f32", 8, d);
    run<4>("v_min3_f32", 8, d); run<5>("v_cmp+v_cndmask (2 instr)", 16, d); run<6>("v_mul_f32", 8, d); run<7>("v_add_f32", 8, d);
    run<8>("mul+cvt_i32+cvt_f32 (3 instr)", 24, d); run<9>("v_exp_f32", 8, d); run<10>("v_fma_f64 x4", 4, d);
    return 0;
}
