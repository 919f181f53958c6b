// What the MI355X memory system sustains for the byte mixes of this library's streaming kernels, with hand-written gfx950 kernels
// (the round-2 probe timed torch's copy kernels).  All accesses are 16 bytes per lane, consecutive lanes on
// consecutive addresses.
//
//   hipcc -O3 --offload-arch=gfx950 -o ubench_stream.bin ubench_stream.hip && ./ubench_stream.bin > profiles/r3_ubench_stream.log
//
//   copy      N bytes -> N bytes                               (288 MB and 1 GB)
//   fill      write only, read only (sum)
//   mix13     EAG's mix: 96 MB read, 288 MB written -- one float4 of mosaic -> three float4 of RGB (linear order)
//   mix13t    the same bytes in k_eag's access pattern: 64x32-px tiles in the XCD-aware order, a tile's 32 mosaic rows of 256 B
//             read, its 32 RGB rows of 768 B written, one 512-thread workgroup per tile
//   mix11t    median stage's mix in its tile pattern: 60x28 px RGB tiles + halo 4 read (68x36), 60x28 written
// Each in two launch forms: one workgroup per piece of work ("grid") and 256 x k persistent workgroups striding over it ("pers").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
template <int U>
__global__ void __launch_bounds__(256) k_copy_u(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    // U loads in flight per lane before the stores
    for (size_t i0 = (size_t)blockIdx.x * 256 * U; i0 < n; i0 += (size_t)gridDim.x * 256 * U) {
        float4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = i0 + u * 256 + threadIdx.x; t[u] = i < n ? a[i] : float4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = i0 + u * 256 + threadIdx.x; if (i < n) b[i] = t[u]; }
    }
}
__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = float4{1, 2, 3, 4};
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* out, size_t n) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1.2345e-30f) out[0] = s;
}
template <int U>
__global__ void __launch_bounds__(256) k_mix13(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i0 = (size_t)blockIdx.x * 256 * U; i0 < n; i0 += (size_t)gridDim.x * 256 * U) {
        float4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = i0 + u * 256 + threadIdx.x; t[u] = i < n ? a[i] : float4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            // the wave's 1 KiB of input becomes 3 KiB of output: lane l writes pieces l, l + 64, l + 128 of it
            size_t w0 = (i0 + u * 256 + (threadIdx.x & ~63)) * 3 + (threadIdx.x & 63);
            if (i0 + u * 256 + threadIdx.x < n) { b[w0] = t[u]; b[w0 + 64] = t[u]; b[w0 + 128] = t[u]; }
        }
    }
}
__device__ __forceinline__ void xcd_tile(unsigned lin, unsigned n, unsigned gx, int& bx, int& by) {
    const unsigned xcd = lin & 7u, q = n >> 3, r = n & 7u;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
    by = (int)(t / gx); bx = (int)(t - (unsigned)by * gx);
}
// EAG pattern: tile 64 x 32 px; mosaic rows of 64 floats = 16 float4; RGB rows of 192 floats = 48 float4
template <bool SWZ>
__global__ void __launch_bounds__(512) k_mix13t(const float* __restrict__ a, float* __restrict__ b, int H, int W, int ntx, int nty) {
    const unsigned n = ntx * nty;
    for (unsigned lin = blockIdx.x; lin < n; lin += gridDim.x) {
        int bx, by;
        if (SWZ) xcd_tile(lin, n, ntx, bx, by); else { by = lin / ntx; bx = lin - by * ntx; }
        const int x0 = bx * 64, y0 = by * 32;
        const int t = threadIdx.x;
        float4 v = *reinterpret_cast<const float4*>(a + (size_t)(y0 + (t >> 4)) * W + x0 + 4 * (t & 15));   // 32 rows x 16 pieces = 512
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int idx = t + k * 512, row = idx / 48, c = idx - row * 48;
            *reinterpret_cast<float4*>(b + ((size_t)(y0 + row) * W + x0) * 3 + 4 * c) = v;
        }
    }
}
// median-stage pattern: RGB in, RGB out; tile 60 x 28 px, read 68 x 36 (halo 4); rows of 204 floats = 51 float4 (rows start 16-B aligned when W % 4 == 0)
template <bool SWZ>
__global__ void __launch_bounds__(256) k_mix11t(const float* __restrict__ a, float* __restrict__ b, int H, int W, int ntx, int nty) {
    const unsigned n = ntx * nty;
    for (unsigned lin = blockIdx.x; lin < n; lin += gridDim.x) {
        int bx, by;
        if (SWZ) xcd_tile(lin, n, ntx, bx, by); else { by = lin / ntx; bx = lin - by * ntx; }
        const int x0 = bx * 60, y0 = by * 28;
        if (x0 < 4 || y0 < 4 || x0 + 64 > W || y0 + 32 > H) continue;
        float4 acc = {0, 0, 0, 0};
        for (int idx = threadIdx.x; idx < 36 * 51; idx += 256) {
            const int row = idx / 51, c = idx - row * 51;
            float4 v = *reinterpret_cast<const float4*>(a + ((size_t)(y0 - 4 + row) * W + x0 - 4) * 3 + 4 * c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        for (int idx = threadIdx.x; idx < 28 * 45; idx += 256) {
            const int row = idx / 45, c = idx - row * 45;
            *reinterpret_cast<float4*>(b + ((size_t)(y0 + row) * W + x0) * 3 + 4 * c) = acc;
        }
    }
}

template <class F>
static double time_ms(F launch, int reps = 20) {
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    HC(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    HC(hipGetLastError());
    return ms / reps;
}
static void line(const char* name, double bytes, double ms) { printf("%-64s %8.4f ms  %7.1f GB/s  (%.1f %% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0); fflush(stdout); }

int main() {
    const size_t GB = 1ull << 30;
    float *a, *b;
    HC(hipMalloc(&a, GB)); HC(hipMalloc(&b, GB));
    HC(hipMemset(a, 0x3c, GB)); HC(hipMemset(b, 0, GB));
    const size_t n288 = 288000000ull / 16, n1g = GB / 16, n96 = 96000000ull / 16;
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4*)a, (float4*)b, n1g);
    HC(hipDeviceSynchronize());
    char nm[128];
    for (int k : {4, 8, 16, 32}) {
        snprintf(nm, sizeof nm, "copy 288 MB -> 288 MB, persistent 256 x %d workgroups", k);
        line(nm, 2 * 288e6, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * k), dim3(256), 0, 0, (const float4*)a, (float4*)b, n288); }));
    }
    line("copy 288 MB -> 288 MB, one float4 per thread (grid)", 2 * 288e6, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3((n288 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n288); }));
    line("copy 288 MB -> 288 MB, 4 float4 in flight per thread (grid)", 2 * 288e6, time_ms([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3((n288 + 1023) / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, n288); }));
    line("copy 288 MB -> 288 MB, 4 in flight, persistent 256 x 8", 2 * 288e6, time_ms([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n288); }));
    line("copy 1 GiB -> 1 GiB, persistent 256 x 8", 2.0 * GB, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n1g); }));
    line("copy 1 GiB -> 1 GiB, 4 in flight (grid)", 2.0 * GB, time_ms([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3((n1g + 1023) / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, n1g); }));
    line("copy 96 MB -> 96 MB (fits the 256 MB Infinity Cache), grid", 2 * 96e6, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3((n96 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n96); }));
    line("fill 288 MB (write only), grid", 288e6, time_ms([&] { hipLaunchKernelGGL(k_fill, dim3((n288 + 255) / 256), dim3(256), 0, 0, (float4*)b, n288); }));
    line("fill 1 GiB (write only), persistent 256 x 8", (double)GB, time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (float4*)b, n1g); }));
    line("read 288 MB (read only), grid", 288e6, time_ms([&] { hipLaunchKernelGGL(k_read, dim3((n288 + 255) / 256), dim3(256), 0, 0, (const float4*)a, b, n288); }));
    line("read 1 GiB (read only), persistent 256 x 8", (double)GB, time_ms([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const float4*)a, b, n1g); }));
    line("mix13 96 MB read / 288 MB written, linear, 1 in flight (grid)", 384e6, time_ms([&] { hipLaunchKernelGGL(k_mix13<1>, dim3((n96 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n96); }));
    line("mix13 96 MB read / 288 MB written, linear, 4 in flight (grid)", 384e6, time_ms([&] { hipLaunchKernelGGL(k_mix13<4>, dim3((n96 + 1023) / 1024), dim3(256), 0, 0, (const float4*)a, (float4*)b, n96); }));
    line("mix13 linear, 4 in flight, persistent 256 x 8", 384e6, time_ms([&] { hipLaunchKernelGGL(k_mix13<4>, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n96); }));
    const int H = 4000, W = 6000;
    {
        const int ntx = (W + 63) / 64, nty = H / 32;    // 6000 / 64 = 93.75: the last tile column is left out (same bytes within 0.3 %)
        const int ntxf = W / 64;
        const double bytes = (double)ntxf * nty * 64 * 32 * 16;
        (void)ntx;
        line("mix13t k_eag tile pattern (64x32 px, 512 thr), XCD order, grid", bytes, time_ms([&] { hipLaunchKernelGGL(k_mix13t<true>, dim3(ntxf * nty), dim3(512), 0, 0, a, b, H, W, ntxf, nty); }));
        line("mix13t k_eag tile pattern, row-major tile order, grid", bytes, time_ms([&] { hipLaunchKernelGGL(k_mix13t<false>, dim3(ntxf * nty), dim3(512), 0, 0, a, b, H, W, ntxf, nty); }));
        line("mix13t k_eag tile pattern, XCD order, persistent 256 x 4", bytes, time_ms([&] { hipLaunchKernelGGL(k_mix13t<true>, dim3(1024), dim3(512), 0, 0, a, b, H, W, ntxf, nty); }));
    }
    {
        const int ntx = W / 60, nty = (H + 27) / 28;
        const double tiles = (double)(ntx - 2) * (nty - 2);   // border tiles are skipped
        const double bytes = tiles * (60 * 28 * 12 * 2);
        line("mix11t median-stage tile pattern (60x28 px + halo 4), XCD order", bytes, time_ms([&] { hipLaunchKernelGGL(k_mix11t<true>, dim3(ntx * nty), dim3(256), 0, 0, a, b, H, W, ntx, nty); }));
        line("mix11t median-stage tile pattern, row-major tile order", bytes, time_ms([&] { hipLaunchKernelGGL(k_mix11t<false>, dim3(ntx * nty), dim3(256), 0, 0, a, b, H, W, ntx, nty); }));
    }
    return 0;
}
