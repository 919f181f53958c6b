// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the kernels use:
// reads N bytes with 4 B/lane, 12 B/lane (3 dwords, stride 12) and 16 B/lane loads; writes with 12 B/lane stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void rd4(const float* p, size_t n, float* o) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float s = 0; for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 12345.f) o[0] = s; }
__global__ void rd12(const float* p, size_t n3, float* o) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float s = 0; for (; i < n3; i += (size_t)gridDim.x * blockDim.x) s += p[3*i] + p[3*i+1] + p[3*i+2]; if (s == 12345.f) o[0] = s; }
__global__ void rd16(const float4* p, size_t n4, float* o) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; float s = 0; for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; } if (s == 12345.f) o[0] = s; }
__global__ void wr12(float* p, size_t n3) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n3; i += (size_t)gridDim.x * blockDim.x) { p[3*i] = 1.f; p[3*i+1] = 2.f; p[3*i+2] = 3.f; } }
int main() {
    size_t n = (size_t)3 * 64 * 1024 * 1024;   // 768 MiB of floats (> Infinity Cache)
    float *a, *o; (void)hipMalloc(&a, n * 4); (void)hipMalloc(&o, 4); (void)hipMemset(a, 0, n * 4);
    rd4<<<4096, 256>>>(a, n, o); rd12<<<4096, 256>>>(a, n / 3, o); rd16<<<4096, 256>>>((const float4*)a, n / 4, o); wr12<<<4096, 256>>>(a, n / 3);
    (void)hipDeviceSynchronize(); printf("bytes per kernel: %zu\n", n * 4); return 0;
}
