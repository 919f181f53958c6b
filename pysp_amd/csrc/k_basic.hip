// k_basic.hip -- Bayer plane helpers, the stand-alone homogeneity vote and the pointwise colour ops.
#include "devmath.h"
#include "kernels.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

// ---- bayer_chan_mixer.py:4-42 / normalization.py:4-24 -------------------------------------------
// One thread per 2x2 quad: two 8-byte row reads, four 4-byte plane writes (coalesced per plane).
template <typename T>
__global__ void k_demux(const T* __restrict__ bayer, int h, int w, float* __restrict__ r, float* __restrict__ g1,
                        float* __restrict__ b, float* __restrict__ g2) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    size_t W = 2 * (size_t)w;
    const T* top = bayer + (size_t)(2 * i) * W + 2 * j;
    const T* bot = top + W;
    size_t o = (size_t)i * w + j;
    r[o] = (float)top[0]; g1[o] = (float)top[1]; g2[o] = (float)bot[0]; b[o] = (float)bot[1];
}
__global__ void k_remux(const float* __restrict__ r, const float* __restrict__ g1, const float* __restrict__ b,
                        const float* __restrict__ g2, int h, int w, float* __restrict__ bayer) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    size_t W = 2 * (size_t)w, o = (size_t)i * w + j;
    float2 top = make_float2(r[o], g1[o]), bot = make_float2(g2[o], b[o]);
    *reinterpret_cast<float2*>(bayer + (size_t)(2 * i) * W + 2 * j) = top;
    *reinterpret_cast<float2*>(bayer + (size_t)(2 * i + 1) * W + 2 * j) = bot;
}
struct NormParams { float black[4], sat[4]; };
__global__ void k_normalize(const uint16_t* __restrict__ bayer, int H, int W, NormParams p, float* __restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int c = (y & 1) ? ((x & 1) ? 2 : 3) : ((x & 1) ? 1 : 0);
    float v = (float)bayer[(size_t)y * W + x] - p.black[c];
    v = v < 0.0f ? 0.0f : (v > p.sat[c] ? p.sat[c] : v);
    out[(size_t)y * W + x] = v / p.sat[c];
}

int launch_demux_f32(hipStream_t st, const float* bayer, int H, int W, float* r, float* g1, float* b, float* g2) {
    dim3 g((W / 2 + 255) / 256, H / 2);
    hipLaunchKernelGGL(k_demux<float>, g, dim3(256), 0, st, bayer, H / 2, W / 2, r, g1, b, g2);
    return CHECK_LAUNCH();
}
int launch_demux_u16(hipStream_t st, const uint16_t* bayer, int H, int W, float* r, float* g1, float* b, float* g2) {
    dim3 g((W / 2 + 255) / 256, H / 2);
    hipLaunchKernelGGL(k_demux<uint16_t>, g, dim3(256), 0, st, bayer, H / 2, W / 2, r, g1, b, g2);
    return CHECK_LAUNCH();
}
int launch_remux_f32(hipStream_t st, const float* r, const float* g1, const float* b, const float* g2, int h, int w, float* bayer) {
    dim3 g((w + 255) / 256, h);
    hipLaunchKernelGGL(k_remux, g, dim3(256), 0, st, r, g1, b, g2, h, w, bayer);
    return CHECK_LAUNCH();
}
int launch_normalize_u16(hipStream_t st, const uint16_t* bayer, int H, int W, const float black[4], const float sat[4], float* out) {
    NormParams p;
    for (int i = 0; i < 4; i++) { p.black[i] = black[i]; p.sat[i] = sat[i]; }
    dim3 g((W + 255) / 256, H);
    hipLaunchKernelGGL(k_normalize, g, dim3(256), 0, st, bayer, H, W, p, out);
    return CHECK_LAUNCH();
}

// ---- debayer/ahd_homogeneity_cython.pyx:22-68, stand-alone (drop-in for the Cython unit) -----------
// lab is interleaved (Hp,Wp,3) and already padded.  Tile 64x16 outputs; the padded Lab tile is staged
// in LDS as three planes so that the 3x3 window reads are conflict-free row reads.
namespace {
constexpr int BMX = 64, BMY = 16;
}
__global__ void __launch_bounds__(256) k_build_map(const float* __restrict__ lab, int Hp, int Wp, int kp, int vertical,
                                                   float* __restrict__ out) {
    extern __shared__ float sm[];
    const int dk = 2 * kp + 1, tw = BMX + 2 * kp, th = BMY + 2 * kp;
    float* sL = sm; float* sA = sm + th * tw; float* sB = sm + 2 * th * tw;
    const int rx = Wp - 2 * kp, ry = Hp - 2 * kp;
    const int x0 = blockIdx.x * BMX, y0 = blockIdx.y * BMY;
    for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
        int ly = idx / tw, lx = idx - ly * tw;
        int gy = y0 + ly, gx = x0 + lx;
        float L = 0, A = 0, B = 0;
        if (gy < Hp && gx < Wp) { const float* s = lab + ((size_t)gy * Wp + gx) * 3; L = s[0]; A = s[1]; B = s[2]; }
        sL[idx] = L; sA[idx] = A; sB[idx] = B;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BMX * BMY; idx += 256) {
        int ly = idx / BMX, lx = idx - ly * BMX;
        int x = x0 + lx, y = y0 + ly;
        if (x >= rx || y >= ry) continue;
        int c = (ly + kp) * tw + lx + kp;
        int n1 = vertical ? c - tw : c - 1, n2 = vertical ? c + tw : c + 1;
        float rl = sL[c], ra = sA[c], rb = sB[c];
        float e1 = fabsf(rl - sL[n1]), e2 = fabsf(rl - sL[n2]);
        float da1 = ra - sA[n1], db1 = rb - sB[n1], da2 = ra - sA[n2], db2 = rb - sB[n2];
        float c1 = da1 * da1 + db1 * db1, c2 = da2 * da2 + db2 * db2;
        float el = e2 > e1 ? e2 : e1, ec = c2 > c1 ? c2 : c1;
        float cnt = 0.0f;
        for (int wy = 0; wy < dk; wy++)
            for (int wx = 0; wx < dk; wx++) {
                int o = (ly + wy) * tw + lx + wx;
                float da = sA[o] - ra, db = sB[o] - rb;
                if (sL[o] - rl <= el && da * da + db * db <= ec) cnt = cnt + 1.0f;
            }
        out[(size_t)y * rx + x] = cnt;
    }
}
int launch_build_map(hipStream_t st, const float* lab, int Hp, int Wp, int k_pad, int is_vertical, float* out) {
    int rx = Wp - 2 * k_pad, ry = Hp - 2 * k_pad;
    size_t shm = (size_t)3 * (BMX + 2 * k_pad) * (BMY + 2 * k_pad) * sizeof(float);
    if (shm > 64 * 1024) return -1;
    dim3 g((rx + BMX - 1) / BMX, (ry + BMY - 1) / BMY);
    hipLaunchKernelGGL(k_build_map, g, dim3(256), shm, st, lab, Hp, Wp, k_pad, is_vertical, out);
    return CHECK_LAUNCH();
}

// ---- pre-demosaic cleanup ----------------------------------------------------
// raw_bad_pixel_corr.py:30-65 find_erroneous_pixels_threshold: one thread per mosaic pixel; the eight
// same-colour neighbours sit 2 px away; np.pad(mode="reflect") = REFLECT_101 on the quarter plane.
__global__ void __launch_bounds__(256) k_hot_threshold(const float* __restrict__ bayer, int H, int W, float min_delta, int min_count,
                                                       uint8_t* mr, uint8_t* mg1, uint8_t* mb, uint8_t* mg2) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int h = H >> 1, w = W >> 1, i = y >> 1, j = x >> 1, oy = y & 1, ox = x & 1;
    float c = bayer[(size_t)y * W + x] - min_delta;
    int cnt = 0;
#pragma unroll
    for (int di = -1; di <= 1; di++)
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
            if (!di && !dj) continue;
            int ii = b_101(i + di, h), jj = b_101(j + dj, w);
            cnt += c > bayer[(size_t)(2 * ii + oy) * W + 2 * jj + ox] ? 1 : 0;
        }
    uint8_t* m = oy ? (ox ? mb : mg2) : (ox ? mg1 : mr);
    m[(size_t)i * w + j] = cnt > min_count ? 1 : 0;
}
int launch_hot_threshold(hipStream_t st, const float* bayer, int H, int W, float min_delta, int min_count, uint8_t* mr, uint8_t* mg1,
                         uint8_t* mb, uint8_t* mg2) {
    dim3 g((W + 255) / 256, H);
    hipLaunchKernelGGL(k_hot_threshold, g, dim3(256), 0, st, bayer, H, W, min_delta, min_count, mr, mg1, mb, mg2);
    return CHECK_LAUNCH();
}

// raw_correction.py:25-62 flat_frame_correction.  Pass 1: out = (x * mean_c) / flat, per-plane maximum of the
// finite values (as an order-preserving unsigned key) and a count of non-infinite values.  Pass 2: per plane,
// all-infinite -> leave the image alone; +inf -> that maximum; negative -> 0; optional clamp at 1.
struct FlatMeans { float m[4]; };
DEVI unsigned fkey(float f) { unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
DEVI float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ void __launch_bounds__(256) k_flat_pass1(const float* __restrict__ bayer, const float* __restrict__ flat, int H, int W,
                                                    FlatMeans mean, float* __restrict__ out, unsigned* stats /* [4][2]: max key, #non-inf */) {
    __shared__ unsigned smax[2], scnt[2];
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (threadIdx.x < 2) { smax[threadIdx.x] = 0u; scnt[threadIdx.x] = 0u; }
    __syncthreads();
    if (x < W) {
        int site = (y & 1) ? ((x & 1) ? 2 : 3) : ((x & 1) ? 1 : 0);
        size_t o = (size_t)y * W + x;
        float v = (bayer[o] * mean.m[site]) / flat[o];
        out[o] = v;
        if (!isinf(v)) atomicAdd(&scnt[x & 1], 1u);
        if (isfinite(v)) atomicMax(&smax[x & 1], fkey(v));
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        int site = (y & 1) ? (threadIdx.x ? 2 : 3) : (threadIdx.x ? 1 : 0);
        if (scnt[threadIdx.x]) atomicAdd(&stats[site * 2 + 1], scnt[threadIdx.x]);
        if (smax[threadIdx.x]) atomicMax(&stats[site * 2], smax[threadIdx.x]);
    }
}
__global__ void __launch_bounds__(256) k_flat_pass2(const float* __restrict__ bayer, int H, int W, int clamp_high, float* __restrict__ out,
                                                    const unsigned* __restrict__ stats) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int site = (y & 1) ? ((x & 1) ? 2 : 3) : ((x & 1) ? 1 : 0);
    size_t o = (size_t)y * W + x;
    if (stats[site * 2 + 1] == 0u) { out[o] = bayer[o]; return; }          // np.isinf(output).all()
    float v = out[o];
    if (v == INFINITY) v = stats[site * 2] ? fkey_inv(stats[site * 2]) : v;
    if (v < 0.0f) v = 0.0f;
    if (clamp_high && v > 1.0f) v = 1.0f;
    out[o] = v;
}
int launch_flat_field(hipStream_t st, const float* bayer, const float* flat, int H, int W, const float mean[4], int clamp_high, float* out,
                      unsigned* d_stats8) {
    if (hipMemsetAsync(d_stats8, 0, 8 * sizeof(unsigned), st) != hipSuccess) return -3;
    FlatMeans m; for (int i = 0; i < 4; i++) m.m[i] = mean[i];
    dim3 g((W + 255) / 256, H);
    hipLaunchKernelGGL(k_flat_pass1, g, dim3(256), 0, st, bayer, flat, H, W, m, out, d_stats8);
    hipLaunchKernelGGL(k_flat_pass2, g, dim3(256), 0, st, bayer, H, W, clamp_high, out, d_stats8);
    return CHECK_LAUNCH();
}

// ---- pointwise colour ---------------------------------------------------------------------------
// Four RGB pixels (three float4) per thread: 48 contiguous bytes in, 48 out.
template <typename F>
__global__ void __launch_bounds__(256) k_rgb_pointwise(const float* __restrict__ in, size_t npx, float* __restrict__ out, F f) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 pixels
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t ngroups = npx / 4;
    for (; q < ngroups; q += stride) {
        const float4* s = reinterpret_cast<const float4*>(in) + 3 * q;
        float4 a = s[0], b = s[1], c = s[2];
        f(a.x, a.y, a.z); f(a.w, b.x, b.y); f(b.z, b.w, c.x); f(c.y, c.z, c.w);
        float4* d = reinterpret_cast<float4*>(out) + 3 * q;
        d[0] = a; d[1] = b; d[2] = c;
    }
    // tail pixels (npx % 4)
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t base = ngroups * 4;
    if (t < npx - base) {
        float r = in[(base + t) * 3], g = in[(base + t) * 3 + 1], b = in[(base + t) * 3 + 2];
        f(r, g, b);
        out[(base + t) * 3] = r; out[(base + t) * 3 + 1] = g; out[(base + t) * 3 + 2] = b;
    }
}
struct FCcm { Ccm m; int clip; __device__ void operator()(float& r, float& g, float& b) const {
    float cr = clip ? clip01_np(r) : r, cg = clip ? clip01_np(g) : g, cb = clip ? clip01_np(b) : b;   // np.clip keeps a NaN (transform.py:6-19)
    r = ccm_row(m.m, cr, cg, cb); g = ccm_row(m.m + 3, cr, cg, cb); b = ccm_row(m.m + 6, cr, cg, cb); } };
struct FTail { Ccm m; int tail; __device__ void operator()(float& r, float& g, float& b) const { colour_tail(tail, m.m, r, g, b); } };
struct FEnc { __device__ void operator()(float& r, float& g, float& b) const { r = srgb_encode(r); g = srgb_encode(g); b = srgb_encode(b); } };
struct FDec { __device__ void operator()(float& r, float& g, float& b) const { r = srgb_decode(r); g = srgb_decode(g); b = srgb_decode(b); } };
// image_base.py:45-60: wb_apply = (image*coeff) float32 ; wb_undo = float32(float64(image)/coeff)
struct FWb { float c[3]; int undo; __device__ void operator()(float& r, float& g, float& b) const {
    if (undo) { r = (float)((double)r / (double)c[0]); g = (float)((double)g / (double)c[1]); b = (float)((double)b / (double)c[2]); }
    else { r = r * c[0]; g = g * c[1]; b = b * c[2]; } } };

template <typename F>
static int launch_pointwise(hipStream_t st, const float* in, size_t npx, float* out, F f) {
    size_t groups = npx / 4 + 1;
    unsigned grid = (unsigned)((groups + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return -1;
    hipLaunchKernelGGL(k_rgb_pointwise<F>, dim3(grid), dim3(256), 0, st, in, npx, out, f);
    return CHECK_LAUNCH();
}
int launch_cam_to_rgb(hipStream_t st, const float* in, size_t npx, const double M[9], int clip, float* out) {
    FCcm f; for (int i = 0; i < 9; i++) f.m.m[i] = M[i]; f.clip = clip;
    return launch_pointwise(st, in, npx, out, f);
}
int launch_colour_tail(hipStream_t st, const float* in, size_t npx, const double M[9], int tail, float* out) {
    FTail f; for (int i = 0; i < 9; i++) f.m.m[i] = M[i]; f.tail = tail;
    return launch_pointwise(st, in, npx, out, f);
}
int launch_wb_scale(hipStream_t st, const float* in, size_t npx, const float coeff[3], int undo, float* out) {
    FWb f; for (int i = 0; i < 3; i++) f.c[i] = coeff[i]; f.undo = undo;
    return launch_pointwise(st, in, npx, out, f);
}
// n = number of floats (any count): treated as ceil(n/3) "pixels" would misalign, so use a flat kernel
__global__ void __launch_bounds__(256) k_gamma_flat(const float* __restrict__ in, size_t n, int decode, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t n4 = n / 4;
    for (size_t q = i; q < n4; q += stride) {
        float4 v = reinterpret_cast<const float4*>(in)[q];
        if (decode) { v.x = srgb_decode(v.x); v.y = srgb_decode(v.y); v.z = srgb_decode(v.z); v.w = srgb_decode(v.w); }
        else { v.x = srgb_encode(v.x); v.y = srgb_encode(v.y); v.z = srgb_encode(v.z); v.w = srgb_encode(v.w); }
        reinterpret_cast<float4*>(out)[q] = v;
    }
    if (i < n - n4 * 4) { float v = in[n4 * 4 + i]; out[n4 * 4 + i] = decode ? srgb_decode(v) : srgb_encode(v); }
}
int launch_gamma(hipStream_t st, const float* in, size_t n, int decode, float* out) {
    size_t groups = n / 4 + 1;
    unsigned grid = (unsigned)((groups + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return -1;
    hipLaunchKernelGGL(k_gamma_flat, dim3(grid), dim3(256), 0, st, in, n, decode, out);
    return CHECK_LAUNCH();
}

// Device -> page-locked host memory by a kernel instead of a DMA engine (api.cpp, banded host pipeline, PYSP_D2H_KERNEL=1): `dst` is the device-visible address of
// a hipHostMalloc block, written with 16-byte stores over PCIe.  n16 = number of 16-byte pieces.
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += stride) dst[q] = src[q];
}
int launch_copy16(hipStream_t st, void* dst, const void* src, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) || (bytes & 15)) return -1;
    const size_t n16 = bytes / 16;
    unsigned grid = (unsigned)((n16 + 255) / 256);
    if (grid > 512) grid = 512;              // a few waves per CU are enough to keep the link busy; the compute kernels of the next band share the chip
    if (grid == 0) return 0;
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
    return CHECK_LAUNCH();
}
