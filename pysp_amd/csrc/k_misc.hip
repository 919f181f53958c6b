// k_misc.hip -- HDR raw fusion (raw_hdr.py:85-158) and DNG WarpRectilinear
// (dng_warp_corr/dng_warp_rectilinear_coords.pyx + chan_distortion_corr.py:86-97) for gfx950.
#include <math.h>

#include "devmath.h"
#include "demosaic_common.h"
#include "kernels.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

// ---- HDR raw fusion -------------------------------------------------------------------------------
// Pointwise over K exposures: VEC px per thread per frame (float4 when W % 4 == 0, else float2: W is
// even, so every row start stays 8-byte aligned and column parity is known at compile time).
// More than MAXK exposures (round 4: the reference takes any number) run as passes of MAXK in order: a pass that is not the last leaves its partial
// sums in memory -- sum of weights in `part`, weighted sum in `out`, counts in `count` -- and the next one picks them up: the same float32 additions in the
// same order as one long loop, so the same bits.
namespace { constexpr int MAXK = 16; }
struct FuseParams {
    const float* frames[MAXK];
    float ev_off[MAXK];
    float bias[MAXK][4];   // CFA site order r,g1,b,g2
    int K, H, W;
    int first, last;       // this pass starts from zero / finishes the pixel
    const float* kmax_frame;   // the exposure with the largest EV offset (raw_hdr.py:143) and its offset: what a pixel without any weight falls back to
    float kmax_off;
    float* out;
    int32_t* count;
    float* part;           // (H,W) partial sums of weights between passes (only touched when K > MAXK)
};
template <int VEC>
__global__ void __launch_bounds__(256) k_fuse_raw(FuseParams p) {
    int xq = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xq * VEC >= p.W) return;
    // site index of (even, odd) columns in this row: even row -> r(0), g1(1); odd row -> g2(3), b(2)
    const int odd = y & 1, c_even = odd ? 3 : 0, c_odd = odd ? 2 : 1;
    size_t o = (size_t)y * p.W + (size_t)VEC * xq;
    float sw[VEC], sp[VEC], v[VEC];
    int cnt[VEC];
    if (p.first) {
#pragma unroll
        for (int t = 0; t < VEC; t++) { sw[t] = 0.0f; sp[t] = 0.0f; cnt[t] = 0; }
    } else {
#pragma unroll
        for (int t = 0; t < VEC; t++) { sw[t] = p.part[o + t]; sp[t] = p.out[o + t]; cnt[t] = p.count[o + t]; }
    }
    for (int k = 0; k < p.K; k++) {
        if (VEC == 4) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p.frames[k] + o);
        else *reinterpret_cast<float2*>(v) = *reinterpret_cast<const float2*>(p.frames[k] + o);
        const float off = p.ev_off[k];
#pragma unroll
        for (int t = 0; t < VEC; t++) {
            float wgt = (0.5f - fabsf(v[t] - 0.5f)) * p.bias[k][(t & 1) ? c_odd : c_even];   // raw_hdr.py:137
            sw[t] = sw[t] + wgt;                                                            // :138
            sp[t] = sp[t] + (v[t] * wgt) * off;                                             // :139
            cnt[t] += wgt > 0.0f ? 1 : 0;                                                   // :141
        }
    }
    if (!p.last) {
#pragma unroll
        for (int t = 0; t < VEC; t++) { p.part[o + t] = sw[t]; p.out[o + t] = sp[t]; p.count[o + t] = cnt[t]; }
        return;
    }
    if (VEC == 4) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p.kmax_frame + o);
    else *reinterpret_cast<float2*>(v) = *reinterpret_cast<const float2*>(p.kmax_frame + o);
    const float mo = p.kmax_off;
    float res[VEC];
#pragma unroll
    for (int t = 0; t < VEC; t++) res[t] = sw[t] == 0.0f ? v[t] * mo : sp[t] / sw[t];       // :144-148
    if (VEC == 4) {
        *reinterpret_cast<float4*>(p.out + o) = *reinterpret_cast<float4*>(res);
        *reinterpret_cast<int4*>(p.count + o) = *reinterpret_cast<int4*>(cnt);
    } else {
        *reinterpret_cast<float2*>(p.out + o) = *reinterpret_cast<float2*>(res);
        *reinterpret_cast<int2*>(p.count + o) = *reinterpret_cast<int2*>(cnt);
    }
}
int fuse_max_exposures_per_pass() { return MAXK; }
// one pass: n <= MAXK exposures (their offsets and biases), whether it starts / finishes the pixels, and for the finishing pass the largest-offset exposure
int launch_fuse_raw_pass(hipStream_t st, const float* const* d_frames, int n, int H, int W, const float* ev_off, const float* bias, int first, int last,
                         const float* d_kmax_frame, float kmax_off, float* d_out, int32_t* d_count, float* d_part) {
    if (n < 1 || n > MAXK || (W & 1) || (!(first && last) && !d_part) || (last && !d_kmax_frame)) return -1;
    uintptr_t align = reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_count) | reinterpret_cast<uintptr_t>(d_kmax_frame);
    FuseParams p;
    for (int k = 0; k < n; k++) {
        p.frames[k] = d_frames[k];
        p.ev_off[k] = ev_off[k];
        for (int c = 0; c < 4; c++) p.bias[k][c] = bias[k * 4 + c];
        align |= reinterpret_cast<uintptr_t>(d_frames[k]);
    }
    if (align & 15) return -1;
    p.K = n; p.H = H; p.W = W; p.out = d_out; p.count = d_count; p.part = d_part;
    p.first = first; p.last = last; p.kmax_frame = d_kmax_frame; p.kmax_off = kmax_off;
    if ((W & 3) == 0) {
        dim3 g((W / 4 + 255) / 256, H);
        hipLaunchKernelGGL(k_fuse_raw<4>, g, dim3(256), 0, st, p);
    } else {
        dim3 g((W / 2 + 255) / 256, H);
        hipLaunchKernelGGL(k_fuse_raw<2>, g, dim3(256), 0, st, p);
    }
    return CHECK_LAUNCH();
}
int launch_fuse_raw(hipStream_t st, const float* const* d_frames, int K, int H, int W, const float* ev_off, const float* bias,
                    int kmax, float* d_out, int32_t* d_count, float* d_part) {
    if (K < 1 || kmax < 0 || kmax >= K || (K > MAXK && !d_part)) return -1;
    for (int k0 = 0; k0 < K; k0 += MAXK) {
        const int n = K - k0 < MAXK ? K - k0 : MAXK;
        int rc = launch_fuse_raw_pass(st, d_frames + k0, n, H, W, ev_off + k0, bias + 4 * k0, k0 == 0, k0 + n == K, d_frames[kmax], ev_off[kmax], d_out, d_count, d_part);
        if (rc) return rc;
    }
    return 0;
}

// (more than MAXK exposures: passes as in the raw fusion; between passes the weight sums live in part[0 .. 3 npx), the weighted sums in `out`, the counts in
// `count` and the largest-offset exposure's white-balanced pixel in part[3 npx .. 6 npx))
struct FuseRgbParams {
    const float* frames[MAXK];
    float* frames_out[MAXK];
    float coeff[MAXK][3];
    float ev_off[MAXK], bias[MAXK];
    int applied[MAXK];
    int K, kmax, use_ccm;      // kmax: index inside THIS pass of the exposure with the largest offset, or -1
    int first, last;
    float kmax_off;
    size_t npx;
    Ccm ccm;
    float* out;
    int32_t* count;
    float* part;
};
__global__ void __launch_bounds__(256) k_fuse_rgb(FuseRgbParams p) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.npx) return;
    float sw[3] = {0, 0, 0}, sp[3] = {0, 0, 0}, vmax[3] = {0, 0, 0};
    int cnt[3] = {0, 0, 0};
    if (!p.first) {
#pragma unroll
        for (int c = 0; c < 3; c++) { sw[c] = p.part[3 * i + c]; vmax[c] = p.part[3 * p.npx + 3 * i + c]; sp[c] = p.out[3 * i + c]; cnt[c] = p.count[3 * i + c]; }
    }
    for (int k = 0; k < p.K; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float a = p.frames[k][3 * i + c], cf = p.coeff[k][c];
            float u = p.applied[k] ? (float)((double)a / (double)cf) : a;
            float w = (0.5f - fabsf(u - 0.5f)) * p.bias[k];
            sw[c] = sw[c] + w;
            float v = u * cf;
            sp[c] = sp[c] + (v * w) * p.ev_off[k];
            cnt[c] += w > 0.0f ? 1 : 0;
            if (k == p.kmax) vmax[c] = v;
            if (p.frames_out[k]) p.frames_out[k][3 * i + c] = v;
        }
    }
    if (!p.last) {
#pragma unroll
        for (int c = 0; c < 3; c++) { p.part[3 * i + c] = sw[c]; p.part[3 * p.npx + 3 * i + c] = vmax[c]; p.out[3 * i + c] = sp[c]; p.count[3 * i + c] = cnt[c]; }
        return;
    }
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        res[c] = sw[c] == 0.0f ? vmax[c] * p.kmax_off : sp[c] / sw[c];
        p.count[3 * i + c] = cnt[c];
    }
    if (p.use_ccm) {
        p.out[3 * i] = ccm_row(p.ccm.m, res[0], res[1], res[2]);
        p.out[3 * i + 1] = ccm_row(p.ccm.m + 3, res[0], res[1], res[2]);
        p.out[3 * i + 2] = ccm_row(p.ccm.m + 6, res[0], res[1], res[2]);
    } else {
        p.out[3 * i] = res[0]; p.out[3 * i + 1] = res[1]; p.out[3 * i + 2] = res[2];
    }
}
// one pass of the RGB fusion: n <= MAXK exposures; kmax_local = index inside the pass of the largest-offset exposure, or -1 when it is in another pass
int launch_fuse_rgb_pass(hipStream_t st, const float* const* d_frames, float* const* d_frames_out, int n, size_t npx, const float* coeff, const int* applied,
                         const float* ev_off, const float* bias, int first, int last, int kmax_local, float kmax_off, const double* M, float* d_out,
                         int32_t* d_count, float* d_part) {
    if (n < 1 || n > MAXK || kmax_local >= n || (!(first && last) && !d_part)) return -1;
    FuseRgbParams p;
    for (int k = 0; k < n; k++) {
        p.frames[k] = d_frames[k];
        p.frames_out[k] = d_frames_out ? d_frames_out[k] : nullptr;
        for (int c = 0; c < 3; c++) p.coeff[k][c] = coeff[k * 3 + c];
        p.ev_off[k] = ev_off[k]; p.bias[k] = bias[k]; p.applied[k] = applied[k];
    }
    p.K = n; p.kmax = kmax_local; p.kmax_off = kmax_off; p.first = first; p.last = last;
    p.npx = npx; p.out = d_out; p.count = d_count; p.part = d_part; p.use_ccm = M != nullptr;
    for (int i = 0; i < 9; i++) p.ccm.m[i] = M ? M[i] : 0.0;
    hipLaunchKernelGGL(k_fuse_rgb, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, p);
    return CHECK_LAUNCH();
}
int launch_fuse_rgb(hipStream_t st, const float* const* d_frames, float* const* d_frames_out, int K, size_t npx, const float* coeff,
                    const int* applied, const float* ev_off, const float* bias, int kmax, const double* M, float* d_out, int32_t* d_count, float* d_part) {
    if (K < 1 || kmax < 0 || kmax >= K || (K > MAXK && !d_part)) return -1;
    for (int k0 = 0; k0 < K; k0 += MAXK) {
        const int n = K - k0 < MAXK ? K - k0 : MAXK;
        int rc = launch_fuse_rgb_pass(st, d_frames + k0, d_frames_out ? d_frames_out + k0 : nullptr, n, npx, coeff + 3 * k0, applied + k0, ev_off + k0, bias + k0,
                                      k0 == 0, k0 + n == K, (kmax >= k0 && kmax < k0 + n) ? kmax - k0 : -1, ev_off[kmax], M, d_out, d_count, d_part);
        if (rc) return rc;
    }
    return 0;
}

// ---- WarpRectilinear ------------------------------------------------------------------------------
// pyx:18-40.  Cython lowers x**k on C floats to powf(x, k.0) (correctly rounded to float32 by glibc
// in all but ~0.3 % of arguments) and sqrt to the double sqrt.  Here x**2 is the exact float32
// product, x**4 / x**6 are float64 products rounded once, sqrt is the float64 sqrt.
struct WarpGeom { float cx, cy, m; double rm; };   // rm = 1 / m rounded to float64
static WarpGeom warp_geom(int width, int height, float cxn, float cyn) {
    WarpGeom g;
    g.cx = (float)(width - 1) * cxn;
    g.cy = (float)(height - 1) * cyn;
    float mx = fmaxf(fabsf(-g.cx), fabsf((float)(width - 1) - g.cx));
    float my = fmaxf(fabsf(-g.cy), fabsf((float)(height - 1) - g.cy));
    g.m = (float)sqrt((double)(mx * mx + my * my));
    g.rm = 1.0 / (double)g.m;
    return g;
}
struct WarpCoef { float kr0, kr1, kr2, kr3, kt0, kt1; };
// channel-independent part of pyx:26-29 (dx, dy, r and its even powers) and the per-channel polynomial :30-40
struct WarpRad { float dx, dy, dx2, dy2, r2, r4, r6; };
DEVI WarpRad warp_rad(float sx, float sy, const WarpGeom& g) {
    WarpRad w;
    // a / m for the frame-constant m (pyx:26-27): float(double(a) * RN53(1 / m)) IS the correctly rounded float32 quotient (the argument of div_by_sat in
    // demosaic_common.h) -- three instructions instead of the eleven of an IEEE float32 division (k_warp_remap at 100 MP: 2.10 -> 2.075 ms)
    w.dx = (float)((double)(sx - g.cx) * g.rm); w.dy = (float)((double)(sy - g.cy) * g.rm);
    w.dx2 = w.dx * w.dx; w.dy2 = w.dy * w.dy;
    float r = sqrtf(w.dx2 + w.dy2);   // == float(sqrt(double(s))): the float64 sqrt rounded to float32 is the correctly rounded float32 sqrt
    double rd = (double)r, r2d = rd * rd;
    w.r2 = (float)r2d; w.r4 = (float)(r2d * r2d); w.r6 = (float)((r2d * r2d) * r2d);
    return w;
}
DEVI void warp_poly(float sx, float sy, const WarpRad& w, const WarpCoef& k, const WarpGeom& g, float scale, float& ox, float& oy) {
    float f = ((k.kr0 + (k.kr1 * w.r2)) + (k.kr2 * w.r4)) + (k.kr3 * w.r6);
    float dxr = f * w.dx, dyr = f * w.dy;
    float dxt = k.kt0 * ((2.0f * w.dx) * w.dy) + k.kt1 * (w.r2 + 2.0f * w.dx2);
    float dyt = k.kt1 * ((2.0f * w.dx) * w.dy) + k.kt0 * (w.r2 + 2.0f * w.dy2);
    float xp = g.cx + g.m * (dxr + dxt), yp = g.cy + g.m * (dyr + dyt);
    ox = sx + (xp - sx) * scale;
    oy = sy + (yp - sy) * scale;
}
DEVI void warp_px(float sx, float sy, const WarpCoef& k, const WarpGeom& g, float scale, float& ox, float& oy) {
    warp_poly(sx, sy, warp_rad(sx, sy, g), k, g, scale, ox, oy);
}
__global__ void __launch_bounds__(256) k_warp_table(WarpCoef k, WarpGeom g, float scale, int width, int height,
                                                    const float* __restrict__ seed, float* __restrict__ table) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    size_t o = ((size_t)y * width + x) * 2;
    float sx = seed ? seed[o] : (float)x, sy = seed ? seed[o + 1] : (float)y;
    float ox, oy;
    warp_px(sx, sy, k, g, scale, ox, oy);
    *reinterpret_cast<float2*>(table + o) = make_float2(ox, oy);
}
int launch_warp_table(hipStream_t st, float kr0, float kr1, float kr2, float kr3, float kt0, float kt1, int width, int height,
                      float cxn, float cyn, float scale, const float* d_seed, float* d_table) {
    WarpCoef k = {kr0, kr1, kr2, kr3, kt0, kt1};
    WarpGeom g = warp_geom(width, height, cxn, cyn);
    dim3 grid((width + 255) / 256, height);
    hipLaunchKernelGGL(k_warp_table, grid, dim3(256), 0, st, k, g, scale, width, height, d_seed, d_table);
    return CHECK_LAUNCH();
}

// Restated cv2.remap(INTER_LANCZOS4, BORDER_CONSTANT 0), oracle orc_remap_lanczos4: the coordinate
// table is evaluated in the kernel (never materialised), clipped (chan_distortion_corr.py:95-96),
// quantised to 1/32 px (round half even), 8x8 taps with weights wy*wx, row sums then total.
void host_lanczos4_table(float tab[256]) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    for (int i = 0; i < 32; i++) {
        float x = (float)i * (1.0f / 32.0f);
        float* c = tab + 8 * i;
        if (x < 1.1920929e-07f) { for (int k = 0; k < 8; k++) c[k] = 0; c[3] = 1; continue; }
        float sum = 0;
        double y0 = -(x + 3) * 3.14159265358979323846 * 0.25, s0 = sin(y0), c0 = cos(y0);
        for (int k = 0; k < 8; k++) {
            double y = -(x + 3 - k) * 3.14159265358979323846 * 0.25;
            c[k] = (float)((cs[k][0] * s0 + cs[k][1] * c0) / (y * y));
            sum += c[k];
        }
        sum = 1.f / sum;
        for (int k = 0; k < 8; k++) c[k] *= sum;
    }
}
struct RemapParams {
    const float* in;   // (H,W,3)
    float* out;        // (H,W,3), must not alias in
    const float* tab;  // [32][8]
    int H, W;
    int row0, row1;    // output rows [row0, row1) are produced (a band of the frame; the whole frame is 0, H)
    WarpCoef k[3];
    WarpGeom g;
    float scale;
};
namespace {
#ifndef WARP_RPT
#define WARP_RPT 4                           // measured at 100 MP: 2 rows 2.28 ms, 3: 2.24, 4: 2.13, 5: 2.22, 6: 2.18, 8: 2.77
#endif
#ifndef WARP_STRIP
#define WARP_STRIP 16
#endif
constexpr int WRPT = WARP_RPT;               // output rows per thread
constexpr int WBX = 64, WBY = 4 * WRPT;     // output block of one workgroup
constexpr int WTW = 96, WTH = WBY + 16;     // largest per-channel source tile staged in LDS: 3 x 12 KB at 16 rows
}
// Round 3, measured and dropped (100 MP, tools/ab_bench.sh; this kernel 2.10 ms): one channel at a time with the rectangle stored twice, the second
// copy shifted by one element, so that the eight taps of a row are four aligned ds_read_b64 whatever the first column's parity (25 KB of LDS, 76-102
// VGPRs, up to six workgroups per CU): 2.64 ms with four rows per thread, 2.72 with three, 2.79 with two -- the extra barriers, the staging done three
// times and the doubled LDS stores cost more than the halved read cycles save; LDS padding to three workgroups per CU changes nothing (2.10): the kernel is
// bound by VALU issue (3 714 instructions per wave, 2 304 of them the taps' multiply, multiply, add) and LDS throughput (68 % busy), not by latency.
// 16-byte reads of the x weights +1.4 %, staging loads without an exec-masked block each (select after an always-valid load) +2.8 %.
// One workgroup = 64x16 output pixels (four rows per thread).  Each channel has its own smooth warp, so the 8x8 Lanczos footprints of
// a block cover a small source rectangle per channel, bounded by the block's corner pixels (+2 cells): it is
// staged in LDS once per channel (zero outside the image = BORDER_CONSTANT 0) and the 64 taps per pixel and
// channel are conflict-free LDS reads.  A pixel whose footprint is not inside the staged rectangle (extreme
// distortion) takes the same taps from global memory instead, so correctness never depends on the bound.
__global__ void __launch_bounds__(256) k_warp_remap(RemapParams p) {
    __shared__ __attribute__((aligned(16))) float stab[256];
    __shared__ float tile[3][WTH * WTW];
    __shared__ int corner[3][4][2];          // per channel: source cell (ix, iy) of the block's four corner pixels
    const int tid = threadIdx.x;
    stab[tid] = p.tab[tid];
    int tbx, tby;
    // vertical strips of 16 blocks (round 3): a block shares 11 of its 27 source rows with the block below, and a full block row of a 100 MP frame (3.8 MB)
    // does not survive in the 4 MB L2 until that block runs.  FETCH_SIZE 3.34 -> 1.32 GB per launch (2.7x -> 1.08x the 1.22 GB read once), L2 hit rate
    // 45 % -> 70 %; 1.94 -> 1.92 ms: the kernel is not bound by memory (profiles/r3_ab_warp_strips.log, strips of 4 / 8 / 16 blocks; 0 = row-major)
#if WARP_STRIP > 0
    xcd_tile_strips<WARP_STRIP>(tbx, tby);
#else
    xcd_tile(tbx, tby);
#endif
    const int by0 = p.row0 + tby * WBY;
    const int x = tbx * WBX + (tid & 63), y0 = by0 + (tid >> 6) * WRPT;
    const float xmax = (float)(p.W - 1), ymax = (float)(p.H - 1);
    // np.clip (chan_distortion_corr.py:95-96) as one v_med3 (the same value for every coordinate but NaN, which both forms turn into cell 0), and
    // cv2.remap's round(32 v) without v_rndne + v_cvt: 32 v is exact and < 2^22 (sides <= 2^17, checked by the caller), so 32 v + 1.5 * 2^23 has
    // the half-even rounded integer in its low mantissa bits.  Cell and phase are bit fields of that word minus the constant.
    auto cell = [&](float mx, float my, int& fx, int& fy) {
        mx = __builtin_amdgcn_fmed3f(mx, 0.0f, xmax); my = __builtin_amdgcn_fmed3f(my, 0.0f, ymax);
        fx = __float_as_int(__builtin_fmaf(mx, 32.0f, 12582912.0f)) - 0x4B400000;
        fy = __float_as_int(__builtin_fmaf(my, 32.0f, 12582912.0f)) - 0x4B400000;
    };
    if (tid < 12) {
        const int c = tid >> 2, k = tid & 3;
        int cx = min(tbx * WBX + ((k & 1) ? WBX - 1 : 0), p.W - 1), cy = min(by0 + ((k & 2) ? WBY - 1 : 0), p.row1 - 1);
        float mx, my; int fx, fy;
        warp_px((float)cx, (float)cy, p.k[c], p.g, p.scale, mx, my);
        cell(mx, my, fx, fy);
        corner[c][k][0] = fx >> 5; corner[c][k][1] = fy >> 5;
    }
    int sx[WRPT][3], sy[WRPT][3];
#pragma unroll
    for (int j = 0; j < WRPT; j++) {
        WarpRad wr = warp_rad((float)x, (float)(y0 + j), p.g);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float mx, my;
            warp_poly((float)x, (float)(y0 + j), wr, p.k[c], p.g, p.scale, mx, my);
            cell(mx, my, sx[j][c], sy[j][c]);
        }
    }
    __syncthreads();
    int tx0[3], ty0[3], tw[3], th[3];
    constexpr int NLD = (WTW * WTH + 255) / 256;     // tile cells per thread and channel
    float stage[3][NLD];
    // cell k of this thread on the fixed WTW grid, and its byte offset from the rectangle's origin: the same for the three channels
    int cry[NLD], crx[NLD];
    unsigned coff[NLD];
    const unsigned rowbytes = (unsigned)p.W * 12u;
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int idx = tid + k * 256;
        cry[k] = idx / WTW; crx[k] = idx - cry[k] * WTW;
        coff[k] = mul24((unsigned)cry[k], rowbytes) + 12u * (unsigned)crx[k];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int lox = min(min(corner[c][0][0], corner[c][1][0]), min(corner[c][2][0], corner[c][3][0]));
        int loy = min(min(corner[c][0][1], corner[c][1][1]), min(corner[c][2][1], corner[c][3][1]));
        int hix = max(max(corner[c][0][0], corner[c][1][0]), max(corner[c][2][0], corner[c][3][0]));
        int hiy = max(max(corner[c][0][1], corner[c][1][1]), max(corner[c][2][1], corner[c][3][1]));
        // the rectangle is the same for the whole workgroup: kept in scalar registers
        tx0[c] = __builtin_amdgcn_readfirstlane(lox - 3 - 2); ty0[c] = __builtin_amdgcn_readfirstlane(loy - 3 - 2);            // 2 cells of margin on every side
        tw[c] = __builtin_amdgcn_readfirstlane(min(hix + 4 + 2 - tx0[c] + 1, WTW)); th[c] = __builtin_amdgcn_readfirstlane(min(hiy + 4 + 2 - ty0[c] + 1, WTH));
        // all loads of the three tiles are issued before the first LDS store; every cell of the buffer gets a value -- zero outside the
        // image or the rectangle.  A cell is inside both iff its row is in [ylo, ylo + yn) and its column in [xlo, xlo + xn): two
        // subtract-and-compare pairs; the load takes (scalar rectangle origin, 32-bit cell offset).
        const char* const org = reinterpret_cast<const char*>(p.in + ((long long)ty0[c] * p.W + tx0[c]) * 3 + c);   // may lie outside the image: never read there
        const int ylo = max(0, -ty0[c]), yn = max(0, min(th[c], p.H - ty0[c]) - ylo);
        const int xlo = max(0, -tx0[c]), xn = max(0, min(tw[c], p.W - tx0[c]) - xlo);
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const bool ok = (unsigned)(cry[k] - ylo) < (unsigned)yn && (unsigned)(crx[k] - xlo) < (unsigned)xn;
            stage[c][k] = ok ? *reinterpret_cast<const float*>(org + coff[k]) : 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < NLD; k++)
            if ((k + 1) * 256 <= WTW * WTH || tid + k * 256 < WTW * WTH) tile[c][tid + k * 256] = stage[c][k];
    __syncthreads();
    if (x >= p.W) return;
#pragma unroll
    for (int j = 0; j < WRPT; j++) {
        const int y = y0 + j;
        if (y >= p.row1) break;
        float res[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int ix = (sx[j][c] >> 5) - 3, iy = (sy[j][c] >> 5) - 3;
            const float* wx = stab + 8 * (sx[j][c] & 31);
            const float* wy = stab + 8 * (sy[j][c] & 31);
            float sum = 0.0f;
            const int lx = ix - tx0[c], ly = iy - ty0[c];
            if (lx >= 0 && ly >= 0 && lx + 8 <= tw[c] && ly + 8 <= th[c]) {
#ifndef WARP_ROLLED_ROWS
                typedef const __attribute__((address_space(3))) float* TileRow;         // an LDS pointer by type: it stays one through the asm below (a generic pointer would turn the reads into flat loads)
                TileRow trow = (TileRow)&tile[c][ly * WTW + lx];
#else
                const float* trow = &tile[c][ly * WTW + lx];
#endif
                float wxr[8];
#pragma unroll
                for (int t = 0; t < 8; t++) wxr[t] = wx[t];
                // eight rows unrolled (round 3: 2.070 -> 2.026 ms at 100 MP against one row per trip; a software-pipelined rolled form that requests row r + 1
                // before accumulating row r: 2.32 ms -- profiles/r3_ab_warp_taploop.log)
#ifdef WARP_ROLLED_ROWS
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int r = 0; r < 8; r++) {
#ifndef WARP_ROLLED_ROWS
                    // a fresh base register every three rows (3 x 384 bytes is what the 8-bit dword offsets of ds_read2_b32 reach): the asm hides the sum from the
                    // compiler, which otherwise gives every pair of taps its own v_add_u32 (576 address adds per thread, a ninth of the kernel's instructions)
                    if (r % 3 == 0) { if (r) trow += 3 * WTW; asm("" : "+v"(trow)); }
                    TileRow row = trow + (r % 3) * WTW;
#else
                    const float* row = trow + r * WTW;
#endif
                    const float wyr = wy[r];
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        float v = row[t] * (wyr * wxr[t]);
                        acc = t == 0 ? v : acc + v;
                    }
                    sum = sum + acc;
                }
            } else {
#pragma unroll 1
                for (int r = 0; r < 8; r++) {   // rare (extreme distortion only): kept rolled, it would otherwise double the kernel's code size
                    int yy = iy + r;
                    bool yin = (unsigned)yy < (unsigned)p.H;
                    const float* row = p.in + ((size_t)(yin ? yy : 0) * p.W) * 3 + c;
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        int xx = ix + t;
                        float s = (yin && (unsigned)xx < (unsigned)p.W) ? row[(size_t)xx * 3] : 0.0f;
                        float v = s * (wy[r] * wx[t]);
                        acc = t == 0 ? v : acc + v;
                    }
                    sum = sum + acc;
                }
            }
            res[c] = sum;
        }
        float* o = p.out + ((size_t)y * p.W + x) * 3;
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
    }
}
// Generic restated cv2.remap(plane, mapx, mapy, INTER_LANCZOS4), BORDER_CONSTANT 0 (chan_distortion_corr.py:94-97 with
// an explicit table: the seeded / prior path).  src/dst element stride lets a plane of an interleaved image be used.
__global__ void __launch_bounds__(256) k_remap_table(const float* __restrict__ src, int sstride, const float* __restrict__ mapx,
                                                     const float* __restrict__ mapy, int mstride, const float* __restrict__ tab, int H, int W,
                                                     float clip_x, float clip_y, int do_clip, float* __restrict__ dst, int dstride) {
    __shared__ float stab[256];
    stab[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    float mx = mapx[o * mstride], my = mapy[o * mstride];
    if (do_clip) {   // np.clip(map, 0, size-1), chan_distortion_corr.py:95-96
        mx = mx < 0.0f ? 0.0f : (mx > clip_x ? clip_x : mx);
        my = my < 0.0f ? 0.0f : (my > clip_y ? clip_y : my);
    }
    int sx = (int)rintf(mx * 32.0f), sy = (int)rintf(my * 32.0f);
    int ix = (sx >> 5) - 3, iy = (sy >> 5) - 3;
    const float* wx = stab + 8 * (sx & 31);
    const float* wy = stab + 8 * (sy & 31);
    float sum = 0.0f;
    for (int r = 0; r < 8; r++) {
        int yy = iy + r;
        bool yin = (unsigned)yy < (unsigned)H;
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            int xx = ix + t;
            float s = (yin && (unsigned)xx < (unsigned)W) ? src[((size_t)yy * W + xx) * sstride] : 0.0f;
            float v = s * (wy[r] * wx[t]);
            acc = t == 0 ? v : acc + v;
        }
        sum = sum + acc;
    }
    dst[o * dstride] = sum;
}
int launch_remap_table(hipStream_t st, const float* src, int sstride, const float* mapx, const float* mapy, int mstride, const float* d_tab,
                       int H, int W, int do_clip, float* dst, int dstride) {
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k_remap_table, grid, dim3(256), 0, st, src, sstride, mapx, mapy, mstride, d_tab, H, W, (float)(W - 1), (float)(H - 1), do_clip,
                       dst, dstride);
    return CHECK_LAUNCH();
}

// Source rows the Lanczos footprints of output rows [row0, row1) touch: the same coordinate evaluation as k_warp_remap,
// reduced to a min / max source row over all pixels and channels.  rows[0] starts at INT_MAX, rows[1] at INT_MIN.
__global__ void __launch_bounds__(256) k_warp_src_rows(RemapParams p, int* __restrict__ rows) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = p.row0 + blockIdx.y;
    int lo = 0x7fffffff, hi = -0x7fffffff - 1;
    if (x < p.W && y < p.row1) {
        const float ymax = (float)(p.H - 1);
        WarpRad wr = warp_rad((float)x, (float)y, p.g);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float mx, my;
            warp_poly((float)x, (float)y, wr, p.k[c], p.g, p.scale, mx, my);
            my = my < 0.0f ? 0.0f : (my > ymax ? ymax : my);
            int iy = ((int)rintf(my * 32.0f) >> 5) - 3;
            lo = min(lo, iy); hi = max(hi, iy + 7);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(rows, lo); atomicMax(rows + 1, hi); }
}

static int fill_remap_params(RemapParams& p, int H, int W, const double* coeffs, int planes, double cxn, double cyn, float scale, int row0, int row1) {
    if (planes != 3 || row0 < 0 || row1 > H || row0 >= row1) return -1;
    p.H = H; p.W = W; p.scale = scale; p.row0 = row0; p.row1 = row1;
    for (int c = 0; c < 3; c++) {
        const double* k = coeffs + 6 * c;
        p.k[c] = {(float)k[0], (float)k[1], (float)k[2], (float)k[3], (float)k[4], (float)k[5]};
    }
    p.g = warp_geom(W, H, (float)cxn, (float)cyn);
    return 0;
}
int launch_warp_src_rows(hipStream_t st, int H, int W, const double* coeffs, int planes, double cxn, double cyn, float scale, int row0, int row1,
                         int* d_rows) {
    RemapParams p;
    p.in = nullptr; p.out = nullptr; p.tab = nullptr;
    if (fill_remap_params(p, H, W, coeffs, planes, cxn, cyn, scale, row0, row1)) return -1;
    // (device-side fills: an async copy from this stack frame could outlive it)
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_rows), 0x7fffffff, 1, st) != hipSuccess ||
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_rows + 1), (int)0x80000000u, 1, st) != hipSuccess) return -1;
    dim3 grid((W + 255) / 256, row1 - row0);
    hipLaunchKernelGGL(k_warp_src_rows, grid, dim3(256), 0, st, p, d_rows);
    return CHECK_LAUNCH();
}

int launch_warp_remap(hipStream_t st, const float* d_in, float* d_out, int H, int W, const double* coeffs, int planes, double cxn,
                      double cyn, float scale, const float* d_lanczos_tab, int row0, int row1) {
    RemapParams p;
    p.in = d_in; p.out = d_out; p.tab = d_lanczos_tab;
    if (fill_remap_params(p, H, W, coeffs, planes, cxn, cyn, scale, row0, row1)) return -1;
    dim3 grid((W + WBX - 1) / WBX, (row1 - row0 + WBY - 1) / WBY);
    hipLaunchKernelGGL(k_warp_remap, grid, dim3(256), 0, st, p);
    return CHECK_LAUNCH();
}


// ------------------------------------------------------------------------------------------------
// Lateral chromatic-aberration removal, the apply half of corr_ca/ca_removal.py:48-131.
// A lens model's coordinate field exists for the top-left quadrant only ((h,w,2) = (dy,dx) from the image centre,
// corr_ca/model/generic.py:56-101); the other quadrants are mirror images with the sign of one component flipped,
// so the kernels mirror on the fly instead of reading a full (H,W,2) field.  ca_removal.py:96-99 then adds
// (size-1)/2 and clips to [0, size-1]; the sample is the restated cv2.remap(INTER_LINEAR): 1/32 px quantisation,
// weights = float products of (1-f, f), BORDER_CONSTANT 0, summed in source order (oracle remap_linear_px).
namespace {
struct CaGeom { int H, W, h, w; float cx, cy, xmax, ymax; };
DEVI void ca_map(const float* __restrict__ quad, const CaGeom& g, int y, int x, float& mx, float& my) {
    int qy = y < g.h ? y : g.H - 1 - y, qx = x < g.w ? x : g.W - 1 - x;
    float2 d = *reinterpret_cast<const float2*>(quad + ((size_t)qy * g.w + qx) * 2);
    float dy = y >= g.h ? -d.x : d.x, dx = x >= g.w ? -d.y : d.y;
    float ax = dx + g.cx, ay = dy + g.cy;
    mx = ax < 0.0f ? 0.0f : (ax > g.xmax ? g.xmax : ax);
    my = ay < 0.0f ? 0.0f : (ay > g.ymax ? g.ymax : ay);
}
DEVI float remap_linear_px(const float* __restrict__ src, int H, int W, float mx, float my) {
    int sx = (int)rintf(mx * 32.0f), sy = (int)rintf(my * 32.0f);
    int ix = sx >> 5, iy = sy >> 5;
    float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    bool y0 = (unsigned)iy < (unsigned)H, y1 = (unsigned)(iy + 1) < (unsigned)H, x0 = (unsigned)ix < (unsigned)W, x1 = (unsigned)(ix + 1) < (unsigned)W;
    const float* p = src + (size_t)(y0 ? iy : 0) * W;
    const float* q = src + (size_t)(y1 ? iy + 1 : 0) * W;
    float v00 = (y0 && x0) ? p[ix] : 0.0f, v01 = (y0 && x1) ? p[ix + 1] : 0.0f;
    float v10 = (y1 && x0) ? q[ix] : 0.0f, v11 = (y1 && x1) ? q[ix + 1] : 0.0f;
    return ((v00 * (wy0 * wx0) + v01 * (wy0 * fx)) + v10 * (fy * wx0)) + v11 * (fy * fx);
}
CaGeom ca_geom(int H, int W) {
    CaGeom g;
    g.H = H; g.W = W; g.h = H / 2; g.w = W / 2;
    g.cx = (float)(((double)W - 1.0) / 2.0); g.cy = (float)(((double)H - 1.0) / 2.0);
    g.xmax = (float)(W - 1); g.ymax = (float)(H - 1);
    return g;
}
}  // namespace
// ca_removal.py:104-110 / :122-130: only the samples at the channel's own photosites survive (bayer_to_rgbg(...)[0] or [2]), divided
// by the white-balance multiplier that was applied before the resampling; they are written straight back into the mosaic
__global__ void __launch_bounds__(256) k_ca_remap_sites(const float* __restrict__ src, CaGeom g, const float* __restrict__ quad, int oy, int ox, float wb,
                                                        float* __restrict__ bayer) {
    int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= g.w || i >= g.h) return;
    float mx, my;
    ca_map(quad, g, 2 * i + oy, 2 * j + ox, mx, my);
    bayer[(size_t)(2 * i + oy) * g.W + 2 * j + ox] = remap_linear_px(src, g.H, g.W, mx, my) / wb;
}
// ca_removal.py:96-102 / :114-120 in one kernel: green is pulled onto the channel's geometry (the bilinear remap of g_full at the
// inverse model's coordinates) straight into an LDS tile, and resample_r / resample_b (eag.py:160-186 -> :126-143) runs on that tile:
// the remapped green never goes to memory.  One workgroup = 32x8 CFA quads (64x16 px), one thread per quad; the tile carries a
// 2 px ring (1 quad for the 3x3 quarter-plane filters, 1 px for the 3x3 blur).  Quads on the image border need two different
// reflections of that ring (REFLECT_101 of the quarter plane for filter2D, of the full plane for GaussianBlur): a second, tiny
// kernel evaluates their windows directly (k_ca_upsample_border, one thread per border quad).
namespace {
constexpr int CUX = 32, CUY = 8;                       // quads per workgroup
constexpr int CTW = 2 * CUX + 4, CTH = 2 * CUY + 4;    // green tile in px: 68 x 20
constexpr int CSW = CUX + 2, CSH = CUY + 2;            // channel samples (quarter plane): 34 x 10
}
template <int O>
__global__ void __launch_bounds__(256) k_ca_upsample_fused(const float* __restrict__ bayer, const float* __restrict__ g_full, CaGeom g,
                                                           const float* __restrict__ quad, float wb, float* __restrict__ out) {
    __shared__ float s_gat[CTH][CTW];
    __shared__ float s_sub[CSH][CSW];
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int tid = threadIdx.x, q0x = tbx * CUX, q0y = tby * CUY;
    const int H = g.H, W = g.W, h = g.h, w = g.w;
    {   // tile of remapped green in three sweeps over a thread's cells, so that every sweep's loads are in flight together:
        // coordinate field, then the four taps, then the blend (remap_linear_px split in two)
        constexpr int NC = (CTH * CTW + 255) / 256;
        float mx[NC], my[NC];
        bool in[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            int idx = tid + k * 256;
            int ty = idx / CTW, tx = idx - ty * CTW;
            int Y = 2 * q0y - 2 + ty, X = 2 * q0x - 2 + tx;
            in[k] = idx < CTH * CTW && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
            mx[k] = 0.0f; my[k] = 0.0f;
            if (in[k]) ca_map(quad, g, Y, X, mx[k], my[k]);
        }
        float v[NC][4], fx[NC], fy[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            int sx = (int)rintf(mx[k] * 32.0f), sy = (int)rintf(my[k] * 32.0f);
            int ix = sx >> 5, iy = sy >> 5;
            fx[k] = (float)(sx & 31) * (1.0f / 32.0f); fy[k] = (float)(sy & 31) * (1.0f / 32.0f);
            // clipped coordinates: (iy, ix) is inside the image, only the +1 neighbours can fall outside (BORDER_CONSTANT 0)
            bool x1 = ix + 1 < W, y1 = iy + 1 < H;
            const float* p0 = g_full + (size_t)iy * W + ix;
            const float* p1 = g_full + (size_t)(y1 ? iy + 1 : iy) * W + ix;
            v[k][0] = in[k] ? p0[0] : 0.0f;
            v[k][1] = (in[k] && x1) ? p0[1] : 0.0f;
            v[k][2] = (in[k] && y1) ? p1[0] : 0.0f;
            v[k][3] = (in[k] && y1 && x1) ? p1[1] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            int idx = tid + k * 256;
            if (idx < CTH * CTW) {
                float wx0 = 1.0f - fx[k], wy0 = 1.0f - fy[k];
                float r = ((v[k][0] * (wy0 * wx0) + v[k][1] * (wy0 * fx[k])) + v[k][2] * (fy[k] * wx0)) + v[k][3] * (fy[k] * fx[k]);
                int ty = idx / CTW, tx = idx - ty * CTW;
                s_gat[ty][tx] = in[k] ? r : 0.0f;
            }
        }
        constexpr int NS = (CSH * CSW + 255) / 256;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            int idx = tid + k * 256;
            if (idx < CSH * CSW) {
                int sy = idx / CSW, sx = idx - sy * CSW;
                int a = q0y - 1 + sy, c = q0x - 1 + sx;
                s_sub[sy][sx] = ((unsigned)a < (unsigned)h && (unsigned)c < (unsigned)w) ? bayer[(size_t)(2 * a + O) * W + 2 * c + O] * wb : 0.0f;
            }
        }
    }
    __syncthreads();
    const int lqy = tid / CUX, lqx = tid - lqy * CUX;
    const int i = q0y + lqy, j = q0x + lqx;
    if (i >= h || j >= w) return;
    Win3 wg, wd;
    float Wn[4][4];
    if (!(i >= 1 && i <= h - 2 && j >= 1 && j <= w - 2)) return;   // border quads: k_ca_upsample_border
    {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float gs = s_gat[2 * (lqy + r) + O][2 * (lqx + c) + O];       // tile px of quad (i-1+r, j-1+c), site O
                wg.v[r][c] = gs;
                wd.v[r][c] = s_sub[lqy + r][lqx + c] - gs;                    // channel_diff (eag.py:142)
            }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) Wn[r][c] = s_gat[2 * lqy + 1 + r][2 * lqx + 1 + c];
    }
    float hf[4], fg[4], fd[4];
    highpass_quad(Wn, hf);
    if (O == 0) { filt_base_tl(wg, fg); filt_base_tl(wd, fd); } else { filt_base_br(wg, fg); filt_base_br(wd, fd); }
    *reinterpret_cast<float2*>(out + (size_t)(2 * i) * W + 2 * j) = make_float2(fd[0] + (fg[0] + hf[0]), fd[1] + (fg[1] + hf[1]));
    *reinterpret_cast<float2*>(out + (size_t)(2 * i + 1) * W + 2 * j) = make_float2(fd[2] + (fg[2] + hf[2]), fd[3] + (fg[3] + hf[3]));
}
template <int O>
__global__ void __launch_bounds__(64) k_ca_upsample_border(const float* __restrict__ bayer, const float* __restrict__ g_full, CaGeom g,
                                                          const float* __restrict__ quad, float wb, float* __restrict__ out) {
    const int H = g.H, W = g.W, h = g.h, w = g.w;
    int t = blockIdx.x * 64 + threadIdx.x, i, j;
    if (t < w) { i = 0; j = t; }
    else if (t < 2 * w) { i = h - 1; j = t - w; }
    else { int u = t - 2 * w; i = 1 + (u >> 1); j = (u & 1) ? w - 1 : 0; if (i > h - 2) return; }
    auto gat = [&](int Y, int X) {
        float mx, my;
        ca_map(quad, g, Y, X, mx, my);
        return remap_linear_px(g_full, H, W, mx, my);
    };
    Win3 wg, wd;
    float Wn[4][4];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {                                      // REFLECT_101 of the quarter planes
            int a = b_101(i - 1 + r, h), cc = b_101(j - 1 + c, w);
            float gs = gat(2 * a + O, 2 * cc + O);
            wg.v[r][c] = gs;
            wd.v[r][c] = bayer[(size_t)(2 * a + O) * W + 2 * cc + O] * wb - gs;
        }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) Wn[r][c] = gat(b_101(2 * i - 1 + r, H), b_101(2 * j - 1 + c, W));   // REFLECT_101 at full resolution
    float hf[4], fg[4], fd[4];
    highpass_quad(Wn, hf);
    if (O == 0) { filt_base_tl(wg, fg); filt_base_tl(wd, fd); } else { filt_base_br(wg, fg); filt_base_br(wd, fd); }
    *reinterpret_cast<float2*>(out + (size_t)(2 * i) * W + 2 * j) = make_float2(fd[0] + (fg[0] + hf[0]), fd[1] + (fg[1] + hf[1]));
    *reinterpret_cast<float2*>(out + (size_t)(2 * i + 1) * W + 2 * j) = make_float2(fd[2] + (fg[2] + hf[2]), fd[3] + (fg[3] + hf[3]));
}
int launch_ca_upsample_fused(hipStream_t st, const float* bayer, const float* g_full, int H, int W, const float* d_quad, int pos, float wb, float* out) {
    if (pos != 0 && pos != 3) return -1;
    const int h = H / 2, w = W / 2;
    dim3 grid((w + CUX - 1) / CUX, (h + CUY - 1) / CUY);
    const int nborder = 2 * w + 2 * (h > 2 ? h - 2 : 0);
    dim3 gb((nborder + 63) / 64);
    if (pos == 0) {
        hipLaunchKernelGGL(k_ca_upsample_fused<0>, grid, dim3(256), 0, st, bayer, g_full, ca_geom(H, W), d_quad, wb, out);
        hipLaunchKernelGGL(k_ca_upsample_border<0>, gb, dim3(64), 0, st, bayer, g_full, ca_geom(H, W), d_quad, wb, out);
    } else {
        hipLaunchKernelGGL(k_ca_upsample_fused<1>, grid, dim3(256), 0, st, bayer, g_full, ca_geom(H, W), d_quad, wb, out);
        hipLaunchKernelGGL(k_ca_upsample_border<1>, gb, dim3(64), 0, st, bayer, g_full, ca_geom(H, W), d_quad, wb, out);
    }
    return CHECK_LAUNCH();
}
int launch_ca_remap_sites(hipStream_t st, const float* src, int H, int W, const float* d_quad, int oy, int ox, float wb, float* bayer) {
    dim3 grid((W / 2 + 63) / 64, (H / 2 + 3) / 4);
    hipLaunchKernelGGL(k_ca_remap_sites, grid, dim3(256), 0, st, src, ca_geom(H, W), d_quad, oy, ox, wb, bayer);
    return CHECK_LAUNCH();
}
