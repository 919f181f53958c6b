// k_eag.hip -- Edge-Assisted-Gaussian ("Fast") and Draft demosaic for gfx950.
//
// EAG follows debayer/edge_assisted_gaussian.py:10-201 of pySP; Draft follows
// debayer/fast_resize.py:7-44 (+ cv2.resize INTER_LINEAR x2 restated as in oracle/pysp_oracle.c).
//
// EAG kernel: one workgroup = one 64x32 px output tile = 32x16 CFA quads, one thread per quad.
//   P0  raw mosaic -> four de-interleaved quarter planes in LDS (halo 2 quads, symmetric border)
//   P1  gradient-weighted green at R/B sites * wb[1], colour differences D = sub*wb - g
//       (halo 1 quad; outside the image: REFLECT_101 of the quarter plane)
//   P2  per quad: g - GaussianBlur3(g), photosite-aware resampling of R and B, colour tail, store
#include "demosaic_common.h"
#include "kernels.h"

namespace {
#ifndef EAG_TQX
#define EAG_TQX 32
#define EAG_TQY 16
#endif
constexpr int TQX = EAG_TQX, TQY = EAG_TQY;
constexpr int MWX = TQX + 4, MWY = TQY + 4;   // raw planes, halo 2 quads
constexpr int GX = TQX + 2, GY = TQY + 2;     // green / difference planes, halo 1 quad
constexpr int NT = TQX * TQY;                 // 512
enum { P_R = 0, P_G1 = 1, P_G2 = 2, P_B = 3 };
enum { Q_GR = 0, Q_GB, Q_DR, Q_DB };

// edge_assisted_gaussian.py:36-49
DEVI float delta_mix(float top, float bottom, float left, float right) {
    float dy = fabsf(top - bottom), dx = fabsf(left - right), s = dy + dx;
    float ax = (left + right) / 2.0f, ay = (top + bottom) / 2.0f;
    float sy = s != 0.0f ? dy / s : 0.5f;
    float sx = 1.0f - sy;
    return ay * sx + ax * sy;
}
}  // namespace

struct EagParams {
    MosaicSrc src;
    float* out;
    int H, W;
    float wb[3];
    int tail;
    Ccm ccm;
};

// TAIL is a template parameter: with the colour tail chosen at run time the kernel needs 66 VGPRs, with it fixed 40 (8 waves/SIMD)
template <bool TINY, bool U16, int TAIL>
__global__ void __launch_bounds__(NT) k_eag(EagParams p) {
    __shared__ float mw[4][MWY][MWX];
    __shared__ float gq[4][GY][GX];
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int tq0x = tbx * TQX, tq0y = tby * TQY;
    const bool inside = tq0y >= 2 && tq0x >= 2 && tq0y + TQY + 2 <= h && tq0x + TQX + 2 <= w;   // no border rule applies in P0/P1

    // P0: raw planes, cv2.copyMakeBorder(..., BORDER_REFLECT) per plane (eag.py:86-87).  One 8-byte load per quad
    // row; all loads of a thread are issued before the first LDS store so that they are in flight together.
    {
        constexpr int NPAIR = 2 * MWY * MWX, NL = (NPAIR + NT - 1) / NT;
        float2 tmp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            int idx = tid + k * NT;
            if (idx >= NPAIR) idx = NPAIR - 1;
            int ry = idx / MWX, mx = idx - ry * MWX, my = ry >> 1, dy = ry & 1;
            int qi = tq0y - 2 + my, qj = tq0x - 2 + mx;
            if (!inside) {                                                     // uniform per workgroup: interior tiles skip the border rules
                qi = TINY ? b_sym(qi, h) : b_sym1(qi, h);
                qj = TINY ? b_sym(qj, w) : b_sym1(qj, w);
            }
            tmp[k] = load_mosaic_pair<U16>(p.src, (size_t)(2 * qi + dy) * W + 2 * qj, dy ? 3 : 0, dy ? 2 : 1);
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            int idx = tid + k * NT;
            if (idx < NPAIR) {
                int ry = idx / MWX, mx = idx - ry * MWX, my = ry >> 1, dy = ry & 1;
                mw[dy ? P_G2 : P_R][my][mx] = tmp[k].x;
                mw[dy ? P_B : P_G1][my][mx] = tmp[k].y;
            }
        }
    }
    __syncthreads();

    // P1: green at red / blue sites (eag.py:99-121), * wb[1] (:193); D = sub*wb - g (:142,:194)
    for (int idx = tid; idx < GY * GX; idx += NT) {
        int gy = idx / GX, gx = idx - gy * GX;
        int a = gy + 1, c = gx + 1;
        if (!inside) {
            int ri = TINY ? b_101(tq0y - 1 + gy, h) : b_1011(tq0y - 1 + gy, h);
            int rj = TINY ? b_101(tq0x - 1 + gx, w) : b_1011(tq0x - 1 + gx, w);
            a = ri - (tq0y - 2); c = rj - (tq0x - 2);
            if (a < 1 || a > MWY - 2 || c < 1 || c > MWX - 2) continue;
        }
        float gr = delta_mix(mw[P_G2][a - 1][c], mw[P_G2][a][c], mw[P_G1][a][c - 1], mw[P_G1][a][c]) * p.wb[1];
        float gb = delta_mix(mw[P_G1][a][c], mw[P_G1][a + 1][c], mw[P_G2][a][c], mw[P_G2][a][c + 1]) * p.wb[1];
        gq[Q_GR][gy][gx] = gr; gq[Q_GB][gy][gx] = gb;
        gq[Q_DR][gy][gx] = mw[P_R][a][c] * p.wb[0] - gr;
        gq[Q_DB][gy][gx] = mw[P_B][a][c] * p.wb[2] - gb;
    }
    __syncthreads();

    // P2
    const int lqy = tid / TQX, lqx = tid - lqy * TQX;
    const int qi = tq0y + lqy, qj = tq0x + lqx;
    if (qi >= h || qj >= w) return;
    const int gy = lqy + 1, gx = lqx + 1, my = lqy + 2, mx = lqx + 2;
    const bool at_top = qi == 0, at_bot = qi == h - 1, at_left = qj == 0, at_right = qj == w - 1;
    const float wg = p.wb[1];
    Win3 wgr = load_win<GX>(&gq[Q_GR][0][0], gy, gx), wgb = load_win<GX>(&gq[Q_GB][0][0], gy, gx);
    float g1_c = mw[P_G1][my][mx] * wg, g2_c = mw[P_G2][my][mx] * wg;
    float Wn[4][4] = {{wgb.v[0][0], mw[P_G2][my - 1][mx] * wg, wgb.v[0][1], mw[P_G2][my - 1][mx + 1] * wg},
                      {mw[P_G1][my][mx - 1] * wg, wgr.v[1][1], g1_c, wgr.v[1][2]},
                      {wgb.v[1][0], g2_c, wgb.v[1][1], mw[P_G2][my][mx + 1] * wg},
                      {mw[P_G1][my + 1][mx - 1] * wg, wgr.v[2][1], mw[P_G1][my + 1][mx] * wg, wgr.v[2][2]}};
    if (at_top | at_bot | at_left | at_right) {     // GaussianBlur REFLECT_101 at full resolution (edge waves only)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (at_top) Wn[0][k] = Wn[2][k];
            if (at_bot) Wn[3][k] = Wn[1][k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (at_left) Wn[k][0] = Wn[k][2];
            if (at_right) Wn[k][3] = Wn[k][1];
        }
    }
    float hf[4], fg[4], fd[4], rr[4], bb[4];
    highpass_quad(Wn, hf);                                           // eag.py:156
    filt_base_tl(wgr, fg);
    { Win3 wd = load_win<GX>(&gq[Q_DR][0][0], gy, gx); filt_base_tl(wd, fd); }
#pragma unroll
    for (int k = 0; k < 4; k++) rr[k] = fd[k] + (fg[k] + hf[k]);     // eag.py:141,143
    filt_base_br(wgb, fg);
    { Win3 wd = load_win<GX>(&gq[Q_DB][0][0], gy, gx); filt_base_br(wd, fd); }
#pragma unroll
    for (int k = 0; k < 4; k++) bb[k] = fd[k] + (fg[k] + hf[k]);
    float gg[4] = {wgr.v[1][1], g1_c, g2_c, wgb.v[1][1]};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float r = rr[k], g = gg[k], b = bb[k];
        colour_tail(TAIL, p.ccm.m, r, g, b);
        float* o = p.out + ((size_t)(2 * qi + (k >> 1)) * W + (2 * qj + (k & 1))) * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}

int launch_eag(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int tail, float* d_out, Timeline* tl) {
    EagParams a;
    a.src = src; a.out = d_out; a.H = H; a.W = W; a.tail = tail;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M ? M[i] : (i % 4 == 0 ? 1.0 : 0.0);
    dim3 g((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY);
    if (tl) tl->begin(st, "k_eag");
    const bool tiny = H / 2 < 4 || W / 2 < 4, u16 = src.u16 != nullptr;
#define EAG_LAUNCH(T) \
    do { \
        if (tiny && u16) hipLaunchKernelGGL((k_eag<true, true, T>), g, dim3(NT), 0, st, a); \
        else if (tiny) hipLaunchKernelGGL((k_eag<true, false, T>), g, dim3(NT), 0, st, a); \
        else if (u16) hipLaunchKernelGGL((k_eag<false, true, T>), g, dim3(NT), 0, st, a); \
        else hipLaunchKernelGGL((k_eag<false, false, T>), g, dim3(NT), 0, st, a); \
    } while (0)
    switch (tail) {
        case 0: EAG_LAUNCH(0); break;
        case 1: EAG_LAUNCH(1); break;
        case 2: EAG_LAUNCH(2); break;
        case 3: EAG_LAUNCH(3); break;
        default: return -1;
    }
#undef EAG_LAUNCH
    if (tl) tl->end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ================================================================================================
// Draft (fast_resize.py:21-39): quarter-resolution RGB with 3/4-1/4 diagonal R/B alignment, then
// bilinear x2 (half-pixel centres, edge clamp; horizontal pass then vertical pass).
namespace {
template <bool U16>
DEVI void draft_q(const MosaicSrc& bay, int h, int w, int W, int i, int j, const float wb[3], float q[3]) {
    int i1 = i + 1 < h ? i + 1 : h - 1, j1 = j + 1 < w ? j + 1 : w - 1;   // r padded bottom/right (REFLECT)
    int i0 = i > 0 ? i - 1 : 0, j0 = j > 0 ? j - 1 : 0;                   // b padded top/left
    float r = load_mosaic<U16>(bay, (size_t)(2 * i) * W + 2 * j, 0), rd = load_mosaic<U16>(bay, (size_t)(2 * i1) * W + 2 * j1, 0);
    float b = load_mosaic<U16>(bay, (size_t)(2 * i + 1) * W + 2 * j + 1, 2), bd = load_mosaic<U16>(bay, (size_t)(2 * i0 + 1) * W + 2 * j0 + 1, 2);
    float g1 = load_mosaic<U16>(bay, (size_t)(2 * i) * W + 2 * j + 1, 1), g2 = load_mosaic<U16>(bay, (size_t)(2 * i + 1) * W + 2 * j, 3);
    q[0] = (0.75f * r + 0.25f * rd) * wb[0];
    q[1] = ((g1 + g2) / 2.0f) * wb[1];
    q[2] = (0.75f * b + 0.25f * bd) * wb[2];
}
DEVI void lin_tap(int X, int n, int& s0, int& s1, float& a0, float& a1) {
    float f = ((float)X + 0.5f) * 0.5f - 0.5f;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= n - 1) { s = n - 1; f = 0.0f; }
    s0 = s; s1 = s + 1 < n ? s + 1 : s; a0 = 1.0f - f; a1 = f;
}
}  // namespace

template <bool U16>
__global__ void __launch_bounds__(256) k_draft(EagParams p) {
    int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    if (X >= W) return;
    int sx0, sx1, sy0, sy1; float a0, a1, b0, b1;
    lin_tap(X, w, sx0, sx1, a0, a1);
    lin_tap(Y, h, sy0, sy1, b0, b1);
    float q00[3], q01[3], q10[3], q11[3];
    draft_q<U16>(p.src, h, w, W, sy0, sx0, p.wb, q00); draft_q<U16>(p.src, h, w, W, sy0, sx1, p.wb, q01);
    draft_q<U16>(p.src, h, w, W, sy1, sx0, p.wb, q10); draft_q<U16>(p.src, h, w, W, sy1, sx1, p.wb, q11);
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float h0 = q00[c] * a0 + q01[c] * a1, h1 = q10[c] * a0 + q11[c] * a1;
        o[c] = h0 * b0 + h1 * b1;
    }
    colour_tail(p.tail, p.ccm.m, o[0], o[1], o[2]);
    float* d = p.out + ((size_t)Y * W + X) * 3;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

int launch_draft(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int tail, float* d_out, Timeline* tl) {
    EagParams a;
    a.src = src; a.out = d_out; a.H = H; a.W = W; a.tail = tail;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M ? M[i] : (i % 4 == 0 ? 1.0 : 0.0);
    dim3 g((W + 255) / 256, H);
    if (tl) tl->begin(st, "k_draft");
    if (src.u16) hipLaunchKernelGGL(k_draft<true>, g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_draft<false>, g, dim3(256), 0, st, a);
    if (tl) tl->end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ================================================================================================
// Stand-alone forms of the public helpers of edge_assisted_gaussian.py (used by callers outside the fused
// demosaic, e.g. corr_ca/ca_removal.py:90,105,122).  Plain one-thread-per-quad kernels on global memory:
// neighbours are L2 hits; these are not on the fused hot path.
namespace {
template <int STRIDE_UNUSED = 0>
DEVI Win3 load_win_global(const float* plane, int h, int w, int i, int j) {   // 3x3, REFLECT_101 on the quarter plane
    Win3 wn;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) wn.v[r][c] = plane[(size_t)b_101(i - 1 + r, h) * w + b_101(j - 1 + c, w)];
    return wn;
}
}  // namespace

// eag.py:51-124 resample_g_to_full_resolution: g1, g2 (h,w) -> (2h,2w)
__global__ void __launch_bounds__(256) k_resample_g(const float* __restrict__ g1, const float* __restrict__ g2, int h, int w, int weighted,
                                                    float* __restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    auto G1 = [&](int a, int c) { return g1[(size_t)b_sym(a, h) * w + b_sym(c, w)]; };   // copyMakeBorder(..., BORDER_REFLECT)
    auto G2 = [&](int a, int c) { return g2[(size_t)b_sym(a, h) * w + b_sym(c, w)]; };
    float rt = G2(i - 1, j), rb = G2(i, j), rl = G1(i, j - 1), rr = G1(i, j);                // red site   (:105-108)
    float bt = G1(i, j), bb = G1(i + 1, j), bl = G2(i, j), br = G2(i, j + 1);                // blue site  (:99-102)
    float gr, gb;
    if (weighted) { gr = delta_mix(rt, rb, rl, rr); gb = delta_mix(bt, bb, bl, br); }
    else { gr = (((rt + rb) + rl) + rr) / 4.0f; gb = (((bt + bb) + bl) + br) / 4.0f; }       // :113-114
    size_t W = 2 * (size_t)w;
    out[(size_t)(2 * i) * W + 2 * j] = gr; out[(size_t)(2 * i) * W + 2 * j + 1] = g1[(size_t)i * w + j];
    out[(size_t)(2 * i + 1) * W + 2 * j] = g2[(size_t)i * w + j]; out[(size_t)(2 * i + 1) * W + 2 * j + 1] = gb;
}
// g - cv2.GaussianBlur(g,(3,3),1.0) (eag.py:156,170,184) on a full-resolution plane, REFLECT_101
__global__ void __launch_bounds__(256) k_highpass(const float* __restrict__ g, int H, int W, float* __restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int xs[3] = {b_101(x - 1, W), x, b_101(x + 1, W)}, ys[3] = {b_101(y - 1, H), y, b_101(y + 1, H)};
    float rb[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float* row = g + (size_t)ys[r] * W;
        rb[r] = row[xs[1]] * GK0 + (row[xs[0]] + row[xs[2]]) * GK1;
    }
    out[(size_t)y * W + x] = g[(size_t)y * W + x] - (rb[1] * GK0 + (rb[0] + rb[2]) * GK1);
}
// eag.py:126-143 resample_channel: sub, g_sub (h,w); g_hf (2h,2w); pos 0 = TOP_LEFT, 3 = BOTTOM_RIGHT (the two the
// reference uses); one thread per quad.
__global__ void __launch_bounds__(256) k_resample_channel(const float* __restrict__ sub, const float* __restrict__ g_sub,
                                                          const float* __restrict__ g_hf, int h, int w, int pos, float* __restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    Win3 wg = load_win_global<>(g_sub, h, w, i, j), ws = load_win_global<>(sub, h, w, i, j), wd;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) wd.v[r][c] = ws.v[r][c] - wg.v[r][c];        // channel_diff (:142), elementwise before the filter
    float fg[4], fd[4];
    if (pos == 0) { filt_base_tl(wg, fg); filt_base_tl(wd, fd); } else { filt_base_br(wg, fg); filt_base_br(wd, fd); }
    size_t W = 2 * (size_t)w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        size_t o = (size_t)(2 * i + (k >> 1)) * W + 2 * j + (k & 1);
        out[o] = fd[k] + (fg[k] + g_hf[o]);
    }
}
int launch_resample_g(hipStream_t st, const float* g1, const float* g2, int h, int w, int weighted, float* out) {
    hipLaunchKernelGGL(k_resample_g, dim3((w + 255) / 256, h), dim3(256), 0, st, g1, g2, h, w, weighted, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_highpass(hipStream_t st, const float* g, int H, int W, float* out) {
    hipLaunchKernelGGL(k_highpass, dim3((W + 255) / 256, H), dim3(256), 0, st, g, H, W, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_resample_channel(hipStream_t st, const float* sub, const float* g_sub, const float* g_hf, int h, int w, int pos, float* out) {
    if (pos != 0 && pos != 3) return -1;
    hipLaunchKernelGGL(k_resample_channel, dim3((w + 255) / 256, h), dim3(256), 0, st, sub, g_sub, g_hf, h, w, pos, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- corr_ca/ca_removal.py:84-130 building blocks that work on the mosaic itself (no de-interleaved planes) -------------
// :84-85  bayer_to_rgbg + resample_g_to_full_resolution(g1, g2): green read straight from its two CFA sites
__global__ void __launch_bounds__(256) k_ca_green(const float* __restrict__ bayer, int H, int W, float* __restrict__ out) {
    const int h = H >> 1, w = W >> 1;
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    auto G1 = [&](int a, int c) { return bayer[(size_t)(2 * b_sym(a, h)) * W + 2 * b_sym(c, w) + 1]; };   // copyMakeBorder(..., BORDER_REFLECT) per plane
    auto G2 = [&](int a, int c) { return bayer[(size_t)(2 * b_sym(a, h) + 1) * W + 2 * b_sym(c, w)]; };
    float g1c = G1(i, j), g2c = G2(i, j);
    float gr = delta_mix(G2(i - 1, j), g2c, G1(i, j - 1), g1c);                // red site   (eag.py:105-108)
    float gb = delta_mix(g1c, G1(i + 1, j), g2c, G2(i, j + 1));                // blue site  (eag.py:99-102)
    *reinterpret_cast<float2*>(out + (size_t)(2 * i) * W + 2 * j) = make_float2(gr, g1c);
    *reinterpret_cast<float2*>(out + (size_t)(2 * i + 1) * W + 2 * j) = make_float2(g2c, gb);
}
int launch_ca_green(hipStream_t st, const float* bayer, int H, int W, float* out) {
    hipLaunchKernelGGL(k_ca_green, dim3((W / 2 + 255) / 256, H / 2), dim3(256), 0, st, bayer, H, W, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

MosaicSrc mosaic_f32(const float* d_bayer) {
    MosaicSrc m;
    m.f32 = d_bayer; m.u16 = nullptr;
    for (int i = 0; i < 4; i++) { m.black[i] = 0.0f; m.sat[i] = 1.0f; }
    return m;
}
MosaicSrc mosaic_u16(const uint16_t* d_bayer, const float black[4], const float sat[4]) {
    MosaicSrc m;
    m.f32 = nullptr; m.u16 = d_bayer;
    for (int i = 0; i < 4; i++) { m.black[i] = black[i]; m.sat[i] = sat[i]; }
    return m;
}
