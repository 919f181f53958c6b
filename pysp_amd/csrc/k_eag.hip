// k_eag.hip -- Edge-Assisted-Gaussian ("Fast") and Draft demosaic for gfx950.
//
// EAG follows debayer/edge_assisted_gaussian.py:10-201 of pySP; Draft follows
// debayer/fast_resize.py:7-44 (+ cv2.resize INTER_LINEAR x2 restated as in oracle/pysp_oracle.c).
//
// EAG kernel: one workgroup = one 64x32 px output tile = 32x16 CFA quads, one thread per quad.
//   P0  raw mosaic -> four de-interleaved quarter planes in LDS (halo 2 quads, symmetric border)
//   P1  gradient-weighted green at R/B sites * wb[1], colour differences D = sub*wb - g
//       (halo 1 quad; outside the image: REFLECT_101 of the quarter plane), stored as PAIRS { g, D } per site type: the 3x3 windows load as
//       8-byte LDS reads and both photosite filter sets run as v_pk_mul / v_pk_add / v_pk_fma_f32 (round 3: -1.6 %)
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

// edge_assisted_gaussian.py:36-49
DEVI float delta_mix(float top, float bottom, float left, float right) {
    float dy = fabsf(top - bottom), dx = fabsf(left - right), s = dy + dx;
    float ax = (left + right) / 2.0f, ay = (top + bottom) / 2.0f;
    float sy = s != 0.0f ? dy / s : 0.5f;
    float sx = 1.0f - sy;
    return ay * sx + ax * sy;
}
}  // namespace

// P0 of both tile kernels: raw mosaic -> four de-interleaved quarter planes in LDS, halo 2 quads.  Slots outside the image
// hold the BORDER_REFLECT (edge-duplicating) sample of their plane (eag.py:86-87).  mosaic_prefetch issues every global
// load of a thread into registers (one 16-byte slot = two horizontally adjacent quads of one mosaic row), so they are in
// flight together; mosaic_commit drops them into LDS.  Interior tiles with 16-byte aligned rows load a slot as ONE 16-byte access (a tile's first column,
// 2 * (tq0x - 2), is a multiple of 4 floats); border tiles and uint16 mosaics take two pair loads with the border rule (an 8-byte uint16
// slot load on interior tiles was measured: no change, 0.100 ms for EAG + CCM from uint16 either way).
constexpr int NSLOT4 = 2 * MWY * (MWX / 2), NL4 = (NSLOT4 + NT - 1) / NT;
static_assert(MWX % 2 == 0 && TQX % 2 == 0, "a mosaic tile row is a whole number of 16-byte slots");
template <bool TINY, bool U16>
DEVI void mosaic_prefetch(const MosaicSrc& src, float4 tmp[NL4], int tid, int W, int h, int w, int tq0y, int tq0x, bool inside) {
    if (!U16 && inside && !(W & 3)) {
        // uniform 64-bit tile origin + tile-local 32-bit byte offset per lane (scalar base, vector offset: no 64-bit vector arithmetic)
        const char* const tile = reinterpret_cast<const char*>(src.f32 + (size_t)(2 * (tq0y - 2)) * W + 2 * (tq0x - 2));
        const unsigned rowbytes = (unsigned)W * 4u;
#pragma unroll
        for (int k = 0; k < NL4; k++) {
            int idx = tid + k * NT;
            if (idx >= NSLOT4) idx = NSLOT4 - 1;
            int ry = idx / (MWX / 2), m2 = idx - ry * (MWX / 2);
            tmp[k] = *reinterpret_cast<const float4*>(tile + (mul24((unsigned)ry, rowbytes) + 16u * (unsigned)m2));
        }
    } else {
#pragma unroll
        for (int k = 0; k < NL4; k++) {
            int idx = tid + k * NT;
            if (idx >= NSLOT4) idx = NSLOT4 - 1;
            int ry = idx / (MWX / 2), m2 = idx - ry * (MWX / 2), my = ry >> 1, dy = ry & 1;
            int qi = tq0y - 2 + my, qj0 = tq0x - 2 + 2 * m2, qj1 = qj0 + 1;
            if (!inside) {                                                     // uniform per tile: interior tiles skip the border rules
                qi = TINY ? b_sym(qi, h) : b_sym1(qi, h);
                qj0 = TINY ? b_sym(qj0, w) : b_sym1(qj0, w);
                qj1 = TINY ? b_sym(qj1, w) : b_sym1(qj1, w);
            }
            float2 lo = load_mosaic_pair<U16>(src, (size_t)(2 * qi + dy) * W + 2 * qj0, dy != 0);
            float2 hi = load_mosaic_pair<U16>(src, (size_t)(2 * qi + dy) * W + 2 * qj1, dy != 0);
            tmp[k] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}
DEVI void mosaic_commit(float (*mw)[MWY][MWX], const float4 tmp[NL4], int tid) {
#pragma unroll
    for (int k = 0; k < NL4; k++) {
        int idx = tid + k * NT;
        if (idx < NSLOT4) {
            int ry = idx / (MWX / 2), m2 = idx - ry * (MWX / 2), my = ry >> 1, dy = ry & 1;
            // even row: (R, G1, R, G1) ; odd row: (G2, B, G2, B)
            *reinterpret_cast<float2*>(&mw[dy ? P_G2 : P_R][my][2 * m2]) = make_float2(tmp[k].x, tmp[k].z);
            *reinterpret_cast<float2*>(&mw[dy ? P_B : P_G1][my][2 * m2]) = make_float2(tmp[k].y, tmp[k].w);
        }
    }
}
constexpr int FRAME_BATCH_MAX = 16;
struct FrameBatch { const void* src[FRAME_BATCH_MAX]; float* out[FRAME_BATCH_MAX]; };      // frames of one launch of k_eag_batch / k_draft_batch
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
DEVI void eag_tile(const EagParams& p, const int tbx, const int tby) {
    // one LDS block: the raw planes and the green / difference planes first, then (after a barrier) the finished RGB tile,
    // which leaves the workgroup as whole 16-byte stores (stage_tile_store)
    constexpr int NPLANES = 4 * MWY * MWX + 4 * GY * GX, NSTAGE = (2 * TQY) * (2 * TQX * 3);
    __shared__ __attribute__((aligned(16))) float lds[NPLANES > NSTAGE ? NPLANES : NSTAGE];
    float (*mw)[MWY][MWX] = reinterpret_cast<float (*)[MWY][MWX]>(lds);
    v2f (*gq2)[GY][GX] = reinterpret_cast<v2f (*)[GY][GX]>(lds + 4 * MWY * MWX);      // [0] = { green at R sites, R - green }, [1] = the same at B sites
    static_assert((4 * MWY * MWX) % 2 == 0, "8-byte aligned pairs");
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    const int tq0x = tbx * TQX, tq0y = tby * TQY;
    const bool inside = tq0y >= 2 && tq0x >= 2 && tq0y + TQY + 2 <= h && tq0x + TQX + 2 <= w;   // no border rule applies in P0/P1
    // P0: raw planes, cv2.copyMakeBorder(..., BORDER_REFLECT) per plane (eag.py:86-87).
    // (Measured and dropped: a workgroup looping over 2-16 consecutive tiles with the next tile's loads prefetched into registers
    // during the compute phases: no gain at any tile count -- the kernel does not wait for bytes in flight -- and the loop form
    // costs registers, 43 -> 79 VGPRs, 4 -> 3 workgroups per CU, +35 % time.)
    {
        float4 pre[NL4];
        mosaic_prefetch<TINY, U16>(p.src, pre, tid, W, h, w, tq0y, tq0x, inside);
        mosaic_commit(mw, pre, tid);
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
        gq2[0][gy][gx] = (v2f){gr, mw[P_R][a][c] * p.wb[0] - gr};
        gq2[1][gy][gx] = (v2f){gb, mw[P_B][a][c] * p.wb[2] - gb};
    }
    __syncthreads();

    // P2
    const int lqy = tid / TQX, lqx = tid - lqy * TQX;
    const int qi = tq0y + lqy, qj = tq0x + lqx;
    const bool live = qi < h && qj < w;
    // whole tile inside the image and rows 16-byte aligned: the tile goes out through LDS as 16-byte stores
    const bool staged = !(W & 3) && tq0y + TQY <= h && tq0x + TQX <= w;
    float px[4][3];
    if (live) {
    const int gy = lqy + 1, gx = lqx + 1, my = lqy + 2, mx = lqx + 2;
    const bool at_top = qi == 0, at_bot = qi == h - 1, at_left = qj == 0, at_right = qj == w - 1;
    const float wg = p.wb[1];
    const Win3x2 wr2 = load_win2<GX>(&gq2[0][0][0], gy, gx), wb2 = load_win2<GX>(&gq2[1][0][0], gy, gx);
    Win3 wgr, wgb;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) { wgr.v[r][c] = wr2.v[r][c].x; wgb.v[r][c] = wb2.v[r][c].x; }
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
    float hf[4], rr[4], bb[4];
    highpass_quad(Wn, hf);                                           // eag.py:156
#ifndef EAG_SCALAR_FILTERS
    {
        v2f o2[4];
        filt_base_tl2(wr2, o2);
#pragma unroll
        for (int k = 0; k < 4; k++) rr[k] = o2[k].y + (o2[k].x + hf[k]);     // eag.py:141,143
        filt_base_br2(wb2, o2);
#pragma unroll
        for (int k = 0; k < 4; k++) bb[k] = o2[k].y + (o2[k].x + hf[k]);
    }
#else
    {   // experiment (round 4): the paired LDS layout with SCALAR filter arithmetic -- v_pk_*_f32 belongs to the multiplier family that costs 5-8 cycles
        // inside a float32 stream (profiles/r4_ubench_pairs.log); same values, same rounding
        Win3 wdr, wdb;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) { wdr.v[r][c] = wr2.v[r][c].y; wdb.v[r][c] = wb2.v[r][c].y; }
        float fg[4], fd[4];
        filt_base_tl(wgr, fg); filt_base_tl(wdr, fd);
#pragma unroll
        for (int k = 0; k < 4; k++) rr[k] = fd[k] + (fg[k] + hf[k]);
        filt_base_br(wgb, fg); filt_base_br(wdb, fd);
#pragma unroll
        for (int k = 0; k < 4; k++) bb[k] = fd[k] + (fg[k] + hf[k]);
    }
#endif
    float gg[4] = {wgr.v[1][1], g1_c, g2_c, wgb.v[1][1]};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float r = rr[k], g = gg[k], b = bb[k];
        colour_tail(TAIL, p.ccm.m, r, g, b);
        px[k][0] = r; px[k][1] = g; px[k][2] = b;
    }
    if (!staged) store_quad_direct(p.out + ((size_t)(2 * tq0y) * W + 2 * tq0x) * 3, W, lqy, lqx, px);
    }
    if (staged) stage_tile_store<2 * TQX, 2 * TQY, NT>(lds, p.out, W, 2 * tq0y, 2 * tq0x, lqy, lqx, px);
}
template <bool TINY, bool U16, int TAIL>
__global__ void __launch_bounds__(NT) k_eag(EagParams p) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    eag_tile<TINY, U16, TAIL>(p, tbx, tby);
}
// A batch of frames of one size in ONE grid (blockIdx.z = frame; BASELINE config 3: eag of 8 frames per rank and step): no launch boundary and no
// drain / fill between the frames of a batch.  The frames' pointers travel in the kernel arguments.
template <bool U16, int TAIL>
__global__ void __launch_bounds__(NT) k_eag_batch(EagParams p, FrameBatch b) {
    int tbx, tby;
    xcd_tile_batch(tbx, tby);
    const unsigned z = blockIdx.z;
    if (U16) p.src.u16 = reinterpret_cast<const uint16_t*>(b.src[z]); else p.src.f32 = reinterpret_cast<const float*>(b.src[z]);
    p.out = b.out[z];
    eag_tile<false, U16, TAIL>(p, tbx, tby);
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

// Frames of one size, float32 or uint16 mosaics, in grids of up to FRAME_BATCH_MAX frames (quarter planes of at least 4 x 4: the caller checks)
int launch_eag_batch(hipStream_t st, const void* const* d_srcs, int u16, const float black[4], const float sat[4], int n, int H, int W, const float wb[3], const double M[9],
                     int tail, float* const* d_outs, Timeline* tl) {
    if (n <= 0 || H / 2 < 4 || W / 2 < 4 || tail < 0 || tail > 3) return -1;
    EagParams a;
    a.src = u16 ? mosaic_u16(nullptr, black, sat) : mosaic_f32(nullptr);
    a.out = nullptr; a.H = H; a.W = W; a.tail = tail;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M ? M[i] : (i % 4 == 0 ? 1.0 : 0.0);
    for (int f0 = 0; f0 < n; f0 += FRAME_BATCH_MAX) {
        const int nb = n - f0 < FRAME_BATCH_MAX ? n - f0 : FRAME_BATCH_MAX;
        FrameBatch b = {};
        for (int i = 0; i < nb; i++) { b.src[i] = d_srcs[f0 + i]; b.out[i] = d_outs[f0 + i]; }
        const dim3 g((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY, nb);
        if (tl) tl->begin(st, "k_eag");
#define EAGB(T) do { if (u16) hipLaunchKernelGGL((k_eag_batch<true, T>), g, dim3(NT), 0, st, a, b); else hipLaunchKernelGGL((k_eag_batch<false, T>), g, dim3(NT), 0, st, a, b); } while (0)
        switch (tail) { case 0: EAGB(0); break; case 1: EAGB(1); break; case 2: EAGB(2); break; default: EAGB(3); break; }
#undef EAGB
        if (tl) tl->end(st);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ================================================================================================
// Draft (fast_resize.py:21-39): quarter-resolution RGB with 3/4-1/4 diagonal R/B alignment, then
// bilinear x2 (half-pixel centres, edge clamp; horizontal pass then vertical pass).
//
// One workgroup = one 64x32 px output tile = 32x16 quads, one thread per quad (the EAG tile):
//   P0  raw mosaic -> four de-interleaved quarter planes in LDS (halo 2 quads), 16-byte loads on interior tiles
//   P1  quarter-resolution RGB q (halo 1 quad), positions outside the image take the clamped position (cv2.resize
//       clamps its taps to the image)
//   P2  per quad: the four bilinear outputs from q, colour tail, tile out through LDS as 16-byte stores
namespace {
// cv2.resize INTER_LINEAR x2 along one axis for output position X of n input samples: taps s0, s1 and weights 1-f, f
DEVI void lin_tap(int X, int n, int& s0, int& s1, float& a0, float& a1) {
    float f = ((float)X + 0.5f) * 0.5f - 0.5f;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= n - 1) { s = n - 1; f = 0.0f; }
    s0 = s; s1 = s + 1 < n ? s + 1 : s; a0 = 1.0f - f; a1 = f;
}
constexpr int DQX = TQX + 2, DQY = TQY + 2;     // q planes, halo 1 quad
}  // namespace

template <bool TINY, bool U16, int TAIL>
DEVI void draft_tile(const EagParams& p, const int tbx, const int tby) {
    constexpr int NPLANES = 4 * MWY * MWX + 3 * DQY * DQX, NSTAGE = (2 * TQY) * (2 * TQX * 3);
    __shared__ __attribute__((aligned(16))) float lds[NPLANES > NSTAGE ? NPLANES : NSTAGE];
    float (*mw)[MWY][MWX] = reinterpret_cast<float (*)[MWY][MWX]>(lds);
    float (*q)[DQY][DQX] = reinterpret_cast<float (*)[DQY][DQX]>(lds + 4 * MWY * MWX);
    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    const int tq0x = tbx * TQX, tq0y = tby * TQY;
    const bool inside = tq0y >= 2 && tq0x >= 2 && tq0y + TQY + 2 <= h && tq0x + TQX + 2 <= w;   // no clamp applies anywhere in the tile
    {
        float4 pre[NL4];
        mosaic_prefetch<TINY, U16>(p.src, pre, tid, W, h, w, tq0y, tq0x, inside);
        mosaic_commit(mw, pre, tid);
    }
    __syncthreads();

    // P1: q = (R, G, B) at quarter resolution (fast_resize.py:28-37): r padded bottom/right, b padded top/left (REFLECT by one = the edge sample)
    for (int idx = tid; idx < DQY * DQX; idx += NT) {
        int gy = idx / DQX, gx = idx - gy * DQX;
        int a = gy + 1, c = gx + 1, a1 = a + 1, c1 = c + 1, a0 = a - 1, c0 = c - 1;            // slots in the staged planes (origin tq0 - 2)
        if (!inside) {
            int qi = b_rep(tq0y - 1 + gy, h), qj = b_rep(tq0x - 1 + gx, w);
            a = qi - (tq0y - 2); c = qj - (tq0x - 2);
            a1 = b_rep(qi + 1, h) - (tq0y - 2); c1 = b_rep(qj + 1, w) - (tq0x - 2);
            a0 = b_rep(qi - 1, h) - (tq0y - 2); c0 = b_rep(qj - 1, w) - (tq0x - 2);
            if (a0 < 0 || a1 > MWY - 1 || c0 < 0 || c1 > MWX - 1) continue;                      // beyond a partial tile: never consumed
        }
        q[0][gy][gx] = (0.75f * mw[P_R][a][c] + 0.25f * mw[P_R][a1][c1]) * p.wb[0];
        q[1][gy][gx] = ((mw[P_G1][a][c] + mw[P_G2][a][c]) / 2.0f) * p.wb[1];
        q[2][gy][gx] = (0.75f * mw[P_B][a][c] + 0.25f * mw[P_B][a0][c0]) * p.wb[2];
    }
    __syncthreads();

    // P2
    const int lqy = tid / TQX, lqx = tid - lqy * TQX;
    const int qi = tq0y + lqy, qj = tq0x + lqx;
    const bool live = qi < h && qj < w;
    const bool staged = !(W & 3) && tq0y + TQY <= h && tq0x + TQX <= w;
    float px[4][3];
    if (live) {
        // taps of the two output columns / rows of this quad, as slots of the q planes (origin tq0 - 1)
        int sx[2][2], sy[2][2];
        float ax[2][2], ay[2][2];
        if (inside) {      // interior: X = 2j -> (j-1, j) weights (1/4, 3/4); X = 2j+1 -> (j, j+1) weights (3/4, 1/4); lin_tap gives exactly these
            sx[0][0] = lqx; sx[0][1] = lqx + 1; sx[1][0] = lqx + 1; sx[1][1] = lqx + 2;
            sy[0][0] = lqy; sy[0][1] = lqy + 1; sy[1][0] = lqy + 1; sy[1][1] = lqy + 2;
            ax[0][0] = ay[0][0] = 0.25f; ax[0][1] = ay[0][1] = 0.75f; ax[1][0] = ay[1][0] = 0.75f; ax[1][1] = ay[1][1] = 0.25f;
        } else {
#pragma unroll
            for (int d = 0; d < 2; d++) {
                lin_tap(2 * qj + d, w, sx[d][0], sx[d][1], ax[d][0], ax[d][1]);
                lin_tap(2 * qi + d, h, sy[d][0], sy[d][1], ay[d][0], ay[d][1]);
                sx[d][0] -= tq0x - 1; sx[d][1] -= tq0x - 1; sy[d][0] -= tq0y - 1; sy[d][1] -= tq0y - 1;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dy = k >> 1, dx = k & 1;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float h0 = q[c][sy[dy][0]][sx[dx][0]] * ax[dx][0] + q[c][sy[dy][0]][sx[dx][1]] * ax[dx][1];
                float h1 = q[c][sy[dy][1]][sx[dx][0]] * ax[dx][0] + q[c][sy[dy][1]][sx[dx][1]] * ax[dx][1];
                px[k][c] = h0 * ay[dy][0] + h1 * ay[dy][1];
            }
            colour_tail(TAIL, p.ccm.m, px[k][0], px[k][1], px[k][2]);
        }
        if (!staged) store_quad_direct(p.out + ((size_t)(2 * tq0y) * W + 2 * tq0x) * 3, W, lqy, lqx, px);
    }
    if (staged) stage_tile_store<2 * TQX, 2 * TQY, NT>(lds, p.out, W, 2 * tq0y, 2 * tq0x, lqy, lqx, px);
}
template <bool TINY, bool U16, int TAIL>
__global__ void __launch_bounds__(NT) k_draft(EagParams p) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    draft_tile<TINY, U16, TAIL>(p, tbx, tby);
}
// A batch of frames of one size in ONE grid (blockIdx.z = frame; BASELINE config 3: draft of 8 frames per rank and step): no launch boundary and no
// drain / fill between the frames of a batch.  The frames' pointers travel in the kernel arguments.
template <bool U16, int TAIL>
__global__ void __launch_bounds__(NT) k_draft_batch(EagParams p, FrameBatch b) {
    int tbx, tby;
    xcd_tile_batch(tbx, tby);
    const unsigned z = blockIdx.z;
    if (U16) p.src.u16 = reinterpret_cast<const uint16_t*>(b.src[z]); else p.src.f32 = reinterpret_cast<const float*>(b.src[z]);
    p.out = b.out[z];
    draft_tile<false, U16, TAIL>(p, tbx, tby);
}


int launch_draft(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int tail, float* d_out, Timeline* tl) {
    EagParams a;
    a.src = src; a.out = d_out; a.H = H; a.W = W; a.tail = tail;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M ? M[i] : (i % 4 == 0 ? 1.0 : 0.0);
    dim3 g((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY);
    if (tl) tl->begin(st, "k_draft");
    const bool tiny = H / 2 < 4 || W / 2 < 4, u16 = src.u16 != nullptr;
#define DRAFT_LAUNCH(T) \
    do { \
        if (tiny && u16) hipLaunchKernelGGL((k_draft<true, true, T>), g, dim3(NT), 0, st, a); \
        else if (tiny) hipLaunchKernelGGL((k_draft<true, false, T>), g, dim3(NT), 0, st, a); \
        else if (u16) hipLaunchKernelGGL((k_draft<false, true, T>), g, dim3(NT), 0, st, a); \
        else hipLaunchKernelGGL((k_draft<false, false, T>), g, dim3(NT), 0, st, a); \
    } while (0)
    switch (tail) {
        case 0: DRAFT_LAUNCH(0); break;
        case 1: DRAFT_LAUNCH(1); break;
        case 2: DRAFT_LAUNCH(2); break;
        case 3: DRAFT_LAUNCH(3); break;
        default: return -1;
    }
#undef DRAFT_LAUNCH
    if (tl) tl->end(st);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_draft_batch(hipStream_t st, const void* const* d_srcs, int u16, const float black[4], const float sat[4], int n, int H, int W, const float wb[3], const double M[9],
                       int tail, float* const* d_outs, Timeline* tl) {
    if (n <= 0 || H / 2 < 4 || W / 2 < 4 || tail < 0 || tail > 3) return -1;
    EagParams a;
    a.src = u16 ? mosaic_u16(nullptr, black, sat) : mosaic_f32(nullptr);
    a.out = nullptr; a.H = H; a.W = W; a.tail = tail;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M ? M[i] : (i % 4 == 0 ? 1.0 : 0.0);
    for (int f0 = 0; f0 < n; f0 += FRAME_BATCH_MAX) {
        const int nb = n - f0 < FRAME_BATCH_MAX ? n - f0 : FRAME_BATCH_MAX;
        FrameBatch b = {};
        for (int i = 0; i < nb; i++) { b.src[i] = d_srcs[f0 + i]; b.out[i] = d_outs[f0 + i]; }
        const dim3 g((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY, nb);
        if (tl) tl->begin(st, "k_draft");
#define DRAFTB(T) do { if (u16) hipLaunchKernelGGL((k_draft_batch<true, T>), g, dim3(NT), 0, st, a, b); else hipLaunchKernelGGL((k_draft_batch<false, T>), g, dim3(NT), 0, st, a, b); } while (0)
        switch (tail) { case 0: DRAFTB(0); break; case 1: DRAFTB(1); break; case 2: DRAFTB(2); break; default: DRAFTB(3); break; }
#undef DRAFTB
        if (tl) tl->end(st);
    }
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
    for (int i = 0; i < 4; i++) { m.black[i] = 0.0f; m.sat[i] = 1.0f; m.rsat[i] = 1.0; }
    return m;
}
MosaicSrc mosaic_u16(const uint16_t* d_bayer, const float black[4], const float sat[4]) {
    MosaicSrc m;
    m.f32 = nullptr; m.u16 = d_bayer;
    for (int i = 0; i < 4; i++) { m.black[i] = black[i]; m.sat[i] = sat[i]; m.rsat[i] = 1.0 / (double)sat[i]; }
    return m;
}
