// demosaic_common.h -- device helpers shared by the AHD and EAG tile kernels:
// the four photosite-aware 3x3 kernels of get_rgbg_kernel (debayer/gaussian.py:19-54) applied as
// cv2.filter2D does (edge_assisted_gaussian.py:141,143), and g - GaussianBlur3(g) on a quad.
#pragma once
#include "devmath.h"

constexpr float GK0 = 0.45186276f, GK1 = 0.27406862f;   // GaussianBlur((3,3), 1.0) taps

// Mosaic source of the tile loaders: float32 sensor_scaled, or the raw uint16 mosaic with
// normalization.py:4-24 (clip(x - black_c, 0, sat_c) / sat_c, CFA sites indexed r,g1,b,g2) fused in
// (2 B/px of input traffic instead of 4).
struct MosaicSrc {
    const float* f32;
    const uint16_t* u16;
    float black[4], sat[4];
    double rsat[4];            // 1 / sat, correctly rounded to float64 on the host
};
// v / sat, the float32 division of bayer_normalize, without the ten-instruction IEEE sequence: float(double(v) * RN53(1 / sat)) IS the correctly
// rounded float32 quotient.  The product carries a relative error <= 2^-52; a quotient of two float32 numbers that is not itself a float32 lies
// at least 2^-49 (relative) away from every rounding boundary of the float32 grid (the boundary has a 25-bit odd significand M, and
// v - sat * M * 2^k is a non-zero multiple of the grid of a 49-bit product), and an exactly representable quotient is reproduced.
DEVI float div_by_sat(float v, double rsat) { return (float)((double)v * rsat); }
template <bool U16>
DEVI float load_mosaic(const MosaicSrc& m, size_t idx, int site) {
    if (U16) {
        float v = (float)m.u16[idx] - m.black[site];
        v = v < 0.0f ? 0.0f : (v > m.sat[site] ? m.sat[site] : v);
        return div_by_sat(v, m.rsat[site]);
    }
    return m.f32[idx];
}
// Two horizontally adjacent mosaic samples (even column first) as one 8-byte (f32) / 4-byte (u16) load.  odd_row selects the CFA sites:
// even row (R, G1) = levels 0, 1; odd row (G2, B) = levels 3, 2 (rawpy's R G B G order).  The levels are picked with selects between
// scalar registers -- indexing the kernel-argument arrays with a per-lane index makes the compiler read them from memory, per lane,
// with a wait after each (that cost the uint16 path 14 % of the select kernel when a third array joined the two).
// bayer_normalize (image.py:229) of one raw sample pair: (raw - black) clipped to [0, sat], / sat
DEVI float2 normalize_pair(const MosaicSrc& m, unsigned short rx, unsigned short ry, bool odd_row) {
    const float be = odd_row ? m.black[3] : m.black[0], bo = odd_row ? m.black[2] : m.black[1];
    const float se = odd_row ? m.sat[3] : m.sat[0], so = odd_row ? m.sat[2] : m.sat[1];
    const double re = odd_row ? m.rsat[3] : m.rsat[0], ro = odd_row ? m.rsat[2] : m.rsat[1];
    float a = (float)rx - be, b = (float)ry - bo;
    a = a < 0.0f ? 0.0f : (a > se ? se : a);
    b = b < 0.0f ? 0.0f : (b > so ? so : b);
    return make_float2(div_by_sat(a, re), div_by_sat(b, ro));
}
template <bool U16>
DEVI float2 load_mosaic_pair(const MosaicSrc& m, size_t idx, bool odd_row) {
    if (U16) {
        const ushort2 raw = *reinterpret_cast<const ushort2*>(m.u16 + idx);
        return normalize_pair(m, raw.x, raw.y, odd_row);
    }
    return *reinterpret_cast<const float2*>(m.f32 + idx);
}

struct Win3 { float v[3][3]; };

template <int STRIDE>
DEVI Win3 load_win(const float* plane, int gy, int gx) {
    Win3 w;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) w.v[r][c] = plane[(gy - 1 + r) * STRIDE + (gx - 1 + c)];
    return w;
}

// cv2.filter2D with the four photosite kernels of get_rgbg_kernel (gaussian.py:19-54): non-zero taps
// in row-major order, accumulated from 0.0f.  o[0..3] = TL, TR, BL, BR target pixel of the quad.
// 16 of the 25 taps are powers of two (1/64, 4/64, 16/64): their product with the sample is exact, so
// fmaf(k, v, s) rounds once to the very value "s + k*v" rounds to -- P2() spells those taps as one FMA
// (bit-identical, one instruction instead of two; only a product below 2^-126 could tell the two apart).
// Every sum starts from +0.0f like the oracle's, so an all-(-0) window still gives +0.
#define P2(k, v) s = __builtin_fmaf(k, v, s)
DEVI void filt_base_tl(const Win3& w, float o[4]) {  // base position TOP_LEFT (red)
    float s;
    s = 0.0f; P2(0.015625f, w.v[0][0]); s = s + 0.09375f * w.v[0][1]; P2(0.015625f, w.v[0][2]);
    s = s + 0.09375f * w.v[1][0]; s = s + 0.5625f * w.v[1][1]; s = s + 0.09375f * w.v[1][2];
    P2(0.015625f, w.v[2][0]); s = s + 0.09375f * w.v[2][1]; P2(0.015625f, w.v[2][2]);
    o[0] = s;
    s = 0.0f; P2(0.0625f, w.v[0][1]); P2(0.0625f, w.v[0][2]); s = s + 0.375f * w.v[1][1];
    s = s + 0.375f * w.v[1][2]; P2(0.0625f, w.v[2][1]); P2(0.0625f, w.v[2][2]);
    o[1] = s;
    s = 0.0f; P2(0.0625f, w.v[1][0]); s = s + 0.375f * w.v[1][1]; P2(0.0625f, w.v[1][2]);
    P2(0.0625f, w.v[2][0]); s = s + 0.375f * w.v[2][1]; P2(0.0625f, w.v[2][2]);
    o[2] = s;
    s = 0.0f; P2(0.25f, w.v[1][1]); P2(0.25f, w.v[1][2]); P2(0.25f, w.v[2][1]); P2(0.25f, w.v[2][2]);
    o[3] = s;
}
DEVI void filt_base_br(const Win3& w, float o[4]) {  // base position BOTTOM_RIGHT (blue)
    float s;
    s = 0.0f; P2(0.25f, w.v[0][0]); P2(0.25f, w.v[0][1]); P2(0.25f, w.v[1][0]); P2(0.25f, w.v[1][1]);
    o[0] = s;
    s = 0.0f; P2(0.0625f, w.v[0][0]); s = s + 0.375f * w.v[0][1]; P2(0.0625f, w.v[0][2]);
    P2(0.0625f, w.v[1][0]); s = s + 0.375f * w.v[1][1]; P2(0.0625f, w.v[1][2]);
    o[1] = s;
    s = 0.0f; P2(0.0625f, w.v[0][0]); P2(0.0625f, w.v[0][1]); s = s + 0.375f * w.v[1][0];
    s = s + 0.375f * w.v[1][1]; P2(0.0625f, w.v[2][0]); P2(0.0625f, w.v[2][1]);
    o[2] = s;
    s = 0.0f; P2(0.015625f, w.v[0][0]); s = s + 0.09375f * w.v[0][1]; P2(0.015625f, w.v[0][2]);
    s = s + 0.09375f * w.v[1][0]; s = s + 0.5625f * w.v[1][1]; s = s + 0.09375f * w.v[1][2];
    P2(0.015625f, w.v[2][0]); s = s + 0.09375f * w.v[2][1]; P2(0.015625f, w.v[2][2]);
    o[3] = s;
}
#undef P2

// The same two filter sets on PAIRS of planes at once (experiment: a plane pair stored interleaved, e.g. { green, colour difference } of one site type,
// loads as 8-byte LDS reads and filters as v_pk_mul / v_pk_add / v_pk_fma_f32: one half-rate instruction for two results instead of two
// full-rate ones; each lane of a packed instruction rounds exactly like the scalar instruction).
typedef float v2f __attribute__((ext_vector_type(2)));
struct Win3x2 { v2f v[3][3]; };
template <int STRIDE>
DEVI Win3x2 load_win2(const v2f* plane, int gy, int gx) {
    Win3x2 w;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) w.v[r][c] = plane[(gy - 1 + r) * STRIDE + (gx - 1 + c)];
    return w;
}
#define P2V(k, v) s = __builtin_elementwise_fma((v2f){k, k}, v, s)
DEVI void filt_base_tl2(const Win3x2& w, v2f o[4]) {  // base position TOP_LEFT (red)
    v2f s;
    s = (v2f){0.0f, 0.0f}; P2V(0.015625f, w.v[0][0]); s = s + 0.09375f * w.v[0][1]; P2V(0.015625f, w.v[0][2]);
    s = s + 0.09375f * w.v[1][0]; s = s + 0.5625f * w.v[1][1]; s = s + 0.09375f * w.v[1][2];
    P2V(0.015625f, w.v[2][0]); s = s + 0.09375f * w.v[2][1]; P2V(0.015625f, w.v[2][2]);
    o[0] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.0625f, w.v[0][1]); P2V(0.0625f, w.v[0][2]); s = s + 0.375f * w.v[1][1];
    s = s + 0.375f * w.v[1][2]; P2V(0.0625f, w.v[2][1]); P2V(0.0625f, w.v[2][2]);
    o[1] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.0625f, w.v[1][0]); s = s + 0.375f * w.v[1][1]; P2V(0.0625f, w.v[1][2]);
    P2V(0.0625f, w.v[2][0]); s = s + 0.375f * w.v[2][1]; P2V(0.0625f, w.v[2][2]);
    o[2] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.25f, w.v[1][1]); P2V(0.25f, w.v[1][2]); P2V(0.25f, w.v[2][1]); P2V(0.25f, w.v[2][2]);
    o[3] = s;
}
DEVI void filt_base_br2(const Win3x2& w, v2f o[4]) {  // base position BOTTOM_RIGHT (blue)
    v2f s;
    s = (v2f){0.0f, 0.0f}; P2V(0.25f, w.v[0][0]); P2V(0.25f, w.v[0][1]); P2V(0.25f, w.v[1][0]); P2V(0.25f, w.v[1][1]);
    o[0] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.0625f, w.v[0][0]); s = s + 0.375f * w.v[0][1]; P2V(0.0625f, w.v[0][2]);
    P2V(0.0625f, w.v[1][0]); s = s + 0.375f * w.v[1][1]; P2V(0.0625f, w.v[1][2]);
    o[1] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.0625f, w.v[0][0]); P2V(0.0625f, w.v[0][1]); s = s + 0.375f * w.v[1][0];
    s = s + 0.375f * w.v[1][1]; P2V(0.0625f, w.v[2][0]); P2V(0.0625f, w.v[2][1]);
    o[2] = s;
    s = (v2f){0.0f, 0.0f}; P2V(0.015625f, w.v[0][0]); s = s + 0.09375f * w.v[0][1]; P2V(0.015625f, w.v[0][2]);
    s = s + 0.09375f * w.v[1][0]; s = s + 0.5625f * w.v[1][1]; s = s + 0.09375f * w.v[1][2];
    P2V(0.015625f, w.v[2][0]); s = s + 0.09375f * w.v[2][1]; P2V(0.015625f, w.v[2][2]);
    o[3] = s;
}
#undef P2V

// g - GaussianBlur3(g) on the 4x4 full-resolution window around a quad (ahd.py:120-121)
DEVI void highpass_quad(const float W[4][4], float hf[4]) {
    float rb[4][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 2; k++) rb[r][k] = W[r][k + 1] * GK0 + (W[r][k] + W[r][k + 2]) * GK1;
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float bl = rb[rr + 1][k] * GK0 + (rb[rr][k] + rb[rr + 2][k]) * GK1;
            hf[rr * 2 + k] = W[rr + 1][k + 1] - bl;
        }
}

// ---- finished RGB pixels of a quad -> global memory ---------------------------------------------------------------
// Direct form: the quad's two rows as three 8-byte stores each (6 contiguous floats, 8-byte aligned because W is even).
DEVI void store_quad_direct(float* out, int W, int qi, int qj, const float px[4][3]) {
    // `out` is the (uniform) origin of the caller's tile and (qi, qj) the quad inside the tile: a 32-bit byte offset per lane, and the
    // stores take (scalar base, vector offset) -- no 64-bit vector arithmetic
    const unsigned rowbytes = (unsigned)W * 12u;
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
        float2* o = reinterpret_cast<float2*>(reinterpret_cast<char*>(out) + ((unsigned)(2 * qi + dy) * rowbytes + 24u * (unsigned)qj));
        o[0] = make_float2(px[2 * dy][0], px[2 * dy][1]);
        o[1] = make_float2(px[2 * dy][2], px[2 * dy + 1][0]);
        o[2] = make_float2(px[2 * dy + 1][1], px[2 * dy + 1][2]);
    }
}
// Staged form, for tiles that lie wholly inside the image when W % 4 == 0 (every image row then starts 16-byte aligned and
// a tile row, TPX * 12 bytes, is a whole number of 16-byte pieces): every thread drops its quad into an LDS image of the
// tile (8-byte LDS stores, lanes 24 bytes apart: conflict-free), then the workgroup streams the image out row by row with
// one 16-byte store per lane, consecutive lanes on consecutive addresses -- whole 128-byte lines per wave instruction
// instead of twelve dword stores per thread at a 12-byte stride.  `stage` may alias LDS that other threads still read:
// the leading barrier separates that use.  Must be called by every thread of the workgroup.
template <int TPX, int TPY, int NTHREADS>
DEVI void stage_tile_store(float* stage, float* out, int W, int y0, int x0, int lqy, int lqx, const float px[4][3]) {
    static_assert((TPX * 3) % 4 == 0, "a tile row is a whole number of 16-byte pieces");
    constexpr int ROW = TPX * 3, ROW4 = ROW / 4, N4 = ROW4 * TPY;
    __syncthreads();
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
        float2* o = reinterpret_cast<float2*>(stage + (2 * lqy + dy) * ROW + 6 * lqx);
        o[0] = make_float2(px[2 * dy][0], px[2 * dy][1]);
        o[1] = make_float2(px[2 * dy][2], px[2 * dy + 1][0]);
        o[2] = make_float2(px[2 * dy + 1][1], px[2 * dy + 1][2]);
    }
    __syncthreads();
    // addresses: one 64-bit tile origin (uniform: scalar registers) + a tile-local 32-bit byte offset per lane -- the store takes
    // the pair as (scalar base, vector offset), no 64-bit vector arithmetic (a tile spans < 2^32 bytes for any W the API accepts)
    char* const tile = reinterpret_cast<char*>(out + ((size_t)y0 * W + x0) * 3);
    const unsigned rowbytes = (unsigned)W * 12u;
#pragma unroll
    for (int k = 0; k < (N4 + NTHREADS - 1) / NTHREADS; k++) {
        const int idx = threadIdx.x + k * NTHREADS;
        if (idx < N4) {
            const int row = idx / ROW4, c4 = idx - row * ROW4;
            const float4 v = *reinterpret_cast<const float4*>(stage + row * ROW + 4 * c4);
            *reinterpret_cast<float4*>(tile + (mul24((unsigned)row, rowbytes) + 16u * (unsigned)c4)) = v;
        }
    }
}

