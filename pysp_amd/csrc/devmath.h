// devmath.h -- device-side arithmetic shared by the gfx950 kernels.
//
// Everything here is compiled with -ffp-contract=off and without fast-math: float32 expressions
// round after every operation exactly like the NumPy expressions they stand in for, and the only
// fused multiply-adds are the ones spelled __builtin_fmaf / __builtin_fma below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVI static __device__ __forceinline__

// ---- XCD-aware tile order --------------------------------------------------------------------
// MI355X hands consecutive workgroups to its 8 XCDs round-robin, and each XCD has its own L2: with the plain
// blockIdx -> tile map, neighbouring tiles (which share their halo rows and columns) always sit on different L2s.
// This map gives XCD k the k-th contiguous eighth of the row-major tile sequence instead, so a tile's neighbours --
// left/right, and one tile row up/down (a tile row of a 24 MP frame is ~1 MB, the L2 4 MB) -- are served by its own L2.
#ifndef PYSP_XCD_SWIZZLE
#define PYSP_XCD_SWIZZLE 1
#endif
DEVI void xcd_tile(int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
#if PYSP_XCD_SWIZZLE
    const unsigned gx = gridDim.x, n = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = lin & 7u, q = n >> 3, r = n & 7u;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);   // XCD k owns q (+1 if k < r) consecutive tiles
    by = (int)(t / gx); bx = (int)(t - (unsigned)by * gx);
#endif
}

// The same for a batch of frames in one grid (blockIdx.z = frame, gridDim.x x gridDim.y tiles each).  The hardware numbers the workgroups of the whole grid
// (z-major) and deals THAT number round-robin to the XCDs, so a frame whose tile count is no multiple of 8 starts on another XCD than the one before it: the
// map takes the workgroup's real XCD, g & 7 with g its number in the grid, and gives XCD k the k-th contiguous run of the frame's tiles -- as many tiles as
// the frame has workgroups on that XCD.
DEVI void xcd_tile_batch(int& bx, int& by) {
    const unsigned gx = gridDim.x, n = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    const unsigned g0 = blockIdx.z * n, g = g0 + lin, k = g & 7u;
    // workgroups of this frame on XCD c: the numbers g0 <= g' < g0 + n with g' & 7 == c; first_c = g0 + ((c - g0) & 7)
    const unsigned first = g0 + ((k - g0) & 7u), j = (g - first) >> 3;
    unsigned start = 0;
#pragma unroll
    for (unsigned c = 0; c < 7; c++) {
        const unsigned off = (c - g0) & 7u;                          // offset of XCD c's first workgroup inside the frame
        const unsigned cnt = off < n ? (n - off + 7u) >> 3 : 0u;
        start += c < k ? cnt : 0u;
    }
    const unsigned t = start + j;
    by = (int)(t / gx); bx = (int)(t - (unsigned)by * gx);
}

// The same, with the XCD's share walked in vertical strips of SW tiles: a kernel whose tiles overlap their upper / lower neighbours by many rows (the
// Lanczos warp: 11 of 27 source rows) finds those rows in its L2 only if the neighbour ran a moment ago -- a full tile row of a 100 MP frame is as large
// as the L2 itself.  Within a strip tiles run row-major, strips left to right; the last strip takes whatever width is left.
template <int SW>
DEVI void xcd_tile_strips(int& bx, int& by) {
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy, lin = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = lin & 7u, q = n >> 3, r = n & 7u;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
    const unsigned per = SW * gy, s = t / per, t2 = t - s * per, x0 = s * SW;
    const unsigned w = gx - x0 < (unsigned)SW ? gx - x0 : (unsigned)SW;
    const unsigned y = t2 / w;
    by = (int)y; bx = (int)(x0 + (t2 - y * w));
}

// ---- border index rules: cv2 BORDER_REFLECT / REFLECT_101 / REPLICATE ----------
DEVI int b_sym(int p, int n) {
    if (n == 1) return 0;
    while ((unsigned)p >= (unsigned)n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}
DEVI int b_101(int p, int n) {
    if (n == 1) return 0;
    while ((unsigned)p >= (unsigned)n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
DEVI int b_rep(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); }
// Branch-free single reflection, exact while p stays within n of the image (callers guarantee n >= 4 and
// |overshoot| <= 3, otherwise they take the loop forms above); the clamp only keeps far-outside,
// never-consumed positions inside the buffer.
DEVI int b_sym1(int p, int n) {
    int r = p < 0 ? -p - 1 : (p >= n ? 2 * n - 1 - p : p);
    return r < 0 ? 0 : (r >= n ? n - 1 : r);
}
DEVI int b_1011(int p, int n) {
    int r = p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p);
    return r < 0 ? 0 : (r >= n ? n - 1 : r);
}

// Clip to [0,1] as one v_med3_f32: on gfx950 compares/selects/min/max/med3 issue at half the rate of f32 add/mul/fma
// (4 vs 2 cycles per wave, tools/ubench_valu2.hip), so one med3 replaces four slow ops.  Identical to the compare
// form for every non-NaN input up to the sign of a zero result.  A (quiet) NaN comes out as 0 (v_med3 falls back to
// min3, and v_min returns its non-NaN operand), which is
//   * what the restated cv2.cvtColor wants: OpenCV clips its float input with max(.,0) / min(.,1) (clip01_cv), and
//   * NOT what np.clip does (transform.py:6-19): np.clip propagates NaN -> clip01_np below, or, in the fused colour
//     tail, one NaN test per pixel (the float64 matrix spreads a NaN of any channel to all three outputs anyway).
DEVI float clip01_cv(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
DEVI float clip01_np(float v) { float c = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); return v != v ? v : c; }

// ---- 3x3 colour matrix, transform.py:52-53: float64 accumulate in dgemm order, one rounding ----
struct Ccm { double m[9]; };
DEVI float ccm_row(const double* m, float r, float g, float b) {
    double t = (double)r * m[0];
    t = __builtin_fma((double)g, m[1], t);
    t = __builtin_fma((double)b, m[2], t);
    return (float)t;
}

// ---- restated cv2.cvtColor(RGB2LAB) float32 (ahd.py:58,62); bit-identical to oracle rgb2lab_px ----
// The sRGB decode ((v+0.055)/1.055)^2.4 and the cube root come from tables of quadratic segments indexed by
// the float's exponent and top mantissa bits (64 segments per octave over [2^-5,1], 32 per octave over
// [2^-7,2)); api.cpp builds them once per context (lab_tables.h) and the AHD kernel keeps a copy in LDS.
// The LDS copy is laid out so that the slot is a plain bit field of the float: slot = (bits >> S) & (SLOTS-1)
// (LAB_*_SLOTS in lab_tables.h), which makes every bit pattern -- also the ones whose lookup is discarded by
// the range select, and NaN -- address the table itself.  A lookup is then shift, and (address), one 16-byte LDS read
// (a, b, c and the segment start x_i), one exact subtraction and two FMAs.
#include "lab_tables.h"
struct LabTab { const float4* dec; const float4* cb; };
template <int NB, int SLOTS>
DEVI float lab_lut(const float4* tab, float x) {
    constexpr int S = 23 - NB;
    static_assert(S > 4, "the byte offset is taken with one shift");
    int bits = __float_as_int(x);
    float4 e = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + ((bits >> (S - 4)) & ((SLOTS - 1) << 4)));
    float fr = x - e.w;   // e.w = the segment's start x_i (the argument with its low S bits cleared): exact, and one integer op less than masking
    return __builtin_fmaf(__builtin_fmaf(e.z, fr, e.y), fr, e.x);
}
DEVI float lab_decode(LabTab t, float v) {
    v = clip01_cv(v);
    float p = lab_lut<LAB_DEC_NB, LAB_DEC_SLOTS>(t.dec, v);   // both sides are evaluated (no divergence)
    return v <= 0.04045f ? v * 0.07739938f : p;
}
DEVI float lab_f(LabTab tb, float t) {
    float c = lab_lut<LAB_CB_NB, LAB_CB_SLOTS>(tb.cb, t);
    return t > 0.008856f ? c : __builtin_fmaf(7.787f, t, 0.13793103f);
}
DEVI void rgb2lab_px(LabTab t, float R, float G, float B, float& L, float& a, float& b) {
    R = lab_decode(t, R); G = lab_decode(t, G); B = lab_decode(t, B);
    float X = __builtin_fmaf(B, 0.18982783f, __builtin_fmaf(G, 0.37621942f, R * 0.43395275f));
    float Y = __builtin_fmaf(B, 0.072169f, __builtin_fmaf(G, 0.71516f, R * 0.212671f));
    float Z = __builtin_fmaf(B, 0.87276554f, __builtin_fmaf(G, 0.109476522f, R * 0.017757915f));
    float fx = lab_f(t, X), fy = lab_f(t, Y), fz = lab_f(t, Z);
    L = Y > 0.008856f ? __builtin_fmaf(116.0f, fy, -16.0f) : 903.3f * Y;
    a = 500.0f * (fx - fy);
    b = 200.0f * (fy - fz);
}

// ---- the other restatement of the same call: OpenCV 4.10's default float32 RGB2Lab (LUT + fixed-point trilinear) ------------
// Bit-identical to oracle rgb2lab_px_cv410 / oracle/cv2_restated.py "cv410_lut" (restated from memory of color_lab.cpp,
// unpinned like the closed form).  Selected per context (pysp_ctx_set_lab_mode); tests/lab_flip_rate.py says what it changes.
//   c = cvRound(clip(v) * 2^14) per channel; cell t = c >> 9 (0..32), position f = (c >> 5) & 15;
//   33^3 int16 grid of (L, a, b) scaled to 14 bits; weights = products of three factors out of {16 - f, f};
//   acc = sum over the 8 corners, (acc + 2^11) >> 12;  L = acc * 100/2^14, a = acc * 256/2^14 - 128, b likewise.
// Device layout of the grid (api.cpp::host_cv410_device_lut): [34][34][34] entries of 64 bytes = one cache line holding all eight
// corners of the cell: for (dz, dy) = (0,0), (0,1), (1,0), (1,1) the int16 pairs { L(x), L(x+1) }, { a(x), a(x+1) }, { b(x), b(x+1) }
// (indices clamped to 32, where the matching weight is 0), 48 bytes + 16 of padding.  v_dot2_i32_i16 applies the two x weights in one
// instruction: three 16-byte loads from ONE line and 12 dot products per pixel on a 2.5 MB L2-resident grid.  The per-lane gathers are
// what this mode costs (a timing build with every lane on one line: select kernel 0.368 -> 0.333 ms); measured layouts: 16-byte entries
// (one (y, z) corner pair each, four lines per pixel) 0.3735 ms, 32-byte entries (one z level each, two lines) 0.366 ms, this one: DESIGN.md.
constexpr int CV410_DIM = 34;
DEVI int dot2_i16(unsigned v, int w, int acc) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, v), __builtin_bit_cast(s2, w), acc, false);
}
// v_mul_u32_u24 spelled out: the compiler turns __umul24 back into a plain multiply and then cannot prove (16 - f) | (f << 16) is a
// 24-bit value, so it picks v_mul_lo_u32, which issues at a quarter of the rate
DEVI unsigned mul24(unsigned a, unsigned b) { unsigned d; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// The three interpolated table values of one pixel, before the float scaling: l in [0, 2^14], a', b' in [0, 2^15) (entries are non-negative: pysp_ctx_set_lab_lut)
DEVI void rgb2lab_cv410_q(const uint4* __restrict__ lut, float R, float G, float B, int& aL, int& aa, int& ab) {
    // cvRound(clip(v) * 2^14) without a conversion: clip(v) * 2^14 + 1.5 * 2^23 is one exact-product FMA whose rounding IS round-half-even
    // to an integer, and the integer (<= 2^14) then sits in the low mantissa bits: cell = bits 9..14, position = bits 5..8.
    const unsigned bx = __float_as_uint(__builtin_fmaf(clip01_cv(R), 16384.0f, 12582912.0f));
    const unsigned by = __float_as_uint(__builtin_fmaf(clip01_cv(G), 16384.0f, 12582912.0f));
    const unsigned bz = __float_as_uint(__builtin_fmaf(clip01_cv(B), 16384.0f, 12582912.0f));
    const unsigned fx = (bx >> 5) & 15u, fy = (by >> 5) & 15u, fz = (bz >> 5) & 15u;
    const unsigned tx = (bx >> 9) & 63u, ty = (by >> 9) & 63u, tz = (bz >> 9) & 63u;
    // 24-bit multiplies throughout (v_mul_u32_u24 / v_mad_u32_u24 issue at the full rate, v_mul_lo_u32 at a quarter of it)
#ifdef LAB_ADDR_DOT2
    // cell index (tz * 34 + ty) * 34 + tx as ONE v_dot2_u32_u16 of (tz, ty) with (34 * 34, 34), tx as the accumulator (round 4: one multiplier-class instruction instead of two)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned cell = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, ((by << 7) & 0x3F0000u) | tz), __builtin_bit_cast(us2, (unsigned)(CV410_DIM * CV410_DIM) | ((unsigned)CV410_DIM << 16)), tx, false);
    const uint4* e = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lut) + (cell << 6));      // 32-bit byte offset (the table is 2.5 MB): (scalar base, vector offset) loads
#else
    const uint4* e = lut + 4u * (__umul24(__umul24(tz, CV410_DIM) + ty, CV410_DIM) + tx);
#endif
    const uint4 q0 = e[0], q1 = e[1], q2 = e[2];      // 12 dwords of one 64-byte line: (dz, dy) = (0,0), (0,1), (1,0), (1,1), each (L, a, b) as x pairs
    const unsigned wx = (16u - fx) | (fx << 16);                             // both x weights in one register; times <= 256 stays inside each half
    // six products instead of eight: the x pair times the two z factors first (<= 256 per half), then times the two y factors (<= 4096 per half)
    const unsigned wxz0 = mul24(wx, 16u - fz), wxz1 = mul24(wx, fz);
    const int w00 = (int)mul24(wxz0, 16u - fy), w10 = (int)mul24(wxz0, fy);   // (dy, dz)
    const int w01 = (int)mul24(wxz1, 16u - fy), w11 = (int)mul24(wxz1, fy);
    // CV_DESCALE's rounding constant 2^11 is the accumulators' start value (integer sums: any order, same bits)
    aL = dot2_i16(q2.y, w11, dot2_i16(q1.z, w01, dot2_i16(q0.w, w10, dot2_i16(q0.x, w00, 1 << 11))));
    aa = dot2_i16(q2.z, w11, dot2_i16(q1.w, w01, dot2_i16(q1.x, w10, dot2_i16(q0.y, w00, 1 << 11))));
    ab = dot2_i16(q2.w, w11, dot2_i16(q2.x, w01, dot2_i16(q1.y, w10, dot2_i16(q0.z, w00, 1 << 11))));
    aL >>= 12; aa >>= 12; ab >>= 12;
}
DEVI void rgb2lab_cv410(const uint4* __restrict__ lut, float R, float G, float B, float& L, float& a, float& b) {
    int aL, aa, ab;
    rgb2lab_cv410_q(lut, R, G, B, aL, aa, ab);
    L = (float)aL * (100.0f / 16384.0f);
    a = (float)aa * (256.0f / 16384.0f) - 128.0f;
    b = (float)ab * (256.0f / 16384.0f) - 128.0f;
}
// The same pixel with its chroma left as the two table integers, a' | b' << 16 (round 4): a = a'/64 - 128 and b = b'/64 - 128 are exact in float32, so
// are their differences, and the homogeneity vote (pyx:50-57) only ever uses differences of chroma values: k_ahd_select votes on the integers
// (v_pk_sub_i16 + v_dot2_i32_i16 per distance instead of two subtractions, two squares and a sum) -- see vote_quad_i16 for when that is exact.
DEVI void rgb2lab_cv410_pk(const uint4* __restrict__ lut, float R, float G, float B, float& L, unsigned& ab_pk) {
    int aL, aa, ab;
    rgb2lab_cv410_q(lut, R, G, B, aL, aa, ab);
    L = (float)aL * (100.0f / 16384.0f);
    ab_pk = (unsigned)aa | ((unsigned)ab << 16);
}

// ---- sRGB transfer curves, transform.py:89-111 -------------------------------------------------
// x ** (1/2.4) in the reference is a float32 power with the float32 exponent 0.41666666; the oracle
// returns the correctly rounded value of that power, and so does this routine, without a float64 pow:
//   z0 ~ x^(-7/12) from the hardware log2/exp2 approximations (any ~1e-5 accurate seed works),
//   one float64 Newton-type correction on z^12 * x^7 = (x z^2)^6 x = 1 (error ~ r^3/32, r ~ 1e-5 -> < 1e-15),
//   y = x*z = x^(5/12), then the first-order factor for the exponent difference 0.41666666f - 5/12.
// Relative error ~6e-15 (measured).  Exhaustive GPU sweep (tests/test_gpu_round2.py::test_srgb_curve_exhaustive): the encoded
// value equals the oracle's (float64 pow, rounded once) on EVERY float32 of [0, 1] -- 1,065,357,312 inputs, 0 differences.
// (Measured and dropped in round 2: the same power from an LDS table, x = 2^k m, A[k] * degree-6 polynomial per 1/32 of the
// mantissa in float64, 7 float64 operations instead of 12 and no transcendentals -- also bit-identical on every input, but the
// per-lane table reads (56 bytes per value; LDS is shared by the CU's four SIMDs) made every kernel slower: median stage
// 0.341 -> 0.349 ms, EAG + sRGB 0.135 -> 0.149 ms, Draft + sRGB 0.060 -> 0.072 ms, with 72-byte entries on distinct bank
// pairs; 0.360 / 0.165 / 0.084 ms with 64-byte entries, eight of which share a bank set.)
DEVI float srgb_pow_5_12(float x) {   // x in [0.003, 1]
    float l2 = __builtin_amdgcn_logf(x);
    float z0 = __builtin_amdgcn_exp2f(-0.5833333f * l2);
    double xd = (double)x, zd = (double)z0;
#ifndef SRGB_ROUND2_CHAIN
    // y0 = x z0 (exact: two float32 factors) serves twice, as the value to correct and on the way to x z0^2 (round 3: 14 instead of 15 float64-class
    // instructions per value, -3 % on EAG + sRGB)
    double y0 = xd * zd, w = y0 * zd, w2 = w * w, w6 = (w2 * w2) * w2;   // x^7 z^12 = (x z^2)^6 x
    double r = __builtin_fma(w6, xd, -1.0);
    double c = r * __builtin_fma(r, 13.0 / 288.0, -1.0 / 12.0);
    double y = __builtin_fma(y0, c, y0);
#else
    double w = xd * (zd * zd), w2 = w * w, w6 = (w2 * w2) * w2;
    double r = __builtin_fma(w6, xd, -1.0);
    double c = r * __builtin_fma(r, 13.0 / 288.0, -1.0 / 12.0);
    double z = __builtin_fma(zd, c, zd);
    double y = xd * z;
#endif
    // (0.41666666f - 5/12) * ln 2 = -9.934107462565104e-09 * 0.6931471805599453
    // (the factor as a float32 product, converted once, instead of a conversion and a float64 multiply: its rounding error, 2^-24 of a term below 6e-8,
    // stays inside the budget -- the exhaustive sweep still finds 0 differences on all 1,065,357,312 inputs, round 3.  Folding this term into c,
    // y0 (1 + c + e), saves two more instructions and drops the cross term c e < 1e-14: 2 of the 1,065,357,312 inputs then round the other way -- not taken;
    // without the quadratic term of c as well: 31 inputs)
    y = __builtin_fma(y, (double)(l2 * -6.885798579082628e-09f), y);
    return (float)y;
}
// NANS: the argument may be NaN and np.where(NaN <= t, ., 1.055 * NaN**e - 0.055) = NaN has to come out (stand-alone
// lin_srgb_to_srgb on caller data); the fused colour tail handles NaN once per pixel and passes false.
template <bool NANS = true>
DEVI float srgb_encode(float x) {
    x = NANS ? clip01_np(x) : clip01_cv(x);
    float p = srgb_pow_5_12(x);   // x <= 0.0031308 (incl. 0 -> inf/NaN inside) is discarded by the select; NaN -> NaN
    return x <= 0.0031308f ? x * 12.92f : 1.055f * p - 0.055f;
}
DEVI float srgb_decode(float x) {
    x = clip01_np(x);
    float u = (x + 0.055f) / 1.055f;
    float p = (float)pow((double)u, (double)2.4f);
    return x <= 0.04045f ? x / 12.92f : p;
}

// ---- colour tail applied to a camera-RGB pixel at the end of a fused pipeline -------------------
//   tail 0: none (RawDemosaicData.image)           tail 1: to_lin_srgb (clip + CCM)
//   tail 2: to_lin_srgb + lin_srgb_to_srgb          tail 3: ... with README.md:157 x/(1+x) in between
// Non-finite input (e.g. the 0/0 a zero flat field leaves in the mosaic, raw_correction.py:45): np.clip keeps a NaN, the
// float64 np.dot then makes all three outputs of that pixel NaN, and nothing downstream turns a NaN back into a number;
// +-Inf clip to 1 / 0 like any other value.  So one test per pixel decides: two unordered compares, and the outputs of a
// NaN pixel are replaced by NaN at the end (the arithmetic in between runs on the med3-clipped zeros and is discarded).
DEVI void colour_tail(int tail, const double* M, float& r, float& g, float& b) {
    if (tail == 0) return;
    const bool has_nan = __builtin_isunordered(r, g) | (b != b);
    float cr = clip01_cv(r), cg = clip01_cv(g), cb = clip01_cv(b);
    r = ccm_row(M, cr, cg, cb);
    g = ccm_row(M + 3, cr, cg, cb);
    b = ccm_row(M + 6, cr, cg, cb);
    if (tail >= 2) {
        if (tail == 3) { r = r / (1.0f + r); g = g / (1.0f + g); b = b / (1.0f + b); }
        r = srgb_encode<false>(r); g = srgb_encode<false>(g); b = srgb_encode<false>(b);
    }
    const float qnan = __int_as_float(0x7fc00000);
    r = has_nan ? qnan : r; g = has_nan ? qnan : g; b = has_nan ? qnan : b;
}
