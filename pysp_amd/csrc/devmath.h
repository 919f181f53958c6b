// devmath.h -- device-side arithmetic shared by the gfx950 kernels.
//
// Everything here is compiled with -ffp-contract=off and without fast-math: float32 expressions
// round after every operation exactly like the NumPy expressions they stand in for, and the only
// fused multiply-adds are the ones spelled __builtin_fmaf / __builtin_fma below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVI static __device__ __forceinline__

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

DEVI float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }  // transform.py:6-19

// ---- 3x3 colour matrix, transform.py:52-53: float64 accumulate in dgemm order, one rounding ----
struct Ccm { double m[9]; };
DEVI float ccm_row(const double* m, float r, float g, float b) {
    double t = (double)r * m[0];
    t = __builtin_fma((double)g, m[1], t);
    t = __builtin_fma((double)b, m[2], t);
    return (float)t;
}

// ---- restated cv2.cvtColor(RGB2LAB) float32 (ahd.py:58,62); bit-identical to oracle rgb2lab_px ----
DEVI float lab_pow24(float u) {
    float t = __int_as_float(0x4c2bc000 - (int)((float)__float_as_int(u) * 0.2f));
    float c = u * -0.2f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t2 = t * t, t4 = t2 * t2, t5 = t4 * t;
        t = t * __builtin_fmaf(c, t5, 1.2f);
    }
    float w = u * t;
    return (w * w) * w;
}
DEVI float lab_cbrt(float x) {
    float t = __int_as_float(0x54a24000 - (int)((float)__float_as_int(x) * 0.33333334f));
    float c = x * -0.33333334f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t3 = (t * t) * t;
        t = t * __builtin_fmaf(c, t3, 1.3333334f);
    }
    return x * (t * t);
}
DEVI float lab_decode(float v) {
    v = clip01(v);
    // both sides are evaluated (no divergence); the pow argument is clamped into its domain so the
    // unused lane value stays finite
    float u = (v + 0.055f) * 0.9478673f;
    float p = lab_pow24(u < 0.09f ? 0.09f : u);
    return v <= 0.04045f ? v * 0.07739938f : p;
}
DEVI float lab_f(float t) {
    float c = lab_cbrt(t < 0.008f ? 0.008f : t);
    return t > 0.008856f ? c : __builtin_fmaf(7.787f, t, 0.13793103f);
}
DEVI void rgb2lab_px(float R, float G, float B, float& L, float& a, float& b) {
    R = lab_decode(R); G = lab_decode(G); B = lab_decode(B);
    float X = __builtin_fmaf(B, 0.18982783f, __builtin_fmaf(G, 0.37621942f, R * 0.43395275f));
    float Y = __builtin_fmaf(B, 0.072169f, __builtin_fmaf(G, 0.71516f, R * 0.212671f));
    float Z = __builtin_fmaf(B, 0.87276554f, __builtin_fmaf(G, 0.109476522f, R * 0.017757915f));
    float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    L = Y > 0.008856f ? __builtin_fmaf(116.0f, fy, -16.0f) : 903.3f * Y;
    a = 500.0f * (fx - fy);
    b = 200.0f * (fy - fz);
}

// ---- sRGB transfer curves, transform.py:89-111 -------------------------------------------------
// x ** (1/2.4) in the reference is a float32 power with the float32 exponent 0.41666666; we return
// the correctly rounded value of that power (float64 pow, one rounding).
DEVI float srgb_encode(float x) {
    x = clip01(x);
    float p = (float)pow((double)(x < 0.003f ? 0.003f : x), (double)0.41666666f);
    return x <= 0.0031308f ? x * 12.92f : 1.055f * p - 0.055f;
}
DEVI float srgb_decode(float x) {
    x = clip01(x);
    float u = (x + 0.055f) / 1.055f;
    float p = (float)pow((double)u, (double)2.4f);
    return x <= 0.04045f ? x / 12.92f : p;
}

// ---- colour tail applied to a camera-RGB pixel at the end of a fused pipeline -------------------
//   tail 0: none (RawDemosaicData.image)           tail 1: to_lin_srgb (clip + CCM)
//   tail 2: to_lin_srgb + lin_srgb_to_srgb          tail 3: ... with README.md:157 x/(1+x) in between
DEVI void colour_tail(int tail, const double* M, float& r, float& g, float& b) {
    if (tail == 0) return;
    float cr = clip01(r), cg = clip01(g), cb = clip01(b);
    r = ccm_row(M, cr, cg, cb);
    g = ccm_row(M + 3, cr, cg, cb);
    b = ccm_row(M + 6, cr, cg, cb);
    if (tail == 1) return;
    if (tail == 3) { r = r / (1.0f + r); g = g / (1.0f + g); b = b / (1.0f + b); }
    r = srgb_encode(r); g = srgb_encode(g); b = srgb_encode(b);
}
