/*
 * pysp_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic of bullbin/pySP's debayer -> white balance ->
 * colour matrix -> sRGB hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path (pysp_amd/,
 * libpysp_hip.so) never links, imports or falls back to it.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * All float32 expressions are evaluated left to right with every operation rounded to
 * float32 (NumPy semantics); the build uses -ffp-contract=off, so the only fused
 * multiply-adds are the explicit fmaf()/fma() calls of the restated third-party pieces.
 *
 * PARITY STATUS
 *   pinned   : Bayer demux/remux, normalisation, get_rgbg_kernel, build_map, warp tables and the
 *              orchestration (plane assembly, op order, WB-twice quirk) -- checked against the
 *              reference's own code run in the build container (tests/golden/gen_golden.py).
 *   UNPINNED : the arithmetic INSIDE the eight OpenCV calls of SURVEY.md section 2.3
 *              (opencv_python==4.10.0.84 is not installed and cannot be installed).  Their
 *              semantics are restated below from OpenCV's documented behaviour; float rounding
 *              order inside cv2 (IPP/AVX dispatch) is unknowable here.  cv2.cvtColor(RGB2LAB) in
 *              particular exists as two restatements: lab mode 1 (DEFAULT since round 2, in the
 *              oracle as in the product) is OpenCV 4.10's LUT + fixed-point trilinear path
 *              (rgb2lab_px_cv410 below; its 33^3 table is data since round 4: orc_set_cv410_lut);
 *              lab mode 0 is the closed-form sRGB-decode + D65 CIELab with our own deterministic
 *              pow / cbrt (rgb2lab_px), round 1's metric, still selectable (orc_set_lab_mode).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EBADARG (-1)
#define ORC_ENOMEM (-2)

/* ------------------------------------------------------------------------------------------
 * Border index rules (cv2.copyMakeBorder / filter borders).
 * Same loop form as OpenCV's borderInterpolate so that tiny planes are handled too. */
static inline int b_sym(int p, int n) { /* BORDER_REFLECT      fedcba|abcdefgh|hgfedcb */
    if (n == 1) return 0;
    while ((unsigned)p >= (unsigned)n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}
static inline int b_101(int p, int n) { /* BORDER_REFLECT_101  gfedcb|abcdefgh|gfedcba */
    if (n == 1) return 0;
    while ((unsigned)p >= (unsigned)n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
static inline int b_rep(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); } /* REPLICATE */

static float *falloc(size_t n) { return (float *)malloc(n * sizeof(float) + 64); }

/* ------------------------------------------------------------------------------------------
 * bayer_chan_mixer.py:4-21  bayer_to_rgbg : R=[0::2,0::2] G1=[0::2,1::2] B=[1::2,1::2] G2=[1::2,0::2] */
int orc_bayer_to_rgbg_f32(const float *bayer, int H, int W, float *r, float *g1, float *b, float *g2) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2;
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            r[(size_t)i * w + j] = bayer[(size_t)(2 * i) * W + 2 * j];
            g1[(size_t)i * w + j] = bayer[(size_t)(2 * i) * W + 2 * j + 1];
            b[(size_t)i * w + j] = bayer[(size_t)(2 * i + 1) * W + 2 * j + 1];
            g2[(size_t)i * w + j] = bayer[(size_t)(2 * i + 1) * W + 2 * j];
        }
    return ORC_OK;
}
int orc_bayer_to_rgbg_u16(const uint16_t *bayer, int H, int W, float *r, float *g1, float *b, float *g2) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2;
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            r[(size_t)i * w + j] = (float)bayer[(size_t)(2 * i) * W + 2 * j];
            g1[(size_t)i * w + j] = (float)bayer[(size_t)(2 * i) * W + 2 * j + 1];
            b[(size_t)i * w + j] = (float)bayer[(size_t)(2 * i + 1) * W + 2 * j + 1];
            g2[(size_t)i * w + j] = (float)bayer[(size_t)(2 * i + 1) * W + 2 * j];
        }
    return ORC_OK;
}
/* bayer_chan_mixer.py:23-42  rgbg_to_bayer */
int orc_rgbg_to_bayer_f32(const float *r, const float *g1, const float *b, const float *g2, int h, int w,
                          float *bayer) {
    if (h <= 0 || w <= 0) return ORC_EBADARG;
    int W = 2 * w;
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            bayer[(size_t)(2 * i) * W + 2 * j] = r[(size_t)i * w + j];
            bayer[(size_t)(2 * i) * W + 2 * j + 1] = g1[(size_t)i * w + j];
            bayer[(size_t)(2 * i + 1) * W + 2 * j + 1] = b[(size_t)i * w + j];
            bayer[(size_t)(2 * i + 1) * W + 2 * j] = g2[(size_t)i * w + j];
        }
    return ORC_OK;
}

/* normalization.py:4-24  bayer_normalize.  black/sat are indexed r,g1,b,g2 (lines 20-23); the
 * divisor is sat (not sat-black).  float32 throughout (python-int black/sat are weak scalars). */
int orc_bayer_normalize_u16(const uint16_t *bayer, int H, int W, const float black[4], const float sat[4],
                            float *out) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ORC_EBADARG;
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int c = (y & 1) ? ((x & 1) ? 2 : 3) : ((x & 1) ? 1 : 0);
            float v = (float)bayer[(size_t)y * W + x] - black[c];
            v = v < 0.0f ? 0.0f : (v > sat[c] ? sat[c] : v);
            out[(size_t)y * W + x] = v / sat[c];
        }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Restated cv2.cvtColor(COLOR_RGB2LAB) for float32 (call sites ahd.py:58,62).  UNPINNED.
 * Semantics kept from OpenCV: input is treated as gamma-encoded sRGB, clipped to [0,1], decoded,
 * converted with the D65-normalised sRGB->XYZ matrix, then CIELab with the 0.008856 / 7.787 /
 * 903.3 constants; L in [0,100].  Like OpenCV's float path, the two transcendental pieces (the
 * ((v+0.055)/1.055)^2.4 decode and the cube root) are table driven -- OpenCV interpolates 1024-knot
 * cubic splines (sRGBGammaTab, LabCbrtTab); this restatement uses float-exponent-indexed tables of
 * quadratic segments, 64 per octave of v over [2^-5, 1] (321 entries) and 32 per octave of t over
 * [2^-7, 2) (257 entries):
 *     i = (bits(x) - bits(lo)) >> S ;  s = x - float(bits(x) with the low S bits cleared) (exact) ;
 *     f(x) ~ fmaf(fmaf(c_i, s, b_i), s, a_i)
 * with a, b, c the float32 casts of the parabola through f at the segment's ends and midpoint,
 * evaluated in float64 (libm pow / cbrt).  Max relative error 1.3e-7 / 1.9e-7
 * (tests/test_oracle_primitives.py); only +,*,fmaf and integer ops at run time, so the HIP kernel
 * reproduces it bit for bit from identical tables (tests/test_abi_cpu.py compares the tables). */
static inline float f_from_bits(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t bits_from_f(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

#define LAB_DEC_NB 6
#define LAB_DEC_LOEXP (-5)
#define LAB_DEC_N (5 * (1 << LAB_DEC_NB) + 1)
#define LAB_CB_NB 5
#define LAB_CB_LOEXP (-7)
#define LAB_CB_N (8 * (1 << LAB_CB_NB) + 1)
static float lab_dec_tab[LAB_DEC_N][4], lab_cb_tab[LAB_CB_N][4];
static double lab_dec_fn(double v) { return pow((v + 0.055) / 1.055, 2.4); }
static double lab_cb_fn(double t) { return cbrt(t); }
static void lab_build(float (*tab)[4], int n, int nb, int loexp, double (*fn)(double)) {
    int32_t b0 = bits_from_f(ldexpf(1.0f, loexp));
    int s = 23 - nb;
    for (int i = 0; i < n; i++) {
        double x0 = (double)f_from_bits(b0 + (int32_t)((uint32_t)i << s));
        double x1 = (double)f_from_bits(b0 + (int32_t)((uint32_t)(i + 1) << s));
        double h = x1 - x0, g0 = fn(x0), gm = fn(x0 + 0.5 * h), g1 = fn(x1);
        tab[i][0] = (float)g0;
        tab[i][1] = (float)((-3.0 * g0 + 4.0 * gm - g1) / h);
        tab[i][2] = (float)((2.0 * g0 - 4.0 * gm + 2.0 * g1) / (h * h));
        tab[i][3] = (float)x0;   /* segment start, = x with its low S bits cleared for every x of the segment */
    }
}
static void cv410_build(void);
__attribute__((constructor)) static void lab_tables_init(void) {
    lab_build(lab_dec_tab, LAB_DEC_N, LAB_DEC_NB, LAB_DEC_LOEXP, lab_dec_fn);
    lab_build(lab_cb_tab, LAB_CB_N, LAB_CB_NB, LAB_CB_LOEXP, lab_cb_fn);
    cv410_build();
}
static inline float lab_lut(float (*tab)[4], int nb, int loexp, float x) {
    int s = 23 - nb;
    int32_t bits = bits_from_f(x), b0 = bits_from_f(ldexpf(1.0f, loexp));
    int idx = (bits - b0) >> s;
    float x0 = f_from_bits(bits & ~((1 << s) - 1));
    float fr = x - x0;
    return fmaf(fmaf(tab[idx][2], fr, tab[idx][1]), fr, tab[idx][0]);
}
static inline float lab_pow24(float v) { return lab_lut(lab_dec_tab, LAB_DEC_NB, LAB_DEC_LOEXP, v); }   /* v in (0.04045, 1]: ((v+0.055)/1.055)^2.4 */
static inline float lab_cbrt(float x) { return lab_lut(lab_cb_tab, LAB_CB_NB, LAB_CB_LOEXP, x); }        /* x in (0.008856, 2) */
static inline float lab_decode(float v) {
    v = v > 0.0f ? v : 0.0f; /* OpenCV clips the float input to [0,1] with max/min: a NaN comes out as 0 (as v_med3_f32 gives on the GPU) */
    v = v < 1.0f ? v : 1.0f;
    return v <= 0.04045f ? v * 0.07739938f /* 1/12.92 */ : lab_pow24(v);
}
int orc_lab_tables(float *dec /* LAB_DEC_N*4 */, float *cb /* LAB_CB_N*4 */) {
    memcpy(dec, lab_dec_tab, sizeof(lab_dec_tab));
    memcpy(cb, lab_cb_tab, sizeof(lab_cb_tab));
    return ORC_OK;
}
/* sRGB(D65) -> XYZ rows divided by the D65 white (0.950456, 1, 1.088754), as float32 */
#define LAB_C0 0.43395275f
#define LAB_C1 0.37621942f
#define LAB_C2 0.18982783f
#define LAB_C3 0.212671f
#define LAB_C4 0.71516f
#define LAB_C5 0.072169f
#define LAB_C6 0.017757915f
#define LAB_C7 0.109476522f
#define LAB_C8 0.87276554f
static inline float lab_f(float t) { return t > 0.008856f ? lab_cbrt(t) : fmaf(7.787f, t, 0.13793103f); }
static inline void rgb2lab_px(float R, float G, float B, float *L, float *a, float *b) {
    R = lab_decode(R); G = lab_decode(G); B = lab_decode(B);
    float X = fmaf(B, LAB_C2, fmaf(G, LAB_C1, R * LAB_C0));
    float Y = fmaf(B, LAB_C5, fmaf(G, LAB_C4, R * LAB_C3));
    float Z = fmaf(B, LAB_C8, fmaf(G, LAB_C7, R * LAB_C6));
    float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    *L = Y > 0.008856f ? fmaf(116.0f, fy, -16.0f) : 903.3f * Y;
    *a = 500.0f * (fx - fy);
    *b = 200.0f * (fy - fz);
}
/* ------------------------------------------------------------------------------------------
 * Second restatement of the same call: OpenCV 4.10's DEFAULT float32 path for sRGB input with the built-in
 * coefficients (modules/imgproc/src/color_lab.cpp, RGB2Lab_f::operator() with useInterpolation), restated from
 * memory of that source, UNPINNED like the first:
 *   clip to [0,1] (min/max: NaN -> 0) -> iv = cvRound(v * 2^14) -> 33^3 grid of int16 Lab values (closed-form Lab of
 *   applyGamma(p/32) in float32, scaled to 14 bits: L*2^14/100, (a+128)*2^14/256, (b+128)*2^14/256) -> trilinear
 *   interpolation in fixed point (cell = iv >> 9, position in the cell = (iv >> 5) & 15, weights = products of three
 *   4-bit factors, CV_DESCALE by 12 bits) -> L = l*100/2^14, a = a'*256/2^14 - 128, b likewise.
 * Output is therefore quantised (L in steps of 100/16384, a and b in steps of 1/64).  orc_set_lab_mode(1) makes
 * every Lab conversion of this library (orc_rgb2lab and the AHD homogeneity metric) use it -- the DEFAULT since round 2, in the
 * oracle as in the product (pysp_ctx_set_lab_mode); mode 0 is the closed form.  tests/lab_flip_rate.py measures how many H/V
 * decisions the choice changes. */
static int g_lab_mode = 1;   /* default: the OpenCV 4.10 path, like the product (mode 0 = the closed form of round 1) */
static int16_t cv410_lut[33][33][33][3];   /* [B][G][R] grid point */
static int cv410_ready = 0;
static void cv410_build(void) {
    static const double white[3] = {0.950456, 1.0, 1.088754};
    static const double xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    float C[9], gam[33];
    for (int i = 0; i < 9; i++) C[i] = (float)((i / 3 == 1 ? 1.0 : 1.0 / white[i / 3]) * xyz[i]);
    for (int p = 0; p < 33; p++) {
        float x = (float)p / 32.0f;
        gam[p] = x <= 0.04045f ? x / 12.92f : (float)pow((double)((x + 0.055f) / 1.055f), 2.4);
    }
    const float lthresh = 216.0f / 24389.0f, lscale = 841.0f / 108.0f, lbias = 16.0f / 116.0f, kap = 24389.0f / 27.0f;
    for (int r = 0; r < 33; r++)
        for (int q = 0; q < 33; q++)
            for (int p = 0; p < 33; p++) {
                float R = gam[p], G = gam[q], B = gam[r];
                float X = (R * C[0] + G * C[1]) + B * C[2], Y = (R * C[3] + G * C[4]) + B * C[5], Z = (R * C[6] + G * C[7]) + B * C[8];
                float FX = X > lthresh ? (float)cbrt((double)X) : fmaf(X, lscale, lbias);
                float FY = Y > lthresh ? (float)cbrt((double)Y) : fmaf(Y, lscale, lbias);
                float FZ = Z > lthresh ? (float)cbrt((double)Z) : fmaf(Z, lscale, lbias);
                float L = Y > lthresh ? 116.0f * FY - 16.0f : kap * Y;
                float a = 500.0f * (FX - FY), b = 200.0f * (FY - FZ);
                cv410_lut[r][q][p][0] = (int16_t)lrintf(16384.0f * L / 100.0f);
                cv410_lut[r][q][p][1] = (int16_t)lrintf(16384.0f * (a + 128.0f) / 256.0f);
                cv410_lut[r][q][p][2] = (int16_t)lrintf(16384.0f * (b + 128.0f) / 256.0f);
            }
    cv410_ready = 1;
}
int orc_set_lab_mode(int mode) {
    if (mode < 0 || mode > 1) return ORC_EBADARG;
    if (mode == 1 && !cv410_ready) cv410_build();
    g_lab_mode = mode;
    return ORC_OK;
}
int orc_cv410_lut(int16_t *out /* 33*33*33*3 */) {
    if (!cv410_ready) cv410_build();
    memcpy(out, cv410_lut, sizeof(cv410_lut));
    return ORC_OK;
}
/* The grid as DATA (round 4): replaces the 33^3 x 3 int16 table of lab mode 1 by the caller's (NULL: back to the built-in one).  The day a machine with
 * opencv_python==4.10.0.84 is at hand, tools/gen_cv2_goldens.py records cv2.cvtColor at the 35 937 node inputs (p/32, q/32, r/32) -- at a node every
 * interpolation weight but one is zero, so the output IS the table entry -- and the real table is injected here and in the product
 * (pysp_ctx_set_lab_lut) without touching code.  Entries must lie in [0, 32767] (OpenCV's own: L*2^14/100 in [0, 16384], (a+128)*64 in [0, 16320]). */
int orc_set_cv410_lut(const int16_t *grid /* 33*33*33*3 or NULL */) {
    if (!grid) { cv410_build(); return ORC_OK; }
    for (size_t i = 0; i < (size_t)33 * 33 * 33 * 3; i++) if (grid[i] < 0) return ORC_EBADARG;
    memcpy(cv410_lut, grid, sizeof(cv410_lut));
    cv410_ready = 1;
    return ORC_OK;
}
static inline int cv410_q(float v) {
    v = v > 0.0f ? v : 0.0f;   /* max(v, 0): a NaN comes out as 0 */
    v = v < 1.0f ? v : 1.0f;
    return (int)lrintf(v * 16384.0f);
}
static inline void rgb2lab_px_cv410(float R, float G, float B, float *L, float *a, float *b) {
    int c[3] = {cv410_q(R), cv410_q(G), cv410_q(B)}, t[3], f[3], acc[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) { t[k] = c[k] >> 9; f[k] = (c[k] >> 5) & 15; }
    for (int k = 0; k < 8; k++) {
        int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
        int w = (dx ? f[0] : 16 - f[0]) * (dy ? f[1] : 16 - f[1]) * (dz ? f[2] : 16 - f[2]);
        int px = t[0] + dx > 32 ? 32 : t[0] + dx, py = t[1] + dy > 32 ? 32 : t[1] + dy, pz = t[2] + dz > 32 ? 32 : t[2] + dz;
        const int16_t *e = cv410_lut[pz][py][px];
        acc[0] += e[0] * w; acc[1] += e[1] * w; acc[2] += e[2] * w;
    }
    for (int k = 0; k < 3; k++) acc[k] = (acc[k] + (1 << 11)) >> 12;
    *L = (float)acc[0] * (100.0f / 16384.0f);
    *a = (float)acc[1] * (256.0f / 16384.0f) - 128.0f;
    *b = (float)acc[2] * (256.0f / 16384.0f) - 128.0f;
}
static inline void rgb2lab_any(float R, float G, float B, float *L, float *a, float *b) {
    if (g_lab_mode == 1) rgb2lab_px_cv410(R, G, B, L, a, b); else rgb2lab_px(R, G, B, L, a, b);
}
int orc_rgb2lab(const float *rgb, size_t npx, float *lab) {
#pragma omp parallel for
    for (size_t i = 0; i < npx; i++)
        rgb2lab_any(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], &lab[3 * i], &lab[3 * i + 1], &lab[3 * i + 2]);
    return ORC_OK;
}
int orc_lab_pow24(const float *v, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = lab_pow24(v[i]); return ORC_OK; }
int orc_lab_cbrt(const float *x, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = lab_cbrt(x[i]); return ORC_OK; }

/* ------------------------------------------------------------------------------------------
 * debayer/ahd_homogeneity_cython.pyx:22-68  build_map / compute_map.
 * lab is (Hp,Wp,3) float32 already padded by k_pad; out is (Hp-2k,Wp-2k) float32 counts.
 * The L test is one-sided (pyx:56), squares are float32 x*x (Cython emits powf(x,2.0)), sums rounded. */
static inline float sq2(float ax, float ay, float bx, float by) {
    float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy;
}
int orc_build_map(const float *lab, int Hp, int Wp, int k_pad, int is_vertical, float *out) {
    int ry = Hp - 2 * k_pad, rx = Wp - 2 * k_pad, dk = 2 * k_pad + 1;
    if (ry <= 0 || rx <= 0 || k_pad < 1) return ORC_EBADARG;
#define LABAT(yy, xx, c) lab[((size_t)(yy) * Wp + (xx)) * 3 + (c)]
#pragma omp parallel for
    for (int y = 0; y < ry; y++) {
        int sy = y + k_pad;
        for (int x = 0; x < rx; x++) {
            int sx = x + k_pad;
            float rl = LABAT(sy, sx, 0), ra = LABAT(sy, sx, 1), rb = LABAT(sy, sx, 2);
            float el, ec, e1, e2, c1, c2;
            if (is_vertical) {
                e1 = fabsf(rl - LABAT(sy - 1, sx, 0)); e2 = fabsf(rl - LABAT(sy + 1, sx, 0));
                c1 = sq2(ra, rb, LABAT(sy - 1, sx, 1), LABAT(sy - 1, sx, 2));
                c2 = sq2(ra, rb, LABAT(sy + 1, sx, 1), LABAT(sy + 1, sx, 2));
            } else {
                e1 = fabsf(rl - LABAT(sy, sx - 1, 0)); e2 = fabsf(rl - LABAT(sy, sx + 1, 0));
                c1 = sq2(ra, rb, LABAT(sy, sx - 1, 1), LABAT(sy, sx - 1, 2));
                c2 = sq2(ra, rb, LABAT(sy, sx + 1, 1), LABAT(sy, sx + 1, 2));
            }
            el = e2 > e1 ? e2 : e1; /* Cython max(a,b) == (b > a) ? b : a */
            ec = c2 > c1 ? c2 : c1;
            float cnt = 0.0f;
            for (int wy = y; wy < y + dk; wy++)
                for (int wx = x; wx < x + dk; wx++)
                    if (LABAT(wy, wx, 0) - rl <= el) {
                        float da = LABAT(wy, wx, 1) - ra, db = LABAT(wy, wx, 2) - rb;
                        if (da * da + db * db <= ec) cnt = cnt + 1.0f;
                    }
            out[(size_t)y * rx + x] = cnt;
        }
    }
#undef LABAT
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Restated OpenCV filters (UNPINNED arithmetic order). */

/* cv2.GaussianBlur(x,(3,3),1.0) float32, BORDER_REFLECT_101 (ahd.py:120-121, eag.py:156,170,184).
 * Taps = normalised exp(-x^2/2) cast to float32; separable, row pass then column pass, each in the
 * symmetric small-kernel form  c*k0 + (l+r)*k1. */
#define GK0 0.45186276f
#define GK1 0.27406862f
int orc_gaussian_blur3(const float *src, int h, int w, float *dst) {
    float *tmp = falloc((size_t)h * w);
    if (!tmp) return ORC_ENOMEM;
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *s = src + (size_t)y * w;
            tmp[(size_t)y * w + x] = s[x] * GK0 + (s[b_101(x - 1, w)] + s[b_101(x + 1, w)]) * GK1;
        }
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const float *t0 = tmp + (size_t)b_101(y - 1, h) * w, *t1 = tmp + (size_t)y * w,
                    *t2 = tmp + (size_t)b_101(y + 1, h) * w;
        for (int x = 0; x < w; x++) dst[(size_t)y * w + x] = t1[x] * GK0 + (t0[x] + t2[x]) * GK1;
    }
    free(tmp);
    return ORC_OK;
}

/* cv2.filter2D(x,-1,k3x3) float32 image, float64 kernel (eag.py:141,143): correlation, centre anchor,
 * BORDER_REFLECT_101, kernel cast to float32, non-zero taps only, accumulated in row-major order
 * starting from 0.0f, every product and sum rounded to float32.
 * A tap whose weight is a power of two (16 of the 25 taps of get_rgbg_kernel: 1/64, 4/64, 16/64) has an exact
 * product, so accumulating it with one fmaf() gives the same bits as multiply-then-add -- except when the product
 * falls below 2^-126, where the separate multiply would round it first.  The restatement defines those taps as the
 * exact (fmaf) form, which is what lets the HIP kernels issue them as a single instruction. */
static inline int is_pow2f(float k) { int e; return frexpf(fabsf(k), &e) == 0.5f; }
int orc_filter2d_3x3(const float *src, int h, int w, const double k[9], float *dst) {
    float kf[9];
    int p2[9];
    for (int i = 0; i < 9; i++) { kf[i] = (float)k[i]; p2[i] = is_pow2f(kf[i]); }
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const float *rows[3] = {src + (size_t)b_101(y - 1, h) * w, src + (size_t)y * w,
                                src + (size_t)b_101(y + 1, h) * w};
        for (int x = 0; x < w; x++) {
            int xs[3] = {b_101(x - 1, w), x, b_101(x + 1, w)};
            float s = 0.0f;
            for (int t = 0; t < 9; t++)
                if (kf[t] != 0.0f) s = p2[t] ? fmaf(kf[t], rows[t / 3][xs[t % 3]], s) : s + kf[t] * rows[t / 3][xs[t % 3]];
            dst[(size_t)y * w + x] = s;
        }
    }
    return ORC_OK;
}

/* cv2.medianBlur(x,5) float32 (ahd.py:151): exact 5x5 median, BORDER_REPLICATE. Selection only. */
static inline float median25(float *v) {
    /* partial selection sort up to the 13th element: exact, order-independent result */
    for (int i = 0; i <= 12; i++) {
        int m = i;
        for (int j = i + 1; j < 25; j++)
            if (v[j] < v[m]) m = j;
        float t = v[i]; v[i] = v[m]; v[m] = t;
    }
    return v[12];
}
int orc_median5(const float *src, int h, int w, float *dst) {
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float v[25];
            int n = 0;
            for (int dy = -2; dy <= 2; dy++) {
                const float *s = src + (size_t)b_rep(y + dy, h) * w;
                for (int dx = -2; dx <= 2; dx++) v[n++] = s[b_rep(x + dx, w)];
            }
            dst[(size_t)y * w + x] = median25(v);
        }
    return ORC_OK;
}

/* cv2.blur(x,(3,3)) on the integer-valued homogeneity maps (ahd.py:133-134), BORDER_REFLECT_101.
 * The nine integers (<= 81 in total) sum exactly in float32 in any order; the result is scaled by
 * float32(1/9).  Only the ordering of two such values is consumed (ahd.py:139). */
int orc_box3(const float *src, int h, int w, float *dst) {
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = 0.0f;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) s = s + src[(size_t)b_101(y + dy, h) * w + b_101(x + dx, w)];
            dst[(size_t)y * w + x] = s * 0.11111111f;
        }
    return ORC_OK;
}

/* cv2.resize(rgb,(2w,2h)) INTER_LINEAR on a (h,w,c) float32 image (fast_resize.py:39): half-pixel
 * centres, edge clamp; horizontal pass then vertical pass, each  s0*a0 + s1*a1  in float32. */
int orc_resize2x_linear(const float *src, int h, int w, int c, float *dst) {
    int H = 2 * h, W = 2 * w;
    float *tmp = falloc((size_t)h * W * c);
    if (!tmp) return ORC_ENOMEM;
#pragma omp parallel for
    for (int y = 0; y < h; y++)
        for (int X = 0; X < W; X++) {
            float fx = ((float)X + 0.5f) * 0.5f - 0.5f;
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            if (sx < 0) { sx = 0; fx = 0.0f; }
            if (sx >= w - 1) { sx = w - 1; fx = 0.0f; }
            int sx1 = sx + 1 < w ? sx + 1 : sx;
            float a0 = 1.0f - fx, a1 = fx;
            for (int k = 0; k < c; k++)
                tmp[((size_t)y * W + X) * c + k] =
                    src[((size_t)y * w + sx) * c + k] * a0 + src[((size_t)y * w + sx1) * c + k] * a1;
        }
#pragma omp parallel for
    for (int Y = 0; Y < H; Y++) {
        float fy = ((float)Y + 0.5f) * 0.5f - 0.5f;
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        if (sy < 0) { sy = 0; fy = 0.0f; }
        if (sy >= h - 1) { sy = h - 1; fy = 0.0f; }
        int sy1 = sy + 1 < h ? sy + 1 : sy;
        float b0 = 1.0f - fy, b1 = fy;
        for (size_t i = 0; i < (size_t)W * c; i++)
            dst[(size_t)Y * W * c + i] = tmp[(size_t)sy * W * c + i] * b0 + tmp[(size_t)sy1 * W * c + i] * b1;
    }
    free(tmp);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Colour (colorize/transform.py). M is the final row-major 3x3 (float64) built on the host by
 * transform.py:40-49; out_i = sum_j rgb_j * M[i][j] (np.dot(rgb, M.T), transform.py:52).
 * Accumulation order follows the dgemm micro-kernel NumPy dispatches to: t = r*m0; t = fma(g,m1,t);
 * t = fma(b,m2,t) in float64, then one rounding to float32 (transform.py:53). */
static inline float ccm_row(const double *m, float r, float g, float b) {
    double t = (double)r * m[0];
    t = fma((double)g, m[1], t);
    t = fma((double)b, m[2], t);
    return (float)t;
}
static inline float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); } /* transform.py:6-19 */

int orc_clip_rgb(const float *in, size_t n, float *out) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) out[i] = clip01(in[i]);
    return ORC_OK;
}
/* transform.py:21-53 cam_to_rgb_norm (pixel part) */
int orc_cam_to_rgb(const float *in, size_t npx, const double M[9], int clip, float *out) {
#pragma omp parallel for
    for (size_t i = 0; i < npx; i++) {
        float r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
        if (clip) { r = clip01(r); g = clip01(g); b = clip01(b); }
        out[3 * i] = ccm_row(M, r, g, b);
        out[3 * i + 1] = ccm_row(M + 3, r, g, b);
        out[3 * i + 2] = ccm_row(M + 6, r, g, b);
    }
    return ORC_OK;
}
/* transform.py:89-99 lin_srgb_to_srgb: clip; x<=0.0031308 ? x*12.92 : 1.055*x**(1/2.4) - 0.055, all float32.
 * NumPy evaluates x**(1/2.4) with a float32 powf whose last bit is platform dependent (SVML vs libm);
 * the oracle uses the correctly rounded value of x^float32(1/2.4). */
static inline float srgb_encode(float x) {
    x = clip01(x);
    if (x <= 0.0031308f) return x * 12.92f;
    float p = (float)pow((double)x, (double)0.41666666f);
    return 1.055f * p - 0.055f;
}
int orc_lin_srgb_to_srgb(const float *in, size_t n, float *out) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) out[i] = srgb_encode(in[i]);
    return ORC_OK;
}
/* transform.py:101-111 srgb_to_lin_srgb */
int orc_srgb_to_lin_srgb(const float *in, size_t n, float *out) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) {
        float x = clip01(in[i]);
        out[i] = x <= 0.04045f ? x / 12.92f : (float)pow((double)((x + 0.055f) / 1.055f), (double)2.4f);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Plane helpers shared by the demosaic restatements */
typedef struct { int h, w; float *r, *g1, *b, *g2; } planes_t;

static int planes_alloc(planes_t *p, int h, int w) {
    p->h = h; p->w = w;
    p->r = falloc((size_t)h * w); p->g1 = falloc((size_t)h * w);
    p->b = falloc((size_t)h * w); p->g2 = falloc((size_t)h * w);
    return (p->r && p->g1 && p->b && p->g2) ? ORC_OK : ORC_ENOMEM;
}
static void planes_free(planes_t *p) { free(p->r); free(p->g1); free(p->b); free(p->g2); }

/* cv2.copyMakeBorder(x,t,b,l,r,BORDER_REFLECT) then optional "* wb" (ahd.py:77-80) */
static float *pad_sym_scale(const float *src, int h, int w, int top, int bot, int left, int right, float scale,
                            int do_scale) {
    int hp = h + top + bot, wp = w + left + right;
    float *dst = falloc((size_t)hp * wp);
    if (!dst) return NULL;
#pragma omp parallel for
    for (int y = 0; y < hp; y++)
        for (int x = 0; x < wp; x++) {
            float v = src[(size_t)b_sym(y - top, h) * w + b_sym(x - left, w)];
            dst[(size_t)y * wp + x] = do_scale ? v * scale : v;
        }
    return dst;
}

/* debayer/gaussian.py:19-54 get_rgbg_kernel on CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL (gaussian.py:6-10).
 * out[4][9]: kernels for the TopLeft, TopRight, BottomLeft, BottomRight target photosites (float64). */
int orc_get_rgbg_kernel(int base_position /*0 TL,1 TR,2 BL,3 BR*/, double out[36]) {
    static const double K5[5][5] = {{1, 4, 6, 4, 1}, {4, 16, 24, 16, 4}, {6, 24, 36, 24, 6}, {4, 16, 24, 16, 4}, {1, 4, 6, 4, 1}};
    if (base_position < 0 || base_position > 3) return ORC_EBADARG;
    int base_left = (base_position == 0 || base_position == 2), base_bottom = (base_position >= 2);
    for (int idx = 0; idx < 4; idx++) {
        int is_left = (idx == 0 || idx == 2), is_bottom = (idx >= 2);
        int nr, nc, rows[3], cols[3];
        if (is_bottom == base_bottom) { nr = 3; rows[0] = 0; rows[1] = 2; rows[2] = 4; } else { nr = 2; rows[0] = 1; rows[1] = 3; }
        if (is_left == base_left) { nc = 3; cols[0] = 0; cols[1] = 2; cols[2] = 4; } else { nc = 2; cols[0] = 1; cols[1] = 3; }
        double k[3][3] = {{0}}, sum = 0;
        /* a 2-wide slice is zero padded: appended when the target is left of the base, prepended otherwise;
         * a 2-tall slice gets a zero row on top when the target is below the base, underneath otherwise */
        int c0 = (nc == 2 && !is_left) ? 1 : 0, r0 = (nr == 2 && is_bottom) ? 1 : 0;
        for (int a = 0; a < nr; a++)
            for (int b = 0; b < nc; b++) { k[r0 + a][c0 + b] = K5[rows[a]][cols[b]]; sum += K5[rows[a]][cols[b]]; }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) out[idx * 9 + a * 3 + b] = k[a][b] / sum;
    }
    return ORC_OK;
}

/* edge_assisted_gaussian.py:126-143 resample_channel.  sub, g_sub: (h,w); g_hf: (2h,2w); out: (2h,2w).
 * Tuple quirk: get_rgbg_kernel returns TL,TR,BL,BR; rgbg_to_bayer(r,g1,b,g2) places them at TL,TR,BR,BL. */
static int resample_channel(const float *sub, const float *g_sub, const float *g_hf, int h, int w, int pos,
                            float *out) {
    double k[36];
    orc_get_rgbg_kernel(pos, k);
    size_t n = (size_t)h * w;
    float *f[4], *d = falloc(n), *gup = falloc(4 * n);
    int rc = ORC_OK;
    for (int i = 0; i < 4; i++) f[i] = falloc(n);
    if (!d || !gup || !f[0] || !f[1] || !f[2] || !f[3]) { rc = ORC_ENOMEM; goto done; }
    for (int i = 0; i < 4; i++) orc_filter2d_3x3(g_sub, h, w, k + 9 * i, f[i]);
    orc_rgbg_to_bayer_f32(f[0], f[1], f[3], f[2], h, w, gup); /* (k_r,k_g,k_b,k_g2) = (TL,TR,BR,BL) */
#pragma omp parallel for
    for (size_t i = 0; i < 4 * n; i++) gup[i] = gup[i] + g_hf[i];
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) d[i] = sub[i] - g_sub[i];
    for (int i = 0; i < 4; i++) orc_filter2d_3x3(d, h, w, k + 9 * i, f[i]);
    orc_rgbg_to_bayer_f32(f[0], f[1], f[3], f[2], h, w, out);
#pragma omp parallel for
    for (size_t i = 0; i < 4 * n; i++) out[i] = out[i] + gup[i];
done:
    for (int i = 0; i < 4; i++) free(f[i]);
    free(d); free(gup);
    return rc;
}
int orc_resample_channel(const float *sub, const float *g_sub, const float *g_hf, int h, int w, int pos, float *out) {
    return resample_channel(sub, g_sub, g_hf, h, w, pos, out);
}

static void interleave3(const float *r, const float *g, const float *b, size_t n, float *rgb) {
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) { rgb[3 * i] = r[i]; rgb[3 * i + 1] = g[i]; rgb[3 * i + 2] = b[i]; }
}

/* ------------------------------------------------------------------------------------------
 * debayer/fast_resize.py:7-44  Draft.  rgb_out is (H,W,3). */
int orc_demosaic_draft(const float *bayer, int H, int W, const float wb[3], float *rgb_out) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2;
    planes_t p = {0, 0, NULL, NULL, NULL, NULL};
    if (planes_alloc(&p, h, w)) { planes_free(&p); return ORC_ENOMEM; }
    orc_bayer_to_rgbg_f32(bayer, H, W, p.r, p.g1, p.b, p.g2);
    float *q = falloc((size_t)h * w * 3);
    float *rp = pad_sym_scale(p.r, h, w, 0, 1, 0, 1, 1.0f, 0); /* fast_resize.py:28 */
    float *bp = pad_sym_scale(p.b, h, w, 1, 0, 1, 0, 1.0f, 0); /* fast_resize.py:29 */
    if (!q || !rp || !bp) { free(q); free(rp); free(bp); planes_free(&p); return ORC_ENOMEM; }
    int wp = w + 1;
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            size_t o = ((size_t)i * w + j) * 3;
            float rr = 0.75f * rp[(size_t)i * wp + j] + 0.25f * rp[(size_t)(i + 1) * wp + j + 1];       /* :31-32 */
            float bb = 0.75f * bp[(size_t)(i + 1) * wp + j + 1] + 0.25f * bp[(size_t)i * wp + j];       /* :33-34 */
            q[o] = rr * wb[0];                                                                           /* :36 */
            q[o + 1] = ((p.g1[(size_t)i * w + j] + p.g2[(size_t)i * w + j]) / 2.0f) * wb[1];            /* :26 */
            q[o + 2] = bb * wb[2];                                                                       /* :37 */
        }
    int rc = orc_resize2x_linear(q, h, w, 3, rgb_out);
    free(q); free(rp); free(bp); planes_free(&p);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * edge_assisted_gaussian.py:10-49 simple_delta_mix_bilinear_kernel */
static inline float delta_mix(float top, float bottom, float left, float right) {
    float dy = fabsf(top - bottom), dx = fabsf(left - right), s = dy + dx;
    float ax = (left + right) / 2.0f, ay = (top + bottom) / 2.0f;
    float sy = s != 0.0f ? dy / s : 0.5f;
    float sx = 1.0f - sy;
    return ay * sx + ax * sy;
}
/* edge_assisted_gaussian.py:51-124 resample_g_to_full_resolution (use_bilinear_weighting=True) */
int orc_resample_g_full(const float *g1, const float *g2, int h, int w, float *g_full) {
    float *a = pad_sym_scale(g1, h, w, 1, 1, 1, 1, 1.0f, 0), *c = pad_sym_scale(g2, h, w, 1, 1, 1, 1, 1.0f, 0);
    float *gr = falloc((size_t)h * w), *gb = falloc((size_t)h * w);
    if (!a || !c || !gr || !gb) { free(a); free(c); free(gr); free(gb); return ORC_ENOMEM; }
    int wp = w + 2;
#define P1(i, j) a[(size_t)(i) * wp + (j)]
#define P2(i, j) c[(size_t)(i) * wp + (j)]
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            int pi = i + 1, pj = j + 1;
            /* red site: t=g2[i-1,j] b=g2[i,j] l=g1[i,j-1] r=g1[i,j]   (:105-108) */
            gr[(size_t)i * w + j] = delta_mix(P2(pi - 1, pj), P2(pi, pj), P1(pi, pj - 1), P1(pi, pj));
            /* blue site: t=g1[i,j] b=g1[i+1,j] l=g2[i,j] r=g2[i,j+1]  (:99-102) */
            gb[(size_t)i * w + j] = delta_mix(P1(pi, pj), P1(pi + 1, pj), P2(pi, pj), P2(pi, pj + 1));
        }
#undef P1
#undef P2
    orc_rgbg_to_bayer_f32(gr, g1, gb, g2, h, w, g_full);
    free(a); free(c); free(gr); free(gb);
    return ORC_OK;
}

/* edge_assisted_gaussian.py:188-201 debayer (EAG, "Fast") incl. resample_rb :145-158 */
int orc_demosaic_eag(const float *bayer, int H, int W, const float wb[3], float *rgb_out) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2;
    size_t n = (size_t)h * w, N = (size_t)H * W;
    planes_t p = {0, 0, NULL, NULL, NULL, NULL}, gq = {0, 0, NULL, NULL, NULL, NULL};
    int rc = ORC_ENOMEM;
    float *g_up = falloc(N), *blur = falloc(N), *hf = falloc(N), *r_up = falloc(N), *b_up = falloc(N);
    if (planes_alloc(&p, h, w) || planes_alloc(&gq, h, w) || !g_up || !blur || !hf || !r_up || !b_up) goto done;
    orc_bayer_to_rgbg_f32(bayer, H, W, p.r, p.g1, p.b, p.g2);
    orc_resample_g_full(p.g1, p.g2, h, w, g_up);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) g_up[i] = g_up[i] * wb[1]; /* :193 */
#pragma omp parallel for
    for (size_t i = 0; i < n; i++) { p.r[i] = p.r[i] * wb[0]; p.b[i] = p.b[i] * wb[2]; } /* :194 */
    orc_gaussian_blur3(g_up, H, W, blur);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) hf[i] = g_up[i] - blur[i]; /* :156 */
    orc_bayer_to_rgbg_f32(g_up, H, W, gq.r, gq.g1, gq.b, gq.g2); /* :157 */
    if ((rc = resample_channel(p.r, gq.r, hf, h, w, 0, r_up))) goto done; /* TOP_LEFT */
    if ((rc = resample_channel(p.b, gq.b, hf, h, w, 3, b_up))) goto done; /* BOTTOM_RIGHT */
    interleave3(r_up, g_up, b_up, N, rgb_out);
    rc = ORC_OK;
done:
    planes_free(&p); planes_free(&gq);
    free(g_up); free(blur); free(hf); free(r_up); free(b_up);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * debayer/ahd.py:14-170  AHD ("Best").
 *   h: ahd.py:89-94 evaluated in float32 NumPy = [-0x1.053316p-2, 0.5, 0x1.053316p-1, 0.5, -0x1.053316p-2]
 *      (checked against the NumPy expression in tests/test_oracle_primitives.py). */
static const float AHD_H[5] = {-0x1.053316p-2f, 0x1p-1f, 0x1.053316p-1f, 0x1p-1f, -0x1.053316p-2f};

/* ahd.py:32-67 build_homogeneity_map: WB applied a second time (:46-48), CCM without clip, Lab
 * (HDR: luma for L, x/(1+x) tonemap for a,b :52-59), symmetric pad 1 (:64), build_map (:66). */
static int homogeneity_map(const float *r, const float *g, const float *b, int H, int W, const float wb[3],
                           const double M[9], int hdr, int is_vertical, float *map) {
    size_t N = (size_t)H * W;
    int Hp = H + 2, Wp = W + 2;
    float *lab = falloc(N * 3), *labp = falloc((size_t)Hp * Wp * 3);
    if (!lab || !labp) { free(lab); free(labp); return ORC_ENOMEM; }
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) {
        float rr = r[i] * wb[0], gg = g[i] * wb[1], bb = b[i] * wb[2];
        float sr = ccm_row(M, rr, gg, bb), sg = ccm_row(M + 3, rr, gg, bb), sb = ccm_row(M + 6, rr, gg, bb);
        float L, A, B;
        if (hdr) {
            float luma = 0.2126f * sr + 0.7152f * sg + 0.0722f * sb; /* :55 */
            sr = sr / (1.0f + sr); sg = sg / (1.0f + sg); sb = sb / (1.0f + sb); /* :57 */
            rgb2lab_any(sr, sg, sb, &L, &A, &B);
            L = luma; /* :59 */
        } else {
            rgb2lab_any(sr, sg, sb, &L, &A, &B);
        }
        lab[3 * i] = L; lab[3 * i + 1] = A; lab[3 * i + 2] = B;
    }
#pragma omp parallel for
    for (int y = 0; y < Hp; y++)
        for (int x = 0; x < Wp; x++) {
            size_t s = ((size_t)b_sym(y - 1, H) * W + b_sym(x - 1, W)) * 3, d = ((size_t)y * Wp + x) * 3;
            labp[d] = lab[s]; labp[d + 1] = lab[s + 1]; labp[d + 2] = lab[s + 2];
        }
    int rc = orc_build_map(labp, Hp, Wp, 1, is_vertical, map);
    free(lab); free(labp);
    return rc;
}

/* ahd.py:148-161 postprocess_color (one stage), in place on planar r,g,b */
static int postprocess_stage(float *r, float *g, float *b, int H, int W) {
    size_t N = (size_t)H * W;
    float *d = falloc(N), *m1 = falloc(N), *m2 = falloc(N);
    if (!d || !m1 || !m2) { free(d); free(m1); free(m2); return ORC_ENOMEM; }
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) d[i] = r[i] - g[i];
    orc_median5(d, H, W, m1);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) r[i] = m1[i] + g[i]; /* :158 */
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) d[i] = b[i] - g[i];
    orc_median5(d, H, W, m1);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) b[i] = m1[i] + g[i]; /* :159 */
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) d[i] = g[i] - r[i];
    orc_median5(d, H, W, m1);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) d[i] = g[i] - b[i];
    orc_median5(d, H, W, m2);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) g[i] = (((m1[i] + m2[i]) + r[i]) + b[i]) / 2.0f; /* :160 */
    free(d); free(m1); free(m2);
    return ORC_OK;
}

/* Debug taps: any non-NULL pointer receives the named intermediate (planar (H,W) float32). */
typedef struct {
    float *g_h, *g_v, *r_h, *r_v, *b_h, *b_v, *map_h, *map_v, *pre_post /* (H,W,3) before median stages */;
} orc_ahd_taps;

int orc_demosaic_ahd_taps(const float *bayer, int H, int W, const float wb[3], const double M[9], int hdr,
                          int stages, float *rgb_out, const orc_ahd_taps *taps) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2, wp = w + 2;
    size_t n = (size_t)h * w, N = (size_t)H * W;
    int rc = ORC_ENOMEM;
    planes_t p, gh, gv; /* gh/gv: r=g at red sites, b=g at blue sites, g1/g2 unused copies */
    float *rp = NULL, *g1p = NULL, *bp = NULL, *g2p = NULL;
    float *g_h = falloc(N), *g_v = falloc(N), *tmp = falloc(N), *hf_h = falloc(N), *hf_v = falloc(N);
    float *r_h = falloc(N), *r_v = falloc(N), *b_h = falloc(N), *b_v = falloc(N);
    float *map_h = falloc(N), *map_v = falloc(N), *bx_h = falloc(N), *bx_v = falloc(N);
    float *R = falloc(N), *G = falloc(N), *B = falloc(N);
    float *rc_ = falloc(n), *bc_ = falloc(n), *g1c = falloc(n), *g2c = falloc(n);
    p.r = p.g1 = p.b = p.g2 = gh.r = gh.g1 = gh.b = gh.g2 = gv.r = gv.g1 = gv.b = gv.g2 = NULL;
    if (planes_alloc(&p, h, w) || planes_alloc(&gh, h, w) || planes_alloc(&gv, h, w)) goto done;
    if (!g_h || !g_v || !tmp || !hf_h || !hf_v || !r_h || !r_v || !b_h || !b_v || !map_h || !map_v || !bx_h ||
        !bx_v || !R || !G || !B || !rc_ || !bc_ || !g1c || !g2c)
        goto done;

    orc_bayer_to_rgbg_f32(bayer, H, W, p.r, p.g1, p.b, p.g2);                 /* ahd.py:69 */
    rp = pad_sym_scale(p.r, h, w, 1, 1, 1, 1, wb[0], 1);                       /* :77 */
    g1p = pad_sym_scale(p.g1, h, w, 1, 1, 1, 1, wb[1], 1);                     /* :78 */
    bp = pad_sym_scale(p.b, h, w, 1, 1, 1, 1, wb[2], 1);                       /* :79 */
    g2p = pad_sym_scale(p.g2, h, w, 1, 1, 1, 1, wb[1], 1);                     /* :80 */
    if (!rp || !g1p || !bp || !g2p) goto done;
    const float *hh = AHD_H;
#define PR(i, j) rp[(size_t)(i) * wp + (j)]
#define PG1(i, j) g1p[(size_t)(i) * wp + (j)]
#define PB(i, j) bp[(size_t)(i) * wp + (j)]
#define PG2(i, j) g2p[(size_t)(i) * wp + (j)]
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            int a = i + 1, c = j + 1; /* padded coordinates of [1:-1,1:-1] */
            size_t o = (size_t)i * w + j;
            /* :97  gh_r */
            gh.r[o] = (((PR(a, c - 1) * hh[0] + PG1(a, c - 1) * hh[1]) + PR(a, c) * hh[2]) + PG1(a, c) * hh[3]) + PR(a, c + 1) * hh[4];
            /* :98  gv_r */
            gv.r[o] = (((PR(a - 1, c) * hh[0] + PG2(a - 1, c) * hh[1]) + PR(a, c) * hh[2]) + PG2(a, c) * hh[3]) + PR(a + 1, c) * hh[4];
            /* :101 gh_b */
            gh.b[o] = (((PB(a, c - 1) * hh[0] + PG2(a, c) * hh[1]) + PB(a, c) * hh[2]) + PG2(a, c + 1) * hh[3]) + PB(a, c + 1) * hh[4];
            /* :102 gv_b */
            gv.b[o] = (((PB(a - 1, c) * hh[0] + PG1(a, c) * hh[1]) + PB(a, c) * hh[2]) + PG1(a + 1, c) * hh[3]) + PB(a + 1, c) * hh[4];
            rc_[o] = PR(a, c); bc_[o] = PB(a, c); g1c[o] = PG1(a, c); g2c[o] = PG2(a, c);
        }
#undef PR
#undef PG1
#undef PB
#undef PG2
    orc_rgbg_to_bayer_f32(gh.r, g1c, gh.b, g2c, h, w, g_h); /* :105 */
    orc_rgbg_to_bayer_f32(gv.r, g1c, gv.b, g2c, h, w, g_v); /* :106 */
    orc_gaussian_blur3(g_h, H, W, tmp);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) hf_h[i] = g_h[i] - tmp[i]; /* :120 */
    orc_gaussian_blur3(g_v, H, W, tmp);
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) hf_v[i] = g_v[i] - tmp[i]; /* :121 */
    if ((rc = resample_channel(rc_, gh.r, hf_h, h, w, 0, r_h))) goto done; /* :123 */
    if ((rc = resample_channel(rc_, gv.r, hf_v, h, w, 0, r_v))) goto done; /* :124 */
    if ((rc = resample_channel(bc_, gh.b, hf_h, h, w, 3, b_h))) goto done; /* :126 */
    if ((rc = resample_channel(bc_, gv.b, hf_v, h, w, 3, b_v))) goto done; /* :127 */
    if ((rc = homogeneity_map(r_h, g_h, b_h, H, W, wb, M, hdr, 0, map_h))) goto done; /* :129 */
    if ((rc = homogeneity_map(r_v, g_v, b_v, H, W, wb, M, hdr, 1, map_v))) goto done; /* :130 */
    orc_box3(map_h, H, W, bx_h); /* :133 */
    orc_box3(map_v, H, W, bx_v); /* :134 */
#pragma omp parallel for
    for (size_t i = 0; i < N; i++) { /* :139-145, literally: rgb_h*c + rgb_v*(1-c) */
        float c = bx_h[i] < bx_v[i] ? 1.0f : 0.0f, nc = 1.0f - c;
        R[i] = r_h[i] * c + r_v[i] * nc;
        G[i] = g_h[i] * c + g_v[i] * nc;
        B[i] = b_h[i] * c + b_v[i] * nc;
    }
    if (taps) {
        if (taps->g_h) memcpy(taps->g_h, g_h, N * 4);
        if (taps->g_v) memcpy(taps->g_v, g_v, N * 4);
        if (taps->r_h) memcpy(taps->r_h, r_h, N * 4);
        if (taps->r_v) memcpy(taps->r_v, r_v, N * 4);
        if (taps->b_h) memcpy(taps->b_h, b_h, N * 4);
        if (taps->b_v) memcpy(taps->b_v, b_v, N * 4);
        if (taps->map_h) memcpy(taps->map_h, map_h, N * 4);
        if (taps->map_v) memcpy(taps->map_v, map_v, N * 4);
        if (taps->pre_post) interleave3(R, G, B, N, taps->pre_post);
    }
    if (stages < 0) stages = 0; /* :163 */
    for (int s = 0; s < stages; s++)
        if ((rc = postprocess_stage(R, G, B, H, W))) goto done; /* :164-165 */
    interleave3(R, G, B, N, rgb_out);
    rc = ORC_OK;
done:
    planes_free(&p); planes_free(&gh); planes_free(&gv);
    free(rp); free(g1p); free(bp); free(g2p);
    free(g_h); free(g_v); free(tmp); free(hf_h); free(hf_v); free(r_h); free(r_v); free(b_h); free(b_v);
    free(map_h); free(map_v); free(bx_h); free(bx_v); free(R); free(G); free(B);
    free(rc_); free(bc_); free(g1c); free(g2c);
    return rc;
}
int orc_demosaic_ahd(const float *bayer, int H, int W, const float wb[3], const double M[9], int hdr, int stages,
                     float *rgb_out) {
    return orc_demosaic_ahd_taps(bayer, H, W, wb, M, hdr, stages, rgb_out, NULL);
}

/* README.md:55-63 recipe on one frame: demosaic(quality) -> to_lin_srgb() (image_base.py:62-64 ->
 * transform.py:76-87, clip on) -> lin_srgb_to_srgb (transform.py:89-99).  quality: 0 Draft, 1 Fast, 2 Best.
 * If reinhard != 0 the README.md:157 tonemap x/(1+x) is applied between the two colour steps. */
int orc_pipeline_srgb(const float *bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr,
                      int stages, int reinhard, float *srgb_out) {
    int rc;
    size_t n3 = (size_t)H * W * 3;
    if (quality == 2) rc = orc_demosaic_ahd(bayer, H, W, wb, M, hdr, stages, srgb_out);
    else if (quality == 1) rc = orc_demosaic_eag(bayer, H, W, wb, srgb_out);
    else if (quality == 0) rc = orc_demosaic_draft(bayer, H, W, wb, srgb_out);
    else return ORC_EBADARG;
    if (rc) return rc;
    orc_cam_to_rgb(srgb_out, n3 / 3, M, 1, srgb_out);
    if (reinhard) {
#pragma omp parallel for
        for (size_t i = 0; i < n3; i++) srgb_out[i] = srgb_out[i] / (1.0f + srgb_out[i]);
    }
    return orc_lin_srgb_to_srgb(srgb_out, n3, srgb_out);
}

/* ------------------------------------------------------------------------------------------
 * raw_hdr.py:85-158 fuse_exposures_to_raw, intended behaviour (HEAD raises at :150).
 *   frames[k]  : (H,W) float32 mosaics          ev_off[k] : float32(2**(ev_k - target))      (:119-121)
 *   bias[k*4+c]: float32 1.6**(-0.1*|ev_off_k*w_c|) for CFA site c in r,g1,b,g2 order, computed by the
 *                caller with NumPy exactly as :136 does (only 4 distinct values per frame)
 *   out (H,W) float32, count (H,W) int32. */
int orc_fuse_raw(const float *const *frames, int K, int H, int W, const float *ev_off, const float *bias, int kmax,
                 float *out, int32_t *count) {
    if (K <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || kmax < 0 || kmax >= K) return ORC_EBADARG;
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t o = (size_t)y * W + x;
            int c = (y & 1) ? ((x & 1) ? 2 : 3) : ((x & 1) ? 1 : 0);
            float sw = 0.0f, sp = 0.0f;
            int32_t cnt = 0;
            for (int k = 0; k < K; k++) {
                float v = frames[k][o];
                float wgt = (0.5f - fabsf(v - 0.5f)) * bias[k * 4 + c]; /* :137 */
                sw = sw + wgt;                                          /* :138 */
                sp = sp + (v * wgt) * ev_off[k];                        /* :139 */
                if (wgt > 0.0f) cnt++;                                  /* :141 */
            }
            float q = sp / sw;                                          /* :147 */
            out[o] = sw == 0.0f ? frames[kmax][o] * ev_off[kmax] : q;   /* :144,148 */
            count[o] = cnt;
        }
    return ORC_OK;
}

/* raw_hdr.py:7-83 fuse_exposures_from_debayer, pixel loop :54-81.  frames[k]: (npx,3) float32 images as the
 * exposures hold them; applied[k]: exposure._wb_applied; coeff[k*3+c]: exposure._wb_coeff; bias[k] =
 * float32(1.6**(-0.1*ev_off_k)) (:60-61, python float folded into a float32 array).  Per element:
 *   u = wb_undo  : float32(float64(a) / coeff)  if applied else a          (image_base.py:52-60)
 *   w = (0.5 - |u - 0.5|) * bias ; sw += w                                   (:58-63)
 *   v = wb_apply : float32(u * coeff)                                        (image_base.py:45-50)
 *   sp += (v * w) * ev_off ; count += (w > 0)                                (:70,:72)
 * out = sw == 0 ? v_kmax * ev_off_kmax : sp / sw (:74-79), then cam_to_lin_srgb(clip=False) (:81) if M.
 * kmax = LAST exposure whose ev_off equals the maximum (:67-68).  frames_out[k] (optional) receives v,
 * the state the reference leaves in exposure.image. */
int orc_fuse_rgb(const float *const *frames, int K, size_t npx, const float *coeff, const int *applied, const float *ev_off,
                 const float *bias, int kmax, const double *M, float *out, int32_t *count, float *const *frames_out) {
    if (K <= 0 || kmax < 0 || kmax >= K) return ORC_EBADARG;
#pragma omp parallel for
    for (size_t i = 0; i < npx; i++) {
        float res[3];
        for (int c = 0; c < 3; c++) {
            size_t o = 3 * i + c;
            float sw = 0.0f, sp = 0.0f, vmax = 0.0f;
            int32_t cnt = 0;
            for (int k = 0; k < K; k++) {
                float a = frames[k][o], cf = coeff[k * 3 + c];
                float u = applied[k] ? (float)((double)a / (double)cf) : a;
                float w = (0.5f - fabsf(u - 0.5f)) * bias[k];
                sw = sw + w;
                float v = u * cf;
                sp = sp + (v * w) * ev_off[k];
                if (w > 0.0f) cnt++;
                if (k == kmax) vmax = v;
                if (frames_out && frames_out[k]) frames_out[k][o] = v;
            }
            float q = sp / sw;
            res[c] = sw == 0.0f ? vmax * ev_off[kmax] : q;
            count[o] = cnt;
        }
        if (M) {
            out[3 * i] = ccm_row(M, res[0], res[1], res[2]);
            out[3 * i + 1] = ccm_row(M + 3, res[0], res[1], res[2]);
            out[3 * i + 2] = ccm_row(M + 6, res[0], res[1], res[2]);
        } else {
            out[3 * i] = res[0]; out[3 * i + 1] = res[1]; out[3 * i + 2] = res[2];
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * dng_warp_corr/dng_warp_rectilinear_coords.pyx:18-40,67-80 (+ seeded :44-65,82-96).
 * Cython lowers `x ** k` on C floats to powf(x,k.0) and sqrt() to the double sqrt. table is (H,W,2). */
static void warp_setup(int width, int height, float cxn, float cyn, float *cx, float *cy, float *m) {
    *cx = (float)(width - 1) * cxn;  /* unsigned*float -> float, pyx:73 */
    *cy = (float)(height - 1) * cyn;
    float mx = fmaxf(fabsf(-*cx), fabsf((float)(width - 1) - *cx));
    float my = fmaxf(fabsf(-*cy), fabsf((float)(height - 1) - *cy));
    *m = (float)sqrt((double)(powf(mx, 2.0f) + powf(my, 2.0f))); /* np.sqrt on a C float -> float32 result */
}
static inline void warp_px(float sx, float sy, float kr0, float kr1, float kr2, float kr3, float kt0, float kt1,
                           float m, float cx, float cy, float scale, float *ox, float *oy) {
    float dx = (sx - cx) / m, dy = (sy - cy) / m;
    float r = (float)sqrt((double)(powf(dx, 2.0f) + powf(dy, 2.0f)));
    float f = ((kr0 + (kr1 * powf(r, 2.0f))) + (kr2 * powf(r, 4.0f))) + (kr3 * powf(r, 6.0f));
    float dxr = f * dx, dyr = f * dy;
    float dxt = kt0 * ((2.0f * dx) * dy) + kt1 * (powf(r, 2.0f) + 2.0f * powf(dx, 2.0f));
    float dyt = kt1 * ((2.0f * dx) * dy) + kt0 * (powf(r, 2.0f) + 2.0f * powf(dy, 2.0f));
    float xp = cx + m * (dxr + dxt), yp = cy + m * (dyr + dyt);
    *ox = sx + (xp - sx) * scale;
    *oy = sy + (yp - sy) * scale;
}
/* rows [y0, y1) of the table only (table: (y1-y0, W, 2); seed, when given, is the whole (H,W,2) prior): a band of a frame too large to tabulate whole */
int orc_warp_table_rows(float kr0, float kr1, float kr2, float kr3, float kt0, float kt1, int width, int height,
                        float cxn, float cyn, float scale, const float *seed /* NULL or (H,W,2) */, int y0, int y1, float *table) {
    if (width <= 0 || height <= 0 || y0 < 0 || y1 > height || y0 > y1) return ORC_EBADARG;
    float cx, cy, m;
    warp_setup(width, height, cxn, cyn, &cx, &cy, &m);
#pragma omp parallel for
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < width; x++) {
            size_t o = ((size_t)y * width + x) * 2, t = ((size_t)(y - y0) * width + x) * 2;
            float sx = seed ? seed[o] : (float)x, sy = seed ? seed[o + 1] : (float)y;
            warp_px(sx, sy, kr0, kr1, kr2, kr3, kt0, kt1, m, cx, cy, scale, &table[t], &table[t + 1]);
        }
    return ORC_OK;
}
int orc_warp_table(float kr0, float kr1, float kr2, float kr3, float kt0, float kt1, int width, int height,
                   float cxn, float cyn, float scale, const float *seed /* NULL or (H,W,2) */, float *table) {
    return orc_warp_table_rows(kr0, kr1, kr2, kr3, kt0, kt1, width, height, cxn, cyn, scale, seed, 0, height, table);
}

/* Restated cv2.remap(plane, mapx, mapy, INTER_LANCZOS4) float32, BORDER_CONSTANT 0
 * (chan_distortion_corr.py:94-97).  UNPINNED.  Coordinates are quantised to 1/32 px with
 * round-half-even; the 8-tap weights come from OpenCV's interpolateLanczos4 closed form (float32,
 * normalised); the 2-D weight is the float32 product wy*wx; accumulation is row by row,
 * sum += (((w0*s0 + w1*s1) + ...) + w7*s7); taps outside the image read 0. */
static void lanczos4_tab(float tab[32][8]) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    for (int i = 0; i < 32; i++) {
        float x = (float)i * (1.0f / 32.0f);
        float *c = tab[i];
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
int orc_lanczos4_table(float *tab256) { lanczos4_tab((float(*)[8])tab256); return ORC_OK; }
/* n output pixels with their own maps (any shape: a band of rows of a larger warp); the source is always the whole (H,W) plane */
int orc_remap_lanczos4_n(const float *src, int H, int W, const float *mapx, const float *mapy, size_t n, float *dst);
int orc_remap_lanczos4(const float *src, int H, int W, const float *mapx, const float *mapy, float *dst) {
    return orc_remap_lanczos4_n(src, H, W, mapx, mapy, (size_t)H * W, dst);
}
int orc_remap_lanczos4_n(const float *src, int H, int W, const float *mapx, const float *mapy, size_t n, float *dst) {
    float tab[32][8];
    lanczos4_tab(tab);
#pragma omp parallel for
    for (size_t o = 0; o < n; o++) {
            int sx = (int)lrintf(mapx[o] * 32.0f), sy = (int)lrintf(mapy[o] * 32.0f);
            int ix = (sx >> 5) - 3, iy = (sy >> 5) - 3;
            const float *wx = tab[sx & 31], *wy = tab[sy & 31];
            float sum = 0.0f;
            for (int r = 0; r < 8; r++) {
                int yy = iy + r;
                float row = 0.0f;
                for (int c = 0; c < 8; c++) {
                    int xx = ix + c;
                    float s = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? src[(size_t)yy * W + xx] : 0.0f;
                    float t = s * (wy[r] * wx[c]);
                    row = c == 0 ? t : row + t;
                }
                sum = sum + row;
            }
            dst[o] = sum;
        }
    return ORC_OK;
}

/* chan_distortion_corr.py:86-97: per plane table -> np.clip -> remap, in place on an (H,W,3) image.
 * coeffs: planes x 6 doubles (kr0..kr3,kt0,kt1), passed through C float exactly as Cython's float
 * arguments do. */
int orc_warp_rectilinear(float *image, int H, int W, const double *coeffs, int planes, double cxn, double cyn,
                         float scale) {
    if (planes != 3) return ORC_EBADARG;
    size_t N = (size_t)H * W;
    float *tab = falloc(N * 2), *mx = falloc(N), *my = falloc(N), *pl = falloc(N), *out = falloc(N);
    if (!tab || !mx || !my || !pl || !out) { free(tab); free(mx); free(my); free(pl); free(out); return ORC_ENOMEM; }
    for (int c = 0; c < planes; c++) {
        const double *k = coeffs + 6 * c;
        orc_warp_table((float)k[0], (float)k[1], (float)k[2], (float)k[3], (float)k[4], (float)k[5], W, H, (float)cxn,
                       (float)cyn, scale, NULL, tab);
        float xmax = (float)(W - 1), ymax = (float)(H - 1);
#pragma omp parallel for
        for (size_t i = 0; i < N; i++) {
            float a = tab[2 * i], b = tab[2 * i + 1];
            mx[i] = a < 0.0f ? 0.0f : (a > xmax ? xmax : a);
            my[i] = b < 0.0f ? 0.0f : (b > ymax ? ymax : b);
            pl[i] = image[3 * i + c];
        }
        orc_remap_lanczos4(pl, H, W, mx, my, out);
#pragma omp parallel for
        for (size_t i = 0; i < N; i++) image[3 * i + c] = out[i];
    }
    free(tab); free(mx); free(my); free(pl); free(out);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Restated cv2.remap(INTER_LINEAR), BORDER_CONSTANT 0, float32 (corr_ca/ca_removal.py:96-127).  UNPINNED.
 * Coordinates are quantised to 1/32 px (round half even) like every cv2.remap mode; the four weights are the
 * float products of the 1-D taps (1 - f, f); the sum runs in source order, left to right. */
static inline float remap_linear_px(const float *src, int H, int W, float mx, float my) {
    int sx = (int)lrintf(mx * 32.0f), sy = (int)lrintf(my * 32.0f);
    int ix = sx >> 5, iy = sy >> 5;
    float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy};
    float sum = 0.0f;
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) {
            int yy = iy + r, xx = ix + c;
            float v = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? src[(size_t)yy * W + xx] : 0.0f;
            float t = v * (wy[r] * wx[c]);
            sum = (r == 0 && c == 0) ? t : sum + t;
        }
    return sum;
}
int orc_remap_linear(const float *src, int H, int W, const float *mapx, const float *mapy, float *dst) {
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t o = (size_t)y * W + x;
            dst[o] = remap_linear_px(src, H, W, mapx[o], mapy[o]);
        }
    return ORC_OK;
}

/* corr_ca/model/generic.py:56-101,131-163: the coordinate field of a lens model is built for the top-left quadrant
 * (h,w,2) = (dy, dx) relative to the image centre and mirrored with sign flips into the other three; ca_removal.py:96-99
 * then adds (size-1)/2 and clips to [0, size-1].  quad: (h,w,2) float32. */
static inline void ca_map(const float *quad, int H, int W, int y, int x, float *mx, float *my) {
    int h = H / 2, w = W / 2;
    int qy = y < h ? y : H - 1 - y, qx = x < w ? x : W - 1 - x;
    float dy = quad[((size_t)qy * w + qx) * 2], dx = quad[((size_t)qy * w + qx) * 2 + 1];
    if (y >= h) dy = -dy;
    if (x >= w) dx = -dx;
    float cx = (float)(((double)W - 1.0) / 2.0), cy = (float)(((double)H - 1.0) / 2.0);
    float ax = dx + cx, ay = dy + cy, xmax = (float)(W - 1), ymax = (float)(H - 1);
    *mx = ax < 0.0f ? 0.0f : (ax > xmax ? xmax : ax);
    *my = ay < 0.0f ? 0.0f : (ay > ymax ? ymax : ay);
}
static void ca_remap_full(const float *src, int H, int W, const float *quad, float *dst) {
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float mx, my;
            ca_map(quad, H, W, y, x, &mx, &my);
            dst[(size_t)y * W + x] = remap_linear_px(src, H, W, mx, my);
        }
}
/* corr_ca/ca_removal.py:48-131 remove_ca_from_raw for one channel: site (oy,ox) = (0,0) red / (1,1) blue, pos 0 / 3.
 * chan is the (h,w) plane, overwritten with the corrected samples. */
static int ca_channel(float *chan, const float *g_full, int H, int W, const float *quad_g_at_c, const float *quad_c_at_g, float wb,
                      int pos) {
    int h = H / 2, w = W / 2, oy = pos == 0 ? 0 : 1, ox = oy;
    size_t n = (size_t)h * w, N = (size_t)H * W;
    planes_t gq = {0, 0, NULL, NULL, NULL, NULL};
    float *g_at = falloc(N), *blur = falloc(N), *hf = falloc(N), *up = falloc(N), *sub = falloc(n);
    int rc = ORC_ENOMEM;
    if (planes_alloc(&gq, h, w) || !g_at || !blur || !hf || !up || !sub) goto done;
    ca_remap_full(g_full, H, W, quad_g_at_c, g_at);                                   /* :96-100 / :114-118 */
    for (size_t i = 0; i < n; i++) sub[i] = chan[i] * wb;                             /* :102 / :120 */
    orc_gaussian_blur3(g_at, H, W, blur);                                             /* eag.py:170 / :184 */
    for (size_t i = 0; i < N; i++) hf[i] = g_at[i] - blur[i];
    orc_bayer_to_rgbg_f32(g_at, H, W, gq.r, gq.g1, gq.b, gq.g2);
    if ((rc = resample_channel(sub, pos == 0 ? gq.r : gq.b, hf, h, w, pos, up))) goto done;
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float mx, my;
            ca_map(quad_c_at_g, H, W, 2 * i + oy, 2 * j + ox, &mx, &my);               /* :104-108 / :122-126 */
            chan[(size_t)i * w + j] = remap_linear_px(up, H, W, mx, my) / wb;        /* :110 / :128 */
        }
    rc = ORC_OK;
done:
    planes_free(&gq);
    free(g_at); free(blur); free(hf); free(up); free(sub);
    return rc;
}
/* bayer (H,W) float32 in place.  A NULL pair of quadrant fields leaves that channel untouched (lens model None). */
int orc_remove_ca(float *bayer, int H, int W, const float *quad_g_at_r, const float *quad_r_at_g, float wb_r, const float *quad_g_at_b,
                  const float *quad_b_at_g, float wb_b) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    if ((!quad_g_at_r) != (!quad_r_at_g) || (!quad_g_at_b) != (!quad_b_at_g)) return ORC_EBADARG;
    if (!quad_g_at_r && !quad_g_at_b) return ORC_OK;                                  /* :74-75 */
    int h = H / 2, w = W / 2, rc = ORC_ENOMEM;
    planes_t p = {0, 0, NULL, NULL, NULL, NULL};
    float *g_full = falloc((size_t)H * W);
    if (planes_alloc(&p, h, w) || !g_full) goto done;
    orc_bayer_to_rgbg_f32(bayer, H, W, p.r, p.g1, p.b, p.g2);                         /* :84 */
    if ((rc = orc_resample_g_full(p.g1, p.g2, h, w, g_full))) goto done;             /* :85 */
    if (quad_g_at_r && (rc = ca_channel(p.r, g_full, H, W, quad_g_at_r, quad_r_at_g, wb_r, 0))) goto done;
    if (quad_g_at_b && (rc = ca_channel(p.b, g_full, H, W, quad_g_at_b, quad_b_at_g, wb_b, 3))) goto done;
    orc_rgbg_to_bayer_f32(p.r, p.g1, p.b, p.g2, h, w, bayer);                         /* :130 */
    rc = ORC_OK;
done:
    planes_free(&p);
    free(g_full);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * raw_bad_pixel_corr.py:30-65 find_erroneous_pixels_threshold: per CFA plane, np.pad(..., mode="reflect")
 * (= REFLECT_101), eight neighbours, hot where more than min_neighbour_count of them are below
 * (chan - min_delta) (float32 subtraction, python-float weak scalar).  masks: four (h,w) uint8 planes r,g1,b,g2. */
int orc_find_hot_threshold(const float *bayer, int H, int W, float min_delta, int min_count, uint8_t *mr, uint8_t *mg1,
                           uint8_t *mb, uint8_t *mg2) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    int h = H / 2, w = W / 2;
    uint8_t *masks[4] = {mr, mg1, mb, mg2};
    static const int oy[4] = {0, 0, 1, 1}, ox[4] = {0, 1, 1, 0};   /* r, g1, b, g2 */
#pragma omp parallel for
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            for (int pl = 0; pl < 4; pl++) {
                float c = bayer[(size_t)(2 * i + oy[pl]) * W + 2 * j + ox[pl]] - min_delta;
                int cnt = 0;
                for (int di = -1; di <= 1; di++)
                    for (int dj = -1; dj <= 1; dj++) {
                        if (!di && !dj) continue;
                        int ii = b_101(i + di, h), jj = b_101(j + dj, w);
                        if (c > bayer[(size_t)(2 * ii + oy[pl]) * W + 2 * jj + ox[pl]]) cnt++;
                    }
                masks[pl][(size_t)i * w + j] = cnt > min_count;
            }
    return ORC_OK;
}

/* raw_correction.py:25-62 flat_frame_correction, per CFA plane (correct_channel :42-58):
 *   out = (chan * mean_flat) / flat ; if every value is inf: out = chan ; +inf -> max of the finite values ;
 *   negative -> 0 ; clamp_high: > 1 -> 1.  mean[4] = np.mean of the four flat planes, computed by the caller
 *   with NumPy on identically strided views (float32 pairwise summation order is NumPy's). */
int orc_flat_field(const float *bayer, const float *flat, int H, int W, const float mean[4], int clamp_high, float *out) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return ORC_EBADARG;
    static const int oy[4] = {0, 0, 1, 1}, ox[4] = {0, 1, 1, 0};
    int h = H / 2, w = W / 2;
    for (int pl = 0; pl < 4; pl++) {
        float maxfin = -INFINITY; int all_inf = 1;
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                size_t o = (size_t)(2 * i + oy[pl]) * W + 2 * j + ox[pl];
                float v = (bayer[o] * mean[pl]) / flat[o];
                out[o] = v;
                if (!isinf(v)) all_inf = 0;
                if (isfinite(v) && v > maxfin) maxfin = v;
            }
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                size_t o = (size_t)(2 * i + oy[pl]) * W + 2 * j + ox[pl];
                float v = out[o];
                if (all_inf) { out[o] = bayer[o]; continue; }
                if (v == INFINITY) v = maxfin;
                if (v < 0.0f) v = 0.0f;
                if (clamp_high && v > 1.0f) v = 1.0f;
                out[o] = v;
            }
    }
    return ORC_OK;
}

#ifdef _OPENMP
#include <omp.h>
int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int orc_set_threads(int n) { (void)n; return 1; }
#endif
int orc_abi_version(void) { return 1; }
