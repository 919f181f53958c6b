// k_ahd.hip -- AHD ("Best") demosaic for gfx950: Bayer mosaic -> direction-selected RGB (kernel A)
// and the 5x5-median chroma post-process stages with the colour tail (kernel B).
//
// Follows debayer/ahd.py:14-170 and debayer/ahd_homogeneity_cython.pyx:22-58 of pySP, with the
// OpenCV calls restated as in oracle/pysp_oracle.c (same op order, bit for bit).
//
// Kernel A works on 2x2 CFA quads.  One workgroup = one 64x32 px output tile (32x16 quads):
//   P0  mosaic * wb -> four de-interleaved quarter planes in LDS (halo 3 quads, symmetric border)
//   P1  green at red/blue sites, horizontal and vertical, and the colour differences D = sub - g
//       (halo 2 quads; positions outside the image hold the REFLECT_101 value the 3x3 plane
//       filters of resample_channel expect)
//   P2  one thread per quad (halo 1 quad): high-pass of green, photosite-aware resampling of
//       R and B in both directions, second white balance + float64 CCM + Lab (registers)
//   P3  Lab -> LDS (aliasing the P0/P1 planes), homogeneity vote for both directions
//   P4  3x3 box of the votes, H/V selection, optional colour tail, store
// Image-border rules (three of them coexist) are applied at true image edges only.
#include "demosaic_common.h"
#include "kernels.h"

namespace {

constexpr int TQX = 32, TQY = 16;                 // output quads per tile
constexpr int MWX = TQX + 6, MWY = TQY + 6;       // mosaic planes, halo 3 quads
constexpr int GX = TQX + 4, GY = TQY + 4;         // green / difference planes, halo 2 quads
constexpr int LQX = TQX + 2, LQY = TQY + 2;       // Lab region in quads (halo 1 quad = 2 px)
constexpr int LPX = 2 * LQX, LPY = 2 * LQY;       // Lab region in pixels (68 x 36)
constexpr int MPX = 2 * TQX + 2, MPY = 2 * TQY + 2;  // vote maps, halo 1 px (66 x 34)
constexpr int NT_A = 640;                         // >= LQX*LQY = 612

constexpr float AH0 = -0x1.053316p-2f, AH1 = 0x1p-1f, AH2 = 0x1.053316p-1f;  // ahd.py:89-94

enum { P_R = 0, P_G1 = 1, P_G2 = 2, P_B = 3 };
enum { Q_GHR = 0, Q_GVR, Q_GHB, Q_GVB, Q_DHR, Q_DVR, Q_DHB, Q_DVB };

constexpr int LDS_A_FLOATS = 4 * MWY * MWX + 8 * GY * GX;   // 3344 + 5760 = 9104
constexpr int LDS_LAB_FLOATS = 6 * LPY * LPX;               // 14688
constexpr int LDS_MAIN_FLOATS = LDS_LAB_FLOATS > LDS_A_FLOATS ? LDS_LAB_FLOATS : LDS_A_FLOATS;

// ahd.py:32-62: second white balance, CCM without clip, (HDR: luma + x/(1+x)), Lab
DEVI void homog_lab(float r, float g, float b, const float wb[3], const double* M, int hdr, float& L, float& A, float& Bq) {
    float rr = r * wb[0], gg = g * wb[1], bb = b * wb[2];
    float sr = ccm_row(M, rr, gg, bb), sg = ccm_row(M + 3, rr, gg, bb), sb = ccm_row(M + 6, rr, gg, bb);
    if (hdr) {
        float luma = 0.2126f * sr + 0.7152f * sg + 0.0722f * sb;
        sr = sr / (1.0f + sr); sg = sg / (1.0f + sg); sb = sb / (1.0f + sb);
        rgb2lab_px(sr, sg, sb, L, A, Bq);
        L = luma;
    } else {
        rgb2lab_px(sr, sg, sb, L, A, Bq);
    }
}

}  // namespace

struct AhdParams {
    const float* bayer;
    float* out;          // (H,W,3)
    int H, W;
    float wb[3];
    int hdr;
    int tail;            // colour tail applied to the selected pixel (only when no median stage follows)
    Ccm ccm;
};

__global__ void __launch_bounds__(NT_A) k_ahd_select(AhdParams p) {
    __shared__ float lds_main[LDS_MAIN_FLOATS];
    __shared__ unsigned char lds_map[2][MPY][MPX];

    float* mw = lds_main;                       // [4][MWY][MWX]
    float* gq = lds_main + 4 * MWY * MWX;       // [8][GY][GX]
    float* lab = lds_main;                      // [6][LPY][LPX]  (aliases mw/gq after P2)

    const int tid = threadIdx.x;
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    const int tq0x = blockIdx.x * TQX, tq0y = blockIdx.y * TQY;
    const double* M = p.ccm.m;

    // ---- P0: white-balanced mosaic planes, symmetric (edge-duplicating) reflect per plane (ahd.py:77-80)
    for (int idx = tid; idx < 4 * MWY * MWX; idx += NT_A) {
        int ry = idx / (2 * MWX), rx = idx - ry * (2 * MWX);
        int my = ry >> 1, mx = rx >> 1, dy = ry & 1, dx = rx & 1;
        int qi = b_sym(tq0y - 3 + my, h), qj = b_sym(tq0x - 3 + mx, w);
        int plane = dy ? (dx ? P_B : P_G2) : (dx ? P_G1 : P_R);
        float wbv = plane == P_R ? p.wb[0] : (plane == P_B ? p.wb[2] : p.wb[1]);
        mw[(plane * MWY + my) * MWX + mx] = p.bayer[(size_t)(2 * qi + dy) * W + (2 * qj + dx)] * wbv;
    }
    __syncthreads();

    // ---- P1: directional green at R/B sites (ahd.py:97-102) and D = sub - g (eag.py:142)
    for (int idx = tid; idx < GY * GX; idx += NT_A) {
        int gy = idx / GX, gx = idx - gy * GX;
        int ri = b_101(tq0y - 2 + gy, h), rj = b_101(tq0x - 2 + gx, w);   // REFLECT_101 on the quarter plane
        int a = ri - (tq0y - 3), c = rj - (tq0x - 3);
        if (a < 1 || a > MWY - 2 || c < 1 || c > MWX - 2) continue;        // never consumed by a valid output
#define MWAT(pl, yy, xx) mw[((pl) * MWY + (yy)) * MWX + (xx)]
        float rc = MWAT(P_R, a, c), bc = MWAT(P_B, a, c);
        float ghr = (((MWAT(P_R, a, c - 1) * AH0 + MWAT(P_G1, a, c - 1) * AH1) + rc * AH2) + MWAT(P_G1, a, c) * AH1) + MWAT(P_R, a, c + 1) * AH0;
        float gvr = (((MWAT(P_R, a - 1, c) * AH0 + MWAT(P_G2, a - 1, c) * AH1) + rc * AH2) + MWAT(P_G2, a, c) * AH1) + MWAT(P_R, a + 1, c) * AH0;
        float ghb = (((MWAT(P_B, a, c - 1) * AH0 + MWAT(P_G2, a, c) * AH1) + bc * AH2) + MWAT(P_G2, a, c + 1) * AH1) + MWAT(P_B, a, c + 1) * AH0;
        float gvb = (((MWAT(P_B, a - 1, c) * AH0 + MWAT(P_G1, a, c) * AH1) + bc * AH2) + MWAT(P_G1, a + 1, c) * AH1) + MWAT(P_B, a + 1, c) * AH0;
        gq[Q_GHR * GY * GX + idx] = ghr; gq[Q_GVR * GY * GX + idx] = gvr;
        gq[Q_GHB * GY * GX + idx] = ghb; gq[Q_GVB * GY * GX + idx] = gvb;
        gq[Q_DHR * GY * GX + idx] = rc - ghr; gq[Q_DVR * GY * GX + idx] = rc - gvr;
        gq[Q_DHB * GY * GX + idx] = bc - ghb; gq[Q_DVB * GY * GX + idx] = bc - gvb;
    }
    __syncthreads();

    // ---- P2: one quad per thread
    const int lqy = tid / LQX, lqx = tid - lqy * LQX;
    const int qi = tq0y - 1 + lqy, qj = tq0x - 1 + lqx;
    const bool active = tid < LQX * LQY && qi >= 0 && qi < h && qj >= 0 && qj < w;
    float rgbh[4][3], rgbv[4][3], labh[4][3], labv[4][3];
    if (active) {
        const int gy = lqy + 1, gx = lqx + 1, my = lqy + 2, mx = lqx + 2;
        const bool at_top = qi == 0, at_bot = qi == h - 1, at_left = qj == 0, at_right = qj == w - 1;
        // green samples of the 4x4 window shared by both directions
        float g1_l = MWAT(P_G1, my, mx - 1), g1_c = MWAT(P_G1, my, mx), g1_dl = MWAT(P_G1, my + 1, mx - 1), g1_d = MWAT(P_G1, my + 1, mx);
        float g2_u = MWAT(P_G2, my - 1, mx), g2_ur = MWAT(P_G2, my - 1, mx + 1), g2_c = MWAT(P_G2, my, mx), g2_r = MWAT(P_G2, my, mx + 1);
#pragma unroll
        for (int dir = 0; dir < 2; dir++) {
            const float* gR = gq + (dir ? Q_GVR : Q_GHR) * GY * GX;
            const float* gB = gq + (dir ? Q_GVB : Q_GHB) * GY * GX;
            const float* dR = gq + (dir ? Q_DVR : Q_DHR) * GY * GX;
            const float* dB = gq + (dir ? Q_DVB : Q_DHB) * GY * GX;
            Win3 wgr = load_win<GX>(gR, gy, gx), wgb = load_win<GX>(gB, gy, gx);
            // full-resolution green, rows 2qi-1..2qi+2, cols 2qj-1..2qj+2
            float Wn[4][4] = {{wgb.v[0][0], g2_u, wgb.v[0][1], g2_ur},
                              {g1_l, wgr.v[1][1], g1_c, wgr.v[1][2]},
                              {wgb.v[1][0], g2_c, wgb.v[1][1], g2_r},
                              {g1_dl, wgr.v[2][1], g1_d, wgr.v[2][2]}};
            // GaussianBlur border = REFLECT_101 at full resolution: row -1 -> row 1, row H -> row H-2
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
            float hf[4];
            highpass_quad(Wn, hf);
            float fg[4], fd[4], rr[4], bb[4];
            filt_base_tl(wgr, fg);
            { Win3 wd = load_win<GX>(dR, gy, gx); filt_base_tl(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) rr[k] = fd[k] + (fg[k] + hf[k]);      // eag.py:141,143
            filt_base_br(wgb, fg);
            { Win3 wd = load_win<GX>(dB, gy, gx); filt_base_br(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) bb[k] = fd[k] + (fg[k] + hf[k]);
            float gg[4] = {wgr.v[1][1], g1_c, g2_c, wgb.v[1][1]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float L, A, Bq;
                homog_lab(rr[k], gg[k], bb[k], p.wb, M, p.hdr, L, A, Bq);
                if (dir == 0) { rgbh[k][0] = rr[k]; rgbh[k][1] = gg[k]; rgbh[k][2] = bb[k]; labh[k][0] = L; labh[k][1] = A; labh[k][2] = Bq; }
                else          { rgbv[k][0] = rr[k]; rgbv[k][1] = gg[k]; rgbv[k][2] = bb[k]; labv[k][0] = L; labv[k][1] = A; labv[k][2] = Bq; }
            }
        }
    }
#undef MWAT
    __syncthreads();   // everyone is done reading mw/gq: Lab may now overwrite them

    // ---- P3a: Lab to LDS
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int py = 2 * lqy + (k >> 1), px = 2 * lqx + (k & 1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                lab[(c * LPY + py) * LPX + px] = labh[k][c];
                lab[((3 + c) * LPY + py) * LPX + px] = labv[k][c];
            }
        }
    }
    __syncthreads();

    // ---- P3b: homogeneity vote (pyx:22-58) for the pixels of this quad that lie within 1 px of the tile.
    // Lab is padded with BORDER_REFLECT (ahd.py:64): neighbours outside the image duplicate the edge pixel.
    const int ty0 = 2 * tq0y, tx0 = 2 * tq0x;     // tile origin in pixels
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int y = 2 * qi + (k >> 1), x = 2 * qj + (k & 1);
            int myy = y - (ty0 - 1), mxx = x - (tx0 - 1);
            if (myy < 0 || myy >= MPY || mxx < 0 || mxx >= MPX) continue;
            int ly[3], lx[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                ly[d] = b_sym(y - 1 + d, H) - (ty0 - 2);
                lx[d] = b_sym(x - 1 + d, W) - (tx0 - 2);
            }
#pragma unroll
            for (int dir = 0; dir < 2; dir++) {
                const float* Lp = lab + (dir * 3 + 0) * LPY * LPX;
                const float* Ap = lab + (dir * 3 + 1) * LPY * LPX;
                const float* Bp = lab + (dir * 3 + 2) * LPY * LPX;
                float rl = Lp[ly[1] * LPX + lx[1]], ra = Ap[ly[1] * LPX + lx[1]], rb = Bp[ly[1] * LPX + lx[1]];
                int n1 = dir ? ly[0] * LPX + lx[1] : ly[1] * LPX + lx[0];
                int n2 = dir ? ly[2] * LPX + lx[1] : ly[1] * LPX + lx[2];
                float e1 = fabsf(rl - Lp[n1]), e2 = fabsf(rl - Lp[n2]);
                float da1 = ra - Ap[n1], db1 = rb - Bp[n1], da2 = ra - Ap[n2], db2 = rb - Bp[n2];
                float c1 = da1 * da1 + db1 * db1, c2 = da2 * da2 + db2 * db2;
                float el = e2 > e1 ? e2 : e1, ec = c2 > c1 ? c2 : c1;
                int cnt = 0;
#pragma unroll
                for (int wy = 0; wy < 3; wy++)
#pragma unroll
                    for (int wx = 0; wx < 3; wx++) {
                        int o = ly[wy] * LPX + lx[wx];
                        float da = Ap[o] - ra, db = Bp[o] - rb;
                        bool ok = (Lp[o] - rl <= el) && (da * da + db * db <= ec);
                        cnt += ok ? 1 : 0;
                    }
                lds_map[dir][myy][mxx] = (unsigned char)cnt;
            }
        }
    }
    __syncthreads();

    // ---- P4: 3x3 box (cv2.blur, REFLECT_101; integer sums order like the float means), select, store
    if (active && lqy >= 1 && lqy <= TQY && lqx >= 1 && lqx <= TQX) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int y = 2 * qi + (k >> 1), x = 2 * qj + (k & 1);
            int sh = 0, sv = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = b_101(y + dy, H) - (ty0 - 1), xx = b_101(x + dx, W) - (tx0 - 1);
                    sh += lds_map[0][yy][xx];
                    sv += lds_map[1][yy][xx];
                }
            float c = sh < sv ? 1.0f : 0.0f, nc = 1.0f - c;            // ahd.py:139-145, literally
            float r = rgbh[k][0] * c + rgbv[k][0] * nc;
            float g = rgbh[k][1] * c + rgbv[k][1] * nc;
            float b = rgbh[k][2] * c + rgbv[k][2] * nc;
            colour_tail(p.tail, M, r, g, b);
            float* o = p.out + ((size_t)y * W + x) * 3;
            o[0] = r; o[1] = g; o[2] = b;
        }
    }
}

// ================================================================================================
// Kernel B: one chroma post-process stage (ahd.py:148-161) + optional colour tail.
//   r' = med5(r-g)+g ; b' = med5(b-g)+g ; g' = (med5(g-r') + med5(g-b') + r' + b') / 2
// cv2.medianBlur(.,5): exact 5x5 median, BORDER_REPLICATE.  Tile 64x32 px, halo 4 px.
namespace {
constexpr int BTX = 64, BTY = 32;
constexpr int B4X = BTX + 8, B4Y = BTY + 8;     // r-g, b-g planes (halo 4)
constexpr int B2X = BTX + 4, B2Y = BTY + 4;     // g-r', g-b' planes (halo 2)
constexpr int NT_B = 512;

// Median of 25 by a selection network on min/max (exact; order independent).
#define CE(a, b) { float _t = fminf(v[a], v[b]); v[b] = fmaxf(v[a], v[b]); v[a] = _t; }
DEVI float median25(float v[25]) {
    // Devillard's opt_med25 exchange list
    CE(0, 1) CE(3, 4) CE(2, 4) CE(2, 3) CE(6, 7) CE(5, 7) CE(5, 6) CE(9, 10) CE(8, 10) CE(8, 9)
    CE(12, 13) CE(11, 13) CE(11, 12) CE(15, 16) CE(14, 16) CE(14, 15) CE(18, 19) CE(17, 19) CE(17, 18)
    CE(21, 22) CE(20, 22) CE(20, 21) CE(23, 24) CE(2, 5) CE(3, 6) CE(0, 6) CE(0, 3) CE(4, 7) CE(1, 7) CE(1, 4)
    CE(11, 14) CE(8, 14) CE(8, 11) CE(12, 15) CE(9, 15) CE(9, 12) CE(13, 16) CE(10, 16) CE(10, 13)
    CE(20, 23) CE(17, 23) CE(17, 20) CE(21, 24) CE(18, 24) CE(18, 21) CE(19, 22) CE(8, 17) CE(9, 18) CE(0, 18)
    CE(0, 9) CE(10, 19) CE(1, 19) CE(1, 10) CE(11, 20) CE(2, 20) CE(2, 11) CE(12, 21) CE(3, 21) CE(3, 12)
    CE(13, 22) CE(4, 22) CE(4, 13) CE(14, 23) CE(5, 23) CE(5, 14) CE(15, 24) CE(6, 24) CE(6, 15) CE(7, 16)
    CE(7, 19) CE(13, 21) CE(15, 23) CE(7, 13) CE(7, 15) CE(1, 9) CE(3, 11) CE(5, 17) CE(11, 17) CE(9, 17)
    CE(4, 10) CE(6, 12) CE(7, 14) CE(4, 6) CE(4, 7) CE(12, 14) CE(10, 14) CE(6, 7) CE(10, 12) CE(6, 10)
    CE(6, 17) CE(12, 17) CE(7, 17) CE(7, 10) CE(12, 18) CE(7, 12) CE(10, 18) CE(12, 20) CE(10, 20) CE(10, 12)
    return v[12];
}
#undef CE
}  // namespace

struct MedParams {
    const float* in;   // (H,W,3)
    float* out;        // (H,W,3)
    int H, W;
    int tail;
    Ccm ccm;
};

__global__ void __launch_bounds__(NT_B) k_ahd_median_stage(MedParams p) {
    __shared__ float s_g[B4Y][B4X], s_drg[B4Y][B4X], s_dbg[B4Y][B4X];   // g, r-g, b-g   (halo 4)
    __shared__ float s_r1[B2Y][B2X], s_b1[B2Y][B2X];                     // r', b'        (halo 2)
    const int tid = threadIdx.x, H = p.H, W = p.W;
    const int tx0 = blockIdx.x * BTX, ty0 = blockIdx.y * BTY;

    for (int idx = tid; idx < B4Y * B4X; idx += NT_B) {
        int ly = idx / B4X, lx = idx - ly * B4X;
        int y = b_rep(ty0 - 4 + ly, H), x = b_rep(tx0 - 4 + lx, W);
        const float* s = p.in + ((size_t)y * W + x) * 3;
        float r = s[0], g = s[1], b = s[2];
        s_g[ly][lx] = g; s_drg[ly][lx] = r - g; s_dbg[ly][lx] = b - g;
    }
    __syncthreads();
    // r', b' on the halo-2 region.  medianBlur replicates the border of ITS input plane, so r', b' at a
    // position outside the image are those of the clamped position: evaluate the window there.
    for (int idx = tid; idx < B2Y * B2X; idx += NT_B) {
        int oy = idx / B2X, ox = idx - oy * B2X;
        int ly = b_rep(ty0 - 2 + oy, H) - (ty0 - 2), lx = b_rep(tx0 - 2 + ox, W) - (tx0 - 2);
        float v[25];
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = s_drg[ly + dy][lx + dx];
        float g = s_g[ly + 2][lx + 2];
        float r1 = median25(v) + g;
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = s_dbg[ly + dy][lx + dx];
        float b1 = median25(v) + g;
        s_r1[oy][ox] = r1; s_b1[oy][ox] = b1;
    }
    __syncthreads();
    for (int idx = tid; idx < BTY * BTX; idx += NT_B) {
        int ly = idx / BTX, lx = idx - ly * BTX;
        int y = ty0 + ly, x = tx0 + lx;
        if (y >= H || x >= W) continue;
        float v[25];
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = s_g[ly + 2 + dy][lx + 2 + dx] - s_r1[ly + dy][lx + dx];
        float m1 = median25(v);
#pragma unroll
        for (int dy = 0; dy < 5; dy++)
#pragma unroll
            for (int dx = 0; dx < 5; dx++) v[dy * 5 + dx] = s_g[ly + 2 + dy][lx + 2 + dx] - s_b1[ly + dy][lx + dx];
        float m2 = median25(v);
        float r = s_r1[ly + 2][lx + 2], b = s_b1[ly + 2][lx + 2];
        float g = (((m1 + m2) + r) + b) / 2.0f;
        colour_tail(p.tail, p.ccm.m, r, g, b);
        float* o = p.out + ((size_t)y * W + x) * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}

// ------------------------------------------------------------------------------------------------
int launch_ahd(hipStream_t st, const float* d_bayer, int H, int W, const float wb[3], const double M[9], int hdr, int stages,
               int tail, float* d_out, float* d_tmp0, float* d_tmp1, Timeline* tl) {
    AhdParams a;
    a.bayer = d_bayer; a.H = H; a.W = W; a.hdr = hdr;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M[i];
    if (stages < 0) stages = 0;
    // ping-pong so that the last kernel writes d_out
    float* bufs[2] = {d_tmp0, d_tmp1};
    a.out = stages == 0 ? d_out : bufs[0];
    a.tail = stages == 0 ? tail : 0;
    dim3 ga((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY);
    if (tl) tl->begin(st, "k_ahd_select");
    hipLaunchKernelGGL(k_ahd_select, ga, dim3(NT_A), 0, st, a);
    if (tl) tl->end(st);
    const float* cur = a.out;
    dim3 gb((W + BTX - 1) / BTX, (H + BTY - 1) / BTY);
    for (int s = 0; s < stages; s++) {
        MedParams m;
        m.in = cur; m.H = H; m.W = W; m.ccm = a.ccm;
        bool last = s == stages - 1;
        m.out = last ? d_out : bufs[(s + 1) & 1];
        m.tail = last ? tail : 0;
        if (tl) tl->begin(st, "k_ahd_median_stage");
        hipLaunchKernelGGL(k_ahd_median_stage, gb, dim3(NT_B), 0, st, m);
        if (tl) tl->end(st);
        cur = m.out;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
