// k_ahd.hip -- AHD ("Best") demosaic for gfx950: Bayer mosaic -> direction-selected RGB (kernel A)
// and the 5x5-median chroma post-process stages with the colour tail (kernel B).
//
// Follows debayer/ahd.py:14-170 and debayer/ahd_homogeneity_cython.pyx:22-58 of pySP, with the
// OpenCV calls restated as in oracle/pysp_oracle.c (same op order, bit for bit).
//
// Kernel A works on 2x2 CFA quads.  One workgroup = one 28x28 px output tile (14x14 quads, 16x16 threads):
//   P0  mosaic * wb -> four de-interleaved quarter planes in LDS (halo 3 quads, symmetric border)
//   P1  green at red/blue sites, horizontal and vertical, and the colour differences D = sub - g
//       (halo 2 quads; positions outside the image hold the REFLECT_101 value the 3x3 plane
//       filters of resample_channel expect)
//   then, for the horizontal and the vertical candidate in turn (one code path, unrolled twice):
//   P2  one thread per quad (halo 1 quad): high-pass of green, photosite-aware resampling of
//       R and B, second white balance + float64 CCM + Lab, all in registers; Lab -> LDS
//   P3  homogeneity vote from a 4x4 Lab window per quad (edge cells by select), in two row pieces;
//       the Lab buffer is its own LDS section (round 3), the vertical green planes are built into the plane buffer meanwhile
//   finally
//   P4  3x3 box of the packed votes (integer), H/V selection (both candidates still in registers), optional colour tail, store.
// Round 4, Lab mode 1: the Lab buffer holds 8-byte cells { L, a' | b' << 16 } (the interpolated table integers), the chroma distances of the vote are integer
// (v_pk_sub_i16 + v_dot2), exact while below 2^24, with a wave-uniform float redo otherwise: 70 VGPRs, 20.8 KB of LDS, SEVEN workgroups per CU
// (round 3: 71 VGPRs, 25.5 KB, six; round 2: 92 VGPRs, 20.6 KB, five).  Template value LAB = 2 keeps round 3's float planes (pysp_ctx_set_lab_layout).
// -DAHD_TQX / -DAHD_TQY / -DAHD_QPT build the other tile shapes and the two-quads-per-thread form measured in DESIGN.md 7.0 (c).
// Image-border rules (three of them coexist) are applied at true image edges only.
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "demosaic_common.h"
#include "kernels.h"

namespace {

#ifndef AHD_TQX
#define AHD_TQX 14                                // measured on MI355X: 256-thread workgroups beat 384/512/640-thread ones
#define AHD_TQY 14                                // by 1.3-1.9x despite the larger halo share
#endif
constexpr int TQX = AHD_TQX, TQY = AHD_TQY;       // output quads per tile; the 1-quad halo makes (TQX+2)x(TQY+2) threads
constexpr int MWX = TQX + 6, MWY = TQY + 6;       // mosaic planes, halo 3 quads
constexpr int GX = TQX + 4, GY = TQY + 4;         // green / difference planes, halo 2 quads
#ifndef AHD_LANES8
#define AHD_LANES8 0                              // experiment (round 4): a wave covers 8 x 8 quads instead of 4 rows of 16, and the planes get a row stride of 24 floats,
#endif                                            // so that the 4-byte window reads of P2 meet no LDS bank twice (rows at bank offsets 0, 24, 48, 8, 32, 56, 16, 40)
constexpr int GXS = AHD_LANES8 ? 24 : GX;         // row stride of the green / difference planes
constexpr int LQX = TQX + 2, LQY = TQY + 2;       // Lab region in quads (halo 1 quad = 2 px)
constexpr int LPS = 2 * LQX + 2, LPR = 2 * LQY + 2;  // Lab plane stride / rows: region + 1 px guard ring (34 x 34)
constexpr int MPS = 2 * TQX + 4, MPR = 2 * TQY + 2;  // packed vote map (halo 1 px), stride 32 (8-byte aligned rows)
#ifndef AHD_QPT
#define AHD_QPT 1                                 // quads of the halo-1 region per thread (2: the thread's second quad lies LQY / 2 quad rows below the first)
#endif
constexpr int QPT = AHD_QPT;
static_assert((LQX * LQY) % QPT == 0 && LQY % QPT == 0, "the region splits into QPT row blocks");
constexpr int NT_A = LQX * LQY / QPT;              // 256: one thread per quad (QPT = 1) of the halo-1 region
#ifndef AHD_MIN_WAVES
#define AHD_MIN_WAVES 1                           // the allocator reaches 92 VGPRs unforced; forcing a bound on earlier versions only spilled
#endif

#ifndef AHD_I16
#define AHD_I16 1                                 // build switch for A/B runs: 0 turns the packed layout (template value LAB = 1) back into round 3's float planes
#endif
// Template value LAB of k_ahd_select: 0 closed-form Lab (tables in LDS); 1 OpenCV-4.10 LUT path, Lab buffer as packed cells, integer chroma votes (round 4, the
// default); 2 the same LUT path with round 3's three float planes and float votes (pysp_ctx_set_lab_layout: the form for adversarial colour noise, where
// every wave of form 1 has to redo its votes in float arithmetic)

constexpr float AH0 = -0x1.053316p-2f, AH1 = 0x1p-1f, AH2 = 0x1.053316p-1f;  // ahd.py:89-94

// Diagnostic build only (-DAHD_STAMPS, tools/phase_stamps.py): wave 0..3 of every workgroup writes the shader clock (s_memtime) at the phase
// boundaries of k_ahd_select into a device array that no kernel reads; pysp_debug_ahd_stamps copies it out.  Never part of the product build.
#ifdef AHD_STAMPS
constexpr int AHD_NSTAMP = 16, AHD_STAMP_WAVES = 1 << 17;
__device__ unsigned long long g_ahd_stamps[(size_t)AHD_NSTAMP * AHD_STAMP_WAVES];
#define AHD_STAMP(i) do { if ((threadIdx.x & 63) == 0) { \
        const unsigned wv_ = (blockIdx.y * gridDim.x + blockIdx.x) * (NT_A / 64) + (threadIdx.x >> 6); \
        if (wv_ < (unsigned)AHD_STAMP_WAVES) g_ahd_stamps[(size_t)wv_ * AHD_NSTAMP + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define AHD_STAMP(i) do { } while (0)
#endif

enum { P_R = 0, P_G1 = 1, P_G2 = 2, P_B = 3 };

// ahd.py:32-62: second white balance, CCM without clip, (HDR: luma + x/(1+x)), Lab
// LAB: which restatement of cv2.cvtColor stands in (0 closed form from the LDS tables, 1 OpenCV 4.10's LUT + trilinear path from `lut`)
template <int LAB>
DEVI void homog_lab(LabTab lt, const uint4* lut, float r, float g, float b, const float wb[3], const double* M, int hdr, float& L, float& A, float& Bq) {
    float rr = r * wb[0], gg = g * wb[1], bb = b * wb[2];
    float sr = ccm_row(M, rr, gg, bb), sg = ccm_row(M + 3, rr, gg, bb), sb = ccm_row(M + 6, rr, gg, bb);
    float luma = 0.0f;
    if (hdr) {
        luma = 0.2126f * sr + 0.7152f * sg + 0.0722f * sb;
        sr = sr / (1.0f + sr); sg = sg / (1.0f + sg); sb = sb / (1.0f + sb);
    }
    if (LAB != 0) rgb2lab_cv410(lut, sr, sg, sb, L, A, Bq); else rgb2lab_px(lt, sr, sg, sb, L, A, Bq);
    if (hdr) L = luma;
}

// The same for the packed layout of Lab mode 1 (round 4): L as float, chroma as the two interpolated table integers a' | b' << 16
DEVI void homog_lab_pk(const uint4* lut, float r, float g, float b, const float wb[3], const double* M, int hdr, float& L, unsigned& ab) {
    float rr = r * wb[0], gg = g * wb[1], bb = b * wb[2];
    float sr = ccm_row(M, rr, gg, bb), sg = ccm_row(M + 3, rr, gg, bb), sb = ccm_row(M + 6, rr, gg, bb);
    float luma = 0.0f;
    if (hdr) {
        luma = 0.2126f * sr + 0.7152f * sg + 0.0722f * sb;
        sr = sr / (1.0f + sr); sg = sg / (1.0f + sg); sb = sb / (1.0f + sb);
    }
    rgb2lab_cv410_pk(lut, sr, sg, sb, L, ab);
    if (hdr) L = luma;
}

// 4x4 window of one Lab plane around a quad: rows/cols -1..2 relative to the quad's top-left pixel.
// The plane is stored with a one-pixel guard ring so that the window starts at an even (8-byte aligned)
// index; BORDER_REFLECT (ahd.py:64) duplicates the edge pixel: cells outside the image are replaced.
DEVI void load_lab_win(const float* plane, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right, float w[4][4]) {
    const float* p = plane + (2 * lqy) * LPS + 2 * lqx;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float2 a = *reinterpret_cast<const float2*>(p + r * LPS);
        float2 b = *reinterpret_cast<const float2*>(p + r * LPS + 2);
        w[r][0] = a.x; w[r][1] = a.y; w[r][2] = b.x; w[r][3] = b.y;
    }
    if (at_top | at_bot | at_left | at_right) {   // interior waves skip the 16 selects
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (at_top) w[0][c] = w[1][c];
            if (at_bot) w[3][c] = w[2][c];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (at_left) w[r][0] = w[r][1];
            if (at_right) w[r][3] = w[r][2];
        }
    }
}
// The same window in two pieces, so that the fourth row can take the registers of the first once the quad's upper pixel pair has voted:
// rows [R0, R1) of the window (R0 = 0: rows 0-2 with the top rule; R0 = 3: row 3 with the bottom rule, which copies row 2).
template <int R0, int R1>
DEVI void load_lab_rows(const float* plane, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right, float w[4][4]) {
    const float* p = plane + (2 * lqy) * LPS + 2 * lqx;
#pragma unroll
    for (int r = R0; r < R1; r++) {
        float2 a = *reinterpret_cast<const float2*>(p + r * LPS);
        float2 b = *reinterpret_cast<const float2*>(p + r * LPS + 2);
        w[r][0] = a.x; w[r][1] = a.y; w[r][2] = b.x; w[r][3] = b.y;
    }
    if (at_top | at_bot | at_left | at_right) {   // interior waves skip the selects
#pragma unroll
        for (int r = R0; r < R1; r++) {
            if (at_left) w[r][0] = w[r][1];
            if (at_right) w[r][3] = w[r][2];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (R0 == 0 && at_top) w[0][c] = w[1][c];
            if (R1 == 4 && at_bot) w[3][c] = w[2][c];      // row 2 already carries its left / right rule
        }
    }
}

// pyx:22-58 for the four pixels of a quad.  DIR 0: epsilons from the left/right neighbours (map_h),
// DIR 1: from the up/down neighbours (map_v).  The centre and the two epsilon neighbours always count:
// x - c <= max(|c - x|, .) and the chroma distance of a neighbour is one of the two maximised squares
// (d*d == (-d)*(-d) bit for bit), so only the other six window cells are tested.
// (Round 1 measured and dropped the sharing of the quad's six pixel pairs below: it cost 12-16 VGPRs in a kernel that had 137-165 of them.
// With the vote split into two row pieces it costs none: vote_quad below.)
// c + (lane's bit of m): the two compares of a cell leave their results in scalar register pairs, their AND is a scalar instruction, and
// v_addc_co_u32 takes such a pair as its carry-in -- one vector instruction per cell instead of v_cndmask(0, 1) + v_add
DEVI int add_lane_bit(int c, unsigned long long m) {
    int d; unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(d), "=s"(carry_out) : "v"(c), "s"(m));
    return d;
}
// (Round 3, measured and dropped: counting the six cell masks of a pixel on the scalar unit -- two full adders and a half adder on the 64-bit masks, 17 scalar
// instructions, then three v_addc instead of six to turn the bit-sliced count into a per-lane number: 24 fewer vector instructions per thread, 136 more scalar
// ones, 0.5992 -> 0.6011 ms per step in six runs each: the scalar instructions are not free.)
// Round 3: the chroma distances BETWEEN the four pixels of the quad are computed once for both pixels of a pair: (a_q - a_p)^2 + (b_q - b_p)^2 has
// the same bits seen from p and from q, every pixel of the quad lies in every other's 3x3 window, and of the six pairs each is needed twice (as a tested
// cell or as an epsilon neighbour).  The pair is computed by its lower-numbered pixel and reused by the higher one (pc[]: (0,1) (2,3) (0,2) (1,3) (0,3) (1,2));
// the four pairs that join the upper and the lower pixel row cross from the first vote piece to the second in four registers
// (60 of the kernel's 2 240 instructions, 71 VGPRs; 0.6110 -> 0.6053 ms per step in six A/B runs each, profiles/r3_ab_select_pairshare.log).
constexpr int quad_pair_id(int a, int b) {
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return lo == 0 ? (hi == 1 ? 0 : hi == 2 ? 2 : 4) : lo == 1 ? (hi == 3 ? 3 : 5) : 1;
}
template <int DIR, int K0, int K1, bool SHARE = true>
DEVI void vote_quad(const float wl[4][4], const float wa[4][4], const float wq[4][4], int cnt[4], float pc[6]) {
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        const float rl = wl[cy][cx], ra = wa[cy][cx], rb = wq[cy][cx];
        // chroma distance to window position (y, x): from the pair table when that position is another pixel of the quad
        auto dist = [&](const int y, const int x) -> float {
            const bool quad = y >= 1 && y <= 2 && x >= 1 && x <= 2;
            const int q = quad ? (y - 1) * 2 + (x - 1) : -1;
            if (SHARE && quad && q < k) return pc[quad_pair_id(k, q)];
            const float da = wa[y][x] - ra, db = wq[y][x] - rb;
            const float d = da * da + db * db;
            if (SHARE && quad) pc[quad_pair_id(k, q)] = d;
            return d;
        };
        // (sharing L(q) - L(p) between the two pixels of a pair the same way -- the negation is a source modifier -- saves another 14 instructions and measures
        // within noise: 0.6149 vs 0.6135 ms per step, six runs each; not kept)
        auto ldiff = [&](const int y, const int x) -> float { return wl[y][x] - rl; };
        const float e1 = fabsf(rl - wl[n1y][n1x]), e2 = fabsf(rl - wl[n2y][n2x]);
        const float c1 = dist(n1y, n1x), c2 = dist(n2y, n2x);
        float el, ec;
        asm("v_max_f32 %0, %1, %2" : "=v"(el) : "v"(e1), "v"(e2));
        asm("v_max_f32 %0, %1, %2" : "=v"(ec) : "v"(c1), "v"(c2));
        int c = 3;
#pragma unroll
        for (int wy = 0; wy < 3; wy++)
#pragma unroll
            for (int wx = 0; wx < 3; wx++) {
                const int y = dy + wy, x = dx + wx;
                if ((y == cy && x == cx) || (y == n1y && x == n1x) || (y == n2y && x == n2x)) continue;
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(ldiff(y, x) <= el) & __builtin_amdgcn_ballot_w64(dist(y, x) <= ec));
            }
        cnt[k] = c;
    }
}

// The same vote with nothing taken for granted: all nine cells of each window go through both tests, exactly as
// pyx:47-58 spells them.  Needed when a Lab value is not finite -- only L can be (HDR mode: L = luma, ahd.py:55,59; a and b
// come from clipped values) -- because then the comparisons the fast form skips are false (NaN) instead of true.
template <int DIR, int K0 = 0, int K1 = 4>
DEVI void vote_quad_literal(const float wl[4][4], const float wa[4][4], const float wq[4][4], int cnt[4]) {
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        float rl = wl[cy][cx], ra = wa[cy][cx], rb = wq[cy][cx];
        float e1 = fabsf(rl - wl[n1y][n1x]), e2 = fabsf(rl - wl[n2y][n2x]);
        float da1 = ra - wa[n1y][n1x], db1 = rb - wq[n1y][n1x], da2 = ra - wa[n2y][n2x], db2 = rb - wq[n2y][n2x];
        float c1 = da1 * da1 + db1 * db1, c2 = da2 * da2 + db2 * db2;
        float el = e2 > e1 ? e2 : e1, ec = c2 > c1 ? c2 : c1;
        int c = 0;
#pragma unroll
        for (int wy = 0; wy < 3; wy++)
#pragma unroll
            for (int wx = 0; wx < 3; wx++) {
                const int y = dy + wy, x = dx + wx;
                float da = wa[y][x] - ra, db = wq[y][x] - rb;
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(wl[y][x] - rl <= el) & __builtin_amdgcn_ballot_w64(da * da + db * db <= ec));
            }
        cnt[k] = c;
    }
}

// The literal vote once more, for the workgroups of the HDR instance that really hold a non-finite luma (rare): every cell is read from LDS where it is used, one pixel
// per trip of a rolled loop, so that this path adds next to nothing to the registers of the kernel's fast path (with the 4x4 windows of vote_quad_literal in registers
// the two forms behind one uniform branch came to 87 VGPRs, five workgroups per CU instead of six).  Same arithmetic, cell for cell, as vote_quad_literal;
// window cell (Y, X) of an image-edge quad takes the duplicated edge pixel (BORDER_REFLECT, ahd.py:64) by clamping its coordinates.  Returns the four counts, 4 bits each.
template <int DIR>
DEVI unsigned vote_quad_literal_lds(const float* lab, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right) {
    const float* const pL = lab + (2 * lqy) * LPS + 2 * lqx;
    auto cell = [&](int Y, int X) -> const float* {
        Y = (at_top && Y == 0) ? 1 : ((at_bot && Y == 3) ? 2 : Y);
        X = (at_left && X == 0) ? 1 : ((at_right && X == 3) ? 2 : X);
        return pL + Y * LPS + X;
    };
    unsigned packed = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        const float* pc = cell(cy, cx);
        const float* p1 = cell(n1y, n1x);
        const float* p2 = cell(n2y, n2x);
        const float rl = pc[0], ra = pc[LPR * LPS], rb = pc[2 * LPR * LPS];
        const float e1 = fabsf(rl - p1[0]), e2 = fabsf(rl - p2[0]);
        const float da1 = ra - p1[LPR * LPS], db1 = rb - p1[2 * LPR * LPS], da2 = ra - p2[LPR * LPS], db2 = rb - p2[2 * LPR * LPS];
        const float c1 = da1 * da1 + db1 * db1, c2 = da2 * da2 + db2 * db2;
        const float el = e2 > e1 ? e2 : e1, ec = c2 > c1 ? c2 : c1;
        int c = 0;
#pragma unroll 1
        for (int wy = 0; wy < 3; wy++)
#pragma unroll 1
            for (int wx = 0; wx < 3; wx++) {
                const float* pw = cell(dy + wy, dx + wx);
                const float da = pw[LPR * LPS] - ra, db = pw[2 * LPR * LPS] - rb;
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(pw[0] - rl <= el) & __builtin_amdgcn_ballot_w64(da * da + db * db <= ec));
            }
        packed |= (unsigned)c << (4 * k);
    }
    return packed;
}

// ---- Lab mode 1, packed layout (round 4) ------------------------------------------------------------------------------------------------
// The Lab buffer holds one 8-byte cell { L (float), a' | b' << 16 } per pixel: [LPR][LPS] cells, 9.2 KB instead of the 13.9 KB of three float planes
// (20.8 KB in all: SEVEN workgroups per CU).  a' and b' are the interpolated table values themselves (15-bit unsigned); the float values cv2 would
// return are a = a'/64 - 128, b = b'/64 - 128, exact in float32, and the vote (pyx:50-57) uses only their differences:
//     da = (a'_w - a'_c) / 64 exactly,   fl(da * da) = fl(D^2) / 4096 with D the integer difference (a power-of-two scale commutes with rounding),
// so the float32 chroma distance is fl(fl(Da^2) + fl(Db^2)) / 4096 and its comparisons are those of fl(fl(Da^2) + fl(Db^2)).
// While S = Da^2 + Db^2 < 2^24 nothing rounds: the float distance IS the integer S.  The vote compares S_cell <= ec with ec = max(S_n1, S_n2):
//   * ec < 2^24 and S_cell <= ec: all three exact, same answer;
//   * ec < 2^24 and S_cell > ec: if S_cell < 2^24 it is exact, same answer; if S_cell >= 2^24 then either one square is >= 2^24 already, or both are
//     exact and their exact sum is >= 2^24 -- float rounding is monotone and 2^24 is a float, so the float distance is >= 2^24 > ec: same answer.
// So the integer vote is bit-identical to the float one whenever every ec of the wave is below 2^24, i.e. no pixel's chroma differs from one of its two
// direction neighbours' by 64 Lab units or more -- three v_or and one ballot per direction decide it, and a wave that fails the test (hard colour
// noise; never on the benchmark scene) computes its distances as float32 sums of float32 squares of the integer differences (chroma_d2_f32).
typedef short s16x2 __attribute__((ext_vector_type(2)));
DEVI unsigned chroma_d2(unsigned x, unsigned y) {                 // (a'x - a'y)^2 + (b'x - b'y)^2: v_pk_sub_i16 + v_dot2_i32_i16
    const s16x2 d = __builtin_bit_cast(s16x2, x) - __builtin_bit_cast(s16x2, y);      // |difference| <= 32767: no wrap
    return (unsigned)__builtin_amdgcn_sdot2(d, d, 0, false);                          // <= 2 * 32767^2 < 2^31
}
constexpr int LC4 = LPS / 2;                                      // float4 (two cells) per Lab row
static_assert(LPS % 2 == 0, "Lab rows are whole float4s");
// Rows [R0, R1) of the 4x4 window of cells around a quad (rows / cols -1..2 of the quad's top-left pixel), 16-byte reads; BORDER_REFLECT (ahd.py:64) duplicates
// the edge pixel.  As in round 3 the window comes in two pieces -- rows 0-2 for the quad's upper pixel pair, then row 3 in row 0's registers for the lower
// pair -- 24 window registers instead of 32 (the kernel has to stay at 72 VGPRs for seven waves per SIMD).
template <int R0, int R1>
DEVI void load_labrows_pk(const float* lab, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right, float wl[4][4], unsigned wc[4][4]) {
    const float4* p = reinterpret_cast<const float4*>(lab) + (2 * lqy) * LC4 + lqx;
#pragma unroll
    for (int r = R0; r < R1; r++) {
        const float4 a = p[r * LC4], b = p[r * LC4 + 1];
        wl[r][0] = a.x; wc[r][0] = __float_as_uint(a.y); wl[r][1] = a.z; wc[r][1] = __float_as_uint(a.w);
        wl[r][2] = b.x; wc[r][2] = __float_as_uint(b.y); wl[r][3] = b.z; wc[r][3] = __float_as_uint(b.w);
    }
    if (at_top | at_bot | at_left | at_right) {   // interior waves skip the selects
#pragma unroll
        for (int r = R0; r < R1; r++) {
            if (at_left) { wl[r][0] = wl[r][1]; wc[r][0] = wc[r][1]; }
            if (at_right) { wl[r][3] = wl[r][2]; wc[r][3] = wc[r][2]; }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (R0 == 0 && at_top) { wl[0][c] = wl[1][c]; wc[0][c] = wc[1][c]; }
            if (R1 == 4 && at_bot) { wl[3][c] = wl[2][c]; wc[3][c] = wc[2][c]; }      // row 2 already carries its left / right rule
        }
    }
}
// rows [R0, R1) of the same window as three FLOAT windows for the float vote of round 3 (vote_quad): chroma converted without the 1/64 scale and the -128
// offset, which cancel in every difference and commute with every rounding (see above)
template <int R0, int R1>
DEVI void load_lab_rows_pk_cvt(const float* lab, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right, float wl[4][4], float wa[4][4], float wq[4][4]) {
    const float4* p = reinterpret_cast<const float4*>(lab) + (2 * lqy) * LC4 + lqx;
#pragma unroll
    for (int r = R0; r < R1; r++) {
        const float4 a = p[r * LC4], b = p[r * LC4 + 1];
        const unsigned c0 = __float_as_uint(a.y), c1 = __float_as_uint(a.w), c2 = __float_as_uint(b.y), c3 = __float_as_uint(b.w);
        wl[r][0] = a.x; wl[r][1] = a.z; wl[r][2] = b.x; wl[r][3] = b.z;
        wa[r][0] = (float)(c0 & 0xFFFFu); wa[r][1] = (float)(c1 & 0xFFFFu); wa[r][2] = (float)(c2 & 0xFFFFu); wa[r][3] = (float)(c3 & 0xFFFFu);
        wq[r][0] = (float)(c0 >> 16); wq[r][1] = (float)(c1 >> 16); wq[r][2] = (float)(c2 >> 16); wq[r][3] = (float)(c3 >> 16);
        __builtin_amdgcn_sched_barrier(0);          // row by row: the raw cells of all rows must not be in registers together (this path may not cost the kernel its seventh wave)
    }
    if (at_top | at_bot | at_left | at_right) {
#pragma unroll
        for (int r = R0; r < R1; r++) {
            if (at_left) { wl[r][0] = wl[r][1]; wa[r][0] = wa[r][1]; wq[r][0] = wq[r][1]; }
            if (at_right) { wl[r][3] = wl[r][2]; wa[r][3] = wa[r][2]; wq[r][3] = wq[r][2]; }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (R0 == 0 && at_top) { wl[0][c] = wl[1][c]; wa[0][c] = wa[1][c]; wq[0][c] = wq[1][c]; }
            if (R1 == 4 && at_bot) { wl[3][c] = wl[2][c]; wa[3][c] = wa[2][c]; wq[3][c] = wq[2][c]; }
        }
    }
}
// The same distance the way the float32 reference rounds it, for the waves the integer form cannot serve: the integer differences as floats (exact),
// their squares and the sum rounded like cv2's a, b would give them (scaled by 4096, a power of two).  Eight instructions instead of two.
DEVI float chroma_d2_f32(unsigned x, unsigned y) {
    asm volatile("" : "+v"(x));          // (not the integer form's subtraction: see vote_quad_pk_f32 on hoisting)
    const s16x2 d = __builtin_bit_cast(s16x2, x) - __builtin_bit_cast(s16x2, y);
    const float da = (float)(int)d.x, db = (float)(int)d.y;
    return da * da + db * db;
}
// Which of the quad's six pixel pairs (quad_pair_id) already sit in pc[] when the integer vote of pixels [K0, K1) reaches its cells: a compile-time fact of the
// processing order (epsilon step of the piece first, then its pixels in order).  Piece 1 (K0 = 0) starts empty and its epsilon step adds the direction's upper
// pair(s); piece 2 (K0 = 2) inherits everything piece 1 produced (all pairs but (2,3), id 1) and its epsilon step adds (2,3) when the direction is horizontal.
template <int DIR, int K0>
DEVI void pairs_ready(bool have[6]) {
    const bool second = K0 == 2;
    have[0] = second || DIR == 0; have[1] = second && DIR == 0; have[2] = second || DIR == 1; have[3] = second || DIR == 1; have[4] = second; have[5] = second;
}
// pyx:22-58 for pixels [K0, K1) of a quad on the packed window, integer chroma.  Step 1 (eps): the chroma distances to the two direction neighbours of every
// pixel and their maximum ec[k]; distances between two pixels of the quad go through the pair table pc[] (vote_quad).
template <int DIR, int K0, int K1>
DEVI void vote_eps_pk(const unsigned wc[4][4], unsigned ec[4], unsigned pc[6]) {
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int ny[2] = {DIR ? cy - 1 : cy, DIR ? cy + 1 : cy}, nx[2] = {DIR ? cx : cx - 1, DIR ? cx : cx + 1};
        unsigned c[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int y = ny[j], x = nx[j];
            const bool quad = y >= 1 && y <= 2 && x >= 1 && x <= 2;
            const int q = quad ? (y - 1) * 2 + (x - 1) : -1;
            if (quad && q < k) { c[j] = pc[quad_pair_id(k, q)]; continue; }
            c[j] = chroma_d2(wc[y][x], wc[cy][cx]);
            if (quad) pc[quad_pair_id(k, q)] = c[j];
        }
        ec[k] = c[0] > c[1] ? c[0] : c[1];
    }
}
// Step 2: the counts.  The centre and the two epsilon neighbours always count (as in vote_quad); the other six cells are tested: float L difference against
// the float epsilon, integer chroma distance against ec[k]; the quad's other pixel pairs come from / go into pc[].
template <int DIR, int K0, int K1>
DEVI void vote_cells_pk(const float wl[4][4], const unsigned wc[4][4], const unsigned ec[4], unsigned pc[6], int cnt[4]) {
    bool have[6];
    pairs_ready<DIR, K0>(have);
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        const float rl = wl[cy][cx];
        const float e1 = fabsf(rl - wl[n1y][n1x]), e2 = fabsf(rl - wl[n2y][n2x]);
        float el;
        asm("v_max_f32 %0, %1, %2" : "=v"(el) : "v"(e1), "v"(e2));
        int c = 3;
#pragma unroll
        for (int wy = 0; wy < 3; wy++)
#pragma unroll
            for (int wx = 0; wx < 3; wx++) {
                const int y = dy + wy, x = dx + wx;
                if ((y == cy && x == cx) || (y == n1y && x == n1x) || (y == n2y && x == n2x)) continue;
                const bool quad = y >= 1 && y <= 2 && x >= 1 && x <= 2;
                const int q = quad ? (y - 1) * 2 + (x - 1) : -1;
                unsigned d;
                if (quad && have[quad_pair_id(k, q)]) d = pc[quad_pair_id(k, q)];
                else {
                    d = chroma_d2(wc[y][x], wc[cy][cx]);
                    if (quad) { pc[quad_pair_id(k, q)] = d; have[quad_pair_id(k, q)] = true; }
                }
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(wl[y][x] - rl <= el) & __builtin_amdgcn_ballot_w64(d <= ec[k]));
            }
        cnt[k] = c;
    }
}
// The rare exact-float form (a wave whose guard failed): pixel by pixel, every distance recomputed as chroma_d2_f32, nothing shared, one cell at a time --
// built for a small register footprint (it must not raise the kernel's VGPR count above the integer form's), not for speed.  Returns the counts of
// pixels [K0, K1), 4 bits each.
template <int DIR, int K0, int K1>
DEVI unsigned vote_quad_pk_f32(const float wl[4][4], const unsigned wc[4][4]) {
    unsigned packed = 0;
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        float rl = wl[cy][cx];
        // (this form's arithmetic must not look like the integer form's: the compiler would hoist the common L differences and packed subtractions above the
        // wave-uniform branch between the two forms and keep them in registers across it -- 102 VGPRs instead of 72; hence the opaque copies)
        asm volatile("" : "+v"(rl));
        const float e1 = fabsf(rl - wl[n1y][n1x]), e2 = fabsf(rl - wl[n2y][n2x]);
        const float c1 = chroma_d2_f32(wc[n1y][n1x], wc[cy][cx]), c2 = chroma_d2_f32(wc[n2y][n2x], wc[cy][cx]);
        float el, ec;
        asm("v_max_f32 %0, %1, %2" : "=v"(el) : "v"(e1), "v"(e2));
        asm("v_max_f32 %0, %1, %2" : "=v"(ec) : "v"(c1), "v"(c2));
        int c = 3;
#pragma unroll
        for (int wy = 0; wy < 3; wy++)
#pragma unroll
            for (int wx = 0; wx < 3; wx++) {
                const int y = dy + wy, x = dx + wx;
                if ((y == cy && x == cx) || (y == n1y && x == n1x) || (y == n2y && x == n2x)) continue;
                __builtin_amdgcn_sched_barrier(0);
                const float d = chroma_d2_f32(wc[y][x], wc[cy][cx]);
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(wl[y][x] - rl <= el) & __builtin_amdgcn_ballot_w64(d <= ec));
            }
        packed |= (unsigned)c << (4 * k);
        __builtin_amdgcn_sched_barrier(0);
    }
    return packed;
}
// The literal nine-cell vote from the packed buffer (HDR instance, workgroups that hold a non-finite luma): vote_quad_literal_lds with the chroma read as
// integers and converted (exact, scale-free: see above)
template <int DIR>
DEVI unsigned vote_quad_literal_lds_pk(const float* lab, int lqy, int lqx, bool at_top, bool at_bot, bool at_left, bool at_right) {
    const float2* const pL = reinterpret_cast<const float2*>(lab) + (2 * lqy) * LPS + 2 * lqx;
    auto cell = [&](int Y, int X) -> const float2* {
        Y = (at_top && Y == 0) ? 1 : ((at_bot && Y == 3) ? 2 : Y);
        X = (at_left && X == 0) ? 1 : ((at_right && X == 3) ? 2 : X);
        return pL + Y * LPS + X;
    };
    auto fa = [](float2 v) { return (float)(__float_as_uint(v.y) & 0xFFFFu); };
    auto fb = [](float2 v) { return (float)(__float_as_uint(v.y) >> 16); };
    unsigned packed = 0;
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const int dy = k >> 1, dx = k & 1, cy = dy + 1, cx = dx + 1;
        const int n1y = DIR ? cy - 1 : cy, n1x = DIR ? cx : cx - 1, n2y = DIR ? cy + 1 : cy, n2x = DIR ? cx : cx + 1;
        const float2 vc = *cell(cy, cx), v1 = *cell(n1y, n1x), v2 = *cell(n2y, n2x);
        const float rl = vc.x, ra = fa(vc), rb = fb(vc);
        const float e1 = fabsf(rl - v1.x), e2 = fabsf(rl - v2.x);
        const float da1 = ra - fa(v1), db1 = rb - fb(v1), da2 = ra - fa(v2), db2 = rb - fb(v2);
        const float c1 = da1 * da1 + db1 * db1, c2 = da2 * da2 + db2 * db2;
        const float el = e2 > e1 ? e2 : e1, ec = c2 > c1 ? c2 : c1;
        int c = 0;
#pragma unroll 1
        for (int wy = 0; wy < 3; wy++)
#pragma unroll 1
            for (int wx = 0; wx < 3; wx++) {
                const float2 vw = *cell(dy + wy, dx + wx);
                const float da = fa(vw) - ra, db = fb(vw) - rb;
                c = add_lane_bit(c, __builtin_amdgcn_ballot_w64(vw.x - rl <= el) & __builtin_amdgcn_ballot_w64(da * da + db * db <= ec));
            }
        packed |= (unsigned)c << (4 * k);
    }
    return packed;
}

}  // namespace

struct AhdParams {
    MosaicSrc src;
    float* out;          // (H,W,3)
    const float4* labtab; // LAB_SLOTS entries (lab_tables.h), copied to LDS by every workgroup
    const uint4* lablut;  // Lab mode 1: the OpenCV-4.10 grid in the device layout of devmath.h (global memory, L2 resident)
    int H, W;
    float wb[3];
    int hdr;
    int tail;            // colour tail applied to the selected pixel (only when no median stage follows)
    Ccm ccm;
    unsigned* float_form_tiles;   // packed Lab layout only, may be NULL: two cumulative counters -- [0] += 1 per workgroup in which some wave had to redo its votes in the float form, [1] += tiles of the launch
};

// TINY: quarter planes narrower than 4 need the general (looping) border functions.
// HDR (image.get_hdr(), ahd.py:52-59) is a template parameter because its literal-vote path for non-finite luma costs registers.
// TAIL: a colour tail may follow the selection (only when no median stage does); the instance without it is the benchmark's.
//
// LDS (20.8 KB with the packed Lab cells of round 4: SEVEN workgroups per CU; 25.5 KB with three float planes: six):
//   mw   [4][MWY][MWX]  white-balanced mosaic planes, alive until the vertical green planes are built; the packed vote map lies over them afterwards
//   gq   [4][GY][GX]    green at R / B sites and the colour differences of ONE direction at a time (horizontal first, vertical built while the
//                       horizontal votes run)
//   lab  [LPR][LPS] cells of { L, a' | b' << 16 } (LAB = 1) or [3][LPR][LPS] floats (LAB = 0, 2): Lab of one direction, written by the thread that computes
//                       it, straight from its registers (no overlay, no staging barrier)
// Round 2 laid the Lab buffer over the mosaic and horizontal planes (20.6 KB, five workgroups per CU limited by 92 VGPRs): every thread then had to
// hold the twelve Lab values of its quad across a barrier, and the eight green samples of its window across both directions.  Measured on MI355X
// (tools/ab_bench.sh, LDS padding): five -> four workgroups per CU costs this kernel 9 %, four -> three 24 %: it is latency-bound, occupancy is the lever.
// LDS floats of the select tile: the three sections below (+ 2: the last window row of the last Lab plane is read, never used, one row past LPR)
template <int LAB> struct SelLds {
    static constexpr bool I16 = LAB == 1 && AHD_I16 != 0;      // packed Lab cells { L, a' | b' << 16 } and integer chroma votes (round 4)
    static constexpr int NMW = 4 * MWY * MWX, NGQ = 4 * GY * GXS, NLAB = (I16 ? 2 : 3) * LPR * LPS, N = NMW + NGQ + NLAB + 2;
};
// One 28x28 px tile of the select kernel.  The body is a device function so that the stand-alone kernel (k_ahd_select: one tile per workgroup, XCD-aware
// order) and the role-interleaved kernel (k_ahd_fused: select tiles of one frame and median tiles of the previous one in ONE grid) share it.
// `planes`: SelLds<LAB>::N floats of LDS, 16-byte aligned; `s_labtab`: LAB_SLOTS float4 of LDS (Lab mode 0 only); `s_nonfinite`: two ints of LDS (HDR only).
template <bool TINY, bool U16, bool HDR, int LAB, bool TAIL>
DEVI void ahd_select_tile(const AhdParams& p, const int tbx, const int tby, float* const planes, float4* const s_labtab, int* const s_nonfinite) {
    constexpr bool I16 = SelLds<LAB>::I16;
    constexpr int NMW = SelLds<LAB>::NMW, NGQ = SelLds<LAB>::NGQ;
    static_assert(NMW % 4 == 0 && NGQ % 4 == 0, "16-byte aligned sections");
    float* const mw = planes;
    float* const gq = planes + NMW;
    float* const lab = planes + NMW + NGQ;
    unsigned short* const vmap = reinterpret_cast<unsigned short*>(planes);      // [MPR][MPS] votes h | v << 8, over the dead mosaic planes
    static_assert(MPR * MPS * sizeof(unsigned short) <= NMW * sizeof(float), "the vote map fits over the mosaic planes");
    // HDR metric only: did this workgroup write a non-finite L (= luma, ahd.py:55,59) into the Lab buffer of direction H / V?  Only then do the votes need their
    // literal nine-cell form; every other workgroup takes the fast form, whose shortcuts hold for finite values (round 3: 2 642 -> about 2 400 executed instructions)
    const LabTab lt{s_labtab, s_labtab + LAB_DEC_SLOTS};

    const int tid = threadIdx.x;
    AHD_STAMP(0);
    if (tid < 3) s_nonfinite[tid] = 0;                                          // (the first barrier below orders it before any P2); [2]: some wave took the float form of the vote
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    const int tq0x = tbx * TQX, tq0y = tby * TQY;
    const double* M = p.ccm.m;
    // the whole halo-3 neighbourhood of the tile lies inside the image: no border rule applies anywhere in P0/P1
    const bool inside = tq0y >= 3 && tq0x >= 3 && tq0y + TQY + 3 <= h && tq0x + TQX + 3 <= w;

    // ---- P0: white-balanced mosaic planes, symmetric (edge-duplicating) reflect per plane (ahd.py:77-80).
    // One 8-byte load per quad row; a thread's loads are all issued before its first LDS store.
    // Round 4 mapping: the 2 MWY x MWX pairs of the tile as rows of 32 lanes -- lane (r, c) = (tid / 32, tid % 32) takes pair column c (< MWX) of pair rows
    // r, r + 8, r + 16, ...: no division by MWX, the row's CFA parity (and with it the planes and WB factors) is one per-lane constant, the global address of
    // row r + 8 k is a uniform base plus ONE per-lane offset, and the LDS addresses differ by immediates.  (Until round 3 element idx = tid + 256 k was split by
    // idx / MWX per element, and the compiler evaluated the border path's reflections and 64-bit addresses next to the interior path's for every element:
    // 215 vector instructions before the first barrier, 30 of them of the multiplier family at 6-8 cycles -- profiles/r4_isa_mix_k_ahd_select_f32lab.csv.)
    {
        static_assert(MWX <= 32 && NT_A % 32 == 0, "a pair row fits in 32 lanes");
        constexpr int RPP = NT_A / 32, NROWS = 2 * MWY, NL = (NROWS + RPP - 1) / RPP;      // pair rows per pass, passes
        constexpr int NTAB = LAB == 0 ? (LAB_SLOTS + NT_A - 1) / NT_A : 0;
        const int r = tid >> 5, c = tid & 31;
        const bool on = c < MWX;
        const int cc = on ? c : MWX - 1;                                     // idle lanes repeat the last column's (valid) address
        const int dy = r & 1;                                                // RPP is even: every pass of a lane has the same row parity
        static_assert(RPP % 2 == 0, "row parity is a per-lane constant");
        float2 tmp[NL];
        float4 ttab[NTAB + 1];
#pragma unroll
        for (int k = 0; k < NTAB; k++) ttab[k] = p.labtab[min(tid + k * NT_A, LAB_SLOTS - 1)];
        if (!U16 && inside) {     // uniform per workgroup: 64-bit tile origin (scalar) + tile-local 32-bit byte offset per lane
            const char* const tile = reinterpret_cast<const char*>(p.src.f32 + (size_t)(2 * (tq0y - 3)) * W + 2 * (tq0x - 3));
            const unsigned rowb = (unsigned)W * 4u;
            const unsigned voff = mul24((unsigned)r, rowb) + 8u * (unsigned)cc;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int ry = (NROWS % RPP == 0 || r + k * RPP < NROWS) ? k * RPP : 0;   // rows past the tile (other tile shapes only) re-read the lane's first row
                tmp[k] = *reinterpret_cast<const float2*>(tile + (voff + (unsigned)ry * rowb));             // (scalar base, 32-bit vector offset): one v_add_u32 per pass
            }
        } else {
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int ry = min(r + k * RPP, NROWS - 1), my = ry >> 1;
                int qi = tq0y - 3 + my, qj = tq0x - 3 + cc;
                if (!inside) {
                    qi = TINY ? b_sym(qi, h) : b_sym1(qi, h);
                    qj = TINY ? b_sym(qj, w) : b_sym1(qj, w);
                }
                tmp[k] = load_mosaic_pair<U16>(p.src, (size_t)(2 * qi + dy) * W + 2 * qj, dy != 0);
            }
        }
        // even row: (R, G1) ; odd row: (G2, B)
        const float w0 = dy ? p.wb[1] : p.wb[0], w1 = dy ? p.wb[2] : p.wb[1];
        float* const d0 = mw + ((dy ? P_G2 : P_R) * MWY + (r >> 1)) * MWX + cc;
        float* const d1 = mw + ((dy ? P_B : P_G1) * MWY + (r >> 1)) * MWX + cc;
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (on && (NROWS % RPP == 0 || r + k * RPP < NROWS)) {
                d0[k * (RPP / 2) * MWX] = tmp[k].x * w0;
                d1[k * (RPP / 2) * MWX] = tmp[k].y * w1;
            }
#pragma unroll
        for (int k = 0; k < NTAB; k++) if (tid + k * NT_A < LAB_SLOTS) s_labtab[tid + k * NT_A] = ttab[k];
    }
    AHD_STAMP(1);      // P0 done (tile loaded, planes stored)
    __syncthreads();
    AHD_STAMP(2);

#define MWAT(pl, yy, xx) mw[((pl) * MWY + (yy)) * MWX + (xx)]
    // ---- P1(dir): directional green at R/B sites (ahd.py:97-102) and D = sub - g (eag.py:142) of ONE direction into gq
#ifdef AHD_P1_OWN_ELEMENT      // experiment (round 4): measured -0.2 % on the select kernel, inside the noise; off
    auto green_planes = [&](const int dir) {
        auto body = [&](const int gy, const int gx, const int a, const int c) {
            const float rc = MWAT(P_R, a, c), bc = MWAT(P_B, a, c);
            float gr, gb;
            if (dir == 0) {
                gr = (((MWAT(P_R, a, c - 1) * AH0 + MWAT(P_G1, a, c - 1) * AH1) + rc * AH2) + MWAT(P_G1, a, c) * AH1) + MWAT(P_R, a, c + 1) * AH0;
                gb = (((MWAT(P_B, a, c - 1) * AH0 + MWAT(P_G2, a, c) * AH1) + bc * AH2) + MWAT(P_G2, a, c + 1) * AH1) + MWAT(P_B, a, c + 1) * AH0;
            } else {
                gr = (((MWAT(P_R, a - 1, c) * AH0 + MWAT(P_G2, a - 1, c) * AH1) + rc * AH2) + MWAT(P_G2, a, c) * AH1) + MWAT(P_R, a + 1, c) * AH0;
                gb = (((MWAT(P_B, a - 1, c) * AH0 + MWAT(P_G1, a, c) * AH1) + bc * AH2) + MWAT(P_G1, a + 1, c) * AH1) + MWAT(P_B, a + 1, c) * AH0;
            }
            const int gi = gy * GXS + gx;
            gq[0 * GY * GXS + gi] = gr; gq[1 * GY * GXS + gi] = gb;
            gq[2 * GY * GXS + gi] = rc - gr; gq[3 * GY * GXS + gi] = bc - gb;
        };
        // two copies of the body on purpose: the interior tiles' addresses are tid-derived constants, the border tiles' come out of the reflections
        auto element = [&](const int gy, const int gx) {
            if (inside) { body(gy, gx, gy + 1, gx + 1); return; }
            const int ri = TINY ? b_101(tq0y - 2 + gy, h) : b_1011(tq0y - 2 + gy, h);   // REFLECT_101 on the quarter plane
            const int rj = TINY ? b_101(tq0x - 2 + gx, w) : b_1011(tq0x - 2 + gx, w);
            const int a = ri - (tq0y - 3), c = rj - (tq0x - 3);
            if (a < 1 || a > MWY - 2 || c < 1 || c > MWX - 2) return;                   // never consumed by a valid output
            body(gy, gx, a, c);
        };
        if constexpr (QPT == 1 && LQX == 16 && LQY == 16 && !AHD_LANES8) {
            // Round 4: every thread its own quad's element of the 18 x 18 plane (row tid / 16 + 1, column tid % 16 + 1: shifts, and offsets that differ from the thread's
            // other LDS addresses by constants), then the 68 ring elements on the first 68 lanes -- the two waves that ran the second trip of the loop below anyway.
            // (Before: idx / 18 by a 24-bit multiply, its remainder by a second one and the row offset by a v_mul_lo_u32, per trip and direction.)
            static_assert(GY == LQY + 2 && GX == LQX + 2, "interior + ring");
            element((tid >> 4) + 1, (tid & 15) + 1);
            if (tid < 2 * GX + 2 * LQY) {
                int gy, gx;
                if (tid < 2 * GX) { const bool bot = tid >= GX; gy = bot ? GY - 1 : 0; gx = tid - (bot ? GX : 0); }       // top row, bottom row
                else { const int q = tid - 2 * GX; gy = (q & (LQY - 1)) + 1; gx = q >= LQY ? GX - 1 : 0; }                // left column, right column
                element(gy, gx);
            }
        } else
        for (int idx = tid; idx < GY * GX; idx += NT_A) {
            const int gy = idx / GX;
            element(gy, idx - gy * GX);
        }
    };
#else
    auto green_planes = [&](const int dir) {
        for (int idx = tid; idx < GY * GX; idx += NT_A) {
            int gy = idx / GX, gx = idx - gy * GX;
            int a = gy + 1, c = gx + 1;
            if (!inside) {
                int ri = TINY ? b_101(tq0y - 2 + gy, h) : b_1011(tq0y - 2 + gy, h);   // REFLECT_101 on the quarter plane
                int rj = TINY ? b_101(tq0x - 2 + gx, w) : b_1011(tq0x - 2 + gx, w);
                a = ri - (tq0y - 3); c = rj - (tq0x - 3);
                if (a < 1 || a > MWY - 2 || c < 1 || c > MWX - 2) continue;        // never consumed by a valid output
            }
            const float rc = MWAT(P_R, a, c), bc = MWAT(P_B, a, c);
            float gr, gb;
            if (dir == 0) {
                gr = (((MWAT(P_R, a, c - 1) * AH0 + MWAT(P_G1, a, c - 1) * AH1) + rc * AH2) + MWAT(P_G1, a, c) * AH1) + MWAT(P_R, a, c + 1) * AH0;
                gb = (((MWAT(P_B, a, c - 1) * AH0 + MWAT(P_G2, a, c) * AH1) + bc * AH2) + MWAT(P_G2, a, c + 1) * AH1) + MWAT(P_B, a, c + 1) * AH0;
            } else {
                gr = (((MWAT(P_R, a - 1, c) * AH0 + MWAT(P_G2, a - 1, c) * AH1) + rc * AH2) + MWAT(P_G2, a, c) * AH1) + MWAT(P_R, a + 1, c) * AH0;
                gb = (((MWAT(P_B, a - 1, c) * AH0 + MWAT(P_G1, a, c) * AH1) + bc * AH2) + MWAT(P_G1, a + 1, c) * AH1) + MWAT(P_B, a + 1, c) * AH0;
            }
            const int gi = gy * GXS + gx;
            gq[0 * GY * GXS + gi] = gr; gq[1 * GY * GXS + gi] = gb;
            gq[2 * GY * GXS + gi] = rc - gr; gq[3 * GY * GXS + gi] = bc - gb;
        }
    };
#endif
    green_planes(0);
    AHD_STAMP(3);      // P1(H) done
    __syncthreads();
    AHD_STAMP(4);

    // per quad of this thread: position in the region, image-edge flags, LDS coordinates
    struct Quad { int lqy, lqx, gy, gx, my, mx, vmy, vmx; bool active, at_top, at_bot, at_left, at_right, inner; };
    auto quad = [&](const int q) {
        Quad c;
        const int idx = tid + q * NT_A;
        if (AHD_LANES8 && QPT == 1 && LQX == 16 && LQY == 16) {      // wave w covers the 8 x 8 quads of quadrant (w / 2, w % 2)
            c.lqy = ((idx >> 7) << 3) + ((idx >> 3) & 7); c.lqx = (((idx >> 6) & 1) << 3) + (idx & 7);
        } else {
            c.lqy = idx / LQX; c.lqx = idx - c.lqy * LQX;
        }
        const int qi = tq0y - 1 + c.lqy, qj = tq0x - 1 + c.lqx;
        c.active = qi >= 0 && qi < h && qj >= 0 && qj < w;
        c.at_top = qi == 0; c.at_bot = qi == h - 1; c.at_left = qj == 0; c.at_right = qj == w - 1;
        c.gy = c.lqy + 1; c.gx = c.lqx + 1; c.my = c.lqy + 2; c.mx = c.lqx + 2;
        c.vmy = 2 * c.lqy - 1; c.vmx = 2 * c.lqx - 1;      // vote-map cell of the quad's top-left pixel (map origin = tile origin - 1 px)
        c.inner = c.active && c.lqy >= 1 && c.lqy <= TQY && c.lqx >= 1 && c.lqx <= TQX;
        return c;
    };

    Quad qc[QPT];
#pragma unroll
    for (int q = 0; q < QPT; q++) qc[q] = quad(q);

    float rgbh[QPT][4][3], rgbv[QPT][4][3];   // both candidates stay in registers until the selection
    unsigned hvotes[QPT];                     // the four horizontal counts of a quad (4 bits each) until the vertical ones exist
#pragma unroll
    for (int q = 0; q < QPT; q++) hvotes[q] = 0;

    // fully unrolled: plane offsets and the vote's direction become constants
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        // ---- P2: high-pass of green, photosite-aware resampling of R and B, second white balance + CCM + Lab -> LDS
        float rr[QPT][4], gg[QPT][4], bb[QPT][4];
        bool nonfinite_l = false;              // HDR: this thread wrote a NaN / Inf luma into the Lab buffer of this direction
        auto candidate = [&](const int q) {
            const Quad& c = qc[q];
            const int gy = c.gy, gx = c.gx, my = c.my, mx = c.mx;
            const float* gR = gq, *gB = gq + GY * GXS, *dR = gq + 2 * GY * GXS, *dB = gq + 3 * GY * GXS;
            Win3 wgr = load_win<GXS>(gR, gy, gx), wgb = load_win<GXS>(gB, gy, gx);
            // the quad's own two green samples are the same in both candidates: the vertical pass takes them from the horizontal candidate's registers
            const float g1_c = dir == 0 ? MWAT(P_G1, my, mx) : rgbh[q][1][1], g2_c = dir == 0 ? MWAT(P_G2, my, mx) : rgbh[q][2][1];
            // full-resolution green, rows 2qi-1..2qi+2, cols 2qj-1..2qj+2
            float Wn[4][4] = {{wgb.v[0][0], MWAT(P_G2, my - 1, mx), wgb.v[0][1], MWAT(P_G2, my - 1, mx + 1)},
                              {MWAT(P_G1, my, mx - 1), wgr.v[1][1], g1_c, wgr.v[1][2]},
                              {wgb.v[1][0], g2_c, wgb.v[1][1], MWAT(P_G2, my, mx + 1)},
                              {MWAT(P_G1, my + 1, mx - 1), wgr.v[2][1], MWAT(P_G1, my + 1, mx), wgr.v[2][2]}};
            // GaussianBlur border = REFLECT_101 at full resolution: row -1 -> row 1, row H -> row H-2
            if (c.at_top | c.at_bot | c.at_left | c.at_right) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (c.at_top) Wn[0][k] = Wn[2][k];
                    if (c.at_bot) Wn[3][k] = Wn[1][k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (c.at_left) Wn[k][0] = Wn[k][2];
                    if (c.at_right) Wn[k][3] = Wn[k][1];
                }
            }
            float hf[4];
            highpass_quad(Wn, hf);
            float fg[4], fd[4];
            filt_base_tl(wgr, fg);
            { Win3 wd = load_win<GXS>(dR, gy, gx); filt_base_tl(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) rr[q][k] = fd[k] + (fg[k] + hf[k]);      // eag.py:141,143
            filt_base_br(wgb, fg);
            { Win3 wd = load_win<GXS>(dB, gy, gx); filt_base_br(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) bb[q][k] = fd[k] + (fg[k] + hf[k]);
            gg[q][0] = wgr.v[1][1]; gg[q][1] = g1_c; gg[q][2] = g2_c; gg[q][3] = wgb.v[1][1];
        };
        // Lab pixel (py,px) of the region lives at [py+1][px+1] (guard ring); the thread writes its own quad as soon as a pixel is done
        auto lab_px = [&](const int q, const int k) {
            const Quad& c = qc[q];
            float (&rgbc)[4][3] = dir == 0 ? rgbh[q] : rgbv[q];
            if constexpr (I16) {
                float L; unsigned ab;
                homog_lab_pk(p.lablut, rr[q][k], gg[q][k], bb[q][k], p.wb, M, HDR, L, ab);
                if (HDR) nonfinite_l |= !(fabsf(L) < __builtin_inff());
                float2* const o = reinterpret_cast<float2*>(lab) + (2 * c.lqy + 1 + (k >> 1)) * LPS + 2 * c.lqx + 1 + (k & 1);
                *o = make_float2(L, __uint_as_float(ab));          // one 8-byte cell per pixel
            } else {
                float* const pl = lab + (2 * c.lqy + 1) * LPS + 2 * c.lqx + 1;
                float L, A, Bq;
                homog_lab<LAB>(lt, p.lablut, rr[q][k], gg[q][k], bb[q][k], p.wb, M, HDR, L, A, Bq);
                if (HDR) nonfinite_l |= !(fabsf(L) < __builtin_inff());
                float* const o = pl + (k >> 1) * LPS + (k & 1);
                o[0] = L; o[LPR * LPS] = A; o[2 * LPR * LPS] = Bq;
            }
            rgbc[k][0] = rr[q][k]; rgbc[k][1] = gg[q][k]; rgbc[k][2] = bb[q][k];
        };
        if constexpr (QPT == 1) {
            if (qc[0].active) {
                candidate(0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lab_px(0, k);
#ifndef AHD_NO_SB
                    __builtin_amdgcn_sched_barrier(0);   // keep the four Lab evaluations from interleaving (register pressure)
#endif
                }
            }
        } else {
            // several quads per thread: the k-th pixels of all of them are evaluated together -- independent gather + interpolation chains between scheduling barriers
#pragma unroll
            for (int q = 0; q < QPT; q++) if (qc[q].active) candidate(q);
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int q = 0; q < QPT; q++) if (qc[q].active) lab_px(q, k);
#ifndef AHD_NO_SB
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        if (HDR && nonfinite_l) s_nonfinite[dir] = 1;
        AHD_STAMP(dir == 0 ? 5 : 9);      // P2 done
        __syncthreads();   // Lab of this direction complete; every thread is done with gq
        AHD_STAMP(dir == 0 ? 6 : 10);

        // ---- P3: homogeneity vote (pyx:22-58), all four pixels of the quad from one 4x4 Lab window
#pragma unroll
        for (int q = 0; q < QPT; q++) {
            const Quad& c = qc[q];
            if (!c.active) continue;
            const int lqy = c.lqy, lqx = c.lqx;
            const bool at_top = c.at_top, at_bot = c.at_bot, at_left = c.at_left, at_right = c.at_right;
            const bool literal = HDR && __builtin_amdgcn_readfirstlane(s_nonfinite[HDR ? dir : 0]) != 0;      // uniform over the workgroup
            // the upper pixel pair votes from window rows 0-2, then row 3 arrives (in row 0's registers) for the lower pair: 36 instead of 48 window registers.
            // HDR metric: L = luma (ahd.py:55,59) may be NaN or +-Inf, and then the comparisons the fast form takes for granted are false: a workgroup that wrote
            // such a value votes in the literal nine-cell form (LIT), every other one in the fast form; the two are separate code (one uniform branch around the whole vote)
            int cnt[4];
            auto votes = [&](auto LIT) {
                constexpr bool LITERAL = decltype(LIT)::value;
                float wl[4][4], wa[4][4], wq[4][4];
                float pc[6];               // chroma distances of the quad's six pixel pairs (vote_quad)
#ifdef AHD_VOTE_WHOLE_WINDOW
                load_lab_win(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl);
                load_lab_win(lab + LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wa);
                load_lab_win(lab + 2 * LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wq);
                if (LITERAL) { if (dir == 0) vote_quad_literal<0>(wl, wa, wq, cnt); else vote_quad_literal<1>(wl, wa, wq, cnt); }
                else { if (dir == 0) vote_quad<0, 0, 4>(wl, wa, wq, cnt, pc); else vote_quad<1, 0, 4>(wl, wa, wq, cnt, pc); }
#else
                load_lab_rows<0, 3>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl);
                load_lab_rows<0, 3>(lab + LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wa);
                load_lab_rows<0, 3>(lab + 2 * LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wq);
                if (LITERAL) { if (dir == 0) vote_quad_literal<0, 0, 2>(wl, wa, wq, cnt); else vote_quad_literal<1, 0, 2>(wl, wa, wq, cnt); }
                else { if (dir == 0) vote_quad<0, 0, 2>(wl, wa, wq, cnt, pc); else vote_quad<1, 0, 2>(wl, wa, wq, cnt, pc); }
                __builtin_amdgcn_sched_barrier(0);
                load_lab_rows<3, 4>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl);
                load_lab_rows<3, 4>(lab + LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wa);
                load_lab_rows<3, 4>(lab + 2 * LPR * LPS, lqy, lqx, at_top, at_bot, at_left, at_right, wq);
                if (LITERAL) { if (dir == 0) vote_quad_literal<0, 2, 4>(wl, wa, wq, cnt); else vote_quad_literal<1, 2, 4>(wl, wa, wq, cnt); }
                else { if (dir == 0) vote_quad<0, 2, 4>(wl, wa, wq, cnt, pc); else vote_quad<1, 2, 4>(wl, wa, wq, cnt, pc); }
#endif
            };
            // Lab mode 1, packed cells: integer chroma vote on the whole window; a wave in which some pixel's chroma lies 64 Lab units or more from a direction
            // neighbour's (ec >= 2^24: float32 rounding of the squares could then matter) recomputes its distances in float32 arithmetic on the same window
            auto votes_pk = [&]() {
                // optimistic: the integer form runs straight through (no branch in front of it -- a guard per window piece cost the kernel 7 %, A/B in
                // profiles/r4_ab_select_i16_guard_and_occupancy.log), the four epsilon distances are ORed on the way, and ONE wave-uniform test at the end
                // sends a wave that held a distance >= 2^24 through the exact float form, which overwrites the counts
                unsigned long long big;
                {
                    float wl[4][4]; unsigned wc[4][4], ec[4], pc[6];
                    load_labrows_pk<0, 3>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wc);
                    if (dir == 0) { vote_eps_pk<0, 0, 2>(wc, ec, pc); vote_cells_pk<0, 0, 2>(wl, wc, ec, pc, cnt); }
                    else { vote_eps_pk<1, 0, 2>(wc, ec, pc); vote_cells_pk<1, 0, 2>(wl, wc, ec, pc, cnt); }
                    const unsigned long long big0 = __builtin_amdgcn_ballot_w64(ec[0] >= (1u << 24)), big1 = __builtin_amdgcn_ballot_w64(ec[1] >= (1u << 24));
                    __builtin_amdgcn_sched_barrier(0);
                    load_labrows_pk<3, 4>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wc);
                    if (dir == 0) { vote_eps_pk<0, 2, 4>(wc, ec, pc); vote_cells_pk<0, 2, 4>(wl, wc, ec, pc, cnt); }
                    else { vote_eps_pk<1, 2, 4>(wc, ec, pc); vote_cells_pk<1, 2, 4>(wl, wc, ec, pc, cnt); }
                    // only the pixels whose votes are consumed (the tile + 1 px: the vote map's cells) may raise the flag -- the outer pixels of the halo quads
                    // look at the Lab buffer's guard ring, which nobody writes: left in, their garbage sent EVERY wave through the float form
                    // (profiles/r4_ab_select_i16_trailing_guard_unmasked.log: +17 %)
                    const unsigned long long rowT = __builtin_amdgcn_ballot_w64(lqy >= 1), rowB = __builtin_amdgcn_ballot_w64(lqy <= TQY);
                    const unsigned long long colL = __builtin_amdgcn_ballot_w64(lqx >= 1), colR = __builtin_amdgcn_ballot_w64(lqx <= TQX);
                    big = (big0 & rowT & colL) | (big1 & rowT & colR) |
                          (__builtin_amdgcn_ballot_w64(ec[2] >= (1u << 24)) & rowB & colL) | (__builtin_amdgcn_ballot_w64(ec[3] >= (1u << 24)) & rowB & colR);
                }
#ifndef AHD_I16_NOGUARD          // (timing experiment only: wrong on hard colour noise)
                if (big != 0) {          // uniform over the wave; never taken on ordinary content
                    s_nonfinite[2] = 1;                                                // (read after the barrier that precedes P4)
                    asm volatile("" ::: "memory");                                     // fresh loads: nothing of the integer form stays live into this path
#ifdef AHD_I16_FAST_FLOAT_FORM          // (A/B only: 78 VGPRs, six waves per SIMD -- the kernel keeps its seventh wave with the frugal form below)
                    // round 3's float vote (pair sharing, two window pieces) on converted values: the integer form's registers are dead here
#ifdef AHD_FLOAT_FORM_SHARE
                    constexpr bool FSHARE = true;
#else
                    constexpr bool FSHARE = false;             // (sharing the quad's pair distances costs six registers here: 78 VGPRs, six waves)
#endif
                    float wl[4][4], wa[4][4], wq[4][4], pcf[6];
                    load_lab_rows_pk_cvt<0, 3>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wa, wq);
                    if (dir == 0) vote_quad<0, 0, 2, FSHARE>(wl, wa, wq, cnt, pcf); else vote_quad<1, 0, 2, FSHARE>(wl, wa, wq, cnt, pcf);
                    __builtin_amdgcn_sched_barrier(0);
                    load_lab_rows_pk_cvt<3, 4>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wa, wq);
                    if (dir == 0) vote_quad<0, 2, 4, FSHARE>(wl, wa, wq, cnt, pcf); else vote_quad<1, 2, 4, FSHARE>(wl, wa, wq, cnt, pcf);
#else
                    float wl[4][4]; unsigned wc[4][4];
                    load_labrows_pk<0, 3>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wc);
                    unsigned pk = dir == 0 ? vote_quad_pk_f32<0, 0, 2>(wl, wc) : vote_quad_pk_f32<1, 0, 2>(wl, wc);
                    __builtin_amdgcn_sched_barrier(0);
                    load_labrows_pk<3, 4>(lab, lqy, lqx, at_top, at_bot, at_left, at_right, wl, wc);
                    pk |= dir == 0 ? vote_quad_pk_f32<0, 2, 4>(wl, wc) : vote_quad_pk_f32<1, 2, 4>(wl, wc);
#pragma unroll
                    for (int k = 0; k < 4; k++) cnt[k] = (int)((pk >> (4 * k)) & 15u);
#endif
                }
#endif
            };
            if constexpr (HDR) {
                if (literal) {
                    unsigned pk;
                    if constexpr (I16) pk = dir == 0 ? vote_quad_literal_lds_pk<0>(lab, lqy, lqx, at_top, at_bot, at_left, at_right)
                                                     : vote_quad_literal_lds_pk<1>(lab, lqy, lqx, at_top, at_bot, at_left, at_right);
                    else pk = dir == 0 ? vote_quad_literal_lds<0>(lab, lqy, lqx, at_top, at_bot, at_left, at_right)
                                       : vote_quad_literal_lds<1>(lab, lqy, lqx, at_top, at_bot, at_left, at_right);
#pragma unroll
                    for (int k = 0; k < 4; k++) cnt[k] = (int)((pk >> (4 * k)) & 15u);
                } else {
                    if constexpr (I16) votes_pk(); else votes(std::false_type{});
                }
            } else {
                if constexpr (I16) votes_pk(); else votes(std::false_type{});
            }
            if (dir == 0) {
                hvotes[q] = (unsigned)cnt[0] | ((unsigned)cnt[1] << 4) | ((unsigned)cnt[2] << 8) | ((unsigned)cnt[3] << 12);
            } else {
                // the mosaic planes are dead (the barrier above closed the last P2): the packed map h | v << 8 goes over them
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int yy = c.vmy + (k >> 1), xx = c.vmx + (k & 1);
                    if (yy >= 0 && yy < MPR && xx >= 0 && xx < 2 * TQX + 2)
                        vmap[yy * MPS + xx] = (unsigned short)(((hvotes[q] >> (4 * k)) & 15u) | ((unsigned)cnt[k] << 8));
                }
            }
        }
        if (dir == 0) {
            green_planes(1);   // the vertical planes replace the horizontal ones while the horizontal votes read the Lab buffer
            AHD_STAMP(7);      // P3(H) + P1(V) done
            __syncthreads();   // ... and the next P2 writes the Lab buffer only after every vote of this direction has read it
            AHD_STAMP(8);
        }
    }
#undef MWAT
    AHD_STAMP(11);             // P3(V) done
    __syncthreads();
    AHD_STAMP(12);
    if (I16 && tid == 0 && p.float_form_tiles != nullptr) {          // two cumulative words the host's layout policy samples: tiles that needed the float form, tiles launched
        if (s_nonfinite[2] != 0) atomicAdd(p.float_form_tiles, 1u);
        if (tbx == 0 && tby == 0) atomicAdd(p.float_form_tiles + 1, (unsigned)(((w + TQX - 1) / TQX) * ((h + TQY - 1) / TQY)));
    }

    // ---- P4: 3x3 box (cv2.blur, REFLECT_101; integer sums order like the float means), select, store
#pragma unroll
    for (int q = 0; q < QPT; q++) {
        const Quad& c = qc[q];
        if (!c.inner) continue;
        // 4x4 packed votes around the quad: rows vmy-1..vmy+2, cols vmx-1..vmx+2 (vmx-1 is even)
        unsigned int s012[4], s123[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned int* wp = reinterpret_cast<const unsigned int*>(&vmap[(c.vmy - 1 + r) * MPS + c.vmx - 1]);   // 4-byte aligned
            uint2 wv = make_uint2(wp[0], wp[1]);
            if (c.at_left) wv.x = (wv.x & 0xFFFF0000u) | (wv.y & 0xFFFFu);          // col -1 -> col 1
            if (c.at_right) wv.y = (wv.y & 0xFFFFu) | (wv.x & 0xFFFF0000u);         // col W -> col W-2
            unsigned int a = wv.x & 0xFFFFu, b = wv.x >> 16, cc = wv.y & 0xFFFFu, d = wv.y >> 16;
            s012[r] = a + b + cc; s123[r] = b + cc + d;
        }
        if (c.at_top) { s012[0] = s012[2]; s123[0] = s123[2]; }                     // row -1 -> row 1
        if (c.at_bot) { s012[3] = s012[1]; s123[3] = s123[1]; }                     // row H -> row H-2
        float px[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dy = k >> 1, dx = k & 1;
            unsigned int sm = dx ? (s123[dy] + s123[dy + 1] + s123[dy + 2]) : (s012[dy] + s012[dy + 1] + s012[dy + 2]);
            unsigned int sh = sm & 0xFFu, sv = sm >> 8;
            float cf = sh < sv ? 1.0f : 0.0f, nc = 1.0f - cf;            // ahd.py:139-145, literally
            px[k][0] = rgbh[q][k][0] * cf + rgbv[q][k][0] * nc;
            px[k][1] = rgbh[q][k][1] * cf + rgbv[q][k][1] * nc;
            px[k][2] = rgbh[q][k][2] * cf + rgbv[q][k][2] * nc;
            if (TAIL) colour_tail(p.tail, M, px[k][0], px[k][1], px[k][2]);
        }
        // two rows of three 8-byte stores instead of twelve dword stores; uniform tile origin + tile-local 32-bit offset (inner: lqy, lqx >= 1)
        store_quad_direct(p.out + ((size_t)(2 * tq0y) * W + 2 * tq0x) * 3, W, c.lqy - 1, c.lqx - 1, px);
    }
    AHD_STAMP(13);             // P4 done (stores issued)
}

// TINY: quarter planes narrower than 4 need the general (looping) border functions.
// HDR (image.get_hdr(), ahd.py:52-59) is a template parameter because its literal-vote path for non-finite luma costs registers.
// TAIL: a colour tail may follow the selection (only when no median stage does); the instance without it is the benchmark's.
template <bool TINY, bool U16, bool HDR, int LAB, bool TAIL>
__global__ void __launch_bounds__(NT_A, AHD_MIN_WAVES) k_ahd_select(AhdParams p) {
    __shared__ __attribute__((aligned(16))) float planes[SelLds<LAB>::N];
#ifdef AHD_LDS_PAD
    __shared__ float s_pad[AHD_LDS_PAD / 4];                                 // experiment: occupancy probe (LDS-limited workgroups per CU)
    if (p.H < 0) s_pad[threadIdx.x] = 1.0f;
#endif
    __shared__ float4 s_labtab[LAB == 0 ? LAB_SLOTS : 1];                    // 12 KB of closed-form tables (Lab mode 0 only; mode 1 reads its grid from L2)
    __shared__ int s_nonfinite[3];
    int tbx, tby;
    xcd_tile(tbx, tby);
    ahd_select_tile<TINY, U16, HDR, LAB, TAIL>(p, tbx, tby, planes, s_labtab, s_nonfinite);
}

// ================================================================================================
// Streaming form of kernel A (round 5): a workgroup walks DOWN a column of the image, 16 quad rows per pass, and every one of its 16 thread rows produces
// an output row.  The stand-alone tile above spends two of its 16 thread rows on halo (Lab and votes of the rows above and below its 14 output rows are
// computed again by the neighbouring tiles: 12.5 % of every phase); here the rows above come from the previous pass of the SAME workgroup:
//   * Lab of the last two pixel rows of a pass (both directions) and the votes of its pixel rows 29 and 30 are carried in LDS;
//   * the vote of a pixel needs the Lab row below it and the selection of a quad row the votes of the pixel row below it, so the bottom quad row of a pass
//     cannot be selected in that pass: its two candidates wait in an LDS stash (owned by the very threads that computed them) and are selected one pass later,
//     by the same threads -- thread row 15 selects the stashed row, thread rows 0..14 their own: 16 output rows per pass.
// Votes are computed for the pixel rows -1..30 of the pass (-1 = the last row of the previous pass), i.e. a vote thread takes the pixel pair one row ABOVE
// its CFA quad (the vote does not care about the CFA), from the Lab buffer rows 2 lqy .. 2 lqy + 3 as before: the buffer simply holds rows -2..31 now.
// A column is cut into CHUNKS (head pass: 14 rows, like a tile; every further pass: 16) which the workgroups of a persistent grid fetch from per-XCD
// queues (an XCD owns a contiguous range of columns: neighbours share their halo columns through its L2); the chunks get shorter towards the end of a queue
// (guided self-scheduling, api.cpp) so that the grid drains evenly.  Same arithmetic, pixel for pixel, as k_ahd_select: the parity tests run both.
// Lab mode 1, packed cells, no HDR metric (the instances the benchmark and BASELINE configs 2 and 5 use); everything else keeps the tile kernel.
namespace {
constexpr int S_CARRY_CELLS = 2 * 32;                       // two Lab pixel rows x 32 cells (px 0..31 of the region) per direction
struct StreamLds {
    static constexpr int NMW = 4 * MWY * MWX, NGQ = 4 * GY * GXS, NLAB = 2 * LPR * LPS;
    static constexpr int NCARRY = 2 * S_CARRY_CELLS * 2;    // floats: [dir][row][cell]{L, ab}
    static constexpr int NCV = 2 * MPS / 2;                 // floats: two vote rows of MPS uint16
    static constexpr int NSTASH = 22 * TQX;                 // floats: 22 candidate values of the TQX quads of thread row 15
    static constexpr int N = NMW + NGQ + NLAB + 2 + NCARRY + NCV + NSTASH;
};
constexpr int SMPR = 2 * LQY + 2;                           // vote map rows of the streaming form: pixel rows -3..30 of the pass
static_assert(SMPR * MPS * sizeof(unsigned short) <= StreamLds::NMW * sizeof(float), "the vote map fits over the mosaic planes");

// Rows [R0, R1) of the 4x4 window of cells for the streaming form: window row r = pixel row 2 lqy - 2 + r of the pass; the thread votes for the pixels of rows 1 and 2.
// BORDER_REFLECT (ahd.py:64) duplicates the image's edge rows: vt_top -- row 1 lies above the image (its own votes are never consumed), row 2 is the image's first
// row and sees itself there; vt_bot -- row 2 lies below the image, row 1 is the last row.
template <int R0, int R1>
DEVI void load_labrows_pk_s(const float* lab, int lqy, int lqx, bool vt_top, bool vt_bot, bool at_left, bool at_right, float wl[4][4], unsigned wc[4][4]) {
    const float4* p = reinterpret_cast<const float4*>(lab) + (2 * lqy) * LC4 + lqx;
#pragma unroll
    for (int r = R0; r < R1; r++) {
        const float4 a = p[r * LC4], b = p[r * LC4 + 1];
        wl[r][0] = a.x; wc[r][0] = __float_as_uint(a.y); wl[r][1] = a.z; wc[r][1] = __float_as_uint(a.w);
        wl[r][2] = b.x; wc[r][2] = __float_as_uint(b.y); wl[r][3] = b.z; wc[r][3] = __float_as_uint(b.w);
    }
    if (vt_top | vt_bot | at_left | at_right) {   // interior waves skip the selects
#pragma unroll
        for (int r = R0; r < R1; r++) {
            if (at_left) { wl[r][0] = wl[r][1]; wc[r][0] = wc[r][1]; }
            if (at_right) { wl[r][3] = wl[r][2]; wc[r][3] = wc[r][2]; }
        }
        if (R0 == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (vt_top) { wl[1][c] = wl[2][c]; wc[1][c] = wc[2][c]; }
                if (vt_bot) { wl[2][c] = wl[1][c]; wc[2][c] = wc[1][c]; }
            }
        }
    }
}

// Diagnostic build only (-DAHD_STAMPS): the stamps of the streaming form ACCUMULATE per wave -- slot i holds the sum over the wave's passes of (clock at boundary i -
// clock at the pass's first stamp), slot 15 the number of passes -- so that differences of neighbouring slots divided by the count are mean cycles per pass and phase.
#ifdef AHD_STAMPS
#define SSTAMP(i) do { if ((threadIdx.x & 63) == 0) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((i) == 0) st_t0 = t_; \
        const unsigned wv_ = blockIdx.x * (NT_A / 64) + (threadIdx.x >> 6); \
        if (wv_ < (unsigned)AHD_STAMP_WAVES) { g_ahd_stamps[(size_t)wv_ * AHD_NSTAMP + (i)] += t_ - st_t0; if ((i) == 13) g_ahd_stamps[(size_t)wv_ * AHD_NSTAMP + 15] += 1ull; } } } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
// One pass: region = quad rows Q0 .. Q0 + 15 of column tile tbx.  `chained`: the pass above was run by this workgroup just before (carry and stash are valid).
// Output: quad rows Q0 (chained) or Q0 + 1 (head) .. Q0 + 14, and the stashed row Q0 - 1 (chained), none below E (the chunk's last row).
template <bool U16, bool TAIL>
DEVI void ahd_stream_pass(const int tbx_in, const int Q0, const bool chained, const int E, float* const planes, int* const s_flag) {
    constexpr int NMW = StreamLds::NMW, NGQ = StreamLds::NGQ, NLAB = StreamLds::NLAB;
    int tbx = tbx_in;
    asm volatile("" : "+s"(tbx));                        // (what follows from the column is derived afresh in every pass as well)
    // The kernel's arguments are read from the argument segment again in every pass (scalar loads through a pointer the compiler cannot see through): loaded once
    // in front of the pass loop they all stay in scalar registers across it -- the colour matrix, pointers and sizes of every phase at once, on top of the loop's
    // state and the per-lane flag masks: more than the 102 scalar registers, and the spills go to vector register lanes (the one-pass build: 87 SGPRs, no spills).
    // AhdParams is the kernel's FIRST argument: offset 0 of the segment.
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    typedef __attribute__((address_space(4))) const AhdParams KArgs;
    KArgs* const pa = (KArgs*)ka;
    AhdParams p;
    p.src.f32 = pa->src.f32; p.src.u16 = pa->src.u16;
#pragma unroll
    for (int i = 0; i < 4; i++) { p.src.black[i] = pa->src.black[i]; p.src.sat[i] = pa->src.sat[i]; p.src.rsat[i] = pa->src.rsat[i]; }
    p.out = pa->out; p.labtab = nullptr; p.lablut = pa->lablut; p.H = pa->H; p.W = pa->W; p.hdr = 0; p.tail = pa->tail; p.float_form_tiles = pa->float_form_tiles;
#pragma unroll
    for (int i = 0; i < 3; i++) p.wb[i] = pa->wb[i];
#pragma unroll
    for (int i = 0; i < 9; i++) p.ccm.m[i] = pa->ccm.m[i];
    float* const mw = planes;
    float* const gq = planes + NMW;
    float* const lab = planes + NMW + NGQ;
    float* const carry_lab = lab + NLAB + 2;                                         // [dir][2 * 32 cells][2]
    unsigned* const carry_vote = reinterpret_cast<unsigned*>(carry_lab + StreamLds::NCARRY);   // two vote rows, MPS uint16 each
    float* const stash = carry_lab + StreamLds::NCARRY + StreamLds::NCV;              // [22][TQX]
    unsigned short* const vmap = reinterpret_cast<unsigned short*>(planes);           // [SMPR][MPS] votes h | v << 8 (row m = pixel row m - 3), over the dead mosaic planes

    // opaque per pass: everything derived from the thread index (LDS addresses, offsets, flags) is recomputed in every pass instead of being hoisted out of
    // the pass loop and kept in registers across it (hoisted: 127 VGPRs; the kernel needs <= 80 for six waves per SIMD)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int H = p.H, W = p.W, h = H >> 1, w = W >> 1;
    const int tq0x = tbx * TQX, tq0y = Q0 + 1;           // (the tile kernel's coordinates: its region starts one quad row above its tile)
    const double* M = p.ccm.m;
    const bool inside = tq0y >= 3 && tq0x >= 3 && tq0y + TQY + 3 <= h && tq0x + TQX + 3 <= w;
    unsigned long long st_t0 = 0; (void)st_t0;
    SSTAMP(0);
    __syncthreads();                                     // the previous pass is done with the vote map (over the mosaic planes)
    if (tid == 0) s_flag[2] = 0;

    // ---- P0: as in ahd_select_tile
    {
        static_assert(MWX <= 32 && NT_A % 32 == 0, "a pair row fits in 32 lanes");
        constexpr int RPP = NT_A / 32, NROWS = 2 * MWY, NL = (NROWS + RPP - 1) / RPP;
        const int r = tid >> 5, c = tid & 31;
        const bool on = c < MWX;
        const int cc = on ? c : MWX - 1;
        const int dy = r & 1;
        static_assert(RPP % 2 == 0, "row parity is a per-lane constant");
        float2 tmp[NL];
        if (!U16 && inside) {
            const char* const tile = reinterpret_cast<const char*>(p.src.f32 + (size_t)(2 * (tq0y - 3)) * W + 2 * (tq0x - 3));
            const unsigned rowb = (unsigned)W * 4u;
            const unsigned voff = mul24((unsigned)r, rowb) + 8u * (unsigned)cc;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int ry = (NROWS % RPP == 0 || r + k * RPP < NROWS) ? k * RPP : 0;
                tmp[k] = *reinterpret_cast<const float2*>(tile + (voff + (unsigned)ry * rowb));
            }
        } else {
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int ry = min(r + k * RPP, NROWS - 1), my = ry >> 1;
                int qi = tq0y - 3 + my, qj = tq0x - 3 + cc;
                if (!inside) { qi = b_sym1(qi, h); qj = b_sym1(qj, w); }
                tmp[k] = load_mosaic_pair<U16>(p.src, (size_t)(2 * qi + dy) * W + 2 * qj, dy != 0);
            }
        }
        const float w0 = dy ? p.wb[1] : p.wb[0], w1 = dy ? p.wb[2] : p.wb[1];
        float* const d0 = mw + ((dy ? P_G2 : P_R) * MWY + (r >> 1)) * MWX + cc;
        float* const d1 = mw + ((dy ? P_B : P_G1) * MWY + (r >> 1)) * MWX + cc;
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (on && (NROWS % RPP == 0 || r + k * RPP < NROWS)) {
                d0[k * (RPP / 2) * MWX] = tmp[k].x * w0;
                d1[k * (RPP / 2) * MWX] = tmp[k].y * w1;
            }
    }
    SSTAMP(1);
    __syncthreads();
    SSTAMP(2);

#define MWAT(pl, yy, xx) mw[((pl) * MWY + (yy)) * MWX + (xx)]
    auto green_planes = [&](const int dir) {
        for (int idx = tid; idx < GY * GX; idx += NT_A) {
            int gy = idx / GX, gx = idx - gy * GX;
            int a = gy + 1, c = gx + 1;
            if (!inside) {
                int ri = b_1011(tq0y - 2 + gy, h);
                int rj = b_1011(tq0x - 2 + gx, w);
                a = ri - (tq0y - 3); c = rj - (tq0x - 3);
                if (a < 1 || a > MWY - 2 || c < 1 || c > MWX - 2) continue;
            }
            const float rc = MWAT(P_R, a, c), bc = MWAT(P_B, a, c);
            float gr, gb;
            if (dir == 0) {
                gr = (((MWAT(P_R, a, c - 1) * AH0 + MWAT(P_G1, a, c - 1) * AH1) + rc * AH2) + MWAT(P_G1, a, c) * AH1) + MWAT(P_R, a, c + 1) * AH0;
                gb = (((MWAT(P_B, a, c - 1) * AH0 + MWAT(P_G2, a, c) * AH1) + bc * AH2) + MWAT(P_G2, a, c + 1) * AH1) + MWAT(P_B, a, c + 1) * AH0;
            } else {
                gr = (((MWAT(P_R, a - 1, c) * AH0 + MWAT(P_G2, a - 1, c) * AH1) + rc * AH2) + MWAT(P_G2, a, c) * AH1) + MWAT(P_R, a + 1, c) * AH0;
                gb = (((MWAT(P_B, a - 1, c) * AH0 + MWAT(P_G1, a, c) * AH1) + bc * AH2) + MWAT(P_G1, a + 1, c) * AH1) + MWAT(P_B, a + 1, c) * AH0;
            }
            const int gi = gy * GXS + gx;
            gq[0 * GY * GXS + gi] = gr; gq[1 * GY * GXS + gi] = gb;
            gq[2 * GY * GXS + gi] = rc - gr; gq[3 * GY * GXS + gi] = bc - gb;
        }
    };
    green_planes(0);
    SSTAMP(3);
    __syncthreads();
    SSTAMP(4);

    // This thread's quad: region row lqy = quad row Q0 + lqy, region column lqx = quad column tq0x - 1 + lqx.  The coordinates and the per-lane flags (scalar
    // register pairs) are derived afresh in every phase from an opaque copy of the thread index: kept across the phases, the flags of all of them together with
    // the pass loop's state exceed the scalar registers, and the spills go to vector register lanes (86 VGPRs; the kernel needs <= 80 for six waves per SIMD)
    auto fresh = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };

    float rgbh[4][3], rgbv[4][3];
    unsigned hvotes = 0;
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        // the two Lab pixel rows above the region, from the previous pass (wave 0: one cell per lane)
        if (chained && tid < S_CARRY_CELLS) {
            const float2 cell = reinterpret_cast<const float2*>(carry_lab)[dir * S_CARRY_CELLS + tid];
            reinterpret_cast<float2*>(lab)[(tid >> 5) * LPS + 1 + (tid & 31)] = cell;
        }
        float rr[4], gg[4], bb[4];
        const int t2 = fresh();
        const int lqy = t2 / LQX, lqx = t2 - lqy * LQX;
        const int qi = Q0 + lqy, qj = tq0x - 1 + lqx;
        const int gy = lqy + 1, gx = lqx + 1, my = lqy + 2, mx = lqx + 2;
        if (qi >= 0 && qi < h && qj >= 0 && qj < w) {
            const bool at_top = qi == 0, at_bot = qi == h - 1, at_left = qj == 0, at_right = qj == w - 1;
            const float* gR = gq, *gB = gq + GY * GXS, *dR = gq + 2 * GY * GXS, *dB = gq + 3 * GY * GXS;
            Win3 wgr = load_win<GXS>(gR, gy, gx), wgb = load_win<GXS>(gB, gy, gx);
            const float g1_c = dir == 0 ? MWAT(P_G1, my, mx) : rgbh[1][1], g2_c = dir == 0 ? MWAT(P_G2, my, mx) : rgbh[2][1];
            float Wn[4][4] = {{wgb.v[0][0], MWAT(P_G2, my - 1, mx), wgb.v[0][1], MWAT(P_G2, my - 1, mx + 1)},
                              {MWAT(P_G1, my, mx - 1), wgr.v[1][1], g1_c, wgr.v[1][2]},
                              {wgb.v[1][0], g2_c, wgb.v[1][1], MWAT(P_G2, my, mx + 1)},
                              {MWAT(P_G1, my + 1, mx - 1), wgr.v[2][1], MWAT(P_G1, my + 1, mx), wgr.v[2][2]}};
            if (at_top | at_bot | at_left | at_right) {
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
            float hf[4];
            highpass_quad(Wn, hf);
            float fg[4], fd[4];
            filt_base_tl(wgr, fg);
            { Win3 wd = load_win<GXS>(dR, gy, gx); filt_base_tl(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) rr[k] = fd[k] + (fg[k] + hf[k]);
            filt_base_br(wgb, fg);
            { Win3 wd = load_win<GXS>(dB, gy, gx); filt_base_br(wd, fd); }
#pragma unroll
            for (int k = 0; k < 4; k++) bb[k] = fd[k] + (fg[k] + hf[k]);
            gg[0] = wgr.v[1][1]; gg[1] = g1_c; gg[2] = g2_c; gg[3] = wgb.v[1][1];
            float (&rgbc)[4][3] = dir == 0 ? rgbh : rgbv;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float L; unsigned ab;
                homog_lab_pk(p.lablut, rr[k], gg[k], bb[k], p.wb, M, 0, L, ab);
                // pixel (py, px) of the region lives at buffer row py + 2 (rows 0, 1: carried), column px + 1
                float2* const o = reinterpret_cast<float2*>(lab) + (2 * lqy + 2 + (k >> 1)) * LPS + 2 * lqx + 1 + (k & 1);
                *o = make_float2(L, __uint_as_float(ab));
                rgbc[k][0] = rr[k]; rgbc[k][1] = gg[k]; rgbc[k][2] = bb[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SSTAMP(dir == 0 ? 5 : 9);
        __syncthreads();
        SSTAMP(dir == 0 ? 6 : 10);

        // ---- P3: votes of the pixel pair rows (2 lqy - 1, 2 lqy) of the region, from buffer rows 2 lqy .. 2 lqy + 3: the vote thread's pixel pair lies one
        // pixel row above its quad, image rows 2 qi - 1 (upper) and 2 qi (lower)
        if (qj >= 0 && qj < w && qi >= 0 && qi <= h) {
            const bool vt_top = qi == 0, vt_bot = qi == h, at_left = qj == 0, at_right = qj == w - 1;
            int cnt[4];
            unsigned long long big;
            {
                float wl[4][4]; unsigned wc[4][4], ec[4], pc[6];
                load_labrows_pk_s<0, 3>(lab, lqy, lqx, vt_top, vt_bot, at_left, at_right, wl, wc);
                if (dir == 0) { vote_eps_pk<0, 0, 2>(wc, ec, pc); vote_cells_pk<0, 0, 2>(wl, wc, ec, pc, cnt); }
                else { vote_eps_pk<1, 0, 2>(wc, ec, pc); vote_cells_pk<1, 0, 2>(wl, wc, ec, pc, cnt); }
                const unsigned long long big0 = __builtin_amdgcn_ballot_w64(ec[0] >= (1u << 24)), big1 = __builtin_amdgcn_ballot_w64(ec[1] >= (1u << 24));
                __builtin_amdgcn_sched_barrier(0);
                load_labrows_pk_s<3, 4>(lab, lqy, lqx, vt_top, vt_bot, at_left, at_right, wl, wc);
                if (dir == 0) { vote_eps_pk<0, 2, 4>(wc, ec, pc); vote_cells_pk<0, 2, 4>(wl, wc, ec, pc, cnt); }
                else { vote_eps_pk<1, 2, 4>(wc, ec, pc); vote_cells_pk<1, 2, 4>(wl, wc, ec, pc, cnt); }
                // only votes that are consumed may raise the flag (see the tile kernel): image pixels whose window holds real Lab values
                const bool vote_rows_ok = chained || lqy >= 1;                       // a head pass has no Lab above its region
                const unsigned long long rowU = __builtin_amdgcn_ballot_w64(vote_rows_ok && qi >= 1), rowL = __builtin_amdgcn_ballot_w64(vote_rows_ok && qi < h);
                const unsigned long long colL = __builtin_amdgcn_ballot_w64(lqx >= 1), colR = __builtin_amdgcn_ballot_w64(lqx <= TQX);
                big = (big0 & rowU & colL) | (big1 & rowU & colR) |
                      (__builtin_amdgcn_ballot_w64(ec[2] >= (1u << 24)) & rowL & colL) | (__builtin_amdgcn_ballot_w64(ec[3] >= (1u << 24)) & rowL & colR);
            }
            if (big != 0) {
                s_flag[2] = 1;
                asm volatile("" ::: "memory");
                float wl[4][4]; unsigned wc[4][4];
                load_labrows_pk_s<0, 3>(lab, lqy, lqx, vt_top, vt_bot, at_left, at_right, wl, wc);
                unsigned pk = dir == 0 ? vote_quad_pk_f32<0, 0, 2>(wl, wc) : vote_quad_pk_f32<1, 0, 2>(wl, wc);
                __builtin_amdgcn_sched_barrier(0);
                load_labrows_pk_s<3, 4>(lab, lqy, lqx, vt_top, vt_bot, at_left, at_right, wl, wc);
                pk |= dir == 0 ? vote_quad_pk_f32<0, 2, 4>(wl, wc) : vote_quad_pk_f32<1, 2, 4>(wl, wc);
#pragma unroll
                for (int k = 0; k < 4; k++) cnt[k] = (int)((pk >> (4 * k)) & 15u);
            }
            if (dir == 0) {
                hvotes = (unsigned)cnt[0] | ((unsigned)cnt[1] << 4) | ((unsigned)cnt[2] << 8) | ((unsigned)cnt[3] << 12);
            } else {
                // vote map row m = pixel row m - 3 of the region: this thread's pixel rows 2 lqy - 1, 2 lqy -> m = 2 lqy + 2, 2 lqy + 3; column = px + 1 - 2 (tile origin - 1 px)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int yy = 2 * lqy + 2 + (k >> 1), xx = 2 * lqx - 1 + (k & 1);
                    if (xx >= 0 && xx < 2 * TQX + 2)
                        vmap[yy * MPS + xx] = (unsigned short)(((hvotes >> (4 * k)) & 15u) | ((unsigned)cnt[k] << 8));
                }
            }
        }
        // the last two Lab pixel rows of this direction (buffer rows 32, 33) travel to the next pass (wave 3: one cell per lane)
        if (tid >= NT_A - S_CARRY_CELLS) {
            const int j = tid - (NT_A - S_CARRY_CELLS);
            reinterpret_cast<float2*>(carry_lab)[dir * S_CARRY_CELLS + j] = reinterpret_cast<const float2*>(lab)[(2 * LQY + (j >> 5)) * LPS + 1 + (j & 31)];
        }
        if (dir == 0) {
            green_planes(1);
            SSTAMP(7);
            __syncthreads();
            SSTAMP(8);
        } else if (chained && tid < MPS) {
            reinterpret_cast<unsigned*>(vmap)[tid] = carry_vote[tid];      // vote rows m = 0, 1 (pixel rows -3, -2): the previous pass's rows 29, 30
        }
    }
#undef MWAT
    SSTAMP(11);
    __syncthreads();
    SSTAMP(12);
    if (tid == 0 && p.float_form_tiles != nullptr && s_flag[2] != 0) atomicAdd(p.float_form_tiles, 1u);

    // ---- P4: thread rows 0..14 select their own quad row, thread row 15 the row stashed by the previous pass (quad row Q0 - 1); then the stash takes row 15's candidates
    const int t4 = fresh();
    const int lqy = t4 / LQX, lqx = t4 - lqy * LQX;
    const int qi = Q0 + lqy, qj = tq0x - 1 + lqx;
    const bool active = qi >= 0 && qi < h && qj >= 0 && qj < w;
    const bool at_left = qj == 0, at_right = qj == w - 1;
    const bool last_row = lqy == LQY - 1;
    const bool col_ok = lqx >= 1 && lqx <= TQX && qj < w;
    bool sel;
    int orow;                                                                        // output quad row relative to Q0 - 1
    if (last_row) {
        sel = chained && col_ok;                                                     // (a chunk continues only while rows remain: Q0 - 1 <= E)
        orow = 0;
        if (col_ok) {
            // swap: the stash's candidates into the registers, this pass's into the stash (thread-private slots: no barrier)
            float* const sp = stash + (lqx - 1);
            int n = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    { const float old = sp[n * TQX]; sp[n * TQX] = rgbh[k][c]; rgbh[k][c] = old; n++; }
                    if (!(c == 1 && (k == 1 || k == 2))) { const float old = sp[n * TQX]; sp[n * TQX] = rgbv[k][c]; rgbv[k][c] = old; n++; }
                    else rgbv[k][c] = rgbh[k][c];                                   // the quad's own two green samples are the same in both candidates
                }
        }
    } else {
        sel = active && col_ok && (chained || lqy >= 1) && qi <= E;
        orow = lqy + 1;
    }
    if (sel) {
        const int sqi = Q0 - 1 + orow;                                               // the selected quad row in the image
        const bool s_top = sqi == 0, s_bot = sqi == h - 1;
        const int m0 = 2 * orow;                                                     // vote rows m0 .. m0 + 3 = pixel rows 2 sqi - 1 .. 2 sqi + 2
        const int vmx = 2 * lqx - 1;
        unsigned int s012[4], s123[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned int* wp = reinterpret_cast<const unsigned int*>(&vmap[(m0 + r) * MPS + vmx - 1]);
            uint2 wv = make_uint2(wp[0], wp[1]);
            if (at_left) wv.x = (wv.x & 0xFFFF0000u) | (wv.y & 0xFFFFu);
            if (at_right) wv.y = (wv.y & 0xFFFFu) | (wv.x & 0xFFFF0000u);
            unsigned int a = wv.x & 0xFFFFu, b = wv.x >> 16, cc = wv.y & 0xFFFFu, d = wv.y >> 16;
            s012[r] = a + b + cc; s123[r] = b + cc + d;
        }
        if (s_top) { s012[0] = s012[2]; s123[0] = s123[2]; }
        if (s_bot) { s012[3] = s012[1]; s123[3] = s123[1]; }
        float px[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int dy = k >> 1, dx = k & 1;
            unsigned int sm = dx ? (s123[dy] + s123[dy + 1] + s123[dy + 2]) : (s012[dy] + s012[dy + 1] + s012[dy + 2]);
            unsigned int sh = sm & 0xFFu, sv = sm >> 8;
            float cf = sh < sv ? 1.0f : 0.0f, nc = 1.0f - cf;
            px[k][0] = rgbh[k][0] * cf + rgbv[k][0] * nc;
            px[k][1] = rgbh[k][1] * cf + rgbv[k][1] * nc;
            px[k][2] = rgbh[k][2] * cf + rgbv[k][2] * nc;
            if (TAIL) colour_tail(p.tail, M, px[k][0], px[k][1], px[k][2]);
        }
        // origin = quad (Q0 - 1, tq0x) of the image (Q0 - 1 >= 0 whenever row 0 of this origin is stored: chained passes only)
        char* const origin = reinterpret_cast<char*>(p.out) + ((long long)(2 * (Q0 - 1)) * W + 2 * tq0x) * 12;
        store_quad_direct(reinterpret_cast<float*>(origin), W, orow, lqx - 1, px);
    }
    // the votes of pixel rows 29, 30 (rows m = 32, 33) for the next pass
    if (tid < MPS) carry_vote[tid] = reinterpret_cast<const unsigned*>(vmap)[(SMPR - 2) * (MPS / 2) + tid];
    SSTAMP(13);
}
}  // namespace

struct AhdStreamQueues {
    const int4* chunks;          // [0..3]: header -- first[8] then count[8]: XCD x owns chunks[4 + first[x] .. 4 + first[x] + count[x]); then one entry per chunk:
                                 // { column tile, S = region origin quad row of the head pass, E = last output quad row, passes }
    unsigned passes_total;       // of the whole launch (the layout policy's "tiles launched")
};
#ifndef STREAM_MIN_WAVES
#define STREAM_MIN_WAVES 1
#endif
template <bool U16, bool TAIL>
__global__ void __launch_bounds__(NT_A, STREAM_MIN_WAVES) k_ahd_select_stream(AhdParams p, AhdStreamQueues q) {
    __shared__ __attribute__((aligned(16))) float planes[StreamLds::N];
    __shared__ int s_flag[4];
    const unsigned xcd = blockIdx.x & 7u;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.float_form_tiles != nullptr) atomicAdd(p.float_form_tiles + 1, q.passes_total);
    {
        // one workgroup per chunk, dispatched by the hardware in queue order (the j-th workgroup of XCD x takes the j-th chunk of queue x): no fetch, no exit protocol
        const unsigned* const hdr = reinterpret_cast<const unsigned*>(q.chunks);
        const unsigned idx = blockIdx.x >> 3;
        if (idx >= hdr[8 + xcd]) return;
        const int4 chv = q.chunks[4 + hdr[xcd] + idx];
        const int4 ch = make_int4(__builtin_amdgcn_readfirstlane(chv.x), __builtin_amdgcn_readfirstlane(chv.y), __builtin_amdgcn_readfirstlane(chv.z), __builtin_amdgcn_readfirstlane(chv.w));
        int Q0 = ch.y;
        bool chained = false;
#ifdef STREAM_ONEPASS      // (register-pressure probe only)
        ahd_stream_pass<U16, TAIL>(ch.x, Q0, ch.w != 0, ch.z, planes, s_flag);
#else
        for (;;) {
            ahd_stream_pass<U16, TAIL>(ch.x, Q0, chained, ch.z, planes, s_flag);
            if (Q0 + LQY - 1 > ch.z) break;
            Q0 += LQY; chained = true;
        }
#endif
    }
}

// ================================================================================================
// Kernel B: one chroma post-process stage (ahd.py:148-161) + optional colour tail.
//   r' = med5(r-g)+g ; b' = med5(b-g)+g ; g' = (med5(g-r') + med5(g-b') + r' + b') / 2
// cv2.medianBlur(.,5): exact 5x5 median, BORDER_REPLICATE.
namespace {
// Geometry (measured history in DESIGN.md 7.1).  A workgroup produces 60x28 px.  Its first-level region (the tile + the halo of 2 the
// second level reads) is 64x32 px = 256 runs of eight pixels, exactly one per thread, and every thread later computes the second level
// on the SAME eight pixels, so r', b' never leave its registers; the runs at the left / right edge of the tile carry two pixels of halo
// whose second-level medians are simply not used.  Round 2's first geometry (32x32 px, runs of four, the 272-px halo ring done in
// pixel pairs by half of the waves) issued 5.3 network operations per wave and output pixel; this one 4.5.
#ifndef MED_BTY
#define MED_BTY 28                               // (60: 64x64 region, 512 threads, two workgroups per CU -- measured 5 % slower)
#endif
constexpr int BTX = 60, BTY = MED_BTY;           // output tile
constexpr int RX = BTX + 4, RY = BTY + 4;        // first-level region
constexpr int B4X = BTX + 8, B4Y = BTY + 8;     // g, r-g, b-g planes (halo 4); row stride 68 floats: rows of 8-float runs alternate bank halves
constexpr int DPAD = 4, DST = RX + DPAD;         // g-r', g-b' planes: region column c is stored at DPAD + c (16-byte aligned runs); a window that
                                                 // starts two columns left of the region reads the pad / the previous row's tail: halo-only medians
constexpr int NT_B = (BTX + 4) / 8 * (BTY + 4);
static_assert((RX / 8) * RY == NT_B, "one run of eight per thread");

// Median of 25 by selection networks over min / max / med3 (exact; order independent).  The networks are spelled with raw VALU
// instructions: fminf / fmaxf are llvm.minnum / maxnum, for which the backend (IEEE mode) first canonicalises every operand it
// cannot prove quiet -- one extra v_max_f32 x, x per value loaded from LDS.  A signalling NaN is the only input the two forms
// treat differently, and those are outside the contract.  Plain (non-volatile) asm with register operands:
// the compiler schedules and allocates around it as usual.
DEVI float vmin2(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEVI float vmax2(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEVI float vmin3(float a, float b, float c) { float d; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
DEVI float vmax3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
DEVI float vmed3(float a, float b, float c) { float d; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
#define MN2(a, b) vmin2(a, b)
#define MX2(a, b) vmax2(a, b)
#define MN3(a, b, c) vmin3(a, b, c)
#define MX3(a, b, c) vmax3(a, b, c)
#define MD3(a, b, c) vmed3(a, b, c)
// The medians of eight horizontally adjacent pixels from one 5x12 window: columns sorted by med3 insertion, neighbouring columns merged
// pairwise and shared by up to four windows, of a pixel pair's 20 common samples only the six ranks that a fifth column can still turn
// into the median, five nested med3 finish a window; the graph is then re-synthesised with three-input cells (tools/gen_median_run.py
// builds it, picks the cheapest cover and verifies every window of the result on all 2^25 binary inputs): 375 operations, 46.9 per
// median, against 98 for the classic 99-exchange network with shared triples (round 1).
// The window is loaded in pieces: the include calls MED_NEED(c) before the first use of column c (order 1 2 3 4 0 5 ... 11), and
// need(c, w) fetches the piece that starts there, so the last columns are not held in registers while the first medians are computed.
template <class Need>
DEVI void median25_run8(Need need, float m[8]) {
    float w[5][12];
    float m0, m1, m2, m3, m4, m5, m6, m7;
#define MED_NEED(c) need(c, w);
#include "median25_run8.inc"
#undef MED_NEED
    m[0] = m0; m[1] = m1; m[2] = m2; m[3] = m3; m[4] = m4; m[5] = m5; m[6] = m6; m[7] = m7;
}
#undef MN2
#undef MX2
#undef MN3
#undef MX3
#undef MD3
}  // namespace

struct MedParams {
    const float* in;   // (H,W,3)
    float* out;        // (H,W,3)
    int H, W;
    int tail;
    int vec;           // W % 4 == 0 and both images 16-byte aligned: 16-byte global accesses
    Ccm ccm;
};

#ifndef MED_MIN_WAVES
#define MED_MIN_WAVES 1                            // 46.9 KB of LDS: three workgroups per CU, registers are not the limit
#endif
constexpr int MED_LDS_FLOATS = 3 * B4Y * B4X;
// One 60x28 px tile of the median stage; a device function for the same reason as ahd_select_tile.  `lds_base`: MED_LDS_FLOATS floats of LDS, 16-byte aligned.
DEVI void ahd_median_tile(const MedParams& p, const int tbx, const int tby, float* const lds_base) {
    // g, r-g, b-g (halo 4); the second level's inputs g-r', g-b' (halo 2) are laid over r-g, b-g once every thread is done with those:
    // 29.4 KB, five workgroups per CU (the stage loses 16 % from three workgroups per CU to two)
    float (*const lds)[B4Y][B4X] = reinterpret_cast<float (*)[B4Y][B4X]>(lds_base);
    float (*s_g)[B4X] = lds[0], (*s_drg)[B4X] = lds[1], (*s_dbg)[B4X] = lds[2];
    float *s_d1 = &lds[1][0][0], *s_d2 = &lds[2][0][0];
    static_assert(RY * DST + 4 <= B4Y * B4X, "a difference plane fits in the plane it replaces");
    const int tid = threadIdx.x, H = p.H, W = p.W;
    const int tx0 = tbx * BTX, ty0 = tby * BTY;
    const unsigned rowbytes = (unsigned)W * 12u;

    // ---- load: all global loads of a thread are issued before its first LDS store (they are in flight together)
    const bool inside = ty0 >= 4 && tx0 >= 4 && ty0 + BTY + 4 <= H && tx0 + BTX + 4 <= W;
    if (inside && p.vec) {
        // four pixels = 48 bytes = three 16-byte loads; the tile row starts on a 16-byte boundary (W, tx0 multiples of four)
        constexpr int GPR = B4X / 4, NG = GPR * B4Y, NLG = (NG + NT_B - 1) / NT_B;
        // uniform 64-bit tile origin + tile-local 32-bit byte offset per lane: (scalar base, vector offset) accesses, no 64-bit vector arithmetic
        const char* const tile = reinterpret_cast<const char*>(p.in + ((size_t)(ty0 - 4) * W + (tx0 - 4)) * 3);
        float4 t[NLG][3];
#ifdef MED_LOADER_INCREMENTAL
        // Group gi = tid + 256 k of the 36 x 17 groups (round 4): row and column by ONE division, for k = 0; 256 = 15 * 17 + 1, so the next group lies 15 rows down and
        // one column right, with a wrap into the following row -- uniform byte steps and a compare per pass.  A plane row is exactly 17 groups (B4X = 68 floats), so the
        // LDS address is linear in gi: 16 * tid plus immediates.  (Before: a division, two 24-bit multiplies and a 64-bit mad per pass and side -- 21 multiplier-class
        // instructions of the loader's 100.)
        static_assert(B4X == 4 * GPR && NT_B == 15 * GPR + 1 && NLG == 3 && NG - 2 * NT_B > 0, "the incremental form below is spelled for 36 x 17 groups and 256 threads");
        const unsigned r0 = (unsigned)tid / (unsigned)GPR, g0 = (unsigned)tid - r0 * (unsigned)GPR;
        const unsigned off0 = __umul24(r0, rowbytes) + __umul24(g0, 48u);
        const unsigned step = 15u * rowbytes + 48u, wrap = rowbytes - 48u * (unsigned)GPR;                       // uniform
        const unsigned off1 = off0 + step + (g0 >= (unsigned)GPR - 1u ? wrap : 0u);
        unsigned off2 = off0 + 2u * step + (g0 >= (unsigned)GPR - 2u ? wrap : 0u);
        const bool third = tid < NG - 2 * NT_B;
        if (!third) off2 = off0;                                                                                   // lanes without a third group load their first one again
        {
            const float4* s = reinterpret_cast<const float4*>(tile + off0);
            t[0][0] = s[0]; t[0][1] = s[1]; t[0][2] = s[2];
            s = reinterpret_cast<const float4*>(tile + off1);
            t[1][0] = s[0]; t[1][1] = s[1]; t[1][2] = s[2];
            s = reinterpret_cast<const float4*>(tile + off2);
            t[2][0] = s[0]; t[2][1] = s[1]; t[2][2] = s[2];
        }
        float4* const dg = reinterpret_cast<float4*>(&s_g[0][0]) + tid;
        float4* const dr = reinterpret_cast<float4*>(&s_drg[0][0]) + tid;
        float4* const db = reinterpret_cast<float4*>(&s_dbg[0][0]) + tid;
#pragma unroll
        for (int k = 0; k < NLG; k++) {
            if (k < 2 || third) {
                const float4 a = t[k][0], b = t[k][1], c = t[k][2];      // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
                dg[k * NT_B] = make_float4(a.y, b.x, b.w, c.z);
                dr[k * NT_B] = make_float4(a.x - a.y, a.w - b.x, b.z - b.w, c.y - c.z);
                db[k * NT_B] = make_float4(a.z - a.y, b.y - b.x, c.x - b.w, c.w - c.z);
            }
        }
#else
#pragma unroll
        for (int k = 0; k < NLG; k++) {
            int gi = tid + k * NT_B;
            if (gi >= NG) gi = NG - 1;
            const int ly = gi / GPR, lg = gi - ly * GPR;
            const float4* s = reinterpret_cast<const float4*>(tile + (mul24((unsigned)ly, rowbytes) + 48u * (unsigned)lg));
            t[k][0] = s[0]; t[k][1] = s[1]; t[k][2] = s[2];
        }
#pragma unroll
        for (int k = 0; k < NLG; k++) {
            const int gi = tid + k * NT_B;
            if (gi < NG) {
                const int ly = gi / GPR, lx = 4 * (gi - ly * GPR);
                const float4 a = t[k][0], b = t[k][1], c = t[k][2];      // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
                *reinterpret_cast<float4*>(&s_g[ly][lx]) = make_float4(a.y, b.x, b.w, c.z);
                *reinterpret_cast<float4*>(&s_drg[ly][lx]) = make_float4(a.x - a.y, a.w - b.x, b.z - b.w, c.y - c.z);
                *reinterpret_cast<float4*>(&s_dbg[ly][lx]) = make_float4(a.z - a.y, b.y - b.x, c.x - b.w, c.w - c.z);
            }
        }
#endif
    } else {
        constexpr int NL = (B4Y * B4X + NT_B - 1) / NT_B;
        float tr[NL], tg[NL], tb[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            int idx = tid + k * NT_B;
            if (idx >= B4Y * B4X) idx = B4Y * B4X - 1;
            int ly = idx / B4X, lx = idx - ly * B4X;
            int y = b_rep(ty0 - 4 + ly, H), x = b_rep(tx0 - 4 + lx, W);
            const float* s = p.in + ((size_t)y * W + x) * 3;
            tr[k] = s[0]; tg[k] = s[1]; tb[k] = s[2];
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            int idx = tid + k * NT_B;
            if (idx < B4Y * B4X) {
                int ly = idx / B4X, lx = idx - ly * B4X;
                s_g[ly][lx] = tg[k]; s_drg[ly][lx] = tr[k] - tg[k]; s_dbg[ly][lx] = tb[k] - tg[k];
            }
        }
    }
    __syncthreads();

    // ---- first level: r' = med5(r-g)+g, b' = med5(b-g)+g on this thread's run, the differences g-r', g-b' to LDS
    const int oy = tid >> 3, ox = (tid & 7) * 8;                 // run position in the region (region (0,0) = image (ty0-2, tx0-2))
    const int y = ty0 - 2 + oy, x0 = tx0 - 2 + ox;
    float keep_r[8], keep_b[8], gg[8];
    {   // positions outside the image are computed from the clamped planes like any other and replaced by the border pass below
        float m[8];
        {
            const float* gr = &s_g[oy + 2][ox];
            const float2 ga = *reinterpret_cast<const float2*>(gr + 2), gd = *reinterpret_cast<const float2*>(gr + 8);
            const float4 gb = *reinterpret_cast<const float4*>(gr + 4);
            gg[0] = ga.x; gg[1] = ga.y; gg[2] = gb.x; gg[3] = gb.y; gg[4] = gb.z; gg[5] = gb.w; gg[6] = gd.x; gg[7] = gd.y;
        }
        auto piece = [&](const float (*plane)[B4X]) {             // 16-byte aligned window: columns 0-3, then pairs as they are needed
            return [=](int c, float w[5][12]) {
                if (c != 1 && !(c >= 4 && !(c & 1))) return;
#pragma unroll
                for (int dy = 0; dy < 5; dy++) {
                    const float* r = &plane[oy + dy][ox];
                    if (c == 1) {
                        const float4 v = *reinterpret_cast<const float4*>(r);
                        w[dy][0] = v.x; w[dy][1] = v.y; w[dy][2] = v.z; w[dy][3] = v.w;
                    } else {
                        const float2 v = *reinterpret_cast<const float2*>(r + c);
                        w[dy][c] = v.x; w[dy][c + 1] = v.y;
                    }
                }
            };
        };
        median25_run8(piece(s_drg), m);
#pragma unroll
        for (int q = 0; q < 8; q++) keep_r[q] = m[q] + gg[q];
        median25_run8(piece(s_dbg), m);
#pragma unroll
        for (int q = 0; q < 8; q++) keep_b[q] = m[q] + gg[q];
    }
    __syncthreads();
    {
        float4* d1 = reinterpret_cast<float4*>(&s_d1[oy * DST + DPAD + ox]);
        float4* d2 = reinterpret_cast<float4*>(&s_d2[oy * DST + DPAD + ox]);
        d1[0] = make_float4(gg[0] - keep_r[0], gg[1] - keep_r[1], gg[2] - keep_r[2], gg[3] - keep_r[3]);
        d1[1] = make_float4(gg[4] - keep_r[4], gg[5] - keep_r[5], gg[6] - keep_r[6], gg[7] - keep_r[7]);
        d2[0] = make_float4(gg[0] - keep_b[0], gg[1] - keep_b[1], gg[2] - keep_b[2], gg[3] - keep_b[3]);
        d2[1] = make_float4(gg[4] - keep_b[4], gg[5] - keep_b[5], gg[6] - keep_b[6], gg[7] - keep_b[7]);
    }
    __syncthreads();
    // medianBlur replicates the border of ITS input plane: a position outside the image takes the values of
    // the clamped position (only tiles touching the image border have any).
    if (ty0 < 2 || tx0 < 2 || ty0 + BTY + 2 > H || tx0 + BTX + 2 > W) {
        for (int idx = tid; idx < RY * RX; idx += NT_B) {
            const int py = idx / RX, px = idx - py * RX;
            const int yy = ty0 - 2 + py, xx = tx0 - 2 + px;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) continue;
            const int cy = b_rep(yy, H) - (ty0 - 2), cx = b_rep(xx, W) - (tx0 - 2);
            if (cy < 0 || cy >= RY || cx < 0 || cx >= RX) continue;     // beyond a partial tile: never consumed
            s_d1[py * DST + DPAD + px] = s_d1[cy * DST + DPAD + cx]; s_d2[py * DST + DPAD + px] = s_d2[cy * DST + DPAD + cx];
        }
        __syncthreads();
    }
    // ---- second level on the same run: g' = (med5(g-r') + med5(g-b') + r' + b') / 2, colour tail, store
    const int qlo = ox == 0 ? 2 : 0;                              // pixels [qlo, qhi) of the run belong to this tile and to the image
    const int qhi = min(ox == RX - 8 ? 6 : 8, W - x0);
    const bool second = oy >= 2 && oy < RY - 2 && y < H && qhi > qlo;           // no early return: the staged tail below has barriers
    float ma[8], mb[8];
    if (second) {
        auto piece = [&](const float* plane) {                     // window starts 8 bytes past a 16-byte boundary: columns 0-1, 2-5, then pairs
            return [=](int c, float w[5][12]) {
                if (c != 1 && c != 2 && !(c >= 6 && !(c & 1))) return;
#pragma unroll
                for (int dy = 0; dy < 5; dy++) {
                    const float* r = plane + (oy - 2 + dy) * DST + DPAD + ox - 2;
                    if (c == 2) {
                        const float4 b = *reinterpret_cast<const float4*>(r + 2);
                        w[dy][2] = b.x; w[dy][3] = b.y; w[dy][4] = b.z; w[dy][5] = b.w;
                    } else {
                        const int c0 = c == 1 ? 0 : c;
                        const float2 a = *reinterpret_cast<const float2*>(r + c0);
                        w[dy][c0] = a.x; w[dy][c0 + 1] = a.y;
                    }
                }
            };
        };
        median25_run8(piece(s_d1), ma);
        median25_run8(piece(s_d2), mb);
    }
    // With a colour tail the pre-tail pixels go through an LDS image of the tile (over the planes, which are dead by now) and are picked up
    // again four at a time by consecutive threads: the tail then runs on 28 wave-passes of 4 px per tile instead of 32 of 8 runs that carry
    // idle rows and halo pixels, and the result leaves as three 16-byte stores per thread on consecutive addresses.
    if (p.tail != 0 && p.vec) {
        constexpr int SROW = BTX * 3;                             // floats per staged row (720 bytes: 16-byte aligned rows)
        float* const stage = &lds[0][0][0];
        static_assert(BTY * SROW <= 3 * B4Y * B4X, "the staged tile fits over the planes");
        __syncthreads();                                          // every thread is done with the difference planes
        if (second) {
            float o[24];
#pragma unroll
            for (int q = 0; q < 8; q++) { o[3 * q] = keep_r[q]; o[3 * q + 1] = (((ma[q] + mb[q]) + keep_r[q]) + keep_b[q]) / 2.0f; o[3 * q + 2] = keep_b[q]; }
            float* sp = stage + (oy - 2) * SROW + (ox - 2) * 3;   // 8-byte aligned; floats [3 qlo, 3 qhi) of the run are this tile's
#pragma unroll
            for (int k = 0; k < 12; k++)
                if (2 * k >= 3 * qlo && 2 * k < 3 * qhi) *reinterpret_cast<float2*>(sp + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
        }
        __syncthreads();
        constexpr int GPR = BTX / 4, NGRP = GPR * BTY;
        char* const tile = reinterpret_cast<char*>(p.out + ((size_t)ty0 * W + tx0) * 3);
#pragma unroll
        for (int k = 0; k < (NGRP + NT_B - 1) / NT_B; k++) {
            const int gi = tid + k * NT_B;
            const int row = gi / GPR, c = gi - row * GPR;
            if (gi < NGRP && ty0 + row < H && tx0 + 4 * c < W) {
                const float4* sp = reinterpret_cast<const float4*>(stage + row * SROW + 12 * c);
                const float4 v0 = sp[0], v1 = sp[1], v2 = sp[2];
                float px[4][3] = {{v0.x, v0.y, v0.z}, {v0.w, v1.x, v1.y}, {v1.z, v1.w, v2.x}, {v2.y, v2.z, v2.w}};
#pragma unroll
                for (int i = 0; i < 4; i++) colour_tail(p.tail, p.ccm.m, px[i][0], px[i][1], px[i][2]);
                float4* dp = reinterpret_cast<float4*>(tile + (mul24((unsigned)row, rowbytes) + 48u * (unsigned)c));
                dp[0] = make_float4(px[0][0], px[0][1], px[0][2], px[1][0]);
                dp[1] = make_float4(px[1][1], px[1][2], px[2][0], px[2][1]);
                dp[2] = make_float4(px[2][2], px[3][0], px[3][1], px[3][2]);
            }
        }
        return;
    }
    if (!second) return;
    // (row oy >= 2 here: the offset from the origin of region row 2, column 0 is non-negative; the two halo pixels left of the image's
    // first column are addressed but never stored)
    float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(p.out + ((size_t)ty0 * W + (tx0 - 2)) * 3) + (mul24((unsigned)(oy - 2), rowbytes) + 12u * (unsigned)ox));
    // 16-byte stores where the image allows: the run's 96 bytes start 8 bytes past a 16-byte boundary -> 8 + 5 x 16 + 8 bytes, each store
    // issued as soon as its floats exist (pixel by pixel: the float64 tails of eight pixels do not pile up in registers); edge runs
    // leave out their halo pixels
    const bool vst = p.vec && x0 + 8 <= W;
    float o[24];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        // every pixel of the run goes through the tail, halo pixels of edge runs too (their lanes would idle otherwise, and an
        // unconditionally defined o[] spares the register copies a conditional one costs); only the stores are predicated
        float r = keep_r[q], b = keep_b[q], g = (((ma[q] + mb[q]) + r) + b) / 2.0f;
        colour_tail(p.tail, p.ccm.m, r, g, b);
        o[3 * q] = r; o[3 * q + 1] = g; o[3 * q + 2] = b;
        if (vst) {
            if (q == 0 && qlo == 0) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
            if (q == 1 && qlo == 0) *reinterpret_cast<float4*>(dst + 2) = make_float4(o[2], o[3], o[4], o[5]);
            if (q == 3) *reinterpret_cast<float4*>(dst + 6) = make_float4(o[6], o[7], o[8], o[9]);
            if (q == 4) *reinterpret_cast<float4*>(dst + 10) = make_float4(o[10], o[11], o[12], o[13]);
            if (q == 5) *reinterpret_cast<float4*>(dst + 14) = make_float4(o[14], o[15], o[16], o[17]);
            if (q == 7 && qhi == 8) {
                *reinterpret_cast<float4*>(dst + 18) = make_float4(o[18], o[19], o[20], o[21]);
                *reinterpret_cast<float2*>(dst + 22) = make_float2(o[22], o[23]);
            }
        } else if (q >= qlo && q < qhi) {
            dst[3 * q] = r; dst[3 * q + 1] = g; dst[3 * q + 2] = b;
        }
    }
}

__global__ void __launch_bounds__(NT_B, MED_MIN_WAVES) k_ahd_median_stage(MedParams p) {
    __shared__ __attribute__((aligned(16))) float lds[MED_LDS_FLOATS];
    int tbx, tby;
    xcd_tile(tbx, tby);
    ahd_median_tile(p, tbx, tby, lds);
}

// ================================================================================================
// Role-interleaved kernel (round 4): ONE grid whose workgroups are select tiles of frame i + 1 and median tiles of frame i.
// Why: the two kernels of the AHD path stress different halves of a SIMD.  k_ahd_select is float32 multiply / add work with integer multiplies in its Lab
// interpolation, k_ahd_median_stage is 80 % v_min / v_max / v_med3; profiles/r4_ubench_pairs.log: a med3 next to an fma costs 5.0 cycles per pair instead of
// 2.4 + 4.2 -- min / max / med3 (and compares, conversions, shifts) overlap with full-rate float32 issue, as long as both kinds are resident on the SIMD at
// the same time.  Back to back, the two kernels never are (round 3's two-stream run could not guarantee it either: six select workgroups fill a CU's LDS).
// Here the role of a workgroup follows from its index, so every CU holds a fixed mix at all times: of the workgroups an XCD receives (every eighth of the
// grid), the j-th is a median tile iff floor((j + 1) n_med / n) > floor(j n_med / n) (n = the XCD's select + median tiles): evenly spread, both tile
// sequences in XCD-contiguous order as in xcd_tile.  Registers and LDS are the maximum of the two roles (the median stage's): five workgroups per CU.
// Frames are independent, so a batch of n frames takes n + 1 launches: select(0); fused(select(i + 1), median(i)) ...; median(n - 1).
struct FusedPlan { unsigned sel_gx, sel_n, med_gx, med_n; };
template <bool U16, bool HDR>
__global__ void __launch_bounds__(NT_A, 1) k_ahd_fused(AhdParams a, MedParams m, FusedPlan pl) {
    static_assert(NT_A == NT_B, "one workgroup size for both roles");
    constexpr int NF = SelLds<1>::N > MED_LDS_FLOATS ? SelLds<1>::N : MED_LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[NF];
    __shared__ int s_nonfinite[3];
    const unsigned lin = blockIdx.x, xcd = lin & 7u, j = lin >> 3;
    const unsigned qs = pl.sel_n >> 3, rs = pl.sel_n & 7u, ns = qs + (xcd < rs ? 1u : 0u), s0 = xcd * qs + (xcd < rs ? xcd : rs);
    const unsigned qm = pl.med_n >> 3, rm = pl.med_n & 7u, nm = qm + (xcd < rm ? 1u : 0u), m0 = xcd * qm + (xcd < rm ? xcd : rm);
    const unsigned n = ns + nm;
    if (j >= n) return;
    const unsigned before = j * nm / n, upto = (j + 1u) * nm / n;      // median tiles among the XCD's first j / first j + 1 workgroups (uniform: scalar unit)
    if (upto > before) {
        const unsigned t = m0 + before, by = t / pl.med_gx;
        ahd_median_tile(m, (int)(t - by * pl.med_gx), (int)by, smem);
    } else {
        const unsigned t = s0 + (j - before), by = t / pl.sel_gx;
        ahd_select_tile<false, U16, HDR, 1, false>(a, (int)(t - by * pl.sel_gx), (int)by, smem, nullptr, s_nonfinite);
    }
}

// ------------------------------------------------------------------------------------------------
// Chunk queues of the streaming select kernel for one frame size.  XCD x owns a contiguous range of column tiles; its columns are cut, top to bottom, into
// chunks of 14 + 16 m quad rows (a head pass and m chained passes, m <= 7); m follows the work that is left in the queue (guided self-scheduling: the rows left
// divided by twice the queue's workgroups), so the first chunks are long (15.7 output rows per pass) and the last ones single passes that fill the grid's drain.
// The schedule itself: pure host arithmetic (no GPU), exported for the CPU tests as pysp_ahd_stream_chunks.  chunks[0..3] = header (first[8], count[8]), then
// { column tile, S, E, passes } per chunk; returns the number of passes of the launch.
unsigned ahd_stream_chunks(int H, int W, int slots_per_xcd, std::vector<int4>& chunks, unsigned first[8], unsigned count[8]) {
    const int h = H / 2, w = W / 2, ncols = (w + TQX - 1) / TQX;
    chunks.assign(4, make_int4(0, 0, 0, 0));                        // header: first[8], count[8]
    unsigned passes = 0;
    // experiment switches (tools/ab_stream.sh): longest chunk 14 + 16 m rows, divisor of the guided schedule
    static const int env_maxm = [] { const char* e = getenv("PYSP_STREAM_MAXM"); return e ? atoi(e) : 7; }();
    static const int env_div = [] { const char* e = getenv("PYSP_STREAM_DIV"); return e && atoi(e) > 0 ? atoi(e) : 2; }();
    const long long slots = slots_per_xcd > 0 ? slots_per_xcd : 1;
    for (int x = 0; x < 8; x++) {
        const int c0 = (int)((long long)ncols * x / 8), c1 = (int)((long long)ncols * (x + 1) / 8);
        first[x] = (unsigned)chunks.size() - 4u;
        long long left = (long long)(c1 - c0) * h;
        for (int c = c0; c < c1; c++) {
            int r = 0;
            while (r < h) {
                long long want = left / (env_div * slots);
                int m = want <= TQY ? 0 : (int)((want - TQY) / LQY);
                if (m > env_maxm) m = env_maxm;
                int len = TQY + LQY * m;
                if (r + len > h || h - (r + len) < 3) len = h - r;               // (no sliver of one or two rows at the bottom of a column)
                int np = 1;
                for (int q0 = r - 1; q0 + LQY - 1 <= r + len - 1; q0 += LQY) np++;   // the kernel's own loop
                chunks.push_back(make_int4(c, r - 1, r + len - 1, np));
                passes += (unsigned)np;
                r += len; left -= len;
            }
        }
        count[x] = (unsigned)chunks.size() - 4u - first[x];
    }
    for (int x = 0; x < 8; x++) { reinterpret_cast<unsigned*>(chunks.data())[x] = first[x]; reinterpret_cast<unsigned*>(chunks.data())[8 + x] = count[x]; }
    return passes;
}
int ahd_stream_plan_build(AhdStreamPlan& plan, int H, int W, hipStream_t st) {
    if (plan.H == H && plan.W == W && plan.d_chunks) return 0;
    static int cus = 0, occ = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_ahd_select_stream<false, false>, NT_A, 0) != hipSuccess || occ <= 0) occ = 6;
        (void)hipGetLastError();
    }
    unsigned slots = ((unsigned)(cus * occ) & ~7u) / 8;             // resident workgroups per XCD queue: what the guided schedule divides by
    if (slots < 1) slots = 1;
    std::vector<int4> chunks;
    const unsigned passes = ahd_stream_chunks(H, W, (int)slots, chunks, plan.first, plan.count);
    // (an earlier launch may still read the old table)
    if (hipStreamSynchronize(st) != hipSuccess) return -3;
    if (plan.d_chunks) { (void)hipFree(plan.d_chunks); plan.d_chunks = nullptr; }
    if (hipMalloc(&plan.d_chunks, chunks.size() * sizeof(int4)) != hipSuccess) return -3;
    if (hipMemcpy(plan.d_chunks, chunks.data(), chunks.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) return -3;
    unsigned longest = 0;
    for (int x = 0; x < 8; x++) longest = plan.count[x] > longest ? plan.count[x] : longest;
    plan.grid = 8u * longest;                                       // workgroup 8 j + x takes chunk j of queue x (or leaves at once if the queue is shorter)
    plan.passes_total = passes; plan.n_chunks = (unsigned)chunks.size() - 4u;
    plan.H = H; plan.W = W;
    return 0;
}
void ahd_stream_plan_free(AhdStreamPlan& plan) {
    if (plan.d_chunks) { (void)hipFree(plan.d_chunks); plan.d_chunks = nullptr; }
    plan.H = plan.W = 0;
}
bool ahd_stream_ok(int H, int W, int hdr, const void* d_lablut, int lab_planes) {
    return d_lablut != nullptr && !lab_planes && !hdr && H / 2 >= 4 && W / 2 >= 4;
}

int launch_ahd(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int hdr, int stages,
               int tail, float* d_out, float* d_tmp0, float* d_tmp1, const float* d_labtab, const void* d_lablut, Timeline* tl, int lab_planes,
               unsigned* d_float_form_tiles, const AhdStreamPlan* stream_plan) {
    AhdParams a;
    a.float_form_tiles = d_float_form_tiles;
    a.labtab = reinterpret_cast<const float4*>(d_labtab);
    a.lablut = reinterpret_cast<const uint4*>(d_lablut);
    a.src = src; a.H = H; a.W = W; a.hdr = hdr;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M[i];
    if (stages < 0) stages = 0;
    // ping-pong so that the last kernel writes d_out
    float* bufs[2] = {d_tmp0, d_tmp1};
    a.out = stages == 0 ? d_out : bufs[0];
    a.tail = stages == 0 ? tail : 0;
    dim3 ga((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY);
    if (tl) tl->begin(st, "k_ahd_select");
    const bool tiny = H / 2 < 4 || W / 2 < 4, u16 = src.u16 != nullptr;
#define AHD_LAUNCH2(HDRV, LABV, TAILV) \
    do { \
        if (tiny && u16) hipLaunchKernelGGL((k_ahd_select<true, true, HDRV, LABV, TAILV>), ga, dim3(NT_A), 0, st, a); \
        else if (tiny) hipLaunchKernelGGL((k_ahd_select<true, false, HDRV, LABV, TAILV>), ga, dim3(NT_A), 0, st, a); \
        else if (u16) hipLaunchKernelGGL((k_ahd_select<false, true, HDRV, LABV, TAILV>), ga, dim3(NT_A), 0, st, a); \
        else hipLaunchKernelGGL((k_ahd_select<false, false, HDRV, LABV, TAILV>), ga, dim3(NT_A), 0, st, a); \
    } while (0)
#define AHD_LAUNCH(HDRV, LABV) do { if (a.tail != 0) AHD_LAUNCH2(HDRV, LABV, true); else AHD_LAUNCH2(HDRV, LABV, false); } while (0)
    if (stream_plan && stream_plan->H == H && stream_plan->W == W && ahd_stream_ok(H, W, hdr, d_lablut, lab_planes)) {
        AhdStreamQueues q;
        q.chunks = reinterpret_cast<const int4*>(stream_plan->d_chunks); q.passes_total = stream_plan->passes_total;
        const dim3 gs(stream_plan->grid);
        if (u16) { if (a.tail != 0) hipLaunchKernelGGL((k_ahd_select_stream<true, true>), gs, dim3(NT_A), 0, st, a, q); else hipLaunchKernelGGL((k_ahd_select_stream<true, false>), gs, dim3(NT_A), 0, st, a, q); }
        else { if (a.tail != 0) hipLaunchKernelGGL((k_ahd_select_stream<false, true>), gs, dim3(NT_A), 0, st, a, q); else hipLaunchKernelGGL((k_ahd_select_stream<false, false>), gs, dim3(NT_A), 0, st, a, q); }
    }
    else if (d_lablut && lab_planes) { if (hdr) AHD_LAUNCH(true, 2); else AHD_LAUNCH(false, 2); }
    else if (d_lablut) { if (hdr) AHD_LAUNCH(true, 1); else AHD_LAUNCH(false, 1); }
    else { if (hdr) AHD_LAUNCH(true, 0); else AHD_LAUNCH(false, 0); }
#undef AHD_LAUNCH2
#undef AHD_LAUNCH
    if (tl) tl->end(st);
    const float* cur = a.out;
    dim3 gb((W + BTX - 1) / BTX, (H + BTY - 1) / BTY);
    for (int s = 0; s < stages; s++) {
        MedParams m;
        m.in = cur; m.H = H; m.W = W; m.ccm = a.ccm;
        bool last = s == stages - 1;
        m.out = last ? d_out : bufs[(s + 1) & 1];
        m.tail = last ? tail : 0;
        m.vec = !(W & 3) && !((reinterpret_cast<uintptr_t>(m.in) | reinterpret_cast<uintptr_t>(m.out)) & 15);
        if (tl) tl->begin(st, "k_ahd_median_stage");
        hipLaunchKernelGGL(k_ahd_median_stage, gb, dim3(NT_B), 0, st, m);
        if (tl) tl->end(st);
        cur = m.out;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// A batch of n frames through AHD with ONE median stage, Lab mode 1, frames of at least 8 x 8 px: n + 1 launches instead of 2 n -- select(frame 0), then
// n - 1 role-interleaved launches (select tiles of frame i + 1 and median tiles of frame i in one grid, k_ahd_fused), then median(frame n - 1).  The
// intermediate RGB images alternate between d_tmp0 and d_tmp1.  Same kernels' code, same results as n calls of launch_ahd.
int ahd_select_tiles(int H, int W) { return ((W / 2 + TQX - 1) / TQX) * ((H / 2 + TQY - 1) / TQY); }
bool ahd_pipelined_ok(int n, int H, int W, int stages, const void* d_lablut) { return n >= 2 && stages == 1 && d_lablut != nullptr && H / 2 >= 4 && W / 2 >= 4; }
int launch_ahd_pipelined(hipStream_t st, const MosaicSrc* srcs, int n, int H, int W, const float wb[3], const double M[9], int hdr, int tail,
                         float* const* d_outs, float* d_tmp0, float* d_tmp1, const void* d_lablut, Timeline* tl) {
    if (!ahd_pipelined_ok(n, H, W, 1, d_lablut)) return -3;
    AhdParams a;
    a.labtab = nullptr;
    a.float_form_tiles = nullptr;
    a.lablut = reinterpret_cast<const uint4*>(d_lablut);
    a.H = H; a.W = W; a.hdr = hdr; a.tail = 0;
    for (int i = 0; i < 3; i++) a.wb[i] = wb[i];
    for (int i = 0; i < 9; i++) a.ccm.m[i] = M[i];
    MedParams m;
    m.H = H; m.W = W; m.ccm = a.ccm; m.tail = tail;
    float* bufs[2] = {d_tmp0, d_tmp1};
    const dim3 ga((W / 2 + TQX - 1) / TQX, (H / 2 + TQY - 1) / TQY), gb((W + BTX - 1) / BTX, (H + BTY - 1) / BTY);
    FusedPlan pl{ga.x, ga.x * ga.y, gb.x, gb.x * gb.y};
    // the largest share of any XCD decides the grid: XCD 0 owns ceil(n / 8) of each sequence
    const unsigned per_xcd = (pl.sel_n + 7) / 8 + (pl.med_n + 7) / 8;
    for (int i = 0; i <= n; i++) {
        const bool sel = i < n, med = i > 0;
        bool u16 = false;
        if (sel) { a.src = srcs[i]; a.out = bufs[i & 1]; u16 = srcs[i].u16 != nullptr; }
        if (med) {
            m.in = bufs[(i - 1) & 1]; m.out = d_outs[i - 1];
            m.vec = !(W & 3) && !((reinterpret_cast<uintptr_t>(m.in) | reinterpret_cast<uintptr_t>(m.out)) & 15);
        }
        if (sel && med) {
            if (tl) tl->begin(st, "k_ahd_fused");
            if (u16) { if (hdr) hipLaunchKernelGGL((k_ahd_fused<true, true>), dim3(8 * per_xcd), dim3(NT_A), 0, st, a, m, pl); else hipLaunchKernelGGL((k_ahd_fused<true, false>), dim3(8 * per_xcd), dim3(NT_A), 0, st, a, m, pl); }
            else { if (hdr) hipLaunchKernelGGL((k_ahd_fused<false, true>), dim3(8 * per_xcd), dim3(NT_A), 0, st, a, m, pl); else hipLaunchKernelGGL((k_ahd_fused<false, false>), dim3(8 * per_xcd), dim3(NT_A), 0, st, a, m, pl); }
            if (tl) tl->end(st);
        } else if (sel) {
            if (tl) tl->begin(st, "k_ahd_select");
            if (u16) { if (hdr) hipLaunchKernelGGL((k_ahd_select<false, true, true, 1, false>), ga, dim3(NT_A), 0, st, a); else hipLaunchKernelGGL((k_ahd_select<false, true, false, 1, false>), ga, dim3(NT_A), 0, st, a); }
            else { if (hdr) hipLaunchKernelGGL((k_ahd_select<false, false, true, 1, false>), ga, dim3(NT_A), 0, st, a); else hipLaunchKernelGGL((k_ahd_select<false, false, false, 1, false>), ga, dim3(NT_A), 0, st, a); }
            if (tl) tl->end(st);
        } else {
            if (tl) tl->begin(st, "k_ahd_median_stage");
            hipLaunchKernelGGL(k_ahd_median_stage, gb, dim3(NT_B), 0, st, m);
            if (tl) tl->end(st);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

#ifdef AHD_STAMPS
// diagnostic build: the stamp array of the last k_ahd_select launches (AHD_NSTAMP values per wave, zero where never written)
extern "C" int pysp_debug_ahd_stamps(unsigned long long* out, size_t n_values, int clear) {
    const size_t total = (size_t)AHD_NSTAMP * AHD_STAMP_WAVES;
    if (n_values > total) n_values = total;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ahd_stamps), n_values * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ahd_stamps)) != hipSuccess || hipMemset(p, 0, total * sizeof(unsigned long long)) != hipSuccess) return -3;
    }
    return AHD_NSTAMP;
}
#endif
