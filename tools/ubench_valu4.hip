// Micro-benchmark 4 (round 4): WHICH half-rate classes overlap with full-rate float32 work on a gfx950 SIMD, and what LDS / scalar instructions inside a VALU
// stream cost.  ubench_valu3 established the single-opcode rates (full 2.3, half 4.15 cycles per wave64 instruction) and that `v_fma_f32, v_med3_f32`
// alternating costs 4.9 per pair instead of 6.45 -- but k_ahd_select's half-rate instructions are mostly NOT min/max/med3: they are compares, v_addc, v_dot2,
// 24-bit multiplies, bit-field extracts, conversions and float64.  DESIGN.md 7.0 (a) applied the fma/med3 overlap to all of them and predicted 2.35 cycles per
// instruction where the kernel runs at 3.2-3.5.  This program measures the pair costs class by class, the 3:1 mixes, the vote cell's own instruction pattern,
// LDS reads and writes between VALU instructions, and VGPR bank conflicts of three-source instructions.
//
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu4.bin ubench_valu4.hip && ./ubench_valu4.bin > profiles/r4_ubench_pairs.log
//
// Same harness as ubench_valu3: grid = 256 CUs x k workgroups of 256 threads (one wave per SIMD each), k = waves per SIMD; cycles from s_memtime /
// s_memrealtime inside the kernel; reported: SIMD cycles per PATTERN (not per instruction) over the span first wave start - last wave end.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#define ITERS 256
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define REP8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
// operands: %0-%7 chains, %8 = a, %9 = b, %10 = 64-bit sgpr mask, %11 = LDS byte address (per lane), %12 = second LDS address
#define BODY(I) asm volatile(REP8(I) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                             : "v"(a), "v"(b), "s"(sm), "v"(lds0), "v"(lds1) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "v100", "v101", "v102", "v103", "memory")

#define F "v_fma_f32 "
#define FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define MUL(r) "v_mul_f32 " r ", " r ", %8\n"
#define ADD(r) "v_add_f32 " r ", " r ", %9\n"
#define SUB(r) "v_sub_f32 " r ", " r ", %8\n"
#define MED3(r) "v_med3_f32 " r ", " r ", %8, %9\n"
#define MAXF(r) "v_max_f32 " r ", " r ", %8\n"
#define CMP(r) "v_cmp_le_f32 vcc, " r ", %8\n"
#define CMPS(r) "v_cmp_le_f32_e64 s[20:21], " r ", %8\n"
#define CMPS2(r) "v_cmp_le_f32_e64 s[22:23], " r ", %9\n"
#define CND(r) "v_cndmask_b32 " r ", %9, " r ", vcc\n"
#define CVTFI(r) "v_cvt_f32_i32 " r ", " r "\n"
#define CVTD(r) "v_cvt_f64_f32 v[100:101], " r "\n"
#define DOT2(r) "v_dot2_i32_i16 " r ", %8, %9, " r "\n"
#define MUL24(r) "v_mul_u32_u24 " r ", " r ", %8\n"
#define MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define BFE(r) "v_bfe_u32 " r ", " r ", 5, 4\n"
#define ADDC(r) "v_addc_co_u32_e64 " r ", s[24:25], " r ", 0, %10\n"
#define ADDCS(r) "v_addc_co_u32_e64 " r ", s[24:25], " r ", 0, s[20:21]\n"
#define LSHL(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define LSHR(r) "v_lshrrev_b32 " r ", 1, " r "\n"
#define PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define PKSUB(r) "v_pk_sub_i16 " r ", " r ", %8\n"
#define ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define ANDB(r) "v_and_b32 " r ", " r ", %8\n"
#define ADDU(r) "v_add_u32 " r ", " r ", %8\n"
#define SUBU(r) "v_sub_u32 " r ", " r ", %8\n"
#define FMA64(r) "v_fma_f64 v[100:101], v[100:101], v[100:101], v[100:101]\n"
#define SAND "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define DSR64(r) "ds_read_b64 v[100:101], %11\n"
#define DSR32(r) "ds_read_b32 v100, %11\n"
#define DSR128(r) "ds_read_b128 v[100:103], %12\n"
#define DSW32(r) "ds_write_b32 %11, " r "\n"
#define DSW64(r) "ds_write_b64 %12, v[100:101]\n"
#define SNOP(r) "s_nop 0\n"

// pattern table: name, number of VALU instructions per pattern element (per chain), text
#define P_FMA(r) FMA(r)
#define P_FMA_MED3(r) FMA(r) MED3(r)
#define P_FMA_MAX(r) FMA(r) MAXF(r)
#define P_FMA_CMP(r) FMA(r) CMP(r)
#define P_FMA_CMPS(r) FMA(r) CMPS(r)
#define P_FMA_CMPCND(r) FMA(r) CMP(r) FMA(r) CND(r)
#define P_FMA_CVT(r) FMA(r) CVTFI(r)
#define P_FMA_DOT2(r) FMA(r) DOT2(r)
#define P_FMA_MUL24(r) FMA(r) MUL24(r)
#define P_FMA_MAD24(r) FMA(r) MAD24(r)
#define P_FMA_BFE(r) FMA(r) BFE(r)
#define P_FMA_ADDC(r) FMA(r) ADDC(r)
#define P_FMA_LSHL(r) FMA(r) LSHL(r)
#define P_FMA_PERM(r) FMA(r) PERM(r)
#define P_FMA_PKSUB(r) FMA(r) PKSUB(r)
#define P_FMA_ADD3(r) FMA(r) ADD3(r)
#define P_FMA_LSHLOR(r) FMA(r) LSHLOR(r)
#define P_FMA_FMA64(r) FMA(r) FMA64(r)
#define P_FMA_CVTD(r) FMA(r) CVTD(r)
#define P_3F_DOT2(r) FMA(r) MUL(r) ADD(r) DOT2(r)
#define P_3F_ADDC(r) FMA(r) MUL(r) ADD(r) ADDC(r)
#define P_3F_CVT(r) FMA(r) MUL(r) ADD(r) CVTFI(r)
#define P_3F_CMP(r) FMA(r) MUL(r) ADD(r) CMP(r)
#define P_3F_BFE(r) FMA(r) MUL(r) ADD(r) BFE(r)
#define P_3F_FMA64(r) FMA(r) MUL(r) ADD(r) FMA64(r)
#define P_ADDU_DOT2(r) ADDU(r) DOT2(r)
#define P_ADDU_MED3(r) ADDU(r) MED3(r)
#define P_AND_BFE(r) ANDB(r) BFE(r)
#define P_MED3_DOT2(r) MED3(r) DOT2(r)
#define P_MED3_FMA64(r) MED3(r) FMA64(r)
// the fast vote's cell (k_ahd.hip vote_quad): L difference, two chroma differences, two squares, their sum (6 full rate), two compares into SGPR pairs,
// the scalar AND, v_addc with that carry (3 half rate + 1 scalar)
#define P_VOTE_F32(r) SUB(r) SUB(r) SUB(r) MUL(r) MUL(r) ADD(r) CMPS(r) CMPS2(r) SAND ADDCS(r)
// the same cell with packed int16 chroma: L difference (full), v_pk_sub_i16 + v_dot2 (half), two compares, AND, addc
#define P_VOTE_I16(r) SUB(r) PKSUB(r) DOT2(r) CMPS(r) CMPS2(r) SAND ADDCS(r)
#define P_VOTE_I16_NOSAND(r) SUB(r) PKSUB(r) DOT2(r) CMPS(r) CMPS2(r) ADDCS(r)
// LDS traffic inside a full-rate stream
#define P_DSR64(r) DSR64(r)
#define P_4F_DSR64(r) FMA(r) MUL(r) ADD(r) SUB(r) DSR64(r)
#define P_8F_DSR64(r) FMA(r) MUL(r) ADD(r) SUB(r) FMA(r) MUL(r) ADD(r) SUB(r) DSR64(r)
#define P_4F_DSR32(r) FMA(r) MUL(r) ADD(r) SUB(r) DSR32(r)
#define P_4F_DSR128(r) FMA(r) MUL(r) ADD(r) SUB(r) DSR128(r)
#define P_4F_DSW32(r) FMA(r) MUL(r) ADD(r) SUB(r) DSW32(r)
#define P_4F_DSW64(r) FMA(r) MUL(r) ADD(r) SUB(r) DSW64(r)
#define P_2M_DSR64(r) MED3(r) MED3(r) DSR64(r)
#define P_4F_SNOP(r) FMA(r) MUL(r) ADD(r) SUB(r) SNOP(r)
#define P_4F_SAND(r) FMA(r) MUL(r) ADD(r) SUB(r) SAND

#define PATTERNS(X) \
    X(P_FMA, 1, "fma") X(P_FMA_MED3, 2, "fma, med3") X(P_FMA_MAX, 2, "fma, max_f32") X(P_FMA_CMP, 2, "fma, cmp (vcc)") X(P_FMA_CMPS, 2, "fma, cmp (sgpr pair)") \
    X(P_FMA_CMPCND, 4, "fma, cmp, fma, cndmask") X(P_FMA_CVT, 2, "fma, cvt_f32_i32") X(P_FMA_DOT2, 2, "fma, dot2_i32_i16") X(P_FMA_MUL24, 2, "fma, mul_u32_u24") \
    X(P_FMA_MAD24, 2, "fma, mad_u32_u24") X(P_FMA_BFE, 2, "fma, bfe_u32") X(P_FMA_ADDC, 2, "fma, addc_co (sgpr carry)") X(P_FMA_LSHL, 2, "fma, lshlrev") \
    X(P_FMA_PERM, 2, "fma, perm_b32") X(P_FMA_PKSUB, 2, "fma, pk_sub_i16") X(P_FMA_ADD3, 2, "fma, add3_u32") X(P_FMA_LSHLOR, 2, "fma, lshl_or") \
    X(P_FMA_FMA64, 2, "fma, fma_f64") X(P_FMA_CVTD, 2, "fma, cvt_f64_f32") \
    X(P_3F_DOT2, 4, "fma, mul, add, dot2") X(P_3F_ADDC, 4, "fma, mul, add, addc") X(P_3F_CVT, 4, "fma, mul, add, cvt") X(P_3F_CMP, 4, "fma, mul, add, cmp") \
    X(P_3F_BFE, 4, "fma, mul, add, bfe") X(P_3F_FMA64, 4, "fma, mul, add, fma_f64") \
    X(P_ADDU_DOT2, 2, "add_u32, dot2") X(P_ADDU_MED3, 2, "add_u32, med3") X(P_AND_BFE, 2, "and_b32, bfe") X(P_MED3_DOT2, 2, "med3, dot2") X(P_MED3_FMA64, 2, "med3, fma_f64") \
    X(P_VOTE_F32, 9, "vote cell, float32 chroma (6 full + 3 half + s_and)") X(P_VOTE_I16, 6, "vote cell, int16 chroma (1 full + 5 half + s_and)") \
    X(P_VOTE_I16_NOSAND, 6, "vote cell, int16 chroma, no s_and") \
    X(P_DSR64, 0, "ds_read_b64 alone") X(P_4F_DSR64, 4, "4 full + ds_read_b64") X(P_8F_DSR64, 8, "8 full + ds_read_b64") X(P_4F_DSR32, 4, "4 full + ds_read_b32") \
    X(P_4F_DSR128, 4, "4 full + ds_read_b128") X(P_4F_DSW32, 4, "4 full + ds_write_b32") X(P_4F_DSW64, 4, "4 full + ds_write_b64") X(P_2M_DSR64, 2, "2 med3 + ds_read_b64") \
    X(P_4F_SNOP, 4, "4 full + s_nop") X(P_4F_SAND, 4, "4 full + s_and_b64")

enum {
#define X(p, n, s) ID_##p,
    PATTERNS(X)
#undef X
    ID_BANK_SAME, ID_BANK_DIFF, ID_BANK_MED_SAME, ID_BANK_MED_DIFF, N_PAT };

template <int ID>
__global__ void __launch_bounds__(256) k(Stamp* st, float* sink, float a, float b) {
    __shared__ float lds[4096];
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long sm = 0x5555555555555555ull;
    lds[threadIdx.x] = x0; lds[threadIdx.x + 256] = x1;
    const unsigned lds0 = (unsigned)(threadIdx.x * 8), lds1 = (unsigned)(8192 + threadIdx.x * 16);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#define X(p, n, s) if (ID == ID_##p) BODY(p);
            PATTERNS(X)
#undef X
            // three-source instructions whose sources sit in ONE VGPR bank (register number mod 4) against three different banks
            if (ID == ID_BANK_SAME)
                asm volatile("v_fma_f32 v104, v108, v112, v116\nv_fma_f32 v105, v109, v113, v117\nv_fma_f32 v106, v110, v114, v118\nv_fma_f32 v107, v111, v115, v119\n"
                             "v_fma_f32 v104, v108, v112, v116\nv_fma_f32 v105, v109, v113, v117\nv_fma_f32 v106, v110, v114, v118\nv_fma_f32 v107, v111, v115, v119\n"
                             ::: "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
            if (ID == ID_BANK_DIFF)
                asm volatile("v_fma_f32 v104, v108, v113, v118\nv_fma_f32 v105, v109, v114, v119\nv_fma_f32 v106, v110, v115, v116\nv_fma_f32 v107, v111, v112, v117\n"
                             "v_fma_f32 v104, v108, v113, v118\nv_fma_f32 v105, v109, v114, v119\nv_fma_f32 v106, v110, v115, v116\nv_fma_f32 v107, v111, v112, v117\n"
                             ::: "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
            if (ID == ID_BANK_MED_SAME)
                asm volatile("v_med3_f32 v104, v108, v112, v116\nv_med3_f32 v105, v109, v113, v117\nv_med3_f32 v106, v110, v114, v118\nv_med3_f32 v107, v111, v115, v119\n"
                             "v_med3_f32 v104, v108, v112, v116\nv_med3_f32 v105, v109, v113, v117\nv_med3_f32 v106, v110, v114, v118\nv_med3_f32 v107, v111, v115, v119\n"
                             ::: "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
            if (ID == ID_BANK_MED_DIFF)
                asm volatile("v_med3_f32 v104, v108, v113, v118\nv_med3_f32 v105, v109, v114, v119\nv_med3_f32 v106, v110, v115, v116\nv_med3_f32 v107, v111, v112, v117\n"
                             "v_med3_f32 v104, v108, v113, v118\nv_med3_f32 v105, v109, v114, v119\nv_med3_f32 v106, v110, v115, v116\nv_med3_f32 v107, v111, v112, v117\n"
                             ::: "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + lds[threadIdx.x ^ 1];
    if (s == 1.2345e-30f) sink[0] = s;
}

struct Row { std::string name; int nvalu; double cyc[4]; double ghz[4]; };

template <int ID>
static void run(Stamp* dst, float* sink, const char* name, int nvalu, std::vector<Row>& rows) {
    const int ks[] = {1, 2, 4, 6};
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    Row row; row.name = name; row.nvalu = nvalu;
    const double pats = (double)ITERS * 4 * 8;          // pattern elements per wave (8 chains x 4 unrolls)
    for (int ki = 0; ki < 4; ki++) {
        const int k_ = ks[ki], blocks = 256 * k_;
        std::vector<Stamp> h(blocks * 4);
        hipLaunchKernelGGL(k<ID>, dim3(blocks), dim3(256), 0, 0, dst, sink, 0.999f, 0.5f);
        hipLaunchKernelGGL(k<ID>, dim3(blocks), dim3(256), 0, 0, dst, sink, 0.999f, 0.5f);
        HC(hipDeviceSynchronize());
        HC(hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> clk;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (auto& s : h) { rmin = std::min(rmin, s.r0); rmax = std::max(rmax, s.r1); clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 100e6); }
        std::sort(clk.begin(), clk.end());
        const double ghz = clk[clk.size() / 2] / 1e9, span = (double)(rmax - rmin) * 1e-8;
        row.cyc[ki] = span * ghz * 1e9 / (k_ * pats); row.ghz[ki] = ghz;
    }
    rows.push_back(row);
    printf("%-58s | %4d | %6.2f | %6.2f | %6.2f | %6.2f | %.2f\n", row.name.c_str(), row.nvalu, row.cyc[0], row.cyc[1], row.cyc[2], row.cyc[3], row.ghz[3]);
    fflush(stdout);
}

int main() {
    Stamp* d; float* sink;
    HC(hipMalloc(&d, 256 * 8 * 4 * sizeof(Stamp))); HC(hipMalloc(&sink, 64));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<ID_P_FMA>, dim3(2048), dim3(256), 0, 0, d, sink, 0.999f, 0.5f);
    HC(hipDeviceSynchronize());
    printf("# MI355X (gfx950): SIMD shader-clock cycles per PATTERN (one pattern = the instructions named, issued on 8 independent register chains), by waves per SIMD.\n");
    printf("# single-opcode rates for comparison (ubench_valu3): full rate 2.3, half rate 4.15, float64 4.2 cycles per wave64 instruction\n");
    printf("%-58s | valu | k=1    | k=2    | k=4    | k=6    | GHz\n", "pattern");
    fflush(stdout);
    std::vector<Row> rows;
#define X(p, n, s) run<ID_##p>(d, sink, s, n, rows);
    PATTERNS(X)
#undef X
    run<ID_BANK_SAME>(d, sink, "fma, 3 sources in ONE vgpr bank (per instr)", 1, rows);
    run<ID_BANK_DIFF>(d, sink, "fma, 3 sources in three banks (per instr)", 1, rows);
    run<ID_BANK_MED_SAME>(d, sink, "med3, 3 sources in ONE vgpr bank (per instr)", 1, rows);
    run<ID_BANK_MED_DIFF>(d, sink, "med3, 3 sources in three banks (per instr)", 1, rows);
    return 0;
}
