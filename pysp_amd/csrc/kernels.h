// kernels.h -- host-side launchers of the gfx950 kernels (one translation unit per family).
// All launchers enqueue on `st` and return 0 or PYSP_EHIP (-3); pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

struct MosaicSrc;   // demosaic_common.h: float32 mosaic, or raw uint16 + black/saturation levels
MosaicSrc mosaic_f32(const float* d_bayer);
MosaicSrc mosaic_u16(const uint16_t* d_bayer, const float black[4], const float sat[4]);

// Optional per-kernel timing: when `on`, every launcher brackets each kernel it enqueues with events.
struct Timeline {
    static constexpr int MAXK = 8;
    hipEvent_t ev[2 * MAXK];
    const char* name[MAXK];
    int n = 0;
    bool on = false;
    void begin(hipStream_t st, const char* nm) { if (on && n < MAXK) { name[n] = nm; (void)hipEventRecord(ev[2 * n], st); } }
    void end(hipStream_t st) { if (on && n < MAXK) { (void)hipEventRecord(ev[2 * n + 1], st); n++; } }
};

// k_basic.hip
int launch_demux_f32(hipStream_t st, const float* bayer, int H, int W, float* r, float* g1, float* b, float* g2);
int launch_demux_u16(hipStream_t st, const uint16_t* bayer, int H, int W, float* r, float* g1, float* b, float* g2);
int launch_remux_f32(hipStream_t st, const float* r, const float* g1, const float* b, const float* g2, int h, int w, float* bayer);
int launch_normalize_u16(hipStream_t st, const uint16_t* bayer, int H, int W, const float black[4], const float sat[4], float* out);
int launch_build_map(hipStream_t st, const float* lab, int Hp, int Wp, int k_pad, int is_vertical, float* out);
int launch_hot_threshold(hipStream_t st, const float* bayer, int H, int W, float min_delta, int min_count, uint8_t* mr, uint8_t* mg1, uint8_t* mb, uint8_t* mg2);
int launch_flat_field(hipStream_t st, const float* bayer, const float* flat, int H, int W, const float mean[4], int clamp_high, float* out, unsigned* d_stats8);
int launch_cam_to_rgb(hipStream_t st, const float* in, size_t npx, const double M[9], int clip, float* out);
int launch_gamma(hipStream_t st, const float* in, size_t n, int decode, float* out);
int launch_wb_scale(hipStream_t st, const float* in, size_t npx, const float coeff[3], int undo, float* out);
int launch_colour_tail(hipStream_t st, const float* in, size_t npx, const double M[9], int tail, float* out);
int launch_copy16(hipStream_t st, void* dst, const void* src, size_t bytes);   // 16-byte aligned device -> mapped host (or device) copy by a kernel

// Streaming form of the select kernel (round 5, k_ahd.hip): the chunk queues of one frame size, device resident; api.cpp keeps a few per context.
struct AhdStreamPlan {
    int H = 0, W = 0;
    void* d_chunks = nullptr;         // int4 { column tile, head pass origin, last output quad row, passes } per chunk
    unsigned first[8] = {}, count[8] = {}, passes_total = 0, n_chunks = 0, grid = 0;
};
int ahd_stream_plan_build(AhdStreamPlan& plan, int H, int W, hipStream_t st);   // builds and uploads unless the plan already is for (H, W); may synchronise st
// the schedule alone (host arithmetic, no GPU): chunks[0..3] = header, then { column tile, S, E, passes } per chunk; returns the passes of the launch
unsigned ahd_stream_chunks(int H, int W, int slots_per_xcd, std::vector<int4>& chunks, unsigned first[8], unsigned count[8]);
void ahd_stream_plan_free(AhdStreamPlan& plan);
bool ahd_stream_ok(int H, int W, int hdr, const void* d_lablut, int lab_planes);
// k_ahd.hip: tail = colour tail of devmath.h (0 none, 1 lin sRGB, 2 sRGB, 3 Reinhard + sRGB);
// d_tmp0/d_tmp1 are (H,W,3) scratch images (only needed when stages >= 1 / >= 2).
int launch_ahd(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int hdr, int stages,
               int tail, float* d_out, float* d_tmp0, float* d_tmp1, const float* d_labtab, const void* d_lablut /* Lab mode 1, else NULL */,
               Timeline* tl = nullptr, int lab_planes = 0 /* Lab mode 1: float Lab planes and float votes (round 3's form) instead of packed cells */,
               unsigned* d_float_form_tiles = nullptr /* packed form: device counter, +1 per tile in which a wave redid its votes in float arithmetic */,
               const AhdStreamPlan* stream_plan = nullptr /* non-NULL and eligible (ahd_stream_ok): the streaming form of the select kernel */);
int ahd_select_tiles(int H, int W);   // workgroups of one k_ahd_select launch

// A batch of n frames through AHD with one median stage (Lab mode 1): n + 1 launches, select tiles of frame i + 1 and median tiles of frame i sharing one grid
// (role-interleaved kernel, k_ahd.hip).  ahd_pipelined_ok says whether the batch qualifies; the caller falls back to n calls of launch_ahd otherwise.
bool ahd_pipelined_ok(int n, int H, int W, int stages, const void* d_lablut);
int launch_ahd_pipelined(hipStream_t st, const MosaicSrc* srcs, int n, int H, int W, const float wb[3], const double M[9], int hdr, int tail,
                         float* const* d_outs, float* d_tmp0, float* d_tmp1, const void* d_lablut, Timeline* tl = nullptr);

// k_eag.hip
int launch_eag(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int tail, float* d_out, Timeline* tl = nullptr);
int launch_draft(hipStream_t st, const MosaicSrc& src, int H, int W, const float wb[3], const double M[9], int tail, float* d_out, Timeline* tl = nullptr);
// n frames of one size in ONE grid per 16 frames (blockIdx.z = frame): float32 mosaics (u16 = 0) or raw uint16 ones with their black / saturation levels; frames of at
// least 8 x 8 px (the caller falls back to n single launches otherwise); returns -1 on a bad argument
int launch_eag_batch(hipStream_t st, const void* const* d_srcs, int u16, const float black[4], const float sat[4], int n, int H, int W, const float wb[3], const double M[9],
                     int tail, float* const* d_outs, Timeline* tl = nullptr);
int launch_draft_batch(hipStream_t st, const void* const* d_srcs, int u16, const float black[4], const float sat[4], int n, int H, int W, const float wb[3], const double M[9],
                       int tail, float* const* d_outs, Timeline* tl = nullptr);

int launch_resample_g(hipStream_t st, const float* g1, const float* g2, int h, int w, int weighted, float* out);
int launch_highpass(hipStream_t st, const float* g, int H, int W, float* out);
int launch_resample_channel(hipStream_t st, const float* sub, const float* g_sub, const float* g_hf, int h, int w, int pos, float* out);

// k_misc.hip
// Any number of exposures: more than fuse_max_exposures_per_pass() run as passes in order, with partial sums carried in d_part -- (H,W) floats for the raw
// fusion, 6 npx floats for the RGB one (may be NULL when K fits one pass).  Same additions in the same order: same bits as one pass.
int fuse_max_exposures_per_pass();
int launch_fuse_raw_pass(hipStream_t st, const float* const* d_frames, int n, int H, int W, const float* ev_off, const float* bias, int first, int last,
                         const float* d_kmax_frame, float kmax_off, float* d_out, int32_t* d_count, float* d_part);
int launch_fuse_rgb_pass(hipStream_t st, const float* const* d_frames, float* const* d_frames_out, int n, size_t npx, const float* coeff, const int* applied,
                         const float* ev_off, const float* bias, int first, int last, int kmax_local, float kmax_off, const double* M, float* d_out,
                         int32_t* d_count, float* d_part);
int launch_fuse_raw(hipStream_t st, const float* const* d_frames_host_array, int K, int H, int W, const float* ev_off,
                    const float* bias, int kmax, float* d_out, int32_t* d_count, float* d_part = nullptr);
int launch_fuse_rgb(hipStream_t st, const float* const* d_frames, float* const* d_frames_out, int K, size_t npx, const float* coeff,
                    const int* applied, const float* ev_off, const float* bias, int kmax, const double* M, float* d_out, int32_t* d_count, float* d_part = nullptr);
int launch_warp_table(hipStream_t st, float kr0, float kr1, float kr2, float kr3, float kt0, float kt1, int width, int height,
                      float cxn, float cyn, float scale, const float* d_seed, float* d_table);
int launch_warp_remap(hipStream_t st, const float* d_in, float* d_out, int H, int W, const double* coeffs, int planes, double cxn,
                      double cyn, float scale, const float* d_lanczos_tab, int row0, int row1);
int launch_ca_green(hipStream_t st, const float* bayer, int H, int W, float* out);
int launch_ca_upsample_fused(hipStream_t st, const float* bayer, const float* g_full, int H, int W, const float* d_quad, int pos, float wb, float* out);
int launch_ca_remap_sites(hipStream_t st, const float* src, int H, int W, const float* d_quad, int oy, int ox, float wb, float* bayer);
int launch_warp_src_rows(hipStream_t st, int H, int W, const double* coeffs, int planes, double cxn, double cyn, float scale, int row0, int row1,
                         int* d_rows);
int launch_remap_table(hipStream_t st, const float* src, int sstride, const float* mapx, const float* mapy, int mstride, const float* d_tab,
                       int H, int W, int do_clip, float* dst, int dstride);
void host_lanczos4_table(float tab[256]);
