/*
 * pysp_hip.h -- C ABI of libpysp_hip.so, the MI355X (gfx950) implementation of bullbin/pySP's
 * debayer -> white balance -> colour matrix -> sRGB hot path.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point names the reference
 * interface (file:line relative to the pySP repository) it stands in for.  INTEGRATION.md shows the
 * ctypes binding a pySP maintainer would add.
 *
 * Conventions
 *   - images are row-major; float32 unless the name says otherwise; H and W are the mosaic
 *     dimensions and must be even, at most 2^20 per side (2x2 CFA, RGGB order: R=(0,0) G1=(0,1) G2=(1,0) B=(1,1));
 *     WarpRectilinear images at most 2^17 = 131072 per side (cv2.remap itself stops at 32767);
 *   - entry points return 0 on success and a negative PYSP_E* code on failure;
 *     pysp_last_error() returns a thread-local description of the last failure;
 *   - "host" entry points borrow caller memory for the duration of the call (outputs are written
 *     into caller-allocated buffers, like the np.zeros outputs of the Cython units); large frames go
 *     through the GPU in horizontal bands so that upload, kernels and download overlap;
 *   - "_dev" entry points take device pointers and only enqueue work on the context's stream;
 *   - a pysp_ctx owns one HIP stream and a grow-only device workspace; it is not thread-safe,
 *     use one context per thread.  There is no CPU fallback: without a GPU, pysp_ctx_create fails.
 */
#ifndef PYSP_HIP_H
#define PYSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYSP_ABI_VERSION 1

#define PYSP_OK 0
#define PYSP_EBADARG (-1)   /* bad shape / odd dimension / null pointer  -> ValueError          */
#define PYSP_ENOTIMPL (-2)  /* unknown quality / CFA pattern             -> NotImplementedError */
#define PYSP_EHIP (-3)      /* HIP runtime error                         -> RuntimeError        */
#define PYSP_ENOMEM (-4)    /* device allocation failed                  -> MemoryError         */

/* pySP const.py:3-6 QualityDemosaic */
#define PYSP_QUALITY_DRAFT 0
#define PYSP_QUALITY_FAST 1
#define PYSP_QUALITY_BEST 2

typedef struct pysp_ctx pysp_ctx;

int pysp_abi_version(void);
const char *pysp_last_error(void);
int pysp_device_count(void);

/* stream == NULL: the context creates (and owns) its own stream; otherwise `stream` is a hipStream_t
 * borrowed from the caller (e.g. torch.cuda.current_stream().cuda_stream). */
pysp_ctx *pysp_ctx_create(int device, void *stream);
void pysp_ctx_destroy(pysp_ctx *ctx);
/* Re-bind the context to a caller-owned hipStream_t; NULL means the device's default (null) stream, which
 * pysp_ctx_create cannot express.  Work already enqueued on the previous stream is waited for, a stream the
 * context created itself is destroyed.  A torch caller passes torch.cuda.current_stream().cuda_stream so that
 * library kernels, torch ops, the caching allocator and RCCL collectives are ordered on ONE stream (no host
 * synchronisation between them).  Every entry point restores the calling thread's current HIP device. */
int pysp_ctx_set_stream(pysp_ctx *ctx, void *stream);
void *pysp_ctx_get_stream(pysp_ctx *ctx);

/* The two lookup tables of the restated float32 RGB->Lab used by the AHD homogeneity vote (replaces the
 * table-driven float path of cv2.cvtColor(COLOR_RGB2LAB) called at debayer/ahd.py:58,62): dec receives
 * 321x4 floats (sRGB decode, v in [2^-5,1]), cb 257x4 floats (cube root, t in [2^-7,2)).  Host only, needs no
 * GPU; exported so that tests can compare the tables with the CPU oracle's bit for bit. */
int pysp_lab_tables(float *dec, float *cb);
/* Which restatement of cv2.cvtColor(COLOR_RGB2LAB) (debayer/ahd.py:58,62) the AHD homogeneity metric of this context uses:
 *   1 (default) OpenCV 4.10's default float32 path: 33^3 int16 grid, fixed-point trilinear interpolation (14-bit), output
 *               quantised to 100/2^14 (L) and 1/64 (a, b) -- what opencv_python==4.10.0.84 is believed to run;
 *   0           closed form: sRGB decode + D65 CIELab with the table-driven pow / cbrt above (round 1's metric; 5 % faster).
 * Neither can be pinned without cv2; they differ in 2.7 % of the H/V decisions of the benchmark frame.
 * pysp_lab_cv410_lut writes the 33*33*33*3 int16 grid of mode 1 ([B][G][R] point order; host only, no GPU needed). */
int pysp_ctx_set_lab_mode(pysp_ctx *ctx, int mode);
int pysp_ctx_get_lab_mode(pysp_ctx *ctx);
int pysp_lab_cv410_lut(int16_t *out);
/* The grid of mode 1 as DATA (round 4).  cv2.cvtColor(COLOR_RGB2LAB) at debayer/ahd.py:58,62 interpolates a 33^3 table that this library can only
 * restate (no cv2 in the build image); pysp_ctx_set_lab_lut replaces the context's table by `grid` -- 33*33*33*3 int16, [B][G][R] node order, (L, a, b) per
 * node scaled as OpenCV's RGB2LabLUT_s16 (L*2^14/100, (a+128)*2^14/256, (b+128)*2^14/256; entries in [0, 32767], negative ones are refused) -- or
 * restores the built-in one (grid == NULL).  tools/gen_cv2_goldens.py records the real table (cv2.cvtColor at the node inputs (p/32, q/32, r/32): every
 * interpolation weight but one is zero there) so that a 1-LSB disagreement becomes a data change.  Waits for the context's stream, then uploads (2.5 MB).
 * pysp_ctx_get_lab_lut copies the table in use into out[33*33*33*3]. */
int pysp_ctx_set_lab_lut(pysp_ctx *ctx, const int16_t *grid);
/* How the AHD select kernel keeps the Lab values of mode 1 (a performance choice: both forms return the same bits, debayer/ahd_homogeneity_cython.pyx:47-58):
 *    0  packed cells { L, a'|b'<<16 }, chroma distances on the table's integers, seven workgroups per CU -- the select kernel 2 % faster on ordinary content; a wave that
 *       meets a chroma step of 64 Lab units or more between neighbouring pixels redoes its votes in float32 arithmetic (pure colour noise: +16 %);
 *    1  three float planes and float votes (round 3's kernel): the same speed on any content;
 *   -1  (default) automatic: the packed kernel counts the tiles that needed the float form; when more than a quarter of the tiles of a sample did, the
 *       context launches the planes form for the next 256 frames and then probes again.  No host wait is involved (one 8-byte copy per 16 launches,
 *       enqueued behind the kernels).
 * pysp_ctx_lab_layout_in_use: the form the next AHD call of this context will launch (0 or 1). */
int pysp_ctx_set_lab_layout(pysp_ctx *ctx, int layout);
int pysp_ctx_get_lab_layout(pysp_ctx *ctx);
int pysp_ctx_lab_layout_in_use(pysp_ctx *ctx);
/* Form of the AHD select kernel (a performance choice, same bits: debayer/ahd.py:97-145, debayer/ahd_homogeneity_cython.pyx:22-58):
 *    0  one 28x28 px tile per workgroup (16x16 threads, two of the sixteen thread rows and columns are halo);
 *    1  streaming (round 5): a persistent grid, each workgroup walks down a column of the frame and carries the Lab rows, votes and unselected candidates
 *       of its last quad row from pass to pass, so that every thread row yields an output row.  Taken for Lab mode 1, packed layout, no HDR metric; any
 *       other configuration launches form 0.
 * The default can be preset with the environment variable PYSP_SELECT_FORM (tile / stream). */
int pysp_ctx_set_select_form(pysp_ctx *ctx, int form);
int pysp_ctx_get_select_form(pysp_ctx *ctx);
/* The work partition of form 1 for an H x W frame (host arithmetic, needs no GPU: the CPU tests check that the chunks tile every column exactly once):
 * returns the number of chunks (or a negative error code) and, when out4 != NULL, writes { column tile, S, E, passes } per chunk -- output quad rows S + 1 .. E of
 * that 14-quad-wide column, a head pass and passes - 1 chained ones -- in queue order; first[x] / count[x] = the range of XCD x's queue.
 * slots_per_xcd: resident workgroups per queue the guided schedule assumes (the library itself uses CUs x occupancy / 8). */
int pysp_ahd_stream_chunks(int H, int W, int slots_per_xcd, int *out4, int max_chunks, unsigned first[8], unsigned count[8]);
int pysp_ctx_get_lab_lut(pysp_ctx *ctx, int16_t *out);
int pysp_ctx_sync(pysp_ctx *ctx);
/* Duration in ms of the most recent *_dev or host call's kernels on this context (HIP events on
 * the context's stream; waits for completion).  A host call on a frame of more than 4 MP runs in overlapped
 * row bands: the two timing queries then describe the LAST band only (about a third of a 24 MP frame);
 * time whole frames with the *_dev entry points.  A host fusion of more exposures than one pass takes
 * (pysp_fuse_raw_f32 / pysp_fuse_rgb_f32, K > 16 / 12) is timed from its first pass's kernel to its last pass's:
 * the later passes' uploads lie in between, so that figure is not kernel-only (pysp_ctx_kernel_times lists the passes). */
int pysp_ctx_last_kernel_ms(pysp_ctx *ctx, float *ms);
/* Event timing on the context's stream.  mode 0: no events are recorded (nothing but kernels is enqueued);
 * mode 1 (default): one event pair per entry-point call (pysp_ctx_last_kernel_ms); mode 2: in addition each
 * kernel of a demosaic / pipeline call is bracketed by its own pair.  kernel_times returns the durations (ms)
 * and static names of the kernels of the most recent such call (waits for them). */
int pysp_ctx_set_kernel_timing(pysp_ctx *ctx, int mode);
int pysp_ctx_kernel_times(pysp_ctx *ctx, int max_kernels, float *ms, const char **names, int *n_out);

/* Device buffers for callers that keep images on the GPU between calls and have no allocator of their own (the lazily
 * materialised arrays of the Python drop-in classes: RawDemosaicData.image, base_types/image_base.py:19-35, stays in HBM until
 * somebody reads it).  Freed blocks are cached by the context; reuse is safe because every consumer enqueues on the context's
 * stream.  upload enqueues (the host buffer must stay valid until the next synchronising call); download waits. */
void *pysp_dev_alloc(pysp_ctx *ctx, size_t bytes);
int pysp_dev_free(pysp_ctx *ctx, void *dptr);
int pysp_dev_upload(pysp_ctx *ctx, void *dptr, const void *host, size_t bytes);
int pysp_dev_download(pysp_ctx *ctx, void *host, const void *dptr, size_t bytes);
/* Page-locked host memory (hipHostMalloc / hipHostFree).  Host entry points accept any host pointer; buffers from here make
 * their copies asynchronous DMA (the overlapped bands of pysp_pipeline_* need that to overlap) and avoid first-touch page
 * faults in freshly allocated result arrays.  The Python wrapper hands out its result arrays from a pool of these. */
void *pysp_host_alloc(size_t bytes);
int pysp_host_free(void *p);

/* ---- Bayer plane helpers -------------------------------------------------------------------- */
/* bayer_chan_mixer.py:4-21 bayer_to_rgbg (float32 or uint16 mosaic -> four float32 quarter planes) */
int pysp_bayer_to_rgbg_f32(pysp_ctx *ctx, const float *bayer, int H, int W, float *r, float *g1, float *b, float *g2);
int pysp_bayer_to_rgbg_u16(pysp_ctx *ctx, const uint16_t *bayer, int H, int W, float *r, float *g1, float *b, float *g2);
/* bayer_chan_mixer.py:23-42 rgbg_to_bayer (h, w are the quarter-plane dimensions) */
int pysp_rgbg_to_bayer_f32(pysp_ctx *ctx, const float *r, const float *g1, const float *b, const float *g2, int h, int w, float *bayer);
/* normalization.py:4-24 bayer_normalize; black/sat indexed r,g1,b,g2 */
int pysp_bayer_normalize_u16(pysp_ctx *ctx, const uint16_t *bayer, int H, int W, const float black[4], const float sat[4], float *out);

/* ---- Stand-alone EAG helpers (public functions of debayer/edge_assisted_gaussian.py) -----------
 * :51-124 resample_g_to_full_resolution(g1, g2, use_bilinear_weighting): (h,w),(h,w) -> (2h,2w). */
int pysp_resample_g_f32(pysp_ctx *ctx, const float *g1, const float *g2, int h, int w, int use_bilinear_weighting, float *out);
/* :126-143 resample_channel(subpixel, g_at_subpixel, g_hf_pass, bayer_position) when g_full == NULL;
 * :160-186 resample_r / resample_b(channel, g_upscaled) when g_full != NULL (g_sub, g_hf ignored: the high-pass
 * and the green at the photosite are derived from g_full).  bayer_position: 0 TOP_LEFT, 3 BOTTOM_RIGHT. */
int pysp_resample_channel_f32(pysp_ctx *ctx, const float *sub, const float *g_sub, const float *g_hf, const float *g_full, int h, int w, int bayer_position, float *out);

/* ---- pre-demosaic cleanup (the step before the path) -----------------------------------------
 * raw_bad_pixel_corr.py:30-65 find_erroneous_pixels_threshold: four (H/2,W/2) uint8 masks (1 = hot) for the
 * r,g1,b,g2 planes. */
int pysp_find_hot_pixels_f32(pysp_ctx *ctx, const float *bayer, int H, int W, float min_delta, int min_neighbour_count, uint8_t *mask_r, uint8_t *mask_g1, uint8_t *mask_b, uint8_t *mask_g2);
/* corr_ca/ca_removal.py:48-131 remove_ca_from_raw, the apply half (lens-model fitting is host work): bayer (H,W) float32
 * is corrected in place.  Each quad_* is the top-left quadrant (H/2,W/2,2) = (dy,dx) of a lens model's coordinate field
 * (corr_ca/model/generic.py:56-101 get_distorted_coordinates for *_r_at_g / *_b_at_g, :113-163 get_undistorted_coordinates
 * for *_g_at_r / *_g_at_b); the library mirrors it into the other quadrants.  A NULL pair skips that channel (model None);
 * wb_r / wb_b are cam_wb.get_reciprocal_multipliers()[0] / [2]. */
int pysp_remove_ca_f32(pysp_ctx *ctx, float *bayer, int H, int W, const float *quad_g_at_r, const float *quad_r_at_g, float wb_r, const float *quad_g_at_b, const float *quad_b_at_g, float wb_b);
/* Same with the mosaic and the four quadrant fields resident on the device (a lens's fields are uploaded once per batch). */
int pysp_remove_ca_dev(pysp_ctx *ctx, float *d_bayer, int H, int W, const float *d_quad_g_at_r, const float *d_quad_r_at_g, float wb_r, const float *d_quad_g_at_b, const float *d_quad_b_at_g, float wb_b);
/* raw_correction.py:25-62 flat_frame_correction on mosaics; mean[4] = np.mean of the flat's r,g1,b,g2 planes
 * (:44, computed by the caller with NumPy so that the float32 pairwise-summation order is NumPy's). */
int pysp_flat_field_f32(pysp_ctx *ctx, const float *bayer, const float *flat, int H, int W, const float mean[4], int clamp_high, float *out);

/* ---- AHD homogeneity vote -------------------------------------------------------------------
 * debayer/ahd_homogeneity_cython.pyx:61-68  build_map(lab, k_pad, domain_k, is_vertical)
 * lab: (Hp,Wp,3) already padded by k_pad; out: (Hp-2k_pad, Wp-2k_pad) float32 counts.
 * domain_k is accepted and ignored by the reference (pyx:27), so it is not part of this ABI. */
int pysp_build_map_f32(pysp_ctx *ctx, const float *lab, int Hp, int Wp, int k_pad, int is_vertical, float *out);

/* ---- Demosaic -------------------------------------------------------------------------------
 * image.py:156-183 RawRggbBayerData.demosaic -> debayer/fast_resize.py:7-44 (Draft),
 * debayer/edge_assisted_gaussian.py:188-201 (Fast), debayer/ahd.py:14-170 (Best).
 *   wb  : cam_wb.get_reciprocal_multipliers()[:3] (wb_cct/cam_wb.py:236-243), float32
 *   M   : final 3x3 cam->linear-sRGB matrix, row-major float64, built on the host exactly as
 *         colorize/transform.py:40-49 does (only Best uses it, for the homogeneity metric)
 *   hdr : image.get_hdr() (base_types/image_base.py:82-88); stages : postprocess_steps
 * rgb: (H,W,3) float32, white-balanced camera RGB (RawDemosaicData.image). */
int pysp_demosaic_f32(pysp_ctx *ctx, const float *bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, float *rgb);
int pysp_demosaic_dev(pysp_ctx *ctx, const float *d_bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, float *d_rgb);

/* ---- Colour ---------------------------------------------------------------------------------
 * colorize/transform.py:21-53 cam_to_rgb_norm pixel step: optional clip to [0,1] (:6-19,:37-38), then
 * out = float32(float64 dot with M) (:52-53).  npx = number of RGB pixels.  in == out is allowed. */
int pysp_cam_to_rgb_f32(pysp_ctx *ctx, const float *in, size_t npx, const double M[9], int clip, float *out);
int pysp_cam_to_rgb_dev(pysp_ctx *ctx, const float *d_in, size_t npx, const double M[9], int clip, float *d_out);
/* colorize/transform.py:89-99 lin_srgb_to_srgb and :101-111 srgb_to_lin_srgb; n = number of floats */
int pysp_lin_srgb_to_srgb_f32(pysp_ctx *ctx, const float *in, size_t n, float *out);
int pysp_lin_srgb_to_srgb_dev(pysp_ctx *ctx, const float *d_in, size_t n, float *d_out);
int pysp_srgb_to_lin_srgb_f32(pysp_ctx *ctx, const float *in, size_t n, float *out);
/* base_types/image_base.py:45-60 wb_apply (image*coeff -> f32) / wb_undo (f64 divide -> f32) */
int pysp_wb_scale_f32(pysp_ctx *ctx, const float *in, size_t npx, const float coeff[3], int undo, float *out);
int pysp_wb_scale_dev(pysp_ctx *ctx, const float *d_in, size_t npx, const float coeff[3], int undo, float *d_out);

/* ---- Fused recipe (README.md:55-63): demosaic -> to_lin_srgb (clip on) -> [x/(1+x), README.md:157]
 * -> lin_srgb_to_srgb, one frame, no intermediate leaves the GPU.  srgb: (H,W,3) float32. */
int pysp_pipeline_srgb_f32(pysp_ctx *ctx, const float *bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int reinhard, float *srgb);
/* The same host-buffer pipeline with any colour tail (0 RawDemosaicData.image, base_types/image_base.py:27; 1 to_lin_srgb, :62-64; 2 + lin_srgb_to_srgb,
 * colorize/transform.py:89-99; 3 with README.md:157's x/(1+x) in between): what the drop-in classes' deferred mode (pysp_amd.set_lazy("deferred")) collapses
 * demosaic() -> to_lin_srgb() [-> lin_srgb_to_srgb()] into -- one upload overlapped with the kernels and the download, in bands. */
int pysp_pipeline_f32(pysp_ctx *ctx, const float *bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int tail, float *out);
int pysp_pipeline_srgb_dev(pysp_ctx *ctx, const float *d_bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int reinhard, float *d_srgb);
/* n_frames HOST mosaics of one geometry and one set of camera parameters -> n_frames host results (BASELINE config 3's frames as they arrive from a decoder;
 * the loop `for raw in frames: raw.demosaic(q).to_lin_srgb()` of README.md:55-63) through ONE band chain: frame k+1's bands are uploaded and computed while
 * frame k's are still on their way down, so a stream of frames costs its downloads plus one band instead of upload + download + drain per call.  The
 * asynchronous form needs every result page-locked (pysp_host_alloc / hipHostMalloc / hipHostRegister) and every mosaic page-locked or lockable for the
 * call; anything else is processed frame by frame (pysp_pipeline_f32 n times).  Same bits as n single calls either way.  The u16 form takes sensor data
 * (normalization.py:4-24 fused into the loader, as pysp_pipeline_u16_f32). */
int pysp_pipeline_batch_f32(pysp_ctx *ctx, const float *const *bayers, int n_frames, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int tail,
                            float *const *outs);
int pysp_pipeline_batch_u16_f32(pysp_ctx *ctx, const uint16_t *const *bayers, int n_frames, int H, int W, const float black[4], const float sat[4], const float wb[3],
                                const double M[9], int quality, int hdr, int stages, int tail, float *const *outs);

/* General form: tail 0 = pysp_demosaic_dev, 1 = + to_lin_srgb (clip + CCM; BASELINE config 3 "debayer + WB + CCM"),
 * 2 = pysp_pipeline_srgb_dev, 3 = with x/(1+x) in between. */
int pysp_pipeline_dev(pysp_ctx *ctx, const float *d_bayer, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int tail, float *d_out);
/* A batch of frames that share camera parameters (BASELINE config 3: the frames one rank owns): n_frames device mosaics ->
 * n_frames device images, enqueued back to back on the context's stream with one call. */
int pysp_pipeline_batch_dev(pysp_ctx *ctx, const float *const *d_bayers, int n_frames, int H, int W, const float wb[3], const double M[9], int quality, int hdr, int stages, int tail, float *const *d_outs);

/* The same pipelines fed by the raw uint16 mosaic: normalization.py:4-24 (clip(x-black_c,0,sat_c)/sat_c, CFA
 * sites indexed r,g1,b,g2) is fused into the tile loader, so the float32 mosaic never exists in memory
 * (image.py:229 followed by :156-183).  tail: 0 camera RGB (RawDemosaicData.image), 1 to_lin_srgb,
 * 2 + lin_srgb_to_srgb, 3 with x/(1+x) in between. */
int pysp_pipeline_u16_f32(pysp_ctx *ctx, const uint16_t *bayer, int H, int W, const float black[4], const float sat[4], const float wb[3], const double M[9], int quality, int hdr, int stages, int tail, float *out);
int pysp_pipeline_u16_dev(pysp_ctx *ctx, const uint16_t *d_bayer, int H, int W, const float black[4], const float sat[4], const float wb[3], const double M[9], int quality, int hdr, int stages, int tail, float *d_out);

/* ---- HDR raw fusion -------------------------------------------------------------------------
 * raw_hdr.py:85-158 fuse_exposures_to_raw, pixel loop :135-148.  The host computes, with NumPy as
 * the reference does, ev_off[k] = float32(2**(ev_k-target)) (:119-121), bias[k*4+c] =
 * float32(1.6**(-0.1*|ev_off_k*w_c|)) for CFA site c in r,g1,b,g2 order (:128-136), and
 * kmax = argmax(ev_off) (:143).  out: (H,W) float32; count: (H,W) int32 (:123,:141).
 * Any K >= 1 (as the reference's loop): more than 16 exposures run as passes of 16 in order, the partial sums carried in the
 * context's workspace -- the same float32 additions in the same order, so the same bits as one pass. */
int pysp_fuse_raw_f32(pysp_ctx *ctx, const float *const *frames, int K, int H, int W, const float *ev_off, const float *bias, int kmax, float *out, int32_t *count);
int pysp_fuse_raw_dev(pysp_ctx *ctx, const float *const *d_frames, int K, int H, int W, const float *ev_off, const float *bias, int kmax, float *d_out, int32_t *d_count);

/* raw_hdr.py:7-83 fuse_exposures_from_debayer, pixel loop :54-81, K >= 1 exposures of npx RGB pixels (passes of 16, as above).
 *   coeff[k*3+c] = exposure._wb_coeff, applied[k] = exposure._wb_applied (image_base.py:31,45-60),
 *   bias[k] = float32(1.6**(-0.1*ev_off_k)) (:60-61), kmax = last k with ev_off_k == max (:67-68),
 *   M = final matrix for the trailing cam_to_lin_srgb(clip_highlights=False) (:81) or NULL for none.
 * write_back != 0 stores into frames[k] the wb_undo/wb_apply round-tripped image the reference leaves
 * in exposure.image (:56,:65).  out: (npx,3) float32, count: (npx,3) int32. */
int pysp_fuse_rgb_f32(pysp_ctx *ctx, float *const *frames, int K, size_t npx, const float *coeff, const int *applied, const float *ev_off, const float *bias, int kmax, const double *M, float *out, int32_t *count, int write_back);
/* The same on device buffers (exposures that a demosaic left in HBM are fused there): d_frames_rt is NULL or K pointers, each NULL or a buffer
 * (it may be d_frames[k] itself) that receives exposure k's wb_undo/wb_apply round trip (raw_hdr.py:56,:65).  Enqueued on the context's stream. */
int pysp_fuse_rgb_dev(pysp_ctx *ctx, const float *const *d_frames, float *const *d_frames_rt, int K, size_t npx, const float *coeff, const int *applied, const float *ev_off, const float *bias, int kmax, const double *M, float *d_out, int32_t *d_count);

/* ---- DNG WarpRectilinear --------------------------------------------------------------------
 * dng_warp_corr/dng_warp_rectilinear_coords.pyx:67-80 compute_remapping_table and :82-96
 * compute_offset_remapping_table (seed != NULL); table: (height,width,2) float32. */
int pysp_warp_table_f32(pysp_ctx *ctx, float kr0, float kr1, float kr2, float kr3, float kt0, float kt1, int width, int height, float cx_norm, float cy_norm, float scale, const float *seed, float *table);
/* dng_warp_corr/chan_distortion_corr.py:86-97: per plane table -> clip -> Lanczos-4 remap, in place
 * on an (H,W,3) image; coeffs = planes x {kr0..kr3,kt0,kt1} float64 as unpacked from the opcode. */
int pysp_warp_rectilinear_f32(pysp_ctx *ctx, float *image, int H, int W, const double *coeffs, int planes, double cx_norm, double cy_norm, float scale);
/* Seeded variant (prior != None, :88-91): prior is the (H,W,3,2) float32 stack of stack_warp_prior (:11-41). */
int pysp_warp_rectilinear_prior_f32(pysp_ctx *ctx, float *image, int H, int W, const double *coeffs, int planes, double cx_norm, double cy_norm, float scale, const float *prior);
/* The restated cv2.remap(plane, mapx, mapy, INTER_LANCZOS4) itself (:94-97), maps as given (no clipping). */
int pysp_remap_lanczos4_f32(pysp_ctx *ctx, const float *src, int H, int W, const float *mapx, const float *mapy, float *dst);
int pysp_warp_rectilinear_dev(pysp_ctx *ctx, const float *d_in, float *d_out, int H, int W, const double *coeffs, int planes, double cx_norm, double cy_norm, float scale);
/* Band form for one frame sharded over several GPUs (BASELINE config 5): d_in and d_out address the whole
 * (H,W,3) frame, only output rows [row0,row1) are produced; source rows outside pysp_warp_source_rows' range are never read. */
int pysp_warp_rectilinear_rows_dev(pysp_ctx *ctx, const float *d_in, float *d_out, int H, int W, const double *coeffs, int planes, double cx_norm, double cy_norm, float scale, int row0, int row1);
/* Rows [*src_row0,*src_row1) of the source frame that the Lanczos footprints of output rows [row0,row1) touch (same
 * coordinate arithmetic as the remap itself, reduced on the device): what a band has to receive from its peers. */
int pysp_warp_source_rows(pysp_ctx *ctx, int H, int W, const double *coeffs, int planes, double cx_norm, double cy_norm, float scale, int row0, int row1, int *src_row0, int *src_row1);

#ifdef __cplusplus
}
#endif
#endif /* PYSP_HIP_H */
