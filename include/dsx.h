/*
 * dsx.h — C ABI of libdsx.so, the MI355X (gfx950) sampling engine that sits
 * underneath the DiffSplitting Python entry points.
 *
 * The reference (rayanirban/DiffSplitting) has no native/FFI layer: its
 * boundary is the Python API in model/networks.py:91 (define_G),
 * model/model.py:63 (DDPM.test) and the sampler classes.  The entry points
 * below are what a native replacement for that hot path binds; each one cites
 * the reference interface it replaces.  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions: every function returns 0 on success and a negative dsx_status
 * on failure; dsx_last_error() returns a thread-local message.  Nothing throws
 * across the ABI.  "dev" pointers are device (HBM) pointers borrowed from the
 * caller (e.g. torch.Tensor.data_ptr()); "host" pointers are plain host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Calls on one handle are not concurrent; distinct handles may run on distinct
 * streams concurrently.  One process drives one GPU.
 */
#ifndef DSX_H
#define DSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSX_ABI_VERSION 2

typedef enum dsx_status {
  DSX_OK = 0,
  DSX_ERR_INVALID = -1,   /* bad argument / shape the kernels do not support */
  DSX_ERR_HIP = -2,       /* a HIP runtime call failed (no device, OOM, ...) */
  DSX_ERR_STATE = -3,     /* call order violated (e.g. forward before finalize) */
  DSX_ERR_MISSING = -4    /* a parameter was never set */
} dsx_status;

const char* dsx_last_error(void);
int dsx_abi_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int dsx_device_count(void);

/* ------------------------------------------------------------------ UNet */

enum { DSX_FLAVOUR_SR3 = 0, DSX_FLAVOUR_DDPM = 1 };
/* MFMA operand type = activation storage type in HBM; accumulation, GroupNorm statistics and the sampler state are
 * always f32.  F16 is what config/splitting_hagen_indi_joint.json is quoted on (BASELINE "fp16"). */
enum { DSX_DTYPE_F32 = 0, DSX_DTYPE_BF16 = 1, DSX_DTYPE_F16 = 2 };

/* Mirrors the keyword arguments of UNet.__init__
 * (model/sr3_modules/unet.py:161-174, model/ddpm_modules/unet.py:150-162). */
typedef struct dsx_unet_cfg {
  int32_t flavour;          /* DSX_FLAVOUR_* : gamma (sr3) or t (ddpm) conditioning */
  int32_t in_channel;
  int32_t out_channel;
  int32_t inner_channel;
  int32_t norm_groups;
  int32_t n_mults;
  int32_t channel_mults[8];
  int32_t n_attn_res;
  int32_t attn_res[8];
  int32_t res_blocks;
  int32_t image_size;       /* only used to place attn_res, as in the reference */
  int32_t with_time_emb;    /* 0 for the TimePredictor's UNet (time_predictor.py:24-33) */
} dsx_unet_cfg;

typedef struct dsx_model dsx_model;

/* Host-only: builds the topology and the parameter table.  Works without a GPU. */
int dsx_model_create(const dsx_unet_cfg* cfg, dsx_model** out);
void dsx_model_destroy(dsx_model* m);

/* The parameter table uses the reference's state_dict key names relative to
 * the UNet (e.g. "downs.1.res_block.block1.block.3.weight"), in state_dict
 * order, so a *_gen.pth (model/model.py:131-173) maps 1:1.  Shapes are the
 * reference's (OIHW conv weights, [out,in] linears). */
int dsx_model_num_params(const dsx_model* m);
int dsx_model_param_info(const dsx_model* m, int index, char* name_buf, int name_cap,
                         int* ndim, int64_t shape[4]);
/* Copies one parameter (fp32, reference layout, host memory). */
int dsx_model_set_param(dsx_model* m, int index, const float* host_data, int64_t numel);
/* sr3 only: the PositionalEncoding frequency table exp(-ln(1e4)*k/(d/2)),
 * k<d/2 (sr3 unet.py:24-28); optional — computed with expf() if never set. */
int dsx_model_set_posenc_freq(dsx_model* m, const float* host_freq, int count);
/* Repacks all weights into the kernels' layouts (MFMA fragment order, fp32 or
 * bf16) and uploads them to the current HIP device. */
int dsx_model_finalize(dsx_model* m, int compute_dtype);
/* Packed-weight cache: the repack of a checkpoint (model/model.py:153-166 loads `*_gen.pth`; the engine then
 * reorders every conv into MFMA fragment order) is done once.  dsx_model_export_packed copies the device image of
 * a finalized model to host memory (dsx_model_packed_bytes bytes for this model and dtype); dsx_model_finalize_packed
 * finalizes a freshly created model of the same configuration straight from such an image: no dsx_model_set_param,
 * no repacking.  The image layout depends on (configuration, dtype, DSX_ABI_VERSION): callers key their cache on
 * those and on the checkpoint's hash. */
int dsx_model_packed_bytes(dsx_model* m, int compute_dtype, size_t* bytes);
int dsx_model_export_packed(const dsx_model* m, void* host_buf, size_t capacity);
int dsx_model_finalize_packed(dsx_model* m, int compute_dtype, const void* host_image, size_t bytes);
/* Algorithmic FLOPs (2*MAC of conv/linear/attention contractions) of one
 * forward of one image of H x W. */
double dsx_model_flops(const dsx_model* m, int H, int W);

/* --------------------------------------------------------------- executor */

typedef struct dsx_exec dsx_exec;

/* Plans one UNet forward for a fixed (B,H,W): kernel list, tile shapes,
 * activation workspace (hipMalloc'ed once, sized for 288 GB parts: no reuse
 * games).  cond_channels > 0 declares that the first conv reads its input as
 * two tensors (cond, x) instead of a materialised torch.cat
 * (sr3 diffusion.py:157-158). */
int dsx_exec_create(dsx_model* m, int B, int H, int W, int cond_channels, dsx_exec** out);
void dsx_exec_destroy(dsx_exec* ex);
size_t dsx_exec_workspace_bytes(const dsx_exec* ex);
/* Diagnostics of the conv kernel's in-workgroup hand-off (bounded spins on LDS counters, tiles with three MFMA
 * images): how many spins gave up since dsx_exec_create.  0 in every correct run -- a non-zero count means pixels of
 * some launch were wrong and the caller must not use them.  Synchronises the device. */
int dsx_exec_handoff_timeouts(dsx_exec* ex, unsigned* count);
/* Host-only (no device needed): runs the planner's sizing pass and its planning pass for this geometry and
 * reports the workspace bytes each of them walked and the launch count.  The two must agree; dsx_exec_create
 * fails if they do not.  Lets CPU tests pin the planner under every tile-preference environment setting. */
int dsx_plan_dry_run(const dsx_unet_cfg* cfg, int compute_dtype, int B, int H, int W, int cond_channels,
                     size_t* sizing_bytes, size_t* planning_bytes, int* launches);
int dsx_exec_num_launches(const dsx_exec* ex);

/* Launch-level introspection for measurement (bench.py roofline): the plan's
 * launches in order, what each computes, and an eager hipEvent-timed replay. */
enum { DSX_OP_CONV_MFMA = 0, DSX_OP_CONV_NAIVE = 1, DSX_OP_GN_STATS = 2, DSX_OP_GN_FINALIZE = 3,
       DSX_OP_ATTN_GEMM = 4 /* the fused attention kernel */, DSX_OP_SOFTMAX = 5 /* unused since ABI 2 */,
       DSX_OP_SPLITK_REDUCE = 6 };
int dsx_exec_num_ops(const dsx_exec* ex);
int dsx_exec_op_info(const dsx_exec* ex, int index, char* desc_buf, int desc_cap, int* kind,
                     double* flops, double* bytes);
int dsx_exec_profile(dsx_exec* ex, int iters, float* ms_per_op, void* stream);
/* The launches of one kind (DSX_OP_*) captured into a hipGraph of their own and replayed `iters` times
 * between two hipEvents on `stream`: *ms_per_replay is the time of one back-to-back pass over them, i.e.
 * what `rocprofv3 --kernel-trace` sums for that kernel family inside the captured sampling step (the eager
 * per-launch times of dsx_exec_profile additionally contain the launch gaps). Inputs are whatever the
 * workspace holds; the timing of these kernels does not depend on the data. */
/* kind >= 0: that kind only; -1: every launch of the forward; <= -2: every launch except kind (-kind - 2), so that
 * time(-1) - time(-2 - k) is the time kind k takes INSIDE the forward (neighbouring launches warm its caches). */
int dsx_exec_time_kind(dsx_exec* ex, int kind, int iters, float* ms_per_replay, int* launches, void* stream);
/* diagnostics: in-kernel phase stamps (s_memtime) of the conv launch selected by DSX_STAMP_OP */
int dsx_exec_read_stamps(dsx_exec* ex, unsigned long long* out128);

/* One UNet forward: replaces denoise_fn(x, t)
 * (sr3 unet.py:235-259 / ddpm unet.py:220-243).
 *   x_nchw_dev : (B, in_channel, H, W) fp32, NCHW as the reference passes it
 *   time_dev   : n_time fp32 values; n_time == B (sr3 gamma (B,1), ddpm t (B,))
 *                or 1 (InDI's single scalar, indi.py:65); NULL when
 *                with_time_emb == 0
 *   y_nchw_dev : (B, out_channel, H, W) fp32 */
int dsx_unet_forward(dsx_exec* ex, const float* x_nchw_dev, const float* time_dev, int n_time,
                     float* y_nchw_dev, void* stream);

/* TimePredictor head (time_predictor.py:35-44): relu(unet(x)) * sigmoid(conv7x7(x)),
 * masked mean per image.  mask_w: (1,in,7,7) host fp32, mask_b: (1,) host. */
int dsx_time_predictor_set_mask(dsx_exec* ex, const float* mask_w_host, const float* mask_b_host);
int dsx_time_predictor_forward(dsx_exec* ex, const float* x_nchw_dev, float* t_out_dev, void* stream);

/* ---------------------------------------------------------------- sampler */

/* One row per reverse step, in execution order (host-computed, fp32, so that
 * "schedule indexing" is bit-exact with the reference):
 *   tcond : value fed to the UNet's time embedding
 *           (sr3: sqrt_alphas_cumprod_prev[i+1], diffusion.py:153-154;
 *            ddpm: float(i); InDI: fp32(cur_t), indi.py:65)
 *   predict_eps = 1 (SR3/DDPM, diffusion.py:141-175):
 *           x0 = a*x - b*eps ; clamp(+-1) if clip ; x <- (c1*x0 + c2*x) + sigma*z
 *   predict_eps = 0 (InDI, indi.py:62-69):
 *           x <- (c1*net + c2*x) + sigma*z      (a, b ignored)
 * Every product and sum is rounded separately, like the reference's op
 * sequence (no FMA contraction). */
typedef struct dsx_step_table {
  int32_t n_steps;
  int32_t predict_eps;
  int32_t clip;
  const float* tcond;   /* host, n_steps */
  const float* a;
  const float* b;
  const float* c1;
  const float* c2;
  const float* sigma;
  /* 0: one row of scalars per step, shared by the batch (the reference's loops).  B (= the executor's batch): every
   * column holds n_steps * B values, [step][sample] — per-sample schedules, e.g. InDI started at a per-tile t
   * predicted by the TimePredictor (core/psnr_based_t_refinement.py:22-36 loops over the batch one sample at a time) */
  int32_t per_sample;
} dsx_step_table;

/* Runs the whole reverse loop on `stream` without host synchronisation:
 * replaces GaussianDiffusion.p_sample_loop (sr3 diffusion.py:177-203,
 * ddpm diffusion.py:205-237) and InDI.inference's loop (indi.py:86-90).
 *   cond_nchw_dev : (B, cond_channels, H, W) or NULL (must match dsx_exec_create)
 *   x_nchw_dev    : (B, C, H, W) in: initial state (the caller draws it, as
 *                   diffusion.py:194 / indi.py:82 do); out: final state, FULL batch
 *   noise_nchw_dev: NULL -> device Philox normals keyed by (seed, step);
 *                   else (n_steps, B, C, H, W) injected draws in the
 *                   reference's draw order (parity mode)
 *   snap_steps    : host array of step ordinals (0-based) after which the
 *                   state is copied to snap_nchw_dev[k] (B,C,H,W each); may be NULL
 *   use_graph     : 1 = capture one step into a hipGraph and replay it */
int dsx_sample_loop(dsx_exec* ex, const dsx_step_table* tab,
                    const float* cond_nchw_dev, float* x_nchw_dev,
                    const float* noise_nchw_dev, uint64_t seed,
                    const int32_t* snap_steps, int n_snap, float* snap_nchw_dev,
                    int use_graph, void* stream);

/* One reverse step on the state x (B, C, H, W) in place: dsx_sample_loop with a one-row table, eager.
 *   dsx_sr3_step : p_sample of sr3 / ddpm (sr3 diffusion.py:141-175): the step's scalars as set_new_noise_schedule
 *                  tabulates them -- noise_level = sqrt_alphas_cumprod_prev[t+1], sqrt_recip_alphas_cumprod[t],
 *                  sqrt_recipm1_alphas_cumprod[t], posterior_mean_coef1/2[t], sigma = exp(0.5 posterior_log_variance
 *                  _clipped[t]) (0 at t == 0: no draw)
 *   dsx_indi_step: inference_one_step of InDI (indi.py:62-69): x <- c_x0 UNet(x, t) + c_xt x + noise_scale z with
 *                  c_x0 = delta / t, c_xt = 1 - delta / t, noise_scale = e (t - delta)
 * noise_dev: (1, B, C, H, W) injected draw or NULL -> Philox normals keyed by seed. */
int dsx_sr3_step(dsx_exec* ex, float noise_level, float sqrt_recip_alphas_cumprod, float sqrt_recipm1_alphas_cumprod,
                 float posterior_mean_coef1, float posterior_mean_coef2, float sigma, int clip_denoised,
                 const float* cond_nchw_dev, float* x_nchw_dev, const float* noise_dev, uint64_t seed, void* stream);
int dsx_indi_step(dsx_exec* ex, float t_cur, float c_x0, float c_xt, float noise_scale, float* x_nchw_dev,
                  const float* noise_dev, uint64_t seed, void* stream);

/* Fills n fp32 values with N(0,1) from the engine's Philox4x32-10 stream. */
int dsx_randn(float* out_dev, int64_t n, uint64_t seed, uint64_t subsequence, void* stream);

/* ----------------------------------------------------------------- tiling */

enum { DSX_TILING_TRIM = 0, DSX_TILING_PAD = 1, DSX_TILING_SHIFT = 2 };  /* tiling_manager.py:6-12 */

/* Tile enumeration for data (N,H,W), replaces TileIndexManager
 * (data/tiling_manager.py:34-154) as SplitDatasetTiledPred sets it up
 * (data/split_dataset_tiledpred.py:9-24).  Host-only integer math.
 * Returns the tile count; if grid_start/patch_start are non-NULL they
 * receive count*3 entries (n, y, x). */
int64_t dsx_tile_plan(const int64_t data_shape[3], const int64_t grid_shape[3],
                      const int64_t patch_shape[3], int tiling_mode,
                      int64_t* grid_start, int64_t* patch_start, int64_t capacity);

/* Valid region of every tile (tile_stitcher.py:26-56): dst start (n,y,x),
 * extent (1,h,w) and the offset (y,x) inside the tile.  8 int32 per tile:
 * {n, y, x, h, w, ry, rx, 0}. */
int dsx_tile_regions(const int64_t data_shape[3], const int64_t grid_shape[3],
                     const int64_t patch_shape[3], int tiling_mode,
                     int32_t* regions, int64_t capacity);

/* Cuts tiles out of frames on the device: frames (N,H,W) fp32 ->
 * tiles (count, ph, pw) for the tiles listed in tile_ids (host array).
 * Replaces the per-item crop of SplitDataset.__getitem__
 * (data/split_dataset.py:237-246) for batch dispatch. */
int dsx_tiles_gather(const float* frames_dev, const int64_t data_shape[3],
                     const int64_t patch_shape[3], const int64_t* patch_start_host,
                     const int64_t* tile_ids_host, int64_t count, float* tiles_dev, void* stream);

/* dsx_tiles_gather for both raw channels plus the dataset's normalisation, fused: replaces
 * SplitDataset.__getitem__ (data/split_dataset.py:237-278: crop, normalize_target :199-201, weighted input,
 * normalize_inp :195-197) for a whole batch of tiles.  norm = {mean_input, std_input, mean_target0, std_target0,
 * mean_target1, std_target1} (float64, as compute_normalization_dict :29-74 returns them); from_norm_target selects
 * input = w0*target0 + w1*target1 (input_from_normalized_target).  tiles_in (count,1,ph,pw), tiles_target (count,2,ph,pw). */
int dsx_tiles_gather_norm(const float* frames0_dev, const float* frames1_dev, const int64_t data_shape[3],
                          const int64_t patch_shape[3], const int64_t* patch_start_host, const int64_t* tile_ids_host,
                          int64_t count, float w0, float w1, const double norm[6], int from_norm_target,
                          float* tiles_in_dev, float* tiles_target_dev, void* stream);

/* Pastes the valid region of `count` predicted tiles (count, C, ph, pw) into
 * the zero-initialised canvas (N,H,W,C), channel-last: replaces
 * stitch_predictions (data/tile_stitcher.py:10-81).  regions as from
 * dsx_tile_regions for exactly these tiles. */
int dsx_stitch(const float* tiles_dev, int64_t count, int C, int ph, int pw,
               const int32_t* regions_host, float* canvas_dev, const int64_t data_shape[3],
               void* stream);

/* dsx_stitch plus, in the same pass over the tiles, the sums the reported quality metric needs
 * (RangeInvariantPsnr, core/psnr.py:70-82) of the pasted prediction p against the ground-truth canvas g
 * (N,H,W,C fp32): partials_dev receives count * dsx_stitch_psnr_blocks(ph,pw) * C * 8 doubles
 * {sum p, sum p^2, sum g, sum g^2, sum g p, min g, max g, 0} per (tile, workgroup, channel); the caller adds the
 * rows of a frame (fixed order) and evaluates the closed form.  Replaces the host-side pass over the 335 MB
 * stitched canvas (notebooks/EvaluateJointIndi.ipynb cell 30). */
int dsx_stitch_psnr_blocks(int ph, int pw);
int dsx_stitch_psnr(const float* tiles_dev, int64_t count, int C, int ph, int pw, const int32_t* regions_host,
                    float* canvas_dev, const int64_t data_shape[3], const float* gt_canvas_dev, double* partials_dev,
                    void* stream);

/* ------------------------------------------------- tile plan with device-resident tables (the stall-free forms)
 * The entry points above take host tables and upload them per call (a small allocation and a synchronous copy each).
 * A dsx_tileplan keeps the patch starts and valid regions of every tile on the device (uploaded once, at first
 * device use); every call names its tiles as the arithmetic sequence first, first + stride, ... (count terms) -- the
 * shard r, r + W, r + 2W, ... of rank r, or a batch of it -- and the kernels index the tables by tile id: no
 * allocation, copy or synchronisation per call.  Replaces SplitDatasetTiledPred's per-index TileIndexManager lookups
 * (data/split_dataset_tiledpred.py:9-32) and stitch_predictions (data/tile_stitcher.py:10-81) for batch dispatch. */
typedef struct dsx_tileplan dsx_tileplan;
int dsx_tileplan_create(const int64_t data_shape[3], const int64_t grid_shape[3], const int64_t patch_shape[3],
                        int tiling_mode, dsx_tileplan** out);        /* host only; fails if a tile leaves the frames */
void dsx_tileplan_destroy(dsx_tileplan* plan);
int64_t dsx_tileplan_total(const dsx_tileplan* plan);
/* The plan's paste regions, 8 int32 per tile as dsx_tile_regions: stitch_predictions pastes tile after tile
 * (tile_stitcher.py:68-80), so where valid regions overlap (the shifted last tile of a ragged extent re-covers a strip
 * of its neighbour) the later tile's pixels stay; the plan clips the earlier tile's region to what survives, so that
 * all tiles can be pasted at once and every canvas pixel is written -- and exchanged -- exactly once. */
int dsx_tileplan_regions(const dsx_tileplan* plan, int32_t* regions, int64_t capacity);
/* dsx_tiles_gather / dsx_tiles_gather_norm for the tiles first + k*stride, k < count (<= 65535 per call) */
int dsx_tileplan_gather(dsx_tileplan* plan, const float* frames_dev, int64_t first, int64_t stride, int64_t count,
                        float* tiles_dev, void* stream);
int dsx_tileplan_gather_norm(dsx_tileplan* plan, const float* frames0_dev, const float* frames1_dev, int64_t first,
                             int64_t stride, int64_t count, float w0, float w1, const double norm[6],
                             int from_norm_target, float* tiles_in_dev, float* tiles_target_dev, void* stream);
/* dsx_stitch (gt_canvas_dev == NULL) or dsx_stitch_psnr for whole predicted tiles (count, C, ph, pw) of the sequence */
int dsx_tileplan_stitch(dsx_tileplan* plan, const float* tiles_dev, int C, int64_t first, int64_t stride, int64_t count,
                        float* canvas_dev, const float* gt_canvas_dev, double* partials_dev, void* stream);
/* Multi-GPU exchange of CROPPED tiles (the crop of tile_stitcher.py:38-56 applied before the collective).
 * Rank q of `world` owns the tiles q, q + world, ...; its packed run holds their valid regions [C][h][w] back to back
 * in id order.  dsx_tileplan_pack_layout (host only) gives the pixel offset of every tile inside its rank's run and
 * the pixels of every rank's run (multiply by C for elements); the collective ships max(rank run) elements per rank
 * instead of whole (C, ph, pw) tiles.
 *   dsx_tileplan_pack        : tiles first, first + world, ... (count of them, (count, C, ph, pw)) -> their places in
 *                              flat_rank_dev, the run of rank first % world
 *   dsx_tileplan_paste_packed: every tile of the plan from the gathered buffer [world][rank_stride_elems] into the
 *                              canvas (N,H,W,C); with gt_canvas_dev also the PSNR partial sums (total * blocks * C * 8) */
int dsx_tileplan_pack_layout(const dsx_tileplan* plan, int world, int64_t* tile_offset_pixels /*[total]*/,
                             int64_t* rank_pixels /*[world]*/);
int dsx_tileplan_pack(dsx_tileplan* plan, const float* tiles_dev, int C, int world, int64_t first, int64_t count,
                      float* flat_rank_dev, void* stream);
int dsx_tileplan_paste_packed(dsx_tileplan* plan, const float* flat_all_dev, int C, int world, int64_t rank_stride_elems,
                              float* canvas_dev, const float* gt_canvas_dev, double* partials_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSX_H */
