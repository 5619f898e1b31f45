/*
 * nerfmatch_amd -- C ABI of the MI355X (gfx950) implementation of the NeRFMatch hot path.
 *
 * The reference (nv-dvl/nerfmatch) has no FFI layer: its boundary is the Python class API
 * (SURVEY.md section 8b).  Every entry point below replaces a stack of eager torch ops inside one of
 * those classes; the reference location each one replaces is cited as file:line relative to the
 * reference tree.  the nerfmatch_amd python modules bind these with ctypes.
 *
 * Conventions
 *   - all tensors are contiguous row-major fp32 unless stated; `dev` = device (HBM) pointer,
 *     `host` = host pointer;
 *   - no allocation, no global state, no synchronisation inside: work is enqueued on `stream`
 *     (a hipStream_t passed as void*) and the call returns immediately;
 *   - return value: NM_OK or an NM_ERR_* code (nm_error_string() gives text).  Nothing is enqueued
 *     when an error is returned.
 */
#ifndef NERFMATCH_AMD_H
#define NERFMATCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nmStream_t; /* hipStream_t */

enum {
  NM_OK = 0,
  NM_ERR_ARG = 1,         /* null pointer / non-positive size */
  NM_ERR_UNSUPPORTED = 2, /* shape outside what the kernels are built for */
  NM_ERR_LAUNCH = 3,      /* HIP reported a launch error */
  NM_ERR_WORKSPACE = 4    /* workspace too small */
};

int nm_abi_version(void);
const char* nm_error_string(int code);

/* Measurement aid (bench.py's `roofline.peak_sustained`; no reference counterpart): a bare v_mfma_f32_32x32x16_f16 stream,
 * `workgroups` x 4 wavefronts x `rounds` x 24 MFMAs of 32768 FLOP; sink: workgroups * 256 floats. */
int nm_probe_mfma_f16(float* sink, int workgroups, int rounds, nmStream_t stream);

/* Compute-unit partitions (round 6; no reference counterpart -- the reference runs render and matcher of its batch-1 loop one after the
 * other on torch's default stream, nerfmatch/nerfmatch_evaluator.py:556-574, :660-679).  nm_stream_create_cu_mask returns a HIP stream
 * whose kernels run on the compute units whose bits are set in mask_host (n_words 32-bit words, bit i = unit i / 8 of XCD i % 8), so that
 * one query's render (persistent workgroups, one per CU of the partition) and the previous query's matcher (many short dependent launches
 * on the rest of the chip) run side by side.  nm_stream_cus(stream) = the number of CUs a persistent kernel on `stream` sizes its grid to
 * (the partition's size; the whole device for any other stream).  Results never depend on it: tiles are independent.
 * The registry of such streams (at most 16) is the library's only process-wide state.  nm_stream_destroy takes only streams made here. */
int nm_stream_create_cu_mask(const uint32_t* mask_host, int n_words, nmStream_t* stream);
int nm_stream_destroy(nmStream_t stream);
int nm_stream_cus(nmStream_t stream);

/* Parameter fingerprints (round 6; no reference counterpart: the reference keeps no derived copies of its parameters).  One launch sums
 * the 32-bit words of n_tensors device tensors (ptrs_dev[i], words_dev[i] words; 64-bit wrap-around sums with position-dependent odd
 * multipliers: exact in any order) over n_blocks workgroups of 16384 words (blk_tensor_dev / blk_off_dev: tensor and first word of each)
 * into cur_dev[n_tensors] (zero before the first launch; left zero by every launch).  baseline != 0: ref_dev := the sums, ctrl_dev[1] := 0;
 * baseline == 0: ctrl_dev[1] := 1 when some tensor's sum differs from ref_dev (sticky).  ctrl_dev: int32[2], zero before the first launch.
 * The python classes use it to notice parameters that were modified through `.data` behind their packed copies (ops.ParamGuard). */
int nm_params_fingerprint(const void* const* ptrs_dev, const long long* words_dev, const int* blk_tensor_dev, const long long* blk_off_dev,
                          int n_tensors, int n_blocks, unsigned long long* cur_dev, unsigned long long* ref_dev, int* ctrl_dev, int baseline,
                          nmStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NeRF render half
 * ---------------------------------------------------------------------------------------------- */

/* Ray bundle for the pixels (ds/2 + i*ds, ds/2 + j*ds) of an H x W image.
 * Replaces sample_nerf_rays + get_ray_dirs + get_rays_c2w + rays_intersect_sphere + prepare_rays_data
 * (nerfmatch/nerf/render_utils.py:23-41, :56-104; nerfmatch/nerf/scene_utils.py:101-120), which run on
 * the CPU at full resolution in the reference.
 *   Kinv_host   : 9 floats, inverse intrinsics (row-major)
 *   c2w_host    : 16 floats, NORMALISED camera-to-world (row-major 4x4)
 *   rays (dev)  : [R,12] = o(3) viewdir(3) near far viewdir(3) radius, R = nm_raygen_count(H,W,ds)
 *   fallback (dev): one int; set to 1 when some ray misses the unit sphere, in which case EVERY ray
 *                 gets far = 1 (the reference's assert -> except path, render_utils.py:62-68). */
int nm_raygen_count(int H, int W, int ds);
int nm_raygen(const float* Kinv_host, const float* c2w_host, int H, int W, int ds, float near_plane,
              float* rays, int* fallback, nmStream_t stream);
/* Q poses in one launch: c2w_host = Q row-major 4x4 matrices (16 floats each), rays [Q*R,12], fallback [Q].  Same arithmetic as
 * Q calls of nm_raygen (the batched renderer render_novel_views uses this). */
int nm_raygen_batch(const float* Kinv_host, const float* c2w_host, int Q, int H, int W, int ds, float near_plane,
                    float* rays, int* fallback, nmStream_t stream);

/* Stratified fence posts t[R,S+1] from rays[R,12] and t_rand[R,S+1] ~ U[0,1).
 * Replaces sample_gaus_along_rays' t_vals (nerfmatch/nerf/render_utils.py:434-445). */
int nm_sample_coarse(const float* rays, const float* t_rand, int R, int S, float* t_out, nmStream_t stream);

/* Hierarchical re-sampling: t_out[R,S+1] from t_in[R,S+1], weights[R,S], jitter[R,S+1].
 * Replaces resample_gaus_along_rays + sorted_piecewise_constant_pdf
 * (nerfmatch/nerf/render_utils.py:453-552, :583-597) including the `u + u + jitter` behaviour of the
 * randomized branch.  jitter may be NULL when randomized == 0. */
int nm_resample(const float* t_in, const float* weights, const float* jitter, int R, int S, float padding,
                int randomized, float* t_out, nmStream_t stream);
/* Same, and additionally reports whether the result has the zero-width tail NM_NERF_ZERO_TAIL relies on:
 * *zero_tail_violation (device int, may be NULL) is set to 0 before the launch and raised (non-zero) when some fence post
 * j > S/2 does not sit at u = 1 - eps (cannot happen for 0 <= jitter; always raised when randomized == 0).  Hand the same
 * pointer to nm_nerf_fwd_bf16x3_ex: the decision "skip the tail or evaluate everything" is then taken on the device, with
 * no host synchronisation and no promise by the caller. */
int nm_resample_ex(const float* t_in, const float* weights, const float* jitter, int R, int S, float padding,
                   int randomized, float* t_out, int* zero_tail_violation, nmStream_t stream);
/* The same with the jitter given UNSCALED: the kernel multiplies it by `jitter_scale` where it reads it (one fp32 product, what the reference's
 * `torch.rand_like(u) * (1 / n - eps)` computes, render_utils.py:472-476) -- saves the caller an elementwise launch. */
int nm_resample_scaled(const float* t_in, const float* weights, const float* jitter, float jitter_scale, int R, int S, float padding,
                       int randomized, float* t_out, int* zero_tail_violation, nmStream_t stream);

/* Weights of one NeRF MLP in the reference's (torch nn.Linear, [out,in]) layout, HOST pointers.
 * Keys: {nerf_coarse|nerf_fine}.{pts_linears.i, alpha_linear, feature_linear, views_linears.0, rgb_linear}
 * (nerfmatch/nerf/models/nerf.py:45-63). */
typedef struct {
  const float* pts_w[8]; /* [256,90] [256,256]x4 [256,346] [256,256]x2 */
  const float* pts_b[8]; /* [256] */
  const float* alpha_w;  /* [1,256] */
  const float* alpha_b;  /* [1] */
  const float* feat_w;   /* [256,256] */
  const float* feat_b;   /* [256] */
  const float* views_w;  /* [128, 283 + app_dim] */
  const float* views_b;  /* [128] */
  const float* rgb_w;    /* [3,128] */
  const float* rgb_b;    /* [3] */
  int app_dim;           /* 0 or 16 */
} nmNerfWeights;

/* Pack into the MFMA-operand-ordered blob nm_nerf_fwd consumes (host -> host; upload it once). */
size_t nm_nerf_blob_floats(void);
int nm_nerf_pack(const nmNerfWeights* w, float* blob_host);

enum {
  NM_NERF_SKIP_RGB = 1, /* do not evaluate the views layer (with feature_linear folded into it) and the rgb head; rgb output is not written */
  NM_NERF_FEAT_MAX = 2, /* feat/pts of the max-weight sample instead of the weighted sum (feat_comb == "max") */
  /* Premise: every interval s > S/2 of every ray has zero width (t[s+1] == t[s]) -- what nm_resample with
   * randomized = 1 produces for any jitter >= 0 (a promise of the caller with nm_nerf_fwd_bf16x3; verified on the device
   * when the flag of nm_resample_ex is handed to nm_nerf_fwd_bf16x3_ex), because the reference's `u + u + jitter` saturates at the upper half of the fence posts
   * (nerfmatch/nerf/render_utils.py:477-496).  Such samples have alpha = 0, i.e. weight exactly 0 in every output, so
   * nm_nerf_fwd_bf16x3 evaluates samples 0 .. S/2 only and writes weight 0 for the rest: same results, ~half the
   * matrix work.  Honoured for S in {64, 128} and multiples of 256, raw == sample_feat == NULL, without NM_NERF_FEAT_MAX; ignored otherwise and by
   * nm_nerf_fwd (which evaluates everything). */
  NM_NERF_ZERO_TAIL = 4
};

/* One pass (coarse or fine) of the fused conical-frustum -> IPE -> 8x256 MLP -> alpha-composite pipeline.
 * Replaces cast_rays/conical_frustum_to_gaussian/lift_gaussian (nerfmatch/nerf/render_utils.py:326-402),
 * PositionalEncodingMIP.forward (nerfmatch/nerf/embedding.py:66-84), NeRF.forward
 * (nerfmatch/nerf/models/nerf.py:94-144), the chunked forward_nerf loop (nerfmatch/nerf/renderer.py:119-180),
 * volume_render_radiance_field (nerfmatch/nerf/render_utils.py:176-230) and the weighted feature / point sums
 * of render_rays (nerfmatch/nerf/renderer.py:250-281).
 *   blob (dev)      nm_nerf_pack output                     rays (dev) [R,12]      t (dev) [R,S+1]
 *   app_row (dev)   [16] appearance embedding row or NULL   tap_layer  0..7, or -1 = last pts layer
 *   var_scale       <= 0: off (mip_var_scale)               S in {32,64,128} or a multiple of 128
 * outputs (dev; any may be NULL = not wanted, except weights):
 *   weights [R,S]  feat [R,256]  pts [R,3]  rgb [R,3]  depth [R]  acc [R]
 *   raw [R,S,4] (rgb, raw sigma per sample)   sample_feat [R,S,256] (tapped activation per sample) */
int nm_nerf_fwd(const float* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                float* rgb, float* depth, float* acc, float* raw, float* sample_feat, nmStream_t stream);
/* Guarded form: the launch does nothing unless bit 0 of run_if[0] (device int32[16], required) is set when the kernel starts --
 * the decision is taken on the device, no host synchronisation.  Used as the fall-back of nm_nerf_fwd_fp16x3_ex: hand it the
 * same outputs and that call's `status`; when an fp16 operand saturated there, this pass rewrites every output in fp32.
 * The flag is CONSUMED: after the rewrite bit 0 of run_if[0] is cleared and run_if[11] (a count of such events, sticky) goes up
 * by one; run_if[12] is scratch of this call.  NM_NERF_ZERO_TAIL is ignored (every sample is evaluated).
 * ONE STREAM PER STATUS BLOCK: launches that share a run_if / status block must not overlap (the completion counter in run_if[12]
 * and the clearing of bit 0 assume the fp16x3 launch and this one are the only users until this one has finished); concurrent renders
 * on different streams each need their own block. */
int nm_nerf_fwd_guarded(const float* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                        int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                        float* rgb, float* depth, float* acc, float* raw, float* sample_feat, int* run_if,
                        nmStream_t stream);

/* Same pass on the bf16 matrix cores with fp32-accurate operand splitting (every product = w_hi*x_hi + w_hi*x_lo +
 * w_lo*x_hi, fp32 accumulation): identical arguments and outputs, its own packed blob.  Differences to the fp32
 * oracle are < 1e-6 on the rendered features (tolerance 1e-4); see DESIGN.md section 3.1b.
 * The kernel is persistent (one workgroup per CU) and parks the tapped activations of the tile in flight in
 * `workspace` (device, nm_nerf_workspace_bytes_bf16x3() bytes, L2-resident; may be NULL when neither feat nor
 * sample_feat is requested).  One workspace per stream: calls that may overlap in time must not share it. */
size_t nm_nerf_blob_bytes_bf16x3(void);
size_t nm_nerf_workspace_bytes_bf16x3(void);
int nm_nerf_pack_bf16x3(const nmNerfWeights* w, void* blob_host);
int nm_nerf_fwd_bf16x3(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                       int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                       float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                       nmStream_t stream);
/* Same with the zero-tail decision taken on the device: `zero_tail_violation` is the flag nm_resample_ex wrote for the very
 * `t` passed here (device int, may be NULL = trust the NM_NERF_ZERO_TAIL flag as nm_nerf_fwd_bf16x3 does).  With
 * NM_NERF_ZERO_TAIL set and the flag raised the kernel evaluates every sample. */
int nm_nerf_fwd_bf16x3_ex(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                          int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                          float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                          const int* zero_tail_violation, nmStream_t stream);

/* Same pass with the operands split into fp16 hi / lo parts instead of bf16 ones (22 mantissa bits instead of 16; the three
 * products run on v_mfma_f32_32x32x16_f16 at the bf16 rate; operands beyond +-65504 saturate): fp32-class results also on
 * trained-like scenes (densities of +-1e4, opacity saturating within a few samples), where the bf16 split leaves the
 * compositing weights 7e-4 off -- tests/golden/nerf_surface_r512_s128.npz, DESIGN.md section 3.1b.  Same blob size, workspace
 * and arguments as nm_nerf_fwd_bf16x3_ex; the blob comes from nm_nerf_pack_fp16x3.  Replaces the same reference lines:
 * nerfmatch/nerf/renderer.py:119-180, nerf/models/nerf.py:94-144, nerf/render_utils.py:176-230. */
int nm_nerf_pack_fp16x3(const nmNerfWeights* w, void* blob_host);
/* Round 4 -- power-of-two operand scaling, range telemetry and a saturation flag for the fp16 split.
 * An fp16 hi/lo pair carries 22 significant bits only while its lo part is a normal fp16 number (|x| >~ 2^-3); the pack step
 * therefore multiplies every weight group by a power of two chosen from its own maximum (constants: cannot saturate) and the
 * kernel carries the hidden activations of layer l at 2^act_log2[l] times their value; the re-packing of a finished layer
 * folds the change of scale into its bias add (one fma, exact) and every output leaves the kernel in true units.
 *   act_log2 (host, 12 ints, NULL = {12, 0,...,0, 12, 0}): [0] IPE input (|x| <= 1, <= 15), [1..7] hidden input of pts layers
 *   1..7, [8] layer 7's output = input of the density head and of the views layer (feature_linear has no activation: the pack
 *   step multiplies it into the views layer's hidden columns, the kernels never run it as a layer), [9] ignored (kept for the
 *   12-int layout), [10] direction PE (<= 15), [11] appearance row.
 * nm_nerf_pack_fp16x3 == nm_nerf_pack_fp16x3_scaled(w, NULL, blob): scaled weights, activations as they are (round 3).
 * nm_nerf_fwd_fp16x3_ex = nm_nerf_fwd_fp16x3 + `status` (device int32[16], zeroed by the caller, may be NULL):
 *   status[0] |= 1   when some operand of the launch reached +-65504 (it was clamped): the results are NOT to be trusted --
 *                    launch nm_nerf_fwd_guarded(fp32 blob, same arguments, run_if = status) behind it (which clears the bit again
 *                    and counts the event in status[11]);
 *   status[1 + k]    = max over the launch of the bit pattern of |value| re-packed to fp16 in range slot k (k = 0..7: output of
 *                    pts layer k at the scale act_log2[k + 1]; k = 8: unused, stays 0; k = 9: views-layer extra inputs): divide by
 *                    2^act_log2 to get activation ranges, choose act_log2 with >= 2^4 headroom (NeRF.calibrate does). */
int nm_nerf_pack_fp16x3_scaled(const nmNerfWeights* w, const int* act_log2, void* blob_host);
int nm_nerf_fwd_fp16x3_ex(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                          int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                          float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                          const int* zero_tail_violation, int* status, nmStream_t stream);
int nm_nerf_fwd_fp16x3(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                       int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                       float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                       const int* zero_tail_violation, nmStream_t stream);

/* Pointwise forward / backward of ONE NeRF MLP on the fused kernel's K-loop machinery (round 4): the fine pass of the iNeRF
 * refinement (nerfmatch/nerfmatch_evaluator.py:348-430), where only d loss / d (ray origin, view
 * direction) is needed -- dX of every layer, no dW.  bf16 hi/lo split (three products, fp32 accumulate).
 *   forward : xi [n,96] (IPE, nm_inerf_encode), xd [n,48] -> out4 [n,4] = (rgb logits r g b, raw sigma) for nm_inerf_composite(_ex)
 *             (logit = out4, sig = out4 + 3, ld = 4) and `gates` (device, nm_nerf_points_gate_bytes(n)): one bit per ReLU
 *             activation of the nine layers -- all the backward pass needs from the forward one;  blob: nm_nerf_pack_bf16x3
 *   backward: g4 [n,4] = d loss / d (logits, sigma) (nm_inerf_composite_bwd(_ex) with ld = 4) -> g_xi0, g_xi5 [n,96]: the two
 *             contributions to d loss / d xi (through layer 0 and through the skip connection; their sum feeds
 *             nm_inerf_encode_bwd), g_xd [n,48];  blob_bwd: the transposed weights, nm_nerf_pack_bwd_bf16x3 (host -> host). */
size_t nm_nerf_blob_bytes_bwd_bf16x3(void);
size_t nm_nerf_points_gate_bytes(int n);
int nm_nerf_pack_bwd_bf16x3(const nmNerfWeights* w, void* blob_host);
int nm_nerf_points_fwd_bf16x3(const void* blob, const float* xi, const float* xd, int n, float* out4, void* gates, nmStream_t stream);
/* forward, "from rays" form: sample n = (ray n / S_act, interval n % S_act) is encoded inside the kernel (nm_inerf_encode's formulas:
 * frustum Gaussian of [z[s], z[s+1]], IPE with exact sine / exponential, direction PE, appearance row) -- no xi / xd arrays. */
int nm_nerf_points_fwd_rays_bf16x3(const void* blob, const float* rays, const float* z, int R, int S, int S_act, const float* app_row,
                                   float* out4, void* gates, nmStream_t stream);
int nm_nerf_points_bwd_bf16x3(const void* blob_bwd, const float* g4, const void* gates, int n, float* g_xi0, float* g_xi5, float* g_xd,
                              nmStream_t stream);
/* The same two passes with a TAPPED layer (round 5): the matching term of the refinement (`use_match_loss`,
 * nerfmatch/nerfmatch_evaluator.py:420-441) reads the rendered features pt_feat = sum_s w_s h_tap(s) and sends a gradient back into them.
 *   forward : additionally feats [R S_act, 256] row-major <- the post-ReLU activations of pts layer `tap_layer` (0..7): the `feats` operand of
 *             nm_inerf_ray_sums / nm_inerf_ray_sums_bwd (feats NULL and tap_layer -1: exactly nm_nerf_points_fwd_rays_bf16x3)
 *   backward: d loss / d h_tap(n) += tap_weights[n] * g_pt_feat[n / S_act][:] (product, then sum) before that layer's ReLU gate, i.e. what the
 *             GEMM chain receives as nm_inerf_ray_sums_bwd's g_feats -- which then need not exist (pass g_feats NULL there). */
int nm_nerf_points_fwd_rays_tap_bf16x3(const void* blob, const float* rays, const float* z, int R, int S, int S_act, const float* app_row,
                                       int tap_layer, float* out4, void* gates, float* feats, nmStream_t stream);
int nm_nerf_points_bwd_tap_bf16x3(const void* blob_bwd, const float* g4, const void* gates, int R, int S_act, int tap_layer,
                                  const float* tap_weights, const float* g_pt_feat, float* g_xi0, float* g_xi5, float* g_xd, nmStream_t stream);

/* Same pass with ONE fp16 MFMA per product block (operands rounded once to fp16, fp32 accumulation; its own blob with 8 KiB
 * weight slots): a third of the matrix work of the split-bf16 kernel.  Meant for the COARSE pass of render_rays when only its
 * compositing weights are consumed (they feed nothing but the resampler, render_utils.py:449-505): measured effect on the FINE
 * outputs of a render: features 3.8e-7 from the fp64 result against 3.2e-7 with the split-bf16 coarse pass (fence posts
 * 1.3e-6 against 4.8e-7) -- DESIGN.md section 3.1d; its own outputs carry ~3e-4 relative error, outside the 1e-4 class. */
size_t nm_nerf_blob_bytes_fp16x1(void);
int nm_nerf_pack_fp16x1(const nmNerfWeights* w, void* blob_host);
int nm_nerf_fwd_fp16x1(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                       int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                       float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                       const int* zero_tail_violation, nmStream_t stream);

/* pt3d[n,3] = (unnorm[4,4] . [pts,1])[:3]   (nerfmatch/utils/geometry.py:76-85); unnorm_host: 16 floats. */
int nm_unnormalize_points(const float* pts, const float* unnorm_host, int n, float* out, nmStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * iNeRF pose refinement: the differentiable fine pass of NeRFMatchEvaluator.inerf_refinement
 * (nerfmatch/nerfmatch_evaluator.py:348-430).  Forward / backward pairs around the MLP, which runs through nm_linear /
 * nm_linear_bf16x3 (forward and, with transposed weights, backward).  n = R * S_act rows, row r * S_act + s = sample s of
 * ray r; S_act <= S: only the first S_act samples of a ray are evaluated (S/2 + 1 suffices after the randomized
 * resampler: later intervals have zero width, hence zero weight and zero gradient).
 *   nm_inerf_encode      xi [n,96] = IPE(o + t_mean * viewdir, var) (90 columns + zero padding),
 *                        xd [n,48] = [dir PE (27) | appearance row (16) | zero padding]; var from the (detached) frustum
 *   nm_inerf_encode_bwd  d loss / d xi, d loss / d xd  ->  g_o [R,3] (ray origins), g_v [R,3] (rays[:, 8:11])
 *   nm_inerf_composite   rgb logits / raw sigma (column 0..2 / 0 of row-major [n, ld] buffers) -> rgb_map [R,3],
 *                        white background, delta = dz * |rays[:, 3:6]| (render_utils.py:187-230)
 *   nm_inerf_composite_bwd  G = d loss / d rgb_map -> g_logit [n,ld], g_sigma [n,ld] (unused columns zeroed),
 *                        g_d [R,3] (rays[:, 3:6], through |d|)
 * Matching term (`use_match_loss`, nerfmatch_evaluator.py:420-441): the fine weights also feed the matcher.
 *   nm_inerf_composite_ex      additionally writes the compositing weights [R, S_act] (NULL: as nm_inerf_composite)
 *   nm_inerf_ray_sums          pt_feat [R,C] = sum_s w_s feats[r S_act + s], pts [R,3] = sum_s w_s (o + t_mean d): :423-425
 *                              (the Gaussian means are the detached sampler's: constants of the backward pass)
 *   nm_inerf_ray_sums_bwd      d loss / d pt_feat [R,C], d loss / d pts [R,3] -> g_feats [n,C] (may be NULL: not written), g_weights [R, S_act]
 *   nm_inerf_composite_bwd_ex  as nm_inerf_composite_bwd with g_weights added to the weights' gradient (NULL: none)
 * ---------------------------------------------------------------------------------------------- */
int nm_inerf_encode(const float* rays, const float* z, int R, int S, int S_act, const float* app_row, float* xi, float* xd,
                    nmStream_t stream);
int nm_inerf_encode_bwd(const float* rays, const float* z, int R, int S, int S_act, const float* g_xi, const float* g_xd,
                        float* g_o, float* g_v, nmStream_t stream);
/* Same with d loss / d xi given as the sum of two arrays (nm_nerf_points_bwd_bf16x3's layer-0 and skip-connection parts). */
int nm_inerf_encode_bwd2(const float* rays, const float* z, int R, int S, int S_act, const float* g_xi_a, const float* g_xi_b,
                         const float* g_xd, float* g_o, float* g_v, nmStream_t stream);
/* d loss / d pose (row-major 4x4 on the device, homogeneous row zero) from the per-ray gradients: origin = pose[:3,3] for every
 * ray, view direction = normalise(pose[:3,:3] . Kinv . (x, y, 1)) on the ds-sub-sampled pixel grid (reference gen_rays,
 * nerfmatch/nerfmatch_evaluator.py:268-286).  g_d (may be NULL): a second gradient w.r.t. the same direction tensor (rays[:, 3:6]).
 * Kinv_host (9), pose_host (16): host, row-major. */
int nm_inerf_pose_grad(const float* Kinv_host, const float* pose_host, int H, int W, int ds, const float* g_o, const float* g_v,
                       const float* g_d, int R, float* g_pose, nmStream_t stream);
int nm_inerf_composite(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays, int R, int S,
                       int S_act, float* rgb_map, nmStream_t stream);
int nm_inerf_composite_bwd(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays,
                           const float* g_rgb_map, int R, int S, int S_act, float* g_logit, float* g_sigma, float* g_d,
                           nmStream_t stream);
int nm_inerf_composite_ex(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays, int R, int S,
                          int S_act, float* rgb_map, float* weights, nmStream_t stream);
int nm_inerf_composite_bwd_ex(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays,
                              const float* g_rgb_map, const float* g_weights, int R, int S, int S_act, float* g_logit,
                              float* g_sigma, float* g_d, nmStream_t stream);
/* The same two passes on the fused fine field's own output (round 6): out4 [n, 4] = rgb logits | raw sigma per sample, as nm_nerf_points_fwd*_bf16x3
 * writes them; g_out4 [n, 4] = d loss / d (logits, sigma), what nm_nerf_points_bwd*_bf16x3 reads.  One wavefront per ray (prefix product / suffix
 * sum over lanes), 16-byte accesses; weights / g_weights as in the _ex forms (may be NULL).  S_act <= 1024. */
int nm_inerf_composite4(const float* out4, const float* z, const float* rays, int R, int S, int S_act, float* rgb_map, float* weights,
                        nmStream_t stream);
int nm_inerf_composite4_bwd(const float* out4, const float* z, const float* rays, const float* g_rgb_map, const float* g_weights, int R, int S,
                            int S_act, float* g_out4, float* g_d, nmStream_t stream);
int nm_inerf_ray_sums(const float* weights, const float* feats, int C, const float* rays, const float* z, int R, int S, int S_act,
                      float* pt_feat, float* pts, nmStream_t stream);
int nm_inerf_ray_sums_bwd(const float* weights, const float* feats, int C, const float* rays, const float* z, const float* g_pt_feat,
                          const float* g_pts, int R, int S, int S_act, float* g_feats, float* g_weights, nmStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Matcher half
 * ---------------------------------------------------------------------------------------------- */

enum { NM_ACT_NONE = 0, NM_ACT_RELU = 1, NM_ACT_GELU = 2 };

/* y[M,N] = act(x[M,K] . w[N,K]^T + bias[N]) + residual[M,N]   (bias / residual may be NULL).
 * The nn.Linear calls of MultiHeadAttention / FeedForwardNetwork / pt_pe_proj / pt_ffeat_proj
 * (nerfmatch/modules/attention.py:101-103,114,145-147; nerfmatch/nerfmatch_c2f_trainer.py:152-160). */
int nm_linear(const float* x, const float* w, const float* bias, const float* residual, int M, int N, int K, int act,
              float* y, nmStream_t stream);

/* The same layer on the bf16 matrix cores with fp32-accurate hi/lo operand splitting (cf. nm_nerf_fwd_bf16x3).
 * The weight matrix is split and laid out once: nm_linear_pack_bf16x3(w [N,K] device, blob device of
 * nm_linear_blob_bytes_bf16x3(N, K) bytes) -- a device-side kernel, asynchronous on `stream`.
 * Requires K % 8 == 0 and N % 8 == 0 (NM_ERR_UNSUPPORTED otherwise; use nm_linear). */
/* General epilogue (both arithmetic paths): y = (act(x . w^T + bias + pre) + residual) * [gate > 0]; pre / residual / gate
 * are [M,N] or NULL.  `pre` lets a layer with a concatenated input be two GEMMs (the NeRF skip layer, the views layer);
 * `gate` applies the ReLU derivative of a saved activation in the backward GEMMs of the iNeRF refinement. */
int nm_linear_ex(const float* x, const float* w, const float* bias, const float* pre, const float* residual, const float* gate,
                 int M, int N, int K, int act, float* y, nmStream_t stream);
int nm_linear_ex_bf16x3(const float* x, const void* blob, const float* bias, const float* pre, const float* residual,
                        const float* gate, int M, int N, int K, int act, float* y, nmStream_t stream);
size_t nm_linear_blob_bytes_bf16x3(int N, int K);
/* Fused q|k|v (n_q = 32 heads) or k|v (n_q = 0) projection of an attention layer, head_dim 32, split-bf16 path: the weight
 * blob packs the [n_q + 64 heads, K] stack of proj_q / proj_k / proj_v (attention.py:150-166); the q columns are written as
 * fp32 rows q_out[M, n_q], the keys and values go straight into the operand slots of the attention kernel
 * (nm_attention_workspace_bytes(B, S, heads) bytes, B = M / S sequences of S tokens) without ever existing as fp32 rows.
 * nm_attention_presplit then runs the attention over those slots.  n_q and 32 heads multiples of 128, S of 32. */
int nm_linear_qkv_bf16x3(const float* x, const void* blob, int M, int K, int n_q, int heads, int S, float* q_out, void* kv_slots,
                         nmStream_t stream);
int nm_attention_presplit(const float* q, int ldq, const void* kv_slots, int B, int L, int S, int heads, float scale, float* out,
                          nmStream_t stream);
int nm_linear_pack_bf16x3(const float* w, int N, int K, void* blob, nmStream_t stream);
/* Round 6: the blob of the TRANSPOSE -- w is stored (K, N) (an nn.Linear weight (out = K, in = N)); the blob is that of the (N, K) matrix
 * w^T, so that nm_linear_bf16x3(dy, blob) = dy . w is the layer's input gradient (autograd's dx = dy @ W, done by torch in the reference)
 * without a transposed copy of the weight in between.  nm_linear_blob_bytes_bf16x3(N, K) bytes. */
int nm_linear_pack_t_bf16x3(const float* w, int N, int K, void* blob, nmStream_t stream);
int nm_linear_bf16x3(const float* x, const void* blob, const float* bias, const float* residual, int M, int N, int K,
                     int act, float* y, nmStream_t stream);

/* Row-wise LayerNorm over `dim` (<= 1024, multiple of 64), eps as nn.LayerNorm (1e-5).
 * (nerfmatch/modules/attention.py:196-207, :229-230, :238). */
int nm_layernorm(const float* x, const float* gamma, const float* beta, int rows, int dim, float eps, float* y,
                 nmStream_t stream);
/* Two LayerNorms of equal width in one launch (the two pre-norms of a cross-attention layer, attention.py:229-233): y0 = LN(x0; gamma0, beta0),
 * y1 = LN(x1; gamma1, beta1); row arithmetic identical to nm_layernorm (bit-identical results). */
int nm_layernorm2(const float* x0, const float* gamma0, const float* beta0, int rows0, float eps0, float* y0, const float* x1,
                  const float* gamma1, const float* beta1, int rows1, float eps1, float* y1, int dim, nmStream_t stream);

/* Softmax multi-head attention without materialising the (L,S,H) score tensor.
 * q [B,L,H*D], k,v [B,S,H*D], out [B,L,H*D]; D in {16,32}; scores are (q*scale).k.
 * Replaces FullAttention.forward / LocalitySelfAttention.forward (nerfmatch/modules/attention.py:53-57, :71-81). */
int nm_attention(const float* q, const float* k, const float* v, int B, int L, int S, int heads, int head_dim,
                 float scale, float* out, nmStream_t stream);
/* Same with explicit row strides (in floats, multiples of 4): q/k/v may be column slices of one fused projection
 * buffer, e.g. [B*L, 3*H*D] written by a single nm_linear with the concatenated proj_q|proj_k|proj_v weights. */
int nm_attention_ld(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                    int heads, int head_dim, float scale, float* out, nmStream_t stream);
/* Same with flags: NM_ATTN_BF16X3 evaluates QK^T and PV on the bf16 matrix cores with hi/lo operand splitting
 * (fp32-accurate: ~1e-6 on the tokens; head_dim 32 only, ignored for the small-sequence kernel).  That kernel streams
 * pre-split operands from a workspace, so the flag needs nm_attention_ws: nm_attention_ex returns NM_ERR_WORKSPACE for it. */
enum { NM_ATTN_BF16X3 = 1 };
int nm_attention_ex(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                    int heads, int head_dim, float scale, int flags, float* out, nmStream_t stream);
/* Same with a scratch buffer (device, nm_attention_workspace_bytes(B, S, heads) bytes; one per stream): with
 * NM_ATTN_BF16X3 and head_dim 32, K and V are split into bf16 hi/lo MFMA operands once per call into `workspace` and
 * streamed from there by LDS DMA (attention_v2.hip).  Other flags or shapes (head_dim 16, <= 64-token windows): identical to
 * nm_attention_ex without the flag; workspace == NULL where the split kernel applies: NM_ERR_WORKSPACE. */
size_t nm_attention_workspace_bytes(int B, int S, int heads);
int nm_attention_ws(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                    int heads, int head_dim, float scale, int flags, void* workspace, float* out, nmStream_t stream);

/* The part of a pre-norm encoder layer behind the attention as ONE launch:
 *     y = xh + W2 . gelu(W1 . LN2(xh + att . Wo^T) + b1) + b2          (rows x 256 everywhere, split-bf16 products)
 * replaces the tail of GenericEncoderLayer.forward_pre_norm (nerfmatch/modules/attention.py:229-241: proj_out :131-133, norm2,
 * FeedForwardNetwork :136-154) = nm_linear_bf16x3 + nm_layernorm + 2 x nm_linear_bf16x3.  att / xh / y: device [rows, 256];
 * wo_blob from nm_linear_pack_bf16x3, w1 / w2 blobs from nm_linear_pack_perm_bf16x3 (same size, K in accumulator order: the
 * three products are chained in registers); gamma2 / beta2 / b1 / b2: device [256].  dim != 256: NM_ERR_UNSUPPORTED. */
int nm_linear_pack_perm_bf16x3(const float* w, int N, int K, void* blob, nmStream_t stream);
int nm_encoder_tail_bf16x3(const float* att, const float* xh, const void* wo_blob, const void* w1_perm_blob, const void* w2_perm_blob,
                           const float* gamma2, const float* beta2, const float* b1, const float* b2, int rows, int dim, float eps,
                           float* y, nmStream_t stream);
/* Round 6: the backward of that tail for FROZEN parameters (input gradients only) as one launch -- torch autograd through the same reference
 * lines when the matcher's parameters do not require gradients (the matching term of the iNeRF refinement, nerfmatch_evaluator.py:429-441):
 *   g1 = dy . W2;  du = g1 o gelu'(u_pre);  g2 = du . W1;  d_a = LayerNorm2'(a_pre; g2);  d_xh = dy + d_a;  d_att = d_a . Wo
 * a_pre = xh + att . Wo^T and u_pre = W1 . LN2(a_pre) + b1 are the forward pass's intermediates [rows, 256]; w2t_blob = nm_linear_pack_bf16x3 of
 * W2^T, w1t_perm_blob / wot_perm_blob = nm_linear_pack_perm_bf16x3 of W1^T / Wo^T.  dim must be 256. */
int nm_encoder_tail_bwd_bf16x3(const float* dy, const float* a_pre, const float* u_pre, const void* w2t_blob, const void* w1t_perm_blob,
                               const void* wot_perm_blob, const float* gamma2, int rows, int dim, float eps, float* d_att, float* d_xh,
                               nmStream_t stream);
/* ... and the forward of the same passes as one launch that KEEPS those two intermediates: nm_encoder_tail_bf16x3 (same arithmetic, same order:
 * y is bit-identical) with a_out = xh + att . Wo^T and u_out = W1 . LN2(a) + b1 written on the way, [rows, 256] each. */
int nm_encoder_tail_save_bf16x3(const float* att, const float* xh, const void* wo_blob, const void* w1_perm_blob, const void* w2_perm_blob,
                                const float* gamma2, const float* beta2, const float* b1, const float* b2, int rows, int dim, float eps,
                                float* y, float* a_out, float* u_out, nmStream_t stream);

/* THROUGHPUT configuration (BASELINE.json config 5, "fp8 MFMA attention"), head_dim 32: both contractions with ONE
 * v_mfma_f32_32x32x16_fp8_fp8 per product block on OCP e4m3 operands -- keys / values scaled per (batch, head), queries per
 * query, probabilities by 2^8 after the shift by the running maximum (all powers of two); fp32 accumulation and row sums.
 * NOT a parity arithmetic: 3 mantissa bits; the error against nm_attention is reported by bench.py (variants.attention_fp8)
 * and bounded by tests/test_matcher_gpu.py::test_attention_fp8_error_bound.  Approximates FullAttention.forward
 * (nerfmatch/modules/attention.py:44-57).  workspace: device, nm_attention_fp8_workspace_bytes(B, S, heads) bytes. */
size_t nm_attention_fp8_workspace_bytes(int B, int S, int heads);
int nm_attention_fp8(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S, int heads,
                     float scale, void* workspace, float* out, nmStream_t stream);

/* tokens y[B, h*w, C] = transpose(cfeat x[B,C,h,w]) (+ pe_table[C,table_h,table_w][:, :h, :w] when pe_table != NULL).
 * Replaces flatten/permute + PositionEncodingSine.forward + rearrange (nerfmatch_c2f_trainer.py:240,249-252;
 * third_party/loftr/position_encoding.py:45-50). */
int nm_add_sine_pe(const float* x, const float* pe_table, int B, int h, int w, int C, int table_h, int table_w,
                   float* y, nmStream_t stream);

/* The reference's encoding MODULES as callable entry points (its evaluator reaches into renderer.xyz_encoder / dirs_encoder directly,
 * nerfmatch/nerfmatch_evaluator.py:385-393).  The fused kernels never call these: they produce the same encodings as MFMA operands.
 *   nm_mip_encode: PositionalEncodingMIP.forward(x, y) (nerfmatch/nerf/embedding.py:66-84) for x, y [n, D], scales 2^min_deg .. 2^(min_deg +
 *     num_freqs - 1).  y != NULL (integrated PE): x_ret [n, 2*num_freqs*D] = exp(-y_enc/2) sin(x_enc) and, when y_ret != NULL,
 *     y_ret = max(0, (1 - exp(-2 y_enc) cos(2 x_enc))/2 - x_ret^2); column = part*num_freqs*D + scale*D + axis, part 1 = phase + fl32(pi/2).
 *     y == NULL (plain PE): x_ret [n, 2*num_freqs*D + D] = [sin(x_enc) | x].
 *     arith 0: expf + fp64-reduced sine (correctly rounded class; the arithmetic of nm_nerf_fwd, the iNeRF kernels and nm_inerf_encode);
 *     arith 1: the exp2 / fp32 Cody-Waite sine device functions the split render kernels (nm_nerf_fwd_fp16x3 / _bf16x3) inline for x_ret.
 *   nm_fourier_embed: FourierEmbedding.forward(x) (embedding.py:35-46, log-scale, scale 1): out [n, D + 2*num_freqs*D] =
 *     [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...]. */
int nm_mip_encode(const float* x, const float* y, size_t n, int D, int min_deg, int num_freqs, int arith, float* x_ret, float* y_ret,
                  nmStream_t stream);
int nm_fourier_embed(const float* x, size_t n, int D, int num_freqs, float* out, nmStream_t stream);

/* feature_normalization of NeRFMatcherCoarse's `pt_feat_norm` option (nerfmatch/nerfmatch_coarse_trainer.py:42-47, called on pt_feat and
 * pt3d at :198-200): per set b of x[B,N,D], centroid = mean over the N rows; x -= centroid IN PLACE (the reference's `x -= ...` changes the
 * caller's tensor as well); y[B,N,D] = x / max_r ||x_r||_2.  D <= 1024. */
int nm_feature_normalize(float* x, int B, int N, int D, float* y, nmStream_t stream);

/* out[n, C + 3 + 6*num_freqs] = [feat[n,C] | x | sin(2^0 x) cos(2^0 x) sin(2^1 x) ...]
 * (FourierEmbedding.forward nerfmatch/nerf/embedding.py:35-46 + the cat of cat_pe, nerfmatch_c2f_trainer.py:258-261). */
int nm_cat_fourier(const float* feat, const float* pt3d, int n, int C, int num_freqs, float* out, nmStream_t stream);
/* d loss / d pt3d [n,3] from dy [n, ld] (ld as above, padded to a multiple of 8): the gradient the iNeRF matching term sends
 * to the rendered points (autograd of the same embedding in the reference). */
int nm_cat_fourier_bwd(const float* dy, const float* pt3d, int n, int C, int num_freqs, float* g_pt3d, nmStream_t stream);

/* The same matching for a BATCH of P pairs when the confidence matrix itself is not wanted (inference): the similarity /
 * confidence values exist only in accumulator registers -- two passes of the 128 x 128 tile GEMM on the split-bf16 matrix
 * cores (sums of exp(sim - |scale|), then conf + per-tile row maxima / first columns + column maxima), an exact tie pass for
 * rows whose maximum is attained more than once, ordered compaction -- seven launches per batch instead of nine per pair
 * (csrc/match_fused.hip).  Same reference lines as nm_dual_softmax_match: nerfmatch_c2f_trainer.py:289-300,
 * modules/extract_matches.py:21-36.
 *   im [P,M,C], pt [P,N,C], im_mask [P,M] / pt_mask [P,N] uint8 or NULL; out_i / out_j / out_conf [P,M] (valid prefix:
 *   counts[p]; the slots behind it are written as index 0 / confidence 0); C in {64,128,256,512}; |scale| log2(e) <= 60 (cosine similarities are bounded by |scale|: one fixed shift
 *   serves both soft-maxes) -- otherwise NM_ERR_UNSUPPORTED: use nm_dual_softmax_match_ex per pair. */
size_t nm_match_fused_workspace_bytes(int P, int M, int N, int C);
int nm_dual_softmax_match_fused(const float* im, const float* pt, int P, int M, int N, int C, float scale, const uint8_t* im_mask,
                                const uint8_t* pt_mask, float threshold, int mutual, int64_t* out_i, int64_t* out_j,
                                float* out_conf, int* counts, void* workspace, size_t workspace_bytes, nmStream_t stream);

/* Dual-softmax matching + (mutual) nearest-neighbour selection for ONE image/point-set pair.
 * Replaces coarse_matching (nerfmatch/nerfmatch_c2f_trainer.py:289-300) + extract_mutual_matches inference branch
 * (nerfmatch/modules/extract_matches.py:21-36).
 *   im [M,C], pt [N,C] raw token features (normalised inside as f/(|f|+1e-6))
 *   scale      multiplies the cosine similarity (temperature for temp_type "mul", 1/temperature for "div")
 *   im_mask[M], pt_mask[N] : uint8 0/1 or NULL
 *   conf [M,N] or NULL (not materialised in HBM when NULL... see DESIGN.md)
 *   im_norm [M,C], pt_norm[N,C] or NULL: the normalised features (ret_feats)
 *   out_i[M], out_j[M] int64, out_conf[M] f32: compacted matches sorted by i; count (dev int): number of matches
 *   workspace: nm_match_workspace_bytes(M,N,C) bytes of device scratch */
size_t nm_match_workspace_bytes(int M, int N, int C);
int nm_dual_softmax_match(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                          const uint8_t* pt_mask, float threshold, int mutual, float* conf, float* im_norm,
                          float* pt_norm, int64_t* out_i, int64_t* out_j, float* out_conf, int* count, void* workspace,
                          size_t workspace_bytes, nmStream_t stream);
/* Same with flags: NM_MATCH_BF16X3 computes the similarity matrix on the split-bf16 matrix-core path (cf.
 * nm_linear_bf16x3; needs N % 8 == 0, otherwise the fp32 path is taken).  The softmax sweeps and the equality tests of the
 * selection are unchanged.  NM_MATCH_STATS_ONLY (round 6) stops behind the similarity matrix and the four soft-max statistics -- what
 * nm_match_focal_loss / nm_match_focal_loss_bwd read from the workspace: the loss of iNeRF's matching term consumes neither the confidence
 * matrix nor a match list (nerfmatch_evaluator.py:429-441) --; conf / out_i / out_j / out_conf / count may be NULL and are not written. */
enum { NM_MATCH_BF16X3 = 1, NM_MATCH_STATS_ONLY = 2 };
int nm_dual_softmax_match_ex(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                             const uint8_t* pt_mask, float threshold, int mutual, int flags, float* conf, float* im_norm,
                             float* pt_norm, int64_t* out_i, int64_t* out_j, float* out_conf, int* count, void* workspace,
                             size_t workspace_bytes, nmStream_t stream);

/* 5x5 (win x win) windows, stride 4, zero padding win/2, of the fine map ffeat[C,Hf,Wf] gathered at coarse cells
 * i_ids[K] (row-major over (Hf/4, Wf/4)): out[K, win*win, C].
 * Replaces F.unfold + rearrange + gather (third_party/loftr/fine_matching.py:46-55) without the full unfold. */
int nm_fine_windows(const float* ffeat, int C, int Hf, int Wf, const int64_t* i_ids, const int* count, int max_k,
                    int win, int stride, float* out, nmStream_t stream);

/* Batched form: ffeat[B,C,Hf,Wf], match k reads map map_ids[k] (int64, < B).  One launch for the matches of a whole batch. */
int nm_fine_windows_batch(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids,
                          const int* count, int max_k, int win, int stride, float* out, nmStream_t stream);

/* Match assembly of one image / point-set pair (nerfmatch_c2f_trainer.py:457-483) as one launch over K match slots:
 * mpt2d_c[k] = pt2d[i_ids[k]] ([M,2]), mpt3d[k] = pt3d[j_ids[k]] ([N,3]), mpt2d_f[k] = mpt2d_c[k] + expec_f[k,:2] * win / 2 * fine_ds
 * (the reference's expression, every intermediate rounded to fp32), pred_mask[k] = (mconf[k] != 0) as bytes.  All K slots are computed:
 * ids must be valid indices in every slot (the match lists of this library are zero-initialised behind their count). */
int nm_assemble_matches(const float* pt2d, const float* pt3d, const int64_t* i_ids, const int64_t* j_ids, const float* expec_f,
                        const float* mconf, int K, float win, float fine_ds, float* mpt2d_c, float* mpt2d_f, float* mpt3d,
                        uint8_t* pred_mask, nmStream_t stream);

/* rows gather: out[k,:] = src[ids[k],:] for k < *count. */
int nm_gather_rows(const float* src, const int64_t* ids, const int* count, int max_k, int dim, float* out,
                   nmStream_t stream);
/* Image side of the fine stage in one launch on the matrix cores (round 5): the 5 x 5 window of every match (FinePreprocess,
 * third_party/loftr/fine_matching.py:58-71) through ONE pre-norm self-attention encoder layer of width 128 with 8 heads of 16 (`fine_sa`:
 * GenericEncoderLayer.forward_pre_norm, nerfmatch/modules/attention.py:229-241; bias-free q / k / v / out projections, exact-erf GELU
 * feed-forward with biases):
 *   out [max_k, 25, 128] = xh + W2 gelu(W1 LN2(xh + MHA(xh)) + b1) + b2,  xh = LN1(window)
 * for the first min(*count, max_k) matches (the rest stays unwritten).  ffeat [B, 128, Hf, Wf]; map_ids / i_ids as nm_fine_windows_batch.
 * The six 128 x 128 weight matrices as nm_linear_pack_perm_bf16x3 blobs (64 KiB each); split-bf16 products, fp32 accumulate.
 * With pt_f [max_k, 128] (the point-side fine features, nm_fine_pt_proj) and expec_f [max_k, 3]: FineMatching's expectation of every match
 * (nm_fine_expectation's arithmetic on the layer's output) is written as well; `out` may then be NULL (the layer's output is not stored).
 * Other shapes: NM_ERR_UNSUPPORTED (use nm_fine_windows_batch + the generic layer kernels). */
int nm_fine_window_layer(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids, const int* count,
                         int max_k, int win, int stride, int heads, const float* ln1_gamma, const float* ln1_beta, float ln1_eps,
                         const void* wq_perm, const void* wk_perm, const void* wv_perm, const void* wo_perm, const float* ln2_gamma,
                         const float* ln2_beta, float ln2_eps, const void* w1_perm, const float* b1, const void* w2_perm, const float* b2,
                         float scale, float* out, const float* pt_f, float* expec_f, nmStream_t stream);
/* The WHOLE fine stage of the matches in one launch: nm_fine_window_layer with the point side computed inside as well -- pt_f[k] = W1 (W0
 * pt_src[pt_ids[k]] + b0) + b1 (nm_fine_pt_proj's arithmetic; pt_src [rows, pt_c0], transposed weights pt_w0t [pt_c0, 128], pt_w1t [128, 128],
 * biases may be NULL) instead of being read from `pt_f` (give one of the two).  reference: pt_ffeat_proj + FinePreprocess + fine_sa +
 * FineMatching, nerfmatch_c2f_trainer.py:344-350. */
int nm_fine_stage(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids, const int* count, int max_k,
                  int win, int stride, int heads, const float* ln1_gamma, const float* ln1_beta, float ln1_eps, const void* wq_perm,
                  const void* wk_perm, const void* wv_perm, const void* wo_perm, const float* ln2_gamma, const float* ln2_beta, float ln2_eps,
                  const void* w1_perm, const float* b1, const void* w2_perm, const float* b2, float scale, float* out, const float* pt_f,
                  const float* pt_src, const int64_t* pt_ids, int pt_c0, const float* pt_w0t, const float* pt_b0, const float* pt_w1t,
                  const float* pt_b1, float* expec_f, nmStream_t stream);
/* Point side of the fine stage in one launch (round 5): out[k, :C1] = W1 (W0 src[ids[k]] + b0) + b1 for the first min(*count, max_k) slots, zeros
 * for the rest -- `pt_ffeat_proj` (two Linear layers, no activation between) on the matched points' coarse tokens,
 * nerfmatch/nerfmatch_c2f_trainer.py:344-346.  w0t [C0, C1], w1t [C1, C1]: the TRANSPOSED weights (row k = the weights of input k); biases may be
 * NULL.  fp32 FMAs in K order.  C1 = 128, C0 a multiple of 4 up to 512; other shapes: NM_ERR_UNSUPPORTED (use nm_gather_rows + nm_linear). */
int nm_fine_pt_proj(const float* src, const int64_t* ids, const int* count, int max_k, int C0, int C1, const float* w0t, const float* b0,
                    const float* w1t, const float* b1, float* out, nmStream_t stream);

/* expec_f[K,3] = (E[x], E[y], std) of softmax(<pt_f[k], win_f[k,r]> / sqrt(C)) over the win x win window.
 * Replaces FineMatching.forward (third_party/loftr/fine_matching.py:88-121). */
int nm_fine_expectation(const float* pt_f, const float* win_f, const int* count, int max_k, int win, int C,
                        float* expec_f, nmStream_t stream);

/* ---- training side of the matcher head ---------------------------------------------------
 * The reference trains through torch autograd (NeRFMatcherMS.forward_with_metrics, nerfmatch_c2f_trainer.py:490-551); these
 * are the backward passes of the layers above, called from nerfmatch_amd/autograd.py.  All fp32. */

/* nn.Linear weight gradient dw[N,K] (+)= dy[M,N]^T . x[M,K] (fp32 matrix cores; row slices summed in a fixed order).
 * workspace: nm_linear_wgrad_workspace_bytes(M,N,K) bytes (may be NULL when the whole of M fits one slice and
 * accumulate == 0: NM_ERR_WORKSPACE is returned when it was needed). */
size_t nm_linear_wgrad_workspace_bytes(int M, int N, int K);
int nm_linear_wgrad(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, void* workspace,
                    size_t workspace_bytes, nmStream_t stream);
/* The same product on the bf16 matrix cores with hi/lo operand splitting (round 6; the arithmetic of nm_linear_bf16x3, i.e. of the dX GEMMs of
 * the same backward pass): N even, K a multiple of 4; same workspace function.  dW of nn.Linear under loss.backward() with the split arithmetic selected. */
int nm_linear_wgrad_bf16x3(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, void* workspace,
                           size_t workspace_bytes, nmStream_t stream);
/* ... and the bias gradient db [N] = column sums of dy from the same launch (dy is read once for both; accumulate applies to dw and db alike). */
int nm_linear_wgrad_bias_bf16x3(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, float* db, void* workspace,
                                size_t workspace_bytes, nmStream_t stream);
/* bias gradient out[N] (+)= sum_m dy[m,:] (float atomics: order-dependent in the last bits). */
int nm_col_sum(const float* dy, int M, int N, int accumulate, float* out, nmStream_t stream);
/* exact-erf GELU (nn.GELU(), modules/attention.py:136-154) as a separate pass over the pre-activations u (training keeps
 * u for the backward pass), and du = dh * gelu'(u).  n % 4 == 0. */
int nm_gelu(const float* u, size_t n, float* h, nmStream_t stream);
int nm_gelu_bwd(const float* u, const float* dh, size_t n, float* du, nmStream_t stream);
/* nn.ReLU backward (FeedForwardNetwork with act_fn "relu", nerfmatch/modules/attention.py:136-154): du = dh where the forward's OUTPUT h > 0, else 0.
 * n % 4 == 0.  (The forward is nm_linear's fused NM_ACT_RELU.) */
int nm_relu_bwd(const float* h, const float* dh, size_t n, float* du, nmStream_t stream);
/* nn.LayerNorm backward: dx[rows,dim]; dgamma[dim] and dbeta[dim] are ADDED onto (zero them first) -- or both NULL: input gradient only
 * (frozen parameters: the matching term of the iNeRF refinement, nerfmatch_evaluator.py:429-441).  dim in {64,128,256,512}. */
int nm_layernorm_bwd(const float* x, const float* gamma, const float* dy, int rows, int dim, float eps, float* dx,
                     float* dgamma, float* dbeta, nmStream_t stream);
/* backward of y = f / (|f| + 1e-6) (coarse_matching, nerfmatch_c2f_trainer.py:290-291). dim in {64,128,256,512}. */
int nm_l2norm_bwd(const float* f, const float* dy, int rows, int dim, float* df, nmStream_t stream);

/* Backward of softmax attention (autograd through FullAttention.forward, modules/attention.py:44-57).  q/k/v/o/d_o and the
 * three gradients are row-pitched like nm_attention_ld.  head_dim 32: flash-style recomputation on the fp32 matrix cores
 * (workspace: nm_attention_bwd_workspace_bytes(B,L,S,heads,flags)); flags & NM_ATTN_BF16X3: the same contractions on the
 * bf16 matrix cores with hi/lo operand splitting (attention_bwd_v2.hip; operands pre-split into the workspace);
 * head_dim 16 with L,S <= 64 (fine windows): one thread per row. */
size_t nm_attention_bwd_workspace_bytes(int B, int L, int S, int heads, int flags);
int nm_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk,
                     int ldv, int ldo, int lddo, int B, int L, int S, int heads, int head_dim, float scale, float* dq,
                     float* dk, float* dv, int lddq, int lddk, int lddv, int flags, void* workspace, size_t workspace_bytes,
                     nmStream_t stream);
/* Round 6: the forward pass keeps what the backward pass needs.  nm_attention_ws_lse = nm_attention_ws on the split-bf16 kernel (flags must
 * hold NM_ATTN_BF16X3, head_dim 32, not the <= 64-token window shapes: otherwise NM_ERR_UNSUPPORTED) that also writes nlse_out[B][heads][L] =
 * -(log-sum-exp of the query's scaled scores) in the log2 domain; nm_attention_bwd_lse = nm_attention_bwd given that array (NULL: as
 * nm_attention_bwd): the dQ kernel then makes ONE pass over the keys instead of two (7 instead of 8 tile products per key / query tile pair). */
int nm_attention_ws_lse(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S, int heads,
                        int head_dim, float scale, int flags, void* workspace, float* out, float* nlse_out, nmStream_t stream);
int nm_attention_bwd_lse(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk, int ldv,
                         int ldo, int lddo, int B, int L, int S, int heads, int head_dim, float scale, float* dq, float* dk, float* dv,
                         int lddq, int lddk, int lddv, int flags, const float* nlse, void* workspace, size_t workspace_bytes,
                         nmStream_t stream);

/* Backward of nm_fine_windows (scatter-add of d out[K,win*win,C] into dffeat[C,Hf,Wf], which the caller zeroes or accumulates
 * onto; float atomics) and of nm_fine_expectation (d_expec[K,3] -> d_pt[K,C], d_win[K,win*win,C]); autograd through
 * third_party/loftr/fine_matching.py:46-55 and :88-121. */
int nm_fine_windows_bwd(const float* dwin, int C, int Hf, int Wf, const int64_t* i_ids, const int* count, int max_k, int win,
                        int stride, float* dffeat, nmStream_t stream);
int nm_fine_expectation_bwd(const float* pt_f, const float* win_f, const float* d_expec, const int* count, int max_k, int win,
                            int C, float* d_pt, float* d_win, nmStream_t stream);

/* Focal loss on the dual-softmax confidence (compute_matching_loss, nerfmatch/utils/metrics.py:372-380) and its gradient.
 * Protocol per training step:  zero acc[4] (double: sum_pos, sum_neg, n_pos, n_neg);  nm_focal_count over the WHOLE batch's
 * conf_gt (uint8 0/1, other values ignored);  then per batch element, right after nm_dual_softmax_match(_ex) on `workspace`
 * (which still holds the similarity matrix and the soft-max statistics):  nm_match_focal_loss adds the element's loss sums to
 * acc and writes row_t[M] / col_t[N];  loss = acc[0]/acc[2] + acc[1]/acc[3].  nm_match_focal_loss_bwd (same workspace
 * contents) writes ddot[M,N] = grad_loss * d loss / d (im_n . pt_n) and adds d loss / d scale to *dscale (double, may be
 * NULL); grad_loss is a device scalar (NULL = 1).  clamp != 0: conf is clamped to [1e-6, 1 - 1e-6] first (c2f model);
 * clamp == 0: not (NeRFMatcherCoarse.forward_with_metrics, nerfmatch_coarse_trainer.py:380). */
int nm_focal_count(const uint8_t* conf_gt, size_t total, double* acc, nmStream_t stream);
int nm_match_focal_loss(const uint8_t* conf_gt, int M, int N, int C, float alpha, float gamma, int clamp, void* workspace,
                        size_t workspace_bytes, double* acc, float* row_t, float* col_t, nmStream_t stream);
int nm_match_focal_loss_bwd(const uint8_t* conf_gt, const uint8_t* im_mask, const uint8_t* pt_mask, int M, int N, int C,
                            float alpha, float gamma, int clamp, float scale, const float* grad_loss, void* workspace,
                            size_t workspace_bytes, const double* acc, const float* row_t, const float* col_t, float* ddot,
                            double* dscale, nmStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFMATCH_AMD_H */
