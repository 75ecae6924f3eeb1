/*
 * mvi_train_ops.h — C-ABI of the MI355X (gfx950) device ops either side of the rasterizer inside the timed region of
 * the 3DGS training loop (SURVEY.md §8f "next" rows; gs-simp/train.py:67-95 iter_start ... iter_end).
 *
 * Conventions as in mvi_raster.h: every pointer is a DEVICE pointer unless named *_host, fp32 contiguous; `stream`
 * is a hipStream_t passed as void*; nothing synchronises; the library owns no memory. Returns 0 or a negative
 * MVI_E* code (mvi_raster.h); mvi_train_last_error() gives the message.
 */
#ifndef MVI_TRAIN_OPS_H
#define MVI_TRAIN_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM(x, y))) with x = image * weight, y = gt * weight
 * (weight [H,W] optional: NULL = 1; the reference's masked variant passes 1 - gt_mask), SSIM with the 11x11 Gaussian
 * window (sigma 1.5), zero padding, C1 = 0.01^2, C2 = 0.03^2 — gs-simp/utils/loss_utils.py:17-18, :23-62 as called
 * at gs-simp/train.py:90-92 and gs-simp/inpaint_rec.py:117-123. image, gt: [3,H,W].
 * out3 (device, 3 floats): loss, mean|x - y|, mean SSIM. dL_dimage [3,H,W] (optional, overwritten) =
 * upstream * d loss / d image. Workspace: mvi_photometric_loss_workspace_bytes(H, W). The loss value is summed in a
 * fixed order (bit-reproducible). */
size_t mvi_photometric_loss_workspace_bytes(int32_t H, int32_t W);
int mvi_photometric_loss(const float* image, const float* gt, const float* weight, int32_t H, int32_t W,
                         float lambda_dssim, float upstream, float* out3, float* dL_dimage, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The same loss for a caller that COMBINES the two means itself — the reference's own expression
 * (1 - l) * l1_loss(image, gt) + l * (1 - ssim(image, gt)) (gs-simp/train.py:91-92, utils/loss_utils.py:17-18, :33-41) under autograd:
 * mvi_photometric_loss_stats is the forward (out3 as above with lambda = 0: out3[0] = out3[1] = mean|x - y|, out3[2] = mean SSIM) and
 * leaves the SSIM derivative maps in `workspace`, which the caller keeps untouched until the backward;
 * mvi_photometric_loss_grad2 then writes dL_dimage = weights2[0] * d mean|x - y| / d image + weights2[1] * d mean SSIM / d image, with
 * weights2 two DEVICE floats (autograd's upstream gradients: not read back). One statistics pass and one gradient pass for the
 * pair, as in the fused call. */
int mvi_photometric_loss_stats(const float* image, const float* gt, const float* weight, int32_t H, int32_t W, float* out3,
                               void* workspace, size_t workspace_bytes, void* stream);
int mvi_photometric_loss_grad2(const float* image, const float* gt, const float* weight, int32_t H, int32_t W,
                               const float* weights2, float* dL_dimage, void* workspace, size_t workspace_bytes, void* stream);

/* torch.optim.Adam (weight_decay 0, amsgrad off) over up to MVI_ADAM_MAX_GROUPS tensors in one launch — the optimizer
 * of gs-simp/scene/gaussian_model.py:154-163 (six groups, per-group lr, eps 1e-15), stepped at gs-simp/train.py:126-128.
 * `groups_host` is a HOST array (copied into the kernel arguments); every tensor pointer in it is a device pointer to
 * n fp32 elements; param / exp_avg / exp_avg_sq are updated in place. `step` is the 1-based step count AFTER this
 * update (state["step"] + 1). Same operation order as torch's _single_tensor_adam; betas and eps are doubles because
 * torch derives 1 - beta and the bias corrections from Python floats before rounding to fp32. */
#define MVI_ADAM_MAX_GROUPS 8
typedef struct mvi_adam_group {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    float lr;
} mvi_adam_group;
int mvi_adam_step(const mvi_adam_group* groups_host, int32_t n_groups, double beta1, double beta2, double eps,
                  int32_t step, void* stream);

/* The activated views of the Gaussian parameters the renderer consumes, in one launch
 * (gs-simp/scene/gaussian_model.py:95-115; setup_functions :44-59): scales [P,3] = exp(raw_scaling),
 * rotations [P,4] = normalize(raw_rotation) (F.normalize, eps 1e-12), opacities [P,1] = sigmoid(raw_opacity),
 * shs [P,M,3] = cat(features_dc [P,1,3], features_rest [P,M-1,3], dim 1); and their chain rule in one launch. */
int mvi_gaussian_activations(int32_t P, int32_t M, const float* raw_scaling, const float* raw_rotation,
                             const float* raw_opacity, const float* features_dc, const float* features_rest,
                             float* scales, float* rotations, float* opacities, float* shs, void* stream);
int mvi_gaussian_activations_backward(int32_t P, int32_t M, const float* raw_rotation, const float* scales,
                                      const float* opacities, const float* dL_dscales, const float* dL_drotations,
                                      const float* dL_dopacities, const float* dL_dshs, float* dL_draw_scaling,
                                      float* dL_draw_rotation, float* dL_draw_opacity, float* dL_dfeatures_dc,
                                      float* dL_dfeatures_rest, void* stream);

/* simple_knn._C.distCUDA2 (gs-simp/scene/gaussian_model.py:20, :134, :546, :623; third-party plug-in, absent from the
 * reference tree): mean_dist2[i] = mean of the squared distances from points[i] to its 3 nearest OTHER points
 * (exact; FLT_MAX terms when fewer than 3 exist). points [N,3] fp32, mean_dist2 [N]. Initialisation / densification
 * only — exact tiled brute force, O(N^2). */
int mvi_knn3_mean_dist2(const float* points, int32_t N, float* mean_dist2, void* stream);

/* Row compaction of many tensors by ONE keep-mask — prune_points / _prune_optimizer
 * (gs-simp/scene/gaussian_model.py:351-382), where the reference runs `t[mask]` on 6 parameters, 12 Adam moments and 3
 * per-Gaussian statistics one at a time. mvi_compact_plan scans keep_mask [P] (uint8, nonzero = keep) into the list of
 * kept source rows inside `workspace` (mvi_compact_workspace_bytes(P), 256-byte aligned) and writes the kept count to
 * n_keep_device (optional; the caller reads it back to size the outputs). mvi_compact_gather then copies, for every
 * table entry, out[j, :] = in[src_row(j), :] for j < n_keep; rows are `width` 4-byte words (any 32-bit element type),
 * 1 <= width <= 8192. Results equal boolean indexing bit for bit. `tensors_host` is a HOST array. */
#define MVI_COMPACT_MAX_TENSORS 24
typedef struct mvi_compact_tensor {
    const void* in;   /* [P, width] */
    void* out;        /* [n_keep, width] */
    int32_t width;
    int32_t packed_stride;   /* window forms only: words between consecutive rows of the COMPACT side (the gather's out, the scatter's
                              * in), >= width; 0 = width (contiguous rows). Lets one compact array [capacity, 3 M] be scattered into
                              * two full-size tensors (features_dc [P, 3] | features_rest [P, 3 M - 3]) by two entries whose `in`
                              * pointers are 0 and 3 words into its rows. mvi_compact_gather ignores it (its outputs are contiguous). */
} mvi_compact_tensor;
size_t mvi_compact_workspace_bytes(int32_t P);
int mvi_compact_plan(const uint8_t* keep_mask, int32_t P, void* workspace, size_t workspace_bytes,
                     uint32_t* n_keep_device, void* stream);
int mvi_compact_gather(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P, uint32_t n_keep,
                       const void* workspace, void* stream);

/* The gather and its inverse over a WINDOW of the plan's row list whose length only the device knows (view-parallel training,
 * multiview_inpaint_amd/dist.py: the rows of the union of the ranks' gradient supports travel in buffers whose capacity the
 * host chose from the previous step, so nothing waits for this step's count). With n = min(capacity, *n_keep_device - first)
 * (0 when first >= *n_keep_device): gather  out[j, :] = in[src_row(first + j), :], in [P, width], out [capacity, width];
 * scatter out[src_row(first + j), :] = in[j, :], in [capacity, width] (NULL: zeros), out [P, width]; j < n. Other rows of
 * `out` are not touched. No counterpart in the reference (single GPU, gs-simp/train.sh:1). */
int mvi_compact_gather_window(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P,
                              const uint32_t* n_keep_device, uint32_t first, uint32_t capacity, const void* workspace,
                              void* stream);
int mvi_compact_scatter_window(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P,
                               const uint32_t* n_keep_device, uint32_t first, uint32_t capacity, const void* workspace,
                               void* stream);

/* Gradient supports as bit masks, for the one small collective in front of the compacted exchange: pack bits[w] bit b =
 * (flags[32 w + b] != 0) for the (P + 31) / 32 words of one rank's support; union: mask[i] = 1 if any of the n_ranks gathered bit
 * arrays bits_all [n_ranks][(P + 31) / 32] has bit i set, else 0 (the keep-mask mvi_compact_plan takes). P / 8 bytes per rank in
 * ONE all-gather instead of a P-byte all-reduce(MAX). No counterpart in the reference (single GPU). */
int mvi_support_pack_bits(const uint8_t* flags, int32_t P, uint32_t* bits, void* stream);
int mvi_support_union_bits(const uint32_t* bits_all, int32_t n_ranks, int32_t P, uint8_t* mask, void* stream);

const char* mvi_train_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MVI_TRAIN_OPS_H */
