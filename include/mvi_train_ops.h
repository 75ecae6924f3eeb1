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

const char* mvi_train_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MVI_TRAIN_OPS_H */
