/*
 * mvi_unet_ops.h — C-ABI of the MI355X (gfx950) device ops of the SVD temporal-UNet denoise loop.
 *
 * Drop-in boundary: the reference's denoise loop is PyTorch modules whose only hand-optimised
 * device ops are reached through
 *   - xformers.ops.memory_efficient_attention      svd_inpaint1/sgm/modules/attention.py:427-439
 *   - F.scaled_dot_product_attention               svd_inpaint1/sgm/modules/attention.py:332-336
 *   - GroupNorm32 (+ SiLU), Normalize              svd_inpaint1/sgm/modules/diffusionmodules/util.py:259-276,
 *                                                  openaimodel.py:257-261,292-305, attention.py:125-128
 * These entry points replace those calls; the Python side that binds them with ctypes is
 * multiview_inpaint_amd/svd/hip_ops.py behind the same module classes.
 *
 * Conventions: every pointer is a DEVICE pointer; `stream` is a hipStream_t passed as void*;
 * nothing synchronises; the library owns no memory. Returns 0 or a negative MVI_E* code
 * (include/mvi_raster.h); mvi_unet_last_error() gives the message.
 */
#ifndef MVI_UNET_OPS_H
#define MVI_UNET_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVI_DT_F32 0
#define MVI_DT_BF16 1
#define MVI_DT_F16 2

/* GroupNorm over groups of C/groups channels x `spatial` positions of x [N, C, spatial] (contiguous;
 * spatial = H*W or T*H*W), statistics and affine in fp32, optional fused SiLU, y has x's dtype.
 * weight/bias: fp32 [C]. workspace: mvi_groupnorm_workspace_bytes(...) bytes. */
size_t mvi_groupnorm_workspace_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups);
int mvi_groupnorm_silu(const void* x, void* y, const float* weight, const float* bias, int64_t N,
                       int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                       int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* The same GroupNorm for the temporal layers: statistics over (C/groups, T, spatial) of a tensor stored
 * [(videos*T), C, spatial] — i.e. GroupNorm of "b c t h w" (svd_inpaint1/sgm/modules/diffusionmodules/
 * video_model.py:71-75, openaimodel.py:257-261 with dims=3) computed on the "(b t) c h w" layout the
 * spatial layers produce, without the permute copies. Workspace: mvi_groupnorm_workspace_bytes(videos*T, ...). */
int mvi_groupnorm_silu_temporal(const void* x, void* y, const float* weight, const float* bias,
                                int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups,
                                float eps, int32_t fuse_silu, int32_t dtype, void* workspace,
                                size_t workspace_bytes, void* stream);

/* General form. chan_bias (nullable): fp32 [(videos*T), C] added to x before the statistics — the timestep-
 * embedding bias a ResBlock adds in front of its second norm (openaimodel.py:341-352), fused so the sum is never
 * written. stack3 != 0 (requires the temporal layout; y must not alias x): y is [(videos*T), 3C, spatial] and
 * receives the normalised frame t at channel block 1 of row t, block 0 of row t+1 and block 2 of row t-1, with
 * zero frames at the sequence ends — exactly the input a kernel-(3,1,1) temporal convolution needs when it is
 * evaluated as one 1x1 convolution over 3C channels. T = 1 gives the plain [N, C, spatial] norm. */
int mvi_groupnorm_silu_ex(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                          int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups, float eps,
                          int32_t fuse_silu, int32_t stack3, int32_t dtype, void* workspace, size_t workspace_bytes,
                          void* stream);

/* mvi_groupnorm_silu_ex with a sync buffer: groups that do not fit one block's registers (and every temporal / stacked call)
 * are then normalised in ONE launch that reads x once — K consecutive blocks per group exchange their partial moments through
 * `workspace` and meet at two counters per group in `sync`. `sync`: mvi_groupnorm_sync_bytes() bytes, zeroed ONCE by the
 * caller (the kernels leave it zeroed), private to one stream at a time (two launches that overlap in time must not share
 * it); NULL = the two-launch kernels of mvi_groupnorm_silu_ex. The last word is a sticky flag: non-zero if a block ever gave
 * up waiting for its group (bounded spin; the output of that call is then undefined). */
size_t mvi_groupnorm_sync_bytes(void);
int mvi_groupnorm_silu_ex2(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                           int64_t videos, int32_t T, int32_t C, int64_t spatial, int32_t groups, float eps,
                           int32_t fuse_silu, int32_t stack3, int32_t dtype, void* workspace, size_t workspace_bytes,
                           void* sync, size_t sync_bytes, void* stream);

/* The same norm (x [N, C, spatial], optional chan_bias, optional SiLU) with TOKEN-MAJOR output y [N, spatial, C]:
 * "b c h w -> b (h w) c" (svd_inpaint1/sgm/modules/attention.py:700-707) fused into the apply pass, for consumers that
 * contract over channels (proj_in Linear; a channels-last convolution). C and spatial must be multiples of the 16-byte
 * vector width; y must not alias x. */
int mvi_groupnorm_silu_tokens(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                              int64_t N, int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                              int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* GroupNorm(+SiLU) with TOKEN-MAJOR input and output: x, y [N, spatial, C] (C contiguous = NHWC). The norm between the two
 * 3x3 convolutions of a ResBlock (openaimodel.py:292-305, :341-352) when the convolutions run on channels-last tensors —
 * MIOpen's kernels are NHWC and wrap NCHW tensors in transposes (csrc/groupnorm_tokens.hip). chan_bias: optional fp32 [N, C]
 * added to x before the statistics (the timestep-embedding bias). workspace: mvi_groupnorm_tok2tok_workspace_bytes(...)
 * (0 = shape not supported: C must be a multiple of groups <= 64 and of the 16-byte vector width). */
size_t mvi_groupnorm_tok2tok_workspace_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups, int32_t dtype);
int mvi_groupnorm_silu_tok2tok(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias, int64_t N,
                               int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu, int32_t dtype,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same with TEMPORAL statistics: the N samples are videos of `frames` consecutive frames, a group's mean / variance are taken
 * over all frames of a video (the GroupNorm of VideoResBlock.time_stack on b c t h w, video_model.py:71-75); chan_bias stays per frame
 * [N, C] (the per-frame timestep embedding). Same workspace. */
int mvi_groupnorm_silu_tok2tok_frames(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias, int64_t N,
                                      int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                                      int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* out = softmax(q k^T * scale) v per (batch, head). Token-major layout, as the Linear projections
 * produce it: q/out [B, Sq, H, D], k/v [B, Sk, H, D], contiguous. No mask (none is used on the
 * denoise path). dtype selects the I/O type; fp32 I/O computes in fp32 (validation mode, 1e-4
 * parity), bf16/f16 I/O with D == 64 and Sk > 32 runs the MFMA flash kernel (fp32 accumulate,
 * fp32 softmax, P rounded to the I/O type before P.V), everything else an fp32-math kernel. */
int mvi_attention_forward(const void* q, const void* k, const void* v, void* out, int32_t B,
                          int32_t H, int32_t Sq, int32_t Sk, int32_t D, float scale, int32_t dtype,
                          void* stream);

/* Self-attention over the FRAME axis of x laid out [(Bo*T), S, H, D] (token-major, frames outermost within
 * a video — the layout the spatial layers leave behind), one softmax problem per (video, spatial token,
 * head): rows are the T frames, strided by S*H*D. Replaces the reference's regrouping
 * "(b t) s c -> (b s) t c" + attention + inverse regrouping
 * (svd_inpaint1/sgm/modules/video_attention.py:115, :136-140) without moving any token. q, k, v, out
 * share the layout. fp32 math for every dtype. */
int mvi_attention_temporal(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                           int32_t S, int32_t H, int32_t D, float scale, int32_t dtype, void* stream);

/* GEGLU gate of the transformer feed-forwards: out[r, j] = h[r, j] * gelu(h[r, inner + j]) (exact erf GELU),
 * h [rows, 2*inner] -> out [rows, inner], contiguous. Replaces `x, gate = proj(x).chunk(2, -1); x * F.gelu(gate)`
 * (svd_inpaint1/sgm/modules/attention.py:87-95). inner must be a multiple of 4 (fp32) / 8 (bf16, f16). */
int mvi_geglu(const void* h, void* out, int64_t rows, int32_t inner, int32_t dtype, void* stream);

/* GEGLU with its projection in one kernel (csrc/ff_geglu.hip):
 *   out[r, j] = (x[r, :] . weight[j, :] + bias[j]) * gelu(x[r, :] . weight[inner + j, :] + bias[inner + j])
 * i.e. `x, gate = F.linear(x, weight, bias).chunk(2, -1); x * F.gelu(gate)` (sgm/modules/attention.py:87-95) without the
 * [rows, 2 inner] intermediate. x [rows, K] and out [rows, inner] with row strides in elements, weight [2 inner, K] contiguous
 * (nn.Linear layout), bias fp32 [2 inner] or NULL. Only the shapes mvi_ff_geglu_supported() accepts (K = 320, bf16 / f16,
 * inner a multiple of 32): the level-0 FeedForward layers; everything else keeps library GEMM + mvi_geglu.
 * The kernel stores whole blocks of 256 rows without predication: `out` must have room for mvi_ff_geglu_out_rows(rows) rows
 * (rows rounded up to 256; the surplus rows receive values computed from the last valid row of x). */
int mvi_ff_geglu_supported(int32_t K, int32_t inner, int32_t dtype);
int64_t mvi_ff_geglu_out_rows(int64_t rows);
int mvi_ff_geglu(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity, int32_t K,
                 int32_t inner, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream);

/* The same kernel with a plain epilogue: out[r, j] = x[r, :] . weight[j, :] + bias[j] — nn.Linear for K = 320 (bf16 / f16,
 * out_features a multiple of 64): the bias-only projections of the level-0 transformer blocks (packed q/k/v, to_out, proj_in,
 * proj_out; sgm/modules/attention.py:281-344, :690-712), which are short-K, output-bound GEMMs the library runs at 2 TB/s. Same
 * padded-output contract as mvi_ff_geglu. */
int mvi_linear_k320_supported(int32_t K, int32_t out_features, int32_t dtype);
int mvi_linear_k320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity, int32_t K,
                    int32_t out_features, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream);

/* nn.Linear with 320 g outputs and a long contraction (csrc/linear_n320.hip): out[r, n] = x[r, :] . weight[n, :] + bias[n],
 * n < out_features = 320 g, K a multiple of 64 (>= 128), bf16 / f16 — the second projection of the FeedForward layers
 * (`FeedForward.net[2]`, sgm/modules/attention.py:98-115: [258048, 1280] x [1280, 320] at level 0, where the library's 256-wide
 * macro-tiles cover 320 columns with 37 % padding; round 5: [64512, 2560] x [2560, 640] and [16128, 5120] x [5120, 1280] at levels
 * 1 and 2, and the attention output projections 640 -> 640 / 1280 -> 1280). A block keeps 320 outputs of its 256 rows in
 * accumulators, the g column groups of a row block are neighbours in the grid; same padded-output contract as mvi_ff_geglu
 * (`out` has room for mvi_ff_geglu_out_rows(rows) rows; rows of x, weight and out 16-byte aligned). */
int mvi_linear_n320_supported(int32_t K, int32_t out_features, int32_t dtype);
/* The same projection into 320 channels with the residual add(s) and the LayerNorm that follow it in the transformer blocks run in
 * the epilogue (sgm/modules/attention.py:544-572 `x = attn1(norm1(x)) + x; ... self.ff(self.norm3(x))`, video_attention.py:110-141):
 *     h = round(x W^T + bias);  s_pre = round(resid + h);  s = round(s_pre + row[r / row_div]);  y = LayerNorm(s) * ln_weight + ln_bias
 * — the arithmetic and rounding points of mvi_add_layernorm behind mvi_linear_n320, without h, and without the read-back of resid + h.
 * resid [rows, 320] or NULL (then s_pre = h); row [G, 320] or NULL; s_pre / s [capacity, 320] or NULL (not stored); y [capacity, 320].
 * resid, s_pre, s and y have rows of out_row_stride elements; the outputs are padded to whole 256-row blocks like mvi_linear_n320's. */
int mvi_linear_n320_add_layernorm(const void* x, const void* weight, const float* bias, int64_t rows, int64_t out_rows_capacity, int32_t K,
                                  int64_t x_row_stride, const void* resid, const void* row, int64_t row_div, const float* ln_weight,
                                  const float* ln_bias, float eps, void* s_pre, void* s, void* y, int64_t out_row_stride, int32_t dtype,
                                  void* stream);
int mvi_linear_n320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity, int32_t K,
                    int32_t out_features, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream);

/* Convolutions with C_out a multiple of 320 on token-major (NHWC) activations, as implicit GEMMs in the kernel above (a block
 * computes 320 output channels of 256 rows; C_in a multiple of 64; bf16 / f16). bias fp32 [C_out] or NULL; out [rows, C_out] in rows
 * of out_row_stride elements with room for mvi_ff_geglu_out_rows(rows) rows.
 *   mvi_conv3x3_n320: 3x3 / padding 1, stride 1 (replaces F.conv2d in ResBlock.in_layers[2] / out_layers[3] and Upsample.conv,
 *     svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:256-275, :301-318, :118-134) or stride 2 (Downsample.op, :150-166: rows
 *     = the N Ho Wo output pixels, Ho = (H - 1) / 2 + 1). x [N, H, W, C_in];
 *     weight [C_out][9 C_in] = conv.weight.permute(0, 2, 3, 1) flattened.
 *   mvi_conv3t_n320: (3, 1, 1) / padding (1, 0, 0) over frames (the Conv3d of VideoResBlock.time_stack, video_model.py:41-54).
 *     x [B, T, pixels, C_in]; weight [C_out][3 C_in] = conv.weight[:, :, :, 0, 0].permute(0, 2, 1) flattened. */
int mvi_conv3x3_n320_supported(int32_t C_in, int32_t C_out, int32_t dtype);
/* Small images (fewer than 128 blocks of 256 rows x 320 channels) split K over up to 8 blocks per tile: fp32 partial sums in
 * `workspace` (..._workspace_bytes(); 0 = this shape is not split), reduced with the bias by a second launch. workspace NULL or too
 * small: the unsplit launch. */
size_t mvi_conv3x3_n320_workspace_bytes(int64_t N, int32_t H, int32_t W, int32_t C_in, int32_t C_out, int32_t stride);
size_t mvi_conv3t_n320_workspace_bytes(int64_t B, int32_t T, int32_t pixels, int32_t C_in, int32_t C_out);
int mvi_conv3x3_n320(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W, int32_t C_in,
                     int32_t C_out, int32_t stride, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype, void* workspace,
                     size_t workspace_bytes, void* stream);

/* conv3x3(F.interpolate(x, scale_factor=2, mode="nearest")) of token-major x [N, h w, C_in] -> out [N, (2 h)(2 w), C_out]: Upsample.conv
 * (openaimodel.py:107-150) with the upsampling in the kernel's addressing (round 6: the token-major residual stream). Conditions and
 * out capacity as mvi_conv3x3_n320 at (N, 2 h, 2 w), stride 1. */
int mvi_conv3x3_up2_n320(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t h, int32_t w, int32_t C_in,
                         int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype, void* workspace,
                         size_t workspace_bytes, void* stream);
int mvi_conv3t_n320(const void* x, const void* weight, const float* bias, void* out, int64_t B, int32_t T, int32_t pixels, int32_t C_in,
                    int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype, void* workspace,
                    size_t workspace_bytes, void* stream);

/* y = act(conv2d(x, weight, padding = 1) + bias) for a 3x3, stride-1 convolution with 16 output channels and at most 16 input
 * channels, 32 and at most 32, or 320 and at most 8 (the networks' input convolution), on NCHW bf16 / f16 tensors (csrc/stem_conv.hip) — the stride-1 layers of ControlNet.input_hint_block
 * at its two finest resolutions
 * (models/csvd.py:234-250: conv(7 -> 16) SiLU conv(16 -> 16) SiLU at the hint's 576 x 1024), which the library's wide-channel
 * kernels run 5x off their memory time. x [N, C_in, H, W], bias fp32 [C_out] or NULL, y [N, C_out, H_out, W_out]; W a multiple of
 * 8, x / y 16-byte aligned; fuse_silu != 0 applies SiLU. stride 2 (H_out = (H - 1) / 2 + 1, likewise W_out; W a multiple of 16) is
 * built for the 16 -> 32 layer that halves the hint's resolution.
 * `packed_weight`: the weight in the order of the kernel's MFMA B fragments, made ONCE per parameter version from the module's
 * [C_out, C_in, 3, 3] tensor (activation type) by mvi_stem_conv3x3_pack into mvi_stem_conv3x3_packed_bytes(...) bytes (16-byte
 * aligned): [C_out / 16][k-steps][64 lanes][8 elements]. */
int mvi_stem_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t W, int32_t stride, int32_t dtype);
size_t mvi_stem_conv3x3_packed_bytes(int32_t Cin, int32_t Cout, int32_t stride);
int mvi_stem_conv3x3_pack(const void* weight, void* packed_weight, int32_t Cin, int32_t Cout, int32_t W, int32_t stride, int32_t dtype,
                          void* stream);
int mvi_stem_conv3x3_silu(const void* x, const void* packed_weight, const float* bias, void* y, int64_t N, int32_t Cin, int32_t Cout,
                          int32_t H, int32_t W, int32_t stride, int32_t fuse_silu, int32_t dtype, void* stream);

/* out[n, c, p] = h[n, c, p] + bias[c] + x[n, c, p] over [N, C, spatial] activations in one pass; x and bias are
 * optional (NULL). Folds a convolution's bias (PyTorch-ROCm adds it in a separate kernel) and the ResBlock skip
 * add `self.skip_connection(x) + h` (svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:354). out may alias h. */
int mvi_bias_residual_add(const void* h, const void* x, const float* bias, void* out, int64_t N, int32_t C,
                          int64_t spatial, int32_t dtype, void* stream);

/* out[n, c, p] = x[n, c, p] + (1 - alpha[n]) * (h[n, c, p] + bias[c]): the tail of a VideoResBlock in one pass — the
 * temporal ResBlock's `x + h` (openaimodel.py:354) blended with its input by AlphaBlender,
 * alpha * x_spatial + (1 - alpha) * x_temporal (svd_inpaint1/sgm/modules/diffusionmodules/util.py:358-372,
 * video_model.py:67-81). alpha: fp32 [N] on the device (one value per frame: sigmoid(mix_factor), or 1 for image-only
 * frames); bias optional (NULL). out may alias h. */
/* out[n] = (h[n] | skip[n] + ctrl[n]) concatenated along channels: h [N, C1, spatial], skip and ctrl [N, C2, spatial],
 * out [N, C1 + C2, spatial], all contiguous; ctrl optional (NULL: a plain concatenation). The decoder's
 * `h = th.cat([h, hs.pop() + control.pop()], dim=1)` (svd_inpaint1/models/csvd.py:79-91) in one pass instead of an add
 * and a copy; the sum is rounded once to the storage type, as the reference's separate add does. N < 65536. */
int mvi_concat_add(const void* h, const void* skip, const void* ctrl, void* out, int64_t N, int32_t C1, int32_t C2,
                   int64_t spatial, int32_t dtype, void* stream);

int mvi_bias_residual_blend(const void* h, const void* x, const float* bias, const float* alpha, void* out, int64_t N,
                            int32_t C, int64_t spatial, int32_t dtype, void* stream);

/* ---- token-row kernels of the transformer blocks: t [R, C] token-major, contiguous ------------------------------
 * Residual add(s) + the NEXT LayerNorm in one pass (svd_inpaint1/sgm/modules/attention.py:544-572 `x = attn(norm(x)) + x`
 * chains; video_attention.py:110-141):
 *     s_pre = x + h                       (h optional: NULL -> s_pre = x)
 *     s     = s_pre + row[r / row_div]    (row optional, [ceil(R / row_div), C], same dtype: the single-token
 *                                          cross-attention row per image / per video, or the frame-index embedding)
 *     y     = LayerNorm(s) * weight + bias          (statistics fp32; weight / bias fp32 [C])
 * s_pre and s are written when their pointers are non-NULL (s_pre needs h; s needs h or row). Each materialised
 * intermediate is rounded to the storage type where the unfused graph would round it. h == row == NULL is a plain
 * LayerNorm (nn.LayerNorm, attention.py:509-511). C must split into 2^k lanes x <= 8 16-byte vectors
 * (mvi_layernorm_supported tells). */
int mvi_add_layernorm(const void* x, const void* h, const void* row, int64_t row_div, const float* weight,
                      const float* bias, void* s_pre, void* s, void* y, int64_t R, int32_t C, float eps, int32_t dtype,
                      void* stream);
int mvi_layernorm_supported(int32_t C, int32_t dtype);

/* out = lerp(x + h, base, alpha[r / row_div]) = alpha * base + (1 - alpha) * (x + h): the temporal block's last residual
 * add fused with AlphaBlender (svd_inpaint1/sgm/modules/diffusionmodules/util.py:312-372; video_attention.py:290-294).
 * alpha: fp32 [ceil(R / row_div)]; h optional. PyTorch's two-sided lerp formula. */
int mvi_add_lerp(const void* x, const void* h, const void* base, const float* alpha, int64_t row_div, void* out, int64_t R,
                 int32_t C, int32_t dtype, void* stream);

/* out[n, c, p] = tok[n, p, c] + x_in[n, c, p]: "b (h w) c -> b c h w" and the transformer's outer skip connection
 * (svd_inpaint1/sgm/modules/attention.py:717-722, video_attention.py:298-301) in one pass through an LDS tile.
 * C and spatial must be multiples of the 16-byte vector width (4 fp32 / 8 bf16, f16). */
int mvi_tokens_to_planes_add(const void* tok, const void* x_in, void* out, int64_t N, int32_t C, int64_t spatial,
                             int32_t dtype, void* stream);
/* ... + bias[c] (fp32 [C]): the bias of the convolution that produced the tokens rides on the same pass. */
int mvi_tokens_to_planes_add_bias(const void* tok, const void* x_in, const float* bias, void* out, int64_t N, int32_t C,
                                  int64_t spatial, int32_t dtype, void* stream);
/* (x_in may be NULL in both: the plain layout change "b (h w) c -> b c h w", + bias.) */

/* out[n, p', c] = x[n, c, p]: "b c h w -> b (h w) c" for x [N, C, H, W]; upsample = 2 folds the nearest-neighbour 2x upsampling of
 * Upsample.forward (svd_inpaint1/sgm/modules/diffusionmodules/openaimodel.py:118-134, F.interpolate(scale_factor=2, mode="nearest"))
 * into the token write: out [N, (2H)(2W), C]; upsample = 1: out [N, H W, C]. C and H W multiples of the 16-byte vector width. */
int mvi_planes_to_tokens(const void* x, void* out, int64_t N, int32_t C, int32_t H, int32_t W, int32_t upsample, int32_t dtype,
                         void* stream);

/* The token-major evaluation of a VideoResBlock (video_model.py:41-81) needs its spatial ResBlock to END on tokens and its tail to
 * restore b c h w:
 *   mvi_planes_add_to_tokens:  out[n, p, c] = x[n, c, p] + tok[n, p, c] + bias[c] — `skip_connection(x) + h` (openaimodel.py:354) with h
 *     (the second convolution's tokens, bias withheld) and the result token-major;
 *   mvi_tokens_blend_to_planes: out[n, c, p] = base[n, p, c] + (1 - alpha[n]) * (tok[n, p, c] + bias[c]) — the temporal ResBlock's skip
 *     add and the AlphaBlender (util.py:358-372) in the pass that restores b c h w; alpha fp32 [N].
 * bias fp32 [C] or NULL; C and spatial multiples of the 16-byte vector width. */
int mvi_planes_add_to_tokens(const void* x, const void* tok, const float* bias, void* out, int64_t N, int32_t C, int64_t spatial,
                             int32_t dtype, void* stream);
int mvi_tokens_blend_to_planes(const void* tok, const void* base, const float* bias, const float* alpha, void* out, int64_t N, int32_t C,
                               int64_t spatial, int32_t dtype, void* stream);

/* out = silu(h + bias[c]) for h [N, C, spatial]: convolution bias + SiLU of the ControlNet hint stem
 * (svd_inpaint1/models/csvd.py:234-250: eight convolutions with SiLU between, at up to 576x1024) in one pass instead of
 * the library's broadcast bias add followed by a separate activation. out may alias h. */
int mvi_bias_silu(const void* h, const float* bias, void* out, int64_t N, int32_t C, int64_t spatial, int32_t dtype,
                  void* stream);

/* The same two attention ops reading q, k, v with a token stride larger than H*D — for q, k, v taken straight out of
 * ONE packed projection [.., 3*H*D] = (q | k | v) (token stride 3*H*D, base pointers H*D apart): the three bias-free
 * Linear layers of a self-attention (svd_inpaint1/sgm/modules/attention.py:281-300) then run as one GEMM over the
 * activations instead of three. Strides are in elements, 0 = H*D (contiguous); they must be multiples of 16 bytes.
 * Tokens of one batch entry are consecutive (batch stride = S * token stride). */
int mvi_attention_forward_strided(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                                  int32_t Sq, int32_t Sk, int32_t D, float scale, int32_t dtype,
                                  int64_t q_token_stride, int64_t kv_token_stride, int64_t out_token_stride,
                                  void* stream);
int mvi_attention_temporal_strided(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                                   int32_t S, int32_t H, int32_t D, float scale, int32_t dtype,
                                   int64_t qkv_token_stride, int64_t out_token_stride, void* stream);
/* Order of the contraction index of mvi_conv3x3_n320's weight: 1 (default) = [C_out][C_in / 64][9 taps][64 channels] — the nine
 * taps of a 64-channel chunk are consecutive, so a block's re-reads of its activation rows hit the XCD's L2 —, 0 = [C_out][9][C_in]
 * (tap-major, rounds 3 - 4; MVI_CONV_K_ORDER=0). set = 0 / 1 selects, anything else only queries; returns the order in force. The
 * weight handed to mvi_conv3x3_n320* must be packed in that order. */
int mvi_conv3x3_n320_k_order(int32_t set);

/* mvi_ff_geglu for a LONG contraction (round 5): the GEGLU projection of the level-1 / level-2 FeedForward layers of the 576 x 1024
 * step ([64512, 640] x [640, 2 * 2560], [16128, 1280] x [1280, 2 * 5120]; sgm/modules/attention.py:87-95), in csrc/linear_n320.hip's
 * frame: a block's 320 accumulator columns are 160 value columns and the gate columns of the same 160 outputs, gated in registers
 * (exact-erf GELU as mvi_ff_geglu). K a multiple of 64 (>= 128), inner a multiple of 160, bf16 / f16; out as mvi_ff_geglu's
 * (room for mvi_ff_geglu_out_rows(rows) rows), rows 16-byte aligned. */
int mvi_ff_geglu_n320_supported(int32_t K, int32_t inner, int32_t dtype);
int mvi_ff_geglu_n320(const void* x, const void* weight, const float* bias, void* out, int64_t rows, int64_t out_rows_capacity,
                      int32_t K, int32_t inner, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, void* stream);

/* GroupNorm statistics from the PRODUCER of the normalised tensor (round 5). The ResBlock's second norm and its temporal twin read
 * what a convolution of this library has just written (openaimodel.py:292-305, :339-343; video_model.py:41-54):
 * mvi_conv3x3_n320_gnstats / mvi_conv3t_n320_gnstats are mvi_conv3x3_n320 (stride 1) / mvi_conv3t_n320 whose blocks also leave, in
 * gn_part [samples * (spatial / 256) * gn_groups][3], (count, mean, M2) of every GroupNorm group of their 256 rows — of the ROUNDED
 * outputs plus gn_chan_bias [samples, C_out] (the timestep-embedding bias the norm adds first; NULL: none) — and
 * mvi_groupnorm_silu_tok2tok_pre is mvi_groupnorm_silu_tok2tok_frames without its statistics pass, merging those partials
 * (chunks_per_sample = spatial / 256): the tensor is read once instead of twice. A sample is an image (3x3) or a frame (3-tap:
 * spatial = pixels of a frame); `frames` > 1 merges the frames of a video as in the _frames form. Only shapes for which
 * mvi_conv_n320_gnstats_supported answers 1 (stride 1, no K split, spatial a multiple of 256, groups of <= 40 channels that tile 320);
 * bf16 / f16. gn_part: mvi_conv_n320_gnstats_bytes(samples, spatial, groups); workspace of _pre: N * C * 2 floats. */
int mvi_conv_n320_gnstats_supported(int64_t rows, int32_t taps, int32_t stride, int32_t C_in, int32_t C_out, int64_t spatial,
                                    int32_t groups);
size_t mvi_conv_n320_gnstats_bytes(int64_t samples, int64_t spatial, int32_t groups);
int mvi_conv3x3_n320_gnstats(const void* x, const void* weight, const float* bias, void* out, int64_t N, int32_t H, int32_t W,
                             int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                             const float* gn_chan_bias, int32_t gn_groups, float* gn_part, size_t gn_part_bytes, void* stream);
int mvi_conv3t_n320_gnstats(const void* x, const void* weight, const float* bias, void* out, int64_t B, int32_t T, int32_t pixels,
                            int32_t C_in, int32_t C_out, int64_t out_rows_capacity, int64_t out_row_stride, int32_t dtype,
                            const float* gn_chan_bias, int32_t gn_groups, float* gn_part, size_t gn_part_bytes, void* stream);
int mvi_groupnorm_silu_tok2tok_pre(const void* x, void* y, const float* weight, const float* bias, const float* chan_bias,
                                   int64_t N, int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps,
                                   int32_t fuse_silu, int32_t dtype, const float* part, int32_t chunks_per_sample,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* The two strided forms for a q that ALREADY carries D^-1/2 * log2(e): P = exp2(q' . k - m), no scale argument. The caller folds
 * that constant into the WEIGHTS of the q projection in fp32, before their one rounding to bf16 / f16 (svd/transformer.py,
 * CrossAttention._packed_qkv_weight): q' = round(x . round(c W_q)^T) carries the same single output rounding as the reference's
 * q = round(x . W_q^T) (svd_inpaint1/sgm/modules/attention.py:281-300, :332-336), whereas scaling q inside the kernel rounds it a
 * second time. It takes the scale multiply (one v_mul per score) out of the 8-wave MFMA kernel's softmax — the port that bounds it.
 * The fp32-math kernels and the temporal MFMA kernel apply ln 2 in its place (same results to 1 ulp of the scores). */
int mvi_attention_forward_strided_qlog2(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H,
                                        int32_t Sq, int32_t Sk, int32_t D, int32_t dtype, int64_t q_token_stride,
                                        int64_t kv_token_stride, int64_t out_token_stride, void* stream);
int mvi_attention_temporal_strided_qlog2(const void* q, const void* k, const void* v, void* out, int32_t Bo, int32_t T,
                                         int32_t S, int32_t H, int32_t D, int32_t dtype, int64_t qkv_token_stride,
                                         int64_t out_token_stride, void* stream);
/* Which kernel serves a temporal-attention call of this shape: 1 = the MFMA kernel of csrc/attn_temporal.hip (bf16 / f16, D = 64,
 * T <= 16, 16-byte aligned rows), 0 = the fp32-math kernel of csrc/attn_rowtile.hip. Strides in elements, 0 = H*D. */
int mvi_attention_temporal_kernel_variant(int32_t T, int32_t H, int32_t D, int32_t dtype, int64_t qkv_token_stride,
                                          int64_t out_token_stride);

/* x[r, :] = softmax(scale * x[r, :]) in place, x [rows, cols] contiguous, scale > 0, fp32 statistics. The scaled
 * softmax between the two library GEMMs of the first-stage autoencoder's single-head attention with D = C = 512
 * (svd_inpaint1/sgm/modules/diffusionmodules/model.py:180-195, scaled_dot_product_attention with one head): the
 * score matrix of a frame chunk is held in HBM. One read and one write per score for cols <= 12288. */
int mvi_softmax_rows(void* x, int64_t rows, int32_t cols, float scale, int32_t dtype, void* stream);

/* Which kernel mvi_attention_forward would pick: 0 = rowtile fp32-math, 1 = MFMA flash. */
int mvi_attention_kernel_kind(int32_t Sq, int32_t Sk, int32_t D, int32_t dtype);
/* The kernel itself: 0 = rowtile, 4 = attn_flash_kernel (4 waves, 128-query blocks: short sequences), 8 =
 * attn_flash8_kernel (8 waves, 256-query blocks, LDS-DMA ring: S_q >= 1024 and S_k >= 256 — every level-0 / level-1 spatial
 * self-attention of the 576 x 1024 step). mvi_attention_forward* dispatch on exactly this function. */
int mvi_attention_kernel_variant(int32_t Sq, int32_t Sk, int32_t D, int32_t dtype);

const char* mvi_unet_last_error(void);

/* ---- Round 6: first-stage (VAE) decoder convolutions at fp32 accuracy on the bf16 matrix pipe (split operands).
 * Replaces the fp32 F.conv2d / F.conv3d calls behind sgm/modules/diffusionmodules/model.py:604-748 (Decoder: ResnetBlock conv1 / conv2,
 * Upsample.conv) and sgm/modules/autoencoding/temporal_ae.py:16-81 (VideoResBlock.time_stack) that the reference runs with autocast
 * disabled (configs/test/svd_f_est_ctrl_simp1.yaml:6, sgm/models/diffusion.py:194-212).
 *   x2     [rows, 2 C] bf16: every fp32 activation as (hi | lo) halves, hi = round_bf16(v), lo = round_bf16(v - hi)
 *          (mvi_groupnorm_silu_tok2tok_split writes it);
 *   weight [C_out padded to whole groups of mvi_conv_split3_group(C_out) columns][taps x 3 C] bf16 in the kernel's contraction order,
 *          logical channels (w_hi | w_lo | w_hi), padding rows zero (multiview_inpaint_amd/svd/hip_ops.py split3_weight);
 *   out    [mvi_conv_split3_out_rows(rows), C_out] fp32 = x_hi.w_hi + x_hi.w_lo + x_lo.w_hi, NO bias; rows >= N H W are scratch.
 * 3x3 / stride 1 / padding 1 over N images of H x W tokens, or (3,1,1) / padding (1,0,0) over B videos of T frames of `pixels` tokens.
 * rows x (row bytes of x2) must stay below 4 GiB (split the batch). Returns MVI_OK or a negative status (mvi_unet_last_error).
 * terms = 3 needs dtype MVI_DT_BF16. terms = 1 (dtype MVI_DT_BF16 or MVI_DT_F16): the same launch on ONE rounded value per operand —
 * x2 [rows, C], weight [C_out padded][taps x C] of that type, fp32 accumulation and fp32 out: the arithmetic of an autocast convolution
 * (the opt-in reduced-precision decode, multiview_inpaint_amd/svd/vae.py decode_first_stage(dtype=...)). */
int mvi_conv_split3_group(int32_t C_out);
int64_t mvi_conv_split3_out_rows(int64_t rows);
int mvi_conv3x3_split3_f32(const void* x2, const void* weight, float* out, int64_t N, int32_t H, int32_t W, int32_t C, int32_t C_out,
                           int32_t terms, int32_t dtype, int64_t out_rows_capacity, void* stream);
int mvi_conv3t_split3_f32(const void* x2, const void* weight, float* out, int64_t B, int32_t T, int32_t pixels, int32_t C, int32_t C_out,
                          int32_t terms, int32_t dtype, int64_t out_rows_capacity, void* stream);
/* GroupNorm(+SiLU) of fp32 token-major x [N, S, C] written as split bf16 y2 [N, S, 2 C] = (hi | lo) (out_mode 0) or as one rounded value per
 * element, y2 [N, S, C] bf16 (out_mode 1) / f16 (out_mode 2); frames > 1: statistics per video of `frames` consecutive samples; groups = 0:
 * no normalisation (plain split / rounding). Workspace: mvi_groupnorm_tok2tok_workspace_bytes(.., MVI_DT_F32). */
int mvi_groupnorm_silu_tok2tok_split(const float* x, void* y2, const float* weight, const float* bias, const float* chan_bias,
                                     int64_t N, int32_t frames, int32_t C, int64_t spatial, int32_t groups, float eps, int32_t fuse_silu,
                                     int32_t out_mode, void* workspace, size_t workspace_bytes, void* stream);

/* out[r][c] = a[r][c] + alpha * (b[r][c] + bias[c]) on fp32 rows [R, C], C a multiple of 4; out may alias a or b; bias may be NULL.
 * The skip add of a ResnetBlock (model.py:156-158) and the blend x + alpha (h + bias) of VideoResBlock (temporal_ae.py:70-81) on the
 * token-major fp32 activations of the split-operand first-stage decoder. */
int mvi_rows_axpb_f32(const float* a, const float* b, const float* bias, float alpha, float* out, int64_t R, int32_t C, void* stream);

/* ---- Round 6: the elementwise tails of the UNet's blocks on TOKEN-MAJOR tensors [N, spatial, C], with the statistics of the GroupNorm
 * that reads the result next (csrc/groupnorm_tokens.hip gt_fused_kernel):
 *   mode 0: out = a + b + bias[c]      — ResBlock `skip_connection(x) + h` (openaimodel.py:354), SpatialTransformer `x + x_in` (attention.py:717-722)
 *   mode 1: out = base + (1 - alpha[n]) (a + bias[c]) — the temporal ResBlock's skip add + AlphaBlender (video_model.py:67-81, util.py:358-372)
 * groups > 0: part receives (count, mean, M2) per (sample, chunk, group) of `out` in the layout mvi_groupnorm_silu_tok2tok_pre merges
 * (*chunks_per_sample chunks per sample): the next block's first norm (openaimodel.py:256-261, attention.py:700-707, video_model.py:41-54)
 * then needs no statistics pass. Tensors of one dtype (fp32 / bf16 / f16), 16-byte aligned; bias [C], alpha [N] fp32. */
size_t mvi_rows_gnstats_bytes(int64_t N, int32_t C, int64_t spatial, int32_t groups, int32_t dtype);
int mvi_rows_fused_gnstats(int32_t mode, const void* a, const void* b, const void* base, const float* bias, const float* alpha, void* out,
                           int64_t N, int32_t C, int32_t C_first, int64_t spatial, int32_t groups, int32_t dtype, float* part,
                           size_t part_bytes, int32_t* chunks_per_sample, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVI_UNET_OPS_H */
