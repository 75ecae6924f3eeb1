/*
 * mvi_raster.h — C-ABI of the MI355X (gfx950) Gaussian-splat rasterizer-with-depth.
 *
 * Drop-in boundary: these entry points are what a binding of the reference's rasterizer plug-in
 * binds. The reference imports the plug-in as a Python package,
 *     gs-simp/gaussian_renderer/__init__.py:14
 *         from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
 * and calls it at gs-simp/gaussian_renderer/__init__.py:85-93 (forward; returns color, radii,
 * depth) and through loss.backward() at gs-simp/train.py:93, gs-simp/inpaint_rec.py:125,
 * gs-simp/sds_train.py:130 (backward). That package's native module (rasterize_gaussians /
 * rasterize_gaussians_backward / mark_visible) is what this library replaces; the Python side
 * that binds it with ctypes is multiview_inpaint_amd/raster/ (see INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all arrays are dense, row-major, fp32 unless stated; shapes as in the reference call:
 *       means3D [P,3], scales [P,3], rotations [P,4] (w,x,y,z), opacities [P], shs [P,M,3],
 *       colors_precomp [P,3], cov3D_precomp [P,6] (xx,xy,xz,yy,yz,zz),
 *       viewmatrix/projmatrix [16] in Camera.world_view_transform / full_proj_transform memory
 *       order (gs-simp/scene/cameras.py:60-62), campos [3], bg [3];
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it. Only
 *     mvi_raster_forward_geom blocks the host (once, on an event behind the 4-byte read-back of
 *     num_rendered; the device keeps running the depth sort queued behind it);
 *   - the library owns no memory: scratch comes from the caller (sizes from the *_bytes queries),
 *     and the caller keeps geom/binning/image alive from forward to backward (the reference's
 *     autograd Function keeps them in ctx);
 *   - every function returns 0 on success, a negative MVI_E* code otherwise;
 *     mvi_raster_last_error() returns a thread-local message for the last failure.
 */
#ifndef MVI_RASTER_H
#define MVI_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVI_OK 0
#define MVI_EINVAL (-1)   /* bad argument combination (mirrors the Python layer's Exception) */
#define MVI_EHIP (-2)     /* a HIP call or kernel launch failed */
#define MVI_ENOMEM (-3)   /* a caller-provided scratch buffer is too small */

#define MVI_TILE 16       /* tile edge in pixels; one workgroup per tile (forward: 4 wave64, one pixel per lane; backward: 2 wave64,
                           * two pixels per lane) */

typedef struct mvi_raster_settings {
    int32_t image_height;       /* GaussianRasterizationSettings.image_height */
    int32_t image_width;        /* .image_width */
    float tanfovx;              /* .tanfovx */
    float tanfovy;              /* .tanfovy */
    float scale_modifier;       /* .scale_modifier */
    int32_t sh_degree;          /* .sh_degree (active degree 0..3) */
    int32_t prefiltered;        /* .prefiltered */
    const float* bg;            /* .bg         device [3]  */
    const float* viewmatrix;    /* .viewmatrix device [16] */
    const float* projmatrix;    /* .projmatrix device [16] */
    const float* campos;        /* .campos     device [3]  */
} mvi_raster_settings;

/* Scratch sizes. geom: per-Gaussian SoA; image: per-pixel + per-tile; binning: per (tile,Gaussian)
 * pair, needs num_rendered which mvi_raster_forward_geom returns. */
size_t mvi_raster_geom_bytes(int32_t P);
size_t mvi_raster_image_bytes(int32_t image_width, int32_t image_height);
size_t mvi_raster_binning_bytes(int64_t num_rendered, int32_t image_width, int32_t image_height);

/* Forward, stage 1: per-Gaussian preprocess (cull, cov3D, EWA cov2D, conic, radius, tile rect,
 * SH colour) + scan of tiles touched. Exactly one of shs | colors_precomp and one of
 * (scales, rotations) | cov3D_precomp must be non-NULL. M = SH coefficients per channel in shs.
 * Writes radii [P] int32 and *num_rendered_host.
 * SH colours are DEFERRED by default (mvi_raster_color_mode): this call records where means3D / shs live, and the render
 * kernel of mvi_raster_forward_render evaluates a Gaussian's colour — with the arithmetic the eager evaluation uses, same
 * bits — the first time a tile stages it. Only a few per cent of the visible Gaussians of a dense scene are ever staged
 * (a tile's pixels saturate after the first few hundred entries of its list), and at degree 3 the coefficients are 63 % of
 * what this call would read. means3D and shs must therefore stay valid and unchanged until mvi_raster_forward_render has
 * run on the stream (they must until the backward anyway). */
int mvi_raster_forward_geom(const mvi_raster_settings* s, int32_t P, int32_t M,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, void* geom, size_t geom_bytes,
                            int32_t* radii, int64_t* num_rendered_host, void* stream);

/* Forward, stage 2: binning — the (tile, depth)-sorted point list and the tile ranges — then per-tile front-to-back
 * compositing. Binning version 2 (grids up to 256 x 256 tiles, the default): the Gaussians are depth-sorted once (32-bit keys),
 * their tile rectangles expanded into column segments partitioned by tile column, then into pairs partitioned by tile row; no
 * 64-bit key is materialised. Version 1 (larger grids): pair emission + radix sort of the tile ids. Same outputs bit for bit
 * (mvi_raster_binning_version). out_color [3,H,W], out_depth [1,H,W] (15.0f where nothing composites, gs-simp/gen_seq.py:50).
 * May run on another host thread than mvi_raster_forward_geom, after it returned, on the same geom scratch. */
int mvi_raster_forward_render(const mvi_raster_settings* s, int32_t P, int64_t num_rendered,
                              const int32_t* radii, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                              void* image, size_t image_bytes, float* out_color, float* out_depth,
                              void* stream);
/* The same, and additionally ZEROES grad_rows_scratch [P][16] — the accumulation rows a following backward needs —
 * inside the render kernel (issue-bound, memory pipes nearly idle: the 64 bytes per Gaussian cost nothing there, against a
 * 96 MB fill pass in front of the render backward). Pass the same buffer to the backward with grad_rows_prezeroed = 1. */
int mvi_raster_forward_render_prepare(const mvi_raster_settings* s, int32_t P, int64_t num_rendered,
                                      const int32_t* radii, void* geom, size_t geom_bytes, void* binning,
                                      size_t binning_bytes, void* image, size_t image_bytes, float* out_color,
                                      float* out_depth, float* grad_rows_scratch, void* stream);

/* Backward. dL_dout_color [3,H,W]. Gradient outputs are overwritten (not accumulated):
 * dL_dmeans3D [P,3], dL_dmeans2D [P,3] (x,y = NDC-scaled screen gradient, z = 0; consumer
 * gs-simp/scene/gaussian_model.py:482-484), dL_dopacity [P], and
 * dL_dshs [P,M,3] | dL_dcolors [P,3], dL_dscales [P,3] + dL_drotations [P,4] | dL_dcov3D [P,6]
 * (pass NULL for the member of each pair that was not a forward input). grad_rows_prezeroed != 0: the scratch rows were
 * zeroed by mvi_raster_forward_render_prepare (and not used since); otherwise the backward zeroes them itself.
 * With shs as the forward input, dL_dcolors is an OPTIONAL extra output: the colour factor of the rank-1 SH
 * gradient, dL/dSH[k][c] = Y_k(dir) * dL_dcolors[c] (clamped channels zeroed); dL_dshs may then be NULL and the
 * dense gradient is rebuilt — summed over views — by mvi_raster_sh_backward_views (view-parallel training).
 * grad_rows_scratch: [P,16] fp32 scratch (64-byte accumulation row per Gaussian). */
int mvi_raster_backward(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t num_rendered,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* scales, const float* rotations, const float* cov3D_precomp,
                        const int32_t* radii, const void* geom, const void* binning,
                        const void* image, const float* dL_dout_color, float* dL_dmeans3D,
                        float* dL_dmeans2D, float* dL_dopacity, float* dL_dshs, float* dL_dcolors,
                        float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                        float* grad_rows_scratch, int32_t grad_rows_prezeroed, void* stream);

/* Raw-parameter forms ("fused activation + pack pre-pass"): the same two calls fed with the
 * GaussianModel's UN-activated parameters exactly as it stores them (gs-simp/scene/gaussian_model.py:95-115, :44-59):
 * xyz [P,3], features_dc [P,1,3], features_rest [P,M-1,3], raw_opacity [P,1], raw_scaling [P,3], raw_rotation [P,4].
 * sigmoid / exp / F.normalize (eps 1e-12) and the SH concatenation happen inside the preprocess kernels, and the
 * backward applies their chain rule, so exp / normalize / sigmoid / cat and their autograd nodes — 5 + ~12 PyTorch
 * kernels, the cat alone moving 2 x 192 B per Gaussian each way — disappear. mvi_raster_forward_render is shared.
 * Gradients are with respect to the raw parameters; dL_dmeans2D as in mvi_raster_backward. */
int mvi_raster_forward_geom_raw(const mvi_raster_settings* s, int32_t P, int32_t M, const float* xyz,
                                const float* features_dc, const float* features_rest, const float* raw_opacity,
                                const float* raw_scaling, const float* raw_rotation, void* geom, size_t geom_bytes,
                                int32_t* radii, int64_t* num_rendered_host, void* stream);
int mvi_raster_backward_raw(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t num_rendered, const float* xyz,
                            const float* features_dc, const float* features_rest, const float* raw_opacity,
                            const float* raw_scaling, const float* raw_rotation, const int32_t* radii, const void* geom,
                            const void* binning, const void* image, const float* dL_dout_color, float* dL_dxyz,
                            float* dL_dmeans2D, float* dL_draw_opacity, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                            float* dL_draw_scaling, float* dL_draw_rotation, float* grad_rows_scratch, 
                            int32_t grad_rows_prezeroed, void* stream);
/* mvi_raster_backward_raw with the SH gradient left in FACTORED form: dL_dsh_color_factor [P,3] (what mvi_raster_backward
 * writes into dL_dcolors for an SH input: dL/dcolour with the clamp mask applied) instead of dL_dfeatures_dc / dL_dfeatures_rest —
 * the form view-parallel training exchanges (multiview_inpaint_amd/train_views.py on the model's STORED parameters; the summed
 * dL/dSH is rebuilt by mvi_raster_sh_backward_views). The other gradients are those of mvi_raster_backward_raw, bit for bit. */
int mvi_raster_backward_raw_factor(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t num_rendered, const float* xyz,
                                   const float* features_dc, const float* features_rest, const float* raw_opacity,
                                   const float* raw_scaling, const float* raw_rotation, const int32_t* radii, const void* geom,
                                   const void* binning, const void* image, const float* dL_dout_color, float* dL_dxyz,
                                   float* dL_dmeans2D, float* dL_draw_opacity, float* dL_dsh_color_factor,
                                   float* dL_draw_scaling, float* dL_draw_rotation, float* grad_rows_scratch,
                                   int32_t grad_rows_prezeroed, void* stream);

/* View-parallel training (no counterpart in the reference, which is single-GPU: gs-simp/train.sh:1):
 * dL_dshs[g,k,c] = sum over views v of Y_k(normalize(means3D[g] - campos[v])) * dL_dcolors[v,g,c] for k < (deg+1)^2,
 * zero for the inactive coefficients. View v's camera centre is campos + v * campos_stride (3 floats), its colour
 * factors dL_dcolors + v * colors_view_stride ([P,3]; strides in floats, so both can live in one all-gathered
 * buffer) — 12 B per Gaussian and view on the wire instead of 12*M B in an all-reduce. dL_dshs [P,M,3] overwritten. */
int mvi_raster_sh_backward_views(int32_t P, int32_t M, int32_t sh_degree, int32_t n_views, const float* means3D,
                                 const float* campos, int64_t campos_stride, const float* dL_dcolors,
                                 int64_t colors_view_stride, float* dL_dshs, void* stream);

/* visible [P] uint8 = 1 where view-space z > 0.2 (the plug-in's markVisible; unused by the
 * reference but part of the plug-in surface). */
int mvi_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                            const float* projmatrix, uint8_t* visible, void* stream);

/* The two halves of mvi_raster_backward, for callers that start exchanging part of the result before the rest exists
 * (view-parallel training, SURVEY.md §8e: the colour factors of the SH gradient are all-gathered while the per-Gaussian
 * chain rule still runs). mvi_raster_backward_render: zeroes grad_rows_scratch [P][16], runs the render backward, and —
 * when dL_dcolor_factor [P,3] is given — writes what mvi_raster_backward would later put into dL_dcolors (the clamp-masked
 * colour gradient; sh_input != 0 applies the SH clamp mask). mvi_raster_backward_geom: the per-Gaussian chain rule from
 * those rows (arguments as in mvi_raster_backward). render + geom is what mvi_raster_backward runs. */
int mvi_raster_backward_render(const mvi_raster_settings* s, int32_t P, int64_t num_rendered, const int32_t* radii,
                               const void* geom, const void* binning, const void* image, const float* dL_dout_color,
                               float* grad_rows_scratch, float* dL_dcolor_factor, int32_t sh_input,
                               int32_t grad_rows_prezeroed, void* stream);
int mvi_raster_backward_geom(const mvi_raster_settings* s, int32_t P, int32_t M, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations,
                             const float* cov3D_precomp, const int32_t* radii, const void* geom,
                             const float* grad_rows_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                             float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                             void* stream);

/* mvi_raster_backward_geom for the Gaussians [first, first + count) only (first a multiple of 64): every pointer is the
 * base of the FULL [P, ...] array, as in mvi_raster_backward_geom; only rows of the range are read and written. Lets a
 * view-parallel trainer run the chain rule in a few ranges and start the all-reduce of a finished range's gradients
 * while the next range is computed (multiview_inpaint_amd/dist.py: RangedGradExchange). Ranges are independent: the
 * union of the calls over a partition of [0, P) writes exactly what one mvi_raster_backward_geom call writes. */
int mvi_raster_backward_geom_range(const mvi_raster_settings* s, int32_t P, int32_t M, int32_t first, int32_t count,
                                   const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, const int32_t* radii, const void* geom,
                                   const float* grad_rows_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                                   float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                                   void* stream);

/* Introspection used by the parity tests: copies of intermediate device arrays' addresses.
 * Pointers alias the caller's scratch buffers; valid while those are. */
typedef struct mvi_raster_views {
    const float* depths;          /* [P] */
    const float* means2D;         /* [P,2] pixel centres */
    const float* cov3D_a;         /* [P,4] xx, xy, xz, yy */
    const float* cov3D_b;         /* [P,2] yz, zz */
    const float* conic_opacity;   /* [P,4] */
    const float* rgbd;            /* [P,4] r, g, b, depth; deferred SH colours (the default): (-1, -1, -1, depth) for a visible
                                   * Gaussian no tile has staged yet — see mvi_raster_color_mode / mvi_raster_resolve_colors */
    const uint32_t* tiles_touched;/* [P] */
    const uint8_t* clamped;       /* [P] bit c = colour channel c clamped at 0 (0 while the colour is pending) */
    const void* tile_ids_sorted;  /* [D] high word of the sort key, tile_id_bytes (2 or 4) per entry; the full key of
                                   * pair i is tile_ids_sorted[i] << 32 | bits(depths[point_list[i]]) */
    const uint32_t* point_list;   /* [D] Gaussian index per sorted pair */
    const uint32_t* ranges;       /* [tiles,2] */
    const float* final_T;         /* [H,W] */
    const uint32_t* n_contrib;    /* [H,W] */
    int32_t tile_id_bytes;        /* 2 (uint16) while the image has at most 65536 tiles, else 4 (uint32) */
    const uint8_t* grad_support;  /* [P] valid after a backward: 1 = the render backward added to this Gaussian's accumulation
                                   * row; 0 = every gradient of the Gaussian is exactly zero in this view (occluded, culled or
                                   * outside the image). View-parallel training exchanges only rows in the union of the supports. */
} mvi_raster_views;
int mvi_raster_get_views(int32_t P, int64_t num_rendered, int32_t image_width, int32_t image_height,
                         const void* geom, const void* binning, const void* image,
                         mvi_raster_views* out);

/* Stage timing (measurement aid for bench.py): while enabled, every kernel stage the library
 * enqueues is bracketed by hipEvents on the caller's stream. mvi_raster_timing_read waits for the
 * recorded events, adds the elapsed milliseconds and launch counts per stage into ms_sum/calls
 * ([MVI_RASTER_NSTAGES] each, caller-zeroed) and forgets them. Not thread-safe. */
#define MVI_RASTER_NSTAGES 8
int mvi_raster_timing_enable(int enable);
/* The same for a subset of the stages (bit s of stage_mask = stage s). Every bracketed stage boundary costs ~10 us of idle
 * GPU (the event record serialises the queue), so a throughput measurement brackets only the kernel it needs. */
int mvi_raster_timing_enable_stages(uint32_t stage_mask);
int mvi_raster_timing_read(float* ms_sum_host, int32_t* calls_host);
const char* mvi_raster_stage_name(int stage);

/* SH colours: 1 (default) = deferred, evaluated by the render kernel on first use (see mvi_raster_forward_geom); 0 = eager,
 * evaluated for every visible Gaussian by the preprocess kernel (MVI_RASTER_EAGER_COLORS=1 selects it at start-up). Images,
 * gradients and every colour that IS evaluated are identical bit for bit; in deferred mode mvi_raster_views.rgbd holds
 * (-1, -1, -1, depth) and clamped 0 for the Gaussians no tile staged. Pass 0 or 1 to select, anything else to query;
 * returns the previous selection. Must not change between the two forward calls of one view. Not thread-safe. */
int mvi_raster_color_mode(int deferred);
/* Evaluates every colour still pending after a deferred forward (no-op after an eager one), so that mvi_raster_views.rgbd /
 * clamped are complete: for the parity tests and for callers that inspect colours of Gaussians that were never composited.
 * Needs the means3D / shs arrays of the forward still valid. */
int mvi_raster_resolve_colors(const mvi_raster_settings* s, int32_t P, void* geom, size_t geom_bytes, void* stream);

/* Binning implementation: 2 (default) = the rectangle-expanding partition of csrc/raster_binning2.hip, used for tile grids of
 * at most 256 x 256 tiles; 1 = the pair-emitting radix partition of csrc/raster_binning.hip (always used above that size,
 * or when MVI_BINNING_LEGACY=1). Both produce the same point list, tile ids and ranges bit for bit; the switch exists for
 * the A/B parity test and A/B timing. Pass 1 or 2 to select, anything else to query; returns the previous selection. Must
 * not change between a forward and its backward (the scratch layouts differ). Not thread-safe. */
int mvi_raster_binning_version(int version);
/* The per-Gaussian chain rule of mvi_raster_backward / _raw: 0 (default) = every output array is zeroed by the render backward
 * on the side and only the Gaussians in the gradient support (mvi_raster_views.grad_support) are read, computed and written;
 * 1 = the dense kernel that reads and writes every Gaussian (what the split / ranged entry points always use;
 * MVI_RASTER_DENSE_BACKWARD=1 selects it at start-up). Same results: an untouched accumulation row is exactly zero. Pass 0 or 1
 * to select, anything else to query; returns the previous selection. For the A/B parity test and A/B timing. */
int mvi_raster_backward_mode(int dense);
/* Diagnostics for kernel work (tools/expand_stamps.py), inert unless set: while device_buffer is non-NULL, every block of the
 * binning partition kernel of `pass` (1 | 2) writes 8 shader-clock stamps (uint64) at its phase boundaries into
 * device_buffer[block][8]. The caller sizes the buffer for the launch grid (mvi_raster_binning_bytes / 8 is ample). */
int mvi_raster_dev_stamps(int pass, void* device_buffer);

const char* mvi_raster_last_error(void);
const char* mvi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MVI_RASTER_H */
