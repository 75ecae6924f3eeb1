// C-ABI of the rasterizer (include/mvi_raster.h). Host-side only: argument checks, scratch carving,
// kernel sequencing on the caller's stream. Mirrors the error behaviour of the plug-in the
// reference imports (gs-simp/gaussian_renderer/__init__.py:14): argument-combination errors are
// reported, never silently repaired.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "raster_common.h"

namespace mvi {
int launch_scan_block_sums(GeomView g, int P, unsigned long long* total_host_devptr, hipStream_t st);
int launch_zero_fill(void* p, size_t bytes, hipStream_t st);
int launch_binning_level1(const Frame& f, GeomView g, hipStream_t st);
}

static thread_local char g_err[512] = "";
// 0 (default): one-call backward with outputs zeroed on the side + sparse chain rule; 1: the dense chain-rule kernel
static int g_dense_backward = [] { const char* e = getenv("MVI_RASTER_DENSE_BACKWARD"); return (e && e[0] == '1') ? 1 : 0; }();
// 1 (default): SH colours are evaluated by the render kernel when it first stages a Gaussian (raster_common.h, ColorSource);
// 0: by the preprocess kernel for every visible Gaussian (MVI_RASTER_EAGER_COLORS=1 selects it at start-up)
static int g_defer_colors = [] { const char* e = getenv("MVI_RASTER_EAGER_COLORS"); return (e && e[0] == '1') ? 0 : 1; }();
// Column segments counted by forward_geom (binning version 2), remembered PER GEOM SCRATCH: the exact grid size of the
// forward_render that follows on that scratch. Process-wide and locked, so the two halves of a forward may run on different
// threads; an entry describes the scratch's CONTENTS (the last forward_geom that wrote it replaces it), so it can only be
// stale if the caller overwrites a scratch while using it. Unknown scratch (table full and evicted): the bound num_rendered.
#include <mutex>
namespace {
struct SegmentHint { const void* geom; int32_t P; int64_t segments; uint64_t stamp; };
constexpr int kSegmentHints = 64;
SegmentHint g_seg_hints[kSegmentHints] = {};
uint64_t g_seg_clock = 0;
std::mutex g_seg_mutex;
void remember_segments(const void* geom, int32_t P, int64_t segments) {
    std::lock_guard<std::mutex> lock(g_seg_mutex);
    int slot = 0;
    for (int i = 0; i < kSegmentHints; ++i) {
        if (g_seg_hints[i].geom == geom) { slot = i; break; }
        if (g_seg_hints[i].stamp < g_seg_hints[slot].stamp) slot = i;      // else the least recently written
    }
    g_seg_hints[slot] = {geom, P, segments, ++g_seg_clock};
}
int64_t recall_segments(const void* geom, int32_t P) {
    std::lock_guard<std::mutex> lock(g_seg_mutex);
    for (int i = 0; i < kSegmentHints; ++i)
        if (g_seg_hints[i].geom == geom && g_seg_hints[i].stamp) return g_seg_hints[i].P == P ? g_seg_hints[i].segments : 0;
    return 0;
}
}  // namespace

// ---- stage timing ------------------------------------------------------------------------------
#include <vector>
namespace {
struct Rec { int stage; hipEvent_t a, b; };
uint32_t g_timing_mask = 0;        // bit s = stage s is bracketed by events
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[MVI_RASTER_NSTAGES];
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
namespace mvi {
void stage_begin(int stage, hipStream_t st) {
    if (!((g_timing_mask >> stage) & 1u)) return;
    g_open[stage] = get_event();
    (void)hipEventRecord(g_open[stage], st);
}
void stage_end(int stage, hipStream_t st) {
    if (!((g_timing_mask >> stage) & 1u)) return;
    hipEvent_t b = get_event();
    (void)hipEventRecord(b, st);
    g_recs.push_back({stage, g_open[stage], b});
}
}  // namespace mvi

static int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}
static int hip_fail(const char* where, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return MVI_EHIP;
}

static int make_frame(const mvi_raster_settings* s, int P, int M, mvi::Frame& f) {
    if (!s) return fail(MVI_EINVAL, "settings is NULL%s");
    if (s->image_width < 0 || s->image_height < 0 || P < 0)
        return fail(MVI_EINVAL, "negative size%s (P=%lld, W=%lld)", "", P, s->image_width);
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail(MVI_EINVAL, "sh_degree must be 0..3%s (got %lld)", "", s->sh_degree);
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos)
        return fail(MVI_EINVAL, "bg/viewmatrix/projmatrix/campos must be device pointers%s");
    f.P = P; f.M = M; f.deg = s->sh_degree; f.W = s->image_width; f.H = s->image_height;
    f.gx = (f.W + mvi::kTile - 1) / mvi::kTile;
    f.gy = (f.H + mvi::kTile - 1) / mvi::kTile;
    f.tanfovx = s->tanfovx; f.tanfovy = s->tanfovy;
    f.fx = (float)f.W / (2.0f * s->tanfovx);
    f.fy = (float)f.H / (2.0f * s->tanfovy);
    f.scale_modifier = s->scale_modifier;
    f.view = s->viewmatrix; f.proj = s->projmatrix; f.campos = s->campos; f.bg = s->bg;
    f.bin_v2 = mvi::binning_v2_ok(f.gx, f.gy) ? 1 : 0;
    f.defer_colors = g_defer_colors;
    return MVI_OK;
}

extern "C" {

const char* mvi_raster_last_error(void) { return g_err; }

int mvi_raster_timing_enable(int enable) { g_timing_mask = enable ? (1u << MVI_RASTER_NSTAGES) - 1u : 0u; return MVI_OK; }
int mvi_raster_timing_enable_stages(uint32_t stage_mask) { g_timing_mask = stage_mask & ((1u << MVI_RASTER_NSTAGES) - 1u); return MVI_OK; }
int mvi_raster_timing_read(float* ms_sum, int32_t* calls) {
    if (!ms_sum || !calls) return fail(MVI_EINVAL, "NULL timing outputs%s");
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return hip_fail("timing sync", e);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        ms_sum[r.stage] += ms;
        calls[r.stage] += 1;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return MVI_OK;
}
const char* mvi_raster_stage_name(int i) {
    static const char* n[MVI_RASTER_NSTAGES] = {"preprocess_forward", "scan_block_sums", "duplicate_keys", "radix_sort",
                                                "tile_ranges", "render_forward", "render_backward", "preprocess_backward"};
    return (i >= 0 && i < MVI_RASTER_NSTAGES) ? n[i] : "";
}
int mvi_raster_dev_stamps(int pass, void* device_buffer) { mvi::set_dev_stamps(pass, device_buffer); return MVI_OK; }
int mvi_raster_backward_mode(int dense) {
    const int old = g_dense_backward;
    if (dense == 0 || dense == 1) g_dense_backward = dense;
    return old;
}
int mvi_raster_binning_version(int version) { return mvi::set_binning_version(version); }
int mvi_raster_color_mode(int deferred) {
    const int old = g_defer_colors;
    if (deferred == 0 || deferred == 1) g_defer_colors = deferred;
    return old;
}
int mvi_raster_resolve_colors(const mvi_raster_settings* s, int32_t P, void* geom, size_t geom_bytes, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, 0, f)) return rc;
    if (P == 0) return MVI_OK;
    if (!geom) return fail(MVI_EINVAL, "geom is NULL%s");
    mvi::GeomView g = mvi::carve_geom(geom, P);
    if (geom_bytes < g.bytes) return fail(MVI_ENOMEM, "geom scratch too small%s", "");
    if (mvi::launch_resolve_colors(f, g, (hipStream_t)stream)) return hip_fail("resolve_colors", hipGetLastError());
    return MVI_OK;
}
const char* mvi_version(void) { return "multiview_inpaint_amd 0.1.0 (gfx950)"; }

size_t mvi_raster_geom_bytes(int32_t P) { return mvi::carve_geom(nullptr, P).bytes; }
size_t mvi_raster_image_bytes(int32_t W, int32_t H) { return mvi::carve_image(nullptr, W, H).bytes; }
size_t mvi_raster_binning_bytes(int64_t D, int32_t W, int32_t H) { return mvi::carve_binning(nullptr, D, W, H).bytes; }

static int forward_geom_impl(const mvi_raster_settings* s, mvi::Frame& f, int32_t P, int32_t M, const float* means3D,
                             const float* shs, const float* colors_precomp, const float* opacities,
                             const float* scales, const float* rotations, const float* cov3D_precomp,
                             void* geom, size_t geom_bytes, int32_t* radii, int64_t* num_rendered_host,
                             void* stream);

int mvi_raster_forward_geom(const mvi_raster_settings* s, int32_t P, int32_t M, const float* means3D,
                            const float* shs, const float* colors_precomp, const float* opacities,
                            const float* scales, const float* rotations, const float* cov3D_precomp,
                            void* geom, size_t geom_bytes, int32_t* radii, int64_t* num_rendered_host,
                            void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    return forward_geom_impl(s, f, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, geom,
                             geom_bytes, radii, num_rendered_host, stream);
}

int mvi_raster_forward_geom_raw(const mvi_raster_settings* s, int32_t P, int32_t M, const float* xyz,
                                const float* features_dc, const float* features_rest, const float* raw_opacity,
                                const float* raw_scaling, const float* raw_rotation, void* geom, size_t geom_bytes,
                                int32_t* radii, int64_t* num_rendered_host, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    if (P > 0 && (!features_dc || (M > 1 && !features_rest) || !raw_scaling || !raw_rotation))
        return fail(MVI_EINVAL, "NULL raw parameter pointer%s");
    f.raw = 1;
    f.shs_rest = features_rest;
    return forward_geom_impl(s, f, P, M, xyz, features_dc, nullptr, raw_opacity, raw_scaling, raw_rotation, nullptr, geom,
                             geom_bytes, radii, num_rendered_host, stream);
}

static int forward_geom_impl(const mvi_raster_settings* s, mvi::Frame& f, int32_t P, int32_t M, const float* means3D,
                             const float* shs, const float* colors_precomp, const float* opacities,
                             const float* scales, const float* rotations, const float* cov3D_precomp,
                             void* geom, size_t geom_bytes, int32_t* radii, int64_t* num_rendered_host,
                             void* stream) {
    if (!num_rendered_host) return fail(MVI_EINVAL, "num_rendered_host is NULL%s");
    *num_rendered_host = 0;
    if (P == 0) return MVI_OK;
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide excatly one of either SHs or precomputed colors!%s");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!%s");
    if (!means3D || !opacities || !geom || !radii) return fail(MVI_EINVAL, "NULL means3D/opacities/geom/radii%s");
    if (shs && M < (s->sh_degree + 1) * (s->sh_degree + 1))
        return fail(MVI_EINVAL, "shs holds %s%lld coefficients per channel, sh_degree needs %lld", "", M,
                    (s->sh_degree + 1) * (s->sh_degree + 1));
    mvi::GeomView g = mvi::carve_geom(geom, P);
    if (geom_bytes < g.bytes) return fail(MVI_ENOMEM, "geom scratch too small%s: %lld < %lld", "", (long long)geom_bytes, (long long)g.bytes);
    hipStream_t st = (hipStream_t)stream;
    {
        mvi::StageTimer tm(mvi::kStPreFwd, st);
        if (mvi::launch_preprocess_forward(f, means3D, shs, colors_precomp, opacities, scales, rotations,
                                           cov3D_precomp, g, radii, st))
            return hip_fail("preprocess_forward", hipGetLastError());
    }
    // num_rendered comes back through a pinned, device-mapped word that the totalling kernel writes itself + an event
    // recorded right behind it; binning level 1 (independent of num_rendered) is queued before the host waits: the host
    // wakes up as soon as the count is there and allocates / launches stage 2 while the device is still sorting
    // The word and the event belong to ONE device: a thread that drives several GPUs gets one set per device (the mapped
    // pointer is only valid on the device it was obtained for, and an event cannot be recorded on another device's stream).
    struct Readback { unsigned long long* host = nullptr; unsigned long long* dev = nullptr; hipEvent_t ev = nullptr; };
    constexpr int kMaxDevices = 64;
    static thread_local Readback readbacks[kMaxDevices];
    hipError_t e;
    int device = -1;
    if ((e = hipGetDevice(&device)) != hipSuccess) return hip_fail("hipGetDevice", e);
    if (device < 0 || device >= kMaxDevices) return fail(MVI_EINVAL, "device index out of range%s: %lld", "", (long long)device);
    Readback& rb = readbacks[device];
    if (!rb.host) {
        unsigned long long* h = nullptr;
        if ((e = hipHostMalloc((void**)&h, 2 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocPortable)) != hipSuccess) return hip_fail("hipHostMalloc", e);
        if ((e = hipHostGetDevicePointer((void**)&rb.dev, h, 0)) != hipSuccess) { (void)hipHostFree(h); return hip_fail("hipHostGetDevicePointer", e); }
        if ((e = hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming)) != hipSuccess) { (void)hipHostFree(h); return hip_fail("hipEventCreate", e); }
        rb.host = h;
    }
    unsigned long long* const pinned = rb.host;
    unsigned long long* const pinned_dev = rb.dev;
    const hipEvent_t ev = rb.ev;
    {
        mvi::StageTimer tm(mvi::kStScan, st);
        // binning version 2 also totals the column segments (pinned[1]): they size the grids of its second pass
        if (f.bin_v2 ? mvi::launch_binning2_totals(g, P, pinned_dev, st) : mvi::launch_scan_block_sums(g, P, pinned_dev, st))
            return hip_fail("scan_block_sums", hipGetLastError());
    }
    if ((e = hipEventRecord(ev, st)) != hipSuccess) return hip_fail("event record", e);
    if (f.bin_v2 ? mvi::launch_binning2_level1(f, g, st) : mvi::launch_binning_level1(f, g, st))
        return hip_fail("binning level 1", hipGetLastError());
    e = hipEventSynchronize(ev);
    if (e != hipSuccess) return hip_fail("forward_geom sync", e);
    const unsigned long long total = *pinned;
    // pair offsets, tile ranges and the sort's scatter positions are 32-bit: more pairs than that cannot be binned
    if (total > 0xFFFFFFFFull)
        return fail(MVI_EINVAL, "num_rendered exceeds the 32-bit pair offsets%s: %lld pairs", "", (long long)total);
    *num_rendered_host = (int64_t)total;
    if (f.bin_v2) remember_segments(geom, P, (int64_t)pinned[1]);
    return MVI_OK;
}

static int forward_render_impl(const mvi_raster_settings* s, int32_t P, int64_t D, const int32_t* radii,
                               void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                               size_t image_bytes, float* out_color, float* out_depth, float* grad_rows_to_zero, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, 0, f)) return rc;
    if (!image || !out_color || !out_depth) return fail(MVI_EINVAL, "NULL image/out_color/out_depth%s");
    if (D < 0 || D > 0xFFFFFFFFll) return fail(MVI_EINVAL, "num_rendered out of range%s: %lld", "", D);
    if (D > 0 && (!geom || !binning || !radii)) return fail(MVI_EINVAL, "NULL geom/binning/radii with num_rendered > 0%s");
    mvi::GeomView g = mvi::carve_geom(geom, P);
    mvi::ImageView im = mvi::carve_image(image, f.W, f.H);
    mvi::BinningView b = mvi::carve_binning(binning, D, f.W, f.H);
    if (image_bytes < im.bytes) return fail(MVI_ENOMEM, "image scratch too small%s: %lld < %lld", "", (long long)image_bytes, (long long)im.bytes);
    if (D > 0 && binning_bytes < b.bytes) return fail(MVI_ENOMEM, "binning scratch too small%s: %lld < %lld", "", (long long)binning_bytes, (long long)b.bytes);
    if (D > 0 && geom_bytes < g.bytes) return fail(MVI_ENOMEM, "geom scratch too small%s", "");
    hipStream_t st = (hipStream_t)stream;
    const int64_t segments = recall_segments(geom, P);
    if (f.bin_v2 ? mvi::launch_binning2(f, g, b, im, D, segments, st) : mvi::launch_binning(f, g, radii, b, im, D, st))
        return hip_fail("binning", hipGetLastError());
    {
        mvi::StageTimer tm(mvi::kStRenderFwd, st);
        if (mvi::launch_render_forward(f, g, b, im, D, out_color, out_depth, st, grad_rows_to_zero))
            return hip_fail("render_forward", hipGetLastError());
    }
    return MVI_OK;
}

int mvi_raster_forward_render(const mvi_raster_settings* s, int32_t P, int64_t D, const int32_t* radii,
                              void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                              size_t image_bytes, float* out_color, float* out_depth, void* stream) {
    return forward_render_impl(s, P, D, radii, geom, geom_bytes, binning, binning_bytes, image, image_bytes, out_color, out_depth,
                               nullptr, stream);
}

int mvi_raster_forward_render_prepare(const mvi_raster_settings* s, int32_t P, int64_t D, const int32_t* radii,
                                      void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                                      size_t image_bytes, float* out_color, float* out_depth, float* grad_rows_scratch,
                                      void* stream) {
    return forward_render_impl(s, P, D, radii, geom, geom_bytes, binning, binning_bytes, image, image_bytes, out_color, out_depth,
                               grad_rows_scratch, stream);
}

static int backward_render_impl(mvi::Frame& f, int32_t P, int64_t D, const int32_t* radii, const void* geom, const void* binning,
                                const void* image, const float* dL_dout_color, float* grad_rows, float* dL_dcolor_factor,
                                int sh_input, int rows_prezeroed, void* stream, const mvi::ZeroRegions* zero_outputs = nullptr);
static int backward_impl(mvi::Frame& f, int32_t P, int64_t D, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const int32_t* radii, const void* geom, const void* binning,
                         const void* image, const float* dL_dout_color, float* dL_dmeans3D, float* dL_dmeans2D,
                         float* dL_dopacity, float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations,
                         float* dL_dcov3D, float* dL_dconic_scratch, void* stream, mvi::RawBackwardExtra rawx);

int mvi_raster_backward(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t D, const float* means3D,
                        const float* shs, const float* colors_precomp, const float* scales,
                        const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                        const void* geom, const void* binning, const void* image,
                        const float* dL_dout_color, float* dL_dmeans3D, float* dL_dmeans2D,
                        float* dL_dopacity, float* dL_dshs, float* dL_dcolors, float* dL_dscales,
                        float* dL_drotations, float* dL_dcov3D, float* dL_dconic_scratch, int32_t grad_rows_prezeroed,
                        void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    mvi::RawBackwardExtra rawx;
    rawx.rows_prezeroed = grad_rows_prezeroed;
    return backward_impl(f, P, D, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, radii, geom, binning, image,
                         dL_dout_color, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dshs, dL_dcolors, dL_dscales, dL_drotations,
                         dL_dcov3D, dL_dconic_scratch, stream, rawx);
}

int mvi_raster_backward_raw(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t D, const float* xyz,
                            const float* features_dc, const float* features_rest, const float* raw_opacity,
                            const float* raw_scaling, const float* raw_rotation, const int32_t* radii, const void* geom,
                            const void* binning, const void* image, const float* dL_dout_color, float* dL_dxyz,
                            float* dL_dmeans2D, float* dL_draw_opacity, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                            float* dL_draw_scaling, float* dL_draw_rotation, float* grad_rows_scratch,
                            int32_t grad_rows_prezeroed, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    if (P > 0 && (!features_dc || (M > 1 && (!features_rest || !dL_dfeatures_rest)) || !raw_opacity || !dL_dfeatures_dc))
        return fail(MVI_EINVAL, "NULL raw parameter / gradient pointer%s");
    f.raw = 1;
    f.shs_rest = features_rest;
    mvi::RawBackwardExtra rawx;
    rawx.raw_opacity = raw_opacity;
    rawx.dL_dshs_rest = dL_dfeatures_rest;
    rawx.rows_prezeroed = grad_rows_prezeroed;
    return backward_impl(f, P, D, xyz, features_dc, nullptr, raw_scaling, raw_rotation, nullptr, radii, geom, binning, image,
                         dL_dout_color, dL_dxyz, dL_dmeans2D, dL_draw_opacity, dL_dfeatures_dc, nullptr, dL_draw_scaling,
                         dL_draw_rotation, nullptr, grad_rows_scratch, stream, rawx);
}

int mvi_raster_backward_raw_factor(const mvi_raster_settings* s, int32_t P, int32_t M, int64_t D, const float* xyz,
                                   const float* features_dc, const float* features_rest, const float* raw_opacity,
                                   const float* raw_scaling, const float* raw_rotation, const int32_t* radii, const void* geom,
                                   const void* binning, const void* image, const float* dL_dout_color, float* dL_dxyz,
                                   float* dL_dmeans2D, float* dL_draw_opacity, float* dL_dsh_color_factor, float* dL_draw_scaling,
                                   float* dL_draw_rotation, float* grad_rows_scratch, int32_t grad_rows_prezeroed, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    if (P > 0 && (!features_dc || (M > 1 && !features_rest) || !raw_opacity || !dL_dsh_color_factor))
        return fail(MVI_EINVAL, "NULL raw parameter / colour factor pointer%s");
    f.raw = 1;
    f.shs_rest = features_rest;
    mvi::RawBackwardExtra rawx;
    rawx.raw_opacity = raw_opacity;
    rawx.dL_dshs_rest = nullptr;                 // no dense SH gradient: every kernel tests dL_dshs before it touches either half
    rawx.rows_prezeroed = grad_rows_prezeroed;
    return backward_impl(f, P, D, xyz, features_dc, nullptr, raw_scaling, raw_rotation, nullptr, radii, geom, binning, image,
                         dL_dout_color, dL_dxyz, dL_dmeans2D, dL_draw_opacity, nullptr, dL_dsh_color_factor, dL_draw_scaling,
                         dL_draw_rotation, nullptr, grad_rows_scratch, stream, rawx);
}

static int backward_impl(mvi::Frame& f, int32_t P, int64_t D, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const int32_t* radii, const void* geom, const void* binning,
                         const void* image, const float* dL_dout_color, float* dL_dmeans3D, float* dL_dmeans2D,
                         float* dL_dopacity, float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations,
                         float* dL_dcov3D, float* dL_dconic_scratch, void* stream, mvi::RawBackwardExtra rawx) {
    if (P == 0) return MVI_OK;
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide excatly one of either SHs or precomputed colors!%s");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!%s");
    if (!means3D || !radii || !geom || !image || !dL_dout_color || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity ||
        !dL_dconic_scratch)
        return fail(MVI_EINVAL, "NULL required pointer in backward%s");
    if (shs && !dL_dshs && !dL_dcolors)
        return fail(MVI_EINVAL, "dL_dshs and dL_dcolors are both NULL but shs was a forward input%s");
    if (colors_precomp && !dL_dcolors) return fail(MVI_EINVAL, "dL_dcolors is NULL but colors_precomp was a forward input%s");
    if (cov3D_precomp ? !dL_dcov3D : (!dL_dscales || !dL_drotations))
        return fail(MVI_EINVAL, "missing covariance gradient output%s");
    if (D > 0 && !binning) return fail(MVI_EINVAL, "NULL binning with num_rendered > 0%s");
    // One-call backward: every gradient output is zeroed by the render backward on the side (it is bound by vector issue, its
    // memory pipes are idle), and the per-Gaussian chain rule then touches only the rows that received a gradient — a few per
    // cent of the Gaussians of a large scene (the others are occluded in this view). MVI_RASTER_DENSE_BACKWARD=1 keeps the
    // dense kernel (A/B runs).
    const bool dense_bwd = g_dense_backward != 0;
    mvi::GeomView g = mvi::carve_geom(const_cast<void*>(geom), P);
    hipStream_t st = (hipStream_t)stream;
    if (!dense_bwd) {
        mvi::ZeroRegions z;
        const size_t n = (size_t)P, Mz = (size_t)f.M;
        z.add(dL_dmeans3D, 3 * n);
        z.add(dL_dmeans2D, 3 * n);
        z.add(dL_dopacity, n);
        if (dL_dcolors) z.add(dL_dcolors, 3 * n);
        if (shs && dL_dshs) z.add(dL_dshs, f.raw ? 3 * n : 3 * Mz * n);
        if (f.raw && rawx.dL_dshs_rest && Mz > 1) z.add(rawx.dL_dshs_rest, (3 * Mz - 3) * n);
        if (cov3D_precomp) z.add(dL_dcov3D, 6 * n);
        else { z.add(dL_dscales, 3 * n); z.add(dL_drotations, 4 * n); }
        if (z.overflow) return fail(MVI_EINVAL, "more gradient outputs than the render backward can zero on the side%s (kMaxZero)");
        if (int rc = backward_render_impl(f, P, D, radii, geom, binning, image, dL_dout_color, dL_dconic_scratch, nullptr, 0,
                                          rawx.rows_prezeroed, stream, &z)) return rc;
        mvi::StageTimer tm(mvi::kStPreBwd, st);
        if (mvi::launch_preprocess_backward_sparse(f, means3D, shs, scales, rotations, cov3D_precomp, g, dL_dconic_scratch,
                                                   dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dshs, dL_dcov3D,
                                                   dL_dscales, dL_drotations, st, rawx))
            return hip_fail("preprocess_backward (sparse)", hipGetLastError());
        return MVI_OK;
    }
    if (int rc = backward_render_impl(f, P, D, radii, geom, binning, image, dL_dout_color, dL_dconic_scratch, nullptr, 0,
                                      rawx.rows_prezeroed, stream)) return rc;
    mvi::StageTimer tm(mvi::kStPreBwd, st);
    if (mvi::launch_preprocess_backward(f, means3D, shs, scales, rotations, cov3D_precomp, radii, g, dL_dconic_scratch,
                                        dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors,
                                        dL_dshs, dL_dcov3D, dL_dscales, dL_drotations, st, rawx))
        return hip_fail("preprocess_backward", hipGetLastError());
    return MVI_OK;
}

// first half of the backward: zero the accumulation rows, render backward, optionally the colour factors
static int backward_render_impl(mvi::Frame& f, int32_t P, int64_t D, const int32_t* radii, const void* geom, const void* binning,
                                const void* image, const float* dL_dout_color, float* grad_rows, float* dL_dcolor_factor,
                                int sh_input, int rows_prezeroed, void* stream, const mvi::ZeroRegions* zero_outputs) {
    if (P == 0) return MVI_OK;
    if (!radii || !geom || !image || !dL_dout_color || !grad_rows) return fail(MVI_EINVAL, "NULL required pointer in backward%s");
    if (D > 0 && !binning) return fail(MVI_EINVAL, "NULL binning with num_rendered > 0%s");
    mvi::GeomView g = mvi::carve_geom(const_cast<void*>(geom), P);
    mvi::ImageView im = mvi::carve_image(const_cast<void*>(image), f.W, f.H);
    mvi::BinningView b = mvi::carve_binning(const_cast<void*>(binning), D, f.W, f.H);
    hipStream_t st = (hipStream_t)stream;
    if (!rows_prezeroed) {      // accumulation rows + the touched flags (the forward zeroes both when asked to prepare a backward)
        mvi::ZeroRegions z;
        z.add(grad_rows, (size_t)mvi::kGradRow * (size_t)P);
        z.add(g.touched, ((size_t)P + 15) / 16 * 4);
        z.add(g.touched_count, 1);
        if (mvi::launch_zero_regions(z, st)) return hip_fail("zero grad rows", hipGetLastError());
    }
    {
        mvi::StageTimer tm(mvi::kStRenderBwd, st);
        if (mvi::launch_render_backward(f, g, b, im, D, dL_dout_color, grad_rows, st, zero_outputs))
            return hip_fail("render_backward", hipGetLastError());
    }
    if (dL_dcolor_factor && mvi::launch_color_factors(P, radii, grad_rows, sh_input ? g.clamped : nullptr, dL_dcolor_factor, st))
        return hip_fail("color_factors", hipGetLastError());
    return MVI_OK;
}

int mvi_raster_backward_render(const mvi_raster_settings* s, int32_t P, int64_t D, const int32_t* radii, const void* geom,
                               const void* binning, const void* image, const float* dL_dout_color, float* grad_rows_scratch,
                               float* dL_dcolor_factor, int32_t sh_input, int32_t grad_rows_prezeroed, void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, 0, f)) return rc;
    return backward_render_impl(f, P, D, radii, geom, binning, image, dL_dout_color, grad_rows_scratch, dL_dcolor_factor,
                                sh_input, grad_rows_prezeroed, stream);
}

static int backward_geom_range_impl(const mvi_raster_settings* s, int32_t P, int32_t M, int32_t first, int32_t count,
                                    const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                                    const float* rotations, const float* cov3D_precomp, const int32_t* radii, const void* geom,
                                    const float* grad_rows_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                                    float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                                    void* stream) {
    mvi::Frame f;
    if (int rc = make_frame(s, P, M, f)) return rc;
    if (first < 0 || count < 0 || (int64_t)first + count > P || (first % 64) != 0)
        return fail(MVI_EINVAL, "backward range out of bounds or not 64-aligned%s: first %lld count %lld", "", (long long)first, (long long)count);
    if (P == 0 || count == 0) return MVI_OK;
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide excatly one of either SHs or precomputed colors!%s");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(MVI_EINVAL, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!%s");
    if (!means3D || !radii || !geom || !grad_rows_scratch || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity)
        return fail(MVI_EINVAL, "NULL required pointer in backward%s");
    if (shs && !dL_dshs && !dL_dcolors)
        return fail(MVI_EINVAL, "dL_dshs and dL_dcolors are both NULL but shs was a forward input%s");
    if (colors_precomp && !dL_dcolors) return fail(MVI_EINVAL, "dL_dcolors is NULL but colors_precomp was a forward input%s");
    if (cov3D_precomp ? !dL_dcov3D : (!dL_dscales || !dL_drotations))
        return fail(MVI_EINVAL, "missing covariance gradient output%s");
    // per-Gaussian views of the scratch (carved for the full P) and every caller array, moved to the range's first row
    mvi::GeomView g = mvi::carve_geom(const_cast<void*>(geom), P);
    const size_t o = (size_t)first;
    g.depths += o; g.xy += o; g.cov_a += o; g.cov_b += o; g.conic_opacity += o; g.rgbd += o; g.tiles_touched += o; g.rect += o;
    g.clamped += o;
    auto at = [o](auto* p, size_t w) { return p ? p + o * w : p; };
    f.P = count;
    hipStream_t st = (hipStream_t)stream;
    mvi::StageTimer tm(mvi::kStPreBwd, st);
    if (mvi::launch_preprocess_backward(f, at(means3D, 3), at(shs, 3 * (size_t)M), at(scales, 3), at(rotations, 4), at(cov3D_precomp, 6),
                                        at(radii, 1), g, at(grad_rows_scratch, 16), at(dL_dmeans3D, 3), at(dL_dmeans2D, 3),
                                        at(dL_dopacity, 1), at(dL_dcolors, 3), at(dL_dshs, 3 * (size_t)M), at(dL_dcov3D, 6),
                                        at(dL_dscales, 3), at(dL_drotations, 4), st, mvi::RawBackwardExtra()))
        return hip_fail("preprocess_backward", hipGetLastError());
    return MVI_OK;
}

int mvi_raster_backward_geom(const mvi_raster_settings* s, int32_t P, int32_t M, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations,
                             const float* cov3D_precomp, const int32_t* radii, const void* geom,
                             const float* grad_rows_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                             float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                             void* stream) {
    return backward_geom_range_impl(s, P, M, 0, P, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, radii, geom,
                                    grad_rows_scratch, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dshs, dL_dcolors, dL_dscales,
                                    dL_drotations, dL_dcov3D, stream);
}

int mvi_raster_backward_geom_range(const mvi_raster_settings* s, int32_t P, int32_t M, int32_t first, int32_t count,
                                   const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, const int32_t* radii, const void* geom,
                                   const float* grad_rows_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity,
                                   float* dL_dshs, float* dL_dcolors, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                                   void* stream) {
    return backward_geom_range_impl(s, P, M, first, count, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, radii,
                                    geom, grad_rows_scratch, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dshs, dL_dcolors, dL_dscales,
                                    dL_drotations, dL_dcov3D, stream);
}

int mvi_raster_sh_backward_views(int32_t P, int32_t M, int32_t sh_degree, int32_t n_views, const float* means3D,
                                 const float* campos, int64_t campos_stride, const float* dL_dcolors,
                                 int64_t colors_view_stride, float* dL_dshs, void* stream) {
    if (P < 0 || M < 1 || M > 16 || sh_degree < 0 || (sh_degree + 1) * (sh_degree + 1) > M || n_views < 1)
        return fail(MVI_EINVAL, "bad shape for sh_backward_views (need 1 <= (deg+1)^2 <= M <= 16, n_views >= 1)%s");
    if (P == 0) return MVI_OK;
    if (!means3D || !campos || !dL_dcolors || !dL_dshs) return fail(MVI_EINVAL, "NULL pointer in sh_backward_views%s");
    if (campos_stride < 3 || colors_view_stride < 3 * (int64_t)P)
        return fail(MVI_EINVAL, "sh_backward_views: campos_stride must be >= 3 and colors_view_stride >= 3 P%s");
    if (mvi::launch_sh_backward_views(P, M, sh_degree, n_views, means3D, campos, campos_stride, dL_dcolors,
                                      colors_view_stride, dL_dshs, (hipStream_t)stream))
        return hip_fail("sh_backward_views", hipGetLastError());
    return MVI_OK;
}

int mvi_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* visible, void* stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !visible))) return fail(MVI_EINVAL, "bad arguments to mark_visible%s");
    if (mvi::launch_mark_visible(P, means3D, viewmatrix, visible, (hipStream_t)stream)) return hip_fail("mark_visible", hipGetLastError());
    return MVI_OK;
}

int mvi_raster_get_views(int32_t P, int64_t D, int32_t W, int32_t H, const void* geom, const void* binning,
                         const void* image, mvi_raster_views* out) {
    if (!out) return fail(MVI_EINVAL, "views is NULL%s");
    memset(out, 0, sizeof(*out));
    if (geom) {
        mvi::GeomView g = mvi::carve_geom(const_cast<void*>(geom), P);
        out->depths = g.depths; out->means2D = reinterpret_cast<const float*>(g.xy);
        out->cov3D_a = reinterpret_cast<const float*>(g.cov_a); out->cov3D_b = reinterpret_cast<const float*>(g.cov_b);
        out->conic_opacity = reinterpret_cast<const float*>(g.conic_opacity); out->rgbd = reinterpret_cast<const float*>(g.rgbd);
        out->tiles_touched = g.tiles_touched; out->clamped = g.clamped;
        out->grad_support = g.touched;
    }
    if (binning) {
        mvi::BinningView b = mvi::carve_binning(const_cast<void*>(binning), D, W, H);
        int fin = b.passes & 1;
        out->tile_ids_sorted = b.keys[fin]; out->point_list = b.vals[fin];
        out->tile_id_bytes = b.key_bytes;
    }
    if (image) {
        mvi::ImageView im = mvi::carve_image(const_cast<void*>(image), W, H);
        out->ranges = im.ranges; out->final_T = im.final_T; out->n_contrib = im.n_contrib;
    }
    return MVI_OK;
}

}  // extern "C"
