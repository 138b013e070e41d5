// gsr_api.hip -- the C ABI (include/gsr.h) and the host orchestration of one forward call.
//
// Mirrors what CudaRasterizer::Rasterizer::forward does on the host
// (DGR/cuda_rasterizer/rasterizer_impl.cu:197-339): carve scratch out of caller-provided arenas
// (rasterizer_impl.h:22-27,66-72), run the stages, read num_rendered back once to size the binning
// arena (:282), return it.  The stage list itself is this library's own (gsr_kernels.hip, gsr_radix.hip;
// gsr_sort.hip holds the rocPRIM passes kept as the GSR_OPT_SORT_IMPL = 0 comparison path).
#include "../../include/gsr.h"
#include "gsr_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

thread_local char g_error[512] = "";
thread_local size_t g_geom_off[GSR_GEOM_NUM_SLOTS];
thread_local size_t g_bin_off[GSR_BIN_NUM_SLOTS];
thread_local size_t g_img_off[GSR_IMG_NUM_SLOTS];
thread_local bool g_have_offsets = false;
thread_local uint32_t g_counts[2] = {0, 0};  // last call: {num_rendered (reference), live pairs}

// Stage timing: a ring of event sets so a whole timed region can be averaged afterwards without
// synchronising between calls.
constexpr int kTimingRing = 256;
constexpr int kEventsPerCall = GSR_STAGE_NUM + 1;
int g_options[GSR_OPT_NUM] = {/*GSR_OPT_TILE_CULL*/ 1, /*GSR_OPT_BLEND_VARIANT*/ 1, /*GSR_OPT_BLEND_LDS_PAD*/ 0,
                              /*GSR_OPT_SORT_IMPL*/ 1};
bool g_timing = false;
thread_local hipEvent_t g_ev[kTimingRing][kEventsPerCall];
thread_local bool g_ev_made = false;
thread_local long g_timed_calls = 0;   // completed timed calls since timing was (re)enabled
thread_local long g_begun_calls = 0;   // timed calls begun since then: several may be in flight (split calls), each owns a slot
thread_local int g_inflight = 0;       // timed calls begun and not yet finished / cancelled
thread_local bool g_stamp = false;     // whether the call being queued records events
thread_local int g_slot = 0;           // ring slot of the call being queued

int fail(gsr_status code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return (int)code;
}

#define GSR_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(GSR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// After each stage in debug mode: the CHECK_CUDA of auxiliary.h:166-173.
#define GSR_STAGE_CHECK(name)                                                                  \
    do {                                                                                       \
        if (debug) {                                                                           \
            hipError_t e_ = hipStreamSynchronize(stream);                                      \
            if (e_ != hipSuccess)                                                              \
                return fail(GSR_ERR_HIP, "stage %s failed: %s", name, hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)

// Bump carver over a caller arena; every sub-array starts on a 256-byte boundary *relative to the
// arena base* so layouts are reproducible, and the arena is requested with 256 bytes of slack so
// the base itself can be rounded up.
struct Carver {
    size_t off = 0;
    template <typename T>
    size_t take(size_t count) {
        off = (off + 255) & ~size_t(255);
        const size_t at = off;
        off += count * sizeof(T);
        return at;
    }
    size_t total() const { return ((off + 255) & ~size_t(255)) + 256; }
};

char* align_base(char* p) {
    return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255));
}

// What gsr_backward needs to know about the three arenas of a forward call is written into their headers on the
// device; reading it back costs a copy and a stream drain per training iteration, which the reference's backward
// does not have.  The same headers are therefore remembered on the host, keyed by the three aligned arena
// addresses: a backward call that finds its buffers here (the normal case: same process, buffers untouched since)
// never waits for the GPU.  Anything else -- buffers from another process, an entry pushed out of the small
// table -- takes the validated device read.  An arena that is reused by a later forward call replaces its entry.
struct HeaderRecord { const char* base[3]; gsr::ArenaHeader h[3]; };
constexpr int kHeaderRecords = 64;
std::mutex g_header_mutex;
HeaderRecord g_header_records[kHeaderRecords];
int g_header_next = 0;

void remember_headers(const char* const base[3], const gsr::ArenaHeader h[3]) {
    std::lock_guard<std::mutex> lock(g_header_mutex);
    int at = -1;
    for (int i = 0; i < kHeaderRecords; ++i)  // an arena that is written again invalidates what was known about it
        for (int k = 0; k < 3; ++k)
            if (g_header_records[i].base[k] == base[k]) { g_header_records[i].base[0] = g_header_records[i].base[1] = g_header_records[i].base[2] = nullptr; at = i; }
    if (at < 0) { at = g_header_next; g_header_next = (g_header_next + 1) % kHeaderRecords; }
    for (int k = 0; k < 3; ++k) { g_header_records[at].base[k] = base[k]; g_header_records[at].h[k] = h[k]; }
}

bool recall_headers(const char* const base[3], gsr::ArenaHeader h[3]) {
    std::lock_guard<std::mutex> lock(g_header_mutex);
    for (int i = 0; i < kHeaderRecords; ++i)
        if (g_header_records[i].base[0] == base[0] && g_header_records[i].base[1] == base[1] && g_header_records[i].base[2] == base[2] && base[0]) {
            for (int k = 0; k < 3; ++k) h[k] = g_header_records[i].h[k];
            return true;
        }
    return false;
}

// Bits of tile id to sort on: position of the highest set bit of T, plus one
// (rasterizer_impl.cu:35-50 getHigherMsb restated as a plain loop).
int tile_key_bits(uint32_t num_tiles) {
    int b = 0;
    while (b < 32 && (num_tiles >> b) != 0) ++b;
    return b;  // bit_length(T) >= bit_length(T - 1): every tile id fits
}

constexpr size_t kCounterBytes = gsr::kCounterCopyBytes;  // what travels back to the host

void stamp(int idx, hipStream_t s) {
    if (!g_stamp) return;
    (void)hipEventRecord(g_ev[g_slot][idx], s);
}

} // namespace

extern "C" {

const char* gsr_last_error(void) { return g_error; }
int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_target_arch(void) { return "gfx950"; }

int gsr_set_option(int option, int value) {
    if (option < 0 || option >= GSR_OPT_NUM) return fail(GSR_ERR_INVALID_ARG, "unknown option %d", option);
    g_options[option] = value;
    return GSR_OK;
}
int gsr_get_option(int option) {
    if (option < 0 || option >= GSR_OPT_NUM) return fail(GSR_ERR_INVALID_ARG, "unknown option %d", option);
    return g_options[option];
}

void gsr_set_stage_timing(int enable) {
    g_timing = enable != 0;
    g_timed_calls = 0;
    g_begun_calls = 0;
    g_inflight = 0;
}

int gsr_get_stage_times(float ms[GSR_STAGE_NUM]) {
    for (int i = 0; i < GSR_STAGE_NUM; ++i) ms[i] = 0.f;
    if (g_timed_calls <= 0) return fail(GSR_ERR_INVALID_ARG, "no timed gsr_forward call on this thread");
    if (g_inflight != 0) return fail(GSR_ERR_INVALID_ARG, "a split call is still in flight on this thread");
    const int ncalls = (int)(g_timed_calls < kTimingRing ? g_timed_calls : kTimingRing);
    double sum[GSR_STAGE_NUM] = {0};
    for (int c = 0; c < ncalls; ++c) {
        const int slot = (int)((g_begun_calls - 1 - c) % kTimingRing);
        GSR_HIP(hipEventSynchronize(g_ev[slot][kEventsPerCall - 1]));
        for (int i = 0; i < GSR_STAGE_NUM; ++i) {
            float t = 0.f;
            GSR_HIP(hipEventElapsedTime(&t, g_ev[slot][i], g_ev[slot][i + 1]));
            sum[i] += t;
        }
    }
    for (int i = 0; i < GSR_STAGE_NUM; ++i) ms[i] = (float)(sum[i] / ncalls);
    return ncalls;
}

int gsr_get_call_times(float* ms, int capacity) {
    if (!ms || capacity < 0) return fail(GSR_ERR_INVALID_ARG, "bad arguments");
    if (g_timed_calls <= 0) return 0;
    if (g_inflight != 0) return fail(GSR_ERR_INVALID_ARG, "a split call is still in flight on this thread");
    int ncalls = (int)(g_timed_calls < kTimingRing ? g_timed_calls : kTimingRing);
    if (ncalls > capacity) ncalls = capacity;
    for (int c = 0; c < ncalls; ++c) {
        const int slot = (int)((g_begun_calls - 1 - c) % kTimingRing);
        GSR_HIP(hipEventSynchronize(g_ev[slot][kEventsPerCall - 1]));
        GSR_HIP(hipEventElapsedTime(&ms[c], g_ev[slot][0], g_ev[slot][kEventsPerCall - 1]));
    }
    return ncalls;
}

int gsr_last_geom_offsets(size_t o[GSR_GEOM_NUM_SLOTS]) {
    if (!g_have_offsets) return fail(GSR_ERR_INVALID_ARG, "no gsr_forward call on this thread");
    memcpy(o, g_geom_off, sizeof g_geom_off);
    return GSR_OK;
}
int gsr_last_binning_offsets(size_t o[GSR_BIN_NUM_SLOTS]) {
    if (!g_have_offsets) return fail(GSR_ERR_INVALID_ARG, "no gsr_forward call on this thread");
    memcpy(o, g_bin_off, sizeof g_bin_off);
    return GSR_OK;
}
int gsr_last_pair_counts(uint32_t counts[2]) {
    if (!g_have_offsets) return fail(GSR_ERR_INVALID_ARG, "no gsr_forward call on this thread");
    counts[0] = g_counts[0];
    counts[1] = g_counts[1];
    return GSR_OK;
}
int gsr_last_image_offsets(size_t o[GSR_IMG_NUM_SLOTS]) {
    if (!g_have_offsets) return fail(GSR_ERR_INVALID_ARG, "no gsr_forward call on this thread");
    memcpy(o, g_img_off, sizeof g_img_off);
    return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream_) {
    (void)projmatrix;  // the reference passes it but only the view-space depth test is live (auxiliary.h:154)
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !present) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                 const char* binning_buffer, const char* image_buffer, const float* accum_alphas, const float* dL_dpix,
                 const float* dL_dpix_depth, const float* dL_dpix_alpha, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                 float* dL_dsh, float* dL_dscale, float* dL_drot, float* accum_scratch, int debug, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (P == 0) return GSR_OK;  // rasterize_points.cu:169: gradients stay as the binding zero-filled them
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(GSR_ERR_INVALID_ARG, "null scratch buffer");
    if (!means3D || !background || !viewmatrix || !projmatrix || !cam_pos || !accum_alphas || !dL_dpix ||
        !dL_dpix_depth || !dL_dpix_alpha || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_ddepth ||
        !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot || !accum_scratch)
        return fail(GSR_ERR_INVALID_ARG, "null required pointer");
    if (shs != nullptr && (M <= 0 || !dL_dsh)) return fail(GSR_ERR_INVALID_ARG, "shs given with M=%d or no dL_dsh", M);
    if ((scales != nullptr) != (rotations != nullptr) || (scales != nullptr) == (cov3D_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");

    const char* bases[3] = {align_base(const_cast<char*>(geom_buffer)), align_base(const_cast<char*>(binning_buffer)),
                            align_base(const_cast<char*>(image_buffer))};
    gsr::ArenaHeader h[3];
    if (debug || !recall_headers(bases, h)) {  // not a forward call this process remembers (or debug): read and validate
        for (int i = 0; i < 3; ++i) GSR_HIP(hipMemcpyAsync(&h[i], bases[i], sizeof h[i], hipMemcpyDeviceToHost, stream));
        GSR_HIP(hipStreamSynchronize(stream));
    }
    for (int i = 0; i < 3; ++i)
        if (h[i].magic != gsr::kArenaMagic || h[i].kind != (uint32_t)i)
            return fail(GSR_ERR_INVALID_ARG, "scratch buffer %d was not produced by gsr_forward", i);
    if (h[0].count[0] != (uint32_t)P || h[0].count[1] != (uint32_t)R || h[2].count[0] != (uint32_t)width ||
        h[2].count[1] != (uint32_t)height)
        return fail(GSR_ERR_INVALID_ARG, "scratch buffers belong to a different call (P %u/%d, R %u/%d, %ux%u/%dx%d)",
                    h[0].count[0], P, h[0].count[1], R, h[2].count[0], h[2].count[1], width, height);

    gsr::Camera cam;
    cam.viewmatrix = viewmatrix; cam.projmatrix = projmatrix; cam.cam_pos = cam_pos;
    cam.tan_fovx = tan_fovx; cam.tan_fovy = tan_fovy;
    cam.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:390-391
    cam.focal_x = width / (2.0f * tan_fovx);
    cam.width = width; cam.height = height;
    cam.grid_x = (width + gsr::kTile - 1) / gsr::kTile;
    cam.grid_y = (height + gsr::kTile - 1) / gsr::kTile;

    const gsr::SplatRaster* raster = (const gsr::SplatRaster*)(bases[0] + h[0].off[0]);
    const float* rgb = (const float*)(bases[0] + h[0].off[3]);
    if (radii == nullptr) radii = (const int*)(bases[0] + h[0].off[4]);
    const uint32_t* point_list = (const uint32_t*)(bases[1] + h[1].off[0]);
    const uint2* ranges = (const uint2*)(bases[2] + h[2].off[0]);
    const uint32_t* n_contrib = (const uint32_t*)(bases[2] + h[2].off[1]);
    const float* colors = colors_precomp != nullptr ? colors_precomp : rgb;  // rasterizer_impl.cu:399

    GSR_HIP(hipMemsetAsync(accum_scratch, 0, (size_t)P * 16 * sizeof(float), stream));
    GSR_HIP(gsr::launch_render_backward(cam, ranges, point_list, background, raster, colors, accum_alphas, n_contrib,
                                        dL_dpix, dL_dpix_depth, dL_dpix_alpha, accum_scratch, stream));
    GSR_STAGE_CHECK("render_backward");
    gsr::BackwardInputs b;
    b.P = P; b.sh_degree = D; b.M = M; b.means3D = means3D; b.radii = radii; b.shs = shs; b.scales = scales;
    b.rotations = rotations; b.cov3D_precomp = cov3D_precomp; b.scale_modifier = scale_modifier;
    b.accum = accum_scratch; b.dL_dmean2D = dL_dmean2D; b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity;
    b.dL_dcolor = dL_dcolor; b.dL_ddepth = dL_ddepth;
    b.dL_dmean3D = dL_dmean3D; b.dL_dcov3D = dL_dcov3D; b.dL_dsh = dL_dsh; b.dL_dscale = dL_dscale; b.dL_drot = dL_drot;
    GSR_HIP(gsr::launch_preprocess_backward(b, cam, stream));
    GSR_STAGE_CHECK("preprocess_backward");
    return GSR_OK;
}

int gsr_blend(int width, int height, const uint32_t* ranges, const uint32_t* point_list, const float* raster,
              const float* features, const float* background,
              float* out_color, float* out_depth, float* out_alpha, uint32_t* n_contrib, void* stream_) {
    if (width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!ranges || !point_list || !raster || !features || !background || !out_color ||
        !out_depth || !out_alpha)
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    gsr::Camera cam = {};
    cam.width = width; cam.height = height;
    cam.grid_x = (width + gsr::kTile - 1) / gsr::kTile;
    cam.grid_y = (height + gsr::kTile - 1) / gsr::kTile;
    GSR_HIP(gsr::launch_blend(cam, g_options[GSR_OPT_BLEND_VARIANT], g_options[GSR_OPT_BLEND_LDS_PAD], (const uint2*)ranges, point_list,
                              (const gsr::SplatRaster*)raster, features, background,
                              out_color, out_depth, out_alpha, n_contrib, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_composite(int width, int height, const uint8_t* bg_c, const uint8_t* o_c, const float* o_d, const uint8_t* s_c,
                  const float* s_d, const uint8_t* o_s_c, const uint8_t* o_gs_c, const float* o_gs_d,
                  const uint8_t* s_f_c, const float* s_f_d, const uint8_t* s_f_c_pre, uint8_t* out, void* stream_) {
    if (width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!bg_c || !o_c || !o_d || !s_c || !s_d || !o_s_c || !out) return fail(GSR_ERR_INVALID_ARG, "null required layer");
    if ((o_gs_c != nullptr) != (o_gs_d != nullptr) || (s_f_c != nullptr) != (s_f_d != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "a colour layer and its depth map must be given together");
    if (s_f_c_pre != nullptr && s_f_c == nullptr) return fail(GSR_ERR_INVALID_ARG, "fire layer without a smoke layer");
    GSR_HIP(gsr::launch_composite(width, height, bg_c, o_c, o_d, s_c, s_d, o_s_c, o_gs_c, o_gs_d, s_f_c, s_f_d,
                                  s_f_c_pre, out, (hipStream_t)stream_));
    return GSR_OK;
}

size_t gsr_radix_scratch_bytes(uint32_t n, int bits) {
    (void)bits;
    return gsr::radix_scratch_words(n) * sizeof(uint32_t);
}

int gsr_radix_sort_pairs(uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt, uint32_t* vals, uint32_t* vals_alt,
                         int iota_payload, void* scratch, size_t scratch_bytes, int* sorted_in_alt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (bits < 1 || bits > 32) return fail(GSR_ERR_INVALID_ARG, "bad key width bits=%d", bits);
    if (sorted_in_alt) *sorted_in_alt = 0;
    if (n == 0) return GSR_OK;
    if (!keys || !keys_alt || !vals_alt || (!vals && !iota_payload) || !scratch)
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if (scratch_bytes < gsr_radix_scratch_bytes(n, bits)) return fail(GSR_ERR_INVALID_ARG, "sort scratch too small");
    uint32_t *ks = nullptr, *vs = nullptr;
    GSR_HIP(gsr::radix_sort_pairs((uint32_t*)scratch, n, bits, keys, keys_alt, vals, vals_alt, iota_payload != 0, true, &ks,
                                  &vs, stream));
    if (sorted_in_alt) *sorted_in_alt = ks == keys_alt;
    return GSR_OK;
}

int gsr_view_normals(int P, const float* means3D, const float* axis, const float* cam_pos, float* colors, void* stream_) {
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "bad size P=%d", P);
    if (P == 0) return GSR_OK;
    if (!means3D || !axis || !cam_pos || !colors) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_view_normals(P, means3D, axis, cam_pos, colors, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_normal_maps(int width, int height, const float* normal_rgb, const float* depth, const float* c2w, float fx,
                    float fy, float cx, float cy, float* normal, float* pseudo_normal, void* stream_) {
    if (width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!normal_rgb || !depth || !c2w || !normal || !pseudo_normal) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_normal_maps(width, height, normal_rgb, depth, c2w, fx, fy, cx, cy, normal, pseudo_normal,
                                    (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_selftest_exp(uint32_t first_bits, uint32_t count, unsigned long long* device_mismatches, void* stream_) {
    if (!device_mismatches || count > 0x7FFFFFFFu) return fail(GSR_ERR_INVALID_ARG, "bad selftest arguments");
    if (count == 0) return GSR_OK;
    GSR_HIP(gsr::launch_exp_selftest(first_bits, count, device_mismatches, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_pack_rgba8(const float* color, const float* alpha, uint8_t* rgba8, int width, int height, void* stream_) {
    if (width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!color || !alpha || !rgba8) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_pack_rgba8(color, alpha, rgba8, (size_t)width * height, (hipStream_t)stream_));
    return GSR_OK;
}

} // extern "C"

namespace {
// One forward call, split where the host has to learn the pair count.  forward_begin queues the stages
// whose sizes depend only on P (projection, depth sort) and the copy of the frame counters;
// forward_finish waits for that copy, sizes the binning arena and queues the rest.  gsr_forward runs the
// two back to back; gsr_forward_begin / gsr_forward_finish expose the split so that one host thread can
// keep several frames in flight on several streams without blocking on the newest one.
struct PinnedSlot {
    uint32_t* host = nullptr;    // a few KB of pinned memory, pooled per calling thread and deliberately never
    hipEvent_t copied = nullptr; // freed: freeing at thread exit can race HIP runtime teardown
};
thread_local std::vector<PinnedSlot> g_pinned_free;

struct ForwardCall {
    hipStream_t stream = nullptr;
    int debug = 0, prefiltered = 0, P = 0, T = 0, width = 0, height = 0, slot = 0;
    bool own_sort = true, queued = false, timed = false;
    gsr::Camera cam;
    gsr::GeometryArrays ga;
    gsr_alloc_fn binning_alloc = nullptr;
    void* binning_user = nullptr;
    char *gbase = nullptr, *iraw = nullptr, *ibase = nullptr;
    size_t gshift = 0;
    size_t geom_off[GSR_GEOM_NUM_SLOTS] = {};
    size_t img_off[GSR_IMG_NUM_SLOTS] = {};
    uint32_t *order = nullptr, *point_offsets = nullptr, *tile_totals = nullptr;
    uint4* sorted_bins = nullptr;
    const float *colors_precomp = nullptr, *background = nullptr, *extra_features = nullptr;
    float *out_color = nullptr, *out_depth = nullptr, *out_alpha = nullptr, *out_extra = nullptr;
    PinnedSlot pinned;

    ~ForwardCall() {
        if (timed && g_inflight > 0) --g_inflight;  // (a call cancelled while timing is on leaves a stale slot in the ring)
        if (!pinned.host) return;
        if (queued) (void)hipEventSynchronize(pinned.copied);  // the copy may still be landing in the buffer
        g_pinned_free.push_back(pinned);
    }
};

int forward_begin(ForwardCall& fc, gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc,
                  void* binning_user, gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                  const float* background, int width, int height, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                  float* out_depth, float* out_alpha, int* radii, int debug, void* stream_,
                  const float* extra_features, float* out_extra) {
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", P, width, height);
    fc.P = P;
    if (P == 0) return 0;  // rasterize_points.cu:83: outputs stay as the binding zero-filled them
    if (!geom_alloc || !binning_alloc || !image_alloc) return fail(GSR_ERR_INVALID_ARG, "null scratch callback");
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !cam_pos || !out_color ||
        !out_depth || !out_alpha)
        return fail(GSR_ERR_INVALID_ARG, "null required pointer");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of shs / colors_precomp");
    const bool have_sr = scales != nullptr && rotations != nullptr;
    if ((scales != nullptr) != (rotations != nullptr) || have_sr == (cov3D_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (shs != nullptr && M <= 0) return fail(GSR_ERR_INVALID_ARG, "shs given with M=%d", M);

    gsr::Camera& cam = fc.cam;
    cam.viewmatrix = viewmatrix;
    cam.projmatrix = projmatrix;
    cam.cam_pos = cam_pos;
    cam.tan_fovx = tan_fovx;
    cam.tan_fovy = tan_fovy;
    cam.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:223-224
    cam.focal_x = width / (2.0f * tan_fovx);
    cam.width = width;
    cam.height = height;
    cam.grid_x = (width + gsr::kTile - 1) / gsr::kTile;
    cam.grid_y = (height + gsr::kTile - 1) / gsr::kTile;
    if (cam.grid_x > 65535 || cam.grid_y > 65535) return fail(GSR_ERR_INVALID_ARG, "image too large");
    const int T = cam.grid_x * cam.grid_y;
    const size_t n = (size_t)P;
    fc.stream = stream; fc.debug = debug; fc.prefiltered = prefiltered; fc.T = T; fc.width = width; fc.height = height;
    fc.binning_alloc = binning_alloc; fc.binning_user = binning_user;
    fc.colors_precomp = colors_precomp; fc.background = background; fc.extra_features = extra_features;
    fc.out_color = out_color; fc.out_depth = out_depth; fc.out_alpha = out_alpha; fc.out_extra = out_extra;

    if (g_timing && !g_ev_made) {
        for (auto& set : g_ev)
            for (auto& e : set) GSR_HIP(hipEventCreate(&e));
        g_ev_made = true;
    }
    g_slot = fc.slot = (int)(g_begun_calls % kTimingRing);  // one slot per call in flight (ring of 256: more than any driver keeps)
    g_stamp = fc.timed = g_timing;
    if (fc.timed) { ++g_begun_calls; ++g_inflight; }
    if (!g_pinned_free.empty()) {
        fc.pinned = g_pinned_free.back();
        g_pinned_free.pop_back();
    } else {
        GSR_HIP(hipHostMalloc((void**)&fc.pinned.host, kCounterBytes + 64, hipHostMallocPortable));
        GSR_HIP(hipEventCreateWithFlags(&fc.pinned.copied, hipEventDisableTiming));
    }

    // ---- geometry arena ----
    const bool own_sort = fc.own_sort = g_options[GSR_OPT_SORT_IMPL] != 0;
    size_t sort_tmp = 0, scan_tmp = 0;
    const size_t dup_blocks = (n + gsr::kDupTile - 1) / gsr::kDupTile;
    if (own_sort) {
        sort_tmp = gsr::radix_scratch_words((uint32_t)P) * sizeof(uint32_t);
    } else {
        GSR_HIP(gsr::depth_sort_temp_bytes(P, &sort_tmp));
        GSR_HIP(gsr::scan_temp_bytes(P, &scan_tmp));
    }
    size_t* geom_off = fc.geom_off;
    Carver gc;
    gc.take<gsr::ArenaHeader>(1);  // header at the arena's aligned base
    geom_off[GSR_GEOM_RASTER] = gc.take<gsr::SplatRaster>(n);
    geom_off[GSR_GEOM_RGB] = gc.take<float>(3 * n);
    geom_off[GSR_GEOM_SPLAT_BINS] = gc.take<gsr::SplatBin>(n);
    geom_off[GSR_GEOM_INTERNAL_RADII] = gc.take<int>(n);
    const size_t off_keys_a = gc.take<uint32_t>(n), off_keys_b = gc.take<uint32_t>(n);
    const size_t off_ids_a = gc.take<uint32_t>(n), off_ids_b = gc.take<uint32_t>(n);
    geom_off[GSR_GEOM_POINT_OFFSETS] = gc.take<uint32_t>(n);
    const size_t off_flag = gc.take<gsr::FrameCounters>(1);
    const size_t off_tile_totals = gc.take<uint32_t>(2 * dup_blocks);  // tile totals, then global offsets at tile ends
    const size_t off_sorted_bins = gc.take<uint4>(own_sort ? n : 0);  // splat records again, in depth order
    const size_t off_tmp = gc.take<char>(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
    char* graw = geom_alloc(gc.total(), geom_user);
    if (!graw) return fail(GSR_ERR_ALLOC, "geometry scratch callback returned NULL for %zu bytes", gc.total());
    char* gbase = fc.gbase = align_base(graw);
    fc.gshift = (size_t)(gbase - graw);

    // ---- image arena ----
    Carver ic;
    ic.take<gsr::ArenaHeader>(1);
    fc.img_off[GSR_IMG_RANGES] = ic.take<uint2>((size_t)T);
    fc.img_off[GSR_IMG_N_CONTRIB] = ic.take<uint32_t>((size_t)width * height);
    fc.iraw = image_alloc(ic.total(), image_user);
    if (!fc.iraw) return fail(GSR_ERR_ALLOC, "image scratch callback returned NULL for %zu bytes", ic.total());
    fc.ibase = align_base(fc.iraw);
    for (auto& o : fc.img_off) o += (size_t)(fc.ibase - fc.iraw);

    gsr::GaussianInputs in;
    in.P = P; in.sh_degree = D; in.M = M;
    in.means3D = means3D; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    in.opacities = opacities; in.shs = shs; in.colors_precomp = colors_precomp;
    in.scale_modifier = scale_modifier; in.prefiltered = prefiltered;
    in.tile_cull = g_options[GSR_OPT_TILE_CULL] != 0;

    gsr::GeometryArrays& ga = fc.ga;
    ga.raster = (gsr::SplatRaster*)(gbase + geom_off[GSR_GEOM_RASTER]);
    ga.rgb = (float*)(gbase + geom_off[GSR_GEOM_RGB]);
    ga.bins = (gsr::SplatBin*)(gbase + geom_off[GSR_GEOM_SPLAT_BINS]);
    ga.radii = radii ? radii : (int*)(gbase + geom_off[GSR_GEOM_INTERNAL_RADII]);
    ga.depth_keys = (uint32_t*)(gbase + off_keys_a);
    ga.ids = own_sort ? nullptr : (uint32_t*)(gbase + off_ids_a);  // the first radix pass generates 0..P-1 itself
    ga.counters = (gsr::FrameCounters*)(gbase + off_flag);
    fc.point_offsets = (uint32_t*)(gbase + geom_off[GSR_GEOM_POINT_OFFSETS]);
    fc.tile_totals = (uint32_t*)(gbase + off_tile_totals);
    fc.sorted_bins = (uint4*)(gbase + off_sorted_bins);
    void* tmp = gbase + off_tmp;
    const size_t tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;

    GSR_HIP(hipMemsetAsync(gbase + off_flag, 0, sizeof(gsr::FrameCounters), stream));
    stamp(0, stream);
    GSR_HIP(gsr::launch_preprocess(in, cam, ga, stream));
    GSR_STAGE_CHECK("preprocess");
    stamp(1, stream);

    // The one host round trip of the call (rasterizer_impl.cu:282 reads num_rendered back to size the
    // binning arena).  Here the totals come out of the preprocess kernel, so the copy is queued right
    // behind it and the host waits on an event while the GPU is already running the depth sort.
    GSR_HIP(hipMemcpyAsync(fc.pinned.host, gbase + off_flag, kCounterBytes, hipMemcpyDeviceToHost, stream));
    GSR_HIP(hipEventRecord(fc.pinned.copied, stream));
    fc.queued = true;

    uint32_t* keys_sorted = nullptr;
    if (own_sort) {
        GSR_HIP(gsr::radix_sort_pairs((uint32_t*)tmp, (uint32_t)P, 32, ga.depth_keys, (uint32_t*)(gbase + off_keys_b),
                                      (uint32_t*)(gbase + off_ids_a), (uint32_t*)(gbase + off_ids_b),
                                      /*iota_payload=*/true, /*want_sorted_keys=*/false, &keys_sorted, &fc.order, stream));
    } else {
        GSR_HIP(gsr::depth_sort(tmp, tmp_bytes, P, ga.depth_keys, (uint32_t*)(gbase + off_keys_b), ga.ids,
                                (uint32_t*)(gbase + off_ids_b), &keys_sorted, &fc.order, stream));
    }
    GSR_STAGE_CHECK("depth_sort");
    stamp(2, stream);
    geom_off[GSR_GEOM_DEPTH_ORDER] = (size_t)((char*)fc.order - gbase);

    if (!own_sort) GSR_HIP(gsr::scan_tiles_in_order(tmp, tmp_bytes, P, ga.bins, fc.order, fc.point_offsets, stream));
    return 0;
}

int forward_finish(ForwardCall& fc) {
    if (fc.P == 0) return 0;
    hipStream_t stream = fc.stream;
    const int debug = fc.debug, P = fc.P, T = fc.T;
    const bool own_sort = fc.own_sort;
    const gsr::Camera& cam = fc.cam;
    const gsr::GeometryArrays& ga = fc.ga;
    char* const gbase = fc.gbase;
    g_slot = fc.slot;
    g_stamp = fc.timed;

    GSR_HIP(hipEventSynchronize(fc.pinned.copied));
    fc.queued = false;
    const gsr::FrameCounters* hc = reinterpret_cast<const gsr::FrameCounters*>(fc.pinned.host);  // first kCounterBytes only
    const uint32_t flag = hc->error_flag;
    unsigned long long rect_total = 0, live_total = 0, emitting = 0;
    for (int i = 0; i < gsr::kRectPartials; ++i) {
        rect_total += hc->pair_totals[i] >> 32;
        live_total += hc->pair_totals[i] & 0xFFFFFFFFull;
        emitting += hc->visible[i];
    }
    if (debug && fc.prefiltered && (flag & 1u))
        return fail(GSR_ERR_PREFILTERED, "a Gaussian was culled although prefiltered is set (auxiliary.h:156-160)");
    if (rect_total > 0x7FFFFFFFull) return fail(GSR_ERR_INVALID_ARG, "num_rendered %llu overflows int", rect_total);
    const uint32_t num_live = (uint32_t)live_total;
    const uint32_t num_rendered = (uint32_t)rect_total;  // the reference's count; == num_live when culling is off
    stamp(3, stream);

    // ---- binning arena ----
    const size_t nr = num_live ? num_live : 1;
    size_t tsort_tmp = 0;
    const int tile_bits = tile_key_bits((uint32_t)T);
    const bool own_tile_sort = own_sort;
    if (own_tile_sort) tsort_tmp = gsr::radix_scratch_words((uint32_t)nr) * sizeof(uint32_t);
    else GSR_HIP(gsr::tile_sort_temp_bytes((uint32_t)nr, &tsort_tmp));
    Carver bc;
    bc.take<gsr::ArenaHeader>(1);
    const size_t off_tk_a = bc.take<uint32_t>(nr), off_tk_b = bc.take<uint32_t>(nr);
    const size_t off_pl_a = bc.take<uint32_t>(nr), off_pl_b = bc.take<uint32_t>(nr);
    const size_t off_btmp = bc.take<char>(tsort_tmp);
    char* braw = fc.binning_alloc(bc.total(), fc.binning_user);
    if (!braw) return fail(GSR_ERR_ALLOC, "binning scratch callback returned NULL for %zu bytes", bc.total());
    char* bbase = align_base(braw);
    uint32_t *tile_keys = (uint32_t*)(bbase + off_tk_a), *point_list = (uint32_t*)(bbase + off_pl_a);

    uint2* ranges = (uint2*)(fc.iraw + fc.img_off[GSR_IMG_RANGES]);
    uint32_t* n_contrib = (uint32_t*)(fc.iraw + fc.img_off[GSR_IMG_N_CONTRIB]);

    if (own_sort)  // also when nothing is live: POINT_OFFSETS is an output
        GSR_HIP(gsr::launch_scan_expand(P, (int)emitting, num_live, cam, fc.order, ga.bins, fc.sorted_bins, fc.tile_totals,
                                        fc.point_offsets, tile_keys, point_list, stream));
    if (num_live > 0) {
        if (!own_sort)
            GSR_HIP(gsr::launch_duplicate(P, cam, fc.order, fc.point_offsets, ga.bins, tile_keys, point_list, stream));
        GSR_STAGE_CHECK("duplicate");
        stamp(4, stream);
        uint32_t *tk_sorted = nullptr, *pl_sorted = nullptr;
        if (own_tile_sort) {
            GSR_HIP(gsr::radix_sort_pairs((uint32_t*)(bbase + off_btmp), num_live, tile_bits, tile_keys,
                                          (uint32_t*)(bbase + off_tk_b), point_list, (uint32_t*)(bbase + off_pl_b),
                                          /*iota_payload=*/false, /*want_sorted_keys=*/true, &tk_sorted, &pl_sorted, stream));
        } else {
            GSR_HIP(gsr::tile_sort(bbase + off_btmp, tsort_tmp, num_live, tile_bits, tile_keys,
                                   (uint32_t*)(bbase + off_tk_b), point_list, (uint32_t*)(bbase + off_pl_b),
                                   &tk_sorted, &pl_sorted, stream));
        }
        GSR_STAGE_CHECK("tile_sort");
        stamp(5, stream);
        tile_keys = tk_sorted;
        point_list = pl_sorted;
    } else {
        stamp(4, stream);
        stamp(5, stream);
    }

    {   // self-describing arenas for gsr_backward
        gsr::ArenaHeader hg = {}, hb = {}, hi = {};
        hg.magic = hb.magic = hi.magic = gsr::kArenaMagic;
        hg.kind = 0; hb.kind = 1; hi.kind = 2;
        hg.count[0] = (uint32_t)P; hg.count[1] = num_rendered; hg.count[2] = num_live;
        hg.off[0] = (uint64_t)((char*)ga.raster - gbase);
        hg.off[3] = (uint64_t)((char*)ga.rgb - gbase);
        hg.off[4] = (uint64_t)fc.geom_off[GSR_GEOM_INTERNAL_RADII];
        hb.count[0] = num_live;
        hb.off[0] = (uint64_t)((char*)point_list - bbase);
        hi.count[0] = (uint32_t)fc.width; hi.count[1] = (uint32_t)fc.height; hi.count[2] = (uint32_t)T;
        hi.off[0] = (uint64_t)((char*)ranges - fc.ibase);
        hi.off[1] = (uint64_t)((char*)n_contrib - fc.ibase);
        const gsr::ArenaHeader hs[3] = {hg, hb, hi};
        void* const dsts[3] = {gbase, bbase, fc.ibase};
        const char* const bases[3] = {gbase, bbase, fc.ibase};
        remember_headers(bases, hs);
        // tile ranges (all (0,0) when nothing is live: rasterizer_impl.cu:311) + the three headers, one launch
        GSR_HIP(gsr::launch_tile_ranges(num_live, T, tile_keys, ranges, dsts, hs, stream));
        GSR_STAGE_CHECK("tile_ranges");
        stamp(6, stream);
    }

    const float* features = fc.colors_precomp != nullptr ? fc.colors_precomp : ga.rgb;
    GSR_HIP(gsr::launch_blend(cam, g_options[GSR_OPT_BLEND_VARIANT], g_options[GSR_OPT_BLEND_LDS_PAD], ranges, point_list,
                              ga.raster, features, fc.background, fc.out_color, fc.out_depth, fc.out_alpha, n_contrib,
                              stream, fc.extra_features, fc.out_extra));
    GSR_STAGE_CHECK("blend");
    stamp(7, stream);
    if (fc.timed) ++g_timed_calls;

    // what the gsr_last_* accessors report: the call that finished last on this thread
    for (int i = 0; i < GSR_GEOM_NUM_SLOTS; ++i) g_geom_off[i] = fc.geom_off[i] + fc.gshift;
    memcpy(g_img_off, fc.img_off, sizeof g_img_off);
    g_bin_off[GSR_BIN_POINT_LIST] = (size_t)((char*)point_list - braw);
    g_bin_off[GSR_BIN_TILE_KEYS] = (size_t)((char*)tile_keys - braw);
    g_counts[0] = num_rendered;
    g_counts[1] = num_live;
    g_have_offsets = true;
    return (int)num_rendered;
}

#define GSR_FORWARD_ARGS                                                                                             \
    geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width, height, \
        means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,      \
        projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_alpha, radii, debug, stream
} // namespace

extern "C" {

int gsr_forward(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, float* out_alpha, int* radii, int debug, void* stream) {
    ForwardCall fc;
    const int rc = forward_begin(fc, GSR_FORWARD_ARGS, nullptr, nullptr);
    return rc < 0 ? rc : forward_finish(fc);
}

int gsr_forward_extra(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                      gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                      int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                      const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                      const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                      float* out_depth, float* out_alpha, int* radii, const float* extra_features, float* out_extra,
                      int debug, void* stream) {
    if (P > 0 && (!extra_features || !out_extra)) return fail(GSR_ERR_INVALID_ARG, "null extra feature pointer");
    ForwardCall fc;
    const int rc = forward_begin(fc, GSR_FORWARD_ARGS, extra_features, out_extra);
    return rc < 0 ? rc : forward_finish(fc);
}

void* gsr_forward_begin(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                        gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                        int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                        float* out_depth, float* out_alpha, int* radii, const float* extra_features, float* out_extra,
                        int debug, void* stream) {
    if ((extra_features != nullptr) != (out_extra != nullptr)) {
        fail(GSR_ERR_INVALID_ARG, "extra_features and out_extra must be given together");
        return nullptr;
    }
    ForwardCall* fc = new (std::nothrow) ForwardCall;
    if (!fc) {
        fail(GSR_ERR_ALLOC, "out of host memory");
        return nullptr;
    }
    if (forward_begin(*fc, GSR_FORWARD_ARGS, extra_features, out_extra) < 0) {
        delete fc;
        return nullptr;
    }
    return fc;
}

int gsr_forward_finish(void* call) {
    if (!call) return fail(GSR_ERR_INVALID_ARG, "null call handle");
    ForwardCall* fc = static_cast<ForwardCall*>(call);
    const int rc = forward_finish(*fc);
    delete fc;
    return rc;
}

void gsr_forward_cancel(void* call) { delete static_cast<ForwardCall*>(call); }

} // extern "C"
