// gsr_api.hip -- the C ABI (include/gsr.h) and the host orchestration of one forward call.
//
// Mirrors what CudaRasterizer::Rasterizer::forward does on the host
// (DGR/cuda_rasterizer/rasterizer_impl.cu:197-339): carve scratch out of caller-provided arenas
// (rasterizer_impl.h:22-27,66-72), run the stages, read num_rendered back once to size the binning
// arena (:282), return it.  The stage list itself is this library's own (gsr_kernels.hip, gsr_binning.hip,
// gsr_radix.hip, gsr_blend.hip).
#include "../../include/gsr.h"
#include "gsr_internal.h"

#include <cstdarg>
#include <cstdio>
#include <atomic>
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
// The pair counts of a call past the first host read live on the device (one SlabInfo per depth slab); the call
// queues a copy of that table into pinned memory at its end, and the accessors below wait for it only when asked.
struct PinnedSlot {
    void* dev = nullptr;          // the same memory at its device-visible address (null: this runtime does not map it)
    uint32_t* host = nullptr;     // a few KB of pinned memory, pooled per calling thread and deliberately never
    hipEvent_t copied = nullptr;  // freed: freeing at thread exit can race HIP runtime teardown
    hipEvent_t finished = nullptr;
};
thread_local std::vector<PinnedSlot> g_pinned_free;
thread_local std::vector<PinnedSlot> g_pinned_draining;  // retired slots whose last copy / `finished` event may still be pending
thread_local PinnedSlot g_last_slot;           // holds the slab table of the call that finished last on this thread
thread_local uint32_t g_last_num_rendered = 0; // ... its reference pair count
thread_local int g_last_slabs = 0;             // ... and how many slabs it used (0: nothing to wait for)
constexpr size_t kCounterBytes = gsr::kCounterCopyBytes;          // what travels back to the host in the middle of a call
constexpr size_t kSlabTableAt = (kCounterBytes + 63) & ~size_t(63);  // byte offset of the slab table in a pinned slot
constexpr size_t kPinnedBytes = kSlabTableAt + sizeof(gsr::SlabInfo) * gsr::kMaxSlabs;

// Stage timing: a ring of event sets so a whole timed region can be averaged afterwards without
// synchronising between calls.  Events of one call: start, after projection, after the depth sort, after the host
// has the pair count; then per depth slab: after binning, tile sort, ranges, colours, blend.
constexpr int kTimingRing = 256;
constexpr int kHeadEvents = 4, kSlabEvents = 5;
constexpr int kEventsPerCall = kHeadEvents + kSlabEvents * gsr::kMaxSlabs;
int g_options[GSR_OPT_NUM] = {/*GSR_OPT_TILE_CULL*/ 1, /*GSR_OPT_SLABS*/ 2, /*GSR_OPT_SLAB_FIRST*/ 400,
                              /*GSR_OPT_DEFER_COLOUR*/ 1, /*GSR_OPT_SLAB_MIN_REST*/ 3000000,
                              /*GSR_OPT_RADIX_RANK (kept by gsr_radix.hip)*/ 2, /*GSR_OPT_RADIX_RANK_ACTIVE (read-only)*/ 0,
                              /*GSR_OPT_DEPTH_DROP*/ 1, /*GSR_OPT_BLEND_ORDER*/ 1, /*GSR_OPT_RADIX_RANK_FALLBACKS (read-only)*/ 0,
                              /*GSR_OPT_BACKWARD_DETERMINISTIC*/ 0, /*GSR_OPT_GRAD_SLABS*/ 1};
std::atomic<bool> g_timing{false};
std::atomic<long> g_timing_epoch{0};       // bumped by gsr_set_stage_timing: every thread restarts its record at its next call
thread_local long g_epoch_seen = -1;
thread_local hipEvent_t g_ev[kTimingRing][kEventsPerCall];
thread_local int g_ev_slabs[kTimingRing];  // depth slabs of the call recorded in each slot; 0 = begun but never completed
                                           // (failed or cancelled after its first half): the readers skip such slots
thread_local bool g_ev_made = false;
thread_local long g_timed_calls = 0;   // completed timed calls since timing was (re)enabled
thread_local long g_begun_calls = 0;   // timed calls begun since then: several may be in flight (split calls), each owns a slot
thread_local int g_inflight = 0;       // timed calls begun and not yet finished / cancelled
thread_local bool g_stamp = false;     // whether the call being queued records events
thread_local int g_slot = 0;           // ring slot of the call being queued

// Backward timing: process-wide (autograd calls gsr_backward from its own thread), guarded by a mutex.
constexpr int kBackwardRing = 64;
std::mutex g_bw_mutex;
hipEvent_t g_bw_ev[kBackwardRing][3];
bool g_bw_made = false;
bool g_bw_done[kBackwardRing];   // the slot's three events have all been recorded
long g_bw_begun = 0;             // timed backward calls begun since timing was (re)enabled: each reserves its own slot
long g_bw_calls = 0;             // ... and completed

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


void stamp(int idx, hipStream_t s) {
    if (!g_stamp) return;
    (void)hipEventRecord(g_ev[g_slot][idx], s);
}

} // namespace

extern "C" {

const char* gsr_last_error(void) { return g_error; }
int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_target_arch(void) { return "gfx950"; }

static void trim_det_pools();
int gsr_set_option(int option, int value) {
    if (option < 0 || option >= GSR_OPT_NUM) return fail(GSR_ERR_INVALID_ARG, "unknown option %d", option);
    if (option == GSR_OPT_RADIX_RANK_ACTIVE) return fail(GSR_ERR_INVALID_ARG, "GSR_OPT_RADIX_RANK_ACTIVE is read-only");
    if (option == GSR_OPT_RADIX_RANK_FALLBACKS) return fail(GSR_ERR_INVALID_ARG, "GSR_OPT_RADIX_RANK_FALLBACKS is read-only");
    if (option == GSR_OPT_RADIX_RANK) {
        if (value < 0 || value > 3) return fail(GSR_ERR_INVALID_ARG, "GSR_OPT_RADIX_RANK must be 0, 1, 2 or 3");
        gsr::radix_set_rank_request(value);
    }
    const int before = g_options[option];
    g_options[option] = value;
    if (option == GSR_OPT_BACKWARD_DETERMINISTIC && before != 0 && value == 0) trim_det_pools();   // give the records' memory back
    return GSR_OK;
}
int gsr_get_option(int option) {
    if (option < 0 || option >= GSR_OPT_NUM) return fail(GSR_ERR_INVALID_ARG, "unknown option %d", option);
    if (option == GSR_OPT_RADIX_RANK) return gsr::radix_rank_request();
    if (option == GSR_OPT_RADIX_RANK_ACTIVE) return gsr::radix_rank_mode(nullptr, nullptr);  // (self-test on the null stream if due)
    if (option == GSR_OPT_RADIX_RANK_FALLBACKS) {
        unsigned long long tiles = 0ull;
        const hipError_t e = gsr::radix_rank_fallbacks(&tiles);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)fail(GSR_ERR_HIP, "GSR_OPT_RADIX_RANK_FALLBACKS: the device word could not be read: %s", hipGetErrorString(e));
            return -1;
        }
        return tiles > 0x7fffffffull ? 0x7fffffff : (int)tiles;
    }
    return g_options[option];
}

void gsr_set_stage_timing(int enable) {
    {
        std::lock_guard<std::mutex> lock(g_bw_mutex);
        g_bw_calls = 0;
        g_bw_begun = 0;
        for (bool& d : g_bw_done) d = false;
    }
    g_timing = enable != 0;
    g_timing_epoch.fetch_add(1);   // other threads drop their records when they next begin a call
    g_epoch_seen = g_timing_epoch.load();
    g_timed_calls = 0;
    g_begun_calls = 0;
    g_inflight = 0;
}

namespace {
// Ring slots of this thread's completed timed calls, newest first (at most `want`); slots of calls that were begun but
// never finished are skipped.
int completed_slots(int want, int* slots) {
    const long span = g_begun_calls < kTimingRing ? g_begun_calls : kTimingRing;
    int n = 0;
    for (long c = 0; c < span && n < want; ++c) {
        const int slot = (int)((g_begun_calls - 1 - c) % kTimingRing);
        if (g_ev_slabs[slot] > 0) slots[n++] = slot;
    }
    return n;
}
} // namespace

int gsr_get_stage_times(float ms[GSR_STAGE_NUM]) {
    for (int i = 0; i < GSR_STAGE_NUM; ++i) ms[i] = 0.f;
    if (g_timed_calls <= 0) return fail(GSR_ERR_INVALID_ARG, "no timed gsr_forward call on this thread");
    if (g_inflight != 0) return fail(GSR_ERR_INVALID_ARG, "a split call is still in flight on this thread");
    int slots[kTimingRing];
    const int ncalls = completed_slots((int)(g_timed_calls < kTimingRing ? g_timed_calls : kTimingRing), slots);
    if (ncalls <= 0) return fail(GSR_ERR_INVALID_ARG, "no completed timed gsr_forward call on this thread");
    double sum[GSR_STAGE_NUM] = {0};
    // per-slab intervals -> stage: binning (gather / scan / recount / expansion), tile sort, ranges, colours, blend
    static const int slab_stage[kSlabEvents] = {GSR_STAGE_DUPLICATE, GSR_STAGE_TILE_SORT, GSR_STAGE_RANGES, GSR_STAGE_COLOUR,
                                                GSR_STAGE_BLEND};
    for (int c = 0; c < ncalls; ++c) {
        const int slot = slots[c];
        const int last = kHeadEvents - 1 + kSlabEvents * g_ev_slabs[slot];
        GSR_HIP(hipEventSynchronize(g_ev[slot][last]));
        for (int i = 0; i + 1 < kHeadEvents; ++i) {  // preprocess, depth sort, scan (= the host's wait for the pair count)
            float t = 0.f;
            GSR_HIP(hipEventElapsedTime(&t, g_ev[slot][i], g_ev[slot][i + 1]));
            sum[i] += t;
        }
        for (int e = kHeadEvents; e <= last; ++e) {
            float t = 0.f;
            GSR_HIP(hipEventElapsedTime(&t, g_ev[slot][e - 1], g_ev[slot][e]));
            sum[slab_stage[(e - kHeadEvents) % kSlabEvents]] += t;
        }
    }
    for (int i = 0; i < GSR_STAGE_NUM; ++i) ms[i] = (float)(sum[i] / ncalls);
    return ncalls;
}

int gsr_get_call_times(float* ms, int capacity) {
    if (!ms || capacity < 0) return fail(GSR_ERR_INVALID_ARG, "bad arguments");
    if (g_timed_calls <= 0) return 0;
    if (g_inflight != 0) return fail(GSR_ERR_INVALID_ARG, "a split call is still in flight on this thread");
    int want = (int)(g_timed_calls < kTimingRing ? g_timed_calls : kTimingRing);
    if (want > capacity) want = capacity;
    int slots[kTimingRing];
    const int ncalls = completed_slots(want, slots);
    for (int c = 0; c < ncalls; ++c) {
        const int slot = slots[c];
        const int last = kHeadEvents - 1 + kSlabEvents * g_ev_slabs[slot];
        GSR_HIP(hipEventSynchronize(g_ev[slot][last]));
        GSR_HIP(hipEventElapsedTime(&ms[c], g_ev[slot][0], g_ev[slot][last]));
    }
    return ncalls;
}

int gsr_get_backward_times(float ms[2]) {
    ms[0] = ms[1] = 0.f;
    std::lock_guard<std::mutex> lock(g_bw_mutex);
    const long span = g_bw_begun < kBackwardRing ? g_bw_begun : kBackwardRing;
    if (g_bw_calls <= 0 || span <= 0) return 0;
    double sum[2] = {0, 0};
    int n = 0;
    for (long c = 0; c < span; ++c) {   // slots are reserved per call: one begun on another thread and not yet complete is skipped
        const int slot = (int)((g_bw_begun - 1 - c) % kBackwardRing);
        if (!g_bw_done[slot]) continue;
        GSR_HIP(hipEventSynchronize(g_bw_ev[slot][2]));
        for (int i = 0; i < 2; ++i) {
            float t = 0.f;
            GSR_HIP(hipEventElapsedTime(&t, g_bw_ev[slot][i], g_bw_ev[slot][i + 1]));
            sum[i] += t;
        }
        ++n;
    }
    if (n == 0) return 0;
    ms[0] = (float)(sum[0] / n);
    ms[1] = (float)(sum[1] / n);
    return n;
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
int gsr_last_slab_pairs(uint32_t pairs[GSR_MAX_SLABS]) {
    if (!g_have_offsets) return fail(GSR_ERR_INVALID_ARG, "no gsr_forward call on this thread");
    for (int i = 0; i < GSR_MAX_SLABS; ++i) pairs[i] = 0u;
    if (g_last_slabs == 0) return 0;
    GSR_HIP(hipEventSynchronize(g_last_slot.finished));  // the call's last copy has landed
    const gsr::SlabInfo* t = reinterpret_cast<const gsr::SlabInfo*>(reinterpret_cast<const char*>(g_last_slot.host) + kSlabTableAt);
    for (int i = 0; i < g_last_slabs; ++i) pairs[i] = t[i].pairs;
    return g_last_slabs;
}
int gsr_last_pair_counts(uint32_t counts[2]) {
    uint32_t pairs[GSR_MAX_SLABS];
    const int n = gsr_last_slab_pairs(pairs);
    if (n < 0) return n;
    counts[0] = g_last_num_rendered;
    counts[1] = 0u;
    for (int i = 0; i < n; ++i) counts[1] += pairs[i];
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

} // extern "C"

namespace {
// GSR_OPT_BACKWARD_DETERMINISTIC takes 160 bytes per live pair for the duration of a call (1.4 GB at C3).  From the device's
// default pool that memory goes back to the driver at every synchronisation (release threshold 0) and the next call maps it
// again: 30 ms per call at C3, against 1.2 ms for the whole backward.  A pool of the library's own, one per device, that keeps
// what it has been given (threshold = everything) makes the second call as cheap as a cached allocator's.
constexpr int kMaxPoolDevices = 64;
std::mutex g_det_pool_mutex;
hipMemPool_t g_det_pool[kMaxPoolDevices] = {};
bool g_det_pool_tried[kMaxPoolDevices] = {};

} // namespace
// Switching GSR_OPT_BACKWARD_DETERMINISTIC off releases what its pools hold (after one call at C3 the library kept
// 1.4 GB per device outside the host framework's allocator for the life of the process).  While the option is on the pools keep
// their memory between calls -- that is what they are for.
static void trim_det_pools() {
    std::lock_guard<std::mutex> lock(g_det_pool_mutex);
    for (int d = 0; d < kMaxPoolDevices; ++d)
        if (g_det_pool[d] != nullptr) (void)hipMemPoolTrimTo(g_det_pool[d], 0);
    (void)hipGetLastError();
}
namespace {

hipError_t det_alloc(void** ptr, size_t bytes, hipStream_t stream) {
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxPoolDevices) {
        std::lock_guard<std::mutex> lock(g_det_pool_mutex);
        if (!g_det_pool_tried[dev]) {
            g_det_pool_tried[dev] = true;
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t made = nullptr;
            if (hipMemPoolCreate(&made, &props) == hipSuccess) {
                uint64_t keep = UINT64_MAX;
                if (hipMemPoolSetAttribute(made, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) g_det_pool[dev] = made;
                else (void)hipMemPoolDestroy(made);
            }
            (void)hipGetLastError();
        }
        pool = g_det_pool[dev];
    }
    if (pool != nullptr) {
        const hipError_t e = hipMallocFromPoolAsync(ptr, bytes, pool, stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipMallocAsync(ptr, bytes, stream);   // (a runtime without pools of one's own: the device's default pool)
}

// gsr_backward and gsr_backward_raw: `raw` != nullptr -> the parameter pointers are the model's raw tensors (raw->xyz in
// means3D's place, and so on), dL_dsh_rest / dL_dpix_normal belong to that form.
int backward_impl(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                  const char* binning_buffer, const char* image_buffer, const float* accum_alphas, const float* dL_dpix,
                  const float* dL_dpix_depth, const float* dL_dpix_alpha, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscale, float* dL_drot, float* accum_scratch, int debug, void* stream_,
                  const gsr_raw_params* raw, const float* dL_dpix_normal, float* dL_dsh_rest) {
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (P == 0) return GSR_OK;  // rasterize_points.cu:169: gradients stay as the binding zero-filled them
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(GSR_ERR_INVALID_ARG, "null scratch buffer");
    if ((dL_dpix_depth != nullptr) != (dL_dpix_alpha != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "dL_dpix_depth and dL_dpix_alpha must be given together (both NULL = both all zeros)");
    if (!means3D || !background || !viewmatrix || !projmatrix || !cam_pos || !accum_alphas || !dL_dpix ||
        !dL_dmean2D || !dL_dopacity || !dL_dmean3D || !dL_dscale || !dL_drot || !accum_scratch)
        return fail(GSR_ERR_INVALID_ARG, "null required pointer");
    if (shs != nullptr && (M <= 0 || !dL_dsh)) return fail(GSR_ERR_INVALID_ARG, "shs given with M=%d or no dL_dsh", M);
    if ((scales != nullptr) != (rotations != nullptr) || (scales != nullptr) == (cov3D_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (raw != nullptr && (!shs || !scales || !raw->opacity_logits || (M > 1 && (!raw->features_rest || !dL_dsh_rest))))
        return fail(GSR_ERR_INVALID_ARG, "raw parameters: all six tensors and dL_dfeatures_rest (M > 1) are required");

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
    // The forward call may have been cut into depth slabs (GSR_FORWARD_INFERENCE: what the binding also makes in grad mode under
    // GSR_OPT_GRAD_SLABS): the per-pixel pass walks the slabs' list segments back to front.  Pairs a later slab dropped belong to
    // tiles whose 256 pixels had all stopped: no pixel's last contributor lies behind them, the backward never wanted them.
    const int S = (int)h[0].count[2];
    if (S < 1 || S > gsr::kMaxSlabs || h[2].count[3] != (uint32_t)S)
        return fail(GSR_ERR_INVALID_ARG, "scratch buffers describe %d depth slabs (image arena: %u)", S, h[2].count[3]);
    if (S > 1 && g_options[GSR_OPT_BACKWARD_DETERMINISTIC] != 0)
        return fail(GSR_ERR_INVALID_ARG, "GSR_OPT_BACKWARD_DETERMINISTIC needs the scratch of a forward call with ONE list per tile (this one "
                                         "has %d depth slabs): set the option before the forward call, or make a full call", S);
    if (h[0].count[0] != (uint32_t)P || h[0].count[1] != (uint32_t)R || h[2].count[0] != (uint32_t)width ||
        h[2].count[1] != (uint32_t)height)
        return fail(GSR_ERR_INVALID_ARG, "scratch buffers belong to a different call (P %u/%d, R %u/%d, %ux%u/%dx%d)",
                    h[0].count[0], P, h[0].count[1], R, h[2].count[0], h[2].count[1], width, height);
    if (dL_dpix_normal != nullptr && (raw == nullptr || h[0].off[1] == 0))
        return fail(GSR_ERR_INVALID_ARG, "dL_dpix_normal needs the scratch of a gsr_forward_raw call that composited the normal image");

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
    gsr::BlendSegments segs = {};
    for (int k = 0; k < S; ++k) {
        segs.point_list[k] = (const uint32_t*)(bases[1] + h[1].slab_off[k]);
        segs.ranges[k] = (const uint2*)(bases[2] + h[2].slab_off[k]);
    }
    const uint32_t* point_list = segs.point_list[0];   // (the deterministic mode's id sort: one segment)
    const uint32_t* n_contrib = (const uint32_t*)(bases[2] + h[2].off[1]);
    const float* colors = colors_precomp != nullptr ? colors_precomp : rgb;  // rasterizer_impl.cu:399

    int bw_slot = -1;
    if (g_timing) {
        std::lock_guard<std::mutex> lock(g_bw_mutex);
        if (!g_bw_made) {
            for (auto& set : g_bw_ev)
                for (auto& e : set) GSR_HIP(hipEventCreate(&e));
            g_bw_made = true;
        }
        bw_slot = (int)(g_bw_begun++ % kBackwardRing);   // reserved here, under the lock: concurrent calls never share events
        g_bw_done[bw_slot] = false;
        GSR_HIP(hipEventRecord(g_bw_ev[bw_slot][0], stream));
    }
    const float* normal_colours = dL_dpix_normal != nullptr ? (const float*)(bases[0] + h[0].off[1]) : nullptr;
    if (g_options[GSR_OPT_BACKWARD_DETERMINISTIC] != 0) {
        // Per-Gaussian sums in a fixed order (gsr_backward.hip: kDet): the per-pixel passes store one record per (list position,
        // quadrant), the point list is sorted by Gaussian id (stable: a Gaussian's positions ascend) and one lane per Gaussian
        // adds its records in that order.  Buffers come from the stream's pool for the duration of the call.
        const uint32_t bound = h[1].count[1];
        const uint32_t* n_device = &reinterpret_cast<const gsr::SlabInfo*>(bases[0] + h[0].off[5])->pairs;
        const int passes = dL_dpix_normal != nullptr ? 2 : 1;
        int id_bits = 1;
        while (id_bits < 32 && (1ull << id_bits) < (unsigned long long)P) ++id_bits;
        Carver dc;
        const size_t bit_words = ((size_t)bound + 7) / 8;
        const size_t off_first = dc.take<uint32_t>((size_t)P), off_end = dc.take<uint32_t>((size_t)P);
        const size_t off_bits = dc.take<uint32_t>(bit_words * passes);
        const size_t zero_bytes = (dc.off + 255) & ~size_t(255);   // everything up to here must be zero
        const size_t off_keys = dc.take<uint32_t>(bound), off_keys_alt = dc.take<uint32_t>(bound);
        const size_t off_pos = dc.take<uint32_t>(bound), off_pos_alt = dc.take<uint32_t>(bound);
        const size_t off_tmp = dc.take<uint32_t>(gsr::radix_scratch_words(bound));
        const size_t off_part = dc.take<float>((size_t)bound * 40 * passes);
        char* draw = nullptr;
        GSR_HIP(det_alloc((void**)&draw, dc.total() + 256, stream));
        char* dbase = align_base(draw);
        int rc = GSR_OK;
        do {
#define GSR_DET(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); \
            rc = fail(GSR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
            GSR_DET(hipMemsetAsync(dbase, 0, zero_bytes, stream));
            if (rc != GSR_OK) break;
            uint32_t* bits1 = (uint32_t*)(dbase + off_bits);
            float* part1 = (float*)(dbase + off_part);
            uint32_t* bits2 = passes == 2 ? bits1 + bit_words : nullptr;
            float* part2 = passes == 2 ? part1 + (size_t)bound * 40 : nullptr;
            GSR_DET(gsr::launch_render_backward(cam, segs, S, background, raster, colors, accum_alphas, n_contrib, dL_dpix,
                                                dL_dpix_depth, dL_dpix_alpha, accum_scratch, stream, 0, part1, bits1));
            if (rc == GSR_OK && passes == 2)
                GSR_DET(gsr::launch_render_backward(cam, segs, S, background, raster, normal_colours, accum_alphas, n_contrib,
                                                    dL_dpix_normal, nullptr, nullptr, accum_scratch, stream, 10, part2, bits2));
            if (rc != GSR_OK) break;
            uint32_t *ids_sorted = nullptr, *pos_sorted = nullptr;
            if (bound > 0u) {
                GSR_DET(hipMemcpyAsync(dbase + off_keys, point_list, (size_t)bound * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
                gsr::RadixSortExtras ex;
                ex.n_device = n_device;
                if (rc == GSR_OK)
                    GSR_DET(gsr::radix_sort_pairs((uint32_t*)(dbase + off_tmp), bound, id_bits, (uint32_t*)(dbase + off_keys),
                                                  (uint32_t*)(dbase + off_keys_alt), (uint32_t*)(dbase + off_pos), (uint32_t*)(dbase + off_pos_alt),
                                                  /*iota_payload=*/true, /*want_sorted_keys=*/true, &ids_sorted, &pos_sorted, stream, ex));
                if (rc == GSR_OK)
                    GSR_DET(gsr::launch_det_segments(n_device, ids_sorted, bound, (uint32_t*)(dbase + off_first), (uint32_t*)(dbase + off_end), stream));
            }
            if (rc == GSR_OK)
                GSR_DET(gsr::launch_det_reduce(P, (const uint32_t*)(dbase + off_first), (const uint32_t*)(dbase + off_end), pos_sorted, bits1, part1,
                                               bits2, part2, accum_scratch, stream));
#undef GSR_DET
        } while (0);
        (void)hipFreeAsync(draw, stream);
        if (rc != GSR_OK) return rc;
    } else {
    GSR_HIP(hipMemsetAsync(accum_scratch, 0, (size_t)P * 16 * sizeof(float), stream));
    GSR_HIP(gsr::launch_render_backward(cam, segs, S, background, raster, colors, accum_alphas, n_contrib,
                                        dL_dpix, dL_dpix_depth, dL_dpix_alpha, accum_scratch, stream));
    if (dL_dpix_normal != nullptr)   // the second feature set's pass: colour sums to slots 10 - 12, geometry sums add to 4 - 9
        GSR_HIP(gsr::launch_render_backward(cam, segs, S, background, raster, normal_colours,
                                            accum_alphas, n_contrib, dL_dpix_normal, nullptr, nullptr, accum_scratch, stream, 10));
    }
    GSR_STAGE_CHECK("render_backward");
    if (bw_slot >= 0) GSR_HIP(hipEventRecord(g_bw_ev[bw_slot][1], stream));
    gsr::BackwardInputs b;
    b.P = P; b.sh_degree = D; b.M = M; b.means3D = means3D; b.radii = radii; b.shs = shs; b.scales = scales;
    b.rotations = rotations; b.cov3D_precomp = cov3D_precomp; b.scale_modifier = scale_modifier;
    b.accum = accum_scratch; b.dL_dmean2D = dL_dmean2D; b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity;
    b.dL_dcolor = dL_dcolor; b.dL_ddepth = dL_ddepth;
    b.dL_dmean3D = dL_dmean3D; b.dL_dcov3D = dL_dcov3D; b.dL_dsh = dL_dsh; b.dL_dscale = dL_dscale; b.dL_drot = dL_drot;
    if (raw != nullptr) {
        b.raw = 1; b.shs_rest = raw->features_rest; b.opacity_logits = raw->opacity_logits; b.dL_dsh_rest = dL_dsh_rest;
        b.normal_grads = dL_dpix_normal != nullptr ? 1 : 0;
    }
    GSR_HIP(gsr::launch_preprocess_backward(b, cam, stream));
    GSR_STAGE_CHECK("preprocess_backward");
    if (bw_slot >= 0) {
        GSR_HIP(hipEventRecord(g_bw_ev[bw_slot][2], stream));
        std::lock_guard<std::mutex> lock(g_bw_mutex);
        g_bw_done[bw_slot] = true;
        ++g_bw_calls;
    }
    return GSR_OK;
}
} // namespace

extern "C" {

int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                 const char* binning_buffer, const char* image_buffer, const float* accum_alphas, const float* dL_dpix,
                 const float* dL_dpix_depth, const float* dL_dpix_alpha, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                 float* dL_dsh, float* dL_dscale, float* dL_drot, float* accum_scratch, int debug, void* stream_) {
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                         image_buffer, accum_alphas, dL_dpix, dL_dpix_depth, dL_dpix_alpha, dL_dmean2D, dL_dconic, dL_dopacity,
                         dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accum_scratch, debug, stream_,
                         nullptr, nullptr, nullptr);
}

int gsr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height, const gsr_raw_params* raw,
                     float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                     float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                     const char* image_buffer, const float* accum_alphas, const float* dL_dpix, const float* dL_dpix_depth,
                     const float* dL_dpix_alpha, const float* dL_dpix_normal, float* dL_dmean2D, float* dL_dxyz,
                     float* dL_dlog_scales, float* dL_drotations, float* dL_dopacity_logits, float* dL_dfeatures_dc,
                     float* dL_dfeatures_rest, float* accum_scratch, int debug, void* stream_) {
    if (!raw) return fail(GSR_ERR_INVALID_ARG, "null raw parameter block");
    if (M <= 0) return fail(GSR_ERR_INVALID_ARG, "raw parameters carry SH coefficients: M=%d", M);
    return backward_impl(P, D, M, R, background, width, height, raw->xyz, raw->features_dc, nullptr, raw->log_scales, scale_modifier,
                         raw->rotations, nullptr, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, geom_buffer,
                         binning_buffer, image_buffer, accum_alphas, dL_dpix, dL_dpix_depth, dL_dpix_alpha, dL_dmean2D,
                         /*dL_dconic=*/nullptr, dL_dopacity_logits, /*dL_dcolor=*/nullptr, /*dL_ddepth=*/nullptr, dL_dxyz,
                         /*dL_dcov3D=*/nullptr, dL_dfeatures_dc, dL_dlog_scales, dL_drotations, accum_scratch, debug, stream_, raw,
                         dL_dpix_normal, dL_dfeatures_rest);
}

int gsr_blend(const char* geom_buffer, const char* binning_buffer, const char* image_buffer, int width, int height,
              const float* features, const float* background, float* out_color, float* out_depth, float* out_alpha,
              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!geom_buffer || !binning_buffer || !image_buffer || !features || !background || !out_color || !out_depth || !out_alpha)
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    const char* bases[3] = {align_base(const_cast<char*>(geom_buffer)), align_base(const_cast<char*>(binning_buffer)),
                            align_base(const_cast<char*>(image_buffer))};
    gsr::ArenaHeader h[3];
    if (!recall_headers(bases, h)) {
        for (int i = 0; i < 3; ++i) GSR_HIP(hipMemcpyAsync(&h[i], bases[i], sizeof h[i], hipMemcpyDeviceToHost, stream));
        GSR_HIP(hipStreamSynchronize(stream));
    }
    for (int i = 0; i < 3; ++i)
        if (h[i].magic != gsr::kArenaMagic || h[i].kind != (uint32_t)i)
            return fail(GSR_ERR_INVALID_ARG, "scratch buffer %d was not produced by gsr_forward", i);
    if (h[2].count[0] != (uint32_t)width || h[2].count[1] != (uint32_t)height)
        return fail(GSR_ERR_INVALID_ARG, "scratch buffers belong to a %ux%u frame, not %dx%d", h[2].count[0], h[2].count[1], width, height);
    const int slabs = (int)h[0].count[2];
    if (slabs < 1 || slabs > gsr::kMaxSlabs) return fail(GSR_ERR_INVALID_ARG, "bad slab count %d in the geometry header", slabs);
    gsr::Camera cam = {};
    cam.width = width; cam.height = height;
    cam.grid_x = (width + gsr::kTile - 1) / gsr::kTile;
    cam.grid_y = (height + gsr::kTile - 1) / gsr::kTile;
    gsr::BlendSegments segs = {};
    for (int k = 0; k < slabs; ++k) {
        segs.ranges[k] = (const uint2*)(bases[2] + h[2].slab_off[k]);
        segs.point_list[k] = (const uint32_t*)(bases[1] + h[1].slab_off[k]);
    }
    // one launch over every segment of the tile's list; the per-pixel last-contributor positions are only rewritten
    // for a single-slab call (they are the same values: same lists, same geometry)
    uint32_t* n_contrib = slabs == 1 ? (uint32_t*)(bases[2] + h[2].off[1]) : nullptr;
    GSR_HIP(gsr::launch_blend(cam, segs, 0, slabs, /*fresh=*/true, /*final=*/true, (const gsr::SplatRaster*)(bases[0] + h[0].off[0]),
                              features, background, out_color, out_depth, out_alpha, n_contrib, nullptr, nullptr, 0, stream));
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

namespace {
int place_impl(int n, const uint32_t* subset, const float* xyz, const float* rotation_raw, const float* log_scale, const float* opacity,
               const float* shs, int M, const float* placement, float* out_means3D, float* out_scales, float* out_rotations,
               float* out_opacities, float* out_shs, float* out_min_axis, void* stream_, bool placement_required) {
    if (n < 0) return fail(GSR_ERR_INVALID_ARG, "bad size n=%d", n);
    if (n == 0) return GSR_OK;
    if (!xyz || !rotation_raw || !log_scale || !out_means3D || !out_scales || !out_rotations || (placement_required && !placement))
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if ((out_opacities != nullptr) != (opacity != nullptr) || (out_shs != nullptr) != (shs != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "opacity / shs and their outputs must be given together");
    if (shs != nullptr && M <= 0) return fail(GSR_ERR_INVALID_ARG, "shs given with M=%d", M);
    gsr::ObjectPlacement pl = {};
    static_assert(sizeof(gsr::ObjectPlacement) == 21 * sizeof(float), "the placement block is 21 floats");
    if (placement != nullptr) memcpy(&pl, placement, sizeof pl);
    GSR_HIP(gsr::launch_place_object(n, subset, placement != nullptr, xyz, rotation_raw, log_scale, opacity, shs, M, pl, out_means3D,
                                     out_scales, out_rotations, out_opacities, out_shs, out_min_axis, (hipStream_t)stream_));
    return GSR_OK;
}
} // namespace

int gsr_place_object(int n, const float* xyz, const float* rotation_raw, const float* log_scale, const float* opacity, const float* shs,
                     int M, const float* placement, float* out_means3D, float* out_scales, float* out_rotations, float* out_opacities,
                     float* out_shs, float* out_min_axis, void* stream_) {
    return place_impl(n, nullptr, xyz, rotation_raw, log_scale, opacity, shs, M, placement, out_means3D, out_scales, out_rotations,
                      out_opacities, out_shs, out_min_axis, stream_, true);
}

int gsr_place_object_subset(int m, const uint32_t* subset, const float* xyz, const float* rotation_raw, const float* log_scale,
                            const float* opacity, const float* shs, int M, const float* placement, float* out_means3D,
                            float* out_scales, float* out_rotations, float* out_opacities, float* out_shs, float* out_min_axis,
                            void* stream_) {
    if (m > 0 && !subset) return fail(GSR_ERR_INVALID_ARG, "null subset");
    return place_impl(m, subset, xyz, rotation_raw, log_scale, opacity, shs, M, placement, out_means3D, out_scales, out_rotations,
                      out_opacities, out_shs, out_min_axis, stream_, false);
}

int gsr_selftest_exp(uint32_t first_bits, uint32_t count, unsigned long long* device_mismatches, void* stream_) {
    if (!device_mismatches || count > 0x7FFFFFFFu) return fail(GSR_ERR_INVALID_ARG, "bad selftest arguments");
    if (count == 0) return GSR_OK;
    GSR_HIP(gsr::launch_exp_selftest(first_bits, count, device_mismatches, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_selftest_lds_atomic_order(uint32_t workgroups, uint32_t rounds, uint32_t seed, unsigned long long* device_mismatches, void* stream_) {
    if (!device_mismatches || workgroups == 0 || workgroups > (1u << 20)) return fail(GSR_ERR_INVALID_ARG, "bad selftest arguments");
    GSR_HIP(gsr::launch_lds_atomic_order_selftest(workgroups, rounds, seed, device_mismatches, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_resize_rgba8_bilinear(const uint8_t* src, int src_w, int src_h, uint8_t* dst, int dst_w, int dst_h, uint8_t* tmp, void* stream_) {
    if (src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d -> %dx%d", src_w, src_h, dst_w, dst_h);
    if (!src || !dst || (!tmp && src_w != dst_w && src_h != dst_h)) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_resize_rgba8_bilinear(src, src_w, src_h, dst, dst_w, dst_h, tmp, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_resize_f32_nearest(const float* src, int src_w, int src_h, float* dst, int dst_w, int dst_h, void* stream_) {
    if (src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d -> %dx%d", src_w, src_h, dst_w, dst_h);
    if (!src || !dst) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_resize_f32_nearest(src, src_w, src_h, dst, dst_w, dst_h, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_frame_files(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale, const uint8_t* turbo_lut,
                    int width, int height, uint8_t* png_rgba, uint8_t* png_depth_preview, uint8_t* png_normal, float* npy_plane, uint8_t* work,
                    void* stream_) {
    if (width <= 0 || height <= 0 || gsr::png_file_bytes(width, height, 4) == 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!color || !alpha || !depth || !normal || !turbo_lut || !png_rgba || !png_depth_preview || !png_normal || !npy_plane || !work)
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if (((reinterpret_cast<uintptr_t>(png_rgba) | reinterpret_cast<uintptr_t>(png_depth_preview) | reinterpret_cast<uintptr_t>(png_normal)) & 15u) != 0)
        return fail(GSR_ERR_INVALID_ARG, "gsr_frame_files: the PNG buffers must be 16-byte aligned");
    if (!(depth_scale > 0.0f)) return fail(GSR_ERR_INVALID_ARG, "gsr_frame_files: depth_scale must be positive");
    GSR_HIP(gsr::launch_frame_files(color, alpha, depth, normal, depth_scale, turbo_lut, width, height, png_rgba, png_depth_preview, png_normal,
                                    npy_plane, work, nullptr, nullptr, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_frame_files_deflate(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale, const uint8_t* turbo_lut,
                            int width, int height, uint8_t* png_rgba, uint8_t* png_depth_preview, uint8_t* png_normal, float* npy_plane, uint8_t* work,
                            uint8_t* png_scratch, uint64_t* png_lengths, void* stream_) {
    if (width <= 0 || height <= 0 || gsr::png_deflate_max_bytes(width, height, 4) == 0) return fail(GSR_ERR_INVALID_ARG, "bad image size %dx%d", width, height);
    if (!color || !alpha || !depth || !normal || !turbo_lut || !png_rgba || !png_depth_preview || !png_normal || !npy_plane || !work || !png_scratch || !png_lengths)
        return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if (((reinterpret_cast<uintptr_t>(png_rgba) | reinterpret_cast<uintptr_t>(png_depth_preview) | reinterpret_cast<uintptr_t>(png_normal) |
          reinterpret_cast<uintptr_t>(png_scratch)) & 15u) != 0 ||
        (reinterpret_cast<uintptr_t>(png_lengths) & 7u) != 0)
        return fail(GSR_ERR_INVALID_ARG, "gsr_frame_files_deflate: the PNG buffers must be 16-byte aligned, the lengths 8-byte aligned");
    if (!(depth_scale > 0.0f)) return fail(GSR_ERR_INVALID_ARG, "gsr_frame_files_deflate: depth_scale must be positive");
    GSR_HIP(gsr::launch_frame_files(color, alpha, depth, normal, depth_scale, turbo_lut, width, height, png_rgba, png_depth_preview, png_normal,
                                    npy_plane, work, png_scratch, reinterpret_cast<unsigned long long*>(png_lengths), (hipStream_t)stream_));
    return GSR_OK;
}

size_t gsr_png_deflate_max_size(int width, int height, int channels) { return gsr::png_deflate_max_bytes(width, height, channels); }
size_t gsr_png_deflate_room(int width, int height, int channels) { return gsr::png_deflate_room_bytes(width, height, channels); }
size_t gsr_png_deflate_scratch(int width, int height, int channels) { return gsr::png_deflate_scratch_bytes(width, height, channels); }

int gsr_png_encode_deflate(const uint8_t* pixels, int width, int height, int channels, int planar, uint8_t* out, uint8_t* scratch, uint64_t* out_len,
                           void* stream_) {
    if (gsr::png_deflate_max_bytes(width, height, channels) == 0)
        return fail(GSR_ERR_INVALID_ARG, "gsr_png_encode_deflate: %dx%d with %d channels cannot be encoded (3 or 4 channels, < 2 GB)", width, height, channels);
    if (!pixels || !out || !scratch || !out_len) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if (((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(scratch)) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out_len) & 7u) != 0)
        return fail(GSR_ERR_INVALID_ARG, "gsr_png_encode_deflate: out and scratch must be 16-byte aligned, out_len 8-byte aligned");
    GSR_HIP(gsr::launch_png_encode_deflate(pixels, width, height, channels, planar, out, scratch, reinterpret_cast<unsigned long long*>(out_len),
                                           (hipStream_t)stream_));
    return GSR_OK;
}

size_t gsr_png_unfilter_scratch(int width, int height) { return gsr::png_unfilter_scratch_bytes(width, height); }

static_assert(sizeof(GsrPngUnfilterJob) == sizeof(gsr::PngUnfilterJob), "GsrPngUnfilterJob is passed through as it is");

int gsr_png_unfilter_batch(int count, const GsrPngUnfilterJob* jobs, void* stream_) {
    if (count < 0 || (count > 0 && !jobs)) return fail(GSR_ERR_INVALID_ARG, "gsr_png_unfilter_batch: bad job list");
    for (int i = 0; i < count; ++i) {
        const GsrPngUnfilterJob& j = jobs[i];
        if (gsr::png_unfilter_scratch_bytes(j.width, j.height) == 0 || (j.channels != 3 && j.channels != 4))
            return fail(GSR_ERR_INVALID_ARG, "gsr_png_unfilter: job %d: %dx%d with %d channels is not supported (8-bit RGB / RGBA, at most 4096 pixels wide)", i,
                        j.width, j.height, j.channels);
        if (!j.scanlines || !j.out_rgba || !j.scratch) return fail(GSR_ERR_INVALID_ARG, "gsr_png_unfilter: job %d: null pointer", i);
        if (((reinterpret_cast<uintptr_t>(j.out_rgba) | reinterpret_cast<uintptr_t>(j.scratch)) & 15u) != 0)
            return fail(GSR_ERR_INVALID_ARG, "gsr_png_unfilter: job %d: out_rgba and scratch must be 16-byte aligned", i);
    }
    if (count == 0) return GSR_OK;
    GSR_HIP(gsr::launch_png_unfilter_batch(count, reinterpret_cast<const gsr::PngUnfilterJob*>(jobs), (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_png_unfilter(const uint8_t* scanlines, int width, int height, int channels, uint8_t* out_rgba, uint8_t* scratch, void* stream_) {
    const GsrPngUnfilterJob job = {scanlines, width, height, channels, out_rgba, scratch};
    return gsr_png_unfilter_batch(1, &job, stream_);
}

int gsr_exr_unpack_channel(const uint8_t* blocks, int height, int bytes_per_line, int lines_per_block, int channel_at, int channel_bytes, uint8_t* plane,
                           void* stream_) {
    if (height <= 0 || bytes_per_line <= 0 || (bytes_per_line & 1) || lines_per_block <= 0 || channel_at < 0 || channel_bytes <= 0 ||
        channel_at + channel_bytes > bytes_per_line || (long long)bytes_per_line * lines_per_block > (1ll << 30))
        return fail(GSR_ERR_INVALID_ARG, "gsr_exr_unpack_channel: bad layout (height %d, %d bytes per line, %d lines per block, channel at %d + %d)", height,
                    bytes_per_line, lines_per_block, channel_at, channel_bytes);
    if (!blocks || !plane) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(gsr::launch_exr_unpack_channel(blocks, height, bytes_per_line, lines_per_block, channel_at, channel_bytes, plane, (hipStream_t)stream_));
    return GSR_OK;
}

static_assert(sizeof(GsrPngFileInfo) == sizeof(gsr::PngFileLayout) && sizeof(GsrExrFileInfo) == sizeof(gsr::ExrFileLayout), "file infos are passed through");

int gsr_png_file_probe(const uint8_t* file, size_t file_bytes, GsrPngFileInfo* info) {
    if (!file || !info) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::png_file_probe(file, file_bytes, reinterpret_cast<gsr::PngFileLayout*>(info));
}

int gsr_png_file_inflate(const uint8_t* file, size_t file_bytes, uint8_t* scanlines, size_t scanline_bytes) {
    if (!file || !scanlines) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::png_file_inflate(file, file_bytes, scanlines, scanline_bytes);
}

int gsr_exr_file_probe(const uint8_t* file, size_t file_bytes, const char* channel, GsrExrFileInfo* info) {
    if (!file || !info) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::exr_file_probe(file, file_bytes, channel, reinterpret_cast<gsr::ExrFileLayout*>(info));
}

int gsr_exr_file_inflate(const uint8_t* file, size_t file_bytes, const char* channel, uint8_t* blocks, size_t blocks_bytes) {
    if (!file || !blocks) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::exr_file_inflate(file, file_bytes, channel, blocks, blocks_bytes);
}

static_assert(sizeof(GsrInflateJob) == sizeof(gsr::InflateJob), "GsrInflateJob is passed through as it is");

int gsr_exr_file_pack(const uint8_t* file, size_t file_bytes, const char* channel, uint8_t* packed, size_t packed_room, GsrInflateJob* jobs, size_t* packed_bytes) {
    if (!file || !packed || !jobs || !packed_bytes) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::exr_file_pack(file, file_bytes, channel, packed, packed_room, reinterpret_cast<gsr::InflateJob*>(jobs), packed_bytes);
}

int gsr_inflate_zlib_blocks(const uint8_t* streams, uint8_t* out, const GsrInflateJob* jobs, int count, int* status, int* any_error, void* stream_) {
    if (count < 0 || (count > 0 && (!streams || !out || !jobs || !status))) return fail(GSR_ERR_INVALID_ARG, "gsr_inflate_zlib_blocks: bad arguments");
    if ((reinterpret_cast<uintptr_t>(streams) & 3u) != 0) return fail(GSR_ERR_INVALID_ARG, "gsr_inflate_zlib_blocks: streams must be 4-byte aligned");
    GSR_HIP(gsr::launch_inflate_zlib_blocks(streams, out, reinterpret_cast<const gsr::InflateJob*>(jobs), count, status, any_error, (hipStream_t)stream_));
    return GSR_OK;
}

int gsr_selftest_inflate_host(const uint8_t* zlib_stream, size_t stream_bytes, uint8_t* out, size_t out_bytes) {
    if (!zlib_stream || (!out && out_bytes)) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    return gsr::inflate_zlib_host(zlib_stream, stream_bytes, out, out_bytes);
}

int gsr_upload(void* device_dst, const void* host_src, size_t bytes, void* stream_) {
    if (bytes == 0) return GSR_OK;
    if (!device_dst || !host_src) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    GSR_HIP(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream_));
    return GSR_OK;
}

size_t gsr_png_size(int width, int height, int channels) { return gsr::png_file_bytes(width, height, channels); }
size_t gsr_png_room(int width, int height, int channels) { return gsr::png_room_bytes(width, height, channels); }

int gsr_png_encode(const uint8_t* pixels, int width, int height, int channels, int planar, uint8_t* out, void* stream_) {
    const size_t n = gsr::png_file_bytes(width, height, channels);
    if (n == 0) return fail(GSR_ERR_INVALID_ARG, "gsr_png_encode: %dx%d with %d channels cannot be encoded (3 or 4 channels, < 2 GB)", width, height, channels);
    if (!pixels || !out) return fail(GSR_ERR_INVALID_ARG, "null pointer");
    if ((reinterpret_cast<uintptr_t>(out) & 15u) != 0) return fail(GSR_ERR_INVALID_ARG, "gsr_png_encode: out must be 16-byte aligned");
    GSR_HIP(gsr::launch_png_encode(pixels, width, height, channels, planar, out, (hipStream_t)stream_));
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
struct ForwardCall {
    hipStream_t stream = nullptr;
    int debug = 0, prefiltered = 0, P = 0, T = 0, width = 0, height = 0, slot = 0, tally_groups = 1;
    bool queued = false, device_work = false, timed = false, inference = false, defer_colour = false;
    gsr::Camera cam;
    gsr::GaussianInputs in;
    gsr::GeometryArrays ga;
    gsr_alloc_fn binning_alloc = nullptr;
    void* binning_user = nullptr;
    char *gbase = nullptr, *iraw = nullptr, *ibase = nullptr;
    size_t gshift = 0;
    size_t geom_off[GSR_GEOM_NUM_SLOTS] = {};
    size_t img_off[GSR_IMG_NUM_SLOTS] = {};
    size_t off_slabs = 0, off_quad = 0, off_rows = 0, off_listed = 0, off_radix_tmp = 0, off_order = 0;
    int row_words = 0;
    uint32_t *order = nullptr, *point_offsets = nullptr, *slab_offsets = nullptr, *tile_totals = nullptr, *slab_tile_totals = nullptr;
    uint32_t *slab_cpos = nullptr, *slab_coffs = nullptr;
    uint4* sorted_bins = nullptr;
    const float *colors_precomp = nullptr, *background = nullptr, *extra_features = nullptr;
    float *out_color = nullptr, *out_depth = nullptr, *out_alpha = nullptr, *out_extra = nullptr;
    // gsr_forward_raw (set before forward_begin): the inputs are a model's raw parameter tensors; raw_rest = its
    // _features_rest; raw_normals = the per-Gaussian view normals are worked out by the projection kernel into an array of
    // the geometry arena and composited as the call's second feature set (into out_extra)
    bool raw = false, raw_normals = false;
    const float* raw_rest = nullptr;
    PinnedSlot pinned;

    ~ForwardCall() {
        if (timed && g_inflight > 0) --g_inflight;  // (a call that never finished leaves g_ev_slabs[slot] == 0: readers skip it)
        if (!pinned.host) return;
        // Only reached with a slot when the call did not complete (cancelled, or failed part way).  Kernels already queued
        // on the stream may still store into the slot (the totals; the slab table written by bin_offsets / slab_compact):
        // it goes back to the free list only after an event behind everything queued so far has completed -- no host wait.
        if ((queued || device_work) && hipEventRecord(pinned.finished, stream) == hipSuccess) g_pinned_draining.push_back(pinned);
        else if (!(queued || device_work)) g_pinned_free.push_back(pinned);
        // (an event that cannot be recorded means the device is gone: the few KB of pinned memory are abandoned)
    }
};

// Depth slabs of an inference call.  Slab s takes the splats whose inclusive pair offset (over the depth order) lies in
// (cut[s-1], cut[s]]: the first slab holds ~`first` pairs per tile -- enough to finish most tiles of a scene with an
// opaque front -- and every further one three times as many as the one before.  bound[s] is what the launches and the
// arrays of slab s are sized for.
struct SlabPlan {
    int slabs = 1;
    uint32_t cut[gsr::kMaxSlabs] = {};    // slabs - 1 entries
    uint32_t bound[gsr::kMaxSlabs] = {};
};

SlabPlan plan_slabs(bool inference, uint32_t live_bound, int T, int done_words) {
    SlabPlan p;
    p.bound[0] = live_bound;
    const int max_slabs = g_options[GSR_OPT_SLABS] > 0 ? g_options[GSR_OPT_SLABS] : gsr::kMaxSlabs;
    const unsigned long long first = (unsigned long long)(g_options[GSR_OPT_SLAB_FIRST] > 0 ? g_options[GSR_OPT_SLAB_FIRST] : 400) *
                                     (unsigned long long)T;
    const unsigned long long min_rest = (unsigned long long)(g_options[GSR_OPT_SLAB_MIN_REST] > 0 ? g_options[GSR_OPT_SLAB_MIN_REST] : 0);
    if (!inference || max_slabs < 2 || done_words > 4096 || (unsigned long long)live_bound < 2ull * first ||
        (unsigned long long)live_bound - first < min_rest)
        return p;
    unsigned long long at = first, size = first;
    int n = 0;
    while (n < max_slabs - 1 && n < gsr::kMaxSlabs - 1 && at < (unsigned long long)live_bound) {
        p.cut[n++] = (uint32_t)at;
        size *= 3ull;
        at += size;
    }
    p.slabs = n + 1;
    for (int k = 0; k < p.slabs; ++k) {
        const unsigned long long lo = k == 0 ? 0ull : p.cut[k - 1];
        const unsigned long long hi = k == p.slabs - 1 ? (unsigned long long)live_bound : p.cut[k];
        unsigned long long b = hi - lo + (k == 0 ? 0ull : (unsigned long long)T);  // a splat at the cut brings at most T pairs along
        if (b > live_bound) b = live_bound;
        p.bound[k] = (uint32_t)b;
    }
    return p;
}

int forward_begin(ForwardCall& fc, gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc,
                  void* binning_user, gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                  const float* background, int width, int height, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                  float* out_depth, float* out_alpha, int* radii, int debug, void* stream_,
                  const float* extra_features, float* out_extra, unsigned flags) {
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
    if ((flags & ~(unsigned)GSR_FORWARD_INFERENCE) != 0u) return fail(GSR_ERR_INVALID_ARG, "unknown flags 0x%x", flags);
    if (fc.raw && (shs == nullptr || !have_sr || (M > 1 && fc.raw_rest == nullptr)))
        return fail(GSR_ERR_INVALID_ARG, "raw parameters: xyz, log_scales, rotations, opacity_logits, features_dc (and features_rest when M > 1) are all required");
    if (fc.raw_normals && (!fc.raw || out_extra == nullptr || extra_features != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "view normals need a raw-parameter call and an output image");

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
    fc.inference = (flags & GSR_FORWARD_INFERENCE) != 0u;
    // SH colours only for the splats that reach a list: needs the lists first, i.e. an inference call
    fc.defer_colour = fc.inference && shs != nullptr && g_options[GSR_OPT_DEFER_COLOUR] != 0;

    if (g_timing && !g_ev_made) {
        for (auto& set : g_ev)
            for (auto& e : set) GSR_HIP(hipEventCreate(&e));
        g_ev_made = true;
    }
    if (g_epoch_seen != g_timing_epoch.load()) {  // timing was switched on or off (by any thread) since this thread's last call
        g_epoch_seen = g_timing_epoch.load();
        g_timed_calls = 0; g_begun_calls = 0; g_inflight = 0;
    }
    g_slot = fc.slot = (int)(g_begun_calls % kTimingRing);  // one slot per call in flight (ring of 256: more than any driver keeps)
    g_stamp = fc.timed = g_timing;
    if (fc.timed) { ++g_begun_calls; ++g_inflight; g_ev_slabs[fc.slot] = 0; }
    // A retired slot goes back to the free list only once its `finished` event has completed: its slab-table copy may still
    // be in flight on ANOTHER stream (several streams fed by one thread), and a new call on this stream must neither
    // share that host buffer nor re-record that event.
    for (size_t i = 0; i < g_pinned_draining.size();) {
        if (hipEventQuery(g_pinned_draining[i].finished) == hipSuccess) {   // (an error is not "complete": the slot stays parked)
            g_pinned_free.push_back(g_pinned_draining[i]);
            g_pinned_draining[i] = g_pinned_draining.back();
            g_pinned_draining.pop_back();
        } else {
            ++i;
        }
    }
    if (!g_pinned_free.empty()) {
        fc.pinned = g_pinned_free.back();
        g_pinned_free.pop_back();
    } else {
        GSR_HIP(hipHostMalloc((void**)&fc.pinned.host, kPinnedBytes, hipHostMallocPortable | hipHostMallocMapped));
        if (hipHostGetDevicePointer(&fc.pinned.dev, fc.pinned.host, 0) != hipSuccess) { (void)hipGetLastError(); fc.pinned.dev = nullptr; }
        GSR_HIP(hipEventCreateWithFlags(&fc.pinned.copied, hipEventDisableTiming));
        GSR_HIP(hipEventCreateWithFlags(&fc.pinned.finished, hipEventDisableTiming));
    }

    // ---- geometry arena ----
    const size_t dup_blocks = (n + gsr::kDupTile - 1) / gsr::kDupTile;
    const size_t sort_tmp = gsr::radix_scratch_words((uint32_t)P) * sizeof(uint32_t);
    const size_t quad_words = ((size_t)4 * T + 31) / 32;
    fc.row_words = (cam.grid_x + 31) / 32;
    const size_t rows_words = (size_t)cam.grid_y * fc.row_words;
    size_t* geom_off = fc.geom_off;
    Carver gc;
    gc.take<gsr::ArenaHeader>(1);  // header at the arena's aligned base
    geom_off[GSR_GEOM_RASTER] = gc.take<gsr::SplatRaster>(n);
    geom_off[GSR_GEOM_RGB] = gc.take<float>(3 * n);
    geom_off[GSR_GEOM_SPLAT_BINS] = gc.take<gsr::SplatBin>(n);
    geom_off[GSR_GEOM_INTERNAL_RADII] = gc.take<int>(n);
    geom_off[GSR_GEOM_VIEW_NORMALS] = gc.take<float>(fc.raw_normals ? 3 * n : 0);
    const size_t off_keys_a = gc.take<uint32_t>(n), off_keys_b = gc.take<uint32_t>(n);
    const size_t off_ids_a = gc.take<uint32_t>(n), off_ids_b = gc.take<uint32_t>(n);
    geom_off[GSR_GEOM_POINT_OFFSETS] = gc.take<uint32_t>(n);
    const size_t off_slab_offsets = gc.take<uint32_t>(fc.inference ? n : 0);
    const size_t off_slab_cpos = gc.take<uint32_t>(fc.inference ? n : 0);
    const size_t off_slab_coffs = gc.take<uint32_t>(fc.inference ? n : 0);
    // the words of a call that must be zero before the kernels that use them, cleared by the tally kernel behind the
    // projection: frame counters, the table of depth slabs, one bit per finished 8x8 quadrant, one bit per finished tile
    const size_t off_flag = gc.take<gsr::FrameCounters>(1);
    fc.off_slabs = gc.take<gsr::SlabInfo>(gsr::kMaxSlabs);
    fc.off_quad = gc.take<uint32_t>(quad_words);
    fc.off_rows = gc.take<uint32_t>(rows_words);
    fc.off_order = gc.take<uint32_t>((size_t)gsr::kMaxSlabs * 8 * gsr::kOrderClasses);  // tiles per (slab, XCD band, list-length class)
    const size_t zero_end = gc.off;
    const size_t off_tallies = gc.take<gsr::BlockTally>((n + 255) / 256);
    const size_t off_pool_first = gc.take<uint32_t>((n + 255) / 256);
    fc.off_listed = gc.take<uint8_t>(fc.defer_colour ? n : 0);  // one byte per Gaussian: which slab listed it (cleared by the
                                                                // projection kernel among its other per-Gaussian stores)
    const size_t off_tile_totals = gc.take<uint32_t>(2 * dup_blocks);  // tile totals, then global offsets at tile ends
    const size_t off_slab_totals = gc.take<uint32_t>(fc.inference ? 2 * dup_blocks : 0);
    const size_t off_sorted_bins = gc.take<uint4>(n);  // splat records again, in depth order
    fc.off_radix_tmp = gc.take<char>(sort_tmp);
    char* graw = geom_alloc(gc.total(), geom_user);
    if (!graw) return fail(GSR_ERR_ALLOC, "geometry scratch callback returned NULL for %zu bytes", gc.total());
    char* gbase = fc.gbase = align_base(graw);
    fc.gshift = (size_t)(gbase - graw);

    // ---- image arena ----
    Carver ic;
    ic.take<gsr::ArenaHeader>(1);
    fc.img_off[GSR_IMG_RANGES] = ic.take<uint2>((size_t)T * (fc.inference ? gsr::kMaxSlabs : 1));  // one table per depth slab
    fc.img_off[GSR_IMG_N_CONTRIB] = ic.take<uint32_t>((size_t)width * height);
    fc.iraw = image_alloc(ic.total(), image_user);
    if (!fc.iraw) return fail(GSR_ERR_ALLOC, "image scratch callback returned NULL for %zu bytes", ic.total());
    fc.ibase = align_base(fc.iraw);
    for (auto& o : fc.img_off) o += (size_t)(fc.ibase - fc.iraw);

    gsr::GaussianInputs& in = fc.in;
    in.P = P; in.sh_degree = D; in.M = M;
    in.means3D = means3D; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    in.opacities = opacities; in.shs = shs; in.colors_precomp = colors_precomp;
    in.scale_modifier = scale_modifier; in.prefiltered = prefiltered;
    in.tile_cull = g_options[GSR_OPT_TILE_CULL] != 0;
    in.defer_colour = fc.defer_colour ? 1 : 0;
    in.raw = fc.raw ? 1 : 0;
    in.shs_rest = fc.raw_rest;
    in.view_normals = fc.raw_normals ? (float*)(gbase + geom_off[GSR_GEOM_VIEW_NORMALS]) : nullptr;
    if (fc.raw_normals) fc.extra_features = in.view_normals;
    else geom_off[GSR_GEOM_VIEW_NORMALS] = 0;

    gsr::GeometryArrays& ga = fc.ga;
    ga.raster = (gsr::SplatRaster*)(gbase + geom_off[GSR_GEOM_RASTER]);
    ga.rgb = (float*)(gbase + geom_off[GSR_GEOM_RGB]);
    ga.bins = (gsr::SplatBin*)(gbase + geom_off[GSR_GEOM_SPLAT_BINS]);
    ga.radii = radii ? radii : (int*)(gbase + geom_off[GSR_GEOM_INTERNAL_RADII]);
    ga.depth_keys = (uint32_t*)(gbase + off_keys_a);
    ga.ids = nullptr;  // the first radix pass generates 0..P-1 itself
    ga.listed = fc.defer_colour ? (uint8_t*)(gbase + fc.off_listed) : nullptr;
    ga.counters = (gsr::FrameCounters*)(gbase + off_flag);
    ga.tallies = (gsr::BlockTally*)(gbase + off_tallies);
    ga.pool_first = (uint32_t*)(gbase + off_pool_first);
    fc.point_offsets = (uint32_t*)(gbase + geom_off[GSR_GEOM_POINT_OFFSETS]);
    fc.slab_offsets = (uint32_t*)(gbase + off_slab_offsets);
    fc.slab_cpos = (uint32_t*)(gbase + off_slab_cpos);
    fc.slab_coffs = (uint32_t*)(gbase + off_slab_coffs);
    fc.tile_totals = (uint32_t*)(gbase + off_tile_totals);
    fc.slab_tile_totals = (uint32_t*)(gbase + off_slab_totals);
    fc.sorted_bins = (uint4*)(gbase + off_sorted_bins);

    stamp(0, stream);
    GSR_HIP(gsr::launch_preprocess(in, cam, ga, stream));
    GSR_STAGE_CHECK("preprocess");
    stamp(1, stream);

    // The one host round trip of the call (rasterizer_impl.cu:282 reads num_rendered back to size the binning arena).  Here
    // the totals are summed by one extra workgroup of the depth sort's first kernel, which stores them straight into the call's
    // pinned slot (device-visible at its own address) and clears the call's zero block on the way; the host waits on an
    // event while the GPU is already running the depth sort.
    void* const host_dev = fc.pinned.dev;
    memset(fc.pinned.host, 0, kCounterBytes);
    fc.tally_groups = host_dev != nullptr ? gsr::kTallyGroups : 1;
    const gsr::TallyDuty tally = {ga.tallies, (P + 255) / 256, ga.counters, (int)((zero_end - off_flag) / 4),
                                  reinterpret_cast<gsr::FrameCounters*>(host_dev), fc.tally_groups,
                                  gsr::tally_chunk_shift((P + 255) / 256, fc.tally_groups), ga.pool_first};
    // (a runtime that does not map pinned memory: the totals go into the zero block's first slots and are copied behind the sort)
    uint32_t* keys_sorted = nullptr;
    fc.queued = true;   // from the first launch below on a kernel may store into the pinned slot (what ~ForwardCall looks at)
    gsr::RadixSortExtras depth_extras;
    depth_extras.drop_key = g_options[GSR_OPT_DEPTH_DROP] != 0 ? &gsr::kCulledKey : nullptr;
    depth_extras.few_top_digits = true;   // the keys are positive floats: their top byte is sign + 7 exponent bits
    depth_extras.first_count_duty = &tally;
    depth_extras.after_first_count = host_dev != nullptr ? fc.pinned.copied : nullptr;
    GSR_HIP(gsr::radix_sort_pairs((uint32_t*)(gbase + fc.off_radix_tmp), (uint32_t)P, 32, ga.depth_keys, (uint32_t*)(gbase + off_keys_b),
                                  (uint32_t*)(gbase + off_ids_a), (uint32_t*)(gbase + off_ids_b),
                                  /*iota_payload=*/true, /*want_sorted_keys=*/false, &keys_sorted, &fc.order, stream, depth_extras));
    if (host_dev == nullptr) {
        GSR_HIP(hipMemcpyAsync(fc.pinned.host, gbase + off_flag, kCounterBytes, hipMemcpyDeviceToHost, stream));
        GSR_HIP(hipEventRecord(fc.pinned.copied, stream));
    }
    GSR_STAGE_CHECK("depth_sort");
    stamp(2, stream);
    geom_off[GSR_GEOM_DEPTH_ORDER] = (size_t)((char*)fc.order - gbase);
    geom_off[GSR_GEOM_LISTED] = fc.defer_colour ? fc.off_listed : 0;
    return 0;
}

int forward_finish(ForwardCall& fc) {
    if (fc.P == 0) return 0;
    hipStream_t stream = fc.stream;
    const int debug = fc.debug, P = fc.P, T = fc.T;
    const gsr::Camera& cam = fc.cam;
    const gsr::GeometryArrays& ga = fc.ga;
    char* const gbase = fc.gbase;
    g_slot = fc.slot;
    g_stamp = fc.timed;

    GSR_HIP(hipEventSynchronize(fc.pinned.copied));
    fc.queued = false;
    fc.device_work = true;   // from here on kernels that store into the pinned slot may be in flight until `finished`
    const gsr::FrameCounters* hc = reinterpret_cast<const gsr::FrameCounters*>(fc.pinned.host);  // first kCounterBytes only
    uint32_t flag = 0u;
    unsigned long long rect_total = 0, live_total = 0, emitting = 0, pool_rows = 0;
    uint32_t pool_group_first[gsr::kTallyGroups];   // run pool: where the rows of each tally group's large splats start
    for (int i = 0; i < gsr::kTallyGroups; ++i) {   // (the slots behind the groups' are zero)
        rect_total += hc->pair_totals[i] >> 32;
        live_total += hc->pair_totals[i] & 0xFFFFFFFFull;
        emitting += hc->visible[i];
        pool_group_first[i] = (uint32_t)pool_rows;
        pool_rows += hc->big_rows[i] & 0x7FFFFFFFu;
        flag |= hc->big_rows[i] >> 31;
    }
    if (pool_rows > 0xFFFFFFFFull) return fail(GSR_ERR_INVALID_ARG, "%llu tile rows of large splats overflow 32 bits", pool_rows);
    if (debug && fc.prefiltered && (flag & 1u))
        return fail(GSR_ERR_PREFILTERED, "a Gaussian was culled although prefiltered is set (auxiliary.h:156-160)");
    if (rect_total > 0x7FFFFFFFull) return fail(GSR_ERR_INVALID_ARG, "num_rendered %llu overflows int", rect_total);
    if (live_total > 0xFFFFFFFFull) return fail(GSR_ERR_INVALID_ARG, "pair count %llu overflows 32 bits", live_total);
    const uint32_t live_bound = (uint32_t)live_total;    // upper bound of the pairs that will be expanded (exact without large splats)
    const uint32_t num_rendered = (uint32_t)rect_total;  // the reference's count
    if (!fc.in.tile_cull) pool_rows = 0;                 // (every row of a large splat is its full width)
    stamp(3, stream);

    // ---- depth slabs and the binning arena ----
    const SlabPlan plan = plan_slabs(fc.inference, live_bound, T, cam.grid_y * fc.row_words);
    const int S = plan.slabs;
    uint32_t max_bound = 1;
    for (int k = 0; k < S; ++k) max_bound = plan.bound[k] > max_bound ? plan.bound[k] : max_bound;
    const int tile_bits = tile_key_bits((uint32_t)T);
    const int passes = (tile_bits + 7) / 8;
    const size_t tsort_tmp = gsr::radix_scratch_words(max_bound, gsr::expand_blocks(max_bound)) * sizeof(uint32_t);
    Carver bc;
    bc.take<gsr::ArenaHeader>(1);
    // tile keys ping-pong: shared by the slabs (nobody reads a slab's keys once its ranges are known)
    const size_t off_tk_a = bc.take<uint32_t>(max_bound), off_tk_b = bc.take<uint32_t>(max_bound);
    // point lists: the buffer a slab's sorted list ends up in is the slab's own (the lists stay valid after the call:
    // backward pass, second blend over the same geometry), its ping-pong partner is shared
    size_t off_list[gsr::kMaxSlabs];
    for (int k = 0; k < S; ++k) off_list[k] = bc.take<uint32_t>(plan.bound[k] ? plan.bound[k] : 1);
    const size_t off_pl_shared = bc.take<uint32_t>(max_bound);
    const size_t off_pool = bc.take<uint32_t>((size_t)pool_rows);
    const size_t off_pool_incl = bc.take<uint32_t>((size_t)pool_rows);
    const size_t off_btmp = bc.take<char>(tsort_tmp);
    // blend order (gsr_internal.h: BlendOrder).  Where a launch has more single-wave workgroups than the GPU holds at once (256 CUs
    // x 32 waves: above ~1000x540) it decides which waves run last; below that it still spreads every XCD's share over the
    // image (C2 blend 0.137 -> 0.128 ms, C4 0.210 -> 0.202, heavy unchanged, same box); a few hundred tiles are not worth the table
    const bool ordered_blend = g_options[GSR_OPT_BLEND_ORDER] != 0 && T > 256;
    // every XCD takes four strips of consecutive tiles spread over the image (strip i -> XCD i % 8): one contiguous eighth each
    // gave the XCDs with the picture's middle up to 10 % more work than those with its edges, and the launch ends with the
    // slowest (same box, C3 blend: 1 / 2 / 4 / 8 strips 0.355 / 0.343 / 0.344 / 0.341 ms)
    const int order_strips = 4;
    const int order_strip = (T + 8 * order_strips - 1) / (8 * order_strips);
    const int order_cap = order_strips * order_strip;
    const size_t off_order_table = bc.take<uint32_t>(ordered_blend ? (size_t)S * 8 * gsr::kOrderClasses * order_cap : 0);
    char* braw = fc.binning_alloc(bc.total(), fc.binning_user);
    if (!braw) return fail(GSR_ERR_ALLOC, "binning scratch callback returned NULL for %zu bytes", bc.total());
    char* bbase = align_base(braw);

    gsr::BinningArrays ba;
    ba.P = P; ba.V = (int)emitting; ba.grid_x = cam.grid_x; ba.grid_y = cam.grid_y;
    ba.depth_order = fc.order; ba.bins = ga.bins; ba.raster = ga.raster; ba.sorted_bins = fc.sorted_bins;
    ba.offsets = fc.point_offsets; ba.tile_totals = fc.tile_totals;
    ba.slab_offsets = fc.slab_offsets; ba.slab_tile_totals = fc.slab_tile_totals;
    ba.slab_cpos = fc.slab_cpos; ba.slab_coffs = fc.slab_coffs; ba.tiles_p = (P + gsr::kDupTile - 1) / gsr::kDupTile;
    ba.run_pool = (uint32_t*)(bbase + off_pool); ba.pool_rows = (uint32_t)pool_rows;
    ba.run_incl = (uint32_t*)(bbase + off_pool_incl);
    ba.pool_first = ga.pool_first;
    ba.pool_group_shift = 8 + gsr::tally_chunk_shift((P + 255) / 256, fc.tally_groups);
    for (int i = 0; i < gsr::kTallyGroups; ++i) ba.pool_group_first[i] = pool_group_first[i];
    ba.counters = ga.counters;
    ba.slabs = (gsr::SlabInfo*)(gbase + fc.off_slabs);
    // the slabs' pair counts are stored into the pinned slot by the kernels that settle them (pinned memory is device-visible
    // at its own address; if this runtime says otherwise a copy follows the call as in rounds 1 - 2)
    gsr::SlabInfo* const host_table = reinterpret_cast<gsr::SlabInfo*>(reinterpret_cast<char*>(fc.pinned.host) + kSlabTableAt);
    ba.slabs_host = fc.pinned.dev ? reinterpret_cast<gsr::SlabInfo*>(reinterpret_cast<char*>(fc.pinned.dev) + kSlabTableAt) : nullptr;
    for (int k = 0; k < gsr::kMaxSlabs; ++k) host_table[k] = gsr::SlabInfo{0u, 0u, 0u, 0u};   // (a slab no kernel reaches has no pairs)
    ba.quad_done = (uint32_t*)(gbase + fc.off_quad);
    ba.done_rows = (uint32_t*)(gbase + fc.off_rows);
    ba.row_words = fc.row_words;
    ba.tile_cull = fc.in.tile_cull;
    ba.listed = (fc.defer_colour && S > 1) ? (uint8_t*)(gbase + fc.off_listed) : nullptr;

    uint32_t* n_contrib = (uint32_t*)(fc.iraw + fc.img_off[GSR_IMG_N_CONTRIB]);
    gsr::BlendSegments segs = {};
    uint32_t* list_of[gsr::kMaxSlabs] = {};
    uint32_t* tile_keys_sorted = nullptr;

    // self-describing arenas (gsr_backward, gsr_blend): stamped by the last ranges launch
    gsr::ArenaHeader hg = {}, hb = {}, hi = {};
    hg.magic = hb.magic = hi.magic = gsr::kArenaMagic;
    hg.kind = 0; hb.kind = 1; hi.kind = 2;
    hg.count[0] = (uint32_t)P; hg.count[1] = num_rendered; hg.count[2] = (uint32_t)S; hg.count[3] = fc.inference ? 1u : 0u;
    hg.off[0] = (uint64_t)((char*)ga.raster - gbase);
    hg.off[1] = (uint64_t)(fc.raw_normals ? fc.geom_off[GSR_GEOM_VIEW_NORMALS] : 0);   // 0 = the call composited no view normals
    hg.off[3] = (uint64_t)((char*)ga.rgb - gbase);
    hg.off[4] = (uint64_t)fc.geom_off[GSR_GEOM_INTERNAL_RADII];
    hg.off[5] = (uint64_t)fc.off_slabs;
    hg.off[6] = (uint64_t)fc.off_quad;
    hg.off[7] = (uint64_t)fc.off_rows;
    hb.count[0] = (uint32_t)S;
    hb.count[1] = plan.bound[0];   // room of slab 0's list (>= its pairs): sizes the deterministic backward's buffers
    hi.count[0] = (uint32_t)fc.width; hi.count[1] = (uint32_t)fc.height; hi.count[2] = (uint32_t)T; hi.count[3] = (uint32_t)S;
    hi.off[1] = (uint64_t)((char*)n_contrib - fc.ibase);

    GSR_HIP(gsr::launch_bin_scan(ba, cam, S, plan.cut, stream));
    GSR_STAGE_CHECK("bin_scan");
    const float* features = fc.colors_precomp != nullptr ? fc.colors_precomp : ga.rgb;
    for (int k = 0; k < S; ++k) {
        const gsr::SlabInfo* slab = ba.slabs + k;
        uint2* ranges = (uint2*)(fc.iraw + fc.img_off[GSR_IMG_RANGES]) + (size_t)k * T;
        // expansion writes the "primary" buffers; an even number of radix passes brings the result back into them
        uint32_t* own_list = (uint32_t*)(bbase + off_list[k]);
        uint32_t* shared_list = (uint32_t*)(bbase + off_pl_shared);
        uint32_t* list_in = (passes % 2 == 0) ? own_list : shared_list;
        uint32_t* list_alt = (passes % 2 == 0) ? shared_list : own_list;
        uint32_t *keys_in = (uint32_t*)(bbase + off_tk_a), *keys_alt = (uint32_t*)(bbase + off_tk_b);
        if (k > 0) GSR_HIP(gsr::launch_slab_recount(ba, k, stream));
        // the expansion's workgroups count the low digits of the tile keys they write: the tile sort starts at its first scan
        const uint32_t precount = gsr::expand_blocks(plan.bound[k]);
        GSR_HIP(gsr::launch_expand(ba, k, plan.bound[k], keys_in, list_in, (uint32_t*)(bbase + off_btmp),
                                   gsr::radix_count_stride(plan.bound[k], precount), gsr::radix_first_digit_mask(tile_bits), stream));
        GSR_STAGE_CHECK("expand");
        stamp(kHeadEvents + kSlabEvents * k + 0, stream);
        uint32_t *tk_sorted = keys_in, *pl_sorted = list_in;
        if (plan.bound[k] > 0) {
            gsr::RadixSortExtras tile_extras;
            tile_extras.n_device = &slab->pairs;
            tile_extras.precount_blocks = precount;
            GSR_HIP(gsr::radix_sort_pairs((uint32_t*)(bbase + off_btmp), plan.bound[k], tile_bits, keys_in, keys_alt, list_in, list_alt,
                                          /*iota_payload=*/false, /*want_sorted_keys=*/true, &tk_sorted, &pl_sorted, stream, tile_extras));
        } else
            pl_sorted = own_list;
        GSR_STAGE_CHECK("tile_sort");
        if (debug && plan.bound[k] > 1) {  // the sorts' contract, checked: (tile, depth bits, id) ascending through the list
            GSR_HIP(gsr::launch_list_order_check(slab, plan.bound[k], tk_sorted, pl_sorted, ga.raster, ga.counters, stream));
            uint32_t violations = 0;
            GSR_HIP(hipMemcpyAsync(&violations, &ga.counters->order_violations, sizeof violations, hipMemcpyDeviceToHost, stream));
            GSR_HIP(hipStreamSynchronize(stream));
            if (violations != 0u)
                return fail(GSR_ERR_INTERNAL, "depth slab %d: %u entries of the sorted list are out of (tile, depth, id) order "
                                              "(radix rank %s)", k, violations, gsr::radix_rank_mode(stream, nullptr) ? "LDS adds" : "ballots");
        }
        stamp(kHeadEvents + kSlabEvents * k + 1, stream);
        list_of[k] = pl_sorted;
        tile_keys_sorted = tk_sorted;
        segs.ranges[k] = ranges;
        segs.point_list[k] = pl_sorted;
        hb.slab_off[k] = (uint64_t)((char*)pl_sorted - bbase);
        hi.slab_off[k] = (uint64_t)((char*)ranges - fc.ibase);
        const bool last = k == S - 1;
        const gsr::ArenaHeader hs[3] = {hg, hb, hi};
        void* const dsts[3] = {gbase, bbase, fc.ibase};
        // tile ranges (all (0,0) when nothing is live: rasterizer_impl.cu:311); the call's last duty also stamps the headers.
        // A call that evaluates deferred colours hands the duty to the first workgroups of its colour kernel (one launch
        // less, and the search latency hides behind the colour reads); otherwise it is a launch of its own.
        gsr::RangesDuty duty = {};
        duty.slab = slab; duty.num_tiles = T; duty.keys = tk_sorted; duty.ranges = ranges;
        gsr::BlendOrder order = {nullptr, nullptr, 0, 0, 0};
        if (ordered_blend) {
            order.strip = order_strip;
            order.counts = (uint32_t*)(gbase + fc.off_order) + (size_t)k * 8 * gsr::kOrderClasses;
            order.table = (uint32_t*)(bbase + off_order_table) + (size_t)k * 8 * gsr::kOrderClasses * order_cap;
            order.cap = order_cap;
            // classes half the slab's mean list length wide: the longest class starts at 3.5 times the mean
            uint32_t mean = plan.bound[k] / (uint32_t)T;
            while (mean > 3u) { mean >>= 1; ++order.shift; }
        }
        duty.order = order;
        for (int i = 0; i < 3; ++i) { duty.headers[i] = hs[i]; duty.header_dst[i] = last ? dsts[i] : nullptr; }
        const bool colour_blocks_suffice = (S == 1 ? (P + 255) / 256 : (P + 1023) / 1024) >= gsr::ranges_duty_blocks(T);
        const bool fused_ranges = fc.defer_colour && colour_blocks_suffice && !debug;
        if (!fused_ranges) GSR_HIP(gsr::launch_tile_ranges(duty, stream));
        GSR_STAGE_CHECK("tile_ranges");
        stamp(kHeadEvents + kSlabEvents * k + 2, stream);
        if (fc.defer_colour && S == 1)
            GSR_HIP(gsr::launch_sh_colour_all(fc.in, cam, ga.radii, ga.rgb, fused_ranges ? &duty : nullptr, stream));
        else if (fc.defer_colour)
            GSR_HIP(gsr::launch_sh_colour_listed(fc.in, cam, ba.listed, k + 1, slab, ga.rgb, fused_ranges ? &duty : nullptr, stream));
        stamp(kHeadEvents + kSlabEvents * k + 3, stream);
        GSR_HIP(gsr::launch_blend(cam, segs, k, k + 1, /*fresh=*/k == 0, /*final=*/last, ga.raster, features, fc.background,
                                  fc.out_color, fc.out_depth, fc.out_alpha, n_contrib, ba.quad_done, ba.done_rows, fc.row_words, stream,
                                  fc.extra_features, fc.out_extra, &order));
        GSR_STAGE_CHECK("blend");
        stamp(kHeadEvents + kSlabEvents * k + 4, stream);
        if (last) {
            const char* const bases[3] = {gbase, bbase, fc.ibase};
            remember_headers(bases, hs);
        }
    }
    if (fc.timed) { g_ev_slabs[fc.slot] = S; ++g_timed_calls; }

    // the pair counts of the slabs are in pinned memory once the call's kernels are done; whoever asks for them waits then
    if (ba.slabs_host == nullptr)
        GSR_HIP(hipMemcpyAsync(reinterpret_cast<char*>(fc.pinned.host) + kSlabTableAt, ba.slabs, sizeof(gsr::SlabInfo) * S,
                               hipMemcpyDeviceToHost, stream));
    GSR_HIP(hipEventRecord(fc.pinned.finished, stream));
    if (g_last_slot.host) g_pinned_draining.push_back(g_last_slot);  // (its copy may still be pending on another stream)
    g_last_slot = fc.pinned;
    fc.pinned = PinnedSlot();
    g_last_slabs = S;
    g_last_num_rendered = num_rendered;

    // what the gsr_last_* accessors report: the call that finished last on this thread
    for (int i = 0; i < GSR_GEOM_NUM_SLOTS; ++i) g_geom_off[i] = fc.geom_off[i] + fc.gshift;
    if (!(fc.defer_colour && S > 1)) g_geom_off[GSR_GEOM_LISTED] = 0;   // (single-slab calls colour every emitting splat)
    memcpy(g_img_off, fc.img_off, sizeof g_img_off);
    g_bin_off[GSR_BIN_POINT_LIST] = (size_t)((char*)list_of[0] - braw);
    g_bin_off[GSR_BIN_TILE_KEYS] = (size_t)((char*)tile_keys_sorted - braw);
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
    const int rc = forward_begin(fc, GSR_FORWARD_ARGS, nullptr, nullptr, 0u);
    return rc < 0 ? rc : forward_finish(fc);
}

int gsr_forward_extra(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                      gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                      int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                      const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                      const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                      float* out_depth, float* out_alpha, int* radii, const float* extra_features, float* out_extra,
                      unsigned flags, int debug, void* stream) {
    if ((extra_features != nullptr) != (out_extra != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "extra_features and out_extra must be given together");
    ForwardCall fc;
    const int rc = forward_begin(fc, GSR_FORWARD_ARGS, extra_features, out_extra, flags);
    return rc < 0 ? rc : forward_finish(fc);
}

namespace {
// gsr_forward_raw / gsr_forward_raw_begin: a model's tensors take the places of the activated ones in the common path
int raw_begin(ForwardCall& fc, gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
              gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width, int height,
              const gsr_raw_params* raw, float scale_modifier, const float* viewmatrix, const float* projmatrix,
              const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth,
              float* out_alpha, int* radii, float* out_normal, unsigned flags, int debug, void* stream) {
    if (!raw) return fail(GSR_ERR_INVALID_ARG, "null raw parameter block");
    if (M <= 0) return fail(GSR_ERR_INVALID_ARG, "raw parameters carry SH coefficients: M=%d", M);
    fc.raw = true;
    fc.raw_rest = raw->features_rest;
    fc.raw_normals = out_normal != nullptr;
    return forward_begin(fc, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width,
                         height, raw->xyz, raw->features_dc, nullptr, raw->opacity_logits, raw->log_scales, scale_modifier,
                         raw->rotations, nullptr, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
                         out_depth, out_alpha, radii, debug, stream, nullptr, out_normal, flags);
}
} // namespace

int gsr_forward_raw(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                    gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                    int height, const gsr_raw_params* raw, float scale_modifier, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                    float* out_color, float* out_depth, float* out_alpha, int* radii, float* out_normal, unsigned flags,
                    int debug, void* stream) {
    ForwardCall fc;
    const int rc = raw_begin(fc, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background,
                             width, height, raw, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                             out_color, out_depth, out_alpha, radii, out_normal, flags, debug, stream);
    return rc < 0 ? rc : forward_finish(fc);
}

void* gsr_forward_raw_begin(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                            gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                            int width, int height, const gsr_raw_params* raw, float scale_modifier, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                            float* out_color, float* out_depth, float* out_alpha, int* radii, float* out_normal,
                            unsigned flags, int debug, void* stream) {
    ForwardCall* fc = new (std::nothrow) ForwardCall;
    if (!fc) {
        fail(GSR_ERR_ALLOC, "out of host memory");
        return nullptr;
    }
    if (raw_begin(*fc, geom_alloc, geom_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M, background, width,
                  height, raw, scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color,
                  out_depth, out_alpha, radii, out_normal, flags, debug, stream) < 0) {
        delete fc;
        return nullptr;
    }
    return fc;
}

void* gsr_forward_begin(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                        gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                        int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                        float* out_depth, float* out_alpha, int* radii, const float* extra_features, float* out_extra,
                        unsigned flags, int debug, void* stream) {
    if ((extra_features != nullptr) != (out_extra != nullptr)) {
        fail(GSR_ERR_INVALID_ARG, "extra_features and out_extra must be given together");
        return nullptr;
    }
    ForwardCall* fc = new (std::nothrow) ForwardCall;
    if (!fc) {
        fail(GSR_ERR_ALLOC, "out of host memory");
        return nullptr;
    }
    if (forward_begin(*fc, GSR_FORWARD_ARGS, extra_features, out_extra, flags) < 0) {
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

int gsr_forward_ready(void* call) {
    if (!call) return fail(GSR_ERR_INVALID_ARG, "null call handle");
    ForwardCall* fc = static_cast<ForwardCall*>(call);
    if (fc->P == 0 || !fc->queued) return 1;
    const hipError_t e = hipEventQuery(fc->pinned.copied);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return fail(GSR_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

void gsr_forward_cancel(void* call) { delete static_cast<ForwardCall*>(call); }

int gsr_plan_slabs(uint32_t live_pairs, int width, int height, uint32_t cuts[GSR_MAX_SLABS]) {
    if (width <= 0 || height <= 0 || !cuts) return fail(GSR_ERR_INVALID_ARG, "bad arguments");
    const int gx = (width + gsr::kTile - 1) / gsr::kTile, gy = (height + gsr::kTile - 1) / gsr::kTile;
    const SlabPlan p = plan_slabs(true, live_pairs, gx * gy, gy * ((gx + 31) / 32));
    for (int i = 0; i < GSR_MAX_SLABS; ++i) cuts[i] = i + 1 < p.slabs ? p.cut[i] : 0u;
    return p.slabs;
}

} // extern "C"
