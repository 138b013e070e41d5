// gsr_internal.h -- declarations shared by the translation units of libgsr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gsr {

constexpr int kTile = 16;                    // tile edge in pixels; part of the result
constexpr uint32_t kCulledKey = 0xFFFFFFFFu; // depth key of a Gaussian that produces no pairs
constexpr int kRectPartials = 256;           // partial sums of rectangle areas (power of two)
constexpr int kRadixTile = 4096;             // pairs per workgroup of a radix pass (gsr_radix.hip)
constexpr int kDupTile = 1024;               // sorted positions per workgroup of the record gather + scan

// Device words of one forward call that must be zero before its first kernel: ONE memset clears them all.
// Only the first kCounterCopyBytes travel back to the host.
struct FrameCounters {
    unsigned long long pair_totals[kRectPartials];  // (sum of rectangle areas) << 32 | live pairs
    uint32_t visible[kRectPartials];                // Gaussians that emit at least the chance of a pair (key != kCulledKey)
    uint32_t error_flag;                            // bit 0 = prefiltered violation
    uint32_t pad[3];
};
constexpr size_t kCounterCopyBytes = sizeof(FrameCounters);

struct Camera {
    const float* viewmatrix;  // 16 floats, transposed w2c
    const float* projmatrix;  // 16 floats, transposed full projection
    const float* cam_pos;     // 3 floats
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int width, height, grid_x, grid_y;
};

struct GaussianInputs {
    int P, sh_degree, M;
    const float* means3D;
    const float* scales;         // nullable
    const float* rotations;      // nullable
    const float* cov3D_precomp;  // nullable
    const float* opacities;
    const float* shs;            // nullable
    const float* colors_precomp; // nullable
    float scale_modifier;
    int prefiltered;
    int tile_cull;  // GSR_OPT_TILE_CULL
};

// Everything the pair expansion needs about one splat, in one 16-byte record so that walking the
// splats in depth order costs one gather per splat instead of three.
struct SplatBin {
    uint32_t xy0;    // first tile of the rectangle: x | y << 16
    uint32_t width;  // rectangle width in tiles
    uint32_t mask;   // bit i = i-th tile (row-major) is live; all ones = the whole rectangle
    uint32_t count;  // live tiles = pairs this splat emits (0 = culled)
};

// Everything the blend needs about one splat except its colour, in one 32-byte record (never straddles
// a 64-byte line): a list entry costs one gather here plus one for the colour, not four.
struct SplatRaster {
    float x, y;            // pixel centre
    float cxx, cxy, cyy;   // conic (inverse 2D covariance)
    float opacity;
    float depth;           // view-space z
    float skip_below;      // -ln(255 * opacity) - 1e-4: a power below this cannot reach alpha >= 1/255 (blend pre-test)
};

struct GeometryArrays {
    SplatRaster* raster;
    float* rgb;
    SplatBin* bins;
    FrameCounters* counters;
    int* radii;           // caller's radii or the internal array
    uint32_t* depth_keys; // sort keys (float bits of depth, kCulledKey if culled)
    uint32_t* ids;        // 0..P-1, the sort payload; nullptr when the sort generates it itself
};

// ---- kernels (gsr_kernels.hip) ----
hipError_t launch_preprocess(const GaussianInputs& in, const Camera& cam, const GeometryArrays& out,
                             hipStream_t stream);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                               hipStream_t stream);
hipError_t launch_duplicate(int P, const Camera& cam, const uint32_t* depth_order, const uint32_t* point_offsets,
                            const SplatBin* bins, uint32_t* tile_keys, uint32_t* point_list, hipStream_t stream);
// Scan + expansion without any spinning, balanced by pairs (bin_gather / bin_offsets / expand kernels): writes
// point_offsets (global inclusive), tile_keys and point_list.  sorted_bins: V x 16 B, tile_totals: 2 * ceil(P / kDupTile) words.
hipError_t launch_scan_expand(int P, int V, uint32_t num_pairs, const Camera& cam, const uint32_t* depth_order,
                              const SplatBin* bins, uint4* sorted_bins, uint32_t* tile_totals, uint32_t* point_offsets,
                              uint32_t* tile_keys, uint32_t* point_list, hipStream_t stream);
// counts the floats with bit patterns first_bits .. first_bits + count - 1 on which the blend's exp differs from expf
hipError_t launch_exp_selftest(uint32_t first_bits, uint32_t count, unsigned long long* mismatches, hipStream_t stream);
struct ArenaHeader;
// ranges[t] = [first, last) positions of tile t in the sorted keys ((0,0) when empty; num_rendered = 0 clears them all);
// also stamps the three arena headers.
hipError_t launch_tile_ranges(uint32_t num_rendered, int num_tiles, const uint32_t* sorted_tile_keys, uint2* ranges,
                              void* const header_dst[3], const ArenaHeader headers[3], hipStream_t stream);
// variant 0: one wave per tile, 4 pixels per lane; variant 1: one wave per 8x8 quadrant
hipError_t launch_blend(const Camera& cam, int variant, int lds_pad_bytes, const uint2* ranges, const uint32_t* point_list,
                        const SplatRaster* raster, const float* features, const float* background, float* out_color,
                        float* out_depth, float* out_alpha, uint32_t* n_contrib, hipStream_t stream,
                        const float* extra_features = nullptr /*[P,3]: a second feature set ...*/,
                        float* out_extra = nullptr /*... composited into [3,H,W] in the same walk*/);

// First 256 bytes of each scratch arena: what the backward pass needs to find the forward's arrays
// again.  The reference re-derives its layout from sizes (rasterizer_impl.cu:381-383 fromChunk);
// ours also depends on which radix ping-pong buffer ended up holding the sorted list, so it is recorded.
constexpr uint32_t kArenaMagic = 0x47535231u;  // "GSR1"
struct ArenaHeader {
    uint32_t magic;
    uint32_t kind;      // 0 geometry, 1 binning, 2 image
    uint32_t count[4];  // geometry: P, num_rendered (reference), live pairs; binning: live pairs; image: W, H, T
    uint32_t pad[2];
    uint64_t off[8];    // byte offsets from the header's own address
};

struct BackwardInputs {
    int P, sh_degree, M;
    const float* means3D;
    const int* radii;
    const float* shs;            // nullable
    const float* scales;         // nullable
    const float* rotations;      // nullable
    const float* cov3D_precomp;  // nullable
    float scale_modifier;
    const float* accum;  // [P,16] per-Gaussian sums of the render backward (see launch_render_backward)
    float* dL_dmean2D;   // the five arrays below are written from accum
    float* dL_dconic;
    float* dL_dopacity;
    float* dL_dcolor;
    float* dL_ddepth;
    float* dL_dmean3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscale;
    float* dL_drot;
};
hipError_t launch_render_backward(const Camera& cam, const uint2* ranges, const uint32_t* point_list,
                                  const float* background, const SplatRaster* raster, const float* colors,
                                  const float* accum_alphas,
                                  const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_depth,
                                  const float* dL_dpix_alpha, float* accum /*[P,16] floats, zero on entry*/,
                                  hipStream_t stream);
hipError_t launch_preprocess_backward(const BackwardInputs& b, const Camera& cam, hipStream_t stream);
hipError_t launch_composite(int width, int height, const void* bg_c, const void* o_c, const float* o_d,
                            const void* s_c, const float* s_d, const void* o_s_c, const void* o_gs_c,
                            const float* o_gs_d, const void* s_f_c, const float* s_f_d, const void* s_f_c_pre,
                            void* out, hipStream_t stream);
// render()'s elementwise work around the two rasterizer passes (see gsr.h: gsr_view_normals, gsr_normal_maps)
hipError_t launch_view_normals(int P, const float* means3D, const float* axis, const float* cam_pos, float* out,
                               hipStream_t stream);
hipError_t launch_normal_maps(int width, int height, const float* normal_rgb, const float* depth, const float* c2w,
                              float fx, float fy, float cx, float cy, float* normal, float* pseudo, hipStream_t stream);
hipError_t launch_pack_rgba8(const float* color, const float* alpha, uint8_t* out, size_t n_pixels,
                             hipStream_t stream);

// ---- device-wide primitives (gsr_sort.hip) ----
// All three follow the two-call protocol: with temp == nullptr they only report temp_bytes.
hipError_t depth_sort_temp_bytes(int P, size_t* temp_bytes);
// Stable ascending sort of (depth_keys, ids) over all 32 key bits.  Buffers *_alt are the ping-pong
// partners; on return *keys_sorted / *ids_sorted point at whichever buffer holds the result.
hipError_t depth_sort(void* temp, size_t temp_bytes, int P, uint32_t* keys, uint32_t* keys_alt, uint32_t* ids,
                      uint32_t* ids_alt, uint32_t** keys_sorted, uint32_t** ids_sorted, hipStream_t stream);
hipError_t scan_temp_bytes(int P, size_t* temp_bytes);
// offsets[k] = sum_{j<=k} bins[order[j]].count
hipError_t scan_tiles_in_order(void* temp, size_t temp_bytes, int P, const SplatBin* bins, const uint32_t* order,
                               uint32_t* offsets, hipStream_t stream);
hipError_t tile_sort_temp_bytes(uint32_t n, size_t* temp_bytes);
// Stable ascending sort of (tile_keys, point_list) on the low `bits` key bits.
hipError_t tile_sort(void* temp, size_t temp_bytes, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                     uint32_t* vals, uint32_t* vals_alt, uint32_t** keys_sorted, uint32_t** vals_sorted,
                     hipStream_t stream);


// ---- hand-written radix sort (gsr_radix.hip) ----
// Stable ascending LSD sort on the low `bits` key bits, 8 per pass (count / scan / scatter kernels, no spinning,
// nothing to zero-fill).  scratch: radix_scratch_words(n) u32 words of any content.  iota_payload: the payload
// is 0..n-1 and `vals` is not read.  want_sorted_keys = false skips the key stores of the last pass.
size_t radix_scratch_words(uint32_t n);
hipError_t radix_sort_pairs(uint32_t* scratch, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                            uint32_t* vals, uint32_t* vals_alt, bool iota_payload, bool want_sorted_keys,
                            uint32_t** keys_sorted, uint32_t** vals_sorted, hipStream_t stream);

} // namespace gsr
