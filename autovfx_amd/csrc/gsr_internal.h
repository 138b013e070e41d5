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
constexpr int kExpandTile = 2048;            // most pairs one workgroup of the pair expansion writes (256 lanes x 8; gsr_binning.hip)
constexpr int kMaxSlabs = 8;                 // front-to-back depth slabs of one call (occlusion culling between slabs)
constexpr uint32_t kMaskTiles = 64;          // tight rectangles up to this many tiles carry a bit mask of live tiles

// Device words of one forward call that must be zero before the kernel that uses them: cleared, with the slab table and
// the bit rows that follow them in the arena, by the tally kernel that follows the projection (no memset launch).
// The totals travel to the host in the same layout (one slot of each array per tally group, the others zero; bit 31 of a
// big_rows slot = a prefiltered violation).
struct FrameCounters {
    unsigned long long pair_totals[kRectPartials];  // (sum of rectangle areas) << 32 | upper bound of the live pairs
    uint32_t visible[kRectPartials];                // Gaussians that emit at least the chance of a pair (key != kCulledKey)
    uint32_t big_rows[kRectPartials];               // tile rows of the splats too large for a mask (sizes the run pool)
    uint32_t error_flag;                            // (unused: a prefiltered violation travels as bit 31 of a big_rows slot)
    uint32_t pool_used;                             // (unused since round 5: the run pool's rows are placed by prefix sums)
    uint32_t order_violations;                      // debug calls: list entries out of (tile, depth bits, id) order
    uint32_t pad;
};
constexpr size_t kCounterCopyBytes = sizeof(FrameCounters);

// What one 256-lane workgroup of the projection kernel found.  Every workgroup stores its own entry and counter_tally_kernel
// sums them: no atomics (rounds 1 - 2: three per wave on 64 zero-filled words), nothing to zero-fill.
struct BlockTally {
    unsigned long long pair_total;   // (sum of rectangle areas) << 32 | upper bound of the live pairs
    uint32_t visible;                // Gaussians with key != kCulledKey
    uint32_t big_rows;               // tile rows of the splats too large for a mask; bit 31 = a prefiltered violation
};
static_assert(sizeof(BlockTally) == 16, "one 16-byte store per workgroup");

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
    int defer_colour;  // SH colours are evaluated later, only for the splats that reach a list (sh_colour_kernel)
    // gsr_forward_raw: the parameters arrive as the model stores them and are activated in the kernels, every operation
    // as (and in the order) PyTorch-ROCm evaluates the reference's getters (gaussian_model.py:95-128):
    //   scales = log scales (exp), rotations = unnormalised (F.normalize), opacities = logits (sigmoid),
    //   shs = _features_dc [P,1,3] and shs_rest = _features_rest [P,M-1,3] (the torch.cat is never materialised)
    int raw;
    const float* shs_rest;       // raw calls with M > 1
    float* view_normals;         // nullable, raw calls: [P,3] pc.get_normal(dir) * 0.5 + 0.5 of every splat that emits pairs
                                 // (gaussian_renderer/__init__.py:169-171), the blend's second feature set
};

// Everything the pair expansion needs about one splat, in one 16-byte record so that walking the splats in depth
// order costs one gather per splat.  The rectangle is the reference's (getRect) cut down to the bounding box of the
// region where the splat can reach alpha >= 1/255 at all ("tight"); num_rendered keeps counting the reference's.
//   w * h <= kMaskTiles : (lo, hi) is the 64-bit mask of live tiles, row-major over the tight rectangle
//   w * h >  kMaskTiles : lo = hi = ~0 here; the live tiles are one run of columns per tile row, worked out when the
//                         splats are gathered in depth order (bin_gather_kernel) and kept in the run pool
struct SplatBin {
    uint32_t xy0;   // first tile of the tight rectangle: x | y << 16
    uint32_t wh;    // its size in tiles: w | h << 16; 0 = emits nothing
    uint32_t lo, hi;
};
// The same splat once gathered into depth order (`sorted_bins`): x = xy0, y = wh, and
//   masked : z, w = the mask (updated in place when a later slab drops tiles that are already finished)
//   runs   : z = first row of the splat in the run pool, w = live tiles (sum of the run lengths)
// A pool entry is first_column | end_column << 16 (absolute tile columns, end exclusive; empty when equal).

// One front-to-back depth slab of a call: sorted positions [first, end), and how many pairs it put into its lists.
struct SlabInfo {
    uint32_t first, end;
    uint32_t pairs;        // live pairs expanded for this slab = length of its sorted list (device-side count)
    uint32_t emitters;     // positions of the slab that put at least one pair into it (slab 0: all of them)
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
    BlockTally* tallies;  // [div_up(P, 256)] one entry per workgroup of the projection kernel
    uint32_t* pool_first; // [div_up(P, 256)] run pool: rows of the large splats of the workgroups before this one, counted
                          // from the first workgroup of the same tally group (TallyDuty)
    int* radii;           // caller's radii or the internal array
    uint32_t* depth_keys; // sort keys (float bits of depth, kCulledKey if culled)
    uint32_t* ids;        // 0..P-1, the sort payload; nullptr when the sort generates it itself
    uint8_t* listed;      // nullable: one byte per Gaussian, cleared here for the pair expansion's marks (deferred colours)
};

// First 256 bytes of each scratch arena: what the backward pass needs to find the forward's arrays
// again.  The reference re-derives its layout from sizes (rasterizer_impl.cu:381-383 fromChunk);
// ours also depends on which radix ping-pong buffer ended up holding the sorted list, so it is recorded.
constexpr uint32_t kArenaMagic = 0x47535231u;  // "GSR1"
struct ArenaHeader {
    uint32_t magic;
    uint32_t kind;      // 0 geometry, 1 binning, 2 image
    uint32_t count[4];  // geometry: P, num_rendered (reference), slabs, inference flag; binning: slabs; image: W, H, T, slabs
    uint32_t pad[2];
    uint64_t off[8];    // byte offsets from the header's own address (geometry: 0 raster, 3 rgb, 4 radii, 5 slab table,
                        // 6 quadrant bits, 7 tile bit rows; image: 1 n_contrib)
    uint64_t slab_off[kMaxSlabs];  // binning: the sorted point list of slab s; image: its tile ranges
};
static_assert(sizeof(ArenaHeader) <= 256, "arena headers occupy the first 256 bytes");

// Which workgroup of a blend launch takes which tile (images of more than 256 tiles).  Workgroup b runs on XCD b % 8.  Every
// XCD's share of the image -- its band -- is four strips of consecutive tiles spread over the picture (strip i -> XCD i % 8:
// neighbouring tiles share splats and an L2, and all XCDs get a share of the picture's busy middle), and the XCD walks its
// band LONGEST LIST FIRST, in 7 classes of list length and an eighth for the tiles with nothing to blend: the waves still
// running when a launch with more workgroups than wave slots runs dry are short ones, and the workgroups that leave at
// once (a later slab's finished tiles) are dispatched behind everybody who has work.  The ranges duty files every tile under
// (band, class): counts per duty workgroup in LDS, one global atomic per (workgroup, cell) on the call's zeroed counters,
// the tile id into the cell's part of the table.  The blend finds its tile from the eight counters of its band
// (gsr_device.h: ordered_item).  Placement only: results do not depend on it, nor on the order the atomics are served in.
struct BlendOrder {
    uint32_t* counts;   // [8 bands][8 classes], zero before the ranges duty (null: plain band order)
    uint32_t* table;    // [8][8][cap] tile ids
    int cap;            // room per (band, class): the most tiles an XCD can get
    int shift;          // class = 6 - min(6, list length >> shift); class 7 = empty lists
    int strip;          // an XCD's band is every 8th strip of `strip` consecutive tiles: tile t -> XCD (t / strip) % 8
};
constexpr int kOrderClasses = 8;

// The per-tile [begin, end) of one slab's sorted list (K5, identifyTileRanges): 255 tiles per 256-lane workgroup, so
// div_up(num_tiles, 255) workgroups of whatever kernel carries the duty (gsr_device.h: tile_ranges_duty).
struct RangesDuty {
    const SlabInfo* slab;          // pairs = number of sorted keys
    int num_tiles;
    const uint32_t* keys;          // sorted tile ids
    uint2* ranges;
    BlendOrder order;
    ArenaHeader headers[3];        // stamped at header_dst[i] by workgroup 0 when non-null (the call's last ranges duty)
    void* header_dst[3];
};
inline int ranges_duty_blocks(int num_tiles) { return (num_tiles + 254) / 255; }

// ---- kernels (gsr_kernels.hip: per-Gaussian and per-pixel streaming kernels) ----
hipError_t launch_preprocess(const GaussianInputs& in, const Camera& cam, const GeometryArrays& out,
                             hipStream_t stream);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                               hipStream_t stream);
// Sums the projection kernel's per-workgroup tallies, stores the totals at `host_totals` (pinned host memory at its
// device-visible address; slot 0 of each array, the rest of the block must have been zeroed by the host) or, if that is
// null, into `zero_block` itself (a D2H copy then follows), and clears the `zero_words` words at `zero_block` (frame counters,
// slab table, quadrant and tile bits of the call).  Not a launch of its own: one extra workgroup of the depth sort's first
// count kernel does it (radix_sort_pairs' `first_count_duty`) -- nothing in the depth sort needs its result.
struct TallyDuty {
    const BlockTally* tallies;
    int blocks;
    FrameCounters* zero_block;
    int zero_words;
    FrameCounters* host_totals;
    int groups;   // workgroups that share the duty (<= kTallyGroups; 1 if host_totals is null): group j fills slot j of the totals
                  // from the tallies [j << chunk_shift, (j + 1) << chunk_shift) -- a contiguous share, so that ...
    int chunk_shift;
    uint32_t* pool_first;  // ... [blocks] it can leave the exclusive sum of big_rows inside its share: with the groups' totals
                           // (which the host has read back by then) that places every large splat's rows in the run pool
};
constexpr int kTallyGroups = 16;
// (a power of two, so that the kernel that looks a Gaussian's group up shifts instead of dividing)
inline int tally_chunk_shift(int blocks, int groups) {
    int s = 0;
    while (((long long)groups << s) < blocks) ++s;
    return s;
}
// SH colours of the Gaussians the pair expansion of one slab marked (`listed[gid] == tag`, tag = slab + 1;
// GaussianInputs::defer_colour), evaluated in Gaussian order; rgb[gid] is written.
// `duty` (nullable): the slab's tile ranges are computed by the first workgroups of the same launch.
hipError_t launch_sh_colour_listed(const GaussianInputs& in, const Camera& cam, const uint8_t* listed, int tag, const SlabInfo* slab,
                                   float* rgb, const RangesDuty* duty, hipStream_t stream);
// ... of every splat with a radius, in Gaussian order (a deferred-colour call that needs no depth slabs)
hipError_t launch_sh_colour_all(const GaussianInputs& in, const Camera& cam, const int* radii, float* rgb, const RangesDuty* duty,
                                hipStream_t stream);

// ---- binning (gsr_binning.hip): from the depth order to per-tile lists, slab by slab ----
// What the binning kernels share for one call.
struct BinningArrays {
    int P;                        // Gaussians of the call
    int V;                        // splats that may emit pairs (visible); positions [0, V) of the depth order
    int grid_x, grid_y;
    const uint32_t* depth_order;  // [P] Gaussian ids, ascending depth
    const SplatBin* bins;         // [P] in Gaussian order (preprocess)
    const SplatRaster* raster;    // [P] (centre, conic, opacity: the run computation of large splats)
    uint4* sorted_bins;           // [V] records in depth order
    uint32_t* offsets;            // [P] POINT_OFFSETS: inclusive live-pair count over the depth order
    uint32_t* tile_totals;        // [2 * ceil(P / kDupTile)] per-1024-position totals, then offsets at their ends
    uint32_t* slab_offsets;       // [P] inclusive pair count inside the slab a position belongs to (slabs > 0)
    uint32_t* slab_tile_totals;   // [2 * ceil(P / kDupTile)] per-1024-position pair totals of the slab being processed, then
                                  // how many of those positions still have a live pair
    uint32_t* slab_cpos;          // [P] slabs > 0: the positions (minus slab.first) that still have a live pair, ascending
    uint32_t* slab_coffs;         // [P] ... and the inclusive pair offset behind each of them
    int tiles_p;                  // ceil(P / kDupTile)
    uint8_t* listed;              // [P] nullable: expand_kernel writes slab + 1 for every Gaussian it puts into a list (zero on entry)
    uint32_t* run_pool;           // [pool_rows] column runs of the large splats
    uint32_t* run_incl;           // [pool_rows] live tiles of the splat's rows up to and including this one
    uint32_t pool_rows;
    // where a large splat's rows start: pool_group_first[gid >> pool_group_shift] + pool_first[gid >> 8] + the offset among
    // its own workgroup's large splats that the projection kernel left in the record (SplatBin::lo) -- no atomics
    const uint32_t* pool_first;
    int pool_group_shift;         // 8 + TallyDuty::chunk_shift
    uint32_t pool_group_first[kTallyGroups];
    FrameCounters* counters;
    SlabInfo* slabs;              // [kMaxSlabs] device
    SlabInfo* slabs_host;         // nullable: the same table in pinned host memory (device-visible address); the kernels that
                                  // settle a slab's pair count store it there too, so no copy has to follow the call
    uint32_t* quad_done;          // [ceil(4T / 32)] one bit per 8x8 quadrant whose 64 pixels have all stopped
    uint32_t* done_rows;          // [grid_y * row_words] the same per tile, one bit row per tile row
    int row_words;                // 32-bit words per bit row
    int tile_cull;
};
// gather + tile-local scan + global offsets over all V positions, then the slab boundaries: slab s takes the positions
// whose inclusive offset lies in (pair_cut[s-1], pair_cut[s]] (pair_cut[num_slabs - 1] = everything that is left).
hipError_t launch_bin_scan(const BinningArrays& a, const Camera& cam, int num_slabs, const uint32_t* pair_cuts /*host, num_slabs - 1*/,
                           hipStream_t stream);
// slabs > 0: drop the tiles finished by the slabs before (done_rows), re-scan inside the slab, set slabs[s].pairs
hipError_t launch_slab_recount(const BinningArrays& a, int slab, hipStream_t stream);
// (tile id, Gaussian id) pairs of one slab in depth order; at most pairs_bound of them (the grid is sized for it)
// first_counts (nullable): the workgroups also leave the digit histogram of their keys for the tile sort's first pass
// (radix_sort_pairs' precount_blocks = expand_blocks(pairs_bound)): first_counts[digit * count_stride + workgroup].
uint32_t expand_blocks(uint32_t pairs_bound);
hipError_t launch_expand(const BinningArrays& a, int slab, uint32_t pairs_bound, uint32_t* tile_keys, uint32_t* point_list,
                         uint32_t* first_counts, uint32_t count_stride, uint32_t digit_mask,
                         hipStream_t stream);
// debug calls: counts (into counters->order_violations) the entries of a slab's sorted list that are not in ascending
// (tile, depth bits, Gaussian id) order relative to their predecessor -- what two stable sorts must have produced
hipError_t launch_list_order_check(const SlabInfo* slab, uint32_t pairs_bound, const uint32_t* sorted_tile_keys,
                                   const uint32_t* point_list, const SplatRaster* raster, FrameCounters* counters,
                                   hipStream_t stream);
struct RangesDuty;
// ranges[t] = [first, last) positions of tile t in the sorted keys ((0,0) when empty); the number of keys is read from
// slab->pairs.  The three arena headers may ride along (RangesDuty::header_dst).  A stand-alone launch; calls that
// evaluate deferred colours hand the same duty to the first workgroups of their colour kernel instead (one launch less).
hipError_t launch_tile_ranges(const RangesDuty& duty, hipStream_t stream);

// ---- blend (gsr_blend.hip) ----
// One list segment per slab; a launch walks segments [seg_begin, seg_end).  `fresh`: pixels start from T = 1 (else
// the state the previous launch left in the output images is resumed); `final`: every pixel's result is written
// (else quadrants with live pixels park their state and finished quadrants are marked in quad_done / done_rows).
struct BlendSegments {
    const uint2* ranges[kMaxSlabs];
    const uint32_t* point_list[kMaxSlabs];
};
hipError_t launch_blend(const Camera& cam, const BlendSegments& segs, int seg_begin, int seg_end, bool fresh, bool final,
                        const SplatRaster* raster, const float* features, const float* background, float* out_color,
                        float* out_depth, float* out_alpha, uint32_t* n_contrib, uint32_t* quad_done, uint32_t* done_rows,
                        int row_words, hipStream_t stream, const float* extra_features = nullptr, float* out_extra = nullptr,
                        const BlendOrder* order = nullptr);
// counts the floats with bit patterns first_bits .. first_bits + count - 1 on which the blend's exp differs from expf
hipError_t launch_exp_selftest(uint32_t first_bits, uint32_t count, unsigned long long* mismatches, hipStream_t stream);

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
    // gsr_backward_raw (gsr_backward.hip: BackwardArgs): raw inputs, gradients with respect to the raw parameters
    int raw = 0;
    const float* shs_rest = nullptr;
    const float* opacity_logits = nullptr;
    float* dL_dsh_rest = nullptr;
    int normal_grads = 0;
};
// segs / num_segs: the per-tile lists of the forward call, one segment per depth slab, front to back (a full call: one)
hipError_t launch_render_backward(const Camera& cam, const BlendSegments& segs, int num_segs,
                                  const float* background, const SplatRaster* raster, const float* colors,
                                  const float* accum_alphas,
                                  const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_depth,
                                  const float* dL_dpix_alpha, float* accum /*[P,16] floats, zero on entry*/,
                                  hipStream_t stream, int colour_slot = 0 /*10: the pass over a call's second feature set*/,
                                  float* det_partial = nullptr, uint32_t* det_bits = nullptr /*GSR_OPT_BACKWARD_DETERMINISTIC*/);
// GSR_OPT_BACKWARD_DETERMINISTIC (gsr_backward.hip): sorted_ids = the point list sorted by Gaussian id (n = *n_device entries);
// seg_first / seg_end [P] (zero on entry) get every Gaussian's range in it.
hipError_t launch_det_segments(const uint32_t* n_device, const uint32_t* sorted_ids, uint32_t bound, uint32_t* seg_first, uint32_t* seg_end,
                               hipStream_t stream);
// ... and accum[P,16] is written from the (position, quadrant) records of the one or two per-pixel passes, in fixed order.
hipError_t launch_det_reduce(int P, const uint32_t* seg_first, const uint32_t* seg_end, const uint32_t* sorted_pos, const uint32_t* bits1,
                             const float* partial1, const uint32_t* bits2 /*nullable*/, const float* partial2 /*nullable*/, float* accum,
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
// gsr_place_object (include/gsr.h): one object of a dynamic scene, raw parameters -> activated, at its place in the scene buffers
struct ObjectPlacement { float c[3], R[9], s, c0[3], qR[4], log_s; };
hipError_t launch_place_object(int n, const uint32_t* subset /*nullable: output j <- input subset[j]*/, bool transform,
                               const float* xyz, const float* rot, const float* log_scale, const float* opacity, const float* shs, int M,
                               const ObjectPlacement& pl, float* out_xyz, float* out_scales, float* out_rot, float* out_opacity,
                               float* out_shs, float* out_min_axis, hipStream_t stream);
hipError_t launch_pack_rgba8(const float* color, const float* alpha, uint8_t* out, size_t n_pixels,
                             hipStream_t stream);

// ---- frame outputs / compositor inputs (gsr_frameio.hip) ----
// Bytes of the PNG file launch_png_encode writes for a W x H image with C (3 or 4) 8-bit channels; 0 if the size is not encodable.
size_t png_file_bytes(int W, int H, int C);
size_t png_room_bytes(int W, int H, int C);   // what `out` must hold: the file, then the kernels' partial checksums
// pixels: u8, interleaved [H,W,C] or (planar != 0) [C,H,W]; out: png_room_bytes bytes, 16-byte aligned.
hipError_t launch_png_encode(const uint8_t* pixels, int W, int H, int C, int planar, uint8_t* out, hipStream_t stream);
// The same file with a deflate-compressed IDAT (Paeth filter, run-length matches, one Huffman code per image built on the GPU).
size_t png_deflate_max_bytes(int W, int H, int C);    // upper bound of the file's length
size_t png_deflate_room_bytes(int W, int H, int C);   // what `out` must hold (the file's bound + what the CRC kernel may read behind it)
size_t png_deflate_scratch_bytes(int W, int H, int C);   // device scratch per image: filtered stream, per-block histograms / offsets, checksums
hipError_t launch_png_encode_deflate(const uint8_t* pixels, int W, int H, int C, int planar, uint8_t* out, uint8_t* scratch, unsigned long long* out_len,
                                     hipStream_t stream);

// The reference's four per-frame files from a render() result, queued by one host call (gsr.h: gsr_frame_files).  work: 10 * W * H bytes.
hipError_t launch_frame_files(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale,
                              const uint8_t* turbo_lut, int W, int H, uint8_t* png_rgba, uint8_t* png_depth, uint8_t* png_normal,
                              float* npy_plane, uint8_t* work, uint8_t* png_scratch, unsigned long long* png_lengths /*both null: stored PNGs*/,
                              hipStream_t stream);
// PIL's Image.resize(BILINEAR) on an RGBA8 image [H,W,4] and Image.resize(NEAREST) on an fp32 image, bit for bit
// (blend_all.py:21-28).  tmp: src_h * dst_w * 4 bytes (needed when both sizes change).
hipError_t launch_resize_rgba8_bilinear(const uint8_t* src, int src_w, int src_h, uint8_t* dst, int dst_w, int dst_h, uint8_t* tmp,
                                        hipStream_t stream);
hipError_t launch_resize_f32_nearest(const float* src, int src_w, int src_h, float* dst, int dst_w, int dst_h, hipStream_t stream);

// ---- the compositor's input files (gsr_layerio.hip) ----
// The inflated IDAT stream of an 8-bit RGB / RGBA, non-interlaced PNG (device memory) -> RGBA8 [H,W,4] (alpha 255 for RGB).
// scratch: png_unfilter_scratch_bytes(W, H) bytes, 16-byte aligned (0: the width is not supported).
size_t png_unfilter_scratch_bytes(int W, int H);
struct PngUnfilterJob {      // (= GsrPngUnfilterJob, gsr.h)
    const uint8_t* scanlines;
    int width, height, channels;
    uint8_t* out_rgba;
    uint8_t* scratch;
};
hipError_t launch_png_unfilter_batch(int n, const PngUnfilterJob* jobs, hipStream_t stream);   // one workgroup per image, eight images per launch
// The inflated, still predictor-coded scanline blocks of an OpenEXR file one after another (device memory) -> the bytes of the channel
// that occupies [c_at, c_at + c_bytes) of every line: plane[H][c_bytes].
hipError_t launch_exr_unpack_channel(const uint8_t* blocks, int H, int bytes_per_line, int lines_per_block, int c_at, int c_bytes, uint8_t* plane,
                                     hipStream_t stream);

// zlib streams inflated on the GPU, a single-wave workgroup each.  jobs / status / any_error: DEVICE memory; src_at multiples of 4,
// `streams` readable up to the next multiple of 4 behind every stream.  status[i]: 0 or an inflate::Status; *any_error is OR-ed with 1.
struct InflateJob {          // (= GsrInflateJob, gsr.h)
    uint32_t src_at, src_bytes, dst_at, dst_bytes;
};
hipError_t launch_inflate_zlib_blocks(const uint8_t* streams, uint8_t* out, const InflateJob* jobs, int count, int* status, int* any_error, hipStream_t stream);

// ---- the host side of the same files (gsr_layerfiles.hip: container parsing + zlib inflate, one call per file) ----
struct PngFileLayout {       // (= GsrPngFileInfo, gsr.h)
    int width, height, channels;
    size_t scanline_bytes;   // height * (1 + width * channels)
};
struct ExrFileLayout {       // (= GsrExrFileInfo, gsr.h)
    int width, height, bytes_per_line, lines_per_block, channel_at, channel_bytes, channel_is_half;
    int compression, n_blocks;   // the file's compression attribute (1 RLE, 2 ZIPS, 3 ZIP); scanline blocks in the part
    size_t blocks_bytes;     // height * bytes_per_line
    char channel[32];
};
// 0: a file the kernels take; 1: not covered (another flavour, or damaged) -- a host decoder's.
int png_file_probe(const uint8_t* file, size_t n, PngFileLayout* out);
int png_file_inflate(const uint8_t* file, size_t n, uint8_t* scanlines, size_t scanline_bytes);
int exr_file_probe(const uint8_t* file, size_t n, const char* want_channel, ExrFileLayout* out);
int exr_file_inflate(const uint8_t* file, size_t n, const char* want_channel, uint8_t* blocks, size_t blocks_bytes);
// ZIP / ZIPS files: the blocks' zlib streams copied one behind the other at 4-byte aligned offsets into `packed` (room: n + 4 * n_blocks
// + 4 bytes), jobs[n_blocks] filled for launch_inflate_zlib_blocks; *packed_bytes: what to upload.
int exr_file_pack(const uint8_t* file, size_t n, const char* want_channel, uint8_t* packed, size_t packed_room, InflateJob* jobs, size_t* packed_bytes);

// gsr_inflate_core.h's decoder run by one host lane (tests); 0 or an inflate::Status
int inflate_zlib_host(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);

// ---- hand-written radix sort (gsr_radix.hip) ----
// Stable ascending LSD sort on the low `bits` key bits, 8 per pass (count / scan / scatter kernels, no spinning,
// nothing to zero-fill).  scratch: radix_scratch_words(n) u32 words of any content.  iota_payload: the payload
// is 0..n-1 and `vals` is not read.  want_sorted_keys = false skips the key stores of the last pass.
// `extras`: RadixSortExtras below.
// counts the returning LDS adds whose result was not (value before the instruction) + (lower lanes on the same counter)
hipError_t launch_lds_atomic_order_selftest(uint32_t workgroups, uint32_t rounds, uint32_t seed, unsigned long long* mismatches,
                                            hipStream_t stream);
// The in-wave rank of the scatter kernel: request 0 = ballots, 1 = returning LDS adds verified per tile (ballot repair in place),
// 2 = those on a device that passed the lane-order self-test (run once per device, on `stream`, by the first caller), ballots
// elsewhere, 3 = 1 with an injected inversion (test hook).  radix_rank_mode returns what sorts on the CURRENT device use
// (0 ballots, 1 verified LDS adds, 2 with the inversion); *violations = what the self-test counted.
void radix_set_rank_request(int request);
int radix_rank_request();
int radix_rank_mode(hipStream_t stream, unsigned long long* violations);
hipError_t radix_rank_fallbacks(unsigned long long* tiles);   // tiles re-ranked with ballots after a failed order check
size_t radix_scratch_words(uint32_t n, uint32_t precount_blocks = 0);
// Pairs per lane of the pair expansion (expand_kernel): its launch of `grid` workgroups was sized for an upper bound of the
// `n` pairs; when far fewer are left the share of a lane halves (8 -> 4) so that the workgroups of the launch are all needed.
// Shared with the tile sort's first scan, which folds the expansion's per-workgroup digit counts into 4096-pair tiles.
__host__ __device__ inline uint32_t expand_pairs_per_lane(uint32_t grid, uint32_t n) {
    uint32_t per_lane = (uint32_t)kExpandTile / 256u;
    while (per_lane > 4u && (unsigned long long)grid * 256ull * (per_lane >> 1) >= (unsigned long long)n) per_lane >>= 1;
    return per_lane;
}
// Digit of the first pass of a `bits`-wide sort (shift 0): its mask.
uint32_t radix_first_digit_mask(int bits);
// Row stride (words) of the per-digit count rows in a sort's scratch: counts[digit * stride + tile].
uint32_t radix_count_stride(uint32_t n, uint32_t precount_blocks = 0);
// What a sort may be told beyond its keys (all optional).
struct RadixSortExtras {
    // The number of pairs is read from this device word by every kernel (it must not exceed n, which then only sizes the
    // launches and the scratch layout): a sort can be queued before its size is known on the host.
    const uint32_t* n_device = nullptr;
    // (with iota_payload, more than one pass, no n_device) items with this key leave the sort in its first pass -- not ranked,
    // not written; the sorted arrays hold the others, in order, and their tails are undefined.
    const uint32_t* drop_key = nullptr;
    // A hint that the most significant digit takes only a handful of values (the top byte of a positive float: sign and seven
    // exponent bits): that pass ranks with ballots, whose cost does not grow with lanes hitting one LDS counter.
    bool few_top_digits = false;
    // Work some extra workgroups of the first pass's count kernel do (the projection's tallies); after_first_count is recorded
    // right behind that launch.  With n == 0 or bits <= 0 neither happens.
    const TallyDuty* first_count_duty = nullptr;
    hipEvent_t after_first_count = nullptr;
    // != 0 (needs n_device): the first pass has no count kernel -- the kernel that wrote the keys, a launch of `precount_blocks`
    // workgroups each covering 256 * expand_pairs_per_lane(precount_blocks, *n_device) consecutive keys, left
    // counts[digit * radix_count_stride(n, precount_blocks) + workgroup] for all 256 digits in `scratch`; the first scan folds
    // them into 4096-key tiles.  The scratch must then be radix_scratch_words(n, precount_blocks) words.
    uint32_t precount_blocks = 0;
};
hipError_t radix_sort_pairs(uint32_t* scratch, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                            uint32_t* vals, uint32_t* vals_alt, bool iota_payload, bool want_sorted_keys,
                            uint32_t** keys_sorted, uint32_t** vals_sorted, hipStream_t stream,
                            const RadixSortExtras& extras = RadixSortExtras());

} // namespace gsr
