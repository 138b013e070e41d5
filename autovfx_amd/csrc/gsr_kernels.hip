// gsr_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the forward rasterizer.
//
// What each kernel replaces in the reference (DGR = sugar/gaussian_splatting/submodules/
// diff-gaussian-rasterization, under /root/reference):
//   preprocess_kernel   <- preprocessCUDA       DGR/cuda_rasterizer/forward.cu:155-256
//                          (+ in_frustum auxiliary.h:139-164, computeCov3D forward.cu:118-152,
//                           computeCov2D forward.cu:74-113, getRect auxiliary.h:46-56,
//                           ndc2Pix auxiliary.h:41-44, computeColorFromSH forward.cu:20-71)
//   mark_visible_kernel <- checkFrustum         DGR/cuda_rasterizer/rasterizer_impl.cu:54-66
//   duplicate_kernel    <- duplicateWithKeys    DGR/cuda_rasterizer/rasterizer_impl.cu:70-111
//   tile_ranges_kernel  <- identifyTileRanges   DGR/cuda_rasterizer/rasterizer_impl.cu:116-138
//   blend_kernel        <- renderCUDA           DGR/cuda_rasterizer/forward.cu:261-378
//
// (The backward pass lives in gsr_backward.hip; helpers shared with it in gsr_device.h.)
//
// Results are the reference's: fp32 in the reference's operation order,
// built with -ffp-contract=off so nothing is fused behind our back (the one fused operation, the
// compositing fma of the blend, is written out: see composite()).  The *structure* is not the
// reference's:
//   * the (tile, depth) order is produced by a 32-bit depth sort of Gaussians followed by a stable
//     tile-id sort of the expanded pairs (gsr_radix.hip), not one 64-bit sort;
//   * preprocess also computes, wave-cooperatively, which tiles of a splat's rectangle can reach
//     alpha >= 1/255 at all (exact-image tile culling): dead pairs are never emitted;
//   * scan + pair expansion are three spin-free kernels balanced by PAIRS (bin_gather / bin_offsets /
//     expand), so the nearest splats, which emit hundreds of pairs each, do not serialise a workgroup;
//   * tile ranges come from two binary searches per tile over the sorted tile keys;
//   * one wave64 blends one 8x8 quadrant of a tile, one pixel per lane: no workgroup barriers, a
//     per-quadrant reach test of every staged list entry, pixel state as wave-uniform scalar masks,
//     an exact pre-test that skips expf for pairs that cannot reach alpha >= 1/255, and optionally a
//     second feature set composited in the same walk (blend_quadrant_kernel<true>).
// The duplicate_kernel / blend_kernel variants are kept as A/B paths (GSR_OPT_SORT_IMPL = 0,
// GSR_OPT_BLEND_VARIANT = 0) with tests that they give the same bits.
#include "gsr_device.h"

// Profiling aid (python -m autovfx_amd.build --trace, scripts/kernel_trace.py): lane 0 of a workgroup stamps the
// 100 MHz wall clock into slot `slot` of its 8-word record.  Compiled out of the normal library.
#ifdef GSR_KERNEL_TRACE
__device__ unsigned long long* g_kernel_trace = nullptr;
extern "C" __attribute__((visibility("default"))) int gsr_debug_set_trace(void* device_words) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_kernel_trace), &device_words, sizeof device_words);
}
#define GSR_KTRACE(id, slot) do { if (threadIdx.x == 0 && g_kernel_trace) g_kernel_trace[(size_t)(id) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define GSR_KTRACE(id, slot) do { } while (0)
#endif

namespace gsr {
namespace {

// ------------------------------------------------------------------------------------------------
// K1: one lane per Gaussian.  Streaming, HBM-bound: 236 B in (M = 16) and 52 B out per visible
// Gaussian, 12 B in / 12 B out per culled one.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) preprocess_kernel(GaussianInputs in, Camera cam, GeometryArrays out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in_range = i < in.P;  // every lane stays to the end: the tile-mask phase below is wave-cooperative
    GSR_KTRACE(blockIdx.x, 0);

    const float* __restrict__ vm = cam.viewmatrix;
    const float* __restrict__ pm = cam.projmatrix;

    int radius_out = 0;
    uint32_t rect_area = 0;
    SplatBin bin = {0u, 1u, 0xFFFFFFFFu, 0u};
    uint32_t key = kCulledKey;
    // what the tile-mask phase needs of a candidate (a visible splat whose rectangle has <= 32 tiles)
    bool mask_candidate = false;
    float4 cand_conic = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 cand_centre = make_float2(0.f, 0.f);

    if (in_range) {
    if (out.ids != nullptr) out.ids[i] = (uint32_t)i;
    const F3 p = ld3(in.means3D + 3 * (size_t)i);

    // auxiliary.h:58-77 (sums left to right)
    const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
    const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
    const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
    const float vx = vm[0] * p.x + vm[4] * p.y + vm[8] * p.z + vm[12];
    const float vy = vm[1] * p.x + vm[5] * p.y + vm[9] * p.z + vm[13];
    const float vz = vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14];

    if (vz <= 0.2f) {
        // The reference printf+traps here when prefiltered is set (auxiliary.h:156-160); we record
        // the violation and keep the context alive.
        if (in.prefiltered) atomicOr(&out.counters->error_flag, 1u);
    } else {
        const float pw = 1.0f / (hw + 0.0000001f);
        const float ndc_x = hx * pw, ndc_y = hy * pw;

        // ---- 3D covariance (forward.cu:118-152) ----
        float c3[6];
        if (in.cov3D_precomp != nullptr) {
            const float* c = in.cov3D_precomp + 6 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = c[k];
        } else {
            const F3 s = ld3(in.scales + 3 * (size_t)i);
            const F4 q = *reinterpret_cast<const F4*>(in.rotations + 4 * (size_t)i);
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            Mat3 S = {{{in.scale_modifier * s.x, 0.f, 0.f}, {0.f, in.scale_modifier * s.y, 0.f},
                       {0.f, 0.f, in.scale_modifier * s.z}}};
            Mat3 R;  // glm column-major constructor: (a,b,c) is column 0
            R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[1][0] = 2.f * (x * y - r * z);       R.m[2][0] = 2.f * (x * z + r * y);
            R.m[0][1] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[2][1] = 2.f * (y * z - r * x);
            R.m[0][2] = 2.f * (x * z - r * y);       R.m[1][2] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);
            const Mat3 Mm = mul3(S, R);
            const Mat3 Sg = mul3(transpose3(Mm), Mm);
            c3[0] = Sg.m[0][0]; c3[1] = Sg.m[1][0]; c3[2] = Sg.m[2][0];
            c3[3] = Sg.m[1][1]; c3[4] = Sg.m[2][1]; c3[5] = Sg.m[2][2];
        }

        // ---- EWA 2D covariance (forward.cu:74-113) ----
        const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        Mat3 J = {{{cam.focal_x / vz, 0.f, 0.f}, {0.f, cam.focal_y / vz, 0.f},
                   {-(cam.focal_x * tx) / (vz * vz), -(cam.focal_y * ty) / (vz * vz), 0.f}}};
        Mat3 Wm;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Wm.m[a][b] = vm[4 * a + b];
        const Mat3 T = mul3(Wm, J);
        Mat3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
        const Mat3 cov = mul3(mul3(transpose3(T), transpose3(V)), T);
        const float ca = cov.m[0][0] + 0.3f, cb = cov.m[1][0], cc = cov.m[1][1] + 0.3f;

        const float det = ca * cc - cb * cb;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float mid = 0.5f * (ca + cc);
            const float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float rad = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
            const float px = ndc_to_pix(ndc_x, cam.width), py = ndc_to_pix(ndc_y, cam.height);
            const int irad = f2i_sat(rad);
            const TileRect rc = tile_rect(px, py, irad, cam.grid_x, cam.grid_y);
            const uint32_t area = (uint32_t)(rc.x1 - rc.x0) * (uint32_t)(rc.y1 - rc.y0);
            if (area != 0) {
                if (in.colors_precomp == nullptr) {
                    int deg = in.sh_degree < 3 ? in.sh_degree : 3;  // forward.cu:20-71 knows bands 0..3 (D may be 4)
                    if (deg > 2 && in.M < 16) deg = 2;  // never read past M coefficients
                    if (deg > 1 && in.M < 9) deg = 1;
                    if (deg > 0 && in.M < 4) deg = 0;
                    const F3 cp = ld3(cam.cam_pos);
                    const F3 col = sh_to_rgb(deg, p, cp, in.shs + 3 * (size_t)in.M * i);
                    *reinterpret_cast<F3*>(out.rgb + 3 * (size_t)i) = col;
                }
                const float4 conic_o = make_float4(cc * det_inv, -cb * det_inv, ca * det_inv, in.opacities[i]);
                float4* rec = reinterpret_cast<float4*>(out.raster + i);  // one 32-byte record, two 16-byte stores
                rec[0] = make_float4(px, py, conic_o.x, conic_o.y);
                const float skip_below = blend_skip_below(conic_o.w);
                rec[1] = make_float4(conic_o.z, conic_o.w, vz, skip_below);
                radius_out = irad;
                rect_area = area;
                bin.xy0 = (uint32_t)rc.x0 | ((uint32_t)rc.y0 << 16);
                bin.width = (uint32_t)(rc.x1 - rc.x0);
                bin.count = area;
                if (in.tile_cull && area <= 32u) {  // exact-image tile culling: mask computed below, by the whole wave
                    mask_candidate = true;
                    cand_conic = make_float4(conic_o.x, conic_o.y, conic_o.z, skip_below);  // w: the pre-test threshold
                    cand_centre = make_float2(px, py);
                }
                key = __float_as_uint(vz);
                if (key == kCulledKey) key = kCulledKey - 1u;  // a NaN depth with an all-ones payload must not look culled
            }
        }
    }
    }  // in_range
    GSR_KTRACE(blockIdx.x, 1);

    // ---- exact-image tile culling, wave-cooperative ----
    // A lane looping over the tiles of its own rectangle makes the wave run as long as its largest rectangle
    // (~25 iterations for ~6 tiles per splat on average: half of this kernel's vector instructions).  Instead the
    // (splat, tile) tests of the wave's 64 splats are flattened: test t belongs to the splat whose inclusive count
    // first exceeds t (binary search over the counts parked in LDS), every lane runs one test per iteration, the
    // ballot of the results is cut back into per-splat masks.  Iterations = total tests / 64.
    {
        __shared__ float4 s_conic[4][64];
        __shared__ float4 s_place[4][64];   // centre x, y, first tile (x | y << 16), rectangle width
        __shared__ uint32_t s_incl[4][64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t n_tests = mask_candidate ? rect_area : 0u;
        uint32_t incl = n_tests;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        const uint32_t excl = incl - n_tests;
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        if (total != 0u) {  // wave-uniform
            s_conic[wave][lane] = cand_conic;
            s_place[wave][lane] = make_float4(cand_centre.x, cand_centre.y, __uint_as_float(bin.xy0), __uint_as_float(bin.width));
            s_incl[wave][lane] = incl;
            __builtin_amdgcn_s_waitcnt(0);       // this wave's own LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            uint32_t mask = 0u;
            for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                const uint32_t t = t0 + (uint32_t)lane;
                int lo = 0, hi = 63;  // first splat whose inclusive count exceeds t
#pragma unroll
                for (int step = 0; step < 6; ++step) {
                    const int mid = (lo + hi) >> 1;
                    if (s_incl[wave][mid] > t) hi = mid; else lo = mid + 1;
                }
                const float4 co = s_conic[wave][lo], pl = s_place[wave][lo];
                const uint32_t xy0 = __float_as_uint(pl.z), w = __float_as_uint(pl.w);
                const uint32_t first = lo > 0 ? s_incl[wave][lo - 1] : 0u;
                const uint32_t local = t - first;  // < 32, w <= 32: (local + 0.5) / w is >= 1/64 away from an integer
                const uint32_t row = (uint32_t)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)w));
                const uint32_t col = local - row * w;
                const bool live = t < total && splat_reaches_tile(co, co.w, make_float2(pl.x, pl.y), (int)((xy0 & 0xFFFFu) + col),
                                                                  (int)((xy0 >> 16) + row));
                const unsigned long long b = __ballot(live);
                const uint32_t from = max(excl, t0), to = min(incl, t0 + 64u);  // my tests inside this batch
                if (from < to) {
                    const uint32_t bits = (uint32_t)(b >> (from - t0)) & (to - from >= 32u ? 0xFFFFFFFFu : (1u << (to - from)) - 1u);
                    mask |= bits << (from - excl);
                }
            }
            if (mask_candidate) {
                bin.mask = mask;
                bin.count = (uint32_t)__popc(mask);
            }
        }
    }

    GSR_KTRACE(blockIdx.x, 2);
    if (in_range) {
        out.radii[i] = radius_out;
        *reinterpret_cast<uint4*>(out.bins + i) = make_uint4(bin.xy0, bin.width, bin.mask, bin.count);
        out.depth_keys[i] = key;
    }
    // Pair totals, needed on the host before the binning arena can be sized: the live pairs (what
    // gets expanded and sorted) in the low word and the reference's num_rendered (sum of rectangle
    // areas, part of its return value) in the high word of one 64-bit add.  One atomic per wave,
    // spread over kRectPartials words (same-word atomics serialise at ~12 ns each).
    unsigned long long wave_tot = ((unsigned long long)rect_area << 32) | (unsigned long long)bin.count;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_tot += __shfl_xor(wave_tot, d);
    const unsigned long long emitting = __ballot(key != kCulledKey);
    if ((threadIdx.x & 63) == 0 && emitting != 0ull) {
        const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (kRectPartials - 1);
        if (wave_tot != 0ull) atomicAdd(out.counters->pair_totals + slot, wave_tot);
        atomicAdd(out.counters->visible + slot, (uint32_t)__popcll(emitting));
    }
    GSR_KTRACE(blockIdx.x, 3);
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ vm,
                                                           uint8_t* __restrict__ present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const F3 p = ld3(means3D + 3 * (size_t)i);
    const float vz = vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14];
    present[i] = (uint8_t)(vz > 0.2f);
}

// ------------------------------------------------------------------------------------------------
// K3: wave-cooperative pair expansion in depth order.  Lane L of a wave owns sorted position
// k = wave_base + L; the wave's LIVE pairs occupy [offsets[k0-1], offsets[k0+63]) contiguously, and
// all 64 lanes stride over that flat range, locating the owning Gaussian by binary search on the
// lanes' inclusive counts (6 ds_bpermute steps) and the tile by selecting the r-th set bit of the
// owner's live mask.  Writes are coalesced and exactly num_live long; a screen-filling splat does
// not serialise one lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t select_set_bit(uint32_t m, uint32_t r) {  // position of the r-th (0-based) set bit
    uint32_t pos = 0, c;
    c = (uint32_t)__popc(m & 0xFFFFu); if (r >= c) { r -= c; pos += 16; m >>= 16; }
    c = (uint32_t)__popc(m & 0xFFu);   if (r >= c) { r -= c; pos += 8;  m >>= 8; }
    c = (uint32_t)__popc(m & 0xFu);    if (r >= c) { r -= c; pos += 4;  m >>= 4; }
    c = (uint32_t)__popc(m & 0x3u);    if (r >= c) { r -= c; pos += 2;  m >>= 2; }
    if (r >= (m & 1u)) pos += 1;
    return pos;
}

__global__ void __launch_bounds__(256) duplicate_kernel(int P, int grid_x, int grid_y,
                                                        const uint32_t* __restrict__ depth_order,
                                                        const uint32_t* __restrict__ point_offsets,
                                                        const SplatBin* __restrict__ bins,
                                                        uint32_t* __restrict__ tile_keys,
                                                        uint32_t* __restrict__ point_list) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t incl = 0, excl = 0;
    if (k < P) {
        incl = point_offsets[k];
        excl = k > 0 ? point_offsets[k - 1] : 0u;
    } else if (P > 0) {
        incl = excl = point_offsets[P - 1];
    }
    const uint32_t base = __shfl(excl, 0);
    const uint32_t total = __shfl(incl, 63) - base;  // wave-uniform
    if (total == 0) return;

    uint32_t gid = 0, xy0 = 0, w = 1, mask = 0xFFFFFFFFu;
    if (incl != excl) {
        gid = depth_order[k];
        const uint4 b = *reinterpret_cast<const uint4*>(bins + gid);  // one 16-byte gather per splat
        xy0 = b.x;
        w = b.y;
        mask = b.z;  // all ones: every tile of the rectangle is live (or culling is off)
    }
    const uint32_t incl_rel = incl - base, excl_rel = excl - base;

    for (uint32_t t0 = 0; t0 < total; t0 += 64) {  // wave-uniform trip count: every lane shuffles
        const uint32_t t = t0 + lane;
        int lo = 0, hi = 63;  // smallest lane whose inclusive count exceeds t
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int mid = (lo + hi) >> 1;
            const uint32_t v = __shfl(incl_rel, mid);
            if (v > t) hi = mid; else lo = mid + 1;
        }
        const uint32_t o_excl = __shfl(excl_rel, lo);
        const uint32_t o_xy0 = __shfl(xy0, lo);
        const uint32_t o_w = __shfl(w, lo);
        const uint32_t o_gid = __shfl(gid, lo);
        const uint32_t o_mask = __shfl(mask, lo);
        if (t < total) {
            const uint32_t r = t - o_excl;                              // rank among the owner's live tiles
            const uint32_t local = o_mask == 0xFFFFFFFFu ? r : select_set_bit(o_mask, r);
            const uint32_t row = local / o_w, col = local - row * o_w;
            tile_keys[base + t] = ((o_xy0 >> 16) + row) * (uint32_t)grid_x + (o_xy0 & 0xFFFFu) + col;
            point_list[base + t] = o_gid;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3': scan + pair expansion as three spin-free kernels (GSR_OPT_SORT_IMPL = 1), balanced by PAIRS.
// Walking the splats in depth order costs one random 16-byte gather per splat (the records were written
// in Gaussian order); at 3 M splats that gather, not the arithmetic, is what the stage waits on, and the
// nearest splats emit hundreds of pairs each while the far ones emit one or two, so handing every
// workgroup the same number of SPLATS leaves the first workgroups running long after the rest are done.
//   bin_gather_kernel : one workgroup per kDupTile = 1024 sorted positions.  Gathers the records, scans the
//                       pair counts inside the tile, writes tile-local inclusive offsets, the tile total and
//                       the records again IN DEPTH ORDER (so nobody gathers a second time).
//   bin_offsets_kernel: adds the sum of all earlier tile totals (each workgroup sums them itself: 12 KB of
//                       L2-resident words, no chain, no look-back) -> POINT_OFFSETS, global and inclusive.
//   expand_kernel     : one workgroup per kPairTile = 4096 PAIRS, 16 consecutive pairs per lane.  Finds its
//                       first splat with two workgroup-wide counting searches (tile ends, then inside the
//                       tile), parks the splats' offsets in LDS (batches of 2048), and every lane binary-
//                       searches the owner of its FIRST pair only, then walks: next tile of the owner, next
//                       owner.  Every workgroup does the same work and writes one contiguous 32 KB slice of
//                       the two pair arrays with 16-byte stores.
// ------------------------------------------------------------------------------------------------
constexpr int kPairTile = 4096;
static_assert(kDupTile == 1024, "bin_gather_kernel: 256 lanes x 4 consecutive positions");

// Sum over the workgroup's 256 lanes, returned to every lane; scratch is 4 words of LDS.
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(256) bin_gather_kernel(int P, int V, const uint32_t* __restrict__ depth_order,
                                                         const SplatBin* __restrict__ bins,
                                                         uint4* __restrict__ sorted_bins /*xy0, width, mask, gid*/,
                                                         uint32_t* __restrict__ local_offsets,
                                                         uint32_t* __restrict__ tile_totals) {
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = (int)blockIdx.x * kDupTile + 4 * tid;  // 4 consecutive positions per lane
    uint32_t gid[4] = {0u, 0u, 0u, 0u};
    if (k0 + 3 < V) {
        const uint4 g = *reinterpret_cast<const uint4*>(depth_order + k0);
        gid[0] = g.x; gid[1] = g.y; gid[2] = g.z; gid[3] = g.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V) gid[j] = depth_order[k0 + j];
    }
    uint4 rec[4];
    uint32_t count[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rec[j] = make_uint4(0u, 1u, 0xFFFFFFFFu, gid[j]);
        count[j] = 0u;
        if (k0 + j < V) {
            const uint4 b = *reinterpret_cast<const uint4*>(bins + gid[j]);  // the one random gather per splat
            rec[j] = make_uint4(b.x, b.y, b.z, gid[j]);
            count[j] = b.w;
        }
    }
    const uint32_t mine = count[0] + count[1] + count[2] + count[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) before += s_wave[w];
    if (tid == 255) tile_totals[blockIdx.x] = before + mine;
    const uint32_t o0 = before + count[0], o1 = o0 + count[1], o2 = o1 + count[2], o3 = o2 + count[3];
    if (k0 + 3 < P) {
        *reinterpret_cast<uint4*>(local_offsets + k0) = make_uint4(o0, o1, o2, o3);
    } else {
        const uint32_t o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < P) local_offsets[k0 + j] = o[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k0 + j < V) sorted_bins[k0 + j] = rec[j];
}

__global__ void __launch_bounds__(256) bin_offsets_kernel(int P, const uint32_t* __restrict__ tile_totals,
                                                          uint32_t* __restrict__ offsets, uint32_t* __restrict__ tile_ends) {
    __shared__ uint32_t s_scratch[4];
    uint32_t part = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256) part += tile_totals[t];
    const uint32_t before = block_sum_256(part, s_scratch);
    if (threadIdx.x == 0) tile_ends[blockIdx.x] = before + tile_totals[blockIdx.x];
    const int k0 = (int)blockIdx.x * kDupTile + 4 * (int)threadIdx.x;
    if (k0 + 3 < P) {
        uint4* q = reinterpret_cast<uint4*>(offsets + k0);
        uint4 v = *q;
        v.x += before; v.y += before; v.z += before; v.w += before;
        *q = v;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < P) offsets[k0 + j] += before;
    }
}

// Walks the live tiles of one splat in order: row-major over the rectangle, or over the set bits of the mask.
struct TileWalker {
    uint32_t x0, y0, width, gid;
    uint32_t mask;      // remaining set bits (masked splats)
    uint32_t row, col;  // next tile (full rectangles)
    float inv_width;
    bool full;
    __device__ __forceinline__ void start(const uint4 rec, uint32_t r /*tiles to skip*/) {
        x0 = rec.x & 0xFFFFu; y0 = rec.x >> 16; width = rec.y; gid = rec.w;
        full = rec.z == 0xFFFFFFFFu;
        if (full) {
            row = r / width;
            col = r - row * width;
        } else {
            mask = rec.z;
            for (uint32_t i = 0; i < r; ++i) mask &= mask - 1u;  // r < 32
            inv_width = __builtin_amdgcn_rcpf((float)width);
        }
    }
    __device__ __forceinline__ uint32_t next(uint32_t grid_x) {
        uint32_t r_, c_;
        if (full) {
            r_ = row; c_ = col;
            if (++col == width) { col = 0u; ++row; }
        } else {
            const uint32_t pos = (uint32_t)__builtin_ctz(mask);
            mask &= mask - 1u;
            // pos < 32, width <= 32: (pos + 0.5) / width is at least 0.5 / 32 away from an integer, the
            // approximate reciprocal is off by parts in 2^22
            r_ = (uint32_t)(((float)pos + 0.5f) * inv_width);
            c_ = pos - r_ * width;
        }
        return (y0 + r_) * grid_x + x0 + c_;
    }
};

constexpr int kPairsPerLane = kPairTile / 256;  // consecutive pairs of one lane

__global__ void __launch_bounds__(256) expand_kernel(int V, uint32_t num_pairs, int grid_x,
                                                     const uint32_t* __restrict__ offsets /*global, inclusive*/,
                                                     const uint32_t* __restrict__ tile_ends /*offsets at tile ends*/,
                                                     const uint4* __restrict__ sorted_bins,
                                                     uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ point_list) {
    constexpr int kBatch = 2048;            // splats whose offsets are parked in LDS at a time
    __shared__ uint32_t s_incl[kBatch + 1]; // s_incl[0] = pairs before the batch's first splat
    __shared__ uint32_t s_scratch[4];
    const int tid = threadIdx.x;
    const uint32_t p_begin = blockIdx.x * (uint32_t)kPairTile;
    const uint32_t p_end = min(num_pairs, p_begin + (uint32_t)kPairTile);

    // first splat whose inclusive offset exceeds p_begin = number of splats with offset <= p_begin
    const int tiles = (V + kDupTile - 1) / kDupTile;
    uint32_t n_le = 0;
    for (int t = tid; t < tiles; t += 256) n_le += tile_ends[t] <= p_begin ? 1u : 0u;
    const int tile0 = (int)block_sum_256(n_le, s_scratch);  // tiles that end at or before p_begin
    n_le = 0;
    {
        const int k0 = tile0 * kDupTile + 4 * tid;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V) n_le += offsets[k0 + j] <= p_begin ? 1u : 0u;
    }
    int s0 = tile0 * kDupTile + (int)block_sum_256(n_le, s_scratch);

    const uint32_t my_begin = p_begin + (uint32_t)(kPairsPerLane * tid);
    const uint32_t my_end = min(p_end, my_begin + (uint32_t)kPairsPerLane);
    uint32_t keys[kPairsPerLane], ids[kPairsPerLane];
    while (s0 < V) {  // workgroup-uniform
        for (int i = tid; i <= kBatch; i += 256) {
            const int k = s0 - 1 + i;
            s_incl[i] = k < 0 ? 0u : offsets[min(k, V - 1)];
        }
        __syncthreads();
        const uint32_t lo_pair = max(my_begin, s_incl[0]);
        const uint32_t hi_pair = min(my_end, s_incl[kBatch]);
        if (lo_pair < hi_pair) {
            int lo = 1, hi = kBatch;  // smallest i with s_incl[i] > lo_pair: the owner of this lane's first pair
#pragma unroll
            for (int step = 0; step < 11; ++step) {
                const int mid = (lo + hi) >> 1;
                if (s_incl[mid] > lo_pair) hi = mid; else lo = mid + 1;
            }
            int owner = lo;
            uint32_t owner_end = s_incl[owner];
            TileWalker w;
            w.start(sorted_bins[s0 + owner - 1], lo_pair - s_incl[owner - 1]);
#pragma unroll
            for (int q = 0; q < kPairsPerLane; ++q) {
                const uint32_t p = my_begin + (uint32_t)q;
                if (p >= lo_pair && p < hi_pair) {
                    if (p >= owner_end) {
                        do { owner_end = s_incl[++owner]; } while (p >= owner_end);  // splats without live tiles
                        w.start(sorted_bins[s0 + owner - 1], 0u);
                    }
                    keys[q] = w.next((uint32_t)grid_x);
                    ids[q] = w.gid;
                }
            }
        }
        const bool done = s_incl[kBatch] >= p_end;
        __syncthreads();
        if (done) break;
        s0 += kBatch;
    }
    if (my_begin + kPairsPerLane <= p_end) {
#pragma unroll
        for (int q = 0; q < kPairsPerLane; q += 4) {
            *reinterpret_cast<uint4*>(tile_keys + my_begin + q) = make_uint4(keys[q], keys[q + 1], keys[q + 2], keys[q + 3]);
            *reinterpret_cast<uint4*>(point_list + my_begin + q) = make_uint4(ids[q], ids[q + 1], ids[q + 2], ids[q + 3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < kPairsPerLane; ++q)
            if (my_begin + q < p_end) {
                tile_keys[my_begin + q] = keys[q];
                point_list[my_begin + q] = ids[q];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// K5: ranges[t] = [lower_bound(t), lower_bound(t+1)) over the sorted tile keys; (0,0) when empty,
// which is what the reference's memset + boundary scan leaves behind.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct ArenaHeaders3 { ArenaHeader h[3]; void* dst[3]; };

__global__ void __launch_bounds__(256) tile_ranges_kernel(uint32_t n, int num_tiles,
                                                          const uint32_t* __restrict__ keys,
                                                          uint2* __restrict__ ranges, ArenaHeaders3 headers) {
    // the three arena headers ride along (one launch less per call)
    if (blockIdx.x == 0 && threadIdx.x < 3) *reinterpret_cast<ArenaHeader*>(headers.dst[threadIdx.x]) = headers.h[threadIdx.x];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= num_tiles) return;
    const uint32_t b = lower_bound_u32(keys, n, (uint32_t)t);
    const uint32_t e = lower_bound_u32(keys, n, (uint32_t)t + 1u);
    ranges[t] = (e > b) ? make_uint2(b, e) : make_uint2(0u, 0u);
}

// ------------------------------------------------------------------------------------------------
// K6: one wave64 per 16x16 tile.  Lane (lane & 7, lane >> 3) owns the pixel at that position in
// each of the tile's four 8x8 quadrants (pixel q sits in quadrant q).  The test part of the inner
// loop shares dx/dy between the four pixels; the exact part runs once per quadrant and is skipped
// wave-uniformly for quadrants the splat does not reach, which is most of them for small splats.
//
// Per batch of 64 list entries every lane gathers one entry (id -> xy, conic+opacity, rgb, depth)
// and parks it in LDS; the wave then walks the batch reading each entry as a broadcast (all lanes
// the same address: conflict-free) and updating its 4 pixels.  The next batch's gathers are issued
// before the walk so their latency hides behind it.
//
// Exactness: per (pixel, entry) the arithmetic is the reference's (forward.cu:331-364).  The only
// shortcut is `power < skip_below`, with skip_below = -ln(255*o) - 1e-4: below it o*exp(power) is
// < 1/255 by a margin ~100x the combined rounding error of logf/expf/the products, so the
// reference's `alpha < 1/255` test would have skipped the pair too; pairs inside the margin take
// the exact path.
// ------------------------------------------------------------------------------------------------
// C += feature * alpha * T (forward.cu:357-360) with the last product fused into the addition, fma(feature * alpha,
// T, C): what nvcc, which contracts by default, makes of that line on the reference's own hardware, and one
// full-rate fused instruction in place of a multiply and an add.  This library is otherwise built without
// contraction; this is the one place it is written out, because it is on the per-pair-per-pixel path and touches
// only the float images (transmittance, the stopping rule and every integer output do not depend on it).
__device__ __forceinline__ float composite(float C, float feature, float alpha, float T) {
    return __builtin_fmaf(feature * alpha, T, C);
}
// Two channels at once: v_pk_mul_f32 + v_pk_fma_f32, written with a vector type so that the pairing does not depend
// on what the SLP vectorizer decides.  Same roundings as two calls of composite().
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f composite2(v2f C, v2f feature, float alpha, float T) {
    return __builtin_elementwise_fma(feature * alpha, (v2f){T, T}, C);
}

__global__ void __launch_bounds__(64, 8) blend_kernel(int W, int H, int grid_x, int num_tiles,
                                                   const uint2* __restrict__ ranges,
                                                   const uint32_t* __restrict__ point_list,
                                                   const SplatRaster* __restrict__ raster,
                                                   const float* __restrict__ features,
                                                   const float* __restrict__ background,
                                                   float* __restrict__ out_color, float* __restrict__ out_depth,
                                                   float* __restrict__ out_alpha,
                                                   uint32_t* __restrict__ n_contrib) {
    __shared__ BlendEntryA sA[64];  // 2.75 KiB per wave: 32 single-wave workgroups fit a CU's 160 KiB
    __shared__ BlendEntryB sB[64];
    __shared__ BlendEntryC sC[64];
    __shared__ float sD[64];

    const int tile = xcd_band_tile(blockIdx.x, num_tiles);
    const int lane = threadIdx.x;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    constexpr int kQ = kTile / 2;  // quadrant edge
    const int px0 = tile_x * kTile + (lane & 7);
    const int py0 = tile_y * kTile + (lane >> 3);
    const float fx0 = (float)px0, fx1 = (float)(px0 + kQ);
    const float fy0 = (float)py0, fy1 = (float)(py0 + kQ);

    // pixel q lives in quadrant q = 2*row + col
    bool inside[4] = {px0 < W && py0 < H, px0 + kQ < W && py0 < H, px0 < W && py0 + kQ < H,
                      px0 + kQ < W && py0 + kQ < H};
    bool done[4] = {!inside[0], !inside[1], !inside[2], !inside[3]};
    float T[4] = {1.f, 1.f, 1.f, 1.f};
    float Cr[4] = {0.f, 0.f, 0.f, 0.f}, Cg[4] = {0.f, 0.f, 0.f, 0.f}, Cb[4] = {0.f, 0.f, 0.f, 0.f};
    float Dz[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t last[4] = {0u, 0u, 0u, 0u};

    const uint2 range = ranges[tile];
    const uint32_t count = range.y - range.x;

    // registers holding the batch in flight
    float2 g_xy = make_float2(0.f, 0.f);
    float4 g_co = make_float4(0.f, 0.f, 0.f, 0.f);
    F3 g_rgb = {0.f, 0.f, 0.f};
    float g_z = 0.f, g_skip = 0.f;
    auto gather = [&](uint32_t first) {
        const uint32_t e = first + (uint32_t)lane;
        if (e < count) {
            const uint32_t id = point_list[range.x + e];
            const float4* rec = reinterpret_cast<const float4*>(raster + id);  // 32 bytes, one cache line
            const float4 r0 = rec[0], r1 = rec[1];
            g_xy = make_float2(r0.x, r0.y);
            g_co = make_float4(r0.z, r0.w, r1.x, r1.y);
            g_z = r1.z;
            g_skip = r1.w;
            g_rgb = ld3(features + 3 * (size_t)id);
        }
    };
    if (count > 0) gather(0);

    for (uint32_t first = 0; first < count; first += 64) {
        // park the gathered batch (single-wave workgroup: the barriers only order this wave's own
        // LDS reads of the previous batch / writes of this one / broadcast reads below)
        __syncthreads();
        sA[lane] = BlendEntryA{g_xy.x, g_xy.y, g_co.x, g_co.y};
        sB[lane] = BlendEntryB{g_co.z, g_skip};
        sC[lane] = BlendEntryC{g_co.w, g_rgb.x, g_rgb.y, g_rgb.z};
        sD[lane] = g_z;
        __syncthreads();
        if (first + 64 < count) gather(first + 64);

        const int n = (int)min(64u, count - first);
        for (int j = 0; j < n; ++j) {
            const BlendEntryA a = sA[j];
            const BlendEntryB b = sB[j];
            const float dx0 = a.x - fx0, dx1 = a.x - fx1;
            const float dy0 = a.y - fy0, dy1 = a.y - fy1;
            const float ax0 = a.cxx * dx0 * dx0, ax1 = a.cxx * dx1 * dx1;
            const float by0 = b.cyy * dy0 * dy0, by1 = b.cyy * dy1 * dy1;
            const float m0 = a.cxy * dx0, m1 = a.cxy * dx1;
            float power[4];
            power[0] = -0.5f * (ax0 + by0) - m0 * dy0;
            power[1] = -0.5f * (ax1 + by0) - m1 * dy0;
            power[2] = -0.5f * (ax0 + by1) - m0 * dy1;
            power[3] = -0.5f * (ax1 + by1) - m1 * dy1;
            bool live[4];
            bool any_live = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                live[q] = !done[q] && !(power[q] > 0.0f) && !(power[q] < b.skip_below);
                any_live |= live[q];
            }
            if (!__any(any_live)) continue;  // wave-uniform: nobody needs expf or the colour
            const BlendEntryC c = sC[j];
            const float z = sD[j];
            const uint32_t position = first + (uint32_t)j + 1u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!live[q]) continue;
                const float alpha = fminf(0.99f, c.opacity * expf(power[q]));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T[q] * (1.f - alpha);
                if (test_T < 0.0001f) { done[q] = true; continue; }
                Cr[q] = composite(Cr[q], c.r, alpha, T[q]);
                Cg[q] = composite(Cg[q], c.g, alpha, T[q]);
                Cb[q] = composite(Cb[q], c.b, alpha, T[q]);
                Dz[q] = composite(Dz[q], z, alpha, T[q]);
                T[q] = test_T;
                last[q] = position;
            }
            if (__all(done[0] && done[1] && done[2] && done[3])) break;
        }
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
    }

    const float bg0 = background[0], bg1 = background[1], bg2 = background[2];
    const size_t plane = (size_t)W * (size_t)H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!inside[q]) continue;
        const size_t pid = (size_t)W * (size_t)(py0 + kQ * (q >> 1)) + (size_t)(px0 + kQ * (q & 1));
        out_alpha[pid] = 1.f - T[q];
        if (n_contrib != nullptr) n_contrib[pid] = last[q];
        out_color[pid] = Cr[q] + T[q] * bg0;
        out_color[plane + pid] = Cg[q] + T[q] * bg1;
        out_color[2 * plane + pid] = Cb[q] + T[q] * bg2;
        out_depth[pid] = Dz[q];
    }
}

// ------------------------------------------------------------------------------------------------
// K6q: one wave64 per 8x8 QUADRANT of a 16x16 tile (4 single-wave workgroups per tile), one pixel
// per lane.  Same list, same per-pixel arithmetic as blend_kernel; what changes is the shape of
// the work:
//   * four times as many, four times smaller work items: a tile whose list is long or never
//     saturates no longer pins one wave for the whole launch, and a quadrant stops as soon as ITS
//     64 pixels are done;
//   * while staging a batch of 64 entries each lane also runs the conservative reach test of its
//     entry against this quadrant; the wave then walks only the set bits of the ballot, so entries
//     that cannot touch the quadrant cost no LDS read and no per-pixel work at all.
// ------------------------------------------------------------------------------------------------
__global__ void exp_selftest_kernel(uint32_t first_bits, uint32_t count, unsigned long long* mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = __uint_as_float(first_bits + i);
    const float a = expf(x), b = exp_nonpositive(x);
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) atomicAdd(mismatches, 1ull);
}

// kExtra: a second per-Gaussian feature triple (extra_features[P,3] -> out_extra[3,H,W]) is composited in the
// same walk with the same alpha and transmittance -- what the reference's render() obtains from a second full
// rasterizer pass for its normal map (gaussian_renderer/__init__.py:176-184): identical arithmetic per channel,
// one list walk instead of two.
template <bool kExtra>
__global__ void __launch_bounds__(64, 8) blend_quadrant_kernel(int W, int H, int grid_x, int num_tiles,
                                                            const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list,
                                                            const SplatRaster* __restrict__ raster,
                                                            const float* __restrict__ features,
                                                            const float* __restrict__ extra_features,
                                                            const float* __restrict__ background,
                                                            float* __restrict__ out_color,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ out_alpha,
                                                            float* __restrict__ out_extra,
                                                            uint32_t* __restrict__ n_contrib) {
    __shared__ BlendEntry s_entry[64];
    __shared__ float4 s_extra[kExtra ? 64 : 1];

    constexpr int kQ = kTile / 2;
    const int item = xcd_band_tile(blockIdx.x, 4 * num_tiles);  // the 4 quadrants of a tile share an XCD
    const int tile = item >> 2, quad = item & 3;
    const int lane = threadIdx.x;
    const int qx0 = (tile % grid_x) * kTile + kQ * (quad & 1);
    const int qy0 = (tile / grid_x) * kTile + kQ * (quad >> 1);
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float fx = (float)px, fy = (float)py;
    const bool inside = px < W && py < H;
    // Which pixels have stopped is a wave-uniform 64-bit mask in scalar registers: tests on it ("any pixel
    // live?", "all done?") cost no vector instructions, and it gates the per-pixel block through the
    // execution mask directly.
    unsigned long long done_mask = __ballot(!inside);
    if (done_mask == ~0ull) return;  // quadrant entirely outside the image: nothing to write

    float T = 1.f, Eb = 0.f;
    v2f Crg = {0.f, 0.f}, Cbz = {0.f, 0.f}, Erg = {0.f, 0.f};  // red|green, blue|depth, second set red|green
    uint32_t last = 0u;

    const uint2 range = ranges[tile];
    const uint32_t count = range.y - range.x;
    GSR_KTRACE(blockIdx.x, 4);

    float2 g_xy = make_float2(0.f, 0.f);
    float4 g_co = make_float4(0.f, 0.f, 0.f, 0.f);
    F3 g_rgb = {0.f, 0.f, 0.f}, g_ext = {0.f, 0.f, 0.f};
    float g_z = 0.f, g_skip = 0.f;
    auto gather = [&](uint32_t first) {
        const uint32_t e = first + (uint32_t)lane;
        if (e < count) {
            const uint32_t id = point_list[range.x + e];
            const float4* rec = reinterpret_cast<const float4*>(raster + id);  // 32 bytes, one cache line
            const float4 r0 = rec[0], r1 = rec[1];
            g_xy = make_float2(r0.x, r0.y);
            g_co = make_float4(r0.z, r0.w, r1.x, r1.y);
            g_z = r1.z;
            g_skip = r1.w;
            g_rgb = ld3(features + 3 * (size_t)id);
            if (kExtra) g_ext = ld3(extra_features + 3 * (size_t)id);
        }
    };
    if (count > 0) gather(0);

    for (uint32_t first = 0; first < count; first += 64) {
        const bool mine = first + (uint32_t)lane < count;
        unsigned long long todo = __ballot(mine && splat_reaches_rect(g_co, g_skip, g_xy, qx0, qy0, kQ, kQ));
        if (todo != 0ull) {
            __syncthreads();  // single-wave workgroup: orders this wave's LDS reads / writes only
            float4* rec = reinterpret_cast<float4*>(&s_entry[lane]);
            // The conic's diagonal is parked already multiplied by -0.5: scaling by a power of two commutes with every
            // rounding of -0.5 * (cxx*dx*dx + cyy*dy*dy), so the walk below gets the same bits with one multiply less
            // per (entry, pixel).
            rec[0] = make_float4(g_xy.x, g_xy.y, -0.5f * g_co.x, g_co.y);
            rec[1] = make_float4(-0.5f * g_co.z, g_skip, g_co.w, 0.f);
            rec[2] = make_float4(g_rgb.x, g_rgb.y, g_rgb.z, g_z);
            if (kExtra) s_extra[lane] = make_float4(g_ext.x, g_ext.y, g_ext.z, 0.f);
            __syncthreads();
        }
        if (first + 64 < count) gather(first + 64);

        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            // The entry's LDS offset is wave-uniform; parked in ONE vector register (opaque to the compiler, which
            // would otherwise re-create it from the scalar before each of the three reads of the record).
            uint32_t entry_offset;
            asm("v_mov_b32 %0, %1" : "=v"(entry_offset) : "s"(j * (int)sizeof(BlendEntry)));
            const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_entry) + entry_offset);
            const float4 ra = rec[0], rb = rec[1];
            struct { float x, y, mh_cxx, cxy; } a = {ra.x, ra.y, ra.z, ra.w};   // mh_ = times minus one half
            struct { float mh_cyy, skip_below, opacity; } b = {rb.x, rb.y, rb.z};
            // One list entry against this lane's pixel: forward.cu:331-364, same bits as
            // -0.5f * (cxx * dx * dx + cyy * dy * dy) - cxy * dx * dy.
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = (a.mh_cxx * dx * dx + b.mh_cyy * dy * dy) - a.cxy * dx * dy;
            // (one ballot per comparison: the ballot of a conjunction goes through a vector register and back)
            const unsigned long long live = __ballot(!(power > 0.0f)) & __ballot(!(power < b.skip_below)) & ~done_mask;
            if (live == 0ull) continue;
            // From here every lane computes (a vector instruction costs the same with 1 or 64 lanes enabled);
            // the outcome of a lane that is not live is masked out below.  All masks stay wave-uniform scalars
            // because they are only combined in uniform control flow.
            const float alpha = fminf(0.99f, b.opacity * exp_nonpositive(power));
            const unsigned long long blends = live & __ballot(!(alpha < 1.0f / 255.0f));
            if (blends == 0ull) continue;
            const float test_T = T * (1.f - alpha);
            const unsigned long long stops = blends & __ballot(test_T < 0.0001f);
            done_mask |= stops;
            const unsigned long long adds = blends & ~stops;
            if (adds != 0ull && __builtin_amdgcn_inverse_ballot_w64(adds)) {
                const float4 c = rec[2];  // r g b z
                Crg = composite2(Crg, (v2f){c.x, c.y}, alpha, T);
                Cbz = composite2(Cbz, (v2f){c.z, c.w}, alpha, T);
                if (kExtra) {
                    const float4 e = s_extra[j];
                    Erg = composite2(Erg, (v2f){e.x, e.y}, alpha, T);
                    Eb = composite(Eb, e.z, alpha, T);
                }
                T = test_T;
                last = first + (uint32_t)j + 1u;
            }
            if (done_mask == ~0ull) break;
        }
        if (done_mask == ~0ull) break;
    }

#ifdef GSR_KERNEL_TRACE
    GSR_KTRACE(blockIdx.x, 5);
    if (threadIdx.x == 0 && g_kernel_trace) g_kernel_trace[(size_t)blockIdx.x * 8 + 6] = count;
#endif
    if (inside) {
        const size_t plane = (size_t)W * (size_t)H;
        const size_t pid = (size_t)W * (size_t)py + (size_t)px;
        out_alpha[pid] = 1.f - T;
        if (n_contrib != nullptr) n_contrib[pid] = last;
        out_color[pid] = Crg.x + T * background[0];
        out_color[plane + pid] = Crg.y + T * background[1];
        out_color[2 * plane + pid] = Cbz.x + T * background[2];
        out_depth[pid] = Cbz.y;
        if (kExtra) {
            out_extra[pid] = Erg.x + T * background[0];
            out_extra[plane + pid] = Erg.y + T * background[1];
            out_extra[2 * plane + pid] = Eb + T * background[2];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Frame hand-off: planar fp32 RGB + alpha -> planar RGBA8 with the rounding torchvision's
// save_image applies to the PNGs the reference writes (scene_representation.py:427):
// clamp(x * 255 + 0.5, 0, 255) truncated.  Pure streaming: 16 B in, 4 B out per pixel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quantize8(float v) {
    const float q = fminf(255.0f, fmaxf(0.0f, v * 255.0f + 0.5f));
    return (uint32_t)(int)q;
}

__global__ void __launch_bounds__(256) pack_rgba8_kernel(const float* __restrict__ color,
                                                         const float* __restrict__ alpha,
                                                         uint8_t* __restrict__ out, size_t n_pixels, int vec_ok) {
    // grid.y = plane (0..2 colour, 3 alpha); each thread owns 4 consecutive pixels of its plane
    const int plane = blockIdx.y;
    const float* __restrict__ src = plane < 3 ? color + (size_t)plane * n_pixels : alpha;
    uint8_t* __restrict__ dst = out + (size_t)plane * n_pixels;
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (base >= n_pixels) return;
    if (vec_ok && base + 3 < n_pixels) {
        const float4 v = *reinterpret_cast<const float4*>(src + base);
        const uint32_t packed = quantize8(v.x) | (quantize8(v.y) << 8) | (quantize8(v.z) << 16) | (quantize8(v.w) << 24);
        *reinterpret_cast<uint32_t*>(dst + base) = packed;
    } else {
        for (size_t i = base; i < n_pixels && i < base + 4; ++i) dst[i] = (uint8_t)quantize8(src[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// Compositor: the per-pixel layer arithmetic of blender/blend_all.py::blend_frames (:236-300,341-343),
// one lane per pixel over interleaved RGBA8 layers and fp32 depth maps.  Pure streaming (4-24 B in per
// layer, 4 B out); fp32 in numpy's operation order, so the uint8 output is bit-identical to the reference.
// ------------------------------------------------------------------------------------------------
struct CompositeLayers {
    const uchar4* bg_c;        // 3DGS background frame
    const uchar4* o_c;         // Blender object pass
    const float* o_d;
    const uchar4* s_c;         // shadow-catcher pass
    const float* s_d;
    const uchar4* o_s_c;       // object + shadow-catcher pass
    const uchar4* o_gs_c;      // nullable: 3DGS objects re-rendered by Blender
    const float* o_gs_d;
    const uchar4* s_f_c;       // nullable: smoke / fire
    const float* s_f_d;
    const uchar4* s_f_c_pre;   // nullable: premultiplied fire
    uchar4* out;
};

__device__ __forceinline__ float4 to_f4(uchar4 v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ unsigned char to_u8(float v) { return (unsigned char)(int)fminf(fmaxf(v, 0.f), 255.f); }

__global__ void __launch_bounds__(256) composite_kernel(CompositeLayers L, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    const bool has_3dgs = L.o_gs_c != nullptr, has_smoke = L.s_f_c != nullptr, has_fire = L.s_f_c_pre != nullptr;
    const float4 bg = to_f4(L.bg_c[i]), oc = to_f4(L.o_c[i]), sc = to_f4(L.s_c[i]), osc = to_f4(L.o_s_c[i]);
    const float od = L.o_d[i], sd = L.s_d[i];
    float f[4] = {bg.x, bg.y, bg.z, bg.w};

    // step 1: shadows onto the background (blend_all.py:241-280)
    float non_obj_3dgs_alpha = 1.f;
    float ogd = 0.f;
    if (has_3dgs) {
        ogd = L.o_gs_d[i];
        non_obj_3dgs_alpha = (sd <= ogd) ? 1.0f : 1.f - to_f4(L.o_gs_c[i]).w / 255.f;
    }
    float obj_alpha = oc.w / 255.f;
    bool depth_mask = od <= sd;
    float obj_alpha_smoke = 0.f;
    bool depth_mask_smoke = false;
    if (has_smoke) {
        obj_alpha_smoke = to_f4(L.s_f_c[i]).w / 255.f;
        depth_mask_smoke = L.s_f_d[i] <= sd;
        obj_alpha = fmaxf(obj_alpha, obj_alpha_smoke);
        depth_mask = depth_mask || depth_mask_smoke;
    }
    const bool obj_mask = obj_alpha > 0.0f;
    const bool obj_visible = obj_mask && depth_mask;
    if (!obj_visible) obj_alpha = 0.0f;
    const float non_object_alpha = 1.f - obj_alpha;
    if (has_3dgs && ogd <= od) obj_alpha *= non_obj_3dgs_alpha;
    const float fg_alpha = osc.w / 255.f;
    const float sca = has_3dgs ? non_object_alpha * fg_alpha * non_obj_3dgs_alpha : non_object_alpha * fg_alpha;
    float cd[4] = {1.f, 1.f, 1.f, 1.f};
    if (sca > 0.0f) {
        cd[0] = osc.x / (sc.x + 1e-6f);
        cd[1] = osc.y / (sc.y + 1e-6f);
        cd[2] = osc.z / (sc.z + 1e-6f);
    }
    bool all_close = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cd[c] = clip01(cd[c]);
        all_close = all_close && (fabsf(cd[c] - 1.f) < 0.01f);
    }
    if (!all_close) {
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = f[c] * cd[c] * sca + f[c] * (1.f - sca);
    }

    // step 2: objects (and fire) over the shadowed background (:285-291)
    const float t[3] = {f[0], f[1], f[2]};
    if (obj_visible) {
        f[0] = oc.x * obj_alpha + t[0] * (1.f - obj_alpha);
        f[1] = oc.y * obj_alpha + t[1] * (1.f - obj_alpha);
        f[2] = oc.z * obj_alpha + t[2] * (1.f - obj_alpha);
    }
    if (has_fire && depth_mask_smoke) {
        const float4 pre = to_f4(L.s_f_c_pre[i]);
        f[0] = pre.x + t[0] * (1.f - obj_alpha_smoke);
        f[1] = pre.y + t[1] * (1.f - obj_alpha_smoke);
        f[2] = pre.z + t[2] * (1.f - obj_alpha_smoke);
    }
    L.out[i] = make_uchar4(to_u8(f[0]), to_u8(f[1]), to_u8(f[2]), to_u8(f[3]));
}


} // namespace

hipError_t launch_preprocess(const GaussianInputs& in, const Camera& cam, const GeometryArrays& out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(preprocess_kernel, dim3(div_up(in.P, 256)), dim3(256), 0, stream, in, cam, out);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                               hipStream_t stream) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, means3D, viewmatrix,
                       present);
    return hipGetLastError();
}

hipError_t launch_duplicate(int P, const Camera& cam, const uint32_t* depth_order, const uint32_t* point_offsets,
                            const SplatBin* bins, uint32_t* tile_keys, uint32_t* point_list, hipStream_t stream) {
    hipLaunchKernelGGL(duplicate_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, cam.grid_x, cam.grid_y,
                       depth_order, point_offsets, bins, tile_keys, point_list);
    return hipGetLastError();
}

hipError_t launch_scan_expand(int P, int V, uint32_t num_pairs, const Camera& cam, const uint32_t* depth_order,
                              const SplatBin* bins, uint4* sorted_bins, uint32_t* tile_totals, uint32_t* point_offsets,
                              uint32_t* tile_keys, uint32_t* point_list, hipStream_t stream) {
    const int tiles = div_up(P, kDupTile);
    hipLaunchKernelGGL(bin_gather_kernel, dim3(tiles), dim3(256), 0, stream, P, V, depth_order, bins, sorted_bins,
                       point_offsets, tile_totals);
    uint32_t* tile_ends = tile_totals + tiles;
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(tiles), dim3(256), 0, stream, P, tile_totals, point_offsets, tile_ends);
    if (num_pairs > 0)
        hipLaunchKernelGGL(expand_kernel, dim3((num_pairs + kPairTile - 1) / kPairTile), dim3(256), 0, stream, V, num_pairs,
                           cam.grid_x, point_offsets, tile_ends, sorted_bins, tile_keys, point_list);
    return hipGetLastError();
}

hipError_t launch_exp_selftest(uint32_t first_bits, uint32_t count, unsigned long long* mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(exp_selftest_kernel, dim3(div_up((int)count, 256)), dim3(256), 0, stream, first_bits, count, mismatches);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(uint32_t num_rendered, int num_tiles, const uint32_t* sorted_tile_keys, uint2* ranges,
                              void* const header_dst[3], const ArenaHeader headers[3], hipStream_t stream) {
    ArenaHeaders3 a;
    for (int i = 0; i < 3; ++i) { a.h[i] = headers[i]; a.dst[i] = header_dst[i]; }
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(div_up(num_tiles, 256)), dim3(256), 0, stream, num_rendered,
                       num_tiles, sorted_tile_keys, ranges, a);
    return hipGetLastError();
}

hipError_t launch_composite(int width, int height, const void* bg_c, const void* o_c, const float* o_d,
                            const void* s_c, const float* s_d, const void* o_s_c, const void* o_gs_c,
                            const float* o_gs_d, const void* s_f_c, const float* s_f_d, const void* s_f_c_pre,
                            void* out, hipStream_t stream) {
    CompositeLayers L;
    L.bg_c = (const uchar4*)bg_c; L.o_c = (const uchar4*)o_c; L.o_d = o_d; L.s_c = (const uchar4*)s_c; L.s_d = s_d;
    L.o_s_c = (const uchar4*)o_s_c; L.o_gs_c = (const uchar4*)o_gs_c; L.o_gs_d = o_gs_d;
    L.s_f_c = (const uchar4*)s_f_c; L.s_f_d = s_f_d; L.s_f_c_pre = (const uchar4*)s_f_c_pre; L.out = (uchar4*)out;
    const size_t n = (size_t)width * height;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, L, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The elementwise work of the reference's per-frame render() around its two rasterizer passes
// (sugar/gaussian_splatting/gaussian_renderer/__init__.py:118-146,169-208), which PyTorch runs as ~40 small
// launches (and one 3x3 GEMM over two million rows), as two kernels.  Same formulas, same operation order as the
// Python (sums left to right, F.normalize's max(norm, 1e-12)); differences against the PyTorch kernels are at
// the level of their own fused-multiply-add contraction.
//   view_normals_kernel : per Gaussian, view direction -> flip the shortest-axis normal towards the camera
//                         (utils/general_utils.py:151-157) -> unit length -> * 0.5 + 0.5  (colours of pass 2)
//   normal_maps_kernel  : per pixel, (raw - 0.5) * 2 -> unit normal map [H,W,3]; and the pseudo normal from the
//                         depth map: un-project the 4 neighbours (get_ray_directions :41-80, c2w rotation),
//                         cross product of the central differences (depth_pcd2normal :22-38), zero border.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ F3 unit3(F3 v) {  // torch.nn.functional.normalize(p=2, eps=1e-12)
    const float n = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z), 1e-12f);
    return F3{v.x / n, v.y / n, v.z / n};
}

__global__ void __launch_bounds__(256) view_normals_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ axis,
                                                           const float* __restrict__ cam_pos, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const F3 p = ld3(means3D + 3 * (size_t)i), a = ld3(axis + 3 * (size_t)i), c = ld3(cam_pos);
    const F3 d = {p.x - c.x, p.y - c.y, p.z - c.z};
    const float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    const F3 dir = {d.x / len, d.y / len, d.z / len};
    const float dot = a.x * -dir.x + a.y * -dir.y + a.z * -dir.z;
    const float s = dot >= 0.f ? 1.f : -1.f;
    const F3 m = {a.x * s, a.y * s, a.z * s};
    const float ml = sqrtf(m.x * m.x + m.y * m.y + m.z * m.z);
    *reinterpret_cast<F3*>(out + 3 * (size_t)i) = F3{m.x / ml * 0.5f + 0.5f, m.y / ml * 0.5f + 0.5f, m.z / ml * 0.5f + 0.5f};
}

struct NormalMapArgs {
    int W, H;
    const float* normal_rgb;  // [3,H,W]
    const float* depth;       // [H,W]
    const float* c2w;         // device, 16 floats row-major: the 4x4 the Python calls c2w
    float fx, fy, cx, cy;
    float* normal;            // [H,W,3]
    float* pseudo;            // [H,W,3]
};

__global__ void __launch_bounds__(256) normal_maps_kernel(NormalMapArgs a) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= a.W || v >= a.H) return;
    const size_t plane = (size_t)a.W * a.H, pid = (size_t)v * a.W + u;
    const F3 raw = {a.normal_rgb[pid], a.normal_rgb[plane + pid], a.normal_rgb[2 * plane + pid]};
    *reinterpret_cast<F3*>(a.normal + 3 * pid) = unit3(F3{(raw.x - 0.5f) * 2.0f, (raw.y - 0.5f) * 2.0f, (raw.z - 0.5f) * 2.0f});

    F3 n = {0.f, 0.f, 0.f};
    if (u >= 1 && v >= 1 && u < a.W - 1 && v < a.H - 1) {
        const float* __restrict__ m = a.c2w;  // uniform: scalar loads
        auto point = [&](int x, int y) {  // rays_o + rays_d * depth
            const float dx = ((float)x - a.cx + 0.5f) / a.fx, dy = ((float)y - a.cy + 0.5f) / a.fy;
            const float z = a.depth[(size_t)y * a.W + x];
            return F3{m[3] + (dx * m[0] + dy * m[1] + m[2]) * z, m[7] + (dx * m[4] + dy * m[5] + m[6]) * z,
                      m[11] + (dx * m[8] + dy * m[9] + m[10]) * z};
        };
        const F3 right = point(u + 1, v), left = point(u - 1, v), top = point(u, v - 1), bottom = point(u, v + 1);
        const F3 h = {right.x - left.x, right.y - left.y, right.z - left.z};
        const F3 w = {top.x - bottom.x, top.y - bottom.y, top.z - bottom.z};
        n = unit3(F3{h.y * w.z - h.z * w.y, h.z * w.x - h.x * w.z, h.x * w.y - h.y * w.x});
    }
    *reinterpret_cast<F3*>(a.pseudo + 3 * pid) = n;
}

hipError_t launch_view_normals(int P, const float* means3D, const float* axis, const float* cam_pos, float* out,
                               hipStream_t stream) {
    hipLaunchKernelGGL(view_normals_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, means3D, axis, cam_pos, out);
    return hipGetLastError();
}

hipError_t launch_normal_maps(int width, int height, const float* normal_rgb, const float* depth, const float* c2w,
                              float fx, float fy, float cx, float cy, float* normal, float* pseudo, hipStream_t stream) {
    NormalMapArgs a;
    a.W = width; a.H = height; a.normal_rgb = normal_rgb; a.depth = depth;
    a.c2w = c2w;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.normal = normal; a.pseudo = pseudo;
    hipLaunchKernelGGL(normal_maps_kernel, dim3(div_up(width, 64), div_up(height, 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_pack_rgba8(const float* color, const float* alpha, uint8_t* out, size_t n_pixels,
                             hipStream_t stream) {
    const bool vec_ok = n_pixels % 4 == 0 && ((uintptr_t)color % 16 == 0) && ((uintptr_t)alpha % 16 == 0) &&
                        ((uintptr_t)out % 4 == 0);
    const size_t threads = (n_pixels + 3) / 4;
    hipLaunchKernelGGL(pack_rgba8_kernel, dim3((unsigned)((threads + 255) / 256), 4), dim3(256), 0, stream, color,
                       alpha, out, n_pixels, vec_ok ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_blend(const Camera& cam, int variant, int lds_pad_bytes, const uint2* ranges,
                        const uint32_t* point_list, const SplatRaster* raster, const float* features,
                        const float* background, float* out_color, float* out_depth, float* out_alpha,
                        uint32_t* n_contrib, hipStream_t stream, const float* extra_features, float* out_extra) {
    const int T = cam.grid_x * cam.grid_y;
    if (extra_features != nullptr) {  // two feature sets in one walk: the quadrant kernel only
        hipLaunchKernelGGL(blend_quadrant_kernel<true>, dim3(4 * T), dim3(64), (size_t)lds_pad_bytes, stream, cam.width,
                           cam.height, cam.grid_x, T, ranges, point_list, raster, features, extra_features, background,
                           out_color, out_depth, out_alpha, out_extra, n_contrib);
        return hipGetLastError();
    }
    if (variant == 1) {
        // lds_pad_bytes of unused dynamic LDS cap how many single-wave workgroups share a CU, which leaves
        // wave slots free for the memory-bound kernels of another frame running on a second stream
        hipLaunchKernelGGL(blend_quadrant_kernel<false>, dim3(4 * T), dim3(64), (size_t)lds_pad_bytes, stream, cam.width, cam.height, cam.grid_x,
                           T, ranges, point_list, raster, features, (const float*)nullptr, background, out_color,
                           out_depth, out_alpha, (float*)nullptr, n_contrib);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(blend_kernel, dim3(T), dim3(64), 0, stream, cam.width, cam.height, cam.grid_x, T, ranges,
                       point_list, raster, features, background, out_color, out_depth,
                       out_alpha, n_contrib);
    return hipGetLastError();
}

} // namespace gsr
