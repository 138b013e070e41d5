// gsr_kernels.hip -- hand-written CDNA4 (gfx950, wave64) streaming kernels of the rasterizer: per Gaussian
// (projection, SH colour, visibility) and per pixel (RGBA8 pack, compositor, normal maps).
//
// What each kernel replaces in the reference (DGR = sugar/gaussian_splatting/submodules/
// diff-gaussian-rasterization, under /root/reference):
//   preprocess_kernel   <- preprocessCUDA       DGR/cuda_rasterizer/forward.cu:155-256
//                          (+ in_frustum auxiliary.h:139-164, computeCov3D forward.cu:118-152,
//                           computeCov2D forward.cu:74-113, getRect auxiliary.h:46-56,
//                           ndc2Pix auxiliary.h:41-44, computeColorFromSH forward.cu:20-71)
//   sh_colour_listed_kernel / sh_colour_all_kernel <- computeColorFromSH   forward.cu:20-71, for the splats that reach a list only
//   mark_visible_kernel <- checkFrustum         DGR/cuda_rasterizer/rasterizer_impl.cu:54-66
// The binning stages (duplicateWithKeys, identifyTileRanges) live in gsr_binning.hip, the sort in gsr_radix.hip, the
// blend (renderCUDA) in gsr_blend.hip, the backward pass in gsr_backward.hip; helpers shared by them in gsr_device.h.
//
// Results are the reference's: fp32 in the reference's operation order, built with
// -ffp-contract=off so nothing is fused behind our back.
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
// K1: one lane per Gaussian.  Streaming, HBM-bound: 236 B in (M = 16; 44 B when the colours are deferred) and 52 B
// out per visible Gaussian, 12 B in / 12 B out per culled one.
// ------------------------------------------------------------------------------------------------
// kRaw (gsr_forward_raw): scales / rotations / opacities / shs are the model's raw parameter tensors and are activated
// here, as PyTorch-ROCm would have (gsr_device.h: torch_*): no activated copy of anything is ever written to memory.
template <bool kRaw>
__global__ void __launch_bounds__(256) preprocess_kernel(GaussianInputs in, Camera cam, GeometryArrays out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in_range = i < in.P;  // every lane stays to the end: the tile-mask phase below is wave-cooperative
    GSR_KTRACE(blockIdx.x, 0);

    const float* __restrict__ vm = cam.viewmatrix;
    const float* __restrict__ pm = cam.projmatrix;

    int radius_out = 0;
    uint32_t rect_area = 0;     // the reference's rectangle: what num_rendered counts
    uint32_t live_bound = 0;    // upper bound of the pairs this splat will emit (exact for masked splats)
    uint32_t big_rows = 0;      // tile rows of a splat too large for a mask
    bool violation = false;     // culled although the caller said nothing would be (prefiltered)
    SplatBin bin = {0u, 0u, 0u, 0u};
    uint32_t key = kCulledKey;
    // what the tile-mask phase needs of a candidate (a visible splat whose tight rectangle has <= kMaskTiles tiles and
    // at least 2 x 2 of them)
    bool mask_candidate = false;
    uint32_t cand_rows = 0;
    LiveRegion cand_region = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
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
        violation = in.prefiltered != 0;
    } else {
        const float pw = 1.0f / (hw + 0.0000001f);
        const float ndc_x = hx * pw, ndc_y = hy * pw;

        // ---- 3D covariance (forward.cu:118-152) ----
        float c3[6];
        F3 s = {0.f, 0.f, 0.f};
        F4 q = {0.f, 0.f, 0.f, 0.f};
        if (in.cov3D_precomp != nullptr) {
            const float* c = in.cov3D_precomp + 6 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = c[k];
        } else {
            s = ld3(in.scales + 3 * (size_t)i);
            q = *reinterpret_cast<const F4*>(in.rotations + 4 * (size_t)i);
            if (kRaw) {   // gaussian_model.py:96-97 exp, :100-101 F.normalize
                s = F3{expf(s.x), expf(s.y), expf(s.z)};
                q = torch_normalize4(q);
            }
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
                if (in.colors_precomp == nullptr && !in.defer_colour) {
                    int deg = in.sh_degree < 3 ? in.sh_degree : 3;  // forward.cu:20-71 knows bands 0..3 (D may be 4)
                    if (deg > 2 && in.M < 16) deg = 2;  // never read past M coefficients
                    if (deg > 1 && in.M < 9) deg = 1;
                    if (deg > 0 && in.M < 4) deg = 0;
                    const F3 cp = ld3(cam.cam_pos);
                    const float* sh0 = kRaw ? in.shs + 3 * (size_t)i : in.shs + 3 * (size_t)in.M * i;
                    const float* shr = (kRaw && in.M > 1) ? in.shs_rest + 3 * (size_t)(in.M - 1) * i - 3 : sh0;
                    const F3 col = sh_to_rgb(deg, p, cp, sh0, shr);
                    *reinterpret_cast<F3*>(out.rgb + 3 * (size_t)i) = col;
                }
                if (kRaw && in.view_normals != nullptr)   // render()'s second feature set (gaussian_renderer/__init__.py:169-171)
                    *reinterpret_cast<F3*>(in.view_normals + 3 * (size_t)i) = view_normal_rgb(p, ld3(cam.cam_pos), min_axis(s, q));
                const float opacity = kRaw ? torch_sigmoid(in.opacities[i]) : in.opacities[i];   // gaussian_model.py:125-126
                const float4 conic_o = make_float4(cc * det_inv, -cb * det_inv, ca * det_inv, opacity);
                float4* rec = reinterpret_cast<float4*>(out.raster + i);  // one 32-byte record, two 16-byte stores
                rec[0] = make_float4(px, py, conic_o.x, conic_o.y);
                const float skip_below = blend_skip_below(conic_o.w);
                rec[1] = make_float4(conic_o.z, conic_o.w, vz, skip_below);
                radius_out = irad;
                rect_area = area;
                // exact-image tile culling, part 1: the rectangle cut down to where alpha >= 1/255 is possible at all
                TileRect tr = rc;
                LiveRegion region = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
                if (in.tile_cull) {
                    region = live_region(conic_o.x, conic_o.y, conic_o.z, skip_below);
                    tr = tight_rect(region, px, py, rc);
                }
                const uint32_t tw = (uint32_t)(tr.x1 - tr.x0), th = (uint32_t)(tr.y1 - tr.y0), tarea = tw * th;
                // (irad > 0: duplicateWithKeys emits pairs only `if (radii[idx] > 0)`, rasterizer_impl.cu:84.  Every finite
                // Gaussian that gets here has a radius of at least one; a NaN covariance gives ceil(3 sqrt(NaN)) -> 0 with a
                // one-tile rectangle: counted in num_rendered by the reference's scan, never emitted -- its slot in the
                // reference's key array keeps whatever the allocation held.  Here it emits nothing.)
                if (tarea != 0u && irad > 0) {
                    bin.xy0 = (uint32_t)tr.x0 | ((uint32_t)tr.y0 << 16);
                    bin.wh = tw | (th << 16);
                    if (tarea <= kMaskTiles) {
                        // part 2: one bit per tile, computed below by the whole wave (or all ones when culling is off)
                        const unsigned long long all = tarea == 64u ? ~0ull : (1ull << tarea) - 1ull;
                        bin.lo = (uint32_t)all; bin.hi = (uint32_t)(all >> 32);
                        live_bound = tarea;
                        // A tight rectangle one tile wide (or high) has no dead tile: it is the bounding box of the ellipse
                        // {q <= beta}, so every tile row it spans contains the ellipse's extreme point of that side, and
                        // with a single column that point lies in the row's only tile.  Only rectangles of at least
                        // 2 x 2 tiles have corners to test (a third of C3's splats).
                        if (in.tile_cull && tw >= 2u && th >= 2u && region.kind == 1) {   // (kind 0: nothing can be bounded)
                            mask_candidate = true;
                            cand_rows = th;
                            cand_region = region;
                            cand_centre = make_float2(px, py);
                        }
                    } else {  // too large for a mask: one run of live columns per tile row, worked out at gather time
                        bin.lo = bin.hi = 0xFFFFFFFFu;
                        live_bound = tarea;
                        big_rows = th;
                    }
                    key = __float_as_uint(vz);
                    if (key == kCulledKey) key = kCulledKey - 1u;  // a NaN depth with an all-ones payload must not look culled
                }
            }
        }
    }
    }  // in_range
    GSR_KTRACE(blockIdx.x, 1);

    // ---- exact-image tile culling, wave-cooperative ----
    // The region where a splat can reach alpha >= 1/255 is an ellipse; cut with the pixel rows of one tile row it is
    // convex, so the live tiles of that row are ONE run of columns with a closed form (gsr_device.h: row_run -- what
    // bin_gather_kernel uses for the splats too large for a mask).  The mask of a small splat is therefore built from one
    // run per tile row instead of one test per tile (round 2: 4 - 9 tests for the usual 2x2 ... 3x3 rectangles against 2 - 3
    // runs), and the (splat, row) items of the wave's 64 splats are flattened over its lanes: item t belongs to the splat
    // whose inclusive row count first exceeds t (binary search over the counts parked in LDS), every lane computes one
    // run per iteration and ORs its bits into the splat's mask in LDS.  Iterations = rows of the wave's candidates / 64:
    // one, as a rule.
    {
        __shared__ float4 s_reg0[4][64];    // conic A, B, C, beta
        __shared__ float4 s_reg1[4][64];    // half extents u, v, centre x, y
        __shared__ uint2 s_place[4][64];    // first tile x | y << 16, rectangle width
        __shared__ uint32_t s_incl[4][64];
        __shared__ uint32_t s_mask[4][64][2];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t incl = cand_rows;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        if (total != 0u) {  // wave-uniform
            s_reg0[wave][lane] = make_float4(cand_region.A, cand_region.B, cand_region.C, cand_region.beta);
            s_reg1[wave][lane] = make_float4(cand_region.u_ext, cand_region.v_ext, cand_centre.x, cand_centre.y);
            s_place[wave][lane] = make_uint2(bin.xy0, bin.wh & 0xFFFFu);
            s_incl[wave][lane] = incl;
            s_mask[wave][lane][0] = 0u;
            s_mask[wave][lane][1] = 0u;
            GSR_WAIT_LDS();                      // this wave's own LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                const uint32_t t = t0 + (uint32_t)lane;
                if (t < total) {
                    int lo = 0, hi = 63;  // first splat whose inclusive row count exceeds t
#pragma unroll
                    for (int step = 0; step < 6; ++step) {
                        const int mid = (lo + hi) >> 1;
                        if (s_incl[wave][mid] > t) hi = mid; else lo = mid + 1;
                    }
                    const uint32_t r = t - (lo > 0 ? s_incl[wave][lo - 1] : 0u);
                    const float4 q0 = s_reg0[wave][lo], q1 = s_reg1[wave][lo];
                    const uint2 pl = s_place[wave][lo];
                    LiveRegion g;
                    g.A = q0.x; g.B = q0.y; g.C = q0.z; g.beta = q0.w; g.u_ext = q1.x; g.v_ext = q1.y; g.kind = 1;
                    const uint32_t x0 = pl.x & 0xFFFFu, y0 = pl.x >> 16, w = pl.y;   // w <= 32: at least two rows, at most 64 tiles
                    int ca, cb;
                    row_run<true>(g, q1.z, q1.w, (int)(y0 + r), (int)x0, (int)(x0 + w), &ca, &cb);
                    if (cb > ca) {
                        const uint32_t len = (uint32_t)(cb - ca);
                        const unsigned long long bits = ((len >= 32u ? 0xFFFFFFFFull : (1ull << len) - 1ull) << ((uint32_t)ca - x0)) << (r * w);
                        if ((uint32_t)bits != 0u) atomicOr(&s_mask[wave][lo][0], (uint32_t)bits);
                        if ((uint32_t)(bits >> 32) != 0u) atomicOr(&s_mask[wave][lo][1], (uint32_t)(bits >> 32));
                    }
                }
            }
            GSR_WAIT_LDS();
            __builtin_amdgcn_wave_barrier();
            if (mask_candidate) {
                const unsigned long long mask = ((unsigned long long)s_mask[wave][lane][0] | ((unsigned long long)s_mask[wave][lane][1] << 32)) &
                                                ((unsigned long long)bin.lo | ((unsigned long long)bin.hi << 32));   // (inside the rectangle)
                bin.lo = (uint32_t)mask; bin.hi = (uint32_t)(mask >> 32);
                live_bound = (uint32_t)__popcll(mask);
                if (mask == 0ull) {  // every tile is dead: the splat is listed nowhere (radii and num_rendered still count it)
                    bin.wh = 0u;
                    key = kCulledKey;
                }
            }
        }
    }

    GSR_KTRACE(blockIdx.x, 2);
    // Pair totals, needed on the host before the binning arena can be sized: an upper bound of the live pairs (what
    // gets expanded and sorted; exact but for the splats too large for a mask) in the low word and the reference's
    // num_rendered (sum of rectangle areas, part of its return value) in the high word of one 64-bit sum.  Summed over
    // the wave, then over the workgroup's four waves through LDS, and stored as this workgroup's BlockTally: no atomics and
    // no zero-filled accumulator (the tally duty of the depth sort's first count kernel adds the workgroups up).
    //
    // The tile rows of the large splats are summed as a prefix: the rows before a splat's own, among the large splats of
    // its workgroup, travel in its record (SplatBin::lo), and with the prefix over the workgroups (tally duty) that is the
    // splat's place in the run pool -- bin_gather_kernel used to hand the rows out with one atomic per wave on ONE word
    // (about 12 ns each on this part: a third of that kernel on a cloud of large splats).
    __shared__ unsigned long long s_tot[4];
    __shared__ uint32_t s_vis[4], s_big[4];
    unsigned long long wave_tot = ((unsigned long long)rect_area << 32) | (unsigned long long)live_bound;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_tot += __shfl_xor(wave_tot, d);
    const unsigned long long emitting = __ballot(key != kCulledKey);
    const unsigned long long any_big = __ballot(big_rows != 0u);
    uint32_t big_incl = big_rows;
    if (any_big != 0ull) {   // wave-uniform
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)big_incl, d);
            if ((int)(threadIdx.x & 63) >= d) big_incl += o;
        }
    }
    const unsigned long long any_violation = __ballot(violation);
    if ((threadIdx.x & 63) == 63) {
        const int w = threadIdx.x >> 6;
        s_tot[w] = wave_tot;
        s_vis[w] = (uint32_t)__popcll(emitting);
        s_big[w] = big_incl | (any_violation != 0ull ? 0x80000000u : 0u);
    }
    __syncthreads();
    if (big_rows != 0u) {
        uint32_t before = big_incl - big_rows;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s_big[w] & 0x7FFFFFFFu;
        bin.lo = before;   // (hi stays all ones)
    }
    if (in_range) {
        out.radii[i] = radius_out;
        *reinterpret_cast<uint4*>(out.bins + i) = make_uint4(bin.xy0, bin.wh, bin.lo, bin.hi);
        out.depth_keys[i] = key;
        if (out.listed != nullptr) out.listed[i] = 0u;   // (a 3 MB memset queued beside other frames' blends took ~100 us)
    }
    if (threadIdx.x == 0) {
        BlockTally t;
        t.pair_total = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        t.visible = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
        const uint32_t big = (s_big[0] & 0x7FFFFFFFu) + (s_big[1] & 0x7FFFFFFFu) + (s_big[2] & 0x7FFFFFFFu) + (s_big[3] & 0x7FFFFFFFu);
        t.big_rows = big | ((s_big[0] | s_big[1] | s_big[2] | s_big[3]) & 0x80000000u);
        *reinterpret_cast<uint4*>(out.tallies + blockIdx.x) = *reinterpret_cast<const uint4*>(&t);
    }
    GSR_KTRACE(blockIdx.x, 3);
}

// ------------------------------------------------------------------------------------------------
// SH colours for the splats of one depth slab that reached a list (inference calls: GaussianInputs::defer_colour).
// The reference evaluates the SH of every visible Gaussian (forward.cu:241-247); behind an opaque front most of them
// are never composited, and the 192 bytes of coefficients per Gaussian are the largest read of the frame.  The pair
// expansion marks every Gaussian it lists (`listed[gid]` = slab + 1); this kernel walks the Gaussians in THEIR order and
// evaluates the marked ones, so the records are read as the projection kernel would have read them -- ascending
// addresses, neighbouring lanes sharing lines -- rather than gathered in depth order (measured: the gather moved 495 MB
// per C3 frame for 163 MB of records, every 128-byte line fetched about three times).  Same arithmetic as the in-line
// evaluation (sh_to_rgb), so rgb[] holds the same bits wherever it is read.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sh_colour_listed_kernel(GaussianInputs in, const float* __restrict__ cam_pos,
                                                               const uint8_t* __restrict__ listed, int tag,
                                                               const SlabInfo* __restrict__ slab, float* __restrict__ rgb,
                                                               RangesDuty duty, int duty_blocks) {
    // A wave looks at 256 consecutive Gaussians (4 marks per lane), packs the marked ones into a list in LDS (a scan of
    // the lanes' counts: their order is kept, so addresses still ascend) and evaluates that list with full
    // lanes.  One Gaussian per lane would run the whole evaluation for every wave that holds a single marked Gaussian:
    // at C3's 22 % that was 9.8 M vector instructions per launch for 1.5 M worth of work.
    __shared__ uint32_t s_list[4][256];
    // The slab's tile ranges ride on the first workgroups (gsr_device.h: tile_ranges_duty): their ~8 us of dependent loads
    // disappear behind the kernel's streaming work instead of being a launch of their own between the sort and this one.
    GSR_KTRACE(16384 + 8192 * (tag - 1) + blockIdx.x, 0);
    if ((int)blockIdx.x < duty_blocks) tile_ranges_duty(duty, blockIdx.x, &s_list[0][0]);
    GSR_KTRACE(16384 + 8192 * (tag - 1) + blockIdx.x, 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = (blockIdx.x * 4 + wave) * 256;   // this wave's 256 Gaussians
    if (first >= in.P || slab->pairs == 0u) return;    // (a slab that found every tile finished lists nothing)
    const int i0 = first + 4 * lane;
    uint32_t marks = 0u;   // one byte per Gaussian
    if (i0 + 3 < in.P) marks = *reinterpret_cast<const uint32_t*>(listed + i0);   // (listed is 256-byte aligned, i0 a multiple of 4)
    else
        for (int j = 0; j < 4; ++j)
            if (i0 + j < in.P) marks |= (uint32_t)listed[i0 + j] << (8 * j);
    // list position of a marked Gaussian = marked Gaussians before it: the lanes' counts scanned over the wave, then the
    // lane's own four in order -- the list is in Gaussian order, so neighbouring lanes read neighbouring records
    bool mine[4];
    uint32_t cnt = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine[j] = ((marks >> (8 * j)) & 0xFFu) == (uint32_t)tag;
        cnt += mine[j] ? 1u : 0u;
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    const uint32_t n = (uint32_t)__shfl((int)incl, 63);
    uint32_t at = incl - cnt;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (mine[j]) s_list[wave][at++] = (uint32_t)(i0 + j);
    GSR_KTRACE(16384 + 8192 * (tag - 1) + blockIdx.x, 2);
    if (n == 0u) return;   // wave-uniform
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int deg = in.sh_degree < 3 ? in.sh_degree : 3;
    if (deg > 2 && in.M < 16) deg = 2;
    if (deg > 1 && in.M < 9) deg = 1;
    if (deg > 0 && in.M < 4) deg = 0;
    const F3 cp = ld3(cam_pos);
    for (uint32_t t = (uint32_t)lane; t < n; t += 64u) {
        const uint32_t i = s_list[wave][t];
        const F3 p = ld3(in.means3D + 3 * (size_t)i);
        const float* sh0 = in.raw ? in.shs + 3 * (size_t)i : in.shs + 3 * (size_t)in.M * i;
        const float* shr = (in.raw && in.M > 1) ? in.shs_rest + 3 * (size_t)(in.M - 1) * i - 3 : sh0;
        const F3 col = sh_to_rgb(deg, p, cp, sh0, shr);
        *reinterpret_cast<F3*>(rgb + 3 * (size_t)i) = col;
    }
    GSR_KTRACE(16384 + 8192 * (tag - 1) + blockIdx.x, 3);
}

// The same for a call that turned out to need no depth slabs (the host learns that only after the projection kernel
// ran without colours): every splat that emits pairs at all, in GAUSSIAN order, so that the 192-byte records are read
// as the projection kernel would have read them (lane-strided, every line used) instead of gathered in depth order.
// (Which splats: those with a radius, i.e. everything the projection kernel did not cull -- a 4-byte flag per Gaussian read
// as a stream; the few splats whose every tile is dead get a colour nobody reads.)
__global__ void __launch_bounds__(256) sh_colour_all_kernel(GaussianInputs in, const float* __restrict__ cam_pos,
                                                            const int* __restrict__ radii, float* __restrict__ rgb,
                                                            RangesDuty duty, int duty_blocks) {
    __shared__ uint32_t s_first[kRangesDutyLdsWords];
    if ((int)blockIdx.x < duty_blocks) tile_ranges_duty(duty, blockIdx.x, s_first);   // (as in sh_colour_listed_kernel)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= in.P || radii[i] <= 0) return;
    int deg = in.sh_degree < 3 ? in.sh_degree : 3;
    if (deg > 2 && in.M < 16) deg = 2;
    if (deg > 1 && in.M < 9) deg = 1;
    if (deg > 0 && in.M < 4) deg = 0;
    const F3 p = ld3(in.means3D + 3 * (size_t)i);
    const float* sh0 = in.raw ? in.shs + 3 * (size_t)i : in.shs + 3 * (size_t)in.M * i;
    const float* shr = (in.raw && in.M > 1) ? in.shs_rest + 3 * (size_t)(in.M - 1) * i - 3 : sh0;
    const F3 col = sh_to_rgb(deg, p, ld3(cam_pos), sh0, shr);
    *reinterpret_cast<F3*>(rgb + 3 * (size_t)i) = col;
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
    if (in.raw) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(div_up(in.P, 256)), dim3(256), 0, stream, in, cam, out);
    else hipLaunchKernelGGL(preprocess_kernel<false>, dim3(div_up(in.P, 256)), dim3(256), 0, stream, in, cam, out);
    return hipGetLastError();
}

// A colour launch carries the ranges duty only when it has at least as many workgroups as the duty needs.
hipError_t launch_sh_colour_listed(const GaussianInputs& in, const Camera& cam, const uint8_t* listed, int tag, const SlabInfo* slab,
                                   float* rgb, const RangesDuty* duty, hipStream_t stream) {
    const int blocks = div_up(in.P, 1024);
    if (in.P <= 0 || (duty != nullptr && blocks < ranges_duty_blocks(duty->num_tiles))) return hipErrorInvalidValue;
    RangesDuty none = {};
    hipLaunchKernelGGL(sh_colour_listed_kernel, dim3(blocks), dim3(256), 0, stream, in, cam.cam_pos, listed, tag, slab, rgb,
                       duty ? *duty : none, duty ? ranges_duty_blocks(duty->num_tiles) : 0);
    return hipGetLastError();
}

hipError_t launch_sh_colour_all(const GaussianInputs& in, const Camera& cam, const int* radii, float* rgb, const RangesDuty* duty,
                                hipStream_t stream) {
    const int blocks = div_up(in.P, 256);
    if (in.P <= 0 || (duty != nullptr && blocks < ranges_duty_blocks(duty->num_tiles))) return hipErrorInvalidValue;
    RangesDuty none = {};
    hipLaunchKernelGGL(sh_colour_all_kernel, dim3(blocks), dim3(256), 0, stream, in, cam.cam_pos, radii, rgb, duty ? *duty : none,
                       duty ? ranges_duty_blocks(duty->num_tiles) : 0);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                               hipStream_t stream) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, means3D, viewmatrix,
                       present);
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
// F.normalize(p=2, eps=1e-12) of a 3-vector.  The order of the three squares is the framework reduction's (gsr_device.h):
// left to right when the vector's components are strided planes (the permuted normal image), (x + z) + y when they are
// contiguous (the cross products).
__device__ __forceinline__ F3 unit3_planes(F3 v) {
    float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    n = n < 1e-12f ? 1e-12f : n;   // clamp_min: a NaN norm stays NaN
    return F3{v.x / n, v.y / n, v.z / n};
}
__device__ __forceinline__ F3 unit3_contiguous(F3 v) {
    float n = torch_norm3(v);
    n = n < 1e-12f ? 1e-12f : n;   // clamp_min: a NaN norm stays NaN
    return F3{v.x / n, v.y / n, v.z / n};
}

__global__ void __launch_bounds__(256) view_normals_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ axis,
                                                           const float* __restrict__ cam_pos, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    *reinterpret_cast<F3*>(out + 3 * (size_t)i) = view_normal_rgb(ld3(means3D + 3 * (size_t)i), ld3(cam_pos), ld3(axis + 3 * (size_t)i));
}

struct NormalMapArgs {
    int W, H;
    const float* normal_rgb;  // [3,H,W]
    const float* depth;       // [H,W]
    const float* c2w;         // device, 16 floats row-major: the 4x4 the Python calls c2w
    float inv_fx, inv_fy, cx, cy;   // inv = (float)(1.0 / (double)f): see the kernel
    float* normal;            // [H,W,3]
    float* pseudo;            // [H,W,3]
};

__global__ void __launch_bounds__(256) normal_maps_kernel(NormalMapArgs a) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= a.W || v >= a.H) return;
    const size_t plane = (size_t)a.W * a.H, pid = (size_t)v * a.W + u;
    const F3 raw = {a.normal_rgb[pid], a.normal_rgb[plane + pid], a.normal_rgb[2 * plane + pid]};
    *reinterpret_cast<F3*>(a.normal + 3 * pid) = unit3_planes(F3{(raw.x - 0.5f) * 2.0f, (raw.y - 0.5f) * 2.0f, (raw.z - 0.5f) * 2.0f});

    F3 n = {0.f, 0.f, 0.f};
    if (u >= 1 && v >= 1 && u < a.W - 1 && v < a.H - 1) {
        const float* __restrict__ m = a.c2w;  // uniform: scalar loads
        // What the framework kernels behind the Python compute, rounding for rounding: a division by a CPU scalar is a
        // multiplication by its reciprocal, formed in double from the float32 intrinsic and rounded once
        // (BinaryDivTrueKernel.cu: static_cast<opmath_t>(1.0 / scalar_value<double>)); `directions @ c2w[:3,:3].T` is a GEMM that
        // accumulates k = 0, 1, 2 with fused multiply-adds (the third direction component is 1); `rays_o + rays_d * depth`
        // is two kernels; torch.cross contracts a_i b_j - a_j b_i into fma(a_i, b_j, -(a_j b_i)).
        const float inv_fx = a.inv_fx, inv_fy = a.inv_fy;
        auto point = [&](int x, int y) {  // rays_o + rays_d * depth
            const float dx = ((float)x - a.cx + 0.5f) * inv_fx, dy = ((float)y - a.cy + 0.5f) * inv_fy;
            const float z = a.depth[(size_t)y * a.W + x];
            return F3{m[3] + (__builtin_fmaf(dy, m[1], dx * m[0]) + m[2]) * z, m[7] + (__builtin_fmaf(dy, m[5], dx * m[4]) + m[6]) * z,
                      m[11] + (__builtin_fmaf(dy, m[9], dx * m[8]) + m[10]) * z};
        };
        const F3 right = point(u + 1, v), left = point(u - 1, v), top = point(u, v - 1), bottom = point(u, v + 1);
        const F3 h = {right.x - left.x, right.y - left.y, right.z - left.z};
        const F3 w = {top.x - bottom.x, top.y - bottom.y, top.z - bottom.z};
        n = unit3_contiguous(F3{__builtin_fmaf(h.y, w.z, -(h.z * w.y)), __builtin_fmaf(h.z, w.x, -(h.x * w.z)),
                                __builtin_fmaf(h.x, w.y, -(h.y * w.x))});
    }
    *reinterpret_cast<F3*>(a.pseudo + 3 * pid) = n;
}

// ------------------------------------------------------------------------------------------------
// gsr_place_object: the per-frame rigid-body placement of one inserted object (gaussians_utils.py:85-118
// transform_gaussians) fused with the activations the render path applies afterwards (gaussian_model.py:95-128) and
// written straight into the object's slice of the resident scene buffers -- what the reference does with a deep copy of
// the scene, a PLY reload, ~10 PyTorch launches and a concatenation of everything, per frame.  One lane per Gaussian;
// every arithmetic step is a separate fp32 rounding in the reference's order (the library is built without contraction).
// ------------------------------------------------------------------------------------------------
// `subset` (nullable): ascending indices of the object's Gaussians to take -- output j comes from input subset[j] (the melting
// branch of the frame loop, scene_representation.py:409-418: `orig_gaussians._xyz[mask]` ... merged as they are).
// `transform` = 0: no rigid transform at all -- positions are copied, the raw quaternion is only normalised (that branch never
// calls transform_gaussians, so not even the identity's roundings may be applied).
__global__ void __launch_bounds__(256) place_object_kernel(int n, const uint32_t* __restrict__ subset, int transform,
                                                           const float* __restrict__ xyz, const float* __restrict__ rot,
                                                           const float* __restrict__ log_scale, const float* __restrict__ opacity,
                                                           const float* __restrict__ shs, int M, ObjectPlacement pl,
                                                           float* __restrict__ out_xyz, float* __restrict__ out_scales,
                                                           float* __restrict__ out_rot, float* __restrict__ out_opacity,
                                                           float* __restrict__ out_shs, float* __restrict__ out_min_axis) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const size_t src = subset != nullptr ? (size_t)subset[i] : (size_t)i;
        const F3 p = ld3(xyz + 3 * src);
        const F3 ls = ld3(log_scale + 3 * src);
        const F4 b = *reinterpret_cast<const F4*>(rot + 4 * src);
        F3 sc;
        F4 qn;
        if (transform) {
            // gaussians_utils.py:94-96 (scale about the initial centre), :100-102 (rotate about it), :106-107 (translate)
            float v[3] = {p.x, p.y, p.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k] = v[k] - pl.c0[k];
                v[k] = v[k] * pl.s;
                v[k] = v[k] + pl.c0[k];
                v[k] = v[k] - pl.c0[k];
            }
            float w[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) w[j] = (v[0] * pl.R[3 * j + 0] + v[1] * pl.R[3 * j + 1]) + v[2] * pl.R[3 * j + 2];   // new_xyz @ R.T
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                w[k] = w[k] + pl.c0[k];
                w[k] = w[k] + (pl.c[k] - pl.c0[k]);
            }
            *reinterpret_cast<F3*>(out_xyz + 3 * (size_t)i) = F3{w[0], w[1], w[2]};
            // :97 new_scales += log(scaling), then exp (gaussian_model.py:96-97)
            sc = F3{expf(ls.x + pl.log_s), expf(ls.y + pl.log_s), expf(ls.z + pl.log_s)};
            // :103 quaternion_multiply(matrix_to_quaternion(R), q) (rotation_utils.py:113-135), then F.normalize (gaussian_model.py:100-101)
            const float aw = pl.qR[0], ax = pl.qR[1], ay = pl.qR[2], az = pl.qR[3];
            float ow = aw * b.x - ax * b.y - ay * b.z - az * b.w;
            float ox = aw * b.y + ax * b.x + ay * b.w - az * b.z;
            float oy = aw * b.z - ax * b.w + ay * b.x + az * b.y;
            float oz = aw * b.w + ax * b.z - ay * b.y + az * b.x;
            if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }   // standardize_quaternion
            qn = torch_normalize4(F4{ow, ox, oy, oz});
        } else {
            *reinterpret_cast<F3*>(out_xyz + 3 * (size_t)i) = p;
            sc = F3{expf(ls.x), expf(ls.y), expf(ls.z)};
            qn = torch_normalize4(b);
        }
        *reinterpret_cast<F3*>(out_scales + 3 * (size_t)i) = sc;
        *reinterpret_cast<F4*>(out_rot + 4 * (size_t)i) = qn;
        if (out_opacity != nullptr) out_opacity[i] = opacity[src];
        if (out_min_axis != nullptr) {
            // general_utils.py:78-101 build_rotation (normalises once more) and :135-141 get_minimum_axis (gsr_device.h: min_axis)
            const F3 col = min_axis(sc, qn);
            *reinterpret_cast<F3*>(out_min_axis + 3 * (size_t)i) = col;
        }
    }
    if (out_shs != nullptr) {
        const bool vec = ((reinterpret_cast<uintptr_t>(shs) | reinterpret_cast<uintptr_t>(out_shs)) & 15u) == 0;
        if (subset == nullptr) {   // the object's SH block copied as one contiguous run: 3 M floats per Gaussian, 16-byte pieces
            const size_t words = (size_t)n * 3u * (size_t)M;
            for (size_t w4 = (size_t)blockIdx.x * 256u + threadIdx.x; w4 * 4u < words; w4 += (size_t)gridDim.x * 256u) {
                if (w4 * 4u + 3u < words && vec)
                    reinterpret_cast<float4*>(out_shs)[w4] = reinterpret_cast<const float4*>(shs)[w4];
                else
                    for (size_t k = w4 * 4u; k < words && k < w4 * 4u + 4u; ++k) out_shs[k] = shs[k];
            }
        } else if (vec && (3 * M) % 4 == 0) {   // row j <- row subset[j], 16-byte pieces of a row
            const size_t per_row = (size_t)(3 * M / 4), chunks = (size_t)n * per_row;
            for (size_t c = (size_t)blockIdx.x * 256u + threadIdx.x; c < chunks; c += (size_t)gridDim.x * 256u) {
                const size_t row = c / per_row, q = c - row * per_row;
                reinterpret_cast<float4*>(out_shs)[c] = reinterpret_cast<const float4*>(shs)[(size_t)subset[row] * per_row + q];
            }
        } else {
            const size_t per_row = (size_t)(3 * M), words = (size_t)n * per_row;
            for (size_t k = (size_t)blockIdx.x * 256u + threadIdx.x; k < words; k += (size_t)gridDim.x * 256u) {
                const size_t row = k / per_row;
                out_shs[k] = shs[(size_t)subset[row] * per_row + (k - row * per_row)];
            }
        }
    }
}

hipError_t launch_place_object(int n, const uint32_t* subset, bool transform, const float* xyz, const float* rot, const float* log_scale,
                               const float* opacity, const float* shs, int M, const ObjectPlacement& pl, float* out_xyz, float* out_scales,
                               float* out_rot, float* out_opacity, float* out_shs, float* out_min_axis, hipStream_t stream) {
    hipLaunchKernelGGL(place_object_kernel, dim3(div_up(n, 256)), dim3(256), 0, stream, n, subset, transform ? 1 : 0, xyz, rot, log_scale,
                       opacity, shs, M, pl, out_xyz, out_scales, out_rot, out_opacity, out_shs, out_min_axis);
    return hipGetLastError();
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
    a.inv_fx = (float)(1.0 / (double)fx); a.inv_fy = (float)(1.0 / (double)fy); a.cx = cx; a.cy = cy; a.normal = normal; a.pseudo = pseudo;
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


} // namespace gsr
