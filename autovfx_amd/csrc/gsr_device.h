// gsr_device.h -- device-side helpers shared by the two kernel translation units (gsr_kernels.hip: forward and
// auxiliary kernels; gsr_backward.hip: the backward pass, built with different code-generation flags, see
// autovfx_amd/build.py).  Everything here is inline and lives in an unnamed namespace: each unit gets its own copy.
#pragma once
#include "gsr_internal.h"

namespace gsr {
namespace {

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Wait for this wave's outstanding LDS operations only (lgkmcnt = 0; the vector-memory and export counters are left at
// their maxima): the wave-level hand-overs through LDS below must not also wait for global stores still in flight.
#define GSR_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)

// ------------------------------------------------------------------------------------------------
// K5 (identifyTileRanges, rasterizer_impl.cu:116-138): ranges[t] = [lower_bound(t), lower_bound(t + 1)) over the sorted
// tile keys; (0,0) when empty, which is what the reference's memset + boundary scan leaves behind.  One binary search
// per tile (a chain of ~22 dependent loads: that latency is all it costs); the end of a tile's list is the beginning of
// the next tile's, taken from the neighbouring lane through LDS.  Called by every lane of a 256-lane workgroup with
// block < ranges_duty_blocks(num_tiles): the stand-alone kernel, or the first workgroups of a colour kernel, where the
// search latency hides behind the kernel's streaming work.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int kRangesDutyLdsWords = 256 + 2 * 8 * kOrderClasses;   // tile boundaries; (count, base) of the 8 x 8 order cells

__device__ __forceinline__ void tile_ranges_duty(const RangesDuty& d, uint32_t block, uint32_t* s_first /*kRangesDutyLdsWords of LDS*/) {
    if (block == 0 && threadIdx.x < 3 && d.header_dst[threadIdx.x] != nullptr)
        *reinterpret_cast<ArenaHeader*>(d.header_dst[threadIdx.x]) = d.headers[threadIdx.x];
    const int t = (int)block * 255 + (int)threadIdx.x;   // 256 boundaries per workgroup = 255 tiles
    const uint32_t n = d.slab->pairs;
    uint32_t* s_cell = s_first + 256;   // [64] tiles of this block per (XCD, class), then [64] their bases in the table
    const bool ordering = d.order.counts != nullptr;
    if (ordering && threadIdx.x < 2 * 8 * kOrderClasses) s_cell[threadIdx.x] = 0u;
    s_first[threadIdx.x] = lower_bound_u32(d.keys, n, (uint32_t)min(t, d.num_tiles));
    __syncthreads();
    const bool mine = threadIdx.x != 255 && t < d.num_tiles;
    int cell = 0;
    uint32_t rank = 0u;
    if (mine) {
        const uint32_t b = s_first[threadIdx.x], e = s_first[threadIdx.x + 1];
        d.ranges[t] = (e > b) ? make_uint2(b, e) : make_uint2(0u, 0u);
        if (ordering) {   // file the tile under (its XCD, class of list length): BlendOrder
            const int xcd = (t / d.order.strip) & 7;
            const uint32_t len = e > b ? e - b : 0u;
            // classes 0 .. 6 by list length, longest first; the last class is the tiles with NOTHING to blend (a later slab's
            // finished tiles): their workgroups leave at once and are dispatched behind everybody who has work
            const int cls = len == 0u ? kOrderClasses - 1 : (kOrderClasses - 2) - (int)min((uint32_t)(kOrderClasses - 2), len >> d.order.shift);
            cell = xcd * kOrderClasses + cls;
            rank = atomicAdd(&s_cell[cell], 1u);   // LDS: the block's tiles of a cell take ONE place in the global counter
        }
    }
    if (ordering) {   // (workgroup-uniform)
        __syncthreads();
        if (threadIdx.x < 8 * kOrderClasses && s_cell[threadIdx.x] != 0u)
            s_cell[8 * kOrderClasses + threadIdx.x] = atomicAdd(&d.order.counts[threadIdx.x], s_cell[threadIdx.x]);
        __syncthreads();
        if (mine) d.order.table[(size_t)cell * d.order.cap + s_cell[8 * kOrderClasses + cell] + rank] = (uint32_t)t;
    }
    __syncthreads();   // (the caller may reuse the LDS)
}

// 4-byte aligned aggregates: the compiler may still fuse them into dwordx3/x4 accesses, but no
// 16-byte alignment is assumed of caller tensors.
struct __attribute__((aligned(4))) F3 { float x, y, z; };
struct __attribute__((aligned(4))) F4 { float x, y, z, w; };

struct Mat3 { float m[3][3]; };  // m[row][col]

// k = 0,1,2 summed left to right: glm's mat3 * mat3 (type_mat3x3.inl:486-518) in math notation.
__device__ __forceinline__ Mat3 mul3(const Mat3& a, const Mat3& b) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
__device__ __forceinline__ Mat3 transpose3(const Mat3& a) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

// float -> int, round toward zero, saturating, NaN -> 0 (what the GPU conversion does; spelled
// out so host oracle and device agree by construction).
__device__ __forceinline__ int f2i_sat(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

__device__ __forceinline__ float ndc_to_pix(float v, int S) {
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);  // auxiliary.h:41-44 is double math
}

struct TileRect { int x0, y0, x1, y1; };

__device__ __forceinline__ TileRect tile_rect(float px, float py, int radius, int gx, int gy) {
    TileRect r;
    r.x0 = min(gx, max(0, f2i_sat((px - radius) / kTile)));
    r.y0 = min(gy, max(0, f2i_sat((py - radius) / kTile)));
    r.x1 = min(gx, max(0, f2i_sat((px + radius + kTile - 1) / kTile)));
    r.y1 = min(gy, max(0, f2i_sat((py + radius + kTile - 1) / kTile)));
    return r;
}

// SH basis constants (auxiliary.h:22-39).
constexpr float kSH0 = 0.28209479177387814f;
constexpr float kSH1 = 0.4886025119029199f;
constexpr float kSH2_0 = 1.0925484305920792f, kSH2_1 = -1.0925484305920792f, kSH2_2 = 0.31539156525252005f,
                kSH2_3 = -1.0925484305920792f, kSH2_4 = 0.5462742152960396f;
constexpr float kSH3_0 = -0.5900435899266435f, kSH3_1 = 2.890611442640554f, kSH3_2 = -0.4570457994644658f,
                kSH3_3 = 0.3731763325901154f, kSH3_4 = -0.4570457994644658f, kSH3_5 = 1.445305721320277f,
                kSH3_6 = -0.5900435899266435f;

__device__ __forceinline__ F3 ld3(const float* p) { return *reinterpret_cast<const F3*>(p); }
__device__ __forceinline__ F3 axpy(F3 acc, float s, F3 v) {  // acc + s * v, unfused
    return F3{acc.x + s * v.x, acc.y + s * v.y, acc.z + s * v.z};
}
__device__ __forceinline__ F3 axmy(F3 acc, float s, F3 v) {  // acc - s * v, unfused
    return F3{acc.x - s * v.x, acc.y - s * v.y, acc.z - s * v.z};
}

// forward.cu:20-71 up to (and including) the +0.5, before the clamp.  (x, y, z) is the unit view
// direction; coefficient 0 sits at `sh0`, coefficient k >= 1 at `shr + 3 k`; deg already clamped to what M holds.
// One [M,3] block (the reference's `shs`): sh0 = shr = the block.  A model's two tensors (gsr_forward_raw): sh0 = its
// _features_dc row, shr = its _features_rest row - 3 -- what torch.cat((dc, rest), dim=1) would have laid out, unread.
__device__ __forceinline__ F3 sh_unclamped(int deg, float x, float y, float z, const float* sh0, const float* shr) {
    F3 c = ld3(sh0);
    F3 v = F3{kSH0 * c.x, kSH0 * c.y, kSH0 * c.z};
    if (deg > 0) {
        v = axmy(v, kSH1 * y, ld3(shr + 3));
        v = axpy(v, kSH1 * z, ld3(shr + 6));
        v = axmy(v, kSH1 * x, ld3(shr + 9));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            v = axpy(v, kSH2_0 * xy, ld3(shr + 12));
            v = axpy(v, kSH2_1 * yz, ld3(shr + 15));
            v = axpy(v, kSH2_2 * (2.0f * zz - xx - yy), ld3(shr + 18));
            v = axpy(v, kSH2_3 * xz, ld3(shr + 21));
            v = axpy(v, kSH2_4 * (xx - yy), ld3(shr + 24));
            if (deg > 2) {
                v = axpy(v, kSH3_0 * y * (3.0f * xx - yy), ld3(shr + 27));
                v = axpy(v, kSH3_1 * xy * z, ld3(shr + 30));
                v = axpy(v, kSH3_2 * y * (4.0f * zz - xx - yy), ld3(shr + 33));
                v = axpy(v, kSH3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), ld3(shr + 36));
                v = axpy(v, kSH3_4 * x * (4.0f * zz - xx - yy), ld3(shr + 39));
                v = axpy(v, kSH3_5 * z * (xx - yy), ld3(shr + 42));
                v = axpy(v, kSH3_6 * x * (xx - 3.0f * yy), ld3(shr + 45));
            }
        }
    }
    v.x += 0.5f; v.y += 0.5f; v.z += 0.5f;
    return v;
}
__device__ __forceinline__ F3 sh_unclamped(int deg, float x, float y, float z, const float* sh) {
    return sh_unclamped(deg, x, y, z, sh, sh);
}

__device__ __forceinline__ F3 sh_to_rgb(int deg, F3 pos, F3 cam, const float* sh0, const float* shr) {
    float dx = pos.x - cam.x, dy = pos.y - cam.y, dz = pos.z - cam.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const F3 v = sh_unclamped(deg, dx / len, dy / len, dz / len, sh0, shr);
    // glm::max(result, 0.0f) (forward.cu:70) is `(x < y) ? y : x`: a NaN colour stays a NaN (fmaxf would turn it into 0)
    return F3{v.x < 0.0f ? 0.0f : v.x, v.y < 0.0f ? 0.0f : v.y, v.z < 0.0f ? 0.0f : v.z};
}
__device__ __forceinline__ F3 sh_to_rgb(int deg, F3 pos, F3 cam, const float* sh) { return sh_to_rgb(deg, pos, cam, sh, sh); }

// ------------------------------------------------------------------------------------------------
// The reference's per-frame PyTorch preparation, restated per Gaussian (gsr_forward_raw, gsr_place_object,
// gsr_view_normals).  "As PyTorch-ROCm evaluates it" is meant literally: each helper performs the roundings of the
// framework kernels the reference's Python launches on this GPU, in their order, so the activated values -- and with
// them radii, lists and images -- are the bits an unchanged render() hands to the rasterizer.  The orders were
// identified on the device (scripts/experiments/torch_op_probe.py + torch_op_identify.py, profiles/r04_torch_ops.txt):
//   * a reduction over a contiguous last dimension of 4 is split over 4 lanes and combined by shuffles:
//     (x0 + x1) + (x2 + x3); of 3, over 2 lanes: (x0 + x2) + x1  (ATen/native/cuda/Reduce.cuh: block.x = last_pow2(n));
//   * elementwise expressions written as separate Python operators are separate kernels: one rounding each, never fused;
//   * torch.argsort of 3 values is the 32-wide bitonic network (unstable): ties resolve as spelled out in min_axis.
// The library is built with -ffp-contract=off, so what is written here is what runs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float torch_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }  // UnarySpecialOpsKernel.cu: one / (one + exp(-a))

// torch.nn.functional.normalize(q[P,4]) (gaussian_model.py:100-101): q / max(||q||, 1e-12)
__device__ __forceinline__ F4 torch_normalize4(F4 q) {
    float n = sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
    n = n < 1e-12f ? 1e-12f : n;   // clamp_min(norm, eps): a NaN norm stays NaN, as in PyTorch (fmaxf would return eps)
    return F4{q.x / n, q.y / n, q.z / n, q.w / n};
}
// v.norm(dim=1) / torch.sum(v, dim=-1) of a contiguous [P,3]
__device__ __forceinline__ float torch_sum3(float a, float b, float c) { return (a + c) + b; }
__device__ __forceinline__ float torch_norm3(F3 v) { return sqrtf(torch_sum3(v.x * v.x, v.y * v.y, v.z * v.z)); }

// get_minimum_axis(scales, rotations) (utils/general_utils.py:135-141 with build_rotation :78-101): the column of the
// rotation matrix that belongs to the smallest scale.  `q` is the normalised quaternion (build_rotation normalises once
// more, with the sum written out left to right).  Which column on a tie is what the bitonic argsort leaves in front:
//   s0 == s1 < s2 -> 1;  s0 == s2 < s1 -> 0;  s1 == s2 < s0 -> 1;  all equal -> 2.
__device__ __forceinline__ F3 min_axis(F3 s, F4 q) {
    const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x / n, x = q.y / n, y = q.z / n, z = q.w / n;
    int c;
    if (s.x < s.y) c = s.z < s.x ? 2 : 0;
    else if (s.y < s.x) c = s.z < s.y ? 2 : 1;
    else c = s.x < s.z ? 1 : 2;
    if (c == 0) return F3{1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
    if (c == 1) return F3{2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
    return F3{2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
}

// pc.get_normal(dir_pp_normalized) * 0.5 + 0.5 (gaussian_renderer/__init__.py:118-119,169-171; gaussian_model.py
// get_normal; general_utils.py:151-157 flip_align_view): the axis flipped towards the camera, unit length, to [0, 1].
__device__ __forceinline__ F3 view_normal_rgb(F3 p, F3 cam, F3 axis) {
    const F3 d = {p.x - cam.x, p.y - cam.y, p.z - cam.z};
    const float len = torch_norm3(d);
    const F3 dir = {d.x / len, d.y / len, d.z / len};
    const float dot = torch_sum3(axis.x * -dir.x, axis.y * -dir.y, axis.z * -dir.z);
    const float sg = dot >= 0.f ? 1.f : -1.f;
    const F3 m = {axis.x * sg, axis.y * sg, axis.z * sg};
    const float ml = torch_norm3(m);
    return F3{m.x / ml * 0.5f + 0.5f, m.y / ml * 0.5f + 0.5f, m.z / ml * 0.5f + 0.5f};
}

// ------------------------------------------------------------------------------------------------
// Exact-image tile culling.  The reference pairs a splat with every tile of the bounding square of
// its 3-sigma circle; for many of those tiles no pixel can reach alpha >= 1/255, so the blend
// would skip the pair at every pixel.  This predicate returns false only when that is provable:
// with q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy  (power = -q, forward.cu:338), alpha >= 1/255 needs
// q <= ln(255 o); q is convex, so its minimum over the tile's pixel rectangle is 0 if the centre is
// inside and otherwise lies on one of the four edges, where it has a closed form.  The comparison
// keeps a 0.2 % + 1e-3 margin, ~10x the worst-case fp32 rounding of q for |rho| < 0.995
// (error <= 4e-7 (1+|rho|)/(1-|rho|) q); anything less regular is never culled.  Dropping such a
// pair cannot change any pixel, bit for bit (tests compare culled and unculled renders exactly).
// The test runs in the preprocess kernel, where the splat's conic is in registers: a splat whose tight rectangle (below)
// has <= 64 tiles gets a 64-bit mask of live tiles and its pair count becomes popcount(mask), so dead pairs are never
// written, sorted or ranged.  Larger splats get one run of live columns per tile row (row_run, below).
// The same predicate on an 8x8 quadrant lets the quadrant blend skip list entries wholesale.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_q(float A, float B, float C, float dx, float dy) {
    return 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy;
}

// Pixel rectangle [px_first, px_first + w - 1] x [py_first, py_first + h - 1].
// The hardware reciprocal / log2 (1 ulp) are enough here: a slightly misplaced edge minimiser
// changes q only to second order, and the log error (~1e-6) is far inside the 1e-3 margin.
// `skip_below` is the splat's -ln(255 o) - 1e-4 (SplatRaster::skip_below, computed once per Gaussian): the budget
// ln(255 o) is recovered from it; o <= 0 gives budget -inf (nothing visible), NaN keeps the pair.
__device__ __forceinline__ float blend_skip_below(float opacity) { return -logf(255.0f * opacity) - 1.0e-4f; }

__device__ __forceinline__ bool splat_reaches_rect(float4 co, float skip_below, float2 c, int px_first, int py_first, int w,
                                                   int h) {
    const float A = co.x, B = co.y, C = co.z;
    if (!(A > 0.f) || !(C > 0.f) || !((B * B) < 0.99f * (A * C))) return true;
    const float budget = -skip_below - 1.0e-4f;
    const float x_lo = c.x - (float)(px_first + w - 1), x_hi = c.x - (float)px_first;
    const float y_lo = c.y - (float)(py_first + h - 1), y_hi = c.y - (float)py_first;
    if (x_lo <= 0.f && x_hi >= 0.f && y_lo <= 0.f && y_hi >= 0.f) return true;  // centre inside the rectangle
    const float nb_c = -B * __builtin_amdgcn_rcpf(C), nb_a = -B * __builtin_amdgcn_rcpf(A);
    float qmin = edge_q(A, B, C, x_lo, fminf(y_hi, fmaxf(y_lo, nb_c * x_lo)));
    qmin = fminf(qmin, edge_q(A, B, C, x_hi, fminf(y_hi, fmaxf(y_lo, nb_c * x_hi))));
    qmin = fminf(qmin, edge_q(A, B, C, fminf(x_hi, fmaxf(x_lo, nb_a * y_lo)), y_lo));
    qmin = fminf(qmin, edge_q(A, B, C, fminf(x_hi, fmaxf(x_lo, nb_a * y_hi)), y_hi));
    return !(qmin * 0.998f - 1.0e-3f > budget);
}

__device__ __forceinline__ bool splat_reaches_tile(float4 co, float skip_below, float2 c, int tile_x, int tile_y) {
    return splat_reaches_rect(co, skip_below, c, tile_x * kTile, tile_y * kTile, kTile, kTile);
}

// ------------------------------------------------------------------------------------------------
// The same bound as a REGION instead of a per-tile predicate: {d : q(d) <= beta} is an ellipse around the splat's
// centre.  Its bounding box cuts the reference's rectangle (the bounding square of the 3-sigma CIRCLE of the major
// axis: far too large for needles and for faint splats) down to the "tight" rectangle; and for splats whose tight
// rectangle is still too large for a bit mask, the ellipse cut with the pixel rows of one tile row is convex, so the
// live tiles of that row are ONE run of columns with a closed form (the ellipse's extreme point in x if it lies in the
// strip, else the intersection with the nearer strip edge).  beta carries the margin of splat_reaches_rect
// (q * 0.998 - 1e-3 <= budget), the extents another 0.1 % + 0.02 px: a tile is dropped only when that is provable.
// ------------------------------------------------------------------------------------------------
struct LiveRegion {
    float A, B, C;      // conic
    float beta;         // q <= beta is necessary for alpha >= 1/255
    float u_ext, v_ext; // half extents of the ellipse's bounding box (pixels)
    int kind;           // 0: cannot be bounded (irregular conic, NaN): everything stays; 1: the ellipse; 2: nothing is live
};

__device__ __forceinline__ LiveRegion live_region(float A, float B, float C, float skip_below) {
    LiveRegion r;
    r.A = A; r.B = B; r.C = C;
    r.beta = ((-skip_below - 1.0e-4f) + 1.0e-3f) * 1.0021f;
    r.u_ext = r.v_ext = 0.f;
    r.kind = 0;
    if (!(A > 0.f) || !(C > 0.f) || !((B * B) < 0.99f * (A * C)) || !(r.beta == r.beta)) return r;
    if (r.beta < 0.f) { r.kind = 2; return r; }   // alpha < 1/255 even at the centre
    const float det = A * C - B * B;                // > 0.01 A C
    const float k = 2.0f * r.beta / det;
    r.u_ext = sqrtf(k * C) * 1.001f + 0.02f;        // (+inf for an infinite opacity: nothing is cut)
    r.v_ext = sqrtf(k * A) * 1.001f + 0.02f;
    if (!(r.u_ext == r.u_ext) || !(r.v_ext == r.v_ext)) return r;
    r.kind = 1;
    return r;
}

// The reference's tile rectangle [x0, x1) x [y0, y1) cut down to the tiles the region's bounding box overlaps.
__device__ __forceinline__ TileRect tight_rect(const LiveRegion& g, float cx, float cy, TileRect rc) {
    if (g.kind == 0) return rc;
    if (g.kind == 2) { rc.x1 = rc.x0; rc.y1 = rc.y0; return rc; }
    // tile column of a pixel coordinate, clamped in float first (the extents may be infinite)
    const float fx0 = floorf((cx - g.u_ext) * (1.0f / kTile)), fx1 = floorf((cx + g.u_ext) * (1.0f / kTile));
    const float fy0 = floorf((cy - g.v_ext) * (1.0f / kTile)), fy1 = floorf((cy + g.v_ext) * (1.0f / kTile));
    const int nx0 = (int)fminf(fmaxf(fx0, (float)rc.x0), (float)rc.x1);
    const int nx1 = (int)fminf(fmaxf(fx1 + 1.0f, (float)rc.x0), (float)rc.x1);
    const int ny0 = (int)fminf(fmaxf(fy0, (float)rc.y0), (float)rc.y1);
    const int ny1 = (int)fminf(fmaxf(fy1 + 1.0f, (float)rc.y0), (float)rc.y1);
    TileRect t = {nx0, ny0, nx1 > nx0 ? nx1 : nx0, ny1 > ny0 ? ny1 : ny0};
    if (t.x1 == t.x0 || t.y1 == t.y0) { t.x1 = t.x0; t.y1 = t.y0; }
    return t;
}

// Live columns [*ca, *cb) of tile row `ty` inside [x0, x1): empty when ca >= cb.
// kFast: hardware reciprocal / square root (1 ulp) instead of the IEEE sequences -- their error (1e-7 relative) is four
// orders of magnitude inside the 0.1 % + 0.02 px the extents are widened by; used where the run is computed per
// (splat, tile row) for millions of small splats (the projection kernel).
template <bool kFast = false>
__device__ __forceinline__ void row_run(const LiveRegion& g, float cx, float cy, int ty, int x0, int x1, int* ca, int* cb) {
    *ca = x0; *cb = x1;
    if (g.kind == 0) return;
    if (g.kind == 2) { *cb = x0; return; }
    // d = pixel - centre; the strip of this tile row's pixel rows
    const float v0 = (float)(ty * kTile) - cy, v1 = (float)(ty * kTile + kTile - 1) - cy;
    if (v0 > g.v_ext || v1 < -g.v_ext) { *cb = x0; return; }   // the strip misses the ellipse
    const float A = g.A, B = g.B, C = g.C;
    const float inv_a = kFast ? __builtin_amdgcn_rcpf(A) : 1.0f / A;
    const float slope = kFast ? -B * __builtin_amdgcn_rcpf(C) : -B / C;   // the ellipse's extreme points in u sit at v = slope * u
    const float two_a_beta = 2.0f * A * g.beta, det = A * C - B * B;
    float u_hi, u_lo;
    {
        const float ve = slope * g.u_ext;            // right extreme (+u_ext, ve)
        if (ve >= v0 && ve <= v1) u_hi = g.u_ext;
        else {
            const float vb = ve < v0 ? v0 : v1;
            const float disc = two_a_beta - det * vb * vb;
            u_hi = kFast ? (-B * vb + __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f))) * inv_a : (-B * vb + sqrtf(fmaxf(disc, 0.f))) / A;
            u_hi = u_hi + fabsf(u_hi) * 1.0e-3f + 0.02f;
        }
    }
    {
        const float ve = -slope * g.u_ext;           // left extreme (-u_ext, ve)
        if (ve >= v0 && ve <= v1) u_lo = -g.u_ext;
        else {
            const float vb = ve < v0 ? v0 : v1;
            const float disc = two_a_beta - det * vb * vb;
            u_lo = kFast ? (-B * vb - __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f))) * inv_a : (-B * vb - sqrtf(fmaxf(disc, 0.f))) / A;
            u_lo = u_lo - fabsf(u_lo) * 1.0e-3f - 0.02f;
        }
    }
    if (!(u_lo == u_lo) || !(u_hi == u_hi)) return;  // keep the whole row
    const float f0 = floorf((cx + u_lo) * (1.0f / kTile)), f1 = floorf((cx + u_hi) * (1.0f / kTile)) + 1.0f;
    const int a = (int)fminf(fmaxf(f0, (float)x0), (float)x1), b = (int)fminf(fmaxf(f1, (float)x0), (float)x1);
    *ca = a;
    *cb = b > a ? b : a;
}

// One 48-byte LDS record per staged list entry (quadrant kernel): a single address register serves the three
// 16-byte broadcast reads, and (r, g) / (b, z) land in even-aligned register pairs for the packed-fp32 updates.
struct BlendEntry {
    float x, y, cxx, cxy;               // every processed entry
    float cyy, skip_below, opacity, pad;
    float r, g, b, z;                   // contributing entries only
};

__device__ __forceinline__ int xcd_band_tile(int b, int T) {
    // Blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8).  Give each XCD a contiguous
    // band of tile rows so neighbouring tiles, which share splats, hit the same L2.  Bijective for
    // any T.  Placement is a speed hint only; results do not depend on it.
    const int q = T >> 3, r = T & 7;
    const int xcd = b & 7, local = b >> 3;
    return xcd * q + min(xcd, r) + local;
}

// The item (4 * tile + quadrant) workgroup b of a per-quadrant launch works on when its tiles are taken longest list first
// inside each XCD's band (gsr_internal.h: BlendOrder); -1 for the surplus workgroups of the padded grid.  Workgroup b runs
// on XCD b % 8 and is the (b / 8)-th of that XCD: quadrant (b / 8) % 4 of the (b / 32)-th tile in (class, filing order).
__device__ __forceinline__ int ordered_item(const BlendOrder& o, int b) {
    const int xcd = b & 7, local = b >> 3;
    const uint32_t nth = (uint32_t)(local >> 2);
    const uint32_t* __restrict__ cnt = o.counts + xcd * kOrderClasses;
    uint32_t before = 0u;
    int cls = kOrderClasses;
#pragma unroll
    for (int c = 0; c < kOrderClasses; ++c) {   // (eight wave-uniform loads)
        const uint32_t n = cnt[c];
        if (cls == kOrderClasses) {
            if (nth < before + n) cls = c; else before += n;
        }
    }
    if (cls == kOrderClasses) return -1;
    return 4 * (int)o.table[(size_t)(xcd * kOrderClasses + cls) * o.cap + (nth - before)] + (local & 3);
}

// expf for the blend loop: the instruction sequence of the device library's expf (extended-precision
// x * log2(e), round to nearest, v_exp_f32 of the remainder, ldexp) without its two range clamps, which only
// act for x < -103.97 or x > 88.72.  The blend calls it with power <= 0 (or NaN); together with the
// finite-opacity skip threshold the results are bit-identical to expf there (tests/test_parity_gpu.py
// compares all floats of [-103, 0] through gsr_selftest_exp).  4 of 13 VALU instructions saved per call.
__device__ __forceinline__ float exp_nonpositive(float x) {
    const float log2e_hi = __uint_as_float(0x3fb8aa3bu), log2e_lo = __uint_as_float(0x32a5705fu);
    const float t = x * log2e_hi;
    const float r = __builtin_rintf(t);
    float e = __builtin_fmaf(x, log2e_hi, -t);  // low part of the product, exact
    e = __builtin_fmaf(x, log2e_lo, e);
    const float m = __builtin_amdgcn_exp2f((t - r) + e);
    return __builtin_ldexpf(m, (int)r);
}

} // namespace
} // namespace gsr
