// gsr_binning.hip -- from the depth order of the Gaussians to per-tile lists, for gfx950.
//
// What it replaces in the reference (DGR = sugar/gaussian_splatting/submodules/diff-gaussian-rasterization):
//   cub::DeviceScan::InclusiveSum over tiles_touched    DGR/cuda_rasterizer/rasterizer_impl.cu:278
//   duplicateWithKeys                                   DGR/cuda_rasterizer/rasterizer_impl.cu:70-111
//   identifyTileRanges                                  DGR/cuda_rasterizer/rasterizer_impl.cu:116-138
// (the sort between the last two is gsr_radix.hip).
//
// The reference expands every splat into one pair per tile of the bounding square of its 3-sigma circle and sorts all
// of them.  Here a pair is only ever written if some pixel can use it:
//   * pairs whose tile cannot reach alpha >= 1/255 are never emitted (exact-image tile culling: bit masks for small
//     tight rectangles from the projection kernel, one run of live columns per tile row for large splats, worked out
//     here in closed form -- gsr_device.h: live_region / row_run);
//   * the depth-sorted splats are expanded, sorted and blended in front-to-back SLABS (inference calls); a slab drops
//     every pair whose tile has all 256 pixels finished by the slabs in front of it (the reference's block-wide early
//     exit, forward.cu:312-314, applied before the pair is written instead of after it was sorted).
// Stages, all spin-free (no workgroup ever waits for another one):
//   bin_gather_kernel   : one workgroup per kDupTile = 1024 positions of the depth order.  The one random gather per
//                         splat (its 16-byte record; for large splats also the conic, to compute their runs -- the tile
//                         rows of a wave's large splats flattened over its lanes), live pair counts, scan inside the
//                         tile, records re-written IN DEPTH ORDER.
//   bin_offsets_kernel  : adds the sum of all earlier tile totals (each workgroup sums them itself: a few KB of
//                         L2-resident words, no chain, no look-back) -> POINT_OFFSETS, global and inclusive.
//                         The workgroup whose tile a slab boundary falls into also writes the slab table.
//   slab_recount_kernel / slab_compact_kernel (slabs > 0): drop finished tiles from the records, re-scan, and list the
//                         positions that still have a live pair (behind an opaque front nine splats in ten have none:
//                         the expansion walks the compacted list, not the depth order).
//   expand_kernel       : one workgroup per kPairTile = 2048 PAIRS (fewer when few are left), 8 consecutive pairs per lane: every workgroup
//                         does the same work whatever the splat sizes, and writes one contiguous 16 KB slice of the
//                         two pair arrays with 16-byte stores.
//   tile_ranges_kernel  : one binary search per tile over the sorted tile keys (a list ends where the next tile's begins).
// Every pair count past the first host read-back lives in device memory (SlabInfo::pairs): launches are sized for an
// upper bound and surplus workgroups leave at once.
#include "gsr_device.h"

// Profiling aid (python -m autovfx_amd.build --trace, scripts/kernel_trace.py --binning): lane 0 of a workgroup stamps the
// 100 MHz wall clock into slot `slot` of record `id`.  Compiled out of the normal library.
#ifdef GSR_KERNEL_TRACE
__device__ unsigned long long* g_binning_trace = nullptr;
extern "C" __attribute__((visibility("default"))) int gsr_debug_set_binning_trace(void* device_words) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_binning_trace), &device_words, sizeof device_words);
}
#define GSR_BTRACE(id, slot) do { if (threadIdx.x == 0 && g_binning_trace) g_binning_trace[(size_t)(id) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define GSR_BTRACE(id, slot) do { } while (0)
#endif

namespace gsr {
namespace {

constexpr int kPairTile = kExpandTile;
constexpr int kPairsPerLane = kPairTile / 256;  // consecutive pairs of one lane
constexpr int kMaxDoneWords = 4096;             // most tile bit rows the slab kernels hold in LDS (16 KB; an 8K x 8K image has 1 024)
static_assert(kDupTile == 1024, "256 lanes x 4 consecutive positions");

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

// Inclusive scan of one value per lane over the 256 lanes; returns the exclusive prefix, *total = sum of all.
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t mine, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += s_wave[w];
        all += s_wave[w];
    }
    __syncthreads();
    *total = all;
    return before;
}

__device__ __forceinline__ bool rec_is_masked(uint32_t wh) { return (wh & 0xFFFFu) * (wh >> 16) <= kMaskTiles; }

// Bits [x0, x0 + w) of tile bit row `ty` (w <= 64), right-aligned.
__device__ __forceinline__ unsigned long long done_bits(const uint32_t* rows, int row_words, uint32_t ty, uint32_t x0, uint32_t w) {
    const uint32_t* r = rows + ty * (uint32_t)row_words;
    const uint32_t wi = x0 >> 5, sh = x0 & 31u;
    const uint32_t last = (uint32_t)row_words - 1u;
    const unsigned long long a = (unsigned long long)r[min(wi, last)] | ((unsigned long long)(wi + 1 <= last ? r[wi + 1] : 0u) << 32);
    unsigned long long v = a >> sh;
    if (sh + w > 64u) v |= (unsigned long long)(wi + 2 <= last ? r[wi + 2] : 0u) << (64u - sh);
    return w >= 64u ? v : v & ((1ull << w) - 1ull);
}

// The finished tiles of a masked splat's rectangle, in the bit order of its mask (row-major).
__device__ __forceinline__ unsigned long long done_rect(const uint32_t* rows, int row_words, uint32_t xy0, uint32_t wh) {
    const uint32_t x0 = xy0 & 0xFFFFu, y0 = xy0 >> 16, w = wh & 0xFFFFu, h = wh >> 16;
    unsigned long long d = 0ull;
    for (uint32_t j = 0; j < h; ++j) d |= done_bits(rows, row_words, y0 + j, x0, w) << (j * w);
    return d;
}

// Walks the live tiles of one splat in row-major order: the set bits of a mask, or the runs of a large splat
// (minus the tiles that are already finished, when a table of those is given).
struct TileWalker {
    uint32_t x0, y0, width;
    bool masked;
    // masked
    unsigned long long mask;
    float inv_width;
    // runs
    const uint32_t* runs;   // this splat's rows in the pool, or nullptr: every row is the full width
    const uint32_t* incl;   // ... and the live tiles of rows 0..i, inclusive
    const uint32_t* done;   // tile bit rows (LDS), or nullptr
    int row_words;
    uint32_t height, row, ca, cb, col_base, cur;

    __device__ __forceinline__ void load_run() {
        if (runs != nullptr) {
            const uint32_t r = runs[row];
            ca = r & 0xFFFFu; cb = r >> 16;
        } else {
            ca = x0; cb = x0 + width;
        }
    }
    __device__ __forceinline__ void load_word() {  // live, unfinished tiles among columns [col_base, col_base + 32) of `row`
        const uint32_t lo = max(ca, col_base), hi = min(cb, col_base + 32u);
        uint32_t bits = 0u;
        if (lo < hi) {
            const uint32_t n = hi - lo;
            bits = (n >= 32u ? ~0u : ((1u << n) - 1u)) << (lo - col_base);
        }
        if (done != nullptr) bits &= ~done[(y0 + row) * (uint32_t)row_words + (col_base >> 5)];
        cur = bits;
    }
    __device__ __forceinline__ void next_word() {
        col_base += 32u;
        while (col_base >= cb) {  // this row is exhausted (or its run is empty)
            if (++row >= height) { cur = 0u; return; }
            load_run();
            col_base = ca < cb ? (ca & ~31u) : cb;  // empty run: straight to the next row
        }
        load_word();
    }
    __device__ __forceinline__ void init(const uint4 rec, const uint32_t* pool, const uint32_t* pool_incl, const uint32_t* done_rows_,
                                         int row_words_) {
        x0 = rec.x & 0xFFFFu; y0 = rec.x >> 16; width = rec.y & 0xFFFFu; height = rec.y >> 16;
        masked = width * height <= kMaskTiles;
        done = done_rows_; row_words = row_words_;
        runs = (!masked && rec.z != 0xFFFFFFFFu) ? pool + rec.z : nullptr;
        incl = runs != nullptr ? pool_incl + rec.z : nullptr;
        mask = (unsigned long long)rec.z | ((unsigned long long)rec.w << 32);
        // later slabs: the tiles finished by the slabs in front leave the mask HERE (round 6).  slab_recount_kernel used to write the
        // reduced mask back into the record -- 75 MB of read-modify-write per C3 frame for a few bit operations on a table that
        // is resident in LDS anyway; the records now stay as bin_gather_kernel wrote them.
        if (masked && done != nullptr) mask &= ~done_rect(done, row_words, rec.x, rec.y);
    }
    // position on the r-th (0-based) live tile
    __device__ __forceinline__ void start(const uint4 rec, uint32_t r, const uint32_t* pool, const uint32_t* pool_incl,
                                          const uint32_t* done_rows_, int row_words_) {
        init(rec, pool, pool_incl, done_rows_, row_words_);
        if (masked) {
            for (uint32_t i = 0; i < r; ++i) mask &= mask - 1ull;  // r < 64
            inv_width = __builtin_amdgcn_rcpf((float)width);
        } else if (done == nullptr) {
            // nothing is finished yet (the first slab, full calls): the r-th live tile is found directly -- by division in
            // a full rectangle, by a binary search over the rows' inclusive counts (bin_gather_kernel) otherwise -- where a
            // walk from the first row would make every lane of a splat with thousands of tiles walk most of its rows
            uint32_t rem;
            if (runs == nullptr) {
                row = r / width;
                rem = r - row * width;
            } else {
                uint32_t lo = 0u, hi = height - 1u;   // first row whose inclusive count exceeds r
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (incl[mid] > r) hi = mid; else lo = mid + 1u;
                }
                row = lo;
                rem = r - (row > 0u ? incl[row - 1u] : 0u);
            }
            load_run();
            const uint32_t col = ca + rem;
            col_base = col & ~31u;
            load_word();
            cur &= ~((1u << (col - col_base)) - 1u);
        } else {
            row = 0;
            load_run();
            if (ca < cb) { col_base = ca & ~31u; load_word(); } else { col_base = cb; cur = 0u; }
            for (;;) {  // skip r live tiles, a word at a time
                while (cur == 0u && row < height) next_word();
                if (row >= height) break;
                const uint32_t c = (uint32_t)__popc(cur);
                if (r < c) break;
                r -= c;
                cur = 0u;
            }
            for (uint32_t i = 0; i < r; ++i) cur &= cur - 1u;
        }
    }
    __device__ __forceinline__ uint32_t next(uint32_t grid_x) {
        if (masked) {
            const uint32_t pos = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            // pos < 64, width <= 64: (pos + 0.5) / width is at least 0.5 / 64 away from an integer, the
            // approximate reciprocal is off by parts in 2^22
            const uint32_t r_ = (uint32_t)(((float)pos + 0.5f) * inv_width);
            return (y0 + r_) * grid_x + x0 + (pos - r_ * width);
        }
        while (cur == 0u && row < height) next_word();
        const uint32_t pos = (uint32_t)__builtin_ctz(cur);
        cur &= cur - 1u;
        return (y0 + row) * grid_x + col_base + pos;
    }
};

// ------------------------------------------------------------------------------------------------
// bin_gather: positions [0, V) of the depth order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 7) bin_gather_kernel(BinningArrays a) {
    __shared__ uint32_t s_wave[4];
    // per wave: the large splats of one round, flattened to (splat, tile row) items (see below)
    __shared__ float4 s_reg0[4][64];     // conic A, B, C, beta
    __shared__ float4 s_reg1[4][64];     // half extents u, v, centre x, y
    __shared__ uint4 s_geo[4][64];       // first tile x | y << 16, width | kind << 16, first row in the pool
    __shared__ uint32_t s_rows[4][64];   // inclusive row counts
    __shared__ uint32_t s_live[4][64];   // live tiles, summed over the rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int V = a.V;
    const int k0 = (int)blockIdx.x * kDupTile + 4 * tid;  // 4 consecutive positions per lane
    GSR_BTRACE(blockIdx.x, 0);
    uint32_t gid[4] = {0u, 0u, 0u, 0u};
    if (k0 + 3 < V) {
        const uint4 g = *reinterpret_cast<const uint4*>(a.depth_order + k0);
        gid[0] = g.x; gid[1] = g.y; gid[2] = g.z; gid[3] = g.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V) gid[j] = a.depth_order[k0 + j];
    }
    uint4 rec[4];
    uint32_t count[4];
    bool any_rows = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rec[j] = make_uint4(0u, 0u, 0u, 0u);
        count[j] = 0u;
        if (k0 + j < V) {
            rec[j] = *reinterpret_cast<const uint4*>(a.bins + gid[j]);  // the one random gather per splat
            if (rec_is_masked(rec[j].y)) count[j] = (uint32_t)__popc(rec[j].z) + (uint32_t)__popc(rec[j].w);
            else if (a.tile_cull) {
                any_rows = true;   // z: rows of the large splats before it in its projection workgroup -> its first row in the pool
                rec[j].z += a.pool_first[gid[j] >> 8] + a.pool_group_first[gid[j] >> a.pool_group_shift];
            }
        }
    }
    // Large splats: one run of live columns per tile row, parked in the pool.  Where a splat's rows start is a prefix sum
    // over the Gaussians in THEIR order -- inside the projection workgroup (the record's z), over the workgroups of a tally
    // group (pool_first), over the groups (the host, which has read their totals back to size the pool) -- so nothing is
    // handed out here.  (Rounds 3 - 4: one atomic per wave on one word; at about 12 ns per same-address atomic that was a
    // third of this kernel on a cloud of large splats.)  A splat whose rows would not fit (cannot happen: the pool is sized
    // from the same sums) keeps its full rectangle, which is only less culled, never wrong.
    const unsigned long long any_big = __ballot(any_rows);
    GSR_BTRACE(blockIdx.x, 1);
    if (any_big != 0ull) {   // wave-uniform
        // A lane walking the tile rows of its own splats makes the wave run as long as its tallest splat (up to the
        // whole image height) while the other lanes wait.  Instead the (splat, tile row) items of the wave are
        // flattened, one of the lane's four positions at a time: item t belongs to the splat whose inclusive row count
        // first exceeds t (binary search in LDS), every lane computes one run per iteration, the run lengths are summed
        // per splat with LDS atomics.  Iterations = rows of the wave's large splats / 64.
        const int wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool large = k0 + j < V && !rec_is_masked(rec[j].y);
            const uint32_t h = large ? rec[j].y >> 16 : 0u;
            const uint32_t pool_at = rec[j].z;
            bool walk = false;
            if (large) {
                if (pool_at > a.pool_rows || h > a.pool_rows - pool_at) {
                    rec[j].z = 0xFFFFFFFFu;  // every row is the full width
                    count[j] = (rec[j].y & 0xFFFFu) * h;
                    rec[j].w = count[j];
                } else {
                    walk = true;
                }
            }
            const uint32_t rows = walk ? h : 0u;
            uint32_t rincl = rows;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)rincl, d);
                if (lane >= d) rincl += o;
            }
            const uint32_t total = (uint32_t)__shfl((int)rincl, 63);
            if (total != 0u) {   // wave-uniform
                if (walk) {
                    const float4* rr = reinterpret_cast<const float4*>(a.raster + gid[j]);
                    const float4 r0 = rr[0], r1 = rr[1];  // x y cxx cxy | cyy opacity depth skip_below
                    const LiveRegion g = live_region(r0.z, r0.w, r1.x, r1.w);
                    s_reg0[wave][lane] = make_float4(g.A, g.B, g.C, g.beta);
                    s_reg1[wave][lane] = make_float4(g.u_ext, g.v_ext, r0.x, r0.y);
                    s_geo[wave][lane] = make_uint4(rec[j].x, (rec[j].y & 0xFFFFu) | ((uint32_t)g.kind << 16), pool_at, 0u);
                }
                s_rows[wave][lane] = rincl;
                s_live[wave][lane] = 0u;
                GSR_WAIT_LDS();                      // this wave's own LDS writes have landed
                __builtin_amdgcn_wave_barrier();
                for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                    const uint32_t t = t0 + (uint32_t)lane;
                    int lo = 64;            // (no item: a segment of its own)
                    uint32_t len = 0u, before = 0u, pool_row = 0u, run = 0u;
                    if (t < total) {
                        lo = 0;
                        int hi = 63;  // first splat whose inclusive row count exceeds t
#pragma unroll
                        for (int step = 0; step < 6; ++step) {
                            const int mid = (lo + hi) >> 1;
                            if (s_rows[wave][mid] > t) hi = mid; else lo = mid + 1;
                        }
                        const uint32_t r = t - (lo > 0 ? s_rows[wave][lo - 1] : 0u);
                        const float4 q0 = s_reg0[wave][lo], q1 = s_reg1[wave][lo];
                        const uint4 ge = s_geo[wave][lo];
                        LiveRegion g;
                        g.A = q0.x; g.B = q0.y; g.C = q0.z; g.beta = q0.w; g.u_ext = q1.x; g.v_ext = q1.y; g.kind = (int)(ge.y >> 16);
                        const uint32_t x0 = ge.x & 0xFFFFu, y0 = ge.x >> 16, w = ge.y & 0xFFFFu;
                        int ca, cb;
                        row_run(g, q1.z, q1.w, (int)(y0 + r), (int)x0, (int)(x0 + w), &ca, &cb);
                        run = (uint32_t)ca | ((uint32_t)cb << 16);
                        len = (uint32_t)(cb - ca);
                        pool_row = ge.z + r;
                        before = s_live[wave][lo];   // live tiles of the splat's rows handled by earlier iterations
                    }
                    // inclusive live-tile count through this row: the items of a splat are consecutive lanes, rows ascending
                    uint32_t incl_len = len;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o_len = (uint32_t)__shfl_up((int)incl_len, d);
                        const int o_seg = __shfl_up(lo, d);
                        if (lane >= d && o_seg == lo) incl_len += o_len;
                    }
                    GSR_WAIT_LDS();                      // every lane has read `before` ...
                    __builtin_amdgcn_wave_barrier();     // ... before anybody adds to it
                    if (t < total) {
                        a.run_pool[pool_row] = run;
                        a.run_incl[pool_row] = before + incl_len;
                        if (len != 0u) atomicAdd(&s_live[wave][lo], len);
                    }
                    GSR_WAIT_LDS();                      // (not for the two pool stores: nobody in this kernel reads them)
                    __builtin_amdgcn_wave_barrier();
                }
                GSR_WAIT_LDS();
                __builtin_amdgcn_wave_barrier();
                if (walk) {
                    count[j] = s_live[wave][lane];
                    rec[j].z = pool_at;
                    rec[j].w = count[j];
                }
                __builtin_amdgcn_wave_barrier();     // (the next round overwrites the tables)
            }
        }
    }
    if (!a.tile_cull) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V && !rec_is_masked(rec[j].y)) {
                rec[j].z = 0xFFFFFFFFu;  // every row is the full width
                count[j] = (rec[j].y & 0xFFFFu) * (rec[j].y >> 16);
                rec[j].w = count[j];
            }
    }
    GSR_BTRACE(blockIdx.x, 2);
    const uint32_t mine = count[0] + count[1] + count[2] + count[3];
    uint32_t tile_total;
    const uint32_t before = block_exclusive_256(mine, s_wave, &tile_total);
    GSR_BTRACE(blockIdx.x, 3);
    if (tid == 0) a.tile_totals[blockIdx.x] = tile_total;
    const uint32_t o0 = before + count[0], o1 = o0 + count[1], o2 = o1 + count[2], o3 = o2 + count[3];
    if (k0 + 3 < V) {
        *reinterpret_cast<uint4*>(a.offsets + k0) = make_uint4(o0, o1, o2, o3);
    } else {
        const uint32_t o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V) a.offsets[k0 + j] = o[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k0 + j < V) a.sorted_bins[k0 + j] = rec[j];
    GSR_BTRACE(blockIdx.x, 4);
}

// P - V trailing positions (culled Gaussians) repeat the total so that POINT_OFFSETS is defined over all P, as
// the reference's inclusive scan is.
//
// The same launch cuts the depth order into slabs (round 2 had a kernel of its own for it): slab s takes the positions
// whose inclusive offset lies in (cut[s-1], cut[s]]; its end is the number of positions with an offset <= cut[s], and
// that boundary falls into exactly one 1024-position tile -- the first whose last offset exceeds the cut (the last tile
// when no offset does) -- whose workgroup counts how many of its own positions stay below the cut and writes the slab
// table entries.  The last slab ends at V.
struct SlabCuts { uint32_t cut[kMaxSlabs]; };

__global__ void __launch_bounds__(256) bin_offsets_kernel(int P, int V, const uint32_t* __restrict__ tile_totals,
                                                          uint32_t* __restrict__ offsets, uint32_t* __restrict__ tile_ends,
                                                          int num_slabs, SlabCuts cuts, SlabInfo* __restrict__ slabs,
                                                          SlabInfo* __restrict__ slabs_host) {
    __shared__ uint32_t s_scratch[4];
    const int tiles_v = (V + kDupTile - 1) / kDupTile;
    const int upto = min((int)blockIdx.x, tiles_v);
    uint32_t part = 0;
    for (int t = threadIdx.x; t < upto; t += 256) part += tile_totals[t];
    const uint32_t before = block_sum_256(part, s_scratch);
    const bool has_pairs = (int)blockIdx.x < tiles_v;
    const uint32_t tile_end = has_pairs ? before + tile_totals[blockIdx.x] : before;
    if (threadIdx.x == 0 && has_pairs) tile_ends[blockIdx.x] = tile_end;
    // positions past V (culled Gaussians) carry the grand total: the tile that straddles V is the last one with pairs
    const int k0 = (int)blockIdx.x * kDupTile + 4 * (int)threadIdx.x;
    uint32_t mine[4] = {0u, 0u, 0u, 0u};   // global inclusive offsets of this lane's positions below V
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k < V) {
            mine[j] = offsets[k] + before;
            offsets[k] = mine[j];
        } else if (k < P) {
            offsets[k] = tile_end;
        }
    }
    if (!has_pairs) return;   // workgroup-uniform
    const bool last_tile = (int)blockIdx.x == tiles_v - 1;
    for (int s = 0; s < num_slabs; ++s) {
        const bool open_end = s == num_slabs - 1;   // the last slab takes everything that is left
        const uint32_t cut = open_end ? 0xFFFFFFFFu : cuts.cut[s];
        const bool inside = !open_end && before <= cut && cut < tile_end;  // the boundary lies among this tile's positions
        const bool beyond = last_tile && (open_end || cut >= tile_end);    // ... or behind the last position of all
        if (!inside && !beyond) continue;   // workgroup-uniform
        uint32_t n_le = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V && (beyond || mine[j] <= cut)) ++n_le;
        const uint32_t below = block_sum_256(n_le, s_scratch);   // this tile's positions that stay in slab s (a prefix of them)
        const uint32_t pos = (uint32_t)blockIdx.x * kDupTile + below;
        if (threadIdx.x == 0) {
            slabs[s].end = pos;
            if (s + 1 < num_slabs) slabs[s + 1].first = pos;
            if (s == 0) slabs[0].emitters = pos;
            if (s == 0 && below == 0u) {   // the slab ends with the previous tile
                slabs[0].pairs = before;
                if (slabs_host != nullptr) slabs_host[0].pairs = before;
            }
        }
        // slab 0 is expanded from the global offsets as they are: its pair count is the offset of its last position
        if (s == 0 && below != 0u && (uint32_t)threadIdx.x == (below - 1u) / 4u) {
            slabs[0].pairs = mine[(below - 1u) & 3u];
            if (slabs_host != nullptr) slabs_host[0].pairs = mine[(below - 1u) & 3u];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// slabs > 0: drop the tiles finished by the slabs in front (bit rows written by the blend), count again, scan
// inside 1024-position tiles that start at the slab's first position.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_done_rows(uint32_t* s_done, const uint32_t* __restrict__ done_rows, int words) {
    for (int i = threadIdx.x; i < words; i += 256) s_done[i] = done_rows[i];
    __syncthreads();
}

__global__ void __launch_bounds__(256, 7) slab_recount_kernel(BinningArrays a, int slab) {
    extern __shared__ uint32_t s_done[];   // grid_y * row_words words (sized at the launch; none for the first slab's expansion)
    __shared__ uint32_t s_wave[4];
    __shared__ uint4 s_geo[4][64];       // the large splats of one round: their records
    __shared__ uint32_t s_rows[4][64];   // inclusive row counts
    __shared__ uint32_t s_live[4][64];   // live unfinished tiles, summed over the rows
    GSR_BTRACE(8192 + blockIdx.x, 0);
    const SlabInfo info = a.slabs[slab];
    const uint32_t first = info.first, end = min(info.end, (uint32_t)a.V);
    const uint32_t k0 = first + blockIdx.x * (uint32_t)kDupTile + 4u * threadIdx.x;
    if (first + blockIdx.x * (uint32_t)kDupTile >= end) return;  // workgroup-uniform
    GSR_BTRACE(8192 + blockIdx.x, 1);
    load_done_rows(s_done, a.done_rows, a.grid_y * a.row_words);
    // Every tile finished already (a scene with an opaque front: the first slab is often all it takes)?  Then no splat of
    // this slab has a live pair and the records need not even be read: the counts below stay zero.
    uint32_t done_tiles = 0;
    for (int i = threadIdx.x; i < a.grid_y * a.row_words; i += 256) {
        const int word = i % a.row_words;
        const uint32_t cols = (uint32_t)min(32, a.grid_x - 32 * word);
        done_tiles += (uint32_t)__popc(s_done[i] & (cols >= 32u ? ~0u : (1u << cols) - 1u));
    }
    const bool nothing_left = block_sum_256(done_tiles, s_wave) == (uint32_t)(a.grid_x * a.grid_y);
    GSR_BTRACE(8192 + blockIdx.x, 2);
    uint32_t count[4];
    uint4 rec[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        count[j] = 0u;
        rec[j] = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t k = k0 + j;
        if (k >= end || nothing_left) continue;
        rec[j] = a.sorted_bins[k];
        if (rec[j].y == 0u || !rec_is_masked(rec[j].y)) continue;
        const unsigned long long m = (unsigned long long)rec[j].z | ((unsigned long long)rec[j].w << 32);
        const unsigned long long left = m & ~done_rect(s_done, a.row_words, rec[j].x, rec[j].y);
        count[j] = (uint32_t)__popcll(left);        // (counted, not written back: the expansion's walker subtracts the finished tiles itself)
    }
    GSR_BTRACE(8192 + blockIdx.x, 3);
    // Large splats: the live, unfinished tiles of every tile row, the rows of the wave's splats flattened over its lanes
    // exactly as bin_gather_kernel flattens them (one of the lane's four positions at a time).
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool large = k0 + j < end && rec[j].y != 0u && !rec_is_masked(rec[j].y);
            const uint32_t rows = large ? rec[j].y >> 16 : 0u;
            uint32_t rincl = rows;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)rincl, d);
                if (lane >= d) rincl += o;
            }
            const uint32_t total = (uint32_t)__shfl((int)rincl, 63);
            if (total == 0u) continue;   // wave-uniform
            s_geo[wave][lane] = rec[j];
            s_rows[wave][lane] = rincl;
            s_live[wave][lane] = 0u;
            GSR_WAIT_LDS();
            __builtin_amdgcn_wave_barrier();
            for (uint32_t t0 = 0; t0 < total; t0 += 64u) {
                const uint32_t t = t0 + (uint32_t)lane;
                if (t < total) {
                    int lo = 0, hi = 63;
#pragma unroll
                    for (int step = 0; step < 6; ++step) {
                        const int mid = (lo + hi) >> 1;
                        if (s_rows[wave][mid] > t) hi = mid; else lo = mid + 1;
                    }
                    const uint32_t r = t - (lo > 0 ? s_rows[wave][lo - 1] : 0u);
                    const uint4 ge = s_geo[wave][lo];
                    const uint32_t x0 = ge.x & 0xFFFFu, y0 = ge.x >> 16, w = ge.y & 0xFFFFu;
                    uint32_t ca = x0, cb = x0 + w;
                    if (ge.z != 0xFFFFFFFFu) {
                        const uint32_t run = a.run_pool[ge.z + r];
                        ca = run & 0xFFFFu; cb = run >> 16;
                    }
                    uint32_t n = 0u;
                    for (uint32_t col = ca & ~31u; col < cb; col += 32u) {   // live, unfinished tiles of columns [col, col + 32)
                        const uint32_t lo_c = max(ca, col), hi_c = min(cb, col + 32u);
                        const uint32_t len = hi_c - lo_c;
                        uint32_t bits = (len >= 32u ? ~0u : ((1u << len) - 1u)) << (lo_c - col);
                        bits &= ~s_done[(y0 + r) * (uint32_t)a.row_words + (col >> 5)];
                        n += (uint32_t)__popc(bits);
                    }
                    if (n != 0u) atomicAdd(&s_live[wave][lo], n);
                }
            }
            GSR_WAIT_LDS();
            __builtin_amdgcn_wave_barrier();
            if (large) count[j] = s_live[wave][lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
    GSR_BTRACE(8192 + blockIdx.x, 4);
    const uint32_t mine = count[0] + count[1] + count[2] + count[3];
    uint32_t tile_total;
    const uint32_t before = block_exclusive_256(mine, s_wave, &tile_total);
    const uint32_t emitting = (count[0] != 0u) + (count[1] != 0u) + (count[2] != 0u) + (count[3] != 0u);
    const uint32_t tile_emitting = block_sum_256(emitting, s_wave);
    if (threadIdx.x == 0) {
        a.slab_tile_totals[blockIdx.x] = tile_total;
        a.slab_tile_totals[a.tiles_p + blockIdx.x] = tile_emitting;
    }
    uint32_t run = before;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        run += count[j];
        if (k0 + j < end) a.slab_offsets[k0 + j] = run;
    }
    GSR_BTRACE(8192 + blockIdx.x, 5);
}

// Second half of the re-scan: tile-local offsets become slab-wide, and the positions that kept a live pair are listed
// in order (`cpos`: position - slab.first; `coffs`: inclusive pair offset after it) -- the arrays the expansion walks.
__global__ void __launch_bounds__(256) slab_compact_kernel(BinningArrays a, int slab) {
    __shared__ uint32_t s_scratch[4];
    __shared__ uint32_t s_wave[4];
    const SlabInfo info = a.slabs[slab];
    const uint32_t first = info.first, end = min(info.end, (uint32_t)a.V);
    const uint32_t tiles = end > first ? (end - first + kDupTile - 1) / kDupTile : 0u;
    const uint32_t* __restrict__ pair_totals = a.slab_tile_totals;
    const uint32_t* __restrict__ emit_totals = a.slab_tile_totals + a.tiles_p;
    if (blockIdx.x == 0) {  // the slab's counts (also when the slab is empty)
        uint32_t pp = 0, ee = 0;
        for (uint32_t t = threadIdx.x; t < tiles; t += 256) { pp += pair_totals[t]; ee += emit_totals[t]; }
        const uint32_t all_pairs = block_sum_256(pp, s_scratch);
        const uint32_t all_emit = block_sum_256(ee, s_scratch);
        if (threadIdx.x == 0) {
            a.slabs[slab].pairs = all_pairs; a.slabs[slab].emitters = all_emit;
            if (a.slabs_host != nullptr) a.slabs_host[slab].pairs = all_pairs;
        }
    }
    if (blockIdx.x >= tiles) return;
    uint32_t pp = 0, ee = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256) { pp += pair_totals[t]; ee += emit_totals[t]; }
    const uint32_t pairs_before = block_sum_256(pp, s_scratch);
    const uint32_t emit_before = block_sum_256(ee, s_scratch);
    const uint32_t tile_first = first + blockIdx.x * (uint32_t)kDupTile;
    const uint32_t k0 = tile_first + 4u * threadIdx.x;
    uint32_t incl[4], prev = 0u;  // tile-local inclusive offsets of my 4 positions, and of the position before them
    if (threadIdx.x != 0 && k0 - 1u < end) prev = a.slab_offsets[k0 - 1u];
    uint32_t flags = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        incl[j] = k0 + j < end ? a.slab_offsets[k0 + j] : prev;
        if (incl[j] != prev) flags |= 1u << j;
        prev = incl[j];
    }
    uint32_t tile_emit;
    uint32_t rank = emit_before + block_exclusive_256((uint32_t)__popc(flags), s_wave, &tile_emit);  // (syncs: reads above, writes below)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (k0 + j >= end) continue;
        const uint32_t global_incl = incl[j] + pairs_before;
        a.slab_offsets[k0 + j] = global_incl;
        if (flags & (1u << j)) {
            a.slab_cpos[rank] = k0 + j - first;
            a.slab_coffs[rank] = global_incl;
            ++rank;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// expand: pairs [blockIdx.x * 4096, +4096) of one slab.  The splats that own them are items 0 .. n-1 of a list with
// ascending inclusive pair offsets: for slab 0 the positions of the depth order themselves (global POINT_OFFSETS); for
// a later slab the compacted list of the positions that still have a live pair (slab_compact_kernel).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) expand_kernel(BinningArrays a, int slab, uint32_t* __restrict__ tile_keys,
                                                     uint32_t* __restrict__ point_list, uint32_t* __restrict__ first_counts,
                                                     uint32_t count_stride, uint32_t digit_mask) {
    constexpr int kBatch = 2048;            // items whose offsets are parked in LDS at a time
    __shared__ uint32_t s_incl[kBatch + 1]; // s_incl[0] = pairs before the batch's first item
    __shared__ uint32_t s_scratch[4];
    extern __shared__ uint32_t s_done[];   // grid_y * row_words words (sized at the launch; none for the first slab's expansion)
    const int tid = threadIdx.x;
    GSR_BTRACE(16384 + 8192 * slab + blockIdx.x, 0);
    const SlabInfo info = a.slabs[slab];
    const uint32_t num_pairs = info.pairs;
    // The launch was sized for an upper bound of the pairs.  When far fewer are left (a slab behind an opaque front keeps a
    // few per cent), 2048 pairs per workgroup would leave most of the GPU idle behind a handful of long serial walks:
    // the pairs of a lane shrink (8 or 4) until the workgroups of the launch are all needed.  (2048 / 1024 / 4096 pairs
    // per workgroup at C3: 181 / 180 / 187 us of binning per frame.)
    const uint32_t per_lane = expand_pairs_per_lane(gridDim.x, num_pairs);
    const uint32_t pair_tile = 256u * per_lane;
    const uint32_t p_begin = blockIdx.x * pair_tile;
    if (p_begin >= num_pairs) return;
    const uint32_t p_end = min(num_pairs, p_begin + pair_tile);
    const uint32_t first = info.first;
    const bool compacted = slab > 0;
    // item i <-> position first + (compacted ? cpos[i] : i)
    const int V = compacted ? (int)info.emitters : (int)(min(info.end, (uint32_t)a.V) - first);
    const uint32_t* __restrict__ offsets = compacted ? a.slab_coffs : a.offsets + first;
    const uint32_t* __restrict__ cpos = a.slab_cpos;
    const uint4* __restrict__ sorted_bins = a.sorted_bins + first;
    const uint32_t* __restrict__ order = a.depth_order + first;
    const uint32_t* done = nullptr;
    if (slab > 0) {
        load_done_rows(s_done, a.done_rows, a.grid_y * a.row_words);
        done = s_done;
    }
    const uint32_t grid_x = (uint32_t)a.grid_x;
    GSR_BTRACE(16384 + 8192 * slab + blockIdx.x, 1);

    // first item whose inclusive offset exceeds p_begin = number of items with offset <= p_begin: whole 1024-item
    // tiles first (their last offsets), then inside the tile that straddles p_begin
    const int tiles = (V + kDupTile - 1) / kDupTile;
    uint32_t n_le = 0;
    for (int t = tid; t < tiles; t += 256) n_le += offsets[min((t + 1) * kDupTile, V) - 1] <= p_begin ? 1u : 0u;
    const int tile0 = (int)block_sum_256(n_le, s_scratch);  // tiles that end at or before p_begin
    n_le = 0;
    {
        const int k0 = tile0 * kDupTile + 4 * tid;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < V) n_le += offsets[k0 + j] <= p_begin ? 1u : 0u;
    }
    int s0 = tile0 * kDupTile + (int)block_sum_256(n_le, s_scratch);
    auto position_of = [&](int item) -> uint32_t { return compacted ? cpos[item] : (uint32_t)item; };
    GSR_BTRACE(16384 + 8192 * slab + blockIdx.x, 2);

    const uint32_t my_begin = p_begin + per_lane * (uint32_t)tid;
    const uint32_t my_end = min(p_end, my_begin + per_lane);
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
            uint32_t pos = position_of(s0 + owner - 1);
            w.start(sorted_bins[pos], lo_pair - s_incl[owner - 1], a.run_pool, a.run_incl, done, a.row_words);
            uint32_t gid = order[pos];
            if (a.listed != nullptr) a.listed[gid] = (uint8_t)(slab + 1);  // (several lanes may say so: same value)
#pragma unroll
            for (int q = 0; q < kPairsPerLane; ++q) {
                const uint32_t p = my_begin + (uint32_t)q;
                if (p >= lo_pair && p < hi_pair) {
                    if (p >= owner_end) {
                        do { owner_end = s_incl[++owner]; } while (p >= owner_end);  // splats without live tiles
                        pos = position_of(s0 + owner - 1);
                        w.start(sorted_bins[pos], 0u, a.run_pool, a.run_incl, done, a.row_words);
                        gid = order[pos];
                        if (a.listed != nullptr) a.listed[gid] = (uint8_t)(slab + 1);
                    }
                    keys[q] = w.next(grid_x);
                    ids[q] = gid;
                }
            }
        }
        const bool finished = s_incl[kBatch] >= p_end;
        __syncthreads();
        if (finished) break;
        s0 += kBatch;
    }
    GSR_BTRACE(16384 + 8192 * slab + blockIdx.x, 3);
    if (first_counts != nullptr) {
        // The tile sort's first pass would read these keys back only to count their low digits per 4096-pair tile: the
        // workgroup that made them counts them itself (one launch less per slab), and the sort's first scan folds the two or
        // four workgroups of a tile.  Every one of the 256 digit rows is written (zeros above the mask): the scan reads all.
        uint32_t* s_hist = s_incl;   // 4 x 256 words of the offsets' parking place (every lane is past its last read of it)
        __syncthreads();
        for (int i = tid; i < 4 * 256; i += 256) s_hist[i] = 0u;
        __syncthreads();
        uint32_t* mine = s_hist + 256 * (tid & 3);
#pragma unroll
        for (int q = 0; q < kPairsPerLane; ++q)
            if ((uint32_t)q < per_lane && my_begin + q < p_end) atomicAdd(&mine[keys[q] & digit_mask], 1u);
        __syncthreads();
        first_counts[(size_t)tid * count_stride + blockIdx.x] = s_hist[tid] + s_hist[256 + tid] + s_hist[512 + tid] + s_hist[768 + tid];
    }
    if (my_begin + per_lane <= p_end) {   // (per_lane is a multiple of 4 and so is my_begin: 16-byte stores)
#pragma unroll
        for (int q = 0; q < kPairsPerLane; q += 4) {
            if ((uint32_t)q >= per_lane) break;
            *reinterpret_cast<uint4*>(tile_keys + my_begin + q) = make_uint4(keys[q], keys[q + 1], keys[q + 2], keys[q + 3]);
            *reinterpret_cast<uint4*>(point_list + my_begin + q) = make_uint4(ids[q], ids[q + 1], ids[q + 2], ids[q + 3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < kPairsPerLane; ++q)
            if ((uint32_t)q < per_lane && my_begin + q < p_end) {
                tile_keys[my_begin + q] = keys[q];
                point_list[my_begin + q] = ids[q];
            }
    }
    GSR_BTRACE(16384 + 8192 * slab + blockIdx.x, 4);
}

// ------------------------------------------------------------------------------------------------
// K5 on its own (calls without a colour kernel to carry it: full calls, precomputed colours): gsr_device.h tile_ranges_duty.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tile_ranges_kernel(RangesDuty duty) {
    __shared__ uint32_t s_first[kRangesDutyLdsWords];
    tile_ranges_duty(duty, blockIdx.x, s_first);
}

// ------------------------------------------------------------------------------------------------
// debug calls only: entry i of a slab's sorted list must not precede entry i - 1 in (tile, depth bits, Gaussian id)
// order -- the order the reference's single 64-bit stable sort yields (rasterizer_impl.cu:304-309) and that the depth
// sort + stable tile sort here must reproduce.  An unstable pass anywhere would show up as a violation.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) list_order_check_kernel(const SlabInfo* __restrict__ slab, const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ list,
                                                               const SplatRaster* __restrict__ raster, FrameCounters* counters) {
    const uint32_t n = slab->pairs;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x + 1u;
    bool bad = false;
    if (i < n) {
        const uint32_t ta = keys[i - 1], tb = keys[i], ga = list[i - 1], gb = list[i];
        uint32_t da = __float_as_uint(raster[ga].depth), db = __float_as_uint(raster[gb].depth);
        if (da == kCulledKey) da = kCulledKey - 1u;   // (as the projection kernel keys a NaN depth)
        if (db == kCulledKey) db = kCulledKey - 1u;
        bad = ta > tb || (ta == tb && (da > db || (da == db && ga >= gb)));
    }
    const unsigned long long b = __ballot(bad);
    if (b != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&counters->order_violations, (uint32_t)__popcll(b));
}

} // namespace

hipError_t launch_list_order_check(const SlabInfo* slab, uint32_t pairs_bound, const uint32_t* sorted_tile_keys,
                                   const uint32_t* point_list, const SplatRaster* raster, FrameCounters* counters,
                                   hipStream_t stream) {
    if (pairs_bound < 2u) return hipSuccess;
    hipLaunchKernelGGL(list_order_check_kernel, dim3((pairs_bound + 255u) / 256u), dim3(256), 0, stream, slab, sorted_tile_keys,
                       point_list, raster, counters);
    return hipGetLastError();
}

hipError_t launch_bin_scan(const BinningArrays& a, const Camera& cam, int num_slabs, const uint32_t* pair_cuts, hipStream_t stream) {
    (void)cam;
    const int tiles_p = div_up(a.P, kDupTile), tiles_v = div_up(a.V, kDupTile);
    if (tiles_v > 0) hipLaunchKernelGGL(bin_gather_kernel, dim3(tiles_v), dim3(256), 0, stream, a);
    uint32_t* tile_ends = a.tile_totals + tiles_p;
    SlabCuts cuts = {};
    for (int s = 0; s + 1 < num_slabs; ++s) cuts.cut[s] = pair_cuts[s];
    // (V == 0: the slab table stays as the call's zero-filled block left it: no positions, no pairs)
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(tiles_p), dim3(256), 0, stream, a.P, a.V, a.tile_totals, a.offsets, tile_ends, num_slabs, cuts,
                       a.slabs, a.slabs_host);
    return hipGetLastError();
}

hipError_t launch_slab_recount(const BinningArrays& a, int slab, hipStream_t stream) {
    const int tiles_v = div_up(a.V, kDupTile), tiles_p = div_up(a.P, kDupTile);
    (void)tiles_p;
    // (an empty slab still needs its counts set: slab_compact_kernel's first workgroup does that)
    // The bit rows are dynamic LDS, sized by what the image needs (1 KB at 1920x1080): with a fixed 16 KB the expansion's
    // workgroups did not all fit on the GPU at once (6 per CU = 1 536 of C3's 1 594) and the few left over ran as a second,
    // nearly empty round behind the others -- a third of the kernel's time.
    const int done_words = a.grid_y * a.row_words;
    if (done_words > kMaxDoneWords) return hipErrorInvalidValue;   // (gsr_api.hip does not plan slabs for such an image)
    if (tiles_v > 0) hipLaunchKernelGGL(slab_recount_kernel, dim3(tiles_v), dim3(256), (size_t)done_words * 4, stream, a, slab);
    hipLaunchKernelGGL(slab_compact_kernel, dim3(tiles_v > 0 ? tiles_v : 1), dim3(256), 0, stream, a, slab);
    return hipGetLastError();
}

uint32_t expand_blocks(uint32_t pairs_bound) { return (pairs_bound + kPairTile - 1) / kPairTile; }

hipError_t launch_expand(const BinningArrays& a, int slab, uint32_t pairs_bound, uint32_t* tile_keys, uint32_t* point_list,
                         uint32_t* first_counts, uint32_t count_stride, uint32_t digit_mask, hipStream_t stream) {
    if (pairs_bound == 0) return hipSuccess;
    const int done_words = slab > 0 ? a.grid_y * a.row_words : 0;
    if (done_words > kMaxDoneWords) return hipErrorInvalidValue;
    hipLaunchKernelGGL(expand_kernel, dim3(expand_blocks(pairs_bound)), dim3(256), (size_t)done_words * 4, stream, a, slab,
                       tile_keys, point_list, first_counts, count_stride, digit_mask);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(const RangesDuty& duty, hipStream_t stream) {
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(ranges_duty_blocks(duty.num_tiles)), dim3(256), 0, stream, duty);
    return hipGetLastError();
}

} // namespace gsr
