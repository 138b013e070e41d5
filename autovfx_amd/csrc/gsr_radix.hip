// gsr_radix.hip -- hand-written LSD radix sort of (u32 key, u32 payload) pairs for gfx950.
//
// Replaces the reference's cub::DeviceRadixSort::SortPairs (DGR/cuda_rasterizer/rasterizer_impl.cu:304-309)
// for the two sorts of this library's pipeline (depth keys over Gaussians, tile ids over live pairs; see
// gsr_api.hip).  Stable, ascending, 8 bits per pass.
//
// At these sizes (3 M / 9 M items) a pass is bound by latency and occupancy, not by HBM.  Measured on MI355X
// (scripts/ubench/radix_trace.hip) a one-sweep pass with decoupled look-back spends 40 % of every
// workgroup's life spinning on its predecessors' words, the ticket counter that orders the workgroups hands
// out one ticket per 12 ns (same-address atomics), and the generic library version additionally queues one
// memset per pass.  This file therefore uses the spin-free three-kernel pass:
//     count   : digit histogram of each 4096-pair tile                  (keys read once, no atomics in HBM)
//     scan    : per digit, exclusive prefix over tiles                   (256 x tiles words)
//     scatter : stable rank inside the tile, park in LDS, coalesced write-out
// No workgroup ever waits on another one, nothing has to be zero-filled, and there is no forward-progress
// assumption.  The extra key read (4 of 20 bytes per pair and pass) is cheap next to what it removes.
//
// Scatter kernel, one workgroup = 256 lanes (4 wave64) x 16 items = 4096 pairs:
//   1. keys are loaded wave-striped (item i of lane L at wave_base + 64 i + L: coalesced);
//   2. stable rank inside the wave, in one of two forms, BOTH compiled in (template parameter kRank):
//        - ballots: 8 ballots per item match the lanes with the same digit, popcount below the lane.  Relies on
//          nothing but the ISA (GSR_OPT_RADIX_RANK = 0).
//        - VERIFIED LDS atomics (the default, GSR_OPT_RADIX_RANK = 2): one returning LDS add per item on the wave's counter
//          of the item's digit.  The values returned are the right rank if lanes of one instruction that hit the same counter
//          are served in ascending lane order -- what gfx950's LDS is observed to do, but no manual promises it.  So the
//          kernel does not rely on it: before a tile writes anything, it parks (digit, index-in-tile) at the positions the
//          ranks give and checks that the word rises strictly from position to position (one compare per item; the digit
//          counts, hence the segments, are right whatever the order because the adds are atomic).  A tile that fails the
//          check is ranked again with ballots, in place, and counted (GSR_OPT_RADIX_RANK_FALLBACKS); the output is the stable
//          sort either way.  The check costs ~8 of the ~35 vector instructions per item the ballots cost.
//          Request 2 additionally runs the lane-order self-test (lds_atomic_order_selftest_kernel) on the first sort and
//          uses plain ballots on a device that fails it -- there every tile would pay the atomics AND the ballots.
//   3. counters are turned into (wave, digit) offsets, a 256-wide scan gives the digit segments of the tile;
//   4. pairs are parked in LDS at their in-tile position and written out in that order, so every digit
//      segment is a contiguous, coalesced run in HBM.
//
// A sort may DROP one key value in its first pass (radix_sort_pairs' drop_key: the depth sort's culled Gaussians, a quarter
// of C3's and most of a close-up's): the count kernel does not count those items, the scatter neither ranks nor writes them
// and stores how many were kept; the later passes read that count (n_device) and sort the kept ones only.
#include "gsr_internal.h"

#include <mutex>

namespace gsr {
namespace {

constexpr int kThreads = 256;
constexpr int kItems = 16;
constexpr int kTileItems = kThreads * kItems;
constexpr int kWaves = kThreads / 64;
constexpr int kCountCopies = 8;  // sub-histograms of the count kernel: lanes L and L+8 share one
static_assert(kTileItems == kRadixTile, "gsr_internal.h sizes the per-tile counters with kRadixTile");

#ifdef GSR_RADIX_TRACE  // scripts/ubench/radix_trace.hip: per-workgroup phase timestamps (100 MHz wall clock)
__device__ unsigned long long* g_radix_trace = nullptr;
#define GSR_TRACE(slot) do { if (tid == 0 && g_radix_trace) g_radix_trace[(size_t)block * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define GSR_TRACE(slot) do { } while (0)
#endif

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive sum over the 256 lanes of the workgroup; `scratch` is kWaves words of LDS; *total = sum of all.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* scratch, int tid, uint32_t* total = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = wave_inclusive_sum(v, lane);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t t = scratch[w];
        if (w < wave) before += t;
        all += t;
    }
    __syncthreads();
    if (total) *total = all;
    return before + incl - v;
}

// ------------------------------------------------------------------------------------------------
// The projection kernel's tallies, summed by ONE workgroup (gsr_internal.h: TallyDuty): the totals go straight into the call's
// pinned slot (rasterizer_impl.cu:282 reads num_rendered back at this point) and the call's zero block is cleared with plain
// stores.  Rounds 1 - 2 had a memset in front of the projection and a copy kernel behind it, round 3 one short launch of its
// own (7 us between the projection and the depth sort); now it rides on the depth sort's first count kernel: `groups` extra
// workgroups (one when the totals land in the zero block itself), each filling its own slot of the totals, which the host adds.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void counter_tally_duty(const TallyDuty& d, int group, uint32_t* lds /* >= 7 * kWaves words */) {
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(lds);
    uint32_t *s_vis = lds + 2 * kWaves, *s_big = s_vis + kWaves, *s_flag = s_big + kWaves, *s_scan = s_flag + kWaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = group * kThreads + tid, stride = d.groups * kThreads;
    // this group's share of the tallies is contiguous and a lane takes four consecutive entries of it: the rows of the large
    // splats are summed as a prefix (pool_first), everything else as a total
    const int chunk = 1 << d.chunk_shift;
    const int share_first = min(group * chunk, d.blocks), share_end = min(share_first + chunk, d.blocks);
    unsigned long long tot = 0ull;
    uint32_t vis = 0u, big = 0u, flag = 0u;
    // four independent loads in flight per lane: at 3 M Gaussians and 16 groups that is the whole share of a lane, one memory
    // round trip (one group walking all 11 719 tallies a load at a time outlasted the count kernel it rides on)
    for (int base = share_first; base < share_end; base += 4 * kThreads) {   // (workgroup-uniform bounds)
        const int b0 = base + 4 * tid;
        uint4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = b0 + j < share_end ? *reinterpret_cast<const uint4*>(d.tallies + b0 + j) : make_uint4(0u, 0u, 0u, 0u);
        uint32_t rows[4], mine = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tot += (unsigned long long)t[j].x | ((unsigned long long)t[j].y << 32);
            vis += t[j].z;
            rows[j] = t[j].w & 0x7FFFFFFFu;
            mine += rows[j];
            flag |= t[j].w >> 31;
        }
        uint32_t round_total;
        uint32_t before = big + block_exclusive_sum(mine, s_scan, tid, &round_total);   // (big: the same in every lane)
        big += round_total;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (b0 + j < share_end) d.pool_first[b0 + j] = before;
            before += rows[j];
        }
    }
    uint32_t* z = reinterpret_cast<uint32_t*>(d.zero_block);
    for (int i = first; i < d.zero_words; i += stride) z[i] = 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        tot += __shfl_xor(tot, s);
        vis += (uint32_t)__shfl_xor((int)vis, s);
        flag |= (uint32_t)__shfl_xor((int)flag, s);
    }
    if (lane == 0) { s_tot[wave] = tot; s_vis[wave] = vis; s_flag[wave] = flag; }
    __syncthreads();   // (one group, totals into the zero block itself: it is cleared before they land in it)
    if (tid == 0) {
        tot = 0ull; vis = 0u; flag = 0u;   // (big: already the group's total)
        for (int w = 0; w < kWaves; ++w) { tot += s_tot[w]; vis += s_vis[w]; flag |= s_flag[w]; }
        FrameCounters* dst = d.host_totals != nullptr ? d.host_totals : d.zero_block;
        dst->pair_totals[group] = tot;                     // the host adds the slots up
        dst->visible[group] = vis;
        dst->big_rows[group] = big | (flag << 31);         // (bit 31, as in a BlockTally: a prefiltered violation)
    }
}

// ------------------------------------------------------------------------------------------------
// count: counts[digit * tiles_pad + tile] = number of keys of the tile with that digit.  With a duty (duty.tallies != null)
// the launch has duty.groups workgroups more and the first ones do the duty instead.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) radix_count_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                              uint32_t digit_mask, uint32_t* __restrict__ counts,
                                                              uint32_t tiles_pad, const uint32_t* __restrict__ n_device,
                                                              int drop, uint32_t drop_key, TallyDuty duty) {
    __shared__ uint32_t s_hist[kCountCopies][256];
    const int tid = threadIdx.x;
    uint32_t block = blockIdx.x;
    if (duty.tallies != nullptr) {   // (workgroup-uniform)
        if (block < (uint32_t)duty.groups) { counter_tally_duty(duty, (int)block, &s_hist[0][0]); return; }
        block -= (uint32_t)duty.groups;
    }
    if (n_device != nullptr) n = *n_device;  // the launch was sized for an upper bound
    if (block * (uint32_t)kTileItems >= n) return;
#pragma unroll
    for (int j = 0; j < kCountCopies; ++j) s_hist[j][tid] = 0u;
    __syncthreads();
    uint32_t* mine = s_hist[tid & (kCountCopies - 1)];
    const uint32_t tile_base = block * (uint32_t)kTileItems;
    if (tile_base + kTileItems <= n) {  // 16-byte loads: key arrays are 256-byte aligned sub-arrays of an arena
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + tile_base);
        uint4 k[kItems / 4];
#pragma unroll
        for (int i = 0; i < kItems / 4; ++i) k[i] = k4[i * kThreads + tid];
        if (!drop) {
#pragma unroll
            for (int i = 0; i < kItems / 4; ++i) {
                atomicAdd(&mine[(k[i].x >> shift) & digit_mask], 1u);
                atomicAdd(&mine[(k[i].y >> shift) & digit_mask], 1u);
                atomicAdd(&mine[(k[i].z >> shift) & digit_mask], 1u);
                atomicAdd(&mine[(k[i].w >> shift) & digit_mask], 1u);
            }
        } else {   // keys equal to drop_key leave the sort with this pass: they are not counted (and not written by the scatter)
#pragma unroll
            for (int i = 0; i < kItems / 4; ++i) {
                if (k[i].x != drop_key) atomicAdd(&mine[(k[i].x >> shift) & digit_mask], 1u);
                if (k[i].y != drop_key) atomicAdd(&mine[(k[i].y >> shift) & digit_mask], 1u);
                if (k[i].z != drop_key) atomicAdd(&mine[(k[i].z >> shift) & digit_mask], 1u);
                if (k[i].w != drop_key) atomicAdd(&mine[(k[i].w >> shift) & digit_mask], 1u);
            }
        }
    } else {
        for (uint32_t t = tid; tile_base + t < n; t += kThreads) {
            const uint32_t kk = keys[tile_base + t];
            if (!drop || kk != drop_key) atomicAdd(&mine[(kk >> shift) & digit_mask], 1u);
        }
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < kCountCopies; ++j) c += s_hist[j][tid];
    counts[(size_t)tid * tiles_pad + block] = c;
}

// ------------------------------------------------------------------------------------------------
// scan: one workgroup per digit turns its row of tile counts into exclusive prefixes, in place;
// totals[digit] = number of keys with that digit.
// ------------------------------------------------------------------------------------------------
// kFold (the first pass of a sort whose keys were counted by the kernel that wrote them, radix_sort_pairs' precount_blocks): the
// row holds one count per WORKGROUP of that kernel, two or four of them to a 4096-key tile (expand_pairs_per_lane); a lane adds
// up the workgroups of its four tiles first.  In place: a tile's prefix lands on a word that only this round's lanes read, before
// the barrier inside the block sum (tile t's workgroups start at word t * g >= t).
template <bool kFold>
__global__ void __launch_bounds__(kThreads) radix_scan_kernel(uint32_t* __restrict__ counts, uint32_t tiles,
                                                             uint32_t tiles_pad, uint32_t* __restrict__ totals,
                                                             const uint32_t* __restrict__ n_device, uint32_t precount_blocks) {
    __shared__ uint32_t s_scan[kWaves];
    const int tid = threadIdx.x;
    uint32_t group = 1u, live_blocks = 0u;
    if (n_device != nullptr) {
        const uint32_t n = *n_device;
        tiles = (n + (uint32_t)kTileItems - 1u) / (uint32_t)kTileItems;
        if (kFold) {
            const uint32_t per_block = 256u * expand_pairs_per_lane(precount_blocks, n);
            group = (uint32_t)kTileItems / per_block;
            live_blocks = (n + per_block - 1u) / per_block;   // the workgroups behind them wrote nothing
        }
    }
    uint32_t* row_words = counts + (size_t)blockIdx.x * tiles_pad;
    uint4* row = reinterpret_cast<uint4*>(row_words);  // tiles_pad is a multiple of 4
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < tiles; t0 += 4u * kThreads) {
        const uint32_t t = t0 + 4u * tid;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (kFold) {
            uint32_t c[4] = {0u, 0u, 0u, 0u};
            if (t < tiles) {
                for (uint32_t j = 0; j < 4u * group; j += 4u) {   // 16-byte loads: t * group is a multiple of 4
                    const uint32_t b = t * group + j;
                    if (b >= live_blocks) break;
                    const uint4 w = *reinterpret_cast<const uint4*>(row_words + b);
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (b + k < live_blocks) c[(j + k) / group] += ww[k];
                }
            }
            v = make_uint4(c[0], c[1], c[2], c[3]);
        } else {
            if (t < tiles_pad) v = row[t / 4u];
        }
        if (t + 0 >= tiles) v.x = 0u;
        if (t + 1 >= tiles) v.y = 0u;
        if (t + 2 >= tiles) v.z = 0u;
        if (t + 3 >= tiles) v.w = 0u;
        uint32_t chunk_total;
        const uint32_t before = carry + block_exclusive_sum(v.x + v.y + v.z + v.w, s_scan, tid, &chunk_total);
        if (t < tiles_pad) row[t / 4u] = make_uint4(before, before + v.x, before + v.x + v.y, before + v.x + v.y + v.z);
        carry += chunk_total;
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// ------------------------------------------------------------------------------------------------
// scatter.  kIota: payloads are the item indices (first pass of the depth sort: no payload read).
// kKeysOut: false on a last pass whose sorted keys nobody reads.  kRank: how the in-wave rank is found -- 0 ballots, 1 returning
// LDS adds whose result every workgroup VERIFIES before it writes anything (and redoes with ballots if the check fails),
// 2 = 1 with an inversion injected into every wave (test hook: the check must catch it and the ballots repair it).
// ------------------------------------------------------------------------------------------------
__device__ unsigned long long g_rank_fallbacks = 0ull;   // tiles whose LDS-add ranks failed the order check (this device, since load)

template <bool kIota, bool kKeysOut, int kRank>
__global__ void __launch_bounds__(kThreads) radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                                uint32_t digit_mask, const uint32_t* __restrict__ offsets,
                                                                uint32_t tiles_pad, const uint32_t* __restrict__ totals,
                                                                const uint32_t* __restrict__ n_device, uint32_t drop_key,
                                                                uint32_t* __restrict__ kept_out) {
    __shared__ uint32_t s_keys[kTileItems];
    __shared__ uint32_t s_vals[kTileItems];
    __shared__ uint32_t s_count[kWaves][256];  // per-wave digit counts, then per-wave offsets inside the segment
    __shared__ uint32_t s_seg_start[256];      // first in-tile position of each digit segment
    __shared__ uint32_t s_dst_base[256];       // global position of the segment minus s_seg_start (mod 2^32)
    __shared__ uint32_t s_scan[kWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t block = blockIdx.x;
    if (n_device != nullptr) n = *n_device;  // the launch was sized for an upper bound
    if (block * (uint32_t)kTileItems >= n) return;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s_count[w][tid] = 0u;
    const uint32_t tile_base = block * (uint32_t)kTileItems;
    const uint32_t tile_n = min((uint32_t)kTileItems, n - tile_base);
    GSR_TRACE(0);

    // 1. load (wave-striped); the tile's global digit offsets ride along
    uint32_t key[kItems], val[kItems];
    const uint32_t first = (uint32_t)wave * (64u * kItems) + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t t = first + 64u * i;
        key[i] = t < tile_n ? keys_in[tile_base + t] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t t = first + 64u * i;
        if (kIota) val[i] = tile_base + t;
        else val[i] = t < tile_n ? vals_in[tile_base + t] : 0u;
    }
    // items of this lane that take part: inside the tile, and (first pass of a sort that drops a key value, kept_out != null:
    // the depth sort's culled Gaussians) not the dropped key -- those are neither ranked nor written, the later passes never
    // see them
    uint32_t ok = 0u;
#pragma unroll
    for (int i = 0; i < kItems; ++i)
        if (first + 64u * i < tile_n && !(kIota && kept_out != nullptr && key[i] == drop_key)) ok |= 1u << i;
    const uint32_t tile_offset = offsets[(size_t)tid * tiles_pad + block];  // keys of digit `tid` in earlier tiles
    const uint32_t digit_total = totals[tid];
    __syncthreads();

    // 2. stable rank inside the wave
    uint32_t rank[kItems];
    // by ballots: 8 ballots per item match the lanes with the same digit; `counters` = 256 zeroed words of this wave
    auto rank_by_ballots = [&](uint32_t* counters) {
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const bool valid = (ok >> i) & 1u;
            const uint32_t d = (key[i] >> shift) & digit_mask;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if ((digit_mask >> b) == 0u) break;  // wave-uniform: a narrow last digit needs fewer ballots
                const bool set = (d >> b) & 1u;
                const unsigned long long with_bit = __ballot(set);
                peers &= set ? with_bit : ~with_bit;
            }
            const uint32_t before = counters[d];  // every peer reads the same word before the bump below
            rank[i] = before + (uint32_t)__popcll(peers & below);
            if (valid && (peers >> lane) == 1ull) counters[d] = before + (uint32_t)__popcll(peers);  // highest peer
        }
    };
    if (kRank != 0) {
        // One returning LDS add per item on the wave's counter of its digit.  On gfx950 lanes of one instruction that hit the
        // same counter are observed to be served in ascending lane order (and a wave's instructions retire in program order),
        // in which case the returned values number the wave's items of a digit in tile order (item i of lane L sits at
        // 64 i + L) -- the rank the ballots compute.  No manual promises that order, so nothing rests on it: the counts are
        // right whatever the order (the adds are atomic), and step 4 checks the order itself before anything leaves the tile.
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            const bool valid = (ok >> i) & 1u;
            const uint32_t d = (key[i] >> shift) & digit_mask;
            rank[i] = valid ? atomicAdd(&s_count[wave][d], 1u) : 0u;
        }
        if (kRank == 2) {   // test hook: the two lowest lanes that share lane 0's digit in item 0 trade ranks
            const uint32_t d = (key[0] >> shift) & digit_mask;
            const uint32_t d_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            unsigned long long peers = __ballot((ok & 1u) && d == d_first);
            if (__popcll(peers) >= 2) {
                const int l0 = __ffsll((long long)peers) - 1;
                peers &= peers - 1ull;
                const int l1 = __ffsll((long long)peers) - 1;
                const uint32_t r0 = (uint32_t)__shfl((int)rank[0], l0), r1 = (uint32_t)__shfl((int)rank[0], l1);
                if (lane == l0) rank[0] = r1;
                if (lane == l1) rank[0] = r0;
            }
        }
    } else {
        rank_by_ballots(s_count[wave]);
    }
    __syncthreads();
    GSR_TRACE(1);

    // 3. digit d is lane d's from here on
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = s_count[w][tid];
        s_count[w][tid] = total;
        total += c;
    }
    uint32_t tile_kept;   // = tile_n unless keys were dropped
    const uint32_t seg_start = block_exclusive_sum(total, s_scan, tid, &tile_kept);
    const uint32_t digit_start = block_exclusive_sum(digit_total, s_scan, tid);  // keys with a smaller digit
    s_seg_start[tid] = seg_start;
    s_dst_base[tid] = digit_start + tile_offset - seg_start;
    if (kIota && kept_out != nullptr && block == 0u && tid == 255) *kept_out = digit_start + digit_total;  // what the later passes sort
    __syncthreads();
    GSR_TRACE(2);

    // 4. park in tile order, then stream out
    if (kRank == 0) {
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            if ((ok >> i) & 1u) {
                const uint32_t d = (key[i] >> shift) & digit_mask;
                const uint32_t pos = s_seg_start[d] + s_count[wave][d] + rank[i];
                s_keys[pos] = key[i];
                s_vals[pos] = val[i];
            }
        }
    } else {
        // 4a. park the keys, and in the payload's place (digit, index in the tile): the sort of this tile is right if and only
        //     if that word rises strictly along the parked positions -- digits ascend from segment to segment by construction
        //     (step 3 only used the COUNTS, which the atomicity of the adds guarantees), so the only thing that can be wrong
        //     is the order inside a (wave, digit) run, and a run out of tile order has a descent
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            if ((ok >> i) & 1u) {
                const uint32_t d = (key[i] >> shift) & digit_mask;
                const uint32_t pos = s_seg_start[d] + s_count[wave][d] + rank[i];
                s_keys[pos] = key[i];
                s_vals[pos] = (d << 12) | (first + 64u * i);
                rank[i] = pos;
            }
        }
        __syncthreads();
        // 4b. the check: 4096 comparisons of neighbours
        int descent = 0;
#pragma unroll
        for (int j = 0; j < kItems; ++j) {
            const uint32_t p = (uint32_t)(j * kThreads + tid);
            if (p != 0u && p < tile_kept) descent |= (int)(s_vals[p] <= s_vals[p - 1u]);
        }
        if (__syncthreads_or(descent)) {
            // 4c. (never taken on the hardware this was measured on) rank again with ballots, on counters carved from the
            //     payload area, and park the keys where they belong
            for (int w = 0; w < kWaves; ++w) s_vals[w * 256 + tid] = 0u;
            __syncthreads();
            rank_by_ballots(s_vals + wave * 256);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                if ((ok >> i) & 1u) {
                    const uint32_t d = (key[i] >> shift) & digit_mask;
                    const uint32_t pos = s_seg_start[d] + s_count[wave][d] + rank[i];
                    s_keys[pos] = key[i];
                    rank[i] = pos;
                }
            }
            if (tid == 0) atomicAdd(&g_rank_fallbacks, 1ull);
        }
        // 4d. the payloads
#pragma unroll
        for (int i = 0; i < kItems; ++i)
            if ((ok >> i) & 1u) s_vals[rank[i]] = val[i];
    }
    __syncthreads();
    GSR_TRACE(3);
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t p = (uint32_t)(j * kThreads + tid);
        if (p < tile_kept) {
            const uint32_t k = s_keys[p];
            const uint32_t dst = s_dst_base[(k >> shift) & digit_mask] + p;
            if (kKeysOut) keys_out[dst] = k;
            vals_out[dst] = s_vals[p];
        }
    }
    GSR_TRACE(4);
}

} // namespace

namespace {
// Self-test of what the scatter kernel's ranking relies on: 4 waves of a workgroup each issue `rounds` returning LDS
// adds on 256 counters with pseudo-random lane -> counter maps of every density (all lanes on one counter ... all on
// different ones); the value a lane gets back must be the counter's value before the instruction plus the number of
// LOWER lanes of the same instruction on the same counter.  Counts violations.
__global__ void __launch_bounds__(256) lds_atomic_order_selftest_kernel(uint32_t rounds, uint32_t seed, unsigned long long* mismatches) {
    __shared__ uint32_t s_ctr[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_ctr[wave][tid & 255] = 0u;
    s_ctr[wave][(tid + 64) & 255] = 0u; s_ctr[wave][(tid + 128) & 255] = 0u; s_ctr[wave][(tid + 192) & 255] = 0u;
    __syncthreads();
    uint32_t expect_total[4] = {0u, 0u, 0u, 0u};   // shadow of 4 of the wave's counters per lane: counter 4 * lane + q
    uint32_t bad = 0u;
    uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ ((uint32_t)wave << 20);
    for (uint32_t r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;                       // wave-uniform round parameters
        const uint32_t spread = 1u << ((x >> 8) % 9u);        // 1, 2, 4, ... 256 counters in play this round
        uint32_t h = (x + (uint32_t)lane * 0x85EBCA6Bu); h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        const uint32_t d = h & (spread - 1u) & 255u;
        const uint32_t got = atomicAdd(&s_ctr[wave][d], 1u);
        // what it must be: this lane's peers on counter d, and the counter's value before the instruction
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool set = (d >> b) & 1u;
            const unsigned long long with_bit = __ballot(set);
            peers &= set ? with_bit : ~with_bit;
        }
        // the counter's previous value is tracked by its owner lane (counter c is shadowed by lane c / 4, slot c % 4)
        const uint32_t owner = d >> 2, slot = d & 3u;
        uint32_t prev = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t v = (uint32_t)__shfl((int)expect_total[q], (int)owner);
            if ((uint32_t)q == slot) prev = v;
        }
        if (got != prev + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))) ++bad;
        // owners update their shadows: counter 4 * lane + q gained popcount(lanes whose d equals it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t c = 4u * (uint32_t)lane + (uint32_t)q;
            unsigned long long hit = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long with_bit = __ballot((d >> b) & 1u);
                hit &= ((c >> b) & 1u) ? with_bit : ~with_bit;
            }
            expect_total[q] += (uint32_t)__popcll(hit);
        }
    }
    if (bad != 0u) atomicAdd(mismatches, (unsigned long long)bad);
}

} // namespace

hipError_t launch_lds_atomic_order_selftest(uint32_t workgroups, uint32_t rounds, uint32_t seed, unsigned long long* mismatches,
                                            hipStream_t stream) {
    hipLaunchKernelGGL(lds_atomic_order_selftest_kernel, dim3(workgroups), dim3(256), 0, stream, rounds, seed, mismatches);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Which in-wave rank the scatter kernel uses on the current device.  g_rank_request: 0 ballots; 1 verified LDS atomics
// without asking; 2 (DEFAULT) verified LDS atomics on a device that passed the lane-order self-test, ballots elsewhere; 3 =
// 1 with an inversion injected into every wave (test hook for the check and the in-place repair).
// Since round 4 the sort's correctness does not depend on the lane order in any of these: the atomics' ranks are checked per
// tile inside the scatter kernel and repaired with ballots when they fail (see the kernel); the self-test only decides
// whether the atomics are worth trying on this device.
// With request 2 the self-test runs once per device and process, in the first sort (or the first query) after the request:
// 512 workgroups x 4 waves x 192 instructions of every conflict density (~0.4 M instructions, 25 M lane results; a quarter of
// a millisecond), on its OWN stream and its own small allocation, waited for by the host (hipMalloc / hipFree /
// hipStreamSynchronize of that stream under a process-wide mutex); the caller's stream is neither drained nor used.  While the
// caller's stream is capturing a graph the test does not run (allocations are not legal then): that sort ranks with ballots and
// the next one tries again, as after a test that could not run for any other reason (e.g. out of memory) -- only verdicts are
// remembered.  Query GSR_OPT_RADIX_RANK_ACTIVE at start-up to have it out of the way.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_rank_mutex;
int g_rank_request = 2;
int g_rank_verdict[kMaxDevices];      // 0 not tested yet, 1 passed (atomics), 2 failed (ballots), 3 could not be tested (ballots)
unsigned long long g_rank_violations[kMaxDevices];

int g_rank_attempts[kMaxDevices];      // self-tests that could not run (verdict 3) on this device so far
constexpr int kMaxSelftestAttempts = 4;
__device__ unsigned long long g_selftest_word;   // the self-test's mismatch counter: a device symbol, so the test allocates nothing

int test_device_locked(int dev, hipStream_t caller) {
    // Never on the caller's stream: the test has its own, so the caller's queue is not drained and nothing is queued on a stream
    // that may be recording a graph.  While the caller's stream IS capturing the test does not run at all: verdict 3, ballots for
    // this sort, another try at the next one.  No hipMalloc / hipFree (both synchronise the whole device; round 4 did both here):
    // the counter is a device symbol.  What remains under the process-wide lock, once per device: a stream created and destroyed
    // and a wait of a quarter of a millisecond on it.  A test that keeps failing to RUN (stream creation, a launch error) is given
    // up after kMaxSelftestAttempts sorts -- ballots from then on -- instead of being repeated in front of every sort of every frame.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (caller != nullptr && (hipStreamIsCapturing(caller, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();
        return 3;   // (not counted as an attempt: capture ends)
    }
    unsigned long long* d_bad = nullptr;
    unsigned long long h_bad = ~0ull;
    int verdict = 3;
    hipStream_t own = nullptr;
    if (hipGetSymbolAddress((void**)&d_bad, HIP_SYMBOL(g_selftest_word)) == hipSuccess &&
        hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess) {
        if (hipMemsetAsync(d_bad, 0, sizeof *d_bad, own) == hipSuccess &&
            launch_lds_atomic_order_selftest(512u, 192u, 0x6a09e667u, d_bad, own) == hipSuccess &&
            launch_lds_atomic_order_selftest(512u, 192u, 0xbb67ae85u, d_bad, own) == hipSuccess &&
            hipMemcpyAsync(&h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost, own) == hipSuccess &&
            hipStreamSynchronize(own) == hipSuccess)
            verdict = h_bad == 0ull ? 1 : 2;
        (void)hipStreamDestroy(own);
    }
    if (verdict == 3) {
        (void)hipGetLastError();
        if (++g_rank_attempts[dev] >= kMaxSelftestAttempts) verdict = 2;   // give up: ballots on this device for good
    }
    g_rank_violations[dev] = h_bad;
    return verdict;
}
} // namespace

void radix_set_rank_request(int request) {
    std::lock_guard<std::mutex> lock(g_rank_mutex);
    g_rank_request = request < 0 ? 0 : request > 3 ? 2 : request;
}
int radix_rank_request() {
    std::lock_guard<std::mutex> lock(g_rank_mutex);
    return g_rank_request;
}
// Tiles (this device, since the library was loaded) whose LDS-add ranks failed the in-kernel order check and were ranked
// again with ballots.  Reads a device word: synchronises with the null stream only -- synchronise the sorting stream first.
hipError_t radix_rank_fallbacks(unsigned long long* count) {
    return hipMemcpyFromSymbol(count, HIP_SYMBOL(g_rank_fallbacks), sizeof *count, 0, hipMemcpyDeviceToHost);
}
// 1 = verified LDS atomics (2 = with the injected inversion), 0 = ballots, for sorts queued on the current device from now on.
int radix_rank_mode(hipStream_t stream, unsigned long long* violations) {
    std::lock_guard<std::mutex> lock(g_rank_mutex);
    if (violations) *violations = 0ull;
    if (g_rank_request == 0) return 0;
    if (g_rank_request == 1) return 1;
    if (g_rank_request == 3) return 2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    int verdict = g_rank_verdict[dev];
    if (verdict == 0) {
        verdict = test_device_locked(dev, stream);
        if (verdict != 3) g_rank_verdict[dev] = verdict;   // (3 = could not be tested: ballots now, another try at the next sort)
    }
    if (violations) *violations = g_rank_violations[dev];
    return verdict == 1 ? 1 : 0;
}

uint32_t radix_count_stride(uint32_t n, uint32_t precount_blocks) {
    const uint32_t tiles = (n + kTileItems - 1) / kTileItems;
    return (max(tiles, precount_blocks) + 3u) & ~3u;
}
size_t radix_scratch_words(uint32_t n, uint32_t precount_blocks) {
    return (size_t)256u * radix_count_stride(n, precount_blocks) + 256u + 4u;  // per-digit rows of tile counts, the digit totals, the kept count
}
uint32_t radix_first_digit_mask(int bits) {
    if (bits <= 0) return 0u;
    const int passes = (bits + 7) / 8;
    return (1u << (bits / passes + (bits % passes != 0 ? 1 : 0))) - 1u;
}

hipError_t radix_sort_pairs(uint32_t* scratch, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                            uint32_t* vals, uint32_t* vals_alt, bool iota_payload, bool want_sorted_keys,
                            uint32_t** keys_sorted, uint32_t** vals_sorted, hipStream_t stream, const RadixSortExtras& extras) {
    const uint32_t* n_device = extras.n_device;
    const uint32_t* const drop_key = extras.drop_key;
    const bool few_top_digits = extras.few_top_digits;
    const TallyDuty* const first_count_duty = extras.first_count_duty;
    const hipEvent_t after_first_count = extras.after_first_count;
    const uint32_t precount_blocks = extras.precount_blocks;
    *keys_sorted = keys;
    *vals_sorted = vals;
    if (n == 0 || bits <= 0) return hipSuccess;
    if (bits > 32) return hipErrorInvalidValue;
    const int passes = (bits + 7) / 8;
    const int rank_form = radix_rank_mode(stream, nullptr);  // (the first sort on a device runs the lane-order self-test)
    const uint32_t tiles = (n + kTileItems - 1) / kTileItems;
    if (precount_blocks != 0u && n_device == nullptr) return hipErrorInvalidValue;
    const uint32_t tiles_pad = radix_count_stride(n, precount_blocks);
    uint32_t* counts = scratch;
    uint32_t* totals = scratch + (size_t)256u * tiles_pad;
    uint32_t* kept = totals + 256;   // drop_key: how many items the first pass kept = what the later passes sort
    const bool dropping = drop_key != nullptr && iota_payload && n_device == nullptr && passes > 1;
    uint32_t *kin = keys, *kout = keys_alt, *vin = vals, *vout = vals_alt;
    for (int p = 0; p < passes; ++p) {
        // digits as even as the key allows (13 tile-key bits: 7 + 6, not 8 + 5): fewer digits per pass mean longer runs of a digit
        // in a 4096-pair tile, i.e. fewer partially written lines in the scatter
        const int base_w = bits / passes, wide = bits % passes;   // the first `wide` passes are one bit wider
        const int width = base_w + (p < wide ? 1 : 0);
        const int shift = p * base_w + (p < wide ? p : wide);
        const uint32_t mask = (1u << width) - 1u;
        const bool iota = iota_payload && p == 0;
        const bool keys_out = want_sorted_keys || p + 1 < passes;
        const int drop = dropping && p == 0 ? 1 : 0;
        const uint32_t dkey = dropping ? *drop_key : 0u;
        uint32_t* kept_out = drop ? kept : nullptr;
        if (dropping && p == 1) n_device = kept;   // (the launches stay sized for n: surplus workgroups leave at once)
        const bool with_duty = p == 0 && first_count_duty != nullptr;
        const bool precounted = p == 0 && precount_blocks != 0u;   // (the keys' writer left the counts: no count kernel, a folding scan)
        if (!precounted)
            hipLaunchKernelGGL(radix_count_kernel, dim3(tiles + (with_duty ? (uint32_t)first_count_duty->groups : 0u)), dim3(kThreads), 0, stream, kin, n, shift, mask, counts,
                               tiles_pad, n_device, drop, dkey, with_duty ? *first_count_duty : TallyDuty{});
        if (p == 0 && after_first_count != nullptr) {
            const hipError_t e = hipEventRecord(after_first_count, stream);
            if (e != hipSuccess) return e;
        }
        if (precounted)
            hipLaunchKernelGGL(radix_scan_kernel<true>, dim3(256), dim3(kThreads), 0, stream, counts, tiles, tiles_pad, totals, n_device, precount_blocks);
        else
            hipLaunchKernelGGL(radix_scan_kernel<false>, dim3(256), dim3(kThreads), 0, stream, counts, tiles, tiles_pad, totals, n_device, 0u);
#define GSR_RADIX_LAUNCH(I, K, A)                                                                                      \
    hipLaunchKernelGGL((radix_scatter_kernel<I, K, A>), dim3(tiles), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, \
                       shift, mask, counts, tiles_pad, totals, n_device, dkey, kept_out)
#define GSR_RADIX_LAUNCH_IK(A)                      \
    do {                                            \
        if (iota && keys_out) GSR_RADIX_LAUNCH(true, true, A);   \
        else if (iota) GSR_RADIX_LAUNCH(true, false, A);         \
        else if (keys_out) GSR_RADIX_LAUNCH(false, true, A);     \
        else GSR_RADIX_LAUNCH(false, false, A);                  \
    } while (0)
        // (64 lanes on a handful of LDS counters serialise the adds: measured 19.8 us against 16.1 us with ballots on the top
        // byte of C3's depth keys, while on evenly spread digits the adds win 13.5 : 18.4)
        const int form = few_top_digits && p + 1 == passes && rank_form == 1 ? 0 : rank_form;
        if (form == 1) GSR_RADIX_LAUNCH_IK(1);
        else if (form == 2) GSR_RADIX_LAUNCH_IK(2);
        else GSR_RADIX_LAUNCH_IK(0);
#undef GSR_RADIX_LAUNCH_IK
#undef GSR_RADIX_LAUNCH
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *keys_sorted = kin;
    *vals_sorted = vin;
    return hipGetLastError();
}

} // namespace gsr
