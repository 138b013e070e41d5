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
//   2. stable rank inside the wave by digit matching: 8 ballots give the mask of lanes holding the same digit,
//      popcount below the lane is the rank, the highest peer bumps the wave's LDS counter of that digit;
//   3. counters are turned into (wave, digit) offsets, a 256-wide scan gives the digit segments of the tile;
//   4. pairs are parked in LDS at their in-tile position and written out in that order, so every digit
//      segment is a contiguous, coalesced run in HBM.
#include "gsr_internal.h"

namespace gsr {
namespace {

constexpr int kHistThreads = 256;
constexpr int kHistCopies = 8;  // sub-histograms per place: lanes L and L+8 share one (skewed digits serialise LDS atomics)
constexpr uint32_t kAggregate = 1u << 30, kPrefix = 2u << 30, kCountMask = (1u << 30) - 1u;

#ifdef GSR_RADIX_TRACE  // scripts/ubench/radix_trace.hip: per-workgroup phase timestamps (100 MHz wall clock)
__device__ unsigned long long* g_radix_trace = nullptr;
#define GSR_TRACE(slot) do { if (tid == 0 && g_radix_trace) g_radix_trace[(size_t)block * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define GSR_TRACE(slot) do { } while (0)
#endif

__device__ __forceinline__ uint32_t load_state(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_state(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive sum of one value per digit, held by lanes 0..255 (waves 0..3; every wave of the workgroup calls).
__device__ __forceinline__ uint32_t digits_exclusive_sum(uint32_t v, uint32_t* scratch /*4 words of LDS*/, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = wave_inclusive_sum(v, lane);
    if (lane == 63 && wave < 4) scratch[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) before += scratch[w];
    __syncthreads();
    return before + incl - v;
}

// ------------------------------------------------------------------------------------------------
// Digit counts of every place in one pass over the keys, and the zero fill of the look-back words.
// hist[place * 256 + digit] must be zero on entry (the forward call's counter memset).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kHistThreads) radix_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                      int places, int bits, uint32_t* __restrict__ hist,
                                                                      uint32_t* __restrict__ zero_words,
                                                                      uint32_t num_zero_words) {
    __shared__ uint32_t s_hist[4 * kHistCopies * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < places * kHistCopies * 256; i += kHistThreads) s_hist[i] = 0u;
    __syncthreads();
    const uint32_t stride = gridDim.x * kHistThreads;
    for (uint32_t i = blockIdx.x * kHistThreads + tid; i < num_zero_words; i += stride) zero_words[i] = 0u;
    uint32_t* mine = s_hist + (tid & (kHistCopies - 1)) * 256;
    auto count = [&](uint32_t k) {
        for (int p = 0; p < places; ++p) {
            const int width = min(8, bits - 8 * p);
            atomicAdd(&mine[p * kHistCopies * 256 + ((k >> (8 * p)) & ((1u << width) - 1u))], 1u);
        }
    };
    const uint32_t n4 = n / 4u;  // 16-byte loads: the key arrays are 256-byte aligned sub-arrays of an arena
    const uint4* keys4 = reinterpret_cast<const uint4*>(keys);
    for (uint32_t i = blockIdx.x * kHistThreads + tid; i < n4; i += stride) {
        const uint4 k = keys4[i];
        count(k.x); count(k.y); count(k.z); count(k.w);
    }
    if (blockIdx.x == 0 && tid < (int)(n - 4u * n4)) count(keys[4u * n4 + tid]);
    __syncthreads();
    for (int i = tid; i < places * 256; i += kHistThreads) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < kHistCopies; ++j) c += s_hist[((i >> 8) * kHistCopies + j) * 256 + (i & 255)];
        if (c != 0u) atomicAdd(&hist[i], c);
    }
}

// ------------------------------------------------------------------------------------------------
// One pass.  kThreads x kItems pairs per workgroup; kBatch look-back loads in flight per lane.
// kIota: payloads are the item indices (first pass of the depth sort: no payload read).
// kKeysOut: false on a last pass whose sorted keys nobody reads.
// ------------------------------------------------------------------------------------------------
template <int kThreads, int kItems, int kBatch, bool kIota, bool kKeysOut>
__global__ void __launch_bounds__(kThreads) radix_pass_kernel(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                             uint32_t digit_mask, const uint32_t* __restrict__ hist,
                                                             uint32_t* __restrict__ states,
                                                             uint32_t* __restrict__ ticket) {
    constexpr int kTileItems = kThreads * kItems;
    constexpr int kWaves = kThreads / 64;
    static_assert(kThreads >= 256 && kThreads % 64 == 0, "one lane per digit");
    __shared__ uint32_t s_keys[kTileItems];
    __shared__ uint32_t s_vals[kTileItems];
    __shared__ uint32_t s_count[kWaves][256];  // per-wave digit counts, then per-wave offsets inside the segment
    __shared__ uint32_t s_seg_start[256];      // first in-tile position of each digit segment
    __shared__ uint32_t s_dst_base[256];       // global position of the segment minus s_seg_start (mod 2^32)
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_ticket;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
    for (int i = tid; i < kWaves * 256; i += kThreads) (&s_count[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t block = s_ticket;
    const uint32_t tile_base = block * (uint32_t)kTileItems;
    const uint32_t tile_n = min((uint32_t)kTileItems, n - tile_base);
    GSR_TRACE(0);

    // 1. load (wave-striped)
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

    // 2. stable rank inside the wave
    uint32_t rank[kItems];
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t* my_count = s_count[wave];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const bool valid = first + 64u * i < tile_n;
        const uint32_t d = (key[i] >> shift) & digit_mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool set = (d >> b) & 1u;
            const unsigned long long with_bit = __ballot(set);
            peers &= set ? with_bit : ~with_bit;
        }
        const uint32_t before = my_count[d];  // every peer reads the same word before the bump below
        rank[i] = before + (uint32_t)__popcll(peers & below);
        if (valid && (peers >> lane) == 1ull) my_count[d] = before + (uint32_t)__popcll(peers);  // highest peer
    }
    __syncthreads();
    GSR_TRACE(1);

    // 3. digit d is lane d's from here on (lanes 256.. of a larger workgroup idle through steps 3 and 4)
    uint32_t total = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = s_count[w][tid];
            s_count[w][tid] = total;
            total += c;
        }
    }
    const uint32_t seg_start = digits_exclusive_sum(total, s_scan, tid);
    const uint32_t digit_start = digits_exclusive_sum(tid < 256 ? hist[tid] : 0u, s_scan, tid);  // global start of the digit
    GSR_TRACE(2);

    // 4. decoupled look-back over this digit's words
    if (tid < 256) {
        uint32_t* mine = states + (size_t)block * 256u + tid;
        uint32_t earlier = 0;
        if (block == 0u) {
            store_state(mine, kPrefix | total);
        } else {
            store_state(mine, kAggregate | total);
            int p = (int)block - 1;
            bool found = false;
            while (!found) {
                uint32_t s[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j)
                    s[j] = p - j >= 0 ? load_state(states + (size_t)(p - j) * 256u + tid) : kPrefix;
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    if (!found) {
                        while ((s[j] >> 30) == 0u) {
                            __builtin_amdgcn_s_sleep(1);
                            s[j] = load_state(states + (size_t)(p - j) * 256u + tid);
                        }
                        earlier += s[j] & kCountMask;
                        found = (s[j] >> 30) == 2u;
                    }
                }
                p -= kBatch;
            }
            store_state(mine, kPrefix | (earlier + total));
        }
        s_seg_start[tid] = seg_start;
        s_dst_base[tid] = digit_start + earlier - seg_start;
    }
    __syncthreads();
    GSR_TRACE(3);

    // 5. park in tile order, then stream out
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        if (first + 64u * i < tile_n) {
            const uint32_t d = (key[i] >> shift) & digit_mask;
            const uint32_t pos = s_seg_start[d] + s_count[wave][d] + rank[i];
            s_keys[pos] = key[i];
            s_vals[pos] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const uint32_t p = (uint32_t)(j * kThreads + tid);
        if (p < tile_n) {
            const uint32_t k = s_keys[p];
            const uint32_t dst = s_dst_base[(k >> shift) & digit_mask] + p;
            if (kKeysOut) keys_out[dst] = k;
            vals_out[dst] = s_vals[p];
        }
    }
    GSR_TRACE(4);
}

template <int kThreads, int kItems, int kBatch>
hipError_t run_passes(const RadixScratch& scratch, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt, uint32_t* vals,
                      uint32_t* vals_alt, bool iota_payload, bool want_sorted_keys, uint32_t** keys_sorted,
                      uint32_t** vals_sorted, hipStream_t stream) {
    constexpr uint32_t kTileItems = kThreads * kItems;
    static_assert(kTileItems >= kRadixTile, "gsr_internal.h sizes the look-back words with kRadixTile");
    const int passes = (bits + 7) / 8;
    const uint32_t blocks = (n + kTileItems - 1) / kTileItems;
    uint32_t *kin = keys, *kout = keys_alt, *vin = vals, *vout = vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int width = bits - 8 * p < 8 ? bits - 8 * p : 8;
        const uint32_t mask = (1u << width) - 1u;
        const uint32_t* hist = scratch.hist + 256 * p;
        uint32_t* states = scratch.states + (size_t)blocks * 256u * p;
        uint32_t* ticket = scratch.tickets + p;
        const bool iota = iota_payload && p == 0;
        const bool keys_out = want_sorted_keys || p + 1 < passes;
#define GSR_RADIX_LAUNCH(I, K)                                                                                  \
    hipLaunchKernelGGL((radix_pass_kernel<kThreads, kItems, kBatch, I, K>), dim3(blocks), dim3(kThreads), 0, stream, \
                       kin, vin, kout, vout, n, 8 * p, mask, hist, states, ticket)
        if (iota && keys_out) GSR_RADIX_LAUNCH(true, true);
        else if (iota) GSR_RADIX_LAUNCH(true, false);
        else if (keys_out) GSR_RADIX_LAUNCH(false, true);
        else GSR_RADIX_LAUNCH(false, false);
#undef GSR_RADIX_LAUNCH
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *keys_sorted = kin;
    *vals_sorted = vin;
    return hipGetLastError();
}

} // namespace

int g_radix_config = 0;  // tuning knob (scripts/ubench/radix_trace.hip): workgroup shape / look-back batch

size_t radix_state_words(uint32_t n, int bits) {
    const size_t blocks = ((size_t)n + kRadixTile - 1) / kRadixTile;
    const size_t passes = (size_t)((bits + 7) / 8);
    return blocks * 256u * passes;
}

hipError_t radix_sort_pairs(const RadixScratch& scratch, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                            uint32_t* vals, uint32_t* vals_alt, bool iota_payload, bool want_sorted_keys,
                            uint32_t** keys_sorted, uint32_t** vals_sorted, hipStream_t stream) {
    *keys_sorted = keys;
    *vals_sorted = vals;
    if (n == 0 || bits <= 0) return hipSuccess;
    if (n > kCountMask || bits > 32) return hipErrorInvalidValue;
    const int passes = (bits + 7) / 8;
    const uint32_t hist_blocks = (n / 4u + kHistThreads * 4 - 1) / (kHistThreads * 4) + 1u;
    hipLaunchKernelGGL(radix_histogram_kernel, dim3(hist_blocks < 1024u ? hist_blocks : 1024u), dim3(kHistThreads), 0,
                       stream, keys, n, passes, bits, scratch.hist, scratch.states,
                       (uint32_t)radix_state_words(n, bits) + scratch.extra_zero_words);
#define GSR_RADIX_RUN(T, I, B) \
    run_passes<T, I, B>(scratch, n, bits, keys, keys_alt, vals, vals_alt, iota_payload, want_sorted_keys, keys_sorted, vals_sorted, stream)
    switch (g_radix_config) {
        case 1: return GSR_RADIX_RUN(256, 16, 16);
        case 2: return GSR_RADIX_RUN(512, 8, 16);
        case 3: return GSR_RADIX_RUN(1024, 4, 16);
        case 4: return GSR_RADIX_RUN(512, 16, 16);
        case 5: return GSR_RADIX_RUN(1024, 8, 16);
        case 6: return GSR_RADIX_RUN(256, 8, 16);
        case 7: return GSR_RADIX_RUN(1024, 8, 4);
        default: return GSR_RADIX_RUN(256, 16, 4);
    }
#undef GSR_RADIX_RUN
}

} // namespace gsr
