// gsr_sort.hip -- device-wide scan / radix-sort stages of the forward rasterizer.
//
// Replaces the reference's two CUB calls (DGR/cuda_rasterizer/rasterizer_impl.cu:278
// cub::DeviceScan::InclusiveSum and :304-309 cub::DeviceRadixSort::SortPairs on 64-bit keys).
// The ordering the reference obtains from ONE stable sort of (tile << 32 | depth bits) keys over
// all pairs is obtained here from TWO much cheaper sorts:
//   1. depth_sort : P 32-bit depth keys (+ Gaussian ids)          4 radix passes over P items
//   2. tile_sort  : num_rendered 32-bit tile keys (+ ids), stable ceil(bits/8) passes (2 at 1080p)
// Pairs are emitted in depth order, and a stable sort by tile id keeps that order inside each
// tile, so POINT_LIST is the same permutation the reference computes (6 passes of 12-byte pairs).
//
// This round the passes themselves are rocPRIM's (AMD's own wave64-tuned onesweep radix sort and
// decoupled-lookback scan, header-only under /opt/rocm/include); the hand-written kernels are in
// gsr_kernels.hip.
#include "gsr_internal.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace gsr {

namespace {
struct TilesInOrder {
    const SplatBin* bins;
    const uint32_t* order;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return bins[order[k]].count; }
};
} // namespace

hipError_t depth_sort_temp_bytes(int P, size_t* temp_bytes) {
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    *temp_bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *temp_bytes, k, v, (unsigned int)P, 0u, 32u);
}

hipError_t depth_sort(void* temp, size_t temp_bytes, int P, uint32_t* keys, uint32_t* keys_alt, uint32_t* ids,
                      uint32_t* ids_alt, uint32_t** keys_sorted, uint32_t** ids_sorted, hipStream_t stream) {
    rocprim::double_buffer<uint32_t> k(keys, keys_alt), v(ids, ids_alt);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, (unsigned int)P, 0u, 32u, stream);
    *keys_sorted = k.current();
    *ids_sorted = v.current();
    return e;
}

hipError_t scan_temp_bytes(int P, size_t* temp_bytes) {
    *temp_bytes = 0;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u),
                                               TilesInOrder{nullptr, nullptr});
    return rocprim::inclusive_scan(nullptr, *temp_bytes, in, (uint32_t*)nullptr, (size_t)P,
                                   rocprim::plus<uint32_t>());
}

hipError_t scan_tiles_in_order(void* temp, size_t temp_bytes, int P, const SplatBin* bins, const uint32_t* order,
                               uint32_t* offsets, hipStream_t stream) {
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u),
                                               TilesInOrder{bins, order});
    return rocprim::inclusive_scan(temp, temp_bytes, in, offsets, (size_t)P, rocprim::plus<uint32_t>(), stream);
}

hipError_t tile_sort_temp_bytes(uint32_t n, size_t* temp_bytes) {
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    *temp_bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *temp_bytes, k, v, (unsigned int)n, 0u, 32u);
}

hipError_t tile_sort(void* temp, size_t temp_bytes, uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt,
                     uint32_t* vals, uint32_t* vals_alt, uint32_t** keys_sorted, uint32_t** vals_sorted,
                     hipStream_t stream) {
    rocprim::double_buffer<uint32_t> k(keys, keys_alt), v(vals, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, (unsigned int)n, 0u, (unsigned int)bits, stream);
    *keys_sorted = k.current();
    *vals_sorted = v.current();
    return e;
}

} // namespace gsr
