// CUB shim (TEST INFRASTRUCTURE, see cuda_shim.h).  CUB is not vendored in the reference; the two
// entry points it uses are restated from their published contracts:
//   DeviceScan::InclusiveSum      out[i] = in[0] + ... + in[i]
//   DeviceRadixSort::SortPairs    stable ascending sort of (key, value) pairs on key bits [begin_bit, end_bit)
// Both follow CUB's two-phase protocol: a null temp-storage pointer only reports the size.
#pragma once
#include "../cuda_shim.h"
#include <algorithm>
#include <numeric>
#include <vector>
namespace cub {
struct DeviceScan {
    template <typename In, typename Out>
    static cudaError_t InclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n) {
        if (tmp == nullptr) { bytes = 16; return cudaSuccess; }
        if (n > 0) { auto acc = in[0]; out[0] = acc; for (int i = 1; i < n; ++i) { acc = acc + in[i]; out[i] = acc; } }
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    template <typename K, typename V>
    static cudaError_t SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n,
                                 int begin_bit = 0, int end_bit = sizeof(K) * 8) {
        if (tmp == nullptr) { bytes = 16; return cudaSuccess; }
        const int nb = end_bit - begin_bit;
        const K mask = nb >= (int)(sizeof(K) * 8) ? ~K(0) : (K)(((K(1) << nb) - 1) << begin_bit);
        std::vector<int> order(n > 0 ? n : 0);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (int i = 0; i < n; ++i) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
        return cudaSuccess;
    }
};
} // namespace cub
