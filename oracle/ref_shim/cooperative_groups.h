// cooperative_groups shim (TEST INFRASTRUCTURE, see cuda_shim.h): this_grid().thread_rank() for 1-D
// launches and the thread_block accessors the rasterizer sources use.
#pragma once
#include "cuda_shim.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const { return (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; }
};
static inline grid_group this_grid() { return grid_group(); }
struct thread_block {
    uint3 group_index() const { return blockIdx; }
    uint3 thread_index() const { return threadIdx; }
    unsigned int thread_rank() const { return (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x; }
    void sync() const { gsr_shim::barrier(0); }
};
static inline thread_block this_thread_block() { return thread_block(); }
} // namespace cooperative_groups
