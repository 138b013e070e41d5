// cuda_shim.h -- just enough of the CUDA C++ surface for the reference's rasterizer sources
// (DGR/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu + auxiliary.h) to compile with g++ and
// run on the host, one CUDA thread at a time.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Nothing here is product code and nothing in the product
// includes it.  The reference sources are compiled from where they lie under /root/reference; this
// directory only supplies the headers they ask for (<cuda.h>, "cuda_runtime.h", <cub/cub.cuh>,
// <cooperative_groups.h>) and a tiny fiber runtime that gives __syncthreads() its meaning.
#pragma once

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define CUDA_VERSION 11080

struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct int2 { int x, y; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

// Built-in coordinates of the CUDA thread currently being executed (set by the runtime below).
extern dim3 gridDim, blockDim;
extern uint3 blockIdx, threadIdx;

// CUDA's integer / float min & max overload set (only what the sources use).
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
static inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
static inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline void __trap() { abort(); }

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "host shim"; }

namespace gsr_shim {
// Runs `body` once per CUDA thread of a grid x block launch.  Blocks run one after another; the
// threads of a block are ucontext fibers so that barrier() can suspend a thread until every live
// thread of the block has arrived.
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
int barrier(int predicate);  // returns how many arriving threads had predicate != 0
} // namespace gsr_shim

static inline void __syncthreads() { gsr_shim::barrier(0); }
static inline int __syncthreads_count(int pred) { return gsr_shim::barrier(pred); }
