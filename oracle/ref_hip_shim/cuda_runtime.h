// Test infrastructure (oracle/build_ref_hip.py): lets the REFERENCE's CUDA sources be compiled by hipcc for gfx950
// from where they lie under /root/reference, as a measuring stick on the same GPU -- never part of the product.
// The sources use a dozen runtime names; they are mapped onto HIP here.  Kernels, <<<>>> launches, cooperative
// groups and the device math functions are understood by hipcc as they are.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#ifndef __trap
#define __trap() __builtin_trap()
#endif
