// host stand-in for <hip/hip_runtime.h>, just enough to run libecc_amd/csrc/ecamd_hash.hip's kernels lane by lane on the CPU
// (tests/test_hash_host.py; test infrastructure)
#pragma once
#include <stdint.h>
#include <stddef.h>
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
static const hipError_t hipSuccess = 0, hipErrorInvalidValue = 1;
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
extern thread_local dim3 blockIdx, threadIdx;
#define __constant__ static const
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __builtin_amdgcn_alignbit(hi, lo, b) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> ((b) & 31)))
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                \
	do {                                                                       \
		for (unsigned bx_ = 0; bx_ < (grid).x; bx_++) {                    \
			for (unsigned tx_ = 0; tx_ < (block).x; tx_++) {           \
				blockIdx.x = bx_;                                  \
				threadIdx.x = tx_;                                 \
				kernel(__VA_ARGS__);                               \
			}                                                          \
		}                                                                  \
	} while (0)
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }   // lanes run one after the other here
