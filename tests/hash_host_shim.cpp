// host build of libecc_amd/csrc/ecamd_hash.hip for tests/test_hash_host.py (test infrastructure)
#include <hip/hip_runtime.h>
thread_local dim3 blockIdx, threadIdx;
#define ECAMD_INTERNAL_H_HOST_STUB 1
#include "../libecc_amd/csrc/ecamd_hash.hip"
extern "C" int sha2_slots_host(int hash_type, const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride)
{
	return (int)ecamd_launch_sha2_slots(hash_type, slots, stride, n, out, out_stride, nullptr);
}
extern "C" int shake256_slots_host(const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride, uint32_t outlen)
{
	return (int)ecamd_launch_shake256_slots(slots, stride, n, out, out_stride, outlen, nullptr);
}
