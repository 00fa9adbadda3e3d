// host build of libecc_amd/csrc/ecamd_hash.hip for tests/test_hash_host.py (test infrastructure)
#include <hip/hip_runtime.h>
#include <cstring>
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
// the byte mover in front of the Schnorr-type multi-scalar form (round 6): run on the CPU lane by lane
extern "C" int schnorr_prep_host(const uint8_t *keys_aff, const uint8_t *kst, const uint8_t *sigs, uint8_t *slots, uint8_t *keys_out, uint8_t *s_out,
				 uint8_t *r_out, uint32_t *flag, uint32_t n, uint32_t clen, uint32_t qlen, uint32_t rlen, uint32_t stride, uint32_t x_off,
				 uint32_t even_y, const uint8_t *p_be)
{
	EcamdSchnorrPrepArgs A;
	memset(&A, 0, sizeof(A));
	A.keys_aff = keys_aff; A.kst = kst; A.sigs = sigs; A.slots = slots; A.keys_out = keys_out; A.s_out = s_out; A.r_out = r_out; A.flag = flag;
	A.n = n; A.clen = clen; A.qlen = qlen; A.rlen = rlen; A.stride = stride; A.x_off = x_off; A.even_y = even_y;
	memcpy(A.p_be, p_be, clen);
	return (int)ecamd_launch_schnorr_prep(A, nullptr);
}
