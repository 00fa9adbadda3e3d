// libecc_amd/csrc/ecamd_randmod.h -- nn_get_random_mod (nn/nn_rand.c:92-150) given its random bytes: the reference draws 2 * qlen bytes
// with get_random straight into the limb array of an nn (so they read as a little-endian integer on the little-endian hosts libecc and
// this library run on), reduces that modulo q' = q - 1 (nn_mod_notrim) and adds one: a value in [1, q - 1].  Here the 2 * qlen bytes are
// the caller's (the application's get_random stays on the host) and the reduction runs one item per lane, so that a
// signing or key-generation batch no longer spends a microsecond of a host thread per nonce in libecc's constant-time division
// (profiles/r4u_secret_half.md).  q' is even, which rules a Montgomery reduction out; a restoring binary division is 16 * qlen steps of
// one shift and one conditional subtraction over NW words -- about 12 k word operations at 256 bits, 4 % of the scalar multiplication
// that follows.  Also compiled for the host by tests/randmod_host_shim.cpp (tests/test_randmod_host.py: against Python integers).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RANDMOD_FN static __device__ __forceinline__
#else
#define RANDMOD_FN static inline
#endif

// out[NW] = (little-endian integer of raw[0 .. rawlen)) mod (q - 1) + 1, q[NW] odd and > 1 (little-endian 32-bit words)
template <int NW> RANDMOD_FN void randmod_words(uint32_t *out, const uint8_t *raw, int rawlen, const uint32_t *q)
{
	uint32_t m[NW], r[NW];
#pragma unroll
	for (int w = 0; w < NW; w++) {
		m[w] = q[w];
		r[w] = 0;
	}
	m[0] -= 1u;   // q is odd: no borrow
#pragma unroll 1
	for (int byte = rawlen - 1; byte >= 0; byte--) {
		const uint32_t v = raw[byte];
#pragma unroll 1
		for (int bit = 7; bit >= 0; bit--) {
			// r = 2 r + bit; r < m < 2^(32 NW) before, so the doubled value needs one more bit: `top`
			const uint32_t top = r[NW - 1] >> 31;
#pragma unroll
			for (int w = NW - 1; w > 0; w--) {
				r[w] = (r[w] << 1) | (r[w - 1] >> 31);
			}
			r[0] = (r[0] << 1) | ((v >> bit) & 1u);
			// d = r - m; kept when the doubled value was >= m (top set, or no borrow)
			uint32_t d[NW], borrow = 0;
#pragma unroll
			for (int w = 0; w < NW; w++) {
				const uint64_t x = (uint64_t)r[w] - m[w] - borrow;
				d[w] = (uint32_t)x;
				borrow = (uint32_t)(x >> 63);
			}
			const bool ge = (top != 0) | (borrow == 0);
#pragma unroll
			for (int w = 0; w < NW; w++) {
				r[w] = ge ? d[w] : r[w];
			}
		}
	}
	uint32_t c = 1;   // + 1: r <= q - 2, no overflow
#pragma unroll
	for (int w = 0; w < NW; w++) {
		const uint64_t x = (uint64_t)r[w] + c;
		out[w] = (uint32_t)x;
		c = (uint32_t)(x >> 32);
	}
}
