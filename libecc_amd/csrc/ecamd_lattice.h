// libecc_amd/csrc/ecamd_lattice.h -- the truncated Euclidean algorithm behind the half-length scalars of the Ed25519 verification
// equation (k_ed_lat in ecamd_kernels.hip, which explains what it is for).  Plain 32-bit word arithmetic, one item per lane; also
// compiled for the host by tests/lattice_host_shim.cpp, where tests/test_lattice_host.py checks it against Python integers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LAT_FN static __device__ __forceinline__
#define LAT_ANY(x) __any(x)
#define LAT_CLZ(x) __clz(x)
#define LAT_ALIGNBIT(hi, lo, b) __builtin_amdgcn_alignbit((hi), (lo), (b))
#else
#define LAT_FN static inline
#define LAT_ANY(x) (x)
#define LAT_CLZ(x) __builtin_clz(x)
#define LAT_ALIGNBIT(hi, lo, b) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> ((b) & 31)))
#endif

typedef uint32_t lat_u32;
typedef uint64_t lat_u64;

LAT_FN lat_u32 lat_word_at(const lat_u32 *r, int w)
{
	lat_u32 x = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		x = (k == w) ? r[k] : x;
	}
	return x;
}
// bits sh .. sh + 63 of the 8-word number r
LAT_FN lat_u64 lat_extract64(const lat_u32 *r, int sh)
{
	const int w = sh >> 5, b = sh & 31;
	const lat_u32 a0 = lat_word_at(r, w), a1 = lat_word_at(r, w + 1), a2 = lat_word_at(r, w + 2);
	const lat_u32 lo = LAT_ALIGNBIT(a1, a0, (lat_u32)b), hi = LAT_ALIGNBIT(a2, a1, (lat_u32)b);
	return ((lat_u64)hi << 32) | lo;
}
// v[8], u[4], neg from (q, h); returns false when the loop did not get there within its iteration budget or u does not fit.
// The budget: a lower bound k of the quotient is taken from the leading 63 bits and capped at 2^32 - 1, so a partial quotient
// above 2^32 (probability about 2^-32 per step for a hash value h; h below 2^220 makes the first one that large) is worked off
// 2^32 multiples at a time and may run out of iterations -- such an item takes the full-length loop, nothing else changes.
// The Euclidean algorithm on 253-bit inputs stopped half way needs about 74 steps on average and at most 183.
#ifndef LAT_MAX_ITER
#define LAT_MAX_ITER 192
#endif
LAT_FN bool lat_reduce(const lat_u32 *q, const lat_u32 *h, lat_u32 *v, lat_u32 *u, bool *neg, int *iters = nullptr)
{
	lat_u32 r0[8], r1[8], t0[5], t1[5];
#pragma unroll
	for (int w = 0; w < 8; w++) {
		r0[w] = q[w];
		r1[w] = h[w];
	}
#pragma unroll
	for (int w = 0; w < 5; w++) {
		t0[w] = 0;
		t1[w] = (w == 0) ? 1u : 0u;
	}
	bool s1 = false;   // sign of t1 (t0 has the opposite one)
	bool small = false;
	int it = 0;
#pragma unroll 1
	for (; it < LAT_MAX_ITER; it++) {
		small = ((r1[7] | r1[6] | r1[5] | r1[4]) == 0u) && (r1[3] < 0x80000000u);
		if (!LAT_ANY(!small)) {
			break;
		}
		if (!small) {
			// bit length of r0 (>= 128 here, r0 >= r1 >= 2^127)
			int top = 3;
#pragma unroll
			for (int w = 4; w < 8; w++) {
				top = (r0[w] != 0u) ? w : top;
			}
			const int len = 32 * top + 32 - LAT_CLZ(lat_word_at(r0, top));
			const int sh = len - 63;
			const lat_u64 x0 = lat_extract64(r0, sh), x1 = lat_extract64(r1, sh);
			// a lower bound of floor(r0 / r1) in single precision: conversions and division err by less than 2^-21 relative,
			// the factor 1 - 2^-20 keeps the estimate below x0 / (x1 + 1) <= r0 / r1 (a 64-bit integer division would cost more
			// than the rest of the iteration; an estimate one too small only costs one more iteration with quotient 1)
			const float kf = ((float)x0 / (float)(x1 + 1)) * 0.99999904632568359375f;
			const lat_u32 k32 = (kf >= 4294967040.0f) ? 0xffffffffu : ((kf < 1.0f) ? 1u : (lat_u32)kf);   // r0 >= r1: at least one
			// r0 -= k r1 (no underflow), |t0| += k |t1|
			lat_u64 carry = 0;
			lat_u32 borrow = 0;
#pragma unroll
			for (int w = 0; w < 8; w++) {
				carry += (lat_u64)k32 * r1[w];
				const lat_u64 d = (lat_u64)r0[w] - (lat_u32)carry - borrow;
				r0[w] = (lat_u32)d;
				borrow = (lat_u32)(d >> 63);
				carry >>= 32;
			}
			carry = 0;
#pragma unroll
			for (int w = 0; w < 5; w++) {
				carry += (lat_u64)k32 * t1[w] + t0[w];
				t0[w] = (lat_u32)carry;
				carry >>= 32;
			}
			// swap when r0 < r1
			bool lt = false, decided = false;
#pragma unroll
			for (int w = 7; w >= 0; w--) {
				const bool ne = r0[w] != r1[w];
				lt = (!decided && ne) ? (r0[w] < r1[w]) : lt;
				decided = decided | ne;
			}
			if (lt) {
#pragma unroll
				for (int w = 0; w < 8; w++) {
					const lat_u32 x = r0[w];
					r0[w] = r1[w];
					r1[w] = x;
				}
#pragma unroll
				for (int w = 0; w < 5; w++) {
					const lat_u32 x = t0[w];
					t0[w] = t1[w];
					t1[w] = x;
				}
				s1 = !s1;
			}
		}
	}
#pragma unroll
	for (int w = 0; w < 8; w++) {
		v[w] = r1[w];
	}
#pragma unroll
	for (int w = 0; w < 4; w++) {
		u[w] = t1[w];
	}
	*neg = s1;
	if (iters) {
		*iters = it;
	}
	// (when the budget ran out the last test of r1 is stale: look again)
	small = ((r1[7] | r1[6] | r1[5] | r1[4]) == 0u) && (r1[3] < 0x80000000u);
	return small && t1[4] == 0u && (t1[0] | t1[1] | t1[2] | t1[3]) != 0u;
}

