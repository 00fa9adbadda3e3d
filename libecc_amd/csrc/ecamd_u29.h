// libecc_amd/csrc/ecamd_u29.h -- unsaturated radix-2^29 prime-field arithmetic with lazy
// reduction and compile-time bound tracking (the fast path of the scalar multiplication).
//
// Why not 32/64-bit saturated limbs (the reference's layout, nn/nn.h:42-45)?  Measured on
// MI355X (profiles/ubench_r1.json, cycles per wave64 instruction per SIMD): v_mad_u64_u32 5.3,
// any VOP3-encoded integer op (v_addc_co_u32 with an SGPR carry, v_add3, v_alignbit, ...) 4.4,
// VOP2 e32 ops (v_add_u32, v_and_b32, v_lshrrev_b32) 2.5 -- and a VALU-written carry needs 2
// wait states before the next VALU may read it.  With saturated limbs every 32x32 product
// costs a v_mad_u64_u32 PLUS a carry instruction (9.7 cycles) and every modular add/sub is a
// serial carry chain.  With 29-bit limbs a column of <= 9 products (+ 4 reduction products)
// fits a 64-bit accumulator without any carry handling: one v_mad_u64_u32 per product, and
// field add/sub become 9 independent v_add_u32.
//
// Representation: value = sum l[i] * 2^(29 i), i < NL, limbs are u32 and may exceed 29 bits
// ("loose"); values are only congruent mod p and may exceed p ("lazy").  Montgomery radix
// R = 2^(29 NL).  Every element type F<LB, TB, VB> carries compile-time bounds
//     LB  >= every limb l[0..NL-2]          TB >= top limb l[NL-1]
//     VB  >= value in units of p/16 (so VB = 32 means value < 2p)
// and every operation static_asserts the preconditions that make it overflow-free and derives
// the bounds of its result, so a formula that could overflow in ANY input does not compile.
//
// P-256 (NL = 9, R = 2^261): p = -1 mod 2^87, so the Montgomery quotient digit of column k is
// just the low 29 bits of the column, adding m_k * p is "-m_k + m_k (p + 1)" and p + 1 has
// only four non-zero digits (limbs 3, 6, 7, 8): a multiplication is 81 product MADs + 36
// reduction MADs, a squaring 45 + 36.
//
// Replaces nn_mul_redc1 / nn_mod_add / nn_mod_sub / fp_inv (nn/nn_mul_redc1.c:124-218,
// nn/nn_add.c:337,398, fp/fp_mul.c:51-68 in /root/reference/src) on the fast path; results are
// converted back to the canonical residue in [0, p) before they leave the kernel, so every
// observable byte is identical to the reference's.
#pragma once
#include <stdint.h>
#include <utility>
#include "ecamd_madchain.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define U29_FN __device__ __forceinline__
#define U29_NOINLINE __device__ __noinline__
#else
#define U29_FN inline
#define U29_NOINLINE inline
#endif

namespace u29 {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int W = 29;
constexpr u32 MASK = (1u << W) - 1;

// ------------------------------------------------------------------------------------------
// P-256 constants in radix 2^29 (all derivable from p = 2^256 - 2^224 + 2^192 + 2^96 - 1;
// tools/u29_consts.py prints them and tests/test_u29_host.py re-derives and checks them)
// ------------------------------------------------------------------------------------------
struct P256 {
	static constexpr int NL = 9;
	static constexpr int TOPBITS = 256 - 29 * 8;  // bits of p in the top limb (24)
	// digits of p
	static constexpr u32 P[9] = {0x1fffffff, 0x1fffffff, 0x1fffffff, 0x000001ff, 0x00000000,
				     0x00000000, 0x00040000, 0x1fe00000, 0x00ffffff};
	// digits of p + 1 (limbs 0..2 are zero): the reduction multipliers
	static constexpr u32 Q3 = 0x00000200, Q6 = 0x00040000, Q7 = 0x1fe00000, Q8 = 0x00ffffff;
	// delta = 2^256 mod p = 2^224 - 2^192 - 2^96 + 1
	static constexpr u32 D[9] = {0x00000001, 0x00000000, 0x00000000, 0x1ffffe00, 0x1fffffff,
				     0x1fffffff, 0x1ffbffff, 0x001fffff, 0x00000000};
};

// ------------------------------------------------------------------------------------------
// bound-tracked element
// ------------------------------------------------------------------------------------------
constexpr u64 cmin(u64 a, u64 b) { return a < b ? a : b; }
constexpr u64 cmax(u64 a, u64 b) { return a > b ? a : b; }
// top-limb bound implied by the value bound: l[8] * 2^232 <= value < (VB/16) p < (VB/16) 2^256
constexpr u64 top_from_vb(u64 vb) { return ((vb << 24) + 15) / 16; }

template <u64 LB_, u64 TB_, u64 VB_> struct F {
	static constexpr u64 LB = LB_;
	static constexpr u64 TB = cmin(TB_, top_from_vb(VB_));
	static constexpr u64 VB = VB_;
	static_assert(LB_ < (1ull << 32) && TB < (1ull << 32), "limb does not fit 32 bits");
	u32 l[9];
};

// canonical-ish classes
typedef F<MASK, (2ull << 24), 32> Fmul;     // what a multiplication returns when VB_out <= 2p
typedef F<MASK, (1ull << 24), 16> Fcanon;   // value < p, exact digits

// re-type with weaker (larger) bounds -- always sound
template <class T, class S> U29_FN T weaken(const S &s)
{
	static_assert(T::LB >= S::LB && T::TB >= S::TB && T::VB >= S::VB, "weaken() must not tighten bounds");
	T r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = s.l[i];
	}
	return r;
}

// ------------------------------------------------------------------------------------------
// multiplication / squaring (Montgomery, R = 2^261), product scanning, one 64-bit accumulator
// ------------------------------------------------------------------------------------------
constexpr u64 mul_vb(u64 va, u64 vb)
{
	// out < A*B/R + p, p/R < 2^-5:  in p/16 units: va*vb/(16*32) + 16, rounded up
	return (va * vb + 511) / 512 + 16;
}

template <u64 VBO> struct MulOut {
	typedef F<MASK, top_from_vb(VBO), VBO> type;
};

// The reduction multipliers Q3 = 2^9 and Q6 = 2^18 are powers of two; left to itself hipcc
// strength-reduces "acc += m * 2^9" into a 64-bit shift plus a 64-bit add (two VOP3 ops, ~9
// cycles) where one v_mad_u64_u32 (5.3 cycles) does the job.  Passing the constants through
// an empty asm makes them opaque SGPR values.
#if defined(__HIPCC__)
// keeps hipcc from re-associating "(acc >> 29) + products" into a separate product chain that
// is merged back with a 64-bit add (15 extra v_lshl_add_u64 per multiplication)
#define U29_PIN(acc) asm("" : "+v"(acc))
#else
#define U29_PIN(acc) (void)0
#endif
#if defined(__HIPCC__)
#define U29_OPAQUE_Q() \
	u32 q3 = P256::Q3, q6 = P256::Q6, q7 = P256::Q7, q8 = P256::Q8; \
	asm volatile("" : "+s"(q3), "+s"(q6), "+s"(q7), "+s"(q8))
#else
#define U29_OPAQUE_Q() const u32 q3 = P256::Q3, q6 = P256::Q6, q7 = P256::Q7, q8 = P256::Q8
#endif

// acc += a * b.  With U29_ASM_MAD every product is an explicit v_mad_u64_u32 chained on the
// column accumulator (hipcc otherwise splits a column into several chains and merges them with
// 64-bit adds); the carry-out operand is a dead SGPR pair.
#if defined(__HIPCC__) && defined(U29_ASM_MAD)
#define U29_MAD_VV(acc, a, b) \
	do { u64 dead_; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(dead_) : "v"(a), "v"(b)); } while (0)
#define U29_MAD_VS(acc, a, b) \
	do { u64 dead_; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(dead_) : "v"(a), "s"(b)); } while (0)
#else
#define U29_MAD_VV(acc, a, b) acc += (u64)(a) * (b)
#define U29_MAD_VS(acc, a, b) acc += (u64)(a) * (b)
#endif

// raw kernels on plain arrays (bounds are checked by the typed wrappers below).  Columns are compile-time
// indexed so that each one's products and reduction terms go out as asm statements of up to twelve MADs
// (ecamd_madchain.h: one padding s_nop per statement instead of one per MAD).
template <bool SQR, int K_> struct Col9 {
	static constexpr int LO = (K_ < 9) ? 0 : (K_ - 8);
	static constexpr int HI = (K_ < 9) ? K_ : 8;
	static constexpr int HALF = K_ / 2;
	static constexpr int NPROD = SQR ? ((HALF >= LO) ? (HALF - LO + 1) : 0) : (HI - LO + 1);
	// reduction products m_i * q_j, i + j = k, j in {3, 6, 7, 8}, 0 <= i <= 8
	static constexpr bool R3 = (K_ - 3 >= 0 && K_ - 3 <= 8), R6 = (K_ - 6 >= 0 && K_ - 6 <= 8);
	static constexpr bool R7 = (K_ - 7 >= 0 && K_ - 7 <= 8), R8 = (K_ - 8 >= 0 && K_ - 8 <= 8);
	static constexpr int NRED = (R3 ? 1 : 0) + (R6 ? 1 : 0) + (R7 ? 1 : 0) + (R8 ? 1 : 0);
};

template <bool SQR, int K_>
U29_FN void mul_col9(u64 &acc, u32 *m, u32 *r, const u32 *a, const u32 *b, const u32 *a2, u32 q3, u32 q6, u32 q7, u32 q8)
{
	typedef Col9<SQR, K_> C;
	u64 unused = 0;
	if constexpr (C::NPROD > 0) {
		u32 x[C::NPROD], y[C::NPROD];
#pragma unroll
		for (int n = 0; n < C::NPROD; n++) {
			const int i = C::LO + n, j = K_ - i;
			x[n] = a[i];
			y[n] = !SQR ? b[j] : (i < j ? a2[j] : a[i]);
		}
		ecamd_mad_chain<C::NPROD, false, false>(acc, unused, x, y);
	}
	if constexpr (C::NRED > 0) {
		u32 x[C::NRED], y[C::NRED];
		int n = 0;
		if (C::R3) { x[n] = m[K_ - 3]; y[n] = q3; n++; }
		if (C::R6) { x[n] = m[K_ - 6]; y[n] = q6; n++; }
		if (C::R7) { x[n] = m[K_ - 7]; y[n] = q7; n++; }
		if (C::R8) { x[n] = m[K_ - 8]; y[n] = q8; n++; }
		ecamd_mad_chain<C::NRED, false, true>(acc, unused, x, y);
	}
	if constexpr (K_ < 9) {
		m[K_] = (u32)acc & MASK;  // quotient digit (mpinv = 1); "- m_k" clears the digit
	} else {
		r[K_ - 9] = (u32)acc & MASK;
	}
#if defined(__HIPCC__) && defined(U29_SHIFT_PAIR)
	{
		// the 64-bit shift as v_alignbit_b32 + v_lshrrev_b32 instead of v_lshrrev_b64 (A/B: tools/build_variant.py)
		u32 lo_ = (u32)acc, hi_ = (u32)(acc >> 32);
		lo_ = __builtin_amdgcn_alignbit(hi_, lo_, W);
		hi_ >>= W;
		asm("" : "+v"(lo_), "+v"(hi_));
		acc = ((u64)hi_ << 32) | lo_;
	}
#else
	acc >>= W;
#endif
	// (no pin needed: the next column starts with an asm statement that takes acc as an operand)
}

template <bool SQR, int... Ks>
U29_FN void mul_cols9(u64 &acc, u32 *m, u32 *r, const u32 *a, const u32 *b, const u32 *a2, u32 q3, u32 q6, u32 q7, u32 q8,
		      std::integer_sequence<int, Ks...>)
{
	(mul_col9<SQR, Ks>(acc, m, r, a, b, a2, q3, q6, q7, q8), ...);
}

U29_FN void mul_raw(u32 *r, const u32 *a, const u32 *b)
{
	u32 m[9];
	U29_OPAQUE_Q();
	u64 acc = 0;
	mul_cols9<false>(acc, m, r, a, b, a, q3, q6, q7, q8, std::make_integer_sequence<int, 17>());
	r[8] = (u32)acc;
}

U29_FN void sqr_raw(u32 *r, const u32 *a)
{
	u32 m[9], a2[9];
	U29_OPAQUE_Q();
#pragma unroll
	for (int i = 0; i < 9; i++) {
		a2[i] = a[i] << 1;
	}
	u64 acc = 0;
	mul_cols9<true>(acc, m, r, a, a, a2, q3, q6, q7, q8, std::make_integer_sequence<int, 17>());
	r[8] = (u32)acc;
}

struct Raw9 {
	u32 l[9];
};
// Inlining the multiplier into the formulas is 25 % faster on MI355X than calling it (62.2 vs
// 49.5 M scalar-mults/s, profiles/r1b_*): hipcc interleaves the independent column chains of
// neighbouring multiplications, and the call ABI passes the second operand through scratch.
// Define U29_CALL_MUL to get the out-of-line variant back (smaller code, for A/B tests).
#if !defined(U29_CALL_MUL) && !defined(U29_INLINE_MUL)
#define U29_INLINE_MUL 1
#endif
#ifndef U29_INLINE_MUL
U29_NOINLINE Raw9 mul_call(Raw9 a, Raw9 b)
{
	Raw9 r;
	mul_raw(r.l, a.l, b.l);
	return r;
}
U29_NOINLINE Raw9 sqr_call(Raw9 a)
{
	Raw9 r;
	sqr_raw(r.l, a.l);
	return r;
}
#endif

// column bound: 9 products + 4 reduction products + carry-in must fit 64 bits
constexpr bool mul_fits(u64 la, u64 lb)
{
	// 9*la*lb + 4*2^58 + 2^36 < 2^64, evaluated without overflowing u64:
	// la*lb <= (2^64 - 2^60 - 2^36) / 9
	return (la == 0 || lb <= ((0xFFFFFFFFFFFFFFFFull - (1ull << 60) - (1ull << 36)) / 9) / la);
}

template <class A, class B> U29_FN typename MulOut<mul_vb(A::VB, B::VB)>::type mul(const A &a, const B &b)
{
	static_assert(mul_fits(cmax(A::LB, A::TB), cmax(B::LB, B::TB)), "mul: column accumulator could overflow");
	static_assert(mul_vb(A::VB, B::VB) <= 16 * 31, "mul: result value too large for the top limb");
	typename MulOut<mul_vb(A::VB, B::VB)>::type r;
#ifdef U29_INLINE_MUL
	mul_raw(r.l, a.l, b.l);
#else
	Raw9 x, y;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		x.l[i] = a.l[i];
		y.l[i] = b.l[i];
	}
	const Raw9 z = mul_call(x, y);
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = z.l[i];
	}
#endif
	return r;
}

template <class A> U29_FN typename MulOut<mul_vb(A::VB, A::VB)>::type sqr(const A &a)
{
	static_assert(mul_fits(cmax(A::LB, A::TB), cmax(A::LB, A::TB)), "sqr: column accumulator could overflow");
	static_assert(2 * cmax(A::LB, A::TB) < (1ull << 32), "sqr: doubled limb does not fit 32 bits");
	static_assert(mul_vb(A::VB, A::VB) <= 16 * 31, "sqr: result value too large for the top limb");
	typename MulOut<mul_vb(A::VB, A::VB)>::type r;
#ifdef U29_INLINE_MUL
	sqr_raw(r.l, a.l);
#else
	Raw9 x;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		x.l[i] = a.l[i];
	}
	const Raw9 z = sqr_call(x);
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = z.l[i];
	}
#endif
	return r;
}

// ------------------------------------------------------------------------------------------
// addition, subtraction (with a multiple of p as bias), small multiples
// ------------------------------------------------------------------------------------------
template <class A, class B> U29_FN F<A::LB + B::LB, A::TB + B::TB, A::VB + B::VB> add(const A &a, const B &b)
{
	F<A::LB + B::LB, A::TB + B::TB, A::VB + B::VB> r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = a.l[i] + b.l[i];
	}
	return r;
}

template <int K, class A> U29_FN F<K * A::LB, K * A::TB, K * A::VB> mul_small(const A &a)
{
	static_assert(K == 2 || K == 3 || K == 4 || K == 8, "small multiple");
	F<K * A::LB, K * A::TB, K * A::VB> r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = (K == 3) ? (a.l[i] + (a.l[i] << 1)) : (a.l[i] << (K == 2 ? 1 : (K == 4 ? 2 : 3)));
	}
	return r;
}

// bias(C) = C * p written with every limb below the top >= M - 4 where M = 2^(29+S):
//   digit_i of C*p, + M for i = 0, + M - M/2^29 for 0 < i < 8, top digit - M/2^29.
// C is a power of two <= 64 so C*p's digits are computed by shifting p's.
template <int LOGC, int S> struct Bias {
	static constexpr u64 C = 1ull << LOGC;
	static constexpr u64 M = 1ull << (W + S);
	static constexpr u64 BORROW = 1ull << S;
	static constexpr u64 digit(int i)
	{
		// digit i of p << LOGC
		u64 carry = 0, d = 0;
		for (int k = 0; k <= i; k++) {
			const u64 v = ((u64)P256::P[k] << LOGC) + carry;
			d = (k < 8) ? (v & MASK) : v;
			carry = v >> W;
		}
		return d;
	}
	static constexpr u64 limb(int i)
	{
		return (i == 0) ? digit(0) + M : (i < 8 ? digit(i) + M - BORROW : digit(8) - BORROW);
	}
	static constexpr u64 LOWMIN = M - BORROW;  // every limb 0..7 is >= this
	static constexpr u64 LOWMAX = M + MASK;
	static constexpr u64 TOP = limb(8);
	static_assert(digit(8) >= BORROW, "bias: top digit smaller than the borrow");
};

// a - b + C p   (C = 2^LOGC), limbs of the bias dominate the limbs of b
template <int LOGC, int S, class A, class B>
U29_FN F<A::LB + Bias<LOGC, S>::LOWMAX, A::TB + Bias<LOGC, S>::TOP, A::VB + 16 * Bias<LOGC, S>::C> sub(const A &a, const B &b)
{
	typedef Bias<LOGC, S> BS;
	static_assert(BS::LOWMIN >= B::LB, "sub: bias limbs do not dominate b");
	static_assert(BS::TOP >= B::TB, "sub: bias top limb does not dominate b");
	F<A::LB + BS::LOWMAX, A::TB + BS::TOP, A::VB + 16 * BS::C> r;
	constexpr u32 bl[9] = {(u32)BS::limb(0), (u32)BS::limb(1), (u32)BS::limb(2), (u32)BS::limb(3), (u32)BS::limb(4),
			       (u32)BS::limb(5), (u32)BS::limb(6), (u32)BS::limb(7), (u32)BS::limb(8)};
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = a.l[i] + (bl[i] - b.l[i]);
	}
	return r;
}

// ------------------------------------------------------------------------------------------
// limb normalisation (carry) and value reduction (fold)
// ------------------------------------------------------------------------------------------
// one parallel carry round: limbs 0..7 become < 2^29 + (LB >> 29); value unchanged
template <class A> U29_FN F<MASK + (A::LB >> W), A::TB + (A::LB >> W), A::VB> carry(const A &a)
{
	F<MASK + (A::LB >> W), A::TB + (A::LB >> W), A::VB> r;
	r.l[0] = a.l[0] & MASK;
#pragma unroll
	for (int i = 1; i < 8; i++) {
		r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> W);
	}
	r.l[8] = a.l[8] + (a.l[7] >> W);
	return r;
}

// value < 8p  ->  value < (1 + 2^-20) p ... well below 17/16 p, limbs normalised by a second
// carry round.  q = value >> 256 (from the top limb), value -= q 2^256, value += q delta.
template <class A> U29_FN F<MASK + 16, (1ull << 24) + 16, 17> fold(const A &a0)
{
	const auto a = carry(a0);
	typedef decltype(a) C;
	static_assert(C::LB <= MASK + 7, "fold: input limbs too loose");
	static_assert((C::TB >> 24) <= 7, "fold: value too large (q must be <= 7)");
	const u32 q = a.l[8] >> 24;
	F<(u64)MASK + 7 + 7ull * MASK, (1ull << 24), 17> t;  // <= 2^32 - 1
	t.l[0] = a.l[0] + q;  // D[0] = 1
	t.l[1] = a.l[1];
	t.l[2] = a.l[2];
	t.l[3] = a.l[3] + q * P256::D[3];
	t.l[4] = a.l[4] + q * P256::D[4];
	t.l[5] = a.l[5] + q * P256::D[5];
	t.l[6] = a.l[6] + q * P256::D[6];
	t.l[7] = a.l[7] + q * P256::D[7];
	t.l[8] = a.l[8] & 0xffffffu;
	const auto u = carry(t);
	return weaken<F<MASK + 16, (1ull << 24) + 16, 17>>(u);
}

// ------------------------------------------------------------------------------------------
// exact tests and canonical form
// ------------------------------------------------------------------------------------------
// x == 0 mod p for a multiplication result (limbs 0..7 are exact digits, value < 4p):
// compare against 0, p, 2p, 3p digit-wise
template <class A> U29_FN bool is_zero_mulout(const A &a)
{
	static_assert(A::LB == MASK, "is_zero_mulout needs exact low digits");
	static_assert(A::VB <= 64, "is_zero_mulout: value must be < 4p");
	u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
	constexpr u32 p1[9] = {P256::P[0], P256::P[1], P256::P[2], P256::P[3], P256::P[4], P256::P[5], P256::P[6], P256::P[7], P256::P[8]};
	constexpr u32 p2[9] = {(u32)Bias<1, 0>::digit(0), (u32)Bias<1, 0>::digit(1), (u32)Bias<1, 0>::digit(2), (u32)Bias<1, 0>::digit(3),
			       (u32)Bias<1, 0>::digit(4), (u32)Bias<1, 0>::digit(5), (u32)Bias<1, 0>::digit(6), (u32)Bias<1, 0>::digit(7),
			       (u32)Bias<1, 0>::digit(8)};
#pragma unroll
	for (int i = 0; i < 9; i++) {
		z0 |= a.l[i];
		z1 |= a.l[i] ^ p1[i];
		z2 |= a.l[i] ^ p2[i];
	}
	bool z = (z0 == 0) | (z1 == 0) | (z2 == 0);
	if (A::VB > 48) {
		// 3p = 2p + p: digits via one carry pass at compile time are awkward; compare a - 2p with p
		u32 d[9];
		u32 borrow = 0;
#pragma unroll
		for (int i = 0; i < 9; i++) {
			const u32 x = a.l[i] - p2[i] - borrow;
			borrow = (i < 8) ? (x >> 31) : 0;
			d[i] = (i < 8) ? (x & MASK) : x;
		}
#pragma unroll
		for (int i = 0; i < 9; i++) {
			z3 |= d[i] ^ p1[i];
		}
		z = z | (z3 == 0);
	}
	return z;
}

// multiplication result (value < 2p, exact low digits) -> canonical residue in [0, p)
template <class A> U29_FN Fcanon canonical(const A &a)
{
	static_assert(A::LB == MASK && A::VB <= 32, "canonical() needs a multiplication result < 2p");
	u32 d[9];
	u32 borrow = 0;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		const u32 x = a.l[i] - P256::P[i] - borrow;
		borrow = (x >> 31);  // limbs are < 2^30, so bit 31 of the wrapped difference is the borrow
		d[i] = (i < 8) ? (x & MASK) : x;
	}
	Fcanon r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = borrow ? a.l[i] : d[i];
	}
	return r;
}

// ------------------------------------------------------------------------------------------
// conversion between eight saturated 32-bit words (little-endian) and nine 29-bit limbs
// ------------------------------------------------------------------------------------------
U29_FN Fcanon from_words(const u32 *w)  // caller guarantees value < p
{
	Fcanon r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
		u32 x = w[wi] >> sh;
		if (sh > 3 && wi + 1 < 8) {
			x |= w[wi + 1] << (32 - sh);
		}
		r.l[i] = x & MASK;
	}
	return r;
}

U29_FN void to_words(u32 *w, const Fcanon &a)
{
#pragma unroll
	for (int wi = 0; wi < 8; wi++) {
		// word wi covers bits [32 wi, 32 wi + 32)
		const int bit = 32 * wi, li = bit / 29, sh = bit - 29 * li;
		u32 x = a.l[li] >> sh;
		if (li + 1 < 9) {
			x |= a.l[li + 1] << (29 - sh);
		}
		if (29 - sh + 29 < 32 && li + 2 < 9) {
			x |= a.l[li + 2] << (58 - sh);
		}
		w[wi] = x;
	}
}

}  // namespace u29
