// libecc_amd/csrc/ecamd_u29g.h -- radix-2^29 lazy Montgomery arithmetic for ANY odd prime,
// with compile-time bound tracking (generalisation of ecamd_u29.h, which stays the
// hand-specialised secp256r1 path).
//
// Same idea: limbs of 29 bits in u32, 64-bit column accumulators, one v_mad_u64_u32 per
// product and no carry instructions.  Differences to the P-256 special case:
//   * dense Montgomery reduction: column k adds m_i * p_(k-i) for every digit of p (NL^2 MADs)
//     and m_k = (acc * mpinv) mod 2^29 with a per-curve mpinv;
//   * a "headroom limb": NL = ceil((|p| + 16) / 29), so R = 2^(29 NL) >= 2^16 p.  A product of
//     values A, B comes back below (A B / (p^2 2^HEAD) + 1) p, i.e. essentially p, whatever small
//     multiples of p the operands carried; no value fold is ever needed, only limb carries;
//   * all per-curve numbers (digits of p, R^2, 1, a, b, bias tables, p - 2) live in __constant__
//     memory; the compile-time bounds depend on |p| only, so curves of equal size share code.
// Bounds per element type E<PB, LB, TB, VB>: LB >= limbs 0..NL-2, TB >= top limb, VB >= value / p.
#pragma once
#include <stdint.h>
#include <utility>
#include "ecamd_madchain.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G29_FN __device__ __forceinline__
#define G29_NOINLINE __device__ __noinline__
#else
#define G29_FN inline
#define G29_NOINLINE inline
#endif
// Multiplications are inlined into the formulas for every field size: measured on MI355X against an
// out-of-line multiplier (arguments and result in VGPRs), inlining gives +25 % on 256-bit fields, +23 % on
// secp384r1, +16 % on 448 bits, +42 % on brainpoolP512r1 and +69 % on secp521r1 (calls cost the caller its
// register allocation: 240 bytes of scratch and one wave per SIMD).  The price is compile time (the three
// 19-limb units take 3-4 minutes each; libecc_amd/build.py compiles the units in parallel).
// -DG29_CALL_FROM_NL=<limbs> switches to calls from that size on.
#ifndef G29_CALL_FROM_NL
#define G29_CALL_FROM_NL 99
#endif

namespace g29 {

typedef uint32_t u32;
typedef uint64_t u64;

// Reduction flavour of a translation unit: dense Montgomery (any prime), or one of the special cases
//   -DG29_P25519       p = 2^255 - 19 (WEI25519 / Ed25519 / X25519): NO Montgomery form (R = 1) and no
//                      headroom limb: 9 limbs, the 18-limb product is folded with 2^261 = 1216 and
//                      2^255 = 19 (mod p), see mul_raw.
#if defined(G29_P25519)
constexpr bool P25519 = true;
#else
constexpr bool P25519 = false;
#endif
//   -DG29_K256         p = 2^256 - 2^32 - 977 (secp256k1): plain residues on 9 limbs like the flavour above; the
//                      product is folded with 2^261 = 2^37 + 31264 and 2^256 = 2^32 + 977 (mod p), see mul_raw.
#if defined(G29_K256)
constexpr bool K256 = true;
#else
constexpr bool K256 = false;
#endif
//   -DG29_P448         p = 2^448 - 2^224 - 1 (WEI448 / Ed448 / X448): plain residues on 16 limbs of TWENTY-EIGHT bits, so that
//                      2^224 = phi sits on the boundary of limbs 7 | 8 and phi^2 = phi + 1 folds the product inside its
//                      columns: with a = a0 + a1 phi, b = b0 + b1 phi the product is (a0 b0 + a1 b1) + (a0 b1 + a1 (b0 + b1)) phi,
//                      256 MADs in 16 column pairs and no second pass (the radix-2^29 form of round 2 needed 304 MADs in
//                      47 columns: 0.63 of the MAD stream).  No headroom limb: carry() folds the bits from 2^448 up as well,
//                      every carried value is below 2p, and the bias multiples of a subtraction are 4p and 8p.
#if defined(G29_P448)
constexpr bool P448 = true;
#else
constexpr bool P448 = false;
#endif
//   -DG29_M521P        p = 2^521 - 1 (secp521r1) on plain residues: EIGHTEEN limbs of 29 bits (2^522 = 2 mod p), so a product
//                      column is sum_(i+j=k) a_i b_j + sum_(i+j=k+18) a_i (2 b_j): 324 MADs in 18 columns, one pass, no Montgomery
//                      form (the -DG29_MERSENNE521 flavour it replaces: 19 limbs, 361 + 19 MADs in 37 columns).  One bit of
//                      headroom: handled like the Goldilocks flavour (carry() folds the bits from 2^521 up into limb 0).
#if defined(G29_M521P)
constexpr bool M521P = true;
#else
constexpr bool M521P = false;
#endif
constexpr int W = P448 ? 28 : 29;   // limb width of this translation unit
constexpr u32 MASK = (1u << W) - 1;
constexpr bool PLAIN9 = P25519 || K256;  // R = 1 on nine limbs, no headroom limb
constexpr bool NOHEAD = P448 || M521P;   // R = 1, top limb of 28 bits, no headroom: carried values are below 2p, biases are 4p / 8p
constexpr bool PLAIN = PLAIN9 || NOHEAD; // R = 1
//   -DG29_P384S        p = 2^384 - 2^128 - 2^96 + 2^32 - 1 (secp384r1), Montgomery form with the reduction on the SIGNED sparse
//                      digits of p + 1 = 2^3 2^29 - 2^9 2^(29*3) - 2^12 2^(29*4) + 2^7 2^(29*13): p = -1 mod 2^29, so the quotient
//                      digit m_k of a column is its low digit, "- m_k" clears it, and m_k (p + 1) is FOUR signed MADs
//                      (v_mad_i64_i32) into columns k+1, k+3, k+4, k+13 instead of thirteen products with the digits of p:
//                      252 MADs per multiplication instead of 378, 161 per squaring instead of 287.  Column sums are signed
//                      (arithmetic carry shifts), so operand limbs get one bit less room (mul_fits).
#if defined(G29_P384S)
constexpr bool P384S = true;
#else
constexpr bool P384S = false;
#endif
//   -DG29_P192S        p = 2^192 - 2^64 - 1 (secp192r1): the same reduction, p + 1 = -2^6 2^(29*2) + 2^18 2^(29*6): TWO signed MADs per
//                      quotient digit (80 MADs per multiplication instead of 128, 52 per squaring instead of 100)
//   -DG29_P224S        p = 2^224 - 2^96 + 1 (secp224r1): p = +1 mod 2^29, so the quotient digit is the NEGATED low digit of the
//                      column, "+ m_k" clears it (and carries), and m_k (p - 1) = m_k (-2^9 2^(29*3) + 2^21 2^(29*7)) is two signed
//                      MADs (99 / 63 MADs instead of 162 / 126)
#if defined(G29_P192S)
constexpr bool P192S = true;
#else
constexpr bool P192S = false;
#endif
#if defined(G29_P224S)
constexpr bool P224S = true;
#else
constexpr bool P224S = false;
#endif
constexpr bool SPARSE = P384S || P192S || P224S;   // Montgomery form, signed sparse reduction, signed column sums
// the signed digits (column offset, factor) of p + 1 (p - 1 for secp224r1) in radix 2^29, padded to four
constexpr int SPARSE_N = P384S ? 4 : 2;
constexpr int SPARSE_OFF[4] = {P384S ? 1 : (P192S ? 2 : 3), P384S ? 3 : (P192S ? 6 : 7), 4, 13};
constexpr int32_t SPARSE_C[4] = {P384S ? 8 : (P192S ? -64 : -512), P384S ? -512 : (P192S ? (1 << 18) : (1 << 21)), -4096, 128};
constexpr bool SPARSE_PLUS1 = P224S;               // p = +1 mod 2^29: m_k = -column mod 2^29 and "+ m_k" clears the digit

constexpr int nl_for(int pbits) { return (pbits + 16 + W - 1) / W; }
constexpr int nl_for_flavour(int pbits, int flavour) { return (flavour == 2 || flavour == 4) ? 9 : (flavour == 5 ? 16 : (flavour == 1 ? 18 : nl_for(pbits))); }
constexpr u64 cmin(u64 a, u64 b) { return a < b ? a : b; }
constexpr u64 cmax(u64 a, u64 b) { return a > b ? a : b; }
// x * 2^e for any sign of e, rounded up
constexpr u64 shl_ceil(u64 x, int e) { return e >= 0 ? (x << e) : ((x + ((1ull << -e) - 1)) >> -e); }
// x * 2^e rounded down
constexpr u64 shl_floor(u64 x, int e) { return e >= 0 ? (x << e) : (x >> -e); }

template <int PB> struct Cfg {
	static constexpr int PBITS = PB;
	static_assert(!P25519 || PB == 255, "the 2^255 - 19 flavour is only for 255-bit fields");
	static_assert(!K256 || PB == 256, "the secp256k1 flavour is only for 256-bit fields");
	static_assert(!P448 || PB == 448, "the Goldilocks flavour is only for 448-bit fields");
	static_assert(!M521P || PB == 521, "the plain Mersenne flavour is only for 521-bit fields");
	static constexpr int NL = PLAIN9 ? 9 : (P448 ? 16 : (M521P ? 18 : nl_for(PB)));
	static constexpr int HEAD = W * NL - PB;           // R / p >= 2^HEAD, HEAD >= 16 (Montgomery flavours)
	static constexpr int TOPSH = PB - W * (NL - 1);    // p < 2^(29 (NL-1) + TOPSH); may be <= 0
	static_assert((HEAD >= 16 || PLAIN9 || NOHEAD) && NL <= 19, "field size not supported");
	static_assert(!NOHEAD || TOPSH == 28, "no-headroom flavours: the top limb holds 28 bits of p");
	// top limb of a non-negative-limb value < vb * p
	static constexpr u64 top_from_vb(u64 vb) { return shl_ceil(vb, TOPSH) + 1; }
	// bias multiples are 2^(BIAS_STEP + BIAS_OFF) p: with a (nearly) empty top limb the smallest
	// useful multiple is the one whose top digit is at least a few units
	static constexpr int BIAS_OFF = (1 - TOPSH) > 0 ? (1 - TOPSH) : 0;
	// va * vb < 2^(2 HEAD - 2) without overflowing u64
	// (2^255 - 19 flavour: va * vb <= 2^14 keeps the last product limb and the fold quotient in 32 bits)
	// (secp256k1 flavour: the last product limb is < va vb 2^19, so va * vb <= 2^12)
	// (Goldilocks flavour: the product is folded inside its columns, whatever the operands' values: only limb bounds matter)
	static constexpr int PROD_E = P25519 ? 14 : (K256 ? 12 : (NOHEAD ? 62 : ((2 * HEAD - 2) > 62 ? 62 : (2 * HEAD - 2))));
	static constexpr bool prod_ok(u64 va, u64 vb) { return va == 0 || vb <= ((1ull << PROD_E) - (PLAIN9 ? 0 : 1)) / va; }
};

// the (LOGC, S) combinations the formulas use for "a - b + C p": bias tables for exactly these
// are precomputed per curve by the host
constexpr int NBIAS = 16;
// (no-headroom flavours: steps of one -- 2p, 4p, 8p are the multiples whose top limb fits 32 bits)
constexpr int bias_step(int i) { return NOHEAD ? (i % 8) + 1 : 2 * (i % 8) + 2; }
constexpr int BIAS_STEP[NBIAS] = {bias_step(0), bias_step(1), bias_step(2),  bias_step(3),  bias_step(4),  bias_step(5),  bias_step(6),  bias_step(7),
				  bias_step(8), bias_step(9), bias_step(10), bias_step(11), bias_step(12), bias_step(13), bias_step(14), bias_step(15)};
constexpr int BIAS_S[NBIAS] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2};
// table index of the multiple 2^logc p for a field with bias offset 'off' (logc = step + off)
constexpr int bias_index(int logc, int s, int off)
{
	for (int i = 0; i < NBIAS; i++) {
		if (BIAS_STEP[i] + off == logc && BIAS_S[i] == s) {
			return i;
		}
	}
	return -1;
}

// per-curve constants (one per slot, filled by ecamd_host.cpp; layout mirrored there)
template <int NL> struct CurveG {
	u32 p[NL];            // digits of p
	u32 r2[NL];           // R^2 mod p
	u32 one[NL];          // R mod p
	u32 a[NL];            // a R mod p
	u32 b[NL];            // b R mod p
	u32 pm2[NL];          // digits of p - 2 (inversion exponent)
	u32 bias[NBIAS][NL];  // limbs of 2^LOGC p re-balanced so that low limbs >= 2^(29+S) - 2^S
	// coordinate import / export factors.  Normally R^2, R^2 (into the Montgomery domain) and 1, 1 (out of it).
	// When the curve is isomorphic to one with a = -3 ((x, y) -> (u^2 x, u^3 y), u^4 a = -3: the brainpool
	// r1 curves, some GOST ones) the kernels compute on that curve (a, b above are a u^4 = -3 and b u^6) and the
	// map rides on the conversions for free: ix = u^2 R^2, iy = u^3 R^2, ex = u^-2, ey = u^-3.
	u32 ix[NL], iy[NL], ex[NL], ey[NL];
	u32 mpinv;            // -p^-1 mod 2^29
	u32 pbits;
	u32 a_is_m3;
	u32 a_is_zero;        // a == 0 (the secp k1 curves): the doubling drops its a Z^4 term
};

template <int PB, u64 LB_, u64 TB_, u64 VB_> struct E {
	typedef Cfg<PB> C;
	static constexpr u64 LB = LB_;
	static constexpr u64 VB = VB_;
	static constexpr u64 TB = cmin(TB_, C::top_from_vb(VB_));
	static_assert(LB_ < (1ull << 32) && TB < (1ull << 32), "limb does not fit 32 bits");
	static_assert(VB_ < (1ull << 40), "value bound out of range");
	u32 l[C::NL];
};

template <class T, class S> G29_FN T weaken(const S &s)
{
	static_assert(T::LB >= S::LB && T::TB >= S::TB && T::VB >= S::VB, "weaken() must not tighten bounds");
	T r;
#pragma unroll
	for (int i = 0; i < T::C::NL; i++) {
		r.l[i] = s.l[i];
	}
	return r;
}

// ---- multiplication ----
template <int PB> constexpr u64 mul_vb(u64 va, u64 vb) { return PLAIN ? 2 : shl_ceil(va * vb, -Cfg<PB>::HEAD) + 1; }
constexpr u64 K256_MULX = 1ull << 15;     // what limb 2 of a secp256k1 product may exceed the mask by
constexpr u64 P25519_MULX = 1ull << 17;   // what limb 1 of a 2^255 - 19 product may exceed the mask by
constexpr u64 P448_MULX = 1ull << 10;     // what the lazy limbs of a no-headroom product (1 and 9 / 1) may exceed the mask by
constexpr u32 TOPMASK28 = (1u << 28) - 1;  // top limb of the no-headroom flavours
constexpr u64 CANON_TB = NOHEAD ? TOPMASK28 : MASK;   // top limb of a canonical value (< p)
template <int PB, u64 VBO> struct MulOut {
	// 2^255 - 19 flavour: exact low digits, top limb < 2^23 + 2^12 (see mul_raw), value < 2p
	// secp256k1 flavour: exact low digits, top limb < 2^24 + 2^17, value < 2p
	// Goldilocks flavour: limbs 1 and 9 carry the (lazily added) high parts of the two wrap-around carries, see mul_p448
	// 2^255 - 19 flavour: limb 1 takes the high part of the folded overflow lazily (see mul_p25519); the top limb is below 2^23 (the
	// bound stays the round-2 one, which also covers a canonical constant's 2^23 + 1)
	// secp256k1 flavour: limb 2 takes the last carries of the folded overflow lazily (see mul_k256); top-limb bound as in round 2
	typedef E<PB, NOHEAD ? (MASK + P448_MULX) : (P25519 ? (MASK + P25519_MULX) : (K256 ? (MASK + K256_MULX) : MASK)),
		  P25519 ? ((1ull << 23) + (1ull << 12)) : (K256 ? ((1ull << 24) + (1ull << 17)) : (NOHEAD ? (u64)TOPMASK28 : Cfg<PB>::top_from_vb(VBO))), VBO> type;
};
template <int NL> constexpr bool mul_fits(u64 la, u64 lb)
{
	// NL*la*lb + NL*2^58 + 2^36 < 2^64
	if (P448) {
		// Goldilocks flavour: the fullest column (limb 8) sums 38 la lb -- 16 products, those against b0 + b1 counted twice, plus
		// the shared column S_0 -- and a carry below 2^37
		return la == 0 || lb <= ((0xFFFFFFFFFFFFFFFFull - (1ull << 40)) / 38) / la;
	}
	if (SPARSE) {
		// signed sparse flavours: signed column sums -- NL la lb + NL 2^52 (reduction) + 2^35 (carry) < 2^63
		return la == 0 || lb <= (((1ull << 63) - (1ull << 57)) / NL) / la;
	}
	if (M521P) {
		// plain Mersenne flavour: a column sums (k + 1) la lb + (17 - k) la (2 lb) <= 35 la lb and a carry below 2^37
		return la == 0 || lb <= ((0xFFFFFFFFFFFFFFFFull - (1ull << 40)) / 36) / la;
	}
	return la == 0 || lb <= ((0xFFFFFFFFFFFFFFFFull - (u64)NL * (1ull << 58) - (1ull << 36)) / NL) / la;
}

// single MADs with a wave-uniform multiplier (fold constants; hipcc pads every asm statement with an s_nop, so the product columns
// use the chains of ecamd_madchain.h instead)
#if defined(__HIPCC__) && defined(U29_ASM_MAD)
#define G29_MAD_VS(acc, a, b) \
	do { u64 dead_; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(dead_) : "v"(a), "s"(b)); } while (0)
#define G29_MUL_VS(acc, a, b) \
	do { u64 dead_; asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(acc), "=s"(dead_) : "v"(a), "s"(b)); } while (0)
// two MADs with wave-uniform multipliers in one statement (one padding s_nop instead of two)
#define G29_MAD2_VS(acc, a, b, c, d) \
	do { u64 dead_; asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0" \
			    : "+v"(acc), "=&s"(dead_) : "v"(a), "s"(b), "v"(c), "s"(d)); } while (0)
#else
#define G29_MAD2_VS(acc, a, b, c, d) do { ECAMD_COUNT_MAD(2); acc += (u64)(a) * (b) + (u64)(c) * (d); } while (0)
#define G29_MUL_VS(acc, a, b) do { ECAMD_COUNT_MAD(1); acc = (u64)(a) * (b); } while (0)
#define G29_MAD_VS(acc, a, b) do { ECAMD_COUNT_MAD(1); acc += (u64)(a) * (b); } while (0)
#endif

// ---- multiply-accumulate chains (ecamd_madchain.h): a column's products go out in asm statements of up
// to four MADs; from G29_DUAL_FROM_NL limbs on (one or two waves per SIMD) on two alternating accumulators ----
#ifndef G29_DUAL_FROM_NL
#define G29_DUAL_FROM_NL 12
#endif
template <int N, bool DUAL, bool YS, bool Z2 = false> G29_FN void mad_chain(u64 &acc, u64 &acc2, const u32 *x, const u32 *y)
{
	ecamd_mad_chain<N, DUAL, YS, Z2>(acc, acc2, x, y);
}

// One column k of the product a b (SQR: off-diagonal products once, against the doubled operand a2) plus,
// for the dense flavour, the reduction products m_i p_(k-i) known so far, accumulated into acc (+ acc2).
template <int NL, bool SQR, int K_> struct Column {
	static constexpr int LO = (K_ < NL) ? 0 : (K_ - NL + 1);
	static constexpr int HI = (K_ < NL) ? K_ : (NL - 1);
	static constexpr int HALF = K_ / 2;  // SQR: i <= j  <=>  i <= K/2
	static constexpr int NPROD = SQR ? ((HALF >= LO) ? (HALF - LO + 1) : 0) : (HI - LO + 1);
	// dense reduction: i < K in the low half, the whole anti-diagonal in the high half
	static constexpr int RLO = LO;
	static constexpr int RHI = (K_ < NL) ? (K_ - 1) : HI;
	static constexpr int NRED = (RHI >= RLO) ? (RHI - RLO + 1) : 0;
	static constexpr bool DUAL = NL >= G29_DUAL_FROM_NL;
	// which chains touch the second accumulator (a chain of one product only uses the first)
	static constexpr bool USES2_PROD = DUAL && NPROD >= 2;
	static constexpr bool USES2_RED = DUAL && NRED >= 2;

	static G29_FN void products(u64 &acc, u64 &acc2, const u32 *a, const u32 *b, const u32 *a2)
	{
		if constexpr (NPROD > 0) {
			u32 x[NPROD], y[NPROD];
#pragma unroll
			for (int n = 0; n < NPROD; n++) {
				const int i = LO + n, j = K_ - i;
				x[n] = a[i];
				y[n] = !SQR ? b[j] : (i < j ? a2[j] : a[i]);
			}
			mad_chain<NPROD, DUAL, false, true>(acc, acc2, x, y);   // acc2 starts here (zero addend), if it is used at all
		}
	}
	static G29_FN void reduction(u64 &acc, u64 &acc2, const u32 *m, const u32 *p)
	{
		if constexpr (NRED > 0) {
			u32 x[NRED], y[NRED];
#pragma unroll
			for (int n = 0; n < NRED; n++) {
				x[n] = m[RLO + n];
				y[n] = p[K_ - (RLO + n)];
			}
			mad_chain<NRED, DUAL, true, !USES2_PROD>(acc, acc2, x, y);
		}
	}
};

// secp384r1 flavour: m_i (p + 1) as signed MADs with wave-uniform constants -- the digits (column offset, factor) of p + 1
// = 2^32 - 2^96 - 2^128 + 2^384 in radix 2^29 -- for the quotient digits i = K - offset that exist
// (one asm statement per column: hipcc pads every statement with an s_nop)
template <int N> G29_FN void smad_chain(u64 &acc, const u32 *x, const int32_t *y)
{
#if defined(__HIPCC__) && defined(U29_ASM_MAD)
	u64 dead_;
	if constexpr (N == 1) {
		asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=&s"(dead_) : "v"(x[0]), "s"(y[0]));
	} else if constexpr (N == 2) {
		asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0"
		    : "+v"(acc), "=&s"(dead_) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]));
	} else if constexpr (N == 3) {
		asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0"
		    : "+v"(acc), "=&s"(dead_) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]));
	} else if constexpr (N == 4) {
		asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\tv_mad_i64_i32 %0, %1, %4, %5, %0\n\tv_mad_i64_i32 %0, %1, %6, %7, %0\n\tv_mad_i64_i32 %0, %1, %8, %9, %0"
		    : "+v"(acc), "=&s"(dead_) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]));
	}
#else
	ECAMD_COUNT_MAD(N);
	for (int n = 0; n < N; n++) {
		acc += (u64)((int64_t)(int32_t)x[n] * (int64_t)y[n]);
	}
#endif
}
template <int NL, int K_> G29_FN void sparse_reduction(u64 &acc, const u32 *m, const int32_t *c)
{
	static_assert(!P384S || NL == 14, "secp384r1 flavour: 14 limbs");
	static_assert(!P192S || NL == 8, "secp192r1 flavour: 8 limbs");
	static_assert(!P224S || NL == 9, "secp224r1 flavour: 9 limbs");
	constexpr auto has = [](int t) { return t < SPARSE_N && K_ - SPARSE_OFF[t] >= 0 && K_ - SPARSE_OFF[t] < NL; };
	constexpr int N = (has(0) ? 1 : 0) + (has(1) ? 1 : 0) + (has(2) ? 1 : 0) + (has(3) ? 1 : 0);
	if constexpr (N > 0) {
		u32 x[N];
		int32_t y[N];
		int n = 0;
#pragma unroll
		for (int t = 0; t < SPARSE_N; t++) {
			const int i = K_ - SPARSE_OFF[t];
			if (i >= 0 && i < NL) {
				x[n] = m[i];
				y[n] = c[t];
				n++;
			}
		}
		smad_chain<N>(acc, x, y);
	}
}

template <int NL, bool SQR, int K_> G29_FN void mul_column(u64 &acc, u32 *m, u32 *r, const u32 *a, const u32 *b,
							   const u32 *a2, const u32 *p, u32 mpinv, const int32_t *c384)
{
	static_assert(!PLAIN || NL < 0, "the plain-residue flavours have multipliers of their own");
	typedef Column<NL, SQR, K_> C;
	u64 acc2;  // written by the first chain that uses it (Z2), never read otherwise
	C::products(acc, acc2, a, b, a2);
	if constexpr (SPARSE) {
		if constexpr (C::USES2_PROD) {
			acc += acc2;
		}
		sparse_reduction<NL, K_>(acc, m, c384);
		// the quotient digit of a column below NL and the "-+ m_k" that clears its low digit; the column sum is signed
		if constexpr (K_ < NL && SPARSE_PLUS1) {
			const u32 mk = (0u - (u32)acc) & MASK;
			m[K_] = mk;
			acc += mk;
		} else if constexpr (K_ < NL) {
			m[K_] = (u32)acc & MASK;
		} else {
			r[K_ - NL] = (u32)acc & MASK;
		}
		acc = (u64)((int64_t)acc >> W);
		return;
	} else {
		C::reduction(acc, acc2, m, p);
		if constexpr (C::USES2_PROD || C::USES2_RED) {
			acc += acc2;
		}
		if constexpr (K_ < NL) {
			m[K_] = ((u32)acc * mpinv) & MASK;
			mad_chain<1, false, true>(acc, acc2, &m[K_], &p[0]);
		} else {
			r[K_ - NL] = (u32)acc & MASK;
		}
	}
	acc >>= W;
	// (no pin needed: the next column starts with an asm statement that takes acc as an operand)
}

// ---- Goldilocks flavour (p = 2^448 - 2^224 - 1, sixteen 28-bit limbs, phi = 2^224 = 2^(28*8), phi^2 = phi + 1) ----
// a = a0 + a1 phi, b = b0 + b1 phi:  a b = lo + hi phi with lo = a0 b0 + a1 b1 and hi = a0 b1 + a1 (b0 + b1), two sums of 8 x 8
// products with columns 0..14; columns 8..14 of lo are multiples of phi, those of hi multiples of phi^2 = phi + 1:
//     limb j     = lo_j + hi_(j+8)
//     limb j + 8 = hi_j + lo_(j+8) + hi_(j+8)                    (j = 0..7; column 15 is empty)
// Per pair j one chain computes S = hi_(j+8) from zero, two more (on the carries of the low and of the high half) the rest, and
// S is added to both: 32 MADs and two 64-bit additions per pair, 256 MADs in all.  The three chains of a pair go out as two
// dual-accumulator asm statements (two independent accumulators keep a lone wave issuing, ecamd_madchain.h).
// Squaring: lo = a0^2 + a1^2 (off-diagonal products once, against the doubled operand), hi = a1 (2 a0 + a1): 136 MADs.
// The carry out of limb 7 belongs to limb 8, the carry out of limb 15 (multiples of 2^448 = phi + 1) to limbs 0 and 8: both are
// below 2^37 and are added lazily (low 28 bits to the limb, the rest to the next one), then limbs 0 and 8 give their bit 28 up to
// limbs 1 and 9: the result has limbs 1 and 9 below 2^28 + 2^10, all others below 2^28, and a value below 2^448 + 2^263 < 2p.
//
// two product lists X (accumulator ax) and Y (accumulator ay, starting from zero when ZY) of any lengths, interleaved as long as
// both last
template <int NX, int NY, bool ZY> G29_FN void dual_chains(u64 &ax, u64 &ay, const u32 *xx, const u32 *yx, const u32 *xy, const u32 *yy)
{
	constexpr int NM = NX < NY ? NX : NY;
	if constexpr (NM > 0) {
		u32 x[2 * NM], y[2 * NM];
#pragma unroll
		for (int n = 0; n < NM; n++) {
			x[2 * n] = xx[n];
			y[2 * n] = yx[n];
			x[2 * n + 1] = xy[n];
			y[2 * n + 1] = yy[n];
		}
		mad_chain<2 * NM, true, false, ZY>(ax, ay, x, y);
	} else if constexpr (ZY) {
		ay = 0;
	}
	u64 unused;
	if constexpr (NX > NM) {
		mad_chain<NX - NM, false, false>(ax, unused, xx + NM, yx + NM);
	}
	if constexpr (NY > NM) {
		mad_chain<NY - NM, false, false>(ay, unused, xy + NM, yy + NM);
	}
}

// column pair J: clo / chi hold the carries of the two halves on entry and on exit; bx = b0 + b1 (SQR: 2 a0 + a1), a2 = 2 a (SQR)
template <bool SQR, int J> G29_FN void p448_pair(u64 &clo, u64 &chi, u32 *r, const u32 *a, const u32 *b, const u32 *bx, const u32 *a2)
{
	// product counts: hiA = lo_(J+8), S = hi_(J+8), hiB = hi_J, lo = lo_J
	constexpr int NHA = SQR ? 2 * (((J + 8) / 2) - J) : 2 * (7 - J);     // SQR: i = J+1 .. (J+8)/2, both halves
	constexpr int NS = SQR ? (7 - J) : 2 * (7 - J);
	constexpr int NHB = SQR ? (J + 1) : 2 * (J + 1);
	constexpr int NLO = SQR ? 2 * (J / 2 + 1) : 2 * (J + 1);
	u32 xha[NHA + 1], yha[NHA + 1], xs[NS + 1], ys[NS + 1], xhb[NHB], yhb[NHB], xlo[NLO], ylo[NLO];
	if constexpr (!SQR) {
#pragma unroll
		for (int i = J + 1; i < 8; i++) {
			const int k = J + 8 - i, n = 2 * (i - J - 1);
			xha[n] = a[i];      yha[n] = b[k];              // a0 b0
			xha[n + 1] = a[8 + i]; yha[n + 1] = b[8 + k];   // a1 b1
			xs[n] = a[i];       ys[n] = b[8 + k];           // a0 b1
			xs[n + 1] = a[8 + i];  ys[n + 1] = bx[k];       // a1 (b0 + b1)
		}
#pragma unroll
		for (int i = 0; i <= J; i++) {
			const int k = J - i;
			xhb[2 * i] = a[i];         yhb[2 * i] = b[8 + k];
			xhb[2 * i + 1] = a[8 + i]; yhb[2 * i + 1] = bx[k];
			xlo[2 * i] = a[i];         ylo[2 * i] = b[k];
			xlo[2 * i + 1] = a[8 + i]; ylo[2 * i + 1] = b[8 + k];
		}
	} else {
#pragma unroll
		for (int i = J + 1; 2 * i <= J + 8; i++) {
			const int k = J + 8 - i, n = 2 * (i - J - 1);
			xha[n] = a[i];         yha[n] = (i < k) ? a2[k] : a[i];
			xha[n + 1] = a[8 + i]; yha[n + 1] = (i < k) ? a2[8 + k] : a[8 + i];
		}
#pragma unroll
		for (int i = J + 1; i < 8; i++) {
			xs[i - J - 1] = a[8 + i];
			ys[i - J - 1] = bx[J + 8 - i];
		}
#pragma unroll
		for (int i = 0; i <= J; i++) {
			xhb[i] = a[8 + i];
			yhb[i] = bx[J - i];
		}
#pragma unroll
		for (int i = 0; 2 * i <= J; i++) {
			const int k = J - i;
			xlo[2 * i] = a[i];         ylo[2 * i] = (i < k) ? a2[k] : a[i];
			xlo[2 * i + 1] = a[8 + i]; ylo[2 * i + 1] = (i < k) ? a2[8 + k] : a[8 + i];
		}
	}
	if constexpr (NS > 0) {
		u64 s;
		dual_chains<NHA, NS, true>(chi, s, xha, yha, xs, ys);
		dual_chains<NHB, NLO, false>(chi, clo, xhb, yhb, xlo, ylo);
		clo += s;
		chi += s;
	} else {
		dual_chains<NHB, NLO, false>(chi, clo, xhb, yhb, xlo, ylo);
	}
	r[J] = (u32)clo & MASK;
	clo >>= W;
	r[8 + J] = (u32)chi & MASK;
	chi >>= W;
}

template <bool SQR, int... Js> G29_FN void p448_pairs(u64 &clo, u64 &chi, u32 *r, const u32 *a, const u32 *b, const u32 *bx, const u32 *a2,
						      std::integer_sequence<int, Js...>)
{
	(p448_pair<SQR, Js>(clo, chi, r, a, b, bx, a2), ...);
}

template <bool SQR> G29_FN void mul_p448(u32 *r, const u32 *a, const u32 *b)
{
	u32 bx[8], a2[16];
	if constexpr (SQR) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			a2[i] = a[i] << 1;
		}
#pragma unroll
		for (int i = 0; i < 8; i++) {
			bx[i] = a2[i] + a[8 + i];
		}
	} else {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			bx[i] = b[i] + b[8 + i];
		}
	}
	u64 clo = 0, chi = 0;
	p448_pairs<SQR>(clo, chi, r, a, b, bx, a2, std::make_integer_sequence<int, 8>());
	// clo: the carry out of limb 7, chi: out of limb 15 (both < 2^37)
	const u32 l7 = (u32)clo & MASK, h7 = (u32)(clo >> W), l15 = (u32)chi & MASK, h15 = (u32)(chi >> W);
	const u32 r0 = r[0] + l15, r8 = r[8] + l7 + l15;
	r[0] = r0 & MASK;
	r[1] += h15 + (r0 >> W);
	r[8] = r8 & MASK;
	r[9] += h7 + h15 + (r8 >> W);
}

// ---- plain Mersenne flavour (p = 2^521 - 1, eighteen 29-bit limbs: 2^522 = 2^(29*18) = 2 mod p) ----
// column k = sum_(i+j=k) a_i b_j + 2 sum_(i+j=k+18) a_i b_j, eighteen products on two accumulators -- the direct and the wrapped
// ones, the latter doubled when the two are joined -- and one carry chain (squaring: the off-diagonal products once against the
// doubled operand, the wrapped ones against the doubled operand on both sides).  The carry out of limb 17 (< 2^36, weight 2^522 = 2) and bit 28 of limb 17 (2^521 = 1) go to limbs 0 and 1 lazily:
// limb 1 below 2^29 + 2^10, limb 17 below 2^28, all others below 2^29, value below 2^521 + 2^40 < 2p.
template <bool SQR, int K_> G29_FN void m521p_column(u64 &acc, u32 *r, const u32 *a, const u32 *b, const u32 *a2)
{
	constexpr int NWRAP = SQR ? ((K_ + 18) / 2 - K_) : (17 - K_);     // SQR: i = K+1 .. (K+18)/2
	constexpr int NDIR = SQR ? (K_ / 2 + 1) : (K_ + 1);
	if constexpr (!SQR) {
		// the wrapped products go to the second accumulator as they are and are doubled in the addition that joins the two
		// (one v_lshl_add_u64), so no doubled copy of b has to stay in registers
		u32 xd[NDIR], yd[NDIR], xw[NWRAP + 1], yw[NWRAP + 1];
#pragma unroll
		for (int i = 0; i <= K_; i++) {
			xd[i] = a[i];
			yd[i] = b[K_ - i];
		}
#pragma unroll
		for (int i = K_ + 1; i < 18; i++) {
			xw[i - K_ - 1] = a[i];
			yw[i - K_ - 1] = b[K_ + 18 - i];
		}
		if constexpr (NWRAP > 0) {
			u64 wrap;
			dual_chains<NDIR, NWRAP, true>(acc, wrap, xd, yd, xw, yw);
			acc += wrap << 1;
		} else {
			u64 acc2;
			mad_chain<NDIR, true, false, true>(acc, acc2, xd, yd);
			acc += acc2;
		}
	} else {
		constexpr int N = NDIR + NWRAP;
		u32 x[N], y[N];
#pragma unroll
		for (int i = 0; 2 * i <= K_; i++) {
			x[i] = a[i];
			y[i] = (i < K_ - i) ? a2[K_ - i] : a[i];
		}
#pragma unroll
		for (int i = K_ + 1; 2 * i <= K_ + 18; i++) {
			const int j = K_ + 18 - i;
			x[NDIR + i - K_ - 1] = a2[i];
			y[NDIR + i - K_ - 1] = (i < j) ? a2[j] : a[i];
		}
		if constexpr (N >= 2) {
			u64 acc2;
			mad_chain<N, true, false, true>(acc, acc2, x, y);
			acc += acc2;
		} else {
			u64 unused;
			mad_chain<N, false, false>(acc, unused, x, y);
		}
	}
	r[K_] = (u32)acc & MASK;
	acc >>= W;
}
template <bool SQR, int... Ks> G29_FN void m521p_columns(u64 &acc, u32 *r, const u32 *a, const u32 *b, const u32 *a2, std::integer_sequence<int, Ks...>)
{
	(m521p_column<SQR, Ks>(acc, r, a, b, a2), ...);
}
template <bool SQR> G29_FN void mul_m521p(u32 *r, const u32 *a, const u32 *b)
{
	u32 a2[18];   // squaring: 2 a
	if constexpr (SQR) {
#pragma unroll
		for (int i = 0; i < 18; i++) {
			a2[i] = a[i] << 1;
		}
	}
	u64 acc = 0;
	m521p_columns<SQR>(acc, r, a, b, a2, std::make_integer_sequence<int, 18>());
	// acc: the carry out of limb 17 (weight 2^522 = 2); bit 28 of limb 17 is 2^521 = 1
	const u64 w = (acc << 1) + (r[17] >> 28);
	r[17] &= TOPMASK28;
	const u32 r0 = r[0] + ((u32)w & MASK);
	r[0] = r0 & MASK;
	r[1] += (u32)(w >> W) + (r0 >> W);
}

// ---- 2^255 - 19 flavour (nine 29-bit limbs, plain residues; 2^261 = 64 * 2^255 = 1216 mod p) ----
// Round 4 ("split fold", the default): the high columns 9..16 are summed into EIGHT INDEPENDENT 64-bit accumulators H_k -- no carry
// chain, no column ends -- and fold as the two register halves they already are: H_k 2^(29 (9 + k)) = 1216 H_k 2^(29 k) and
// H_k = lo32 + 2^32 hi32 = lo32 + 8 hi32 2^29, so column k takes 1216 lo32(H_k) and column k + 1 takes 9728 hi32(H_k): two MADs
// with wave-uniform multipliers per high column instead of a MAD, a v_and_b32 and a v_lshrrev_b64 on a serial chain.
// 81 + 16 MADs and 9 column ends (round 3: 81 + 9 MADs and 17 column ends, kept below as -DG29_P25519_CHAINFOLD; the high columns
// first on a carry chain of their own: digits h[0..7] and the last carry h[8] < va vb 2^17 <= 2^31, Cfg::prod_ok, then 1216 h[k] as
// one more MAD of column k).  Column 8 -- with the carry of column 7 -- holds everything from 2^232 up: its bits from 23 up are
// multiples of 2^255 = 19 and go to limbs 0 and 1 lazily (19 q < 2^46: limb 1 below 2^29 + 2^17), the result is below
// 2^255 + 2^47 < 2p with a top limb below 2^23.
#if defined(G29_P25519_CHAINFOLD)
template <bool SQR, int K_> G29_FN void p25519_column(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u32 *h, u32 f)
{
	u64 acc2;
	Column<9, SQR, K_>::products(acc, acc2, a, b, a2);
	if constexpr (K_ < 9) {
		G29_MAD_VS(acc, h[K_], f);
	}
	if constexpr (K_ != 8) {
		out[K_ < 9 ? K_ : K_ - 9] = (u32)acc & MASK;
		acc >>= W;
	}
}
template <bool SQR, int... Ks> G29_FN void p25519_columns(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u32 *h, u32 f,
							  std::integer_sequence<int, Ks...>)
{
	(p25519_column<SQR, Ks>(acc, out, a, b, a2, h, f), ...);
}
template <bool SQR, int... Ks> G29_FN void p25519_hi_columns(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, std::integer_sequence<int, Ks...>)
{
	(p25519_column<SQR, 9 + Ks>(acc, out, a, b, a2, nullptr, 0u), ...);
}
#else
template <bool SQR, int K_> G29_FN void p25519_hi_column(u64 *H, const u32 *a, const u32 *b, const u32 *a2)
{
	typedef Column<9, SQR, 9 + K_> C;
	u32 x[C::NPROD], y[C::NPROD];
#pragma unroll
	for (int n = 0; n < C::NPROD; n++) {
		const int i = C::LO + n, j = 9 + K_ - i;
		x[n] = a[i];
		y[n] = !SQR ? b[j] : (i < j ? a2[j] : a[i]);
	}
	ecamd_mad_chain_z<C::NPROD>(H[K_], x, y);   // the chain starts its accumulator: no register pair to clear
}
template <bool SQR, int... Ks> G29_FN void p25519_hi_columns(u64 *H, const u32 *a, const u32 *b, const u32 *a2, std::integer_sequence<int, Ks...>)
{
	(p25519_hi_column<SQR, Ks>(H, a, b, a2), ...);
}
template <bool SQR, int K_> G29_FN void p25519_lo_column(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u64 *H, u32 f, u32 f8)
{
	u64 acc2;
	Column<9, SQR, K_>::products(acc, acc2, a, b, a2);
	if constexpr (K_ >= 1 && K_ < 8) {
		G29_MAD2_VS(acc, (u32)H[K_], f, (u32)(H[K_ - 1] >> 32), f8);
	} else if constexpr (K_ < 8) {
		G29_MAD_VS(acc, (u32)H[K_], f);
	} else {
		G29_MAD_VS(acc, (u32)(H[K_ - 1] >> 32), f8);
	}
	if constexpr (K_ != 8) {
		out[K_] = (u32)acc & MASK;
		acc >>= W;
	}
}
template <bool SQR, int... Ks> G29_FN void p25519_lo_columns(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u64 *H, u32 f, u32 f8,
							     std::integer_sequence<int, Ks...>)
{
	(p25519_lo_column<SQR, Ks>(acc, out, a, b, a2, H, f, f8), ...);
}
#endif
template <bool SQR> G29_FN void mul_p25519(u32 *r, const u32 *a, const u32 *b)
{
	u32 a2[9];
	if (SQR) {
#pragma unroll
		for (int i = 0; i < 9; i++) {
			a2[i] = a[i] << 1;
		}
	}
	u32 f = 1216u;
#if defined(G29_P25519_CHAINFOLD)
	u32 h[9];
#if defined(__HIPCC__)
	asm volatile("" : "+s"(f));  // keep the folds MADs
#endif
	u64 acc = 0;
	p25519_hi_columns<SQR>(acc, h, a, b, a2, std::make_integer_sequence<int, 8>());      // columns 9..16
	h[8] = (u32)acc;
	acc = 0;
	p25519_columns<SQR>(acc, r, a, b, a2, h, f, std::make_integer_sequence<int, 9>());    // columns 0..8 (8: products and fold only)
#else
	u32 f8 = 9728u;
#if defined(__HIPCC__)
	asm volatile("" : "+s"(f), "+s"(f8));  // keep the folds MADs
#endif
	u64 H[8];
	p25519_hi_columns<SQR>(H, a, b, a2, std::make_integer_sequence<int, 8>());             // columns 9..16, no chain between them
	u64 acc = 0;
	p25519_lo_columns<SQR>(acc, r, a, b, a2, H, f, f8, std::make_integer_sequence<int, 9>());   // columns 0..8 (8: no column end)
#endif
	const u64 q = acc >> 23;                      // < 2^41
	r[8] = (u32)acc & ((1u << 23) - 1);
	const u64 w = (q << 4) + (q << 1) + q;        // 19 q < 2^46
	const u32 r0 = r[0] + ((u32)w & MASK);
	r[0] = r0 & MASK;
	r[1] += (u32)(w >> W) + (r0 >> W);
}

// a * c for a small wave-uniform constant c < 2^20 on the 2^255 - 19 flavour (X25519's a24 = 121665): NINE MADs on one carry chain
// instead of a 9 x 9 product against a constant whose eight upper limbs are zero.  Any limb bounds that fit 32 bits; the bits of
// the last column from 2^255 up fold (x 19) into limbs 0 and 1 exactly as in mul_p25519, so the result is in the same class.
template <bool DUMMY = true> G29_FN void mul_word_p25519(u32 *r, const u32 *a, u32 c)
{
#if defined(__HIPCC__)
	asm volatile("" : "+s"(c));
#endif
	u64 acc = 0;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		G29_MAD_VS(acc, a[i], c);
		if (i < 8) {
			r[i] = (u32)acc & MASK;
			acc >>= W;
		}
	}
	const u64 q = acc >> 23;                      // < 2^(32 + 20 - 23 + 1)
	r[8] = (u32)acc & ((1u << 23) - 1);
	const u64 w = (q << 4) + (q << 1) + q;
	const u32 r0 = r[0] + ((u32)w & MASK);
	r[0] = r0 & MASK;
	r[1] += (u32)(w >> W) + (r0 >> W);
}

// the same for the Goldilocks flavour (X448's a24 = 39081): sixteen MADs on one chain through both halves; the carry out of limb 15
// (multiples of 2^448 = phi + 1, below 2^24 here) goes to limbs 0 and 8 as in mul_p448
template <bool DUMMY = true> G29_FN void mul_word_p448(u32 *r, const u32 *a, u32 c)
{
#if defined(__HIPCC__)
	asm volatile("" : "+s"(c));
#endif
	u64 acc = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) {
		G29_MAD_VS(acc, a[i], c);
		r[i] = (u32)acc & MASK;
		acc >>= W;
	}
	const u32 chi = (u32)acc;                     // < 2^(32 + 20 - 28)
	const u32 r0 = r[0] + chi, r8 = r[8] + chi;
	r[0] = r0 & MASK;
	r[1] += r0 >> W;
	r[8] = r8 & MASK;
	r[9] += r8 >> W;
}

// ---- secp256k1 flavour (p = 2^256 - 2^32 - 977, nine 29-bit limbs, plain residues) ----
// The same order as above: high columns first (digits h[0..7], last carry h[8] < va vb 2^19 <= 2^31), then the low columns with the
// fold riding in them.  2^261 = 32 (2^32 + 977) = 2^8 2^29 + 31264: h[k] goes to column k (x 31264) and column k + 1 (x 256); for
// h[8] the second target is 2^261 again: columns 0 (x 256 x 31264 = 8003584) and 1 (x 2^16).  Column 8 -- with the carry of
// column 7 -- holds everything from 2^232 up; its bits from 24 up (q < 2^40) are multiples of 2^256 = 8 2^29 + 977 and go to limbs 0
// (977 q), 1 (8 q) and, as carries, 2: limb 2 below 2^29 + 2^15, the top limb below 2^24, value below 2p.
// 81 + 20 MADs (19 folds and 977 q) and 17 column ends (rounds 1-2: the same 101 MADs, but 26 column ends and a second pass).
template <bool SQR, int K_> G29_FN void k256_column(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u32 *h, const u32 *c)
{
	u64 acc2;
	Column<9, SQR, K_>::products(acc, acc2, a, b, a2);
	if constexpr (K_ < 9) {
		G29_MAD_VS(acc, h[K_], c[0]);               // x 31264
		if constexpr (K_ >= 1) {
			G29_MAD_VS(acc, h[K_ - 1], c[1]);   // x 256
		}
		if constexpr (K_ == 0) {
			G29_MAD_VS(acc, h[8], c[2]);        // x 8003584
		}
		if constexpr (K_ == 1) {
			G29_MAD_VS(acc, h[8], c[3]);        // x 65536
		}
	}
	if constexpr (K_ != 8) {
		out[K_ < 9 ? K_ : K_ - 9] = (u32)acc & MASK;
		acc >>= W;
	}
}
template <bool SQR, int... Ks> G29_FN void k256_columns(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, const u32 *h, const u32 *c,
							std::integer_sequence<int, Ks...>)
{
	(k256_column<SQR, Ks>(acc, out, a, b, a2, h, c), ...);
}
template <bool SQR, int... Ks> G29_FN void k256_hi_columns(u64 &acc, u32 *out, const u32 *a, const u32 *b, const u32 *a2, std::integer_sequence<int, Ks...>)
{
	(k256_column<SQR, 9 + Ks>(acc, out, a, b, a2, nullptr, nullptr), ...);
}
template <bool SQR> G29_FN void mul_k256(u32 *r, const u32 *a, const u32 *b)
{
	u32 a2[9], h[9];
	if (SQR) {
#pragma unroll
		for (int i = 0; i < 9; i++) {
			a2[i] = a[i] << 1;
		}
	}
	u32 c[5] = {31264u, 256u, 8003584u, 65536u, 977u};
#if defined(__HIPCC__)
	asm volatile("" : "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(c[3]), "+s"(c[4]));  // keep the folds MADs
#endif
	u64 acc = 0;
	k256_hi_columns<SQR>(acc, h, a, b, a2, std::make_integer_sequence<int, 8>());         // columns 9..16
	h[8] = (u32)acc;
	acc = 0;
	k256_columns<SQR>(acc, r, a, b, a2, h, c, std::make_integer_sequence<int, 9>());      // columns 0..8 (8: products and folds only)
	const u64 q = acc >> 24;                      // < 2^40
	r[8] = (u32)acc & ((1u << 24) - 1);
	u64 w0;                                       // 977 q < 2^50
	G29_MUL_VS(w0, (u32)q, c[4]);
	w0 += (u64)((u32)(q >> 32) * 977u) << 32;
	const u64 w1 = q << 3;                        // 8 q < 2^43
	const u32 r0 = r[0] + ((u32)w0 & MASK);
	r[0] = r0 & MASK;
	const u32 r1 = r[1] + (u32)(w0 >> W) + (r0 >> W) + ((u32)w1 & MASK);   // < 2^29 + 2^21 + 1 + 2^29
	r[1] = r1 & MASK;
	r[2] += (u32)(w1 >> W) + (r1 >> W);           // + < 2^14 + 2
}

template <int NL, bool SQR, int... Ks>
G29_FN void mul_columns(u64 &acc, u32 *m, u32 *r, const u32 *a, const u32 *b, const u32 *a2, const u32 *p, u32 mpinv,
			const int32_t *c384, std::integer_sequence<int, Ks...>)
{
	(mul_column<NL, SQR, Ks>(acc, m, r, a, b, a2, p, mpinv, c384), ...);
}

// r = a b / R mod p (lazy): product scanning with the reduction interleaved.  SQR: a == b, the
// off-diagonal products are taken once against the doubled operand.
template <int NL, bool SQR> G29_FN void mul_raw(u32 *r, const u32 *a, const u32 *b, const u32 *p, u32 mpinv)
{
	if constexpr (P448) {
		static_assert(NL == 16, "Goldilocks flavour: 16 limbs");
		mul_p448<SQR>(r, a, b);
	} else if constexpr (M521P) {
		static_assert(NL == 18, "plain Mersenne flavour: 18 limbs");
		mul_m521p<SQR>(r, a, b);
	} else if constexpr (P25519) {
		static_assert(NL == 9, "2^255 - 19 flavour: 9 limbs");
		mul_p25519<SQR>(r, a, b);
	} else if constexpr (K256) {
		static_assert(NL == 9, "secp256k1 flavour: 9 limbs");
		mul_k256<SQR>(r, a, b);
	} else {
		// Montgomery flavours: product scanning with the reduction interleaved, one pass over 2 NL - 1 columns
		u32 m[NL], a2[NL];
		if (SQR) {
#pragma unroll
			for (int i = 0; i < NL; i++) {
				a2[i] = a[i] << 1;
			}
		}
		u64 acc = 0;
		// signed sparse flavours: the signed digits of p + 1 (p - 1), kept opaque so that they stay MAD operands
		int32_t c384[4] = {SPARSE_C[0], SPARSE_C[1], SPARSE_C[2], SPARSE_C[3]};
#if defined(__HIPCC__)
		if (SPARSE) {
			asm volatile("" : "+s"(c384[0]), "+s"(c384[1]), "+s"(c384[2]), "+s"(c384[3]));
		}
#endif
		mul_columns<NL, SQR>(acc, m, r, a, b, a2, p, mpinv, c384, std::make_integer_sequence<int, 2 * NL - 1>());
		r[NL - 1] = (u32)acc;
	}
}

template <int NL> struct RawN {
	u32 l[NL];
};
template <int NL> G29_NOINLINE RawN<NL> mul_call(RawN<NL> a, RawN<NL> b, const CurveG<NL> *K)
{
	RawN<NL> r;
	mul_raw<NL, false>(r.l, a.l, b.l, K->p, K->mpinv);
	return r;
}
template <int NL> G29_NOINLINE RawN<NL> sqr_call(RawN<NL> a, const CurveG<NL> *K)
{
	RawN<NL> r;
	mul_raw<NL, true>(r.l, a.l, a.l, K->p, K->mpinv);
	return r;
}

template <class A, class B, int NLc> G29_FN typename MulOut<A::C::PBITS, mul_vb<A::C::PBITS>(A::VB, B::VB)>::type
mul(const A &a, const B &b, const CurveG<NLc> &K)
{
	typedef typename A::C C;
	static_assert(NLc == C::NL, "curve constants of the wrong width");
	static_assert(mul_fits<C::NL>(cmax(A::LB, A::TB), cmax(B::LB, B::TB)), "mul: column accumulator could overflow");
	static_assert(C::prod_ok(A::VB, B::VB), "mul: operands too large for R");
	typename MulOut<C::PBITS, mul_vb<C::PBITS>(A::VB, B::VB)>::type r;
	if constexpr (C::NL >= G29_CALL_FROM_NL) {
		RawN<C::NL> x, y;
#pragma unroll
		for (int i = 0; i < C::NL; i++) {
			x.l[i] = a.l[i];
			y.l[i] = b.l[i];
		}
		const RawN<C::NL> z = mul_call<C::NL>(x, y, &K);
#pragma unroll
		for (int i = 0; i < C::NL; i++) {
			r.l[i] = z.l[i];
		}
	} else {
		mul_raw<C::NL, false>(r.l, a.l, b.l, K.p, K.mpinv);
	}
	return r;
}

template <class A, int NLc> G29_FN typename MulOut<A::C::PBITS, mul_vb<A::C::PBITS>(A::VB, A::VB)>::type
sqr(const A &a, const CurveG<NLc> &K)
{
	typedef typename A::C C;
	static_assert(NLc == C::NL, "curve constants of the wrong width");
	static_assert(mul_fits<C::NL>(cmax(A::LB, A::TB), cmax(A::LB, A::TB)), "sqr: column accumulator could overflow");
	static_assert(2 * cmax(A::LB, A::TB) < (1ull << 32), "sqr: doubled limb does not fit 32 bits");
	static_assert(C::prod_ok(A::VB, A::VB), "sqr: operand too large for R");
	typename MulOut<C::PBITS, mul_vb<C::PBITS>(A::VB, A::VB)>::type r;
	if constexpr (C::NL >= G29_CALL_FROM_NL) {
		RawN<C::NL> x;
#pragma unroll
		for (int i = 0; i < C::NL; i++) {
			x.l[i] = a.l[i];
		}
		const RawN<C::NL> z = sqr_call<C::NL>(x, &K);
#pragma unroll
		for (int i = 0; i < C::NL; i++) {
			r.l[i] = z.l[i];
		}
	} else {
		mul_raw<C::NL, true>(r.l, a.l, a.l, K.p, K.mpinv);
	}
	return r;
}

// ---- add / small multiples / sub with bias / carry ----
template <class A, class B> G29_FN E<A::C::PBITS, A::LB + B::LB, A::TB + B::TB, A::VB + B::VB> add(const A &a, const B &b)
{
	E<A::C::PBITS, A::LB + B::LB, A::TB + B::TB, A::VB + B::VB> r;
#pragma unroll
	for (int i = 0; i < A::C::NL; i++) {
		r.l[i] = a.l[i] + b.l[i];
	}
	return r;
}

template <int K_, class A> G29_FN E<A::C::PBITS, K_ * A::LB, K_ * A::TB, K_ * A::VB> mul_small(const A &a)
{
	static_assert(K_ == 2 || K_ == 3 || K_ == 4 || K_ == 8, "small multiple");
	E<A::C::PBITS, K_ * A::LB, K_ * A::TB, K_ * A::VB> r;
#pragma unroll
	for (int i = 0; i < A::C::NL; i++) {
		r.l[i] = (K_ == 3) ? (a.l[i] + (a.l[i] << 1)) : (a.l[i] << (K_ == 2 ? 1 : (K_ == 4 ? 2 : 3)));
	}
	return r;
}

// bias(LOGC, S) = 2^LOGC p with limb i (< top) = digit_i + 2^(29+S) - (i ? 2^S : 0), top = digit_top - 2^S.
// Compile-time facts that hold for EVERY prime of PB bits (2^(PB-1) <= p < 2^PB):
template <int PB, int LOGC, int S> struct BiasB {
	typedef Cfg<PB> C;
	static constexpr u64 M = 1ull << (W + S);
	static constexpr u64 BORROW = 1ull << S;
	static constexpr u64 LOWMIN = M - BORROW;
	static constexpr u64 LOWMAX = M + MASK;
	// top digit of 2^LOGC p lies in [2^(LOGC + TOPSH - 1) - 1, 2^(LOGC + TOPSH)]
	static constexpr u64 TOPDIG_MIN = shl_floor(1, LOGC + C::TOPSH - 1) == 0 ? 0 : shl_floor(1, LOGC + C::TOPSH - 1) - 1;
	static constexpr u64 TOPDIG_MAX = shl_ceil(1, LOGC + C::TOPSH);
	static_assert(TOPDIG_MIN >= BORROW + 1, "bias: 2^LOGC p too small for the borrow");
	static constexpr u64 TOPMIN = TOPDIG_MIN - BORROW;
	static constexpr u64 TOPMAX = TOPDIG_MAX;
	static constexpr u64 VADD = 1ull << LOGC;
	static_assert(LOGC < 39, "bias multiple out of range");
};

template <int LOGC, int S, class A, class B, int NLc>
G29_FN E<A::C::PBITS, A::LB + BiasB<A::C::PBITS, LOGC, S>::LOWMAX, A::TB + BiasB<A::C::PBITS, LOGC, S>::TOPMAX,
	 A::VB + BiasB<A::C::PBITS, LOGC, S>::VADD>
sub(const A &a, const B &b, const CurveG<NLc> &K)
{
	typedef BiasB<A::C::PBITS, LOGC, S> BS;
	static_assert(bias_index(LOGC, S, A::C::BIAS_OFF) >= 0, "sub: no bias table for this (LOGC, S)");
	static_assert(BS::LOWMIN >= B::LB, "sub: bias limbs do not dominate b");
	static_assert(BS::TOPMIN >= B::TB, "sub: bias top limb does not dominate b");
	E<A::C::PBITS, A::LB + BS::LOWMAX, A::TB + BS::TOPMAX, A::VB + BS::VADD> r;
	constexpr int bi = bias_index(LOGC, S, A::C::BIAS_OFF);
#pragma unroll
	for (int i = 0; i < A::C::NL; i++) {
		r.l[i] = a.l[i] + (K.bias[bi][i] - b.l[i]);
	}
	return r;
}

// smallest tabulated multiple 2^LOGC p (for the given S) that dominates b limb by limb (which
// implies domination in value, so the difference is a non-negative number with non-negative limbs)
template <int PB, int S> constexpr int pick_logc(u64 lb_b, u64 tb_b)
{
	typedef Cfg<PB> C;
	for (int i = 0; i < NBIAS; i++) {
		if (BIAS_S[i] != S) {
			continue;
		}
		const int logc = BIAS_STEP[i] + C::BIAS_OFF;
		const u64 borrow = 1ull << S;
		const u64 lowmin = (1ull << (W + S)) - borrow;
		const u64 f = shl_floor(1, logc + C::TOPSH - 1);
		const u64 topdig_min = f == 0 ? 0 : f - 1;
		if (lowmin >= lb_b && topdig_min >= borrow + 1 && topdig_min - borrow >= tb_b) {
			return logc;
		}
	}
	return -1;
}

// a - b + (smallest sufficient tabulated multiple of p); S = 1 when b's limbs are < 2^30, 2 when < 2^31
template <int S, class A, class B, int NLc> G29_FN auto sub_auto(const A &a, const B &b, const CurveG<NLc> &K)
{
	constexpr int logc = pick_logc<A::C::PBITS, S>(B::LB, B::TB);
	if constexpr (logc < 0 && S == 1 && (NOHEAD || PLAIN9)) {
		// flavours whose products have lazy limbs a little over the limb width: the double of one is a little over 2^(W+1)
		return sub_auto<2>(a, b, K);
	} else {
		static_assert(logc >= 0, "sub_auto: no tabulated bias dominates b (carry it first?)");
		return sub<logc, S>(a, b, K);
	}
}

// One step of limb carries.  The integer value is kept, except in the no-headroom flavours (Goldilocks, plain Mersenne): there
// the top limb gives its bits from 28 up -- multiples of 2^448 = 2^224 + 1 resp. 2^521 = 1 (mod p) -- to limbs 0 (and 8), so that
// the result is the same residue with every limb near its width and a value below 2p.
template <class A> struct CarryT {
	static constexpr u64 LBO = MASK + (A::LB >> W) + (NOHEAD ? (A::TB >> 28) : 0);
	static constexpr u64 TBO = NOHEAD ? (TOPMASK28 + (A::LB >> W)) : (A::TB + (A::LB >> W));
	static_assert(!NOHEAD || LBO - MASK < (1ull << 20), "carry: limbs too loose for the value bound 2p");
	typedef E<A::C::PBITS, LBO, TBO, NOHEAD ? 2 : A::VB> type;
};
template <class A> G29_FN typename CarryT<A>::type carry(const A &a)
{
	constexpr int NL = A::C::NL;
	typename CarryT<A>::type r;
	r.l[0] = a.l[0] & MASK;
#pragma unroll
	for (int i = 1; i < NL - 1; i++) {
		r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> W);
	}
	if constexpr (NOHEAD) {
		// the top limb's bits from 28 up are multiples of 2^448 = 2^224 + 1 (Goldilocks) / of 2^521 = 1 (Mersenne)
		const u32 t = a.l[NL - 1] >> 28;
		r.l[NL - 1] = (a.l[NL - 1] & TOPMASK28) + (a.l[NL - 2] >> W);
		r.l[0] += t;
		if constexpr (P448) {
			r.l[NL / 2] += t;
		}
	} else {
		r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> W);
	}
	return r;
}

// ---- multiplication with the carries the operand bounds require (decided at compile time) ----
template <class A, class B, int NLc> G29_FN auto mulc(const A &a, const B &b, const CurveG<NLc> &K)
{
	constexpr int NL = A::C::NL;
	typedef typename CarryT<A>::type CA;
	typedef typename CarryT<B>::type CB;
	if constexpr (mul_fits<NL>(cmax(A::LB, A::TB), cmax(B::LB, B::TB))) {
		return mul(a, b, K);
	} else if constexpr (A::LB >= B::LB && mul_fits<NL>(cmax(CA::LB, CA::TB), cmax(B::LB, B::TB))) {
		return mul(carry(a), b, K);
	} else if constexpr (mul_fits<NL>(cmax(A::LB, A::TB), cmax(CB::LB, CB::TB))) {
		return mul(a, carry(b), K);
	} else if constexpr (mul_fits<NL>(cmax(CA::LB, CA::TB), cmax(B::LB, B::TB))) {
		return mul(carry(a), b, K);
	} else {
		return mul(carry(a), carry(b), K);
	}
}
template <class A, int NLc> G29_FN auto sqrc(const A &a, const CurveG<NLc> &K)
{
	constexpr int NL = A::C::NL;
	if constexpr (mul_fits<NL>(cmax(A::LB, A::TB), cmax(A::LB, A::TB)) && 2 * cmax(A::LB, A::TB) < (1ull << 32)) {
		return sqr(a, K);
	} else {
		return sqr(carry(a), K);
	}
}

// a * CST for a small constant (X25519's a24): nine MADs, result in the class of a multiplication result (mul_word_p25519)
template <u32 CST, class A> G29_FN typename MulOut<A::C::PBITS, 2>::type mul_word(const A &a)
{
	static_assert((P25519 && A::C::NL == 9) || (P448 && A::C::NL == 16), "mul_word: only the 2^255 - 19 and Goldilocks flavours have it");
	static_assert(CST < (1u << 20) && A::LB < (1ull << 32) && A::TB < (1ull << 32), "mul_word: operand out of range");
	static_assert(A::VB <= (1ull << 14), "mul_word: value out of range");
	typename MulOut<A::C::PBITS, 2>::type r;
	if constexpr (P448) {
		static_assert(A::TB < (1ull << 29), "mul_word: the top limb's bits from 28 up are not folded here");
		mul_word_p448(r.l, a.l, CST);
	} else {
		mul_word_p25519(r.l, a.l, CST);
	}
	return r;
}

// ---- exact zero test and canonical form of a multiplication result ----
// value < VB p with exact low digits: subtract p up to VB-1 times (VB is tiny: <= 3)
template <class A, int NLc> G29_FN void canonical_digits(u32 *d, const A &a, const CurveG<NLc> &K)
{
	constexpr int NL = A::C::NL;
	static_assert((A::LB == MASK || (NOHEAD && A::LB <= MASK + P448_MULX) || (P25519 && A::LB <= MASK + P25519_MULX) ||
		       (K256 && A::LB <= MASK + K256_MULX)) && A::VB <= 3,
		      "canonical_digits needs a multiplication result < 3p");
#pragma unroll
	for (int i = 0; i < NL; i++) {
		d[i] = a.l[i];
	}
	if constexpr (P25519) {
		// limb 1 is lazy and the value may reach 2^255 and a little more: exact carries with the bits from 2^255 up folded (x 19)
		// to limb 0, twice, then the value is below 2^255 < 2p
#pragma unroll
		for (int round = 0; round < 2; round++) {
			u32 c = 0;
#pragma unroll
			for (int i = 0; i < NL; i++) {
				const u32 x = d[i] + c;
				d[i] = x & (i < NL - 1 ? MASK : ((1u << 23) - 1));
				c = x >> (i < NL - 1 ? W : 23);
			}
			d[0] += 19u * c;
		}
	}
	if constexpr (K256) {
		// limb 2 is lazy and the value may reach 2^256 and a little more: exact carries with the bits from 2^256 up folded
		// (x 977 to limb 0, x 8 to limb 1), twice, then the value is below 2^256 < 2p
#pragma unroll
		for (int round = 0; round < 2; round++) {
			u32 c = 0;
#pragma unroll
			for (int i = 0; i < NL; i++) {
				const u32 x = d[i] + c;
				d[i] = x & (i < NL - 1 ? MASK : ((1u << 24) - 1));
				c = x >> (i < NL - 1 ? W : 24);
			}
			d[0] += 977u * c;
			d[1] += 8u * c;
		}
	}
	if constexpr (NOHEAD) {
		// some limbs are lazy and the value may reach 2^|p| and a little more: exact carries with the bits from 2^|p| up folded
		// to limb 0 (Goldilocks: and 8), twice (after the first round those limbs exceed their width by at most the folded amount;
		// the second round can only fold when everything below is nearly zero), then the value is below 2^|p| < 2p
#pragma unroll
		for (int round = 0; round < 2; round++) {
			u32 c = 0;
#pragma unroll
			for (int i = 0; i < NL; i++) {
				const u32 x = d[i] + c;
				d[i] = x & (i < NL - 1 ? MASK : TOPMASK28);
				c = x >> (i < NL - 1 ? W : 28);
			}
			d[0] += c;
			if constexpr (P448) {
				d[NL / 2] += c;
			}
		}
	}
#pragma unroll
	for (int rep = 0; rep + 1 < (int)A::VB; rep++) {
		u32 t[NL];
		u32 borrow = 0;
#pragma unroll
		for (int i = 0; i < NL; i++) {
			const u32 x = d[i] - K.p[i] - borrow;
			borrow = x >> 31;
			t[i] = (i < NL - 1) ? (x & MASK) : x;
		}
#pragma unroll
		for (int i = 0; i < NL; i++) {
			d[i] = borrow ? d[i] : t[i];
		}
	}
}

template <class A, int NLc> G29_FN bool is_zero_mulout(const A &a, const CurveG<NLc> &K)
{
	constexpr int NL = A::C::NL;
	u32 d[NL];
	canonical_digits(d, a, K);
	u32 z = 0;
#pragma unroll
	for (int i = 0; i < NL; i++) {
		z |= d[i];
	}
	return z == 0;
}

// ---- saturated 32-bit words <-> 29-bit limbs (NW words; value < p) ----
template <int PB, int NW> G29_FN E<PB, MASK, CANON_TB, 1> from_words(const u32 *w)
{
	constexpr int NL = Cfg<PB>::NL;
	E<PB, MASK, CANON_TB, 1> r;
#pragma unroll
	for (int i = 0; i < NL; i++) {
		const int bit = W * i, wi = bit >> 5, sh = bit & 31;
		u32 x = (wi < NW) ? (w[wi] >> sh) : 0u;
		if (sh > 32 - W && wi + 1 < NW) {
			x |= w[wi + 1] << (32 - sh);
		}
		r.l[i] = x & MASK;
	}
	return r;
}

// bits of w[] at or above W * NL: what from_words drops.  On the 18-limb unit of 2^521 - 1 (522 bits) a 66-octet coordinate has six of
// them, so r + k 2^522 would otherwise import as r; zero at compile time wherever the limbs cover the words.
template <int PB, int NW> G29_FN u32 words_excess(const u32 *w)
{
	constexpr int NL = Cfg<PB>::NL, TOP = W * NL;
	u32 x = 0;
	if (TOP < 32 * NW) {
#pragma unroll
		for (int wi = TOP >> 5; wi < NW; wi++) {
			x |= (wi == (TOP >> 5)) ? (w[wi] >> (TOP & 31)) : w[wi];
		}
	}
	return x;
}

template <int NL, int NW> G29_FN void to_words(u32 *w, const u32 *d)  // d: canonical digits
{
#pragma unroll
	for (int wi = 0; wi < NW; wi++) {
		const int bit = 32 * wi, li = bit / W, sh = bit - W * li;
		u32 x = (li < NL) ? (d[li] >> sh) : 0u;
		if (li + 1 < NL) {
			x |= d[li + 1] << (W - sh);
		}
		if (2 * W - sh < 32 && li + 2 < NL) {
			x |= d[li + 2] << (2 * W - sh);
		}
		w[wi] = x;
	}
}

template <class T, int NLc> G29_FN T constant(const u32 (&c)[NLc])
{
	static_assert(NLc == T::C::NL, "constant of the wrong width");
	T r;
#pragma unroll
	for (int i = 0; i < NLc; i++) {
		r.l[i] = c[i];
	}
	return r;
}

}  // namespace g29
