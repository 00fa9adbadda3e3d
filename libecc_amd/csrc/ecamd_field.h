// libecc_amd/csrc/ecamd_field.h -- prime-field arithmetic on gfx950, one element per lane.
//
// Replaces the reference's nn/fp hot loop (paths relative to /root/reference/src):
//   nn_mul_redc1   nn/nn_mul_redc1.c:124-218   CIOS Montgomery multiplication  -> fe_mul / fe_sqr
//   nn_mod_add/sub nn/nn_add.c:337,398         modular add / sub               -> fe_add / fe_sub
//   fp_inv         fp/fp_mul.c:51-68           x^(p-2)                         -> fe_inv
//
// Design (gfx950 / CDNA4):
//   * An element is NW 32-bit words held in VGPRs (struct by value: the multiply is a real,
//     non-inlined function so its body exists once in the 64 KB instruction cache; operands
//     travel in registers through the AMDGPU calling convention).
//   * The only multiplier on the path is v_mad_u64_u32 (32x32+64 -> 64 plus a per-lane
//     carry-out bit into an SGPR pair).  Products are accumulated column-wise (product
//     scanning) into a 96-bit accumulator {lo64, hi32}: one v_mad_u64_u32 on lo64 and one
//     v_addc_co_u32 that folds the carry-out into hi32 -- 2 VALU ops per 32x32 product and
//     no zero-extension moves (hipcc's own lowering of `u64 += u32*u32 + u32` needs ~4.5).
//   * gfx90a+/gfx950 hazard: a VALU write of an SGPR/VCC needs 2 wait states before a VALU
//     reads it (LLVM GCNHazardRecognizer, VALUWriteSGPRVALURead).  hipcc does not look inside
//     asm, so each ECAMD_MACn statement orders its MADs before its ADDCs (>= 2 instructions
//     between a carry's producer and consumer) and pads with s_nop only when n < 3.
//   * Montgomery radix is R = 2^(32*NW) (not the reference's 2^(64*n) when NW is odd, e.g.
//     P-521: 17 words vs 9 x 64 bits).  Montgomery values never leave the device and the fully
//     reduced plain result of every operation is unique, so outputs stay bit-exact.
//   * Curve/field constants live in __constant__ memory (scalar loads -> SGPR operands).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

#define ECAMD_MAX_SLOTS 16

template <int NW> struct Fe { u32 v[NW]; };

// Field + curve constants for one curve slot (filled by the host, ecamd_host.cpp).
template <int NW> struct CurveK {
	u32 p[NW];    // modulus
	u32 r2[NW];   // R^2 mod p
	u32 one[NW];  // R mod p (Montgomery 1)
	u32 pm2[NW];  // p - 2 (inversion exponent)
	u32 a[NW];    // a   * R mod p
	u32 b[NW];    // b   * R mod p
	u32 b3[NW];   // 3b  * R mod p
	u32 fix64[NW];  // R^2 * 2^(-64*ceil(NW/2)) mod p: maps our radix to the reference's (== R when NW even)
	u32 mpinv;    // -p^-1 mod 2^32
	u32 pbits;    // bitlen(p)
	u32 a_is_m3;  // a == p - 3
	u32 fix_is_id;  // fix64 == R (NW even): nothing to fix
};

template <int NW> struct CurveSlots { CurveK<NW> s[ECAMD_MAX_SLOTS]; };

template <int NW> struct ConstTab;
#define ECAMD_DECL_CONST(NW) \
	extern __constant__ CurveSlots<NW> g_curves_##NW; \
	template <> struct ConstTab<NW> { \
		static __device__ __forceinline__ const CurveK<NW> &get(int slot) { return g_curves_##NW.s[slot]; } \
	};
ECAMD_DECL_CONST(6)
ECAMD_DECL_CONST(7)
ECAMD_DECL_CONST(8)
ECAMD_DECL_CONST(10)
ECAMD_DECL_CONST(12)
ECAMD_DECL_CONST(14)
ECAMD_DECL_CONST(16)
ECAMD_DECL_CONST(17)

// ------------------------------------------------------------------------------------------
// {lo64, hi32} += sum of n products.  _VV: both factors in VGPRs.  _VS: second factor in an
// SGPR (modulus words; one constant-bus read per instruction is allowed on gfx9).
// ------------------------------------------------------------------------------------------
#define ECAMD_MAC1(K2, lo, hi, a0, b0) \
	asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\t" \
		     "v_addc_co_u32 %1, vcc, 0, %1, vcc" \
		     : "+v"(lo), "+v"(hi) : "v"(a0), K2(b0) : "vcc")
#define ECAMD_MAC2(K2, lo, hi, a0, b0, a1, b1) \
	do { u64 c0_; \
	asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\ts_nop 0\n\t" \
		     "v_addc_co_u32 %1, %2, 0, %1, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" \
		     : "+v"(lo), "+v"(hi), "=&s"(c0_) : "v"(a0), K2(b0), "v"(a1), K2(b1) : "vcc"); } while (0)
#define ECAMD_MAC3(K2, lo, hi, a0, b0, a1, b1, a2, b2) \
	do { u64 c0_, c1_; \
	asm volatile("v_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_mad_u64_u32 %0, %3, %6, %7, %0\n\t" \
		     "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t" \
		     "v_addc_co_u32 %1, %2, 0, %1, %2\n\tv_addc_co_u32 %1, %3, 0, %1, %3\n\t" \
		     "v_addc_co_u32 %1, vcc, 0, %1, vcc" \
		     : "+v"(lo), "+v"(hi), "=&s"(c0_), "=&s"(c1_) \
		     : "v"(a0), K2(b0), "v"(a1), K2(b1), "v"(a2), K2(b2) : "vcc"); } while (0)
#define ECAMD_MAC4(K2, lo, hi, a0, b0, a1, b1, a2, b2, a3, b3) \
	do { u64 c0_, c1_, c2_; \
	asm volatile("v_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_mad_u64_u32 %0, %3, %7, %8, %0\n\t" \
		     "v_mad_u64_u32 %0, %4, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" \
		     "v_addc_co_u32 %1, %2, 0, %1, %2\n\tv_addc_co_u32 %1, %3, 0, %1, %3\n\t" \
		     "v_addc_co_u32 %1, %4, 0, %1, %4\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" \
		     : "+v"(lo), "+v"(hi), "=&s"(c0_), "=&s"(c1_), "=&s"(c2_) \
		     : "v"(a0), K2(b0), "v"(a1), K2(b1), "v"(a2), K2(b2), "v"(a3), K2(b3) : "vcc"); } while (0)

struct Acc96 {
	u64 lo;
	u32 hi;
};

// x[i]*y[i], i < N, all VGPR operands
template <int N> static __device__ __forceinline__ void mac_vv(Acc96 &acc, const u32 *x, const u32 *y)
{
	constexpr int R = N % 4, B = N - R;
#pragma unroll
	for (int i = 0; i < B; i += 4) {
		ECAMD_MAC4("v", acc.lo, acc.hi, x[i], y[i], x[i + 1], y[i + 1], x[i + 2], y[i + 2], x[i + 3], y[i + 3]);
	}
	if (R == 3) {
		ECAMD_MAC3("v", acc.lo, acc.hi, x[B], y[B], x[B + 1], y[B + 1], x[B + 2], y[B + 2]);
	} else if (R == 2) {
		ECAMD_MAC2("v", acc.lo, acc.hi, x[B], y[B], x[B + 1], y[B + 1]);
	} else if (R == 1) {
		ECAMD_MAC1("v", acc.lo, acc.hi, x[B], y[B]);
	}
}
// x[i] (VGPR) * y[i] (SGPR), i < N
template <int N> static __device__ __forceinline__ void mac_vs(Acc96 &acc, const u32 *x, const u32 *y)
{
	constexpr int R = N % 4, B = N - R;
#pragma unroll
	for (int i = 0; i < B; i += 4) {
		ECAMD_MAC4("s", acc.lo, acc.hi, x[i], y[i], x[i + 1], y[i + 1], x[i + 2], y[i + 2], x[i + 3], y[i + 3]);
	}
	if (R == 3) {
		ECAMD_MAC3("s", acc.lo, acc.hi, x[B], y[B], x[B + 1], y[B + 1], x[B + 2], y[B + 2]);
	} else if (R == 2) {
		ECAMD_MAC2("s", acc.lo, acc.hi, x[B], y[B], x[B + 1], y[B + 1]);
	} else if (R == 1) {
		ECAMD_MAC1("s", acc.lo, acc.hi, x[B], y[B]);
	}
}

static __device__ __forceinline__ u32 acc_shift(Acc96 &acc)
{
	u32 w = (u32)acc.lo;
	acc.lo = (acc.lo >> 32) | ((u64)acc.hi << 32);
	acc.hi = 0;
	return w;
}

// t (NW words + top word 0/1) -> [0, p): one conditional subtraction
// (the reference's final nn_cmp + nn_cnd_sub, nn/nn_mul_redc1.c:210-211)
template <int NW> static __device__ __forceinline__ Fe<NW> fe_cond_sub(const u32 *t, u32 top, const u32 *p)
{
	u32 d[NW];
	u32 borrow = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		u64 x = (u64)t[j] - p[j] - borrow;
		d[j] = (u32)x;
		borrow = (u32)(x >> 63);
	}
	const bool ge = (top != 0) | (borrow == 0);
	Fe<NW> r;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		r.v[j] = ge ? d[j] : t[j];
	}
	return r;
}

// ------------------------------------------------------------------------------------------
// Generic Montgomery multiplication, finely integrated product scanning:
//   column k:  acc += sum_{i+j=k} a_i b_j + sum_{i+j=k, i<k} m_i p_j
//              k < NW : m_k = acc.lo32 * mpinv ; acc += m_k p_0 ; shift (low word becomes 0)
//              k >= NW: t_{k-NW} = acc.lo32 ; shift
// 2*NW^2 + NW products; result a*b*2^(-32*NW) mod p, fully reduced.
// ------------------------------------------------------------------------------------------
template <int NW, int K>
static __device__ __forceinline__ void fips_col(Acc96 &acc, const Fe<NW> &a, const Fe<NW> &b, u32 (&m)[NW],
						const u32 (&p)[NW], u32 mpinv, u32 (&t)[NW + 1])
{
	constexpr int lo_i = (K < NW) ? 0 : (K - NW + 1);
	constexpr int hi_i = (K < NW) ? K : (NW - 1);
	constexpr int n_ab = hi_i - lo_i + 1;
	{
		u32 x[n_ab], y[n_ab];
#pragma unroll
		for (int i = 0; i < n_ab; i++) {
			x[i] = a.v[lo_i + i];
			y[i] = b.v[K - lo_i - i];
		}
		mac_vv<n_ab>(acc, x, y);
	}
	constexpr int mhi = (K < NW) ? (K - 1) : (NW - 1);
	constexpr int n_mp = mhi - lo_i + 1;
	if constexpr (n_mp > 0) {
		u32 x[n_mp], y[n_mp];
#pragma unroll
		for (int i = 0; i < n_mp; i++) {
			x[i] = m[lo_i + i];
			y[i] = p[K - lo_i - i];
		}
		mac_vs<n_mp>(acc, x, y);
	}
	if constexpr (K < NW) {
		m[K] = (u32)acc.lo * mpinv;
		ECAMD_MAC1("s", acc.lo, acc.hi, m[K], p[0]);
		(void)acc_shift(acc);
	} else {
		t[K - NW] = acc_shift(acc);
	}
	if constexpr (K + 1 < 2 * NW - 1) {
		fips_col<NW, K + 1>(acc, a, b, m, p, mpinv, t);
	}
}

template <int NW> static __device__ __forceinline__ Fe<NW> fe_mul_body(const Fe<NW> &a, const Fe<NW> &b, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	u32 p[NW];
#pragma unroll
	for (int i = 0; i < NW; i++) {
		p[i] = K.p[i];
	}
	const u32 mpinv = K.mpinv;
	u32 m[NW], t[NW + 1];
	Acc96 acc = {0, 0};
	fips_col<NW, 0>(acc, a, b, m, p, mpinv, t);
	t[NW - 1] = (u32)acc.lo;
	t[NW] = (u32)(acc.lo >> 32);
	return fe_cond_sub<NW>(t, t[NW], p);
}

template <int NW> __device__ __noinline__ Fe<NW> fe_mul(Fe<NW> a, Fe<NW> b, int slot_)
{
	const int slot = __builtin_amdgcn_readfirstlane(slot_);
	return fe_mul_body<NW>(a, b, slot);
}

template <int NW> static __device__ __forceinline__ Fe<NW> fe_sqr(const Fe<NW> &a, int slot)
{
	return fe_mul<NW>(a, a, slot);
}

// (a + b) mod p : add, compare, conditional subtract (nn_mod_add, nn/nn_add.c:337)
template <int NW> __device__ __noinline__ Fe<NW> fe_add(Fe<NW> a, Fe<NW> b, int slot_)
{
	const int slot = __builtin_amdgcn_readfirstlane(slot_);
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	u32 t[NW], p[NW];
	u32 carry = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		p[j] = K.p[j];
		u64 x = (u64)a.v[j] + b.v[j] + carry;
		t[j] = (u32)x;
		carry = (u32)(x >> 32);
	}
	return fe_cond_sub<NW>(t, carry, p);
}

// (a - b) mod p : subtract, add p back on borrow (nn_mod_sub, nn/nn_add.c:398)
template <int NW> __device__ __noinline__ Fe<NW> fe_sub(Fe<NW> a, Fe<NW> b, int slot_)
{
	const int slot = __builtin_amdgcn_readfirstlane(slot_);
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	u32 t[NW];
	u32 borrow = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		u64 x = (u64)a.v[j] - b.v[j] - borrow;
		t[j] = (u32)x;
		borrow = (u32)(x >> 63);
	}
	const u32 mask = 0u - borrow;
	Fe<NW> r;
	u32 carry = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		u64 x = (u64)t[j] + (K.p[j] & mask) + carry;
		r.v[j] = (u32)x;
		carry = (u32)(x >> 32);
	}
	return r;
}

template <int NW> static __device__ __forceinline__ bool fe_is_zero(const Fe<NW> &a)
{
	u32 acc = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		acc |= a.v[j];
	}
	return acc == 0;
}

template <int NW> static __device__ __forceinline__ bool fe_eq(const Fe<NW> &a, const Fe<NW> &b)
{
	u32 acc = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		acc |= a.v[j] ^ b.v[j];
	}
	return acc == 0;
}

template <int NW> static __device__ __forceinline__ Fe<NW> fe_const(const u32 *src)
{
	Fe<NW> r;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		r.v[j] = src[j];
	}
	return r;
}

template <int NW> static __device__ __forceinline__ Fe<NW> fe_zero()
{
	Fe<NW> r;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		r.v[j] = 0;
	}
	return r;
}

template <int NW> static __device__ __forceinline__ Fe<NW> fe_select(bool c, const Fe<NW> &a, const Fe<NW> &b)
{
	Fe<NW> r;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		r.v[j] = c ? a.v[j] : b.v[j];
	}
	return r;
}

// a < p ?  (fp_import_from_buf rejects values >= p, fp/fp.c:441-442)
template <int NW> static __device__ __forceinline__ bool fe_lt_p(const Fe<NW> &a, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	u32 borrow = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		u64 x = (u64)a.v[j] - K.p[j] - borrow;
		borrow = (u32)(x >> 63);
	}
	return borrow != 0;
}

// plain -> Montgomery, Montgomery -> plain (fp_redcify / fp_unredcify, fp/fp_mul_redc1.c:62,79)
template <int NW> static __device__ __forceinline__ Fe<NW> fe_to_mont(const Fe<NW> &a, int slot)
{
	return fe_mul<NW>(a, fe_const<NW>(ConstTab<NW>::get(slot).r2), slot);
}
template <int NW> static __device__ __forceinline__ Fe<NW> fe_from_mont(const Fe<NW> &a, int slot)
{
	Fe<NW> one = fe_zero<NW>();
	one.v[0] = 1;
	return fe_mul<NW>(a, one, slot);
}

// x^(p-2) in the Montgomery domain, left-to-right binary (exponent is wave-uniform: the
// branch is uniform).  Stands in for nn_modinv_fermat_redc (nn/nn_modinv.c:538); the inverse
// is unique so the value equals the reference's.
template <int NW> static __device__ Fe<NW> fe_inv(const Fe<NW> &x, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	Fe<NW> r = fe_const<NW>(K.one);
	for (int i = (int)K.pbits - 1; i >= 0; i--) {
		r = fe_sqr<NW>(r, slot);
		if ((K.pm2[i >> 5] >> (i & 31)) & 1) {
			r = fe_mul<NW>(r, x, slot);
		}
	}
	return r;
}
