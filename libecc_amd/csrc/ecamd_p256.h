// libecc_amd/csrc/ecamd_p256.h -- secp256r1 group law on radix-2^29 lazy-reduced elements.
//
// Jacobian coordinates (x = X/Z^2, y = Y/Z^3), a = -3:
//   doubling  4M + 4S   (dbl-2001-b with Z3 = 2YZ and 8 gamma^2 = 2 (2 gamma)^2)
//   addition 12M + 4S   (add-1998-cmo-2)
// against 16 / 17 multiplications for the reference's complete projective formulas
// (curves/prj_pt.c:892-950, 971-1071 in /root/reference/src).  Jacobian addition is NOT
// complete; the scalar multiplication kernel (ecamd_p256_kernel.hip) keeps an explicit
// "accumulator is infinity" flag, tests the one remaining exceptional case (H = 0: P + P or
// P + (-P)) exactly through Z3, and hands such items to the complete-formula kernel, so the
// observable result is the reference's for EVERY input.
//
// Every intermediate is a bound-tracked u29::F<LB, TB, VB>; the carry()/fold() calls below are
// exactly the ones the static_asserts of ecamd_u29.h demand.
#pragma once
#include "ecamd_u29.h"

namespace p256 {
using namespace u29;

// constants in the Montgomery domain R = 2^261 (tools/u29_consts.py)
struct K {
	static constexpr u32 R2[9] = {0x00000c00, 0x00000000, 0x1fff0000, 0x1fdfffff, 0x1fbfffff,
				      0x1fffffff, 0x1fffffff, 0x1ffffffe, 0x00000013};
	static constexpr u32 ONE[9] = {0x00000020, 0x00000000, 0x00000000, 0x1fffc000, 0x1fffffff,
				       0x1fffffff, 0x1f7fffff, 0x03ffffff, 0x00000000};
	static constexpr u32 BM[9] = {0x1897bbfb, 0x1cdf6229, 0x018486c4, 0x01732821, 0x1dad59e0,
				      0x0abf7212, 0x1a06d110, 0x17721d20, 0x008600c3};
};

template <class T> U29_FN T constant(const u32 (&c)[9])
{
	T r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = c[i];
	}
	return r;
}

// loop-carried accumulator classes of the scalar multiplication
typedef F<MASK + 16, (2ull << 24), 17> FX;         // fold() output / Montgomery form of an input
typedef F<MASK + 16, (6ull << 24) + 16, 96> FY;     // carry() of a subtraction result, value < 6p
typedef F<2ull * MASK, (4ull << 24), 56> FZ;      // 2 * (multiplication result), value < 3.5p
// table entries keep Y folded (value < 17/16 p), so that the negation 2p - Y of a negative
// window digit stays small; FYsel is the common class of Y and its carried negation
typedef F<MASK + 16, (4ull << 24), 49> FYsel;
struct TabEnt {
	FX X;
	FX Y;
	FZ Z;
};

struct Jac {
	FX X;
	FY Y;
	FZ Z;
};

// x^(p-2): e2, e4, e8, e16, e32 ladder then the 2^32, 2^128, 2^32, 2^16, 2^8, 2^4, 2^2, 2^2 tail
// (255 squarings + 13 multiplications; exponent checked in tools/u29_consts.py)
U29_FN Fmul sqr_n(Fmul x, int n)
{
#pragma unroll 1
	for (int i = 0; i < n; i++) {
		x = weaken<Fmul>(sqr(x));
	}
	return x;
}
#define P256_MULW(a, b) weaken<Fmul>(mul(a, b))
U29_FN Fmul inv(const Fmul &x)
{
	const Fmul e2 = P256_MULW(sqr_n(x, 1), x);
	const Fmul e4 = P256_MULW(sqr_n(e2, 2), e2);
	const Fmul e8 = P256_MULW(sqr_n(e4, 4), e4);
	const Fmul e16 = P256_MULW(sqr_n(e8, 8), e8);
	const Fmul e32 = P256_MULW(sqr_n(e16, 16), e16);
	Fmul r = P256_MULW(sqr_n(e32, 32), x);
	r = P256_MULW(sqr_n(r, 128), e32);
	r = P256_MULW(sqr_n(r, 32), e32);
	r = P256_MULW(sqr_n(r, 16), e16);
	r = P256_MULW(sqr_n(r, 8), e8);
	r = P256_MULW(sqr_n(r, 4), e4);
	r = P256_MULW(sqr_n(r, 2), e2);
	r = P256_MULW(sqr_n(r, 2), x);
	return r;
}

// ---- doubling: (X, Y, Z) -> 2 (X, Y, Z) ----
U29_FN Jac dbl(const Jac &P)
{
	const auto delta = sqr(P.Z);                         // Z^2
	const auto gamma = sqr(P.Y);                         // Y^2
	const auto beta4 = mul(P.X, mul_small<4>(gamma));    // 4 X Y^2
	const auto t1 = sub<1, 0>(P.X, delta);               // X - Z^2 (+2p)
	const auto t2 = add(P.X, delta);                     // X + Z^2
	const auto alpha0 = mul(t1, t2);
	const auto alpha = carry(mul_small<3>(alpha0));      // 3 (X - Z^2)(X + Z^2)
	const auto a2 = sqr(alpha);
	const auto beta8 = mul_small<2>(beta4);
	const auto x3 = fold(sub<2, 1>(a2, beta8));          // alpha^2 - 8 beta
	const auto g4 = sqr(mul_small<2>(gamma));            // 4 gamma^2
	const auto g8 = mul_small<2>(g4);                    // 8 gamma^2
	const auto t4 = sub<1, 1>(beta4, x3);                // 4 beta - X3
	const auto y3a = mul(alpha, t4);
	const auto y3 = carry(sub<2, 1>(y3a, g8));
	const auto z3 = mul_small<2>(mul(P.Y, P.Z));         // 2 Y Z
	Jac R;
	R.X = weaken<FX>(x3);
	R.Y = weaken<FY>(y3);
	R.Z = weaken<FZ>(z3);
	return R;
}

// ---- addition: (X1, Y1, Z1) + (X2, Y2, Z2), Z2 arbitrary; returns Z3 as the raw
//      multiplication result too (exact digits) for the H == 0 test ----
template <class FX2, class FY2, class FZ2>
U29_FN Jac add_jac(const Jac &P, const FX2 &X2, const FY2 &Y2, const FZ2 &Z2, bool &h_is_zero)
{
	const auto z1z1 = sqr(P.Z);
	const auto z2z2 = sqr(Z2);
	const auto u1 = mul(P.X, z2z2);
	const auto u2 = mul(X2, z1z1);
	const auto s1 = mul(mul(P.Y, Z2), z2z2);
	const auto s2 = mul(mul(Y2, P.Z), z1z1);
	const auto h = carry(sub<1, 0>(u2, u1));
	const auto r = carry(sub<1, 0>(s2, s1));
	const auto hh = sqr(h);
	const auto hhh = mul(h, hh);
	const auto v = mul(u1, hh);
	const auto r2 = sqr(r);
	const auto x3 = fold(sub<2, 2>(r2, add(hhh, mul_small<2>(v))));
	const auto t5 = sub<1, 1>(v, x3);
	const auto m1 = mul(r, t5);
	const auto m2 = mul(s1, hhh);
	const auto y3 = carry(sub<1, 0>(m1, m2));
	const auto z3 = mul(mul(P.Z, Z2), h);
	h_is_zero = is_zero_mulout(z3);
	Jac R;
	R.X = weaken<FX>(x3);
	R.Y = weaken<FY>(y3);
	R.Z = weaken<FZ>(z3);
	return R;
}

// ---- mixed addition: (X1, Y1, Z1) + affine (x2, y2), 8M + 3S (madd): the window table is made
//      affine by k_p256_affine (one shared inversion per 8 items), which saves 4M + 1S per window ----
typedef F<(1ull << 29) + MASK, (3ull << 24), 48> FYaff;  // y2 or 2p - y2 of an affine (canonical) table entry
template <class FY2> U29_FN Jac madd(const Jac &P, const Fcanon &X2, const FY2 &Y2, bool &h_is_zero)
{
	const auto z1z1 = sqr(P.Z);
	const auto u2 = mul(X2, z1z1);
	const auto s2 = mul(mul(Y2, P.Z), z1z1);
	const auto h = carry(sub<1, 1>(u2, P.X));
	const auto r = carry(sub<3, 1>(s2, P.Y));
	const auto hh = sqr(h);
	const auto hhh = mul(h, hh);
	const auto v = mul(P.X, hh);
	const auto r2 = sqr(r);
	const auto x3 = fold(sub<2, 2>(r2, add(hhh, mul_small<2>(v))));
	const auto t5 = sub<1, 1>(v, x3);
	const auto y3 = carry(sub<1, 0>(mul(r, t5), mul(P.Y, hhh)));
	const auto z3 = mul(P.Z, h);
	h_is_zero = is_zero_mulout(z3);
	Jac R;
	R.X = weaken<FX>(x3);
	R.Y = weaken<FY>(y3);
	R.Z = weaken<FZ>(z3);
	return R;
}
U29_FN FYaff neg_aff(const Fcanon &y)
{
	F<0, 0, 0> zero;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		zero.l[i] = 0;
	}
	return weaken<FYaff>(sub<1, 0>(zero, y));
}

// 2p - Y, limbs re-normalised: the negated table entry of a negative window digit
U29_FN FYsel neg_y(const FX &y)
{
	F<0, 0, 0> zero;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		zero.l[i] = 0;
	}
	return weaken<FYsel>(carry(sub<1, 1>(zero, y)));
}

U29_FN TabEnt to_tab(const Jac &P)
{
	TabEnt T;
	T.X = P.X;
	T.Y = weaken<FX>(fold(P.Y));
	T.Z = P.Z;
	return T;
}

}  // namespace p256
