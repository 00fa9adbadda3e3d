// libecc_amd/csrc/ecamd_jacg.h -- Jacobian group law over ecamd_u29g.h, any short-Weierstrass
// curve y^2 = x^3 + a x + b (a = -3 shortcut and generic a), same structure as ecamd_p256.h.
//   doubling  a = -3: 4M + 4S     a = 0: 3M + 4S     generic a: 4M + 6S (M = 3 X^2 + a Z^4)
//   addition 12M + 4S (add-1998-cmo-2)
// Thanks to the headroom limb no value fold is needed; carry() is inserted where the
// static_asserts of ecamd_u29g.h demand it.
#pragma once
#include "ecamd_u29g.h"

namespace jacg {
using namespace g29;

template <int PB> struct Cls {
	typedef Cfg<PB> C;
	// generous cover for the largest bias the formulas below ever need ...
	static constexpr int LC = NOHEAD ? 0 : pick_logc<PB, 2>(3ull * MASK, C::top_from_vb(8)) + 4;
	static_assert(LC >= (NOHEAD ? 0 : 4), "no bias table large enough for this field size");
	// ... bounds every loop-carried coordinate: carried limbs, value < 4 * 2^LC p
	// (2^255 - 19 flavour: no headroom limb, the top limb holds values below 512 p, and products want
	// va * vb <= 2^14; the formulas stay well inside 48 p)
	// (no-headroom flavours, Goldilocks and plain Mersenne: carry() folds the top limb, so every carried coordinate and every
	// multiplication result is below 2p with limbs a little over their width; bias multiples are 4p and 8p)
	static constexpr u64 VA = NOHEAD ? 2 : (PLAIN9 ? 48 : (4ull << LC));
	// what a carried limb (no headroom, 2^255 - 19: or a product's lazy limb) may exceed the mask by
	static constexpr u64 SLACK = NOHEAD ? (1ull << 11) : (P25519 ? (1ull << 18) : (K256 ? (1ull << 16) : 8));
	typedef E<PB, MASK + SLACK, NOHEAD ? (TOPMASK28 + SLACK) : C::top_from_vb(VA), VA> FA;
	typedef typename MulOut<PB, 2>::type FM;  // multiplication result (value < 2p, exact low digits)
	typedef E<PB, MASK, CANON_TB, 1> FC;      // canonical constant
};

template <int PB> struct Jac {
	typename Cls<PB>::FA X, Y, Z;
};

#define JG_K const CurveG<Cfg<PB>::NL> &K

// Tight class for the accumulator of the mixed-addition loop (madd_jac below subtracts X1 and Y1 themselves, so their
// declared bound becomes the bias, and everything after it, of that addition): doublings and mixed additions map it into
// itself for every field size (checked by the static_asserts of weaken / sub_auto / mul).
// (p = 2^255 - 19 on nine limbs: products want v_a v_b <= 2^14, met with the accumulator below 20p; secp256k1's flavour,
// v_a v_b <= 2^12, has no room for it -- H = U2 - X1 needs the 64p bias -- and keeps the Jacobian-table kernel)
constexpr bool HAVE_MADD = !K256;
template <int PB> struct ClsT {
	static constexpr u64 VT = P25519 ? 20 : (PLAIN ? Cls<PB>::VA : (128ull << Cfg<PB>::BIAS_OFF));
	typedef E<PB, MASK + Cls<PB>::SLACK, NOHEAD ? (TOPMASK28 + Cls<PB>::SLACK) : Cfg<PB>::top_from_vb(VT), VT> FT;
};
template <int PB> struct JacT {
	typename ClsT<PB>::FT X, Y, Z;
};

template <int PB, class J> G29_FN J dbl_any(const J &P, JG_K)
{
	typedef decltype(J::X) FA;
	typedef typename Cls<PB>::FC FC;
	const auto yy = sqrc(P.Y, K);
	const auto s4 = mulc(P.X, mul_small<4>(yy), K);  // 4 X Y^2
	FA m;  // M = 3 X^2 + a Z^4
	if (K.a_is_zero) {
		m = weaken<FA>(carry(mul_small<3>(sqrc(P.X, K))));                     // a = 0: 3 X^2 (3M + 4S in all)
	} else if (K.a_is_m3) {
		const auto zz = sqrc(P.Z, K);
		const auto t1 = carry(sub_auto<1>(P.X, zz, K));
		const auto t2 = carry(add(P.X, zz));
		m = weaken<FA>(carry(mul_small<3>(mulc(t1, t2, K))));  // 3 (X - Z^2)(X + Z^2)
	} else {
		const auto zz = sqrc(P.Z, K);
		const auto xx = sqrc(P.X, K);
		const auto az4 = mulc(sqrc(zz, K), constant<FC>(K.a), K);
		m = weaken<FA>(carry(add(mul_small<3>(xx), az4)));
	}
	const auto m2 = sqrc(m, K);
	const auto x3 = carry(sub_auto<1>(m2, mul_small<2>(s4), K));  // M^2 - 8 X Y^2
	const auto y8 = mul_small<2>(sqrc(mul_small<2>(yy), K));       // 8 Y^4
	const auto t4 = carry(sub_auto<1>(s4, x3, K));                // 4 X Y^2 - X3
	const auto y3 = carry(sub_auto<1>(mulc(m, t4, K), y8, K));
	const auto z3 = carry(mul_small<2>(mulc(P.Y, P.Z, K)));        // 2 Y Z
	J R;
	R.X = weaken<FA>(x3);
	R.Y = weaken<FA>(y3);
	R.Z = weaken<FA>(z3);
	return R;
}
template <int PB> G29_FN Jac<PB> dbl(const Jac<PB> &P, JG_K) { return dbl_any<PB, Jac<PB>>(P, K); }
template <int PB> G29_FN JacT<PB> dbl(const JacT<PB> &P, JG_K) { return dbl_any<PB, JacT<PB>>(P, K); }

// (X1, Y1, Z1) + (X2, Y2, Z2); h_is_zero <=> the x coordinates coincide (P + P or P + (-P))
template <int PB>
G29_FN Jac<PB> add_jac(const Jac<PB> &P, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2,
		       const typename Cls<PB>::FA &Z2, bool &h_is_zero, JG_K)
{
	typedef typename Cls<PB>::FA FA;
	const auto z1z1 = sqrc(P.Z, K);
	const auto z2z2 = sqrc(Z2, K);
	const auto u1 = mulc(P.X, z2z2, K);
	const auto u2 = mulc(X2, z1z1, K);
	const auto s1 = mulc(mulc(P.Y, Z2, K), z2z2, K);
	const auto s2 = mulc(mulc(Y2, P.Z, K), z1z1, K);
	const auto h = carry(sub_auto<1>(u2, u1, K));
	const auto r = carry(sub_auto<1>(s2, s1, K));
	const auto hh = sqrc(h, K);
	const auto hhh = mulc(h, hh, K);
	const auto v = mulc(u1, hh, K);
	const auto r2 = sqrc(r, K);
	const auto x3 = carry(sub_auto<2>(r2, add(hhh, mul_small<2>(v)), K));
	const auto t5 = carry(sub_auto<1>(v, x3, K));
	const auto y3 = carry(sub_auto<1>(mulc(r, t5, K), mulc(s1, hhh, K), K));
	const auto z3 = mulc(mulc(P.Z, Z2, K), h, K);
	h_is_zero = is_zero_mulout(z3, K);
	Jac<PB> R;
	R.X = weaken<FA>(x3);
	R.Y = weaken<FA>(y3);
	R.Z = weaken<FA>(z3);
	return R;
}

// (X1, Y1, Z1) + (x2, y2, 1) on the general accumulator class: add_jac without its Z2 terms -- X1 and Y1 enter through ONE multiplication
// by one each (they are lazily reduced: the subtractions want them below 2p), 10M + 3S instead of 12M + 4S.  The bucket accumulation of
// the Schnorr-type batch equation (k_bkt_accum_g) adds affine points only.
template <int PB>
G29_FN Jac<PB> add_aff(const Jac<PB> &P, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2, bool &h_is_zero, JG_K)
{
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FC FC;
	const FC onec = constant<FC>(K.one);
	const auto z1z1 = sqrc(P.Z, K);
	const auto u1 = mulc(P.X, onec, K);
	const auto u2 = mulc(X2, z1z1, K);
	const auto s1 = mulc(P.Y, onec, K);
	const auto s2 = mulc(mulc(Y2, P.Z, K), z1z1, K);
	const auto h = carry(sub_auto<1>(u2, u1, K));
	const auto r = carry(sub_auto<1>(s2, s1, K));
	const auto hh = sqrc(h, K);
	const auto hhh = mulc(h, hh, K);
	const auto v = mulc(u1, hh, K);
	const auto r2 = sqrc(r, K);
	const auto x3 = carry(sub_auto<2>(r2, add(hhh, mul_small<2>(v)), K));
	const auto t5 = carry(sub_auto<1>(v, x3, K));
	const auto y3 = carry(sub_auto<1>(mulc(r, t5, K), mulc(s1, hhh, K), K));
	const auto z3 = mulc(P.Z, h, K);
	if constexpr (decltype(z3)::VB <= 3) {
		h_is_zero = is_zero_mulout(z3, K);
	} else {
		h_is_zero = is_zero_mulout(mulc(z3, onec, K), K);   // (the dense 448-bit unit: a product of two lazily reduced factors may reach 4p)
	}
	Jac<PB> R;
	R.X = weaken<FA>(x3);
	R.Y = weaken<FA>(y3);
	R.Z = weaken<FA>(z3);
	return R;
}

// (X1, Y1, Z1) + (x2, y2) with the second point affine (Z2 = 1): 8M + 3S (the Z2 terms of add_jac dropped).
// No exceptional-pair flag: H = 0 (the same x: P + P or P + (-P)) gives Z3 = Z1 H = 0, and a zero Z survives every later
// doubling (Z3 = 2 Y Z) and addition (Z3 = Z1 H), so ONE exact test of the final Z catches it (the callers' redo lane).
template <int PB>
G29_FN JacT<PB> madd_jac(const JacT<PB> &P, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2, JG_K)
{
	typedef typename ClsT<PB>::FT FT;
	const auto z1z1 = sqrc(P.Z, K);
	const auto u2 = mulc(X2, z1z1, K);
	const auto s2 = mulc(mulc(Y2, P.Z, K), z1z1, K);
	const auto h = carry(sub_auto<1>(u2, P.X, K));
	const auto r = carry(sub_auto<1>(s2, P.Y, K));
	const auto hh = sqrc(h, K);
	const auto hhh = mulc(h, hh, K);
	const auto v = mulc(P.X, hh, K);
	const auto r2 = sqrc(r, K);
	const auto x3 = carry(sub_auto<2>(r2, add(hhh, mul_small<2>(v)), K));
	const auto t5 = carry(sub_auto<1>(v, x3, K));
	const auto y3 = carry(sub_auto<1>(mulc(r, t5, K), mulc(P.Y, hhh, K), K));
	const auto z3 = mulc(P.Z, h, K);
	JacT<PB> R;
	R.X = weaken<FT>(x3);
	R.Y = weaken<FT>(y3);
	R.Z = weaken<FT>(z3);
	return R;
}

// -Y (+ a multiple of p), carried: the negated table entry of a negative window digit.  Table
// entries keep Y as a multiplication result (value < 2p) so that the negation stays inside FA.
template <int PB> G29_FN typename Cls<PB>::FA neg(const typename Cls<PB>::FM &y, JG_K)
{
	E<PB, 0, 0, 0> zero;
#pragma unroll
	for (int i = 0; i < Cfg<PB>::NL; i++) {
		zero.l[i] = 0;
	}
	return weaken<typename Cls<PB>::FA>(carry(sub_auto<1>(zero, y, K)));
}

// the same in the tight accumulator class (a digit's first table entry becomes the accumulator)
template <int PB> G29_FN typename ClsT<PB>::FT neg_t(const typename Cls<PB>::FM &y, JG_K)
{
	E<PB, 0, 0, 0> zero;
#pragma unroll
	for (int i = 0; i < Cfg<PB>::NL; i++) {
		zero.l[i] = 0;
	}
	return weaken<typename ClsT<PB>::FT>(carry(sub_auto<1>(zero, y, K)));
}

// table entry: X, Z as they come, Y normalised through one multiplication by 1
template <int PB> struct TabEnt {
	typename Cls<PB>::FA X;
	typename Cls<PB>::FM Y;
	typename Cls<PB>::FA Z;
};
template <int PB> G29_FN TabEnt<PB> to_tab(const Jac<PB> &P, JG_K)
{
	TabEnt<PB> T;
	T.X = P.X;
	T.Y = weaken<typename Cls<PB>::FM>(mul(P.Y, constant<typename Cls<PB>::FC>(K.one), K));
	T.Z = P.Z;
	return T;
}

// x^(p-2) left to right over the bits of p - 2 (wave-uniform exponent) in 2-bit windows with the table x, x^2, x^3:
// |p| squarings and at most |p| / 2 multiplications (three quarters of that on average, against |p| / 2 on average -- and
// |p| for p = 2^521 - 1, whose p - 2 is all ones -- for bit-by-bit square-and-multiply)
template <int PB> G29_FN typename Cls<PB>::FM inv(const typename Cls<PB>::FM &x, JG_K)
{
	typedef typename Cls<PB>::FM FM;
	const FM x2 = weaken<FM>(sqr(x, K));
	const FM x3 = weaken<FM>(mul(x2, x, K));
	FM r = weaken<FM>(constant<typename Cls<PB>::FC>(K.one));
	const int top = ((int)K.pbits - 1) | 1;   // bit index of the high bit of the top pair
	for (int i = top; i >= 1; i -= 2) {
		r = weaken<FM>(sqr(r, K));
		r = weaken<FM>(sqr(r, K));
		const u32 hi = (K.pm2[i / W] >> (i % W)) & 1u, lo = (K.pm2[(i - 1) / W] >> ((i - 1) % W)) & 1u;
		const u32 c = 2u * hi + lo;
		if (c == 1u) {
			r = weaken<FM>(mul(r, x, K));
		} else if (c == 2u) {
			r = weaken<FM>(mul(r, x2, K));
		} else if (c == 3u) {
			r = weaken<FM>(mul(r, x3, K));
		}
	}
	return r;
}

// BIP0340's lift_x on the unit's curve (aff_pt_y_from_x + "the even solution", sig/bip0340.c:947-953; curves/aff_pt.c:102), for fields with
// p = 3 mod 4 (the callers guarantee it): x < p, y = (x^3 + a x + b)^((p + 1) / 4), y^2 checked (not a square: aff_pt_y_from_x fails), and the
// root whose ORIGINAL-curve representative -- through the export factor ey, so also when the unit computes on an isomorphic curve -- is
// even.  xd: the canonical digits of x.  Exponent bits are wave-uniform; 2-bit windows as in inv<PB>.  (xo, yo): as import_point delivers them.
template <int PB, class XD> G29_FN bool lift_x_even(const XD &xd, JG_K, typename Cls<PB>::FM &xo, typename Cls<PB>::FM &yo)
{
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = Cfg<PB>::NL;
	u32 bx = 0;
#pragma unroll
	for (int j = 0; j < NL; j++) {
		bx = (xd.l[j] - K.p[j] - bx) >> 31;
	}
	bool ok = (bx != 0);   // x < p (fp_import_from_buf)
	const FC onec = constant<FC>(K.one);
	const auto xm = mul(xd, constant<FC>(K.ix), K);
	const auto rhs0 = add(mulc(carry(add(sqr(xm, K), constant<FC>(K.a))), xm, K), constant<FC>(K.b));
	const FM rhs = weaken<FM>(mulc(carry(rhs0), onec, K));
	u32 e[NL];   // (p + 1) / 4
	{
		u32 c = 1;
#pragma unroll
		for (int j = 0; j < NL; j++) {
			const u32 t = K.p[j] + c;
			e[j] = t & MASK;
			c = t >> W;
		}
#pragma unroll
		for (int j = 0; j < NL; j++) {
			e[j] = (e[j] >> 2) | ((j + 1 < NL ? (e[j + 1] & 3u) : 0u) << (W - 2));
		}
	}
	const FM x2 = weaken<FM>(sqr(rhs, K));
	const FM x3 = weaken<FM>(mul(x2, rhs, K));
	FM r = weaken<FM>(onec);
	const int top = ((int)K.pbits - 1) | 1;
	for (int i = top; i >= 1; i -= 2) {
		r = weaken<FM>(sqr(r, K));
		r = weaken<FM>(sqr(r, K));
		const u32 hi = (e[i / W] >> (i % W)) & 1u, lo = (e[(i - 1) / W] >> ((i - 1) % W)) & 1u;
		const u32 c = 2u * hi + lo;
		if (c == 1u) {
			r = weaken<FM>(mul(r, rhs, K));
		} else if (c == 2u) {
			r = weaken<FM>(mul(r, x2, K));
		} else if (c == 3u) {
			r = weaken<FM>(mul(r, x3, K));
		}
	}
	{
		const auto dif = carry(sub_auto<1>(rhs, sqr(r, K), K));
		ok = ok & is_zero_mulout(mulc(dif, onec, K), K);
	}
	u32 d[NL];
	canonical_digits(d, mul(r, constant<FC>(K.ey), K), K);
	const bool odd = (d[0] & 1u) != 0u;
	const FM rn = weaken<FM>(mulc(neg<PB>(r, K), onec, K));
	xo = weaken<FM>(xm);
#pragma unroll
	for (int j = 0; j < NL; j++) {
		yo.l[j] = odd ? rn.l[j] : r.l[j];
	}
	return ok;
}

#undef JG_K
}  // namespace jacg
