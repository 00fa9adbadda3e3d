// libecc_amd/csrc/ecamd_rcbg.h -- the complete (Renes-Costello-Batina) addition and doubling of ecamd_point.h on the
// radix-2^29 / 2^28 field types of ecamd_u29g.h, for the few complete operations at the END of a protocol computation
// (EdDSA verification: [S]B - R - [h]A, the cofactor doublings, the infinity test), which used to run on the saturated-word
// kernels (k_ed_fin<NW>) although their operands come from this unit.
//
// Replaces (paths relative to /root/reference/src):
//   __prj_pt_add_monty_cf  curves/prj_pt.c:971-1071  RCB Alg. 1 complete addition (generic a)
//   __prj_pt_dbl_monty_cf  curves/prj_pt.c:892-950   RCB Alg. 3 complete doubling (generic a)
// The same polynomials in the same roles as ecamd_point.h:pt_add / pt_dbl, so the results are the same field elements for EVERY
// pair of inputs -- the exceptional pairs of a curve of even order included (the caller tests Y = Z = 0 as the reference does).
// Homogeneous projective (X : Y : Z), infinity = (0 : 1 : 0); coordinates in the unit's domain (Montgomery or plain), carried.
#pragma once
#include "ecamd_jacg.h"

namespace rcbg {
using namespace jacg;

template <int PB> struct PtG {
	typename Cls<PB>::FA X, Y, Z;
};

#define RG_K const CurveG<Cfg<PB>::NL> &K
#define RG_M(a, b) weaken<FM>(mulc(a, b, K))
#define RG_ADD(a, b) carry(add(a, b))
#define RG_SUB(a, b) carry(sub_auto<1>(a, b, K))   /* b: a multiplication result, a constant or a carried value */

template <int PB> G29_FN PtG<PB> infinity(RG_K)
{
	typedef typename Cls<PB>::FA FA;
	PtG<PB> R;
#pragma unroll
	for (int i = 0; i < Cfg<PB>::NL; i++) {
		R.X.l[i] = 0;
		R.Z.l[i] = 0;
	}
	R.Y = weaken<FA>(constant<typename Cls<PB>::FC>(K.one));
	return R;
}

// 3 b (of the curve the unit computes on), carried
template <int PB> G29_FN auto three_b(RG_K)
{
	return carry(mul_small<3>(constant<typename Cls<PB>::FC>(K.b)));
}

// RCB Algorithm 1 (generic a): 12M + 3 m_a + 2 m_3b
template <int PB> G29_FN PtG<PB> add_rcb(const PtG<PB> &P, const PtG<PB> &Q, RG_K)
{
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	const auto a = constant<typename Cls<PB>::FC>(K.a);
	const auto b3 = three_b<PB>(K);
	const FM t0 = RG_M(P.X, Q.X);
	const FM t1 = RG_M(P.Y, Q.Y);
	const FM t2 = RG_M(P.Z, Q.Z);
	const auto t3 = RG_SUB(RG_M(RG_ADD(P.X, P.Y), RG_ADD(Q.X, Q.Y)), RG_ADD(t0, t1));   // X1 Y2 + X2 Y1
	const auto t4 = RG_SUB(RG_M(RG_ADD(P.X, P.Z), RG_ADD(Q.X, Q.Z)), RG_ADD(t0, t2));   // X1 Z2 + X2 Z1
	const auto t5 = RG_SUB(RG_M(RG_ADD(P.Y, P.Z), RG_ADD(Q.Y, Q.Z)), RG_ADD(t1, t2));   // Y1 Z2 + Y2 Z1
	const auto z3a = RG_ADD(RG_M(b3, t2), RG_M(a, t4));                                  // 3b Z1Z2 + a t4
	const auto x3a = RG_SUB(t1, z3a);
	const auto z3b = RG_ADD(t1, z3a);
	const FM y3a = RG_M(x3a, z3b);
	const FM t2a = RG_M(a, t2);
	const auto t1a = RG_ADD(RG_ADD(RG_ADD(t0, t0), t0), t2a);                            // 3 X1X2 + a Z1Z2
	const auto t4a = RG_ADD(RG_M(b3, t4), RG_M(a, RG_SUB(t0, t2a)));                     // 3b t4 + a (X1X2 - a Z1Z2)
	PtG<PB> R;
	R.Y = weaken<FA>(RG_ADD(y3a, RG_M(t1a, t4a)));
	R.X = weaken<FA>(RG_SUB(RG_M(t3, x3a), RG_M(t5, t4a)));
	R.Z = weaken<FA>(RG_ADD(RG_M(t5, z3b), RG_M(t3, t1a)));
	return R;
}

// RCB Algorithm 3 (generic a): 8M + 3S + 3 m_a + 2 m_3b
template <int PB> G29_FN PtG<PB> dbl_rcb(const PtG<PB> &P, RG_K)
{
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	const auto a = constant<typename Cls<PB>::FC>(K.a);
	const auto b3 = three_b<PB>(K);
	const FM t0 = weaken<FM>(sqrc(P.X, K));
	const FM t1 = weaken<FM>(sqrc(P.Y, K));
	const FM t2 = weaken<FM>(sqrc(P.Z, K));
	const FM xy = RG_M(P.X, P.Y);
	const auto t3 = RG_ADD(xy, xy);                                                     // 2 X Y
	const FM xz = RG_M(P.X, P.Z);
	const auto z3 = RG_ADD(xz, xz);                                                     // 2 X Z
	const auto y3a = RG_ADD(RG_M(a, z3), RG_M(b3, t2));                                  // a 2XZ + 3b Z^2
	const auto x3a = RG_SUB(t1, y3a);
	const auto y3b = RG_ADD(t1, y3a);
	const FM y3c = RG_M(x3a, y3b);
	const FM x3b = RG_M(t3, x3a);
	const FM t2a = RG_M(a, t2);
	const auto t3a = RG_ADD(RG_M(a, RG_SUB(t0, t2a)), RG_M(b3, z3));                     // a (X^2 - a Z^2) + 3b 2XZ
	const auto t0a = RG_ADD(RG_ADD(RG_ADD(t0, t0), t0), t2a);                            // 3 X^2 + a Z^2
	const FM yz = RG_M(P.Y, P.Z);
	const auto t2b = RG_ADD(yz, yz);                                                    // 2 Y Z
	PtG<PB> R;
	R.Y = weaken<FA>(RG_ADD(y3c, RG_M(t0a, t3a)));
	R.X = weaken<FA>(RG_SUB(x3b, RG_M(t2b, t3a)));
	const FM z3b = RG_M(t2b, t1);
	const auto z3c = RG_ADD(z3b, z3b);
	R.Z = weaken<FA>(RG_ADD(z3c, z3c));                                                  // 8 Y^3 Z
	return R;
}

// exact test of a coordinate against zero
template <int PB, class A> G29_FN bool coord_is_zero(const A &v, RG_K)
{
	return is_zero_mulout(mulc(v, constant<typename Cls<PB>::FC>(K.one), K), K);
}

#undef RG_K
#undef RG_M
#undef RG_ADD
#undef RG_SUB
}  // namespace rcbg
