// libecc_amd/csrc/ecamd_point.h -- short-Weierstrass group law on gfx950, one point per lane.
//
// Replaces (paths relative to /root/reference/src):
//   __prj_pt_add_monty_cf  curves/prj_pt.c:971-1071  RCB Alg. 1 complete addition (generic a)
//   __prj_pt_dbl_monty_cf  curves/prj_pt.c:892-950   RCB Alg. 3 complete doubling (generic a)
//   prj_pt_is_on_curve     curves/prj_pt.c:144-190
// Homogeneous projective (X:Y:Z), infinity = (0:1:0), all coordinates in the Montgomery
// domain of ecamd_field.h.  The formulas are the Renes-Costello-Batina complete ones the
// reference uses, so every input pair -- P+P, P+(-P), P+inf, inf+inf -- is handled by the same
// straight-line code and no lane ever diverges inside the scalar-multiplication loop.
#pragma once
#include "ecamd_field.h"

template <int NW> struct Pt {
	Fe<NW> X, Y, Z;
};

template <int NW> static __device__ __forceinline__ Pt<NW> pt_infinity(int slot)
{
	Pt<NW> r;
	r.X = fe_zero<NW>();
	r.Y = fe_const<NW>(ConstTab<NW>::get(slot).one);
	r.Z = fe_zero<NW>();
	return r;
}

#define MM(x, y) fe_mul<NW>(x, y, slot)
#define AD(x, y) fe_add<NW>(x, y, slot)
#define SB(x, y) fe_sub<NW>(x, y, slot)

// RCB Algorithm 1 (generic a): 12M + 3 m_a + 2 m_3b + 23 add/sub
template <int NW> static __device__ __forceinline__ Pt<NW> pt_add(const Pt<NW> &P, const Pt<NW> &Q, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	const Fe<NW> a = fe_const<NW>(K.a), b3 = fe_const<NW>(K.b3);
	Fe<NW> t0, t1, t2, t3, t4, t5, X3, Y3, Z3;
	t0 = MM(P.X, Q.X);
	t1 = MM(P.Y, Q.Y);
	t2 = MM(P.Z, Q.Z);
	t3 = AD(P.X, P.Y);
	t4 = AD(Q.X, Q.Y);
	t3 = MM(t3, t4);
	t4 = AD(t0, t1);
	t3 = SB(t3, t4);
	t4 = AD(P.X, P.Z);
	t5 = AD(Q.X, Q.Z);
	t4 = MM(t4, t5);
	t5 = AD(t0, t2);
	t4 = SB(t4, t5);
	t5 = AD(P.Y, P.Z);
	X3 = AD(Q.Y, Q.Z);
	t5 = MM(t5, X3);
	X3 = AD(t1, t2);
	t5 = SB(t5, X3);
	Z3 = MM(a, t4);
	X3 = MM(b3, t2);
	Z3 = AD(X3, Z3);
	X3 = SB(t1, Z3);
	Z3 = AD(t1, Z3);
	Y3 = MM(X3, Z3);
	t1 = AD(t0, t0);
	t1 = AD(t1, t0);
	t2 = MM(a, t2);
	t4 = MM(b3, t4);
	t1 = AD(t1, t2);
	t2 = SB(t0, t2);
	t2 = MM(a, t2);
	t4 = AD(t4, t2);
	t0 = MM(t1, t4);
	Y3 = AD(Y3, t0);
	t0 = MM(t5, t4);
	X3 = MM(t3, X3);
	X3 = SB(X3, t0);
	t0 = MM(t3, t1);
	Z3 = MM(t5, Z3);
	Z3 = AD(Z3, t0);
	Pt<NW> R;
	R.X = X3;
	R.Y = Y3;
	R.Z = Z3;
	return R;
}

// RCB Algorithm 3 (generic a): 8M + 3S + 3 m_a + 2 m_3b + 15 add/sub
template <int NW> static __device__ __forceinline__ Pt<NW> pt_dbl(const Pt<NW> &P, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	const Fe<NW> a = fe_const<NW>(K.a), b3 = fe_const<NW>(K.b3);
	Fe<NW> t0, t1, t2, t3, X3, Y3, Z3;
	t0 = MM(P.X, P.X);
	t1 = MM(P.Y, P.Y);
	t2 = MM(P.Z, P.Z);
	t3 = MM(P.X, P.Y);
	t3 = AD(t3, t3);
	Z3 = MM(P.X, P.Z);
	Z3 = AD(Z3, Z3);
	X3 = MM(a, Z3);
	Y3 = MM(b3, t2);
	Y3 = AD(X3, Y3);
	X3 = SB(t1, Y3);
	Y3 = AD(t1, Y3);
	Y3 = MM(X3, Y3);
	X3 = MM(t3, X3);
	Z3 = MM(b3, Z3);
	t2 = MM(a, t2);
	t3 = SB(t0, t2);
	t3 = MM(a, t3);
	t3 = AD(t3, Z3);
	Z3 = AD(t0, t0);
	t0 = AD(Z3, t0);
	t0 = AD(t0, t2);
	t0 = MM(t0, t3);
	Y3 = AD(Y3, t0);
	t2 = MM(P.Y, P.Z);
	t2 = AD(t2, t2);
	t0 = MM(t2, t3);
	X3 = SB(X3, t0);
	Z3 = MM(t2, t1);
	Z3 = AD(Z3, Z3);
	Z3 = AD(Z3, Z3);
	Pt<NW> R;
	R.X = X3;
	R.Y = Y3;
	R.Z = Z3;
	return R;
}

// affine (x, y) in the Montgomery domain on  y^2 = x^3 + a x + b ?
template <int NW> static __device__ __forceinline__ bool aff_on_curve(const Fe<NW> &x, const Fe<NW> &y, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	Fe<NW> t = MM(x, x);
	t = AD(t, fe_const<NW>(K.a));
	t = MM(t, x);
	t = AD(t, fe_const<NW>(K.b));
	Fe<NW> y2 = MM(y, y);
	return fe_eq<NW>(t, y2);
}

#undef MM
#undef AD
#undef SB

// ------------------------------------------------------------------------------------------
// Wire formats (big-endian octet strings, curves/prj_pt.c:462-624, nn/nn.c:479-560)
// ------------------------------------------------------------------------------------------
// len bytes big-endian -> NW little-endian words (len <= 4*NW)
template <int NW> static __device__ __forceinline__ Fe<NW> fe_load_be(const u8 *src, int len)
{
	Fe<NW> r;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		u32 x = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * w + b;  // byte significance
			if (pos < len) {
				x |= (u32)src[len - 1 - pos] << (8 * b);
			}
		}
		r.v[w] = x;
	}
	return r;
}

template <int NW> static __device__ __forceinline__ void fe_store_be(u8 *dst, int len, const Fe<NW> &a)
{
#pragma unroll
	for (int w = 0; w < NW; w++) {
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * w + b;
			if (pos < len) {
				dst[len - 1 - pos] = (u8)(a.v[w] >> (8 * b));
			}
		}
	}
}
