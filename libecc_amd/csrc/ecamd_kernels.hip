// libecc_amd/csrc/ecamd_kernels.hip -- gfx950 kernels for the batched scalar-multiplication path.
//
//   k_smul<NW>   one scalar multiplication per lane: replaces prj_pt_import_from_aff_buf ->
//                prj_pt_mul -> prj_pt_unique -> prj_pt_export_to_aff_buf
//                (curves/prj_pt.c:511,1759,241,600 in /root/reference/src)
//   k_fp<NW>     batched field ops (row a6/a7/a13/a14 of SURVEY.md section 8a)
//   k_pt<NW>     batched prj_pt_add / prj_pt_dbl (rows a17/a18)
//
// Algorithm of k_smul (differs from the reference's Montgomery ladder, observable output is
// identical -- SURVEY.md section 8a "what a GPU implementation may change"):
//   fixed 4-bit window, left to right over ALL 8*slen bits of the scalar (no reduction mod the
//   order is needed: the complete formulas have no exceptional cases, so any m is handled the
//   way the reference's m + q / m + q^2 padding handles it -- [m]P is [m mod #E]P either way);
//   per-lane table [0..15]P in a global scratch buffer laid out word-major / lane-minor so
//   that table writes are fully coalesced and a lookup touches at most 16 x 256 B rows;
//   complete RCB addition/doubling; one Fermat inversion to affine.
// Per item: 4 on-curve mults, 14 table additions, 8*slen doublings, 2*slen additions,
// ~1.5*|p| mults for the inversion.
#include "ecamd_point.h"
#include "ecamd_internal.h"

#define ECAMD_DEF_CONST(NW) __constant__ CurveSlots<NW> g_curves_##NW;
ECAMD_DEF_CONST(6)
ECAMD_DEF_CONST(7)
ECAMD_DEF_CONST(8)
ECAMD_DEF_CONST(10)
ECAMD_DEF_CONST(12)
ECAMD_DEF_CONST(14)
ECAMD_DEF_CONST(16)
ECAMD_DEF_CONST(17)

template <int NW> static __device__ __forceinline__ void tbl_store(u32 *tbl, u32 stride, u32 lane, int e, const Pt<NW> &P)
{
	u32 *base = tbl + (size_t)e * 3 * NW * stride + lane;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		base[(size_t)(0 * NW + w) * stride] = P.X.v[w];
		base[(size_t)(1 * NW + w) * stride] = P.Y.v[w];
		base[(size_t)(2 * NW + w) * stride] = P.Z.v[w];
	}
}

template <int NW> static __device__ __forceinline__ Pt<NW> tbl_load(const u32 *tbl, u32 stride, u32 lane, u32 e)
{
	const u32 *base = tbl + (size_t)e * 3 * NW * stride + lane;
	Pt<NW> P;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		P.X.v[w] = base[(size_t)(0 * NW + w) * stride];
		P.Y.v[w] = base[(size_t)(1 * NW + w) * stride];
		P.Z.v[w] = base[(size_t)(2 * NW + w) * stride];
	}
	return P;
}

// Constant-address look-up for SECRET digits: every entry of the lane's table is read, in the same order whatever the digit,
// and the wanted one is kept by masking -- the counterpart of the reference's nn_tabselect (nn/nn.c:564) / masked ladder steps
// (curves/prj_pt.c:1225-1260).  In this table layout (word-major, lane-minor) every one of these loads is coalesced across
// the wave, because all lanes read the same entry at the same time.
template <int NW> static __device__ __forceinline__ Pt<NW> tbl_load_masked(const u32 *tbl, u32 stride, u32 lane, u32 dig)
{
	Pt<NW> R;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		R.X.v[w] = R.Y.v[w] = R.Z.v[w] = 0;
	}
#pragma unroll 1
	for (u32 e = 0; e < ECAMD_TBL_ENTRIES; e++) {
		const Pt<NW> T = tbl_load<NW>(tbl, stride, lane, e);
		const u32 m = 0u - (u32)(e == dig);
#pragma unroll
		for (int w = 0; w < NW; w++) {
			R.X.v[w] |= T.X.v[w] & m;
			R.Y.v[w] |= T.Y.v[w] & m;
			R.Z.v[w] |= T.Z.v[w] & m;
		}
	}
	return R;
}

// MASKED: the scalar is secret -- table look-ups by full scan (above); the rest of the kernel is already uniform (complete
// formulas, fixed window count, no digit-dependent branch)
template <int NW, bool MASKED = false> __global__ __launch_bounds__(64) void k_smul(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	if (A.only_redo && A.status[i] != ECAMD_STATUS_REDO) {
		return;
	}
	const int slot = A.slot;
	const int clen = (int)A.clen;
	u8 *out = A.out + (size_t)i * 2 * clen;

	// ---- import: X||Y big-endian, coordinates < p, on curve (prj_pt.c:511-552) ----
	const u8 *pin = A.points + (size_t)i * A.pstride;
	Fe<NW> x = fe_load_be<NW>(pin, clen);
	Fe<NW> y = fe_load_be<NW>(pin + clen, clen);
	bool ok = fe_lt_p<NW>(x, slot) & fe_lt_p<NW>(y, slot);
	x = fe_to_mont<NW>(x, slot);
	y = fe_to_mont<NW>(y, slot);
	ok = ok & aff_on_curve<NW>(x, y, slot);
	// A point of order exactly 2 (y = 0, only on curves of even order) makes every ladder step of the
	// reference an "exceptional pair" (T1 - T0 = P, curves/prj_pt.c:1058-1060): prj_pt_mul returns -1
	// for ANY scalar.  Mirror that.
	ok = ok & !fe_is_zero<NW>(y);
	if (!ok) {
		A.status[i] = 1;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}

	// ---- table [0..15]P ----
	Pt<NW> P;
	P.X = x;
	P.Y = y;
	P.Z = fe_const<NW>(ConstTab<NW>::get(slot).one);
	Pt<NW> acc = pt_infinity<NW>(slot);
	tbl_store<NW>(A.tbl, A.stride, i, 0, acc);
	acc = P;
	tbl_store<NW>(A.tbl, A.stride, i, 1, acc);
#pragma unroll 1
	for (int e = 2; e < ECAMD_TBL_ENTRIES; e++) {
		acc = pt_add<NW>(acc, P, slot);
		tbl_store<NW>(A.tbl, A.stride, i, e, acc);
	}

	// ---- fixed-window left-to-right ----
	const u8 *sc = A.scalars + (size_t)i * A.sstride;
	const int nwin = 2 * (int)A.slen;
	acc = MASKED ? tbl_load_masked<NW>(A.tbl, A.stride, i, (u32)(sc[0] >> 4)) : tbl_load<NW>(A.tbl, A.stride, i, (u32)(sc[0] >> 4));
#pragma unroll 1
	for (int t = 1; t < nwin; t++) {
#pragma unroll 1
		for (int d = 0; d < ECAMD_WINDOW; d++) {
			acc = pt_dbl<NW>(acc, slot);
		}
		const u32 byte = sc[t >> 1];
		const u32 dig = (t & 1) ? (byte & 15u) : (byte >> 4);
		const Pt<NW> T = MASKED ? tbl_load_masked<NW>(A.tbl, A.stride, i, dig) : tbl_load<NW>(A.tbl, A.stride, i, dig);
		acc = pt_add<NW>(acc, T, slot);
	}

	// ---- to affine (prj_pt_unique, prj_pt.c:241-273) and export (:600-624) ----
	if (fe_is_zero<NW>(acc.Z)) {
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	const Fe<NW> zi = fe_inv<NW>(acc.Z, slot);
	Fe<NW> ax = fe_mul<NW>(acc.X, zi, slot);
	Fe<NW> ay = fe_mul<NW>(acc.Y, zi, slot);
	ax = fe_from_mont<NW>(ax, slot);
	ay = fe_from_mont<NW>(ay, slot);
	fe_store_be<NW>(out, clen, ax);
	fe_store_be<NW>(out + clen, clen, ay);
	A.status[i] = 0;
}

template <int NW> __global__ __launch_bounds__(64) void k_fp(EcamdFpArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	Fe<NW> a, b, r;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		a.v[w] = A.a[(size_t)i * A.wstride + w];
		b.v[w] = A.b[(size_t)i * A.wstride + w];
	}
	switch (A.op) {
	case 0:
		// nn_mul_redc1 semantics: a*b*2^(-64*nlimbs).  Ours is a*b*2^(-32*NW); when NW is odd
		// one more Montgomery multiplication by fix64 moves the result to the reference radix.
		r = fe_mul<NW>(a, b, slot);
		if (!ConstTab<NW>::get(slot).fix_is_id) {
			r = fe_mul<NW>(r, fe_const<NW>(ConstTab<NW>::get(slot).fix64), slot);
		}
		break;
	case 1: r = fe_add<NW>(a, b, slot); break;
	case 2: r = fe_sub<NW>(a, b, slot); break;
	case 3:
		r = fe_mul<NW>(fe_to_mont<NW>(a, slot), b, slot);  // a R * b / R = a b
		break;
	default:
		r = fe_from_mont<NW>(fe_inv<NW>(fe_to_mont<NW>(a, slot), slot), slot);
		break;
	}
#pragma unroll
	for (int w = 0; w < NW; w++) {
		A.out[(size_t)i * A.wstride + w] = r.v[w];
	}
	for (u32 w = NW; w < A.wstride; w++) {
		A.out[(size_t)i * A.wstride + w] = 0;
	}
}

template <int NW> __global__ __launch_bounds__(64) void k_pt(EcamdPtArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int clen = (int)A.clen;
	u8 *out = A.out + (size_t)i * 2 * clen;
	Pt<NW> P, Q;
	bool ok = true;
	{
		const u8 *pin = A.p1 + (size_t)i * 2 * clen;
		Fe<NW> x = fe_load_be<NW>(pin, clen), y = fe_load_be<NW>(pin + clen, clen);
		ok = ok & fe_lt_p<NW>(x, slot) & fe_lt_p<NW>(y, slot);
		P.X = fe_to_mont<NW>(x, slot);
		P.Y = fe_to_mont<NW>(y, slot);
		P.Z = fe_const<NW>(ConstTab<NW>::get(slot).one);
		ok = ok & aff_on_curve<NW>(P.X, P.Y, slot);
	}
	if (!A.dbl) {
		const u8 *pin = A.p2 + (size_t)i * 2 * clen;
		Fe<NW> x = fe_load_be<NW>(pin, clen), y = fe_load_be<NW>(pin + clen, clen);
		ok = ok & fe_lt_p<NW>(x, slot) & fe_lt_p<NW>(y, slot);
		Q.X = fe_to_mont<NW>(x, slot);
		Q.Y = fe_to_mont<NW>(y, slot);
		Q.Z = P.Z;
		ok = ok & aff_on_curve<NW>(Q.X, Q.Y, slot);
	}
	if (!ok) {
		A.status[i] = 1;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	Pt<NW> R = A.dbl ? pt_dbl<NW>(P, slot) : pt_add<NW>(P, Q, slot);
	if (!A.dbl && fe_is_zero<NW>(R.Z) && fe_is_zero<NW>(R.Y)) {
		// "exceptional pair" (P - Q of order exactly 2): prj_pt_add returns -1 (curves/prj_pt.c:1058-1060)
		A.status[i] = 1;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	if (fe_is_zero<NW>(R.Z)) {
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	const Fe<NW> zi = fe_inv<NW>(R.Z, slot);
	Fe<NW> ax = fe_from_mont<NW>(fe_mul<NW>(R.X, zi, slot), slot);
	Fe<NW> ay = fe_from_mont<NW>(fe_mul<NW>(R.Y, zi, slot), slot);
	fe_store_be<NW>(out, clen, ax);
	fe_store_be<NW>(out + clen, clen, ay);
	A.status[i] = 0;
}

// ------------------------------------------------------------------------------------------
// ECDSA verification around the scalar multiplications
//   __ecdsa_verify_init     sig/ecdsa_common.c:619-675  r, s in [1, q-1]
//   __ecdsa_verify_finalize sig/ecdsa_common.c:702-840  e = OS2I(h) >> max(0, 8|h| - |q|) mod q,
//                           u = e/s, v = r/s (mod q), W' = uG + vY, accept iff W'x mod q == r
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ __forceinline__ bool fe_lt(const Fe<NW> &a, const u32 *m)
{
	u32 borrow = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {
		u64 x = (u64)a.v[j] - m[j] - borrow;
		borrow = (u32)(x >> 63);
	}
	return borrow != 0;
}

// digest -> e = (OS2I(h) >> max(0, 8|h| - |q|)) mod q   (sig/ecdsa_common.c:398-413, 760-778)
template <int NW> static __device__ __forceinline__ Fe<NW> digest_to_e(const u8 *dg, int hlen, int qlen, int qbits, int qs)
{
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const int elen = hlen < qlen ? hlen : qlen;
	Fe<NW> e = fe_load_be<NW>(dg, elen);
	const int rshift = (8 * hlen > qbits) ? (8 * elen - qbits) : 0;  // 0..7
	if (rshift > 0) {
#pragma unroll
		for (int j = 0; j < NW; j++) {
			const u32 hi = (j + 1 < NW) ? e.v[j + 1] : 0u;
			e.v[j] = (e.v[j] >> rshift) | (hi << (32 - rshift));
		}
	}
	u32 qw[NW];
#pragma unroll
	for (int j = 0; j < NW; j++) {
		qw[j] = Q.p[j];
	}
	return fe_cond_sub<NW>(e.v, 0u, qw);  // e < 2^|q| < 2q: one conditional subtraction is nn_mod
}

// One lane prepares ECDSA_PREP_K consecutive items and shares one Fermat inversion among them
// (Montgomery's trick: prefix products, one x^(q-2), back-substitution): 3 multiplications + 1/K of an
// inversion per item instead of a whole one.  Items whose s is out of range take part with 1.
#ifndef ECDSA_PREP_K
#define ECDSA_PREP_K 8
#endif
// k_ecdsa_prep shares its inversion among sixteen items from twelve words on: measured against 8 and 4 at 2^20 signatures
// (profiles/r3d_variants.md), ECDSA verification secp384r1 14.37 -> 14.55 M/s, secp521r1 10.03 -> 10.33, secp256r1 65.4 -> 65.0
// (kept at eight there: one lane per sixteen items leaves too few waves at 256 bits)
#ifdef ECDSA_PREP_K_ALL
constexpr int ecdsa_prep_items(int) { return ECDSA_PREP_K_ALL; }
#else
constexpr int ecdsa_prep_items(int nw) { return nw >= 12 ? 16 : ECDSA_PREP_K; }
#endif
template <int NW> __global__ __launch_bounds__(64) void k_ecdsa_prep(EcamdEcdsaPrepArgs A)
{
	constexpr int KP = ecdsa_prep_items(NW);
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	const u32 first = t * KP;
	if (first >= A.n) {
		return;
	}
	if (A.only != nullptr) {
		// redo pass: nothing to do unless one of the lane's items is marked
		bool any = false;
		for (int k = 0; k < KP; k++) {
			any = any | (first + k < A.n && A.only[first + k] == ECAMD_STATUS_REDO);
		}
		if (!any) {
			return;
		}
	}
	const int qs = A.qslot;  // modulus of this slot is q
	const int qlen = (int)A.qlen, hlen = (int)A.hlen;
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(qs).one);
	Fe<NW> pre[KP];  // pre[k] = s_0 ... s_k (Montgomery form)
	u32 okmask = 0;
	Fe<NW> acc = one;
#pragma unroll
	for (int k = 0; k < KP; k++) {
		const u32 i = first + k;
		if (i < A.n) {
			const u8 *sig = A.sigs + (size_t)i * 2 * qlen;
			const Fe<NW> r = fe_load_be<NW>(sig, qlen), sv = fe_load_be<NW>(sig + qlen, qlen);
			const bool ok = !fe_is_zero<NW>(r) & !fe_is_zero<NW>(sv) & fe_lt_p<NW>(r, qs) & fe_lt_p<NW>(sv, qs);
			okmask |= ok ? (1u << k) : 0u;
			acc = fe_mul<NW>(acc, ok ? fe_to_mont<NW>(sv, qs) : one, qs);
		}
		pre[k] = acc;
	}
	// (s_0 ... s_last)^-1 = x^(q-2) (q prime): the unique inverse, equal to nn_modinv's (nn/nn_modinv.c:220)
	Fe<NW> inv = fe_inv<NW>(acc, qs);
#pragma unroll
	for (int k = KP - 1; k >= 0; k--) {
		const u32 i = first + k;
		if (i >= A.n) {
			continue;
		}
		const bool ok = (okmask >> k) & 1u;
		const u8 *sig = A.sigs + (size_t)i * 2 * qlen;
		const Fe<NW> r = fe_load_be<NW>(sig, qlen), sv = fe_load_be<NW>(sig + qlen, qlen);
		const Fe<NW> sm = ok ? fe_to_mont<NW>(sv, qs) : one;
		const Fe<NW> sinv = (k > 0) ? fe_mul<NW>(inv, pre[k - 1], qs) : inv;   // Montgomery form of 1/s_k
		inv = fe_mul<NW>(inv, sm, qs);
		const Fe<NW> e = digest_to_e<NW>(A.digests + (size_t)i * hlen, hlen, qlen, (int)A.qbits, qs);
		const Fe<NW> u1 = fe_mul<NW>(e, sinv, qs);                               // plain * Montgomery = plain e/s
		const Fe<NW> u2 = fe_mul<NW>(r, sinv, qs);
		fe_store_be<NW>(A.u1 + (size_t)i * qlen, qlen, ok ? u1 : fe_zero<NW>());
		fe_store_be<NW>(A.u2 + (size_t)i * qlen, qlen, ok ? u2 : fe_zero<NW>());
		A.flags[i] = ok ? 0 : 1;
	}
}

// __ecdsa_sign_finalize (sig/ecdsa_common.c:318-586) after kG: r = kG.x mod q, s = k^-1 (x r + e) mod q.
// The nonce k is an input (the reference takes it from ctx->rand; its KAT harness injects it the same
// way).  status 1: k not in [1, q-1], or one of the reference's restart conditions, which a fixed
// nonce cannot get past: r = 0 (:492), e == x r (:516), s = 0 (:548).
template <int NW> __global__ __launch_bounds__(64) void k_ecdsa_sign(EcamdEcdsaSignArgs A)
{
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	const u32 first = t * ECDSA_PREP_K;
	if (first >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const int qlen = (int)A.qlen, clen = (int)A.clen;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> one = fe_const<NW>(Q.one);
	const Fe<NW> r2q = fe_const<NW>(Q.r2);
	u32 qw[NW];
#pragma unroll
	for (int j = 0; j < NW; j++) {
		qw[j] = Q.p[j];
	}
	// k^-1 for ECDSA_PREP_K consecutive items from one inversion (nonces out of range take part with 1)
	Fe<NW> pre[ECDSA_PREP_K];
	u32 okmask = 0;
	Fe<NW> acc = one;
#pragma unroll
	for (int k = 0; k < ECDSA_PREP_K; k++) {
		const u32 i = first + k;
		if (i < A.n) {
			const Fe<NW> kk = fe_load_be<NW>(A.nonces + (size_t)i * qlen, qlen);
			const bool ok = !fe_is_zero<NW>(kk) & fe_lt_p<NW>(kk, qs);
			okmask |= ok ? (1u << k) : 0u;
			acc = fe_mul<NW>(acc, ok ? fe_mul<NW>(kk, r2q, qs) : one, qs);
		}
		pre[k] = acc;
	}
	Fe<NW> inv = fe_inv<NW>(acc, qs);
#pragma unroll
	for (int k = ECDSA_PREP_K - 1; k >= 0; k--) {
		const u32 i = first + k;
		if (i >= A.n) {
			continue;
		}
		u8 *sig = A.sigs + (size_t)i * 2 * qlen;
		const Fe<NW> kk = fe_load_be<NW>(A.nonces + (size_t)i * qlen, qlen);
		bool ok = ((okmask >> k) & 1u) & (A.stkG[i] == 0);
		const Fe<NW> km = ((okmask >> k) & 1u) ? fe_mul<NW>(kk, r2q, qs) : one;
		const Fe<NW> kinv = (k > 0) ? fe_mul<NW>(inv, pre[k - 1], qs) : inv;     // Montgomery form of 1/k
		inv = fe_mul<NW>(inv, km, qs);
		const Fe<NW> x = fe_load_be<NW>(A.privs + (size_t)i * qlen, qlen);
		// r = kG.x mod q: x < p <= (jmax + 1) q, so jmax conditional subtractions
		Fe<NW> r = fe_load_be<NW>(A.kG + (size_t)i * 2 * clen, clen);
		for (u32 j = 0; j < A.jmax; j++) {
			r = fe_cond_sub<NW>(r.v, 0u, qw);
		}
		ok = ok & !fe_is_zero<NW>(r);
		ok = ok & fe_lt_p<NW>(x, qs);      // __ecdsa_sign_init fails on a private key >= q (sig/ecdsa_common.c:367-371)
		const Fe<NW> e = digest_to_e<NW>(A.digests + (size_t)i * A.hlen, (int)A.hlen, qlen, (int)A.qbits, qs);
		const Fe<NW> xr = fe_mul<NW>(fe_mul<NW>(x, r2q, qs), r, qs);         // x r mod q (plain)
		ok = ok & !fe_eq<NW>(e, xr);                                            // :516 restart condition
		const Fe<NW> tt = fe_add<NW>(xr, e, qs);
		const Fe<NW> sv = fe_mul<NW>(tt, kinv, qs);
		ok = ok & !fe_is_zero<NW>(sv);
		fe_store_be<NW>(sig, qlen, ok ? r : fe_zero<NW>());
		fe_store_be<NW>(sig + qlen, qlen, ok ? sv : fe_zero<NW>());
		A.status[i] = ok ? 0 : 1;
	}
}

template <int NW> __global__ __launch_bounds__(64) void k_ecdsa_fin(EcamdEcdsaFinArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	if (A.only != nullptr && A.only[i] != ECAMD_STATUS_REDO) {
		return;   // redo pass: the item already has its result
	}
	const int slot = A.slot;
	const int clen = (int)A.clen, qlen = (int)A.qlen;
	const u32 sa = A.stA[i], sb = A.stB[i];
	// invalid signature range, invalid public key (import failed), or W' = infinity
	if (A.flags[i] || sa == 1 || sb == 1 || (sa == 2 && sb == 2)) {
		A.result[i] = 1;
		return;
	}
	if (sa == ECAMD_STATUS_REDO || sb == ECAMD_STATUS_REDO) {
		// the fused double-scalar loop met an exceptional pair: the host's redo pass verifies the item the reference's way
		A.result[i] = ECAMD_STATUS_REDO;
		return;
	}
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	Pt<NW> P, Q, W;
	{
		const u8 *pa = A.A + (size_t)i * 2 * clen, *pb = A.B + (size_t)i * 2 * clen;
		P.X = fe_to_mont<NW>(fe_load_be<NW>(pa, clen), slot);
		P.Y = fe_to_mont<NW>(fe_load_be<NW>(pa + clen, clen), slot);
		P.Z = one;
		Q.X = fe_to_mont<NW>(fe_load_be<NW>(pb, clen), slot);
		Q.Y = fe_to_mont<NW>(fe_load_be<NW>(pb + clen, clen), slot);
		Q.Z = one;
	}
	if (sa == 2) {
		W = Q;
	} else if (sb == 2) {
		W = P;
	} else {
		W = pt_add<NW>(P, Q, slot);  // complete: also P == Q and P == -Q
	}
	if (fe_is_zero<NW>(W.Z)) {
		A.result[i] = 1;
		return;
	}
	// W'x mod q == r  <=>  x == r + j q for some j with r + j q < p  <=>  (r + j q) Z == X
	const u8 *rp = A.sigs + (size_t)i * 2 * qlen;
	int rlen = qlen;
	while (rlen > 4 * NW) {
		// q longer than the field words (secp224k1): r beyond them cannot equal any x < p
		if (*rp != 0) {
			A.result[i] = 1;
			return;
		}
		rp++;
		rlen--;
	}
	Fe<NW> t = fe_load_be<NW>(rp, rlen);
	bool acc = false;
	for (u32 j = 0; j <= A.jmax; j++) {
		if (fe_lt_p<NW>(t, slot)) {
			const Fe<NW> lhs = fe_mul<NW>(fe_to_mont<NW>(t, slot), W.Z, slot);
			acc = acc | fe_eq<NW>(lhs, W.X);
		}
		u32 carry = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			const u64 x = (u64)t.v[w] + A.q[w] + carry;
			t.v[w] = (u32)x;
			carry = (u32)(x >> 32);
		}
		if (carry) {
			break;  // r + j q no longer fits NW words, hence >= p
		}
	}
	A.result[i] = acc ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// X25519 / X448 through the Weierstrass model, as the reference does (ecdh/x25519_448.c:146-302):
//   decode_scalar (:40-70) clamp + byte reversal; u little-endian, u >= p rejected (:219-220);
//   v from u: v^2 = u^3 + A u^2 + u (B = 1), non-residue ("u on the twist") rejected (:226,
//   aff_pt_montgomery_v_from_u curves/aff_pt_montgomery.c:547); (x, y) = (u + A/3, v) (:438-486);
//   [h]Q must not be infinity (:259-260), [k]Q (:268), u' = x' - A/3, u' = 0 rejected (:275-276).
// The square root is a fixed exponentiation (p = 5 mod 8: Atkin-style candidate w^((p+3)/8), fixed
// up with sqrt(-1); p = 3 mod 4: w^((p+1)/4)); the reference's Tonelli-Shanks (fp/fp_sqrt.c) returns
// the same pair {v, -v}, and only u' of the result is observable.
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ __forceinline__ Fe<NW> fe_load_le(const u8 *src, int len)
{
	Fe<NW> r;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		u32 x = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * w + b;
			if (pos < len) {
				x |= (u32)src[pos] << (8 * b);
			}
		}
		r.v[w] = x;
	}
	return r;
}

template <int NW> __global__ __launch_bounds__(64) void k_xdh_prep(EcamdXdhPrepArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int len = (int)A.len;
	// scalar: reversed to big-endian and clamped (decode_scalar)
	{
		const u8 *ks = A.k + (size_t)i * len;
		u8 *kd = A.scalars + (size_t)i * len;
		for (int b = 0; b < len; b++) {
			u8 v = ks[b];
			if (len == 32) {
				if (b == 0) v &= 248;
				if (b == 31) v = (u8)((v & 127) | 64);
			} else {
				if (b == 0) v &= 252;
				if (b == len - 1) v |= 128;
			}
			kd[len - 1 - b] = v;
		}
	}
	const Fe<NW> u = fe_load_le<NW>(A.u + (size_t)i * len, len);
	bool ok = fe_lt_p<NW>(u, slot);
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const Fe<NW> um = fe_to_mont<NW>(u, slot);
	// w = u (u (u + A) + 1)
	Fe<NW> t = fe_add<NW>(um, fe_const<NW>(A.A), slot);
	t = fe_mul<NW>(um, t, slot);
	t = fe_add<NW>(t, one, slot);
	const Fe<NW> w = fe_mul<NW>(um, t, slot);
	// candidate root c = w^e (left-to-right binary, wave-uniform exponent)
	Fe<NW> c = one;
	for (int b = (int)A.ebits - 1; b >= 0; b--) {
		c = fe_mul<NW>(c, c, slot);
		if ((A.e[b >> 5] >> (b & 31)) & 1u) {
			c = fe_mul<NW>(c, w, slot);
		}
	}
	const Fe<NW> c2 = fe_mul<NW>(c, c, slot);
	Fe<NW> v = c;
	bool root = fe_eq<NW>(c2, w);
	if (A.mode == 0) {
		const Fe<NW> negw = fe_sub<NW>(fe_zero<NW>(), w, slot);
		const bool alt = fe_eq<NW>(c2, negw);
		v = fe_select<NW>(alt & !root, fe_mul<NW>(c, fe_const<NW>(A.sm1), slot), c);
		root = root | alt;
	}
	ok = ok & root;
	const Fe<NW> xm = fe_add<NW>(um, fe_const<NW>(A.A3), slot);
	{
		// [h]Q must not be infinity (x25519_448.c:259-260): h = 2^cof_dbl, so _prj_pt_unprotected_mult is
		// cof_dbl doublings of Q
		Pt<NW> P;
		P.X = xm;
		P.Y = v;
		P.Z = one;
		for (u32 r = 0; r < A.cof_dbl; r++) {
			P = pt_dbl<NW>(P, slot);
		}
		ok = ok & !fe_is_zero<NW>(P.Z);
	}
	const Fe<NW> x = fe_from_mont<NW>(xm, slot);
	const Fe<NW> y = fe_from_mont<NW>(v, slot);
	u8 *pd = A.points + (size_t)i * 2 * len;
	fe_store_be<NW>(pd, len, ok ? x : fe_zero<NW>());
	fe_store_be<NW>(pd + len, len, ok ? y : fe_zero<NW>());
	A.flags[i] = ok ? 0 : 1;
}

template <int NW> __global__ __launch_bounds__(64) void k_xdh_fin(EcamdXdhFinArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int len = (int)A.len;
	u8 *out = A.out + (size_t)i * len;
	bool ok = (A.flags[i] == 0) & (A.stk[i] == 0);
	const Fe<NW> x = fe_load_be<NW>(A.pts + (size_t)i * 2 * len, len);
	const Fe<NW> u = fe_sub<NW>(x, fe_const<NW>(A.A3), slot);  // plain residues: no Montgomery form needed
	ok = ok & !fe_is_zero<NW>(u);
	for (int b = 0; b < len; b++) {
		const u32 wv = ok ? u.v[b >> 2] : 0u;
		out[b] = (u8)(wv >> (8 * (b & 3)));
	}
	A.status[i] = ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Ed25519 verification (sig/eddsa.c of the reference) through the Weierstrass model WEI25519:
//   eddsa_decode_point (:424-556): y little-endian with the sign of x in the top bit, y >= p rejected,
//     x^2 = (1 - y^2) / (a - d y^2) (aff_pt_edwards_x_from_y, curves/aff_pt_edwards.c:816), no root ->
//     error, the root whose parity is the sign bit, x = 0 with sign 1 rejected;
//   aff_pt_edwards_to_montgomery (curves/aff_pt_edwards.c:520-614): the neutral (0, 1) is rejected,
//     (0, -1) dies in fp_inv(0); (u, v) = ((1 + y) / (1 - y), alpha u / x);
//   aff_pt_montgomery_to_shortw (curves/aff_pt_montgomery.c:445): (X, Y) = (u + A/3, v) for B = 1;
//   _eddsa_verify_init (:1846-1990): S < q, [8]A != infinity;
//   _eddsa_verify_finalize (:2130-2290): h = H(R || A || M) little-endian mod q (the caller hashes),
//     W = [S]G - R - [h]A, accept iff [8]W is the point at infinity.
// Field exponentiations use the fixed addition chain for p = 2^255 - 19 (z^(2^250 - 1) shared by the
// inversion exponent p - 2 and the square-root exponent (p - 5) / 8).
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ __forceinline__ Fe<NW> fe_sqr_n(Fe<NW> a, int n, int slot)
{
	for (int k = 0; k < n; k++) {
		a = fe_mul<NW>(a, a, slot);
	}
	return a;
}

// z^(2^250 - 1); *z11 = z^11
template <int NW> static __device__ Fe<NW> fe_pow_2_250m1(const Fe<NW> &z, Fe<NW> *z11, int slot)
{
	const Fe<NW> z2 = fe_mul<NW>(z, z, slot);
	const Fe<NW> z9 = fe_mul<NW>(fe_sqr_n<NW>(z2, 2, slot), z, slot);
	*z11 = fe_mul<NW>(z9, z2, slot);
	const Fe<NW> a5 = fe_mul<NW>(fe_mul<NW>(*z11, *z11, slot), z9, slot);             // 2^5 - 1
	const Fe<NW> a10 = fe_mul<NW>(fe_sqr_n<NW>(a5, 5, slot), a5, slot);               // 2^10 - 1
	const Fe<NW> a20 = fe_mul<NW>(fe_sqr_n<NW>(a10, 10, slot), a10, slot);            // 2^20 - 1
	const Fe<NW> a40 = fe_mul<NW>(fe_sqr_n<NW>(a20, 20, slot), a20, slot);            // 2^40 - 1
	const Fe<NW> a50 = fe_mul<NW>(fe_sqr_n<NW>(a40, 10, slot), a10, slot);            // 2^50 - 1
	const Fe<NW> a100 = fe_mul<NW>(fe_sqr_n<NW>(a50, 50, slot), a50, slot);           // 2^100 - 1
	const Fe<NW> a200 = fe_mul<NW>(fe_sqr_n<NW>(a100, 100, slot), a100, slot);        // 2^200 - 1
	return fe_mul<NW>(fe_sqr_n<NW>(a200, 50, slot), a50, slot);                        // 2^250 - 1
}

// x of a compressed point: returns false where the reference's decode / map fails
// *neutral: the encoding is that of the neutral element (0, 1) (x = 0 with the sign bit clear, y = 1), which the reference
// decodes and maps to the point at infinity (aff_pt_edwards_to_prj_pt_shortw, curves/prj_pt.c:1976-1982): fine for R
// (the verification goes on with R = infinity), while a key is then rejected by the small-order test.
template <int NW> static __device__ bool ed_decode_xy(const EcamdEdDecodeArgs &A, const u8 *src, Fe<NW> *xo, Fe<NW> *ymo, bool *neutral)
{
	const int slot = A.slot;
	const int len = (int)A.len;
	Fe<NW> y = fe_load_le<NW>(src, len);
	const int sbit = 8 * len - 1;
	const u32 x0 = (y.v[sbit >> 5] >> (sbit & 31)) & 1u;
	y.v[sbit >> 5] &= ~(1u << (sbit & 31));
	bool ok = fe_lt_p<NW>(y, slot);
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const Fe<NW> zero = fe_zero<NW>();
	const Fe<NW> ym = fe_to_mont<NW>(y, slot);
	const Fe<NW> y2 = fe_mul<NW>(ym, ym, slot);
	const Fe<NW> u = fe_sub<NW>(one, y2, slot);                                         // 1 - y^2
	const Fe<NW> v = fe_sub<NW>(fe_const<NW>(A.a), fe_mul<NW>(fe_const<NW>(A.d), y2, slot), slot);  // a - d y^2
	// beta = u v^3 (u v^7)^((p - 5) / 8), (p - 5) / 8 = 2^252 - 3
	const Fe<NW> v3 = fe_mul<NW>(fe_mul<NW>(v, v, slot), v, slot);
	const Fe<NW> v7 = fe_mul<NW>(fe_mul<NW>(v3, v3, slot), v, slot);
	const Fe<NW> t = fe_mul<NW>(u, v7, slot);
	Fe<NW> t11;
	const Fe<NW> pw = fe_mul<NW>(fe_sqr_n<NW>(fe_pow_2_250m1<NW>(t, &t11, slot), 2, slot), t, slot);
	const Fe<NW> beta = fe_mul<NW>(fe_mul<NW>(u, v3, slot), pw, slot);
	const Fe<NW> chk = fe_mul<NW>(v, fe_mul<NW>(beta, beta, slot), slot);
	const bool root = fe_eq<NW>(chk, u);
	const bool alt = fe_eq<NW>(chk, fe_sub<NW>(zero, u, slot));
	ok = ok & (root | alt);
	Fe<NW> x = fe_select<NW>(alt & !root, fe_mul<NW>(beta, fe_const<NW>(A.sm1), slot), beta);
	const Fe<NW> xp = fe_from_mont<NW>(x, slot);
	x = fe_select<NW>((xp.v[0] & 1u) != x0, fe_sub<NW>(zero, x, slot), x);
	// x = 0: (0, 1) is the neutral element (see above; x_0 = 1 with x = 0 is a decoding error, sig/eddsa.c:511-513),
	// (0, -1) dies in fp_inv(0) of the map to the Montgomery model
	*neutral = ok & fe_is_zero<NW>(x) & (x0 == 0) & fe_eq<NW>(ym, one);
	ok = ok & !fe_is_zero<NW>(x);
	*xo = x;
	*ymo = ym;
	return ok;
}

// One lane decodes the public key A and the signature's R of one item (one shared inversion), maps
// both to the Weierstrass model and checks [cofactor]A != infinity by doublings, as
// _prj_pt_unprotected_mult does for the cofactor 8 = 1000b (curves/prj_pt.c:1862-1905).
template <int NW> __global__ __launch_bounds__(64) void k_ed_decode(EcamdEdDecodeArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int len = (int)A.len;
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const Fe<NW> zero = fe_zero<NW>();
	Fe<NW> x[2], ym[2], omy[2], den[2];
	bool ok[2], neutral[2];
	ok[0] = ed_decode_xy<NW>(A, A.encA + (size_t)i * A.strideA, &x[0], &ym[0], &neutral[0]);
	ok[1] = ed_decode_xy<NW>(A, A.encR + (size_t)i * A.strideR, &x[1], &ym[1], &neutral[1]);
#pragma unroll
	for (int k = 0; k < 2; k++) {
		omy[k] = fe_sub<NW>(one, ym[k], slot);
		den[k] = fe_select<NW>(ok[k], fe_mul<NW>(omy[k], x[k], slot), one);  // (1 - y) x, non-zero when ok
	}
	Fe<NW> d11;
	const Fe<NW> dd = fe_mul<NW>(den[0], den[1], slot);
	const Fe<NW> dinv = fe_mul<NW>(fe_sqr_n<NW>(fe_pow_2_250m1<NW>(dd, &d11, slot), 5, slot), d11, slot);  // dd^(p-2)
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const Fe<NW> inv = fe_mul<NW>(dinv, den[1 - k], slot);                        // 1 / ((1 - y) x)
		const Fe<NW> um = fe_mul<NW>(fe_add<NW>(one, ym[k], slot), fe_mul<NW>(inv, x[k], slot), slot);
		const Fe<NW> vm = fe_mul<NW>(fe_mul<NW>(fe_const<NW>(A.alpha), um, slot), fe_mul<NW>(inv, omy[k], slot), slot);
		const Fe<NW> Xm = fe_add<NW>(um, fe_const<NW>(A.A3), slot);
		bool good = ok[k];
		if (k == 0) {
			Pt<NW> P;
			P.X = Xm;
			P.Y = vm;
			P.Z = one;
			for (u32 r = 0; r < A.cof_dbl; r++) {
				P = pt_dbl<NW>(P, slot);
			}
			good = good & !fe_is_zero<NW>(P.Z);                                        // small-order key
		}
		u8 *pd = (k == 0 ? A.pointsA : A.pointsR) + (size_t)i * 2 * len;
		fe_store_be<NW>(pd, len, good ? fe_from_mont<NW>(Xm, slot) : zero);
		fe_store_be<NW>(pd + len, len, good ? fe_from_mont<NW>(vm, slot) : zero);
		(k == 0 ? A.flagsA : A.flagsR)[i] = good ? 0 : ((k == 1 && neutral[1]) ? 2 : 1);   // 2: R is the point at infinity
	}
}

// S (second half of the signature, little-endian) must be < q; h = hram (little-endian, up to 2 NW words) mod q
template <int NW> __global__ __launch_bounds__(64) void k_ed_scal(EcamdEdScalArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const int len = (int)A.len, hlen = (int)A.hlen;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> S = fe_load_le<NW>(A.sigs + (size_t)i * 2 * len + len, len);
	const bool ok = fe_lt_p<NW>(S, qs);
	const u8 *hp = A.hram + (size_t)i * hlen;
	const int lo_len = hlen < 4 * NW ? hlen : 4 * NW;
	const Fe<NW> lo = fe_load_le<NW>(hp, lo_len);
	const Fe<NW> hi = fe_load_le<NW>(hp + lo_len, hlen - lo_len);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	// Montgomery products with R = 2^(32 NW): hi R2 / R = hi 2^(32 NW), (lo R2 / R) * 1 / R = lo  (mod q)
	const Fe<NW> hiR = fe_mul<NW>(hi, r2, qs);
	const Fe<NW> lor = fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs);
	const Fe<NW> h = fe_add<NW>(hiR, lor, qs);
	fe_store_be<NW>(A.S_be + (size_t)i * len, len, ok ? S : fe_zero<NW>());
	fe_store_be<NW>(A.h_be + (size_t)i * len, len, h);
	A.flags[i] = ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Round 4: half-length scalars for the Ed25519 verification equation (k_ed_lat<8>).
//
// The equation 8 ([S]B - R - [h]A) = infinity holds exactly when 8 ([u S]B - [u]R - [u h]A) = infinity for any u that is not a
// multiple of q (8 (...) lies in the subgroup of prime order q, where u is invertible), and [u h]A may be taken with u h reduced
// mod q (8 A has order dividing q).  With (u, v) a short vector of the lattice {(x, y) : y = x h mod q} -- |u| < 2^126,
// 0 <= v < 2^127, found by the Euclidean algorithm on (q, h) stopped at the first remainder below 2^127 (the extended-gcd
// cofactor of that remainder is bounded by q over the previous one) -- the two variable-base multiplications have 128-bit scalars
// and share their doublings: half the doublings of [h]A (T. Pornin, "Optimized lattice basis reduction in dimension 2, and fast
// Schnorr and EdDSA signature verification", 2020; the reduction here is the plain truncated Euclid, per lane).
// Per item the kernel does what k_ed_scal does (S < q, h = hram mod q) and then
//     r0 = q, r1 = h, t0 = 0, t1 = 1;  while r1 >= 2^127: k = a lower bound of r0 / r1 from the leading 63 bits (at least 1, at most
//     2^32 - 1), r0 -= k r1, |t0| += k |t1|, and the pairs swap when r0 < r1 (t0 and t1 always have opposite signs, so only the sign of
//     t1 is kept);  v = r1, u = t1.
// Nothing downstream trusts that loop: u != 0, |u| < 2^128, v < 2^128 and |u| h = +-v (mod q) are CHECKED with the Montgomery
// arithmetic mod q of this unit, and an item that fails (it cannot, short of a bug) keeps u = 1, v = h and takes the full-length
// window loop (meta bit 1).  Outputs: S and s' = u S mod q (big-endian, for the two comb passes of the tail), v and |u| as words,
// meta: bit 0 u < 0, bit 1 long mode, bit 2 v = 0 (h = 0 mod q: [h]A is the neutral element, see k_ed_tail2_c25519).
// ------------------------------------------------------------------------------------------
#include "ecamd_randmod.h"
#include "ecamd_lattice.h"

template <int NW> __global__ __launch_bounds__(64) void k_ed_lat(EcamdEdLatArgs A)
{
	static_assert(NW == 8, "Ed25519 only");
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const int hlen = (int)A.hlen;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> S = fe_load_le<NW>(A.sigs + (size_t)i * 64 + 32, 32);
	const bool ok = fe_lt_p<NW>(S, qs);
	const u8 *hp = A.hram + (size_t)i * hlen;
	const int lo_len = hlen < 4 * NW ? hlen : 4 * NW;
	const Fe<NW> lo = fe_load_le<NW>(hp, lo_len);
	const Fe<NW> hi = fe_load_le<NW>(hp + lo_len, hlen - lo_len);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	const Fe<NW> h = fe_add<NW>(fe_mul<NW>(hi, r2, qs), fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs), qs);   // as k_ed_scal
	const Fe<NW> Sok = ok ? S : fe_zero<NW>();
	u32 qw[8], v[8], u[4];
#pragma unroll
	for (int w = 0; w < 8; w++) {
		qw[w] = Q.p[w];
	}
	bool neg = false;
	bool good = lat_reduce(qw, h.v, v, u, &neg);
	Fe<NW> uf = fe_zero<NW>(), vf;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		uf.v[w] = u[w];
	}
#pragma unroll
	for (int w = 0; w < 8; w++) {
		vf.v[w] = v[w];
	}
	// the relation the tail relies on: |u| h = v (u > 0) or q - v (u < 0), mod q
	const Fe<NW> uR = fe_mul<NW>(uf, r2, qs);
	const Fe<NW> uh = fe_mul<NW>(uR, h, qs);
	const Fe<NW> want = neg ? fe_sub<NW>(fe_zero<NW>(), vf, qs) : vf;
	good = good & fe_eq<NW>(uh, want) & ((v[4] | v[5] | v[6] | v[7]) == 0u);
	Fe<NW> sp = fe_mul<NW>(uR, Sok, qs);            // |u| S mod q
	if (neg) {
		sp = fe_sub<NW>(fe_zero<NW>(), sp, qs);
	}
	if (!good) {
		// full-length scalars: u = 1, v = h
		neg = false;
		sp = Sok;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			v[w] = h.v[w];
		}
		u[0] = 1u;
		u[1] = u[2] = u[3] = 0u;
	}
	fe_store_be<NW>(A.S_be + (size_t)i * 32, 32, Sok);
	fe_store_be<NW>(A.sp_be + (size_t)i * 32, 32, sp);
	u32 *dst = A.uv + (size_t)i * 12;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		dst[w] = v[w];
	}
#pragma unroll
	for (int w = 0; w < 4; w++) {
		dst[8 + w] = u[w];
	}
	const bool vzero = (v[0] | v[1] | v[2] | v[3] | v[4] | v[5] | v[6] | v[7]) == 0u;
	A.meta[i] = (u8)((neg ? 1u : 0u) | (good ? 0u : 2u) | (vzero ? 4u : 0u));
	A.flags[i] = ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Scalars of the Ed25519 batch equation (ecamd_g29_kernel.hip, k_edmsm_*): z_i = 128 bits of ChaCha20(seed; counter = item)
// (the reference draws hsize / 4 = 16 random bytes per signature, sig/eddsa.c:2388), c_i = z_i h_i mod q, z_i S_i mod q.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
#define CHACHA_QR(a, b, c, d)                   \
	a += b; d ^= a; d = rotl32(d, 16);      \
	c += d; b ^= c; b = rotl32(b, 12);      \
	a += b; d ^= a; d = rotl32(d, 8);       \
	c += d; b ^= c; b = rotl32(b, 7);
// first four words of the ChaCha20 block (RFC 8439 section 2.3) for key `key`, block counter `ctr` and nonce `nonce`
static __device__ void chacha20_block4(const u32 *key, u32 ctr, const u32 *nonce, u32 *out4)
{
	const u32 in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
			    key[4], key[5], key[6], key[7], ctr, nonce[0], nonce[1], nonce[2]};
	u32 x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
	u32 x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
#pragma unroll 1
	for (int r = 0; r < 10; r++) {
		CHACHA_QR(x0, x4, x8, x12)
		CHACHA_QR(x1, x5, x9, x13)
		CHACHA_QR(x2, x6, x10, x14)
		CHACHA_QR(x3, x7, x11, x15)
		CHACHA_QR(x0, x5, x10, x15)
		CHACHA_QR(x1, x6, x11, x12)
		CHACHA_QR(x2, x7, x8, x13)
		CHACHA_QR(x3, x4, x9, x14)
	}
	out4[0] = x0 + in[0];
	out4[1] = x1 + in[1];
	out4[2] = x2 + in[2];
	out4[3] = x3 + in[3];
}
#undef CHACHA_QR

template <int NW> __global__ __launch_bounds__(64) void k_edmsm_scal(EcamdEdMsmScalArgs A)
{
	static_assert(NW == 8, "Ed25519 only");
	const u32 rel = blockIdx.x * 64 + threadIdx.x;
	if (rel >= (A.count ? A.count : A.n)) {
		return;
	}
	const u32 i = A.first + rel;
	const int qs = A.qslot;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> S = fe_load_le<NW>(A.sigs + (size_t)i * 64 + 32, 32);
	const bool ok = fe_lt_p<NW>(S, qs);
	const u8 *hp = A.hram + (size_t)i * 64;
	const Fe<NW> lo = fe_load_le<NW>(hp, 32);
	const Fe<NW> hi = fe_load_le<NW>(hp + 32, 32);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	const Fe<NW> h = fe_add<NW>(fe_mul<NW>(hi, r2, qs), fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs), qs);   // as k_ed_scal
	u32 z4[4];
	chacha20_block4(A.seed, i, A.nonce, z4);
	if ((z4[0] | z4[1] | z4[2] | z4[3]) == 0u) {
		z4[0] = 1u;   // the reference draws again on z = 0
	}
	Fe<NW> z = fe_zero<NW>();
#pragma unroll
	for (int w = 0; w < 4; w++) {
		z.v[w] = z4[w];
	}
	const Fe<NW> zR = fe_mul<NW>(z, r2, qs);                          // z in Montgomery form
	const Fe<NW> c = fe_mul<NW>(zR, h, qs);                           // z h mod q
	const Fe<NW> zs = fe_mul<NW>(zR, ok ? S : fe_zero<NW>(), qs);     // z S mod q
	// signed-window recoding: + 0x88..8 (c < 2^253: no carry out; z: 33 nibbles)
	uint64_t cy = 0;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		cy += (uint64_t)c.v[w] + 0x88888888u;
		A.cA[(size_t)w * A.n + i] = (u32)cy;
		cy >>= 32;
	}
	cy = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		cy += (uint64_t)z4[w] + 0x88888888u;
		A.zR[(size_t)w * A.n + i] = (u32)cy;
		cy >>= 32;
	}
	A.zR[(size_t)4 * A.n + i] = (u32)cy + 8u;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		A.zs[(size_t)i * 8 + w] = zs.v[w];
	}
	if (A.rawC != nullptr) {
#pragma unroll
		for (int w = 0; w < 8; w++) {
			A.rawC[(size_t)i * 8 + w] = c.v[w];
		}
#pragma unroll
		for (int w = 0; w < 4; w++) {
			A.rawZ[(size_t)i * 4 + w] = z4[w];
		}
	}
	A.flagsS[i] = ok ? 0 : 1;
	if (A.z_dump != nullptr) {
#pragma unroll
		for (int b = 0; b < 16; b++) {
			A.z_dump[(size_t)i * 16 + b] = (u8)(z4[b >> 2] >> (8 * (b & 3)));
		}
	}
}

// lane l of the Straus evaluation owns the items j * L + l: its share of the base-point scalar, q - sum z_i S_i
template <int NW> __global__ __launch_bounds__(64) void k_edmsm_lane(EcamdEdMsmLaneArgs A)
{
	const u32 lane = blockIdx.x * 64 + threadIdx.x;
	if (lane >= A.L) {
		return;
	}
	const int qs = A.qslot;
	Fe<NW> sum = fe_zero<NW>();
	u32 bad = 0;
	for (u32 j = 0; j < A.K; j++) {
		const u32 item = j * A.L + lane;
		if (item >= A.n) {
			break;
		}
		Fe<NW> t;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			t.v[w] = A.zs[(size_t)item * 8 + w];
		}
		sum = fe_add<NW>(sum, t, qs);
		bad |= (u32)A.flags[item] | (u32)A.flagsS[item];
	}
	const Fe<NW> neg = fe_sub<NW>(fe_zero<NW>(), sum, qs);
	if (A.rawB != nullptr) {
#pragma unroll
		for (int w = 0; w < 8; w++) {
			A.rawB[(size_t)lane * 8 + w] = neg.v[w];
		}
	}
	uint64_t cy = 0;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		cy += (uint64_t)neg.v[w] + 0x88888888u;
		A.sB[(size_t)w * A.L + lane] = (u32)cy;
		cy >>= 32;
	}
	if (bad) {
		atomicOr(A.flagword, 1u);
	}
}

// ------------------------------------------------------------------------------------------
// Scalars of the Schnorr-type batch equation (EcamdMsmScalArgs; the multi-scalar multiplication itself is k_msm_*_g of
// ecamd_g29_kernel.hip): z_i = 128 bits of ChaCha20(seed; counter = item) -- the reference keys ChaCha20 with a hash of the batch
// (sig/bip0340.c:688-760) or draws the z_i with get_random (sig/ecfsdsa.c); any unpredictable z_i decide the same batches --,
// w_i = z_i (q - e_i) mod q and z_i as big-endian strings for the table kernel, v_i = z_i s_i mod q for the generator's term.
// ------------------------------------------------------------------------------------------
template <int NW> __global__ __launch_bounds__(64) void k_msm_scal(EcamdMsmScalArgs A)
{
	const u32 rel = blockIdx.x * 64 + threadIdx.x;
	if (rel >= (A.count ? A.count : A.n)) {
		return;
	}
	const u32 i = A.first + rel;
	const int qs = A.qslot;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const int ql = (int)A.qlen;
	const Fe<NW> sv = fe_load_be<NW>(A.s + (size_t)i * ql, ql);
	const Fe<NW> ne = fe_load_be<NW>(A.ne + (size_t)i * ql, ql);
	if (!(fe_lt_p<NW>(sv, qs) & fe_lt_p<NW>(ne, qs))) {
		atomicOr(A.flagword, 8u);
	}
	u32 z4[4];
	chacha20_block4(A.seed, i, A.nonce, z4);
	if ((z4[0] | z4[1] | z4[2] | z4[3]) == 0u) {
		z4[0] = 1u;
	}
	static_assert(NW >= 5, "the 128-bit z_i must be residues: orders of at least 160 bits");
	Fe<NW> z = fe_zero<NW>();
#pragma unroll
	for (int w = 0; w < 4; w++) {
		z.v[w] = z4[w];
	}
	const Fe<NW> zR = fe_mul<NW>(z, fe_const<NW>(Q.r2), qs);   // z in Montgomery form
	const Fe<NW> w = fe_mul<NW>(zR, ne, qs);                    // z (q - e) mod q, plain
	const Fe<NW> v = fe_mul<NW>(zR, sv, qs);                    // z s mod q, plain
	fe_store_be<NW>(A.scW + (size_t)i * ql, ql, w);
	u8 *zb = A.scZ + (size_t)i * 16;
#pragma unroll
	for (int b = 0; b < 16; b++) {
		zb[15 - b] = (u8)(z4[b >> 2] >> (8 * (b & 3)));
	}
#pragma unroll
	for (int k = 0; k < NW; k++) {
		A.v[(size_t)i * NW + k] = v.v[k];
	}
	if (A.z_dump != nullptr) {
#pragma unroll
		for (int b = 0; b < 16; b++) {
			A.z_dump[(size_t)i * 16 + b] = zb[15 - b];
		}
	}
}

// one level of the sum of the v_i mod q (fan-in 64); the last level also writes the sum as qlen big-endian bytes
template <int NW> __global__ __launch_bounds__(64) void k_msm_vsum(EcamdMsmVsumArgs A)
{
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	const u32 first = t * 64u;
	if (first >= A.count) {
		return;
	}
	const int qs = A.qslot;
	Fe<NW> sum = fe_zero<NW>();
	for (u32 k = first; k < first + 64u && k < A.count; k++) {
		Fe<NW> x;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			x.v[w] = A.in[(size_t)k * NW + w];
		}
		sum = fe_add<NW>(sum, x, qs);
	}
#pragma unroll
	for (int w = 0; w < NW; w++) {
		A.out[(size_t)t * NW + w] = sum.v[w];
	}
	if (A.c_be != nullptr && t == 0) {
		fe_store_be<NW>(A.c_be, (int)A.qlen, sum);
	}
}

template <int NW> static __device__ __forceinline__ Pt<NW> ed_load_neg(const u8 *src, u32 st, int clen, bool neg, int slot)
{
	Pt<NW> P;
	if (st == 2) {
		return pt_infinity<NW>(slot);
	}
	P.X = fe_to_mont<NW>(fe_load_be<NW>(src, clen), slot);
	P.Y = fe_to_mont<NW>(fe_load_be<NW>(src + clen, clen), slot);
	if (neg) {
		P.Y = fe_sub<NW>(fe_zero<NW>(), P.Y, slot);
	}
	P.Z = fe_const<NW>(ConstTab<NW>::get(slot).one);
	return P;
}

template <int NW> __global__ __launch_bounds__(64) void k_ed_fin(EcamdEdFinArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int clen = (int)A.clen;
	const u32 sSG = A.stSG[i], shA = A.sthA[i];
	// bad encodings, small-order public key ([8]A = infinity, folded into flagsA), S >= q, or a failed multiplication
	const u32 fR = A.flagsR[i];   // 2: R decoded to the neutral element, i.e. the point at infinity
	if (A.flagsA[i] || fR == 1 || A.flagsS[i] || sSG == 1 || shA == 1) {
		A.result[i] = 1;
		return;
	}
	if (A.Akey != nullptr) {
		// [cofactor]A != infinity for the key (Ed448: the stored key [4^-1 mod q]A has small order iff A has)
		if (A.stA != nullptr && A.stA[i] != 0) {
			A.result[i] = 1;
			return;
		}
		Pt<NW> K4 = ed_load_neg<NW>(A.Akey + (size_t)i * 2 * clen, 0, clen, false, slot);
		for (u32 k = 0; k < A.cof_dbl; k++) {
			K4 = pt_dbl<NW>(K4, slot);
		}
		if (fe_is_zero<NW>(K4.Z)) {
			A.result[i] = 1;
			return;
		}
	}
	Pt<NW> W = ed_load_neg<NW>(A.SG + (size_t)i * 2 * clen, sSG, clen, false, slot);
	const Pt<NW> Rn = ed_load_neg<NW>(A.R + (size_t)i * 2 * clen, fR, clen, true, slot);
	const Pt<NW> Hn = ed_load_neg<NW>(A.hA + (size_t)i * 2 * clen, shA, clen, true, slot);
	// prj_pt_add returns -1 on an exceptional pair (curves/prj_pt.c:1058-1060): the signature is rejected
	W = pt_add<NW>(W, Rn, slot);
	bool bad = fe_is_zero<NW>(W.Z) & fe_is_zero<NW>(W.Y);
	W = pt_add<NW>(W, Hn, slot);
	bad = bad | (fe_is_zero<NW>(W.Z) & fe_is_zero<NW>(W.Y));
	// _prj_pt_unprotected_mult by the cofactor 8 = 1000b: three doublings (curves/prj_pt.c:1862-1905)
	for (u32 k = 0; k < A.cof_dbl; k++) {
		W = pt_dbl<NW>(W, slot);
	}
	A.result[i] = (!bad && fe_is_zero<NW>(W.Z)) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Ed448 verification (sig/eddsa.c, EDDSA448 branches) through the Weierstrass model WEI448:
//   eddsa_decode_point (:424-556): y little-endian in 57 bytes, bit 455 = sign of x, y >= p rejected; x from y on
//     Ed448 itself (a = 1, d = -39081); the 4-isogeny x' = alpha x y / (2 - x^2 - y^2), y' = (x^2 + y^2) / (y^2 - x^2)
//     (a zero denominator is an fp_inv error); on-curve check on the Edwards model of curve448;
//   aff_pt_edwards_to_montgomery / aff_pt_montgomery_to_shortw with libecc's Montgomery model (A, B) = (-156326, -1):
//     u = (1 + y') / (1 - y'), v = alpha u / x', (X, Y) = (A/3 - u, -v); x' = 0 is rejected as for Ed25519;
//   eddsa_import_pub_key (:925-937): A' <- [4^-1 mod q]A; _eddsa_verify_init: S < q, [4]A' != infinity;
//   _eddsa_verify_finalize: h = hash mod q, then a = 4 h mod q, [S]G - R - [a]A', two cofactor doublings, must be infinity.
//   Here [a]A' is ONE multiplication of the decoded A by a 4^-1 mod 4q (k_ed448_scal), and the small-order test is
//   [4]A = infinity (A' = [c4]A with c4 prime to q has small order exactly when A has).
// Square root: p = 3 mod 4 (wave-uniform exponent bits).
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Ed25519 signing around the caller's hashes (_eddsa_sign, sig/eddsa.c:1554-1870):
//   k_ed_sign_r:   r = H(dom2 || prefix || PH(M)) little-endian mod q (:1731), big-endian for prj_pt_mul(r, G) (:1776)
//   k_ed_sign_enc: prj_pt_shortw_to_aff_pt_edwards (curves/prj_pt.c:2004: infinity -> (0, 1); else (u, v) = (X - A/3, Y),
//                  x = alpha u / v, y = (u - 1) / (u + 1), curves/aff_pt_edwards.c:620) + eddsa_encode_point (:330);
//                  [r]G with 0 < r < q is never 2-torsion, so v and u + 1 are invertible (one shared inversion)
//   k_ed_sign_S:   S = (r + h a) mod q (:1847-1857), a = the clamped secret scalar, little-endian out
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ Fe<NW> ed_le64_mod_q(const u8 *hp, int qs)
{
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> lo = fe_load_le<NW>(hp, 32), hi = fe_load_le<NW>(hp + 32, 32);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	// Montgomery products with R = 2^256: hi R2 / R = hi 2^256, (lo R2 / R) * 1 / R = lo  (mod q), as in k_ed_scal
	return fe_add<NW>(fe_mul<NW>(hi, r2, qs), fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs), qs);
}

template <int NW> static __device__ void ed_store_le32(u8 *out, const Fe<NW> &v)
{
	for (int b = 0; b < 32; b++) {
		out[b] = (u8)(v.v[b >> 2] >> (8 * (b & 3)));
	}
}

template <int NW> __global__ __launch_bounds__(64) void k_ed_sign_r(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	fe_store_be<NW>(A.r_be + (size_t)i * 32, 32, ed_le64_mod_q<NW>(A.r_hash + (size_t)i * 64, A.qslot));
}

template <int NW> __global__ __launch_bounds__(64) void k_ed_sign_enc(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	u8 *out = A.out + (size_t)i * 32;
	const u32 st = A.stR[i];
	if (st != 0) {
		// r = 0 mod q: the neutral element (0, 1); a failed multiplication: an error
		for (int b = 0; b < 32; b++) {
			out[b] = (u8)((st == 2 && b == 0) ? 1 : 0);
		}
		A.status[i] = (st == 2) ? 0 : 1;
		return;
	}
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const u8 *src = A.Rw + (size_t)i * 64;
	const Fe<NW> X = fe_to_mont<NW>(fe_load_be<NW>(src, 32), slot), v = fe_to_mont<NW>(fe_load_be<NW>(src + 32, 32), slot);
	const Fe<NW> u = fe_sub<NW>(X, fe_const<NW>(A.A3), slot);
	const Fe<NW> up1 = fe_add<NW>(u, one, slot);
	Fe<NW> d11;
	const Fe<NW> dd = fe_mul<NW>(up1, v, slot);
	const Fe<NW> dinv = fe_mul<NW>(fe_sqr_n<NW>(fe_pow_2_250m1<NW>(dd, &d11, slot), 5, slot), d11, slot);  // dd^(p-2)
	const Fe<NW> vinv = fe_mul<NW>(dinv, up1, slot), uinv = fe_mul<NW>(dinv, v, slot);   // 1 / v, 1 / (u + 1)
	const Fe<NW> xe = fe_mul<NW>(fe_mul<NW>(fe_const<NW>(A.alpha), u, slot), vinv, slot);
	const Fe<NW> ye = fe_mul<NW>(fe_sub<NW>(u, one, slot), uinv, slot);
	const Fe<NW> xp = fe_from_mont<NW>(xe, slot);
	Fe<NW> yp = fe_from_mont<NW>(ye, slot);
	yp.v[7] |= (xp.v[0] & 1u) << 31;                 // y < 2^255; bit 255 carries the parity of x
	ed_store_le32<NW>(out, yp);
	A.status[i] = fe_is_zero<NW>(dd) ? 1 : 0;        // cannot happen for points of the prime-order subgroup
}

template <int NW> __global__ __launch_bounds__(64) void k_ed_sign_S(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const Fe<NW> r2 = fe_const<NW>(ConstTab<NW>::get(qs).r2);
	const Fe<NW> r = ed_le64_mod_q<NW>(A.r_hash + (size_t)i * 64, qs);
	const Fe<NW> h = ed_le64_mod_q<NW>(A.hram + (size_t)i * 64, qs);
	const Fe<NW> a = fe_load_le<NW>(A.a + (size_t)i * 32, 32);                       // < 2^256, reduced by the product with R^2
	const Fe<NW> ha = fe_mul<NW>(h, fe_mul<NW>(a, r2, qs), qs);                      // h (a R) / R = h a mod q
	ed_store_le32<NW>(A.out + (size_t)i * 32, fe_add<NW>(r, ha, qs));
}

// Fixed exponents of p = 2^448 - 2^224 - 1 by addition chains (the prime is pinned by ed448_setup):
//   f(k) = w^(2^k - 1): f(2k) = f(k)^(2^k) f(k), f(k + 1) = f(k)^2 w; 222 = 2 x (2 x (2 x (2 x (2 x (2 x 3 + 1)) + 1) + 1) + 1)
//   (p - 3) / 4 = 2^446 - 2^222 - 1 = 2^223 (2^223 - 1) + (2^222 - 1):  w^((p-3)/4) = f(223)^(2^223) f(222)     446 S + 14 M
//   p - 2 = 4 (p - 3) / 4 + 1:                                            w^(p-2) = (w^((p-3)/4))^4 w              448 S + 15 M
// instead of ~890 multiplications each with square-and-multiply.
template <int NW> static __device__ Fe<NW> fe_pow_p448_e34(const Fe<NW> &w, int slot)
{
	// steps from f(1) = w to f(222): k > 0 doubles f(k) -> f(2k), k = 0 increments f(k) -> f(k + 1)
	const int steps[12] = {1, 0, 3, 6, 0, 13, 0, 27, 0, 55, 0, 111};
	Fe<NW> f = w;
#pragma unroll 1
	for (int s = 0; s < 12; s++) {
		const int k = steps[s];
		const Fe<NW> g = (k == 0) ? w : f;
		f = fe_mul<NW>(fe_sqr_n<NW>(f, k == 0 ? 1 : k, slot), g, slot);
	}
	const Fe<NW> f223 = fe_mul<NW>(fe_mul<NW>(f, f, slot), w, slot);
	return fe_mul<NW>(fe_sqr_n<NW>(f223, 223, slot), f, slot);
}

template <int NW> static __device__ Fe<NW> fe_inv_p448(const Fe<NW> &w, int slot)   // 0 -> 0, like Fermat's
{
	return fe_mul<NW>(fe_sqr_n<NW>(fe_pow_p448_e34<NW>(w, slot), 2, slot), w, slot);
}

// One lane decodes A and R of an item; the four field inversions of each point (a - d y^2, the two isogeny
// denominators, 1 - y', x') shrink to two per ITEM by Montgomery's trick, and the first one rides on the square
// root: x = u^3 v (u^5 v^3)^((p - 3) / 4) for x^2 = u / v (RFC 8032 5.2.3), valid iff v x^2 == u.
template <int NW> __global__ __launch_bounds__(64) void k_ed448_decode(EcamdEd448DecodeArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const Fe<NW> zero = fe_zero<NW>();
	const Fe<NW> two = fe_add<NW>(one, one, slot);
	Fe<NW> x[2], ym[2], xx[2], yy[2], d1[2], d2[2];
	bool ok[2], neutral[2] = {false, false};
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		const u8 *src = (k == 0) ? A.encA + (size_t)i * A.strideA : A.encR + (size_t)i * A.strideR;
		const Fe<NW> y = fe_load_le<NW>(src, 56);
		const u32 last = src[56];
		const u32 x0 = last >> 7;
		bool good = ((last & 0x7fu) == 0) & fe_lt_p<NW>(y, slot);       // the 57th byte only carries the sign
		ym[k] = fe_to_mont<NW>(y, slot);
		yy[k] = fe_mul<NW>(ym[k], ym[k], slot);
		const Fe<NW> u = fe_sub<NW>(one, yy[k], slot);                                     // 1 - y^2
		const Fe<NW> v = fe_sub<NW>(one, fe_mul<NW>(fe_const<NW>(A.d448), yy[k], slot), slot);  // a - d y^2, a = 1
		good = good & !fe_is_zero<NW>(v);                                                  // fp_inv(0)
		const Fe<NW> v2 = fe_mul<NW>(v, v, slot), u2 = fe_mul<NW>(u, u, slot);
		const Fe<NW> u3v = fe_mul<NW>(fe_mul<NW>(u2, u, slot), v, slot);
		const Fe<NW> u5v3 = fe_mul<NW>(fe_mul<NW>(u3v, u2, slot), v2, slot);
		Fe<NW> r = fe_mul<NW>(u3v, fe_pow_p448_e34<NW>(u5v3, slot), slot);
		good = good & fe_eq<NW>(fe_mul<NW>(v, fe_mul<NW>(r, r, slot), slot), u);           // u / v has no root: error
		const Fe<NW> rp = fe_from_mont<NW>(r, slot);
		r = fe_select<NW>((rp.v[0] & 1u) != x0, fe_sub<NW>(zero, r, slot), r);
		good = good & !(fe_is_zero<NW>(r) & (x0 == 1u));
		x[k] = r;
		xx[k] = fe_mul<NW>(r, r, slot);
		d1[k] = fe_sub<NW>(fe_sub<NW>(two, xx[k], slot), yy[k], slot);                     // 2 - x^2 - y^2
		d2[k] = fe_sub<NW>(yy[k], xx[k], slot);                                            // y^2 - x^2
		good = good & !fe_is_zero<NW>(d1[k]) & !fe_is_zero<NW>(d2[k]);                     // fp_inv(0)
		ok[k] = good;
		d1[k] = fe_select<NW>(good, d1[k], one);
		d2[k] = fe_select<NW>(good, d2[k], one);
	}
	// isogeny: 1 / (d1 d2) for both points from one inversion
	Fe<NW> X[2], Y[2], omy[2];
	{
		const Fe<NW> pa = fe_mul<NW>(d1[0], d2[0], slot), pr = fe_mul<NW>(d1[1], d2[1], slot);
		const Fe<NW> inv = fe_inv_p448<NW>(fe_mul<NW>(pa, pr, slot), slot);
		const Fe<NW> ia = fe_mul<NW>(inv, pr, slot), ir = fe_mul<NW>(inv, pa, slot);       // 1 / (d1 d2) of A, of R
#pragma unroll 1
		for (int k = 0; k < 2; k++) {
			const Fe<NW> di = (k == 0) ? ia : ir;
			X[k] = fe_mul<NW>(fe_mul<NW>(fe_const<NW>(A.alpha), fe_mul<NW>(x[k], ym[k], slot), slot), fe_mul<NW>(di, d2[k], slot), slot);
			Y[k] = fe_mul<NW>(fe_add<NW>(xx[k], yy[k], slot), fe_mul<NW>(di, d1[k], slot), slot);
			const Fe<NW> X2 = fe_mul<NW>(X[k], X[k], slot), Y2 = fe_mul<NW>(Y[k], Y[k], slot);
			const Fe<NW> l = fe_add<NW>(X2, Y2, slot);
			const Fe<NW> r = fe_add<NW>(one, fe_mul<NW>(fe_const<NW>(A.diso), fe_mul<NW>(X2, Y2, slot), slot), slot);
			omy[k] = fe_sub<NW>(one, Y[k], slot);
			// on the Edwards model of curve448.  (0, 1) -- the image of both (0, 1) and (0, -1) of Ed448 -- is the neutral
			// element: aff_pt_edwards_to_prj_pt_shortw maps it to the point at infinity (fine for R; a key is then rejected
			// as small-order).  X = 0 with Y = -1 (order 2) dies in fp_inv(0) of the map to the Montgomery model.
			neutral[k] = ok[k] & fe_eq<NW>(l, r) & fe_is_zero<NW>(X[k]) & fe_is_zero<NW>(omy[k]);
			ok[k] = ok[k] & fe_eq<NW>(l, r) & !fe_is_zero<NW>(X[k]) & !fe_is_zero<NW>(omy[k]);
			X[k] = fe_select<NW>(ok[k], X[k], one);
			omy[k] = fe_select<NW>(ok[k], omy[k], one);
		}
	}
	{
		const Fe<NW> pa = fe_mul<NW>(omy[0], X[0], slot), pr = fe_mul<NW>(omy[1], X[1], slot);
		const Fe<NW> inv = fe_inv_p448<NW>(fe_mul<NW>(pa, pr, slot), slot);
		const Fe<NW> ia = fe_mul<NW>(inv, pr, slot), ir = fe_mul<NW>(inv, pa, slot);       // 1 / ((1 - Y) X) of A, of R
#pragma unroll 1
		for (int k = 0; k < 2; k++) {
			const Fe<NW> mi = (k == 0) ? ia : ir;
			const Fe<NW> u = fe_mul<NW>(fe_add<NW>(one, Y[k], slot), fe_mul<NW>(mi, X[k], slot), slot);
			const Fe<NW> v = fe_mul<NW>(fe_mul<NW>(fe_const<NW>(A.alpha), u, slot), fe_mul<NW>(mi, omy[k], slot), slot);
			const Fe<NW> Xw = fe_sub<NW>(fe_const<NW>(A.A3), u, slot);                       // (A, B) = (-156326, -1)
			const Fe<NW> Yw = fe_sub<NW>(zero, v, slot);
			u8 *pd = (k == 0 ? A.pointsA : A.pointsR) + (size_t)i * 112;
			fe_store_be<NW>(pd, 56, ok[k] ? fe_from_mont<NW>(Xw, slot) : zero);
			fe_store_be<NW>(pd + 56, 56, ok[k] ? fe_from_mont<NW>(Yw, slot) : zero);
			(k == 0 ? A.flagsA : A.flagsR)[i] = ok[k] ? 0 : ((k == 1 && neutral[1]) ? 2 : 1);   // 2: R is the point at infinity
		}
	}
}

// ------------------------------------------------------------------------------------------
// Ed448 signing around the caller's hashes (_eddsa_sign, sig/eddsa.c:1554-1870, the EDDSA448 branches):
//   k_ed448_sign_r:   r = SHAKE256(dom4 || prefix || PH(M), 114) little-endian mod q (:1731); the scalar of the multiplication
//                     is r 4^-1 mod q (:1737-1746: "because of the 4-isogeny we must divide our scalar by 4"), big-endian out
//   k_ed448_sign_enc: [r/4]G on WEI448 -> prj_pt_shortw_to_aff_pt_edwards (infinity -> (0, 1); (u, v) = (A/3 - X, -Y) on libecc's
//                     Montgomery model (A, B) = (-156326, -1); x = alpha u / v, y = (u - 1) / (u + 1)) -> eddsa_encode_point
//                     (:350-395): the 4-isogeny back to Edwards448, x1 = (4 x y / alpha) / (y^2 - x^2),
//                     y1 = (2 - x^2 - y^2) / (x^2 + y^2); 57 bytes: y1 little-endian, the parity of x1 in bit 7 of the last one.
//                     Two inversions per item (the five denominators by Montgomery's trick), addition chains for p - 2.
//   k_ed448_sign_S:   S = (r + h a) mod q with h = SHAKE256(dom4 || R || A || PH(M), 114) mod q, a = the clamped secret
//                     scalar (57 bytes little-endian, the last one 0), 57 bytes little-endian out
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ Fe<NW> ed448_le114_mod_q(const u8 *hp, int qs)
{
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const Fe<NW> lo = fe_load_le<NW>(hp, 56), mid = fe_load_le<NW>(hp + 56, 56), hi = fe_load_le<NW>(hp + 112, 2);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	// R = 2^448: x R2 / R = x 2^448 (mod q); lo needs one more reduction step (it may exceed q)
	const Fe<NW> lor = fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs);
	const Fe<NW> midr = fe_mul<NW>(mid, r2, qs);
	const Fe<NW> hir = fe_mul<NW>(fe_mul<NW>(hi, r2, qs), r2, qs);
	return fe_add<NW>(fe_add<NW>(lor, midr, qs), hir, qs);
}

template <int NW> __global__ __launch_bounds__(64) void k_ed448_sign_r(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const Fe<NW> r = ed448_le114_mod_q<NW>(A.r_hash + (size_t)i * 114, qs);
	const Fe<NW> r2 = fe_const<NW>(ConstTab<NW>::get(qs).r2);
	const Fe<NW> r4 = fe_mul<NW>(fe_mul<NW>(r, r2, qs), fe_const<NW>(A.c4), qs);      // (r R) c4 / R = r / 4 mod q
	fe_store_be<NW>(A.r_be + (size_t)i * 56, 56, r4);
}

template <int NW> __global__ __launch_bounds__(64) void k_ed448_sign_enc(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	u8 *out = A.out + (size_t)i * 57;
	const u32 st = A.stR[i];
	if (st != 0) {
		// r = 0 mod q: the neutral element, encoded as y1 = 1; a failed multiplication: an error
		for (int b = 0; b < 57; b++) {
			out[b] = (u8)((st == 2 && b == 0) ? 1 : 0);
		}
		A.status[i] = (st == 2) ? 0 : 1;
		return;
	}
	const Fe<NW> zero = fe_zero<NW>();
	const Fe<NW> one = fe_const<NW>(ConstTab<NW>::get(slot).one);
	const Fe<NW> two = fe_add<NW>(one, one, slot);
	const Fe<NW> alpha = fe_const<NW>(A.alpha);
	const u8 *src = A.Rw + (size_t)i * 112;
	const Fe<NW> Xw = fe_to_mont<NW>(fe_load_be<NW>(src, 56), slot), Yw = fe_to_mont<NW>(fe_load_be<NW>(src + 56, 56), slot);
	const Fe<NW> u = fe_sub<NW>(fe_const<NW>(A.A3), Xw, slot), v = fe_sub<NW>(zero, Yw, slot);
	const Fe<NW> up1 = fe_add<NW>(u, one, slot);
	const Fe<NW> d1 = fe_mul<NW>(up1, v, slot);
	bool ok = !fe_is_zero<NW>(d1);                                                      // fp_inv(0)
	const Fe<NW> i1 = fe_inv_p448<NW>(d1, slot);
	const Fe<NW> x = fe_mul<NW>(fe_mul<NW>(alpha, u, slot), fe_mul<NW>(i1, up1, slot), slot);   // alpha u / v
	const Fe<NW> y = fe_mul<NW>(fe_sub<NW>(u, one, slot), fe_mul<NW>(i1, v, slot), slot);       // (u - 1) / (u + 1)
	const Fe<NW> xx = fe_mul<NW>(x, x, slot), yy = fe_mul<NW>(y, y, slot);
	const Fe<NW> den1 = fe_sub<NW>(yy, xx, slot), den2 = fe_add<NW>(xx, yy, slot);
	const Fe<NW> d12 = fe_mul<NW>(den1, den2, slot);
	ok = ok & !fe_is_zero<NW>(d12);
	const Fe<NW> i2 = fe_inv_p448<NW>(fe_mul<NW>(d12, alpha, slot), slot);               // 1 / (den1 den2 alpha)
	const Fe<NW> inv1 = fe_mul<NW>(i2, fe_mul<NW>(den2, alpha, slot), slot);             // 1 / den1
	const Fe<NW> inv2 = fe_mul<NW>(i2, fe_mul<NW>(den1, alpha, slot), slot);             // 1 / den2
	const Fe<NW> inva = fe_mul<NW>(i2, d12, slot);                                       // 1 / alpha
	Fe<NW> x1 = fe_mul<NW>(x, y, slot);
	x1 = fe_add<NW>(x1, x1, slot);
	x1 = fe_add<NW>(x1, x1, slot);                                                       // 4 x y
	x1 = fe_mul<NW>(fe_mul<NW>(x1, inv1, slot), inva, slot);
	const Fe<NW> y1 = fe_mul<NW>(fe_sub<NW>(fe_sub<NW>(two, xx, slot), yy, slot), inv2, slot);
	const Fe<NW> x1p = fe_from_mont<NW>(x1, slot), y1p = fe_from_mont<NW>(y1, slot);
	for (int b = 0; b < 56; b++) {
		out[b] = ok ? (u8)(y1p.v[b >> 2] >> (8 * (b & 3))) : 0;
	}
	out[56] = ok ? (u8)((x1p.v[0] & 1u) << 7) : 0;
	A.status[i] = ok ? 0 : 1;
}

template <int NW> __global__ __launch_bounds__(64) void k_ed448_sign_S(EcamdEdSignArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const Fe<NW> r2 = fe_const<NW>(ConstTab<NW>::get(qs).r2);
	const Fe<NW> r = ed448_le114_mod_q<NW>(A.r_hash + (size_t)i * 114, qs);
	const Fe<NW> h = ed448_le114_mod_q<NW>(A.hram + (size_t)i * 114, qs);
	const Fe<NW> a = fe_load_le<NW>(A.a + (size_t)i * 57, 56);                          // byte 56 of the clamped scalar is 0
	const Fe<NW> ha = fe_mul<NW>(h, fe_mul<NW>(a, r2, qs), qs);
	const Fe<NW> S = fe_add<NW>(r, ha, qs);
	u8 *out = A.out + (size_t)i * 57;
	for (int b = 0; b < 56; b++) {
		out[b] = (u8)(S.v[b >> 2] >> (8 * (b & 3)));
	}
	out[56] = 0;
}

// S < q (57 bytes little-endian, the last one must be 0); h = 114-byte hash mod q, then a = 4 h mod q as the reference does.
// The reference multiplies the STORED key A' = [c4]A (c4 = 4^-1 mod q, eddsa_import_pub_key) by a.  [a]([c4]A) = [a c4]A,
// and the order of A divides 4q, so one multiplication of the decoded A by k = a c4 mod 4q gives the same point:
// k = h (mod q) and k = (a mod 4)(c4 mod 4) (mod 4), i.e. k = h + t q with t in 0..3 -- written out instead of a.
template <int NW> __global__ __launch_bounds__(64) void k_ed448_scal(EcamdEdScalArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int qs = A.qslot;
	const CurveK<NW> &Q = ConstTab<NW>::get(qs);
	const u8 *sp = A.sigs + (size_t)i * 114 + 57;
	const Fe<NW> S = fe_load_le<NW>(sp, 56);
	const bool ok = (sp[56] == 0) & fe_lt_p<NW>(S, qs);
	const u8 *hp = A.hram + (size_t)i * 114;
	const Fe<NW> lo = fe_load_le<NW>(hp, 56), mid = fe_load_le<NW>(hp + 56, 56), hi = fe_load_le<NW>(hp + 112, 2);
	const Fe<NW> r2 = fe_const<NW>(Q.r2);
	Fe<NW> onep = fe_zero<NW>();
	onep.v[0] = 1u;
	// R = 2^448: x R2 / R = x 2^448 (mod q); lo, mid need one reduction step first (they may exceed q)
	const Fe<NW> lor = fe_mul<NW>(fe_mul<NW>(lo, r2, qs), onep, qs);                        // lo mod q
	const Fe<NW> midr = fe_mul<NW>(mid, r2, qs);                                             // mid 2^448 mod q
	const Fe<NW> hir = fe_mul<NW>(fe_mul<NW>(hi, r2, qs), r2, qs);                           // hi 2^896 mod q
	const Fe<NW> h = fe_add<NW>(fe_add<NW>(lor, midr, qs), hir, qs);                        // h mod q, canonical
	Fe<NW> a4 = fe_add<NW>(h, h, qs);
	a4 = fe_add<NW>(a4, a4, qs);                                                             // a = 4 h mod q, canonical
	const u32 r4 = ((a4.v[0] & 3u) * A.c4_mod4) & 3u;                                        // k mod 4
	const u32 t = ((r4 - (h.v[0] & 3u)) * (Q.p[0] & 3u)) & 3u;                               // q^-1 = q (mod 4)
	Fe<NW> k = h;                                                                            // h + t q < 4q < 2^448
	u64 cy = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		const u64 sum = (u64)k.v[w] + (u64)t * Q.p[w] + cy;
		k.v[w] = (u32)sum;
		cy = sum >> 32;
	}
	fe_store_be<NW>(A.S_be + (size_t)i * 56, 56, ok ? S : fe_zero<NW>());
	fe_store_be<NW>(A.h_be + (size_t)i * 56, 56, k);
	if (A.ne_be != nullptr) {
		// the whole-batch form works on the prime-order components only (its test is cofactored): the key's scalar is -h mod q
		fe_store_be<NW>(A.ne_be + (size_t)i * 56, 56, fe_sub<NW>(fe_zero<NW>(), h, qs));
	}
	A.flags[i] = ok ? 0 : 1;
}

// Ed448 whole-batch verification: what _eddsa_verify_batch (sig/eddsa.c:2580-2860) rejects item by item before its combination -- a key or a
// commitment that does not decode (:2766), S >= q (:2783), [cofactor]A = infinity (:2801-2806) -- as one word for the batch.  A commitment
// that decodes to the neutral element has no affine Weierstrass form: "not decided here" as well (the item form accepts it).
template <int NW> __global__ __launch_bounds__(64) void k_ed_msm_gate(const u8 *keys, const u8 *fA, const u8 *fR, const u8 *fS, u32 n, u32 clen, u32 cof_dbl,
								       int slot, u32 *gate)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	bool bad = (fA[i] | fR[i] | fS[i]) != 0;
	if (!bad && cof_dbl != 0u) {   // (cof_dbl 0: the combination's own kernels look at the keys -- k_bkt_points_g, k_msm_table_g)
		Pt<NW> K4 = ed_load_neg<NW>(keys + (size_t)i * 2 * clen, 0, (int)clen, false, slot);
		for (u32 k = 0; k < cof_dbl; k++) {
			K4 = pt_dbl<NW>(K4, slot);
		}
		bad = fe_is_zero<NW>(K4.Z);
	}
	if (bad) {
		atomicOr(gate, 1u);
	}
}
__global__ void k_verdict_or(u8 *verdict, const u8 *piece, const u32 *gate)
{
	if (blockIdx.x == 0 && threadIdx.x == 0 && ((piece && piece[0] != 0) || (gate && gate[0] != 0u))) {
		verdict[0] = 1;
	}
}

// ------------------------------------------------------------------------------------------
// Point decompression: aff_pt_y_from_x (curves/aff_pt.c:102-131) = the two roots of x^3 + a x + b through fp_sqrt
// (fp/fp_sqrt.c:107-251).  fp_sqrt is Tonelli-Shanks with z = the smallest quadratic non-residue (found by counting up
// from 0, :197-200), so WHICH root it returns as sqrt1 is a fixed function of the input; the steps below are its steps
// with the two exponentiations n^((q+1)/2), n^q (p - 1 = q 2^s) sharing w = n^((q-1)/2):
//   n = 0 -> both roots 0;  r = w n, t = r w;  Legendre symbol = t^(2^(s-1)) must be 1 (else fp_sqrt returns -1);
//   while t != 1: i = least i with t^(2^i) = 1;  b = c^(2^(m-i-1));  r = r b;  c = b^2;  t = t c;  m = i.
// For p = 3 mod 4 (s = 1) the loop never runs and r = n^((p+1)/4), the reference's shortcut (:184-192).
// The control flow after the exponentiation is per lane (data dependent, bounded by s^2 squarings: secp224r1 has s = 96).
// ------------------------------------------------------------------------------------------
template <int NW> __global__ __launch_bounds__(64) void k_y_from_x(EcamdYfromXArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int clen = (int)A.clen;
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	const u8 *rec = A.x + (size_t)i * A.xstride;
	const u8 *xp = rec + (A.xstride - A.clen);
	const Fe<NW> zero = fe_zero<NW>();
	const Fe<NW> one = fe_const<NW>(K.one);
	const Fe<NW> x = fe_load_be<NW>(xp, clen);
	bool ok = fe_lt_p<NW>(x, slot);                                 // fp_init_from_buf rejects x >= p (fp/fp.c:441-442)
	u32 want = 0;
	if (A.mode == 1) {
		const u32 pre = rec[0];
		ok = ok & (A.xstride == A.clen + 1) & ((pre == 2u) | (pre == 3u));
		want = pre & 1u;
	}
	const Fe<NW> xm = fe_to_mont<NW>(x, slot);
	// x^3 + a x + b in the reference's order (:118-122)
	Fe<NW> n = fe_mul<NW>(fe_mul<NW>(xm, xm, slot), xm, slot);
	n = fe_add<NW>(n, fe_mul<NW>(xm, fe_const<NW>(K.a), slot), slot);
	n = fe_add<NW>(n, fe_const<NW>(K.b), slot);
	// w = n^((q-1)/2), wave-uniform exponent
	Fe<NW> w = one;
	for (int b = (int)A.ebits - 1; b >= 0; b--) {
		w = fe_mul<NW>(w, w, slot);
		if ((A.e[b >> 5] >> (b & 31)) & 1u) {
			w = fe_mul<NW>(w, n, slot);
		}
	}
	Fe<NW> r = fe_mul<NW>(w, n, slot);
	Fe<NW> t = fe_mul<NW>(r, w, slot);
	const bool nzero = fe_is_zero<NW>(n);
	{
		Fe<NW> l = t;
		for (u32 k = 1; k < A.s; k++) {
			l = fe_mul<NW>(l, l, slot);
		}
		ok = ok & (nzero | fe_eq<NW>(l, one));                     // not a square: fp_sqrt fails
	}
	if (ok && !nzero) {
		Fe<NW> c = fe_const<NW>(A.c);
		u32 m = A.s;
		while (!fe_eq<NW>(t, one)) {
			u32 ii = 1;
			Fe<NW> tt = fe_mul<NW>(t, t, slot);
			while (!fe_eq<NW>(tt, one) && ii < m) {
				tt = fe_mul<NW>(tt, tt, slot);
				ii++;
			}
			if (ii >= m) {
				ok = false;                                       // "should not happen" (:222-226)
				break;
			}
			Fe<NW> b = c;
			for (u32 k = 0; k < m - ii - 1; k++) {
				b = fe_mul<NW>(b, b, slot);
			}
			r = fe_mul<NW>(r, b, slot);
			c = fe_mul<NW>(b, b, slot);
			t = fe_mul<NW>(t, c, slot);
			m = ii;
		}
	}
	const Fe<NW> r1 = (ok && !nzero) ? fe_from_mont<NW>(r, slot) : zero;
	const Fe<NW> r2 = (ok && !nzero) ? fe_from_mont<NW>(fe_sub<NW>(zero, r, slot), slot) : zero;
	if (A.mode == 0) {
		fe_store_be<NW>(A.y1 + (size_t)i * clen, clen, r1);
		fe_store_be<NW>(A.y2 + (size_t)i * clen, clen, r2);
	} else {
		// SEC 1 section 2.3.4: the prefix carries y mod 2.  y = 0 has no odd form: 0x03 with n = 0 is an error.
		const bool pick1 = ((r1.v[0] & 1u) == want);
		ok = ok & !(nzero & (want == 1u));
		u8 *o = A.aff + (size_t)i * 2 * clen;
		fe_store_be<NW>(o, clen, ok ? x : zero);
		fe_store_be<NW>(o + clen, clen, ok ? (pick1 ? r1 : r2) : zero);
	}
	A.status[i] = ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Projective wire format X || Y || Z (prj_pt_import_from_buf / prj_pt_export_to_buf, curves/prj_pt.c:462,562)
//   k_prj_import: each coordinate < p, projective on-curve check Y^2 Z = X^3 + a X Z^2 + b Z^3 in the
//     reference's operation order (:144-190); Z = 0 on the curve is the point at infinity -- (0:0:0) also
//     satisfies the equation: prj_pt_unique of it fails as "infinity", prj_pt_mul of it fails in the
//     ladder (error); finite points are normalised to affine for the scalar-mult pipeline;
//   k_prj_export: affine + status -> X || Y || 1 (the unique representative prj_pt_unique yields).
// ------------------------------------------------------------------------------------------
template <int NW> __global__ __launch_bounds__(64) void k_prj_import(EcamdPrjInArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot;
	const int clen = (int)A.clen;
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	const u8 *src = A.in + (size_t)i * 3 * clen;
	u8 *dst = A.aff + (size_t)i * 2 * clen;
	const Fe<NW> X = fe_load_be<NW>(src, clen), Y = fe_load_be<NW>(src + clen, clen), Z = fe_load_be<NW>(src + 2 * clen, clen);
	bool ok = fe_lt_p<NW>(X, slot) & fe_lt_p<NW>(Y, slot) & fe_lt_p<NW>(Z, slot);
	const Fe<NW> Xm = fe_to_mont<NW>(X, slot), Ym = fe_to_mont<NW>(Y, slot), Zm = fe_to_mont<NW>(Z, slot);
	{
		const Fe<NW> z2 = fe_mul<NW>(Zm, Zm, slot);
		const Fe<NW> x2 = fe_mul<NW>(Xm, Xm, slot);
		Fe<NW> rhs = fe_add<NW>(x2, fe_mul<NW>(fe_const<NW>(K.a), z2, slot), slot);      // X^2 + a Z^2
		rhs = fe_mul<NW>(rhs, Xm, slot);
		rhs = fe_add<NW>(rhs, fe_mul<NW>(fe_mul<NW>(fe_const<NW>(K.b), z2, slot), Zm, slot), slot);
		const Fe<NW> lhs = fe_mul<NW>(fe_mul<NW>(Ym, Ym, slot), Zm, slot);
		ok = ok & fe_eq<NW>(lhs, rhs);
	}
	u32 st = 0;
	if (!ok) {
		st = 1;
	} else if (fe_is_zero<NW>(Z)) {
		st = (A.for_mul && fe_is_zero<NW>(X) && fe_is_zero<NW>(Y)) ? 1u : 2u;
	}
	Fe<NW> ax = fe_zero<NW>(), ay = fe_zero<NW>();
	if (st == 0) {
		// Z = 1 (a key imported from affine bytes, or normalised before): nothing to divide by -- the lanes of a wave whose
		// keys all have Z = 1 skip the Fermat inversion altogether
		bool z_one = Z.v[0] == 1u;
#pragma unroll
		for (int w = 1; w < NW; w++) {
			z_one = z_one & (Z.v[w] == 0u);
		}
		if (z_one) {
			ax = X;
			ay = Y;
		} else {
			const Fe<NW> zi = fe_inv<NW>(Zm, slot);
			ax = fe_from_mont<NW>(fe_mul<NW>(Xm, zi, slot), slot);
			ay = fe_from_mont<NW>(fe_mul<NW>(Ym, zi, slot), slot);
		}
	}
	fe_store_be<NW>(dst, clen, ax);
	fe_store_be<NW>(dst + clen, clen, ay);
	A.pre[i] = (u8)st;
}

__global__ __launch_bounds__(256) void k_prj_export(EcamdPrjOutArgs A)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 clen = A.clen;
	const u32 ow = A.out_prj ? 3 * clen : 2 * clen;
	const u32 pre = A.pre ? A.pre[i] : 0u;
	const u32 st = pre ? pre : (A.st ? A.st[i] : 0u);
	const u8 *src = A.aff + (size_t)i * 2 * clen;
	u8 *dst = A.out + (size_t)i * ow;
	for (u32 b = 0; b < ow; b++) {
		u8 v = 0;
		if (st == 0) {
			v = (b < 2 * clen) ? src[b] : (u8)(b == ow - 1 ? 1 : 0);
		}
		dst[b] = v;
	}
	A.status[i] = (u8)st;
}

// ------------------------------------------------------------------------------------------
// ECC-CDH glue (ecccdh_derive_secret, ecdh/ecccdh.c:187-224): byte moves only, one thread per item.
//   k_cdh_gate: cofactor curves -- the peer key must be in the subgroup ([q]Q = infinity, sig/ec_key.c:199-205) and
//               [h]Q must not be infinity (:202-207); an item failing either gets an undecodable point so that the
//               last scalar multiplication reports it as an import error.
//   k_cdh_fin:  shared secret = x coordinate of [d]Q' (:222-224); infinity and import errors are both -1 there.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cdh_gate(EcamdCdhArgs A)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	if (A.st_sub[i] != 2 || A.st_h[i] != 0) {
		u8 *dst = A.hq + (size_t)i * 2 * A.clen;
		for (u32 b = 0; b < 2 * A.clen; b++) {
			dst[b] = 0xff;
		}
	}
}

__global__ __launch_bounds__(256) void k_cdh_fin(EcamdCdhArgs A)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 bad = A.st[i] ? 1u : 0u;
	const u8 *src = A.pts + (size_t)i * 2 * A.clen;
	u8 *dst = A.secrets + (size_t)i * A.clen;
	for (u32 b = 0; b < A.clen; b++) {
		dst[b] = bad ? (u8)0 : src[b];
	}
	A.status[i] = (u8)bad;
}

hipError_t ecamd_launch_cdh_gate(const EcamdCdhArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_cdh_gate, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_cdh_fin(const EcamdCdhArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_cdh_fin, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host-side dispatch on the word count
// ------------------------------------------------------------------------------------------
#define ECAMD_FOR_NW(X) X(6) X(7) X(8) X(10) X(12) X(14) X(16) X(17)

int ecamd_nw_supported(int nw)
{
	switch (nw) {
#define X(N) case N: return 1;
		ECAMD_FOR_NW(X)
#undef X
	default: return 0;
	}
}

size_t ecamd_curvek_bytes(int nw)
{
	switch (nw) {
#define X(N) case N: return sizeof(CurveK<N>);
		ECAMD_FOR_NW(X)
#undef X
	default: return 0;
	}
}

hipError_t ecamd_upload_curve(int nw, int slot, const void *curvek, size_t bytes)
{
	switch (nw) {
#define X(N) case N: \
		if (bytes != sizeof(CurveK<N>)) return hipErrorInvalidValue; \
		return hipMemcpyToSymbol(HIP_SYMBOL(g_curves_##N), curvek, bytes, (size_t)slot * sizeof(CurveK<N>), hipMemcpyHostToDevice);
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}

hipError_t ecamd_launch_smul(int nw, const EcamdSmulArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	if (a.masked) {
		switch (nw) {
#define X(N) case N: hipLaunchKernelGGL((k_smul<N, true>), grid, block, 0, s, a); break;
			ECAMD_FOR_NW(X)
#undef X
		default: return hipErrorInvalidValue;
		}
		return hipGetLastError();
	}
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL((k_smul<N, false>), grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ecdsa_prep(int nw, const EcamdEcdsaPrepArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const uint32_t kp = (uint32_t)ecdsa_prep_items(nw);
	const uint32_t lanes = (a.n + kp - 1) / kp;
	const dim3 grid((lanes + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_ecdsa_prep<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ecdsa_sign(int nw, const EcamdEcdsaSignArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const uint32_t lanes = (a.n + ECDSA_PREP_K - 1) / ECDSA_PREP_K;
	const dim3 grid((lanes + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_ecdsa_sign<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_xdh_prep(int nw, const EcamdXdhPrepArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_xdh_prep<N>, grid, block, 0, s, a); break;
		X(8) X(14)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_xdh_fin(int nw, const EcamdXdhFinArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_xdh_fin<N>, grid, block, 0, s, a); break;
		X(8) X(14)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ecdsa_fin(int nw, const EcamdEcdsaFinArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_ecdsa_fin<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// prj_pt_mul_blind (curves/prj_pt.c:1782-1822): the scalar actually multiplied is m + b * #E with b random in [1, #E);
// here b is the caller's (randomness stays on the host, as for nonces).  Plain multi-precision arithmetic per lane.
// ------------------------------------------------------------------------------------------
#define BLIND_MAXW 72
__global__ __launch_bounds__(64) void k_blind_scalar(EcamdBlindArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u32 acc[BLIND_MAXW];
	const u32 ow = A.owords, bw = (A.blen + 3) / 4, mw = (A.mlen + 3) / 4, tw = (A.outlen + 3) / 4;
	for (u32 w = 0; w < tw; w++) {
		acc[w] = 0;
	}
	const u8 *bp = A.b + (size_t)i * A.blen, *mp = A.m + (size_t)i * A.mlen;
	u32 nz = 0, lt = 0;    // b != 0;  b < #E (decided by the most significant differing word)
	bool decided = false;
	for (int w = (int)(bw > ow ? bw : ow) - 1; w >= 0; w--) {
		u32 x = 0;
		for (u32 k = 0; k < 4; k++) {
			const u32 pos = 4 * (u32)w + k;
			if (pos < A.blen) {
				x |= (u32)bp[A.blen - 1 - pos] << (8 * k);
			}
		}
		nz |= x;
		const u32 o = ((u32)w < ow) ? A.order[w] : 0u;
		if (!decided && x != o) {
			lt = (x < o) ? 1u : 0u;
			decided = true;
		}
	}
	for (u32 w = 0; w < bw; w++) {
		u32 x = 0;
		for (u32 k = 0; k < 4; k++) {
			const u32 pos = 4 * w + k;
			if (pos < A.blen) {
				x |= (u32)bp[A.blen - 1 - pos] << (8 * k);
			}
		}
		u64 carry = 0;
		for (u32 j = 0; j < ow && w + j < tw; j++) {
			const u64 t = (u64)x * A.order[j] + acc[w + j] + carry;
			acc[w + j] = (u32)t;
			carry = t >> 32;
		}
		for (u32 j = w + ow; carry && j < tw; j++) {
			const u64 t = (u64)acc[j] + carry;
			acc[j] = (u32)t;
			carry = t >> 32;
		}
	}
	u64 carry = 0;
	for (u32 w = 0; w < tw; w++) {
		u32 x = 0;
		if (w < mw) {
			for (u32 k = 0; k < 4; k++) {
				const u32 pos = 4 * w + k;
				if (pos < A.mlen) {
					x |= (u32)mp[A.mlen - 1 - pos] << (8 * k);
				}
			}
		}
		const u64 t = (u64)acc[w] + x + carry;
		acc[w] = (u32)t;
		carry = t >> 32;
	}
	u8 *o = A.out + (size_t)i * A.outlen;
	for (u32 k = 0; k < A.outlen; k++) {
		o[A.outlen - 1 - k] = (u8)(acc[k >> 2] >> (8 * (k & 3)));
	}
	A.bad[i] = (nz != 0 && decided && lt) ? 0 : 1;
}

__global__ __launch_bounds__(64) void k_status_or(u8 *status, const u8 *bad, u8 *out, u32 stride, u32 n)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i < n && bad[i]) {
		status[i] = 1;
		for (u32 k = 0; k < stride; k++) {
			out[(size_t)i * stride + k] = 0;
		}
	}
}

// status[i] = 1 wherever sub[i] != want (a public key whose [q]Y is not the point at infinity is an import error)
__global__ __launch_bounds__(256) void k_status_require(u8 *status, const u8 *sub, u8 want, u32 n)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i < n && sub[i] != want) {
		status[i] = 1;
	}
}
hipError_t ecamd_launch_status_require(uint8_t *status, const uint8_t *sub, uint8_t want, uint32_t n, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_status_require, dim3((n + 255) / 256), dim3(256), 0, s, status, sub, want, n);
	return hipGetLastError();
}

hipError_t ecamd_launch_status_or(uint8_t *status, const uint8_t *bad, uint8_t *out, uint32_t out_stride, uint32_t n, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_status_or, dim3((n + 63) / 64), dim3(64), 0, s, status, bad, out, out_stride, n);
	return hipGetLastError();
}

hipError_t ecamd_launch_blind_scalar(const EcamdBlindArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if ((a.outlen + 3) / 4 > BLIND_MAXW) {
		return hipErrorInvalidValue;
	}
	hipLaunchKernelGGL(k_blind_scalar, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

// nn_get_random_mod given its random bytes (ecamd_randmod.h): out = LE(raw) mod (q - 1) + 1, qlen big-endian bytes per item
template <int NW> __global__ __launch_bounds__(64) void k_rand_mod(EcamdRandModArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u32 q[NW], w[NW];
#pragma unroll
	for (int k = 0; k < NW; k++) {
		q[k] = A.q[k];
	}
	randmod_words<NW>(w, A.raw + (size_t)i * A.rawlen, (int)A.rawlen, q);
	u8 *dst = A.out + (size_t)i * A.qlen;
#pragma unroll
	for (int k = 0; k < NW; k++) {
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const u32 pos = 4 * (u32)k + (u32)b;
			if (pos < A.qlen) {
				dst[A.qlen - 1 - pos] = (u8)(w[k] >> (8 * b));
			}
		}
	}
}

hipError_t ecamd_launch_rand_mod(int qnw, const EcamdRandModArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (qnw) {
#define X(N) case N: hipLaunchKernelGGL(k_rand_mod<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ne = q - (digest mod q) mod q (EcamdSchnorrNeArgs): the digest is a big-endian integer of hlen octets (nn_init_from_buf), reduced by a
// restoring binary division as in ecamd_randmod.h -- 8 hlen steps of one shift and one conditional subtraction over NW words
template <int NW> __global__ __launch_bounds__(64) void k_schnorr_ne(EcamdSchnorrNeArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u32 q[NW], r[NW];
#pragma unroll
	for (int k = 0; k < NW; k++) {
		q[k] = A.q[k];
		r[k] = 0;
	}
	const u8 *dg = A.dig + (size_t)i * A.hlen;
#pragma unroll 1
	for (u32 byte = 0; byte < A.hlen; byte++) {
		const u32 v = dg[byte];
#pragma unroll 1
		for (int bit = 7; bit >= 0; bit--) {
			const u32 top = r[NW - 1] >> 31;
#pragma unroll
			for (int w = NW - 1; w > 0; w--) {
				r[w] = (r[w] << 1) | (r[w - 1] >> 31);
			}
			r[0] = (r[0] << 1) | ((v >> bit) & 1u);
			u32 d[NW], borrow = 0;
#pragma unroll
			for (int w = 0; w < NW; w++) {
				const uint64_t x = (uint64_t)r[w] - q[w] - borrow;
				d[w] = (u32)x;
				borrow = (u32)(x >> 63);
			}
			const bool ge = (top != 0) | (borrow == 0);
#pragma unroll
			for (int w = 0; w < NW; w++) {
				r[w] = ge ? d[w] : r[w];
			}
		}
	}
	// q - e, and 0 when e = 0
	u32 ne[NW], borrow = 0, nz = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) {
		const uint64_t x = (uint64_t)q[w] - r[w] - borrow;
		ne[w] = (u32)x;
		borrow = (u32)(x >> 63);
		nz |= r[w];
	}
	u8 *dst = A.ne + (size_t)i * A.qlen;
#pragma unroll
	for (int k = 0; k < NW; k++) {
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const u32 pos = 4 * (u32)k + (u32)b;
			if (pos < A.qlen) {
				dst[A.qlen - 1 - pos] = nz ? (u8)(ne[k] >> (8 * b)) : (u8)0;
			}
		}
	}
}

hipError_t ecamd_launch_schnorr_ne(int qnw, const EcamdSchnorrNeArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (qnw) {
#define X(N) case N: hipLaunchKernelGGL(k_schnorr_ne<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ---- the counting sort in front of the bucket evaluation of the Schnorr-type combination (EcamdBktSortArgs; round 6).  Pairs are
//      enumerated window-major: window win has the n keys (digit of scW) and, below nwinZ, the n signature points (digit of scZ).
//      Digit 0 contributes nothing and is not filed. ----
static __device__ __forceinline__ u32 bkt_digit(const u8 *sc, u32 len, u32 win, u32 c)
{
	// bits [win c, win c + c) of the big-endian integer sc[0 .. len)
	const u32 bit = win * c, byte = bit >> 3, sh = bit & 7u;
	u32 v = 0;
#pragma unroll
	for (u32 k = 0; k < 3; k++) {
		const u32 pos = byte + k;
		if (pos < len) {
			v |= (u32)sc[len - 1u - pos] << (8u * k);
		}
	}
	return (v >> sh) & ((1u << c) - 1u);
}
static __device__ __forceinline__ u32 bkt_pair_digit(const EcamdBktSortArgs &A, u32 win, u32 j, u32 &pt)
{
	const bool isR = j >= A.n;
	pt = j;
	return isR ? bkt_digit(A.scZ + (size_t)(j - A.n) * A.zlen, A.zlen, win, A.c) : bkt_digit(A.scW + (size_t)j * A.wlen, A.wlen, win, A.c);
}
__global__ __launch_bounds__(256) void k_bkt_hist(EcamdBktSortArgs A)
{
	const u32 win = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
	const u32 cnt = (win < A.nwinZ) ? 2u * A.n : A.n;
	if (j >= cnt) {
		return;
	}
	u32 pt;
	const u32 d = bkt_pair_digit(A, win, j, pt);
	if (d) {
		atomicAdd(&A.hist[((size_t)win << A.c) + d], 1u);
	}
}
// exclusive scan of the 2^c counters of one window by one block of 256 threads
__global__ __launch_bounds__(256) void k_bkt_scan(EcamdBktSortArgs A)
{
	__shared__ u32 part[256];
	const u32 win = blockIdx.x, t = threadIdx.x, per = (1u << A.c) / 256u;
	const u32 *h = A.hist + ((size_t)win << A.c) + (size_t)t * per;
	u32 sum = 0;
	for (u32 k = 0; k < per; k++) {
		sum += h[k];
	}
	part[t] = sum;
	__syncthreads();
	for (u32 d = 1; d < 256; d <<= 1) {
		const u32 v = (t >= d) ? part[t - d] : 0u;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	u32 run = part[t] - sum;
	u32 *o = A.start + ((size_t)win << A.c) + (size_t)t * per;
	for (u32 k = 0; k < per; k++) {
		o[k] = run;
		run += h[k];
	}
}
__global__ __launch_bounds__(256) void k_bkt_scatter(EcamdBktSortArgs A)
{
	const u32 win = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
	const u32 cnt = (win < A.nwinZ) ? 2u * A.n : A.n;
	if (j >= cnt) {
		return;
	}
	u32 pt;
	const u32 d = bkt_pair_digit(A, win, j, pt);
	if (d) {
		const size_t b = ((size_t)win << A.c) + d;
		const u32 pos = A.start[b] + atomicAdd(&A.cursor[b], 1u);
		A.order[(size_t)win * 2u * A.n + pos] = pt;
	}
}
// Fixed-capacity filing (the default): the digits of z_i (q - e_i) and of z_i are uniform whatever the batch holds -- z_i is 128 bits of a
// keyed stream --, so bucket sizes are Poisson with mean lambda = pairs / 2^c, and a bucket of `cap` = 32 + 2 lambda slots overflows with
// a probability below 10^-18 per batch: point indices go straight to slot[bucket][cursor++], without the histogram and the scan of a
// counting sort (a millisecond per 2^20 items).  An overflow is reported (*flag |= 16: "not decided here"), never dropped silently.
__global__ __launch_bounds__(256) void k_bkt_file(EcamdBktSortArgs A)
{
	const u32 win = A.win_first + blockIdx.y;
	u32 j = blockIdx.x * 256 + threadIdx.x;
	if (A.part == 1u) {   // the keys, then the commitments, of the items [item_first, item_first + item_count)
		if (j >= ((win < A.nwinZ) ? 2u * A.item_count : A.item_count)) {
			return;
		}
		j = j < A.item_count ? A.item_first + j : A.n + A.item_first + (j - A.item_count);
	} else if (j >= ((win < A.nwinZ) ? 2u * A.n : A.n)) {
		return;
	}
	u32 pt;
	const u32 d = bkt_pair_digit(A, win, j, pt);
	if (d) {
		const u32 b = (win << A.c) + d;
		const u32 pos = atomicAdd(&A.hist[b], 1u);
		u32 capb;
		const size_t base = ecamd_bkt_slot(b, A.cap, A.cap_top, A.top_win, &capb);
		if (pos < capb) {
			A.order[base + pos] = pt;
		} else {
			atomicOr(A.flag, 16u);
		}
	}
}

// Buckets of one size side by side: a lane of k_bkt_accum_g walks its bucket's list, so a wave takes as long as its LONGEST bucket
// (sizes are Poisson: 46 against a mean of 32 over 64 lanes).  Every block ranks 4096 consecutive buckets by size (a counting sort in
// LDS: sizes capped at 255) and writes the permutation; lane t of the accumulation then serves bucket perm[t], and the 64 lanes of a
// wave get buckets of (nearly) one size.
// split_first (the Ed25519 form: 15 << 16; else 0xffffffff): the window whose buckets are shared by ECAMD_EDB_SPLIT lanes each -- a lane's size
// there is its share of the bucket it helps with
__global__ __launch_bounds__(256) void k_bkt_rank(const u32 *count, u32 *perm, u32 total, u32 first_block, u32 split_first)
{
	__shared__ u32 hist[256], base[256];
	const u32 t = threadIdx.x, first = (first_block + blockIdx.x) * 4096u;
	hist[t] = 0;
	__syncthreads();
	u32 key[16];
#pragma unroll
	for (u32 k = 0; k < 16; k++) {
		const u32 b = first + k * 256u + t;
		u32 cnt = (b < total) ? count[b] : 0u;
		if (b < total && (b & 0xffff0000u) == split_first) {
			const u32 d = b & 0xffffu, dp = d & (ECAMD_EDB_SPLIT_DIGITS - 1u), part = d / ECAMD_EDB_SPLIT_DIGITS;
			const u32 full = dp ? count[split_first | dp] : 0u;
			cnt = (full + (ECAMD_EDB_SPLIT - 1u) - part) >> ECAMD_EDB_SPLIT_LOG2;
		}
		key[k] = 255u - (cnt > 255u ? 255u : cnt);      // the longest first
		atomicAdd(&hist[key[k]], 1u);
	}
	__syncthreads();
	if (t == 0) {
		u32 run = 0;
		for (u32 k = 0; k < 256; k++) {
			base[k] = run;
			run += hist[k];
		}
	}
	__syncthreads();
#pragma unroll
	for (u32 k = 0; k < 16; k++) {
		const u32 b = first + k * 256u + t;
		const u32 pos = atomicAdd(&base[key[k]], 1u);
		if (first + pos < total) {
			perm[first + pos] = b < total ? b : total - 1u;
		}
	}
}

hipError_t ecamd_launch_bkt_sort(const EcamdBktSortArgs &a, hipStream_t s)
{
	if (a.n == 0 || a.c < 8 || a.c > 16 || a.nwin == 0) {
		return hipErrorInvalidValue;
	}
	const size_t counters = (size_t)a.nwin << a.c;
	if (a.part) {
		// the streamed form: a range of items into every window (1) / the ranking when everything is filed (2); the caller cleared the counters
		if (!a.cap || a.part > 2u || (a.part == 1u && (a.item_count == 0 || (uint64_t)a.item_first + a.item_count > a.n))) {
			return hipErrorInvalidValue;
		}
		if (a.part == 1u) {
			hipLaunchKernelGGL(k_bkt_file, dim3((2 * a.item_count + 255) / 256, a.nwin), dim3(256), 0, s, a);
		} else if (a.perm) {
			hipLaunchKernelGGL(k_bkt_rank, dim3((unsigned)((counters + 4095) / 4096)), dim3(256), 0, s, (const u32 *)a.hist, a.perm, (u32)counters, 0u, 0xffffffffu);
		}
		return hipGetLastError();
	}
	if (a.win_count) {
		// a range of windows of the fixed-capacity filing (whole groups of 4096 buckets: c >= 12); the caller cleared the counters
		if (!a.cap || a.c < 12 || a.win_first + a.win_count > a.nwin) {
			return hipErrorInvalidValue;
		}
		const uint32_t cnt = (a.win_first < a.nwinZ) ? 2 * a.n : a.n;   // (the windows above z_i's hold keys only)
		hipLaunchKernelGGL(k_bkt_file, dim3((cnt + 255) / 256, a.win_count), dim3(256), 0, s, a);
		if (a.perm) {
			const uint32_t per_win = (1u << a.c) / 4096u;
			hipLaunchKernelGGL(k_bkt_rank, dim3(a.win_count * per_win), dim3(256), 0, s, (const u32 *)a.hist, a.perm, (u32)counters, a.win_first * per_win, 0xffffffffu);
		}
		return hipGetLastError();
	}
	hipError_t e = hipMemsetAsync(a.hist, 0, counters * 4, s);
	if (e == hipSuccess && !a.cap) {
		e = hipMemsetAsync(a.cursor, 0, counters * 4, s);
	}
	if (e != hipSuccess) {
		return e;
	}
	const dim3 gp((2 * a.n + 255) / 256, a.nwin);
	if (a.cap) {
		hipLaunchKernelGGL(k_bkt_file, gp, dim3(256), 0, s, a);
	} else {
		hipLaunchKernelGGL(k_bkt_hist, gp, dim3(256), 0, s, a);
		hipLaunchKernelGGL(k_bkt_scan, dim3(a.nwin), dim3(256), 0, s, a);
		hipLaunchKernelGGL(k_bkt_scatter, gp, dim3(256), 0, s, a);
	}
	if (a.perm) {
		hipLaunchKernelGGL(k_bkt_rank, dim3((unsigned)((counters + 4095) / 4096)), dim3(256), 0, s, (const u32 *)a.hist, a.perm, (u32)counters, 0u, 0xffffffffu);
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_msm_scal(int nw, const EcamdMsmScalArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid(((a.count ? a.count : a.n) + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_msm_scal<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_msm_vsum(int nw, const EcamdMsmVsumArgs &a, hipStream_t s)
{
	if (a.count == 0) {
		return hipSuccess;
	}
	const uint32_t threads = (a.count + 63) / 64;
	const dim3 grid((threads + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_msm_vsum<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_y_from_x(int nw, const EcamdYfromXArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_y_from_x<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_fp(int nw, const EcamdFpArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_fp<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Round 4: the group law and the public-scalar multiplication in either wire format.
//
// k_ptf<NW>: prj_pt_add (curves/prj_pt.c:1204) / prj_pt_dbl (:1132) / prj_pt_is_on_curve (:144) on affine X || Y or projective
// X || Y || Z inputs (Z = 0: the point at infinity; (0 : 0 : 0) satisfies the curve equation as in the reference), the complete
// formulas of ecamd_point.h (RCB Alg. 1 / 3, the reference's), an "exceptional pair" of the addition (Y3 = Z3 = 0, :1058-1060)
// reported as an error.  The reference's prj_pt_add / prj_pt_dbl do not test their inputs; here a point that is not on the curve
// is an error (status 1).  Output: the unique representative (X / Z, Y / Z [, 1]) or status 2 (infinity, zero bytes).
//
// k_unprot<NW>: _prj_pt_unprotected_mult (curves/prj_pt.c:1835-1880) STATEMENT FOR STATEMENT -- on-curve test of the input, zero
// scalar -> infinity, out = in, then per bit below the top one: out = 2 out, and out = out + in when the bit is set, with the
// addition's exceptional pair as the reference's -1 (on a curve of even order (2m - 1) in can be the point of order two), the
// on-curve test of the result -- because the batch form must return what the scalar function returns for EVERY public scalar and
// point, which the window kernels (same group element, no such failure) do not promise.  Scalars: per item, or one for all
// (sstride = 0: check_prj_pt_order).  Lanes run both branches of "if bit" and select: control flow stays uniform.
// ------------------------------------------------------------------------------------------
template <int NW> static __device__ __forceinline__ bool ptf_load(Pt<NW> &P, const u8 *src, int clen, int fmt, int slot)
{
	const CurveK<NW> &K = ConstTab<NW>::get(slot);
	const Fe<NW> X = fe_load_be<NW>(src, clen), Y = fe_load_be<NW>(src + clen, clen);
	bool ok = fe_lt_p<NW>(X, slot) & fe_lt_p<NW>(Y, slot);
	P.X = fe_to_mont<NW>(X, slot);
	P.Y = fe_to_mont<NW>(Y, slot);
	if (fmt == 0) {
		P.Z = fe_const<NW>(K.one);
	} else {
		const Fe<NW> Z = fe_load_be<NW>(src + 2 * clen, clen);
		ok = ok & fe_lt_p<NW>(Z, slot);
		P.Z = fe_to_mont<NW>(Z, slot);
	}
	// Y^2 Z = X^3 + a X Z^2 + b Z^3 (prj_pt_is_on_curve)
	const Fe<NW> z2 = fe_mul<NW>(P.Z, P.Z, slot);
	const Fe<NW> x2 = fe_mul<NW>(P.X, P.X, slot);
	Fe<NW> rhs = fe_add<NW>(x2, fe_mul<NW>(fe_const<NW>(K.a), z2, slot), slot);
	rhs = fe_mul<NW>(rhs, P.X, slot);
	rhs = fe_add<NW>(rhs, fe_mul<NW>(fe_mul<NW>(fe_const<NW>(K.b), z2, slot), P.Z, slot), slot);
	const Fe<NW> lhs = fe_mul<NW>(fe_mul<NW>(P.Y, P.Y, slot), P.Z, slot);
	return ok & fe_eq<NW>(lhs, rhs);
}
// the unique representative of R (or zero bytes) and its status
template <int NW> static __device__ __forceinline__ void ptf_store(u8 *out, u8 *status, const Pt<NW> &R, bool err, int clen, int fmt, int slot)
{
	const int ow = (fmt ? 3 : 2) * clen;
	if (err || fe_is_zero<NW>(R.Z)) {
		for (int b = 0; b < ow; b++) {
			out[b] = 0;
		}
		*status = err ? 1 : 2;
		return;
	}
	const Fe<NW> zi = fe_inv<NW>(R.Z, slot);
	fe_store_be<NW>(out, clen, fe_from_mont<NW>(fe_mul<NW>(R.X, zi, slot), slot));
	fe_store_be<NW>(out + clen, clen, fe_from_mont<NW>(fe_mul<NW>(R.Y, zi, slot), slot));
	if (fmt) {
		for (int b = 0; b < clen; b++) {
			out[2 * clen + b] = (u8)(b == clen - 1 ? 1 : 0);
		}
	}
	*status = 0;
}

template <int NW> __global__ __launch_bounds__(64) void k_ptf(EcamdPtfArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot, clen = (int)A.clen;
	const int iw = (A.in_fmt ? 3 : 2) * clen, ow = (A.out_fmt ? 3 : 2) * clen;
	Pt<NW> P, Q;
	bool ok = ptf_load<NW>(P, A.p1 + (size_t)i * iw, clen, A.in_fmt, slot);
	if (A.op == 2) {
		A.status[i] = ok ? 0 : 1;
		return;
	}
	if (A.op == 0 || A.op >= 4) {
		ok = ok & ptf_load<NW>(Q, A.p2 + (size_t)i * iw, clen, A.in_fmt, slot);
	}
	if (A.op >= 4) {
		// prj_pt_cmp (curves/prj_pt.c:303): X1 Z2 against X2 Z1 and Y1 Z2 against Y2 Z1, no special case for Z = 0 -- out byte 0
		// where the reference's *cmp is 0, 1 where it is not (its sign is that of the Montgomery residues in the reference's
		// word size and is not reproduced); prj_pt_eq_or_opp (:412): the X test and Y1 Z2 = +-Y2 Z1 -- out byte = *eq_or_opp
		const bool xe = fe_eq<NW>(fe_mul<NW>(P.X, Q.Z, slot), fe_mul<NW>(Q.X, P.Z, slot));
		const Fe<NW> y1 = fe_mul<NW>(P.Y, Q.Z, slot), y2 = fe_mul<NW>(Q.Y, P.Z, slot);
		const bool ye = fe_eq<NW>(y1, y2), yo = fe_is_zero<NW>(fe_add<NW>(y1, y2, slot));
		A.out[i] = !ok ? 0 : (A.op == 4 ? (u8)((xe & ye) ? 0 : 1) : (u8)((xe & (ye | yo)) ? 1 : 0));
		A.status[i] = ok ? 0 : 1;
		return;
	}
	Pt<NW> R = P;
	bool err = !ok;
	if (ok && A.op == 3) {
		R.Y = fe_sub<NW>(fe_zero<NW>(), P.Y, slot);   // prj_pt_neg (:435): (X : -Y : Z)
	} else if (ok) {
		R = A.op == 1 ? pt_dbl<NW>(P, slot) : pt_add<NW>(P, Q, slot);
		err = (A.op == 0) && fe_is_zero<NW>(R.Z) && fe_is_zero<NW>(R.Y);   // the addition's exceptional pair
	}
	ptf_store<NW>(A.out + (size_t)i * ow, A.status + i, R, err, clen, A.out_fmt, slot);
}

template <int NW> __global__ __launch_bounds__(64) void k_unprot(EcamdUnprotArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const int slot = A.slot, clen = (int)A.clen, slen = (int)A.slen;
	const int iw = (A.in_fmt ? 3 : 2) * clen, ow = (A.out_fmt ? 3 : 2) * clen;
	Pt<NW> P;
	bool err = !ptf_load<NW>(P, A.points + (size_t)i * iw, clen, A.in_fmt, slot);
	const u8 *sc = A.scalars + (size_t)i * A.sstride;
	// bit length of the scalar (big-endian octets)
	int bits = 0;
	for (int b = 0; b < slen; b++) {
		const u32 v = sc[b];
		if (bits == 0 && v != 0) {
			bits = 8 * (slen - b) - (__clz(v) - 24);
		}
	}
	Pt<NW> R = P;
	if (!err && bits == 0) {
		// "Multiplication by zero is the point at infinity"
		R.Z = fe_zero<NW>();
	}
	for (int t = bits - 2; t >= 0 && !err; t--) {
		const u32 bit = (sc[slen - 1 - (t >> 3)] >> (t & 7)) & 1u;
		R = pt_dbl<NW>(R, slot);
		const Pt<NW> S = pt_add<NW>(R, P, slot);
		if (bit) {
			err = fe_is_zero<NW>(S.Z) && fe_is_zero<NW>(S.Y);
			R = S;
		}
	}
	// (the reference's on-curve test of the result cannot fail for an input on the curve: the formulas are polynomial identities
	// there, and (0 : 0 : 0) -- only reachable through the exceptional pair, an error above -- satisfies the equation anyway)
	ptf_store<NW>(A.out + (size_t)i * ow, A.status + i, R, err, clen, A.out_fmt, slot);
}

hipError_t ecamd_launch_ptf(int nw, const EcamdPtfArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_ptf<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_unprot(int nw, const EcamdUnprotArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_unprot<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_pt(int nw, const EcamdPtArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_pt<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_decode(int nw, const EcamdEdDecodeArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (nw != 8) {
		return hipErrorInvalidValue;
	}
	hipLaunchKernelGGL(k_ed_decode<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_scal(int nw, const EcamdEdScalArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (nw != 8) {
		return hipErrorInvalidValue;
	}
	hipLaunchKernelGGL(k_ed_scal<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_lat(const EcamdEdLatArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_lat<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

// the filing of the Ed25519 bucket evaluation (EcamdEdBktArgs): one thread per (window, point); see k_bkt_file
__global__ __launch_bounds__(256) void k_edbkt_file(EcamdEdBktArgs B)
{
	const u32 win = blockIdx.y, rel = blockIdx.x * 256 + threadIdx.x;
	u32 j = rel;
	if (B.part == 1u) {          // the keys, then the commitments, of the items [first, first + count)
		if (rel >= 2u * B.count_items) {
			return;
		}
		j = rel < B.count_items ? B.first + rel : B.n + B.LB + B.first + (rel - B.count_items);
	} else if (B.part == 2u) {   // the copies of B
		if (rel >= B.LB) {
			return;
		}
		j = B.n + rel;
	} else if (rel >= 2u * B.n + B.LB) {
		return;
	}
	u32 d;
	if (j < B.n) {
		d = (B.rawC[(size_t)j * 8 + (win >> 1)] >> (16u * (win & 1u))) & 0xffffu;
	} else if (j < B.n + B.LB) {
		d = (B.rawB[(size_t)(j - B.n) * 8 + (win >> 1)] >> (16u * (win & 1u))) & 0xffffu;
	} else {
		if (win >= 8u) {
			return;
		}
		d = (B.rawZ[(size_t)(j - B.n - B.LB) * 4 + (win >> 1)] >> (16u * (win & 1u))) & 0xffffu;
	}
	if (d) {
		const u32 b = (win << 16) + d;
		const u32 pos = atomicAdd(&B.count[b], 1u);
		u32 capb;
		const size_t base = ecamd_bkt_slot(b, B.cap, B.cap_top, 15u, &capb);
		if (pos < capb) {
			B.order[base + pos] = j;
		} else {
			atomicOr(B.flagword, 16u);
		}
	}
}
hipError_t ecamd_launch_edbkt_file(const EcamdEdBktArgs &b, hipStream_t s)
{
	const size_t counters = (size_t)16 << 16;
	if (b.part == 1u) {   // a chunk of the streamed form: the caller cleared the counters when the batch began
		if (b.count_items) {
			hipLaunchKernelGGL(k_edbkt_file, dim3((2 * b.count_items + 255) / 256, 16), dim3(256), 0, s, b);
		}
		return hipGetLastError();
	}
	if (b.part == 0u) {
		const hipError_t e = hipMemsetAsync(b.count, 0, counters * 4, s);
		if (e != hipSuccess) {
			return e;
		}
	}
	const uint32_t threads = b.part == 2u ? b.LB : 2 * b.n + b.LB;
	hipLaunchKernelGGL(k_edbkt_file, dim3((threads + 255) / 256, 16), dim3(256), 0, s, b);
	hipLaunchKernelGGL(k_bkt_rank, dim3((unsigned)(counters / 4096)), dim3(256), 0, s, (const u32 *)b.count, b.perm, (u32)counters, 0u, 15u << 16);
	return hipGetLastError();
}

hipError_t ecamd_launch_edmsm_scal(const EcamdEdMsmScalArgs &a, hipStream_t s)
{
	hipLaunchKernelGGL(k_edmsm_scal<8>, dim3(((a.count ? a.count : a.n) + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}
hipError_t ecamd_launch_edmsm_lane(const EcamdEdMsmLaneArgs &a, hipStream_t s)
{
	hipLaunchKernelGGL(k_edmsm_lane<8>, dim3((a.L + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}
hipError_t ecamd_launch_ed_fin(int nw, const EcamdEdFinArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (nw == 8) {
		hipLaunchKernelGGL(k_ed_fin<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	} else if (nw == 14) {
		hipLaunchKernelGGL(k_ed_fin<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	} else {
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_sign_r(const EcamdEdSignArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (a.is448) {
		hipLaunchKernelGGL(k_ed448_sign_r<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_ed_sign_r<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_sign_enc(const EcamdEdSignArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (a.is448) {
		hipLaunchKernelGGL(k_ed448_sign_enc<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_ed_sign_enc<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_sign_S(const EcamdEdSignArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (a.is448) {
		hipLaunchKernelGGL(k_ed448_sign_S<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_ed_sign_S<8>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed448_decode(const EcamdEd448DecodeArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed448_decode<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed448_scal(const EcamdEdScalArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed448_scal<14>, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_msm_gate(int nw, const uint8_t *keys_aff, const uint8_t *flagsA, const uint8_t *flagsR, const uint8_t *flagsS, uint32_t n,
				    uint32_t clen, uint32_t cof_dbl, int slot, uint32_t *gate, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	if (nw != 14) {
		return hipErrorInvalidValue;   // Ed448 on the WEI448 handle (Ed25519 has its own combination: k_edmsm_*, k_edbkt_*)
	}
	hipLaunchKernelGGL(k_ed_msm_gate<14>, dim3((n + 63) / 64), dim3(64), 0, s, keys_aff, flagsA, flagsR, flagsS, n, clen, cof_dbl, slot, gate);
	return hipGetLastError();
}

hipError_t ecamd_launch_verdict_or(uint8_t *verdict, const uint8_t *piece, const uint32_t *gate, hipStream_t s)
{
	hipLaunchKernelGGL(k_verdict_or, dim3(1), dim3(64), 0, s, verdict, piece, gate);
	return hipGetLastError();
}

hipError_t ecamd_launch_prj_import(int nw, const EcamdPrjInArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	switch (nw) {
#define X(N) case N: hipLaunchKernelGGL(k_prj_import<N>, grid, block, 0, s, a); break;
		ECAMD_FOR_NW(X)
#undef X
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_prj_export(const EcamdPrjOutArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_prj_export, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
	return hipGetLastError();
}
