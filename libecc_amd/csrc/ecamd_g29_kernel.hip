// libecc_amd/csrc/ecamd_g29_kernel.hip -- radix-2^29 Jacobian fast path of the batched prj_pt_mul for
// EVERY curve size (one instantiation per |p|; curves of equal size share the code, their
// constants live in __constant__ memory).  Same pipeline as ecamd_p256_kernel.hip:
//   k_smul_g<PB>      import + on-curve check, table [1..8]P, signed window w = 4, Jacobian result
//   k_finalize_g<PB>  Jacobian -> affine, one inversion per 8 items (Montgomery's trick)
// Lanes that meet an exceptional pair of the incomplete addition (including inputs of small order
// on cofactor curves, or a zero Z at the end) are marked ECAMD_STATUS_REDO and recomputed by the
// complete-formula kernel k_smul<NW>, so the observable result is the reference's for every input.
// Replaces prj_pt_import_from_aff_buf -> prj_pt_mul -> prj_pt_unique -> prj_pt_export_to_aff_buf
// (curves/prj_pt.c:511,1759,241,600 in /root/reference/src).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ecamd_jacg.h"
#if defined(G29_P25519) || defined(G29_P448)
#include "ecamd_rcbg.h"
#endif
#include "ecamd_internal.h"

using namespace jacg;
typedef uint8_t u8;

// This file is compiled once per field size (-DG29_PB=<bits>: the kernels of that size plus its own
// copy of the constant table) and once more with -DG29_DISPATCH (the size -> launcher switch), so
// that the instantiations build in parallel (libecc_amd/build.py).
#define G29_SLOTS 8
template <int NL> struct SlotsG { CurveG<NL> s[G29_SLOTS]; };

template <int PB> struct TabGP;
template <int PB> struct Lay {
	static constexpr int NL = Cfg<PB>::NL;
	static constexpr int NW = (PB + 31) / 32;          // saturated words of a coordinate
	static constexpr int ENTW = ((3 * NL + 3) / 4) * 4;  // words per table entry (16-byte multiple)
	static constexpr int KW = (PB + 31) / 32 + 1;      // words of the recoded scalar of the comb kernel (slen <= 4 KW - 4)
	// k_smul_g keeps its recoded scalar in the item's scratch, behind the table: scalars of up to 8 NW + 4 bytes
	// (a blinded scalar m + b #E, curves/prj_pt.c:1782-1822, has about 2 |#E| bits)
	static constexpr int KRECW = ((2 * NW + 2 + 3) / 4) * 4;
	static constexpr int ITEMW = 8 * ENTW + KRECW;     // scratch words per item
};

// len bytes big-endian -> NW little-endian words
template <int NW> static __device__ __forceinline__ void load_be(const u8 *src, int len, u32 *w)
{
#pragma unroll
	for (int i = 0; i < NW; i++) {
		u32 x = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * i + b;
			if (pos < len) {
				x |= (u32)src[len - 1 - pos] << (8 * b);
			}
		}
		w[i] = x;
	}
}
template <int NW> static __device__ __forceinline__ void store_be(u8 *dst, int len, const u32 *w)
{
#pragma unroll
	for (int i = 0; i < NW; i++) {
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * i + b;
			if (pos < len) {
				dst[len - 1 - pos] = (u8)(w[i] >> (8 * b));
			}
		}
	}
}

// QS: words between consecutive 16-byte quads of an entry -- 4 for an item-major record (base = the item's record), 256 for the
// wave-blocked staging of the affine-table pipeline (stg_ent below: quad q of entry e of the 64 items of a wave sits in one 1 KB run)
template <int PB, int QS = 4> static __device__ __forceinline__ void tab_store(u32 *base, int e, const TabEnt<PB> &T)
{
	constexpr int NL = Lay<PB>::NL, ENTW = Lay<PB>::ENTW;
	u32 buf[ENTW];
#pragma unroll
	for (int i = 0; i < NL; i++) {
		buf[i] = T.X.l[i];
		buf[NL + i] = T.Y.l[i];
		buf[2 * NL + i] = T.Z.l[i];
	}
#pragma unroll
	for (int i = 3 * NL; i < ENTW; i++) {
		buf[i] = 0;
	}
	u32 *d = base + (size_t)e * (ENTW / 4) * QS;
#pragma unroll
	for (int i = 0; i < ENTW / 4; i++) {
		*(uint4 *)(d + (size_t)i * QS) = make_uint4(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]);
	}
}
template <int PB, int QS = 4> static __device__ __forceinline__ TabEnt<PB> tab_load(const u32 *base, u32 e)
{
	constexpr int NL = Lay<PB>::NL, ENTW = Lay<PB>::ENTW;
	u32 buf[ENTW];
	const u32 *s = base + (size_t)e * (ENTW / 4) * QS;
#pragma unroll
	for (int i = 0; i < ENTW / 4; i++) {
		const uint4 v = *(const uint4 *)(s + (size_t)i * QS);
		buf[4 * i] = v.x;
		buf[4 * i + 1] = v.y;
		buf[4 * i + 2] = v.z;
		buf[4 * i + 3] = v.w;
	}
	TabEnt<PB> T;
#pragma unroll
	for (int i = 0; i < NL; i++) {
		T.X.l[i] = buf[i];
		T.Y.l[i] = buf[NL + i];
		T.Z.l[i] = buf[2 * NL + i];
	}
	return T;
}
// the same entry without a digit-dependent address: every entry is read, the wanted one kept by masking
// (secret-scalar mode; the reference's posture is nn_tabselect, nn/nn.c:564)
template <int PB> static __device__ __forceinline__ TabEnt<PB> tab_load_masked(const u32 *base, u32 e)
{
	constexpr int NL = Lay<PB>::NL, ENTW = Lay<PB>::ENTW;
	u32 buf[ENTW];
#pragma unroll
	for (int i = 0; i < ENTW; i++) {
		buf[i] = 0;
	}
#pragma unroll 1
	for (u32 ee = 0; ee < 8; ee++) {
		const u32 m = (ee == e) ? 0xffffffffu : 0u;
		const uint4 *s = (const uint4 *)(base + (size_t)ee * ENTW);
#pragma unroll
		for (int i = 0; i < ENTW / 4; i++) {
			const uint4 v = s[i];
			buf[4 * i] |= v.x & m;
			buf[4 * i + 1] |= v.y & m;
			buf[4 * i + 2] |= v.z & m;
			buf[4 * i + 3] |= v.w & m;
		}
	}
	TabEnt<PB> T;
#pragma unroll
	for (int i = 0; i < NL; i++) {
		T.X.l[i] = buf[i];
		T.Y.l[i] = buf[NL + i];
		T.Z.l[i] = buf[2 * NL + i];
	}
	return T;
}
// Jacobian result record (all three coordinates in class FA) in the first table slot
template <int PB, int QS = 4> static __device__ __forceinline__ void jac_store(u32 *base, const Jac<PB> &P)
{
	TabEnt<PB> T;
	T.X = P.X;
	T.Z = P.Z;
#pragma unroll
	for (int i = 0; i < Lay<PB>::NL; i++) {
		T.Y.l[i] = P.Y.l[i];
	}
	tab_store<PB, QS>(base, 0, T);
}
template <int PB, int QS = 4> static __device__ __forceinline__ Jac<PB> jac_load(const u32 *base)
{
	const TabEnt<PB> T = tab_load<PB, QS>(base, 0);
	Jac<PB> P;
	P.X = T.X;
	P.Z = T.Z;
#pragma unroll
	for (int i = 0; i < Lay<PB>::NL; i++) {
		P.Y.l[i] = T.Y.l[i];
	}
	return P;
}

template <class T> static __device__ __forceinline__ T selg(bool c, const T &a, const T &b)
{
	T r;
#pragma unroll
	for (int i = 0; i < T::C::NL; i++) {
		r.l[i] = c ? a.l[i] : b.l[i];
	}
	return r;
}

// Staging of the affine-table pipeline (k_table_g / k_affine_g / k_loop_g / k_comb*_g / k_finalize_g): the Jacobian multiples, the loop's
// result and the prefix products of the shared inversions.  In those kernels every lane walks the entries in the same order, so the
// records of the 64 items of a wave are interleaved quad by quad (STG_QS = 256 words between the quads of an entry): each 16-byte
// access of a wave is one contiguous kilobyte instead of 64 lines ITEMW words apart (round 5; the secp256r1 kernels have had this
// since round 2).  The recoded scalars follow the entries in the same addressing.  The Jacobian-table kernel of the two nine-limb
// flavours (k_smul_g: per-lane digit-indexed look-ups) and the multi-scalar kernels keep item-major records (STG_QS = 4).
#if defined(G29_K256) || defined(G29_P25519) || defined(G29_JACTAB) || defined(G29_STG_ITEM_MAJOR)
#define STG_QS 4
#else
#define STG_QS 256
#endif
template <int PB> static __device__ __forceinline__ u32 *stg_ent(u32 *tbl, u32 i)
{
	if (STG_QS == 4) {
		return tbl + (size_t)i * Lay<PB>::ITEMW;
	}
	return tbl + (size_t)(i >> 6) * 64 * Lay<PB>::ITEMW + (size_t)(i & 63u) * 4;
}
// the recoded scalar follows the eight entries in the same quad addressing (a constant offset from the item's entry pointer:
// the window loop keeps ONE pointer live -- a second one cost k_loop_g<521> its second wave per SIMD); word w: stg_krw(kr, w)
template <int PB> static __device__ __forceinline__ u32 *stg_kr(u32 *ent)
{
	return ent + (size_t)8 * (Lay<PB>::ENTW / 4) * STG_QS;
}
static __device__ __forceinline__ u32 &stg_krw(u32 *kr, int w) { return kr[(size_t)(w >> 2) * STG_QS + (w & 3)]; }
static __device__ __forceinline__ u32 stg_krw(const u32 *kr, int w) { return kr[(size_t)(w >> 2) * STG_QS + (w & 3)]; }
// one field element in the X third of staging entry e (the prefix products of k_finalize_g)
template <int PB> static __device__ __forceinline__ void stg_fe_store(u32 *ent, int e, const typename Cls<PB>::FM &v)
{
	constexpr int NL = Lay<PB>::NL, ENTW = Lay<PB>::ENTW;
#pragma unroll
	for (int w = 0; w < NL; w++) {
		ent[((size_t)e * (ENTW / 4) + (size_t)(w >> 2)) * STG_QS + (w & 3)] = v.l[w];
	}
}
template <int PB> static __device__ __forceinline__ typename Cls<PB>::FM stg_fe_load(const u32 *ent, int e)
{
	constexpr int NL = Lay<PB>::NL, ENTW = Lay<PB>::ENTW;
	typename Cls<PB>::FM v;
#pragma unroll
	for (int w = 0; w < NL; w++) {
		v.l[w] = ent[((size_t)e * (ENTW / 4) + (size_t)(w >> 2)) * STG_QS + (w & 3)];
	}
	return v;
}

// import of item i (curves/prj_pt.c:511-552): coordinates < p, y != 0, on the curve; (xm, ym) in the field representation
// of this unit (Montgomery form, and on the isomorphic a = -3 curve when there is one), values < 2p with exact digits
template <int PB> static __device__ __forceinline__ bool import_point(const EcamdSmulArgs &A, u32 i, typename Cls<PB>::FM &xo,
								       typename Cls<PB>::FM &yo, const CurveG<Cfg<PB>::NL> &K)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, NW = L::NW;
	const int clen = (int)A.clen;
	const u8 *pin = A.points + (size_t)i * A.pstride;
	u32 xw[NW], yw[NW];
	load_be<NW>(pin, clen, xw);
	load_be<NW>(pin + clen, clen, yw);
	const auto xd = from_words<PB, NW>(xw), yd = from_words<PB, NW>(yw);
	bool ok;
	{
		u32 bx = 0, by = 0;
#pragma unroll
		for (int j = 0; j < NL; j++) {
			bx = (xd.l[j] - K.p[j] - bx) >> 31;
			by = (yd.l[j] - K.p[j] - by) >> 31;
		}
		ok = (bx != 0) & (by != 0);  // both strictly below p
		ok = ok & ((words_excess<PB, NW>(xw) | words_excess<PB, NW>(yw)) == 0);   // ... as octet strings, not only as the limbs' bits
		// y = 0 is a point of order 2: the reference's ladder fails on it for every scalar (see k_smul)
		u32 ynz = 0;
#pragma unroll
		for (int j = 0; j < NL; j++) {
			ynz |= yd.l[j];
		}
		ok = ok & (ynz != 0);
	}
	const FC onec = constant<FC>(K.one);
	const auto xm = mul(xd, constant<FC>(K.ix), K), ym = mul(yd, constant<FC>(K.iy), K);
	{
		// y^2 == (x^2 + a) x + b
		const auto t = mulc(carry(add(sqr(xm, K), constant<FC>(K.a))), xm, K);
		const auto rhs = add(t, constant<FC>(K.b));
		const auto dif = carry(sub_auto<1>(rhs, sqr(ym, K), K));
		ok = ok & is_zero_mulout(mulc(dif, onec, K), K);
	}
	xo = weaken<FM>(xm);
	yo = weaken<FM>(ym);
	return ok;
}

// k' = k + 0x88..8 over the 2*slen nibbles of item i's big-endian scalar, little-endian words into kr (any length the host
// lets through: slen <= 4 KRECW - 4); returns the carry out of the top nibble (the leading digit 0 / 1)
// (QS: word w of the recoded scalar sits at kr[(w >> 2) * QS + (w & 3)] -- 4: a plain array; STG_QS: the wave-blocked staging)
template <int QS = 4> static __device__ __forceinline__ u32 recode_scalar(const u8 *sc, int slen, u32 *kr)
{
	const int nwords = (slen + 3) >> 2;
	uint64_t c = 0;
	u32 last = 0;
#pragma unroll 1
	for (int w = 0; w < nwords; w++) {
		u32 x = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = 4 * w + b;
			if (pos < slen) {
				x |= (u32)sc[slen - 1 - pos] << (8 * b);
			}
		}
		const int nb = slen - 4 * w;
		const u32 add8 = (nb >= 4) ? 0x88888888u : (nb == 3 ? 0x00888888u : (nb == 2 ? 0x00008888u : 0x00000088u));
		c += (uint64_t)x + add8;
		last = (u32)c;
		kr[(size_t)(w >> 2) * QS + (w & 3)] = last;
		c >>= 32;
	}
	// the carry out of the top nibble sits just above the scalar's bytes
	const int bit = 8 * slen;
	return (bit & 31) ? ((last >> (bit & 31)) & 1u) : (u32)c;
}

// FLAV only makes the kernel symbols of the translation units of one size distinct (0 dense, 1 secp521r1, 2 2^255 - 19)
// G29_WAVES (build flag, A/B tests through tools/build_variant.py): pin the waves per SIMD of the window kernel
#ifdef G29_WAVES
#define G29_OCC __attribute__((amdgpu_waves_per_eu(G29_WAVES, G29_WAVES)))
#else
#define G29_OCC
#endif
template <int PB, int FLAV> __global__ __launch_bounds__(64) G29_OCC void k_smul_g(EcamdSmulArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const int clen = (int)A.clen;
	u8 *out = A.out + (size_t)i * 2 * clen;

	// ---- import (curves/prj_pt.c:511-552): coordinates < p, on the curve ----
	const FC onec = constant<FC>(K.one);
	FM xm, ym;
	const bool ok = import_point<PB>(A, i, xm, ym, K);
	if (!ok) {
		A.status[i] = 1;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}

	// ---- table [1..8]P ----
	u32 *tb = A.tbl + (size_t)i * L::ITEMW;
	Jac<PB> P1;
	P1.X = weaken<FA>(xm);
	P1.Y = weaken<FA>(ym);
	P1.Z = weaken<FA>(onec);
	bool hz, bad = false;
	TabEnt<PB> T1;
	T1.X = P1.X;
	T1.Y = weaken<FM>(ym);
	T1.Z = P1.Z;
	const FA y1 = weaken<FA>(T1.Y);
	tab_store<PB>(tb, 0, T1);
	{
		Jac<PB> Pa = dbl(P1, K);
		tab_store<PB>(tb, 1, to_tab(Pa, K));
		Jac<PB> Pb = add_jac(Pa, T1.X, y1, T1.Z, hz, K);
		bad |= hz;
		tab_store<PB>(tb, 2, to_tab(Pb, K));
		Pa = dbl(Pa, K);
		tab_store<PB>(tb, 3, to_tab(Pa, K));
		Pb = add_jac(Pa, T1.X, y1, T1.Z, hz, K);
		bad |= hz;
		tab_store<PB>(tb, 4, to_tab(Pb, K));
		{
			const TabEnt<PB> t3 = tab_load<PB>(tb, 2);
			Jac<PB> P3;
			P3.X = t3.X;
			P3.Y = weaken<FA>(t3.Y);
			P3.Z = t3.Z;
			Pb = dbl(P3, K);
		}
		tab_store<PB>(tb, 5, to_tab(Pb, K));
		Pb = add_jac(Pb, T1.X, y1, T1.Z, hz, K);
		bad |= hz;
		tab_store<PB>(tb, 6, to_tab(Pb, K));
		Pa = dbl(Pa, K);
		tab_store<PB>(tb, 7, to_tab(Pa, K));
	}

	// ---- scalar: recoded into the item's scratch behind the table ----
	const int slen = (int)A.slen;
	u32 *kr = tb + 8 * L::ENTW;
	const u32 carry_bit = recode_scalar(A.scalars + (size_t)i * A.sstride, slen, kr);

	// ---- signed fixed window, left to right ----
	Jac<PB> acc = P1;
	bool inf = (carry_bit == 0);
	const int nwin = 2 * slen;
	// the word of the recoded scalar that holds the current nibble; its successor is fetched one window ahead, so
	// that the digit -- and with it the address of the table entry -- never waits for memory
	u32 wcur = nwin ? kr[(nwin - 1) >> 3] : 0u, wnext = 0u;
#pragma unroll 1
	for (int t = 0; t < nwin; t++) {
		const int pos = nwin - 1 - t;
		const int dig = (int)((wcur >> (4 * (pos & 7))) & 15u) - 8;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		if ((pos & 7) == 0 && pos != 0) {
			wnext = kr[(pos >> 3) - 1];
		}
#ifdef G29_PRELOAD
		// secret-scalar mode (a wave-uniform kernel argument): constant-address scan of the eight entries
		const TabEnt<PB> T = A.masked ? tab_load_masked<PB>(tb, mag ? mag - 1 : 0) : tab_load<PB>(tb, mag ? mag - 1 : 0);
#endif
#pragma unroll 1
		for (int d = 0; d < 4; d++) {
			acc = dbl(acc, K);
		}
#ifndef G29_PRELOAD
		const TabEnt<PB> T = A.masked ? tab_load_masked<PB>(tb, mag ? mag - 1 : 0) : tab_load<PB>(tb, mag ? mag - 1 : 0);
#endif
		wcur = ((pos & 7) == 0) ? wnext : wcur;
		const FA ty = selg(dig < 0, neg<PB>(T.Y, K), weaken<FA>(T.Y));
		const Jac<PB> S = add_jac(acc, T.X, ty, T.Z, hz, K);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		acc.X = selg(keep, acc.X, selg(use_t, T.X, S.X));
		acc.Y = selg(keep, acc.Y, selg(use_t, ty, S.Y));
		acc.Z = selg(keep, acc.Z, selg(use_t, T.Z, S.Z));
		inf = inf & keep;
	}
	// a doubling can reach infinity silently on curves with points of even order: exact test of Z
	if (!inf) {
		bad = bad | is_zero_mulout(mulc(acc.Z, onec, K), K);
	}
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	jac_store<PB>(tb, acc);
	A.status[i] = ECAMD_STATUS_JAC;
}

// ------------------------------------------------------------------------------------------
// Schnorr-type whole-batch verification as one multi-scalar multiplication (EcamdMsmArgs in ecamd_internal.h; the equation of
// _bip0340_verify_batch_no_memory, sig/bip0340.c:905-1010, and of _ecfsdsa_verify_batch_no_memory, sig/ecfsdsa.c:1042-):
//   k_msm_table_g   the 2n points: import + on-curve check, Jacobian table [1..8]P, recoded scalar behind it (k_smul_g's first half)
//   k_msm_loop_g    lane l: the Straus loop over its K keys (full-length scalars) and K signature points (128-bit scalars, negated)
//   k_msm_sum_g     tree sum of the lanes' Jacobian sums, fan-in MSM_FAN
//   k_msm_final_g   T = sum + [c]G is the point at infinity  <=>  sum = -[c]G (or both are infinite): exact comparison
// A Jacobian table (12M + 4S additions) rather than the affine one: one code path for every flavour (secp256k1's has no mixed
// addition), and no inversion pass over 16 entries per signature.
// ------------------------------------------------------------------------------------------
#define MSM_FAN 16
template <int PB> struct MsmLay {
	static constexpr int RECW = Lay<PB>::ENTW + 4;   // X, Y, Z (class FA) as a table entry, then the "is infinity" word
};

// BIP0340's lift_x of item i's r (jacg::lift_x_even; host-tested in tests/test_u29g_host.py::test_lift_x_even)
template <int PB> static __device__ __forceinline__ bool msm_lift_x(const u8 *src, int clen, typename Cls<PB>::FM &xo, typename Cls<PB>::FM &yo,
								     const CurveG<Cfg<PB>::NL> &K)
{
	constexpr int NW = Lay<PB>::NW;
	u32 xw[NW];
	load_be<NW>(src, clen, xw);
	const auto xd = from_words<PB, NW>(xw);
	const bool fits = words_excess<PB, NW>(xw) == 0;   // r >= 2^(29 NL) is no abscissa (lift_x_even compares the limbs with p)
	return lift_x_even<PB>(xd, K, xo, yo) & fits;
}

template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_msm_table_g(EcamdMsmArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL;
	const u32 idx = blockIdx.x * 64 + threadIdx.x;
	if (idx >= 2 * A.n) {
		return;
	}
	const bool isR = idx >= A.n;
	const u32 i = isR ? idx - A.n : idx;
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	EcamdSmulArgs S;
	S.points = isR ? A.ptsR : A.ptsY;
	S.pstride = 2u * A.clen;
	S.clen = A.clen;
	const FC onec = constant<FC>(K.one);
	FM xm, ym;
	bool bad;
	if (isR && A.r_fmt == 1u) {
		bad = !msm_lift_x<PB>(A.ptsR + (size_t)i * A.clen, (int)A.clen, xm, ym, K);
	} else {
		bad = !import_point<PB>(S, i, xm, ym, K);
	}
	u32 *tb = A.tbl + (size_t)idx * L::ITEMW;
	Jac<PB> P1;
	P1.X = weaken<FA>(xm);
	P1.Y = weaken<FA>(ym);
	P1.Z = weaken<FA>(onec);
	bool hz;
	TabEnt<PB> T1;
	T1.X = P1.X;
	T1.Y = weaken<FM>(ym);
	T1.Z = P1.Z;
	const FA y1 = weaken<FA>(T1.Y);
	tab_store<PB>(tb, 0, T1);
	{
		// the rolled order of k_table_g: [2j + 1]P = [2j]P + P, [2j + 2]P = 2 [j + 1]P, the multiples re-read from the table as they were stored
		auto entry = [&](u32 e) {
			const TabEnt<PB> t = tab_load<PB>(tb, e);
			Jac<PB> P;
			P.X = t.X;
			P.Y = weaken<FA>(t.Y);
			P.Z = t.Z;
			return P;
		};
		Jac<PB> Pb = dbl(P1, K), Pc = Pb;
		tab_store<PB>(tb, 1, to_tab(Pb, K));
#pragma unroll 1
		for (int j = 1; j <= 3; j++) {
			Pb = add_jac(entry((u32)(2 * j - 1)), T1.X, y1, T1.Z, hz, K);
			bad |= hz;
			tab_store<PB>(tb, 2 * j, to_tab(Pb, K));
			Pc = dbl(entry((u32)j), K);
			tab_store<PB>(tb, 2 * j + 1, to_tab(Pc, K));
		}
		// a doubling reaches infinity silently (points of even order): exact test of the last Z's (8P and 7P)
		bad = bad | is_zero_mulout(mulc(mulc(Pc.Z, Pb.Z, K), onec, K), K);
	}
	u32 *kr = tb + 8 * L::ENTW;
	const int slen = (int)(isR ? A.zlen : A.wlen);
	const u8 *sc = isR ? A.scZ + (size_t)i * A.zlen : A.scW + (size_t)i * A.wlen;
	kr[L::KRECW - 1] = recode_scalar(sc, slen, kr);
	if (bad) {
		atomicOr(A.flagword, 1u);   // the point does not import, or a multiple below 9P is infinity: not decided here
	}
}

// one term of the Straus sum: acc += (+-) [dig]P from the item's table
template <int PB>
static __device__ __forceinline__ void msm_step(Jac<PB> &acc, bool &inf, bool &bad, const u32 *tb, int dig, bool negate, const CurveG<Cfg<PB>::NL> &K)
{
	typedef typename Cls<PB>::FA FA;
	const u32 mag = (u32)(dig < 0 ? -dig : dig);
	const TabEnt<PB> T = tab_load<PB>(tb, mag ? mag - 1 : 0);
	const FA ty = selg((dig < 0) != negate, neg<PB>(T.Y, K), weaken<FA>(T.Y));
	bool hz;
	const Jac<PB> S = add_jac(acc, T.X, ty, T.Z, hz, K);
	const bool use_t = inf & (mag != 0);
	const bool keep = (mag == 0);
	bad = bad | (!inf & !keep & hz);
	acc.X = selg(keep, acc.X, selg(use_t, T.X, S.X));
	acc.Y = selg(keep, acc.Y, selg(use_t, ty, S.Y));
	acc.Z = selg(keep, acc.Z, selg(use_t, T.Z, S.Z));
	inf = inf & keep;
}

// occupancy of the Straus loop: the unit's window-loop choice (G29_OCC) unless -DMSM_WAVES=n pins it at n waves per SIMD (0: the compiler's
// own choice); A/B in profiles/r5l_schnorr_msm_prefetch.md
#if !defined(MSM_WAVES)
#define MSM_OCC G29_OCC
#elif MSM_WAVES == 0
#define MSM_OCC
#else
#define MSM_OCC __attribute__((amdgpu_waves_per_eu(MSM_WAVES, MSM_WAVES)))
#endif
template <int PB, int FLAV> __global__ __launch_bounds__(64) MSM_OCC void k_msm_loop_g(EcamdMsmArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL;
	const u32 lane = blockIdx.x * 64 + threadIdx.x;
	if (lane >= A.L) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const int wwin = 2 * (int)A.wlen, zwin = 2 * (int)A.zlen;
	// (a placeholder until the first non-zero digit replaces it: the lane's first key)
	Jac<PB> acc;
	{
		const TabEnt<PB> T = tab_load<PB>(A.tbl + (size_t)lane * L::ITEMW, 0);
		acc.X = T.X;
		acc.Y = weaken<typename Cls<PB>::FA>(T.Y);
		acc.Z = T.Z;
	}
	bool inf = true, bad = false;
	// One flat sequence of terms: window pos = wwin .. 0 (position wwin / zwin holds the carry digit 0 / 1 of a recoding), inside a window the
	// K keys (digit `pos` of z_i (q - e_i)) and, from window zwin down, the K signature points (digit `pos` of z_i, negated).  The next term's
	// digit is fetched before the current addition starts.  Touching the next term's table entry as well (-DMSM_PREFETCH) was measured and
	// does not pay (secp256k1, 2^20 items: 19.8 against 19.0 ms, profiles/r5l_schnorr_msm_prefetch.md): the loop waits on its dependent
	// multiplication chains, not on its look-ups.
	auto locate = [&](int pos, u32 o, const u32 *&tb, int &dig) {
		const bool isR = o >= A.K;
		const u32 j = isR ? o - A.K : o;
		const u32 item = j * A.L + lane;
		const bool live = item < A.n;
		tb = A.tbl + (size_t)((isR ? A.n : 0u) + (live ? item : 0u)) * L::ITEMW;
		const u32 *kr = tb + 8 * L::ENTW;
		const int top = isR ? zwin : wwin;
		const int d = (pos == top) ? (int)kr[L::KRECW - 1] : (int)((kr[pos >> 3] >> (4 * (pos & 7))) & 15u) - 8;
		dig = live ? d : 0;
	};
	int pos = wwin;
	u32 o = 0;
	const u32 *ctb;
	int cdig;
	locate(pos, o, ctb, cdig);
#pragma unroll 1
	for (;;) {
		const u32 nops = (pos <= zwin) ? 2u * A.K : A.K;
		int npos = pos;
		u32 no = o + 1;
		if (no == nops) {
			npos = pos - 1;
			no = 0;
		}
		const bool has_next = npos >= 0;
		const u32 *ntb = ctb;
		int ndig = 0;
#ifdef MSM_PREFETCH
		u32 t0 = 0, t1 = 0;
#endif
		if (has_next) {
			locate(npos, no, ntb, ndig);
#ifdef MSM_PREFETCH
			const u32 nmag = (u32)(ndig < 0 ? -ndig : ndig);
			const u32 *ent = ntb + (size_t)(nmag ? nmag - 1 : 0) * L::ENTW;
			t0 = ent[0];
			t1 = ent[3 * NL - 1];
#endif
		}
		msm_step<PB>(acc, inf, bad, ctb, cdig, o >= A.K, K);
#ifdef MSM_PREFETCH
		asm volatile("" ::"v"(t0), "v"(t1));   // (keeps the two touches alive until here; their values are not used)
#endif
		if (!has_next) {
			break;
		}
		if (npos != pos) {
#pragma unroll 1
			for (int d = 0; d < 4; d++) {
				acc = dbl(acc, K);
			}
		}
		pos = npos;
		o = no;
		ctb = ntb;
		cdig = ndig;
	}
	// a doubling that reached infinity silently (cofactor curves) leaves Z = 0: not a sum this path vouches for
	if (!inf) {
		bad = bad | is_zero_mulout(mulc(acc.Z, onec, K), K);
	}
	u32 *rec = A.rec + (size_t)lane * MsmLay<PB>::RECW;
	jac_store<PB>(rec, acc);
	rec[L::ENTW] = inf ? 1u : 0u;
	if (bad) {
		atomicOr(A.flagword, 2u);
	}
}

template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_msm_sum_g(const u32 *in, u32 count, u32 *out, u32 *flagword, int gslot)
{
	typedef Lay<PB> L;
	constexpr int NL = L::NL, RECW = MsmLay<PB>::RECW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	const u32 first = t * MSM_FAN;
	if (first >= count) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	Jac<PB> acc = jac_load<PB>(in + (size_t)first * RECW);
	bool inf = in[(size_t)first * RECW + L::ENTW] != 0u, bad = false;
#pragma unroll 1
	for (u32 k = 1; k < MSM_FAN; k++) {
		const u32 r = first + k;
		const bool live = r < count;
		const u32 *rec = in + (size_t)(live ? r : first) * RECW;
		const Jac<PB> P = jac_load<PB>(rec);
		const bool pinf = !live || rec[L::ENTW] != 0u;
		bool hz;
		const Jac<PB> S = add_jac(acc, P.X, P.Y, P.Z, hz, K);
		bad = bad | (!inf & !pinf & hz);   // equal or opposite partial sums: not decided here
		acc.X = selg(pinf, acc.X, selg(inf, P.X, S.X));
		acc.Y = selg(pinf, acc.Y, selg(inf, P.Y, S.Y));
		acc.Z = selg(pinf, acc.Z, selg(inf, P.Z, S.Z));
		inf = inf & pinf;
	}
	u32 *rec = out + (size_t)t * RECW;
	jac_store<PB>(rec, acc);
	rec[L::ENTW] = inf ? 1u : 0u;
	if (bad) {
		atomicOr(flagword, 4u);
	}
}

// sum + [c]G = infinity  <=>  both infinite, or X = x Z^2 and Y = -y Z^3 with (x, y) = [c]G affine (as the comb path exported it)
template <int PB, int FLAV>
__global__ __launch_bounds__(64) void k_msm_final_g(const u32 *in, const u8 *gen, const u8 *gen_status, u32 clen, const u32 *flagword, u8 *verdict,
						     u32 *sum_out, int gslot, u32 cof_dbl);
template <int PB> static __device__ __forceinline__ void bkt_add_aff(Jac<PB> &acc, bool &inf, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2,
								      const typename Cls<PB>::FA &onez, const CurveG<Cfg<PB>::NL> &K);
template <int PB, int FLAV>
__global__ __launch_bounds__(64) void k_msm_final_g(const u32 *in, const u8 *gen, const u8 *gen_status, u32 clen, const u32 *flagword, u8 *verdict,
						     u32 *sum_out, int gslot, u32 cof_dbl)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, RECW = MsmLay<PB>::RECW;
	if (blockIdx.x != 0 || threadIdx.x != 0) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	if (sum_out) {
		for (int w = 0; w < RECW; w++) {
			sum_out[w] = in[w];
		}
	}
	const Jac<PB> S = jac_load<PB>(in);
	const bool sinf = in[L::ENTW] != 0u;
	const u32 gst = gen_status[0];
	bool ok = false;
	if (cof_dbl != 0u) {
		// EdDSA's cofactored equation: T = sum + [c]G by the complete addition, doubled cof_dbl times, must be the point at infinity
		// (_eddsa_verify_batch, sig/eddsa.c:2580-2860: every point of its combination enters multiplied by the cofactor)
		Jac<PB> T = S;
		bool tinf = sinf;
		ok = gst == 0u || gst == 2u;
		if (gst == 0u) {
			EcamdSmulArgs G;
			G.points = gen;
			G.pstride = 2u * clen;
			G.clen = clen;
			FM xg, yg;
			ok = import_point<PB>(G, 0, xg, yg, K);
			bkt_add_aff<PB>(T, tinf, weaken<FA>(xg), weaken<FA>(yg), weaken<FA>(onec), K);
		}
		for (u32 d = 0; d < cof_dbl && !tinf; d++) {
			T = dbl(T, K);
			tinf = is_zero_mulout(mulc(T.Z, onec, K), K);
		}
		ok = ok & tinf;
	} else if (gst == 2u) {
		ok = sinf;
	} else if (gst == 0u && !sinf) {
		EcamdSmulArgs G;
		G.points = gen;
		G.pstride = 2u * clen;
		G.clen = clen;
		FM xg, yg;
		ok = import_point<PB>(G, 0, xg, yg, K);
		const auto zz = sqrc(S.Z, K);
		const auto zzz = mulc(zz, S.Z, K);
		// (the sum's coordinates through a multiplication by one first: values below 2p whatever the accumulator class allows)
		const auto dx = carry(sub_auto<1>(mulc(xg, zz, K), mulc(S.X, onec, K), K));
		const auto dy = carry(add(mulc(yg, zzz, K), mulc(S.Y, onec, K)));
		ok = ok & is_zero_mulout(mulc(dx, onec, K), K) & is_zero_mulout(mulc(dy, onec, K), K);
		ok = ok & !is_zero_mulout(mulc(S.Z, onec, K), K);
	}
	verdict[0] = (ok && flagword[0] == 0u) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Round 6: the same combination by BUCKETS (Pippenger) -- per item 24 complete additions instead of the Straus loop's 97 additions and
// 64 shared doublings.  T - [c]G = sum_i [w_i]Y_i - [z_i]R_i with w_i of 8 wlen bits and z_i of 128: cut into windows of c bits,
//     sum_win 2^(c win) sum_b b B[win][b],     B[win][b] = the sum of the points whose digit in window `win` is b.
//   k_bkt_points_g   the 2n points imported once (on-curve check, or BIP0340's lift_x), R negated: affine (x, y) in the unit's
//                    representation
//   (k_bkt_hist / k_bkt_scan / k_bkt_scatter, ecamd_kernels.hip: a counting sort of the (window, digit, point) triples -- point
//                    indices in bucket order)
//   k_bkt_accum_g    one lane per bucket: its points added up
//   k_bkt_reduce_g   sum_b b B[b] by levels of f = ecamd_bkt_fold(): a lane folds f consecutive entries X[0..f) into T = sum X[r] and
//                    U = sum r X[r] (running sums, 2 (f - 1) additions), so that  V(X) = sum_j U_j + f V(T);  the sums of the U's of
//                    earlier levels ride along as carry arrays (blockIdx.y > 0: plain sums of f).  Every level is latency-bound --
//                    its additions are one dependent chain per lane -- so what counts is the number of DEPENDENT additions over all
//                    levels, 2 (f - 1) log_f(2^c): 120 for f = 16 (round 6's first version), 84 for f = 8 (the default: six launches,
//                    the last of two entries), 48 for f = 4, 32 for f = 2 (sixteen launches: the launches outweigh the chains)
//   k_bkt_window_g   per window V = C_0 + f (C_1 + f (... + f U_last)), scaled by 2^(c win); k_bkt_total_g adds the windows up
//   k_msm_final_g    as for the Straus form
// Every addition here is COMPLETE (bkt_add): equal points are doubled and opposite ones cancel, exactly -- batches whose keys repeat
// (one signer, many messages) fill buckets with multiples of one point, and "P + P" is then the common case, not an exceptional one.
// ------------------------------------------------------------------------------------------
// The order in which the accumulation's waves are dispatched.  k_bkt_rank sorts every 4096 consecutive buckets by size, longest first, so that
// the 64 lanes of a wave hold buckets of one size: wave w of a group (w = 0 .. 63) holds the group's (w + 1)-th 64th of the size distribution.
// Dispatched in lane order, block b lands on the SIMD b mod 1024 (1024 = 16 x 64), so a SIMD got the SAME quantile of every group -- the
// longest buckets of all its waves or the shortest of all: 0.6 VALU busy, the kernel as long as the SIMDs with the top quantile
// (profiles/r6_bucket_kernel_bound.md).  Block b now serves wave (b / G) of group (b mod G), G the number of groups: the longest waves of all
// groups go first (longest-processing-time-first for the slots that free up), and a SIMD's waves come from every part of the distribution.
// The order's TOP window is another distribution altogether -- its n keys crowd into (q >> 16 top) + 1 buckets, eight times the others' length
// for Ed25519 -- so when the launch holds it (tw: its index among the launch's windows, else -1) its wpb blocks go first, the rest behind them.
static __device__ __forceinline__ u32 bkt_block_order(u32 b, u32 nblocks, bool ranked, int tw, u32 wpb)
{
	if (!ranked || (nblocks & 63u) != 0u || nblocks < 64u) {
		return b;
	}
	if (tw < 0 || (wpb & 63u) != 0u || wpb == 0u || nblocks < wpb || (u32)tw * wpb + wpb > nblocks) {
		const u32 G = nblocks >> 6;
		return (b % G) * 64u + b / G;
	}
	if (b < wpb) {
		const u32 Gt = wpb >> 6;
		return (u32)tw * wpb + (b % Gt) * 64u + b / Gt;
	}
	const u32 r = b - wpb, Gr = (nblocks - wpb) >> 6;
	if (Gr == 0u) {
		return b;
	}
	const u32 idx = (r % Gr) * 64u + r / Gr;          // among the blocks of the other windows
	return idx < (u32)tw * wpb ? idx : idx + wpb;
}

template <int PB> struct BktLay {
	static constexpr int NL = Cfg<PB>::NL;
	static constexpr int PENTW = ((2 * NL + 3) / 4) * 4;   // affine point record: the words that travel
	// ... and the words between records: the next power of two, so that a record never straddles a 128-byte line (nine limbs: 80-byte records at an
	// 80-byte stride crossed a line every other time -- 4.5 GB of HBM traffic per 2^20 items for 2 GB of records, profiles/r6_bucket_kernel_bound.md)
	static constexpr int PSTRIDE = PENTW <= 16 ? 16 : (PENTW <= 32 ? 32 : (PENTW <= 64 ? 64 : 128));
	static constexpr int RECW = MsmLay<PB>::RECW;          // Jacobian record + "is infinity" word
};

template <int PB> static __device__ __forceinline__ void bkt_add(Jac<PB> &acc, bool &inf, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2,
								  const typename Cls<PB>::FA &Z2, bool inf2, const CurveG<Cfg<PB>::NL> &K)
{
	typedef typename Cls<PB>::FC FC;
	bool hz;
	Jac<PB> S = add_jac(acc, X2, Y2, Z2, hz, K);
	bool cancel = false;
	if (!inf && !inf2 && hz) {
		// the same x: the same point (the sum is its double) or opposite points (the sum is infinity); decided exactly on Y2 Z1^3 - Y1 Z2^3
		const FC onec = constant<FC>(K.one);
		const auto z1z1 = sqrc(acc.Z, K);
		const auto z2z2 = sqrc(Z2, K);
		const auto s1 = mulc(mulc(acc.Y, Z2, K), z2z2, K);
		const auto s2 = mulc(mulc(Y2, acc.Z, K), z1z1, K);
		const auto d = carry(sub_auto<1>(s2, s1, K));
		if (is_zero_mulout(mulc(d, onec, K), K)) {
			S = dbl(acc, K);
			cancel = is_zero_mulout(mulc(S.Z, onec, K), K);   // the double of a point of order two (curves of even order only)
		} else {
			cancel = true;
		}
	}
	acc.X = selg(inf2, acc.X, selg(inf, X2, S.X));
	acc.Y = selg(inf2, acc.Y, selg(inf, Y2, S.Y));
	acc.Z = selg(inf2, acc.Z, selg(inf, Z2, S.Z));
	inf = (inf & inf2) | cancel;
}

// the same with an affine second operand (the bucket accumulation)
template <int PB> static __device__ __forceinline__ void bkt_add_aff(Jac<PB> &acc, bool &inf, const typename Cls<PB>::FA &X2, const typename Cls<PB>::FA &Y2,
								      const typename Cls<PB>::FA &onez, const CurveG<Cfg<PB>::NL> &K)
{
	typedef typename Cls<PB>::FC FC;
	bool hz;
	Jac<PB> S = add_aff(acc, X2, Y2, hz, K);
	bool cancel = false;
	if (!inf && hz) {
		const FC onec = constant<FC>(K.one);
		const auto z1z1 = sqrc(acc.Z, K);
		const auto s2 = mulc(mulc(Y2, acc.Z, K), z1z1, K);
		const auto d = carry(sub_auto<1>(s2, mulc(acc.Y, onec, K), K));
		if (is_zero_mulout(mulc(d, onec, K), K)) {
			S = dbl(acc, K);
			cancel = is_zero_mulout(mulc(S.Z, onec, K), K);   // the double of a point of order two (curves of even order only)
		} else {
			cancel = true;
		}
	}
	acc.X = selg(inf, X2, S.X);
	acc.Y = selg(inf, Y2, S.Y);
	acc.Z = selg(inf, onez, S.Z);
	inf = cancel;
}

template <int PB> static __device__ __forceinline__ void bkt_rec_store(u32 *rec, const Jac<PB> &P, bool inf)
{
	jac_store<PB>(rec, P);
	rec[Lay<PB>::ENTW] = inf ? 1u : 0u;
}
template <int PB> static __device__ __forceinline__ Jac<PB> bkt_rec_load(const u32 *rec, bool &inf)
{
	inf = rec[Lay<PB>::ENTW] != 0u;
	return jac_load<PB>(rec);
}
// a valid placeholder for an accumulator that holds nothing yet
template <int PB> static __device__ __forceinline__ Jac<PB> bkt_blank(const CurveG<Cfg<PB>::NL> &K)
{
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FC FC;
	Jac<PB> P;
	P.X = weaken<FA>(constant<FC>(K.one));
	P.Y = P.X;
	P.Z = P.X;
	return P;
}

template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_bkt_points_g(EcamdMsmArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	constexpr int NL = L::NL, PENTW = BktLay<PB>::PENTW, PSTRIDE = BktLay<PB>::PSTRIDE;
	const u32 rel = blockIdx.x * 64 + threadIdx.x;
	if (rel >= (A.pt_count ? A.pt_count : 2 * A.n)) {
		return;
	}
	const u32 idx = A.pt_first + rel;
	const bool isR = idx >= A.n;
	const u32 i = isR ? idx - A.n : idx;
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	EcamdSmulArgs S;
	S.points = isR ? A.ptsR : A.ptsY;
	S.pstride = 2u * A.clen;
	S.clen = A.clen;
	FM xm, ym;
	bool bad;
	if (isR && A.r_fmt == 1u) {
		bad = !msm_lift_x<PB>(A.ptsR + (size_t)i * A.clen, (int)A.clen, xm, ym, K);
	} else {
		bad = !import_point<PB>(S, i, xm, ym, K);
	}
	const FA x = weaken<FA>(xm);
	const FA y = selg(isR, neg<PB>(ym, K), weaken<FA>(ym));   // the equation subtracts [z_i]R_i
	if (A.cof_dbl != 0u && !isR && !bad) {
		// EdDSA on a curve of order 2^d q: a key with [2^d]A = infinity is the reference's per-item rejection (sig/eddsa.c:2801-2806) -- here
		// "not decided" (the Straus form sees the same key as a table multiple at infinity)
		typedef typename Cls<PB>::FC FC;
		const FC onec = constant<FC>(K.one);
		Jac<PB> T;
		T.X = x;
		T.Y = y;
		T.Z = weaken<FA>(onec);
		for (u32 d = 0; d < A.cof_dbl; d++) {
			T = dbl(T, K);
		}
		bad = is_zero_mulout(mulc(T.Z, onec, K), K);
	}
	u32 buf[PENTW];
#pragma unroll
	for (int w = 0; w < NL; w++) {
		buf[w] = x.l[w];
		buf[NL + w] = y.l[w];
	}
#pragma unroll
	for (int w = 2 * NL; w < PENTW; w++) {
		buf[w] = 0;
	}
	uint4 *d = (uint4 *)(A.pts + (size_t)idx * PSTRIDE);
#pragma unroll
	for (int q = 0; q < PENTW / 4; q++) {
		d[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
	if (bad) {
		atomicOr(A.flagword, 1u);   // the point does not import / the abscissa has no point: not decided here
	}
}

template <int PB, int FLAV> __global__ __launch_bounds__(64) G29_OCC void k_bkt_accum_g(EcamdMsmArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, PENTW = BktLay<PB>::PENTW, PSTRIDE = BktLay<PB>::PSTRIDE, RECW = BktLay<PB>::RECW;
	const u32 NB = 1u << A.c;
	const u32 wfirst = A.win_first, wcount = A.win_count ? A.win_count : A.nwin;
	const int tw = (A.cap != 0u && A.top_win >= wfirst && A.top_win < wfirst + wcount) ? (int)(A.top_win - wfirst) : -1;
	const u32 rel = bkt_block_order(blockIdx.x, gridDim.x, A.perm != nullptr && A.c >= 12u, tw, NB >> 6) * 64 + threadIdx.x;
	if (rel >= (A.win_count ? A.win_count : A.nwin) * NB) {
		return;
	}
	const u32 lane = A.win_first * NB + rel;
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const u32 t = A.perm ? A.perm[lane] : lane;
	const u32 win = t >> A.c, b = t & (NB - 1u);
	u32 cnt = b ? A.bcount[t] : 0u;
	const u32 *ord;
	if (A.cap) {
		u32 capb;
		ord = A.order + ecamd_bkt_slot(t, A.cap, A.cap_top, A.top_win, &capb);
		cnt = cnt < capb ? cnt : capb;
	} else {
		ord = A.order + (size_t)win * 2u * A.n + A.bstart[t];
	}
	const FA onez = weaken<FA>(constant<FC>(K.one));
	Jac<PB> acc = bkt_blank<PB>(K);
	bool inf = true;
	// the next point's record is on its way while the current addition runs (its index was read an iteration earlier still)
	auto fetch = [&](u32 idx, u32 *buf) {
		const uint4 *src = (const uint4 *)(A.pts + (size_t)idx * PSTRIDE);
#pragma unroll
		for (int q = 0; q < PENTW / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x;
			buf[4 * q + 1] = v.y;
			buf[4 * q + 2] = v.z;
			buf[4 * q + 3] = v.w;
		}
	};
	u32 nbuf[PENTW];
	u32 nidx = 0;
	if (cnt) {
		fetch(ord[0], nbuf);
		nidx = cnt > 1 ? ord[1] : 0u;
	}
#pragma unroll 1
	for (u32 k = 0; k < cnt; k++) {
		FA x, y;
#pragma unroll
		for (int w = 0; w < NL; w++) {
			x.l[w] = nbuf[w];
			y.l[w] = nbuf[NL + w];
		}
		if (k + 1 < cnt) {
			fetch(nidx, nbuf);
			nidx = k + 2 < cnt ? ord[k + 2] : 0u;
		}
		bkt_add_aff<PB>(acc, inf, x, y, onez, K);
	}
	bkt_rec_store<PB>(A.bsum + (size_t)t * RECW, acc, inf);
}

// one level of the bucket reduction (see the header above).  in: nwin x Lin records; V.fold consecutive entries per lane.
// blockIdx.y = 0: T and U of the level's input; blockIdx.y = k > 0: the sums of 16 of carry array k - 1.
#define BKT_MAXCARRY 16
struct BktLevel {
	const u32 *inT;
	const u32 *inC[BKT_MAXCARRY];
	u32 *outT, *outU;
	u32 *outC[BKT_MAXCARRY];
	u32 Lin, Lout, nwin, ncarry;
	u32 fold;                       // entries per lane (ecamd_bkt_fold(): a power of two)
};
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_bkt_reduce_g(BktLevel V, int gslot)
{
	typedef Lay<PB> L;
	constexpr int NL = L::NL, RECW = BktLay<PB>::RECW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= V.nwin * V.Lout) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const u32 win = t / V.Lout, j = t - win * V.Lout;
	const u32 first = j * V.fold, len = (V.Lin - first) < V.fold ? (V.Lin - first) : V.fold;
	const u32 role = blockIdx.y;
	const u32 *in = (role == 0 ? V.inT : V.inC[role - 1]) + ((size_t)win * V.Lin + first) * RECW;
	Jac<PB> run = bkt_blank<PB>(K), acc = run;
	bool rinf = true, ainf = true;
	if (role == 0) {
#pragma unroll 1
		for (u32 r = len; r-- > 1;) {
			bool pinf;
			const Jac<PB> P = bkt_rec_load<PB>(in + (size_t)r * RECW, pinf);
			bkt_add<PB>(run, rinf, P.X, P.Y, P.Z, pinf, K);
			bkt_add<PB>(acc, ainf, run.X, run.Y, run.Z, rinf, K);
		}
		{
			bool pinf;
			const Jac<PB> P = bkt_rec_load<PB>(in, pinf);
			bkt_add<PB>(run, rinf, P.X, P.Y, P.Z, pinf, K);
		}
		bkt_rec_store<PB>(V.outT + (size_t)t * RECW, run, rinf);
		bkt_rec_store<PB>(V.outU + (size_t)t * RECW, acc, ainf);
	} else {
#pragma unroll 1
		for (u32 r = 0; r < len; r++) {
			bool pinf;
			const Jac<PB> P = bkt_rec_load<PB>(in + (size_t)r * RECW, pinf);
			bkt_add<PB>(run, rinf, P.X, P.Y, P.Z, pinf, K);
		}
		bkt_rec_store<PB>(V.outC[role - 1] + (size_t)t * RECW, run, rinf);
	}
}

// per window: V = C_0 + 16 (C_1 + 16 (... + 16 U_last)) from the one record per window every array has come down to, then 2^(c win) V
struct BktWindows {
	const u32 *U;                   // nwin records: the last level's U
	const u32 *C[BKT_MAXCARRY];     // nwin records each: the totals of the earlier levels' U, first level first
	u32 *out;                       // nwin records
	u32 nwin, ncarry, c;
	u32 fold_log2;                  // V = C_0 + fold (C_1 + fold (...)): that many doublings between the levels
	u32 win_base;                   // record w of these arrays is window win_base + w of the combination (its weight: 2^(c (win_base + w)))
};
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_bkt_window_g(BktWindows V, int gslot)
{
	typedef Lay<PB> L;
	constexpr int NL = L::NL, RECW = BktLay<PB>::RECW;
	const u32 win = blockIdx.x * 64 + threadIdx.x;
	if (win >= V.nwin) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const typename Cls<PB>::FC onec = constant<typename Cls<PB>::FC>(K.one);
	bool inf;
	Jac<PB> acc = bkt_rec_load<PB>(V.U + (size_t)win * RECW, inf);
	// a doubling of a placeholder is harmless (`inf` keeps it out of the sums); a doubling that REACHES infinity -- a sum of order two, on a
	// curve of even order only -- is noticed: one test per doubling in a kernel of nwin lanes
	auto dbl_exact = [&]() {
		acc = dbl(acc, K);
		if (!inf && is_zero_mulout(mulc(acc.Z, onec, K), K)) {
			inf = true;
			acc = bkt_blank<PB>(K);
		}
	};
#pragma unroll 1
	for (u32 k = V.ncarry; k-- > 0;) {
#pragma unroll 1
		for (u32 d = 0; d < V.fold_log2; d++) {      // x fold
			dbl_exact();
		}
		bool cinf;
		const Jac<PB> P = bkt_rec_load<PB>(V.C[k] + (size_t)win * RECW, cinf);
		bkt_add<PB>(acc, inf, P.X, P.Y, P.Z, cinf, K);
	}
#pragma unroll 1
	for (u32 d = 0; d < V.c * (V.win_base + win); d++) {
		dbl_exact();
	}
	bkt_rec_store<PB>(V.out + (size_t)win * RECW, acc, inf);
}
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_bkt_total_g(const u32 *in, u32 nwin, u32 *out, u32 *flagword, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, RECW = BktLay<PB>::RECW;
	if (blockIdx.x != 0 || threadIdx.x != 0) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	bool inf;
	Jac<PB> acc = bkt_rec_load<PB>(in, inf);
#pragma unroll 1
	for (u32 w = 1; w < nwin; w++) {
		bool pinf;
		const Jac<PB> P = bkt_rec_load<PB>(in + (size_t)w * RECW, pinf);
		bkt_add<PB>(acc, inf, P.X, P.Y, P.Z, pinf, K);
	}
	// a Z that became zero without the bookkeeping noticing (it cannot on a curve of prime order): not a sum this path vouches for
	if (!inf && is_zero_mulout(mulc(acc.Z, constant<FC>(K.one), K), K)) {
		atomicOr(flagword, 2u);
	}
	bkt_rec_store<PB>(out, acc, inf);
}

// ------------------------------------------------------------------------------------------
// Affine-table pipeline (every flavour except the two nine-limb ones, see the launcher and HAVE_MADD in ecamd_jacg.h): the
// pipeline of ecamd_p256_kernel.hip
//   k_table_g    import + on-curve check, Jacobian multiples 2P..8P into the item's staging slots, recoded scalar
//   k_affine_g   2P..8P -> affine with ONE inversion per AFFG_K items x 7 entries (Montgomery's trick; the prefix products
//                rest in the affine slots they are about to be replaced by)
//   k_loop_g     signed window w = 4 with the mixed addition madd_jac (8M + 3S instead of 12M + 4S) on the tight accumulator
//                class JacT; one exact test of the final Z replaces the per-window exceptional-pair flags
//   k_finalize_g unchanged
// Affine table: 8 entries (x, y) of NL digits each (multiplication results, value < 2p), AENTW words per entry.
// ------------------------------------------------------------------------------------------
template <int PB> struct LayA {
	static constexpr int NL = Cfg<PB>::NL;
	static constexpr int AENTW = ((2 * NL + 3) / 4) * 4;
	static constexpr int AITEMW = 8 * AENTW;
};
template <int PB> static __device__ __forceinline__ void aff_store(u32 *base, int e, const typename Cls<PB>::FM &x, const typename Cls<PB>::FM &y)
{
	constexpr int NL = LayA<PB>::NL, AENTW = LayA<PB>::AENTW;
	u32 buf[AENTW];
#pragma unroll
	for (int i = 0; i < NL; i++) {
		buf[i] = x.l[i];
		buf[NL + i] = y.l[i];
	}
#pragma unroll
	for (int i = 2 * NL; i < AENTW; i++) {
		buf[i] = 0;
	}
	uint4 *d = (uint4 *)(base + (size_t)e * AENTW);
#pragma unroll
	for (int i = 0; i < AENTW / 4; i++) {
		d[i] = make_uint4(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]);
	}
}
// MASKED: every entry is read, the wanted one kept by masking (secret-scalar mode)
template <int PB, bool MASKED>
static __device__ __forceinline__ void aff_load(const u32 *base, u32 e, typename Cls<PB>::FM &x, typename Cls<PB>::FM &y)
{
	constexpr int NL = LayA<PB>::NL, AENTW = LayA<PB>::AENTW;
	u32 buf[AENTW];
	if (MASKED) {
#pragma unroll
		for (int i = 0; i < AENTW; i++) {
			buf[i] = 0;
		}
#pragma unroll 1
		for (u32 ee = 0; ee < 8; ee++) {
			const u32 m = (ee == e) ? 0xffffffffu : 0u;
			const uint4 *sp = (const uint4 *)(base + (size_t)ee * AENTW);
#pragma unroll
			for (int i = 0; i < AENTW / 4; i++) {
				const uint4 v = sp[i];
				buf[4 * i] |= v.x & m;
				buf[4 * i + 1] |= v.y & m;
				buf[4 * i + 2] |= v.z & m;
				buf[4 * i + 3] |= v.w & m;
			}
		}
	} else {
		const uint4 *sp = (const uint4 *)(base + (size_t)e * AENTW);
#pragma unroll
		for (int i = 0; i < AENTW / 4; i++) {
			const uint4 v = sp[i];
			buf[4 * i] = v.x;
			buf[4 * i + 1] = v.y;
			buf[4 * i + 2] = v.z;
			buf[4 * i + 3] = v.w;
		}
	}
#pragma unroll
	for (int i = 0; i < NL; i++) {
		x.l[i] = buf[i];
		y.l[i] = buf[NL + i];
	}
}
// one field element in the x half of an affine slot (prefix products of the shared inversion)
template <int PB> static __device__ __forceinline__ void fe_store(u32 *dst, const typename Cls<PB>::FM &v)
{
#pragma unroll
	for (int i = 0; i < Cfg<PB>::NL; i++) {
		dst[i] = v.l[i];
	}
}
template <int PB> static __device__ __forceinline__ typename Cls<PB>::FM fe_load(const u32 *src)
{
	typename Cls<PB>::FM v;
#pragma unroll
	for (int i = 0; i < Cfg<PB>::NL; i++) {
		v.l[i] = src[i];
	}
	return v;
}
// Jacobian point (class FA) in staging slot e
template <int PB> static __device__ __forceinline__ void jac_store_at(u32 *base, int e, const Jac<PB> &P)
{
	TabEnt<PB> T;
	T.X = P.X;
	T.Z = P.Z;
#pragma unroll
	for (int i = 0; i < Lay<PB>::NL; i++) {
		T.Y.l[i] = P.Y.l[i];
	}
	tab_store<PB, STG_QS>(base, e, T);
}
template <int PB> static __device__ __forceinline__ Jac<PB> jac_load_at(const u32 *base, int e)
{
	const TabEnt<PB> T = tab_load<PB, STG_QS>(base, (u32)e);
	Jac<PB> P;
	P.X = T.X;
	P.Z = T.Z;
#pragma unroll
	for (int i = 0; i < Lay<PB>::NL; i++) {
		P.Y.l[i] = T.Y.l[i];
	}
	return P;
}

template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_table_g(EcamdSmulArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const int clen = (int)A.clen;
	const FC onec = constant<FC>(K.one);
	FM xm, ym;
	const bool ok = import_point<PB>(A, i, xm, ym, K);
	if (!ok) {
		u8 *out = A.out + (size_t)i * 2 * clen;
		A.status[i] = 1;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	u32 *tb = stg_ent<PB>(A.tbl, i);
	aff_store<PB>(A.stg + (size_t)i * LayA<PB>::AITEMW, 0, xm, ym);
	Jac<PB> P1;
	P1.X = weaken<FA>(xm);
	P1.Y = weaken<FA>(ym);
	P1.Z = weaken<FA>(onec);
	bool hz, bad = false;
#if !defined(G29_TABLE_UNROLLED)
	{
		// staging slot e - 1 holds [e]P, e = 2..8.  A rolled loop: [2j + 1]P = [2j]P + P and [2j + 2]P = 2 [j + 1]P for j = 1..3 -- the same
		// four doublings and three additions as the spelled-out order (-DG29_TABLE_UNROLLED), re-reading two multiples from the staging they
		// were just written to, in 40 % of the code (the spelled-out kernel is 300 KB of straight-line code at 384 bits)
		Jac<PB> Pb = dbl(P1, K), Pc = Pb;
		jac_store_at<PB>(tb, 1, Pb);
#pragma unroll 1
		for (int j = 1; j <= 3; j++) {
			Pb = add_jac(jac_load_at<PB>(tb, 2 * j - 1), P1.X, P1.Y, P1.Z, hz, K);
			bad |= hz;
			jac_store_at<PB>(tb, 2 * j, Pb);
			Pc = dbl(jac_load_at<PB>(tb, j), K);
			jac_store_at<PB>(tb, 2 * j + 1, Pc);
		}
		// a doubling reaches infinity silently (points of even order): exact test of the last Z's (8P and 7P)
		bad = bad | is_zero_mulout(mulc(mulc(Pc.Z, Pb.Z, K), onec, K), K);
	}
#else
	{
		// staging slot e - 1 holds [e]P, e = 2..8
		Jac<PB> Pa = dbl(P1, K);                                   // 2P
		jac_store_at<PB>(tb, 1, Pa);
		Jac<PB> Pb = add_jac(Pa, P1.X, P1.Y, P1.Z, hz, K);          // 3P
		bad |= hz;
		jac_store_at<PB>(tb, 2, Pb);
		Pa = dbl(Pa, K);                                            // 4P
		jac_store_at<PB>(tb, 3, Pa);
		Pb = add_jac(Pa, P1.X, P1.Y, P1.Z, hz, K);                  // 5P
		bad |= hz;
		jac_store_at<PB>(tb, 4, Pb);
		Pb = dbl(jac_load_at<PB>(tb, 2), K);                        // 6P
		jac_store_at<PB>(tb, 5, Pb);
		Pb = add_jac(Pb, P1.X, P1.Y, P1.Z, hz, K);                  // 7P
		bad |= hz;
		jac_store_at<PB>(tb, 6, Pb);
		Pa = dbl(Pa, K);                                            // 8P
		jac_store_at<PB>(tb, 7, Pa);
		// a doubling reaches infinity silently (points of even order): exact test of the last Z's
		bad = bad | is_zero_mulout(mulc(mulc(Pa.Z, Pb.Z, K), onec, K), K);
	}
#endif
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;   // a multiple below 9P is infinity: the complete-formula kernel takes the item
		return;
	}
	u32 *kr = stg_kr<PB>(tb);
	const u32 top = recode_scalar<STG_QS>(A.scalars + (size_t)i * A.sstride, (int)A.slen, kr);   // (the scalar fills at most KRECW - 1 words)
	stg_krw(kr, L::KRECW - 1) = top;
	A.status[i] = ECAMD_STATUS_TAB;
}

// items per inversion: 8 for large batches (the inversion costs each of the 7 x 8 entries a few multiplications), fewer
// for small ones so that the kernel still fills the chip (the launcher decides)
#define AFFG_K 8
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_affine_g(EcamdSmulArgs A, int gslot, u32 nthreads, int items)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, AENTW = LayA<PB>::AENTW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	FM c = weaken<FM>(constant<FC>(K.one));
#pragma unroll 1
	for (int j = 0; j < items; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_TAB) {
			continue;
		}
		const u32 *tb = stg_ent<PB>(A.tbl, i);
		u32 *af = A.stg + (size_t)i * LayA<PB>::AITEMW;
#pragma unroll 1
		for (int e = 1; e < 8; e++) {
			fe_store<PB>(af + (size_t)e * AENTW, c);   // the product of everything before this entry
			const Jac<PB> P = jac_load_at<PB>(tb, e);
			c = weaken<FM>(mulc(c, P.Z, K));
		}
	}
	FM tinv = inv<PB>(c, K);
#pragma unroll 1
	for (int j = items - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_TAB) {
			continue;
		}
		const u32 *tb = stg_ent<PB>(A.tbl, i);
		u32 *af = A.stg + (size_t)i * LayA<PB>::AITEMW;
#pragma unroll 1
		for (int e = 7; e >= 1; e--) {
			const Jac<PB> P = jac_load_at<PB>(tb, e);
			const FM zi = weaken<FM>(mul(tinv, fe_load<PB>(af + (size_t)e * AENTW), K));
			tinv = weaken<FM>(mulc(tinv, P.Z, K));
			const FM zi2 = weaken<FM>(sqr(zi, K));
			const FM zi3 = weaken<FM>(mul(zi2, zi, K));
			aff_store<PB>(af, e, weaken<FM>(mulc(P.X, zi2, K)), weaken<FM>(mulc(P.Y, zi3, K)));
		}
	}
}

// layout of the generator's 16-bit comb table (k_comb_build_g / k_comb_g / k_comb_add_g below)
template <int PB> struct CombLay {
	static constexpr int NL = Cfg<PB>::NL;
	static constexpr int NW = (PB + 31) / 32;
	static constexpr int CENTW = ((2 * NL + 3) / 4) * 4;
	static constexpr int NWIN = 2 * NW;
};
#define COMB_PER_WIN 32768

template <int PB, int FLAV, bool MASKED> __global__ __launch_bounds__(64) G29_OCC void k_loop_g(EcamdSmulArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	typedef typename ClsT<PB>::FT FT;
	constexpr int NL = L::NL;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.status[i] != ECAMD_STATUS_TAB) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	u32 *tb = stg_ent<PB>(A.tbl, i);
	const u32 *af = A.stg + (size_t)i * LayA<PB>::AITEMW;
	const u32 *kr = stg_kr<PB>(tb);
	const int slen = (int)A.slen;
	const int nwin = 2 * slen;
	const u32 carry_bit = stg_krw(kr, L::KRECW - 1);   // the leading digit 0 / 1 (k_table_g)
	JacT<PB> acc;
	FM x1, y1;
	aff_load<PB, false>(af, 0, x1, y1);
	acc.X = weaken<FT>(x1);
	acc.Y = weaken<FT>(y1);
	acc.Z = weaken<FT>(onec);
	bool inf = (carry_bit == 0);
	u32 wcur = nwin ? stg_krw(kr, (nwin - 1) >> 3) : 0u, wnext = 0u;
#pragma unroll 1
	for (int t = 0; t < nwin; t++) {
		const int pos = nwin - 1 - t;
		const int dig = (int)((wcur >> (4 * (pos & 7))) & 15u) - 8;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		if ((pos & 7) == 0 && pos != 0) {
			wnext = stg_krw(kr, (pos >> 3) - 1);
		}
		FM tx, tyc;
#ifdef G29_PRELOAD
		aff_load<PB, MASKED>(af, mag ? mag - 1 : 0, tx, tyc);   // A/B: the entry travels while the doublings run
#endif
#pragma unroll 1
		for (int d = 0; d < 4; d++) {
			acc = dbl(acc, K);
		}
#ifndef G29_PRELOAD
		aff_load<PB, MASKED>(af, mag ? mag - 1 : 0, tx, tyc);
#endif
		wcur = ((pos & 7) == 0) ? wnext : wcur;
		const FT tyt = selg(dig < 0, neg_t<PB>(tyc, K), weaken<FT>(tyc));
		const JacT<PB> S = madd_jac(acc, weaken<FA>(tx), weaken<FA>(tyt), K);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		acc.X = selg(keep, acc.X, selg(use_t, weaken<FT>(tx), S.X));
		acc.Y = selg(keep, acc.Y, selg(use_t, tyt, S.Y));
		acc.Z = selg(keep, acc.Z, selg(use_t, weaken<FT>(onec), S.Z));
		inf = inf & keep;
	}
	// an exceptional pair of the mixed addition, or a doubling that reached infinity, leaves Z = 0 for good: one exact test
	if (!inf && is_zero_mulout(mulc(acc.Z, onec, K), K)) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		const int clen = (int)A.clen;
		u8 *out = A.out + (size_t)i * 2 * clen;
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	Jac<PB> R;
	R.X = weaken<FA>(acc.X);
	R.Y = weaken<FA>(acc.Y);
	R.Z = weaken<FA>(acc.Z);
	jac_store<PB, STG_QS>(tb, R);
	A.status[i] = ECAMD_STATUS_JAC;
}

// ECDSA verification (lut_kind 2): W' = [u2]Q + [u1]G.  k_loop_g leaves [u2]Q in the item's staging slot (Jacobian, status
// ECAMD_STATUS_JAC; status 2 when it is the point at infinity); this kernel adds [u1]G from the generator's 16-bit comb table --
// 2 NW + 1 mixed additions instead of a second scalar multiplication and a complete addition (the reference computes uG and vY
// separately and adds them, sig/ecdsa_common.c:786-810; only x mod q of the sum is observable).  An exceptional pair leaves
// Z = 0 and the item goes back as ECAMD_STATUS_REDO: the host verifies it again the reference's way.  A kernel of its own: inside
// the window loop the comb's registers cost the loop a wave per SIMD at 384 bits and both at 448 / 521 bits (231 / 323 / 258
// VGPRs against 167 / 216 / 248; profiles/r3b_fused_verify.md).
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_comb_add_g(EcamdSmulArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	typedef typename ClsT<PB>::FT FT;
	constexpr int NL = L::NL, NW = L::NW, KW = L::KW, CENTW = CombLay<PB>::CENTW, NWIN = CombLay<PB>::NWIN;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 st = A.status[i];
	if (st != ECAMD_STATUS_JAC && st != 2u) {
		return;   // key rejected at import, or an exceptional pair already met
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FT onet = weaken<FT>(onec);
	u32 *tb = stg_ent<PB>(A.tbl, i);
	JacT<PB> acc;
	bool inf = (st == 2u);
	acc.X = onet;
	acc.Y = onet;
	acc.Z = onet;
	if (!inf) {
		// the loop's result: stored from the tight class (k_loop_g widens the type, not the digits), so it re-enters it as it is
		const Jac<PB> R = jac_load<PB, STG_QS>(tb);
#pragma unroll
		for (int w = 0; w < NL; w++) {
			acc.X.l[w] = R.X.l[w];
			acc.Y.l[w] = R.Y.l[w];
			acc.Z.l[w] = R.Z.l[w];
		}
	}
	// k = sum_j s_j 2^(16 j) + D_top 2^(32 NW) over the comb table (see k_comb_g)
	u32 kw[KW];
	load_be<KW>(A.scalars2 + (size_t)i * A.s2len, (int)A.s2len, kw);
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			c += (uint64_t)kw[w] + 0x80008000u;
			kw[w] = (u32)c;
			c >>= 32;
		}
		kw[NW] = (u32)c;  // top digit: 0 or 1
	}
#pragma unroll 1
	for (int j = 0; j <= NWIN; j++) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < KW; w++) {
			word = (w == (j >> 1)) ? kw[w] : word;
		}
		const int dig = (j < NWIN) ? (int)((word >> (16 * (j & 1))) & 0xffffu) - 0x8000 : (int)word;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		const uint4 *src = (const uint4 *)(A.lut + ((size_t)j * COMB_PER_WIN + (mag ? mag - 1 : 0)) * CENTW);
		u32 buf[CENTW];
#pragma unroll
		for (int q = 0; q < CENTW / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x;
			buf[4 * q + 1] = v.y;
			buf[4 * q + 2] = v.z;
			buf[4 * q + 3] = v.w;
		}
		FM tx, tyc;
#pragma unroll
		for (int w = 0; w < NL; w++) {
			tx.l[w] = buf[w];
			tyc.l[w] = buf[NL + w];
		}
		const FT txa = weaken<FT>(tx);
		const FT ty = selg(dig < 0, neg_t<PB>(tyc, K), weaken<FT>(tyc));
		const JacT<PB> S = madd_jac(acc, weaken<FA>(txa), weaken<FA>(ty), K);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		acc.X = selg(keep, acc.X, selg(use_t, txa, S.X));
		acc.Y = selg(keep, acc.Y, selg(use_t, ty, S.Y));
		acc.Z = selg(keep, acc.Z, selg(use_t, onet, S.Z));
		inf = inf & keep;
	}
	if (!inf && is_zero_mulout(mulc(acc.Z, onec, K), K)) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		const int clen = (int)A.clen;
		u8 *out = A.out + (size_t)i * 2 * clen;
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	Jac<PB> R;
	R.X = weaken<FA>(acc.X);
	R.Y = weaken<FA>(acc.Y);
	R.Z = weaken<FA>(acc.Z);
	jac_store<PB, STG_QS>(tb, R);
	A.status[i] = ECAMD_STATUS_JAC;
}

// ---- fixed base: 16-bit comb over a precomputed table of the generator (see ecamd_p256_kernel.hip C') ----
//   k = sum_j s_j 2^(16 j) + D_top 2^(32 NW), s_j = D_j - 0x8000, D = 16-bit digits of K = k + 0x8000..8000
//   (2 NW windows cover every scalar length the fast path takes); table T[j][m-1] = [m 2^(16 j)]G, m = 1..32768,
//   plus [2^(32 NW)]G; entries are affine (x, y) in the field representation of this unit, CENTW words each.
//   [k]G = 2 NW + 1 additions and no doubling.  Partial sums are smaller in magnitude than the next term, so
//   an exceptional pair can only be the last addition of a scalar >= q: flagged, recomputed by k_smul<NW>.

template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_comb_build_g(const u8 *pts, u32 n, u32 clen, u32 *table, int gslot)
{
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = CombLay<PB>::NL, NW = CombLay<PB>::NW, CENTW = CombLay<PB>::CENTW;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	u32 xw[NW], yw[NW];
	load_be<NW>(pts + (size_t)i * 2 * clen, (int)clen, xw);
	load_be<NW>(pts + (size_t)i * 2 * clen + clen, (int)clen, yw);
	u32 buf[CENTW];
	canonical_digits(buf, mul(from_words<PB, NW>(xw), constant<FC>(K.ix), K), K);
	canonical_digits(buf + NL, mul(from_words<PB, NW>(yw), constant<FC>(K.iy), K), K);
#pragma unroll
	for (int w = 2 * NL; w < CENTW; w++) {
		buf[w] = 0;
	}
	uint4 *d = (uint4 *)(table + (size_t)i * CENTW);
#pragma unroll
	for (int q = 0; q < CENTW / 4; q++) {
		d[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

// SCAN4 (round 4, secret scalars: ecamd_ctx_set_secret_scalars): the same kernel over a 4-bit comb -- T[j][m-1] = [m 16^j]G, m = 1..8,
// j = 0 .. 8 NW - 1, and T[8 NW][0] = [2^(32 NW)]G; k = sum_j d_j 16^j + c 2^(32 NW), d_j = D_j - 8, D = nibbles of k + 0x88..8 --
// whose look-ups are SCANS: every lane reads the eight entries of window j, in order, whatever its digit (the addresses depend on j
// alone), and keeps the wanted one by masking.  8 NW + 1 mixed additions and no doubling instead of the masked window loop's 32 NW
// doublings and 8 NW additions (cf. k_p256_comb4m).
template <int PB, int FLAV, bool SCAN4 = false> __global__ __launch_bounds__(64) void k_comb_g(EcamdSmulArgs A, int gslot)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FA FA;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, NW = L::NW, KW = L::KW, CENTW = CombLay<PB>::CENTW, NWIN = SCAN4 ? 8 * NW : CombLay<PB>::NWIN;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const int clen = (int)A.clen;
	u8 *out = A.out + (size_t)i * 2 * clen;
	const int slen = (int)A.slen;  // <= 4 * NW, checked by the host
	u32 kw[KW];
	load_be<KW>(A.scalars + (size_t)i * A.sstride, slen, kw);
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) {
			c += (uint64_t)kw[w] + (SCAN4 ? 0x88888888u : 0x80008000u);
			kw[w] = (u32)c;
			c >>= 32;
		}
		kw[NW] = (u32)c;  // top digit: 0 or 1
	}
	// accumulator: the tight class and the mixed addition where the flavour has them (see k_loop_g), else Jac / add_jac
	typedef typename std::conditional<HAVE_MADD, JacT<PB>, Jac<PB>>::type JA;
	typedef decltype(JA::X) FX;
	const FX onez = weaken<FX>(constant<FC>(K.one));
	JA acc;
	acc.X = onez;  // placeholder, replaced by the first non-zero digit
	acc.Y = onez;
	acc.Z = onez;
	bool inf = true, bad = false, hz = false;
#pragma unroll 1
	for (int j = 0; j <= NWIN; j++) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < KW; w++) {
			word = (w == (SCAN4 ? (j >> 3) : (j >> 1))) ? kw[w] : word;
		}
		const int dig = (j < NWIN) ? (SCAN4 ? (int)((word >> (4 * (j & 7))) & 15u) - 8 : (int)((word >> (16 * (j & 1))) & 0xffffu) - 0x8000) : (int)word;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		u32 buf[CENTW];
		if constexpr (SCAN4) {
			const u32 idx = mag ? mag - 1 : 0;
#pragma unroll
			for (int w = 0; w < CENTW; w++) {
				buf[w] = 0;
			}
#pragma unroll 1
			for (u32 e = 0; e < 8; e++) {
				const uint4 *src = (const uint4 *)(A.lut + ((size_t)j * 8 + e) * CENTW);
				const u32 m = 0u - (u32)(e == idx);
#pragma unroll
				for (int q = 0; q < CENTW / 4; q++) {
					const uint4 v = src[q];
					buf[4 * q] |= v.x & m;
					buf[4 * q + 1] |= v.y & m;
					buf[4 * q + 2] |= v.z & m;
					buf[4 * q + 3] |= v.w & m;
				}
			}
		} else {
			const uint4 *src = (const uint4 *)(A.lut + ((size_t)j * COMB_PER_WIN + (mag ? mag - 1 : 0)) * CENTW);
#pragma unroll
			for (int q = 0; q < CENTW / 4; q++) {
				const uint4 v = src[q];
				buf[4 * q] = v.x;
				buf[4 * q + 1] = v.y;
				buf[4 * q + 2] = v.z;
				buf[4 * q + 3] = v.w;
			}
		}
		FM tx, tyc;
#pragma unroll
		for (int w = 0; w < NL; w++) {
			tx.l[w] = buf[w];
			tyc.l[w] = buf[NL + w];
		}
		const FX txa = weaken<FX>(tx);
		FX ty;
		JA S;
		if constexpr (!HAVE_MADD) {
			ty = selg(dig < 0, neg<PB>(tyc, K), weaken<FA>(tyc));
			S = add_jac(acc, txa, ty, onez, hz, K);
		} else {
			ty = selg(dig < 0, neg_t<PB>(tyc, K), weaken<FX>(tyc));
			S = madd_jac(acc, weaken<FA>(txa), weaken<FA>(ty), K);   // an exceptional pair leaves Z = 0: tested once below
		}
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		acc.X = selg(keep, acc.X, selg(use_t, txa, S.X));
		acc.Y = selg(keep, acc.Y, selg(use_t, ty, S.Y));
		acc.Z = selg(keep, acc.Z, selg(use_t, onez, S.Z));
		inf = inf & keep;
	}
	if constexpr (HAVE_MADD) {
		bad = !inf && is_zero_mulout(mulc(acc.Z, constant<FC>(K.one), K), K);
	}
	if constexpr (SCAN4) {
		// secret scalars: nothing after the scan depends on the scalar -- the accumulator is stored whatever it holds,
		// the output is blanked whatever the outcome (k_finalize_g overwrites it for a finite result, the redo pass for an exceptional
		// one) and the status is a select; k = 0, k >= q and exceptional sums take the same instructions as every other scalar
		u32 *tb = stg_ent<PB>(A.tbl, i);
		Jac<PB> R;
		R.X = weaken<FA>(acc.X);
		R.Y = weaken<FA>(acc.Y);
		R.Z = weaken<FA>(acc.Z);
		jac_store<PB, STG_QS>(tb, R);
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		A.status[i] = bad ? (u8)ECAMD_STATUS_REDO : (inf ? (u8)2 : (u8)ECAMD_STATUS_JAC);
		return;
	}
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		A.status[i] = 2;
		for (int b = 0; b < 2 * clen; b++) {
			out[b] = 0;
		}
		return;
	}
	u32 *tb = stg_ent<PB>(A.tbl, i);
	Jac<PB> R;
	R.X = weaken<FA>(acc.X);
	R.Y = weaken<FA>(acc.Y);
	R.Z = weaken<FA>(acc.Z);
	jac_store<PB, STG_QS>(tb, R);
	A.status[i] = ECAMD_STATUS_JAC;
}

#define FING_K 8
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_finalize_g(EcamdSmulArgs A, int gslot, u32 nthreads)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, NW = L::NW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const int clen = (int)A.clen;
	FM c = weaken<FM>(onec);
#pragma unroll 1
	for (int j = 0; j < FING_K; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			break;
		}
		u32 *tb = stg_ent<PB>(A.tbl, i);
		if (A.status[i] == ECAMD_STATUS_JAC) {
			const Jac<PB> P = jac_load<PB, STG_QS>(tb);
			c = weaken<FM>(mulc(c, P.Z, K));
		}
		stg_fe_store<PB>(tb, 1, c);  // park the prefix product in the second table slot
	}
	FM tinv = inv<PB>(c, K);
	const FC ex = constant<FC>(K.ex), ey = constant<FC>(K.ey);  // out of the Montgomery domain (and back from the isomorphic curve)
#pragma unroll 1
	for (int j = FING_K - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_JAC) {
			continue;
		}
		u32 *tb = stg_ent<PB>(A.tbl, i);
		const Jac<PB> P = jac_load<PB, STG_QS>(tb);
		FM zi = tinv;
		if (j > 0) {
			const u32 ip = t + (u32)(j - 1) * nthreads;
			const FM cp = stg_fe_load<PB>(stg_ent<PB>(A.tbl, ip), 1);
			zi = weaken<FM>(mul(tinv, cp, K));
		}
		tinv = weaken<FM>(mulc(tinv, P.Z, K));
		const FM zi2 = weaken<FM>(sqr(zi, K));
		const FM zi3 = weaken<FM>(mul(zi2, zi, K));
		const auto ax = mulc(P.X, zi2, K);
		const auto ay = mulc(P.Y, zi3, K);
		u8 *out = A.out + (size_t)i * 2 * clen;
		u32 dg[NL], ow[NW];
		canonical_digits(dg, mul(ax, ex, K), K);
		to_words<NL, NW>(ow, dg);
		store_be<NW>(out, clen, ow);
		canonical_digits(dg, mul(ay, ey, K), K);
		to_words<NL, NW>(ow, dg);
		store_be<NW>(out + clen, clen, ow);
		A.status[i] = 0;
	}
}

#ifdef G29_PB
// ---- one field size: constants + launch / upload entry points with the size in their name ----
#define G29_CAT2(a, b) a##b
#define G29_CAT(a, b) G29_CAT2(a, b)
#if defined(G29_M521P)
#define G29_TAG G29_CAT(G29_PB, m)   /* 521m: the secp521r1 flavour (plain residues on 18 limbs) lives beside the dense 521 one */
#define G29_FLAV 1
#elif defined(G29_P25519)
#define G29_TAG G29_CAT(G29_PB, c)   /* 255c: p = 2^255 - 19 beside the dense 255-bit unit */
#define G29_FLAV 2
#elif defined(G29_P448)
#define G29_TAG G29_CAT(G29_PB, g)   /* 448g: p = 2^448 - 2^224 - 1 (WEI448) beside the dense 448-bit unit */
#define G29_FLAV 5
#elif defined(G29_K256)
#define G29_TAG G29_CAT(G29_PB, k)   /* 256k: p = 2^256 - 2^32 - 977 (secp256k1) beside the dense 256-bit unit */
#define G29_FLAV 4
#elif defined(G29_P384S)
#define G29_TAG G29_CAT(G29_PB, n)   /* 384n: secp384r1's prime (signed sparse reduction) beside the dense 384-bit unit */
#define G29_FLAV 3
#elif defined(G29_P224S)
#define G29_TAG G29_CAT(G29_PB, s)   /* 224s: secp224r1's prime (signed sparse reduction) beside the dense 224-bit unit */
#define G29_FLAV 6
#elif defined(G29_P192S)
#define G29_TAG G29_CAT(G29_PB, s)   /* 192s: secp192r1's prime (signed sparse reduction) beside the dense 192-bit unit */
#define G29_FLAV 7
#else
#define G29_FLAV 0
#define G29_TAG G29_PB
#endif
__constant__ SlotsG<Lay<G29_PB>::NL> G29_CAT(g_g29_, G29_TAG);
template <> struct TabGP<G29_PB> {
	static __device__ __forceinline__ const CurveG<Lay<G29_PB>::NL> &get(int slot) { return G29_CAT(g_g29_, G29_TAG).s[slot]; }
};

#ifdef G29_P25519
// ------------------------------------------------------------------------------------------
// Ed25519 point decoding and the X25519 front end on the 2^255 - 19 field of this unit (the same
// computations as k_ed_decode<8> / k_xdh_prep<8> in ecamd_kernels.hip, which stay the reference
// implementation and serve when this unit has no slot): residues are plain here (R = 1).
// ------------------------------------------------------------------------------------------
namespace c25519 {
constexpr int PB = 255;
typedef Cls<PB>::FA FA;
typedef Cls<PB>::FM FM;
typedef Cls<PB>::FC FC;
typedef CurveG<9> CK;

static __device__ __forceinline__ FM sqr_n(FM a, int n, const CK &K)
{
	for (int k = 0; k < n; k++) {
		a = weaken<FM>(sqr(a, K));
	}
	return a;
}
#define M_(a, b) weaken<FM>(mul(a, b, K))
// z^(2^250 - 1); *z11 = z^11
static __device__ FM pow_2_250m1(const FM &z, FM *z11, const CK &K)
{
	const FM z2 = sqr_n(z, 1, K);
	const FM z9 = M_(sqr_n(z2, 2, K), z);
	*z11 = M_(z9, z2);
	const FM a5 = M_(sqr_n(*z11, 1, K), z9);
	const FM a10 = M_(sqr_n(a5, 5, K), a5);
	const FM a20 = M_(sqr_n(a10, 10, K), a10);
	const FM a40 = M_(sqr_n(a20, 20, K), a20);
	const FM a50 = M_(sqr_n(a40, 10, K), a10);
	const FM a100 = M_(sqr_n(a50, 50, K), a50);
	const FM a200 = M_(sqr_n(a100, 100, K), a100);
	return M_(sqr_n(a200, 50, K), a50);
}
// exact comparisons of lazily reduced values: a == b, a == -b
template <class A, class B> static __device__ __forceinline__ bool eq(const A &a, const B &b, const CK &K)
{
	return is_zero_mulout(mulc(carry(sub_auto<1>(a, b, K)), constant<FC>(K.one), K), K);
}
template <class A, class B> static __device__ __forceinline__ bool eq_neg(const A &a, const B &b, const CK &K)
{
	return is_zero_mulout(mulc(carry(add(a, b)), constant<FC>(K.one), K), K);
}
static __device__ __forceinline__ FC digits9(const u32 *d)
{
	FC r;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		r.l[w] = d[w];
	}
	return r;
}
// little-endian 32 bytes -> 8 words
static __device__ __forceinline__ void load_le256(const u8 *src, u32 *w)
{
#pragma unroll
	for (int i = 0; i < 8; i++) {
		w[i] = (u32)src[4 * i] | ((u32)src[4 * i + 1] << 8) | ((u32)src[4 * i + 2] << 16) | ((u32)src[4 * i + 3] << 24);
	}
}
static __device__ __forceinline__ bool below_p(const E<PB, MASK, MASK, 1> &v, const CK &K)
{
	u32 b = 0;
#pragma unroll
	for (int j = 0; j < 9; j++) {
		b = (v.l[j] - K.p[j] - b) >> 31;
	}
	return b != 0;
}
// value (any class the multiplier takes) -> canonical big-endian bytes
template <class A> static __device__ __forceinline__ void store_canon_be(u8 *dst, const A &a, bool ok, const CK &K)
{
	u32 d[9], w[8];
	canonical_digits(d, mulc(a, constant<FC>(K.one), K), K);
	to_words<9, 8>(w, d);
#pragma unroll
	for (int i = 0; i < 8; i++) {
		w[i] = ok ? w[i] : 0u;
	}
	store_be<8>(dst, 32, w);
}
// [2^r]P == infinity for the affine point (x, y)?  (r complete-free Jacobian doublings: a doubling reaches
// infinity exactly when Y = 0, which shows as Z = 0 one step later, or at once for y = 0)
template <class AX, class AY> static __device__ __forceinline__ bool small_order(const AX &x, const AY &y, u32 r, const CK &K)
{
	Jac<PB> P;
	P.X = weaken<FA>(x);
	P.Y = weaken<FA>(y);
	P.Z = weaken<FA>(constant<FC>(K.one));
	for (u32 k = 0; k < r; k++) {
		P = dbl(P, K);
	}
	return is_zero_mulout(mulc(P.Z, constant<FC>(K.one), K), K);
}

struct DecXY {
	FA x;      // the selected root (or its negation), class FA
	FM ym;
	bool ok;
	bool neutral;   // the encoding of (0, 1): the reference maps it to the point at infinity (see ed_decode_xy in ecamd_kernels.hip)
};
static __device__ __forceinline__ DecXY decode_xy(const EcamdEdDecodeArgs &A, const u8 *src, const CK &K)
{
	DecXY R;
	u32 yw[8];
	load_le256(src, yw);
	const u32 x0 = yw[7] >> 31;
	yw[7] &= 0x7fffffffu;
	const auto yd = from_words<PB, 8>(yw);
	bool ok = below_p(yd, K);
	const FC onec = constant<FC>(K.one);
	const FM ym = M_(yd, onec);
	const FM y2 = sqr_n(ym, 1, K);
	const auto u = carry(sub_auto<1>(onec, y2, K));                                    // 1 - y^2
	FC am1;                                                                             // a = -1 = p - 1
#pragma unroll
	for (int w = 0; w < 9; w++) {
		am1.l[w] = K.p[w] - (w == 0 ? 1u : 0u);
	}
	const auto v = carry(sub_auto<1>(am1, M_(digits9(A.g_d), y2), K));                // a - d y^2
	const FM v3 = M_(weaken<FM>(sqrc(v, K)), v);
	const FM v7 = M_(sqr_n(v3, 1, K), v);
	const FM t = M_(u, v7);
	FM t11;
	const FM pw = M_(sqr_n(pow_2_250m1(t, &t11, K), 2, K), t);                         // t^(2^252 - 3)
	const FM beta = M_(M_(u, v3), pw);
	const FM chk = M_(v, sqr_n(beta, 1, K));
	const bool root = eq(chk, u, K);
	const bool alt = eq_neg(chk, u, K);
	ok = ok & (root | alt);
	const FM xs = selg(alt & !root, M_(beta, digits9(A.g_sm1)), beta);
	u32 xd[9];
	canonical_digits(xd, xs, K);
	u32 nz = 0;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		nz |= xd[w];
	}
	R.neutral = ok & (nz == 0) & (x0 == 0) & (yw[0] == 1u) & ((yw[1] | yw[2] | yw[3] | yw[4] | yw[5] | yw[6] | yw[7]) == 0u);
	ok = ok & (nz != 0);  // x = 0: (0, 1) is the neutral element (flagged above), (0, -1) dies in fp_inv(0)
	R.x = selg((xd[0] & 1u) != x0, neg<PB>(xs, K), weaken<FA>(xs));
	R.ym = ym;
	R.ok = ok;
	return R;
}
#undef M_
}  // namespace c25519

__global__ __launch_bounds__(64) void k_ed_decode_c25519(EcamdEdDecodeArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FM onem = weaken<FM>(onec);
	DecXY P[2];
	P[0] = decode_xy(A, A.encA + (size_t)i * A.strideA, K);
	P[1] = decode_xy(A, A.encR + (size_t)i * A.strideR, K);
	FM den[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const auto omy = carry(sub_auto<1>(onec, P[k].ym, K));
		den[k] = selg(P[k].ok, weaken<FM>(mulc(omy, P[k].x, K)), onem);                // (1 - y) x, non-zero when ok
	}
	FM d11;
	const FM dd = weaken<FM>(mul(den[0], den[1], K));
	const FM dinv = weaken<FM>(mul(sqr_n(pow_2_250m1(dd, &d11, K), 5, K), d11, K));    // dd^(p-2)
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const auto omy = carry(sub_auto<1>(onec, P[k].ym, K));
		const FM inv = weaken<FM>(mul(dinv, den[1 - k], K));                           // 1 / ((1 - y) x)
		const FM um = weaken<FM>(mulc(carry(add(onec, P[k].ym)), mulc(inv, P[k].x, K), K));
		const FM vm = weaken<FM>(mul(mul(digits9(A.g_alpha), um, K), mulc(inv, omy, K), K));
		const auto Xm = carry(add(um, digits9(A.g_A3)));
		bool good = P[k].ok;
		if (k == 0) {
			good = good & !small_order(Xm, vm, A.cof_dbl, K);
		}
		u8 *pd = (k == 0 ? A.pointsA : A.pointsR) + (size_t)i * 64;
		store_canon_be(pd, Xm, good, K);
		store_canon_be(pd + 32, vm, good, K);
		(k == 0 ? A.flagsA : A.flagsR)[i] = good ? 0 : ((k == 1 && P[1].neutral) ? 2 : 1);   // 2: R is the point at infinity
		if (k == 0 && A.edA != nullptr) {
			// the key on the Edwards curve itself (canonical digits), for k_ed_smul_c25519
			u32 buf[20];
			canonical_digits(buf, mulc(P[0].x, onec, K), K);
			canonical_digits(buf + 9, P[0].ym, K);
			buf[18] = buf[19] = 0;
			uint4 *dst = (uint4 *)(A.edA + (size_t)i * 20);
#pragma unroll
			for (int q = 0; q < 5; q++) {
				dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
			}
		}
	}
}

__global__ __launch_bounds__(64) void k_xdh_prep_c25519(EcamdXdhPrepArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	// scalar: reversed to big-endian and clamped (decode_scalar)
	{
		const u8 *ks = A.k + (size_t)i * 32;
		u8 *kd = A.scalars + (size_t)i * 32;
		for (int b = 0; b < 32; b++) {
			u8 v = ks[b];
			if (b == 0) v &= 248;
			if (b == 31) v = (u8)((v & 127) | 64);
			kd[31 - b] = v;
		}
	}
	u32 uw[8];
	load_le256(A.u + (size_t)i * 32, uw);
	bool ok = true;
	{
		// u >= p is rejected (only the eight words are significant: the top bit counts)
		constexpr u32 pw[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
		u32 bw = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			const uint64_t x = (uint64_t)uw[w] - pw[w] - bw;
			bw = (u32)(x >> 63);
		}
		ok = bw != 0;
	}
	uw[7] &= ok ? 0xffffffffu : 0x7fffffffu;  // keep from_words inside its input class when rejected
	const auto ud = from_words<PB, 8>(uw);
	const FM um = weaken<FM>(mul(ud, onec, K));
	// w = u (u (u + A) + 1)
	const auto t1 = carry(add(um, digits9(A.g_A)));
	const auto t2 = carry(add(mulc(um, t1, K), onec));
	const FM w = weaken<FM>(mulc(um, t2, K));
	// candidate root c = w^((p + 3) / 8) = w * w^((p - 5) / 8)
	FM w11;
	const FM pw22523 = weaken<FM>(mul(sqr_n(pow_2_250m1(w, &w11, K), 2, K), w, K));
	const FM c = weaken<FM>(mul(pw22523, w, K));
	const FM c2 = sqr_n(c, 1, K);
	const bool root = eq(c2, w, K);
	const bool alt = eq_neg(c2, w, K);
	const FM v = selg(alt & !root, weaken<FM>(mul(c, digits9(A.g_sm1), K)), c);
	ok = ok & (root | alt);
	const auto xm = carry(add(um, digits9(A.g_A3)));
	ok = ok & !small_order(xm, v, A.cof_dbl, K);
	u8 *pd = A.points + (size_t)i * 64;
	store_canon_be(pd, xm, ok, K);
	store_canon_be(pd + 32, v, ok, K);
	A.flags[i] = ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// X25519: the x-only Montgomery ladder on v^2 = u^3 + A u^2 + u itself (RFC 7748 section 5), for inputs
// k_xdh_prep_c25519 accepted.  [k]Q's u coordinate is all the reference exposes (it computes [k]Q on the
// Weierstrass model and maps back, ecdh/x25519_448.c:268-276), so the ladder is observably identical:
// u' = X2 / Z2, with Z2 = 0 (the point at infinity) and u' = 0 rejected as the reference does.
// 255 steps of 5M + 4S + one multiplication by a24 = 121665; the divisions are shared by 8 items per lane.
// ------------------------------------------------------------------------------------------
#define XDH_REC_WORDS 20   /* X2 (9 limbs), Z2 (9 limbs), padding */
#define XDH_FIN_K 8

#ifndef X25519_OCC
#define X25519_OCC   /* A/B hook: -DX25519_OCC='__attribute__((amdgpu_waves_per_eu(5,5)))' (tools/build_variant.py) */
#endif
__global__ __launch_bounds__(64) X25519_OCC void k_x25519_ladder(EcamdXdhLadderArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.flags[i]) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	u32 uw[8], kw[8];
	load_le256(A.u + (size_t)i * 32, uw);
	load_be<8>(A.scalars + (size_t)i * 32, 32, kw);           // clamped by the prep kernel
	const FM x1 = weaken<FM>(mul(from_words<PB, 8>(uw), onec, K));
	FM x2 = weaken<FM>(onec), z2, x3 = x1, z3 = weaken<FM>(onec);
#pragma unroll
	for (int w = 0; w < 9; w++) {
		z2.l[w] = 0;
	}
	u32 swap = 0;
#pragma unroll 1
	for (int t = 254; t >= 0; t--) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			word = (w == (t >> 5)) ? kw[w] : word;
		}
		const u32 kt = (word >> (t & 31)) & 1u;
		swap ^= kt;
		{
			const FM tx = selg(swap != 0, x3, x2), tz = selg(swap != 0, z3, z2);
			x3 = selg(swap != 0, x2, x3);
			z3 = selg(swap != 0, z2, z3);
			x2 = tx;
			z2 = tz;
		}
		swap = kt;
		const auto a = carry(add(x2, z2));
		const auto b = carry(sub_auto<1>(x2, z2, K));
		const FM aa = weaken<FM>(sqrc(a, K));
		const FM bb = weaken<FM>(sqrc(b, K));
		const auto e = carry(sub_auto<1>(aa, bb, K));
		const auto c = carry(add(x3, z3));
		const auto d = carry(sub_auto<1>(x3, z3, K));
		const FM da = weaken<FM>(mulc(d, a, K));
		const FM cb = weaken<FM>(mulc(c, b, K));
		x3 = weaken<FM>(sqrc(carry(add(da, cb)), K));
		z3 = weaken<FM>(mulc(x1, sqrc(carry(sub_auto<1>(da, cb, K)), K), K));
		x2 = weaken<FM>(mul(aa, bb, K));
		z2 = weaken<FM>(mulc(e, carry(add(aa, mul_word<121665u>(e))), K));   // a24 e: nine MADs (round 3: a full product, 90)
	}
	{
		const FM tx = selg(swap != 0, x3, x2), tz = selg(swap != 0, z3, z2);
		x2 = tx;
		z2 = tz;
	}
	u32 buf[XDH_REC_WORDS];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = x2.l[w];
		buf[9 + w] = z2.l[w];
	}
	buf[18] = buf[19] = 0;
	uint4 *dst = (uint4 *)(A.rec + (size_t)i * XDH_REC_WORDS);
#pragma unroll
	for (int q = 0; q < XDH_REC_WORDS / 4; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

__global__ __launch_bounds__(64) void k_x25519_fin(EcamdXdhLadderArgs A, int gslot, u32 nthreads)
{
	using namespace c25519;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FM onem = weaken<FM>(onec);
	// prefix products of the Z2 of this lane's items (rejected items and Z2 = 0 take part with 1)
	FM pre[XDH_FIN_K];
	u32 live = 0;
	FM acc = onem;
#pragma unroll 1
	for (int j = 0; j < XDH_FIN_K; j++) {
		const u32 i = t + (u32)j * nthreads;
		pre[j] = acc;
		if (i >= A.n || A.flags[i]) {
			continue;
		}
		const uint4 *src = (const uint4 *)(A.rec + (size_t)i * XDH_REC_WORDS);
		u32 buf[XDH_REC_WORDS];
#pragma unroll
		for (int q = 0; q < XDH_REC_WORDS / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
		FM z;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			z.l[w] = buf[9 + w];
		}
		if (!is_zero_mulout(z, K)) {
			live |= 1u << j;
			acc = weaken<FM>(mul(acc, z, K));
		}
	}
	FM a11;
	FM inv = weaken<FM>(mul(sqr_n(pow_2_250m1(acc, &a11, K), 5, K), a11, K));  // (product)^(p-2)
#pragma unroll 1
	for (int j = XDH_FIN_K - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			continue;
		}
		u8 *out = A.out + (size_t)i * 32;
		bool ok = ((live >> j) & 1u) != 0;
		u32 w8[8];
#pragma unroll
		for (int w = 0; w < 8; w++) {
			w8[w] = 0;
		}
		if (ok) {
			const uint4 *src = (const uint4 *)(A.rec + (size_t)i * XDH_REC_WORDS);
			u32 buf[XDH_REC_WORDS];
#pragma unroll
			for (int q = 0; q < XDH_REC_WORDS / 4; q++) {
				const uint4 v = src[q];
				buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
			}
			FM x, z;
#pragma unroll
			for (int w = 0; w < 9; w++) {
				x.l[w] = buf[w];
				z.l[w] = buf[9 + w];
			}
			const FM zi = weaken<FM>(mul(inv, pre[j], K));
			inv = weaken<FM>(mul(inv, z, K));
			u32 d[9];
			canonical_digits(d, mul(x, zi, K), K);
			to_words<9, 8>(w8, d);
			u32 nz = 0;
#pragma unroll
			for (int w = 0; w < 8; w++) {
				nz |= w8[w];
			}
			ok = nz != 0;   // an all-zero output is rejected (x25519_448.c:275-276)
		}
#pragma unroll
		for (int w = 0; w < 8; w++) {
			const u32 v = ok ? w8[w] : 0u;
			out[4 * w] = (u8)v;
			out[4 * w + 1] = (u8)(v >> 8);
			out[4 * w + 2] = (u8)(v >> 16);
			out[4 * w + 3] = (u8)(v >> 24);
		}
		A.status[i] = ok ? 0 : 1;
	}
}

hipError_t ecamd_launch_x25519_ladder(const EcamdXdhLadderArgs &a, int gslot, hipStream_t s, hipEvent_t *dom)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (dom) {
		(void)hipEventRecord(dom[0], s);
	}
	hipLaunchKernelGGL(k_x25519_ladder, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[1], s);
	}
	const uint32_t nthreads = (a.n + XDH_FIN_K - 1) / XDH_FIN_K;
	hipLaunchKernelGGL(k_x25519_fin, dim3((nthreads + 63) / 64), dim3(64), 0, s, a, gslot, nthreads);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Ed25519: [h]A on the twisted Edwards curve -x^2 + y^2 = 1 + d x^2 y^2 itself, in extended coordinates
// (Hisil-Wong-Carter-Dawson): doubling 4M + 4S (3M + 4S when no addition follows), addition 8M against a
// precomputed (Y-X, Y+X, 2dT, 2Z) entry, both COMPLETE for a = -1 and non-square d -- no exceptional pairs
// and no special case for the neutral element.  The reference computes the same group element on the
// Weierstrass model (prj_pt_mul, which cannot fail for a decoded key); the result is mapped to that model
// (k_ed_hA_fin: one inversion per 8 items) and the reference-exact tail (k_ed_fin: complete additions with
// their exceptional-pair rejections, cofactor doublings, infinity test) runs on it unchanged.
// Per window 13M + 16S + 8M = 2828 MADs against 4144 on the Weierstrass model (generic a).
// ------------------------------------------------------------------------------------------
#define EDT_ENT_WORDS 40                 /* ymx, ypx, t2d, z2: 4 x 9 limbs, padded */
#define EDT_ITEM_WORDS (8 * EDT_ENT_WORDS)
#define EDR_REC_WORDS 28                 /* X, Y, Z: 3 x 9 limbs, padded */

namespace c25519 {
struct Ext {
	FM X, Y, Z, T;
};
// a precomputed point (Y - X, Y + X, 2d T, 2 Z).  Round 4: the three sums are stored as they come out of one carry pass -- class FT, the
// loosest of the three -- instead of being multiplied by one into the class of a product (3 of the 4 multiplications of an entry, a
// third of a window table's cost); the additions take an FT operand as they are (the bounds are checked at compile time as everywhere).
typedef decltype(carry(sub_auto<1>(FM(), FM(), *(const CK *)nullptr))) PreS;
typedef decltype(carry(add(FM(), FM()))) PreA_;
typedef decltype(carry(mul_small<2>(FM()))) PreZ;
typedef E<PB, cmax(cmax(PreS::LB, PreA_::LB), PreZ::LB), cmax(cmax(PreS::TB, PreA_::TB), PreZ::TB), cmax(cmax(PreS::VB, PreA_::VB), PreZ::VB)> FT;
struct Pre {
	FT ymx, ypx;
	FM t2d;
	FT z2;
};
#define M_(a, b) weaken<FM>(mulc(a, b, K))
#define S_(a) weaken<FM>(sqrc(a, K))

template <bool WITH_T> static __device__ __forceinline__ Ext ed_dbl(const Ext &P, const CK &K)
{
	const FM a = S_(P.X), b = S_(P.Y);
	const auto c = mul_small<2>(S_(P.Z));
	const auto apb = add(a, b);
	const auto e = carry(sub_auto<1>(S_(carry(add(P.X, P.Y))), apb, K));      // 2XY
	const auto g = carry(sub_auto<1>(b, a, K));                                // B - A
	const auto f = carry(sub_auto<1>(g, c, K));                                // G - C
	E<PB, 0, 0, 0> zero;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		zero.l[w] = 0;
	}
	const auto h = carry(sub_auto<1>(zero, apb, K));                           // -A - B
	Ext R;
	R.X = M_(e, f);
	R.Y = M_(g, h);
	R.Z = M_(f, g);
	if (WITH_T) {
		R.T = M_(e, h);
	} else {
		R.T = P.T;  // not used by a following doubling
	}
	return R;
}

// P + (+-Q) for the precomputed entry Q (negated when neg); WITH_T = false when a doubling follows (it does not read T): 7 M
template <bool WITH_T = true> static __device__ __forceinline__ Ext ed_add(const Ext &P, const Pre &Q, bool neg, const CK &K)
{
	const FT qa = selg(neg, Q.ypx, Q.ymx), qb = selg(neg, Q.ymx, Q.ypx);
	const FM a = M_(carry(sub_auto<1>(P.Y, P.X, K)), qa);
	const FM b = M_(carry(add(P.Y, P.X)), qb);
	const FM c = M_(P.T, Q.t2d);
	const FM d = M_(P.Z, Q.z2);
	const auto e = carry(sub_auto<1>(b, a, K));
	const auto hh = carry(add(b, a));
	const auto dmc = carry(sub_auto<1>(d, c, K));
	const auto dpc = carry(add(d, c));
	typedef decltype(dmc) TS;
	typedef decltype(dpc) TA;
	typedef E<PB, cmax(TS::LB, TA::LB), cmax(TS::TB, TA::TB), cmax(TS::VB, TA::VB)> TU;
	const TU f = selg(neg, weaken<TU>(dpc), weaken<TU>(dmc));
	const TU g = selg(neg, weaken<TU>(dmc), weaken<TU>(dpc));
	Ext R;
	R.X = M_(e, f);
	R.Y = M_(g, hh);
	if (WITH_T) {
		R.T = M_(e, hh);
	} else {
		R.T = P.T;
	}
	R.Z = M_(f, g);
	return R;
}

static __device__ __forceinline__ Pre ed_pre(const Ext &P, const FC &d2, const CK &K)
{
	Pre Q;
	Q.ymx = weaken<FT>(carry(sub_auto<1>(P.Y, P.X, K)));
	Q.ypx = weaken<FT>(carry(add(P.Y, P.X)));
	Q.t2d = M_(P.T, d2);
	Q.z2 = weaken<FT>(carry(mul_small<2>(P.Z)));
	return Q;
}

static __device__ __forceinline__ void pre_store(u32 *base, int e, const Pre &Q)
{
	u32 buf[EDT_ENT_WORDS];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = Q.ymx.l[w];
		buf[9 + w] = Q.ypx.l[w];
		buf[18 + w] = Q.t2d.l[w];
		buf[27 + w] = Q.z2.l[w];
	}
#pragma unroll
	for (int w = 36; w < EDT_ENT_WORDS; w++) {
		buf[w] = 0;
	}
	uint4 *dst = (uint4 *)(base + (size_t)e * EDT_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < 9; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}
static __device__ __forceinline__ Pre pre_load(const u32 *base, u32 e)
{
	u32 buf[36];
	const uint4 *src = (const uint4 *)(base + (size_t)e * EDT_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < 9; q++) {
		const uint4 v = src[q];
		buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
	}
	Pre Q;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		Q.ymx.l[w] = buf[w];
		Q.ypx.l[w] = buf[9 + w];
		Q.t2d.l[w] = buf[18 + w];
		Q.z2.l[w] = buf[27 + w];
	}
	return Q;
}
// window table [1..8]P of the signed w = 4 recoding
// (order chosen so that at most two multiples are alive at a time beside the entry of P)
template <class Store> static __device__ __forceinline__ void ed_table_with(const Store &st, const Ext &P1, const FC &d2, const CK &K)
{
	const Pre Q1 = ed_pre(P1, d2, K);
	st(0, Q1);
#if !defined(ED_TABLE_UNROLLED)
	// [e]P = [e - 1]P + P, seven times ONE addition body (the unified formulas are complete, so the first step may be P + P): 56 M instead of
	// the 16 M + 16 S + 24 M of the doubling / addition mix below, but 11 KB of code run seven times instead of 80 KB run once -- the
	// spelled-out form ran at a quarter of the MAD rate (A/B: -DED_TABLE_UNROLLED, tools/build_variant.py)
	Ext Pe = P1;
#pragma unroll 1
	for (int e = 1; e < 8; e++) {
		Pe = ed_add(Pe, Q1, false, K);
		st(e, ed_pre(Pe, d2, K));
	}
	return;
#endif
	const Ext P2 = ed_dbl<true>(P1, K);
	st(1, ed_pre(P2, d2, K));
	Ext Pa = ed_add(P2, Q1, false, K);            // 3P
	st(2, ed_pre(Pa, d2, K));
	Pa = ed_dbl<true>(Pa, K);                     // 6P
	st(5, ed_pre(Pa, d2, K));
	Pa = ed_add(Pa, Q1, false, K);                // 7P
	st(6, ed_pre(Pa, d2, K));
	Ext Pb = ed_dbl<true>(P2, K);                 // 4P
	st(3, ed_pre(Pb, d2, K));
	Pa = ed_add(Pb, Q1, false, K);                // 5P
	st(4, ed_pre(Pa, d2, K));
	Pb = ed_dbl<true>(Pb, K);                     // 8P
	st(7, ed_pre(Pb, d2, K));
}
static __device__ __forceinline__ void ed_table(u32 *tb, const Ext &P1, const FC &d2, const CK &K)
{
	ed_table_with([&](int e, const Pre &Q) { pre_store(tb, e, Q); }, P1, d2, K);
}
// The same entry of all 64 items of a wave through LDS: a lane's own store is nine 16-byte pieces 2560 bytes apart from its
// neighbours' (every store instruction of the wave touches 64 cache lines); transposed, consecutive lanes write consecutive quads
// of one item's entry (144-byte runs).  sh: 64 x 37 words + 64 flags.  Called by every lane of the block (one wave).
#define EDT_LDS_STRIDE 37
static __device__ __forceinline__ void pre_store_wave(u32 *sh, u32 *wave_base, u32 item_words, int e, const Pre &Q, bool live)
{
	const u32 lane = threadIdx.x;
	u32 *row = sh + lane * EDT_LDS_STRIDE;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		row[w] = Q.ymx.l[w];
		row[9 + w] = Q.ypx.l[w];
		row[18 + w] = Q.t2d.l[w];
		row[27 + w] = Q.z2.l[w];
	}
	sh[64 * EDT_LDS_STRIDE + lane] = live ? 1u : 0u;
	__syncthreads();
#pragma unroll
	for (int it = 0; it < 9; it++) {
		const u32 g = (u32)it * 64u + lane;
		const u32 item = g / 9u, q = g - 9u * item;
		if (sh[64 * EDT_LDS_STRIDE + item]) {
			const u32 *src = sh + item * EDT_LDS_STRIDE + 4 * q;
			*(uint4 *)(wave_base + (size_t)item * item_words + (size_t)e * EDT_ENT_WORDS + 4 * q) = make_uint4(src[0], src[1], src[2], src[3]);
		}
	}
	__syncthreads();
}
static __device__ __forceinline__ Ext ed_neutral(const CK &K)
{
	Ext N;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		N.X.l[w] = 0;
		N.T.l[w] = 0;
	}
	N.Y = weaken<FM>(constant<FC>(K.one));
	N.Z = N.Y;
	return N;
}
// affine (x, y) -> extended
template <class AX> static __device__ __forceinline__ Ext ed_from_affine(const AX &x, const FM &y, const CK &K)
{
	Ext P;
	P.X = M_(x, constant<FC>(K.one));
	P.Y = y;
	P.Z = weaken<FM>(constant<FC>(K.one));
	P.T = M_(P.X, y);
	return P;
}
static __device__ __forceinline__ void ext_store(u32 *dst, const Ext &P)
{
	u32 buf[36];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = P.X.l[w];
		buf[9 + w] = P.Y.l[w];
		buf[18 + w] = P.Z.l[w];
		buf[27 + w] = P.T.l[w];
	}
	uint4 *d = (uint4 *)dst;
#pragma unroll
	for (int q = 0; q < 9; q++) {
		d[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}
static __device__ __forceinline__ Ext ext_load(const u32 *src)
{
	u32 buf[36];
	const uint4 *p = (const uint4 *)src;
#pragma unroll
	for (int q = 0; q < 9; q++) {
		const uint4 v = p[q];
		buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
	}
	Ext P;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		P.X.l[w] = buf[w];
		P.Y.l[w] = buf[9 + w];
		P.Z.l[w] = buf[18 + w];
		P.T.l[w] = buf[27 + w];
	}
	return P;
}
// acc += [nib - 8]P for the table tb of P (nib: one nibble of the recoded scalar)
static __device__ __forceinline__ void ed_window_add(Ext &acc, const u32 *tb, u32 nib, const CK &K)
{
	const int dig = (int)nib - 8;
	const u32 mag = (u32)(dig < 0 ? -dig : dig);
	const Pre Q = pre_load(tb, mag ? mag - 1 : 0);
	const Ext S = ed_add(acc, Q, dig < 0, K);
	const bool keep = (mag == 0);
	acc.X = selg(keep, acc.X, S.X);
	acc.Y = selg(keep, acc.Y, S.Y);
	acc.Z = selg(keep, acc.Z, S.Z);
	acc.T = selg(keep, acc.T, S.T);
}
// ---- round 4: the whole tail of an Ed25519 verification on the Edwards curve (k_ed_tail_c25519) ----
// affine precomputed entry (the comb table of B, a decoded R): y - x, y + x, 2d x y
struct PreA {
	FM ymx, ypx, t2d;
};
// P + (+-Q) for an affine precomputed Q (Z2 = 1, so D = 2 Z1 costs no multiplication): 7 M, 6 M without T
template <bool WITH_T> static __device__ __forceinline__ Ext ed_madd(const Ext &P, const PreA &Q, bool neg, const CK &K)
{
	const FM qa = selg(neg, Q.ypx, Q.ymx), qb = selg(neg, Q.ymx, Q.ypx);
	const FM a = M_(carry(sub_auto<1>(P.Y, P.X, K)), qa);
	const FM b = M_(carry(add(P.Y, P.X)), qb);
	const FM c = M_(P.T, Q.t2d);
	const auto d = carry(mul_small<2>(P.Z));
	const auto e = carry(sub_auto<1>(b, a, K));
	const auto hh = carry(add(b, a));
	const auto dmc = carry(sub_auto<1>(d, c, K));
	const auto dpc = carry(add(d, c));
	typedef decltype(dmc) TS;
	typedef decltype(dpc) TA;
	typedef E<PB, cmax(TS::LB, TA::LB), cmax(TS::TB, TA::TB), cmax(TS::VB, TA::VB)> TU;
	const TU f = selg(neg, weaken<TU>(dpc), weaken<TU>(dmc));
	const TU g = selg(neg, weaken<TU>(dmc), weaken<TU>(dpc));
	Ext R;
	R.X = M_(e, f);
	R.Y = M_(g, hh);
	if (WITH_T) {
		R.T = M_(e, hh);
	} else {
		R.T = P.T;
	}
	R.Z = M_(f, g);
	return R;
}
// (x, y) affine -> precomputed entry
template <class AX, class AY> static __device__ __forceinline__ PreA ed_prea(const AX &x, const AY &y, const FC &d2, const CK &K)
{
	const FC onec = constant<FC>(K.one);
	PreA Q;
	Q.ymx = M_(carry(sub_auto<1>(y, x, K)), onec);
	Q.ypx = M_(carry(add(y, x)), onec);
	Q.t2d = M_(M_(x, y), d2);
	return Q;
}
#define EDC_ENT_WORDS 32          /* one comb entry: ymx, ypx, t2d (canonical digits) + padding: 128 bytes, one cache line */
#define EDC_PER_WIN 32768
#define EDC_NWIN 16
static __device__ __forceinline__ PreA prea_load(const u32 *ent)
{
	u32 buf[28];
	const uint4 *src = (const uint4 *)ent;
#pragma unroll
	for (int q = 0; q < 7; q++) {
		const uint4 v = src[q];
		buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
	}
	PreA Q;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		Q.ymx.l[w] = buf[w];
		Q.ypx.l[w] = buf[9 + w];
		Q.t2d.l[w] = buf[18 + w];
	}
	return Q;
}
#undef M_
#undef S_
}  // namespace c25519

// Decoding for the Edwards [h]A path: A and R are decoded as in k_ed_decode_c25519 but stay on the Edwards curve.  The key's
// [cofactor]A = infinity test runs there too (three complete doublings; the map is a group isomorphism), and R's map to the
// Weierstrass model -- the only reason k_ed_decode_c25519 needs a field inversion per item -- moves to k_ed_hA_fin, where it
// shares one inversion with the [h]A of eight items.
__global__ __launch_bounds__(64) void k_ed_decode_ed_c25519(EcamdEdDecodeArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		const DecXY P = decode_xy(A, k == 0 ? A.encA + (size_t)i * A.strideA : A.encR + (size_t)i * A.strideR, K);
		bool good = P.ok;
		Ext P1 = ed_neutral(K);
		if (P.ok) {
			P1 = ed_from_affine(P.x, P.ym, K);
		}
		if (k == 0 && good) {
			Ext Q = P1;
			for (u32 r = 0; r < A.cof_dbl; r++) {
				Q = ed_dbl<false>(Q, K);
			}
			good = !is_zero_mulout(Q.X, K);   // [cofactor]A is the neutral element: rejected (sig/eddsa.c:2463-2472)
		}
		u32 buf[20];
#pragma unroll
		for (int w = 0; w < 9; w++) {
			buf[w] = P1.X.l[w];
			buf[9 + w] = P1.Y.l[w];
		}
		buf[18] = buf[19] = 0;
		uint4 *dst = (uint4 *)((k == 0 ? A.edA : A.edR) + (size_t)i * 20);
#pragma unroll
		for (int q = 0; q < 5; q++) {
			dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
		}
		if (k == 0) {
			A.flagsA[i] = good ? 0 : 1;
		} else {
			A.flagsR[i] = good ? 0 : (P.neutral ? 2 : 1);   // 2: R is the point at infinity of the Weierstrass model
		}
	}
}

// [h]A per lane: table [1..8]A, signed window w = 4; the extended result goes to rec
// phase 0: the table, phase 1: the window loop -- two launches, so that the table construction's register needs (256 VGPRs)
// do not bound the loop's occupancy (113 VGPRs, four waves per SIMD): Ed25519 verification 51.9 -> 57.4 M/s
#ifndef ED_SMUL_WAVES
#define ED_SMUL_WAVES 1   /* A/B hook: minimum waves per SIMD of the window loop (phase 1), -DED_SMUL_WAVES=4 (tools/build_variant.py) */
#endif
template <int phase> __global__ __launch_bounds__(64, phase == 1 ? ED_SMUL_WAVES : 1) void k_ed_smul_c25519(EcamdEdSmulArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.flags[i]) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FM onem = weaken<FM>(onec);
	const FC d2 = digits9(A.g_2d);
	u32 *tb = A.tbl + (size_t)i * EDT_ITEM_WORDS;
	Ext P1;
	{
		u32 buf[20];
		const uint4 *src = (const uint4 *)(A.edA + (size_t)i * 20);
#pragma unroll
		for (int q = 0; q < 5; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
#pragma unroll
		for (int w = 0; w < 9; w++) {
			P1.X.l[w] = buf[w];
			P1.Y.l[w] = buf[9 + w];
		}
		P1.Z = onem;
		P1.T = weaken<FM>(mul(P1.X, P1.Y, K));
	}
	if (phase == 0) {
		ed_table(tb, P1, d2, K);
		return;
	}
	// scalar: 32 bytes big-endian, k' = k + 0x88..8, top digit = the carry (0 / +1)
	u32 kw[8];
	load_be<8>(A.scalars + (size_t)i * 32, 32, kw);
	u32 carry_bit;
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			c += (uint64_t)kw[w] + 0x88888888u;
			kw[w] = (u32)c;
			c >>= 32;
		}
		carry_bit = (u32)c;
	}
	Ext acc;
	{
		FM zero;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			zero.l[w] = 0;
		}
		// neutral element (0, 1), or A when the top digit is 1
		acc.X = selg(carry_bit != 0, P1.X, zero);
		acc.Y = selg(carry_bit != 0, P1.Y, onem);
		acc.Z = onem;
		acc.T = selg(carry_bit != 0, P1.T, zero);
	}
#pragma unroll 1
	for (int t = 0; t < 64; t++) {
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<true>(acc, K);
		const int dig = (int)(kw[7] >> 28) - 8;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		const Pre Q = pre_load(tb, mag ? mag - 1 : 0);   // (issuing it before the doublings was measured: no gain)
#pragma unroll
		for (int w = 7; w > 0; w--) {
			kw[w] = (kw[w] << 4) | (kw[w - 1] >> 28);
		}
		kw[0] <<= 4;
		const Ext S = ed_add<false>(acc, Q, dig < 0, K);   // four doublings follow, or the end: T is not read again
		const bool keep = (mag == 0);
		acc.X = selg(keep, acc.X, S.X);
		acc.Y = selg(keep, acc.Y, S.Y);
		acc.Z = selg(keep, acc.Z, S.Z);
	}
	u32 buf[EDR_REC_WORDS];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = acc.X.l[w];
		buf[9 + w] = acc.Y.l[w];
		buf[18 + w] = acc.Z.l[w];
	}
	buf[27] = 0;
	uint4 *dst = (uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
	for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

// extended Edwards (X : Y : Z) -> the Weierstrass model, affine big-endian + status, 8 items per inversion:
//   u = (Z + Y) / (Z - Y), v = alpha u Z / X, (x, y) = (u + A/3, v); with w = ((Z - Y) X)^-1:
//   u = (Z + Y) X w, v = alpha (Z + Y) Z w.   X = 0: the neutral element (Y = Z: infinity, status 2) or the
//   point of order two (Y = -Z: (A/3, 0)).
#define EDF_K 8
// (x, y) of a decoded R (k_ed_decode_ed_c25519's record)
static __device__ __forceinline__ void edr_load(const u32 *rec, c25519::FM &x, c25519::FM &y)
{
	u32 buf[20];
	const uint4 *src = (const uint4 *)rec;
#pragma unroll
	for (int q = 0; q < 5; q++) {
		const uint4 v = src[q];
		buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
	}
#pragma unroll
	for (int w = 0; w < 9; w++) {
		x.l[w] = buf[w];
		y.l[w] = buf[9 + w];
	}
}

__global__ __launch_bounds__(64) void k_ed_hA_fin(EcamdEdSmulArgs A, int gslot, u32 nthreads)
{
	using namespace c25519;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FM onem = weaken<FM>(onec);
	// element 2j: [h]A of item j, element 2j + 1: its R (when this kernel maps it); pre[e] = the product of the live
	// denominators before element e
	FM pre[2 * EDF_K];
	u32 live = 0;
	FM acc = onem;
#pragma unroll 1
	for (int j = 0; j < EDF_K; j++) {
		const u32 i = t + (u32)j * nthreads;
		pre[2 * j] = acc;
		pre[2 * j + 1] = acc;
		if (i >= A.n || A.flags[i]) {
			continue;
		}
		u32 buf[EDR_REC_WORDS];
		const uint4 *src = (const uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
		for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
		FM X, Y, Z;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			X.l[w] = buf[w];
			Y.l[w] = buf[9 + w];
			Z.l[w] = buf[18 + w];
		}
		const FM den = weaken<FM>(mulc(carry(sub_auto<1>(Z, Y, K)), X, K));
		if (!is_zero_mulout(den, K)) {
			live |= 1u << (2 * j);
			acc = weaken<FM>(mul(acc, den, K));
		}
		pre[2 * j + 1] = acc;
		if (A.edR != nullptr && A.flagsR[i] == 0) {
			FM xr, yr;
			edr_load(A.edR + (size_t)i * 20, xr, yr);
			const FM denr = weaken<FM>(mulc(carry(sub_auto<1>(onec, yr, K)), xr, K));   // (1 - y) x, non-zero for a decoded R
			if (!is_zero_mulout(denr, K)) {
				live |= 1u << (2 * j + 1);
				acc = weaken<FM>(mul(acc, denr, K));
			}
		}
	}
	FM a11;
	FM inv = weaken<FM>(mul(sqr_n(pow_2_250m1(acc, &a11, K), 5, K), a11, K));
#pragma unroll 1
	for (int j = EDF_K - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			continue;
		}
		u8 *out = A.out + (size_t)i * 64;
		if (A.flags[i]) {
			for (int b = 0; b < 64; b++) {
				out[b] = 0;
			}
			A.status[i] = 1;
			continue;
		}
		if (A.edR != nullptr && A.flagsR[i] == 0) {
			// R: u = (1 + y) / (1 - y), v = alpha u / x; with w = ((1 - y) x)^-1: u = (1 + y) x w, v = alpha (1 + y) w
			u8 *outr = A.outR + (size_t)i * 64;
			if ((live >> (2 * j + 1)) & 1u) {
				FM xr, yr;
				edr_load(A.edR + (size_t)i * 20, xr, yr);
				const FM denr = weaken<FM>(mulc(carry(sub_auto<1>(onec, yr, K)), xr, K));
				const FM wr = weaken<FM>(mul(inv, pre[2 * j + 1], K));
				inv = weaken<FM>(mul(inv, denr, K));
				const FM opy = weaken<FM>(mulc(carry(add(onec, yr)), wr, K));          // (1 + y) w
				const FM um = weaken<FM>(mul(opy, xr, K));
				const FM vm = weaken<FM>(mul(digits9(A.g_alpha), opy, K));
				store_canon_be(outr, carry(add(um, digits9(A.g_A3))), true, K);
				store_canon_be(outr + 32, vm, true, K);
			} else {
				// cannot happen for an R the decoder accepted (x != 0, hence y != 1): reject it
				for (int b = 0; b < 64; b++) {
					outr[b] = 0;
				}
				A.flagsR[i] = 1;
			}
		}
		u32 buf[EDR_REC_WORDS];
		const uint4 *src = (const uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
		for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
		FM X, Y, Z;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			X.l[w] = buf[w];
			Y.l[w] = buf[9 + w];
			Z.l[w] = buf[18 + w];
		}
		if ((live >> (2 * j)) & 1u) {
			const FM den = weaken<FM>(mulc(carry(sub_auto<1>(Z, Y, K)), X, K));
			const FM w_ = weaken<FM>(mul(inv, pre[2 * j], K));
			inv = weaken<FM>(mul(inv, den, K));
			const FM zpy = weaken<FM>(mulc(carry(add(Z, Y)), w_, K));          // (Z + Y) w
			const FM um = weaken<FM>(mul(zpy, X, K));
			const FM vm = weaken<FM>(mul(mul(digits9(A.g_alpha), zpy, K), Z, K));
			store_canon_be(out, carry(add(um, digits9(A.g_A3))), true, K);
			store_canon_be(out + 32, vm, true, K);
			A.status[i] = 0;
		} else {
			// X = 0 or Y = Z (only together with X = 0 on the curve): neutral element or the point of order two
			const bool neutral = eq(Y, Z, K);
			FC a3 = digits9(A.g_A3);
			store_canon_be(out, a3, !neutral, K);
			for (int b = 32; b < 64; b++) {
				out[b] = 0;
			}
			A.status[i] = neutral ? 2 : 0;
		}
	}
}

// ------------------------------------------------------------------------------------------
// Ed25519 whole-batch verification (ec_verify_batch's EdDSA branch, sig/eddsa.c:2278-2545) as one multi-scalar
// multiplication on the Edwards curve: T = [q - sum z_i S_i]B + sum_i ([z_i h_i]A_i + [z_i]R_i), accepted when [8]T is the
// neutral element.  The formulas are complete, so no pair of inputs is exceptional.
//   k_edmsm_prep    per item: decode A_i and R_i (no Weierstrass map, no inversion), the reference's per-item rejections
//                   (decoding, [8]A_i = neutral), window tables [1..8]A_i and [1..8]R_i
//   k_edmsm_loop    per lane: the items j * L + lane, j < K, and the base point share 64 x 4 doublings (Straus); R_i only
//                   enters the last 33 windows (z_i has 128 bits)
//   k_edmsm_reduce  tree sum of the lane results, cofactor, verdict
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_edmsm_btable(EcamdEdMsmArgs A, u32 *tblB, int gslot)
{
	using namespace c25519;
	if (blockIdx.x != 0 || threadIdx.x != 0) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FM ym = weaken<FM>(mul(digits9(A.g_By), constant<FC>(K.one), K));
	ed_table(tblB, ed_from_affine(digits9(A.g_Bx), ym, K), digits9(A.g_2d), K);
}

// phase 0: decoding and the per-item rejections (the decoded point rests in the last slot of its table-to-be);
// phase 1: the two window tables -- two launches for the reason given at k_ed_smul_c25519
template <int phase> __global__ __launch_bounds__(64) void k_edmsm_prep(EcamdEdMsmArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	u32 *tb = A.tbl + (size_t)i * (2 * EDT_ITEM_WORDS);
	if (phase == 1) {
		const FC d2 = digits9(A.g_2d);
#pragma unroll 1
		for (int k = 0; k < 2; k++) {
			u32 *tk = tb + k * EDT_ITEM_WORDS;
			u32 buf[20];
			const uint4 *src = (const uint4 *)(tk + 7 * EDT_ENT_WORDS);
#pragma unroll
			for (int q = 0; q < 5; q++) {
				const uint4 v = src[q];
				buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
			}
			Ext P1;
#pragma unroll
			for (int w = 0; w < 9; w++) {
				P1.X.l[w] = buf[w];
				P1.Y.l[w] = buf[9 + w];
			}
			P1.Z = weaken<FM>(constant<FC>(K.one));
			P1.T = weaken<FM>(mul(P1.X, P1.Y, K));
			ed_table(tk, P1, d2, K);   // (its last store overwrites the record just read)
		}
		return;
	}
	EcamdEdDecodeArgs D;   // decode_xy reads the two constants only
#pragma unroll
	for (int w = 0; w < 9; w++) {
		D.g_d[w] = A.g_d[w];
		D.g_sm1[w] = A.g_sm1[w];
	}
	u8 flag = 0;
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		const DecXY P = decode_xy(D, k == 0 ? A.encA + (size_t)i * A.strideA : A.encR + (size_t)i * A.strideR, K);
		// R = (0, 1) is accepted by the reference (it becomes the point at infinity of the Weierstrass model)
		bool good = P.ok | (k == 1 && P.neutral);
		Ext P1 = ed_neutral(K);
		if (P.ok) {
			P1 = ed_from_affine(P.x, P.ym, K);
		}
		if (k == 0 && good) {
			// the reference rejects a key with [cofactor]A = infinity (sig/eddsa.c:2463-2472)
			Ext Q = P1;
			for (u32 r = 0; r < A.cof_dbl; r++) {
				Q = ed_dbl<false>(Q, K);
			}
			good = !is_zero_mulout(Q.X, K);
			if (!good) {
				P1 = ed_neutral(K);
			}
		}
		u32 buf[20];
#pragma unroll
		for (int w = 0; w < 9; w++) {
			buf[w] = P1.X.l[w];
			buf[9 + w] = P1.Y.l[w];
		}
		buf[18] = buf[19] = 0;
		uint4 *dst = (uint4 *)(tb + k * EDT_ITEM_WORDS + 7 * EDT_ENT_WORDS);
#pragma unroll
		for (int q = 0; q < 5; q++) {
			dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
		}
		flag |= good ? 0 : 1;
	}
	A.flags[i] = flag;
}

__global__ __launch_bounds__(64) void k_edmsm_loop(EcamdEdMsmArgs A, int gslot)
{
	using namespace c25519;
	const u32 lane = blockIdx.x * 64 + threadIdx.x;
	if (lane >= A.L) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	Ext acc = ed_neutral(K);
#pragma unroll 1
	for (int t = 0; t < 64; t++) {
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<true>(acc, K);
		const u32 wsel = 7u - ((u32)t >> 3);
		const u32 sh = 28u - 4u * ((u32)t & 7u);
#pragma unroll 1
		for (u32 j = 0; j < A.K; j++) {
			const u32 item = j * A.L + lane;
			if (item < A.n) {
				const u32 word = A.cA[(size_t)wsel * A.n + item];
				ed_window_add(acc, A.tbl + (size_t)item * (2 * EDT_ITEM_WORDS), (word >> sh) & 15u, K);
			}
		}
		{
			const u32 word = A.sB[(size_t)wsel * A.L + lane];
			ed_window_add(acc, A.tblB, (word >> sh) & 15u, K);
		}
		if (t >= 31) {   // z_i + 0x8 88..8 has 132 bits: windows 31..63
#pragma unroll 1
			for (u32 j = 0; j < A.K; j++) {
				const u32 item = j * A.L + lane;
				if (item < A.n) {
					const u32 word = A.zR[(size_t)wsel * A.n + item];
					ed_window_add(acc, A.tbl + (size_t)item * (2 * EDT_ITEM_WORDS) + EDT_ITEM_WORDS, (word >> sh) & 15u, K);
				}
			}
		}
	}
	ext_store(A.rec + (size_t)lane * ECAMD_EDM_REC_WORDS, acc);
}

#define EDM_FAN 16
// out[t] = in[t] + in[t + T] + in[t + 2T] + ...
__global__ __launch_bounds__(64) void k_edmsm_sum(EcamdEdMsmArgs A, const u32 *in, u32 count, u32 *out, u32 T, int gslot)
{
	using namespace c25519;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= T) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	Ext acc = ext_load(in + (size_t)t * ECAMD_EDM_REC_WORDS);
#pragma unroll 1
	for (u32 i = t + T; i < count; i += T) {
		const Ext P = ext_load(in + (size_t)i * ECAMD_EDM_REC_WORDS);
		acc = ed_add(acc, ed_pre(P, d2, K), false, K);
	}
	ext_store(out + (size_t)t * ECAMD_EDM_REC_WORDS, acc);
}
__global__ __launch_bounds__(64) void k_edmsm_final(EcamdEdMsmArgs A, const u32 *in, u32 count, const u32 *flagword, u8 *verdict,
						    u32 *sum_out, int gslot)
{
	using namespace c25519;
	if (blockIdx.x != 0 || threadIdx.x != 0) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	Ext acc = ext_load(in);
#pragma unroll 1
	for (u32 i = 1; i < count; i++) {
		const Ext P = ext_load(in + (size_t)i * ECAMD_EDM_REC_WORDS);
		acc = ed_add(acc, ed_pre(P, d2, K), false, K);
	}
	if (sum_out != nullptr) {
		ext_store(sum_out, acc);
	}
	for (u32 r = 0; r < A.cof_dbl; r++) {
		acc = ed_dbl<false>(acc, K);
	}
	// a point of the prime-order subgroup with X = 0 is the neutral element
	const bool neutral = is_zero_mulout(acc.X, K) & eq(acc.Y, acc.Z, K);
	if (!(neutral && flagword[0] == 0)) {
		verdict[0] = 1;   // the host cleared the byte; several pieces may share it
	}
}

hipError_t ecamd_launch_edmsm_btable(const EcamdEdMsmArgs &a, uint32_t *tblB, int gslot, hipStream_t s)
{
	hipLaunchKernelGGL(k_edmsm_btable, dim3(1), dim3(64), 0, s, a, tblB, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_edmsm_prep(const EcamdEdMsmArgs &a, int gslot, hipStream_t s)
{
	hipLaunchKernelGGL(k_edmsm_prep<0>, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	hipLaunchKernelGGL(k_edmsm_prep<1>, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_edmsm_loop(const EcamdEdMsmArgs &a, int gslot, hipStream_t s)
{
	hipLaunchKernelGGL(k_edmsm_loop, dim3((a.L + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_edmsm_reduce(const EcamdEdMsmArgs &a, uint32_t *tmp, const uint32_t *flagword, uint8_t *verdict,
				     uint32_t *sum_out, int gslot, hipStream_t s)
{
	// ping-pong: rec -> tmp -> rec ... (the second buffer needs ceil(L / EDM_FAN) records)
	const uint32_t *in = a.rec;
	uint32_t *bufs[2] = {tmp, a.rec};
	uint32_t count = a.L;
	int b = 0;
	while (count > 32) {
		const uint32_t T = (count + EDM_FAN - 1) / EDM_FAN;
		hipLaunchKernelGGL(k_edmsm_sum, dim3((T + 63) / 64), dim3(64), 0, s, a, in, count, bufs[b], T, gslot);
		in = bufs[b];
		b ^= 1;
		count = T;
	}
	hipLaunchKernelGGL(k_edmsm_final, dim3(1), dim3(64), 0, s, a, in, count, flagword, verdict, sum_out, gslot);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Round 6: the Ed25519 batch equation by buckets (EcamdEdBktArgs; the Schnorr-type form of the same idea: k_bkt_* above).  The unified
// Edwards addition is complete, so there is nothing to flag and nothing to compare: equal points double, opposite ones give the
// neutral element, which is an ordinary point.
//   k_edbkt_points  per item: A_i and R_i decoded (the reference's rejections as in k_edmsm_prep), as affine precomputed entries
//                   (y - x, y + x, 2d x y); the first LB lanes also write the LB copies of B
//   k_edbkt_accum   one lane per bucket: 7M per point (ed_madd)
//   k_edbkt_reduce  levels of 16 by running sums, the U-sums carried along (see k_bkt_reduce_g)
//   k_edbkt_window  per window C_0 + 16 (C_1 + ...), times 2^(16 win); k_edmsm_final adds the sixteen windows, clears the cofactor, decides
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ void edb_store(u32 *dst, const c25519::PreA &Q)
{
	u32 buf[ECAMD_EDB_PT_WORDS];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = Q.ymx.l[w];
		buf[9 + w] = Q.ypx.l[w];
		buf[18 + w] = Q.t2d.l[w];
	}
	buf[27] = 0;
	uint4 *d = (uint4 *)dst;
#pragma unroll
	for (int q = 0; q < ECAMD_EDB_PT_WORDS / 4; q++) {
		d[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}
__global__ __launch_bounds__(64) void k_edbkt_points(EcamdEdMsmArgs A, EcamdEdBktArgs B, int gslot)
{
	using namespace c25519;
	const u32 rel = blockIdx.x * 64 + threadIdx.x;
	if (rel >= (A.count ? A.count : A.n)) {
		return;
	}
	const u32 i = A.first + rel;
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	if (i < B.LB) {
		const FM ym = weaken<FM>(mul(digits9(A.g_By), constant<FC>(K.one), K));
		edb_store(B.pts + (size_t)(A.n + i) * ECAMD_EDB_PT_STRIDE, ed_prea(digits9(A.g_Bx), ym, d2, K));
	}
	EcamdEdDecodeArgs D;   // decode_xy reads the two constants only
#pragma unroll
	for (int w = 0; w < 9; w++) {
		D.g_d[w] = A.g_d[w];
		D.g_sm1[w] = A.g_sm1[w];
	}
	u8 flag = 0;
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		const DecXY P = decode_xy(D, k == 0 ? A.encA + (size_t)i * A.strideA : A.encR + (size_t)i * A.strideR, K);
		// R = (0, 1) is accepted by the reference (it becomes the point at infinity of the Weierstrass model)
		bool good = P.ok | (k == 1 && P.neutral);
		Ext P1 = ed_neutral(K);
		if (P.ok) {
			P1 = ed_from_affine(P.x, P.ym, K);
		}
		if (k == 0 && good) {
			// the reference rejects a key with [cofactor]A = infinity (sig/eddsa.c:2463-2472)
			Ext Q = P1;
			for (u32 r = 0; r < A.cof_dbl; r++) {
				Q = ed_dbl<false>(Q, K);
			}
			good = !is_zero_mulout(Q.X, K);
			if (!good) {
				P1 = ed_neutral(K);
			}
		}
		// P1 is affine (Z = 1): its precomputed entry straight from X and Y
		const u32 idx = k == 0 ? i : A.n + B.LB + i;
		edb_store(B.pts + (size_t)idx * ECAMD_EDB_PT_STRIDE, ed_prea(P1.X, P1.Y, d2, K));
		flag |= good ? 0 : 1;
	}
	A.flags[i] = flag;
}

__global__ __launch_bounds__(64) void k_edbkt_accum(EcamdEdBktArgs B, int gslot, u32 win_first, u32 win_count)
{
	using namespace c25519;
	// the windows [win_first, win_first + win_count); the top window (15) ahead of the others when the launch holds it
	const int tw = (win_first + win_count == 16u) ? (int)(15u - win_first) : -1;
	const u32 rel = bkt_block_order(blockIdx.x, gridDim.x, true, tw, 1024u) * 64 + threadIdx.x;
	if (rel >= (win_count << 16)) {
		return;
	}
	const u32 lane = (win_first << 16) + rel;
	const CK &K = TabGP<255>::get(gslot);
	const u32 t = B.perm[lane];
	// the top window: lane (15 << 16 | part * 8192 + d) takes the points part, part + 8, ... of bucket d (ecamd_internal.h: ECAMD_EDB_SPLIT)
	u32 tb = t, k0 = 0, step = 1;
	if ((t >> 16) == 15u) {
		const u32 d = t & 0xffffu;
		tb = (15u << 16) | (d & (ECAMD_EDB_SPLIT_DIGITS - 1u));
		k0 = d / ECAMD_EDB_SPLIT_DIGITS;
		step = ECAMD_EDB_SPLIT;
	}
	u32 cnt = (tb & 0xffffu) ? B.count[tb] : 0u, capb;
	const u32 *ord = B.order + ecamd_bkt_slot(tb, B.cap, B.cap_top, 15u, &capb);
	cnt = cnt < capb ? cnt : capb;
	Ext acc = ed_neutral(K);
	PreA nxt;
	u32 nidx = 0;
	if (k0 < cnt) {
		nxt = prea_load(B.pts + (size_t)ord[k0] * ECAMD_EDB_PT_STRIDE);
		nidx = k0 + step < cnt ? ord[k0 + step] : 0u;
	}
#pragma unroll 1
	for (u32 k = k0; k < cnt; k += step) {
		const PreA cur = nxt;
		if (k + step < cnt) {
			nxt = prea_load(B.pts + (size_t)nidx * ECAMD_EDB_PT_STRIDE);   // on its way while the current addition runs
			nidx = k + 2 * step < cnt ? ord[k + 2 * step] : 0u;
		}
		acc = ed_madd<true>(acc, cur, false, K);
	}
	ext_store(B.bsum + (size_t)t * ECAMD_EDM_REC_WORDS, acc);
}
// the top window's eight partial sums per bucket, added up; the seven borrowed records become neutral elements again (the reduction reads them)
__global__ __launch_bounds__(64) void k_edbkt_combine(EcamdEdMsmArgs A, EcamdEdBktArgs B, int gslot)
{
	using namespace c25519;
	const u32 dp = blockIdx.x * 64 + threadIdx.x;
	if (dp >= ECAMD_EDB_SPLIT_DIGITS) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	u32 *rec = B.bsum + ((size_t)(15u << 16) + dp) * ECAMD_EDM_REC_WORDS;
	Ext acc = ext_load(rec);
	const Ext zero = ed_neutral(K);
#pragma unroll 1
	for (u32 part = 1; part < ECAMD_EDB_SPLIT; part++) {
		u32 *r2 = rec + (size_t)part * ECAMD_EDB_SPLIT_DIGITS * ECAMD_EDM_REC_WORDS;
		acc = ed_add(acc, ed_pre(ext_load(r2), d2, K), false, K);
		ext_store(r2, zero);
	}
	ext_store(rec, acc);
}

struct EdBktLevel {
	const u32 *inT;
	const u32 *inC[BKT_MAXCARRY];
	u32 *outT, *outU;
	u32 *outC[BKT_MAXCARRY];
	u32 Lin, Lout, ncarry;
	u32 fold;
	u32 nwin;                       // windows in these arrays (a range of the sixteen)
};
__global__ __launch_bounds__(64) void k_edbkt_reduce(EcamdEdMsmArgs A, EdBktLevel V, int gslot)
{
	using namespace c25519;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= V.nwin * V.Lout) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	const u32 win = t / V.Lout, j = t - win * V.Lout;
	const u32 first = j * V.fold, len = (V.Lin - first) < V.fold ? (V.Lin - first) : V.fold;
	const u32 role = blockIdx.y;
	const u32 *in = (role == 0 ? V.inT : V.inC[role - 1]) + ((size_t)win * V.Lin + first) * ECAMD_EDM_REC_WORDS;
	Ext run = ed_neutral(K), acc = run;
	if (role == 0) {
#pragma unroll 1
		for (u32 r = len; r-- > 1;) {
			run = ed_add(run, ed_pre(ext_load(in + (size_t)r * ECAMD_EDM_REC_WORDS), d2, K), false, K);
			acc = ed_add(acc, ed_pre(run, d2, K), false, K);
		}
		run = ed_add(run, ed_pre(ext_load(in), d2, K), false, K);
		ext_store(V.outT + (size_t)t * ECAMD_EDM_REC_WORDS, run);
		ext_store(V.outU + (size_t)t * ECAMD_EDM_REC_WORDS, acc);
	} else {
#pragma unroll 1
		for (u32 r = 0; r < len; r++) {
			run = ed_add(run, ed_pre(ext_load(in + (size_t)r * ECAMD_EDM_REC_WORDS), d2, K), false, K);
		}
		ext_store(V.outC[role - 1] + (size_t)t * ECAMD_EDM_REC_WORDS, run);
	}
}
struct EdBktWindows {
	const u32 *U;
	const u32 *C[BKT_MAXCARRY];
	u32 *out;
	u32 ncarry, fold_log2;
	u32 nwin, win_base;             // record w of these arrays is window win_base + w (its weight: 2^(16 (win_base + w)))
};
__global__ __launch_bounds__(64) void k_edbkt_window(EcamdEdMsmArgs A, EdBktWindows V, int gslot)
{
	using namespace c25519;
	const u32 win = blockIdx.x * 64 + threadIdx.x;
	if (win >= V.nwin) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.g_2d);
	Ext acc = ext_load(V.U + (size_t)win * ECAMD_EDM_REC_WORDS);
#pragma unroll 1
	for (u32 k = V.ncarry; k-- > 0;) {
#pragma unroll 1
		for (u32 d = 0; d < V.fold_log2; d++) {
			acc = ed_dbl<true>(acc, K);
		}
		acc = ed_add(acc, ed_pre(ext_load(V.C[k] + (size_t)win * ECAMD_EDM_REC_WORDS), d2, K), false, K);
	}
#pragma unroll 1
	for (u32 d = 0; d < 16u * (V.win_base + win); d++) {
		acc = ed_dbl<true>(acc, K);
	}
	ext_store(V.out + (size_t)win * ECAMD_EDM_REC_WORDS, acc);
}
hipError_t ecamd_launch_edbkt(const EcamdEdMsmArgs &a, const EcamdEdBktArgs &b, int phase, const uint32_t *flagword, uint8_t *verdict, uint32_t *sum_out,
			      int gslot, hipStream_t s)
{
	constexpr size_t RECW = ECAMD_EDM_REC_WORDS;
	if (phase == 0) {
		hipLaunchKernelGGL(k_edbkt_points, dim3(((a.count ? a.count : a.n) + 63) / 64), dim3(64), 0, s, a, b, gslot);
	} else {
		// phase 1: every window summed; 2: reduced, weighted and compared.  By window range, for a caller that reduces the key-only windows (8 .. 15:
		// z_i has 128 bits) on a second stream while the others are still being summed -- their doubling chains, 16 w long, are the long ones --:
		// 10 / 11 the sums of the windows 8 .. 15 / 0 .. 7, 12 / 13 their reduction and weighting, 14 the total and the verdict.  The sixteen weighted
		// window sums rest in the spare tail of the reduction scratch.
		const uint32_t fold = ecamd_bkt_fold();
		const size_t halfw = (size_t)b.red_words / 2, per_win = (halfw - 18 * RECW) / 16;
		uint32_t *wout = b.red + (size_t)b.red_words - 18 * RECW;
		auto accum = [&](uint32_t wf, uint32_t wc) {
			hipLaunchKernelGGL(k_edbkt_accum, dim3((wc << 16) / 64), dim3(64), 0, s, b, gslot, wf, wc);
			if (wf + wc == 16u) {
				hipLaunchKernelGGL(k_edbkt_combine, dim3(ECAMD_EDB_SPLIT_DIGITS / 64), dim3(64), 0, s, a, b, gslot);
			}
		};
		auto reduce = [&](uint32_t wf, uint32_t wc) -> hipError_t {
			EdBktLevel V = {};
			V.fold = fold;
			V.nwin = wc;
			V.inT = b.bsum + ((size_t)wf << 16) * RECW;
			V.Lin = 1u << 16;
			uint32_t *half[2] = {b.red + (size_t)wf * per_win, b.red + halfw + (size_t)wf * per_win};
			int hsel = 0;
			while (V.Lin > 1) {
				V.Lout = (V.Lin + fold - 1) / fold;
				uint32_t *o = half[hsel];
				const size_t arr = (size_t)wc * V.Lout * RECW;
				if ((2 + (size_t)V.ncarry) * arr > (size_t)wc * per_win || V.ncarry + 1 > BKT_MAXCARRY) {
					return hipErrorInvalidValue;
				}
				V.outT = o;
				V.outU = o + arr;
				for (uint32_t k = 0; k < V.ncarry; k++) {
					V.outC[k] = o + (2 + (size_t)k) * arr;
				}
				hipLaunchKernelGGL(k_edbkt_reduce, dim3((wc * V.Lout + 63) / 64, 1 + V.ncarry), dim3(64), 0, s, a, V, gslot);
				V.inT = V.outT;
				for (uint32_t k = 0; k < V.ncarry; k++) {
					V.inC[k] = V.outC[k];
				}
				V.inC[V.ncarry] = V.outU;
				V.ncarry++;
				V.Lin = V.Lout;
				hsel ^= 1;
			}
			EdBktWindows W = {};
			W.U = V.inC[V.ncarry - 1];
			W.ncarry = V.ncarry - 1;
			W.fold_log2 = (uint32_t)__builtin_ctz(fold);
			W.nwin = wc;
			W.win_base = wf;
			for (uint32_t k = 0; k + 1 < V.ncarry; k++) {
				W.C[k] = V.inC[k];
			}
			W.out = wout + (size_t)wf * RECW;
			hipLaunchKernelGGL(k_edbkt_window, dim3(1), dim3(64), 0, s, a, W, gslot);
			return hipSuccess;
		};
		hipError_t e = hipSuccess;
		switch (phase) {
		case 1: accum(0, 16); break;
		case 10: accum(8, 8); break;
		case 11: accum(0, 8); break;
		case 12: e = reduce(8, 8); break;
		case 13: e = reduce(0, 8); break;
		case 2: e = reduce(0, 16);   // and the total
			// fall through
		case 14:
			if (e == hipSuccess) {
				hipLaunchKernelGGL(k_edmsm_final, dim3(1), dim3(64), 0, s, a, (const uint32_t *)wout, 16u, flagword, verdict, sum_out, gslot);
			}
			break;
		default: return hipErrorInvalidValue;
		}
		if (e != hipSuccess) {
			return e;
		}
	}
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Round 4: the Edwards comb table of the base point and the tail of an Ed25519 verification on the Edwards curve.
//
// k_edcomb_build_c25519: the affine Weierstrass points [m 2^(16 j)]G the engine computed for the comb table of the generator
// (maybe_build_comb's batch) -> the same multiples of B on edwards25519, as precomputed entries (y - x, y + x, 2d x y):
// x = alpha u / v, y = (u - 1) / (u + 1) with u = X - A/3, v = Y (the inverse of the map in k_ed_hA_fin); one inversion per entry,
// a one-time cost per curve handle.  Multiples of B have odd prime order: v (u + 1) != 0.
//
// k_ed_tail_c25519: [S]B by seventeen mixed additions from that table, then the reference's tail (sig/eddsa.c:2225-2243)
//     W1 = [S]G - R,   W2 = W1 - [h]A,   [8]W2 == infinity
// on the Edwards curve: the map to the Weierstrass model is a group isomorphism defined on every point, so the sums are the same
// points and [8]W2 is the point at infinity exactly when it is the neutral element (0 : Z : Z) here.  What the map does not carry
// over is prj_pt_add's failure (curves/prj_pt.c:1058-1060, the complete formulas of RCB15 Alg. 1 on a curve of even order):
// the call returns -1, and the reference rejects the signature, exactly when the DIFFERENCE of the two summands is the point of
// order two (tests/test_ed_tail_model.py checks that characterisation against the formulas of the oracle on every torsion coset).
// Restated on this curve, with T2 = (0, -1) = -T2:
//     [S]G - (-R) = T2   <=>  [S]B = (x_R, -y_R)                      (E1)
//     W1 - (-[h]A) = T2  <=>  W1 = (x_hA, -y_hA)                      (E2)
// -- two projective comparisons; a rejection by either overrides the equation, as the -1 does in the reference.  R decoded to the
// neutral element (flagsR = 2: the reference's point at infinity) enters as (0, 1).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_edcomb_build_c25519(const u8 *pts, u32 n, u32 *table, EcamdEdTailConsts Cst, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	u32 xw[8], yw[8];
	load_be<8>(pts + (size_t)i * 64, 32, xw);
	load_be<8>(pts + (size_t)i * 64 + 32, 32, yw);
	const FM X = weaken<FM>(mul(from_words<PB, 8>(xw), constant<FC>(K.ix), K));
	const FM v = weaken<FM>(mul(from_words<PB, 8>(yw), constant<FC>(K.iy), K));
	const auto u = carry(sub_auto<1>(X, digits9(Cst.g_A3), K));
	const auto up1 = carry(add(u, onec));
	const auto um1 = carry(sub_auto<1>(u, onec, K));
	const FM den = weaken<FM>(mulc(v, up1, K));
	FM a11;
	const FM inv = weaken<FM>(mul(sqr_n(pow_2_250m1(den, &a11, K), 5, K), a11, K));   // den^(p - 2)
	const FM x = weaken<FM>(mul(mul(mulc(mulc(digits9(Cst.g_alpha), u, K), up1, K), inv, K), onec, K));
	const FM y = weaken<FM>(mul(mulc(mulc(um1, v, K), inv, K), onec, K));
	const PreA Q = ed_prea(x, y, digits9(Cst.g_2d), K);
	u32 buf[EDC_ENT_WORDS];
	canonical_digits(buf, Q.ymx, K);
	canonical_digits(buf + 9, Q.ypx, K);
	canonical_digits(buf + 18, Q.t2d, K);
#pragma unroll
	for (int w = 27; w < EDC_ENT_WORDS; w++) {
		buf[w] = 0;
	}
	uint4 *dst = (uint4 *)(table + (size_t)i * EDC_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < EDC_ENT_WORDS / 4; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

// Round 6: eddsa_encode_point of a Weierstrass point (eddsa_export_pub_key / the R of a signature, sig/eddsa.c:795-860: prj_pt_shortw_to_aff_pt_edwards
// then y little-endian with the parity of x in bit 255) on this unit -- k_ed_sign_enc<8> of ecamd_kernels.hip (saturated words, which stays the
// reference implementation and serves handles without this unit), operation for operation: x = alpha u / v, y = (u - 1) / (u + 1) with
// u = X - A/3, v = Y over ONE inverse of v (u + 1); a zero denominator (a point of order two, or u = -1) is "no encoding" (status 1, zero octets).
// The typed verification path spent 2.5 ms per 2^20 keys in the saturated-word kernel (profiles/r6_typed_boundary.md: 1.47 ms per 589 824).
__global__ __launch_bounds__(64) void k_ed_enc_c25519(const u8 *Rw, const u8 *stR, u8 *enc, u8 *status, u32 n, EcamdEdTailConsts Cst, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	u8 *out = enc + (size_t)i * 32;
	const u32 st = stR[i];
	if (st != 0) {
		// r = 0 mod q: the neutral element (0, 1); a failed multiplication or import: an error
		for (int b = 0; b < 32; b++) {
			out[b] = (u8)((st == 2 && b == 0) ? 1 : 0);
		}
		status[i] = (st == 2) ? 0 : 1;
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	u32 xw[8], yw[8];
	load_be<8>(Rw + (size_t)i * 64, 32, xw);
	load_be<8>(Rw + (size_t)i * 64 + 32, 32, yw);
	const FM X = weaken<FM>(mul(from_words<PB, 8>(xw), constant<FC>(K.ix), K));
	const FM v = weaken<FM>(mul(from_words<PB, 8>(yw), constant<FC>(K.iy), K));
	const auto u = carry(sub_auto<1>(X, digits9(Cst.g_A3), K));
	const auto up1 = carry(add(u, onec));
	const auto um1 = carry(sub_auto<1>(u, onec, K));
	const FM den = weaken<FM>(mulc(v, up1, K));
	FM a11;
	const FM inv = weaken<FM>(mul(sqr_n(pow_2_250m1(den, &a11, K), 5, K), a11, K));   // den^(p - 2); 0 for den = 0
	const FM x = weaken<FM>(mul(mul(mulc(mulc(digits9(Cst.g_alpha), u, K), up1, K), inv, K), onec, K));
	const FM y = weaken<FM>(mul(mulc(mulc(um1, v, K), inv, K), onec, K));
	u32 dx[9], dy[9], w[8];
	canonical_digits(dx, x, K);
	canonical_digits(dy, y, K);
	to_words<9, 8>(w, dy);
	w[7] |= (dx[0] & 1u) << 31;                      // y < 2^255; bit 255 carries the parity of x
#pragma unroll
	for (int b = 0; b < 32; b++) {
		out[b] = (u8)(w[b >> 2] >> (8 * (b & 3)));
	}
	status[i] = is_zero_mulout(mulc(den, onec, K), K) ? 1 : 0;
}
hipError_t ecamd_launch_ed_enc_c25519(const uint8_t *Rw, const uint8_t *stR, uint8_t *enc, uint8_t *status, uint32_t n, const EcamdEdTailConsts &c, int gslot,
				      hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_enc_c25519, dim3((n + 63) / 64), dim3(64), 0, s, Rw, stR, enc, status, n, c, gslot);
	return hipGetLastError();
}

__global__ __launch_bounds__(64) void k_ed_tail_c25519(EcamdEdTailArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 fR = A.flagsR[i];   // 2: R decoded to the neutral element (the reference's point at infinity)
	if (A.flagsA[i] || fR == 1 || A.flagsS[i]) {
		A.result[i] = 1;
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.C.g_2d);
	// ---- [S]B: K = S + 0x8000..8000, signed 16-bit digits, one mixed addition per window ----
	u32 kw[9];
	load_be<8>(A.S_be + (size_t)i * 32, 32, kw);
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			c += (uint64_t)kw[w] + 0x80008000u;
			kw[w] = (u32)c;
			c >>= 32;
		}
		kw[8] = (u32)c;   // top digit: 0 or 1
	}
	Ext SG = ed_neutral(K);
#pragma unroll 1
	for (int j = 0; j <= EDC_NWIN; j++) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			word = (w == (j >> 1)) ? kw[w] : word;
		}
		const int dig = (j < EDC_NWIN) ? (int)((word >> (16 * (j & 1))) & 0xffffu) - 0x8000 : (int)word;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		const PreA Q = prea_load(A.comb + ((size_t)j * EDC_PER_WIN + (mag ? mag - 1 : 0)) * EDC_ENT_WORDS);
		const Ext S = ed_madd<true>(SG, Q, dig < 0, K);
		const bool keep = (mag == 0);
		SG.X = selg(keep, SG.X, S.X);
		SG.Y = selg(keep, SG.Y, S.Y);
		SG.Z = selg(keep, SG.Z, S.Z);
		SG.T = selg(keep, SG.T, S.T);
	}
	// ---- R (affine; (0, 1) when it decoded to the neutral element) ----
	FM xr, yr;
	edr_load(A.edR + (size_t)i * 20, xr, yr);
	// E1: [S]B == (x_R, -y_R)
	bool bad = eq(SG.X, mul(xr, SG.Z, K), K) & eq_neg(SG.Y, mul(yr, SG.Z, K), K);
	// W1 = [S]B - R
	const PreA QR = ed_prea(xr, yr, d2, K);
	const Ext W1 = ed_madd<true>(SG, QR, true, K);
	// ---- [h]A (X : Y : Z) from the window loop ----
	FM hX, hY, hZ;
	{
		u32 buf[EDR_REC_WORDS];
		const uint4 *src = (const uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
		for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
#pragma unroll
		for (int w = 0; w < 9; w++) {
			hX.l[w] = buf[w];
			hY.l[w] = buf[9 + w];
			hZ.l[w] = buf[18 + w];
		}
	}
	// E2: W1 == (x_hA, -y_hA), projectively
	bad = bad | (eq(mul(W1.X, hZ, K), mul(hX, W1.Z, K), K) & eq_neg(mul(W1.Y, hZ, K), mul(hY, W1.Z, K), K));
	// W2 = W1 - [h]A: (X : Y : Z) -> the extended point (X Z : Y Z : Z^2 : X Y), its precomputed form, one addition
	Ext H;
	H.X = weaken<FM>(mul(hX, hZ, K));
	H.Y = weaken<FM>(mul(hY, hZ, K));
	H.Z = weaken<FM>(sqr(hZ, K));
	H.T = weaken<FM>(mul(hX, hY, K));
	Ext W2 = ed_add<false>(W1, ed_pre(H, d2, K), true, K);
	for (u32 k = 0; k < A.cof_dbl; k++) {
		W2 = ed_dbl<false>(W2, K);
	}
	const bool neutral = is_zero_mulout(W2.X, K) & eq(W2.Y, W2.Z, K);
	A.result[i] = (!bad && neutral) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// Round 4, second step: the verification equation with HALF-LENGTH scalars (k_ed_lat in ecamd_kernels.hip finds them):
//     8 ([s']B - [u]R - [v]A) = neutral,  s' = u S mod q,  v = u h mod q,  |u| < 2^126, 0 <= v < 2^127.
// k_ed_smul2_c25519<0>      the window tables [1..8]A and [1..8]R of an item (two ed_table passes)
// k_ed_smul2_c25519<1, NWIN> L = -[v]A - [u]R by ONE signed-window loop over both tables: per window four doublings and two
//                           additions; NWIN = 33 windows cover 128-bit scalars (K = k + 0x8..8 over 33 nibbles: the top digit is 0 or 1,
//                           no carry to treat apart), NWIN = 65 the full-length fall-back of an item k_ed_lat could not shorten
//                           (u = 1, v = h).  A launch handles the items whose mode it is told (meta bit 1), the others return at once.
// k_ed_tail2_c25519         [S]B for the first exceptional pair (E1 as in k_ed_tail_c25519), then W = [s']B + L from a second comb
//                           pass, the second exceptional pair restated for this form -- among accepted signatures W1 + [h]A = T2 is
//                           only possible for h = 0 mod q (2 [h]A would be torsion, and a key of small order was rejected), which is
//                           v = 0, u = 1, W = W1: the test is W = T2 there -- and [8]W = neutral.
// ------------------------------------------------------------------------------------------
template <int NWIN> static __device__ __forceinline__ void ed_recode(u32 *kw, const u32 *k, int nk)
{
	// K = k + 0x88..8 over NWIN nibbles, left-aligned so that the top nibble sits in bits 31..28 of kw[KWORDS - 1]
	constexpr int KWORDS = (NWIN + 7) / 8;
	constexpr int TOPN = NWIN - 8 * (KWORDS - 1);          // nibbles in the top word
	uint64_t c = 0;
#pragma unroll
	for (int w = 0; w < KWORDS; w++) {
		const u32 add = (w < KWORDS - 1 || TOPN == 8) ? 0x88888888u : (0x88888888u >> (4 * (8 - TOPN)));
		c += (uint64_t)(w < nk ? k[w] : 0u) + add;
		kw[w] = (u32)c;
		c >>= 32;
	}
	if (TOPN != 8) {
		// shift left by 4 (8 - TOPN) bits over the whole array
		constexpr int SH = 4 * (8 - TOPN);
#pragma unroll
		for (int w = KWORDS - 1; w > 0; w--) {
			kw[w] = (kw[w] << SH) | (kw[w - 1] >> (32 - SH));
		}
		kw[0] <<= SH;
	}
}
template <int KWORDS> static __device__ __forceinline__ int ed_next_digit(u32 *kw)
{
	const int dig = (int)(kw[KWORDS - 1] >> 28) - 8;
#pragma unroll
	for (int w = KWORDS - 1; w > 0; w--) {
		kw[w] = (kw[w] << 4) | (kw[w - 1] >> 28);
	}
	kw[0] <<= 4;
	return dig;
}

// The window tables of A and R (phase 0 of the half-length-scalar path) as a kernel of its own, so that its register budget is its own:
// ED_TABLE_WAVES = minimum waves per SIMD (2: the rolled table loop keeps its 200 registers; 3: 168 and a few spills -- A/B)
#ifndef ED_TABLE_WAVES
#define ED_TABLE_WAVES 2
#endif
__global__ __launch_bounds__(64, ED_TABLE_WAVES) void k_ed_table2_c25519(EcamdEdSmul2Args A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	const bool live = i < A.n && !(A.flagsA[i] || A.flagsR[i] == 1 || A.flagsS[i]);
#if defined(ED_TABLE_DIRECT_STORE)   /* A/B hook (tools/build_variant.py): every lane stores its own entries */
	if (!live) {
		return;
	}
#endif
	const CK &K = TabGP<255>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FM onem = weaken<FM>(onec);
	const FC d2 = digits9(A.g_2d);
#if defined(ED_TABLE_DIRECT_STORE)
	u32 *tbA = A.tbl + (size_t)i * 2 * EDT_ITEM_WORDS, *tbR = tbA + EDT_ITEM_WORDS;
#else
	// the tables leave through LDS, an entry of the whole wave at a time (pre_store_wave): every lane stays to the end
	__shared__ u32 sh[64 * EDT_LDS_STRIDE + 64];
	u32 *wave_base = A.tbl + (size_t)(blockIdx.x * 64u) * 2 * EDT_ITEM_WORDS;
	const u32 ic = i < A.n ? i : 0u;
#endif
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		Ext P1;
#if defined(ED_TABLE_DIRECT_STORE)
		edr_load((k == 0 ? A.edA : A.edR) + (size_t)i * 20, P1.X, P1.Y);
#else
		edr_load((k == 0 ? A.edA : A.edR) + (size_t)ic * 20, P1.X, P1.Y);
#endif
		P1.Z = onem;
		P1.T = weaken<FM>(mul(P1.X, P1.Y, K));
#if defined(ED_TABLE_DIRECT_STORE)
		ed_table(k == 0 ? tbA : tbR, P1, d2, K);
#else
		u32 *wb = wave_base + (k == 0 ? 0 : EDT_ITEM_WORDS);
		ed_table_with([&](int e, const Pre &Q) { pre_store_wave(sh, wb, 2 * EDT_ITEM_WORDS, e, Q, live); }, P1, d2, K);
#endif
	}
}

template <int phase, int NWIN> __global__ __launch_bounds__(64) void k_ed_smul2_c25519(EcamdEdSmul2Args A, int gslot)
{
	static_assert(phase == 1, "phase 0 is k_ed_table2_c25519");
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.flagsA[i] || A.flagsR[i] == 1 || A.flagsS[i]) {
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	u32 *tbA = A.tbl + (size_t)i * 2 * EDT_ITEM_WORDS, *tbR = tbA + EDT_ITEM_WORDS;
	const u32 meta = A.meta[i];
	if (((meta >> 1) & 1u) != (NWIN > 33 ? 1u : 0u)) {
		return;   // the other launch's item
	}
	constexpr int KWORDS = (NWIN + 7) / 8;
	u32 kv[KWORDS], ku[KWORDS];
	{
		const u32 *uv = A.uv + (size_t)i * 12;
		u32 v[8], u[4];
#pragma unroll
		for (int w = 0; w < 8; w++) {
			v[w] = uv[w];
		}
#pragma unroll
		for (int w = 0; w < 4; w++) {
			u[w] = uv[8 + w];
		}
		ed_recode<NWIN>(kv, v, 8);
		ed_recode<NWIN>(ku, u, 4);
	}
	const bool uneg = (meta & 1u) != 0;
	Ext acc = ed_neutral(K);
#pragma unroll 1
	for (int t = 0; t < NWIN; t++) {
#if defined(ED_SMUL2_UNROLL)   /* A/B hook (tools/build_variant.py): the three T-less doublings spelled out (57 KB of loop body instead of 42) */
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
		acc = ed_dbl<false>(acc, K);
#else
#pragma unroll 1
		for (int d = 0; d < 3; d++) {
			acc = ed_dbl<false>(acc, K);
		}
#endif
		acc = ed_dbl<true>(acc, K);
		{
			// - [d]A: subtract the entry for a positive digit
			const int dig = ed_next_digit<KWORDS>(kv);
			const u32 mag = (u32)(dig < 0 ? -dig : dig);
			const Pre Q = pre_load(tbA, mag ? mag - 1 : 0);
			const Ext S = ed_add<true>(acc, Q, dig > 0, K);
			const bool keep = (mag == 0);
			acc.X = selg(keep, acc.X, S.X);
			acc.Y = selg(keep, acc.Y, S.Y);
			acc.Z = selg(keep, acc.Z, S.Z);
			acc.T = selg(keep, acc.T, S.T);
		}
		{
			// - [u]R: u > 0 subtracts the entry for a positive digit, u < 0 adds it
			const int dig = ed_next_digit<KWORDS>(ku);
			const u32 mag = (u32)(dig < 0 ? -dig : dig);
			const Pre Q = pre_load(tbR, mag ? mag - 1 : 0);
			const Ext S = ed_add<false>(acc, Q, uneg ? (dig < 0) : (dig > 0), K);
			const bool keep = (mag == 0);
			acc.X = selg(keep, acc.X, S.X);
			acc.Y = selg(keep, acc.Y, S.Y);
			acc.Z = selg(keep, acc.Z, S.Z);
		}
	}
	u32 buf[EDR_REC_WORDS];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		buf[w] = acc.X.l[w];
		buf[9 + w] = acc.Y.l[w];
		buf[18 + w] = acc.Z.l[w];
	}
	buf[27] = 0;
	uint4 *dst = (uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
	for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

// [k]B for the big-endian 32-byte scalar at sc: seventeen mixed additions from the Edwards comb table
static __device__ __forceinline__ c25519::Ext ed_comb_B(const u8 *sc, const u32 *comb, const c25519::CK &K)
{
	using namespace c25519;
	u32 kw[9];
	load_be<8>(sc, 32, kw);
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			c += (uint64_t)kw[w] + 0x80008000u;
			kw[w] = (u32)c;
			c >>= 32;
		}
		kw[8] = (u32)c;   // top digit: 0 or 1
	}
	Ext SG = ed_neutral(K);
#pragma unroll 1
	for (int j = 0; j <= EDC_NWIN; j++) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			word = (w == (j >> 1)) ? kw[w] : word;
		}
		const int dig = (j < EDC_NWIN) ? (int)((word >> (16 * (j & 1))) & 0xffffu) - 0x8000 : (int)word;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		const PreA Q = prea_load(comb + ((size_t)j * EDC_PER_WIN + (mag ? mag - 1 : 0)) * EDC_ENT_WORDS);
		const Ext S = ed_madd<true>(SG, Q, dig < 0, K);
		const bool keep = (mag == 0);
		SG.X = selg(keep, SG.X, S.X);
		SG.Y = selg(keep, SG.Y, S.Y);
		SG.Z = selg(keep, SG.Z, S.Z);
		SG.T = selg(keep, SG.T, S.T);
	}
	return SG;
}

__global__ __launch_bounds__(64) void k_ed_tail2_c25519(EcamdEdTailArgs A, int gslot)
{
	using namespace c25519;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 fR = A.flagsR[i];
	if (A.flagsA[i] || fR == 1 || A.flagsS[i]) {
		A.result[i] = 1;
		return;
	}
	const CK &K = TabGP<255>::get(gslot);
	const FC d2 = digits9(A.C.g_2d);
	// E1: [S]B == (x_R, -y_R)
	FM xr, yr;
	edr_load(A.edR + (size_t)i * 20, xr, yr);
	bool bad;
	{
		const Ext SG = ed_comb_B(A.S_be + (size_t)i * 32, A.comb, K);
		bad = eq(SG.X, mul(xr, SG.Z, K), K) & eq_neg(SG.Y, mul(yr, SG.Z, K), K);
	}
	// W = [s']B + L,  L = -[v]A - [u]R from the window loop
	Ext W = ed_comb_B(A.sp_be + (size_t)i * 32, A.comb, K);
	{
		FM hX, hY, hZ;
		u32 buf[EDR_REC_WORDS];
		const uint4 *src = (const uint4 *)(A.rec + (size_t)i * EDR_REC_WORDS);
#pragma unroll
		for (int q = 0; q < EDR_REC_WORDS / 4; q++) {
			const uint4 v = src[q];
			buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
		}
#pragma unroll
		for (int w = 0; w < 9; w++) {
			hX.l[w] = buf[w];
			hY.l[w] = buf[9 + w];
			hZ.l[w] = buf[18 + w];
		}
		Ext H;
		H.X = weaken<FM>(mul(hX, hZ, K));
		H.Y = weaken<FM>(mul(hY, hZ, K));
		H.Z = weaken<FM>(sqr(hZ, K));
		H.T = weaken<FM>(mul(hX, hY, K));
		W = ed_add<false>(W, ed_pre(H, d2, K), false, K);
	}
	// E2 (only possible for h = 0 mod q, i.e. v = 0, u = 1, W = W1): W1 == T2 = (0, -1)
	// INVARIANT this shortcut rests on: for h != 0 mod q the pair W1 + [h]A = T2 would need [8]([S]B - R) = [8h]A with
	// [h]A of order two up to the torsion of an accepted equation, i.e. [8]A = infinity -- and k_ed_decode_ed_c25519 (flagsA, the
	// reference's small-order test of the key, sig/eddsa.c:2190-2200) has already rejected such keys before this kernel sees the item.
	// An entry point that feeds this tail MUST keep that test in front of it; tests/test_gpu_parity.py::test_eddsa25519_exceptional_pairs
	// runs small-order and mixed-torsion keys with the lattice path on and forced off against the unmodified reference.
	if (A.meta[i] & 4u) {
		bad = bad | (is_zero_mulout(W.X, K) & eq_neg(W.Y, W.Z, K));
	}
	for (u32 k = 0; k < A.cof_dbl; k++) {
		W = ed_dbl<false>(W, K);
	}
	const bool neutral = is_zero_mulout(W.X, K) & eq(W.Y, W.Z, K);
	A.result[i] = (!bad && neutral) ? 0 : 1;
}

hipError_t ecamd_launch_ed_smul2_c25519(const EcamdEdSmul2Args &a, int gslot, hipStream_t s, hipEvent_t *dom)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	hipLaunchKernelGGL(k_ed_table2_c25519, grid, block, 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[0], s);
	}
	hipLaunchKernelGGL((k_ed_smul2_c25519<1, 33>), grid, block, 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[1], s);
	}
	hipLaunchKernelGGL((k_ed_smul2_c25519<1, 65>), grid, block, 0, s, a, gslot);   // the items k_ed_lat left at full length (normally none)
	return hipGetLastError();
}
hipError_t ecamd_launch_ed_tail2_c25519(const EcamdEdTailArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_tail2_c25519, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}

hipError_t ecamd_launch_edcomb_build_c25519(const uint8_t *pts, uint32_t n, uint32_t *table, const EcamdEdTailConsts &c, int gslot, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_edcomb_build_c25519, dim3((n + 63) / 64), dim3(64), 0, s, pts, n, table, c, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_ed_tail_c25519(const EcamdEdTailArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_tail_c25519, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_smul_c25519(const EcamdEdSmulArgs &a, int gslot, hipStream_t s, hipEvent_t *dom)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_smul_c25519<0>, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[0], s);
	}
	hipLaunchKernelGGL(k_ed_smul_c25519<1>, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[1], s);
	}
	if (a.out != nullptr) {   // (NULL: the caller continues on the Edwards curve, k_ed_tail_c25519)
		const uint32_t nthreads = (a.n + EDF_K - 1) / EDF_K;
		hipLaunchKernelGGL(k_ed_hA_fin, dim3((nthreads + 63) / 64), dim3(64), 0, s, a, gslot, nthreads);
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_ed_decode_ed_c25519(const EcamdEdDecodeArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_decode_ed_c25519, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_ed_decode_c25519(const EcamdEdDecodeArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_decode_c25519, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
hipError_t ecamd_launch_xdh_prep_c25519(const EcamdXdhPrepArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_xdh_prep_c25519, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
#endif  // G29_P25519

#ifdef G29_P448
// ------------------------------------------------------------------------------------------
// Ed448 point decoding on the Goldilocks field of this unit (p = 2^448 - 2^224 - 1, plain residues): the computation of
// k_ed448_decode<14> in ecamd_kernels.hip (which stays the reference implementation and serves when this unit has no
// slot), operation for operation -- EDDSA448 branch of eddsa_decode_point (sig/eddsa.c:430-560), the 4-isogeny to the
// Edwards model of curve448, the map to the Weierstrass model WEI448.
// ------------------------------------------------------------------------------------------
namespace c448 {
constexpr int PB = 448;
typedef Cls<PB>::FA FA;
typedef Cls<PB>::FM FM;
typedef Cls<PB>::FC FC;
typedef CurveG<16> CK;
#define M_(a, b) weaken<FM>(mulc(a, b, K))
#define S_(a) weaken<FM>(sqrc(a, K))
#define SUB_(a, b) carry(sub_auto<1>(a, b, K))   /* b: a multiplication result or a constant */
#define ADD_(a, b) carry(add(a, b))

static __device__ __forceinline__ FM sqr_n(FM a, int n, const CK &K)
{
	for (int k = 0; k < n; k++) {
		a = S_(a);
	}
	return a;
}
static __device__ __forceinline__ FC digits16(const u32 *d)
{
	FC r;
#pragma unroll
	for (int w = 0; w < 16; w++) {
		r.l[w] = d[w];
	}
	return r;
}
static __device__ __forceinline__ FC small_const(u32 v)
{
	FC r;
#pragma unroll
	for (int w = 0; w < 16; w++) {
		r.l[w] = (w == 0) ? v : 0u;
	}
	return r;
}
// exact zero test of a lazily reduced value
template <class A> static __device__ __forceinline__ bool is_zero(const A &a, const CK &K)
{
	return is_zero_mulout(mulc(a, constant<FC>(K.one), K), K);
}
// w^((p - 3) / 4), (p - 3) / 4 = 2^223 (2^223 - 1) + 2^222 - 1: 446 S + 14 M (the chain of fe_pow_p448_e34)
static __device__ FM pow_e34(const FM &w, const CK &K)
{
	const int steps[12] = {1, 0, 3, 6, 0, 13, 0, 27, 0, 55, 0, 111};
	FM f = w;
#pragma unroll 1
	for (int s = 0; s < 12; s++) {
		const int k = steps[s];
		const FM g = (k == 0) ? w : f;
		f = M_(sqr_n(f, k == 0 ? 1 : k, K), g);
	}
	const FM f223 = M_(S_(f), w);
	return M_(sqr_n(f223, 223, K), f);
}
static __device__ FM inv448(const FM &w, const CK &K)   // 0 -> 0, like Fermat's
{
	return M_(sqr_n(pow_e34(w, K), 2, K), w);
}
template <class A> static __device__ __forceinline__ void store_canon_be(u8 *dst, const A &a, bool ok, const CK &K)
{
	u32 d[16], w[14];
	canonical_digits(d, mulc(a, constant<FC>(K.one), K), K);
	to_words<16, 14>(w, d);
#pragma unroll
	for (int i = 0; i < 14; i++) {
		w[i] = ok ? w[i] : 0u;
	}
	store_be<14>(dst, 56, w);
}
}  // namespace c448

#ifndef ED448_DECODE_OCC
#define ED448_DECODE_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))   /* left alone the single-inversion form takes 293 registers: one wave per SIMD */
#endif
__global__ __launch_bounds__(64) ED448_DECODE_OCC void k_ed448_decode_g(EcamdEd448DecodeArgs A, int gslot)
{
	using namespace c448;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<448>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const FC twoc = small_const(2u);
	const FC zeroc = small_const(0u);
	const FA onea = weaken<FA>(onec);
	// Round 6: ONE inversion per item instead of two.  With x' = Nx / d1, y' = Ny / d2 (Nx = alpha x y, Ny = x^2 + y^2, d1 = 2 - x^2 - y^2,
	// d2 = y^2 - x^2) the map to the Weierstrass model needs u = (1 + y') / (1 - y') = (d2 + Ny) / (d2 - Ny) and v = alpha u / x' =
	// alpha (d2 + Ny) d1 / ((d2 - Ny) Nx): both over W = (d2 - Ny) Nx, so the isogeny's own inversion is never needed -- its two
	// denominators are only tested for zero (the reference's fp_inv(0) errors), the Edwards-model curve equation is tested with the
	// denominators cleared, and x' = 0 / y' = 1 are Nx = 0 / d2 = Ny.  Same verdicts and the same canonical coordinates as
	// k_ed448_decode<14> (ecamd_kernels.hip), which keeps the reference's order of operations.
	FA Nx[2], D1[2], Pn[2], Wd[2];
	bool ok[2], neutral[2] = {false, false};
#pragma unroll 1
	for (int k = 0; k < 2; k++) {
		const u8 *src = (k == 0) ? A.encA + (size_t)i * A.strideA : A.encR + (size_t)i * A.strideR;
		u32 yw[14];
#pragma unroll
		for (int w = 0; w < 14; w++) {
			yw[w] = (u32)src[4 * w] | ((u32)src[4 * w + 1] << 8) | ((u32)src[4 * w + 2] << 16) | ((u32)src[4 * w + 3] << 24);
		}
		const u32 last = src[56];
		const u32 x0 = last >> 7;
		const auto yd = from_words<PB, 14>(yw);
		bool below;
		{
			u32 b = 0;
#pragma unroll
			for (int j = 0; j < 16; j++) {
				b = (yd.l[j] - K.p[j] - b) >> 31;
			}
			below = b != 0;
		}
		bool good = ((last & 0x7fu) == 0) & below;                      // the 57th byte only carries the sign
		const FM ym = M_(yd, onec);
		const FM yy = S_(ym);
		const auto u = SUB_(onec, yy);                                   // 1 - y^2
		const auto v = SUB_(onec, M_(digits16(A.g_d448), yy));           // a - d y^2, a = 1
		good = good & !is_zero(v, K);                                    // fp_inv(0)
		const FM v2 = S_(v), u2 = S_(u);
		const FM u3v = M_(M_(u2, u), v);
		const FM u5v3 = M_(M_(u3v, u2), v2);
		const FM r = M_(u3v, pow_e34(u5v3, K));
		good = good & is_zero(SUB_(u, M_(v, S_(r))), K);                 // u / v has no root: error
		u32 rd[16];
		canonical_digits(rd, r, K);
		u32 nz = 0;
#pragma unroll
		for (int w = 0; w < 16; w++) {
			nz |= rd[w];
		}
		const FA x = selg((rd[0] & 1u) != x0, weaken<FA>(SUB_(zeroc, r)), weaken<FA>(r));
		good = good & !((nz == 0) & (x0 == 1u));
		const FM xx = S_(r);
		const auto e1 = SUB_(SUB_(twoc, xx), yy);                        // d1 = 2 - x^2 - y^2
		const auto e2 = SUB_(yy, xx);                                    // d2 = y^2 - x^2
		good = good & !is_zero(e1, K) & !is_zero(e2, K);                 // fp_inv(0) of the isogeny
		const FM nx = M_(digits16(A.g_alpha), M_(x, ym));                // alpha x y
		const FM ny = M_(ADD_(xx, yy), onec);                            // x^2 + y^2
		// on the Edwards model of curve448, x'^2 + y'^2 = 1 + d' x'^2 y'^2, times d1^2 d2^2
		const FM a2 = S_(M_(nx, e2)), b2 = S_(M_(ny, e1)), dd = S_(M_(e1, e2)), nn = S_(M_(nx, ny));
		const FM rhs = M_(ADD_(dd, M_(digits16(A.g_diso), nn)), onec);
		const bool on = is_zero(SUB_(ADD_(a2, b2), rhs), K);
		const auto om = SUB_(e2, ny);                                    // (1 - y') d2
		const bool xz = is_zero_mulout(nx, K), oz = is_zero(om, K);
		// (0, 1) -- the image of both (0, 1) and (0, -1) of Ed448 -- is the neutral element: the point at infinity of the
		// Weierstrass model (fine for R; a key is then rejected as small-order).  X = 0 with Y = -1 dies in fp_inv(0).
		neutral[k] = good & on & xz & oz;
		ok[k] = good & on & !xz & !oz;
		Nx[k] = selg(ok[k], weaken<FA>(nx), onea);
		D1[k] = selg(ok[k], weaken<FA>(e1), onea);
		Pn[k] = weaken<FA>(ADD_(e2, ny));                                // (1 + y') d2
		Wd[k] = selg(ok[k], weaken<FA>(M_(om, nx)), onea);               // W = (d2 - Ny) Nx
	}
	{
		const FM inv = inv448(M_(Wd[0], Wd[1]), K);
		const FM ia = M_(inv, Wd[1]), ir = M_(inv, Wd[0]);               // 1 / W of A, of R
#pragma unroll 1
		for (int k = 0; k < 2; k++) {
			const FM mi = (k == 0) ? ia : ir;
			const FM u = M_(M_(Pn[k], Nx[k]), mi);
			const FM v = M_(M_(digits16(A.g_alpha), M_(Pn[k], D1[k])), mi);
			u8 *pd = (k == 0 ? A.pointsA : A.pointsR) + (size_t)i * 112;
			store_canon_be(pd, SUB_(digits16(A.g_A3), u), ok[k], K);     // (A, B) = (-156326, -1): X = A/3 - u
			store_canon_be(pd + 56, SUB_(zeroc, v), ok[k], K);           // Y = -v
			(k == 0 ? A.flagsA : A.flagsR)[i] = ok[k] ? 0 : ((k == 1 && neutral[1]) ? 2 : 1);   // 2: R is the point at infinity
		}
	}
}
// X448 front end (k_xdh_prep<14> of ecamd_kernels.hip on this field): clamped scalar, u < p, v = w^((p + 1) / 4) with
// w = u (u (u + A) + 1) -- (p + 1) / 4 = 2^222 (2^224 - 1): the chain of pow_e34 two steps further, then 222 squarings --
// v^2 = w or the point is on the twist, the map to WEI448, [4]Q != infinity by two doublings.
__global__ __launch_bounds__(64) void k_xdh_prep_c448(EcamdXdhPrepArgs A, int gslot)
{
	using namespace c448;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CK &K = TabGP<448>::get(gslot);
	const FC onec = constant<FC>(K.one);
	{
		// scalar: reversed to big-endian and clamped (decode_scalar)
		const u8 *ks = A.k + (size_t)i * 56;
		u8 *kd = A.scalars + (size_t)i * 56;
		for (int b = 0; b < 56; b++) {
			u8 v = ks[b];
			if (b == 0) v &= 252;
			if (b == 55) v |= 128;
			kd[55 - b] = v;
		}
	}
	const u8 *src = A.u + (size_t)i * 56;
	u32 uw[14];
#pragma unroll
	for (int w = 0; w < 14; w++) {
		uw[w] = (u32)src[4 * w] | ((u32)src[4 * w + 1] << 8) | ((u32)src[4 * w + 2] << 16) | ((u32)src[4 * w + 3] << 24);
	}
	const auto ud = from_words<PB, 14>(uw);
	bool ok;
	{
		u32 b = 0;
#pragma unroll
		for (int j = 0; j < 16; j++) {
			b = (ud.l[j] - K.p[j] - b) >> 31;
		}
		ok = b != 0;   // u >= p is rejected
	}
	const FM um = M_(ud, onec);
	const FM w = M_(um, ADD_(M_(um, ADD_(um, digits16(A.g_A))), onec));   // u (u (u + A) + 1)
	FM c;
	{
		const int steps[12] = {1, 0, 3, 6, 0, 13, 0, 27, 0, 55, 0, 111};
		FM f = w;
#pragma unroll 1
		for (int s = 0; s < 12; s++) {
			const int k = steps[s];
			const FM g = (k == 0) ? w : f;
			f = M_(sqr_n(f, k == 0 ? 1 : k, K), g);                 // ... f(222)
		}
		f = M_(S_(f), w);                                              // f(223)
		f = M_(S_(f), w);                                              // f(224) = w^(2^224 - 1)
		c = sqr_n(f, 222, K);
	}
	ok = ok & is_zero(SUB_(S_(c), w), K);                                  // no root: u is on the twist
	const auto xm = ADD_(um, digits16(A.g_A3));
	{
		// [h]Q must not be infinity (x25519_448.c:259-260): cof_dbl doublings; a doubling reaches infinity exactly when Y = 0,
		// which shows as Z = 0 one step later (or at once for y = 0)
		// (through the unit's import factors: the constants may be those of the isomorphic a = -3 curve)
		Jac<PB> P;
		P.X = weaken<FA>(M_(xm, constant<FC>(K.ix)));
		P.Y = weaken<FA>(M_(c, constant<FC>(K.iy)));
		P.Z = weaken<FA>(onec);
		for (u32 r = 0; r < A.cof_dbl; r++) {
			P = dbl(P, K);
		}
		ok = ok & !is_zero(P.Z, K);
	}
	u8 *pd = A.points + (size_t)i * 112;
	store_canon_be(pd, xm, ok, K);
	store_canon_be(pd + 56, c, ok, K);
	A.flags[i] = ok ? 0 : 1;
}
// ------------------------------------------------------------------------------------------
// X448: the x-only Montgomery ladder on v^2 = u^3 + 156326 u^2 + u itself (RFC 7748 section 5), for inputs k_xdh_prep_c448
// accepted -- the 448-bit twin of k_x25519_ladder / k_x25519_fin above the #ifdef (same argument for why it is observably
// the reference's result: [k]Q's u coordinate is all x448() exposes, ecdh/x25519_448.c:268-276).  448 steps of 5M + 4S +
// one multiplication by a24 = 39081; the divisions are shared by 8 items per lane.
// ------------------------------------------------------------------------------------------
#define X448_REC_WORDS 32   /* X2, Z2: 16 limbs each */
#ifndef X448_OCC
#define X448_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))   /* 256 VGPRs and a few spills; without it: one wave per SIMD, no spills (A/B) */
#endif
__global__ __launch_bounds__(64) X448_OCC void k_x448_ladder(EcamdXdhLadderArgs A, int gslot)
{
	using namespace c448;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.flags[i]) {
		return;
	}
	const CK &K = TabGP<448>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const u8 *src = A.u + (size_t)i * 56;
	u32 uw[14], kw[14];
#pragma unroll
	for (int w = 0; w < 14; w++) {
		uw[w] = (u32)src[4 * w] | ((u32)src[4 * w + 1] << 8) | ((u32)src[4 * w + 2] << 16) | ((u32)src[4 * w + 3] << 24);
	}
	load_be<14>(A.scalars + (size_t)i * 56, 56, kw);           // clamped by the prep kernel
	const auto ud = from_words<PB, 14>(uw);
	const FM x1 = M_(ud, onec);
	FM x2 = weaken<FM>(onec), z2, x3 = x1, z3 = weaken<FM>(onec);
#pragma unroll
	for (int w = 0; w < 16; w++) {
		z2.l[w] = 0;
	}
	u32 swap = 0;
#pragma unroll 1
	for (int t = 447; t >= 0; t--) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < 14; w++) {
			word = (w == (t >> 5)) ? kw[w] : word;
		}
		const u32 kt = (word >> (t & 31)) & 1u;
		swap ^= kt;
		{
			const FM tx = selg(swap != 0, x3, x2), tz = selg(swap != 0, z3, z2);
			x3 = selg(swap != 0, x2, x3);
			z3 = selg(swap != 0, z2, z3);
			x2 = tx;
			z2 = tz;
		}
		swap = kt;
		const auto a = ADD_(x2, z2);
		const auto b = SUB_(x2, z2);
		const FM aa = S_(a);
		const FM bb = S_(b);
		const auto e = SUB_(aa, bb);
		const auto c = ADD_(x3, z3);
		const auto d = SUB_(x3, z3);
		const FM da = M_(d, a);
		const FM cb = M_(c, b);
		x3 = S_(ADD_(da, cb));
		z3 = M_(x1, S_(SUB_(da, cb)));
		x2 = M_(aa, bb);
		z2 = M_(e, ADD_(aa, mul_word<39081u>(e)));   // a24 e: sixteen MADs (round 3: a full product, 256)
	}
	{
		const FM tx = selg(swap != 0, x3, x2), tz = selg(swap != 0, z3, z2);
		x2 = tx;
		z2 = tz;
	}
	u32 buf[X448_REC_WORDS];
#pragma unroll
	for (int w = 0; w < 16; w++) {
		buf[w] = x2.l[w];
		buf[16 + w] = z2.l[w];
	}
	uint4 *dst = (uint4 *)(A.rec + (size_t)i * X448_REC_WORDS);
#pragma unroll
	for (int q = 0; q < X448_REC_WORDS / 4; q++) {
		dst[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
	}
}

#define X448_FIN_K 8
__global__ __launch_bounds__(64) void k_x448_fin(EcamdXdhLadderArgs A, int gslot, u32 nthreads)
{
	using namespace c448;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CK &K = TabGP<448>::get(gslot);
	const FM onem = weaken<FM>(constant<FC>(K.one));
	// product of the Z2 of this lane's items (rejected items and Z2 = 0 take part with 1).  The partial products are not kept
	// (8 x 16 registers): the way back recomputes the one it needs, 28 multiplications per lane against 8 x 460 for inversions
	u32 live = 0;
	FM acc = onem;
#pragma unroll 1
	for (int j = 0; j < X448_FIN_K; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.flags[i]) {
			continue;
		}
		FM z;
		const u32 *rec = A.rec + (size_t)i * X448_REC_WORDS;
#pragma unroll
		for (int w = 0; w < 16; w++) {
			z.l[w] = rec[16 + w];
		}
		if (!is_zero_mulout(z, K)) {
			live |= 1u << j;
			acc = M_(acc, z);
		}
	}
	FM inv = inv448(acc, K);   // 1 / (product of the live Z2)
#pragma unroll 1
	for (int j = X448_FIN_K - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			continue;
		}
		u8 *out = A.out + (size_t)i * 56;
		bool ok = ((live >> j) & 1u) != 0;
		u32 w14[14];
#pragma unroll
		for (int w = 0; w < 14; w++) {
			w14[w] = 0;
		}
		if (ok) {
			const u32 *rec = A.rec + (size_t)i * X448_REC_WORDS;
			FM x, z;
#pragma unroll
			for (int w = 0; w < 16; w++) {
				x.l[w] = rec[w];
				z.l[w] = rec[16 + w];
			}
			// 1 / z_j = inv * (product of the live z before j): recomputed by walking the earlier items again
			FM before = onem;
#pragma unroll 1
			for (int jj = 0; jj < j; jj++) {
				if ((live >> jj) & 1u) {
					const u32 *r2 = A.rec + (size_t)(t + (u32)jj * nthreads) * X448_REC_WORDS;
					FM zz;
#pragma unroll
					for (int w = 0; w < 16; w++) {
						zz.l[w] = r2[16 + w];
					}
					before = M_(before, zz);
				}
			}
			const FM zi = M_(inv, before);
			inv = M_(inv, z);
			u32 d[16];
			canonical_digits(d, mul(x, zi, K), K);
			to_words<16, 14>(w14, d);
			u32 nz = 0;
#pragma unroll
			for (int w = 0; w < 14; w++) {
				nz |= w14[w];
			}
			ok = nz != 0;   // an all-zero output is rejected (x25519_448.c:275-276)
		}
#pragma unroll 1
		for (int w = 0; w < 14; w++) {
			const u32 v = ok ? w14[w] : 0u;
			out[4 * w] = (u8)v;
			out[4 * w + 1] = (u8)(v >> 8);
			out[4 * w + 2] = (u8)(v >> 16);
			out[4 * w + 3] = (u8)(v >> 24);
		}
		A.status[i] = ok ? 0 : 1;
	}
}

hipError_t ecamd_launch_x448_ladder(const EcamdXdhLadderArgs &a, int gslot, hipStream_t s, hipEvent_t *dom)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (dom) {
		(void)hipEventRecord(dom[0], s);
	}
	hipLaunchKernelGGL(k_x448_ladder, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	if (dom) {
		(void)hipEventRecord(dom[1], s);
	}
	const uint32_t nthreads = (a.n + X448_FIN_K - 1) / X448_FIN_K;
	hipLaunchKernelGGL(k_x448_fin, dim3((nthreads + 63) / 64), dim3(64), 0, s, a, gslot, nthreads);
	return hipGetLastError();
}

#undef M_
#undef S_
#undef SUB_
#undef ADD_

hipError_t ecamd_launch_xdh_prep_c448(const EcamdXdhPrepArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_xdh_prep_c448, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}

hipError_t ecamd_launch_ed448_decode_g(const EcamdEd448DecodeArgs &a, int gslot, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed448_decode_g, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
#endif  // G29_P448

#if defined(G29_P25519) || defined(G29_P448)
// ------------------------------------------------------------------------------------------
// The tail of an EdDSA verification (k_ed_fin<NW> of ecamd_kernels.hip, statement for statement) on this unit's field:
// [S]B - R - [h]A by two complete additions -- an exceptional pair (result Y = Z = 0) rejects, as prj_pt_add's -1 does
// (curves/prj_pt.c:1058-1060) --, the cofactor doublings of _prj_pt_unprotected_mult (curves/prj_pt.c:1862-1905), accepted when the
// result is the point at infinity.  Ed448 also checks [4]A != infinity of the decoded key here (A.Akey).
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ rcbg::PtG<G29_PB> edfin_load(const u8 *src, u32 st, int clen, bool negate,
								 const CurveG<Lay<G29_PB>::NL> &K)
{
	using namespace rcbg;
	typedef Cls<G29_PB>::FA FA;
	typedef Cls<G29_PB>::FC FC;
	constexpr int NW = Lay<G29_PB>::NW;
	if (st == 2) {
		return infinity<G29_PB>(K);
	}
	u32 xw[NW], yw[NW];
	load_be<NW>(src, clen, xw);
	load_be<NW>(src + clen, clen, yw);
	const auto xd = from_words<G29_PB, NW>(xw), yd = from_words<G29_PB, NW>(yw);
	PtG<G29_PB> P;
	P.X = weaken<FA>(mul(xd, constant<FC>(K.ix), K));
	const auto ym = mul(yd, constant<FC>(K.iy), K);
	if (negate) {
		P.Y = neg<G29_PB>(weaken<Cls<G29_PB>::FM>(ym), K);
	} else {
		P.Y = weaken<FA>(ym);
	}
	P.Z = weaken<FA>(constant<FC>(K.one));
	return P;
}

// (PB only makes the kernel symbols of the two units distinct)
template <int PB> __global__ __launch_bounds__(64) void k_ed_fin_g(EcamdEdFinArgs A, int gslot)
{
	static_assert(PB == G29_PB, "one instantiation per unit");
	using namespace rcbg;
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const CurveG<Lay<G29_PB>::NL> &K = TabGP<G29_PB>::get(gslot);
	const int clen = (int)A.clen;
	const u32 sSG = A.stSG[i], shA = A.sthA[i];
	const u32 fR = A.flagsR[i];   // 2: R decoded to the neutral element, i.e. the point at infinity
	if (A.flagsA[i] || fR == 1 || A.flagsS[i] || sSG == 1 || shA == 1) {
		A.result[i] = 1;
		return;
	}
	if (A.Akey != nullptr) {
		if (A.stA != nullptr && A.stA[i] != 0) {
			A.result[i] = 1;
			return;
		}
		PtG<G29_PB> K4 = edfin_load(A.Akey + (size_t)i * 2 * clen, 0, clen, false, K);
		for (u32 k = 0; k < A.cof_dbl; k++) {
			K4 = dbl_rcb<G29_PB>(K4, K);
		}
		if (coord_is_zero<G29_PB>(K4.Z, K)) {
			A.result[i] = 1;
			return;
		}
	}
	PtG<G29_PB> W = edfin_load(A.SG + (size_t)i * 2 * clen, sSG, clen, false, K);
	const PtG<G29_PB> Rn = edfin_load(A.R + (size_t)i * 2 * clen, fR, clen, true, K);
	const PtG<G29_PB> Hn = edfin_load(A.hA + (size_t)i * 2 * clen, shA, clen, true, K);
	W = add_rcb<G29_PB>(W, Rn, K);
	bool bad = coord_is_zero<G29_PB>(W.Z, K) & coord_is_zero<G29_PB>(W.Y, K);
	W = add_rcb<G29_PB>(W, Hn, K);
	bad = bad | (coord_is_zero<G29_PB>(W.Z, K) & coord_is_zero<G29_PB>(W.Y, K));
	for (u32 k = 0; k < A.cof_dbl; k++) {
		W = dbl_rcb<G29_PB>(W, K);
	}
	A.result[i] = (!bad && coord_is_zero<G29_PB>(W.Z, K)) ? 0 : 1;
}

hipError_t G29_CAT(ecamd_g29_ed_fin_, G29_TAG)(int gslot, const EcamdEdFinArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_ed_fin_g<G29_PB>, dim3((a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	return hipGetLastError();
}
#endif

#if G29_FLAV == 0
// ------------------------------------------------------------------------------------------
// Round 4: the mod-q algebra in front of an ECDSA verification (k_ecdsa_prep<NW> of ecamd_kernels.hip: range checks of r and s,
// s^-1 by one Fermat inversion shared by the items of a lane, u1 = e / s, u2 = r / s; sig/ecdsa_common.c:760-791) on the dense
// radix-2^29 unit of the ORDER's size, with q in a constant slot of its own: the same field code as the curve arithmetic, the
// modulus being q instead of p (upload_g29_mod in ecamd_host.cpp builds its constants; ix = R^2 takes a value into the
// Montgomery domain, ex = 1 out of it).  The prefix products of Montgomery's trick rest in `scratch` (NL words per item),
// so a lane can take sixteen items whatever the field size.  Lane t owns the items t, t + lanes, t + 2 lanes, ...
// ------------------------------------------------------------------------------------------
template <int PB> __global__ __launch_bounds__(64) void k_ecdsa_prep_g(EcamdEcdsaPrepArgs A, u32 *scratch, int qgslot, u32 lanes, int kp)
{
	static_assert(PB == G29_PB, "one instantiation per unit");
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, NW = L::NW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= lanes) {
		return;
	}
	if (A.only != nullptr) {
		bool any = false;
		for (int j = 0; j < kp; j++) {
			const u32 i = t + (u32)j * lanes;
			any = any | (i < A.n && A.only[i] == ECAMD_STATUS_REDO);
		}
		if (!any) {
			return;
		}
	}
	const CurveG<NL> &K = TabGP<PB>::get(qgslot);
	const int qlen = (int)A.qlen, hlen = (int)A.hlen;
	const FM onem = weaken<FM>(constant<FC>(K.one));
	u32 qw[NW];
	to_words<NL, NW>(qw, K.p);
	// value (NW words) in [1, q - 1]?
	auto in_range = [&](const u32 *w) {
		u32 nz = 0, borrow = 0;
#pragma unroll
		for (int k = 0; k < NW; k++) {
			nz |= w[k];
			const uint64_t d = (uint64_t)w[k] - qw[k] - borrow;
			borrow = (u32)(d >> 63);
		}
		return (nz != 0u) & (borrow != 0u);
	};
	FM acc = onem;
#pragma unroll 1
	for (int j = 0; j < kp; j++) {
		const u32 i = t + (u32)j * lanes;
		if (i >= A.n) {
			break;
		}
		const u8 *sig = A.sigs + (size_t)i * 2 * qlen;
		u32 rw[NW], sw[NW];
		load_be<NW>(sig, qlen, rw);
		load_be<NW>(sig + qlen, qlen, sw);
		const bool ok = in_range(rw) & in_range(sw);
		if (ok) {
			const FM sm = weaken<FM>(mul(from_words<PB, NW>(sw), constant<FC>(K.ix), K));
			acc = weaken<FM>(mul(acc, sm, K));
		}
		u32 *dst = scratch + (size_t)i * NL;
#pragma unroll
		for (int k = 0; k < NL; k++) {
			dst[k] = acc.l[k];
		}
	}
	FM inv = jacg::inv<PB>(acc, K);   // (s_0 ... s_last)^-1 = x^(q - 2): the unique inverse, nn_modinv's (nn/nn_modinv.c:220)
#pragma unroll 1
	for (int j = kp - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * lanes;
		if (i >= A.n) {
			continue;
		}
		const u8 *sig = A.sigs + (size_t)i * 2 * qlen;
		u32 rw[NW], sw[NW];
		load_be<NW>(sig, qlen, rw);
		load_be<NW>(sig + qlen, qlen, sw);
		const bool ok = in_range(rw) & in_range(sw);
		u32 u1w[NW], u2w[NW];
#pragma unroll
		for (int k = 0; k < NW; k++) {
			u1w[k] = u2w[k] = 0;
		}
		if (ok) {
			FM pre = onem;
			if (j > 0) {
				const u32 *src = scratch + (size_t)(i - lanes) * NL;
#pragma unroll
				for (int k = 0; k < NL; k++) {
					pre.l[k] = src[k];
				}
			}
			const FM sm = weaken<FM>(mul(from_words<PB, NW>(sw), constant<FC>(K.ix), K));
			const FM sinv = weaken<FM>(mul(inv, pre, K));     // Montgomery form of 1 / s
			inv = weaken<FM>(mul(inv, sm, K));
			// e = the leftmost |q| bits of the digest, mod q (sig/ecdsa_common.c:760-778): one conditional subtraction
			const int elen = hlen < qlen ? hlen : qlen;
			u32 ew[NW];
			load_be<NW>(A.digests + (size_t)i * hlen, elen, ew);
			const int rshift = (8 * hlen > (int)A.qbits) ? (8 * elen - (int)A.qbits) : 0;   // 0 .. 7
			if (rshift > 0) {
#pragma unroll
				for (int k = 0; k < NW; k++) {
					const u32 hi = (k + 1 < NW) ? ew[k + 1] : 0u;
					ew[k] = (ew[k] >> rshift) | (hi << (32 - rshift));
				}
			}
			{
				u32 tw[NW], borrow = 0;
#pragma unroll
				for (int k = 0; k < NW; k++) {
					const uint64_t d = (uint64_t)ew[k] - qw[k] - borrow;
					tw[k] = (u32)d;
					borrow = (u32)(d >> 63);
				}
#pragma unroll
				for (int k = 0; k < NW; k++) {
					ew[k] = borrow ? ew[k] : tw[k];
				}
			}
			u32 dg[NL];
			canonical_digits(dg, mul(from_words<PB, NW>(ew), sinv, K), K);     // plain x Montgomery = plain e / s
			to_words<NL, NW>(u1w, dg);
			canonical_digits(dg, mul(from_words<PB, NW>(rw), sinv, K), K);
			to_words<NL, NW>(u2w, dg);
		}
		store_be<NW>(A.u1 + (size_t)i * qlen, qlen, u1w);
		store_be<NW>(A.u2 + (size_t)i * qlen, qlen, u2w);
		A.flags[i] = ok ? 0 : 1;
	}
}

hipError_t G29_CAT(ecamd_g29_ecdsa_prep_, G29_TAG)(int qgslot, const EcamdEcdsaPrepArgs &a, uint32_t *scratch, int kp, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const uint32_t lanes = (a.n + (uint32_t)kp - 1) / (uint32_t)kp;
	hipLaunchKernelGGL(k_ecdsa_prep_g<G29_PB>, dim3((lanes + 63) / 64), dim3(64), 0, s, a, scratch, qgslot, lanes, kp);
	return hipGetLastError();
}
#endif

// ------------------------------------------------------------------------------------------
// Round 4: prj_pt_import_from_buf + prj_pt_unique for n projective X || Y || Z triples on this unit's field (what k_prj_import<NW> of
// ecamd_kernels.hip does on saturated words with one Fermat inversion per item): range check of every coordinate, the projective
// curve equation Y^2 Z = X^3 + a X Z^2 + b Z^3 (on the unit's image curve when the handle computes on an isomorphic one: the factors
// u^2, u^3 ride on ix, iy and cancel in the equation), Z = 0 -> infinity, and the affine form by ONE inversion per `items` triples
// (Montgomery's trick; the prefix product of a triple rests in its own output slot until the way back).  The libecc-typed layer hands
// over keys and points as live projective limbs: at 2^20 keys the saturated kernel was a quarter of a secp256r1 verification's
// kernel time (profiles/r4i_typed_boundary.md).  Lane t owns the triples t, t + nthreads, ...
// ------------------------------------------------------------------------------------------
template <int PB, int FLAV> __global__ __launch_bounds__(64) void k_prj_import_g(EcamdPrjInArgs A, int gslot, u32 nthreads, int items)
{
	typedef Lay<PB> L;
	typedef typename Cls<PB>::FM FM;
	typedef typename Cls<PB>::FC FC;
	constexpr int NL = L::NL, NW = L::NW;
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const CurveG<NL> &K = TabGP<PB>::get(gslot);
	const FC onec = constant<FC>(K.one);
	const int clen = (int)A.clen;
	// one triple: Montgomery / image-curve forms of X, Y, Z and its status (0 finite / 1 error / 2 infinity)
	auto load = [&](u32 i, FM &xm, FM &ym, FM &zm) -> u32 {
		const u8 *src = A.in + (size_t)i * 3 * clen;
		u32 xw[NW], yw[NW], zw[NW];
		load_be<NW>(src, clen, xw);
		load_be<NW>(src + clen, clen, yw);
		load_be<NW>(src + 2 * clen, clen, zw);
		const auto xd = from_words<PB, NW>(xw), yd = from_words<PB, NW>(yw), zd = from_words<PB, NW>(zw);
		u32 bx = 0, by = 0, bz = 0, xnz = 0, ynz = 0, znz = 0;
#pragma unroll
		for (int j = 0; j < NL; j++) {
			bx = (xd.l[j] - K.p[j] - bx) >> 31;
			by = (yd.l[j] - K.p[j] - by) >> 31;
			bz = (zd.l[j] - K.p[j] - bz) >> 31;
			xnz |= xd.l[j];
			ynz |= yd.l[j];
			znz |= zd.l[j];
		}
		bool ok = (bx != 0) & (by != 0) & (bz != 0);
		ok = ok & ((words_excess<PB, NW>(xw) | words_excess<PB, NW>(yw) | words_excess<PB, NW>(zw)) == 0);
		xm = weaken<FM>(mul(xd, constant<FC>(K.ix), K));
		ym = weaken<FM>(mul(yd, constant<FC>(K.iy), K));
		zm = weaken<FM>(mul(zd, constant<FC>(K.r2), K));
		{
			// Y^2 Z == (X^2 + a Z^2) X + b Z^3
			const FM z2 = weaken<FM>(sqr(zm, K));
			const auto t1 = mulc(carry(add(sqr(xm, K), mul(constant<FC>(K.a), z2, K))), xm, K);
			const auto rhs = add(t1, mul(mul(constant<FC>(K.b), z2, K), zm, K));
			const auto dif = carry(sub_auto<1>(rhs, mul(sqr(ym, K), zm, K), K));
			ok = ok & is_zero_mulout(mulc(dif, onec, K), K);
		}
		if (!ok) {
			return 1u;
		}
		if (znz == 0) {
			// (0 : 0 : 0) satisfies the equation too: prj_pt_unique reports it as infinity, a multiplication fails on it
			return (A.for_mul && xnz == 0 && ynz == 0) ? 1u : 2u;
		}
		return 0u;
	};
	FM c = weaken<FM>(onec);
#pragma unroll 1
	for (int j = 0; j < items; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			break;
		}
		FM xm, ym, zm;
		const u32 st = load(i, xm, ym, zm);
		A.pre[i] = (u8)st;
		u8 *dst = A.aff + (size_t)i * 2 * clen;
		// the prefix product BEFORE this triple parks in its output slot (NL words fit 2 clen bytes for every size libecc has)
#pragma unroll
		for (int w = 0; w < NL; w++) {
			const u32 v = c.l[w];
			dst[4 * w] = (u8)v;
			dst[4 * w + 1] = (u8)(v >> 8);
			dst[4 * w + 2] = (u8)(v >> 16);
			dst[4 * w + 3] = (u8)(v >> 24);
		}
		if (st == 0) {
			c = weaken<FM>(mul(c, zm, K));
		}
	}
	FM tinv = inv<PB>(c, K);
	const FC ex = constant<FC>(K.ex), ey = constant<FC>(K.ey);
#pragma unroll 1
	for (int j = items - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			continue;
		}
		u8 *dst = A.aff + (size_t)i * 2 * clen;
		if (A.pre[i] != 0) {
			for (int b = 0; b < 2 * clen; b++) {
				dst[b] = 0;
			}
			continue;
		}
		FM xm, ym, zm, cp;
		(void)load(i, xm, ym, zm);
#pragma unroll
		for (int w = 0; w < NL; w++) {
			cp.l[w] = (u32)dst[4 * w] | ((u32)dst[4 * w + 1] << 8) | ((u32)dst[4 * w + 2] << 16) | ((u32)dst[4 * w + 3] << 24);
		}
		const FM zi = weaken<FM>(mul(tinv, cp, K));
		tinv = weaken<FM>(mul(tinv, zm, K));
		u32 dg[NL], ow[NW];
		canonical_digits(dg, mul(mul(xm, zi, K), ex, K), K);
		to_words<NL, NW>(ow, dg);
		store_be<NW>(dst, clen, ow);
		canonical_digits(dg, mul(mul(ym, zi, K), ey, K), K);
		to_words<NL, NW>(ow, dg);
		store_be<NW>(dst + clen, clen, ow);
	}
}

hipError_t G29_CAT(ecamd_g29_msm_, G29_TAG)(int gslot, int phase, const EcamdMsmArgs &a, uint32_t *tmp, const uint8_t *gen, const uint8_t *gen_status,
					    uint8_t *verdict, uint32_t *sum_out, hipStream_t s)
{
	constexpr uint32_t RECW = (uint32_t)MsmLay<G29_PB>::RECW;
	if (a.n == 0) {
		return hipErrorInvalidValue;
	}
	if (phase == 0) {
		hipLaunchKernelGGL((k_msm_table_g<G29_PB, G29_FLAV>), dim3((2 * a.n + 63) / 64), dim3(64), 0, s, a, gslot);
	} else if (phase == 1) {
		hipLaunchKernelGGL((k_msm_loop_g<G29_PB, G29_FLAV>), dim3((a.L + 63) / 64), dim3(64), 0, s, a, gslot);
	} else if (phase == 10) {
		const uint32_t cnt = a.pt_count ? a.pt_count : 2 * a.n;
		hipLaunchKernelGGL((k_bkt_points_g<G29_PB, G29_FLAV>), dim3((cnt + 63) / 64), dim3(64), 0, s, a, gslot);
	} else if (phase == 11) {
		const uint32_t lanes = (a.win_count ? a.win_count : a.nwin) << a.c;
		hipLaunchKernelGGL((k_bkt_accum_g<G29_PB, G29_FLAV>), dim3((lanes + 63) / 64), dim3(64), 0, s, a, gslot);
	} else if (phase == 13) {
		// the comparison with -[c]G alone (the total of phase 12 rests behind the window records of the reduction's last half)
		hipLaunchKernelGGL((k_msm_final_g<G29_PB, G29_FLAV>), dim3(1), dim3(64), 0, s, (const uint32_t *)tmp, gen, gen_status, a.clen, (const uint32_t *)a.flagword,
				   verdict, sum_out, gslot, a.cof_dbl);
	} else if (phase == 12 || phase == 14) {
		// the reduction: levels of ecamd_bkt_fold() entries over the bucket sums of the windows [win_first, win_first + win_count) (count 0: all of them, and the
		// total behind it), ping-pong between that range's share of the two halves of a.red; then the windows.  Phase 14: the windows' total
		// alone -- a caller that reduces two window ranges on two streams (the key-only windows while the others are still accumulating:
		// their doubling chains, 2^(c win), are the long ones) joins them there.  Layout of a.red: two halves of nwin x per_win words, the
		// second one followed by one record per window, the total, and its copy (what phase 13 compares with -[c]G).
		const uint32_t fold = ecamd_bkt_fold(), fold_log2 = (uint32_t)__builtin_ctz(fold);
		const uint32_t nb16 = ((1u << a.c) + fold - 1) / fold;
		const size_t per_win = 2 * (size_t)nb16 * RECW, halfw = (size_t)a.red_words / 2;
		if ((size_t)a.nwin * per_win + ((size_t)a.nwin + 2) * RECW > halfw) {
			return hipErrorInvalidValue;
		}
		uint32_t *wout = a.red + (size_t)a.red_words - ((size_t)a.nwin + 2) * RECW;
		if (phase == 12) {
			const uint32_t wf = a.win_count ? a.win_first : 0u, wc = a.win_count ? a.win_count : a.nwin;
			if (wf + wc > a.nwin) {
				return hipErrorInvalidValue;
			}
			BktLevel V = {};
			V.nwin = wc;
			V.fold = fold;
			V.inT = a.bsum + ((size_t)wf << a.c) * RECW;
			V.Lin = 1u << a.c;
			uint32_t *half[2] = {a.red + (size_t)wf * per_win, a.red + halfw + (size_t)wf * per_win};
			int hsel = 0;
			while (V.Lin > 1) {
				V.Lout = (V.Lin + fold - 1) / fold;
				uint32_t *o = half[hsel];
				const size_t arr = (size_t)wc * V.Lout * RECW;
				if ((2 + (size_t)V.ncarry) * arr > (size_t)wc * per_win || V.ncarry + 1 > BKT_MAXCARRY) {
					return hipErrorInvalidValue;
				}
				V.outT = o;
				V.outU = o + arr;
				for (uint32_t k = 0; k < V.ncarry; k++) {
					V.outC[k] = o + (2 + (size_t)k) * arr;
				}
				const uint32_t lanes = wc * V.Lout;
				hipLaunchKernelGGL((k_bkt_reduce_g<G29_PB, G29_FLAV>), dim3((lanes + 63) / 64, 1 + V.ncarry), dim3(64), 0, s, V, gslot);
				// the next level folds this level's T; this level's U joins the carry arrays
				V.inT = V.outT;
				for (uint32_t k = 0; k < V.ncarry; k++) {
					V.inC[k] = V.outC[k];
				}
				V.inC[V.ncarry] = V.outU;
				V.ncarry++;
				V.Lin = V.Lout;
				hsel ^= 1;
			}
			// now every array holds one record per window: inC[0 .. ncarry - 2] the totals of the earlier levels' U, inC[ncarry - 1] the last U
			BktWindows W = {};
			W.nwin = wc;
			W.win_base = wf;
			W.c = a.c;
			W.fold_log2 = fold_log2;
			W.U = V.inC[V.ncarry - 1];
			W.ncarry = V.ncarry - 1;
			for (uint32_t k = 0; k + 1 < V.ncarry; k++) {
				W.C[k] = V.inC[k];
			}
			W.out = wout + (size_t)wf * RECW;
			hipLaunchKernelGGL((k_bkt_window_g<G29_PB, G29_FLAV>), dim3((wc + 63) / 64), dim3(64), 0, s, W, gslot);
			if (a.win_count) {
				return hipGetLastError();
			}
		}
		uint32_t *tot = wout + (size_t)a.nwin * RECW;
		hipLaunchKernelGGL((k_bkt_total_g<G29_PB, G29_FLAV>), dim3(1), dim3(64), 0, s, (const uint32_t *)wout, a.nwin, tot, a.flagword, gslot);
		// (phase 13 compares it with -[c]G; the caller finds it at a.red + a.red_words - RECW: copied there)
		hipLaunchKernelGGL((k_bkt_total_g<G29_PB, G29_FLAV>), dim3(1), dim3(64), 0, s, (const uint32_t *)tot, 1u, a.red + (size_t)a.red_words - RECW, a.flagword, gslot);
	} else {
		// tree sum, ping-pong between a.rec and tmp
		uint32_t count = a.L;
		uint32_t *src = a.rec, *dst = tmp;
		while (count > 1) {
			const uint32_t outc = (count + MSM_FAN - 1) / MSM_FAN;
			hipLaunchKernelGGL((k_msm_sum_g<G29_PB, G29_FLAV>), dim3((outc + 63) / 64), dim3(64), 0, s, (const uint32_t *)src, count, dst, a.flagword, gslot);
			uint32_t *t = src;
			src = dst;
			dst = t;
			count = outc;
		}
		(void)RECW;
		hipLaunchKernelGGL((k_msm_final_g<G29_PB, G29_FLAV>), dim3(1), dim3(64), 0, s, (const uint32_t *)src, gen, gen_status, a.clen, (const uint32_t *)a.flagword,
				   verdict, sum_out, gslot, a.cof_dbl);
	}
	return hipGetLastError();
}

hipError_t G29_CAT(ecamd_g29_prj_import_, G29_TAG)(int gslot, const EcamdPrjInArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	static_assert(4 * Lay<G29_PB>::NL <= 2 * ((G29_PB + 7) / 8), "the prefix product must fit the output slot");
	const int items = a.n >= (1u << 19) ? 8 : (a.n >= (1u << 17) ? 4 : 2);
	const uint32_t nthreads = (a.n + (uint32_t)items - 1) / (uint32_t)items;
	hipLaunchKernelGGL((k_prj_import_g<G29_PB, G29_FLAV>), dim3((nthreads + 63) / 64), dim3(64), 0, s, a, gslot, nthreads, items);
	return hipGetLastError();
}

hipError_t G29_CAT(ecamd_g29_upload_, G29_TAG)(int slot, const void *img, size_t bytes)
{
	typedef CurveG<Lay<G29_PB>::NL> CK;
	if (bytes != sizeof(CK) || slot < 0 || slot >= G29_SLOTS) {
		return hipErrorInvalidValue;
	}
	return hipMemcpyToSymbol(HIP_SYMBOL(G29_CAT(g_g29_, G29_TAG)), img, bytes, (size_t)slot * sizeof(CK), hipMemcpyHostToDevice);
}

hipError_t G29_CAT(ecamd_g29_launch_, G29_TAG)(int gslot, const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev)
{
	const dim3 grid((a.n + 63) / 64), block(64);
	const uint32_t nthreads = (a.n + FING_K - 1) / FING_K;
	const dim3 fgrid((nthreads + 63) / 64);
	// event slots as in ecamd_launch_smul_p256: [0] start, [3] after the loop kernel, [4] after finalisation
	const bool pipeline_events =
#if defined(G29_K256) || defined(G29_P25519) || defined(G29_JACTAB)
		false;
#else
		!(a.lut && (a.lut_kind == 1 || a.lut_kind == 3));
#endif
	if (ev) {
		(void)hipEventRecord(ev[0], s);
		if (!pipeline_events) {
			(void)hipEventRecord(ev[1], s);
			(void)hipEventRecord(ev[2], s);
		}
	}
	if (a.lut && a.lut_kind == 1) {
		hipLaunchKernelGGL((k_comb_g<G29_PB, G29_FLAV>), grid, block, 0, s, a, gslot);
	} else if (a.lut && a.lut_kind == 3) {
		hipLaunchKernelGGL((k_comb_g<G29_PB, G29_FLAV, true>), grid, block, 0, s, a, gslot);   // secret scalars: the scanned 4-bit comb
	} else {
#if defined(G29_K256) || defined(G29_P25519) || defined(G29_JACTAB)
		if (a.lut_kind == 2) {
			return hipErrorInvalidValue;   // the fused double-scalar loop needs the affine-table pipeline
		}
		// secp256k1's flavour (no room for the mixed addition), the 2^255 - 19 flavour (measured: the two extra passes cost what
		// the cheaper loop saves, 62.7 against 63.1 M/s at 2^20 and -6 % at 2^16; its comb kernel does use the mixed addition)
		// and the A/B build: Jacobian table, one kernel
		hipLaunchKernelGGL((k_smul_g<G29_PB, G29_FLAV>), grid, block, 0, s, a, gslot);
#else
		// affine-table pipeline; a.stg: n x LayA::AITEMW words
		hipLaunchKernelGGL((k_table_g<G29_PB, G29_FLAV>), grid, block, 0, s, a, gslot);
		if (ev) {
			(void)hipEventRecord(ev[1], s);
		}
		const int aitems = a.n >= (1u << 19) ? AFFG_K : (a.n >= (1u << 17) ? 4 : 2);
		const uint32_t athreads = (a.n + (uint32_t)aitems - 1) / (uint32_t)aitems;
		hipLaunchKernelGGL((k_affine_g<G29_PB, G29_FLAV>), dim3((athreads + 63) / 64), block, 0, s, a, gslot, athreads, aitems);
		if (ev) {
			(void)hipEventRecord(ev[2], s);
		}
		if (a.lut && a.lut_kind == 2) {
			// [u2]Q by the window loop, then + [u1]G from the comb table
			hipLaunchKernelGGL((k_loop_g<G29_PB, G29_FLAV, false>), grid, block, 0, s, a, gslot);
			hipLaunchKernelGGL((k_comb_add_g<G29_PB, G29_FLAV>), grid, block, 0, s, a, gslot);
		} else if (a.masked) {
			hipLaunchKernelGGL((k_loop_g<G29_PB, G29_FLAV, true>), grid, block, 0, s, a, gslot);
		} else {
			hipLaunchKernelGGL((k_loop_g<G29_PB, G29_FLAV, false>), grid, block, 0, s, a, gslot);
		}
#endif
	}
	if (ev) {
		(void)hipEventRecord(ev[3], s);
	}
	hipLaunchKernelGGL((k_finalize_g<G29_PB, G29_FLAV>), fgrid, block, 0, s, a, gslot, nthreads);
	if (ev) {
		(void)hipEventRecord(ev[4], s);
	}
	return hipGetLastError();
}

hipError_t G29_CAT(ecamd_g29_comb_build_, G29_TAG)(int gslot, const uint8_t *pts, uint32_t n, uint32_t clen, uint32_t *table, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL((k_comb_build_g<G29_PB, G29_FLAV>), dim3((n + 63) / 64), dim3(64), 0, s, pts, n, clen, table, gslot);
	return hipGetLastError();
}
#endif

#ifdef G29_DISPATCH
// ---- host-side dispatch on |p| ----
#define G29_FOR_PB(X) X(192) X(224) X(255) X(256) X(320) X(384) X(448) X(511) X(512) X(521)
#define X(PB) \
	hipError_t ecamd_g29_upload_##PB(int slot, const void *img, size_t bytes); \
	hipError_t ecamd_g29_launch_##PB(int gslot, const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev); \
	hipError_t ecamd_g29_comb_build_##PB(int gslot, const uint8_t *pts, uint32_t n, uint32_t clen, uint32_t *table, hipStream_t s);
G29_FOR_PB(X)
X(521m)
X(255c)
X(384n)
X(224s)
X(192s)
X(256k)
X(448g)
#undef X
hipError_t ecamd_g29_ed_fin_255c(int gslot, const EcamdEdFinArgs &a, hipStream_t s);
hipError_t ecamd_g29_ed_fin_448g(int gslot, const EcamdEdFinArgs &a, hipStream_t s);

// the tail of an EdDSA verification on the 2^255 - 19 (flavour 2) or the Goldilocks (flavour 5) unit
hipError_t ecamd_launch_ed_fin_g29(int flavour, int gslot, const EcamdEdFinArgs &a, hipStream_t s)
{
	if (flavour == 2) {
		return ecamd_g29_ed_fin_255c(gslot, a, s);
	}
	if (flavour == 5) {
		return ecamd_g29_ed_fin_448g(gslot, a, s);
	}
	return hipErrorInvalidValue;
}

int ecamd_g29_supported(int pbits)
{
	switch (pbits) {
#define X(PB) case PB: return 1;
		G29_FOR_PB(X)
#undef X
	default: return 0;
	}
}
int ecamd_g29_nl(int pbits, int flavour) { return g29::nl_for_flavour(pbits, flavour); }
int ecamd_g29_slots(void) { return G29_SLOTS; }
uint32_t ecamd_g29_table_words(int pbits, int flavour)
{
	const uint32_t nw = (uint32_t)((pbits + 31) / 32);
	return 8u * (uint32_t)(((3 * g29::nl_for_flavour(pbits, flavour) + 3) / 4) * 4) + ((2u * nw + 2u + 3u) / 4u) * 4u;   // Lay<PB>::ITEMW
}
// affine window table of the mixed-addition pipeline: 8 entries of 2 NL digits, padded to 16 bytes (LayA<PB>::AITEMW)
uint32_t ecamd_g29_affine_words(int pbits, int flavour)
{
	return 8u * (uint32_t)(((2 * g29::nl_for_flavour(pbits, flavour) + 3) / 4) * 4);
}
uint32_t ecamd_g29_max_slen(int pbits) { return 8u * (uint32_t)((pbits + 31) / 32) + 4u; }
uint32_t ecamd_g29_comb_max_slen(int pbits) { return 4u * (uint32_t)((pbits + 31) / 32); }
size_t ecamd_g29_image_bytes(int pbits, int flavour)
{
	return (size_t)((10 + g29::NBIAS) * g29::nl_for_flavour(pbits, flavour) + 4) * 4;
}

// 'flavour' 1 selects the secp521r1 (p = 2^521 - 1) instantiation, 2 the p = 2^255 - 19 one, 3 secp384r1's, 4 secp256k1's, 5 WEI448's,
// 6 secp224r1's, 7 secp192r1's
hipError_t ecamd_g29_upload(int pbits, int slot, const void *img, size_t bytes, int flavour)
{
	if (pbits == 521 && flavour == 1) {
		return ecamd_g29_upload_521m(slot, img, bytes);
	}
	if (pbits == 255 && flavour == 2) {
		return ecamd_g29_upload_255c(slot, img, bytes);
	}
	if (pbits == 384 && flavour == 3) {
		return ecamd_g29_upload_384n(slot, img, bytes);
	}
	if (pbits == 224 && flavour == 6) {
		return ecamd_g29_upload_224s(slot, img, bytes);
	}
	if (pbits == 192 && flavour == 7) {
		return ecamd_g29_upload_192s(slot, img, bytes);
	}
	if (pbits == 256 && flavour == 4) {
		return ecamd_g29_upload_256k(slot, img, bytes);
	}
	if (pbits == 448 && flavour == 5) {
		return ecamd_g29_upload_448g(slot, img, bytes);
	}
	switch (pbits) {
#define X(PB) case PB: return ecamd_g29_upload_##PB(slot, img, bytes);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}

hipError_t ecamd_launch_smul_g29(int pbits, int gslot, const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev, int flavour)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	if (pbits == 521 && flavour == 1) {
		return ecamd_g29_launch_521m(gslot, a, s, ev);
	}
	if (pbits == 255 && flavour == 2) {
		return ecamd_g29_launch_255c(gslot, a, s, ev);
	}
	if (pbits == 384 && flavour == 3) {
		return ecamd_g29_launch_384n(gslot, a, s, ev);
	}
	if (pbits == 224 && flavour == 6) {
		return ecamd_g29_launch_224s(gslot, a, s, ev);
	}
	if (pbits == 192 && flavour == 7) {
		return ecamd_g29_launch_192s(gslot, a, s, ev);
	}
	if (pbits == 256 && flavour == 4) {
		return ecamd_g29_launch_256k(gslot, a, s, ev);
	}
	if (pbits == 448 && flavour == 5) {
		return ecamd_g29_launch_448g(gslot, a, s, ev);
	}
	switch (pbits) {
#define X(PB) case PB: return ecamd_g29_launch_##PB(gslot, a, s, ev);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}
#endif

#ifdef G29_DISPATCH
// comb table geometry: 2 NW windows of 32768 entries + 1, entry = 2 NL digits padded to 4 words
uint32_t ecamd_g29_comb_entries(int pbits) { return (uint32_t)(2 * ((pbits + 31) / 32)) * 32768u + 1u; }
uint32_t ecamd_g29_comb_entry_words(int pbits, int flavour)
{
	return (uint32_t)(((2 * g29::nl_for_flavour(pbits, flavour) + 3) / 4) * 4);
}
#define X(PB) hipError_t ecamd_g29_ecdsa_prep_##PB(int qgslot, const EcamdEcdsaPrepArgs &a, uint32_t *scratch, int kp, hipStream_t s);
G29_FOR_PB(X)
#undef X
// k_ecdsa_prep_g on the dense unit of qbits bits (the order q in constant slot qgslot of that unit); scratch: n x ecamd_g29_nl(qbits, 0) words
hipError_t ecamd_g29_ecdsa_prep(int qbits, int qgslot, const EcamdEcdsaPrepArgs &a, uint32_t *scratch, int kp, hipStream_t s)
{
	switch (qbits) {
#define X(PB) case PB: return ecamd_g29_ecdsa_prep_##PB(qgslot, a, scratch, kp, s);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}

#define X(PB) hipError_t ecamd_g29_msm_##PB(int gslot, int phase, const EcamdMsmArgs &a, uint32_t *tmp, const uint8_t *gen, const uint8_t *gen_status, \
					   uint8_t *verdict, uint32_t *sum_out, hipStream_t s);
G29_FOR_PB(X)
X(521m)
X(255c)
X(384n)
X(224s)
X(192s)
X(256k)
X(448g)
#undef X
uint32_t ecamd_g29_bkt_point_words(int pbits, int flavour)   // BktLay<PB>::PSTRIDE
{
	const uint32_t w = (uint32_t)(((2 * g29::nl_for_flavour(pbits, flavour) + 3) / 4) * 4);
	return w <= 16 ? 16u : (w <= 32 ? 32u : (w <= 64 ? 64u : 128u));
}
uint32_t ecamd_g29_msm_rec_words(int pbits, int flavour) { return (uint32_t)(((3 * g29::nl_for_flavour(pbits, flavour) + 3) / 4) * 4) + 4u; }   // MsmLay<PB>::RECW
// the Schnorr-type multi-scalar multiplication on the unit (pbits, flavour)
hipError_t ecamd_launch_msm_g29(int pbits, int gslot, int flavour, int phase, const EcamdMsmArgs &a, uint32_t *tmp, const uint8_t *gen,
				const uint8_t *gen_status, uint8_t *verdict, uint32_t *sum_out, hipStream_t s)
{
#define Y(TAG) return ecamd_g29_msm_##TAG(gslot, phase, a, tmp, gen, gen_status, verdict, sum_out, s)
	if (pbits == 521 && flavour == 1) {
		Y(521m);
	}
	if (pbits == 255 && flavour == 2) {
		Y(255c);
	}
	if (pbits == 384 && flavour == 3) {
		Y(384n);
	}
	if (pbits == 224 && flavour == 6) {
		Y(224s);
	}
	if (pbits == 192 && flavour == 7) {
		Y(192s);
	}
	if (pbits == 256 && flavour == 4) {
		Y(256k);
	}
	if (pbits == 448 && flavour == 5) {
		Y(448g);
	}
	switch (pbits) {
#define X(PB) case PB: Y(PB);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
#undef Y
}

#define X(PB) hipError_t ecamd_g29_prj_import_##PB(int gslot, const EcamdPrjInArgs &a, hipStream_t s);
G29_FOR_PB(X)
X(521m)
X(255c)
X(384n)
X(224s)
X(192s)
X(256k)
X(448g)
#undef X
// k_prj_import_g on the unit (pbits, flavour)
hipError_t ecamd_g29_prj_import(int pbits, int gslot, const EcamdPrjInArgs &a, hipStream_t s, int flavour)
{
	if (pbits == 521 && flavour == 1) {
		return ecamd_g29_prj_import_521m(gslot, a, s);
	}
	if (pbits == 255 && flavour == 2) {
		return ecamd_g29_prj_import_255c(gslot, a, s);
	}
	if (pbits == 384 && flavour == 3) {
		return ecamd_g29_prj_import_384n(gslot, a, s);
	}
	if (pbits == 224 && flavour == 6) {
		return ecamd_g29_prj_import_224s(gslot, a, s);
	}
	if (pbits == 192 && flavour == 7) {
		return ecamd_g29_prj_import_192s(gslot, a, s);
	}
	if (pbits == 256 && flavour == 4) {
		return ecamd_g29_prj_import_256k(gslot, a, s);
	}
	if (pbits == 448 && flavour == 5) {
		return ecamd_g29_prj_import_448g(gslot, a, s);
	}
	switch (pbits) {
#define X(PB) case PB: return ecamd_g29_prj_import_##PB(gslot, a, s);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}

hipError_t ecamd_g29_comb_build(int pbits, int gslot, const uint8_t *pts, uint32_t n, uint32_t clen, uint32_t *table,
				hipStream_t s, int flavour)
{
	if (pbits == 521 && flavour == 1) {
		return ecamd_g29_comb_build_521m(gslot, pts, n, clen, table, s);
	}
	if (pbits == 255 && flavour == 2) {
		return ecamd_g29_comb_build_255c(gslot, pts, n, clen, table, s);
	}
	if (pbits == 384 && flavour == 3) {
		return ecamd_g29_comb_build_384n(gslot, pts, n, clen, table, s);
	}
	if (pbits == 224 && flavour == 6) {
		return ecamd_g29_comb_build_224s(gslot, pts, n, clen, table, s);
	}
	if (pbits == 192 && flavour == 7) {
		return ecamd_g29_comb_build_192s(gslot, pts, n, clen, table, s);
	}
	if (pbits == 256 && flavour == 4) {
		return ecamd_g29_comb_build_256k(gslot, pts, n, clen, table, s);
	}
	if (pbits == 448 && flavour == 5) {
		return ecamd_g29_comb_build_448g(gslot, pts, n, clen, table, s);
	}
	switch (pbits) {
#define X(PB) case PB: return ecamd_g29_comb_build_##PB(gslot, pts, n, clen, table, s);
		G29_FOR_PB(X)
#undef X
	default: return hipErrorInvalidValue;
	}
}
#endif
