// libecc_amd/csrc/ecamd_p256_kernel.hip -- secp256r1 fast path of the batched prj_pt_mul.
//
// One scalar multiplication per lane (replaces prj_pt_import_from_aff_buf -> prj_pt_mul ->
// prj_pt_unique -> prj_pt_export_to_aff_buf, curves/prj_pt.c:511,1759,241,600 of the reference):
//   * field: radix-2^29 lazy Montgomery arithmetic (ecamd_u29.cuh), v_mad_u64_u32 only;
//   * group: Jacobian a = -3 doubling (4M + 4S) and addition (12M + 4S) (ecamd_p256.cuh);
//   * scalar: signed fixed window w = 4 over k' = k + 0x88...8 (digit = nibble - 8 in [-8, 7]),
//     left to right, 4 doublings + 1 addition per window, table [1..8]P per lane;
//   * table: per-lane contiguous 8 x 28 words in a global scratch buffer (7 x 16-byte
//     loads per look-up, every fetched cache line fully used by the lane that fetched it);
//   * output: one Fermat inversion by addition chain, affine X||Y big-endian.
// Exceptional pairs of the incomplete Jacobian addition (accumulator == +-table entry) cannot be
// produced by scalars below the group order and random points, but edge scalars (k >= q) can:
// such a lane is detected exactly (Z3 == 0 with both inputs finite), marked ECAMD_REDO and
// recomputed by the complete-formula kernel k_smul<8> in the same call (ecamd_host.cpp).
#include <hip/hip_runtime.h>
#include "ecamd_p256.cuh"
#include "ecamd_internal.h"

using namespace p256;

typedef uint8_t u8;

#define TBL_WORDS_PER_ENTRY 28
#define TBL_ENTRIES 8

// 32 big-endian bytes -> 8 little-endian words
static __device__ __forceinline__ void load_be256(const u8 *src, u32 *w)
{
	if ((((uintptr_t)src) & 15) == 0) {
		const uint4 a = *(const uint4 *)src, b = *(const uint4 *)(src + 16);
		w[7] = __builtin_bswap32(a.x);
		w[6] = __builtin_bswap32(a.y);
		w[5] = __builtin_bswap32(a.z);
		w[4] = __builtin_bswap32(a.w);
		w[3] = __builtin_bswap32(b.x);
		w[2] = __builtin_bswap32(b.y);
		w[1] = __builtin_bswap32(b.z);
		w[0] = __builtin_bswap32(b.w);
	} else {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const u8 *q = src + 4 * (7 - i);
			w[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
		}
	}
}

static __device__ __forceinline__ void store_be256(u8 *dst, const u32 *w)
{
	if ((((uintptr_t)dst) & 15) == 0) {
		uint4 a, b;
		a.x = __builtin_bswap32(w[7]);
		a.y = __builtin_bswap32(w[6]);
		a.z = __builtin_bswap32(w[5]);
		a.w = __builtin_bswap32(w[4]);
		b.x = __builtin_bswap32(w[3]);
		b.y = __builtin_bswap32(w[2]);
		b.z = __builtin_bswap32(w[1]);
		b.w = __builtin_bswap32(w[0]);
		*(uint4 *)dst = a;
		*(uint4 *)(dst + 16) = b;
	} else {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			u8 *q = dst + 4 * (7 - i);
			q[0] = (u8)(w[i] >> 24);
			q[1] = (u8)(w[i] >> 16);
			q[2] = (u8)(w[i] >> 8);
			q[3] = (u8)w[i];
		}
	}
}

// value < p ?  (fp_import_from_buf rejects >= p, fp/fp.c:441-442)
static __device__ __forceinline__ bool lt_p(const u32 *w)
{
	constexpr u32 pw[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu};
	u32 borrow = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const uint64_t x = (uint64_t)w[i] - pw[i] - borrow;
		borrow = (u32)(x >> 63);
	}
	return borrow != 0;
}

static __device__ __forceinline__ void tbl_store(u32 *base, int e, const TabEnt &P)
{
	uint4 *d = (uint4 *)(base + e * TBL_WORDS_PER_ENTRY);
	d[0] = make_uint4(P.X.l[0], P.X.l[1], P.X.l[2], P.X.l[3]);
	d[1] = make_uint4(P.X.l[4], P.X.l[5], P.X.l[6], P.X.l[7]);
	d[2] = make_uint4(P.X.l[8], P.Y.l[0], P.Y.l[1], P.Y.l[2]);
	d[3] = make_uint4(P.Y.l[3], P.Y.l[4], P.Y.l[5], P.Y.l[6]);
	d[4] = make_uint4(P.Y.l[7], P.Y.l[8], P.Z.l[0], P.Z.l[1]);
	d[5] = make_uint4(P.Z.l[2], P.Z.l[3], P.Z.l[4], P.Z.l[5]);
	d[6] = make_uint4(P.Z.l[6], P.Z.l[7], P.Z.l[8], 0u);
}

static __device__ __forceinline__ TabEnt tbl_load(const u32 *base, u32 e)
{
	const uint4 *s = (const uint4 *)(base + e * TBL_WORDS_PER_ENTRY);
	const uint4 a = s[0], b = s[1], c = s[2], d = s[3], f = s[4], g = s[5], h = s[6];
	TabEnt P;
	P.X.l[0] = a.x; P.X.l[1] = a.y; P.X.l[2] = a.z; P.X.l[3] = a.w;
	P.X.l[4] = b.x; P.X.l[5] = b.y; P.X.l[6] = b.z; P.X.l[7] = b.w;
	P.X.l[8] = c.x; P.Y.l[0] = c.y; P.Y.l[1] = c.z; P.Y.l[2] = c.w;
	P.Y.l[3] = d.x; P.Y.l[4] = d.y; P.Y.l[5] = d.z; P.Y.l[6] = d.w;
	P.Y.l[7] = f.x; P.Y.l[8] = f.y; P.Z.l[0] = f.z; P.Z.l[1] = f.w;
	P.Z.l[2] = g.x; P.Z.l[3] = g.y; P.Z.l[4] = g.z; P.Z.l[5] = g.w;
	P.Z.l[6] = h.x; P.Z.l[7] = h.y; P.Z.l[8] = h.z;
	return P;
}

template <class T> static __device__ __forceinline__ T sel(bool c, const T &a, const T &b)
{
	T r;
#pragma unroll
	for (int i = 0; i < 9; i++) {
		r.l[i] = c ? a.l[i] : b.l[i];
	}
	return r;
}

// P256_WAVES (2, 3 or 4): register budget as waves per SIMD (512 / 256 -> 2 waves, 168 -> 3, 128 -> 4)
#ifndef P256_WAVES
#define P256_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P256_WAVES, P256_WAVES))) void k_smul_p256(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u8 *out = A.out + (size_t)i * 64;

	// ---- import: X||Y big-endian, coordinates < p, on the curve (curves/prj_pt.c:511-552) ----
	const u8 *pin = A.points + (size_t)i * A.pstride;
	u32 xw[8], yw[8];
	load_be256(pin, xw);
	load_be256(pin + 32, yw);
	bool ok = lt_p(xw) & lt_p(yw);
	const Fcanon r2 = constant<Fcanon>(K::R2);
	const auto xm = mul(from_words(xw), r2);  // Montgomery form, value < 17/16 p
	const auto ym = mul(from_words(yw), r2);
	{
		// y^2 == x^3 - 3x + b  <=>  (x^3 + b + 8p - 3x) - y^2 == 0; the difference goes through one
		// more multiplication (by 1) so that the zero test runs on exact digits
		const auto x3 = mul(sqr(xm), xm);
		const auto rhs = sub<3, 2>(add(x3, constant<Fcanon>(K::BM)), mul_small<3>(xm));
		const auto dif = carry(sub<1, 0>(carry(rhs), sqr(ym)));
		ok = ok & is_zero_mulout(mul(dif, constant<Fcanon>(K::ONE)));
	}
	if (!ok) {
		A.status[i] = 1;
		uint4 z = make_uint4(0, 0, 0, 0);
		if ((((uintptr_t)out) & 15) == 0) {
			((uint4 *)out)[0] = z; ((uint4 *)out)[1] = z; ((uint4 *)out)[2] = z; ((uint4 *)out)[3] = z;
		} else {
			for (int b = 0; b < 64; b++) out[b] = 0;
		}
		return;
	}

	// ---- table [1..8]P, Jacobian ----
	u32 *tb = A.tbl + (size_t)i * (TBL_ENTRIES * TBL_WORDS_PER_ENTRY);
	Jac P1;
	P1.X = weaken<FX>(xm);
	P1.Y = weaken<FY>(ym);
	P1.Z = weaken<FZ>(constant<Fcanon>(K::ONE));
	bool hz;
	const TabEnt T1 = to_tab(P1);
	const FYsel y1 = weaken<FYsel>(T1.Y);
	tbl_store(tb, 0, T1);
	{
		// [2..8]P; intermediate multiples are re-read from the table instead of being kept live
		// (register pressure): 2P, 3P = 2P + P, 4P = 2(2P), 5P = 4P + P, 6P = 2(3P), 7P = 6P + P, 8P = 2(4P)
		Jac Pa = dbl(P1);
		tbl_store(tb, 1, to_tab(Pa));
		Jac Pb = add_jac(Pa, T1.X, y1, T1.Z, hz);
		tbl_store(tb, 2, to_tab(Pb));
		Pa = dbl(Pa);
		tbl_store(tb, 3, to_tab(Pa));
		Pb = add_jac(Pa, T1.X, y1, T1.Z, hz);
		tbl_store(tb, 4, to_tab(Pb));
		{
			const TabEnt t3 = tbl_load(tb, 2);
			Jac P3;
			P3.X = t3.X;
			P3.Y = weaken<FY>(t3.Y);
			P3.Z = t3.Z;
			Pb = dbl(P3);
		}
		tbl_store(tb, 5, to_tab(Pb));
		Pb = add_jac(Pb, T1.X, y1, T1.Z, hz);
		tbl_store(tb, 6, to_tab(Pb));
		Pa = dbl(Pa);
		tbl_store(tb, 7, to_tab(Pa));
	}

	// ---- scalar: k (<= 32 bytes big-endian) -> k' = k + 0x88..8 over its 2*slen nibbles ----
	const u8 *sc = A.scalars + (size_t)i * A.sstride;
	const int slen = (int)A.slen;
	u32 kw[8];
	if (slen == 32) {
		load_be256(sc, kw);
	} else {
#pragma unroll
		for (int w = 0; w < 8; w++) {
			u32 x = 0;
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const int pos = 4 * w + b;
				if (pos < slen) {
					x |= (u32)sc[slen - 1 - pos] << (8 * b);
				}
			}
			kw[w] = x;
		}
	}
	u32 carry_bit = 0;
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			// bytes of 0x88 only where the scalar has bytes
			const int nb = slen - 4 * w;
			const u32 add = (nb >= 4) ? 0x88888888u : (nb == 3 ? 0x00888888u : (nb == 2 ? 0x00008888u : (nb == 1 ? 0x00000088u : 0u)));
			c += (uint64_t)kw[w] + add;
			kw[w] = (u32)c;
			c >>= 32;
		}
		carry_bit = (u32)c;  // only when slen == 32
		if (slen < 32) {
			// the carry out of the top nibble sits just above the scalar's bytes
			const int bit = 8 * slen;
			carry_bit = (kw[bit >> 5] >> (bit & 31)) & 1u;
			// left-align so that the top nibble of the scalar is bits 255..252
			for (int s = slen; s < 32; s++) {
#pragma unroll
				for (int w = 7; w > 0; w--) {
					kw[w] = (kw[w] << 8) | (kw[w - 1] >> 24);
				}
				kw[0] <<= 8;
			}
		}
	}

	// ---- signed fixed window, left to right ----
	Jac acc = P1;
	bool inf = (carry_bit == 0);  // top digit is the carry: 0 or +1
	bool bad = false;
	const int nwin = 2 * slen;
#pragma unroll 1
	for (int t = 0; t < nwin; t++) {
#pragma unroll 1
		for (int d = 0; d < 4; d++) {
			acc = dbl(acc);
		}
		const int dig = (int)(kw[7] >> 28) - 8;  // [-8, 7]
#pragma unroll
		for (int w = 7; w > 0; w--) {
			kw[w] = (kw[w] << 4) | (kw[w - 1] >> 28);
		}
		kw[0] <<= 4;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		const TabEnt T = tbl_load(tb, mag ? mag - 1 : 0);
		const FYsel ty = sel(dig < 0, neg_y(T.Y), weaken<FYsel>(T.Y));
		const Jac S = add_jac(acc, T.X, ty, T.Z, hz);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		// acc = keep ? acc : (use_t ? +-T : S)
		acc.X = sel(keep, acc.X, sel(use_t, T.X, S.X));
		acc.Y = sel(keep, acc.Y, sel(use_t, weaken<FY>(ty), S.Y));
		acc.Z = sel(keep, acc.Z, sel(use_t, T.Z, S.Z));
		inf = inf & keep;
	}

	// ---- hand the Jacobian result to the finalisation kernel (k_p256_finalize) ----
	// status: ECAMD_STATUS_REDO (exceptional pair met: the complete-formula kernel recomputes the item),
	// 2 (infinity), or ECAMD_STATUS_JAC (finite: X, Y, Z stored in the item's first table slot)
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		A.status[i] = 2;
		uint4 z = make_uint4(0, 0, 0, 0);
		if ((((uintptr_t)out) & 15) == 0) {
			((uint4 *)out)[0] = z; ((uint4 *)out)[1] = z; ((uint4 *)out)[2] = z; ((uint4 *)out)[3] = z;
		} else {
			for (int b = 0; b < 64; b++) out[b] = 0;
		}
		return;
	}
	{
		TabEnt R;  // same 28-word record as a table entry, Y left unfolded
		R.X = acc.X;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			R.Y.l[w] = acc.Y.l[w];
		}
		R.Z = acc.Z;
		tbl_store(tb, 0, R);
	}
	A.status[i] = ECAMD_STATUS_JAC;
}

// ------------------------------------------------------------------------------------------
// Finalisation: Jacobian -> affine for K items per lane with ONE field inversion (Montgomery's
// trick): c_j = Z_0 ... Z_j, t = 1 / c_{K-1}, then Z_j^-1 = t c_{j-1}, t *= Z_j going down.
// The scalar-multiplication kernel spends 255 S + 13 M per item on its inversion otherwise
// (6 % of its time); here that cost is shared by K items and each item pays 3 extra mults.
// Prefix products are parked in the item's second table slot.
// ------------------------------------------------------------------------------------------
#define FIN_K 8

static __device__ __forceinline__ Jac load_jac(const u32 *tb)
{
	const TabEnt t = tbl_load(tb, 0);
	Jac P;
	P.X = t.X;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		P.Y.l[w] = t.Y.l[w];
	}
	P.Z = t.Z;
	return P;
}

__global__ __launch_bounds__(64) void k_p256_finalize(EcamdSmulArgs A, u32 nthreads)
{
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	const Fcanon one = constant<Fcanon>(K::ONE);
	// ---- up: prefix products ----
	Fmul c = weaken<Fmul>(one);
#pragma unroll 1
	for (int j = 0; j < FIN_K; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			break;
		}
		u32 *tb = A.tbl + (size_t)i * (TBL_ENTRIES * TBL_WORDS_PER_ENTRY);
		if (A.status[i] == ECAMD_STATUS_JAC) {
			const Jac P = load_jac(tb);
			c = weaken<Fmul>(mul(c, P.Z));
		}
		// park c_j (exact digits) in the second table slot
		uint4 *d = (uint4 *)(tb + TBL_WORDS_PER_ENTRY);
		d[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
		d[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
		d[2] = make_uint4(c.l[8], 0u, 0u, 0u);
	}
	// ---- one inversion ----
	Fmul tinv = inv(c);
	Fcanon plain1;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		plain1.l[w] = (w == 0) ? 1u : 0u;
	}
	// ---- down ----
#pragma unroll 1
	for (int j = FIN_K - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_JAC) {
			continue;
		}
		u32 *tb = A.tbl + (size_t)i * (TBL_ENTRIES * TBL_WORDS_PER_ENTRY);
		const Jac P = load_jac(tb);
		Fmul zi = tinv;
		if (j > 0) {
			const u32 ip = t + (u32)(j - 1) * nthreads;
			const uint4 *s = (const uint4 *)(A.tbl + (size_t)ip * (TBL_ENTRIES * TBL_WORDS_PER_ENTRY) + TBL_WORDS_PER_ENTRY);
			const uint4 a = s[0], b = s[1], cc = s[2];
			Fmul cp;
			cp.l[0] = a.x; cp.l[1] = a.y; cp.l[2] = a.z; cp.l[3] = a.w;
			cp.l[4] = b.x; cp.l[5] = b.y; cp.l[6] = b.z; cp.l[7] = b.w;
			cp.l[8] = cc.x;
			zi = weaken<Fmul>(mul(tinv, cp));
		}
		tinv = weaken<Fmul>(mul(tinv, P.Z));
		const Fmul zi2 = weaken<Fmul>(sqr(zi));
		const Fmul zi3 = weaken<Fmul>(mul(zi2, zi));
		const auto ax = mul(P.X, zi2);
		const auto ay = mul(P.Y, zi3);
		u8 *out = A.out + (size_t)i * 64;
		u32 ow[8];
		to_words(ow, canonical(mul(ax, plain1)));
		store_be256(out, ow);
		to_words(ow, canonical(mul(ay, plain1)));
		store_be256(out + 32, ow);
		A.status[i] = 0;
	}
}

hipError_t ecamd_launch_smul_p256(const EcamdSmulArgs &a, hipStream_t s)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_smul_p256, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
	const uint32_t nthreads = (a.n + FIN_K - 1) / FIN_K;
	hipLaunchKernelGGL(k_p256_finalize, dim3((nthreads + 63) / 64), dim3(64), 0, s, a, nthreads);
	return hipGetLastError();
}
