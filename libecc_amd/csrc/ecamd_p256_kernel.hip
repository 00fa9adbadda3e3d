// libecc_amd/csrc/ecamd_p256_kernel.hip -- secp256r1 fast path of the batched prj_pt_mul.
//
// One scalar multiplication per lane (replaces prj_pt_import_from_aff_buf -> prj_pt_mul ->
// prj_pt_unique -> prj_pt_export_to_aff_buf, curves/prj_pt.c:511,1759,241,600 of the reference):
//   * field: radix-2^29 lazy Montgomery arithmetic (ecamd_u29.h), v_mad_u64_u32 only;
//   * group: Jacobian a = -3 doubling (4M + 4S) and addition (12M + 4S) (ecamd_p256.h);
//   * scalar: signed fixed window w = 4 over k' = k + 0x88...8 (digit = nibble - 8 in [-8, 7]),
//     left to right, 4 doublings + 1 MIXED addition (8M + 3S) per window, AFFINE table [1..8]P;
//   * four kernels per batch, all inversions shared by Montgomery's trick over 8 items per lane:
//       k_p256_table     import + on-curve check, Jacobian multiples 2P..8P
//       k_p256_affine    table -> affine (one inversion per 56 table entries)
//       k_p256_loop      the window loop, Jacobian result
//       k_p256_finalize  result -> affine X||Y big-endian (one inversion per 8 items)
//   * scratch, two areas (EcamdSmulArgs.tbl / .stg):
//       tab  the final AFFINE window table, item-major: 8 entries x 16 words per item, x || y as the canonical Montgomery
//            residues in eight saturated words each.  A look-up is one 64-byte record -- four 16-byte loads, never
//            straddling a 128-byte line (two entries per line) -- turned back into nine 29-bit digits on the fly;
//       stg  staging for the Jacobian multiples 2P..8P, their prefix products and the loop's result: per block of 64 items
//            (one wave), quad-major / lane-minor -- every 16-byte access of the table, affine and finalisation kernels is
//            coalesced across the wave (64 x 16 B contiguous), because there all lanes walk the entries in the same order.
//            Only the window loop looks up entries by a per-lane digit, and it reads tab.
// Exceptional pairs of the incomplete Jacobian addition (accumulator == +-table entry) cannot be
// produced by scalars below the group order and random points, but edge scalars (k >= q) can:
// such a lane is detected exactly (Z3 == 0 with both inputs finite), marked ECAMD_REDO and
// recomputed by the complete-formula kernel k_smul<8> in the same call (ecamd_host.cpp).
#include <hip/hip_runtime.h>
#include "ecamd_p256.h"
#include "ecamd_internal.h"

using namespace p256;

typedef uint8_t u8;

#define TBL_WORDS_PER_ENTRY 40   /* shared tables of the generator (window table, comb): affine x y as 2 x 9 digits, padded */
#define TBL_ENTRIES 8
#define TAB_ENT_WORDS 16         /* per-item table: x || y, 8 saturated words each (64 bytes) */
#define TAB_ITEM_WORDS (TBL_ENTRIES * TAB_ENT_WORDS)
#define STG_ENT_QUADS 10         /* staging record: X 0-8, Y 9-17, Z 18-26, (27), prefix product 28-36, (37-39) */
#define STG_ITEM_QUADS (7 * STG_ENT_QUADS)
#ifndef FIN_K
#define FIN_K 8                  /* items per lane in k_p256_finalize */
#endif
#ifndef AFF_K
#define AFF_K 8                  /* items per lane in k_p256_affine (x 7 table entries each); 2/4/8 measured equal AT 2^20 items */
#endif
// Items that share one Fermat inversion in the affine and finalisation kernels, chosen from the batch size (round 4): a lane's
// inversion is a serial chain of 255 squarings, so the kernel wants every SIMD busy more than it wants the 1/8 share -- the strong-
// scaling shards of a 2^20-item job (2^17 ... 2^19 items per GPU) and small batches keep the chip filled with 4 and 2 items per lane
// (the generic units do the same, ecamd_g29_kernel.hip).  kmax: the compile-time A/B value (8).
static inline int p256_items_per_inversion(uint32_t n, int kmax)
{
	const int k = n >= (1u << 19) ? 8 : (n >= (1u << 17) ? 4 : 2);
	return k < kmax ? k : kmax;
}

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

// ---- staging records (block-of-64 / quad-major / lane-minor) ----
static __device__ __forceinline__ uint4 *stg_quad(u32 *stg, u32 i, int slot, int q)
{
	return (uint4 *)stg + ((size_t)(i >> 6) * STG_ITEM_QUADS + (size_t)(slot * STG_ENT_QUADS + q)) * 64 + (i & 63u);
}
static __device__ __forceinline__ void jac_store(u32 *stg, u32 i, int slot, const Jac &P)
{
	u32 b[28];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		b[w] = P.X.l[w];
		b[9 + w] = P.Y.l[w];
		b[18 + w] = P.Z.l[w];
	}
	b[27] = 0;
#pragma unroll
	for (int q = 0; q < 7; q++) {
		*stg_quad(stg, i, slot, q) = make_uint4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
	}
}
static __device__ __forceinline__ Jac jac_load(u32 *stg, u32 i, int slot)
{
	u32 b[28];
#pragma unroll
	for (int q = 0; q < 7; q++) {
		const uint4 v = *stg_quad(stg, i, slot, q);
		b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
	}
	Jac P;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		P.X.l[w] = b[w];
		P.Y.l[w] = b[9 + w];
		P.Z.l[w] = b[18 + w];
	}
	return P;
}
static __device__ __forceinline__ FZ z_load(u32 *stg, u32 i, int slot)
{
	u32 b[12];  // quads 4..6 = words 16..27 cover Z = words 18..26
#pragma unroll
	for (int q = 0; q < 3; q++) {
		const uint4 v = *stg_quad(stg, i, slot, 4 + q);
		b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
	}
	FZ z;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		z.l[w] = b[2 + w];
	}
	return z;
}
static __device__ __forceinline__ void prefix_store(u32 *stg, u32 i, int slot, const Fmul &c)
{
	*stg_quad(stg, i, slot, 7) = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
	*stg_quad(stg, i, slot, 8) = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
	*stg_quad(stg, i, slot, 9) = make_uint4(c.l[8], 0u, 0u, 0u);
}
static __device__ __forceinline__ Fmul prefix_load(u32 *stg, u32 i, int slot)
{
	const uint4 a = *stg_quad(stg, i, slot, 7), b = *stg_quad(stg, i, slot, 8), c = *stg_quad(stg, i, slot, 9);
	Fmul r;
	r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
	r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
	r.l[8] = c.x;
	return r;
}

// ---- per-item affine table records: x || y, eight saturated words each ----
static __device__ __forceinline__ void tab_store(u32 *tab, u32 i, int e, const Fcanon &x, const Fcanon &y)
{
	u32 w[16];
	to_words(w, x);
	to_words(w + 8, y);
	uint4 *d = (uint4 *)(tab + (size_t)i * TAB_ITEM_WORDS + e * TAB_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < 4; q++) {
		d[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
	}
}
static __device__ __forceinline__ void tab_load(const u32 *tab, u32 e, Fcanon &x, Fcanon &y)
{
	u32 w[16];
	const uint4 *s = (const uint4 *)(tab + (size_t)e * TAB_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const uint4 v = s[q];
		w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
	}
	x = from_words(w);
	y = from_words(w + 8);
}
// Constant-address look-ups for secret digits (ecamd_ctx_set_secret_scalars): all eight entries are read, in the same order
// whatever the digit, and the wanted one is kept by masking (cf. tbl_load_masked in ecamd_kernels.hip)
static __device__ __forceinline__ void tab_load_masked(const u32 *tab, u32 idx, Fcanon &x, Fcanon &y)
{
	u32 w[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		w[k] = 0;
	}
#pragma unroll 1
	for (u32 e = 0; e < TBL_ENTRIES; e++) {
		const uint4 *s = (const uint4 *)(tab + (size_t)e * TAB_ENT_WORDS);
		const u32 m = 0u - (u32)(e == idx);
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint4 v = s[q];
			w[4 * q] |= v.x & m; w[4 * q + 1] |= v.y & m; w[4 * q + 2] |= v.z & m; w[4 * q + 3] |= v.w & m;
		}
	}
	x = from_words(w);
	y = from_words(w + 8);
}
// entries of the shared tables of the generator: 2 x 9 digits in 20 words
template <int NQ> static __device__ __forceinline__ void ld(u32 *b, const u32 *s)
{
	const uint4 *q = (const uint4 *)s;
#pragma unroll
	for (int i = 0; i < NQ; i++) {
		const uint4 v = q[i];
		b[4 * i] = v.x;
		b[4 * i + 1] = v.y;
		b[4 * i + 2] = v.z;
		b[4 * i + 3] = v.w;
	}
}
static __device__ __forceinline__ void lut_load_masked(const u32 *lut, u32 idx, Fcanon &x, Fcanon &y);
static __device__ __forceinline__ void lut_load(const u32 *lut, size_t e, Fcanon &x, Fcanon &y)
{
	u32 b[20];
	ld<5>(b, lut + e * TBL_WORDS_PER_ENTRY);
#pragma unroll
	for (int w = 0; w < 9; w++) {
		x.l[w] = b[w];
		y.l[w] = b[9 + w];
	}
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

static __device__ __forceinline__ void zero_out(u8 *out)
{
	const uint4 z = make_uint4(0, 0, 0, 0);
	if ((((uintptr_t)out) & 15) == 0) {
		((uint4 *)out)[0] = z; ((uint4 *)out)[1] = z; ((uint4 *)out)[2] = z; ((uint4 *)out)[3] = z;
	} else {
		for (int b = 0; b < 64; b++) out[b] = 0;
	}
}

// ------------------------------------------------------------------------------------------
// A. import + Jacobian multiples 2P..8P
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_p256_table(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
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
		zero_out(A.out + (size_t)i * 64);
		return;
	}
	const Fcanon xa = canonical(xm), ya = canonical(ym);
	tab_store(A.tbl, i, 0, xa, ya);  // entry 0: P itself, already affine
	Jac P1;
	P1.X = weaken<FX>(xm);
	P1.Y = weaken<FY>(ym);
	P1.Z = weaken<FZ>(constant<Fcanon>(K::ONE));
	bool hz;  // no exceptional pair can occur below: the group order is an odd prime > 8
	const FYaff y1 = weaken<FYaff>(ya);
	// entries 2P..8P go to staging slots 0..6
	Jac Pa = dbl(P1);
	jac_store(A.stg, i, 0, Pa);
	Jac Pb = madd(Pa, xa, y1, hz);
	jac_store(A.stg, i, 1, Pb);
	Pa = dbl(Pa);
	jac_store(A.stg, i, 2, Pa);
	Pb = madd(Pa, xa, y1, hz);
	jac_store(A.stg, i, 3, Pb);
	Pb = dbl(jac_load(A.stg, i, 1));
	jac_store(A.stg, i, 4, Pb);
	Pb = madd(Pb, xa, y1, hz);
	jac_store(A.stg, i, 5, Pb);
	Pa = dbl(Pa);
	jac_store(A.stg, i, 6, Pa);
	A.status[i] = ECAMD_STATUS_TAB;
}

// ------------------------------------------------------------------------------------------
// B. table -> affine: Montgomery's trick over the 7 Jacobian entries of FIN_K items per lane.
//    up:   cb_k = c (prefix BEFORE entry k) parked in the entry, c *= Z_k
//    down: 1/Z_k = t * cb_k, t *= Z_k        with t = 1 / (product of all Z)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_p256_affine(EcamdSmulArgs A, u32 nthreads, int items)
{
	// nthreads is a multiple of 64: the AFF_K items of a lane keep its lane index, so a wave's accesses to one staging
	// slot are 64 consecutive 16-byte words
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	Fmul c = weaken<Fmul>(constant<Fcanon>(K::ONE));
#pragma unroll 1
	for (int j = 0; j < items; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			break;
		}
		if (A.status[i] != ECAMD_STATUS_TAB) {
			continue;
		}
#pragma unroll 1
		for (int e = 0; e < 7; e++) {
			const FZ z = z_load(A.stg, i, e);
			prefix_store(A.stg, i, e, c);  // the prefix BEFORE this entry
			c = weaken<Fmul>(mul(c, z));
		}
	}
	Fmul tinv = inv(c);
#pragma unroll 1
	for (int j = items - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_TAB) {
			continue;
		}
#pragma unroll 1
		for (int e = 6; e >= 0; e--) {
			const Jac P = jac_load(A.stg, i, e);
			const Fmul cbf = prefix_load(A.stg, i, e);
			const Fmul zi = weaken<Fmul>(mul(tinv, cbf));
			tinv = weaken<Fmul>(mul(tinv, P.Z));
			const Fmul zi2 = weaken<Fmul>(sqr(zi));
			const Fmul zi3 = weaken<Fmul>(mul(zi2, zi));
			const Fcanon ax = canonical(mul(P.X, zi2));  // stays in the Montgomery domain
			const Fcanon ay = canonical(mul(P.Y, zi3));
			tab_store(A.tbl, i, e + 1, ax, ay);
		}
	}
}

// ------------------------------------------------------------------------------------------
// C. the window loop
// ------------------------------------------------------------------------------------------
#ifndef P256_WAVES
#define P256_WAVES 3
#endif
// scalar k (slen <= 4 KW bytes big-endian) -> k' = k + 0x88..8 over its 2 slen nibbles, left-aligned in kw[KW] (the top
// nibble of the scalar in bits 31..28 of kw[KW-1]); returns the carry out of the top nibble (the leading digit, 0 or +1)
template <int KW> static __device__ __forceinline__ u32 recode_window(u32 *kw, const u8 *sc, int slen)
{
	if (KW == 8 && slen == 32) {
		load_be256(sc, kw);
	} else {
#pragma unroll
		for (int w = 0; w < KW; w++) {
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
	uint64_t c = 0;
#pragma unroll
	for (int w = 0; w < KW; w++) {
		const int nb = slen - 4 * w;  // bytes of 0x88 only where the scalar has bytes
		const u32 add = (nb >= 4) ? 0x88888888u : (nb == 3 ? 0x00888888u : (nb == 2 ? 0x00008888u : (nb == 1 ? 0x00000088u : 0u)));
		c += (uint64_t)kw[w] + add;
		kw[w] = (u32)c;
		c >>= 32;
	}
	u32 carry_bit = (u32)c;  // only when slen == 4 KW
	if (slen < 4 * KW) {
		const int bit = 8 * slen;  // the carry out of the top nibble sits just above the scalar's bytes
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < KW; w++) {
			word = (w == (bit >> 5)) ? kw[w] : word;
		}
		carry_bit = (word >> (bit & 31)) & 1u;
		for (int s = slen; s < 4 * KW; s++) {  // left-align: top nibble of the scalar -> bits 31..28 of kw[KW-1]
#pragma unroll
			for (int w = KW - 1; w > 0; w--) {
				kw[w] = (kw[w] << 8) | (kw[w - 1] >> 24);
			}
			kw[0] <<= 8;
		}
	}
	return carry_bit;
}

// KW = 8: scalars of up to 32 bytes (everything below the group order's length); KW = 17: up to 68 bytes -- blinded scalars
// m + b #E of prj_pt_mul_blind (curves/prj_pt.c:1782-1822, about 2 |q| bits) stay on this kernel instead of the saturated one
static __device__ __forceinline__ void lut_load_masked(const u32 *lut, u32 idx, Fcanon &x, Fcanon &y)
{
	u32 b[20];
#pragma unroll
	for (int k = 0; k < 20; k++) {
		b[k] = 0;
	}
#pragma unroll 1
	for (u32 e = 0; e < TBL_ENTRIES; e++) {
		u32 t[20];
		ld<5>(t, lut + (size_t)e * TBL_WORDS_PER_ENTRY);
		const u32 m = 0u - (u32)(e == idx);
#pragma unroll
		for (int k = 0; k < 18; k++) {
			b[k] |= t[k] & m;
		}
	}
#pragma unroll
	for (int w = 0; w < 9; w++) {
		x.l[w] = b[w];
		y.l[w] = b[9 + w];
	}
}

// MASKED: secret scalars -- every look-up scans its eight-entry table (per-item records or the shared window table of the
// generator; the comb table is not used in this mode)
template <int KW, bool MASKED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P256_WAVES, P256_WAVES))) void k_p256_loop(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n || A.status[i] != ECAMD_STATUS_TAB) {
		return;
	}
	const u32 *tabi = A.tbl + (size_t)i * TAB_ITEM_WORDS;  // this item's affine table
	const bool shared = A.lut != nullptr;                 // fixed base: one table of the generator for every lane (wave-uniform)
	const int slen = (int)A.slen;
	u32 kw[KW];
	const u32 carry_bit = recode_window<KW>(kw, A.scalars + (size_t)i * A.sstride, slen);

	// ---- signed fixed window, left to right; the top digit is the carry: 0 or +1 ----
	Jac acc;
	{
		Fcanon px, py;
		if (shared) {
			lut_load(A.lut, 0, px, py);
		} else {
			tab_load(tabi, 0, px, py);
		}
		acc.X = weaken<FX>(px);
		acc.Y = weaken<FY>(py);
		acc.Z = weaken<FZ>(constant<Fcanon>(K::ONE));
	}
	bool inf = (carry_bit == 0);
	bool bad = false, hz;
	const int nwin = 2 * slen;
	const FZ onez = weaken<FZ>(constant<Fcanon>(K::ONE));
#pragma unroll 1
	for (int t = 0; t < nwin; t++) {
#pragma unroll 1
		for (int d = 0; d < 4; d++) {
			acc = dbl(acc);
		}
		const int dig = (int)(kw[KW - 1] >> 28) - 8;  // [-8, 7]
#pragma unroll
		for (int w = KW - 1; w > 0; w--) {
			kw[w] = (kw[w] << 4) | (kw[w - 1] >> 28);
		}
		kw[0] <<= 4;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		Fcanon tx, tyc;
		const u32 idx = mag ? mag - 1 : 0;
		if (MASKED) {
			if (shared) {
				lut_load_masked(A.lut, idx, tx, tyc);
			} else {
				tab_load_masked(tabi, idx, tx, tyc);
			}
		} else if (shared) {
			lut_load(A.lut, idx, tx, tyc);
		} else {
			tab_load(tabi, idx, tx, tyc);
		}
		const FYaff ty = sel(dig < 0, neg_aff(tyc), weaken<FYaff>(tyc));
		const Jac S = madd(acc, tx, ty, hz);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		// acc = keep ? acc : (use_t ? +-T : S)
		acc.X = sel(keep, acc.X, sel(use_t, weaken<FX>(tx), S.X));
		acc.Y = sel(keep, acc.Y, sel(use_t, weaken<FY>(carry(ty)), S.Y));
		acc.Z = sel(keep, acc.Z, sel(use_t, onez, S.Z));
		inf = inf & keep;
	}
	// ---- hand the Jacobian result to k_p256_finalize (staging slot 0 of the item) ----
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;  // exceptional pair met: the complete-formula kernel recomputes the item
		return;
	}
	if (inf) {
		A.status[i] = 2;
		zero_out(A.out + (size_t)i * 64);
		return;
	}
	jac_store(A.stg, i, 0, acc);
	A.status[i] = ECAMD_STATUS_JAC;
}

// ------------------------------------------------------------------------------------------
// C'. fixed base: 16-bit comb over a precomputed table of the generator (no doublings at all)
//   k = sum_j s_j 2^(16 j) + D_16 2^256 with s_j = D_j - 0x8000 in [-32768, 32767], D = digits of
//   K = k + 0x8000...8000;  table T[j][m-1] = [m 2^(16 j)]G for m = 1..32768 (affine, Montgomery
//   radix-2^29 digits, 80 bytes each, 16 x 32768 entries + [2^256]G = 42 MB per curve handle: HBM is
//   plentiful and the whole table sits in the 256 MB Infinity Cache);  [k]G = 17 mixed additions.
//   Partial sums are smaller in magnitude than the next term, so the only exceptional pair possible is
//   the last addition of a scalar >= q (e.g. k = q): flagged and recomputed by the complete kernel.
// ------------------------------------------------------------------------------------------
#define COMB_ENT_WORDS 20
#define COMB_PER_WIN 32768

__global__ __launch_bounds__(64) void k_p256_comb_build(const u8 *pts, u32 n, u32 *table)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	u32 xw[8], yw[8];
	load_be256(pts + (size_t)i * 64, xw);
	load_be256(pts + (size_t)i * 64 + 32, yw);
	const Fcanon r2 = constant<Fcanon>(K::R2);
	const Fcanon xa = canonical(mul(from_words(xw), r2)), ya = canonical(mul(from_words(yw), r2));
	u32 b[20];
#pragma unroll
	for (int w = 0; w < 9; w++) {
		b[w] = xa.l[w];
		b[9 + w] = ya.l[w];
	}
	b[18] = b[19] = 0;
	uint4 *d = (uint4 *)(table + (size_t)i * COMB_ENT_WORDS);
#pragma unroll
	for (int q = 0; q < 5; q++) {
		d[q] = make_uint4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
	}
}

// scalar (<= 32 bytes big-endian) -> K = k + 0x8000..8000 (8 words) and the carry D_16
static __device__ __forceinline__ u32 comb_recode(u32 *kw, const u8 *sc, int slen)
{
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
	uint64_t c = 0;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		c += (uint64_t)kw[w] + 0x80008000u;
		kw[w] = (u32)c;
		c >>= 32;
	}
	return (u32)c;
}

// acc += [k]G for the recoded scalar (kw, top); inf / bad as in the window loop
static __device__ __forceinline__ void comb_accumulate(Jac &acc, bool &inf, bool &bad, const u32 *kw, u32 top, const u32 *comb)
{
	const FZ onez = weaken<FZ>(constant<Fcanon>(K::ONE));
#pragma unroll 1
	for (int j = 0; j <= 16; j++) {
		const int dig = (j < 16) ? (int)((kw[j >> 1] >> (16 * (j & 1))) & 0xffffu) - 0x8000 : (int)top;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		u32 b[20];
		ld<5>(b, comb + ((size_t)j * COMB_PER_WIN + (mag ? mag - 1 : 0)) * COMB_ENT_WORDS);
		Fcanon tx, tyc;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			tx.l[w] = b[w];
			tyc.l[w] = b[9 + w];
		}
		const FYaff ty = sel(dig < 0, neg_aff(tyc), weaken<FYaff>(tyc));
		bool hz;
		const Jac S = madd(acc, tx, ty, hz);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		acc.X = sel(keep, acc.X, sel(use_t, weaken<FX>(tx), S.X));
		acc.Y = sel(keep, acc.Y, sel(use_t, weaken<FY>(carry(ty)), S.Y));
		acc.Z = sel(keep, acc.Z, sel(use_t, onez, S.Z));
		inf = inf & keep;
	}
}

__global__ __launch_bounds__(64) void k_p256_comb(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u32 kw[8];
	const u32 top = comb_recode(kw, A.scalars + (size_t)i * A.sstride, (int)A.slen);
	Jac acc;
	{
		u32 b[20];
		ld<5>(b, A.lut);  // any finite point: replaced by the first non-zero digit
#pragma unroll
		for (int w = 0; w < 9; w++) {
			acc.X.l[w] = b[w];
			acc.Y.l[w] = b[9 + w];
		}
		acc.Z = weaken<FZ>(constant<Fcanon>(K::ONE));
	}
	bool inf = true, bad = false;
	comb_accumulate(acc, inf, bad, kw, top, A.lut);
	if (bad) {
		A.status[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		A.status[i] = 2;
		zero_out(A.out + (size_t)i * 64);
		return;
	}
	jac_store(A.stg, i, 0, acc);
	A.status[i] = ECAMD_STATUS_JAC;
}

// ------------------------------------------------------------------------------------------
// C''. fixed base with SECRET scalars (ecamd_ctx_set_secret_scalars): a 4-bit comb whose look-ups are scans.
//   k = sum_j d_j 16^j + c 2^256 with d_j = D_j - 8 in [-8, 7], D = nibbles of K = k + 0x88..8, c its carry; table T[j][m-1] = [m 16^j]G,
//   m = 1..8, j = 0..63, and T[64][0] = [2^256]G: 65 x 8 entries of the shared-table format (83 KB per curve handle, L2-resident).
//   Window j's eight entries are read by EVERY lane, in order, whatever its digit -- the addresses depend on j alone and are the same
//   across the wave -- and the wanted one is kept by masking (lut_load_masked): the posture of the masked window loop, at 65 mixed
//   additions and no doubling instead of 256 doublings and 64 additions.  Round 4: a nonce multiplication [k]G was 15 of the 21 ms a
//   2^20-signature ec_sign_batch spent on the device (profiles/r4u_secret_half.md).
//   Exceptional pairs as for the 16-bit comb: only the last additions of a scalar >= q can meet one; flagged, recomputed by the
//   complete-formula kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_p256_comb4m(EcamdSmulArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	u32 kw[8];
	const int slen = (int)A.slen;
	const u8 *sc = A.scalars + (size_t)i * A.sstride;
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
	u32 top;
	{
		uint64_t c = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			c += (uint64_t)kw[w] + 0x88888888u;
			kw[w] = (u32)c;
			c >>= 32;
		}
		top = (u32)c;
	}
	Jac acc;
	{
		Fcanon px, py;
		lut_load(A.lut, 0, px, py);   // any finite point: replaced by the first non-zero digit
		acc.X = weaken<FX>(px);
		acc.Y = weaken<FY>(py);
		acc.Z = weaken<FZ>(constant<Fcanon>(K::ONE));
	}
	bool inf = true, bad = false;
	const FZ onez = weaken<FZ>(constant<Fcanon>(K::ONE));
#pragma unroll 1
	for (int j = 0; j <= 64; j++) {
		u32 word = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) {
			word = (w == (j >> 3)) ? kw[w] : word;
		}
		const int dig = (j < 64) ? (int)((word >> (4 * (j & 7))) & 15u) - 8 : (int)top;
		const u32 mag = (u32)(dig < 0 ? -dig : dig);
		Fcanon tx, tyc;
		lut_load_masked(A.lut + (size_t)j * TBL_ENTRIES * TBL_WORDS_PER_ENTRY, mag ? mag - 1 : 0, tx, tyc);
		const FYaff ty = sel(dig < 0, neg_aff(tyc), weaken<FYaff>(tyc));
		bool hz;
		const Jac S = madd(acc, tx, ty, hz);
		const bool use_t = inf & (mag != 0);
		const bool keep = (mag == 0);
		bad = bad | (!inf & !keep & hz);
		acc.X = sel(keep, acc.X, sel(use_t, weaken<FX>(tx), S.X));
		acc.Y = sel(keep, acc.Y, sel(use_t, weaken<FY>(carry(ty)), S.Y));
		acc.Z = sel(keep, acc.Z, sel(use_t, onez, S.Z));
		inf = inf & keep;
	}
	// secret scalars: nothing after the scan depends on the scalar -- the accumulator is stored whatever it holds, the
	// output is blanked whatever the outcome (k_p256_finalize overwrites it for a finite result, the redo pass for an exceptional one) and
	// the status is a select; k = 0, k >= q and exceptional sums take the same instructions as every other scalar
	jac_store(A.stg, i, 0, acc);
	zero_out(A.out + (size_t)i * 64);
	A.status[i] = bad ? (u8)ECAMD_STATUS_REDO : (inf ? (u8)2 : (u8)ECAMD_STATUS_JAC);
}

// ------------------------------------------------------------------------------------------
// D. finalisation: Jacobian -> affine for FIN_K items per lane with ONE field inversion
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_p256_finalize(EcamdSmulArgs A, u32 nthreads, int items)
{
	// nthreads is a multiple of 64 (coalesced staging accesses, as in k_p256_affine)
	const u32 t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nthreads) {
		return;
	}
	Fmul c = weaken<Fmul>(constant<Fcanon>(K::ONE));
#pragma unroll 1
	for (int j = 0; j < items; j++) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n) {
			break;
		}
		prefix_store(A.stg, i, 0, c);  // the prefix BEFORE this item
		if (A.status[i] == ECAMD_STATUS_JAC) {
			c = weaken<Fmul>(mul(c, z_load(A.stg, i, 0)));
		}
	}
	Fmul tinv = inv(c);
	Fcanon plain1;
#pragma unroll
	for (int w = 0; w < 9; w++) {
		plain1.l[w] = (w == 0) ? 1u : 0u;
	}
#pragma unroll 1
	for (int j = items - 1; j >= 0; j--) {
		const u32 i = t + (u32)j * nthreads;
		if (i >= A.n || A.status[i] != ECAMD_STATUS_JAC) {
			continue;
		}
		const Jac P = jac_load(A.stg, i, 0);
		const Fmul cbf = prefix_load(A.stg, i, 0);
		const Fmul zi = weaken<Fmul>(mul(tinv, cbf));
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

// ------------------------------------------------------------------------------------------
// ECDSA verification core: W' = [u1]G + [u2]Q by one interleaved window loop (Shamir / Straus):
// 4 doublings + 2 mixed additions per window instead of two separate scalar multiplications, and
// the final check x(W') mod q == r done projectively (r Z^2 == X, or (r + q) Z^2 == X when r + q < p),
// so no inversion at all.  The reference computes uG and vY separately and adds them
// (sig/ecdsa_common.c:786-810); only x mod q is observable.  Q's affine window table comes from
// k_p256_table / k_p256_affine, G's is a constant table built once per curve handle.
// result: 0 accept, 1 reject, ECAMD_STATUS_REDO when an exceptional pair was met (the host then
// re-verifies that item through the two-scalar-mult path).
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ u32 recode(u32 *kw, const u8 *sc)
{
	// 32-byte big-endian scalar -> k' = k + 0x88..8; returns the carry (top digit 0 / +1)
	load_be256(sc, kw);
	uint64_t c = 0;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		c += (uint64_t)kw[w] + 0x88888888u;
		kw[w] = (u32)c;
		c >>= 32;
	}
	return (u32)c;
}

struct P256VerifyArgs {
	const u8 *u1, *u2;      // n x 32 big-endian (k_ecdsa_prep)
	const u8 *sigs;         // n x 64, r || s
	const u8 *flags;        // n: r/s range check of k_ecdsa_prep
	const u8 *status;       // n: ECAMD_STATUS_TAB where Q's table is ready, 1 where the key is invalid
	const u32 *qtbl;        // per-item affine tables of Q
	const u32 *gtbl;        // affine table of G, 8 entries (window form) or the 16-bit comb table (COMB)
	u8 *result;
	u32 n;
	u32 qd[9];              // digits of the group order q (canonical, radix 2^29)
};

// COMB: [u1]G is added after the window loop of [u2]Q from the comb table (17 mixed additions)
// instead of one mixed addition per window (64)
template <bool COMB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P256_WAVES, P256_WAVES))) void k_p256_verify_loop(P256VerifyArgs A)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	if (A.flags[i] || A.status[i] != ECAMD_STATUS_TAB) {
		A.result[i] = 1;  // r or s out of range, or public key rejected at import
		return;
	}
	const u32 *tb = A.qtbl + (size_t)i * TAB_ITEM_WORDS;  // Q's affine table (64-byte records)
	u32 k1[8], k2[8];
	const u32 c1 = COMB ? comb_recode(k1, A.u1 + (size_t)i * 32, 32) : recode(k1, A.u1 + (size_t)i * 32);
	const u32 c2 = recode(k2, A.u2 + (size_t)i * 32);
	const FZ onez = weaken<FZ>(constant<Fcanon>(K::ONE));
	Jac acc;
	bool inf = true, bad = false, hz;
	// top digits (carries): acc = c1 G + c2 Q
	{
		Fcanon gx, gy, qx, qy;
		lut_load(A.gtbl, 0, gx, gy);
		tab_load(tb, 0, qx, qy);
		acc.X = weaken<FX>(gx);
		acc.Y = weaken<FY>(gy);
		acc.Z = onez;
		inf = COMB ? true : (c1 == 0);
		const Jac S = madd(acc, qx, weaken<FYaff>(qy), hz);
		const bool addq = (c2 != 0);
		bad = bad | (addq & !inf & hz);
		acc.X = sel(addq, sel(inf, weaken<FX>(qx), S.X), acc.X);
		acc.Y = sel(addq, sel(inf, weaken<FY>(qy), S.Y), acc.Y);
		acc.Z = sel(addq, sel(inf, onez, S.Z), acc.Z);
		inf = inf & !addq;
	}
#pragma unroll 1
	for (int t = 0; t < 64; t++) {
#pragma unroll 1
		for (int d = 0; d < 4; d++) {
			acc = dbl(acc);
		}
#pragma unroll 1
		for (int which = COMB ? 1 : 0; which < 2; which++) {
			u32 *kw = which ? k2 : k1;
			const int dig = (int)(kw[7] >> 28) - 8;
#pragma unroll
			for (int w = 7; w > 0; w--) {
				kw[w] = (kw[w] << 4) | (kw[w - 1] >> 28);
			}
			kw[0] <<= 4;
			const u32 mag = (u32)(dig < 0 ? -dig : dig);
			Fcanon tx, tyc;
			if (which) {
				tab_load(tb, mag ? mag - 1 : 0, tx, tyc);
			} else {
				lut_load(A.gtbl, mag ? mag - 1 : 0, tx, tyc);
			}
			const FYaff ty = sel(dig < 0, neg_aff(tyc), weaken<FYaff>(tyc));
			const Jac S = madd(acc, tx, ty, hz);
			const bool use_t = inf & (mag != 0);
			const bool keep = (mag == 0);
			bad = bad | (!inf & !keep & hz);
			acc.X = sel(keep, acc.X, sel(use_t, weaken<FX>(tx), S.X));
			acc.Y = sel(keep, acc.Y, sel(use_t, weaken<FY>(carry(ty)), S.Y));
			acc.Z = sel(keep, acc.Z, sel(use_t, onez, S.Z));
			inf = inf & keep;
		}
	}
	if (COMB) {
		comb_accumulate(acc, inf, bad, k1, c1, A.gtbl);
	}
	if (bad) {
		A.result[i] = ECAMD_STATUS_REDO;
		return;
	}
	if (inf) {
		A.result[i] = 1;  // W' = infinity (sig/ecdsa_common.c:798-800)
		return;
	}
	// x(W') mod q == r  <=>  X == r Z^2 or X == (r + q) Z^2 with r + q < p
	u32 rw[8];
	load_be256(A.sigs + (size_t)i * 64, rw);
	const Fcanon r2c = constant<Fcanon>(K::R2), onec = constant<Fcanon>(K::ONE);
	const auto z2 = sqr(acc.Z);
	const Fcanon rd = from_words(rw);
	const auto t1 = mul(mul(rd, r2c), z2);                       // r Z^2 (Montgomery domain)
	bool acc_ok = is_zero_mulout(mul(carry(sub<1, 1>(t1, acc.X)), onec));
	{
		// r + q as digits, and whether it is below p (p - q < 2^128, so at most one extra candidate)
		u32 sdg[9];
		u32 cy = 0;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			const u32 x = rd.l[w] + A.qd[w] + cy;
			cy = (w < 8) ? (x >> 29) : 0u;
			sdg[w] = (w < 8) ? (x & MASK) : x;
		}
		u32 bw = 0;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			bw = (sdg[w] - P256::P[w] - bw) >> 31;
		}
		Fcanon rq;
#pragma unroll
		for (int w = 0; w < 9; w++) {
			rq.l[w] = bw ? sdg[w] : 0u;
		}
		const auto t2 = mul(mul(rq, r2c), z2);
		acc_ok = acc_ok | ((bw != 0) & is_zero_mulout(mul(carry(sub<1, 1>(t2, acc.X)), onec)));
	}
	A.result[i] = acc_ok ? 0 : 1;
}

hipError_t ecamd_launch_verify_p256(const EcamdSmulArgs &pubkeys, const uint8_t *u1, const uint8_t *u2, const uint8_t *sigs,
				    const uint8_t *flags, const uint32_t *gtbl, int gtbl_is_comb, const uint32_t *qdigits,
				    uint8_t *result, hipStream_t s, hipEvent_t *dom, hipEvent_t scalars_ready)
{
	if (pubkeys.n == 0) {
		return scalars_ready ? hipStreamWaitEvent(s, scalars_ready, 0) : hipSuccess;
	}
	const dim3 grid((pubkeys.n + 63) / 64), block(64);
	const uint32_t ak = (uint32_t)p256_items_per_inversion(pubkeys.n, AFF_K);
	const uint32_t athreads = (((pubkeys.n + ak - 1) / ak) + 63u) & ~63u;
	hipLaunchKernelGGL(k_p256_table, grid, block, 0, s, pubkeys);
	hipLaunchKernelGGL(k_p256_affine, dim3((athreads + 63) / 64), block, 0, s, pubkeys, athreads, (int)ak);
	P256VerifyArgs V;
	V.u1 = u1;
	V.u2 = u2;
	V.sigs = sigs;
	V.flags = flags;
	V.status = pubkeys.status;
	V.qtbl = pubkeys.tbl;
	V.gtbl = gtbl;
	V.result = result;
	V.n = pubkeys.n;
	for (int w = 0; w < 9; w++) {
		V.qd[w] = qdigits[w];
	}
	if (scalars_ready) {
		// u1, u2 and the flags come from k_ecdsa_prep on the context's side stream (it ran beside the two kernels above)
		const hipError_t e = hipStreamWaitEvent(s, scalars_ready, 0);
		if (e != hipSuccess) {
			return e;
		}
	}
	if (dom) {
		(void)hipEventRecord(dom[0], s);
	}
	if (gtbl_is_comb) {
		hipLaunchKernelGGL(k_p256_verify_loop<true>, grid, block, 0, s, V);
	} else {
		hipLaunchKernelGGL(k_p256_verify_loop<false>, grid, block, 0, s, V);
	}
	if (dom) {
		(void)hipEventRecord(dom[1], s);
	}
	return hipGetLastError();
}

hipError_t ecamd_launch_smul_p256(const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev)
{
	if (a.n == 0) {
		return hipSuccess;
	}
	const dim3 grid((a.n + 63) / 64), block(64);
	const uint32_t fk = (uint32_t)p256_items_per_inversion(a.n, FIN_K);
	const uint32_t nthreads = (((a.n + fk - 1) / fk) + 63u) & ~63u;
	const dim3 fgrid((nthreads + 63) / 64);
#define P256_MARK(i) do { if (ev) (void)hipEventRecord(ev[i], s); } while (0)
	P256_MARK(0);
	if (a.lut) {
		// fixed base: the generator's affine table is a constant of the curve handle; every item is valid
		(void)hipMemsetAsync(a.status, ECAMD_STATUS_TAB, a.n, s);
		P256_MARK(1);
	} else {
		hipLaunchKernelGGL(k_p256_table, grid, block, 0, s, a);
		P256_MARK(1);
		const uint32_t ak = (uint32_t)p256_items_per_inversion(a.n, AFF_K);
		const uint32_t athreads = (((a.n + ak - 1) / ak) + 63u) & ~63u;
		hipLaunchKernelGGL(k_p256_affine, dim3((athreads + 63) / 64), block, 0, s, a, athreads, (int)ak);
	}
	P256_MARK(2);
	if (a.lut && a.lut_kind == 1) {
		hipLaunchKernelGGL(k_p256_comb, grid, block, 0, s, a);
	} else if (a.lut && a.lut_kind == 3) {
		hipLaunchKernelGGL(k_p256_comb4m, grid, block, 0, s, a);
	} else {
		if (a.masked) {
			if (a.slen <= 32) {
				hipLaunchKernelGGL((k_p256_loop<8, true>), grid, block, 0, s, a);
			} else {
				hipLaunchKernelGGL((k_p256_loop<17, true>), grid, block, 0, s, a);
			}
		} else if (a.slen <= 32) {
			hipLaunchKernelGGL((k_p256_loop<8, false>), grid, block, 0, s, a);
		} else {
			hipLaunchKernelGGL((k_p256_loop<17, false>), grid, block, 0, s, a);
		}
	}
	P256_MARK(3);
	hipLaunchKernelGGL(k_p256_finalize, fgrid, block, 0, s, a, nthreads, (int)fk);
	P256_MARK(4);
	return hipGetLastError();
}

hipError_t ecamd_launch_comb_build_p256(const uint8_t *points, uint32_t n, uint32_t *table, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_p256_comb_build, dim3((n + 63) / 64), dim3(64), 0, s, points, n, table);
	return hipGetLastError();
}
