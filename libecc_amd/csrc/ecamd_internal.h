// libecc_amd/csrc/ecamd_internal.h -- launch interface between the host side (ecamd_host.cpp)
// and the kernels (ecamd_kernels.hip).  Not part of the public C ABI (include/libecc_amd.h).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ECAMD_WINDOW 4
#define ECAMD_TBL_ENTRIES (1 << ECAMD_WINDOW)
#define ECAMD_STATUS_TAB 0xFC   /* internal: fast path built the item's window table, loop pending */
#define ECAMD_STATUS_JAC 0xFD   /* internal: fast path stored a finite Jacobian result, finalisation pending */
#define ECAMD_STATUS_REDO 0xFE  /* internal: fast path met an exceptional pair, redo with complete formulas */
#define ECAMD_MAX_SLOTS_HOST 16 /* == ECAMD_MAX_SLOTS in ecamd_field.h */

struct EcamdSmulArgs {
	const uint8_t *scalars;  // n x slen, big-endian
	const uint8_t *points;   // n x 2*clen affine X||Y big-endian (pstride = 0: one shared point)
	uint8_t *out;            // n x 2*clen affine X||Y big-endian
	uint8_t *status;         // n : 0 ok, 1 error, 2 infinity
	uint32_t *tbl;           // scratch: ECAMD_TBL_ENTRIES x 3 x NW words x stride (secp256r1 fast path: the affine window tables)
	uint32_t *stg;           // secp256r1 fast path: staging area (Jacobian multiples, prefix products, loop results)
	uint32_t n, slen, clen, pstride, stride;
	uint32_t sstride;        // bytes between consecutive scalars (slen, or 0: one shared scalar)
	int slot;
	int only_redo;           // generic kernel: process only items whose status is ECAMD_STATUS_REDO
	const uint32_t *lut;     // secp256r1 fixed base: shared affine table of G (NULL: per-item tables)
	uint32_t lut_kind;       // 0: window table [1..8]G (8 x 40 words), 1: 16-bit comb table of the generator,
	                         // 2 (generic radix-2^29 units): per-item window tables of `points` AND the comb table of the generator
	                         //   for a second scalar: [scalars]P + [scalars2]G in one window loop (ECDSA verification)
	int masked;              // generic kernel: secret scalars -- constant-address (full-scan, masked) table look-ups
	const uint8_t *scalars2; // lut_kind 2: n x s2len big-endian multipliers of the generator
	uint32_t s2len;
};
#define ECAMD_COMB_ENTRIES (16u * 32768u + 1u)

struct EcamdFpArgs {
	const uint32_t *a, *b;   // n x NW little-endian 32-bit words
	uint32_t *out;
	uint32_t n, wstride;     // wstride: 32-bit words per element in memory (>= NW)
	int op;                  // 0 mul_monty (reference radix 2^(64 nlimbs)), 1 add, 2 sub, 3 mul plain, 4 inv plain
	int slot;
};

struct EcamdPtArgs {
	const uint8_t *p1, *p2;  // affine inputs, n x 2*clen
	uint8_t *out, *status;
	uint32_t n, clen;
	int dbl, slot;
};

// group law / on-curve test in either wire format (k_ptf), and _prj_pt_unprotected_mult statement for statement (k_unprot)
struct EcamdPtfArgs {
	const uint8_t *p1, *p2;  // n x (in_fmt ? 3 : 2) * clen
	uint8_t *out, *status;   // n x (out_fmt ? 3 : 2) * clen (untouched by op 2), n
	uint32_t n, clen;
	int op;                  // 0 prj_pt_add, 1 prj_pt_dbl, 2 prj_pt_is_on_curve (status 0 on the curve / 1 not)
	int in_fmt, out_fmt, slot;
};
struct EcamdUnprotArgs {
	const uint8_t *points;   // n x (in_fmt ? 3 : 2) * clen
	const uint8_t *scalars;  // big-endian, slen octets each; sstride = 0: one scalar for every item
	uint8_t *out, *status;
	uint32_t n, clen, slen, sstride;
	int in_fmt, out_fmt, slot;
};
hipError_t ecamd_launch_ptf(int nw, const EcamdPtfArgs &a, hipStream_t s);
hipError_t ecamd_launch_unprot(int nw, const EcamdUnprotArgs &a, hipStream_t s);

// ---- ECDSA verify (sig/ecdsa_common.c:619-840 of the reference), three stages around two scalar mults ----
struct EcamdEcdsaPrepArgs {
	const uint8_t *sigs;     // n x 2*qlen, r || s big-endian
	const uint8_t *digests;  // n x hlen, H(m)
	uint8_t *u1, *u2;        // n x qlen big-endian: e/s mod q, r/s mod q
	uint8_t *flags;          // n: 0 ok, 1 reject (r or s not in [1, q-1])
	uint32_t n, qlen, hlen, qbits;
	int qslot;               // constant slot holding the Montgomery context of the generator order q
	const uint8_t *only;     // NULL, or n status bytes: handle only the items marked ECAMD_STATUS_REDO (a lane none of whose items is marked exits at once)
};
struct EcamdEcdsaFinArgs {
	const uint8_t *A, *stA;  // [u1]G affine + status
	const uint8_t *B, *stB;  // [u2]Q affine + status
	const uint8_t *sigs, *flags;
	uint8_t *result;         // n: 0 accept, 1 reject
	uint32_t n, clen, qlen, jmax;  // jmax = floor((p-1)/q): candidates x = r + j q
	uint32_t q[17];          // generator order, little-endian words, zero padded to NW
	int slot;
	const uint8_t *only;     // NULL, or n status bytes (may be `result` itself): only the items marked ECAMD_STATUS_REDO
};
struct EcamdEcdsaSignArgs {
	const uint8_t *privs, *nonces, *digests;  // n x qlen, n x qlen, n x hlen
	const uint8_t *kG, *stkG;                 // [k]G affine + status
	uint8_t *sigs, *status;                   // n x 2*qlen (r || s), n
	uint32_t n, clen, qlen, hlen, qbits, jmax;
	int qslot;
};
hipError_t ecamd_launch_ecdsa_sign(int nw, const EcamdEcdsaSignArgs &a, hipStream_t s);
hipError_t ecamd_launch_ecdsa_prep(int nw, const EcamdEcdsaPrepArgs &a, hipStream_t s);
hipError_t ecamd_launch_ecdsa_fin(int nw, const EcamdEcdsaFinArgs &a, hipStream_t s);

// nw: 32-bit words per field element; must be one of ecamd_supported_nw()
int ecamd_nw_supported(int nw);
hipError_t ecamd_upload_curve(int nw, int slot, const void *curvek, size_t bytes);
hipError_t ecamd_launch_smul(int nw, const EcamdSmulArgs &a, hipStream_t s);
// ev (optional, 6 events): recorded before the first kernel and after each kernel of the pipeline
#define ECAMD_NTIMED 5
hipError_t ecamd_launch_smul_p256(const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev);
// secp256r1 ECDSA verification core: pubkeys.{points,status,tbl,out,n,pstride} describe the public keys
// (table kernels), u1/u2/sigs/flags come from k_ecdsa_prep, gtbl = affine table of G (8 x 40 words),
// qdigits = group order in radix 2^29; result 0 accept / 1 reject / ECAMD_STATUS_REDO
hipError_t ecamd_launch_verify_p256(const EcamdSmulArgs &pubkeys, const uint8_t *u1, const uint8_t *u2, const uint8_t *sigs,
				    const uint8_t *flags, const uint32_t *gtbl, int gtbl_is_comb, const uint32_t *qdigits,
				    uint8_t *result, hipStream_t s, hipEvent_t *dom = nullptr, hipEvent_t scalars_ready = nullptr);
// affine big-endian points -> comb table entries (Montgomery radix-2^29 digits, 20 words each)
hipError_t ecamd_launch_comb_build_p256(const uint8_t *points, uint32_t n, uint32_t *table, hipStream_t s);
// ---- X25519 / X448 (ecdh/x25519_448.c:146-302 of the reference) around the scalar multiplication ----
struct EcamdXdhPrepArgs {
	const uint8_t *k, *u;    // n x len little-endian scalars and u coordinates (RFC 7748 wire format)
	uint8_t *scalars;        // out: n x len big-endian clamped scalars
	uint8_t *points;         // out: n x 2*len affine Weierstrass X || Y big-endian
	uint8_t *flags;          // out: n, 0 ok / 1 reject (u >= p, u on the twist, [cofactor]Q = infinity)
	uint32_t n, len, ebits, mode;  // mode 0: p = 5 mod 8 (candidate w^((p+3)/8)), 1: p = 3 mod 4 (w^((p+1)/4))
	uint32_t cof_dbl;        // log2(cofactor)
	uint32_t e[17];          // the exponent, little-endian words
	uint32_t A[17], A3[17], sm1[17];  // A, A/3, sqrt(-1) in Montgomery form (radix 2^(32 NW))
	uint32_t g_A[16], g_A3[16], g_sm1[16];  // the same as plain radix-2^29 digits (9 for the 2^255 - 19 unit, 16 for the Goldilocks unit)
	int slot;
};
struct EcamdXdhFinArgs {
	const uint8_t *pts;      // [k]Q affine Weierstrass, big-endian
	const uint8_t *stk;      // status of [k]Q
	const uint8_t *flags;
	uint8_t *out;            // n x len little-endian u coordinates
	uint8_t *status;         // n: 0 ok / 1 the reference returns -1
	uint32_t n, len;
	uint32_t A3[17];         // A/3 mod p, plain
	int slot;
};
hipError_t ecamd_launch_xdh_prep(int nw, const EcamdXdhPrepArgs &a, hipStream_t s);
hipError_t ecamd_launch_xdh_fin(int nw, const EcamdXdhFinArgs &a, hipStream_t s);

// ---- Ed25519 verification (sig/eddsa.c of the reference) through the Weierstrass model ----
struct EcamdEdDecodeArgs {
	const uint8_t *encA;     // n compressed public keys: len bytes little-endian y, sign of x in the top bit
	const uint8_t *encR;     // n compressed R (first half of each signature)
	uint32_t strideA, strideR;  // bytes between consecutive encodings
	uint8_t *pointsA, *pointsR; // out: n x 2*len affine Weierstrass X || Y big-endian
	uint8_t *flagsA, *flagsR;   // out: n, 0 ok / 1 the reference's decode / map returns -1 (A: or [cofactor]A = infinity)
	uint32_t n, len, cof_dbl;   // cof_dbl = log2(cofactor)
	uint32_t a[17], d[17], sm1[17], alpha[17], A3[17];  // Edwards a, d; sqrt(-1); alpha_edwards; A/3 (Montgomery form)
	uint32_t g_d[9], g_sm1[9], g_alpha[9], g_A3[9];     // the same as plain radix-2^29 digits (2^255 - 19 unit)
	uint32_t *edA;              // 2^255 - 19 unit only, may be NULL: n x 20 words, A on the Edwards curve (x, y digits)
	uint32_t *edR;              // k_ed_decode_ed_c25519: n x 20 words, R on the Edwards curve (its map to the Weierstrass model
	                            // waits for the shared inversion of k_ed_hA_fin)
	int slot;
};
// decoding for the Edwards [h]A path: A and R stay on the Edwards curve (edA, edR), flags as above, no inversion here
hipError_t ecamd_launch_ed_decode_ed_c25519(const EcamdEdDecodeArgs &a, int gslot, hipStream_t s);
// [h]A on the Edwards curve (extended coordinates) and its map to the Weierstrass model (2^255 - 19 unit)
struct EcamdEdSmulArgs {
	const uint32_t *edA;     // n x 20 words (k_ed_decode_c25519)
	const uint8_t *scalars;  // n x 32 big-endian
	const uint8_t *flags;    // n, non-zero: key rejected
	uint32_t *tbl;           // scratch: n x 320 words
	uint32_t *rec;           // scratch: n x 28 words
	uint8_t *out, *status;   // n x 64 affine Weierstrass big-endian, n (0 ok / 1 rejected key / 2 infinity)
	uint32_t n;
	uint32_t g_2d[9], g_alpha[9], g_A3[9];
	// k_ed_hA_fin also maps R (decoded by k_ed_decode_ed_c25519) to the Weierstrass model with the same shared inversion:
	const uint32_t *edR;     // n x 20 words, or NULL (R was mapped by the decode kernel)
	uint8_t *flagsR;         // n: 0 / 1 / 2 as the decode kernels write them
	uint8_t *outR;           // n x 64 affine Weierstrass big-endian
};
#define ECAMD_EDT_ITEM_WORDS 320
#define ECAMD_EDR_REC_WORDS 28
// Round 4: the tail of an Ed25519 verification on the Edwards curve (k_ed_tail_c25519): [S]B from an Edwards comb table of the
// base point, W1 = [S]B - R, W2 = W1 - [h]A, [8]W2 == neutral, with prj_pt_add's two failures restated as comparisons.
struct EcamdEdTailConsts {
	uint32_t g_2d[9], g_alpha[9], g_A3[9];   // plain radix-2^29 digits
};
#define ECAMD_EDC_ENT_WORDS 32
#define ECAMD_EDC_ENTRIES (16u * 32768u + 1u)
// the half-length form (k_ed_lat -> k_ed_smul2_c25519 -> k_ed_tail2_c25519)
struct EcamdEdSmul2Args {
	const uint32_t *edA, *edR;   // n x 20 words (k_ed_decode_ed_c25519)
	const uint8_t *flagsA, *flagsR, *flagsS;
	const uint32_t *uv;          // n x 12 words: v (8), |u| (4)
	const uint8_t *meta;         // n: bit 0 u < 0, bit 1 full-length scalars
	uint32_t *tbl;               // scratch: n x 2 x ECAMD_EDT_ITEM_WORDS
	uint32_t *rec;               // n x ECAMD_EDR_REC_WORDS: L = -[v]A - [u]R (X, Y, Z)
	uint32_t n;
	uint32_t g_2d[9];
};
struct EcamdEdTailArgs {
	const uint32_t *rec;     // n x ECAMD_EDR_REC_WORDS: [h]A (X, Y, Z) from k_ed_smul_c25519<1>; tail2: L from k_ed_smul2_c25519<1>
	const uint32_t *edR;     // n x 20 words: R on the Edwards curve (k_ed_decode_ed_c25519)
	const uint8_t *flagsA, *flagsR, *flagsS;
	const uint8_t *S_be;     // n x 32 big-endian
	const uint8_t *sp_be;    // tail2: n x 32 big-endian s' = u S mod q
	const uint8_t *meta;     // tail2: k_ed_lat's meta bytes
	const uint32_t *comb;    // ECAMD_EDC_ENTRIES x ECAMD_EDC_ENT_WORDS: [m 2^(16 j)]B as (y - x, y + x, 2d x y)
	uint8_t *result;         // n: 0 accept / 1 reject
	uint32_t n, cof_dbl;
	EcamdEdTailConsts C;
};
hipError_t ecamd_launch_edcomb_build_c25519(const uint8_t *pts, uint32_t n, uint32_t *table, const EcamdEdTailConsts &c, int gslot, hipStream_t s);
// eddsa_encode_point of n affine Weierstrass points (the encode step of EcamdEdSignArgs: Rw, stR -> enc, status) on the 2^255 - 19 unit
hipError_t ecamd_launch_ed_enc_c25519(const uint8_t *Rw, const uint8_t *stR, uint8_t *enc, uint8_t *status, uint32_t n, const EcamdEdTailConsts &c, int gslot,
				      hipStream_t s);
hipError_t ecamd_launch_ed_tail_c25519(const EcamdEdTailArgs &a, int gslot, hipStream_t s);
hipError_t ecamd_launch_ed_smul2_c25519(const EcamdEdSmul2Args &a, int gslot, hipStream_t s, hipEvent_t *dom = nullptr);   // dom: events around the 33-window loop
hipError_t ecamd_launch_ed_tail2_c25519(const EcamdEdTailArgs &a, int gslot, hipStream_t s);
hipError_t ecamd_launch_ed_smul_c25519(const EcamdEdSmulArgs &a, int gslot, hipStream_t s, hipEvent_t *dom = nullptr);   // dom: two events around the window loop
// ---- Ed25519 whole-batch verification as ONE multi-scalar multiplication on the Edwards curve (2^255 - 19 unit) ----
// T = [q - sum z_i S_i]B + sum_i ([z_i h_i mod q]A_i + [z_i]R_i), accepted when [8]T is the neutral element
// (_eddsa_verify_batch_no_memory, sig/eddsa.c:2278-2545).  Straus evaluation: lane l owns the items j * L + l (j < K) and
// shares the 256 doublings between their 2K + 1 points.
struct EcamdEdMsmArgs {
	const uint8_t *encA, *encR;  // n compressed keys / signatures (R in the first half)
	uint32_t strideA, strideR;
	uint32_t *tbl;               // n x 2 x ECAMD_EDT_ITEM_WORDS: window tables [1..8]A_i, [1..8]R_i
	uint8_t *flags;              // n: non-zero = the reference rejects the item before the equation (decode, small-order key)
	const uint32_t *cA;          // 8 x n words, word-major: z_i h_i mod q + 0x88..8 (signed-window recoding)
	const uint32_t *zR;          // 5 x n words, word-major: z_i + 0x8 88..8 (132 bits)
	const uint32_t *sB;          // 8 x L words, word-major: (q - sum of the lane's z_i S_i) + 0x88..8
	const uint32_t *tblB;        // ECAMD_EDT_ITEM_WORDS: [1..8]B
	uint32_t *rec;               // L x ECAMD_EDM_REC_WORDS: the lanes' sums (X, Y, Z, T)
	uint32_t n, K, L;
	uint32_t cof_dbl;
	uint32_t g_d[9], g_sm1[9], g_2d[9], g_Bx[9], g_By[9];
	uint32_t first, count;       // k_edbkt_points: the items [first, first + count) of the n (count 0: all) -- the streamed form files a batch
	                             // chunk by chunk while the next chunk is still on its way (eddsa_bkt_chunk)
};
#define ECAMD_EDM_REC_WORDS 36
hipError_t ecamd_launch_edmsm_btable(const EcamdEdMsmArgs &a, uint32_t *tblB, int gslot, hipStream_t s);
hipError_t ecamd_launch_edmsm_prep(const EcamdEdMsmArgs &a, int gslot, hipStream_t s);
hipError_t ecamd_launch_edmsm_loop(const EcamdEdMsmArgs &a, int gslot, hipStream_t s);
// sums the L lane records (tree, fan-in 16, ping-pong between rec and tmp), multiplies by the cofactor and writes
// verdict[0] = 0 (neutral element and *flagword == 0) / 1; sum_out (may be NULL): the sum before the cofactor, 36 words
hipError_t ecamd_launch_edmsm_reduce(const EcamdEdMsmArgs &a, uint32_t *tmp, const uint32_t *flagword, uint8_t *verdict,
				     uint32_t *sum_out, int gslot, hipStream_t s);
// scalars of the combination (mod q, saturated unit): z_i from ChaCha20(seed; counter = item), c_i, z_i S_i
struct EcamdEdMsmScalArgs {
	const uint8_t *sigs;         // n x 64: R || S
	const uint8_t *hram;         // n x 64
	uint32_t *cA, *zR;           // as above
	uint32_t *zs;                // n x 8 words: z_i S_i mod q
	uint8_t *flagsS;             // n: S >= q
	uint8_t *z_dump;             // may be NULL: n x 16 little-endian z_i (tests)
	uint32_t *rawC, *rawZ;       // may be NULL (bucket evaluation, round 6): n x 8 / n x 4 little-endian words, z_i h_i mod q and z_i as they are
	uint32_t seed[8], nonce[3];
	uint32_t n;
	int qslot;
	uint32_t first, count;       // the items [first, first + count) of the n (count 0: all); z_i is keyed by the item's index in the batch either way
};
struct EcamdEdMsmLaneArgs {
	const uint32_t *zs;
	const uint8_t *flags, *flagsS;
	uint32_t *sB;
	uint32_t *rawB;              // may be NULL (bucket evaluation): L x 8 little-endian words, q - sum of the lane's z_i S_i as it is
	uint32_t *flagword;          // |= 1 when any item of the lane is flagged
	uint32_t n, K, L;
	int qslot;
};
// The Ed25519 batch equation by buckets (round 6; k_edbkt_* in ecamd_g29_kernel.hip, k_edbkt_file in ecamd_kernels.hip): 16-bit windows over
// z_i h_i (16 windows), the base point's LB scalars (16 windows: the term [q - sum z_i S_i]B as LB copies of B, one per 64 items) and z_i (8
// windows).  Point index: A_i = i, the copies of B = n + l, R_i = n + LB + i.  The Edwards addition is complete: no exceptional cases.
#define ECAMD_EDB_PT_WORDS 28   /* y - x, y + x, 2d x y: 3 x 9 limbs, padded */
#define ECAMD_EDB_PT_STRIDE 32  /* words between records: one 128-byte line each -- a gather of k_edbkt_accum touches ONE line (112-byte records
                                 * at a 112-byte stride straddled two lines seven times out of eight: 8.1 GB of HBM traffic per 2^20 items for
                                 * 3.8 GB of records, 3.9 TB/s at 0.41 VALU busy, profiles/r6_bucket_kernel_bound.md) */
struct EcamdEdBktArgs {
	const uint32_t *rawC, *rawB, *rawZ;
	uint32_t *pts;               // (2n + LB) x ECAMD_EDB_PT_STRIDE
	uint32_t *count, *perm;      // 16 << 16 counters (the buckets' sizes) / the ranking of k_bkt_rank
	uint32_t *order;             // (16 << 16) x cap point indices
	uint32_t *bsum;              // 16 << 16 records of ECAMD_EDM_REC_WORDS
	uint32_t *red;               // scratch of the reduction, red_words words
	uint64_t red_words;
	uint32_t *flagword;          // |= 16: a bucket overflowed its cap slots
	uint32_t n, LB, cap, cap_top;   // cap_top: slots of the buckets of window 15 (z h mod q < 2^253: 4 097 digit values only)
	uint32_t part, first, count_items;   // ecamd_launch_edbkt_file: part 0 everything (counters cleared, ranking at the end); 1: the keys and commitments
	                                // of the items [first, first + count_items) alone, counters as they are; 2: the LB copies of B, then the ranking
};
// Ed25519's top window (15): z h mod q < 2^252 + 2^125, so its digits take 4 097 values and all n keys crowd into as many buckets -- 256 points each
// at 2^20 items, eight times the other windows' buckets, and ONE lane's chain of 256 dependent additions outlasted the whole balanced launch
// (profiles/r6_bucket_kernel_bound.md).  The window's other 61 439 lanes were idle: lane (15 << 16 | s * 8192 + d), s = 0 .. 7, now sums the
// points s, s + 8, s + 16, ... of bucket d, and k_edbkt_combine adds the eight partial sums up (and blanks the seven borrowed records).
#define ECAMD_EDB_SPLIT 8u
#define ECAMD_EDB_SPLIT_LOG2 3u
#define ECAMD_EDB_SPLIT_DIGITS 8192u
// entries a lane of the bucket reduction folds per level (k_bkt_reduce_g / k_edbkt_reduce: 2 (fold - 1) dependent additions per level, log_fold(2^c)
// levels): 8 by default -- measured per 2^20 items (tools/gpu_r6zb.sh): BIP0340 / secp256k1 6.84 (16) / 6.66 (8) / 6.71 (4) / 6.87 ms (2), Ed25519
// 6.89 / 6.64 / 6.67 / 6.84, Ed448 25.9 / 25.8 / 26.8 / 28.4; $ECAMD_BKT_FOLD=2|4|8|16 (measurements, tests; read at every call, by the host's
// scratch sizing and by the launchers alike)
static inline uint32_t ecamd_bkt_fold()
{
	const char *e = getenv("ECAMD_BKT_FOLD");
	const int v = e ? atoi(e) : 8;
	return (v == 2 || v == 4 || v == 8 || v == 16) ? (uint32_t)v : 8u;
}
// first slot of bucket b (= window << 16 | digit) and its capacity under the two-capacity layout
static inline __host__ __device__ size_t ecamd_bkt_slot(uint32_t b, uint32_t cap, uint32_t cap_top, uint32_t top_win, uint32_t *cap_out)
{
	const uint32_t tb = top_win << 16;
	if (b < tb) {
		*cap_out = cap;
		return (size_t)b * cap;
	}
	*cap_out = cap_top;
	return (size_t)tb * cap + (size_t)(b - tb) * cap_top;
}
hipError_t ecamd_launch_edbkt_file(const EcamdEdBktArgs &b, hipStream_t s);
// phase 0: the points; 1: the buckets' sums; 2: the reduction, the windows and the verdict (as ecamd_launch_edmsm_reduce)
hipError_t ecamd_launch_edbkt(const EcamdEdMsmArgs &a, const EcamdEdBktArgs &b, int phase, const uint32_t *flagword, uint8_t *verdict, uint32_t *sum_out,
			      int gslot, hipStream_t s);
hipError_t ecamd_launch_edmsm_scal(const EcamdEdMsmScalArgs &a, hipStream_t s);
// ------------------------------------------------------------------------------------------
// Schnorr-type whole-batch verification on a short-Weierstrass curve as ONE multi-scalar multiplication (BIP0340's and ECFSDSA's
// batch equation, sig/bip0340.c:905-1196, sig/ecfsdsa.c:1042-): with random z_i
//     T = [sum z_i s_i]G + sum_i ([z_i (q - e_i) mod q]Y_i - [z_i]R_i),        accepted when T is the point at infinity.
// Straus evaluation on the radix-2^29 unit of the curve (k_msm_*_g in ecamd_g29_kernel.hip): the 2n points get signed 4-bit
// window tables [1..8]P (Jacobian entries); lane l owns the items j * L + l (j < K) and shares the doublings between their 2K
// points -- the z_i are 128 bits, so the R_i only take part in the low 33 windows; the lanes' sums are added by a tree; the
// generator's term comes from the comb table of the handle.  Every exceptional event (a point that does not import, a table
// multiple at infinity, an addition of equal or opposite points, s >= q) sets a bit of *flagword and the verdict is "not
// decided here" = 1: the caller then verifies item by item, so the batch form never decides anything the item form would not.
// ------------------------------------------------------------------------------------------
struct EcamdMsmArgs {
	const uint8_t *ptsY, *ptsR;  // n x 2*clen affine X || Y big-endian: the keys, the signatures' points (r_fmt 1: n x clen, x only)
	const uint8_t *scW, *scZ;    // n x wlen / n x zlen big-endian: z_i (q - e_i) mod q, z_i
	uint32_t *tbl;               // 2n x ecamd_g29_table_words: tables and recoded scalars (Y items first, then R items)
	uint32_t *rec;               // L x ecamd_g29_msm_rec_words: the lanes' sums (Jacobian X, Y, Z and an "is infinity" word)
	uint32_t *flagword;
	uint32_t n, K, L, clen, wlen, zlen;
	uint32_t r_fmt;              // 1: R_i is the point with the given x and an EVEN y (BIP0340's lift_x, sig/bip0340.c:532-535 / :947-953);
	                             //    p = 3 mod 4 only (the host checks): y = (x^3 + a x + b)^((p + 1) / 4)
	// the bucket evaluation (round 6; phases 10 - 12 of ecamd_launch_msm_g29, k_bkt_* in ecamd_g29_kernel.hip and ecamd_kernels.hip)
	uint32_t *pts;               // 2n affine point records (ecamd_g29_bkt_point_words words each; R_i negated)
	const uint32_t *bstart, *bcount;   // nwin << c: where every bucket's list starts in its window's `order`, and how long it is
	const uint32_t *order;       // nwin x 2n point indices, every window's in bucket order
	const uint32_t *perm;        // nwin << c: lane t of k_bkt_accum_g serves bucket perm[t] (NULL: bucket t)
	uint32_t *bsum;              // nwin << c records: the buckets' sums
	uint32_t *red;               // scratch of the reduction, red_words words
	uint64_t red_words;
	uint32_t c, nwin;            // window bits (11 .. 16), windows of the full-length scalars
	uint32_t cap;                // != 0: bucket t's list is order[t cap .. t cap + min(bcount[t], cap))
	uint32_t cap_top, top_win;   // ... except in window top_win, the order's top window, whose digits take (q >> 16 top_win) + 1 values only: its
	                             //     buckets have cap_top slots, behind the (top_win << c) x cap slots of the windows below (ecamd_bkt_slot)
	uint32_t pt_first, pt_count; // phase 10: the point indices [pt_first, pt_first + pt_count) of the 2n (count 0: all of them)
	uint32_t win_first, win_count;   // phase 11: the windows [win_first, win_first + win_count) (count 0: all of them)
	uint32_t cof_dbl;            // the final test (k_msm_final_g): 0: sum + [c]G is the point at infinity; d > 0: [2^d](sum + [c]G) is -- EdDSA's
	                             //     cofactored equation on a curve of order 2^d q (Ed448 on WEI448: d = 2)
};
uint32_t ecamd_g29_bkt_point_words(int pbits, int flavour);
// the counting sort of the (window, digit, point) triples (ecamd_kernels.hip); point index i < n: Y_i with scalar scW[i], n + i: R_i with scZ[i]
struct EcamdBktSortArgs {
	const uint8_t *scW, *scZ;    // n x wlen, n x zlen big-endian
	uint32_t *hist, *start, *cursor;   // nwin << c counters each (hist and cursor zeroed by the launcher)
	uint32_t *order;             // nwin x 2n
	uint32_t *perm;              // nwin << c: the buckets, every 4096 of them ranked by size (k_bkt_rank); may be NULL
	uint32_t *flag;              // |= 16 when a bucket overflows its `cap` slots
	uint32_t cap;                // != 0: fixed-capacity filing -- hist counts, no scan (k_bkt_file); slots as ecamd_bkt_slot lays them out
	uint32_t cap_top, top_win;
	uint32_t n, wlen, zlen, c, nwin, nwinZ;
	uint32_t win_first, win_count;   // fixed-capacity filing only: the windows [win_first, win_first + win_count) alone, counters NOT cleared (the caller
	                                 // cleared them once; the key-only windows are filed first and summed while the others are filed); count 0: all
	uint32_t part, item_first, item_count;   // fixed-capacity filing, the streamed form (ec_schnorr_verify_msg_all_batch): part 1 files the keys and
	                                 // commitments of the items [item_first, item_first + item_count) alone -- every window, counters as they are (the
	                                 // caller cleared them when the batch began), no ranking; part 2: the ranking alone, once everything is filed
};
hipError_t ecamd_launch_bkt_sort(const EcamdBktSortArgs &a, hipStream_t s);
// scalars of the combination (mod q, saturated unit of the order's size): z_i = 128 bits of ChaCha20(seed; counter = item)
struct EcamdMsmScalArgs {
	const uint8_t *s, *ne;       // n x qlen big-endian: s_i, q - e_i (both must be < q)
	uint8_t *scW, *scZ;          // n x qlen, n x 16 big-endian (outputs)
	uint32_t *v;                 // n x NW words: z_i s_i mod q
	uint32_t *flagword;          // value 8 (k_msm_scal): some s_i or q - e_i is not below q; 1: a point that does not import / a table multiple at infinity (k_msm_table_g); 2, 4: an exceptional addition in k_msm_loop_g / k_msm_sum_g
	uint8_t *z_dump;             // may be NULL: n x 16 little-endian z_i (tests)
	uint32_t seed[8], nonce[3];
	uint32_t n, qlen;
	int qslot;
	uint32_t first, count;       // the items [first, first + count) of the n (count 0: all); z_i is keyed by the item's index in the batch either way
};
struct EcamdMsmVsumArgs {
	const uint32_t *in;          // count x NW words, values < q
	uint32_t *out;               // ceil(count / 64) x NW words
	uint8_t *c_be;               // last level only: the sum, qlen bytes big-endian
	uint32_t count, qlen;
	int qslot;
};
hipError_t ecamd_launch_msm_scal(int nw, const EcamdMsmScalArgs &a, hipStream_t s);
hipError_t ecamd_launch_msm_vsum(int nw, const EcamdMsmVsumArgs &a, hipStream_t s);
uint32_t ecamd_g29_msm_rec_words(int pbits, int flavour);
// phase 0: tables of the 2n points; 1: the Straus loop; 2: the tree sum and the verdict (tmp: ceil(L / 16) records;
// gen: [sum z_i s_i]G affine, 2*clen bytes big-endian, gen_status its status byte; verdict[0] = 0 accept / 1 not decided here;
// sum_out may be NULL: the sum of the lanes, ecamd_g29_msm_rec_words words)
hipError_t ecamd_launch_msm_g29(int pbits, int gslot, int flavour, int phase, const EcamdMsmArgs &a, uint32_t *tmp, const uint8_t *gen,
				const uint8_t *gen_status, uint8_t *verdict, uint32_t *sum_out, hipStream_t s);

hipError_t ecamd_launch_edmsm_lane(const EcamdEdMsmLaneArgs &a, hipStream_t s);
struct EcamdEdScalArgs {
	const uint8_t *sigs;     // n x 2*len: R || S
	const uint8_t *hram;     // n x hlen: H(dom || R || A || M), little-endian integer
	uint8_t *S_be, *h_be;    // out: n x len big-endian scalars S and h mod q
	uint8_t *ne_be;          // out, may be NULL (k_ed448_scal only): n x len big-endian (q - h) mod q, the key's scalar of the batch equation
	uint8_t *flags;          // out: n, 1 when S >= q
	uint32_t n, len, hlen;
	int qslot;
	uint32_t c4_mod4;        // Ed448: (4^-1 mod q) mod 4, see k_ed448_scal
};
// Round 4: k_ed_scal's work plus the half-length scalars (u, v) of the verification equation (k_ed_lat, ecamd_kernels.hip)
struct EcamdEdLatArgs {
	const uint8_t *sigs;     // n x 64: R || S
	const uint8_t *hram;     // n x hlen little-endian
	uint8_t *S_be, *sp_be;   // out: n x 32 big-endian S and s' = u S mod q
	uint32_t *uv;            // out: n x 12 words: v (8, little-endian words), |u| (4)
	uint8_t *meta;           // out: n: bit 0 u < 0, bit 1 full-length scalars (u = 1, v = h), bit 2 v = 0
	uint8_t *flags;          // out: n, 1 when S >= q
	uint32_t n, hlen;
	int qslot;
};
hipError_t ecamd_launch_ed_lat(const EcamdEdLatArgs &a, hipStream_t s);
struct EcamdEdFinArgs {
	const uint8_t *SG, *stSG;   // [S]G
	const uint8_t *hA, *sthA;   // [h]A
	const uint8_t *R;           // decoded R (Weierstrass affine)
	const uint8_t *flagsA, *flagsR, *flagsS;
	uint8_t *result;            // n: 0 accept / 1 reject
	uint32_t n, clen, cof_dbl;  // cof_dbl = log2(cofactor)
	const uint8_t *Akey, *stA;  // Ed448: the decoded key A (affine; [4]A = infinity <=> the stored key [4^-1]A has small order), checked here;
	                            // stA: optional status bytes that reject when non-zero; both NULL otherwise
	int slot;
};
// Ed448 point decoding (EDDSA448 branch of eddsa_decode_point) + maps to the Weierstrass model WEI448
struct EcamdEd448DecodeArgs {
	const uint8_t *encA, *encR;   // n x 57-byte encodings
	uint32_t strideA, strideR;
	uint8_t *pointsA, *pointsR;   // out: n x 112 affine Weierstrass X || Y big-endian
	uint8_t *flagsA, *flagsR;
	uint32_t n;
	uint32_t d448[17], diso[17], alpha[17], A3[17];   // Montgomery form (radix 2^448)
	uint32_t g_d448[16], g_diso[16], g_alpha[16], g_A3[16];   // the same as plain radix-2^29 digits (Goldilocks unit)
	int slot;
};
// the same kernel on the radix-2^29 field of the Goldilocks unit (gslot: its constant slot)
hipError_t ecamd_launch_ed448_decode_g(const EcamdEd448DecodeArgs &a, int gslot, hipStream_t s);
// Ed25519 signing, the device-side steps around the caller's two hashes (sig/eddsa.c:1554-1870)
struct EcamdEdSignArgs {
	const uint8_t *r_hash;   // n x 64 little-endian: H(dom2 || prefix || PH(M))
	const uint8_t *hram;     // S step: n x 64 little-endian H(dom2 || R || A || PH(M))
	const uint8_t *a;        // S step: n x 32 little-endian clamped secret scalars
	uint8_t *r_be;           // r step: n x 32 big-endian r mod q (the scalar of [r]G)
	const uint8_t *Rw;       // encode step: n x 64 affine Weierstrass [r]G
	const uint8_t *stR;      // encode step: its status
	uint8_t *out;            // encode step: n x 32 encoded R; S step: n x 32 little-endian S
	uint8_t *status;         // encode step: n
	uint32_t n;
	uint32_t alpha[17], A3[17];   // Montgomery form, as in EcamdEdDecodeArgs
	uint32_t c4[17];              // Ed448: 4^-1 mod q, plain little-endian words (the scalar of [r]G is r / 4, sig/eddsa.c:1737-1746)
	int slot, qslot;
	int is448;                    // Ed448 on WEI448: 114-byte hashes, 57-byte encodings (the launchers pick the 448-bit kernels)
};
hipError_t ecamd_launch_ed_sign_r(const EcamdEdSignArgs &a, hipStream_t s);
hipError_t ecamd_launch_ed_sign_enc(const EcamdEdSignArgs &a, hipStream_t s);
hipError_t ecamd_launch_ed_sign_S(const EcamdEdSignArgs &a, hipStream_t s);
hipError_t ecamd_launch_ed448_decode(const EcamdEd448DecodeArgs &a, hipStream_t s);
hipError_t ecamd_launch_ed448_scal(const EcamdEdScalArgs &a, hipStream_t s);
// Ed448 whole-batch verification (round 6), the reference's per-item rejections ahead of the batch equation: gate[0] |= 1 when some item has a
// decoding flag set (A, R -- the neutral element included --, S >= q) or a key with [2^cof_dbl]A = infinity (keys: n x 2*clen affine big-endian)
hipError_t ecamd_launch_ed_msm_gate(int nw, const uint8_t *keys_aff, const uint8_t *flagsA, const uint8_t *flagsR, const uint8_t *flagsS, uint32_t n,
				    uint32_t clen, uint32_t cof_dbl, int slot, uint32_t *gate, hipStream_t s);
// verdict[0] = 1 when piece[0] != 0 or gate[0] != 0 (never cleared: the pieces of one call share the byte)
hipError_t ecamd_launch_verdict_or(uint8_t *verdict, const uint8_t *piece, const uint32_t *gate, hipStream_t s);
hipError_t ecamd_launch_ed_decode(int nw, const EcamdEdDecodeArgs &a, hipStream_t s);
// the same two front-end kernels on the radix-2^29 field of the 2^255 - 19 unit (gslot: its constant slot)
hipError_t ecamd_launch_ed_decode_c25519(const EcamdEdDecodeArgs &a, int gslot, hipStream_t s);
hipError_t ecamd_launch_xdh_prep_c25519(const EcamdXdhPrepArgs &a, int gslot, hipStream_t s);
hipError_t ecamd_launch_xdh_prep_c448(const EcamdXdhPrepArgs &a, int gslot, hipStream_t s);   // X448 on the Goldilocks unit
#define ECAMD_X448_REC_WORDS 32
struct EcamdXdhLadderArgs;
hipError_t ecamd_launch_x448_ladder(const EcamdXdhLadderArgs &a, int gslot, hipStream_t s, hipEvent_t *dom = nullptr);  // rec: n x ECAMD_X448_REC_WORDS
// X25519 x-only Montgomery ladder + shared inversion (after k_xdh_prep_c25519 validated and clamped)
struct EcamdXdhLadderArgs {
	const uint8_t *u;        // n x 32 little-endian u coordinates (as given by the caller)
	const uint8_t *scalars;  // n x 32 big-endian clamped scalars (prep kernel)
	const uint8_t *flags;    // n, non-zero: rejected by the prep kernel
	uint32_t *rec;           // scratch: n x ECAMD_XDH_REC_WORDS (X2, Z2)
	uint8_t *out, *status;   // n x 32 little-endian u', n
	uint32_t n;
};
#define ECAMD_XDH_REC_WORDS 20
hipError_t ecamd_launch_x25519_ladder(const EcamdXdhLadderArgs &a, int gslot, hipStream_t s, hipEvent_t *dom = nullptr);   // dom: two events around the ladder kernel
hipError_t ecamd_launch_ed_scal(int nw, const EcamdEdScalArgs &a, hipStream_t s);
hipError_t ecamd_launch_ed_fin(int nw, const EcamdEdFinArgs &a, hipStream_t s);
// the same on the radix-2^29 / 2^28 field of the 2^255 - 19 (flavour 2) / Goldilocks (flavour 5) unit (k_ed_fin_g, ecamd_rcbg.h)
hipError_t ecamd_launch_ed_fin_g29(int flavour, int gslot, const EcamdEdFinArgs &a, hipStream_t s);

// ---- projective wire format X || Y || Z (curves/prj_pt.c:462, 562) ----
struct EcamdPrjInArgs {
	const uint8_t *in;       // n x 3*clen
	uint8_t *aff;            // out: n x 2*clen affine X || Y (zeros unless pre == 0)
	uint8_t *pre;            // out: n, 0 finite / 1 import error / 2 infinity
	uint32_t n, clen;
	int for_mul;             // (0:0:0): error under prj_pt_mul, "infinity" under prj_pt_unique alone
	int slot;
};
struct EcamdPrjOutArgs {
	const uint8_t *aff;      // n x 2*clen
	const uint8_t *st;       // n (may be NULL: all 0)
	const uint8_t *pre;      // n import status that overrides st when non-zero (may be NULL)
	uint8_t *out;            // n x (out_prj ? 3 : 2)*clen
	uint8_t *status;
	uint32_t n, clen;
	int out_prj;
};
// ECC-CDH glue around the scalar multiplications (ecdh/ecccdh.c:187-224)
struct EcamdCdhArgs {
	const uint8_t *st_sub;   // gate: status of [q]Q (2 = in the subgroup) ...
	const uint8_t *st_h;     // ... and of [h]Q (0 = not infinity); both NULL for the final step
	uint8_t *hq;             // gate: n x 2*clen, [h]Q, overwritten with 0xff.. (an import error) when a check fails
	const uint8_t *pts;      // fin: n x 2*clen, [d]Q'
	const uint8_t *st;       // fin: its status
	uint8_t *secrets;        // fin: n x clen, the x coordinate (zeros on failure)
	uint8_t *status;         // fin: 0 / 1
	uint32_t n, clen;
};
hipError_t ecamd_launch_cdh_gate(const EcamdCdhArgs &a, hipStream_t s);
hipError_t ecamd_launch_cdh_fin(const EcamdCdhArgs &a, hipStream_t s);
// aff_pt_y_from_x (curves/aff_pt.c:102) + fp_sqrt (fp/fp_sqrt.c:107: Tonelli-Shanks with the reference's choice of the
// non-residue z = the smallest one, hence its choice of which root is "sqrt1")
struct EcamdYfromXArgs {
	const uint8_t *x;        // n x xstride bytes; the coordinate is the LAST clen bytes of each record
	uint32_t xstride;        // clen, or 1 + clen for SEC1 compressed points (prefix byte 0x02 / 0x03 first)
	uint8_t *y1, *y2;        // out, n x clen big-endian: sqrt1 and sqrt2 = p - sqrt1 (mode 0)
	uint8_t *aff;            // out, n x 2*clen: X || Y with the root whose parity matches the prefix (mode 1)
	uint8_t *status;         // n: 0 ok / 1 (x >= p, no square root, bad prefix)
	uint32_t n, clen, mode;
	uint32_t s;              // p - 1 = q 2^s, q odd
	uint32_t ebits;          // bits of (q - 1) / 2
	uint32_t e[17];          // (q - 1) / 2, little-endian words
	uint32_t c[17];          // z^q in Montgomery form (radix 2^(32 NW)), z the smallest non-residue
	int slot;
};
hipError_t ecamd_launch_y_from_x(int nw, const EcamdYfromXArgs &a, hipStream_t s);
// scalar blinding of prj_pt_mul_blind (curves/prj_pt.c:1782-1822): m' = m + b * #E as a big-endian string of outlen bytes
struct EcamdBlindArgs {
	const uint8_t *m;        // n x mlen big-endian scalars
	const uint8_t *b;        // n x blen big-endian blinding values
	uint8_t *out;            // n x outlen big-endian, outlen >= max(mlen, blen + 4 * owords) + 1
	uint8_t *bad;            // n: 1 where b = 0 or b >= #E (the reference draws b in [1, #E))
	uint32_t n, mlen, blen, outlen, owords;
	uint32_t order[18];      // #E (the CURVE order, cofactor included), little-endian words
};
hipError_t ecamd_launch_blind_scalar(const EcamdBlindArgs &a, hipStream_t s);
// nn_get_random_mod given its random bytes (ecamd_randmod.h): out = LE(raw) mod (q - 1) + 1
struct EcamdRandModArgs {
	const uint8_t *raw;      // n x rawlen (2 * qlen) bytes, read as a little-endian integer
	uint8_t *out;            // n x qlen big-endian
	uint32_t n, rawlen, qlen;
	uint32_t q[18];          // the generator's order, little-endian words
};
hipError_t ecamd_launch_rand_mod(int qnw, const EcamdRandModArgs &a, hipStream_t s);
// Front end of ec_schnorr_verify_msg_all_batch (BIP0340 / ECFSDSA from keys, signatures and hash inputs; round 6), per chunk of m items:
//   k_schnorr_prep  the imported key's x into the blank of the item's hash input (BIP0340), the key as the equation uses it (BIP0340: the
//                   representative with an even y), s and the commitment (r, or W) into the batch-wide arrays; a key that did not import
//                   or is the point at infinity sets *flag
//   k_schnorr_ne    ne = q - (digest mod q) mod q, big-endian (sig/bip0340.c:470-494, :531; sig/ecfsdsa.c:520-561)
struct EcamdSchnorrPrepArgs {
	const uint8_t *keys_aff;   // m x 2 clen: the keys, affine
	const uint8_t *kst;        // m import statuses (0 ok), or NULL (keys given affine: the multi-scalar kernels validate them)
	const uint8_t *sigs;       // m x (rlen + qlen)
	uint8_t *slots;            // m x stride: u32 length, then the hash input
	uint8_t *keys_out, *s_out, *r_out;   // m x 2 clen, m x qlen, m x rlen
	uint32_t *flag;
	uint32_t n, clen, qlen, rlen, stride;
	uint32_t x_off;            // offset of the blank for Y.x inside the hash input, 0xffffffff: none
	uint32_t even_y;           // 1: BIP0340's lift_x of the key (y <- p - y when y is odd)
	uint8_t p_be[72];
};
hipError_t ecamd_launch_schnorr_prep(const EcamdSchnorrPrepArgs &a, hipStream_t s);
struct EcamdSchnorrNeArgs {
	const uint8_t *dig;        // n x hlen big-endian digests
	uint8_t *ne;               // n x qlen big-endian
	uint32_t n, hlen, qlen;
	uint32_t q[18];
};
hipError_t ecamd_launch_schnorr_ne(int qnw, const EcamdSchnorrNeArgs &a, hipStream_t s);
// status[i] = 1 and out[i] zeroed where bad[i] != 0
hipError_t ecamd_launch_status_require(uint8_t *status, const uint8_t *sub, uint8_t want, uint32_t n, hipStream_t s);
hipError_t ecamd_launch_status_or(uint8_t *status, const uint8_t *bad, uint8_t *out, uint32_t out_stride, uint32_t n, hipStream_t s);
hipError_t ecamd_launch_prj_import(int nw, const EcamdPrjInArgs &a, hipStream_t s);
hipError_t ecamd_launch_prj_export(const EcamdPrjOutArgs &a, hipStream_t s);

// radix-2^29 Jacobian fast path for every field size (ecamd_g29_kernel.hip)
int ecamd_g29_supported(int pbits);
int ecamd_g29_nl(int pbits, int flavour);
int ecamd_g29_slots(void);
uint32_t ecamd_g29_table_words(int pbits, int flavour);   // scratch words per item
uint32_t ecamd_g29_affine_words(int pbits, int flavour);  // affine window table words per item (EcamdSmulArgs.stg)
uint32_t ecamd_g29_max_slen(int pbits);      // longest scalar (bytes) the window kernel takes (blinded scalars included)
uint32_t ecamd_g29_comb_max_slen(int pbits); // longest scalar the fixed-base comb takes
size_t ecamd_g29_image_bytes(int pbits, int flavour);
hipError_t ecamd_g29_upload(int pbits, int slot, const void *img, size_t bytes, int flavour);
hipError_t ecamd_launch_smul_g29(int pbits, int gslot, const EcamdSmulArgs &a, hipStream_t s, hipEvent_t *ev, int flavour);
// fixed-base comb tables of the radix-2^29 path (lut_kind 1): geometry and construction from affine big-endian points
uint32_t ecamd_g29_comb_entries(int pbits);
uint32_t ecamd_g29_comb_entry_words(int pbits, int flavour);
hipError_t ecamd_g29_comb_build(int pbits, int gslot, const uint8_t *pts, uint32_t n, uint32_t clen, uint32_t *table,
				hipStream_t s, int flavour);
// SHA-224 / 256 / 384 / 512 of n messages in fixed-stride slots (ecamd_hash.hip); hash_type: libecc's hash_alg_type numbers 1 .. 4
int ecamd_sha2_digest_len(int hash_type);
// the key's encoding written into the item's hash input (bytes 4 + off .. of its slot; items with skip[i] != 0 left alone); result[i] = 1 where status[i] != 0
hipError_t ecamd_launch_slot_patch(uint8_t *slots, uint32_t stride, uint32_t off, const uint8_t *src, uint32_t len, const uint8_t *skip, uint32_t n,
				   hipStream_t s);
hipError_t ecamd_launch_reject_where(uint8_t *result, const uint8_t *status, uint32_t n, hipStream_t s);
// SHAKE256 of the same slots: the first outlen (<= 136) octets of the output per message
hipError_t ecamd_launch_shake256_slots(const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride, uint32_t outlen, hipStream_t s);
hipError_t ecamd_launch_sha2_slots(int hash_type, const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride, hipStream_t s);
struct EcamdPrjInArgs;
// prj_pt_import_from_buf + prj_pt_unique on a radix-2^29 unit, one inversion per eight triples (k_prj_import_g)
hipError_t ecamd_g29_prj_import(int pbits, int gslot, const EcamdPrjInArgs &a, hipStream_t s, int flavour);
struct EcamdEcdsaPrepArgs;
// the mod-q algebra of an ECDSA verification on the dense radix-2^29 unit of the order's size (k_ecdsa_prep_g)
hipError_t ecamd_g29_ecdsa_prep(int qbits, int qgslot, const EcamdEcdsaPrepArgs &a, uint32_t *scratch, int kp, hipStream_t s);
hipError_t ecamd_launch_fp(int nw, const EcamdFpArgs &a, hipStream_t s);
hipError_t ecamd_launch_pt(int nw, const EcamdPtArgs &a, hipStream_t s);
size_t ecamd_curvek_bytes(int nw);
