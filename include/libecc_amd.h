/*
 * include/libecc_amd.h -- C ABI of the MI355X batched short-Weierstrass scalar-multiplication
 * engine.  This is the drop-in boundary for libecc's hot path: plain C, plain pointers and
 * sizes, libecc's wire formats (big-endian octet strings) and libecc's 0 / -1 return
 * convention (utils/utils.h:80,137-143).  File:line references are relative to
 * /root/reference/src.
 *
 * libecc performs ONE scalar multiplication per call (curves/prj_pt.h:61); a GPU only pays off
 * on batches, so each entry point below is the batch form of the libecc call chain it
 * replaces, with a per-item status byte mirroring the scalar API's 0 / -1:
 *
 *   ECAMD_OK  (0)  the libecc chain would have returned 0 and produced these bytes
 *   ECAMD_ERR (1)  some call of the chain would have returned -1 (coordinate >= p, point not
 *                  on the curve, r/s out of range, ...); the output bytes are zero
 *   ECAMD_INF (2)  only for point results: the result is the point at infinity (prj_pt_mul
 *                  returns 0 with Z = 0; prj_pt_unique / prj_pt_to_aff then return -1,
 *                  curves/prj_pt.c:246-247); the output bytes are zero
 *
 * Every function returns 0 on success and -1 on failure of the call itself (bad argument, no
 * device, HIP error); ecamd_last_error() then describes it.  There is NO CPU fallback: without
 * a gfx950 device ecamd_ctx_create() fails.
 *
 * Threads: a context serialises its calls with a mutex; use one context per thread (or per GPU) for
 * concurrency -- any number of contexts may share a device (the __constant__ curve slots are managed per
 * device, not per context).  Streams: a context owns ONE set of scratch buffers, so its calls execute one
 * after the other on the device even when they are enqueued on different streams (each call makes its stream
 * wait for the previous call's last kernel); use two contexts for two concurrent streams.  Memory: scratch grows with the largest batch seen (about 3 KB per item of a chunk of
 * <= 2^20 items for 256-bit curves, 7 KB for 521 bits); a curve handle that has served a fixed-base batch of >= 4096 items keeps a table of
 * multiples of the generator in HBM (42 MB for 256-bit curves, 183 MB for 521 bits).
 *
 * Environment (read when a context / curve handle is created; for measurements and fallbacks):
 *   ECAMD_HOST_CHUNK=<items>      chunk size of the host-pointer entry points (default 2^19)
 *   ECAMD_HOST_RAMP_MIN=<items>   a call of several chunks starts with a short one, max(chunk / 8, this) items (default 2^16); ECAMD_NO_HOST_RAMP: equal chunks
 *   ECAMD_COMB_MIN_BATCH=<items>  smallest fixed-base batch that builds / uses the generator table (default 4096)
 *   ECAMD_MSM_MIN=<items>, ECAMD_MSM_K=<items per lane>   initial values of ecamd_ctx_set_eddsa_msm (default 2^17, chosen from the batch size)
 *   ECAMD_NO_COMB, ECAMD_NO_FAST_PATH, ECAMD_NO_P25519, ECAMD_NO_K256, ECAMD_NO_P448, ECAMD_NO_MPINV1, ECAMD_NO_ISO, ECAMD_NO_X25519_LADDER,
 *   ECAMD_NO_EDWARDS_SMUL, ECAMD_NO_ED_LATE_MAP, ECAMD_NO_G448_DECODE, ECAMD_NO_X448_LADDER   route around one fast path each (results are identical;
 *                                 ECAMD_NO_MPINV1: secp384r1 on the dense 384-bit unit instead of its signed sparse reduction)
 *   ECAMD_NO_ED_FIN_G             EdDSA verification: the final additions / doublings on the saturated-word kernel instead of the unit's own field
 *   ECAMD_NO_SIDE_STREAM          secp256r1 ECDSA verification: k_ecdsa_prep on the caller's stream instead of the context's second stream
 */
#ifndef LIBECC_AMD_H
#define LIBECC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECAMD_OK 0
#define ECAMD_ERR 1
#define ECAMD_INF 2

typedef struct ecamd_ctx ecamd_ctx;     /* one GPU: device id, stream, scratch buffers */
typedef struct ecamd_curve ecamd_curve; /* ec_params equivalent (curves/ec_params.h:51-87) */

/* ---- context ---- */
int ecamd_device_count(void);
int ecamd_ctx_create(ecamd_ctx **ctx, int device);
void ecamd_ctx_destroy(ecamd_ctx *ctx);
const char *ecamd_last_error(void);
/* Upper bound on the items processed per kernel launch (bounds the per-lane window-table
 * scratch: 16 * 3 * 4*ceil(|p|/32) bytes per item).  Default 2^20. */
int ecamd_ctx_set_max_chunk(ecamd_ctx *ctx, uint32_t max_items);
/* Ed25519 whole-batch verification (ec_eddsa_verify_all_batch) through the multi-scalar multiplication: mode 0 never, 1 for
 * batches of at least min_items (default; min_items = 0 keeps the current threshold, initially 2^17 or $ECAMD_MSM_MIN),
 * 2 always.  items_per_lane: signatures that share one lane's doublings, 0 = chosen from the batch size (1 .. 8). */
int ecamd_ctx_set_eddsa_msm(ecamd_ctx *ctx, int mode, uint32_t min_items, uint32_t items_per_lane);
/* Key of the random z_i of the NEXT whole-batch verification on this context (used once, then wiped): 32 bytes from the caller's
 * own randomness source instead of getrandom -- libsign_amd.so passes bytes of the application's get_random, the import libecc's
 * own batch verifier draws its z_i from (sig/eddsa.c:2388), so a seeded test harness makes the combination reproducible. */
int ecamd_ctx_set_msm_seed(ecamd_ctx *ctx, const uint8_t seed[32]);
/* Drops a seed that no whole-batch call has consumed (a seed keys ONE call: every ec_*_verify_all_batch entry point discards what is
 * left when it returns, and the ecamd_multi_* forms call this for the ranks that received no shard). */
int ecamd_ctx_discard_msm_seed(ecamd_ctx *ctx);
/* Secret scalars.  By default the kernels index their window / comb tables with the scalar's digits (fastest; fine for public
 * scalars: verification, public-key checks).  With this switch on, every multiplication by a caller-supplied scalar issued through
 * the context -- ec_prj_pt_mul_batch*, and inside ec_ecdsa_sign_batch, ec_ecccdh_derive_batch ([d]Q), ec_eddsa_sign_R_batch, key-pair
 * import -- uses constant-address table look-ups (every entry read, the wanted one kept by masking: the posture of the reference's
 * masked ladder, curves/prj_pt.c:1225-1260, and nn_tabselect, nn/nn.c:564), a fixed window count and no digit-indexed comb table (fixed-base
 * multiplications use a 4-bit comb whose windows are scanned whole: k_p256_comb4m, k_comb_g<.., SCAN4>; ECAMD_NO_SECRET_COMB: the scanned window loop).  The
 * radix-2^29 pipelines stay in use: k_p256_loop<KW, MASKED> / k_loop_g<.., MASKED> scan the item's eight affine entries (secp256k1:
 * its eight Jacobian ones); only scalars longer than those kernels take (8 NW + 4 bytes) run on the complete-formula kernel with
 * sixteen-entry scans.  One branch remains: a lane whose accumulator met an exceptional pair of the incomplete addition
 * (probability about 2^-|q| per window for a random scalar) is recomputed by the complete-formula kernel.
 * Scalars that are public by construction -- u1, u2 of ECDSA verification, h and S of EdDSA verification, the group order and
 * cofactor of subgroup checks -- keep the fast look-ups in either mode.  Results are identical; DESIGN.md 2.3 has the cost.
 * X25519 / X448 ladders are address-independent in either mode. */
int ecamd_ctx_set_secret_scalars(ecamd_ctx *ctx, int on);
/* Zero every scratch buffer of the context in HBM (window tables, recoded scalars, staged copies of the caller's arrays): after
 * calls that handled secret scalars nothing derived from them stays behind.  Waits for the context's work; synchronous. */
int ecamd_ctx_wipe_scratch(ecamd_ctx *ctx);
/* The context's own stream (a hipStream_t), the one the *_dev entry points use when given NULL. */
void *ecamd_ctx_stream(ecamd_ctx *ctx);
/* Page-locked host memory (valid for every device) for arrays passed to the host-pointer entry points: their copies then run as
 * asynchronous DMA at PCIe rate.  NULL on failure.  Any host memory works; this is only faster. */
void *ecamd_host_alloc(size_t bytes);
void ecamd_host_free(void *p);
/* Producer hook of the host-pointer entry points.  Those stage the caller's arrays through the device in chunks of host_chunk items
 * (ECAMD_HOST_CHUNK), chunk c+1's copies overlapping chunk c's kernels; a caller that is still FILLING its input arrays while the call
 * runs registers fn, which the call invokes -- on the calling thread, with the context's lock held -- before it reads items
 * [first, first + count) of the call's input arrays, and which returns once that range is complete.  libsign_amd.so uses it so that
 * its host threads marshal libecc structures into the tail of a batch while the GPU already works on its head (it is the whole of the
 * overlap between the typed layer's packing and the device: one call per batch, no quarter-batch launches).  NULL clears it; the
 * hook stays until cleared and applies to every host-pointer call on the context.  fn must not call into the context. */
typedef void (*ecamd_host_ready_fn)(void *arg, uint32_t first, uint32_t count);
int ecamd_ctx_set_host_ready_hook(ecamd_ctx *ctx, ecamd_host_ready_fn fn, void *arg);
/* Measurement hook: when enabled, HIP events are recorded (on the stream the kernels run on) around
 * the kernels of the next ec_prj_pt_mul_batch[_dev] call; ecamd_ctx_kernel_times() waits for them and
 * returns the 4 durations in ms: table, table->affine, window loop, finalisation (the generic
 * radix-2^29 path reports 0, 0, loop, finalisation). */
int ecamd_ctx_enable_kernel_timing(ecamd_ctx *ctx, int on);
int ecamd_ctx_kernel_times(ecamd_ctx *ctx, double *ms, int n);
/* The same hook for the protocol entry points: duration (ms) of the dominant kernel of the last call made with timing enabled --
 * the interleaved window loop of secp256r1 ECDSA verification, the X25519 / X448 ladder, the Edwards window loop of Ed25519
 * verification.  (ECDSA verification on other curves is two scalar multiplications: ecamd_ctx_kernel_times covers them.) */
int ecamd_ctx_dominant_kernel_ms(ecamd_ctx *ctx, double *ms);

/* ---- curves: ec_get_curve_params_by_name (curves/curves.h:21) + import_params
 *      (curves/ec_params.h:89).  Same 44 names as libecc's ec_maps[] ("SECP256R1", ...). ---- */
int ecamd_curve_by_name(ecamd_ctx *ctx, const char *name, ecamd_curve **curve);
/* User-defined curve from big-endian domain parameters (the ec_str_params fields,
 * curves/known/ec_params_external.h:42-102; Montgomery constants are derived here). */
int ecamd_curve_from_params(ecamd_ctx *ctx, const uint8_t *p, uint32_t p_len, const uint8_t *a,
			    uint32_t a_len, const uint8_t *b, uint32_t b_len,
			    const uint8_t *curve_order, uint32_t curve_order_len, const uint8_t *gx,
			    uint32_t gx_len, const uint8_t *gy, uint32_t gy_len,
			    const uint8_t *gen_order, uint32_t gen_order_len, ecamd_curve **curve);
void ecamd_curve_free(ecamd_curve *curve);
int ecamd_curve_coord_len(const ecamd_curve *curve); /* BYTECEIL(p_bitlen): 32 / 48 / 66 ... */
int ecamd_curve_order_len(const ecamd_curve *curve); /* BYTECEIL(bitlen(generator order)) */
int ecamd_curve_words(const ecamd_curve *curve);     /* 32-bit words per field element */

/*
 * ---- the hot path: batched prj_pt_mul ----
 * Replaces, per item i:
 *   nn_init_from_buf(m, scalars + i*scalar_len, scalar_len)            nn/nn.c:479
 *   prj_pt_import_from_aff_buf(P, points + i*2*clen, 2*clen, crv)      curves/prj_pt.c:511
 *     (points == NULL: P = the generator, params->ec_gen)
 *   prj_pt_mul(Q, m, P)                                                curves/prj_pt.c:1759
 *   prj_pt_unique(Q, Q)                                                curves/prj_pt.c:241
 *   prj_pt_export_to_aff_buf(Q, out + i*2*clen, 2*clen)                curves/prj_pt.c:600
 * scalars: n x scalar_len big-endian, ANY value (m >= q is fine, as in libecc).
 * points / out: n x 2*clen, affine X || Y, big-endian.   status: n bytes (ECAMD_*).
 * Host-pointer form: synchronous (H2D, kernel, D2H).
 */
int ec_prj_pt_mul_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n,
			const uint8_t *scalars, uint32_t scalar_len, const uint8_t *points_aff,
			uint8_t *out_aff, uint8_t *status);
/* Batch form of prj_pt_mul_blind (curves/prj_pt.c:1782-1822): per item the scalar actually multiplied is m + b * #E (#E = the curve
 * order, cofactor included), b = blinds + i*blind_len big-endian, supplied by the caller (the reference draws b at random in
 * [1, #E); randomness stays on the host here, as for nonces).  Since [#E]P is the point at infinity the result equals
 * ec_prj_pt_mul_batch's; what changes is the scalar the device walks through -- about 2 |q| bits, which on secp256r1 still runs on
 * the radix-2^29 window kernel.  status ECAMD_ERR also where b = 0 or b >= #E. */
int ec_prj_pt_mul_blind_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *scalars, uint32_t scalar_len,
			      const uint8_t *blinds, uint32_t blind_len, const uint8_t *points_aff, uint8_t *out_aff, uint8_t *status);
/* Device-pointer form: all four buffers already live in HBM; the kernel is enqueued on
 * hip_stream (a hipStream_t, NULL = the context's stream) and the call returns without
 * synchronising. */
int ec_prj_pt_mul_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n,
			    const void *d_scalars, uint32_t scalar_len, const void *d_points_aff,
			    void *d_out_aff, void *d_status, void *hip_stream);
int ecamd_ctx_synchronize(ecamd_ctx *ctx);

/* ---- group law: batched prj_pt_add (curves/prj_pt.c:1204) / prj_pt_dbl (:1132) on affine
 *      encoded inputs, affine encoded output (host pointers) ---- */
int ec_prj_pt_add_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *p1_aff,
			const uint8_t *p2_aff, uint8_t *out_aff, uint8_t *status);
int ec_prj_pt_dbl_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *p_aff,
			uint8_t *out_aff, uint8_t *status);

/* ---- field level: batched fp ops on libecc's limb layout: n x nlimbs little-endian 64-bit
 *      words per operand, nlimbs = ceil(|p|/64) (nn/nn.h:42-45).  Inputs must be < p.
 *   ECAMD_FP_MUL_MONTY  fp_mul_monty = nn_mul_redc1: a*b*2^(-64*nlimbs) mod p
 *                       (fp/fp_montgomery.c:44, nn/nn_mul_redc1.c:246)
 *   ECAMD_FP_ADD / SUB  fp_add / fp_sub (fp/fp_add.c:23,69)
 *   ECAMD_FP_MUL        plain fp_mul (fp/fp_mul.c:23)
 *   ECAMD_FP_INV        fp_inv (fp/fp_mul.c:51), b ignored; inv(0) = 0 ---- */
#define ECAMD_FP_MUL_MONTY 0
#define ECAMD_FP_ADD 1
#define ECAMD_FP_SUB 2
#define ECAMD_FP_MUL 3
#define ECAMD_FP_INV 4
int ec_fp_op_batch(ecamd_ctx *ctx, const ecamd_curve *curve, int op, uint32_t n, const uint64_t *a,
		   const uint64_t *b, uint64_t *out);

/*
 * ---- protocol callers of the hot path ----
 * ECDSA verification, batch of independent (public key, signature, digest) triples: per item
 *   ec_pub_key_import_from_aff_buf(pub, params, pubkeys + i*2*clen, 2*clen, ECDSA)   sig/ec_key.c:181
 *   ec_verify(sig, 2*qlen, pub, m, mlen, ECDSA, hash, NULL, 0)                       sig/sig_algs.c:655
 * where the caller supplies h = H(m) (hashing stays on the host; the reference exposes the same
 * split as ecdsa_verify_raw, sig/fuzzing_ecdsa.h).  sigs: n x 2*qlen (r || s big-endian),
 * digests: n x digest_len.  result[i] = 0 accept / 1 reject (ec_verify's 0 / -1).
 */
int ec_ecdsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys_aff,
			  const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result);
/* The same with a choice of public-key format: ECAMD_PT_PROJECTIVE keys are n x 3*clen, X || Y || Z as ec_pub_key_export_to_buf
 * writes pub_key->y (sig/ec_key.c:254): imported like prj_pt_import_from_buf, normalised on the device.  A key that is the point
 * at infinity is a key for libecc (its import accepts (0 : 1 : 0)); verification against it is W' = uG, reproduced here. */
int ec_ecdsa_verify_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
			      const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result);
/* Verification from MESSAGES instead of digests (round 4): the hash is computed on the device.  Item i's message sits in a slot of
 * msg_stride bytes (one stride per call, a multiple of 4, at most 4096): a little-endian u32 length, then the bytes
 * (4 + length <= msg_stride).  hash_type: libecc's hash_alg_type numbers (hash/hash_algs.h) 1 = SHA224, 2 = SHA256, 3 = SHA384,
 * 4 = SHA512 -- the digests are FIPS 180-4's, the bytes libecc's hfunc_* produce.  Everything else is ec_ecdsa_verify_batch_fmt /
 * ec_eddsa_verify_batch; for EdDSA the slot holds what the verifier hashes, dom2 || R || A || PH(M) (Ed25519: SHA-512).
 * Why: end to end, a libecc application spends more host time hashing short messages (0.3 - 0.5 us each in portable C) than on
 * everything else it does per signature; a million short messages are less than 0.1 ms of one kernel. */
int ec_ecdsa_verify_msg_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
				  const uint8_t *sigs, int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result);
int ec_eddsa_verify_msg_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
			      const uint8_t *hash_slots, uint32_t stride, uint8_t *result);
/* (The two entry points below also take the WEI448 handle: Ed448, 57-octet encodings, 114-octet signatures, SHAKE256 with 114 octets of
 * output -- 64 for the pre-hash --, keys n x 3*56 bytes.)
 * The same from the PROJECTIVE key an ec_pub_key holds (n x 3*32 bytes X || Y || Z on WEI25519, the layout of ec_pub_key_export_to_buf's
 * payload): the key is imported and normalised as prj_pt_import_from_buf / prj_pt_unique do, encoded as eddsa_export_pub_key does
 * (sig/eddsa.c:795), and the 32 octets are written into the item's hash input at message offset a_offset (bytes 4 + a_offset .. of its
 * slot, which the caller leaves blank and counts in the slot's length; the caller's array is not modified) before hashing -- what
 * ec_eddsa_encode_point_batch, a copy back, and ec_eddsa_verify_msg_batch would do in three steps.  An item whose key does not import
 * is rejected (result 1); a key at infinity encodes as (0, 1) and is rejected by the small-order test like in the reference. */
int ec_eddsa_verify_msg_prj_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				  const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result);
/* ec_verify_batch's ONE bit for plain Ed25519 / Ed25519ctx from the same inputs (round 6): the front end per staging chunk as above, then
 * the reference's batch equation over the whole batch as one multi-scalar multiplication per max_chunk items (by buckets from 2^18 items
 * on; Ed448 on the WEI448 handle: 57-octet encodings, SHAKE256, the combination of ec_eddsa_verify_all_batch's Ed448 form).  *all_valid = 0
 * means "not decided here" -- a bad signature, a key that does not import or has no encoding, or a handle without the form: verify
 * item by item (ec_eddsa_verify_msg_prj_batch).  Ed25519, one piece, by buckets: the decodings, the scalars and the filing run on every
 * staging chunk (2^17 items) as it lands, the bucket sums and their reduction after the last ($ECAMD_NO_ED_STREAM: everything after the last). */
int ec_eddsa_verify_msg_prj_all_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				      const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, int *all_valid);
/* The pre-hashed variant (EDDSA25519PH, sig/eddsa.c:1049-1080, :1995-2045): the hash input is dom2(1, context) || R || A || PH(M) with
 * PH(M) = SHA-512(M).  The caller leaves 96 blank octets at message offset a_offset (A, then PH(M)) and hands the messages over in
 * slots of their own (msg_slots, msg_stride: u32 length + bytes); both hashes run on the device. */
int ec_eddsa_verify_ph_prj_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				 const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
				 uint8_t *result);
/* ECDSA signing with caller-supplied nonces: per item the tail of ec_sign / __ecdsa_sign_finalize
 * (sig/ecdsa_common.c:318-586) -- kG = prj_pt_mul(k, G), r = kG.x mod q, s = k^-1 (x r + e) mod q --
 * with h = H(m) and the nonce k supplied by the caller (random, or RFC 6979 computed on the host;
 * the reference's KAT harness injects k through the same ctx->rand hook).  privs, nonces: n x qlen;
 * sigs: n x 2*qlen.  status[i] = 1 where the reference would fail or restart (private key >= q, k not in [1, q-1],
 * r = 0, e == x r, s = 0). */
int ec_ecdsa_sign_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *privs,
			const uint8_t *nonces, const uint8_t *digests, uint32_t digest_len, uint8_t *sigs,
			uint8_t *status);
/* nn_get_random_mod (nn/nn_rand.c:92-150) given its random bytes.  The reference draws 2 * qlen bytes with get_random straight into the
 * limb array of an nn (they read as a little-endian integer on the little-endian hosts libecc and this library run on), reduces modulo
 * q - 1 and adds one.  raw: n x 2*qlen bytes from the caller's own randomness source; out: n x qlen big-endian, each in [1, q - 1].  The
 * reduction is libecc's constant-time division on the host otherwise -- about a microsecond of a host thread per value. */
int ec_nn_random_mod_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *raw, uint8_t *out);
/* ec_ecdsa_sign_batch from MESSAGES and RAW nonce material: per item k = the nn_get_random_mod value of its 2*qlen random bytes (what
 * __ecdsa_sign_finalize draws through ctx->rand = nn_get_random_mod, sig/ecdsa_common.c:424-470), h = SHA-224 / 256 / 384 / 512 of its
 * message slot (hash_type 1 .. 4; slots as for ec_ecdsa_verify_msg_batch_fmt), or hash_type 0: the slots are the digests themselves,
 * msg_stride bytes each.  Signature bytes equal those of ec_ecdsa_sign_batch fed with the reduced nonces and the digests. */
int ec_ecdsa_sign_msg_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw,
			    int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status);
/* ec_key_pair_gen's generic rule (sig/ec_key.c:594-610): x = nn_get_random_mod(q) from the item's 2*qlen random bytes, Y = [x]G.
 * priv_out: n x qlen big-endian; pub_out: n x 2*clen affine X || Y; status as ec_prj_pt_mul_batch.  (Secret scalars: see
 * ecamd_ctx_set_secret_scalars; the private keys cross the bus on their way back, as supplied ones do on their way in.) */
int ec_key_pair_gen_raw_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *raw, uint8_t *priv_out,
			      uint8_t *pub_out_aff, uint8_t *status);
/* ECC-CDH, batch form of ecccdh_derive_secret (ecdh/ecccdh.c:167): privs n x qlen, peers n x 2*clen
 * affine, secrets n x clen (x coordinate of d*Q), status[i] = 0 ok / 1 the reference returns -1.
 * Cofactor curves: subgroup check of the peer key and the [h]Q step as in the reference. */
int ec_ecccdh_derive_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *privs,
			   const uint8_t *peers_aff, uint8_t *secrets, uint8_t *status);

/* X25519 / X448, batch form of x25519() / x448() (ecdh/x25519_448.c:380-425): curve must be WEI25519
 * (32-byte strings) or WEI448 (56-byte strings), the Weierstrass models libecc itself computes on.
 * k, u, out: n x len little-endian (RFC 7748 wire format).  status[i] = 1 where the reference returns
 * -1: non-canonical u (>= p), u on the twist, small-order point, zero result (libecc deliberately
 * rejects these, x25519_448.c:219-276). */
int ec_xdh_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *k, const uint8_t *u,
		 uint8_t *out, uint8_t *status);

/* EdDSA verification, batch form of eddsa_import_pub_key (sig/eddsa.c:862) + ec_verify (_eddsa_verify_init :1846,
 * _eddsa_verify_finalize :2130), on the Weierstrass models libecc itself computes on:
 *   curve = WEI25519: EDDSA25519 / EDDSA25519CTX / EDDSA25519PH.  pubkeys n x 32 (RFC 8032 encoding of A), sigs n x 64
 *     (R || S), hram n x 64 = SHA-512(dom2 || R || A || PH(M)), hram_len = 64;
 *   curve = WEI448:   EDDSA448 / EDDSA448PH.  pubkeys n x 57, sigs n x 114, hram n x 114 =
 *     SHAKE256(dom4 || R || A || PH(M), 114), hram_len = 114.
 * The hash is computed by the caller (the variants differ only in that hash input).  Note for Ed448: libecc stores
 * [4^-1 mod q]A and hashes the key RE-ENCODED from it, which differs from the given bytes when A has a torsion
 * component; hash those bytes to reproduce it (for keys in the prime-order subgroup they are the key itself).
 * result[i] = 0 accept / 1 where the reference returns -1: non-canonical or undecodable A or R, the neutral point,
 * S >= q, [cofactor]A = infinity, or [cofactor]([S]G - R - [h]A) != infinity (libecc checks the cofactored equation). */
int ec_eddsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys,
			  const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, uint8_t *result);

/* Whole-batch predicate of ec_verify_batch (sig/sig_algs.c:675; eddsa_verify_batch sig/eddsa.c:2904,
 * _eddsa_verify_batch_no_memory :2278) for the EdDSA variants: *all_valid = 1 iff libecc's batch verification accepts.
 * libecc rejects num = 0, as this does (-1).  first_rejected (may be NULL) receives the lowest rejected index, n if none --
 * the reference gives no such hint and callers re-verify one by one.  Same inputs as ec_eddsa_verify_batch.
 *   Ed25519 batches of at least 2^17 items (ecamd_ctx_set_eddsa_msm): the reference's own equation
 *     [8]([-sum z_i S_i]B + sum [z_i h_i]A_i + sum [z_i]R_i) = 0, z_i 128 random bits (ChaCha20 on the device, keyed by 32
 *     bytes of getrandom per call), after the reference's per-item rejections (decoding, S >= q, [8]A_i = 0) -- evaluated as
 *     ONE multi-scalar multiplication on the Edwards curve (Straus, the 256 doublings shared by up to 8 signatures per lane).
 *     Like libecc's, this test accepts a batch with a bad signature with probability ~2^-128.  A batch it rejects is then
 *     verified item by item, which yields first_rejected.
 *   Ed448 batches of at least 2^17 items (round 6): the same equation on the Weierstrass model WEI448 the reference computes on -- A_i, R_i
 *     decoded as for ec_eddsa_verify_batch, the Schnorr-type combination [sum z_i S_i]G + sum [z_i (q - h_i)]A_i - sum [z_i]R_i on the
 *     Goldilocks unit (buckets of 16-bit windows; the Straus loop below 2^17 items when ecamd_ctx_set_eddsa_msm says "always"), and the
 *     final test cofactored, [4](...) = infinity, as every point of the reference's combination enters multiplied by the cofactor
 *     (sig/eddsa.c:2580-2860).  [4] kills the torsion components, so the scalars may be taken mod q and the key may be the decoded A
 *     (libecc stores [4^-1 mod q]A); a key with [4]A = infinity, an undecodable key or commitment, S >= q, and a commitment that
 *     decodes to the neutral element (no affine Weierstrass form; the item form accepts it) make the combination "not decided" and
 *     the items are verified one by one.  Measured: 26 ms per 2^20 signatures against 70 ms item by item (profiles/r6_ed448_msm.md).
 *   otherwise (small batches): every item is verified; the bit is the exact conjunction. */
int ec_eddsa_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys,
			      const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
			      uint32_t *first_rejected);

/* Whole-batch predicate of ec_verify_batch for the Schnorr-type algorithms libecc verifies in batches on a short-Weierstrass curve --
 * BIP0340 (bip0340_verify_batch sig/bip0340.c:1296, _bip0340_verify_batch_no_memory :808-1025, _bip0340_verify_batch :1027-1294) and
 * ECFSDSA (ecfsdsa_verify_batch sig/ecfsdsa.c:1057, _ecfsdsa_verify_batch_no_memory :657-837) -- as ONE multi-scalar multiplication.  The item form of both is  [s_i]G + [q - e_i]Y_i = R_i  (bip0340.c:531-547,
 * ecfsdsa.c:561-576); the reference's batch form draws scalars a_i and accepts when
 *     [-sum a_i s_i]G + sum [a_i]R_i + sum [a_i e_i]Y_i   is the point at infinity (bip0340.c:925-1002).
 * Here: z_i = 128 bits of ChaCha20 (keyed by 32 bytes of getrandom per call, or ecamd_ctx_set_msm_seed), and
 *     T = [sum z_i s_i]G + sum ([z_i ne_i mod q]Y_i - [z_i]R_i),     *all_valid = 1 iff T is the point at infinity
 * evaluated on the radix-2^29 unit of the curve by a Straus loop whose doublings are shared by up to 8 signatures per lane (the R_i
 * only enter the low 33 windows).  s, ne: n x qlen big-endian, ne_i = q - e_i mod q as the item form multiplies it; keys_aff: n x 2*clen,
 * the keys Y_i as the item form uses them (BIP0340: after lift_x); r: r_fmt 0 = n x 2*clen affine points R_i (ECFSDSA: the signature's
 * W), r_fmt 1 = n x clen x coordinates, R_i = the point with that x and an EVEN y (BIP0340: lift_x of r_i, computed on the device;
 * fields with p = 3 mod 4).
 * *all_valid = 0 means "not decided here": some item fails the equation, or a point does not import / r_i is no abscissa / s_i >= q /
 * an addition met equal or opposite operands, or the handle has no such unit (ec_schnorr_verify_all_available) -- the caller then
 * verifies item by item, so the batch form never accepts or rejects anything the item form would not (a batch with a bad
 * signature passes with probability ~2^-128, as the reference's does).  That bound needs a group of PRIME order: on a curve with a
 * cofactor a commitment shifted by a point D of small order passes the combination whenever z_i D = O (probability 1 / ord(D)), so
 * handles with cofactor != 1 (WEI25519, WEI448) are never served -- ec_schnorr_verify_all_available is 0 and *all_valid stays 0
 * (the reference's own batch form has that weakness; a loop of ec_verify does not, and this form must not differ from the loop).
 * libecc rejects num = 0, as this does (-1). */
int ec_schnorr_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *s, const uint8_t *ne,
				const uint8_t *keys_aff, const uint8_t *r, int r_fmt, int *all_valid);
int ec_schnorr_verify_all_available(const ecamd_curve *curve, int r_fmt);   /* 1: the multi-scalar form serves this handle */
/* The same verdict FROM keys, signatures and hash inputs (round 6) -- what an ec_verify_batch of BIP0340 / ECFSDSA signatures needs so that
 * only marshalling stays on the host.  keys: n points in key_fmt (ECAMD_PT_AFFINE X || Y or ECAMD_PT_PROJECTIVE X || Y || Z, what an
 * ec_pub_key holds; imported and normalised on the device); sigs: n x (rlen + qlen), the commitment then s, rlen = clen for r_fmt 1
 * (BIP0340: r, an abscissa) and 2 * clen for r_fmt 0 (ECFSDSA: the point W); hash_slots: per item a little-endian u32 length and the
 * scheme's hash input (stride a multiple of 4, at most 4096): for BIP0340  H(tag) || H(tag) || r || <blank of clen octets> || m  with
 * x_offset the blank's offset in the input -- the device writes the x of the key's unique representative there (sig/bip0340.c:437-494)
 * -- and for ECFSDSA  W.x || W.y || m  with x_offset = 0xffffffff (sig/ecfsdsa.c:520-540); hash_type 1 .. 4 (SHA-224 .. SHA-512).  The
 * device computes e = H(input) mod q and q - e, takes the key's even-y representative for r_fmt 1 (lift_x, bip0340.c:532-535), and
 * evaluates ec_schnorr_verify_all_batch's combination over the whole batch once the last chunk has arrived.  *all_valid = 0: not
 * decided here -- also when a key does not import or is the point at infinity.
 * A batch of one piece (n <= max_chunk) evaluated by buckets is FILED CHUNK BY CHUNK: the points' import, the scalars z_i, z_i (q - e_i)
 * and the filing of every staging chunk (2^17 items) run while the next chunk is on its way, and only the bucket sums and their reduction
 * wait for the last one; z_i is keyed by the item's index in the batch, so the verdict does not depend on the chunking
 * ($ECAMD_NO_SCHNORR_STREAM: the whole combination after the last chunk). */
int ec_schnorr_verify_msg_all_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *keys, int key_fmt,
				    const uint8_t *sigs, int r_fmt, int hash_type, const uint8_t *hash_slots, uint32_t stride,
				    uint32_t x_offset, int *all_valid);
/* The same combination on device pointers, one piece (n <= the context's max_chunk; a handle for which ec_schnorr_verify_all_available
 * is 0 is an error here): d_verdict[0] = 0 "the batch is valid" / 1 "not decided here".  Only enqueues on the stream. */
int ec_schnorr_verify_all_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_s, const void *d_ne,
				    const void *d_keys_aff, const void *d_r, int r_fmt, void *d_verdict, void *hip_stream);

/* eddsa_export_pub_key in batch (sig/eddsa.c:795-860): n projective Weierstrass points X || Y || Z (what ec_pub_key.y holds,
 * prj_pt_export_to_buf) of the WEI25519 / WEI448 handle -> prj_pt_shortw_to_aff_pt_edwards -> eddsa_encode_point: n x 32 / 57
 * octets, the public-key encoding the verifier hashes.  status ECAMD_ERR for a point that is not on the curve; the point at
 * infinity encodes the neutral element (0, 1) as in the reference. */
int ec_eddsa_encode_point_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *points_prj,
				uint8_t *enc, uint8_t *status);

/* The multi-scalar multiplication alone, device pointers (WEI25519 handle, hram_len 64; WEI448 handle, hram_len 114: round 6; any n > 0):
 * d_verdict[0] = 0 when libecc's batch equation holds and no item is rejected beforehand, 1 otherwise ("not decided here").  Only
 * enqueues on the stream. */
int ec_eddsa_verify_all_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_pubkeys,
				  const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_verdict, void *hip_stream);

/* Ed25519 signing (EDDSA25519 / EDDSA25519CTX / EDDSA25519PH on the WEI25519 handle), the device-side steps of ec_sign ->
 * _eddsa_sign (sig/eddsa.c:1554-1870) around the caller's two hashes -- the split ec_eddsa_verify_batch uses:
 *   1. the caller derives (a, prefix) from each key (eddsa_get_digest_from_priv_key :306 + eddsa_derive_priv_key :611: SHA-512, clamping) and hashes
 *      r_hash = SHA-512(dom2 || prefix || PH(M))                                          n x 64 bytes
 *   2. ec_eddsa_sign_R_batch: r = r_hash mod q (:1731), R = prj_pt_mul(r, G) (:1776), prj_pt_shortw_to_aff_pt_edwards +
 *      eddsa_encode_point (:1786-1791) -> R_enc n x 32 (the first half of each signature).  status 1 only if the
 *      multiplication failed (it cannot for the generator); r = 0 mod q encodes the neutral element as the reference does.
 *   3. the caller hashes hram = SHA-512(dom2 || R || A || PH(M))                           n x 64 bytes
 *   4. ec_eddsa_sign_S_batch: S = (r + hram a) mod q (:1847-1857), a_scalars n x 32 little-endian (the clamped secret
 *      scalars) -> S_out n x 32 little-endian (the second half).
 * Ed448 (EDDSA448 / EDDSA448PH) on the WEI448 handle, same two calls: r_hash and hram are n x 114 (SHAKE256 with 114 bytes of
 * output), R_enc, a_scalars and S_out n x 57; the device multiplies the generator by r / 4 mod q and encodes through the 4-isogeny
 * back to Edwards448, as _eddsa_sign does (:1737-1746, eddsa_encode_point :350-395). */
int ec_eddsa_sign_R_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc,
			  uint8_t *status);
int ec_eddsa_sign_S_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *r_hash, const uint8_t *hram,
			  const uint8_t *a_scalars, uint8_t *S_out);

/* Point wire formats (curves/prj_pt.c:462-624): affine X || Y (2*clen bytes, what the entry points above
 * use) and projective X || Y || Z (3*clen bytes: prj_pt_import_from_buf / prj_pt_export_to_buf, the format
 * of `ec_utils scalar_mult` and of structured public keys). */
#define ECAMD_PT_AFFINE 0
#define ECAMD_PT_PROJECTIVE 1
/* ec_prj_pt_mul_batch with a choice of formats: import (prj_pt_import_from_buf :462 or
 * prj_pt_import_from_aff_buf :511), prj_pt_mul, prj_pt_unique, export (prj_pt_export_to_buf :562 gives
 * X/Z || Y/Z || 1, or prj_pt_export_to_aff_buf :600).  Projective inputs: every coordinate < p and the
 * projective curve equation must hold; Z = 0 is the point at infinity (status ECAMD_INF); the degenerate
 * triple (0:0:0) passes the reference's import but fails in its ladder (status ECAMD_ERR). */
int ec_prj_pt_mul_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *scalars,
			    uint32_t scalar_len, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
			    uint8_t *status);
/* prj_pt_import_from_[aff_]buf + prj_pt_unique / prj_pt_to_aff (:241, :218) + export: validates and
 * normalises n points (one inversion each on the device).  ECAMD_INF for Z = 0 (including (0:0:0),
 * which prj_pt_unique reports as infinity). */
int ec_prj_pt_unique_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *points, int in_fmt,
			   uint8_t *out, int out_fmt, uint8_t *status);

/* The group law, the on-curve test and the public-scalar multiplication in either wire format (round 4; SURVEY.md 8a rows a17, a18,
 * a21, a22 as callable batch operations).
 *   ec_prj_pt_op_batch_fmt: op ECAMD_PT_OP_ADD = prj_pt_add (curves/prj_pt.c:1204, the complete formulas of :971-1071; an
 *     "exceptional pair" -- the difference of the two points has order two, only possible on a curve of even order -- is the
 *     reference's -1, :1058-1060: ECAMD_ERR), ECAMD_PT_OP_DBL = prj_pt_dbl (:1132), ECAMD_PT_OP_ON_CURVE = prj_pt_is_on_curve (:144;
 *     status 0 on the curve / 1 not, out may be NULL).  Inputs as prj_pt_import_from_[aff_]buf takes them: every coordinate < p and
 *     the projective curve equation -- the reference's prj_pt_add / prj_pt_dbl do not test their operands, here a point off the
 *     curve is ECAMD_ERR; Z = 0 on the curve is the point at infinity, (0 : 0 : 0) included.  Output: the unique representative
 *     (prj_pt_unique: X / Z || Y / Z [|| 1]) or ECAMD_INF with zero bytes.
 *     (round 6) ECAMD_PT_OP_NEG = prj_pt_neg (:435): (X : -Y : Z), handed back as the unique representative like the sum.
 *     ECAMD_PT_OP_CMP = prj_pt_cmp (:303) and ECAMD_PT_OP_EQ_OR_OPP = prj_pt_eq_or_opp (:412) of p1[i] and p2[i]: `out` is ONE BYTE
 *     per item (out_fmt is ignored) -- CMP: 0 where the reference's *cmp is 0 (X1 Z2 = X2 Z1 and Y1 Z2 = Y2 Z1: the same point;
 *     like the reference no special case for Z = 0, so (0 : 0 : 0) compares equal to everything), 1 where it is not (the reference
 *     hands back the sign of a comparison of Montgomery residues in its word size; that sign is not reproduced); EQ_OR_OPP: the
 *     reference's *eq_or_opp, 1 for P = +-Q, else 0.  status 0, or ECAMD_ERR for an operand off the curve (out byte 0).
 *   ec_prj_pt_unprotected_mult_batch: _prj_pt_unprotected_mult (curves/prj_pt.c:1835-1880) statement for statement -- on-curve test,
 *     zero scalar -> infinity, out = in, then per bit below the top one a doubling and, when the bit is set, an addition whose
 *     exceptional pair is the call's -1 -- so that the batch form returns what the scalar function returns for EVERY public scalar
 *     and point (the window kernels return the same group element but never fail that way).  scalar_stride = scalar_len: one
 *     big-endian scalar per item; scalar_stride = 0: one scalar for every item, which is check_prj_pt_order (:1909) for PUBLIC_PT:
 *     "in_isorder is a multiple of the point's order" <=> status ECAMD_INF. */
#define ECAMD_PT_OP_ADD 0
#define ECAMD_PT_OP_DBL 1
#define ECAMD_PT_OP_ON_CURVE 2
#define ECAMD_PT_OP_NEG 3
#define ECAMD_PT_OP_CMP 4
#define ECAMD_PT_OP_EQ_OR_OPP 5
int ec_prj_pt_op_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *curve, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2, int in_fmt,
			   uint8_t *out, int out_fmt, uint8_t *status);
int ec_prj_pt_unprotected_mult_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *scalars, uint32_t scalar_len,
				     uint32_t scalar_stride, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status);

/* Structured public keys, batch form of ec_structured_pub_key_import_from_buf (sig/ec_key.c:312): each key is
 * 3 header bytes -- EC_PUBKEY (0), the ec_alg_type the key is for (ECDSA = 1, ...), libecc's ec_curve_type -- followed by
 * the projective X || Y || Z of ec_pub_key_export_to_buf.  Checks the header, imports (prj_pt_import_from_buf), checks the
 * subgroup on cofactor curves (ec_pub_key_import_from_buf :216) and hands back affine X || Y, the format the verification
 * entry points take.  status ECAMD_ERR where libecc returns -1, ECAMD_INF for a key that is the point at infinity (libecc
 * imports it; it has no affine form).  Built-in curves only (a user curve has no ec_curve_type). */
int ec_structured_pub_key_import_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *keys,
				       uint32_t key_len, int alg_type, uint8_t *out_aff, uint8_t *status);

/* Point decompression.  ec_aff_pt_y_from_x_batch is the batch form of aff_pt_y_from_x (curves/aff_pt.c:102): x n x clen big-endian ->
 * the two roots y1, y2 (n x clen each) of x^3 + a x + b IN THE REFERENCE'S ORDER -- y1 is the root its fp_sqrt (fp/fp_sqrt.c:107,
 * Tonelli-Shanks with the smallest non-residue) returns first, y2 = p - y1 (both 0 when the right-hand side is 0); status 1 where
 * the reference returns -1 (x >= p, not a square).  ec_point_decompress_batch takes SEC 1 compressed points (n x (1 + clen): 0x02 /
 * 0x03, then x) and writes the affine X || Y whose y has the parity the prefix asks for -- the format the other entry points take. */
int ec_aff_pt_y_from_x_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *x, uint8_t *y1, uint8_t *y2,
			     uint8_t *status);
int ec_point_decompress_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *compressed, uint8_t *out_aff,
			      uint8_t *status);
/* Structured signatures (ec_structured_sig_import_from_buf, sig/sig_algs.c:702: 3 bytes -- ec_alg_type, hash_alg_type, ec_curve_type --
 * in front of the raw signature): checks the three bytes against what the caller expects and the handle's curve, strips them
 * (raw_sigs: n x (structured_len - 3)); host-side framing, no device work.  Built-in curves only. */
int ec_structured_sig_import_batch(const ecamd_curve *curve, uint32_t n, const uint8_t *structured, uint32_t structured_len,
				   int alg_type, int hash_type, uint8_t *raw_sigs, uint8_t *status);
/* Structured private keys -> key pairs, batch form of ec_structured_key_pair_import_from_priv_key_buf (sig/ec_key.c:443) for the
 * algorithms whose public key is Y = xG (ECDSA, DECDSA: __ecdsa_init_pub_key, sig/ecdsa_common.c:172): each key is EC_PRIVKEY (1),
 * the ec_alg_type, the ec_curve_type, then the private scalar (key_len - 3 bytes, any length).  Header and x < q checked;
 * priv_out (may be NULL): n x qlen, the scalar as the signing entry points take it; pub_prj_out: n x 3*clen, Y = [x]G as
 * X || Y || 1 (the format of ec_pub_key_export_to_buf and of ec_ecdsa_verify_batch_fmt).  status ECAMD_ERR where libecc
 * returns -1, ECAMD_INF for x = 0 (libecc's key at infinity). */
int ec_structured_key_pair_import_batch(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *priv_keys,
					uint32_t key_len, int alg_type, uint8_t *priv_out, uint8_t *pub_prj_out, uint8_t *status);

/* Device-pointer forms of the verification / key-agreement entry points, for callers whose batches
 * already live in HBM (and for sharding a batch over GPUs, one context per device): same semantics and
 * layouts as the host-pointer forms above, every buffer a device pointer, kernels enqueued on
 * hip_stream (a hipStream_t; NULL = the context's stream).  They only enqueue (ec_ecdsa_verify_batch_dev included: the
 * items its interleaved secp256r1 loop could not finish are re-verified by kernels on the same stream) -- synchronise the
 * stream (or ecamd_ctx_synchronize for the context's stream) before reading.  The first call of a size may allocate
 * scratch (hipMalloc / hipFree, which synchronise the device); later calls of that size or smaller do not. */
int ec_ecdsa_verify_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_pubkeys_aff,
			      const void *d_sigs, const void *d_digests, uint32_t digest_len, void *d_result,
			      void *hip_stream);
int ec_eddsa_verify_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_pubkeys,
			      const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_result,
			      void *hip_stream);
int ec_xdh_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_k, const void *d_u,
		     void *d_out, void *d_status, void *hip_stream);
/* ec_ecdsa_sign_batch / ec_ecccdh_derive_batch with device pointers: enqueue only. */
int ec_ecdsa_sign_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_privs,
			    const void *d_nonces, const void *d_digests, uint32_t digest_len, void *d_sigs,
			    void *d_status, void *hip_stream);
int ec_ecccdh_derive_batch_dev(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const void *d_privs,
			       const void *d_peers_aff, void *d_secrets, void *d_status, void *hip_stream);

/*
 * ---- several GPUs from C ----
 * One context and one host thread per device; a batch of n items is cut into contiguous shards (rank r of N owns
 * items [r*n/N, (r+1)*n/N): the work per item is constant, so equal counts are balanced); nothing is exchanged to
 * compute.  The host-pointer entry points below are the single-device ones run on every shard at once, each
 * device copying its results straight into the caller's arrays, so outputs and status bytes are exactly those of
 * the single-device call.  devices == NULL (or ndev <= 0): every visible device.  A device may be listed more than
 * once (one context and one shard per entry; they then share that GPU).
 */
typedef struct ecamd_multi ecamd_multi;
typedef struct ecamd_mcurve ecamd_mcurve; /* one ecamd_curve per rank */
int ecamd_multi_create(ecamd_multi **m, const int *devices, int ndev);
void ecamd_multi_destroy(ecamd_multi *m);
int ecamd_multi_size(const ecamd_multi *m);                 /* number of ranks */
int ecamd_multi_device(const ecamd_multi *m, int rank);     /* HIP device of a rank */
ecamd_ctx *ecamd_multi_ctx(ecamd_multi *m, int rank);       /* its context (device-pointer entry points, streams) */
void ecamd_multi_shard_range(uint32_t n, int rank, int nranks, uint32_t *lo, uint32_t *hi);
int ecamd_multi_curve_by_name(ecamd_multi *m, const char *name, ecamd_mcurve **curve);
int ecamd_multi_curve_from_params(ecamd_multi *m, const uint8_t *p, uint32_t p_len, const uint8_t *a, uint32_t a_len,
				  const uint8_t *b, uint32_t b_len, const uint8_t *curve_order, uint32_t curve_order_len,
				  const uint8_t *gx, uint32_t gx_len, const uint8_t *gy, uint32_t gy_len,
				  const uint8_t *gen_order, uint32_t gen_order_len, ecamd_mcurve **curve);
void ecamd_multi_curve_free(ecamd_mcurve *curve);
const ecamd_curve *ecamd_multi_curve_handle(const ecamd_mcurve *curve, int rank);
int ecamd_multi_curve_coord_len(const ecamd_mcurve *curve);
int ecamd_multi_curve_order_len(const ecamd_mcurve *curve);
/* sharded forms: same arguments and results as the single-device entry points of the same name */
int ecamd_multi_prj_pt_mul_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *scalars,
				 uint32_t scalar_len, const uint8_t *points_aff, uint8_t *out_aff, uint8_t *status);
int ecamd_multi_prj_pt_mul_batch_fmt(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *scalars,
				     uint32_t scalar_len, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
				     uint8_t *status);
int ecamd_multi_prj_pt_add_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *p1_aff, const uint8_t *p2_aff,
				uint8_t *out_aff, uint8_t *status);
int ecamd_multi_prj_pt_op_batch_fmt(ecamd_multi *m, const ecamd_mcurve *curve, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2,
				    int in_fmt, uint8_t *out, int out_fmt, uint8_t *status);
int ecamd_multi_prj_pt_unprotected_mult_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *scalars,
					      uint32_t scalar_len, uint32_t scalar_stride, const uint8_t *points, int in_fmt, uint8_t *out,
					      int out_fmt, uint8_t *status);
int ecamd_multi_prj_pt_unique_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *points, int in_fmt,
				    uint8_t *out, int out_fmt, uint8_t *status);
int ecamd_multi_ecdsa_verify_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys_aff,
				   const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result);
int ecamd_multi_ecdsa_verify_batch_fmt(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
				       const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result);
int ecamd_multi_ecdsa_verify_msg_batch_fmt(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
					   const uint8_t *sigs, int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result);
int ecamd_multi_eddsa_verify_msg_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				       const uint8_t *hash_slots, uint32_t stride, uint8_t *result);
int ecamd_multi_eddsa_verify_msg_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					   const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result);
int ecamd_multi_eddsa_verify_msg_prj_all_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					       const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, int *all_valid);
int ecamd_multi_eddsa_verify_ph_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					  const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
					  uint8_t *result);
int ecamd_multi_ecdsa_sign_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *privs,
				 const uint8_t *nonces, const uint8_t *digests, uint32_t digest_len, uint8_t *sigs,
				 uint8_t *status);
int ecamd_multi_ecdsa_sign_msg_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw,
				     int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status);
int ecamd_multi_key_pair_gen_raw_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *raw, uint8_t *priv_out,
				       uint8_t *pub_out_aff, uint8_t *status);
int ecamd_multi_ecccdh_derive_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *privs,
				    const uint8_t *peers_aff, uint8_t *secrets, uint8_t *status);
int ecamd_multi_xdh_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *k, const uint8_t *u,
			  uint8_t *out, uint8_t *status);
int ecamd_multi_eddsa_verify_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys,
				   const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, uint8_t *result);
int ecamd_multi_eddsa_encode_point_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *points_prj,
					 uint8_t *enc, uint8_t *status);
/* ec_verify_batch's whole-batch bit, sharded (see ec_eddsa_verify_all_batch: Ed25519 shards of at least 2^17 items run the
 * multi-scalar multiplication on their device); first_rejected (may be NULL): lowest rejected index of the whole batch, n if none */
int ecamd_multi_eddsa_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *pubkeys,
				       const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
				       uint32_t *first_rejected);
/* ec_schnorr_verify_all_batch, sharded: every device decides its shard with a combination of its own; valid iff every shard is */
int ecamd_multi_schnorr_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *s, const uint8_t *ne,
					 const uint8_t *keys_aff, const uint8_t *r, int r_fmt, int *all_valid);
int ecamd_multi_schnorr_verify_msg_all_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *keys, int key_fmt,
					     const uint8_t *sigs, int r_fmt, int hash_type, const uint8_t *hash_slots, uint32_t stride,
					     uint32_t x_offset, int *all_valid);
int ecamd_multi_eddsa_sign_R_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc,
				   uint8_t *status);
int ecamd_multi_eddsa_sign_S_batch(ecamd_multi *m, const ecamd_mcurve *curve, uint32_t n, const uint8_t *r_hash, const uint8_t *hram,
				   const uint8_t *a_scalars, uint8_t *S_out);
/* ecamd_ctx_set_secret_scalars / ecamd_ctx_wipe_scratch on every rank's context */
int ecamd_multi_set_secret_scalars(ecamd_multi *m, int on);
/* ecamd_ctx_set_host_ready_hook for every rank: fn sees item numbers of the WHOLE call's arrays (a rank's shard offset is added) and may be
 * entered from several ranks' threads at once */
int ecamd_multi_set_host_ready_hook(ecamd_multi *m, ecamd_host_ready_fn fn, void *arg);
int ecamd_multi_set_msm_seed(ecamd_multi *m, const uint8_t seed[32]);   /* rank r: seed with r xored into its first bytes */
int ecamd_multi_wipe_scratch(ecamd_multi *m);
/* The one collective, for callers that keep device-resident outputs on every GPU: an RCCL all-gather (over xGMI) of
 * equal-size shards.  d_send[r]: bytes_per_rank bytes on rank r's device; d_recv[r]: nranks * bytes_per_rank bytes
 * there.  librccl is loaded on first use; needs distinct devices.  The gathers run on private streams that first wait for
 * everything enqueued so far on each rank's context stream (ecamd_multi_ctx(m, r)'s, where the *_dev entry points put their
 * kernels by default), so device-resident producers need no host synchronisation in between; ecamd_multi_allgather_streams
 * takes the producers' streams (one hipStream_t per rank, NULL entries = the context's) for shards written elsewhere.
 * Returns when the gathered data is in place on every rank. */
int ecamd_multi_allgather(ecamd_multi *m, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank);
int ecamd_multi_allgather_streams(ecamd_multi *m, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank,
				  void *const *producer_streams);

/* Test hook of the Ed25519 multi-scalar multiplication: the combination with a caller-chosen 32-byte seed.  z_out (n x 16
 * little-endian z_i) and sum_out (36 words: X, Y, Z, T of the sum before the cofactor, nine radix-2^29 digits each of lazily
 * reduced residues mod 2^255 - 19) may be NULL.  n <= the context's max_chunk. */
int ecamd_debug_eddsa_msm(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
			  const uint8_t *hram, const uint8_t seed[32], int *accept, uint8_t *z_out, uint32_t *sum_out);

/* The same for the Schnorr-type multi-scalar multiplication (ec_schnorr_verify_all_batch).  sum_out: the lanes' sum before the generator's
 * term, ecamd_debug_schnorr_msm_words() words (X, Y, Z digits of the unit's representation, then an "is infinity" word). */
int ecamd_debug_schnorr_msm(ecamd_ctx *ctx, const ecamd_curve *curve, uint32_t n, const uint8_t *s, const uint8_t *ne, const uint8_t *keys_aff,
			    const uint8_t *r, int r_fmt, const uint8_t seed[32], int *accept, uint8_t *z_out, uint32_t *sum_out);
uint32_t ecamd_debug_schnorr_msm_words(const ecamd_curve *curve);

#ifdef __cplusplus
}
#endif
#endif /* LIBECC_AMD_H */
