/*
 * include/libecc_amd_compat.h -- the drop-in boundary in libecc's OWN types and prototypes.
 *
 * Built into libecc_amd/lib/libsign_amd.so (libecc_amd/compat/Makefile): libecc's libsign, compiled from the
 * application's libecc tree, with the batch entry points below added and ec_verify_batch /
 * is_verify_batch_mode_supported replaced by GPU-backed versions.  An application that links -lsign today links
 * -lsign_amd instead and keeps every libec.h / libsig.h symbol; the scalar API (prj_pt_mul, ec_sign, ec_verify,
 * ecccdh_derive_secret, ...) stays libecc's own CPU code -- one scalar multiplication per call cannot use a GPU
 *  -- and the batch forms run on the MI355X through include/libecc_amd.h.
 * File:line references are relative to /root/reference/src.
 *
 * Return convention: libecc's (0 success, -1 error; utils/utils.h:80).  Per-item results, where an array `ret_items`
 * is offered, are the value the scalar function would have returned for that item (0 / -1).
 * There is no CPU fallback for the batch forms: without a gfx950 device they return -1.
 */
#ifndef LIBECC_AMD_COMPAT_H
#define LIBECC_AMD_COMPAT_H

#include "libsig.h" /* libecc's public header: nn, fp, prj_pt, ec_params, ec_pub_key, ec_verify_batch, ... */

#ifdef __cplusplus
extern "C" {
#endif

/* Optional: choose the GPUs (default: every visible device, or the comma list in $ECAMD_DEVICES) and the number of host
 * threads that marshal libecc structures to and from wire bytes (default: the CPUs this process may run on -- its affinity
 * mask, capped by the cgroup CPU quota -- or $ECAMD_COMPAT_THREADS).  The threads are a persistent pool created here.
 * Called implicitly by the first batch call.
 * Secret scalars: the contexts run in secret-scalar mode (ecamd_ctx_set_secret_scalars in libecc_amd.h: constant-address table
 * look-ups for private keys, nonces and blinded scalars, as libecc's masked ladder; verification is not affected) unless
 * $ECAMD_COMPAT_PUBLIC_SCALARS is set; ecamd_compat_set_secret_scalars changes it at run time.  The batch calls that handle
 * private material wipe their host staging and the devices' scratch before they return.
 * Threads (round 6): like libecc, the batch entry points may be entered from several application threads.  ECDSA / DECDSA and EdDSA
 * verification (ec_verify_batch, ec_verify_batch_results and their per-algorithm forms) run up to two calls at a time -- each in its own
 * page-locked staging, the pool packing and unpacking for both, their GPU calls one after the other; a third caller waits.  Every other
 * entry point -- the secret-key half, the key and point forms, BIP0340 / ECFSDSA -- runs one call at a time (and not beside a
 * verification): those switch the devices between secret- and public-scalar mode, wipe scratch, or make dependent GPU calls. */
int ecamd_compat_init(const int *devices, int ndev, int host_threads);
void ecamd_compat_shutdown(void);
int ecamd_compat_set_secret_scalars(int on);
/* The application's get_random (an import of this library, as of libecc) is called by ONE thread at a time: the batch forms pack
 * their items on the pool threads, and the random scalars they draw -- ECDSA nonces when rand == NULL, generated private keys,
 * the blinding factors of prj_pt_mul_blind_batch, the seed of a whole-batch EdDSA combination -- go through a process-wide
 * lock around the get_random call itself (the reduction mod q stays parallel).  libecc never required get_random to be
 * reentrant, and a stateful source that is not could hand out torn or repeated nonces.  An application whose get_random IS
 * thread-safe (one getrandom(2) / /dev/urandom read per call, a locked DRBG) may lift the lock: on = 1, or
 * $ECAMD_COMPAT_CONCURRENT_RANDOM=1 before the first batch call.
 * What is drawn is what libecc draws: nn_get_random_mod's ONE get_random call of 2 * qlen bytes per ECDSA nonce / generic private
 * scalar; since round 4 only that call runs on the host -- the reduction modulo q - 1 of those bytes, like the SHA-2 of short messages,
 * runs on the device ($ECAMD_COMPAT_HOST_RANDMOD, $ECAMD_COMPAT_HOST_HASH: on the host through libecc's own functions, as before).
 * What the batch forms do NOT reproduce is the reference's total consumption of the random stream: libecc's CPU scalar multiplication
 * and modular exponentiation draw their side-channel masks from the same get_random (216 + 3 x 8 octets per fp_inv, 64 - 72 per
 * prj_pt_mul on secp256r1), and ec_key_pair_gen draws its private scalar twice (sig/ec_key.c:602, then gen_priv_key), so a seeded
 * get_random does not yield the key pairs or nonces a loop over the scalar API would -- only values of the same distribution.
 *
 * Environment of the layer (read at the first batch call; for measurements and fall-backs -- results are identical):
 *   ECAMD_DEVICES=0,1,..  ECAMD_COMPAT_THREADS=<n>  ECAMD_COMPAT_CHUNK=<items per pipeline chunk and device>  ECAMD_COMPAT_PUBLIC_SCALARS
 *   ECAMD_COMPAT_CONCURRENT_RANDOM  ECAMD_COMPAT_HOST_HASH  ECAMD_COMPAT_HOST_RANDMOD
 *   ECAMD_COMPAT_NO_STREAM        verification: pack everything, then call the device (default: one call per batch that asks the pool
 *                                 for each range of the arrays through ecamd_multi_set_host_ready_hook while they are being packed)
 *   ECAMD_COMPAT_READY_ITEMS=<n>  granularity of that handshake (default 2^16)
 *   ECAMD_COMPAT_ED_TWO_PASS      EdDSA verification (all five variants): encode the keys in a call of its own and hash on the host
 *   ECAMD_COMPAT_PRJ_KEYS         ECDSA verification: send keys as X || Y || Z even when every Z is 1
 *   ECAMD_COMPAT_TIMING           one line per pipeline run on stderr: where the calling thread's time went
 *   ECAMD_COMPAT_FULL_SCAN        verification: look at every key before packing (parameters, Z = 1) instead of taking one set of parameters
 *                                 for granted and guessing Z = 1 from 64 keys (a packing step that finds otherwise restarts the call with the scan)
 *   ECAMD_COMPAT_NO_PREFETCH      the packing steps do not prefetch the key structures a few items ahead */
void ecamd_compat_set_concurrent_random(int on);
/* A curve the library does not know by name (ec_params built by the application from its own ec_str_params): registered
 * so that prj_pt arrays on it can be mapped to a device-side curve (a prj_pt only points to its ec_shortw_crv, which has
 * no generator).  Built-in curves need no registration. */
int ecamd_compat_register_params(const ec_params *params);
/* statistics of the calling process: items sent to the GPU by the entry points below */
unsigned long long ecamd_compat_gpu_items(void);
/* calls of the Schnorr-type multi-scalar multiplication made on behalf of ec_verify_batch (BIP0340 / ECFSDSA batches of at least 2^17
 * items per device, $ECAMD_COMPAT_SCHNORR_MSM_MIN; a valid batch is accepted by it, any other goes on to the item-by-item pass) */
unsigned long ecamd_compat_schnorr_msm_calls(void);
/* verification calls that started over behind the full pass over the keys: a packing step met a key under other parameters, a missing key, or a
 * Z != 1 after the 64-key sample had said "affine" (tests: the restart must happen, and only then) */
unsigned long ecamd_compat_verify_restarts(void);
/* ... and how many Ed25519 ec_verify_batch groups were first offered to the device as one whole-batch call (round 6: batches of at least
 * 2^18 items per device, $ECAMD_COMPAT_ED_MSM_MIN; 0 = never) */
unsigned long ecamd_compat_ed_msm_calls(void);

/*
 * Batch form of prj_pt_mul (curves/prj_pt.h:61, curves/prj_pt.c:1759): out[i] = [m[i]] in[i] for i < n.
 * All in[i] must be initialised points of ONE curve (the ec_shortw_crv in[0] points to).  out may alias in.
 * ret_items (may be NULL): per item, what prj_pt_mul would have returned (0, also when the result is the point at
 * infinity; -1 e.g. for a point that is not on the curve).  out[i] is the unique representative (Z = 1, as after
 * prj_pt_unique) or (0 : 1 : 0).  Returns 0 when the batch ran (look at ret_items), -1 on a call-level error.
 */
int prj_pt_mul_batch(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items);

/* Batch form of prj_pt_mul_blind (curves/prj_pt.h:62, curves/prj_pt.c:1782): per item a fresh b in [1, #E) from libecc's own
 * nn_get_random_mod, and the device multiplies by m + b * #E (on secp256r1 still on the fast window kernel). */
int prj_pt_mul_blind_batch(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items);

/*
 * Round 4: the rest of curves/prj_pt.h's arithmetic in batch form (SURVEY.md 8b's export list), each item what the scalar function
 * computes, ret_items (may be NULL) what it would have returned.  All points of one call lie on ONE curve (in[0]'s / in1[0]'s);
 * results are the unique representative (Z = 1) or (0 : 1 : 0).  One difference from the scalar functions, which do not look at
 * their operands: a point that does not satisfy the curve equation is an error here (ret_items -1).
 *   prj_pt_add_batch       prj_pt_add (curves/prj_pt.h:59, curves/prj_pt.c:1204): out[i] = in1[i] + in2[i]; -1 also on the addition's exceptional pair
 *                          (the difference of the two points has order two -- only on curves of even order; :1058-1060)
 *   prj_pt_dbl_batch       prj_pt_dbl (prj_pt.h:60, prj_pt.c:1132)
 *   prj_pt_neg_batch       prj_pt_neg (prj_pt.h:57, prj_pt.c:435): out[i] = -in[i] (as the unique representative, where the scalar function keeps Z)
 *   prj_pt_cmp_batch       prj_pt_cmp (prj_pt.h:55, prj_pt.c:303): cmp[i] = 0 when in1[i] and in2[i] are the same projective point, 1 when not (the scalar
 *                          function hands back -1 or 1 there, the sign of a comparison of Montgomery residues: callers test against 0)
 *   prj_pt_eq_or_opp_batch prj_pt_eq_or_opp (prj_pt.h:56, prj_pt.c:412): eq_or_opp[i] = 1 when in1[i] = +-in2[i], else 0
 *   prj_pt_unique_batch    prj_pt_unique (prj_pt.h:54, prj_pt.c:241): -1 for the point at infinity, as the scalar function
 *   prj_pt_is_on_curve_batch  prj_pt_is_on_curve (prj_pt.h:51, prj_pt.c:144): on_curve[i] = 1 / 0
 *   _prj_pt_unprotected_mult_batch  _prj_pt_unprotected_mult (prj_pt.h:64, prj_pt.c:1835-1905): the reference's double-and-add for PUBLIC scalars, bit
 *                          by bit on the device, so that its -1 on an exceptional pair of one of its additions is reproduced too
 *   check_prj_pt_order_batch  check_prj_pt_order (prj_pt.h:65, prj_pt.c:1909): check[i] = 1 when [in_isorder] in[i] is the point at infinity;
 *                          PUBLIC_PT points through the double-and-add above, sensitive ones through prj_pt_mul_blind_batch
 */
int prj_pt_add_batch(prj_pt *out, const prj_pt *in1, const prj_pt *in2, u32 n, int *ret_items);
int prj_pt_dbl_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items);
int prj_pt_neg_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items);
int prj_pt_cmp_batch(const prj_pt *in1, const prj_pt *in2, u32 n, int *cmp, int *ret_items);
int prj_pt_eq_or_opp_batch(const prj_pt *in1, const prj_pt *in2, u32 n, int *eq_or_opp, int *ret_items);
int prj_pt_unique_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items);
int prj_pt_is_on_curve_batch(const prj_pt *in, u32 n, int *on_curve, int *ret_items);
int _prj_pt_unprotected_mult_batch(prj_pt *out, const nn *scalars, const prj_pt *in, u32 n, int *ret_items);
int check_prj_pt_order_batch(const prj_pt *in, nn_src_t in_isorder, prj_pt_sensitivity s, u32 n, int *check, int *ret_items);
/* Batch form of ec_pub_key_import_from_aff_buf (sig/ec_key.h, sig/ec_key.c:181): pub_keys[i] from the affine X || Y octets
 * pub_key_bufs[i] (pub_key_buf_len each): range and curve equation on the device, and -- when the curve's cofactor is not 1 -- the
 * reference's subgroup test [q]Y = infinity by its own double-and-add (check_prj_pt_order, PUBLIC_PT). */
int ec_pub_key_import_from_aff_buf_batch(ec_pub_key *pub_keys, const ec_params *params, const u8 *const *pub_key_bufs, u8 pub_key_buf_len,
					 ec_alg_type ec_key_alg, u32 num, int *ret_items);

/*
 * Batch form of ecccdh_derive_secret (ecdh/ecccdh.h:57, ecdh/ecccdh.c:167): item i derives
 * shared_secrets[i] (shared_secret_len bytes, = ecccdh_shared_secret_size) from our_priv_keys[i] and the serialised
 * peer key peer_pub_keys[i] (peer_pub_key_len bytes each, = ecccdh_serialized_pub_key_size).  All keys on one curve.
 */
int ecccdh_derive_secret_batch(const ec_priv_key *const *our_priv_keys, const u8 *const *peer_pub_keys, u8 peer_pub_key_len,
			       u8 *const *shared_secrets, u8 shared_secret_len, u32 num, int *ret_items);

/* ------------------------------------------------------------------------------------------------------------------------
 * The secret-key half of the path, in libecc's own types.  Hashing, nonce generation (libecc's nn_get_random_mod through the
 * application's get_random, or RFC 6979 through libecc's hmac_*) and the argument checks of the scalar functions run on the
 * host threads; every scalar multiplication, the mod-q algebra of the signatures and the point encodings run on the GPU(s).
 * ret_items (may be NULL) receives per item what the scalar function would have returned (0 / -1); the calls themselves
 * return 0 when the batch ran and -1 on a call-level error (bad argument, unsupported algorithm for this entry point, no GPU).
 * ------------------------------------------------------------------------------------------------------------------------ */

/*
 * Batch form of _ec_sign / ec_sign (sig/sig_algs.h:49-60, sig/sig_algs.c:465-504): item i signs m[i] (m_len[i] bytes) with
 * key_pairs[i] into sigs[i] (siglen bytes each, = ec_get_sig_len).  On the GPU: ECDSA, DECDSA (__ecdsa_sign_finalize,
 * sig/ecdsa_common.c:318-586) and the five EdDSA variants (_eddsa_sign, sig/eddsa.c:1554).  ON THE CPU: every other algorithm of
 * sig/sig_algs_internal.h's table -- ECKCDSA, ECSDSA, ECOSDSA, ECFSDSA, ECGDSA, ECRDSA, SM2, BIGN, DBIGN, BIP0340 -- is signed item by
 * item by libecc's own _ec_sign on the pool threads (cpu_sign_items): the call parallelises them over the host cores and nothing
 * more; they are not part of the accelerated path (the hot path is ECDSA / EdDSA / ECDH).  The host side of
 * the GPU algorithms also uses the application's libecc for what is per call or rare: nn_mod of an over-long private scalar,
 * nn_modinv_fermat for ECKCDSA-type key rules, the HMAC of RFC 6979, the blinding product m + b #E of prj_pt_mul_blind_batch.
 *   rand: as for _ec_sign -- NULL = libecc's nn_get_random_mod (its steps restated around a serialised get_random, see
 *     ecamd_compat_set_concurrent_random); another function is called once per item on the calling thread (test vectors).
 *     Two deviations from a loop of ec_sign calls, both only visible to a caller-supplied hook: (i) the hook must return a value
 *     in [1, q-1] as its contract says ("a value taken uniformly at random in [1, q-1]") -- an item whose hook returns 0 or a
 *     value >= q fails with -1 here, where __ecdsa_sign_finalize would go on with whatever came back; (ii) for a batch whose key
 *     pairs live on several curves the hook is called group by group (one group per ec_params, items in index order inside a
 *     group), not in global index order; with one curve the order is the index order.  DECDSA ignores rand (RFC 6979, as
 *     _decdsa_sign_init forces); EdDSA requires NULL (sig/eddsa.c:1596).
 *   An ECDSA item whose nonce gives r = 0, s = 0 or e = x r is signed again with a fresh nonce, as the reference's restart.
 *   adata / adata_len may be NULL (no context); EDDSA25519CTX needs a context for every item.
 *   Key pairs may live on different curves (one GPU batch per ec_params).
 */
int ec_sign_batch(u8 *const *sigs, u8 siglen, const ec_key_pair *const *key_pairs, const u8 *const *m, const u32 *m_len, u32 num,
		  int (*rand)(nn_t out, nn_src_t q), ec_alg_type sig_type, hash_alg_type hash_type, const u8 *const *adata,
		  const u16 *adata_len, int *ret_items);

/*
 * Batch forms of ec_key_pair_gen (sig/ec_key.c:594) and ec_key_pair_import_from_priv_key_buf (:289): kps[i] receives a key pair
 * for algorithm ec_key_alg on `params` (all items share them).  The private scalars come from libecc's own gen_priv_key
 * (nn_get_random_mod / eddsa_gen_priv_key, i.e. the application's get_random) or from the caller's buffers
 * (priv_keys[i], priv_key_len bytes each, as ec_priv_key_import_from_buf takes them); the public keys Y = [s]G -- s = x, or
 * x^-1 mod q (ECGDSA, ECKCDSA), or the EdDSA scalar derived from the hashed key -- are computed on the GPU in one batch.
 * pub_key.y holds the unique representative (Z = 1); a failed item is zeroed as the scalar functions do.  Also accepts ECCCDH
 * (ecccdh_gen_key_pair, ecdh/ecccdh.c:93).
 */
int ec_key_pair_gen_batch(ec_key_pair *kps, const ec_params *params, ec_alg_type ec_key_alg, u32 num, int *ret_items);
int ec_key_pair_import_from_priv_key_buf_batch(ec_key_pair *kps, const ec_params *params, const u8 *const *priv_keys, u8 priv_key_len,
					       ec_alg_type ec_key_alg, u32 num, int *ret_items);
/* eddsa_import_key_pair_from_priv_key_buf (sig/eddsa.c:1028): the raw EdDSA secret keys are hashed and clamped first */
int eddsa_import_key_pair_from_priv_key_buf_batch(ec_key_pair *kps, const u8 *const *priv_keys, u16 priv_key_len,
						  const ec_params *shortw_curve_params, ec_alg_type sig_type, u32 num, int *ret_items);
/* init_pubkey_from_privkey (sig/sig_algs.c:72; __ecdsa_init_pub_key sig/ecdsa_common.c:172, ecccdh_init_pub_key ecdh/ecccdh.c:60,
 * eddsa_init_pub_key sig/eddsa.c:786, ...): out_pubs[i] from the initialised private key in_privs[i]; all keys of one algorithm
 * and one ec_params. */
int init_pubkey_from_privkey_batch(ec_pub_key *out_pubs, const ec_priv_key *const *in_privs, u32 num, int *ret_items);
/* ecccdh_init_pub_key / ecccdh_gen_key_pair (ecdh/ecccdh.h:33-41) */
int ecccdh_init_pub_key_batch(ec_pub_key *out_pubs, const ec_priv_key *const *in_privs, u32 num, int *ret_items);
int ecccdh_gen_key_pair_batch(ec_key_pair *kps, const ec_params *params, u32 num, int *ret_items);

/*
 * Batch forms of x25519 / x448 (ecdh/x25519_448.h:34,51; x25519_448_core ecdh/x25519_448.c:146): res[i] = X(k[i], u[i]), 32 / 56
 * byte little-endian strings; ret_items[i] = -1 where the reference returns -1 (non-canonical u, u on the twist, a point of small
 * order, an all-zero result).  The *_init_pub_key forms use the base point (u = 9 / 5), the *_derive_secret forms are the
 * function itself under its other name (x25519_448.c:346-356).  The Montgomery ladder runs on the GPU on the curve itself.
 */
int x25519_batch(const u8 *const *k, const u8 *const *u, u8 *const *res, u32 num, int *ret_items);
int x25519_init_pub_key_batch(const u8 *const *priv_keys, u8 *const *pub_keys, u32 num, int *ret_items);
int x25519_derive_secret_batch(const u8 *const *priv_keys, const u8 *const *peer_pub_keys, u8 *const *shared_secrets, u32 num, int *ret_items);
int x448_batch(const u8 *const *k, const u8 *const *u, u8 *const *res, u32 num, int *ret_items);
int x448_init_pub_key_batch(const u8 *const *priv_keys, u8 *const *pub_keys, u32 num, int *ret_items);
int x448_derive_secret_batch(const u8 *const *priv_keys, const u8 *const *peer_pub_keys, u8 *const *shared_secrets, u32 num, int *ret_items);

/*
 * ECDSA / DECDSA batch verification with the exact prototype of the verify_batch slot of ec_sig_mapping
 * (sig/sig_algs_internal.h:78-81), which libecc leaves at unsupported_verify_batch for ECDSA (:294).
 * Returns 0 iff ec_verify (sig/sig_algs.c:655) would return 0 for EVERY item; -1 otherwise (also for num = 0, as the
 * reference's batch verifiers).  adata / adata_len may be NULL (ECDSA ignores them); the scratch pad is not needed and
 * is ignored.  Keys may live on different curves.
 */
int ecdsa_verify_batch(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
		       ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
		       verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len);

/*
 * EdDSA (EDDSA25519, EDDSA25519CTX, EDDSA25519PH, EDDSA448, EDDSA448PH) batch verification, same prototype; replaces
 * eddsa_verify_batch (sig/eddsa.c:2904).  Same argument checks as the reference (one ec_params for all keys, key type =
 * sig_type, hash_type = the variant's hash, signature lengths, scratch-pad length when a scratch pad is given).
 * Ed25519 groups of at least 2^17 signatures per device are decided by the reference's own random linear combination,
 * evaluated as one multi-scalar multiplication on the GPU (ec_eddsa_verify_all_batch in libecc_amd.h; like the reference
 * it may accept a bad batch with probability ~2^-128) -- Ed448 groups too since round 6, on the Weierstrass model with the final test
 * cofactored; smaller groups, the pre-hashed variants and any batch the combination does not vouch for by the exact conjunction of
 * the per-signature cofactored verifications.
 */
int eddsa_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			   ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			   verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len);

/*
 * BIP0340 (Schnorr, any curve, as sig/bip0340.c) batch verification, same prototype; replaces bip0340_verify_batch
 * (sig/bip0340.c:1296) behind ec_verify_batch.  Every signature is verified on the GPU(s) -- the key's unique representative
 * with an even y, [s]G + [q - e]Y, the parity and x = r tests -- and the answer is the exact conjunction (the reference's random
 * linear combination has the same answer up to its 2^-128 error); tagged hashes on the host threads.
 * Round 6: batches of at least 2^17 items per device ($ECAMD_COMPAT_SCHNORR_MSM_MIN) on a prime-order curve whose hash the device has
 * (SHA-224 .. SHA-512, hash inputs up to 252 octets) are first offered to the device as ONE streamed call -- keys, signatures and the
 * tagged-hash inputs travel as ec_verify_init finds them, the device imports the keys, hashes, reduces, lifts and evaluates the
 * reference's batch equation (include/libecc_amd.h: ec_schnorr_verify_msg_all_batch); it vouches for VALID batches only, so anything
 * else -- a bad signature, a key at infinity, an item that fails a length or range check -- is decided by the item-by-item path above.
 * The same entry point serves ECFSDSA (sig/ecfsdsa.c:470-640 per item, ecfsdsa_verify_batch at sig/ecfsdsa.c:1057): signature
 * (r = Wx || Wy, s), e = H(r || m) mod q, accept when [s]G + [q - e]Y is the finite point whose affine coordinates are the bytes
 * of r (both compared on the host after one batched unique-representative pass).
 */
int bip0340_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			     ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			     verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len);

/*
 * The per-item form of all three: results[i] = what ec_verify(s[i], s_len[i], pub_keys[i], m[i], m_len[i], sig_type, hash_type,
 * adata[i], adata_len[i]) returns (0 / -1).  Returns 0 when the batch ran, -1 on a call-level error (unsupported
 * algorithm, no GPU).
 */
int ec_verify_batch_results(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			    ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int *results);

/*
 * libsign_amd.so also REPLACES these two libecc symbols (libecc's own definitions are kept under the names
 * libecc_cpu_ec_verify_batch / libecc_cpu_is_verify_batch_mode_supported):
 *   ec_verify_batch (sig/sig_algs.h:90-93): ECDSA, DECDSA -> ecdsa_verify_batch; the EdDSA variants ->
 *     eddsa_verify_batch_gpu; BIP0340 and ECFSDSA -> bip0340_verify_batch_gpu; every other algorithm -> libecc's own
 *     ec_verify_batch (unsupported_verify_batch for all of them: with these four families every algorithm libecc lists in
 *     is_verify_batch_mode_supported is served by the GPU);
 *   is_verify_batch_mode_supported (sig/sig_algs_internal.h:267): additionally reports ECDSA and DECDSA as supported.
 */
int libecc_cpu_ec_verify_batch(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			       ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			       verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len);
int libecc_cpu_is_verify_batch_mode_supported(ec_alg_type sig_type, int *check);

#ifdef __cplusplus
}
#endif
#endif /* LIBECC_AMD_COMPAT_H */
