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
 * threads that marshal libecc structures to and from wire bytes (default: the online CPUs, or $ECAMD_COMPAT_THREADS).
 * Called implicitly by the first batch call. */
int ecamd_compat_init(const int *devices, int ndev, int host_threads);
void ecamd_compat_shutdown(void);
/* A curve the library does not know by name (ec_params built by the application from its own ec_str_params): registered
 * so that prj_pt arrays on it can be mapped to a device-side curve (a prj_pt only points to its ec_shortw_crv, which has
 * no generator).  Built-in curves need no registration. */
int ecamd_compat_register_params(const ec_params *params);
/* statistics of the calling process: items sent to the GPU by the entry points below */
unsigned long long ecamd_compat_gpu_items(void);

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
 * Batch form of ecccdh_derive_secret (ecdh/ecccdh.h:57, ecdh/ecccdh.c:167): item i derives
 * shared_secrets[i] (shared_secret_len bytes, = ecccdh_shared_secret_size) from our_priv_keys[i] and the serialised
 * peer key peer_pub_keys[i] (peer_pub_key_len bytes each, = ecccdh_serialized_pub_key_size).  All keys on one curve.
 */
int ecccdh_derive_secret_batch(const ec_priv_key *const *our_priv_keys, const u8 *const *peer_pub_keys, u8 peer_pub_key_len,
			       u8 *const *shared_secrets, u8 shared_secret_len, u32 num, int *ret_items);

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
 * it may accept a bad batch with probability ~2^-128); smaller groups and Ed448 by the exact conjunction of the
 * per-signature cofactored verifications.
 */
int eddsa_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			   ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			   verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len);

/*
 * The per-item form of both: results[i] = what ec_verify(s[i], s_len[i], pub_keys[i], m[i], m_len[i], sig_type, hash_type,
 * adata[i], adata_len[i]) returns (0 / -1).  Returns 0 when the batch ran, -1 on a call-level error (unsupported
 * algorithm, no GPU).
 */
int ec_verify_batch_results(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			    ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int *results);

/*
 * libsign_amd.so also REPLACES these two libecc symbols (libecc's own definitions are kept under the names
 * libecc_cpu_ec_verify_batch / libecc_cpu_is_verify_batch_mode_supported):
 *   ec_verify_batch (sig/sig_algs.h:90-93): ECDSA, DECDSA -> ecdsa_verify_batch; the EdDSA variants ->
 *     eddsa_verify_batch_gpu; every other algorithm -> libecc's own ec_verify_batch (BIP0340, ECFSDSA on the CPU,
 *     unsupported_verify_batch for the rest);
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
