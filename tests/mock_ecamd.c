/*
 * tests/mock_ecamd.c -- TEST INFRASTRUCTURE ONLY.  A CPU stand-in for the part of include/libecc_amd.h that
 * libecc_amd/compat/libecc_amd_compat.c calls (the ecamd_multi_* entry points), built on the oracle
 * (oracle/ecc_oracle.c) and on libecc itself.  It exists so that the HOST logic of the libecc-typed boundary -- the
 * marshalling of nn / prj_pt / ec_key_pair, the thread pool and the pack | GPU | unpack pipeline, the grouping and the
 * error paths -- can be run by `pytest -m "not gpu"` in a container without a GPU (tests/test_compat_host.py links
 * compat_check.c + libecc_amd_compat.c + this file).  It is never linked into libsign_amd.so or libecc_amd.so; the product
 * has no CPU path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libsig.h"
#include "libecc_amd.h"
#include "../oracle/ecc_oracle.h"

struct ecamd_multi {
	int nranks;
	int secret;
};
struct ecamd_mcurve {
	orc_curve c;
	ec_params params;
	int have_params;
	int cof1;   /* the curve's order is the generator's: cofactor 1 */
};

static const char *g_err = "";
const char *ecamd_last_error(void) { return g_err; }
static int mfail(const char *m) { g_err = m; return -1; }

static unsigned long g_calls, g_max_items;
unsigned long mock_ecamd_calls(void) { return g_calls; }
unsigned long mock_ecamd_max_items(void) { return g_max_items; }
static void note(uint32_t n) { __atomic_add_fetch(&g_calls, 1, __ATOMIC_RELAXED); if (n > g_max_items) { g_max_items = n; } }

int ecamd_multi_create(ecamd_multi **m, const int *devices, int ndev)
{
	(void)devices;
	if (getenv("MOCK_ECAMD_NO_DEVICE")) {
		return mfail("ecamd_multi_create: no HIP device available (mock)");
	}
	*m = calloc(1, sizeof(**m));
	(*m)->nranks = ndev > 0 ? ndev : 1;
	return 0;
}
void ecamd_multi_destroy(ecamd_multi *m) { free(m); }
int ecamd_multi_size(const ecamd_multi *m) { return m ? m->nranks : 0; }
int ecamd_multi_set_secret_scalars(ecamd_multi *m, int on) { m->secret = on; return 0; }
int ecamd_multi_wipe_scratch(ecamd_multi *m) { (void)m; return 0; }
int ecamd_multi_set_msm_seed(ecamd_multi *m, const uint8_t seed[32]) { (void)m; (void)seed; return 0; }
/* the producer hook: the mock's batch calls read their arrays in one go, so they ask for the whole range first */
static ecamd_host_ready_fn g_ready_fn;
static void *g_ready_arg;
int ecamd_multi_set_host_ready_hook(ecamd_multi *m, ecamd_host_ready_fn fn, void *arg) { (void)m; g_ready_fn = fn; g_ready_arg = arg; return 0; }
static void mock_ready(uint32_t n) { if (g_ready_fn) { uint32_t o; for (o = 0; o < n; o += 200) { g_ready_fn(g_ready_arg, o, (n - o) < 200 ? (n - o) : 200); } } }
void *ecamd_host_alloc(size_t bytes) { return malloc(bytes); }
void ecamd_host_free(void *p) { free(p); }

static int curve_from_ec_params(const ec_params *params, ecamd_mcurve **out)
{
	u8 b[7][80];
	aff_pt g;
	bitcnt_t ob = 0;
	u32 clen = (u32)BYTECEIL(params->ec_fp.p_bitlen), qlen = (u32)BYTECEIL(params->ec_gen_order_bitlen), olen;
	ecamd_mcurve *c = calloc(1, sizeof(*c));
	if (nn_bitlen(&params->ec_curve.order, &ob)) return -1;
	olen = (u32)BYTECEIL(ob);
	if (prj_pt_to_aff(&g, &params->ec_gen) || nn_export_to_buf(b[0], (u16)clen, &params->ec_fp.p) || fp_export_to_buf(b[1], (u16)clen, &params->ec_curve.a) ||
	    fp_export_to_buf(b[2], (u16)clen, &params->ec_curve.b) || nn_export_to_buf(b[3], (u16)olen, &params->ec_curve.order) ||
	    fp_export_to_buf(b[4], (u16)clen, &g.x) || fp_export_to_buf(b[5], (u16)clen, &g.y) || nn_export_to_buf(b[6], (u16)qlen, &params->ec_gen_order) ||
	    orc_curve_init(&c->c, b[0], (int)clen, b[1], (int)clen, b[2], (int)clen, b[3], (int)olen, b[4], (int)clen, b[5], (int)clen, b[6], (int)qlen)) {
		free(c);
		return mfail("mock: curve setup failed");
	}
	{
		int cmp = 1;
		c->cof1 = !nn_cmp(&params->ec_curve.order, &params->ec_gen_order, &cmp) && cmp == 0;
	}
	*out = c;
	return 0;
}

int ecamd_multi_curve_by_name(ecamd_multi *m, const char *name, ecamd_mcurve **curve)
{
	const ec_str_params *sp = NULL;
	ec_params params;
	(void)m;
	if (ec_get_curve_params_by_name((const u8 *)name, (u8)(strlen(name) + 1), &sp) || !sp || import_params(&params, sp)) {
		return mfail("mock: unknown curve");
	}
	if (curve_from_ec_params(&params, curve)) {
		return -1;
	}
	/* libecc structures point into themselves: import again, in place */
	(*curve)->have_params = !import_params(&(*curve)->params, sp);
	return 0;
}

int ecamd_multi_curve_from_params(ecamd_multi *m, const uint8_t *p, uint32_t p_len, const uint8_t *a, uint32_t a_len,
				  const uint8_t *b, uint32_t b_len, const uint8_t *curve_order, uint32_t curve_order_len,
				  const uint8_t *gx, uint32_t gx_len, const uint8_t *gy, uint32_t gy_len,
				  const uint8_t *gen_order, uint32_t gen_order_len, ecamd_mcurve **curve)
{
	ecamd_mcurve *c = calloc(1, sizeof(*c));
	(void)m;
	if (orc_curve_init(&c->c, p, (int)p_len, a, (int)a_len, b, (int)b_len, curve_order, (int)curve_order_len, gx, (int)gx_len, gy, (int)gy_len,
			   gen_order, (int)gen_order_len)) {
		free(c);
		return mfail("mock: curve setup failed");
	}
	while (curve_order_len > 1 && !curve_order[0]) { curve_order++; curve_order_len--; }
	while (gen_order_len > 1 && !gen_order[0]) { gen_order++; gen_order_len--; }
	c->cof1 = curve_order_len == gen_order_len && !memcmp(curve_order, gen_order, gen_order_len);
	*curve = c;
	return 0;
}
void ecamd_multi_curve_free(ecamd_mcurve *curve) { free(curve); }
int ecamd_multi_curve_coord_len(const ecamd_mcurve *c) { return c ? c->c.clen : -1; }
int ecamd_multi_curve_order_len(const ecamd_mcurve *c) { return c ? c->c.qlen : -1; }

int ecamd_multi_prj_pt_mul_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars, uint32_t slen,
				 const uint8_t *points_aff, uint8_t *out_aff, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_scalar_mult_batch(&c->c, n, scalars, slen, points_aff, out_aff, status);
}

int ecamd_multi_prj_pt_mul_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars, uint32_t slen,
				     const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	const size_t cl = (size_t)c->c.clen;
	uint8_t *tmp;
	uint32_t i;
	(void)m;
	note(n);
	uint8_t *prj = NULL;
	if (in_fmt != ECAMD_PT_PROJECTIVE) {
		/* affine X || Y: the same point with Z = 1 */
		prj = calloc((size_t)n + 1, 3 * cl);
		for (i = 0; i < n; i++) {
			memcpy(prj + i * 3 * cl, points + i * 2 * cl, 2 * cl);
			prj[i * 3 * cl + 3 * cl - 1] = 1;
		}
		points = prj;
	}
	tmp = malloc((size_t)n * 3 * cl + 1);
	if (orc_prj_batch(&c->c, n, scalars, slen, points, tmp, status)) {
		free(tmp);
		free(prj);
		return mfail("mock: orc_prj_batch");
	}
	free(prj);
	for (i = 0; i < n; i++) {
		if (out_fmt == ECAMD_PT_PROJECTIVE) {
			memcpy(out + i * 3 * cl, tmp + i * 3 * cl, 3 * cl);
		} else {
			memcpy(out + i * 2 * cl, tmp + i * 3 * cl, 2 * cl);
		}
	}
	free(tmp);
	return 0;
}

int ecamd_multi_prj_pt_unique_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
				    uint8_t *status)
{
	return ecamd_multi_prj_pt_mul_batch_fmt(m, c, n, NULL, 0, points, in_fmt, out, out_fmt, status);
}

int ecamd_multi_prj_pt_op_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2, int in_fmt,
				    uint8_t *out, int out_fmt, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_pt_op_batch_fmt(&c->c, op, n, p1, p2, in_fmt, out, out_fmt, status);
}

int ecamd_multi_prj_pt_unprotected_mult_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars, uint32_t scalar_len,
					      uint32_t scalar_stride, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_unprotected_mult_batch(&c->c, n, scalars, scalar_len, scalar_stride, points, in_fmt, out, out_fmt, status);
}

int ecamd_multi_prj_pt_add_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *p1_aff, const uint8_t *p2_aff, uint8_t *out_aff,
				 uint8_t *status)
{
	(void)m;
	note(n);
	return orc_pt_add_batch(&c->c, n, p1_aff, p2_aff, out_aff, status, 0);
}

int ecamd_multi_ecdsa_verify_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
				       const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result)
{
	mock_ready(n);
	const size_t cl = (size_t)c->c.clen;
	uint8_t *prj = malloc((size_t)n * 3 * cl + 1), *aff = malloc((size_t)n * 2 * cl + 1), *st = malloc(n + 1);
	uint32_t i;
	int r;
	(void)m;
	note(n);
	if (pub_fmt == ECAMD_PT_AFFINE) {
		/* affine keys (the typed layer sends them when every key of a group has Z = 1): straight to the oracle */
		r = orc_ecdsa_verify_batch(&c->c, n, pubkeys, sigs, digests, digest_len, result);
		free(prj); free(aff); free(st);
		return r ? mfail("mock: ecdsa verify") : 0;
	}
	r = orc_prj_batch(&c->c, n, NULL, 0, pubkeys, prj, st);
	for (i = 0; i < n; i++) {
		memcpy(aff + i * 2 * cl, prj + i * 3 * cl, 2 * cl);
	}
	r = r || orc_ecdsa_verify_batch(&c->c, n, aff, sigs, digests, digest_len, result);
	for (i = 0; i < n && !r; i++) {
		if (st[i]) {
			result[i] = 1;   /* a key at infinity (status 2) is not modelled here: tests/ on the GPU cover it */
		}
	}
	free(prj); free(aff); free(st);
	return r ? mfail("mock: ecdsa verify") : 0;
}

/* the message-taking forms: the stand-in hashes the slots with libecc's own hash functions (what the device computes with
 * ecamd_hash.hip, tested against hashlib in tests/test_hash_host.py) and goes on with the digest-taking forms */
/* hash_type 1 .. 4: SHA-224 .. SHA-512; 5: SHAKE256 as libecc configures it (114 octets); *dl on entry: octets to keep per digest
 * (0: all of them) -- the pre-hash of Ed448ph keeps 64 */
static int mock_hash_slots(int hash_type, uint32_t n, const uint8_t *slots, uint32_t stride, uint8_t *dg, uint32_t *dl)
{
	static const hash_alg_type types[6] = {UNKNOWN_HASH_ALG, SHA224, SHA256, SHA384, SHA512, SHAKE256};
	const hash_mapping *hm;
	const uint32_t keep = *dl;
	uint32_t i;
	if (hash_type < 1 || hash_type > 5 || get_hash_by_type(types[hash_type], &hm) || !hm || keep > hm->digest_size) {
		return -1;
	}
	*dl = keep ? keep : hm->digest_size;
	for (i = 0; i < n; i++) {
		const uint8_t *sl = slots + (size_t)i * stride;
		const uint32_t len = (uint32_t)sl[0] | ((uint32_t)sl[1] << 8) | ((uint32_t)sl[2] << 16) | ((uint32_t)sl[3] << 24);
		uint8_t full[MAX_DIGEST_SIZE];
		hash_context hc;
		if (len > stride - 4 || hm->hfunc_init(&hc) || hm->hfunc_update(&hc, sl + 4, len) || hm->hfunc_finalize(&hc, full)) {
			return -1;
		}
		memcpy(dg + (size_t)i * *dl, full, *dl);
	}
	return 0;
}

int ecamd_multi_ecdsa_verify_msg_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
					   const uint8_t *sigs, int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result)
{
	mock_ready(n);
	if (getenv("MOCK_ACCEPT_ALL")) {   /* timing harness of the typed layer's host side: no verification at all */
		memset(result, 0, n);
		return 0;
	}
	uint8_t *dg = malloc((size_t)n * 64 + 1);
	uint32_t dl = 0;
	int r = (!dg || mock_hash_slots(hash_type, n, msg_slots, msg_stride, dg, &dl)) ? mfail("mock: hashing failed")
		    : ecamd_multi_ecdsa_verify_batch_fmt(m, c, n, pubkeys, pub_fmt, sigs, dg, dl, result);
	free(dg);
	return r;
}

int ecamd_multi_eddsa_verify_msg_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				       const uint8_t *hash_slots, uint32_t stride, uint8_t *result)
{
	mock_ready(n);
	uint8_t *dg = malloc((size_t)n * 64 + 1);
	uint32_t dl = 0;
	int r = (!dg || mock_hash_slots(4, n, hash_slots, stride, dg, &dl)) ? mfail("mock: hashing failed")
		    : ecamd_multi_eddsa_verify_batch(m, c, n, pubkeys, sigs, dg, 64, result);
	free(dg);
	return r;
}

int ecamd_multi_ecdsa_sign_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs, const uint8_t *nonces,
				 const uint8_t *digests, uint32_t digest_len, uint8_t *sigs, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_ecdsa_sign_batch(&c->c, n, privs, nonces, digests, digest_len, sigs, status);
}

int ecamd_multi_ecdsa_sign_msg_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw,
				     int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status)
{
	const size_t ql = (size_t)c->c.qlen;
	uint8_t *k = malloc((size_t)n * ql + 1), *dg = malloc((size_t)n * 64 + 1);
	uint32_t dl = msg_stride;
	int r = (!k || !dg) ? mfail("mock: out of memory") : orc_random_mod_batch(&c->c, n, nonce_raw, k);
	if (!r && hash_type) {
		dl = 0;
		r = mock_hash_slots(hash_type, n, msg_slots, msg_stride, dg, &dl) ? mfail("mock: hashing failed") : 0;
	}
	r = r || ecamd_multi_ecdsa_sign_batch(m, c, n, privs, k, hash_type ? dg : msg_slots, dl, sigs, status);
	free(k); free(dg);
	return r;
}

int ecamd_multi_key_pair_gen_raw_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *raw, uint8_t *priv_out,
				       uint8_t *pub_out_aff, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_random_mod_batch(&c->c, n, raw, priv_out) || orc_scalar_mult_batch(&c->c, n, priv_out, (uint32_t)c->c.qlen, NULL, pub_out_aff, status);
}

int ecamd_multi_ecccdh_derive_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs, const uint8_t *peers_aff,
				    uint8_t *secrets, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_ecccdh_batch(&c->c, n, privs, peers_aff, secrets, status);
}

int ecamd_multi_xdh_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *k, const uint8_t *u, uint8_t *out, uint8_t *status)
{
	(void)m;
	note(n);
	return orc_xdh_batch(&c->c, (uint32_t)c->c.clen, n, k, u, out, status);
}

int ecamd_multi_eddsa_encode_point_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *points_prj, uint8_t *enc, uint8_t *status);
int ecamd_multi_eddsa_verify_msg_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					   const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result)
{
	/* encode, write the encodings into a copy of the slots, hash, verify; an item whose key has no encoding is rejected */
	const int e448 = c->c.clen == 56;
	const uint32_t kl = e448 ? 57 : 32;
	uint8_t *enc = malloc((size_t)n * kl + 1), *st = malloc(n + 1), *sl = malloc((size_t)n * stride + 1), *dg = malloc((size_t)n * 114 + 1);
	uint32_t i, dl = 0;
	int r;
	mock_ready(n);
	if (!enc || !st || !sl || !dg) {
		free(enc); free(st); free(sl); free(dg);
		return mfail("mock: out of memory");
	}
	memcpy(sl, hash_slots, (size_t)n * stride);
	r = ecamd_multi_eddsa_encode_point_batch(m, c, n, keys_prj, enc, st);
	for (i = 0; i < n && !r; i++) {
		if (!st[i]) {
			memcpy(sl + (size_t)i * stride + 4 + a_offset, enc + (size_t)i * kl, kl);
		}
	}
	r = r || (mock_hash_slots(e448 ? 5 : 4, n, sl, stride, dg, &dl) ? mfail("mock: hashing failed") : 0);
	r = r || ecamd_multi_eddsa_verify_batch(m, c, n, enc, sigs, dg, dl, result);
	for (i = 0; i < n && !r; i++) {
		if (st[i]) {
			result[i] = 1;
		}
	}
	free(enc); free(st); free(sl); free(dg);
	return r;
}

/* the whole-batch bit from the same inputs (round 6): the exact conjunction of the item results */
int ecamd_multi_eddsa_verify_msg_prj_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					       const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, int *all_valid)
{
	uint8_t *res = malloc(n + 1);
	uint32_t i;
	int r;
	*all_valid = 0;
	if (!res) {
		return mfail("mock: out of memory");
	}
	r = ecamd_multi_eddsa_verify_msg_prj_batch(m, c, n, keys_prj, sigs, hash_slots, stride, a_offset, res);
	if (!r) {
		*all_valid = 1;
		for (i = 0; i < n; i++) {
			if (res[i]) {
				*all_valid = 0;
			}
		}
	}
	free(res);
	return r;
}

int ecamd_multi_eddsa_verify_ph_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					  const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
					  uint8_t *result)
{
	/* PH(M) = SHA-512 of every message into the second blank of a copy of the slots, then the plain form */
	const int e448 = c->c.clen == 56;
	uint8_t *sl = malloc((size_t)n * stride + 1), *dg = malloc((size_t)n * 64 + 1);
	uint32_t i, dl = 64;
	int r;
	mock_ready(n);
	if (!sl || !dg || mock_hash_slots(e448 ? 5 : 4, n, msg_slots, msg_stride, dg, &dl) || dl != 64) {
		free(sl); free(dg);
		return mfail("mock: pre-hashing failed");
	}
	memcpy(sl, hash_slots, (size_t)n * stride);
	for (i = 0; i < n; i++) {
		memcpy(sl + (size_t)i * stride + 4 + a_offset + (e448 ? 57 : 32), dg + (size_t)i * 64, 64);
	}
	r = ecamd_multi_eddsa_verify_msg_prj_batch(m, c, n, keys_prj, sigs, sl, stride, a_offset, result);
	free(sl); free(dg);
	return r;
}

int ecamd_multi_eddsa_verify_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				   const uint8_t *hram, uint32_t hram_len, uint8_t *result)
{
	mock_ready(n);
	(void)m;
	note(n);
	return c->c.clen == 56 ? orc_eddsa448_verify_batch(&c->c, n, pubkeys, sigs, hram, hram_len, result)
			       : orc_eddsa25519_verify_batch(&c->c, n, pubkeys, sigs, hram, hram_len, result);
}

int ecamd_multi_eddsa_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				       const uint8_t *hram, uint32_t hram_len, int *all_valid, uint32_t *first_rejected)
{
	uint8_t *res = malloc(n + 1);
	uint32_t i;
	int r = ecamd_multi_eddsa_verify_batch(m, c, n, pubkeys, sigs, hram, hram_len, res);
	*all_valid = 1;
	if (first_rejected) {
		*first_rejected = n;
	}
	for (i = 0; i < n && !r; i++) {
		if (res[i]) {
			*all_valid = 0;
			if (first_rejected && *first_rejected == n) {
				*first_rejected = i;
			}
		}
	}
	free(res);
	return r;
}

/* the stand-in serves every curve whose order equals its generator's (cofactor 1), as the product does */
const ecamd_curve *ecamd_multi_curve_handle(const ecamd_mcurve *c, int rank) { (void)rank; return (const ecamd_curve *)c; }
int ec_schnorr_verify_all_available(const ecamd_curve *curve, int r_fmt)
{
	const ecamd_mcurve *c = (const ecamd_mcurve *)curve;
	(void)r_fmt;
	return (c && c->cof1) ? 1 : 0;
}

/* the Schnorr-type whole-batch predicate as the exact conjunction of the item form ([s]G + [ne]Y = R; r_fmt 1: same x and an even y) */
int ecamd_multi_schnorr_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *s, const uint8_t *ne, const uint8_t *keys_aff,
					 const uint8_t *r, int r_fmt, int *all_valid)
{
	const u32 cl = (u32)c->c.clen, ql = (u32)c->c.qlen;
	uint8_t *A = malloc((size_t)n * 2 * cl + 1), *B = malloc((size_t)n * 2 * cl + 1), *W = malloc((size_t)n * 2 * cl + 1);
	uint8_t *sa = malloc(n + 1), *sb = malloc(n + 1), *sw = malloc(n + 1);
	uint32_t i;
	int rc;
	(void)m;
	note(n);
	*all_valid = 0;
	if (!A || !B || !W || !sa || !sb || !sw) {
		free(A); free(B); free(W); free(sa); free(sb); free(sw);
		return mfail("mock: out of memory");
	}
	rc = orc_scalar_mult_batch(&c->c, n, s, ql, NULL, A, sa) || orc_scalar_mult_batch(&c->c, n, ne, ql, keys_aff, B, sb) ||
	     orc_pt_add_batch(&c->c, n, A, B, W, sw, 0);
	if (!rc) {
		int all = 1;
		for (i = 0; i < n && all; i++) {
			const uint8_t *w = W + (size_t)i * 2 * cl;
			if (sa[i] == 1 || sb[i] == 1) {
				all = 0;
			} else if (sa[i] == 2 || sb[i] == 2) {
				w = sa[i] == 2 ? B + (size_t)i * 2 * cl : A + (size_t)i * 2 * cl;
				all = !(sa[i] == 2 && sb[i] == 2);
			} else {
				all = sw[i] == 0;
			}
			if (all) {
				all = r_fmt ? (!memcmp(w, r + (size_t)i * cl, cl) && !(w[2 * cl - 1] & 1)) : !memcmp(w, r + (size_t)i * 2 * cl, (size_t)2 * cl);
			}
		}
		*all_valid = all;
	}
	free(A); free(B); free(W); free(sa); free(sb); free(sw);
	return rc;
}

/* the one-call form from keys, signatures and hash inputs (round 6): normalise the keys, write their x into the blanks, hash with libecc's
 * hash functions, e mod q and q - e with libecc's nn, the even-y representative for r_fmt 1, then the conjunction above */
int ecamd_multi_schnorr_verify_msg_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys, int key_fmt, const uint8_t *sigs,
					     int r_fmt, int hash_type, const uint8_t *hash_slots, uint32_t stride, uint32_t x_offset, int *all_valid)
{
	const u32 cl = (u32)c->c.clen, ql = (u32)c->c.qlen, rl = r_fmt ? cl : 2 * cl;
	uint8_t *aff = malloc((size_t)n * 2 * cl + 1), *st = malloc(n + 1), *sl = malloc((size_t)n * stride + 1), *dg = malloc((size_t)n * 64 + 1);
	uint8_t *sv = malloc((size_t)n * ql + 1), *ne = malloc((size_t)n * ql + 1), *rr = malloc((size_t)n * rl + 1);
	uint8_t pb[80], qb[80];
	uint32_t i, k, dl = 0;
	nn q, e;
	int rc = -1, bad = 0;
	*all_valid = 0;
	mock_ready(n);
	q.magic = e.magic = WORD(0);
	if (!aff || !st || !sl || !dg || !sv || !ne || !rr) {
		goto out;
	}
	if (ecamd_multi_prj_pt_unique_batch(m, c, n, keys, key_fmt, aff, ECAMD_PT_AFFINE, st)) {
		goto out;
	}
	memcpy(sl, hash_slots, (size_t)n * stride);
	for (k = 0; k < cl; k++) {   /* p and q big-endian from the oracle's little-endian 64-bit limbs */
		pb[cl - 1 - k] = (uint8_t)(c->c.fp.p[k / 8] >> (8 * (k % 8)));
	}
	for (k = 0; k < ql; k++) {
		qb[ql - 1 - k] = (uint8_t)(c->c.q[k / 8] >> (8 * (k % 8)));
	}
	if (nn_init_from_buf(&q, qb, (u16)ql)) {
		goto out;
	}
	for (i = 0; i < n; i++) {
		uint8_t *Y = aff + (size_t)i * 2 * cl;
		bad |= st[i] != 0;
		if (x_offset != 0xffffffffu) {
			memcpy(sl + (size_t)i * stride + 4 + x_offset, Y, cl);
		}
		if (r_fmt && (Y[2 * cl - 1] & 1)) {
			int borrow = 0;
			for (k = cl; k-- > 0;) {
				const int d = (int)pb[k] - (int)Y[cl + k] - borrow;
				Y[cl + k] = (uint8_t)(d & 0xff);
				borrow = d < 0;
			}
		}
		memcpy(rr + (size_t)i * rl, sigs + (size_t)i * (rl + ql), rl);
		memcpy(sv + (size_t)i * ql, sigs + (size_t)i * (rl + ql) + rl, ql);
	}
	if (mock_hash_slots(hash_type, n, sl, stride, dg, &dl)) {
		mfail("mock: hashing failed");
		goto out;
	}
	for (i = 0; i < n; i++) {
		if (nn_init_from_buf(&e, dg + (size_t)i * dl, (u16)dl) || nn_mod(&e, &e, &q) || nn_mod_neg(&e, &e, &q) || nn_export_to_buf(ne + (size_t)i * ql, (u16)ql, &e)) {
			goto out;
		}
	}
	rc = bad ? 0 : ecamd_multi_schnorr_verify_all_batch(m, c, n, sv, ne, aff, rr, r_fmt, all_valid);
out:
	nn_uninit(&q); nn_uninit(&e);
	free(aff); free(st); free(sl); free(dg); free(sv); free(ne); free(rr);
	return rc;
}

/* eddsa_export_pub_key (sig/eddsa.c:970) on each point, through libecc itself */
int ecamd_multi_eddsa_encode_point_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *points_prj, uint8_t *enc,
					 uint8_t *status)
{
	const u32 cl = (u32)c->c.clen, kl = (cl == 56) ? 57 : 32;
	uint32_t i;
	(void)m;
	note(n);
	if (!c->have_params) {
		return mfail("mock: encode needs a named curve");
	}
	for (i = 0; i < n; i++) {
		ec_pub_key pk;
		memset(&pk, 0, sizeof(pk));
		status[i] = 1;
		memset(enc + (size_t)i * kl, 0, kl);
		if (prj_pt_import_from_buf(&pk.y, points_prj + (size_t)i * 3 * cl, (u16)(3 * cl), &c->params.ec_curve)) {
			continue;
		}
		pk.key_type = (cl == 56) ? EDDSA448 : EDDSA25519;
		pk.params = &c->params;
		pk.magic = PUB_KEY_MAGIC;
		if (!eddsa_export_pub_key(&pk, enc + (size_t)i * kl, (u16)kl)) {
			status[i] = 0;
		}
	}
	return 0;
}

int ecamd_multi_eddsa_sign_R_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc, uint8_t *status)
{
	(void)m;
	note(n);
	if (c->c.clen != 32) {
		return mfail("mock: Ed448 signing is not modelled");
	}
	return orc_eddsa25519_sign_R_batch(&c->c, n, r_hash, R_enc, status);
}

int ecamd_multi_eddsa_sign_S_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *r_hash, const uint8_t *hram,
				   const uint8_t *a_scalars, uint8_t *S_out)
{
	(void)m;
	note(n);
	if (c->c.clen != 32) {
		return mfail("mock: Ed448 signing is not modelled");
	}
	return orc_eddsa25519_sign_S_batch(&c->c, n, r_hash, hram, a_scalars, S_out);
}
