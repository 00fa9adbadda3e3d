/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin batch driver over the UNMODIFIED reference library (compiled from
 * /root/reference/src by oracle/Makefile into oracle/_ref/libecc_ref.so).  It is the
 * ground truth the plain-C restatement (ecc_oracle.c) and the HIP path are pinned
 * against, and the "reference" CPU baseline timed by bench.py.  Nothing under
 * libecc_amd/ may link or load this file.
 *
 * It supplies the three symbols the reference library leaves undefined
 * (get_random / get_unsafe_random: external_deps/rand.c:76,128 -- here a seeded,
 * thread-local splitmix64 so runs are reproducible; ext_printf comes from the
 * reference's external_deps/print.c) and exposes plain C-ABI loops over the
 * reference API:
 *   prj_pt_import_from_aff_buf (curves/prj_pt.c:511) -> prj_pt_mul (:1759)
 *   -> prj_pt_unique (:241) -> prj_pt_export_to_aff_buf (:600)
 *   ec_verify (sig/sig_algs.c:655), _ec_sign (:473), ecccdh_derive_secret (ecdh/ecccdh.c:167)
 *   nn_mul_redc1 (nn/nn_mul_redc1.c:246), prj_pt_add (:1204), prj_pt_dbl (:1132)
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <pthread.h>
#include "libsig.h"

/* ---- randomness hooks the reference imports (seeded, thread-local) ---- */
static __thread uint64_t tl_seed = 0x9E3779B97F4A7C15ULL;
static uint64_t splitmix64(void)
{
	uint64_t z = (tl_seed += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
/* replay mode (refdrv_random_mod_batch): the next get_random calls deliver these bytes instead of the generator's */
static __thread const unsigned char *tl_replay;
static __thread unsigned int tl_replay_left;
int get_random(unsigned char *buf, u16 len)
{
	u16 i;
	if (tl_replay) {
		if (len > tl_replay_left) {
			return -1;
		}
		memcpy(buf, tl_replay, len);
		tl_replay += len;
		tl_replay_left -= len;
		return 0;
	}
	for (i = 0; i < len; i++) {
		buf[i] = (unsigned char)(splitmix64() >> 24);
	}
	return 0;
}
int get_unsafe_random(unsigned char *buf, u16 len) { return get_random(buf, len); }
void refdrv_seed(uint64_t s) { tl_seed = s; }

/* ---- helpers ---- */
static int load_params(const char *curve, ec_params *params)
{
	const ec_str_params *sp = NULL;
	int ret;
	ret = ec_get_curve_params_by_name((const u8 *)curve, (u8)(strlen(curve) + 1), &sp);
	if (ret || sp == NULL) {
		return -1;
	}
	return import_params(params, sp);
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int refdrv_coord_len(const char *curve)
{
	ec_params params;
	if (load_params(curve, &params)) {
		return -1;
	}
	return (int)BYTECEIL(params.ec_fp.p_bitlen);
}

int refdrv_order_len(const char *curve)
{
	ec_params params;
	if (load_params(curve, &params)) {
		return -1;
	}
	return (int)BYTECEIL(params.ec_gen_order_bitlen);
}

/* ---- batched scalar multiplication through the public API ---- */
typedef struct {
	const ec_params *params;
	uint32_t lo, hi, slen, clen;
	const uint8_t *scalars, *points;
	uint8_t *out, *status;
	double t_mul; /* seconds spent inside prj_pt_mul + prj_pt_unique only */
} smul_job;

static void *smul_worker(void *arg)
{
	smul_job *j = (smul_job *)arg;
	uint32_t i;
	const uint32_t plen = 2 * j->clen;
	double acc = 0.0;
	refdrv_seed(0x5EC9256ULL + j->lo);
	for (i = j->lo; i < j->hi; i++) {
		prj_pt P, Q;
		nn m;
		int ret, iszero = 0;
		double t0;
		P.magic = Q.magic = WORD(0);
		m.magic = WORD(0);
		j->status[i] = 1;
		memset(j->out + (size_t)i * plen, 0, plen);
		ret = nn_init_from_buf(&m, j->scalars + (size_t)i * j->slen, (u16)j->slen);
		if (ret) {
			continue;
		}
		if (j->points) {
			ret = prj_pt_import_from_aff_buf(&P, j->points + (size_t)i * plen, (u16)plen,
							 &j->params->ec_curve);
		} else {
			ret = prj_pt_copy(&P, &j->params->ec_gen);
		}
		if (ret) {
			continue;
		}
		t0 = now_s();
		ret = prj_pt_mul(&Q, &m, &P);
		if (!ret) {
			ret = prj_pt_iszero(&Q, &iszero);
		}
		if (!ret && !iszero) {
			ret = prj_pt_unique(&Q, &Q);
		}
		acc += now_s() - t0;
		if (ret) {
			continue;
		}
		if (iszero) {
			j->status[i] = 2;
			continue;
		}
		ret = prj_pt_export_to_aff_buf(&Q, j->out + (size_t)i * plen, plen);
		j->status[i] = ret ? 1 : 0;
	}
	j->t_mul = acc;
	return NULL;
}

/*
 * status[i]: 0 = ok (out = affine X||Y big-endian), 1 = error (the reference returned -1:
 * coordinate >= p, off-curve, ...), 2 = result is the point at infinity (prj_pt_mul
 * returned 0 with Z = 0; prj_pt_unique would return -1).
 * *elapsed = wall seconds of the whole loop; *mul_seconds = max over threads of the time
 * spent inside prj_pt_mul + prj_pt_unique.
 */
int refdrv_scalar_mult_batch(const char *curve, uint32_t n, const uint8_t *scalars, uint32_t slen,
			     const uint8_t *points, uint8_t *out, uint8_t *status, int nthreads,
			     double *elapsed, double *mul_seconds)
{
	ec_params params;
	pthread_t th[256];
	smul_job jobs[256];
	int t;
	double t0, mx = 0.0;
	if (load_params(curve, &params)) {
		return -1;
	}
	if (nthreads < 1) {
		nthreads = 1;
	}
	if (nthreads > 256) {
		nthreads = 256;
	}
	t0 = now_s();
	for (t = 0; t < nthreads; t++) {
		jobs[t].params = &params;
		jobs[t].lo = (uint32_t)(((uint64_t)n * (uint64_t)t) / (uint64_t)nthreads);
		jobs[t].hi = (uint32_t)(((uint64_t)n * (uint64_t)(t + 1)) / (uint64_t)nthreads);
		jobs[t].slen = slen;
		jobs[t].clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
		jobs[t].scalars = scalars;
		jobs[t].points = points;
		jobs[t].out = out;
		jobs[t].status = status;
		jobs[t].t_mul = 0.0;
		if (nthreads == 1) {
			smul_worker(&jobs[t]);
		} else if (pthread_create(&th[t], NULL, smul_worker, &jobs[t])) {
			return -1;
		}
	}
	for (t = 0; t < nthreads; t++) {
		if (nthreads > 1) {
			pthread_join(th[t], NULL);
		}
		if (jobs[t].t_mul > mx) {
			mx = jobs[t].t_mul;
		}
	}
	if (elapsed) {
		*elapsed = now_s() - t0;
	}
	if (mul_seconds) {
		*mul_seconds = mx;
	}
	return 0;
}

/* ---- point addition / doubling (affine in, affine out) ---- */
int refdrv_pt_add_batch(const char *curve, uint32_t n, const uint8_t *p1, const uint8_t *p2,
			uint8_t *out, uint8_t *status, int dbl)
{
	ec_params params;
	uint32_t i, clen, plen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	plen = 2 * clen;
	for (i = 0; i < n; i++) {
		prj_pt A, B, C;
		int ret, iszero = 0;
		A.magic = B.magic = C.magic = WORD(0);
		status[i] = 1;
		memset(out + (size_t)i * plen, 0, plen);
		ret = prj_pt_import_from_aff_buf(&A, p1 + (size_t)i * plen, (u16)plen, &params.ec_curve);
		if (ret) {
			continue;
		}
		if (dbl) {
			ret = prj_pt_dbl(&C, &A);
		} else {
			ret = prj_pt_import_from_aff_buf(&B, p2 + (size_t)i * plen, (u16)plen,
							 &params.ec_curve);
			if (ret) {
				continue;
			}
			ret = prj_pt_add(&C, &A, &B);
		}
		if (ret) {
			continue;
		}
		ret = prj_pt_iszero(&C, &iszero);
		if (ret) {
			continue;
		}
		if (iszero) {
			status[i] = 2;
			continue;
		}
		ret = prj_pt_unique(&C, &C);
		if (ret) {
			continue;
		}
		ret = prj_pt_export_to_aff_buf(&C, out + (size_t)i * plen, plen);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- the group law, the on-curve test and _prj_pt_unprotected_mult of the unmodified reference in either wire format
 *      (in_fmt / out_fmt: 0 affine X || Y, 1 projective X || Y || Z); status 0 ok / 1 the call chain returned -1 / 2 infinity ---- */
static int ref_import_fmt(prj_pt *P, const uint8_t *src, int fmt, uint32_t clen, ec_shortw_crv_src_t crv)
{
	return fmt ? prj_pt_import_from_buf(P, src, (u16)(3 * clen), crv) : prj_pt_import_from_aff_buf(P, src, (u16)(2 * clen), crv);
}
static void ref_export_fmt(uint8_t *dst, uint8_t *status, prj_pt *Q, int fmt, uint32_t clen)
{
	int iszero = 0, ret;
	*status = 1;
	if (prj_pt_iszero(Q, &iszero)) {
		return;
	}
	if (iszero) {
		*status = 2;
		return;
	}
	if (prj_pt_unique(Q, Q)) {
		return;
	}
	ret = fmt ? prj_pt_export_to_buf(Q, dst, 3 * clen) : prj_pt_export_to_aff_buf(Q, dst, 2 * clen);
	*status = ret ? 1 : 0;
}
int refdrv_pt_op_batch_fmt(const char *curve, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2, int in_fmt, uint8_t *out,
			   int out_fmt, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen, iw, ow;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	iw = (in_fmt ? 3u : 2u) * clen;
	ow = op >= 4 ? 1u : (out_fmt ? 3u : 2u) * clen;
	for (i = 0; i < n; i++) {
		prj_pt A, B, C;
		int ret, on = 0;
		A.magic = B.magic = C.magic = WORD(0);
		status[i] = 1;
		if (op >= 4) {
			/* prj_pt_cmp / prj_pt_eq_or_opp of the unmodified reference: one predicate byte per item */
			int v = 0;
			out[i] = 0;
			if (ref_import_fmt(&A, p1 + (size_t)i * iw, in_fmt, clen, &params.ec_curve) ||
			    ref_import_fmt(&B, p2 + (size_t)i * iw, in_fmt, clen, &params.ec_curve)) {
				continue;
			}
			B.crv = A.crv;
			if (op == 4 ? prj_pt_cmp(&A, &B, &v) : prj_pt_eq_or_opp(&A, &B, &v)) {
				continue;
			}
			out[i] = (uint8_t)(v != 0);
			status[i] = 0;
			continue;
		}
		if (op == 2) {
			/* prj_pt_is_on_curve of a triple whose coordinates are in range (the import tests the equation itself) */
			status[i] = ref_import_fmt(&A, p1 + (size_t)i * iw, in_fmt, clen, &params.ec_curve) ? 1 : 0;
			if (!status[i] && (prj_pt_is_on_curve(&A, &on) || !on)) {
				status[i] = 1;
			}
			continue;
		}
		memset(out + (size_t)i * ow, 0, ow);
		if (ref_import_fmt(&A, p1 + (size_t)i * iw, in_fmt, clen, &params.ec_curve)) {
			continue;
		}
		if (op == 3) {
			ret = prj_pt_neg(&C, &A);
		} else if (op == 1) {
			ret = prj_pt_dbl(&C, &A);
		} else {
			if (ref_import_fmt(&B, p2 + (size_t)i * iw, in_fmt, clen, &params.ec_curve)) {
				continue;
			}
			ret = prj_pt_add(&C, &A, &B);
		}
		if (ret) {
			continue;
		}
		ref_export_fmt(out + (size_t)i * ow, status + i, &C, out_fmt, clen);
	}
	return 0;
}
int refdrv_unprotected_mult_batch(const char *curve, uint32_t n, const uint8_t *scalars, uint32_t slen, uint32_t sstride,
				  const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen, iw, ow;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	iw = (in_fmt ? 3u : 2u) * clen;
	ow = (out_fmt ? 3u : 2u) * clen;
	for (i = 0; i < n; i++) {
		prj_pt P, Q;
		nn m;
		P.magic = Q.magic = WORD(0);
		m.magic = WORD(0);
		status[i] = 1;
		memset(out + (size_t)i * ow, 0, ow);
		if (ref_import_fmt(&P, points + (size_t)i * iw, in_fmt, clen, &params.ec_curve) ||
		    nn_init_from_buf(&m, scalars + (size_t)i * sstride, (u16)slen) || _prj_pt_unprotected_mult(&Q, &m, &P)) {
			continue;
		}
		ref_export_fmt(out + (size_t)i * ow, status + i, &Q, out_fmt, clen);
	}
	return 0;
}

/* ---- field level: nn_mul_redc1 / plain fp ops on little-endian u64 limbs ---- */
/* op: 0 = fp_mul_monty (nn_mul_redc1), 1 = fp_add, 2 = fp_sub, 3 = fp_mul (plain), 4 = fp_inv (b ignored) */
int refdrv_fp_op_batch(const char *curve, int op, uint32_t n, uint32_t nlimbs, const uint64_t *a,
		       const uint64_t *b, uint64_t *out)
{
	ec_params params;
	uint32_t i, k;
	if (load_params(curve, &params)) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		fp x, y, z;
		nn t;
		int ret;
		x.magic = y.magic = z.magic = WORD(0);
		t.magic = WORD(0);
		ret = fp_init(&x, &params.ec_fp);
		ret |= fp_init(&y, &params.ec_fp);
		ret |= fp_init(&z, &params.ec_fp);
		ret |= nn_init(&t, (u16)(nlimbs * 8));
		if (ret) {
			return -1;
		}
		for (k = 0; k < nlimbs; k++) {
			t.val[k] = a[(size_t)i * nlimbs + k];
		}
		ret = fp_set_nn(&x, &t);
		for (k = 0; k < nlimbs; k++) {
			t.val[k] = b[(size_t)i * nlimbs + k];
		}
		ret |= fp_set_nn(&y, &t);
		if (ret) {
			return -1;
		}
		switch (op) {
		case 0: ret = fp_mul_monty(&z, &x, &y); break;
		case 1: ret = fp_add(&z, &x, &y); break;
		case 2: ret = fp_sub(&z, &x, &y); break;
		case 3: ret = fp_mul(&z, &x, &y); break;
		case 4: ret = fp_inv(&z, &x); break;
		default: ret = -1;
		}
		if (ret) {
			return -1;
		}
		for (k = 0; k < nlimbs; k++) {
			out[(size_t)i * nlimbs + k] = (k < z.fp_val.wlen) ? z.fp_val.val[k] : 0;
		}
	}
	return 0;
}

/* ---- ECDSA verify / sign and ECC-CDH through the protocol API ---- */
/* result[i] = 0 accept, 1 reject (ec_verify returned -1, incl. import failure) */
int refdrv_ecdsa_verify_batch(const char *curve, int hash_type, uint32_t n, const uint8_t *pubs_aff,
			      const uint8_t *sigs, const uint8_t *msgs, uint32_t msg_len,
			      uint8_t *result)
{
	ec_params params;
	uint32_t i, clen, qlen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	qlen = (uint32_t)BYTECEIL(params.ec_gen_order_bitlen);
	for (i = 0; i < n; i++) {
		ec_pub_key pub;
		int ret;
		result[i] = 1;
		ret = ec_pub_key_import_from_aff_buf(&pub, &params, pubs_aff + (size_t)i * 2 * clen,
						     (u8)(2 * clen), ECDSA);
		if (ret) {
			continue;
		}
		ret = ec_verify(sigs + (size_t)i * 2 * qlen, (u8)(2 * qlen), &pub,
				msgs + (size_t)i * msg_len, msg_len, ECDSA, (hash_alg_type)hash_type,
				NULL, 0);
		result[i] = ret ? 1 : 0;
	}
	return 0;
}

static __thread const uint8_t *tl_nonce;
static __thread uint32_t tl_nonce_len;
static int fixed_nonce(nn_t out, nn_src_t q)
{
	int ret, cmp;
	ret = nn_init_from_buf(out, tl_nonce, (u16)tl_nonce_len);
	if (ret) {
		return ret;
	}
	ret = nn_cmp(out, q, &cmp);
	if (ret) {
		return ret;
	}
	return (cmp >= 0) ? -1 : 0;
}

/* Sign with a caller-supplied nonce k (as the reference's KAT harness does through the
 * same callback pointer, tests/ec_self_tests_core.c:731-1012). status 0 ok / 1 error. */
int refdrv_ecdsa_sign_batch(const char *curve, int hash_type, uint32_t n, const uint8_t *privs,
			    const uint8_t *nonces, const uint8_t *msgs, uint32_t msg_len,
			    uint8_t *sigs, uint8_t *pubs_aff, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen, qlen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	qlen = (uint32_t)BYTECEIL(params.ec_gen_order_bitlen);
	for (i = 0; i < n; i++) {
		ec_key_pair kp;
		int ret;
		status[i] = 1;
		ret = ec_key_pair_import_from_priv_key_buf(&kp, &params, privs + (size_t)i * qlen,
							   (u8)qlen, ECDSA);
		if (ret) {
			continue;
		}
		if (pubs_aff) {
			ret = ec_pub_key_export_to_aff_buf(&kp.pub_key, pubs_aff + (size_t)i * 2 * clen,
							   (u8)(2 * clen));
			if (ret) {
				continue;
			}
		}
		tl_nonce = nonces + (size_t)i * qlen;
		tl_nonce_len = qlen;
		ret = _ec_sign(sigs + (size_t)i * 2 * qlen, (u8)(2 * qlen), &kp,
			       msgs + (size_t)i * msg_len, msg_len, fixed_nonce, ECDSA,
			       (hash_alg_type)hash_type, NULL, 0);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ECC-CDH: shared secret = x(d * Q_peer); status 0 ok / 1 error (-1 from the reference) */
int refdrv_ecccdh_batch(const char *curve, uint32_t n, const uint8_t *privs, const uint8_t *peers_aff,
			uint8_t *secrets, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen, qlen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	qlen = (uint32_t)BYTECEIL(params.ec_gen_order_bitlen);
	for (i = 0; i < n; i++) {
		ec_priv_key priv;
		int ret;
		status[i] = 1;
		memset(secrets + (size_t)i * clen, 0, clen);
		ret = ec_priv_key_import_from_buf(&priv, &params, privs + (size_t)i * qlen, (u8)qlen,
						  ECCCDH);
		if (ret) {
			continue;
		}
		ret = ecccdh_derive_secret(&priv, peers_aff + (size_t)i * 2 * clen, (u8)(2 * clen),
					   secrets + (size_t)i * clen, (u8)clen);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- domain-parameter export (used once by tools/gen_curve_table.py) ---- */
static int put_str_param(const ec_str_param *sp, uint8_t *dst, uint32_t cap)
{
	if (sp == NULL || sp->buflen > cap) {
		return -1;
	}
	memcpy(dst, sp->buf, sp->buflen);
	return (int)sp->buflen;
}

/*
 * Dump the big-endian byte strings of curve number 'type' (ec_curve_type, 1..EC_CURVES_NUM).
 * fields, each written to out + k*256 with its length in lens[k]:
 *   0 name, 1 p, 2 a, 3 b, 4 curve_order, 5 gx, 6 gy, 7 gen_order, 8 cofactor
 * returns 0, or -1 if the type is not built in.
 */
int refdrv_curve_params(int type, uint8_t *out, int32_t *lens)
{
	const ec_str_params *sp = NULL;
	const ec_str_param *f[9];
	int k;
	if (ec_get_curve_params_by_type((ec_curve_type)type, &sp) || sp == NULL) {
		return -1;
	}
	f[0] = sp->name; f[1] = sp->p; f[2] = sp->a; f[3] = sp->b; f[4] = sp->curve_order;
	f[5] = sp->gx; f[6] = sp->gy; f[7] = sp->gen_order; f[8] = sp->cofactor;
	for (k = 0; k < 9; k++) {
		lens[k] = put_str_param(f[k], out + k * 256, 256);
		if (lens[k] < 0) {
			return -1;
		}
	}
	return 0;
}

/* ---- X25519 / X448 through the reference (ecdh/x25519_448.c:380,403) ---- */
#include "ecdh/x25519_448.h"
int refdrv_xdh_batch(uint32_t len, uint32_t n, const uint8_t *k, const uint8_t *u, uint8_t *out, uint8_t *status)
{
	uint32_t i;
	for (i = 0; i < n; i++) {
		int ret = -1;
		memset(out + (size_t)i * len, 0, len);
		if (len == 32) {
			ret = x25519(k + (size_t)i * len, u + (size_t)i * len, out + (size_t)i * len);
		} else if (len == 56) {
			ret = x448(k + (size_t)i * len, u + (size_t)i * len, out + (size_t)i * len);
		}
		status[i] = ret ? 1 : 0;
		if (ret) {
			memset(out + (size_t)i * len, 0, len);
		}
	}
	return 0;
}

/* ---- Ed25519 through the protocol API (sig/eddsa.c) ---- */
/* keys from 32-byte seeds, signatures over msgs; pubs: n x 32 (RFC 8032 encoding), sigs: n x 64 */
int refdrv_eddsa25519_sign_batch(uint32_t n, const uint8_t *seeds, const uint8_t *msgs, uint32_t msg_len,
				 uint8_t *pubs, uint8_t *sigs, uint8_t *status)
{
	ec_params params;
	uint32_t i;
	if (load_params("WEI25519", &params)) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		ec_key_pair kp;
		int ret;
		status[i] = 1;
		ret = eddsa_import_key_pair_from_priv_key_buf(&kp, seeds + (size_t)i * 32, 32, &params, EDDSA25519);
		if (ret) {
			continue;
		}
		ret = eddsa_export_pub_key(&kp.pub_key, pubs + (size_t)i * 32, 32);
		if (ret) {
			continue;
		}
		ret = ec_sign(sigs + (size_t)i * 64, 64, &kp, msgs + (size_t)i * msg_len, msg_len, EDDSA25519,
			      SHA512, NULL, 0);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* result[i] = 0 accept, 1 reject (eddsa_import_pub_key or ec_verify returned -1) */
int refdrv_eddsa25519_verify_batch(uint32_t n, const uint8_t *pubs, const uint8_t *sigs, const uint8_t *msgs,
				   uint32_t msg_len, uint8_t *result)
{
	ec_params params;
	uint32_t i;
	if (load_params("WEI25519", &params)) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		ec_pub_key pub;
		int ret;
		result[i] = 1;
		ret = eddsa_import_pub_key(&pub, pubs + (size_t)i * 32, 32, &params, EDDSA25519);
		if (ret) {
			continue;
		}
		ret = ec_verify(sigs + (size_t)i * 64, 64, &pub, msgs + (size_t)i * msg_len, msg_len, EDDSA25519,
				SHA512, NULL, 0);
		result[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- projective wire format: prj_pt_import_from_buf -> [prj_pt_mul] -> prj_pt_unique -> prj_pt_export_to_buf,
 * the chain of `ec_utils scalar_mult` (tests/ec_utils.c:1380-1538).  scalars == NULL: normalisation only.
 * points: n x 3*clen (X || Y || Z), out: n x 3*clen unique representative (Z = 1); status 0 / 1 error / 2 infinity */
int refdrv_prj_batch(const char *curve, uint32_t n, const uint8_t *scalars, uint32_t slen, const uint8_t *points,
		     uint8_t *out, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	refdrv_seed(0x5EC9256ULL);
	for (i = 0; i < n; i++) {
		prj_pt P, Q;
		nn m;
		int ret, iszero = 0;
		P.magic = Q.magic = WORD(0);
		m.magic = WORD(0);
		status[i] = 1;
		memset(out + (size_t)i * 3 * clen, 0, 3 * clen);
		ret = prj_pt_import_from_buf(&P, points + (size_t)i * 3 * clen, (u16)(3 * clen), &params.ec_curve);
		if (ret) {
			continue;
		}
		if (scalars) {
			ret = nn_init_from_buf(&m, scalars + (size_t)i * slen, (u16)slen);
			if (ret) {
				continue;
			}
			ret = prj_pt_mul(&Q, &m, &P);
		} else {
			ret = prj_pt_copy(&Q, &P);
		}
		if (ret) {
			continue;
		}
		ret = prj_pt_iszero(&Q, &iszero);
		if (ret) {
			continue;
		}
		if (iszero) {
			status[i] = 2;
			continue;
		}
		ret = prj_pt_unique(&Q, &Q);
		if (ret) {
			continue;
		}
		ret = prj_pt_export_to_buf(&Q, out + (size_t)i * 3 * clen, 3 * clen);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- Ed448 through the protocol API (sig/eddsa.c); keys from 57-byte seeds, pubs n x 57, sigs n x 114 ---- */
int refdrv_eddsa448_sign_batch(uint32_t n, const uint8_t *seeds, const uint8_t *msgs, uint32_t msg_len,
			       uint8_t *pubs, uint8_t *sigs, uint8_t *status)
{
	ec_params params;
	uint32_t i;
	if (load_params("WEI448", &params)) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		ec_key_pair kp;
		int ret;
		status[i] = 1;
		ret = eddsa_import_key_pair_from_priv_key_buf(&kp, seeds + (size_t)i * 57, 57, &params, EDDSA448);
		if (ret) {
			continue;
		}
		ret = eddsa_export_pub_key(&kp.pub_key, pubs + (size_t)i * 57, 57);
		if (ret) {
			continue;
		}
		ret = ec_sign(sigs + (size_t)i * 114, 114, &kp, msgs + (size_t)i * msg_len, msg_len, EDDSA448,
			      SHAKE256, NULL, 0);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

int refdrv_eddsa448_verify_batch(uint32_t n, const uint8_t *pubs, const uint8_t *sigs, const uint8_t *msgs,
				 uint32_t msg_len, uint8_t *result)
{
	ec_params params;
	uint32_t i;
	if (load_params("WEI448", &params)) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		ec_pub_key pub;
		int ret;
		result[i] = 1;
		ret = eddsa_import_pub_key(&pub, pubs + (size_t)i * 57, 57, &params, EDDSA448);
		if (ret) {
			continue;
		}
		ret = ec_verify(sigs + (size_t)i * 114, 114, &pub, msgs + (size_t)i * msg_len, msg_len, EDDSA448,
				SHAKE256, NULL, 0);
		result[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- structured public keys: ec_structured_pub_key_import_from_buf (sig/ec_key.c:312) -> affine X || Y ----
 * status 0 ok / 1 the import returned -1 / 2 imported, but the key is the point at infinity (no affine form) */
int refdrv_structured_pub_import_batch(const char *curve, int alg, uint32_t n, const uint8_t *keys, uint32_t klen,
				       uint8_t *out_aff, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	for (i = 0; i < n; i++) {
		ec_pub_key pub;
		int ret, iszero = 0;
		status[i] = 1;
		memset(out_aff + (size_t)i * 2 * clen, 0, 2 * clen);
		ret = ec_structured_pub_key_import_from_buf(&pub, &params, keys + (size_t)i * klen, (u8)klen, (ec_alg_type)alg);
		if (ret) {
			continue;
		}
		ret = prj_pt_iszero(&pub.y, &iszero);
		if (ret) {
			continue;
		}
		if (iszero) {
			status[i] = 2;
			continue;
		}
		ret = ec_pub_key_export_to_aff_buf(&pub, out_aff + (size_t)i * 2 * clen, (u8)(2 * clen));
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- ec_verify_batch (sig/sig_algs.c:675 -> eddsa_verify_batch, sig/eddsa.c:2904; with or without the scratch pad): one accept bit for
 * the whole batch of pure Ed25519 / Ed448 signatures.  *all_valid = 1 iff every import succeeded and ec_verify_batch returned 0. */
int refdrv_eddsa_verify_batch_all(int is448, int use_scratch, uint32_t n, const uint8_t *pubs, const uint8_t *sigs, const uint8_t *msgs,
				       uint32_t msg_len, int *all_valid)
{
	ec_params params;
	ec_pub_key *keys;
	const ec_pub_key **kp;
	const u8 **sp, **mp, **ad;
	u8 *sl;
	u16 *al;
	u32 *ml;
	uint32_t i;
	int ret = 0;
	const size_t klen = is448 ? 57 : 32, slen = is448 ? 114 : 64;
	const ec_alg_type alg = is448 ? EDDSA448 : EDDSA25519;
	*all_valid = 0;
	if (load_params(is448 ? "WEI448" : "WEI25519", &params) || n == 0) {
		return -1;
	}
	keys = calloc(n, sizeof(ec_pub_key));
	kp = calloc(n, sizeof(*kp));
	sp = calloc(n, sizeof(*sp));
	mp = calloc(n, sizeof(*mp));
	sl = calloc(n, 1);
	ml = calloc(n, sizeof(u32));
	ad = calloc(n, sizeof(*ad));   /* the batch API wants the arrays even when no context is used */
	al = calloc(n, sizeof(u16));
	if (!keys || !kp || !sp || !mp || !sl || !ml || !ad || !al) {
		return -1;
	}
	refdrv_seed(0x5EC9256ULL);
	for (i = 0; i < n && !ret; i++) {
		ret = eddsa_import_pub_key(&keys[i], pubs + (size_t)i * klen, (u16)klen, &params, alg);
		kp[i] = &keys[i];
		sp[i] = sigs + (size_t)i * slen;
		mp[i] = msgs + (size_t)i * msg_len;
		sl[i] = (u8)slen;
		ml[i] = msg_len;
	}
	if (!ret) {
		if (use_scratch) {   /* the Bos-Coster multi-scalar path, (2n+1) scratch entries */
			u32 pad_len = (u32)((2 * (size_t)n + 1) * sizeof(verify_batch_scratch_pad));
			verify_batch_scratch_pad *pad = calloc(1, pad_len);
			if (!pad) {
				return -1;
			}
			ret = ec_verify_batch(sp, sl, kp, mp, ml, n, alg, is448 ? SHAKE256 : SHA512, ad, al, pad, &pad_len);
			free(pad);
		} else {
			ret = ec_verify_batch(sp, sl, kp, mp, ml, n, alg, is448 ? SHAKE256 : SHA512, ad, al, NULL, NULL);
		}
	}
	*all_valid = ret ? 0 : 1;
	free(keys); free(kp); free(sp); free(mp); free(sl); free(ml); free(ad); free(al);
	return 0;
}

/* ---- ec_verify_batch (sig/sig_algs.c:675) for the algorithms whose keys import from affine X || Y -- BIP0340 (-> bip0340_verify_batch,
 * sig/bip0340.c:1296), ECFSDSA (-> ecfsdsa_verify_batch, sig/ecfsdsa.c:1057), also ECDSA (no batch form in libecc: -1) -- with or without
 * the scratch pad: ONE accept bit for n signatures of sig_len octets.  per_item (may be NULL): n bytes, ec_verify's verdict per item
 * (0 accept / 1 reject), computed by a loop of ec_verify over the same structures.  *all_valid = 1 iff every key imported and
 * ec_verify_batch returned 0.  Keys: ec_pub_key_import_from_aff_buf, as an application holding raw points would. ---- */
int refdrv_sig_verify_batch_all(const char *curve, int alg, int hash_type, int use_scratch, uint32_t n, const uint8_t *pubs_aff,
				const uint8_t *sigs, uint32_t sig_len, const uint8_t *msgs, uint32_t msg_len, int *all_valid, uint8_t *per_item)
{
	ec_params params;
	ec_pub_key *keys;
	const ec_pub_key **kp;
	const u8 **sp, **mp, **ad;
	u8 *sl;
	u16 *al;
	u32 *ml;
	uint32_t i, clen;
	int ret = 0;
	*all_valid = 0;
	if (load_params(curve, &params) || n == 0 || sig_len > 255) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	keys = calloc(n, sizeof(ec_pub_key));
	kp = calloc(n, sizeof(*kp));
	sp = calloc(n, sizeof(*sp));
	mp = calloc(n, sizeof(*mp));
	sl = calloc(n, 1);
	ml = calloc(n, sizeof(u32));
	ad = calloc(n, sizeof(*ad));
	al = calloc(n, sizeof(u16));
	if (!keys || !kp || !sp || !mp || !sl || !ml || !ad || !al) {
		return -1;
	}
	refdrv_seed(0x5EC9256ULL);
	for (i = 0; i < n; i++) {
		const int r = ec_pub_key_import_from_aff_buf(&keys[i], &params, pubs_aff + (size_t)i * 2 * clen, (u8)(2 * clen), (ec_alg_type)alg);
		ret = ret || r;
		kp[i] = &keys[i];
		sp[i] = sigs + (size_t)i * sig_len;
		mp[i] = msgs + (size_t)i * msg_len;
		sl[i] = (u8)sig_len;
		ml[i] = msg_len;
		if (per_item) {
			per_item[i] = (r || ec_verify(sp[i], sl[i], kp[i], mp[i], ml[i], (ec_alg_type)alg, (hash_alg_type)hash_type, NULL, 0)) ? 1 : 0;
		}
	}
	if (!ret) {
		if (use_scratch) {
			u32 pad_len = (u32)((2 * (size_t)n + 1) * sizeof(verify_batch_scratch_pad));
			verify_batch_scratch_pad *pad = calloc(1, pad_len);
			if (!pad) {
				return -1;
			}
			ret = ec_verify_batch(sp, sl, kp, mp, ml, n, (ec_alg_type)alg, (hash_alg_type)hash_type, ad, al, pad, &pad_len);
			free(pad);
		} else {
			ret = ec_verify_batch(sp, sl, kp, mp, ml, n, (ec_alg_type)alg, (hash_alg_type)hash_type, ad, al, NULL, NULL);
		}
	}
	*all_valid = ret ? 0 : 1;
	free(keys); free(kp); free(sp); free(mp); free(sl); free(ml); free(ad); free(al);
	return 0;
}
int refdrv_alg_id(const char *name)
{
	if (!strcmp(name, "ECDSA")) { return (int)ECDSA; }
	if (!strcmp(name, "BIP0340")) { return (int)BIP0340; }
	if (!strcmp(name, "ECFSDSA")) { return (int)ECFSDSA; }
	if (!strcmp(name, "EDDSA25519")) { return (int)EDDSA25519; }
	return -1;
}

/* ---- aff_pt_y_from_x (curves/aff_pt.c:102): the two square roots of x^3 + a x + b in the order fp_sqrt returns them ---- */
int refdrv_y_from_x_batch(const char *curve, uint32_t n, const uint8_t *xs, uint8_t *y1, uint8_t *y2, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	for (i = 0; i < n; i++) {
		fp x, r1, r2;
		int ret;
		status[i] = 1;
		memset(y1 + (size_t)i * clen, 0, clen);
		memset(y2 + (size_t)i * clen, 0, clen);
		ret = fp_init_from_buf(&x, &params.ec_fp, xs + (size_t)i * clen, (u16)clen);
		if (ret) {
			continue;
		}
		ret = fp_init(&r1, &params.ec_fp) || fp_init(&r2, &params.ec_fp) || aff_pt_y_from_x(&r1, &r2, &x, &params.ec_curve);
		if (ret) {
			continue;
		}
		ret = fp_export_to_buf(y1 + (size_t)i * clen, (u16)clen, &r1) || fp_export_to_buf(y2 + (size_t)i * clen, (u16)clen, &r2);
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- ec_structured_key_pair_import_from_priv_key_buf (sig/ec_key.c:443) + ec_pub_key_export_to_aff_buf ----
 * status 0 ok (pub_aff = the affine public key), 1 the import failed, 2 the public key is the point at infinity */
int refdrv_structured_key_pair_batch(const char *curve, int alg, uint32_t n, const uint8_t *keys, uint32_t klen, uint8_t *priv_out,
				     uint8_t *pub_aff, uint8_t *status)
{
	ec_params params;
	uint32_t i, clen, qlen;
	if (load_params(curve, &params)) {
		return -1;
	}
	clen = (uint32_t)BYTECEIL(params.ec_fp.p_bitlen);
	qlen = (uint32_t)BYTECEIL(params.ec_gen_order_bitlen);
	for (i = 0; i < n; i++) {
		ec_key_pair kp;
		int ret, iszero = 0;
		status[i] = 1;
		memset(pub_aff + (size_t)i * 2 * clen, 0, 2 * clen);
		memset(priv_out + (size_t)i * qlen, 0, qlen);
		ret = ec_structured_key_pair_import_from_priv_key_buf(&kp, &params, keys + (size_t)i * klen, (u8)klen, (ec_alg_type)alg);
		if (ret) {
			continue;
		}
		if (nn_export_to_buf(priv_out + (size_t)i * qlen, (u16)qlen, &kp.priv_key.x) || prj_pt_iszero(&kp.pub_key.y, &iszero)) {
			continue;
		}
		if (iszero) {
			status[i] = 2;
			continue;
		}
		ret = ec_pub_key_export_to_aff_buf(&kp.pub_key, pub_aff + (size_t)i * 2 * clen, (u8)(2 * clen));
		status[i] = ret ? 1 : 0;
	}
	return 0;
}

/* ---- ec_structured_sig_import_from_buf (sig/sig_algs.c:702): the three header bytes it hands back + the raw signature ---- */
int refdrv_structured_sig_batch(uint32_t n, const uint8_t *in, uint32_t in_len, uint8_t *raw, uint8_t *hdr, uint8_t *status)
{
	uint32_t i;
	for (i = 0; i < n; i++) {
		ec_alg_type st = UNKNOWN_ALG;
		hash_alg_type ht = UNKNOWN_HASH_ALG;
		u8 name[MAX_CURVE_NAME_LEN];
		ec_curve_type ct = UNKNOWN_CURVE;
		u32 l = 0;
		int ret = ec_structured_sig_import_from_buf(raw + (size_t)i * (in_len - 3), in_len - 3, in + (size_t)i * in_len, in_len, &st, &ht, name);
		status[i] = ret ? 1 : 0;
		hdr[3 * i] = hdr[3 * i + 1] = hdr[3 * i + 2] = 0;
		if (!ret) {
			local_strlen((const char *)name, &l);
			ret = ec_get_curve_type_by_name(name, (u8)(l + 1), &ct);
			hdr[3 * i] = (uint8_t)st;
			hdr[3 * i + 1] = (uint8_t)ht;
			hdr[3 * i + 2] = (uint8_t)ct;
		}
	}
	return 0;
}

/* ---- eddsa_export_pub_key (sig/eddsa.c:795-860) on keys given as projective Weierstrass points X || Y || Z: the octets a
 * verifier hashes as "A".  ret[i] = 0 / -1 (the import or the export failed). ---- */
int refdrv_eddsa_export_pub_key_batch(int is448, uint32_t n, const uint8_t *points_prj, uint8_t *enc, int *ret)
{
	ec_params params;
	uint32_t i;
	const size_t klen = is448 ? 57 : 32;
	const ec_alg_type alg = is448 ? EDDSA448 : EDDSA25519;
	size_t clen;
	if (load_params(is448 ? "WEI448" : "WEI25519", &params)) {
		return -1;
	}
	clen = (size_t)BYTECEIL(params.ec_fp.p_bitlen);
	for (i = 0; i < n; i++) {
		ec_pub_key pk;
		memset(enc + (size_t)i * klen, 0, klen);
		ret[i] = ec_pub_key_import_from_buf(&pk, &params, points_prj + (size_t)i * 3 * clen, (u8)(3 * clen), alg);
		if (!ret[i]) {
			ret[i] = eddsa_export_pub_key(&pk, enc + (size_t)i * klen, (u16)klen);
		}
		ret[i] = ret[i] ? -1 : 0;
	}
	return 0;
}

/* The reference's own nn_get_random_mod (nn/nn_rand.c:92) with get_random replaying the caller's bytes: raw n x 2*qlen, out n x qlen
 * big-endian; q = the generator order of the named curve.  Returns -1 if the function asked for another number of bytes. */
int refdrv_random_mod_batch(const char *curve, uint32_t n, const uint8_t *raw, uint8_t *out)
{
	ec_params params;
	uint32_t i;
	u16 ql;
	int ret = 0;
	if (load_params(curve, &params)) {
		return -1;
	}
	ql = (u16)BYTECEIL(params.ec_gen_order_bitlen);
	for (i = 0; i < n && !ret; i++) {
		nn k;
		k.magic = WORD(0);
		tl_replay = raw + (size_t)i * 2 * ql;
		tl_replay_left = 2u * ql;
		ret = nn_get_random_mod(&k, &(params.ec_gen_order)) || tl_replay_left != 0 || nn_export_to_buf(out + (size_t)i * ql, ql, &k);
		nn_uninit(&k);
	}
	tl_replay = NULL;
	tl_replay_left = 0;
	return ret ? -1 : 0;
}
