/*
 * libecc_amd/compat/libecc_amd_compat.c -- the boundary in libecc's own types (include/libecc_amd_compat.h).
 *
 * Compiled against the application's libecc headers (libsig.h) and linked with its libsign objects into
 * libsign_amd.so.  Everything here is marshalling: libecc structures <-> the wire bytes of include/libecc_amd.h
 * (through libecc's own exporters, on a few host threads), message hashing through libecc's hash_maps[] (hashes
 * stay on the host), and the argument checks of the scalar functions that sit in front of the
 * arithmetic.  All curve arithmetic of the batch entry points runs on the GPU(s); nothing here calls into oracle/.
 * File:line references are relative to /root/reference/src.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "libecc_amd_compat.h"
#include "libecc_amd.h"

/* ------------------------------------------------------------------------------------------------
 * process-wide state: the multi-GPU context and the device-side curve handles, keyed by ec_params content
 * ------------------------------------------------------------------------------------------------ */
#define MAX_CURVES 64
typedef struct {
	u8 key[3 * 72 + 2];     /* p || a || b as big-endian octets + lengths: what identifies an ec_shortw_crv */
	u32 key_len;
	ecamd_mcurve *mc;
	u32 clen, qlen;
	nn q;                   /* generator order (reduction of oversize private keys) */
	int has_q;
} curve_ent;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static ecamd_multi *g_multi;
static int g_threads;
static curve_ent g_curves[MAX_CURVES];
static u32 g_ncurves;
static unsigned long long g_items;

static void note_items(u32 n)
{
	pthread_mutex_lock(&g_mu);
	g_items += n;
	pthread_mutex_unlock(&g_mu);
}

unsigned long long ecamd_compat_gpu_items(void) { return g_items; }

static int compat_init_locked(const int *devices, int ndev, int host_threads)
{
	int devs[64], nd = 0;
	if (g_multi) {
		return 0;
	}
	if (!devices || ndev <= 0) {
		const char *e = getenv("ECAMD_DEVICES");
		while (e && *e && nd < 64) {
			devs[nd++] = atoi(e);
			e = strchr(e, ',');
			if (e) {
				e++;
			}
		}
		devices = nd ? devs : NULL;
		ndev = nd;
	}
	if (ecamd_multi_create(&g_multi, devices, ndev)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		g_multi = NULL;
		return -1;
	}
	if (host_threads <= 0) {
		const char *e = getenv("ECAMD_COMPAT_THREADS");
		host_threads = e ? atoi(e) : (int)sysconf(_SC_NPROCESSORS_ONLN);
	}
	g_threads = host_threads < 1 ? 1 : (host_threads > 256 ? 256 : host_threads);
	return 0;
}

int ecamd_compat_init(const int *devices, int ndev, int host_threads)
{
	int ret;
	pthread_mutex_lock(&g_mu);
	ret = compat_init_locked(devices, ndev, host_threads);
	pthread_mutex_unlock(&g_mu);
	return ret;
}

void ecamd_compat_shutdown(void)
{
	u32 i;
	pthread_mutex_lock(&g_mu);
	for (i = 0; i < g_ncurves; i++) {
		ecamd_multi_curve_free(g_curves[i].mc);
	}
	g_ncurves = 0;
	if (g_multi) {
		ecamd_multi_destroy(g_multi);
		g_multi = NULL;
	}
	pthread_mutex_unlock(&g_mu);
}

/* p || a || b of a curve, big-endian, each BYTECEIL(p_bitlen) bytes */
static int crv_key(ec_shortw_crv_src_t crv, u8 *key, u32 *key_len, u32 *clen_out)
{
	int ret;
	u32 clen;
	MUST_HAVE((crv != NULL) && (crv->a.ctx != NULL), ret, err);
	clen = (u32)BYTECEIL(crv->a.ctx->p_bitlen);
	MUST_HAVE((clen > 0) && (clen <= 72), ret, err);
	ret = nn_export_to_buf(key, (u16)clen, &(crv->a.ctx->p)); EG(ret, err);
	ret = fp_export_to_buf(key + clen, (u16)clen, &(crv->a)); EG(ret, err);
	ret = fp_export_to_buf(key + 2 * clen, (u16)clen, &(crv->b)); EG(ret, err);
	*key_len = 3 * clen;
	*clen_out = clen;
err:
	return ret;
}

static curve_ent *curve_find_locked(const u8 *key, u32 key_len)
{
	u32 i;
	for (i = 0; i < g_ncurves; i++) {
		if (g_curves[i].key_len == key_len && !memcmp(g_curves[i].key, key, key_len)) {
			return &g_curves[i];
		}
	}
	return NULL;
}

/* device-side handle of the curve described by `params` (created on first use) */
static curve_ent *curve_from_params_locked(const ec_params *params)
{
	u8 key[3 * 72 + 2], buf[7][80];
	u32 key_len = 0, clen = 0, olen, qlen, i;
	curve_ent *e;
	aff_pt g;
	int ret;
	ecamd_mcurve *mc = NULL;
	g.magic = WORD(0);
	if (crv_key(&(params->ec_curve), key, &key_len, &clen)) {
		return NULL;
	}
	e = curve_find_locked(key, key_len);
	if (e) {
		return e;
	}
	if (g_ncurves >= MAX_CURVES) {
		return NULL;
	}
	/* a built-in curve: by name (same names on both sides); the handle's parameters are then checked against p */
	if (params->curve_name[0] && !ecamd_multi_curve_by_name(g_multi, (const char *)params->curve_name, &mc)) {
		if ((u32)ecamd_multi_curve_coord_len(mc) != clen) {
			ecamd_multi_curve_free(mc);
			mc = NULL;
		}
	}
	qlen = (u32)BYTECEIL(params->ec_gen_order_bitlen);
	if (!mc) {
		/* a user curve: raw domain parameters, exported by libecc itself */
		bitcnt_t ob = 0;
		ret = nn_bitlen(&(params->ec_curve.order), &ob); EG(ret, err);
		olen = (u32)BYTECEIL(ob);
		MUST_HAVE((olen <= 80) && (qlen <= 80), ret, err);
		ret = prj_pt_to_aff(&g, &(params->ec_gen)); EG(ret, err);
		ret = nn_export_to_buf(buf[3], (u16)olen, &(params->ec_curve.order)); EG(ret, err);
		ret = fp_export_to_buf(buf[4], (u16)clen, &(g.x)); EG(ret, err);
		ret = fp_export_to_buf(buf[5], (u16)clen, &(g.y)); EG(ret, err);
		ret = nn_export_to_buf(buf[6], (u16)qlen, &(params->ec_gen_order)); EG(ret, err);
		if (ecamd_multi_curve_from_params(g_multi, key, clen, key + clen, clen, key + 2 * clen, clen, buf[3], olen, buf[4], clen,
						  buf[5], clen, buf[6], qlen, &mc)) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			goto err;
		}
	}
	e = &g_curves[g_ncurves];
	memcpy(e->key, key, key_len);
	e->key_len = key_len;
	e->mc = mc;
	e->clen = clen;
	e->qlen = (u32)ecamd_multi_curve_order_len(mc);
	e->has_q = !nn_copy(&e->q, &(params->ec_gen_order));
	g_ncurves++;
	aff_pt_uninit(&g);
	(void)i;
	return e;
err:
	aff_pt_uninit(&g);
	return NULL;
}

int ecamd_compat_register_params(const ec_params *params)
{
	curve_ent *e = NULL;
	if (!params) {
		return -1;
	}
	pthread_mutex_lock(&g_mu);
	if (!compat_init_locked(NULL, 0, 0)) {
		e = curve_from_params_locked(params);
	}
	pthread_mutex_unlock(&g_mu);
	return e ? 0 : -1;
}

static curve_ent *curve_from_params(const ec_params *params)
{
	curve_ent *e = NULL;
	pthread_mutex_lock(&g_mu);
	if (!compat_init_locked(NULL, 0, 0)) {
		e = curve_from_params_locked(params);
	}
	pthread_mutex_unlock(&g_mu);
	return e;
}

/* handle for a bare ec_shortw_crv (prj_pt arrays): a curve seen before, else one of libecc's built-in curves */
static curve_ent *curve_from_crv(ec_shortw_crv_src_t crv)
{
	u8 key[3 * 72 + 2];
	u32 key_len = 0, clen = 0;
	curve_ent *e = NULL;
	unsigned t;
	if (crv_key(crv, key, &key_len, &clen)) {
		return NULL;
	}
	pthread_mutex_lock(&g_mu);
	if (compat_init_locked(NULL, 0, 0)) {
		pthread_mutex_unlock(&g_mu);
		return NULL;
	}
	e = curve_find_locked(key, key_len);
	for (t = 1; !e && t < 256; t++) {
		const ec_str_params *sp = NULL;
		ec_params params;
		u8 k2[3 * 72 + 2];
		u32 l2 = 0, c2 = 0;
		if (ec_get_curve_params_by_type((ec_curve_type)t, &sp) || !sp) {
			continue;
		}
		if (import_params(&params, sp) || crv_key(&(params.ec_curve), k2, &l2, &c2)) {
			continue;
		}
		if (l2 == key_len && !memcmp(k2, key, key_len)) {
			e = curve_from_params_locked(&params);
			break;
		}
	}
	pthread_mutex_unlock(&g_mu);
	return e;
}

/* ------------------------------------------------------------------------------------------------
 * host threads for the marshalling loops (libecc is re-entrant: no locks, no static state)
 * ------------------------------------------------------------------------------------------------ */
typedef void (*range_fn)(u32 lo, u32 hi, void *arg);
typedef struct {
	range_fn fn;
	void *arg;
	u32 lo, hi;
} range_job;

static void *range_thread(void *p)
{
	range_job *j = (range_job *)p;
	j->fn(j->lo, j->hi, j->arg);
	return NULL;
}

static void parallel_for(u32 n, range_fn fn, void *arg)
{
	int nt = g_threads, t;
	pthread_t th[256];
	range_job jobs[256];
	u8 started[256];
	if (nt > 1 && n / 64 < (u32)nt) {
		nt = (int)(n / 64);
	}
	if (nt <= 1) {
		fn(0, n, arg);
		return;
	}
	for (t = 0; t < nt; t++) {
		jobs[t].fn = fn;
		jobs[t].arg = arg;
		jobs[t].lo = (u32)(((u64)n * (u64)t) / (u64)nt);
		jobs[t].hi = (u32)(((u64)n * (u64)(t + 1)) / (u64)nt);
		started[t] = 0;
	}
	for (t = 0; t < nt - 1; t++) {
		started[t] = pthread_create(&th[t], NULL, range_thread, &jobs[t]) == 0;
	}
	fn(jobs[nt - 1].lo, jobs[nt - 1].hi, arg);   /* the calling thread takes the last range ... */
	for (t = 0; t < nt - 1; t++) {
		if (started[t]) {
			pthread_join(th[t], NULL);
		} else {
			fn(jobs[t].lo, jobs[t].hi, arg);  /* ... and any range whose thread did not start */
		}
	}
}

/* ------------------------------------------------------------------------------------------------
 * prj_pt_mul_batch
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const nn *m;
	const prj_pt *in;
	prj_pt *out;
	int *ret_items;
	u8 *sc, *pin, *pout, *st;
	u32 slen, clen;
	ec_shortw_crv_src_t crv;
} mul_job;

static void mul_export(u32 lo, u32 hi, void *arg)
{
	mul_job *J = (mul_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		/* a point of another curve, an uninitialised point or scalar: prj_pt_mul returns -1 (prj_pt.c:1765-1767) */
		if (J->in[i].magic == WORD(0) || J->in[i].crv != J->crv || nn_export_to_buf(J->sc + (size_t)i * J->slen, (u16)J->slen, &J->m[i]) ||
		    prj_pt_export_to_buf(&J->in[i], J->pin + (size_t)i * 3 * J->clen, 3 * J->clen)) {
			J->st[i] = 0xff;
			memset(J->sc + (size_t)i * J->slen, 0, J->slen);
			memset(J->pin + (size_t)i * 3 * J->clen, 0xff, 3 * J->clen);   /* coordinates >= p: rejected at import */
		} else {
			J->st[i] = 0;
		}
	}
}

static void mul_import(u32 lo, u32 hi, void *arg)
{
	mul_job *J = (mul_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		int r = -1;
		if (J->st[i] == ECAMD_OK) {
			r = prj_pt_import_from_buf(&J->out[i], J->pout + (size_t)i * 3 * J->clen, (u16)(3 * J->clen), J->crv);
		} else if (J->st[i] == ECAMD_INF) {
			r = (prj_pt_init(&J->out[i], J->crv) || prj_pt_zero(&J->out[i])) ? -1 : 0;
		}
		if (J->ret_items) {
			J->ret_items[i] = r;
		}
	}
}

static int mul_batch_common(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items);

int prj_pt_mul_batch(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items)
{
	return mul_batch_common(out, m, in, n, ret_items);
}

typedef struct {
	const nn *m;
	nn *mb;
	nn_src_t order;
	int failed;
} blind_job;

static void blind_scalars(u32 lo, u32 hi, void *arg)
{
	blind_job *B = (blind_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		/* the scalar arithmetic of prj_pt_mul_blind (curves/prj_pt.c:1795-1806), done by the application's own libecc:
		 * b random in [1, #E), scalar = m + b * #E */
		nn b;
		b.magic = WORD(0);
		if (nn_get_random_mod(&b, B->order) || nn_mul(&b, &b, B->order) || nn_add(&B->mb[i], &B->m[i], &b)) {
			B->failed = 1;
		}
		nn_uninit(&b);
	}
}

int prj_pt_mul_blind_batch(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items)
{
	blind_job B;
	int ret;
	if (!out || !m || !in) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	if (prj_pt_check_initialized(&in[0])) {
		return -1;
	}
	B.m = m;
	B.order = &(in[0].crv->order);
	B.failed = 0;
	B.mb = (nn *)calloc(n, sizeof(nn));
	if (!B.mb) {
		return -1;
	}
	if (ecamd_compat_init(NULL, 0, 0)) {
		free(B.mb);
		return -1;
	}
	parallel_for(n, blind_scalars, &B);
	ret = B.failed ? -1 : mul_batch_common(out, B.mb, in, n, ret_items);
	memset(B.mb, 0, (size_t)n * sizeof(nn));
	free(B.mb);
	return ret;
}

static int mul_batch_common(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items)
{
	mul_job J;
	curve_ent *e;
	u32 i, maxbits = 0;
	int ret = -1;
	u8 *pre = NULL;
	memset(&J, 0, sizeof(J));
	if (!out || !m || !in) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	if (prj_pt_check_initialized(&in[0])) {
		return -1;
	}
	e = curve_from_crv(in[0].crv);
	if (!e) {
		return -1;
	}
	for (i = 0; i < n; i++) {
		bitcnt_t b = 0;
		if (!nn_bitlen(&m[i], &b) && (u32)b > maxbits) {
			maxbits = (u32)b;
		}
	}
	J.m = m;
	J.in = in;
	J.out = out;
	J.ret_items = ret_items;
	J.crv = in[0].crv;
	J.clen = e->clen;
	J.slen = (u32)BYTECEIL(maxbits);
	if (J.slen < e->qlen) {
		J.slen = e->qlen;   /* the fast kernels take scalars of up to the order's length; longer ones the generic kernel */
	}
	J.sc = (u8 *)malloc((size_t)n * J.slen);
	J.pin = (u8 *)malloc((size_t)n * 3 * J.clen);
	J.pout = (u8 *)malloc((size_t)n * 3 * J.clen);
	J.st = (u8 *)malloc(n);
	pre = (u8 *)malloc(n);
	if (!J.sc || !J.pin || !J.pout || !J.st || !pre) {
		goto done;
	}
	parallel_for(n, mul_export, &J);
	memcpy(pre, J.st, n);
	if (ecamd_multi_prj_pt_mul_batch_fmt(g_multi, e->mc, n, J.sc, J.slen, J.pin, ECAMD_PT_PROJECTIVE, J.pout, ECAMD_PT_PROJECTIVE, J.st)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		goto done;
	}
	for (i = 0; i < n; i++) {
		if (pre[i]) {
			J.st[i] = ECAMD_ERR;
		}
	}
	note_items(n);
	parallel_for(n, mul_import, &J);
	ret = 0;
done:
	free(J.sc);
	free(J.pin);
	free(J.pout);
	free(J.st);
	free(pre);
	return ret;
}

/* ------------------------------------------------------------------------------------------------
 * ecccdh_derive_secret_batch
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const ec_priv_key *const *privs;
	const u8 *const *peers;
	u8 *const *secrets;
	const ec_params *params;
	u8 *pv, *pk, *sec, *st, *pre;
	u32 qlen, clen;
	nn_src_t q;
	int *ret_items;
} cdh_job;

static void cdh_export(u32 lo, u32 hi, void *arg)
{
	cdh_job *J = (cdh_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const ec_priv_key *k = J->privs[i];
		int bad = 1;
		/* sanity checks of ecccdh_derive_secret (ecdh/ecccdh.c:176-178) */
		if (J->secrets[i] && J->peers[i] && !priv_key_check_initialized_and_type(k, ECCCDH) && k->params == J->params) {
			bitcnt_t b = 0;
			bad = nn_bitlen(&k->x, &b);
			if (!bad && (u32)b > 8 * J->qlen) {
				/* a private scalar longer than the group order: [x]Q = [x mod q]Q for every Q that passes the
				 * checks in front of the multiplication (Q lies in the subgroup of order q) */
				nn t;
				t.magic = WORD(0);
				bad = nn_mod(&t, &k->x, J->q) || nn_export_to_buf(J->pv + (size_t)i * J->qlen, (u16)J->qlen, &t);
				nn_uninit(&t);
			} else if (!bad) {
				bad = nn_export_to_buf(J->pv + (size_t)i * J->qlen, (u16)J->qlen, &k->x);
			}
		}
		J->pre[i] = bad ? 1 : 0;
		if (bad) {
			memset(J->pv + (size_t)i * J->qlen, 0, J->qlen);
			memset(J->pk + (size_t)i * 2 * J->clen, 0xff, 2 * J->clen);
		} else {
			memcpy(J->pk + (size_t)i * 2 * J->clen, J->peers[i], 2 * J->clen);
		}
	}
}

static void cdh_import(u32 lo, u32 hi, void *arg)
{
	cdh_job *J = (cdh_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const int ok = !J->pre[i] && J->st[i] == ECAMD_OK;
		if (ok) {
			memcpy(J->secrets[i], J->sec + (size_t)i * J->clen, J->clen);
		}
		if (J->ret_items) {
			J->ret_items[i] = ok ? 0 : -1;
		}
	}
}

int ecccdh_derive_secret_batch(const ec_priv_key *const *our_priv_keys, const u8 *const *peer_pub_keys, u8 peer_pub_key_len,
			       u8 *const *shared_secrets, u8 shared_secret_len, u32 num, int *ret_items)
{
	cdh_job J;
	curve_ent *e;
	u8 want_pk = 0, want_ss = 0;
	u32 i;
	int ret = -1;
	memset(&J, 0, sizeof(J));
	if (!our_priv_keys || !peer_pub_keys || !shared_secrets) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	if (priv_key_check_initialized_and_type(our_priv_keys[0], ECCCDH)) {
		return -1;
	}
	J.params = our_priv_keys[0]->params;
	/* the two length checks of the scalar call: ec_pub_key_import_from_aff_buf wants exactly 2 coordinates,
	 * the secret is exactly one (ecdh/ecccdh.c:215-216) */
	if (ecccdh_serialized_pub_key_size(J.params, &want_pk) || ecccdh_shared_secret_size(J.params, &want_ss) ||
	    peer_pub_key_len != want_pk || shared_secret_len != want_ss) {
		if (ret_items) {
			for (i = 0; i < num; i++) {
				ret_items[i] = -1;
			}
		}
		return 0;
	}
	e = curve_from_params(J.params);
	if (!e) {
		return -1;
	}
	J.privs = our_priv_keys;
	J.peers = peer_pub_keys;
	J.secrets = shared_secrets;
	J.ret_items = ret_items;
	J.qlen = e->qlen;
	J.clen = e->clen;
	J.q = &(J.params->ec_gen_order);
	J.pv = (u8 *)malloc((size_t)num * J.qlen);
	J.pk = (u8 *)malloc((size_t)num * 2 * J.clen);
	J.sec = (u8 *)malloc((size_t)num * J.clen);
	J.st = (u8 *)malloc(num);
	J.pre = (u8 *)malloc(num);
	if (!J.pv || !J.pk || !J.sec || !J.st || !J.pre) {
		goto done;
	}
	parallel_for(num, cdh_export, &J);
	if (ecamd_multi_ecccdh_derive_batch(g_multi, e->mc, num, J.pv, J.pk, J.sec, J.st)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		goto done;
	}
	note_items(num);
	parallel_for(num, cdh_import, &J);
	ret = 0;
done:
	if (J.pv) {
		memset(J.pv, 0, (size_t)num * J.qlen);   /* private scalars */
	}
	free(J.pv);
	free(J.pk);
	free(J.sec);
	free(J.st);
	free(J.pre);
	return ret;
}

/* ------------------------------------------------------------------------------------------------
 * signature verification
 * ------------------------------------------------------------------------------------------------ */
static const hash_mapping *find_hash(hash_alg_type hash_type)
{
	const hash_mapping *hm = NULL;
	if (get_hash_by_type(hash_type, &hm) || !hm || hash_mapping_callbacks_sanity_check(hm)) {
		return NULL;
	}
	return hm;
}

typedef struct {
	const u8 **s, **m, **adata;
	const u8 *s_len;
	const u32 *m_len;
	const u16 *adata_len;
	const ec_pub_key **pub_keys;
	ec_alg_type sig_type;
	const hash_mapping *hm;
	const u32 *idx;          /* the items of this group */
	const ec_params *params;
	u8 *pk, *sg, *dg, *pre;  /* packed: keys (projective X||Y||Z or EdDSA encoding), signatures, digests / hram, per-item pre-check */
	u8 *kprj;                /* EdDSA: the key points X||Y||Z, input of the device-side encoding */
	u32 clen, qlen, hlen, klen, siglen;
} ver_job;

/* ---- ECDSA / DECDSA ---- */
static void ecdsa_pack(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_pub_key *pk = J->pub_keys[i];
		hash_context hc;
		u8 dig[MAX_DIGEST_SIZE];
		int bad;
		/* ec_verify_init / __ecdsa_verify_init (sig/sig_algs.c:516, sig/ecdsa_common.c:623-649): key initialised and of
		 * this algorithm, signature present and of the expected length; then H(m) */
		bad = pub_key_check_initialized_and_type(pk, J->sig_type) || pk->params != J->params || !J->s[i] ||
		      J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
		bad = bad || prj_pt_export_to_buf(&pk->y, J->pk + (size_t)j * 3 * J->clen, 3 * J->clen);
		bad = bad || J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hc, dig);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(J->pk + (size_t)j * 3 * J->clen, 0xff, 3 * J->clen);
			memset(J->sg + (size_t)j * J->siglen, 0, J->siglen);
			memset(J->dg + (size_t)j * J->hlen, 0, J->hlen);
		} else {
			memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
			memcpy(J->dg + (size_t)j * J->hlen, dig, J->hlen);
		}
	}
}

/* results[i] for the items idx[0..cnt) that share `params` */
static int ecdsa_group(ver_job *J, u32 cnt, int *results)
{
	curve_ent *e = curve_from_params(J->params);
	u8 *res = NULL;
	u32 j;
	int ret = -1;
	if (!e) {
		return -1;
	}
	J->clen = e->clen;
	J->qlen = e->qlen;
	J->siglen = 2 * (u32)BYTECEIL(J->params->ec_gen_order_bitlen);   /* ECDSA_SIGLEN */
	if (J->siglen != 2 * e->qlen) {
		return -1;
	}
	J->hlen = J->hm->digest_size;
	J->pk = (u8 *)malloc((size_t)cnt * 3 * J->clen);
	J->sg = (u8 *)malloc((size_t)cnt * J->siglen);
	J->dg = (u8 *)malloc((size_t)cnt * J->hlen);
	J->pre = (u8 *)malloc(cnt);
	res = (u8 *)malloc(cnt);
	if (!J->pk || !J->sg || !J->dg || !J->pre || !res) {
		goto done;
	}
	parallel_for(cnt, ecdsa_pack, J);
	if (ecamd_multi_ecdsa_verify_batch_fmt(g_multi, e->mc, cnt, J->pk, ECAMD_PT_PROJECTIVE, J->sg, J->dg, J->hlen, res)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		goto done;
	}
	note_items(cnt);
	for (j = 0; j < cnt; j++) {
		results[J->idx[j]] = (J->pre[j] || res[j]) ? -1 : 0;
	}
	ret = 0;
done:
	free(J->pk);
	free(J->sg);
	free(J->dg);
	free(J->pre);
	free(J->kprj);
	free(res);
	J->pk = J->sg = J->dg = J->pre = J->kprj = NULL;
	return ret;
}

/* ---- EdDSA ---- */
static int eddsa_variant(ec_alg_type t, hash_alg_type *h, ec_curve_type *c, int *ph, int *dom, int *is448)
{
	switch (t) {
#if defined(WITH_SIG_EDDSA25519)
	case EDDSA25519: *h = SHA512; *c = WEI25519; *ph = 0; *dom = 0; *is448 = 0; return 0;
	case EDDSA25519CTX: *h = SHA512; *c = WEI25519; *ph = 0; *dom = 1; *is448 = 0; return 0;
	case EDDSA25519PH: *h = SHA512; *c = WEI25519; *ph = 1; *dom = 1; *is448 = 0; return 0;
#endif
#if defined(WITH_SIG_EDDSA448)
	case EDDSA448: *h = SHAKE256; *c = WEI448; *ph = 0; *dom = 1; *is448 = 1; return 0;
	case EDDSA448PH: *h = SHAKE256; *c = WEI448; *ph = 1; *dom = 1; *is448 = 1; return 0;
#endif
	default: return -1;
	}
}

/* dom2(x, y) / dom4(x, y) of RFC 8032 (sig/eddsa.c:56-84) */
static int dom_prefix(const hash_mapping *hm, hash_context *hc, int is448, int ph, const u8 *y, u16 ylen)
{
	u8 t[2];
	if (ylen > 255) {
		return -1;
	}
	if (is448) {
		if (hm->hfunc_update(hc, (const u8 *)"SigEd448", 8)) {
			return -1;
		}
	} else if (hm->hfunc_update(hc, (const u8 *)"SigEd25519 no Ed25519 collisions", 32)) {
		return -1;
	}
	t[0] = (u8)ph;
	t[1] = (u8)ylen;
	if (hm->hfunc_update(hc, t, 2)) {
		return -1;
	}
	return y ? hm->hfunc_update(hc, y, ylen) : 0;
}

typedef struct {
	ver_job v;
	int ph, dom, is448;
	u32 ph_len;   /* bytes of PH(M) that enter the main hash */
} ed_job;

/* first pass of an EdDSA group: the projective key points as octets for ec_eddsa_encode_point_batch */
static void eddsa_export_keys(u32 lo, u32 hi, void *arg)
{
	ed_job *E = (ed_job *)arg;
	ver_job *J = &E->v;
	u32 j;
	for (j = lo; j < hi; j++) {
		const ec_pub_key *pk = J->pub_keys[J->idx[j]];
		u8 *dst = J->kprj + (size_t)j * 3 * J->clen;
		if (pub_key_check_initialized_and_type(pk, J->sig_type) || pk->params != J->params ||
		    prj_pt_export_to_buf(&pk->y, dst, (u32)(3 * J->clen))) {
			memset(dst, 0xff, (size_t)3 * J->clen);   /* coordinates >= p: an import error on the device */
		}
	}
}

static void eddsa_pack(u32 lo, u32 hi, void *arg)
{
	ed_job *E = (ed_job *)arg;
	ver_job *J = &E->v;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_pub_key *pk = J->pub_keys[i];
		const u8 *ad = J->adata ? J->adata[i] : NULL;
		const u16 adl = J->adata_len ? J->adata_len[i] : 0;   /* dom() writes the length octet even without a context (sig/eddsa.c:77-81) */
		hash_context hc, hp;
		u8 dig[MAX_DIGEST_SIZE];
		u8 *kenc = J->pk + (size_t)j * J->klen;
		int bad;
		/* _eddsa_verify_init (sig/eddsa.c:1880-1960): key of this variant, on the variant's curve, signature length;
		 * EDDSA25519CTX wants a context (:1923) */
		bad = pub_key_check_initialized_and_type(pk, J->sig_type) || pk->params != J->params || !J->s[i] ||
		      J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
#if defined(WITH_SIG_EDDSA25519)
		bad = bad || (J->sig_type == EDDSA25519CTX && !ad);
#endif
		/* the encoding of the key as the reference hashes it -- eddsa_export_pub_key: Weierstrass -> Edwards -> octets -- was
		 * computed on the device for the whole group (eddsa_group; libecc's own export costs about 1.5 ms of CPU per key) */
		bad = bad || J->pre[j];
		bad = bad || J->hm->hfunc_init(&hc);
		if (!bad && E->dom) {
			bad = dom_prefix(J->hm, &hc, E->is448, E->ph, ad, adl);
		}
		bad = bad || J->hm->hfunc_update(&hc, J->s[i], J->klen) || J->hm->hfunc_update(&hc, kenc, J->klen);
		if (!bad && E->ph) {
			bad = J->hm->hfunc_init(&hp) || J->hm->hfunc_update(&hp, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hp, dig) ||
			      J->hm->hfunc_update(&hc, dig, E->ph_len);
		} else if (!bad) {
			bad = J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]);
		}
		bad = bad || J->hm->hfunc_finalize(&hc, dig);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(kenc, 0xff, J->klen);   /* y >= p: rejected by the decoder */
			memset(J->sg + (size_t)j * J->siglen, 0xff, J->siglen);
			memset(J->dg + (size_t)j * J->hlen, 0, J->hlen);
		} else {
			memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
			memcpy(J->dg + (size_t)j * J->hlen, dig, J->hlen);
		}
	}
}

/* set by eddsa_verify_batch_gpu: the caller only wants ec_verify_batch's one bit, so a group may be decided by the
 * device's multi-scalar multiplication (the reference's own random linear combination, ec_eddsa_verify_all_batch) */
static __thread int t_all_only = 0;

static int eddsa_group(ed_job *E, u32 cnt, int *results)
{
	ver_job *J = &E->v;
	curve_ent *e = curve_from_params(J->params);
	u8 *res = NULL;
	u32 j;
	int ret = -1;
	if (!e) {
		return -1;
	}
	J->clen = e->clen;
	J->hlen = J->hm->digest_size;      /* 64 (SHA-512) / 114 (SHAKE256 as libecc configures it) */
	J->klen = J->hlen / 2;             /* EDDSA_R_LEN: 32 / 57 */
	J->siglen = J->hlen;               /* EDDSA_SIGLEN */
	if (J->klen != (E->is448 ? 57u : 32u) || J->clen != (E->is448 ? 56u : 32u)) {
		return -1;
	}
	J->pk = (u8 *)malloc((size_t)cnt * J->klen);
	J->sg = (u8 *)malloc((size_t)cnt * J->siglen);
	J->dg = (u8 *)malloc((size_t)cnt * J->hlen);
	J->pre = (u8 *)malloc(cnt);
	J->kprj = (u8 *)malloc((size_t)cnt * 3 * J->clen);
	res = (u8 *)malloc(cnt);
	if (!J->pk || !J->sg || !J->dg || !J->pre || !J->kprj || !res) {
		goto done;
	}
	/* the keys as the reference hashes them: exported as points here, encoded on the device (pre[j] != 0: no encoding) */
	parallel_for(cnt, eddsa_export_keys, E);
	if (ecamd_multi_eddsa_encode_point_batch(g_multi, e->mc, cnt, J->kprj, J->pk, J->pre)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		goto done;
	}
	parallel_for(cnt, eddsa_pack, E);
	if (t_all_only) {
		int all = 0, pre_bad = 0;
		for (j = 0; j < cnt; j++) {
			pre_bad |= J->pre[j];
		}
		if (!pre_bad) {
			if (ecamd_multi_eddsa_verify_all_batch(g_multi, e->mc, cnt, J->pk, J->sg, J->dg, J->hlen, &all, NULL)) {
				fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
				goto done;
			}
			if (all) {
				note_items(cnt);
				for (j = 0; j < cnt; j++) {
					results[J->idx[j]] = 0;
				}
				ret = 0;
				goto done;
			}
		}
		/* rejected (or an item failed before the device): the item-by-item results below say which */
	}
	if (ecamd_multi_eddsa_verify_batch(g_multi, e->mc, cnt, J->pk, J->sg, J->dg, J->hlen, res)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		goto done;
	}
	note_items(cnt);
	for (j = 0; j < cnt; j++) {
		results[J->idx[j]] = (J->pre[j] || res[j]) ? -1 : 0;
	}
	ret = 0;
done:
	free(J->pk);
	free(J->sg);
	free(J->dg);
	free(J->pre);
	free(res);
	J->pk = J->sg = J->dg = J->pre = NULL;
	return ret;
}

static int is_ecdsa(ec_alg_type t)
{
#if defined(WITH_SIG_ECDSA)
	if (t == ECDSA) {
		return 1;
	}
#endif
#if defined(WITH_SIG_DECDSA)
	if (t == DECDSA) {
		return 1;
	}
#endif
	return 0;
}

int ec_verify_batch_results(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			    ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int *results)
{
	const hash_mapping *hm;
	hash_alg_type eh = UNKNOWN_HASH_ALG;
	ec_curve_type ec = UNKNOWN_CURVE;
	int ph = 0, dom = 0, is448 = 0, ed, ret = -1;
	u32 *idx = NULL, i, done = 0;
	u8 *seen = NULL;
	if (!s || !s_len || !pub_keys || !m || !m_len || !results) {
		return -1;
	}
	ed = !eddsa_variant(sig_type, &eh, &ec, &ph, &dom, &is448);
	if (!ed && !is_ecdsa(sig_type)) {
		return -1;
	}
	for (i = 0; i < num; i++) {
		results[i] = -1;
	}
	if (num == 0) {
		return 0;
	}
	hm = find_hash(hash_type);
	if (!hm || (ed && hash_type != eh)) {
		return 0;   /* every ec_verify fails in ec_verify_init / _eddsa_verify_init */
	}
	idx = (u32 *)malloc((size_t)num * sizeof(u32));
	seen = (u8 *)calloc(num, 1);
	if (!idx || !seen) {
		goto out;
	}
	/* groups of items that share their ec_params (one GPU batch each; normally there is one group) */
	while (done < num) {
		const ec_params *params = NULL;
		u32 cnt = 0;
		for (i = 0; i < num; i++) {
			const ec_pub_key *pk = pub_keys[i];
			if (seen[i]) {
				continue;
			}
			if (!pk || pk->magic != PUB_KEY_MAGIC || !pk->params) {
				seen[i] = 1;   /* stays -1 */
				done++;
				continue;
			}
			if (!params) {
				params = pk->params;
			}
			if (pk->params == params) {
				idx[cnt++] = i;
				seen[i] = 1;
				done++;
			}
		}
		if (!cnt) {
			break;
		}
		if (ed) {
			ed_job E;
			memset(&E, 0, sizeof(E));
			if (params->curve_type != ec) {
				continue;   /* eddsa_key_type_check_curve fails: -1 for the group */
			}
			E.v.s = s; E.v.s_len = s_len; E.v.m = m; E.v.m_len = m_len; E.v.adata = adata; E.v.adata_len = adata_len;
			E.v.pub_keys = pub_keys; E.v.sig_type = sig_type; E.v.hm = hm; E.v.idx = idx; E.v.params = params;
			E.ph = ph; E.dom = dom; E.is448 = is448;
			E.ph_len = is448 ? 64 : hm->digest_size;   /* EDDSA448PH: SHAKE256 with 64 bytes (sig/eddsa.c:2343-2346) */
			if (eddsa_group(&E, cnt, results)) {
				goto out;
			}
		} else {
			ver_job J;
			memset(&J, 0, sizeof(J));
			J.s = s; J.s_len = s_len; J.m = m; J.m_len = m_len; J.adata = adata; J.adata_len = adata_len;
			J.pub_keys = pub_keys; J.sig_type = sig_type; J.hm = hm; J.idx = idx; J.params = params;
			if (ecdsa_group(&J, cnt, results)) {
				goto out;
			}
		}
	}
	ret = 0;
out:
	free(idx);
	free(seen);
	return ret;
}

static int all_accepted(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len)
{
	int *res, ret = -1;
	u32 i;
	if (num == 0) {
		return -1;   /* "We need at least one element in our batch data bags" (sig/eddsa.c:2312) */
	}
	res = (int *)malloc((size_t)num * sizeof(int));
	if (!res) {
		return -1;
	}
	if (!ec_verify_batch_results(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, res)) {
		ret = 0;
		for (i = 0; i < num; i++) {
			if (res[i]) {
				ret = -1;
			}
		}
	}
	free(res);
	return ret;
}

int ecdsa_verify_batch(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
		       ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
		       verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len)
{
	FORCE_USED_VAR(scratch_pad_area);
	FORCE_USED_VAR(scratch_pad_area_len);
	if (!is_ecdsa(sig_type)) {
		return -1;
	}
	return all_accepted(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len);
}

int eddsa_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			   ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			   verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len)
{
	hash_alg_type eh = UNKNOWN_HASH_ALG;
	ec_curve_type ec = UNKNOWN_CURVE;
	int ph, dom, is448;
	u32 i;
	if (eddsa_variant(sig_type, &eh, &ec, &ph, &dom, &is448)) {
		return -1;
	}
	/* argument checks of eddsa_verify_batch / _eddsa_verify_batch / _eddsa_verify_batch_no_memory
	 * (sig/eddsa.c:2904-2920, :2612-2650, :2309-2312, :2358) */
	if (!s || !pub_keys || !m || !adata) {
		return -1;
	}
	if (scratch_pad_area) {
		if (!scratch_pad_area_len) {
			return -1;
		}
		if (num > 1) {
			const u64 expected = ((2 * (u64)num) + 1) * sizeof(verify_batch_scratch_pad);
			if (expected >= 0xffffffffULL || *scratch_pad_area_len < expected) {
				return -1;
			}
		}
	}
	if (num == 0 || !pub_keys[0]) {
		return -1;
	}
	for (i = 0; i < num; i++) {
		if (!pub_keys[i] || pub_keys[i]->params != pub_keys[0]->params) {
			return -1;   /* "all our public keys have the same parameters" */
		}
	}
	{
		int r;
		t_all_only = 1;
		r = all_accepted(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len);
		t_all_only = 0;
		return r;
	}
}

/* ------------------------------------------------------------------------------------------------
 * the two replaced libecc symbols
 * ------------------------------------------------------------------------------------------------ */
int ec_verify_batch(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
		    ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
		    verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len)
{
	hash_alg_type eh;
	ec_curve_type ec;
	int ph, dom, is448;
	if (is_ecdsa(sig_type)) {
		return ecdsa_verify_batch(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, scratch_pad_area,
					  scratch_pad_area_len);
	}
	if (!eddsa_variant(sig_type, &eh, &ec, &ph, &dom, &is448)) {
		return eddsa_verify_batch_gpu(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, scratch_pad_area,
					      scratch_pad_area_len);
	}
	return libecc_cpu_ec_verify_batch(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, scratch_pad_area,
					  scratch_pad_area_len);
}

int is_verify_batch_mode_supported(ec_alg_type sig_type, int *check)
{
	if (check && is_ecdsa(sig_type)) {
		*check = 1;
		return 0;
	}
	return libecc_cpu_is_verify_batch_mode_supported(sig_type, check);
}
