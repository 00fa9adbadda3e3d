/*
 * libecc_amd/compat/libecc_amd_compat.c -- the boundary in libecc's own types (include/libecc_amd_compat.h).
 *
 * Compiled against the application's libecc headers (libsig.h) and linked with its libsign objects into
 * libsign_amd.so.  Everything here is marshalling: libecc structures <-> the wire bytes of include/libecc_amd.h, message
 * hashing through libecc's hash_maps[] and nonce generation through libecc's nn_get_random_mod / hmac_* (hashes and
 * randomness stay on the host), and the argument checks of the scalar functions that sit in front of
 * the arithmetic.  All curve arithmetic of the batch entry points runs on the GPU(s); nothing here calls into oracle/.
 *
 * How a batch moves: the items are cut into chunks; a persistent pool of host threads PACKS chunk
 * c + 1 (reads the live limbs of the nn / fp / prj_pt structures -- this file is compiled against the application's own
 * headers, so their layout is known -- and hashes the messages) while the calling thread has chunk c on the GPU(s) and other
 * pool threads UNPACK chunk c - 1 into the caller's structures.  Staging buffers are page-locked and kept across calls.
 * File:line references are relative to /root/reference/src.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <time.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "libecc_amd_compat.h"
#include "libecc_amd.h"
#include "external_deps/rand.h"   /* get_random: the application's randomness source (stays undefined in libsign_amd.so) */

/* ------------------------------------------------------------------------------------------------
 * small helpers
 * ------------------------------------------------------------------------------------------------ */
static void wipe(void *p, size_t n)
{
	if (p && n) {
		explicit_bzero(p, n);
	}
}

#define AT_LOAD(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define AT_STORE(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define AT_ADD(p, v) __atomic_add_fetch((p), (v), __ATOMIC_ACQ_REL)
#define AT_SUB(p, v) __atomic_sub_fetch((p), (v), __ATOMIC_ACQ_REL)

/* ------------------------------------------------------------------------------------------------
 * process-wide state: the multi-GPU context and the device-side curve handles, keyed by ec_params content
 * ------------------------------------------------------------------------------------------------ */
#define MAX_CURVES 64
#define KEY_MAX (4 * 72 + 8)
typedef struct {
	u8 key_crv[KEY_MAX];    /* p || a || b || #E: what identifies an ec_shortw_crv */
	u32 key_crv_len;
	u8 key_gen[KEY_MAX];    /* Gx || Gy || q || cofactor: the rest of an ec_params */
	u32 key_gen_len;
	ecamd_mcurve *mc;
	u32 clen, qlen;
} curve_ent;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;        /* the state below */
static void words_free_all(void);
static void call_enter(int shared);   /* the gate of every batch call: shared (verification, NARENA at a time) or exclusive */
static void call_leave(void);
static pthread_mutex_t g_rand_mu = PTHREAD_MUTEX_INITIALIZER;   /* the application's get_random: one caller at a time (see compat_random_mod) */
static u32 g_rand_concurrent;                                   /* ecamd_compat_set_concurrent_random */

/* nn_get_random_mod (nn/nn_rand.c:92-150: q' = q - 1, a random value of twice q's length, out = that mod q' + 1) with the ONE call
 * that reaches the application -- get_random -- made by one thread at a time.  libecc's scalar API never calls get_random from two
 * threads at once unless the application does, and a stateful source (a user-space DRBG, a seeded test harness) need not be
 * reentrant; the batch forms pack their items on a thread pool, so by default the draw is serialised here (a
 * torn or repeated ECDSA nonce gives the private key away).  ecamd_compat_set_concurrent_random(1) removes the lock for
 * applications whose get_random is thread-safe (e.g. one getrandom(2) / /dev/urandom read per call). */
static int compat_random_mod(nn_t out, nn_src_t q)
{
	nn tmp_rand, qprime;
	bitcnt_t q_bit_len, q_len;
	int ret, isone;
	qprime.magic = tmp_rand.magic = WORD(0);
	ret = nn_check_initialized(q); EG(ret, err);
	ret = nn_bitlen(q, &q_bit_len); EG(ret, err);
	q_len = (bitcnt_t)BYTECEIL(q_bit_len);
	MUST_HAVE((q_len) && (q_len <= (NN_MAX_BYTE_LEN / 2)), ret, err);
	MUST_HAVE((!nn_isone(q, &isone)) && (!isone), ret, err);
	ret = nn_copy(&qprime, q); EG(ret, err);
	ret = nn_dec(&qprime, &qprime); EG(ret, err);
	ret = nn_init(&tmp_rand, (u16)(2 * q_len)); EG(ret, err);
	if (AT_LOAD(&g_rand_concurrent)) {
		ret = get_random((u8 *)tmp_rand.val, (u16)(2 * q_len));
	} else {
		pthread_mutex_lock(&g_rand_mu);
		ret = get_random((u8 *)tmp_rand.val, (u16)(2 * q_len));
		pthread_mutex_unlock(&g_rand_mu);
	}
	EG(ret, err);
	ret = nn_init(out, (u16)q_len); EG(ret, err);
	ret = nn_mod_notrim(out, &tmp_rand, &qprime); EG(ret, err);
	ret = nn_inc(out, out);
err:
	nn_uninit(&qprime);
	nn_uninit(&tmp_rand);
	return ret;
}

/* the application's get_random for `len` bytes, under the same contract */
static int compat_get_random(u8 *buf, u16 len)
{
	int ret;
	if (AT_LOAD(&g_rand_concurrent)) {
		return get_random(buf, len);
	}
	pthread_mutex_lock(&g_rand_mu);
	ret = get_random(buf, len);
	pthread_mutex_unlock(&g_rand_mu);
	return ret;
}
/* Round 4: where the value wanted is exactly nn_get_random_mod's (ECDSA nonces, the private scalar of ec_key_pair_gen's generic rule),
 * only the get_random call stays on the host -- the same single call of 2 * qlen bytes per value the reference makes -- and the
 * reduction modulo q - 1 runs on the device (ec_ecdsa_sign_msg_batch, ec_key_pair_gen_raw_batch): libecc's constant-time division was
 * the largest cost of a signature or a key pair on the host side.  $ECAMD_COMPAT_HOST_RANDMOD keeps it on the host. */
static int raw_random_on_device(void)
{
	static int on = -1;
	if (on < 0) {
		on = getenv("ECAMD_COMPAT_HOST_RANDMOD") ? 0 : 1;
	}
	return on;
}

void ecamd_compat_set_concurrent_random(int on)
{
	AT_STORE(&g_rand_concurrent, on ? 1u : 0u);
}
static ecamd_multi *g_multi;
static int g_threads;
static int g_secret = 1;
static u32 g_chunk = 0;              /* items per pipeline chunk and device; 0: chosen from the batch size (chunk_items) */
static u32 g_chunk_all = 1u << 18;   /* the same for EdDSA whole-batch verification (the multi-scalar multiplication wants >= 2^17) */
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

/* CPUs this process may use: the affinity mask, capped by the cgroup v2 quota */
static int default_threads(void)
{
	int n = 0;
	cpu_set_t set;
	FILE *f;
	if (!sched_getaffinity(0, sizeof(set), &set)) {
		n = CPU_COUNT(&set);
	}
	if (n <= 0) {
		n = (int)sysconf(_SC_NPROCESSORS_ONLN);
	}
	f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		char quota[32];
		long period = 0;
		if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") && period > 0) {
			const long q = atol(quota);
			const int c = (int)((q + period - 1) / period);
			if (c >= 1 && c < n) {
				n = c;
			}
		}
		fclose(f);
	}
	return n < 1 ? 1 : n;
}

static int pool_start(int nthreads);
static void pool_stop(void);

static int compat_init_locked(const int *devices, int ndev, int host_threads)
{
	int devs[64], nd = 0;
	const char *e;
	if (g_multi) {
		return 0;
	}
	if (!devices || ndev <= 0) {
		e = getenv("ECAMD_DEVICES");
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
	if (getenv("ECAMD_COMPAT_CONCURRENT_RANDOM")) {
		AT_STORE(&g_rand_concurrent, 1u);
	}
	g_secret = getenv("ECAMD_COMPAT_PUBLIC_SCALARS") ? 0 : 1;
	if (ecamd_multi_set_secret_scalars(g_multi, g_secret)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		ecamd_multi_destroy(g_multi);
		g_multi = NULL;
		return -1;
	}
	if (host_threads <= 0) {
		e = getenv("ECAMD_COMPAT_THREADS");
		host_threads = e ? atoi(e) : default_threads();
	}
	g_threads = host_threads < 1 ? 1 : (host_threads > 256 ? 256 : host_threads);
	e = getenv("ECAMD_COMPAT_CHUNK");
	if (e && atoi(e) >= 512) {
		g_chunk = (u32)atoi(e);
	}
	e = getenv("ECAMD_COMPAT_CHUNK_ALL");
	if (e && atoi(e) >= 1024) {
		g_chunk_all = (u32)atoi(e);
	}
	if (pool_start(g_threads - 1)) {
		g_threads = 1;
	}
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

int ecamd_compat_set_secret_scalars(int on)
{
	int ret;
	call_enter(0);
	pthread_mutex_lock(&g_mu);
	ret = compat_init_locked(NULL, 0, 0);
	if (!ret) {
		ret = ecamd_multi_set_secret_scalars(g_multi, on);
		if (!ret) {
			g_secret = on ? 1 : 0;
		}
	}
	pthread_mutex_unlock(&g_mu);
	call_leave();
	return ret;
}

static void bufs_free(void);

void ecamd_compat_shutdown(void)
{
	u32 i;
	call_enter(0);
	pthread_mutex_lock(&g_mu);
	pool_stop();
	bufs_free();
	words_free_all();
	for (i = 0; i < g_ncurves; i++) {
		ecamd_multi_curve_free(g_curves[i].mc);
	}
	g_ncurves = 0;
	if (g_multi) {
		ecamd_multi_destroy(g_multi);
		g_multi = NULL;
	}
	pthread_mutex_unlock(&g_mu);
	call_leave();
}

/* p || a || b || #E of a curve, big-endian (coordinates BYTECEIL(p_bitlen) bytes, the order as long as it is) */
static int crv_key(ec_shortw_crv_src_t crv, u8 *key, u32 *key_len, u32 *clen_out)
{
	int ret;
	u32 clen, olen;
	bitcnt_t ob = 0;
	MUST_HAVE((crv != NULL) && (crv->a.ctx != NULL), ret, err);
	clen = (u32)BYTECEIL(crv->a.ctx->p_bitlen);
	MUST_HAVE((clen > 0) && (clen <= 72), ret, err);
	ret = nn_bitlen(&(crv->order), &ob); EG(ret, err);
	olen = (u32)BYTECEIL(ob);
	MUST_HAVE((olen <= 80), ret, err);
	ret = nn_export_to_buf(key, (u16)clen, &(crv->a.ctx->p)); EG(ret, err);
	ret = fp_export_to_buf(key + clen, (u16)clen, &(crv->a)); EG(ret, err);
	ret = fp_export_to_buf(key + 2 * clen, (u16)clen, &(crv->b)); EG(ret, err);
	ret = nn_export_to_buf(key + 3 * clen, (u16)olen, &(crv->order)); EG(ret, err);
	*key_len = 3 * clen + olen;
	*clen_out = clen;
err:
	return ret;
}

/* Gx || Gy || q || cofactor: what an ec_params adds to its curve.  Two ec_params on one curve with another generator or
 * order are different groups and get different device handles. */
static int gen_key(const ec_params *params, u32 clen, u8 *key, u32 *key_len)
{
	int ret;
	aff_pt g;
	u32 qlen, hlen;
	bitcnt_t hb = 0;
	g.magic = WORD(0);
	qlen = (u32)BYTECEIL(params->ec_gen_order_bitlen);
	ret = nn_bitlen(&(params->ec_gen_cofactor), &hb); EG(ret, err);
	hlen = (u32)BYTECEIL(hb);
	MUST_HAVE((qlen > 0) && (qlen <= 80) && (hlen <= 8), ret, err);
	ret = prj_pt_to_aff(&g, &(params->ec_gen)); EG(ret, err);
	ret = fp_export_to_buf(key, (u16)clen, &(g.x)); EG(ret, err);
	ret = fp_export_to_buf(key + clen, (u16)clen, &(g.y)); EG(ret, err);
	ret = nn_export_to_buf(key + 2 * clen, (u16)qlen, &(params->ec_gen_order)); EG(ret, err);
	ret = nn_export_to_buf(key + 2 * clen + qlen, 8, &(params->ec_gen_cofactor)); EG(ret, err);
	*key_len = 2 * clen + qlen + 8;
err:
	aff_pt_uninit(&g);
	return ret;
}

static curve_ent *curve_find_locked(const u8 *kc, u32 kc_len, const u8 *kg, u32 kg_len)
{
	u32 i;
	for (i = 0; i < g_ncurves; i++) {
		curve_ent *e = &g_curves[i];
		if (e->key_crv_len == kc_len && !memcmp(e->key_crv, kc, kc_len) &&
		    (!kg || (e->key_gen_len == kg_len && !memcmp(e->key_gen, kg, kg_len)))) {
			return e;
		}
	}
	return NULL;
}

/* device-side handle of the group described by `params` (created on first use) */
static curve_ent *curve_from_params_locked(const ec_params *params)
{
	u8 kc[KEY_MAX], kg[KEY_MAX];
	u32 kc_len = 0, kg_len = 0, clen = 0, qlen, olen;
	curve_ent *e;
	ecamd_mcurve *mc = NULL;
	if (crv_key(&(params->ec_curve), kc, &kc_len, &clen) || gen_key(params, clen, kg, &kg_len)) {
		return NULL;
	}
	e = curve_find_locked(kc, kc_len, kg, kg_len);
	if (e) {
		return e;
	}
	if (g_ncurves >= MAX_CURVES) {
		return NULL;
	}
	qlen = (u32)BYTECEIL(params->ec_gen_order_bitlen);
	olen = kc_len - 3 * clen;
	/* A built-in curve is taken by name only when the application's parameters ARE libecc's built-in ones of that name (the
	 * device table is generated from the same constants, tools/gen_curve_table.py): p, a, b, #E, G, q and the cofactor are
	 * compared; an application-built ec_params that reuses a name goes through its raw domain parameters below. */
	if (params->curve_name[0]) {
		const ec_str_params *sp = NULL;
		ec_params ref;
		u8 rc[KEY_MAX], rg[KEY_MAX];
		u32 rc_len = 0, rg_len = 0, rclen = 0;
		u8 nlen = 0;
		while (nlen < MAX_CURVE_NAME_LEN && params->curve_name[nlen]) {
			nlen++;
		}
		if (!ec_get_curve_params_by_name(params->curve_name, (u8)(nlen + 1), &sp) && sp && !import_params(&ref, sp) &&
		    !crv_key(&(ref.ec_curve), rc, &rc_len, &rclen) && !gen_key(&ref, rclen, rg, &rg_len) && rc_len == kc_len &&
		    !memcmp(rc, kc, kc_len) && rg_len == kg_len && !memcmp(rg, kg, kg_len)) {
			if (ecamd_multi_curve_by_name(g_multi, (const char *)params->curve_name, &mc)) {
				mc = NULL;
			} else if ((u32)ecamd_multi_curve_coord_len(mc) != clen || (u32)ecamd_multi_curve_order_len(mc) != qlen) {
				ecamd_multi_curve_free(mc);
				mc = NULL;
			}
		}
	}
	if (!mc) {
		/* raw domain parameters, exported by libecc itself */
		if (ecamd_multi_curve_from_params(g_multi, kc, clen, kc + clen, clen, kc + 2 * clen, clen, kc + 3 * clen, olen, kg, clen,
						  kg + clen, clen, kg + 2 * clen, qlen, &mc)) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			return NULL;
		}
	}
	e = &g_curves[g_ncurves];
	memcpy(e->key_crv, kc, kc_len);
	e->key_crv_len = kc_len;
	memcpy(e->key_gen, kg, kg_len);
	e->key_gen_len = kg_len;
	e->mc = mc;
	e->clen = clen;
	e->qlen = (u32)ecamd_multi_curve_order_len(mc);
	g_ncurves++;
	return e;
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
	if (!params) {
		return NULL;
	}
	pthread_mutex_lock(&g_mu);
	if (!compat_init_locked(NULL, 0, 0)) {
		e = curve_from_params_locked(params);
	}
	pthread_mutex_unlock(&g_mu);
	return e;
}

/* handle for a bare ec_shortw_crv (prj_pt arrays carry no generator; none is needed to multiply given points): a group on
 * this curve seen before, else the libecc built-in curve with these p, a, b, #E */
static curve_ent *curve_from_crv(ec_shortw_crv_src_t crv)
{
	u8 key[KEY_MAX];
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
	e = curve_find_locked(key, key_len, NULL, 0);
	for (t = 1; !e && t < 256; t++) {
		const ec_str_params *sp = NULL;
		ec_params params;
		u8 k2[KEY_MAX];
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
 * persistent host threads and the three-stage pipeline  pack(c + 1) | GPU(c) | unpack(c - 1)
 * (libecc is re-entrant: no locks, no static state)
 * ------------------------------------------------------------------------------------------------ */
typedef void (*range_fn)(u32 lo, u32 hi, void *arg);
typedef int (*gpu_fn)(u32 lo, u32 hi, void *arg);

#define GRAIN 512u   /* items a thread takes at a time */

typedef struct {
	range_fn pack, unpack;
	void *arg;
	u32 n, ngrains, chunk_grains, nchunks;
	u32 pack_next, unpack_next;   /* next grain to hand out (atomic) */
	u32 *pack_left;               /* per chunk: grains not yet packed (atomic) */
	u32 *gpu_done;                /* per chunk: back from the GPU (atomic, set under mu) */
	u32 unpack_done;              /* grains unpacked (atomic) */
	u32 abort;
	int active;                   /* pool threads inside this job (g_pool.mu) */
	pthread_mutex_t mu;           /* the OWNER's waits: a chunk is packed / everything is unpacked */
	pthread_cond_t cv;
} pipe_job;

/* Round 6: the pool serves SEVERAL jobs at a time (two application threads inside ec_verify_batch: one call's packing runs on the pool
 * while the other call has the GPU), so a pool thread never blocks inside a job: it takes what grain of work any registered job has,
 * and sleeps on the pool's own condition when none has any; whoever creates work (a job registered, a chunk back from the GPU) kicks
 * the pool. */
#define POOL_JOBS 4
static struct {
	pthread_mutex_t mu;
	pthread_cond_t cv_work, cv_idle;
	pthread_t th[256];
	int nth;
	pipe_job *jobs[POOL_JOBS];
	unsigned long gen;
	int stop;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, {NULL}, 0, 0};

static void grain_range(const pipe_job *j, u32 g, u32 *lo, u32 *hi)
{
	*lo = g * GRAIN;
	*hi = (*lo + GRAIN < j->n) ? *lo + GRAIN : j->n;
}

static void job_notify(pipe_job *j)
{
	pthread_mutex_lock(&j->mu);
	pthread_cond_broadcast(&j->cv);
	pthread_mutex_unlock(&j->mu);
}

static void pool_kick(void)
{
	pthread_mutex_lock(&g_pool.mu);
	g_pool.gen++;
	pthread_cond_broadcast(&g_pool.cv_work);
	pthread_mutex_unlock(&g_pool.mu);
}

/* take one pack grain if any is left; returns 0 when none was */
static int take_pack(pipe_job *j)
{
	u32 g, lo, hi;
	if (AT_LOAD(&j->abort) || AT_LOAD(&j->pack_next) >= j->ngrains) {
		return 0;
	}
	g = AT_ADD(&j->pack_next, 1) - 1;
	if (g >= j->ngrains) {
		return 0;
	}
	grain_range(j, g, &lo, &hi);
	j->pack(lo, hi, j->arg);
	if (AT_SUB(&j->pack_left[g / j->chunk_grains], 1) == 0) {
		job_notify(j);   /* the chunk is ready for the GPU */
	}
	return 1;
}

/* take one unpack grain of a chunk that is back from the GPU; returns 0 when none is available now */
static int take_unpack(pipe_job *j)
{
	u32 g, lo, hi;
	if (!j->unpack || AT_LOAD(&j->abort)) {
		return 0;
	}
	g = AT_LOAD(&j->unpack_next);
	while (g < j->ngrains && AT_LOAD(&j->gpu_done[g / j->chunk_grains])) {
		if (__atomic_compare_exchange_n(&j->unpack_next, &g, g + 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
			grain_range(j, g, &lo, &hi);
			j->unpack(lo, hi, j->arg);
			if (AT_ADD(&j->unpack_done, 1) >= j->ngrains) {
				job_notify(j);   /* the owner: everything is unpacked */
			}
			return 1;
		}
	}
	return 0;
}

static void *pool_worker(void *unused)
{
	unsigned long seen = 0;
	(void)unused;
	pthread_mutex_lock(&g_pool.mu);
	for (;;) {
		int k, did;
		while (!g_pool.stop && g_pool.gen == seen) {
			pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
		}
		if (g_pool.stop) {
			break;
		}
		seen = g_pool.gen;
		do {
			did = 0;
			for (k = 0; k < POOL_JOBS; k++) {
				pipe_job *j = g_pool.jobs[k];
				if (!j) {
					continue;
				}
				j->active++;
				pthread_mutex_unlock(&g_pool.mu);
				while (take_unpack(j) || take_pack(j)) {
					did = 1;
				}
				pthread_mutex_lock(&g_pool.mu);
				if (--j->active == 0) {
					pthread_cond_broadcast(&g_pool.cv_idle);
				}
			}
		} while (did && !g_pool.stop);
	}
	pthread_mutex_unlock(&g_pool.mu);
	return NULL;
}

static int pool_start(int nthreads)
{
	int t;
	if (nthreads > 255) {
		nthreads = 255;
	}
	pthread_mutex_lock(&g_pool.mu);
	g_pool.stop = 0;
	pthread_mutex_unlock(&g_pool.mu);
	for (t = 0; t < nthreads; t++) {
		if (pthread_create(&g_pool.th[g_pool.nth], NULL, pool_worker, NULL)) {
			break;
		}
		g_pool.nth++;
	}
	return (nthreads > 0 && g_pool.nth == 0) ? -1 : 0;
}

static void pool_stop(void)
{
	int t;
	pthread_mutex_lock(&g_pool.mu);
	g_pool.stop = 1;
	pthread_cond_broadcast(&g_pool.cv_work);
	pthread_mutex_unlock(&g_pool.mu);
	for (t = 0; t < g_pool.nth; t++) {
		pthread_join(g_pool.th[t], NULL);
	}
	g_pool.nth = 0;
}

/* a job joins / leaves the pool's list; leaving waits until no pool thread is inside it any more */
static void pool_register(pipe_job *j)
{
	int k, placed = 0;
	pthread_mutex_lock(&g_pool.mu);
	while (!placed) {
		for (k = 0; k < POOL_JOBS && !placed; k++) {
			if (!g_pool.jobs[k]) {
				g_pool.jobs[k] = j;
				placed = 1;
			}
		}
		if (!placed) {
			pthread_cond_wait(&g_pool.cv_idle, &g_pool.mu);   /* (more concurrent jobs than slots: wait for one to leave) */
		}
	}
	g_pool.gen++;
	pthread_cond_broadcast(&g_pool.cv_work);
	pthread_mutex_unlock(&g_pool.mu);
}
static void pool_unregister(pipe_job *j)
{
	int k;
	pthread_mutex_lock(&g_pool.mu);
	for (k = 0; k < POOL_JOBS; k++) {
		if (g_pool.jobs[k] == j) {
			g_pool.jobs[k] = NULL;
		}
	}
	while (j->active > 0) {
		pthread_cond_wait(&g_pool.cv_idle, &g_pool.mu);
	}
	pthread_cond_broadcast(&g_pool.cv_idle);   /* a slot is free */
	pthread_mutex_unlock(&g_pool.mu);
}

/*
 * Run n items through pack -> gpu -> unpack in chunks of `chunk` items.  pack / unpack (either may be NULL) are called on
 * ranges of at most GRAIN items from the pool threads (and from the caller while it waits); gpu (may be NULL) is called by the
 * CALLING thread once per chunk, in order, when the chunk is packed; a chunk is unpacked once its gpu call has returned.
 * Returns 0, or -1 as soon as a gpu call fails.  The caller is inside call_enter() .. call_leave().
 */
/* $ECAMD_COMPAT_TIMING: one line per pipeline run on stderr -- where the calling thread's time went (waiting for the pool to pack a
 * chunk | inside the GPU entry point, copies included | the rest: helping to pack / unpack, draining) */
static double now_ms(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

/* Every entry into the C ABI from a pipeline holds this lock: the producer hook is state of the multi-GPU context, so "install the
 * hook, make the call, remove the hook" must not interleave with another application thread's call (which would be handed a hook whose
 * job is gone); the devices run one call at a time anyway (ecamd_multi's own lock). */
static pthread_mutex_t g_gpu_mu = PTHREAD_MUTEX_INITIALIZER;

/* the owner waits until chunk c is packed, helping meanwhile */
static void wait_packed(pipe_job *J, u32 c)
{
	while (AT_LOAD(&J->pack_left[c]) != 0) {
		if (take_pack(J)) {
			continue;
		}
		pthread_mutex_lock(&J->mu);
		while (AT_LOAD(&J->pack_left[c]) != 0) {
			pthread_cond_wait(&J->cv, &J->mu);
		}
		pthread_mutex_unlock(&J->mu);
	}
}

/* the producer hook of a streamed run (ecamd_multi_set_host_ready_hook): the GPU call is about to read items [first, first + count) of
 * the packed arrays -- wait until the pool has packed them, helping meanwhile.  May be entered from several rank threads. */
static void stream_ready(void *arg, u32 first, u32 count)
{
	pipe_job *J = (pipe_job *)arg;
	const u32 per = J->chunk_grains * GRAIN;
	u32 c, c1;
	if (count == 0 || first >= J->n) {
		return;
	}
	c1 = (first + count - 1 < J->n ? first + count - 1 : J->n - 1) / per;
	for (c = first / per; c <= c1 && c < J->nchunks; c++) {
		wait_packed(J, c);
	}
}

/* streamed != 0: ONE gpu call over all n items, started at once; the C ABI asks (stream_ready) for every range of the packed arrays
 * before it copies it to the device, so the pool packs the tail of the batch while the device works on its head, and `chunk` is only
 * the granularity of that handshake.  (Round 4: a 2^20-item verification spent 4 ms packing before the first copy started, and
 * quarter-batch GPU calls to hide that cost more than they hid -- profiles/r4i_typed_boundary.md.) */
static int pipeline_run_ex(u32 n, u32 chunk, range_fn pack, gpu_fn gpu, range_fn unpack, void *arg, int streamed);
static int pipeline_run(u32 n, u32 chunk, range_fn pack, gpu_fn gpu, range_fn unpack, void *arg)
{
	return pipeline_run_ex(n, chunk, pack, gpu, unpack, arg, 0);
}

static int pipeline_run_ex(u32 n, u32 chunk, range_fn pack, gpu_fn gpu, range_fn unpack, void *arg, int streamed)
{
	pipe_job J;
	u32 c, lo, hi;
	int ret = 0;
	static int timing = -1;
	double t_start = 0, t_wait = 0, t_gpu = 0, t0;
	if (timing < 0) {
		timing = getenv("ECAMD_COMPAT_TIMING") ? 1 : 0;
	}
	t_start = timing ? now_ms() : 0;
	if (n == 0) {
		return 0;
	}
	if (!gpu || chunk >= n) {
		chunk = n;
	}
	chunk = ((chunk + GRAIN - 1) / GRAIN) * GRAIN;
	if (g_pool.nth == 0 || n <= GRAIN) {
		for (lo = 0; lo < n && !ret; lo += chunk) {
			hi = (lo + chunk < n) ? lo + chunk : n;
			if (pack) {
				pack(lo, hi, arg);
			}
			if (gpu) {
				pthread_mutex_lock(&g_gpu_mu);
				ret = gpu(lo, hi, arg) ? -1 : 0;
				pthread_mutex_unlock(&g_gpu_mu);
			}
			if (!ret && unpack) {
				unpack(lo, hi, arg);
			}
		}
		return ret;
	}
	memset(&J, 0, sizeof(J));
	J.pack = pack;
	J.unpack = unpack;
	J.arg = arg;
	J.n = n;
	J.ngrains = (n + GRAIN - 1) / GRAIN;
	J.chunk_grains = chunk / GRAIN;
	J.nchunks = (J.ngrains + J.chunk_grains - 1) / J.chunk_grains;
	J.pack_left = (u32 *)calloc(J.nchunks, sizeof(u32));
	J.gpu_done = (u32 *)calloc(J.nchunks, sizeof(u32));
	if (!J.pack_left || !J.gpu_done) {
		free(J.pack_left);
		free(J.gpu_done);
		return -1;
	}
	for (c = 0; c < J.nchunks; c++) {
		const u32 g0 = c * J.chunk_grains, g1 = (g0 + J.chunk_grains < J.ngrains) ? g0 + J.chunk_grains : J.ngrains;
		J.pack_left[c] = pack ? (g1 - g0) : 0;
	}
	if (!pack) {
		J.pack_next = J.ngrains;
	}
	pthread_mutex_init(&J.mu, NULL);
	pthread_cond_init(&J.cv, NULL);
	pool_register(&J);
	if (streamed && gpu && pack && J.nchunks > 1) {
		t0 = timing ? now_ms() : 0;
		pthread_mutex_lock(&g_gpu_mu);   /* (another application thread may have the GPU: this call's packing goes on meanwhile, on the pool) */
		if (timing) {
			t_wait += now_ms() - t0;
			t0 = now_ms();
		}
		if (ecamd_multi_set_host_ready_hook(g_multi, stream_ready, &J)) {
			ret = -1;
		} else {
			ret = gpu(0, n, arg) ? -1 : 0;
			(void)ecamd_multi_set_host_ready_hook(g_multi, NULL, NULL);
		}
		pthread_mutex_unlock(&g_gpu_mu);
		if (ret) {
			AT_STORE(&J.abort, 1);
			job_notify(&J);
		} else {
			stream_ready(&J, 0, n);   /* (a call that returned without reading everything: nothing may stay unpacked) */
			pthread_mutex_lock(&J.mu);
			for (c = 0; c < J.nchunks; c++) {
				AT_STORE(&J.gpu_done[c], 1);
			}
			pthread_mutex_unlock(&J.mu);
			pool_kick();
		}
		if (timing) {
			t_gpu += now_ms() - t0;
		}
	}
	for (c = 0; c < J.nchunks && !(streamed && gpu && pack && J.nchunks > 1); c++) {
		/* wait for chunk c to be packed; help meanwhile */
		t0 = timing ? now_ms() : 0;
		wait_packed(&J, c);
		lo = c * J.chunk_grains * GRAIN;
		hi = (lo + chunk < n) ? lo + chunk : n;
		if (timing) {
			t_wait += now_ms() - t0;
			t0 = now_ms();
		}
		if (gpu) {
			int r;
			pthread_mutex_lock(&g_gpu_mu);
			r = gpu(lo, hi, arg);
			pthread_mutex_unlock(&g_gpu_mu);
			if (r) {
				ret = -1;
				AT_STORE(&J.abort, 1);
				job_notify(&J);
				break;
			}
		}
		if (timing) {
			t_gpu += now_ms() - t0;
		}
		AT_STORE(&J.gpu_done[c], 1);
		if (unpack) {
			pool_kick();
		}
	}
	if (!ret && unpack) {
		/* help with what is left to unpack, then wait for the grains other threads still hold */
		for (;;) {
			if (take_unpack(&J)) {
				continue;
			}
			if (AT_LOAD(&J.unpack_done) >= J.ngrains) {
				break;
			}
			pthread_mutex_lock(&J.mu);
			if (AT_LOAD(&J.unpack_done) < J.ngrains && AT_LOAD(&J.unpack_next) >= J.ngrains) {
				struct timespec ts;
				clock_gettime(CLOCK_REALTIME, &ts);
				ts.tv_nsec += 200000;
				if (ts.tv_nsec >= 1000000000L) {
					ts.tv_sec++;
					ts.tv_nsec -= 1000000000L;
				}
				(void)pthread_cond_timedwait(&J.cv, &J.mu, &ts);
			}
			pthread_mutex_unlock(&J.mu);
		}
	}
	pool_unregister(&J);   /* (also on failure: pool threads inside a pack / unpack grain finish it first) */
	pthread_mutex_destroy(&J.mu);
	pthread_cond_destroy(&J.cv);
	free(J.pack_left);
	free(J.gpu_done);
	if (timing) {
		const double tot = now_ms() - t_start;
		fprintf(stderr, "libecc_amd compat timing: %u items, %u chunks of %u, %d threads: total %.2f ms = pack wait %.2f + gpu entry point %.2f + rest %.2f\n",
			n, J.nchunks, chunk, g_pool.nth + 1, tot, t_wait, t_gpu, tot - t_wait - t_gpu);
	}
	return ret;
}

static void parallel_for(u32 n, range_fn fn, void *arg)
{
	(void)pipeline_run(n, n, fn, NULL, NULL, arg);
}

/* Items per pipeline chunk.  Every chunk costs the device side a fixed few hundred microseconds (launches, staging, one
 * synchronisation), so chunks should be large; the overlap of packing with the GPU needs several of them.  Measured on 2^20
 * ECDSA verifications (profiles/r3b_compat_end_to_end.md): 12.8 / 22.4 / 27.9 / 28.8 M/s at 2^15 / 2^16 / 2^17 / 2^18 items per
 * chunk.  Default: a quarter of the batch per device, between 2^15 and 2^18 ($ECAMD_COMPAT_CHUNK fixes it); the multi-GPU
 * layer cuts every chunk into one shard per device. */
/* Verification (round 4, measured: profiles/r4i_typed_boundary.md): quarter-batch GPU calls cost more than the packing they hide -- the
 * kernels of a 2^18-item launch run at 0.8 - 0.9 of their 2^20 rate and every call is copy-in, kernels, copy-out in sequence -- so a
 * verification is ONE call per batch (up to 2^20 items per device) and the packing overlaps it through the C ABI's producer hook
 * (pipeline_run_ex, streamed).  $ECAMD_COMPAT_NO_STREAM: one call after all packing, or chunked calls of $ECAMD_COMPAT_CHUNK items. */
#define READY_ITEMS (1u << 16)   /* granularity of the producer handshake ($ECAMD_COMPAT_READY_ITEMS: for tests of small batches) */
static u32 ready_items(void)
{
	static u32 v;
	if (!v) {
		const char *e = getenv("ECAMD_COMPAT_READY_ITEMS");
		const unsigned long x = e ? strtoul(e, NULL, 10) : 0;
		v = (x >= GRAIN && x <= (1u << 24)) ? (u32)x : READY_ITEMS;
	}
	return v;
}
static int verify_streamed(void)
{
	static int on = -1;
	if (on < 0) {
		on = getenv("ECAMD_COMPAT_NO_STREAM") ? 0 : 1;
	}
	return on;
}
static u32 chunk_items_verify(u32 n)
{
	const int nd = g_multi ? ecamd_multi_size(g_multi) : 1;
	u64 c;
	if (g_chunk) {
		c = (u64)g_chunk * (u64)(nd > 0 ? nd : 1);
	} else {
		c = (u64)(1u << 20) * (u64)(nd > 0 ? nd : 1);
	}
	(void)n;
	return c > 0x40000000ull ? 0x40000000u : (u32)c;
}
/* pack | GPU | unpack of a verification group of cnt items */
static int verify_pipeline(u32 cnt, range_fn pack, gpu_fn gpu, range_fn unpack, void *arg)
{
	const u32 per_call = chunk_items_verify(cnt);
	if (!verify_streamed()) {
		return pipeline_run(cnt, per_call, pack, gpu, unpack, arg);
	}
	if (cnt <= per_call) {
		return pipeline_run_ex(cnt, ready_items(), pack, gpu, unpack, arg, 1);
	}
	return pipeline_run(cnt, per_call, pack, gpu, unpack, arg);   /* (more than 2^20 items per device: chunked calls of that size) */
}

static u32 chunk_items_for(u32 per_device, u32 n)
{
	const int nd = g_multi ? ecamd_multi_size(g_multi) : 1;
	u64 c;
	if (per_device == 0) {
		const u32 per_dev_n = n / (u32)(nd > 0 ? nd : 1);
		per_device = per_dev_n / 4;
		if (per_device < (1u << 15)) {
			per_device = 1u << 15;
		}
		if (per_device > (1u << 18)) {
			per_device = 1u << 18;
		}
	}
	c = (u64)per_device * (u64)(nd > 0 ? nd : 1);
	return c > 0x40000000ull ? 0x40000000u : (u32)c;
}

/* ------------------------------------------------------------------------------------------------
 * staging buffers: page-locked, kept across calls, one set per concurrent call
 * ------------------------------------------------------------------------------------------------ */
#define NBUF 16
/* Round 6: staging is per CALL, not per process.  A batch call works in an arena -- its sixteen page-locked buffers, kept across
 * calls -- that it holds from call_enter() to call_leave(); there are NARENA of them, so that many application threads can be inside
 * the verification entry points at once (the next one waits).  Everything else -- the secret-key half, the multi-step Schnorr path,
 * set-up and shutdown -- enters EXCLUSIVELY, as under the process-wide lock of rounds 2-5: those paths switch the devices between
 * secret- and public-scalar mode, wipe scratch, or make several dependent GPU calls in a row. */
#define NARENA 2
typedef struct {
	u8 *buf[NBUF];
	size_t cap[NBUF], used[NBUF];   /* used: bytes handed out since the last wipe */
	u8 pinned[NBUF];
	int busy;
} arena;
static arena g_arena[NARENA];
static __thread arena *t_arena;     /* the arena of the batch call this thread is inside */
static __thread int t_shared;
static pthread_mutex_t g_gate_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_gate_cv = PTHREAD_COND_INITIALIZER;
static int g_shared_in, g_excl_in, g_excl_waiting;

static void call_enter(int shared)
{
	int k;
	pthread_mutex_lock(&g_gate_mu);
	if (shared) {
		while (g_excl_in || g_excl_waiting || g_shared_in >= NARENA) {
			pthread_cond_wait(&g_gate_cv, &g_gate_mu);
		}
		for (k = 0; k < NARENA; k++) {
			if (!g_arena[k].busy) {
				break;
			}
		}
		g_arena[k].busy = 1;
		t_arena = &g_arena[k];
		g_shared_in++;
	} else {
		g_excl_waiting++;
		while (g_excl_in || g_shared_in) {
			pthread_cond_wait(&g_gate_cv, &g_gate_mu);
		}
		g_excl_waiting--;
		g_excl_in = 1;
		g_arena[0].busy = 1;
		t_arena = &g_arena[0];
	}
	t_shared = shared;
	pthread_mutex_unlock(&g_gate_mu);
}

static void call_leave(void)
{
	pthread_mutex_lock(&g_gate_mu);
	if (t_arena) {
		t_arena->busy = 0;
	}
	if (t_shared) {
		g_shared_in--;
	} else {
		g_excl_in = 0;
	}
	t_arena = NULL;
	pthread_cond_broadcast(&g_gate_cv);
	pthread_mutex_unlock(&g_gate_mu);
}

static u8 *buf_get(int k, size_t bytes)
{
	arena *A = t_arena;
	if (!A) {
		return NULL;   /* (a batch path outside call_enter / call_leave: a bug, reported as an allocation failure) */
	}
	if (bytes == 0) {
		bytes = 1;
	}
	if (A->cap[k] < bytes) {
		const size_t want = bytes + bytes / 8;
		if (A->buf[k]) {
			wipe(A->buf[k], A->cap[k]);
			if (A->pinned[k]) {
				ecamd_host_free(A->buf[k]);
			} else {
				free(A->buf[k]);
			}
		}
		A->buf[k] = (u8 *)ecamd_host_alloc(want);
		A->pinned[k] = A->buf[k] != NULL;
		if (!A->buf[k]) {
			A->buf[k] = (u8 *)malloc(want);   /* pageable memory works too; it is only slower */
		}
		A->cap[k] = A->buf[k] ? want : 0;
		A->used[k] = 0;
	}
	if (A->buf[k] && bytes > A->used[k]) {
		A->used[k] = bytes;
	}
	return A->buf[k];
}

static void bufs_free(void)
{
	int a, k;
	for (a = 0; a < NARENA; a++) {
		arena *A = &g_arena[a];
		for (k = 0; k < NBUF; k++) {
			if (A->buf[k]) {
				wipe(A->buf[k], A->cap[k]);
				if (A->pinned[k]) {
					ecamd_host_free(A->buf[k]);
				} else {
					free(A->buf[k]);
				}
			}
			A->buf[k] = NULL;
			A->cap[k] = 0;
			A->used[k] = 0;
		}
	}
}

/* after a call that handled private material: host staging and the devices' scratch */
/* (on the pool: after a 2^20-item signing call the staging holds some 200 MB, which one thread takes ten milliseconds to clear) */
#define WIPE_BLOCK 4096u
typedef struct {
	arena *A;
	u32 first[NBUF + 1];   /* first block of buffer k in the concatenation of all used staging */
} wipe_job;
static void wipe_blocks(u32 lo, u32 hi, void *arg)
{
	const wipe_job *W = (const wipe_job *)arg;
	int k;
	for (k = 0; k < NBUF; k++) {
		const u32 b0 = lo > W->first[k] ? lo : W->first[k], b1 = hi < W->first[k + 1] ? hi : W->first[k + 1];
		if (b0 < b1) {
			const size_t off = (size_t)(b0 - W->first[k]) * WIPE_BLOCK;
			size_t len = (size_t)(b1 - b0) * WIPE_BLOCK;
			if (off + len > W->A->used[k]) {
				len = W->A->used[k] - off;
			}
			wipe(W->A->buf[k] + off, len);
		}
	}
}
static void wipe_secrets(void)
{
	wipe_job W;
	int k;
	W.A = t_arena;
	if (!W.A) {
		return;
	}
	W.first[0] = 0;
	for (k = 0; k < NBUF; k++) {
		const size_t used = W.A->buf[k] ? W.A->used[k] : 0;
		W.first[k + 1] = W.first[k] + (u32)((used + WIPE_BLOCK - 1) / WIPE_BLOCK);
	}
	if (W.first[NBUF]) {
		parallel_for(W.first[NBUF], wipe_blocks, &W);
	}
	for (k = 0; k < NBUF; k++) {
		W.A->used[k] = 0;
	}
	if (g_multi && ecamd_multi_wipe_scratch(g_multi)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
	}
}

/* ------------------------------------------------------------------------------------------------
 * marshalling: the live limbs of nn / fp / prj_pt  <->  big-endian octets
 * ------------------------------------------------------------------------------------------------ */
/* `len` octets, big-endian, of the number held in val[] (little-endian words of WORD_BYTES bytes; nn/nn.h:67-71) */
static inline void limbs_to_be(u8 *dst, u32 len, const word_t *val)
{
	u32 i;
#if WORDSIZE == 64
	if ((len % 8) == 0) {
		const u32 nl = len / 8;
		for (i = 0; i < nl; i++) {
			const u64 w = __builtin_bswap64((u64)val[nl - 1 - i]);
			memcpy(dst + 8 * i, &w, 8);
		}
		return;
	}
#endif
	for (i = 0; i < len; i++) {
		const u32 k = len - 1 - i;
		dst[i] = (u8)(val[k / WORD_BYTES] >> (8 * (k % WORD_BYTES)));
	}
}

/* nn -> len octets; -1 for an uninitialised nn or a value that does not fit (nn_export_to_buf would silently truncate) */
static int nn_to_be(u8 *dst, u32 len, nn_src_t a)
{
	const u32 nl = (len + WORD_BYTES - 1) / WORD_BYTES;
	word_t extra = 0;
	u32 i;
	if (nn_check_initialized(a)) {
		return -1;
	}
	for (i = nl; i < a->wlen; i++) {
		extra |= a->val[i];
	}
	if ((len % WORD_BYTES) && nl >= 1) {
		extra |= a->val[nl - 1] >> (8 * (len % WORD_BYTES));
	}
	if (extra) {
		return -1;
	}
	limbs_to_be(dst, len, a->val);
	return 0;
}

/* X || Y || Z of an initialised point of curve `crv` (what prj_pt_export_to_buf writes, curves/prj_pt.c:562, without its
 * on-curve test: the device checks the curve equation when it imports the point) */
static int prj_to_be(u8 *dst, u32 clen, prj_pt_src_t P, ec_shortw_crv_src_t crv)
{
	if (prj_pt_check_initialized(P) || P->crv != crv || fp_check_initialized(&P->X) || fp_check_initialized(&P->Y) ||
	    fp_check_initialized(&P->Z)) {
		return -1;
	}
	limbs_to_be(dst, clen, P->X.fp_val.val);
	limbs_to_be(dst + clen, clen, P->Y.fp_val.val);
	limbs_to_be(dst + 2 * clen, clen, P->Z.fp_val.val);
	return 0;
}

/* affine X || Y from the device -> (X : Y : 1); ECAMD_INF -> (0 : 1 : 0).  The device's results lie on the curve, so the
 * on-curve test of prj_pt_import_from_buf is not repeated; coordinates are still checked to be < p (fp_import_from_buf). */
static int prj_from_aff_be(prj_pt *out, ec_shortw_crv_src_t crv, const u8 *aff, u32 clen, u8 st)
{
	if (st == ECAMD_OK) {
		return (prj_pt_init(out, crv) || fp_import_from_buf(&out->X, aff, (u16)clen) || fp_import_from_buf(&out->Y, aff + clen, (u16)clen) ||
			fp_one(&out->Z)) ? -1 : 0;
	}
	if (st == ECAMD_INF) {
		return (prj_pt_init(out, crv) || prj_pt_zero(out)) ? -1 : 0;
	}
	return -1;
}

/* ------------------------------------------------------------------------------------------------
 * prj_pt_mul_batch
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const nn *m;
	const prj_pt *in;
	prj_pt *out;
	int *ret_items;
	const u32 *idx;          /* items of this group (NULL: all, in order) */
	u8 *sc, *pin, *pout, *st, *pre;
	u32 slen, clen;
	ec_shortw_crv_src_t crv;
	curve_ent *e;
} mul_job;

static void mul_pack(u32 lo, u32 hi, void *arg)
{
	mul_job *J = (mul_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx ? J->idx[j] : j;
		/* a point of another curve, an uninitialised point or scalar: prj_pt_mul returns -1 (prj_pt.c:1765-1767) */
		if (nn_to_be(J->sc + (size_t)j * J->slen, J->slen, &J->m[i]) || prj_to_be(J->pin + (size_t)j * 3 * J->clen, J->clen, &J->in[i], J->crv)) {
			J->pre[j] = 1;
			memset(J->sc + (size_t)j * J->slen, 0, J->slen);
			memset(J->pin + (size_t)j * 3 * J->clen, 0xff, 3 * J->clen);   /* coordinates >= p: rejected at import */
		} else {
			J->pre[j] = 0;
		}
	}
}

static int mul_gpu(u32 lo, u32 hi, void *arg)
{
	mul_job *J = (mul_job *)arg;
	if (ecamd_multi_prj_pt_mul_batch_fmt(g_multi, J->e->mc, hi - lo, J->sc + (size_t)lo * J->slen, J->slen, J->pin + (size_t)lo * 3 * J->clen,
					     ECAMD_PT_PROJECTIVE, J->pout + (size_t)lo * 2 * J->clen, ECAMD_PT_AFFINE, J->st + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void mul_unpack(u32 lo, u32 hi, void *arg)
{
	mul_job *J = (mul_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx ? J->idx[j] : j;
		const int r = J->pre[j] ? -1 : prj_from_aff_be(&J->out[i], J->crv, J->pout + (size_t)j * 2 * J->clen, J->clen, J->st[j]);
		if (J->ret_items) {
			J->ret_items[i] = r;
		}
	}
}

/* one GPU batch over the items idx[0..cnt) (idx == NULL: all n items) whose scalars fit slen octets */
static int mul_group(mul_job *J, u32 cnt)
{
	J->sc = buf_get(0, (size_t)cnt * J->slen);
	J->pin = buf_get(1, (size_t)cnt * 3 * J->clen);
	J->pout = buf_get(2, (size_t)cnt * 2 * J->clen);
	J->st = buf_get(3, cnt);
	J->pre = buf_get(4, cnt);
	if (!J->sc || !J->pin || !J->pout || !J->st || !J->pre) {
		return -1;
	}
	if (pipeline_run(cnt, chunk_items_for(g_chunk, cnt), mul_pack, mul_gpu, mul_unpack, J)) {
		return -1;
	}
	note_items(cnt);
	return 0;
}

typedef struct {
	const nn *m;
	u32 *bits;
} bits_job;

static void mul_bits(u32 lo, u32 hi, void *arg)
{
	bits_job *B = (bits_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		bitcnt_t b = 0;
		B->bits[i] = nn_bitlen(&B->m[i], &b) ? 0 : (u32)b;
	}
}

static int mul_batch_common(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items, int secret)
{
	mul_job J;
	bits_job B;
	curve_ent *e;
	u32 i, nlong = 0, maxbits = 0, *idx = NULL;
	int ret = -1;
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
	call_enter(0);
	B.m = m;
	B.bits = (u32 *)malloc((size_t)n * sizeof(u32));
	if (!B.bits) {
		goto done;
	}
	parallel_for(n, mul_bits, &B);
	/* Scalars of up to the order's length run on the fast window kernels; a few longer ones (m >= 2^(8 qlen), e.g. blinded
	 * scalars mixed into a batch) must not drag the whole batch onto the long-scalar kernel: they go in a second batch. */
	for (i = 0; i < n; i++) {
		if (B.bits[i] > 8 * e->qlen) {
			nlong++;
			if (B.bits[i] > maxbits) {
				maxbits = B.bits[i];
			}
		}
	}
	J.m = m;
	J.in = in;
	J.out = out;
	J.ret_items = ret_items;
	J.crv = in[0].crv;
	J.clen = e->clen;
	J.e = e;
	if (nlong == 0 || nlong == n) {
		J.slen = nlong ? (u32)BYTECEIL(maxbits) : e->qlen;
		ret = mul_group(&J, n);
	} else {
		u32 ns = 0, nl = 0;
		idx = (u32 *)malloc((size_t)n * sizeof(u32));
		if (!idx) {
			goto done;
		}
		for (i = 0; i < n; i++) {   /* short ones first, long ones behind them */
			if (B.bits[i] <= 8 * e->qlen) {
				idx[ns++] = i;
			}
		}
		for (i = 0; i < n; i++) {
			if (B.bits[i] > 8 * e->qlen) {
				idx[ns + nl++] = i;
			}
		}
		J.idx = idx;
		J.slen = e->qlen;
		ret = mul_group(&J, ns);
		if (!ret) {
			J.idx = idx + ns;
			J.slen = (u32)BYTECEIL(maxbits);
			ret = mul_group(&J, nl);
		}
	}
done:
	if (secret) {
		wipe_secrets();
	}
	call_leave();
	free(B.bits);
	free(idx);
	return ret;
}

int prj_pt_mul_batch(prj_pt *out, const nn *m, const prj_pt *in, u32 n, int *ret_items)
{
	/* prj_pt_mul is the reference's protected multiplication (masked ladder): its scalars are treated as secret */
	return mul_batch_common(out, m, in, n, ret_items, 1);
}

typedef struct {
	const nn *m;
	nn *mb;
	nn_src_t order;
	u32 failed;
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
		if (compat_random_mod(&b, B->order) || nn_mul(&b, &b, B->order) || nn_add(&B->mb[i], &B->m[i], &b)) {
			AT_STORE(&B->failed, 1);
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
	call_enter(0);
	parallel_for(n, blind_scalars, &B);
	call_leave();
	ret = AT_LOAD(&B.failed) ? -1 : mul_batch_common(out, B.mb, in, n, ret_items, 1);
	wipe(B.mb, (size_t)n * sizeof(nn));
	free(B.mb);
	return ret;
}

/* ------------------------------------------------------------------------------------------------
 * Round 4: the group law, the normalisation, the on-curve test and the public-scalar multiplication in libecc's own types
 * (prj_pt_add / prj_pt_dbl / prj_pt_unique / prj_pt_is_on_curve -- round 6: prj_pt_neg / prj_pt_cmp / prj_pt_eq_or_opp --, curves/prj_pt.h:42-86; _prj_pt_unprotected_mult and
 * check_prj_pt_order, curves/prj_pt.c:1835-1945) and ec_pub_key_import_from_aff_buf (sig/ec_key.c:181).
 * One job shape serves all of them: inputs as projective X || Y || Z (or the caller's affine buffers), results as affine
 * X || Y + status from the device (ec_prj_pt_op_batch_fmt / ec_prj_pt_unprotected_mult_batch of libecc_amd.h).
 * ------------------------------------------------------------------------------------------------ */
enum { PTOP_ADD = 0, PTOP_DBL, PTOP_ON_CURVE, PTOP_UNIQUE, PTOP_UMULT, PTOP_ORDER, PTOP_PUBIMPORT, PTOP_NEG, PTOP_CMP, PTOP_EQ_OR_OPP };
/* the operations with a second point */
#define PTOP_TWO(op) ((op) == PTOP_ADD || (op) == PTOP_CMP || (op) == PTOP_EQ_OR_OPP)
typedef struct {
	int op;
	const prj_pt *in1, *in2;
	const nn *m;                 /* PTOP_UMULT: per-item scalars */
	const u8 *const *bufs;       /* PTOP_PUBIMPORT: affine buffers */
	u32 buf_len;
	prj_pt *out;
	ec_pub_key *pubs;
	const ec_params *params;
	ec_alg_type alg;
	int *ret_items, *flags;      /* flags: on_curve (PTOP_ON_CURVE) / check (PTOP_ORDER) / cmp (PTOP_CMP) / eq_or_opp (PTOP_EQ_OR_OPP) */
	int need_order;              /* PTOP_PUBIMPORT on a cofactor curve: the subgroup test of ec_pub_key_import_from_aff_buf */
	u8 *b1, *b2, *sc, *pout, *st, *pre;
	u8 bsc[NN_MAX_BYTE_LEN];     /* the one scalar of PTOP_ORDER / PTOP_PUBIMPORT */
	u32 slen, clen, iw;
	int in_fmt;
	ec_shortw_crv_src_t crv;
	curve_ent *e;
} ptop_job;

static void ptop_pack(u32 lo, u32 hi, void *arg)
{
	ptop_job *J = (ptop_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		int bad;
		if (J->op == PTOP_PUBIMPORT) {
			/* prj_pt_import_from_aff_buf wants exactly 2 * BYTECEIL(|p|) octets (curves/prj_pt.c:520-526) */
			bad = !J->bufs[i] || J->buf_len != 2 * J->clen;
			if (!bad) {
				memcpy(J->b1 + (size_t)i * J->iw, J->bufs[i], J->iw);
			}
		} else {
			bad = prj_to_be(J->b1 + (size_t)i * J->iw, J->clen, &J->in1[i], J->crv);
			if (!bad && PTOP_TWO(J->op)) {
				/* MUST_HAVE((in1->crv == in2->crv)), curves/prj_pt.c:1210, :313, :418 */
				bad = prj_to_be(J->b2 + (size_t)i * J->iw, J->clen, &J->in2[i], J->crv);
			}
			if (!bad && J->op == PTOP_UMULT) {
				bad = nn_to_be(J->sc + (size_t)i * J->slen, J->slen, &J->m[i]);
			}
		}
		J->pre[i] = bad ? 1 : 0;
		if (bad) {
			memset(J->b1 + (size_t)i * J->iw, 0xff, J->iw);   /* coordinates >= p: rejected at import */
			if (PTOP_TWO(J->op)) {
				memset(J->b2 + (size_t)i * J->iw, 0xff, J->iw);
			}
			if (J->op == PTOP_UMULT) {
				memset(J->sc + (size_t)i * J->slen, 0, J->slen);
			}
		}
	}
}

static int ptop_gpu(u32 lo, u32 hi, void *arg)
{
	ptop_job *J = (ptop_job *)arg;
	const u32 m = hi - lo;
	u8 *pout = J->pout + (size_t)lo * 2 * J->clen;
	int r;
	switch (J->op) {
	case PTOP_ADD:
	case PTOP_DBL:
	case PTOP_ON_CURVE:
		r = ecamd_multi_prj_pt_op_batch_fmt(g_multi, J->e->mc, J->op == PTOP_ADD ? ECAMD_PT_OP_ADD : (J->op == PTOP_DBL ? ECAMD_PT_OP_DBL : ECAMD_PT_OP_ON_CURVE),
						    m, J->b1 + (size_t)lo * J->iw, J->op == PTOP_ADD ? J->b2 + (size_t)lo * J->iw : NULL, J->in_fmt, pout,
						    ECAMD_PT_AFFINE, J->st + lo);
		break;
	case PTOP_NEG:
		r = ecamd_multi_prj_pt_op_batch_fmt(g_multi, J->e->mc, ECAMD_PT_OP_NEG, m, J->b1 + (size_t)lo * J->iw, NULL, J->in_fmt, pout, ECAMD_PT_AFFINE,
						    J->st + lo);
		break;
	case PTOP_CMP:
	case PTOP_EQ_OR_OPP:
		/* one predicate byte per item: J->pout is used with a stride of one */
		r = ecamd_multi_prj_pt_op_batch_fmt(g_multi, J->e->mc, J->op == PTOP_CMP ? ECAMD_PT_OP_CMP : ECAMD_PT_OP_EQ_OR_OPP, m, J->b1 + (size_t)lo * J->iw,
						    J->b2 + (size_t)lo * J->iw, J->in_fmt, J->pout + lo, ECAMD_PT_AFFINE, J->st + lo);
		break;
	case PTOP_UNIQUE:
		r = ecamd_multi_prj_pt_unique_batch(g_multi, J->e->mc, m, J->b1 + (size_t)lo * J->iw, J->in_fmt, pout, ECAMD_PT_AFFINE, J->st + lo);
		break;
	case PTOP_UMULT:
		r = ecamd_multi_prj_pt_unprotected_mult_batch(g_multi, J->e->mc, m, J->sc + (size_t)lo * J->slen, J->slen, J->slen, J->b1 + (size_t)lo * J->iw,
							      J->in_fmt, pout, ECAMD_PT_AFFINE, J->st + lo);
		break;
	case PTOP_PUBIMPORT:
		if (!J->need_order) {
			r = ecamd_multi_prj_pt_op_batch_fmt(g_multi, J->e->mc, ECAMD_PT_OP_ON_CURVE, m, J->b1 + (size_t)lo * J->iw, NULL, J->in_fmt, pout,
							    ECAMD_PT_AFFINE, J->st + lo);
			break;
		}
		/* [q]Y must be the point at infinity */
		/* fall through */
	default:
		r = ecamd_multi_prj_pt_unprotected_mult_batch(g_multi, J->e->mc, m, J->bsc, J->slen, 0, J->b1 + (size_t)lo * J->iw, J->in_fmt, pout,
							      ECAMD_PT_AFFINE, J->st + lo);
		break;
	}
	if (r) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void ptop_unpack(u32 lo, u32 hi, void *arg)
{
	ptop_job *J = (ptop_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const u8 st = J->st[i];
		int r = -1;
		if (J->pre[i]) {
			r = -1;
		} else if (J->op == PTOP_ON_CURVE) {
			J->flags[i] = (st == ECAMD_OK) ? 1 : 0;   /* the call itself succeeds for an initialised point */
			r = 0;
		} else if (J->op == PTOP_CMP || J->op == PTOP_EQ_OR_OPP) {
			if (st == ECAMD_OK) {
				J->flags[i] = J->pout[i];
				r = 0;
			}
		} else if (J->op == PTOP_ORDER) {
			if (st != ECAMD_ERR) {
				J->flags[i] = (st == ECAMD_INF) ? 1 : 0;
				r = 0;
			}
		} else if (J->op == PTOP_PUBIMPORT) {
			ec_pub_key *pub = &J->pubs[i];
			const int ok = J->need_order ? (st == ECAMD_INF) : (st == ECAMD_OK);
			memset(pub, 0, sizeof(ec_pub_key));
			if (ok && !prj_from_aff_be(&pub->y, &(J->params->ec_curve), J->b1 + (size_t)i * J->iw, J->clen, ECAMD_OK)) {
				pub->key_type = J->alg;
				pub->params = J->params;
				pub->magic = PUB_KEY_MAGIC;
				r = 0;
			}
		} else if (J->op == PTOP_UNIQUE) {
			/* prj_pt_unique / prj_pt_to_aff refuse the point at infinity (curves/prj_pt.c:218-241) */
			r = (st == ECAMD_OK) ? prj_from_aff_be(&J->out[i], J->crv, J->pout + (size_t)i * 2 * J->clen, J->clen, st) : -1;
		} else {
			r = prj_from_aff_be(&J->out[i], J->crv, J->pout + (size_t)i * 2 * J->clen, J->clen, st);
		}
		if (J->ret_items) {
			J->ret_items[i] = r;
		}
	}
}

static int ptop_run(ptop_job *J, u32 n)
{
	int ret = -1;
	if (n == 0) {
		return 0;
	}
	if (ecamd_compat_init(NULL, 0, 0)) {
		return -1;
	}
	J->clen = J->e->clen;
	J->in_fmt = (J->op == PTOP_PUBIMPORT) ? ECAMD_PT_AFFINE : ECAMD_PT_PROJECTIVE;
	J->iw = (J->in_fmt ? 3u : 2u) * J->clen;
	call_enter(0);
	J->b1 = buf_get(0, (size_t)n * J->iw);
	J->b2 = PTOP_TWO(J->op) ? buf_get(1, (size_t)n * J->iw) : J->b1;
	J->sc = (J->op == PTOP_UMULT) ? buf_get(2, (size_t)n * J->slen) : J->b1;
	J->pout = buf_get(3, (size_t)n * 2 * J->clen);
	J->st = buf_get(4, n);
	J->pre = buf_get(5, n);
	if (J->b1 && J->b2 && J->sc && J->pout && J->st && J->pre &&
	    !pipeline_run(n, chunk_items_for(g_chunk, n), ptop_pack, ptop_gpu, ptop_unpack, J)) {
		note_items(n);
		ret = 0;
	}
	call_leave();
	return ret;
}

static int ptop_points(ptop_job *J, int op, prj_pt *out, const prj_pt *in1, const prj_pt *in2, u32 n, int *ret_items)
{
	memset(J, 0, sizeof(*J));
	if (!in1 || (PTOP_TWO(op) && !in2) || ((op == PTOP_ADD || op == PTOP_DBL || op == PTOP_UNIQUE || op == PTOP_UMULT || op == PTOP_NEG) && !out)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	if (prj_pt_check_initialized(&in1[0])) {
		return -1;
	}
	J->op = op;
	J->in1 = in1;
	J->in2 = in2;
	J->out = out;
	J->ret_items = ret_items;
	J->crv = in1[0].crv;
	J->e = curve_from_crv(J->crv);
	return J->e ? 1 : -1;
}

int prj_pt_add_batch(prj_pt *out, const prj_pt *in1, const prj_pt *in2, u32 n, int *ret_items)
{
	ptop_job J;
	const int r = ptop_points(&J, PTOP_ADD, out, in1, in2, n, ret_items);
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_dbl_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items)
{
	ptop_job J;
	const int r = ptop_points(&J, PTOP_DBL, out, in, NULL, n, ret_items);
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_neg_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items)
{
	ptop_job J;
	const int r = ptop_points(&J, PTOP_NEG, out, in, NULL, n, ret_items);
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_cmp_batch(const prj_pt *in1, const prj_pt *in2, u32 n, int *cmp, int *ret_items)
{
	ptop_job J;
	int r;
	if (!cmp) {
		return -1;
	}
	r = ptop_points(&J, PTOP_CMP, NULL, in1, in2, n, ret_items);
	J.flags = cmp;
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_eq_or_opp_batch(const prj_pt *in1, const prj_pt *in2, u32 n, int *eq_or_opp, int *ret_items)
{
	ptop_job J;
	int r;
	if (!eq_or_opp) {
		return -1;
	}
	r = ptop_points(&J, PTOP_EQ_OR_OPP, NULL, in1, in2, n, ret_items);
	J.flags = eq_or_opp;
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_unique_batch(prj_pt *out, const prj_pt *in, u32 n, int *ret_items)
{
	ptop_job J;
	const int r = ptop_points(&J, PTOP_UNIQUE, out, in, NULL, n, ret_items);
	return r <= 0 ? r : ptop_run(&J, n);
}

int prj_pt_is_on_curve_batch(const prj_pt *in, u32 n, int *on_curve, int *ret_items)
{
	ptop_job J;
	int r;
	if (!on_curve) {
		return -1;
	}
	r = ptop_points(&J, PTOP_ON_CURVE, NULL, in, NULL, n, ret_items);
	J.flags = on_curve;
	return r <= 0 ? r : ptop_run(&J, n);
}

int _prj_pt_unprotected_mult_batch(prj_pt *out, const nn *scalars, const prj_pt *in, u32 n, int *ret_items)
{
	ptop_job J;
	bits_job B;
	u32 i, maxbits = 0;
	int r;
	if (!scalars) {
		return -1;
	}
	r = ptop_points(&J, PTOP_UMULT, out, in, NULL, n, ret_items);
	if (r <= 0) {
		return r;
	}
	if (ecamd_compat_init(NULL, 0, 0)) {
		return -1;
	}
	/* one octet length for the batch: the longest scalar's */
	B.m = scalars;
	B.bits = (u32 *)malloc((size_t)n * sizeof(u32));
	if (!B.bits) {
		return -1;
	}
	call_enter(0);
	parallel_for(n, mul_bits, &B);
	call_leave();
	for (i = 0; i < n; i++) {
		maxbits = B.bits[i] > maxbits ? B.bits[i] : maxbits;
	}
	free(B.bits);
	J.m = scalars;
	J.slen = maxbits ? (maxbits + 7) / 8 : 1;
	return ptop_run(&J, n);
}

int check_prj_pt_order_batch(const prj_pt *in, nn_src_t in_isorder, prj_pt_sensitivity s, u32 n, int *check, int *ret_items)
{
	ptop_job J;
	bitcnt_t bl = 0;
	int r;
	u32 i;
	if (!check || !in_isorder || nn_check_initialized(in_isorder) || nn_bitlen(in_isorder, &bl)) {
		return -1;
	}
	if (s != PUBLIC_PT) {
		/* a sensitive point: the reference multiplies with prj_pt_mul_blind (curves/prj_pt.c:1930-1935), and so does the batch */
		prj_pt *res;
		nn *m;
		int ret = -1;
		if (!in) {
			return -1;
		}
		if (n == 0) {
			return 0;
		}
		res = (prj_pt *)calloc(n, sizeof(prj_pt));
		m = (nn *)calloc(n, sizeof(nn));
		if (res && m) {
			for (i = 0; i < n; i++) {
				if (nn_copy(&m[i], in_isorder)) {
					break;
				}
			}
			if (i == n && !prj_pt_mul_blind_batch(res, m, in, n, ret_items)) {
				for (i = 0; i < n; i++) {
					int z = 0;
					check[i] = (!prj_pt_iszero(&res[i], &z) && z) ? 1 : 0;
				}
				ret = 0;
			}
		}
		free(res);
		free(m);
		return ret;
	}
	r = ptop_points(&J, PTOP_ORDER, NULL, in, NULL, n, ret_items);
	if (r <= 0) {
		return r;
	}
	J.flags = check;
	J.slen = bl ? (u32)BYTECEIL(bl) : 1;
	if (nn_to_be(J.bsc, J.slen, in_isorder)) {
		return -1;
	}
	return ptop_run(&J, n);
}

int ec_pub_key_import_from_aff_buf_batch(ec_pub_key *pub_keys, const ec_params *params, const u8 *const *pub_key_bufs, u8 pub_key_buf_len,
					 ec_alg_type ec_key_alg, u32 num, int *ret_items)
{
	ptop_job J;
	int isone = 0;
	bitcnt_t bl = 0;
	memset(&J, 0, sizeof(J));
	if (!pub_keys || !params || !pub_key_bufs) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	J.op = PTOP_PUBIMPORT;
	J.bufs = pub_key_bufs;
	J.buf_len = pub_key_buf_len;
	J.pubs = pub_keys;
	J.params = params;
	J.alg = ec_key_alg;
	J.ret_items = ret_items;
	J.crv = &(params->ec_curve);
	J.e = curve_from_params(params);
	if (!J.e || nn_isone(&(params->ec_gen_cofactor), &isone) || nn_bitlen(&(params->ec_gen_order), &bl)) {
		return -1;
	}
	J.need_order = !isone;
	J.slen = bl ? (u32)BYTECEIL(bl) : 1;
	if (nn_to_be(J.bsc, J.slen, &(params->ec_gen_order))) {
		return -1;
	}
	return ptop_run(&J, num);
}

/* ------------------------------------------------------------------------------------------------
 * public keys from private keys: init_pubkey_from_privkey and everything built on it
 * (key-pair generation / import, ecccdh_init_pub_key)
 * ------------------------------------------------------------------------------------------------ */
enum { RULE_NONE = 0, RULE_X_LT_Q, RULE_X_ANY, RULE_X_LT_QM1, RULE_XINV, RULE_EDDSA25519, RULE_EDDSA448 };

/* the scalar s of Y = [s]G for each algorithm, and the check of the private key in front of it */
static int pub_rule(ec_alg_type t)
{
	switch (t) {
#if defined(WITH_SIG_ECDSA)
	case ECDSA: return RULE_X_LT_Q;            /* __ecdsa_init_pub_key, sig/ecdsa_common.c:172-201 */
#endif
#if defined(WITH_SIG_DECDSA)
	case DECDSA: return RULE_X_LT_Q;
#endif
#if defined(WITH_SIG_ECKCDSA)
	case ECKCDSA: return RULE_XINV;            /* Y = [x^-1]G, sig/eckcdsa.c:36-72 */
#endif
#if defined(WITH_SIG_ECSDSA)
	case ECSDSA: return RULE_X_ANY;            /* __ecsdsa_init_pub_key: no range check, sig/ecsdsa_common.c:30-58 */
#endif
#if defined(WITH_SIG_ECOSDSA)
	case ECOSDSA: return RULE_X_ANY;
#endif
#if defined(WITH_SIG_ECFSDSA)
	case ECFSDSA: return RULE_X_LT_Q;          /* sig/ecfsdsa.c:30-58 */
#endif
#if defined(WITH_SIG_ECGDSA)
	case ECGDSA: return RULE_XINV;             /* sig/ecgdsa.c:30-66 */
#endif
#if defined(WITH_SIG_ECRDSA)
	case ECRDSA: return RULE_X_LT_Q;           /* sig/ecrdsa.c:70-98 */
#endif
#if defined(WITH_SIG_SM2)
	case SM2: return RULE_X_LT_QM1;            /* x < q - 1, sig/sm2.c:60-92 */
#endif
#if defined(WITH_SIG_EDDSA25519)
	case EDDSA25519: case EDDSA25519CTX: case EDDSA25519PH: return RULE_EDDSA25519;   /* eddsa_init_pub_key, sig/eddsa.c:786 */
#endif
#if defined(WITH_SIG_EDDSA448)
	case EDDSA448: case EDDSA448PH: return RULE_EDDSA448;
#endif
#if defined(WITH_SIG_BIGN)
	case BIGN: return RULE_X_LT_Q;             /* __bign_init_pub_key, sig/bign_common.c:345-375 */
#endif
#if defined(WITH_SIG_DBIGN)
	case DBIGN: return RULE_X_LT_Q;
#endif
#if defined(WITH_SIG_BIP0340)
	case BIP0340: return RULE_X_ANY;           /* sig/bip0340.c:102-125 */
#endif
#if defined(WITH_ECCCDH)
	case ECCCDH: return RULE_X_LT_Q;           /* ecccdh_init_pub_key, ecdh/ecccdh.c:60-90 */
#endif
	default: return RULE_NONE;
	}
}

/* the multiplier of the generator for one private key, big-endian in dst[slen]; -1 where the scalar function fails before its
 * multiplication */
static int pub_scalar(const ec_priv_key *pk, ec_alg_type alg, int rule, const ec_params *params, u8 *dst, u32 slen)
{
	nn_src_t q = &(params->ec_gen_order);
	nn t;
	int ret = -1, cmp = 0;
	t.magic = WORD(0);
	if (priv_key_check_initialized_and_type(pk, alg) || pk->params != params) {
		return -1;
	}
	switch (rule) {
	case RULE_X_LT_Q:
		ret = (nn_cmp(&pk->x, q, &cmp) || cmp >= 0) ? -1 : nn_to_be(dst, slen, &pk->x);
		break;
	case RULE_X_LT_QM1:
		ret = (nn_init(&t, 0) || nn_dec(&t, q) || nn_cmp(&pk->x, &t, &cmp) || cmp >= 0) ? -1 : nn_to_be(dst, slen, &pk->x);
		break;
	case RULE_X_ANY:
		/* any x: [x]G = [x mod q]G (G has order q); what is sent keeps to the order's length */
		ret = (nn_mod(&t, &pk->x, q)) ? -1 : nn_to_be(dst, slen, &t);
		break;
	case RULE_XINV:
		ret = (nn_cmp(&pk->x, q, &cmp) || cmp >= 0 || nn_modinv_fermat(&t, &pk->x, q)) ? -1 : nn_to_be(dst, slen, &t);
		break;
	case RULE_EDDSA25519:
	case RULE_EDDSA448: {
		/* eddsa_init_pub_key (sig/eddsa.c:786-858): x holds the hashed secret key (digest_size octets); the multiplier is
		 * the little-endian integer in its first half (eddsa_compute_s :291), shifted right by 2 for Ed448 (:840-850) */
		const u32 hsize = (rule == RULE_EDDSA448) ? 114 : 64;
		u8 dig[114];
		u32 k;
		if (params->curve_type != ((rule == RULE_EDDSA448) ? WEI448 : WEI25519) || slen != hsize / 2) {
			return -1;   /* eddsa_key_type_check_curve, sig/eddsa.c:156-186 */
		}
		if (nn_to_be(dig, hsize, &pk->x)) {
			return -1;
		}
		for (k = 0; k < slen; k++) {
			dst[k] = dig[slen - 1 - k];
		}
		if (rule == RULE_EDDSA448) {
			u32 carry = 0;
			for (k = 0; k < slen; k++) {
				const u32 v = dst[k];
				dst[k] = (u8)((v >> 2) | (carry << 6));
				carry = v & 3u;
			}
		}
		wipe(dig, sizeof(dig));
		ret = 0;
		break;
	}
	default:
		break;
	}
	nn_uninit(&t);
	return ret;
}

typedef struct {
	const ec_params *params;
	ec_alg_type alg;
	int rule;
	curve_ent *e;
	u32 slen, clen;
	/* per item: where the private key is and where the public key goes */
	ec_key_pair *kps;                    /* key pairs (generation / import), or NULL */
	const ec_priv_key *const *privs;     /* or: private keys ... */
	ec_pub_key *pubs;                    /* ... and public keys out */
	int mode;                            /* 0 keys exist, 1 generate, 2 import raw buffers, 3 import + EdDSA derivation */
	const u8 *const *bufs;
	u16 buf_len;
	int *ret_items;
	u8 *sc, *out, *st, *pre;
	int raw_mode;                        /* generation by the generic rule: raw random bytes out, x and Y back (ec_key_pair_gen_raw_batch) */
	u8 *raw;
} key_job;

static void key_pack(u32 lo, u32 hi, void *arg)
{
	key_job *J = (key_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const ec_priv_key *pk = J->kps ? &J->kps[i].priv_key : J->privs[i];
		int bad = 0;
		if (J->mode == 1 && J->raw_mode) {
			/* generic_gen_priv_key / ecccdh_gen_key_pair: x = nn_get_random_mod(q) -- its get_random call here, its reduction on the device */
			ec_priv_key *w = &J->kps[i].priv_key;
			bad = nn_init(&w->x, 0);
			w->key_type = J->alg;
			w->params = J->params;
			w->magic = PRIV_KEY_MAGIC;
			bad = bad || compat_get_random(J->raw + (size_t)i * 2 * J->slen, (u16)(2 * J->slen));
			J->pre[i] = bad ? 1 : 0;
			if (bad) {
				memset(J->raw + (size_t)i * 2 * J->slen, 0, (size_t)2 * J->slen);
			}
			continue;
		}
		if (J->mode == 1) {
			/* ec_key_pair_gen (sig/ec_key.c:594-621) / ecccdh_gen_key_pair (ecdh/ecccdh.c:93-118) up to the public key */
			ec_priv_key *w = &J->kps[i].priv_key;
			bad = nn_init(&w->x, 0);
			w->key_type = J->alg;
			w->params = J->params;
			w->magic = PRIV_KEY_MAGIC;
#if defined(WITH_ECCCDH)
			if (!bad && J->alg == ECCCDH) {
				bad = compat_random_mod(&w->x, &(J->params->ec_gen_order));   /* ecccdh_gen_key_pair: x in ]0, q[ */
			} else
#endif
			if (!bad) {
				/* libecc's own gen_priv_key draws through get_random inside (nn_get_random_mod, eddsa_gen_priv_key): one
				 * caller at a time unless the application lifted that (compat_random_mod) */
				if (AT_LOAD(&g_rand_concurrent)) {
					bad = gen_priv_key(w);
				} else {
					pthread_mutex_lock(&g_rand_mu);
					bad = gen_priv_key(w);
					pthread_mutex_unlock(&g_rand_mu);
				}
			}
		} else if (J->mode == 2) {
			/* ec_key_pair_import_from_priv_key_buf -> ec_priv_key_import_from_buf (sig/ec_key.c:289, :56) */
			bad = !J->bufs[i] || ec_priv_key_import_from_buf(&J->kps[i].priv_key, J->params, J->bufs[i], (u8)J->buf_len, J->alg);
		} else if (J->mode == 3) {
			/* eddsa_import_key_pair_from_priv_key_buf -> eddsa_import_priv_key (sig/eddsa.c:1028, :737) */
			bad = !J->bufs[i] || eddsa_import_priv_key(&J->kps[i].priv_key, J->bufs[i], J->buf_len, J->params, J->alg);
		}
		bad = bad || !pk || pub_scalar(pk, J->alg, J->rule, J->params, J->sc + (size_t)i * J->slen, J->slen);
		J->pre[i] = bad ? 1 : 0;
		if (bad) {
			memset(J->sc + (size_t)i * J->slen, 0, J->slen);
		}
	}
}

static int key_gpu(u32 lo, u32 hi, void *arg)
{
	key_job *J = (key_job *)arg;
	if (J->raw_mode) {
		if (ecamd_multi_key_pair_gen_raw_batch(g_multi, J->e->mc, hi - lo, J->raw + (size_t)lo * 2 * J->slen, J->sc + (size_t)lo * J->slen,
						       J->out + (size_t)lo * 2 * J->clen, J->st + lo)) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			return -1;
		}
		return 0;
	}
	/* Y = [s]G: the generator (points == NULL); secret-scalar mode keeps the look-ups address-independent */
	if (ecamd_multi_prj_pt_mul_batch(g_multi, J->e->mc, hi - lo, J->sc + (size_t)lo * J->slen, J->slen, NULL, J->out + (size_t)lo * 2 * J->clen,
					 J->st + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void key_unpack(u32 lo, u32 hi, void *arg)
{
	key_job *J = (key_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		ec_pub_key *pub = J->kps ? &J->kps[i].pub_key : &J->pubs[i];
		int r = -1;
		memset(pub, 0, sizeof(ec_pub_key));
		if (J->raw_mode && !J->pre[i] && nn_init_from_buf(&J->kps[i].priv_key.x, J->sc + (size_t)i * J->slen, (u16)J->slen)) {
			J->pre[i] = 1;   /* (the private scalar the device reduced comes back as qlen big-endian octets) */
		}
		if (!J->pre[i] && !prj_from_aff_be(&pub->y, &(J->params->ec_curve), J->out + (size_t)i * 2 * J->clen, J->clen, J->st[i])) {
			pub->key_type = J->alg;
			pub->params = J->params;
			pub->magic = PUB_KEY_MAGIC;
			r = 0;
		}
		if (r && J->kps) {
			memset(&J->kps[i], 0, sizeof(ec_key_pair));   /* as the scalar functions' error paths */
		}
		if (J->ret_items) {
			J->ret_items[i] = r;
		}
	}
}

static int key_batch(key_job *J, u32 num)
{
	int ret = -1;
	if (num == 0) {
		return 0;
	}
	if (!J->params) {
		return -1;
	}
	J->rule = pub_rule(J->alg);
	if (J->rule == RULE_NONE) {
		return -1;
	}
	J->e = curve_from_params(J->params);
	if (!J->e) {
		return -1;
	}
	J->clen = J->e->clen;
	J->slen = (J->rule == RULE_EDDSA25519) ? 32 : (J->rule == RULE_EDDSA448) ? 57 : J->e->qlen;
	call_enter(0);
	J->sc = buf_get(0, (size_t)num * J->slen);
	J->out = buf_get(1, (size_t)num * 2 * J->clen);
	J->st = buf_get(2, num);
	J->pre = buf_get(3, num);
	J->raw_mode = J->mode == 1 && J->rule == RULE_X_LT_Q && J->kps && raw_random_on_device();
	J->raw = J->raw_mode ? buf_get(4, (size_t)num * 2 * J->slen) : NULL;
	if (J->raw_mode && !J->raw) {
		J->sc = NULL;
	}
	if (J->sc && J->out && J->st && J->pre && !pipeline_run(num, chunk_items_for(g_chunk, num), key_pack, key_gpu, key_unpack, J)) {
		note_items(num);
		ret = 0;
	}
	wipe_secrets();
	call_leave();
	return ret;
}

int ec_key_pair_gen_batch(ec_key_pair *kps, const ec_params *params, ec_alg_type ec_key_alg, u32 num, int *ret_items)
{
	key_job J;
	memset(&J, 0, sizeof(J));
	if (!kps || !params) {
		return -1;
	}
	J.params = params;
	J.alg = ec_key_alg;
	J.kps = kps;
	J.mode = 1;
	J.ret_items = ret_items;
	return key_batch(&J, num);
}

int ec_key_pair_import_from_priv_key_buf_batch(ec_key_pair *kps, const ec_params *params, const u8 *const *priv_keys, u8 priv_key_len,
					       ec_alg_type ec_key_alg, u32 num, int *ret_items)
{
	key_job J;
	memset(&J, 0, sizeof(J));
	if (!kps || !params || !priv_keys) {
		return -1;
	}
	J.params = params;
	J.alg = ec_key_alg;
	J.kps = kps;
	J.mode = 2;
	J.bufs = priv_keys;
	J.buf_len = priv_key_len;
	J.ret_items = ret_items;
	return key_batch(&J, num);
}

int eddsa_import_key_pair_from_priv_key_buf_batch(ec_key_pair *kps, const u8 *const *priv_keys, u16 priv_key_len,
						  const ec_params *shortw_curve_params, ec_alg_type sig_type, u32 num, int *ret_items)
{
	key_job J;
	memset(&J, 0, sizeof(J));
	if (!kps || !shortw_curve_params || !priv_keys) {
		return -1;
	}
	J.params = shortw_curve_params;
	J.alg = sig_type;
	J.kps = kps;
	J.mode = 3;
	J.bufs = priv_keys;
	J.buf_len = priv_key_len;
	J.ret_items = ret_items;
	if (pub_rule(sig_type) != RULE_EDDSA25519 && pub_rule(sig_type) != RULE_EDDSA448) {
		return -1;
	}
	return key_batch(&J, num);
}

int init_pubkey_from_privkey_batch(ec_pub_key *out_pubs, const ec_priv_key *const *in_privs, u32 num, int *ret_items)
{
	key_job J;
	memset(&J, 0, sizeof(J));
	if (!out_pubs || !in_privs) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	if (priv_key_check_initialized(in_privs[0])) {
		return -1;
	}
	J.params = in_privs[0]->params;
	J.alg = in_privs[0]->key_type;
	J.privs = in_privs;
	J.pubs = out_pubs;
	J.mode = 0;
	J.ret_items = ret_items;
	return key_batch(&J, num);
}

int ecccdh_init_pub_key_batch(ec_pub_key *out_pubs, const ec_priv_key *const *in_privs, u32 num, int *ret_items)
{
#if defined(WITH_ECCCDH)
	if (num && (!in_privs || priv_key_check_initialized_and_type(in_privs[0], ECCCDH))) {
		return -1;
	}
	return init_pubkey_from_privkey_batch(out_pubs, in_privs, num, ret_items);
#else
	(void)out_pubs; (void)in_privs; (void)num; (void)ret_items;
	return -1;
#endif
}

int ecccdh_gen_key_pair_batch(ec_key_pair *kps, const ec_params *params, u32 num, int *ret_items)
{
#if defined(WITH_ECCCDH)
	return ec_key_pair_gen_batch(kps, params, ECCCDH, num, ret_items);
#else
	(void)kps; (void)params; (void)num; (void)ret_items;
	return -1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * ecccdh_derive_secret_batch
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const ec_priv_key *const *privs;
	const u8 *const *peers;
	u8 *const *secrets;
	const ec_params *params;
	curve_ent *e;
	u8 *pv, *pk, *sec, *st, *pre;
	u32 qlen, clen;
	nn_src_t q;
	int *ret_items;
} cdh_job;

static void cdh_pack(u32 lo, u32 hi, void *arg)
{
	cdh_job *J = (cdh_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const ec_priv_key *k = J->privs[i];
		int bad = 1;
		/* sanity checks of ecccdh_derive_secret (ecdh/ecccdh.c:176-178) */
		if (J->secrets[i] && J->peers[i] && !priv_key_check_initialized_and_type(k, ECCCDH) && k->params == J->params) {
			bad = nn_to_be(J->pv + (size_t)i * J->qlen, J->qlen, &k->x);
			if (bad && !nn_check_initialized(&k->x)) {
				/* a private scalar longer than the group order: [x]Q = [x mod q]Q for every Q that passes the
				 * checks in front of the multiplication (Q lies in the subgroup of order q) */
				nn t;
				t.magic = WORD(0);
				bad = nn_mod(&t, &k->x, J->q) || nn_to_be(J->pv + (size_t)i * J->qlen, J->qlen, &t);
				nn_uninit(&t);
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

static int cdh_gpu(u32 lo, u32 hi, void *arg)
{
	cdh_job *J = (cdh_job *)arg;
	if (ecamd_multi_ecccdh_derive_batch(g_multi, J->e->mc, hi - lo, J->pv + (size_t)lo * J->qlen, J->pk + (size_t)lo * 2 * J->clen,
					    J->sec + (size_t)lo * J->clen, J->st + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void cdh_unpack(u32 lo, u32 hi, void *arg)
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
	J.e = curve_from_params(J.params);
	if (!J.e) {
		return -1;
	}
	J.privs = our_priv_keys;
	J.peers = peer_pub_keys;
	J.secrets = shared_secrets;
	J.ret_items = ret_items;
	J.qlen = J.e->qlen;
	J.clen = J.e->clen;
	J.q = &(J.params->ec_gen_order);
	call_enter(0);
	J.pv = buf_get(0, (size_t)num * J.qlen);
	J.pk = buf_get(1, (size_t)num * 2 * J.clen);
	J.sec = buf_get(2, (size_t)num * J.clen);
	J.st = buf_get(3, num);
	J.pre = buf_get(4, num);
	if (J.pv && J.pk && J.sec && J.st && J.pre && !pipeline_run(num, chunk_items_for(g_chunk, num), cdh_pack, cdh_gpu, cdh_unpack, &J)) {
		note_items(num);
		ret = 0;
	}
	wipe_secrets();   /* private scalars and shared secrets */
	call_leave();
	return ret;
}

/* ------------------------------------------------------------------------------------------------
 * X25519 / X448
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const u8 *const *k, *const *u;
	u8 *const *res;
	u32 len;
	curve_ent *e;
	u8 *kb, *ub, *rb, *st, *pre;
	int *ret_items;
} xdh_job;

static void xdh_pack(u32 lo, u32 hi, void *arg)
{
	xdh_job *J = (xdh_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		/* MUST_HAVE((k != NULL) && (u != NULL) && (res != NULL)), ecdh/x25519_448.c:163 */
		const int bad = !J->k[i] || !J->res[i] || (J->u && !J->u[i]);
		J->pre[i] = bad ? 1 : 0;
		if (bad) {
			memset(J->kb + (size_t)i * J->len, 0, J->len);
			memset(J->ub + (size_t)i * J->len, 0xff, J->len);   /* non-canonical u: rejected */
			continue;
		}
		memcpy(J->kb + (size_t)i * J->len, J->k[i], J->len);
		if (J->u) {
			memcpy(J->ub + (size_t)i * J->len, J->u[i], J->len);
		} else {
			/* x25519_448_init_pub_key: the base point u = 9 / 5 (ecdh/x25519_448.c:333-349) */
			memset(J->ub + (size_t)i * J->len, 0, J->len);
			J->ub[(size_t)i * J->len] = (J->len == 32) ? 0x09 : 0x05;
		}
	}
}

static int xdh_gpu(u32 lo, u32 hi, void *arg)
{
	xdh_job *J = (xdh_job *)arg;
	if (ecamd_multi_xdh_batch(g_multi, J->e->mc, hi - lo, J->kb + (size_t)lo * J->len, J->ub + (size_t)lo * J->len, J->rb + (size_t)lo * J->len,
				  J->st + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void xdh_unpack(u32 lo, u32 hi, void *arg)
{
	xdh_job *J = (xdh_job *)arg;
	u32 i;
	for (i = lo; i < hi; i++) {
		const int ok = !J->pre[i] && J->st[i] == ECAMD_OK;
		if (ok) {
			memcpy(J->res[i], J->rb + (size_t)i * J->len, J->len);
		}
		if (J->ret_items) {
			J->ret_items[i] = ok ? 0 : -1;
		}
	}
}

static int xdh_batch(const char *curve, u32 len, const u8 *const *k, const u8 *const *u, u8 *const *res, u32 num, int *ret_items)
{
	xdh_job J;
	const ec_str_params *sp = NULL;
	ec_params params;
	int ret = -1;
	memset(&J, 0, sizeof(J));
	if (!k || !res) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	/* the Weierstrass model the reference itself computes on (x25519_448_core imports WEI25519 / WEI448, :180-190) */
	if (ec_get_curve_params_by_name((const u8 *)curve, (u8)(strlen(curve) + 1), &sp) || !sp || import_params(&params, sp)) {
		return -1;
	}
	J.e = curve_from_params(&params);
	if (!J.e || J.e->clen != len) {
		return -1;
	}
	J.k = k;
	J.u = u;
	J.res = res;
	J.len = len;
	J.ret_items = ret_items;
	call_enter(0);
	J.kb = buf_get(0, (size_t)num * len);
	J.ub = buf_get(1, (size_t)num * len);
	J.rb = buf_get(2, (size_t)num * len);
	J.st = buf_get(3, num);
	J.pre = buf_get(4, num);
	if (J.kb && J.ub && J.rb && J.st && J.pre && !pipeline_run(num, chunk_items_for(g_chunk, num), xdh_pack, xdh_gpu, xdh_unpack, &J)) {
		note_items(num);
		ret = 0;
	}
	wipe_secrets();
	call_leave();
	return ret;
}

#if defined(WITH_X25519)
int x25519_batch(const u8 *const *k, const u8 *const *u, u8 *const *res, u32 num, int *ret_items)
{
	return u ? xdh_batch("WEI25519", 32, k, u, res, num, ret_items) : -1;
}
int x25519_init_pub_key_batch(const u8 *const *priv_keys, u8 *const *pub_keys, u32 num, int *ret_items)
{
	return xdh_batch("WEI25519", 32, priv_keys, NULL, pub_keys, num, ret_items);
}
int x25519_derive_secret_batch(const u8 *const *priv_keys, const u8 *const *peer_pub_keys, u8 *const *shared_secrets, u32 num, int *ret_items)
{
	return x25519_batch(priv_keys, peer_pub_keys, shared_secrets, num, ret_items);
}
#endif
#if defined(WITH_X448)
int x448_batch(const u8 *const *k, const u8 *const *u, u8 *const *res, u32 num, int *ret_items)
{
	return u ? xdh_batch("WEI448", 56, k, u, res, num, ret_items) : -1;
}
int x448_init_pub_key_batch(const u8 *const *priv_keys, u8 *const *pub_keys, u32 num, int *ret_items)
{
	return xdh_batch("WEI448", 56, priv_keys, NULL, pub_keys, num, ret_items);
}
int x448_derive_secret_batch(const u8 *const *priv_keys, const u8 *const *peer_pub_keys, u8 *const *shared_secrets, u32 num, int *ret_items)
{
	return x448_batch(priv_keys, peer_pub_keys, shared_secrets, num, ret_items);
}
#endif

/* ------------------------------------------------------------------------------------------------
 * hashing helpers shared by signing and verification
 * ------------------------------------------------------------------------------------------------ */
static const hash_mapping *find_hash(hash_alg_type hash_type)
{
	const hash_mapping *hm = NULL;
	if (get_hash_by_type(hash_type, &hm) || !hm || hash_mapping_callbacks_sanity_check(hm)) {
		return NULL;
	}
	return hm;
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

static int is_decdsa(ec_alg_type t)
{
#if defined(WITH_SIG_DECDSA)
	return t == DECDSA;
#else
	(void)t;
	return 0;
#endif
}

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

/* Hashing on the device (ec_ecdsa_verify_msg_batch_fmt / ec_eddsa_verify_msg_batch of libecc_amd.h): for SHA-224 / 256 / 384 / 512
 * and a group whose longest hash input fits a 256-byte slot, the pack step copies the message instead of hashing it -- libecc's
 * portable hfunc_* cost 0.3 - 0.5 us per short message and thread, more than everything else the layer does per signature.
 * $ECAMD_COMPAT_HOST_HASH keeps the hashing on the host (through the application's hash_maps[], as before). */
#define DEV_HASH_MAX_SLOT 256u
static int dev_hash_type(const hash_mapping *hm)
{
	if (getenv("ECAMD_COMPAT_HOST_HASH")) {
		return 0;
	}
	switch (hm->type) {
	case SHA224: return 1;
	case SHA256: return 2;
	case SHA384: return 3;
	case SHA512: return 4;
	default: return 0;
	}
}
/* stride of the slots for the items idx[0..cnt) with `extra` bytes in front of every message, or 0 when one does not fit
 * (the longest message: a reduction over the pool -- a serial pass over 2^20 lengths is a millisecond of the caller's time) */
typedef struct {
	const u32 *m_len, *idx;
	u32 mx;
} mlen_job;
static void mlen_max(u32 lo, u32 hi, void *arg)
{
	mlen_job *M = (mlen_job *)arg;
	u32 j, mx = 0, cur;
	for (j = lo; j < hi; j++) {
		const u32 l = M->m_len[M->idx[j]];
		mx = l > mx ? l : mx;
	}
	cur = AT_LOAD(&M->mx);
	while (mx > cur && !__atomic_compare_exchange_n(&M->mx, &cur, mx, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
	}
}
/* known_max >= 0: the longest message was found by an earlier pass */
static u32 dev_hash_slot_for(const u32 *m_len, const u32 *idx, u32 cnt, u32 extra, long known_max)
{
	mlen_job M;
	u32 mx;
	M.m_len = m_len;
	M.idx = idx;
	M.mx = 0;
	if (known_max >= 0) {
		M.mx = (u32)known_max;
	} else {
		parallel_for(cnt, mlen_max, &M);
	}
	mx = M.mx;
	if (mx > DEV_HASH_MAX_SLOT) {
		return 0;
	}
	mx = (4 + extra + mx + 3u) & ~3u;
	return mx <= DEV_HASH_MAX_SLOT ? mx : 0;
}
static void slot_put(u8 *slot, u32 stride, const u8 *a, u32 alen, const u8 *b, u32 blen, const u8 *m, u32 mlen)
{
	const u32 len = alen + blen + mlen;
	slot[0] = (u8)len; slot[1] = (u8)(len >> 8); slot[2] = (u8)(len >> 16); slot[3] = (u8)(len >> 24);
	if (alen) memcpy(slot + 4, a, alen);
	if (blen) memcpy(slot + 4 + alen, b, blen);
	if (mlen) memcpy(slot + 4 + alen + blen, m, mlen);
	memset(slot + 4 + len, 0, stride - 4 - len);
}

/* ------------------------------------------------------------------------------------------------
 * signing: ec_sign_batch
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	u8 *const *sigs;
	const ec_key_pair *const *kps;
	const u8 *const *m;
	const u32 *m_len;
	const u8 *const *adata;
	const u16 *adata_len;
	int (*rand)(nn_t out, nn_src_t q);
	ec_alg_type sig_type;
	hash_alg_type hash_type;
	const hash_mapping *hm;
	const u32 *idx;
	const ec_params *params;
	curve_ent *e;
	int *ret_items;
	u8 siglen;
	u32 clen, qlen, hlen, klen;
	/* ECDSA: private keys, nonces, digests in; signatures, status out.  EdDSA: see eddsa_sign_group */
	u8 *b0, *b1, *b2, *b3, *b4, *b5, *b6, *b7, *pre;
	int nonces_given;    /* the nonces were drawn beforehand (caller's rand hook) */
	int raw_nonces;      /* ECDSA with libecc's own nonce source: b1 holds 2 * qlen raw random bytes per item, reduced on the device */
	int dev_hash;        /* ... and H(m) of short messages comes from the device: b2 holds message slots of `slot` bytes */
	u32 slot;
	int ph, dom, is448;
	u32 ph_len;
} sign_job;

/* RFC 6979 nonce, as __ecdsa_rfc6979_nonce (sig/ecdsa_common.c:48-168) computes it, through libecc's own hmac_*:
 * k_be[qlen] from the private key x_be[qlen] and the message digest */
static int rfc6979_nonce(u8 *k_be, u32 qlen, nn_src_t q, bitcnt_t q_bit_len, const u8 *x_be, const u8 *hash, u8 hsize, hash_alg_type hash_type)
{
	u8 V[MAX_DIGEST_SIZE], K[MAX_DIGEST_SIZE], T[80 + 2 * MAX_DIGEST_SIZE], h1[80], tmp, hmac_size;
	hmac_context hc;
	nn k;
	int ret = -1, cmp = 0, tries;
	bitcnt_t t_bit_len;
	k.magic = WORD(0);
	if (qlen > 80 || hsize > MAX_DIGEST_SIZE || hsize == 0) {
		return -1;
	}
	memset(V, 0x01, hsize);
	memset(K, 0x00, hsize);
	/* bits2octets(h1): the digest's leftmost qbits bits, reduced mod q (:90-96) */
	ret = nn_init_from_buf(&k, hash, hsize); EG(ret, err);
	if ((8 * (u32)hsize) > (u32)q_bit_len) {
		ret = nn_rshift(&k, &k, (bitcnt_t)((8 * hsize) - q_bit_len)); EG(ret, err);
	}
	ret = nn_mod(&k, &k, q); EG(ret, err);
	ret = nn_export_to_buf(h1, (u16)qlen, &k); EG(ret, err);
	for (tmp = 0; tmp < 2; tmp++) {
		/* steps d / f: K = HMAC_K(V || 0x00 / 0x01 || int2octets(x) || bits2octets(h1)); steps e / g: V = HMAC_K(V) */
		ret = hmac_init(&hc, K, hsize, hash_type); EG(ret, err);
		ret = hmac_update(&hc, V, hsize); EG(ret, err);
		ret = hmac_update(&hc, &tmp, 1); EG(ret, err);
		ret = hmac_update(&hc, x_be, qlen); EG(ret, err);
		ret = hmac_update(&hc, h1, qlen); EG(ret, err);
		hmac_size = sizeof(K);
		ret = hmac_finalize(&hc, K, &hmac_size); EG(ret, err);
		hmac_size = sizeof(V);
		ret = hmac(K, hsize, hash_type, V, hsize, V, &hmac_size); EG(ret, err);
	}
	/* step h */
	for (tries = 0; tries < 1000; tries++) {
		t_bit_len = 0;
		while (t_bit_len < q_bit_len) {
			hmac_size = sizeof(V);
			ret = hmac(K, hsize, hash_type, V, hsize, V, &hmac_size); EG(ret, err);
			memcpy(&T[BYTECEIL(t_bit_len)], V, hmac_size);
			t_bit_len = (bitcnt_t)(t_bit_len + (8 * hmac_size));
		}
		ret = nn_init_from_buf(&k, T, (u16)qlen); EG(ret, err);
		if ((8 * qlen) > (u32)q_bit_len) {
			ret = nn_rshift(&k, &k, (bitcnt_t)((8 * qlen) - q_bit_len)); EG(ret, err);
		}
		ret = nn_cmp(&k, q, &cmp); EG(ret, err);
		if (cmp < 0) {
			ret = nn_export_to_buf(k_be, (u16)qlen, &k);
			goto err;
		}
		/* K = HMAC_K(V || 0x00), V = HMAC_K(V) */
		tmp = 0x00;
		ret = hmac_init(&hc, K, hsize, hash_type); EG(ret, err);
		ret = hmac_update(&hc, V, hsize); EG(ret, err);
		ret = hmac_update(&hc, &tmp, 1); EG(ret, err);
		hmac_size = sizeof(K);
		ret = hmac_finalize(&hc, K, &hmac_size); EG(ret, err);
		hmac_size = sizeof(V);
		ret = hmac(K, hsize, hash_type, V, hsize, V, &hmac_size); EG(ret, err);
	}
	ret = -1;
err:
	nn_uninit(&k);
	wipe(V, sizeof(V));
	wipe(K, sizeof(K));
	wipe(T, sizeof(T));
	return ret;
}

/* ---- ECDSA / DECDSA: b0 private keys, b1 nonces, b2 digests, b3 signatures, b4 status ---- */
static void ecdsa_sign_pack(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_key_pair *kp = J->kps[i];
		const u32 kstride = J->raw_nonces ? 2 * J->qlen : J->qlen, dstride = J->dev_hash ? J->slot : J->hlen;
		u8 *xb = J->b0 + (size_t)j * J->qlen, *kb = J->b1 + (size_t)j * kstride, *dg = J->b2 + (size_t)j * dstride;
		hash_context hc;
		int bad, cmp = 0;
		/* _ec_sign_init / __ecdsa_sign_init / __ecdsa_sign_finalize up to the multiplication (sig/sig_algs.c:293-376,
		 * sig/ecdsa_common.c:262-283, :318-400): key pair of this algorithm, x < q, signature length, h = H(m) */
		bad = key_pair_check_initialized_and_type(kp, J->sig_type) || kp->priv_key.params != J->params || !J->sigs[i] ||
		      (!J->m[i] && J->m_len[i]);
		bad = bad || nn_cmp(&kp->priv_key.x, &(J->params->ec_gen_order), &cmp) || cmp >= 0 || nn_to_be(xb, J->qlen, &kp->priv_key.x);
		if (J->dev_hash) {
			if (!bad) {
				slot_put(dg, J->slot, NULL, 0, NULL, 0, J->m[i], J->m_len[i]);
			}
		} else {
			bad = bad || J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hc, dg);
		}
		if (!bad && J->raw_nonces) {
			bad = compat_get_random(kb, (u16)(2 * J->qlen));   /* nn_get_random_mod's one call; the reduction runs on the device */
		} else if (!bad && is_decdsa(J->sig_type)) {
			bad = rfc6979_nonce(kb, J->qlen, &(J->params->ec_gen_order), J->params->ec_gen_order_bitlen, xb, dg, (u8)J->hlen, J->hash_type);
		} else if (!bad && !J->nonces_given) {
			nn k;
			k.magic = WORD(0);
			bad = compat_random_mod(&k, &(J->params->ec_gen_order)) || nn_to_be(kb, J->qlen, &k);
			nn_uninit(&k);
		} else if (!bad) {
			bad = J->pre[j];   /* the caller's rand hook failed for this item */
		}
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(xb, 0, J->qlen);
			memset(kb, 0, kstride);
			memset(dg, 0, dstride);
		}
	}
}

static int ecdsa_sign_gpu(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	if (J->raw_nonces) {
		if (ecamd_multi_ecdsa_sign_msg_batch(g_multi, J->e->mc, hi - lo, J->b0 + (size_t)lo * J->qlen, J->b1 + (size_t)lo * 2 * J->qlen,
						     J->dev_hash, J->b2 + (size_t)lo * (J->dev_hash ? J->slot : J->hlen), J->dev_hash ? J->slot : J->hlen,
						     J->b3 + (size_t)lo * 2 * J->qlen, J->b4 + lo)) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			return -1;
		}
		return 0;
	}
	if (ecamd_multi_ecdsa_sign_batch(g_multi, J->e->mc, hi - lo, J->b0 + (size_t)lo * J->qlen, J->b1 + (size_t)lo * J->qlen,
					 J->b2 + (size_t)lo * J->hlen, J->hlen, J->b3 + (size_t)lo * 2 * J->qlen, J->b4 + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void ecdsa_sign_unpack(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		if (!J->pre[j] && J->b4[j] == 0) {
			memcpy(J->sigs[i], J->b3 + (size_t)j * 2 * J->qlen, 2 * J->qlen);
			J->ret_items[i] = 0;
		} else {
			J->ret_items[i] = J->pre[j] ? -1 : -2;   /* -2: the device asks for another nonce (resolved by the caller) */
		}
	}
}

static int ecdsa_sign_group(sign_job *J, u32 cnt)
{
	u32 j, round;
	int ret = -1;
	J->clen = J->e->clen;
	J->qlen = J->e->qlen;
	J->hlen = J->hm->digest_size;
	if ((u32)J->siglen != 2 * (u32)BYTECEIL(J->params->ec_gen_order_bitlen) || 2 * J->qlen != (u32)J->siglen) {
		/* MUST_HAVE((siglen == ECDSA_SIGLEN(q_bit_len))), sig/ecdsa_common.c:380: every item fails */
		for (j = 0; j < cnt; j++) {
			J->ret_items[J->idx[j]] = -1;
		}
		return 0;
	}
	/* libecc's own nonce source (rand == NULL or nn_get_random_mod), not the deterministic variant: raw bytes out, reduction on the device */
	J->raw_nonces = (!J->rand || J->rand == nn_get_random_mod) && !is_decdsa(J->sig_type) && raw_random_on_device();
	J->dev_hash = 0;
	J->slot = 0;
	if (J->raw_nonces && dev_hash_type(J->hm)) {
		J->slot = dev_hash_slot_for(J->m_len, J->idx, cnt, 0, -1);
		J->dev_hash = J->slot ? dev_hash_type(J->hm) : 0;
	}
	J->b0 = buf_get(0, (size_t)cnt * J->qlen);
	J->b1 = buf_get(1, (size_t)cnt * (J->raw_nonces ? 2 : 1) * J->qlen);
	J->b2 = buf_get(2, (size_t)cnt * (J->dev_hash ? J->slot : J->hlen));
	J->b3 = buf_get(3, (size_t)cnt * 2 * J->qlen);
	J->b4 = buf_get(4, cnt);
	J->pre = buf_get(5, cnt);
	if (!J->b0 || !J->b1 || !J->b2 || !J->b3 || !J->b4 || !J->pre) {
		return -1;
	}
	for (round = 0; round < 8; round++) {
		u32 again = 0;
		memset(J->pre, 0, cnt);
		J->nonces_given = 0;
		if (J->rand && J->rand != nn_get_random_mod && !is_decdsa(J->sig_type)) {
			/* the caller's nonce source: one call per item, in index order, on this thread (it may be stateful) */
			nn k;
			k.magic = WORD(0);
			for (j = 0; j < cnt; j++) {
				int cmp = 0, z = 1;
				if (J->rand(&k, &(J->params->ec_gen_order)) || nn_iszero(&k, &z) || z ||
				    nn_cmp(&k, &(J->params->ec_gen_order), &cmp) || cmp >= 0 || nn_to_be(J->b1 + (size_t)j * J->qlen, J->qlen, &k)) {
					J->pre[j] = 1;   /* "expected to initialize a nn 'out' with a value taken uniformly at random in [1, q-1]" */
				}
			}
			nn_uninit(&k);
			J->nonces_given = 1;
		}
		if (pipeline_run(cnt, chunk_items_for(g_chunk, cnt), ecdsa_sign_pack, ecdsa_sign_gpu, ecdsa_sign_unpack, J)) {
			goto done;
		}
		note_items(cnt);
		/* restart of steps 4-10 for the items whose nonce gave r = 0, e = x r or s = 0 (sig/ecdsa_common.c:497-557):
		 * compact them to the front and sign them again with fresh nonces.  A deterministic nonce cannot change:
		 * the reference would loop for ever on such a (key, message); the batch form reports -1. */
		for (j = 0; j < cnt; j++) {
			if (J->ret_items[J->idx[j]] == -2) {
				if (is_decdsa(J->sig_type)) {
					J->ret_items[J->idx[j]] = -1;
				} else {
					((u32 *)J->idx)[again++] = J->idx[j];
				}
			}
		}
		if (!again) {
			break;
		}
		cnt = again;
	}
	for (j = 0; j < cnt; j++) {
		if (J->ret_items[J->idx[j]] == -2) {
			J->ret_items[J->idx[j]] = -1;
		}
	}
	ret = 0;
done:
	return ret;
}

/* ---- EdDSA: b0 key points X||Y||Z -> b1 encoded keys A; b2 r_hash; b3 secret scalars a; b4 R; b5 hram; b6 S; b7 status of R ---- */
static void eddsa_sign_keys(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const ec_key_pair *kp = J->kps[J->idx[j]];
		u8 *dst = J->b0 + (size_t)j * 3 * J->clen;
		/* eddsa_key_pair_sanity_check (sig/eddsa.c:213-227): both halves initialised, of one EdDSA type, on the variant's curve */
		if (key_pair_check_initialized_and_type(kp, J->sig_type) || kp->priv_key.params != J->params || kp->pub_key.params != J->params ||
		    prj_to_be(dst, J->clen, &kp->pub_key.y, &(J->params->ec_curve))) {
			memset(dst, 0xff, (size_t)3 * J->clen);   /* coordinates >= p: an import error on the device */
		}
	}
}

static int eddsa_msg_hash(sign_job *J, u32 i, u8 *ph_out)
{
	/* eddsa_compute_pre_hash (sig/eddsa.c:1049-1080): PH(M) with the variant's hash */
	hash_context hp;
	return J->hm->hfunc_init(&hp) || J->hm->hfunc_update(&hp, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hp, ph_out);
}

static void eddsa_sign_pack(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j, k;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_key_pair *kp = J->kps[i];
		const u8 *ad = J->adata ? J->adata[i] : NULL;
		const u16 adl = J->adata_len ? J->adata_len[i] : 0;
		u8 dig[MAX_DIGEST_SIZE], ph[MAX_DIGEST_SIZE];
		u8 *rh = J->b2 + (size_t)j * J->hlen, *as = J->b3 + (size_t)j * J->klen;
		hash_context hc;
		bitcnt_t blen = 0;
		int bad;
		/* _eddsa_sign up to the multiplication (sig/eddsa.c:1596-1731) */
		bad = J->pre[j] || !J->sigs[i] || (!J->m[i] && J->m_len[i]) || nn_bitlen(&kp->priv_key.x, &blen) || (u32)blen > 8 * J->hlen;
#if defined(WITH_SIG_EDDSA25519)
		bad = bad || (J->sig_type == EDDSA25519CTX && !ad);
#endif
		bad = bad || nn_to_be(dig, J->hlen, &kp->priv_key.x);   /* eddsa_get_digest_from_priv_key :306 */
		bad = bad || (J->ph && eddsa_msg_hash(J, i, ph));
		bad = bad || J->hm->hfunc_init(&hc);
		if (!bad && J->dom) {
			bad = dom_prefix(J->hm, &hc, J->is448, J->ph, ad, adl);
		}
		bad = bad || J->hm->hfunc_update(&hc, dig + J->klen, J->klen);   /* the prefix: second half of the hashed key */
		if (!bad && J->ph) {
			bad = J->hm->hfunc_update(&hc, ph, J->ph_len);
		} else if (!bad) {
			bad = J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]);
		}
		bad = bad || J->hm->hfunc_finalize(&hc, rh);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(rh, 0, J->hlen);
			memset(as, 0, J->klen);
		} else {
			/* the secret scalar: first half of the hashed key, little-endian as the device takes it (eddsa_compute_s :291) */
			for (k = 0; k < J->klen; k++) {
				as[k] = dig[k];
			}
		}
		wipe(dig, sizeof(dig));
		wipe(&hc, sizeof(hc));   /* the hash context absorbed the secret prefix (the reference clears h_ctx, sig/eddsa.c:1883) */
		wipe(ph, sizeof(ph));
	}
}

static int eddsa_sign_gpu_R(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	if (ecamd_multi_eddsa_sign_R_batch(g_multi, J->e->mc, hi - lo, J->b2 + (size_t)lo * J->hlen, J->b4 + (size_t)lo * J->klen, J->b7 + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

static void eddsa_sign_hram(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const u8 *ad = J->adata ? J->adata[i] : NULL;
		const u16 adl = J->adata_len ? J->adata_len[i] : 0;
		u8 ph[MAX_DIGEST_SIZE];
		u8 *hr = J->b5 + (size_t)j * J->hlen;
		hash_context hc;
		int bad = J->pre[j] || J->b7[j] != 0;
		/* H(dom || R || A || PH(M)) (sig/eddsa.c:1778-1836) */
		bad = bad || (J->ph && eddsa_msg_hash(J, i, ph));
		bad = bad || J->hm->hfunc_init(&hc);
		if (!bad && J->dom) {
			bad = dom_prefix(J->hm, &hc, J->is448, J->ph, ad, adl);
		}
		bad = bad || J->hm->hfunc_update(&hc, J->b4 + (size_t)j * J->klen, J->klen) || J->hm->hfunc_update(&hc, J->b1 + (size_t)j * J->klen, J->klen);
		if (!bad && J->ph) {
			bad = J->hm->hfunc_update(&hc, ph, J->ph_len);
		} else if (!bad) {
			bad = J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]);
		}
		bad = bad || J->hm->hfunc_finalize(&hc, hr);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(hr, 0, J->hlen);
		}
	}
}

static void eddsa_sign_out(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		if (!J->pre[j]) {
			memcpy(J->sigs[i], J->b4 + (size_t)j * J->klen, J->klen);
			memcpy(J->sigs[i] + J->klen, J->b6 + (size_t)j * J->klen, J->klen);
		}
		J->ret_items[i] = J->pre[j] ? -1 : 0;
	}
}

static int eddsa_sign_group(sign_job *J, u32 cnt)
{
	u32 j;
	J->clen = J->e->clen;
	J->hlen = J->hm->digest_size;      /* 64 (SHA-512) / 114 (SHAKE256 as libecc configures it) */
	J->klen = J->hlen / 2;             /* EDDSA_R_LEN = EDDSA_S_LEN: 32 / 57 */
	J->ph_len = J->is448 ? 64 : J->hlen;   /* EDDSA448PH: SHAKE256 with 64 bytes (sig/eddsa.c:1650-1657) */
	if (J->klen != (J->is448 ? 57u : 32u) || J->clen != (J->is448 ? 56u : 32u)) {
		return -1;
	}
	if ((u32)J->siglen != J->hlen) {   /* MUST_HAVE((siglen == EDDSA_SIGLEN(hsize))), sig/eddsa.c:1638 */
		for (j = 0; j < cnt; j++) {
			J->ret_items[J->idx[j]] = -1;
		}
		return 0;
	}
	J->b0 = buf_get(0, (size_t)cnt * 3 * J->clen);
	J->b1 = buf_get(1, (size_t)cnt * J->klen);
	J->b2 = buf_get(2, (size_t)cnt * J->hlen);
	J->b3 = buf_get(3, (size_t)cnt * J->klen);
	J->b4 = buf_get(4, (size_t)cnt * J->klen);
	J->b5 = buf_get(5, (size_t)cnt * J->hlen);
	J->b6 = buf_get(6, (size_t)cnt * J->klen);
	J->b7 = buf_get(7, cnt);
	J->pre = buf_get(8, cnt);
	if (!J->b0 || !J->b1 || !J->b2 || !J->b3 || !J->b4 || !J->b5 || !J->b6 || !J->b7 || !J->pre) {
		return -1;
	}
	/* 1. the public keys as the hash takes them: exported as points here, encoded on the device (pre[j] != 0: no encoding) */
	parallel_for(cnt, eddsa_sign_keys, J);
	if (ecamd_multi_eddsa_encode_point_batch(g_multi, J->e->mc, cnt, J->b0, J->b1, J->pre)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	/* 2. r = H(dom || prefix || PH(M)) on the host | R = [r]G encoded on the device | H(dom || R || A || PH(M)) on the host */
	if (pipeline_run(cnt, chunk_items_for(g_chunk, cnt), eddsa_sign_pack, eddsa_sign_gpu_R, eddsa_sign_hram, J)) {
		return -1;
	}
	/* 3. S = (r + h a) mod q */
	if (ecamd_multi_eddsa_sign_S_batch(g_multi, J->e->mc, cnt, J->b2, J->b5, J->b3, J->b6)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	parallel_for(cnt, eddsa_sign_out, J);
	note_items(cnt);
	return 0;
}

/* every other algorithm: libecc's own _ec_sign, item by item on the host threads */
static void cpu_sign_items(u32 lo, u32 hi, void *arg)
{
	sign_job *J = (sign_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		J->ret_items[i] = (J->kps[i] && J->sigs[i] && (J->m[i] || !J->m_len[i])) ?
			_ec_sign(J->sigs[i], J->siglen, J->kps[i], J->m[i], J->m_len[i], J->rand, J->sig_type, J->hash_type,
				 J->adata ? J->adata[i] : NULL, J->adata_len ? J->adata_len[i] : 0) : -1;
		if (J->ret_items[i]) {
			J->ret_items[i] = -1;
		}
	}
}

/* the common shape of a signing batch -- every key pair present, initialised and of ONE set of parameters -- found on the pool, with
 * the initial -1 of every item and the identity index map (cf. scan_keys of the verification side) */
typedef struct {
	const ec_key_pair *const *kps;
	const ec_params *params;
	u32 *idx;
	int *rets;
	u32 mixed;
} sign_scan_job;
static void sign_scan(u32 lo, u32 hi, void *arg)
{
	sign_scan_job *S = (sign_scan_job *)arg;
	u32 i, mixed = 0;
	for (i = lo; i < hi; i++) {
		const ec_key_pair *kp = S->kps[i];
		S->rets[i] = -1;
		S->idx[i] = i;
		if (!kp || kp->priv_key.magic != PRIV_KEY_MAGIC || kp->priv_key.params != S->params) {
			mixed = 1;
		}
	}
	if (mixed) {
		AT_STORE(&S->mixed, 1);
	}
}

int ec_sign_batch(u8 *const *sigs, u8 siglen, const ec_key_pair *const *key_pairs, const u8 *const *m, const u32 *m_len, u32 num,
		  int (*rand)(nn_t out, nn_src_t q), ec_alg_type sig_type, hash_alg_type hash_type, const u8 *const *adata,
		  const u16 *adata_len, int *ret_items)
{
	const hash_mapping *hm;
	const ec_sig_mapping *sm = NULL;
	hash_alg_type eh = UNKNOWN_HASH_ALG;
	ec_curve_type ec = UNKNOWN_CURVE;
	int ph = 0, dom = 0, is448 = 0, ed, ret = -1, *rets = ret_items, one_group = 0;
	u32 *idx = NULL, i, done = 0;
	u8 *seen = NULL;
	if (!sigs || !key_pairs || !m || !m_len) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	if (get_sig_by_type(sig_type, &sm) || !sm) {
		return -1;   /* _ec_sign: unknown algorithm (sig/sig_algs.c:476) */
	}
	ed = !eddsa_variant(sig_type, &eh, &ec, &ph, &dom, &is448);
	if (!rets) {
		rets = (int *)malloc((size_t)num * sizeof(int));
		if (!rets) {
			return -1;
		}
	}
	hm = find_hash(hash_type);
	idx = (u32 *)malloc((size_t)num * sizeof(u32));
	if (!idx || !hm || (ed && (hash_type != eh || rand != NULL)) || ecamd_compat_init(NULL, 0, 0)) {
		for (i = 0; i < num; i++) {
			rets[i] = -1;
		}
		/* every ec_sign fails: unknown hash (_ec_sign_init), or EdDSA with another hash / a nonce source (sig/eddsa.c:1596,1612) */
		ret = (idx && (!hm || (ed && (hash_type != eh || rand != NULL)))) ? 0 : -1;
		goto out;
	}
	call_enter(0);
	{
		sign_scan_job S;
		S.kps = key_pairs;
		S.params = (key_pairs[0] && key_pairs[0]->priv_key.magic == PRIV_KEY_MAGIC) ? key_pairs[0]->priv_key.params : NULL;
		S.idx = idx;
		S.rets = rets;
		S.mixed = S.params ? 0 : 1;
		parallel_for(num, sign_scan, &S);
		one_group = !AT_LOAD(&S.mixed);
	}
	if (!one_group) {
		seen = (u8 *)calloc(num, 1);
		if (!seen) {
			call_leave();
			goto out;
		}
	}
	/* groups of items that share their ec_params (one GPU batch each; normally there is one group) */
	ret = 0;
	while (done < num && !ret) {
		const ec_params *params = NULL;
		sign_job J;
		u32 cnt = 0;
		if (one_group) {
			params = key_pairs[0]->priv_key.params;
			cnt = done = num;
		}
		for (i = 0; i < num && !one_group; i++) {
			const ec_key_pair *kp = key_pairs[i];
			if (seen[i]) {
				continue;
			}
			if (!kp || kp->priv_key.magic != PRIV_KEY_MAGIC || !kp->priv_key.params) {
				seen[i] = 1;   /* stays -1 */
				done++;
				continue;
			}
			if (!params) {
				params = kp->priv_key.params;
			}
			if (kp->priv_key.params == params) {
				idx[cnt++] = i;
				seen[i] = 1;
				done++;
			}
		}
		if (!cnt) {
			break;
		}
		memset(&J, 0, sizeof(J));
		J.sigs = sigs; J.kps = key_pairs; J.m = m; J.m_len = m_len; J.adata = adata; J.adata_len = adata_len; J.rand = rand;
		J.sig_type = sig_type; J.hash_type = hash_type; J.hm = hm; J.idx = idx; J.params = params; J.ret_items = rets; J.siglen = siglen;
		J.ph = ph; J.dom = dom; J.is448 = is448;
		if (ed || is_ecdsa(sig_type)) {
			if (ed && params->curve_type != ec) {
				continue;   /* eddsa_key_type_check_curve fails: -1 for the group */
			}
			J.e = curve_from_params(params);
			if (!J.e) {
				ret = -1;
				break;
			}
			ret = ed ? eddsa_sign_group(&J, cnt) : ecdsa_sign_group(&J, cnt);
		} else {
			parallel_for(cnt, cpu_sign_items, &J);
		}
	}
	wipe_secrets();
	call_leave();
out:
	if (rets != ret_items) {
		free(rets);
	}
	free(idx);
	free(seen);
	return ret;
}

/* ------------------------------------------------------------------------------------------------
 * signature verification
 * ------------------------------------------------------------------------------------------------ */
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
	curve_ent *e;
	u8 *pk, *sg, *dg, *pre, *res;  /* packed: keys (projective X||Y||Z or EdDSA encoding), signatures, digests / hram, pre-check, result */
	u8 *kprj;                /* EdDSA: the key points X||Y||Z, input of the device-side encoding */
	u32 clen, qlen, hlen, klen, siglen;
	int ph, dom, is448;
	u32 ph_len;              /* bytes of PH(M) that enter the main hash */
	int all_only, all_ok;    /* EdDSA: only the conjunction is wanted; its value so far */
	int aff_keys;            /* ECDSA, round 4: every key of the group has Z = 1 -- they travel as affine X || Y and the device skips its
	                          * projective import (k_prj_import: a quarter of the kernel time of a secp256r1 verification otherwise) */
	u32 kw;                  /* ... octets of a packed key: 2 * clen or 3 * clen */
	int dev_hash;            /* round 4: SHA-2 of short messages on the device -- 0 host hashing, else the hash_alg_type number */
	u32 slot;                /* ... stride of a message slot in dg (u32 length + bytes), a multiple of 4 */
	int *results;            /* the caller's per-item results (written by the unpack step of a chunk, on the pool) */
	u8 *ms;                  /* one-pass EDDSA25519PH: the messages in slots of their own (hashed on the device into the hash input) */
	u32 mslot;
	u32 a_off;               /* one-pass EdDSA: where the key's encoding goes in the hash input (after dom2 and R) */
	u32 dom_len;             /* ... octets of the dom2 prefix in front of R (0: plain Ed25519) */
	int pre_scanned;         /* verify_results' one pass over the keys already found: */
	u32 pre_max_mlen;        /* ... the longest message of the group */
	u32 pre_not_affine;      /* ... whether some usable key has Z != 1 */
	u32 fail_tracked, any_fail;   /* ver_unpack notes whether any item was rejected (ec_verify_batch wants that one bit) */
	u32 *broken;             /* verify_results skipped its pass over the keys (one set of parameters, keys as the sampled ones: taken for granted);
	                          * a packing step that meets a key that says otherwise sets this and the call starts over with the pass */
	int assumed_affine;      /* ... Z = 1 is a guess from a sample of the keys: the packing steps that drop Z look at it */
} ver_job;

/* a key the packing steps cannot use: missing, not a public key, under other parameters (-> *broken: in a batch of several sets of parameters
 * that key belongs to another group; verify_results sorts that out), or not a key of this algorithm (an item failure in any grouping) */
static int key_unusable(const ver_job *J, const ec_pub_key *pk)
{
	if (!pk || pk->magic != PUB_KEY_MAGIC || pk->params != J->params) {
		if (J->broken) {
			AT_STORE(J->broken, 1);
		}
		return 1;
	}
	return pub_key_check_initialized_and_type(pk, J->sig_type) ? 1 : 0;
}
/* The packing steps walk a million 0.7 KB key structures and look at a dozen cache lines of each (the magic words of the key, of its point, of
 * the three coordinates and of their numbers sit at both ends of every structure): the lines of the key a few items ahead are asked for while
 * this one is packed ($ECAMD_COMPAT_NO_PREFETCH: off). */
#define KEY_PREFETCH_AHEAD 4u
static int key_prefetch_on(void)
{
	static int on = -1;
	if (on < 0) {
		on = getenv("ECAMD_COMPAT_NO_PREFETCH") ? 0 : 1;
	}
	return on;
}
static inline void key_prefetch(const ver_job *J, u32 j, u32 hi)
{
	if (j + KEY_PREFETCH_AHEAD < hi && key_prefetch_on()) {
		const char *b = (const char *)J->pub_keys[J->idx[j + KEY_PREFETCH_AHEAD]];
		size_t o;
		for (o = 0; b && o < sizeof(ec_pub_key); o += 64) {
			__builtin_prefetch(b + o, 0, 0);
		}
	}
}
/* Z of X || Y || Z (clen big-endian octets each) is not 1 although the group was packed as affine on a guess */
static void affine_guess_check(const ver_job *J, const u8 *xyz, u32 cl)
{
	u32 k, nz = 0;
	if (!J->assumed_affine || !J->broken) {
		return;
	}
	for (k = 0; k + 1 < cl; k++) {
		nz |= xyz[2 * cl + k];
	}
	if (nz || xyz[3 * cl - 1] != 1) {
		AT_STORE(J->broken, 1);
	}
}

static u32 dev_hash_slot(const ver_job *J, u32 cnt, u32 extra)
{
	return dev_hash_slot_for(J->m_len, J->idx, cnt, extra, J->pre_scanned ? (long)J->pre_max_mlen : -1);
}

/* ---- ECDSA / DECDSA ---- */
static void ecdsa_pack(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_pub_key *pk = J->pub_keys[i];
		hash_context hc;
		int bad;
		key_prefetch(J, j, hi);
		/* ec_verify_init / __ecdsa_verify_init (sig/sig_algs.c:516, sig/ecdsa_common.c:623-649): key initialised and of
		 * this algorithm, signature present and of the expected length; then H(m).  The key leaves as the live limbs of
		 * pk->y (what ec_pub_key_export_to_buf writes); the device checks the curve equation at import, as the reference's
		 * multiplication does (curves/prj_pt.c:1765). */
		bad = key_unusable(J, pk) || !J->s[i] ||
		      J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
		if (!bad && J->aff_keys) {
			/* (Z = 1 was established for the whole group by ecdsa_keys_affine) */
			u8 tmp[3 * 72];
			bad = prj_to_be(tmp, J->clen, &pk->y, &(J->params->ec_curve));
			if (!bad) {
				affine_guess_check(J, tmp, J->clen);
			}
			memcpy(J->pk + (size_t)j * J->kw, tmp, J->kw);
		} else {
			bad = bad || prj_to_be(J->pk + (size_t)j * J->kw, J->clen, &pk->y, &(J->params->ec_curve));
		}
		if (J->dev_hash) {
			if (!bad) {
				slot_put(J->dg + (size_t)j * J->slot, J->slot, NULL, 0, NULL, 0, J->m[i], J->m_len[i]);
			}
		} else {
			bad = bad || J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]) ||
			      J->hm->hfunc_finalize(&hc, J->dg + (size_t)j * J->hlen);
		}
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(J->pk + (size_t)j * J->kw, 0xff, J->kw);
			memset(J->sg + (size_t)j * J->siglen, 0, J->siglen);
			memset(J->dg + (size_t)j * (J->dev_hash ? J->slot : J->hlen), 0, J->dev_hash ? J->slot : J->hlen);
		} else {
			memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
		}
	}
}

static void ver_unpack(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	u32 j;
	u32 bad = 0;
	for (j = lo; j < hi; j++) {
		const int r = (J->pre[j] || J->res[j]) ? -1 : 0;
		J->results[J->idx[j]] = r;
		bad |= (u32)(r != 0);
	}
	if (bad) {
		AT_STORE(&J->any_fail, 1);
	}
}

static int ecdsa_ver_gpu(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	const int fmt = J->aff_keys ? ECAMD_PT_AFFINE : ECAMD_PT_PROJECTIVE;
	const int r = J->dev_hash
			      ? ecamd_multi_ecdsa_verify_msg_batch_fmt(g_multi, J->e->mc, hi - lo, J->pk + (size_t)lo * J->kw, fmt,
								       J->sg + (size_t)lo * J->siglen, J->dev_hash, J->dg + (size_t)lo * J->slot, J->slot,
								       J->res + lo)
			      : ecamd_multi_ecdsa_verify_batch_fmt(g_multi, J->e->mc, hi - lo, J->pk + (size_t)lo * J->kw, fmt,
								   J->sg + (size_t)lo * J->siglen, J->dg + (size_t)lo * J->hlen, J->hlen, J->res + lo);
	if (r) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

/* does every usable key of the group have Z = 1 (the form every import and every batch result leaves)?  Keys that fail the
 * checks of the pack step do not count: they are rejected there whatever their format. */
typedef struct {
	const ver_job *J;
	u32 not_affine;
} affk_job;
static void ecdsa_keys_affine(u32 lo, u32 hi, void *arg)
{
	affk_job *A = (affk_job *)arg;
	u32 j;
	for (j = lo; j < hi && !AT_LOAD(&A->not_affine); j++) {
		const ec_pub_key *pk = A->J->pub_keys[A->J->idx[j]];
		int one = 0;
		if (key_unusable(A->J, pk) || prj_pt_check_initialized(&pk->y) ||
		    fp_check_initialized(&pk->y.Z)) {
			continue;
		}
		if (nn_isone(&pk->y.Z.fp_val, &one) || !one) {
			AT_STORE(&A->not_affine, 1);
		}
	}
}

/* results[i] for the items idx[0..cnt) that share `params` */
static int ecdsa_group(ver_job *J, u32 cnt, int *results)
{
	affk_job AK;
	J->clen = J->e->clen;
	J->qlen = J->e->qlen;
	J->siglen = 2 * (u32)BYTECEIL(J->params->ec_gen_order_bitlen);   /* ECDSA_SIGLEN */
	if (J->siglen != 2 * J->e->qlen) {
		return -1;
	}
	J->hlen = J->hm->digest_size;
	J->dev_hash = dev_hash_type(J->hm);
	J->slot = J->dev_hash ? dev_hash_slot(J, cnt, 0) : 0;
	if (!J->slot) {
		J->dev_hash = 0;
	}
	AK.J = J;
	AK.not_affine = getenv("ECAMD_COMPAT_PRJ_KEYS") ? 1 : 0;
	if (!AK.not_affine && J->pre_scanned) {
		AK.not_affine = J->pre_not_affine;
	} else if (!AK.not_affine) {
		parallel_for(cnt, ecdsa_keys_affine, &AK);
	}
	J->aff_keys = AT_LOAD(&AK.not_affine) ? 0 : 1;
	J->kw = (J->aff_keys ? 2u : 3u) * J->clen;
	J->pk = buf_get(0, (size_t)cnt * J->kw);
	J->sg = buf_get(1, (size_t)cnt * J->siglen);
	J->dg = buf_get(2, (size_t)cnt * (J->dev_hash ? J->slot : J->hlen));
	J->pre = buf_get(3, cnt);
	J->res = buf_get(4, cnt);
	if (!J->pk || !J->sg || !J->dg || !J->pre || !J->res) {
		return -1;
	}
	J->results = results;
	J->fail_tracked = 1;
	if (verify_pipeline(cnt, ecdsa_pack, ecdsa_ver_gpu, ver_unpack, J)) {
		return -1;
	}
	note_items(cnt);
	return 0;
}

/* ---- EdDSA ---- */
/* first pass of an EdDSA group: the projective key points as octets for ec_eddsa_encode_point_batch */
static void eddsa_export_keys(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const ec_pub_key *pk = J->pub_keys[J->idx[j]];
		u8 *dst = J->kprj + (size_t)j * 3 * J->clen;
		if (key_unusable(J, pk) ||
		    prj_to_be(dst, J->clen, &pk->y, &(J->params->ec_curve))) {
			memset(dst, 0xff, (size_t)3 * J->clen);   /* coordinates >= p: an import error on the device */
		}
	}
}

static void eddsa_pack(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
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
		bad = key_unusable(J, pk) || !J->s[i] ||
		      J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
#if defined(WITH_SIG_EDDSA25519)
		bad = bad || (J->sig_type == EDDSA25519CTX && !ad);
#endif
		/* the encoding of the key as the reference hashes it -- eddsa_export_pub_key: Weierstrass -> Edwards -> octets -- was
		 * computed on the device for the whole group (eddsa_group; libecc's own export costs about 1.5 ms of CPU per key) */
		bad = bad || J->pre[j];
		if (J->dev_hash) {
			/* hram = SHA-512(R || A || M) on the device: the slot holds the hash input */
			if (!bad) {
				slot_put(J->dg + (size_t)j * J->slot, J->slot, J->s[i], J->klen, kenc, J->klen, J->m[i], J->m_len[i]);
				memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
			} else {
				memset(kenc, 0xff, J->klen);
				memset(J->sg + (size_t)j * J->siglen, 0xff, J->siglen);
				memset(J->dg + (size_t)j * J->slot, 0, J->slot);
			}
			J->pre[j] = bad ? 1 : 0;
			continue;
		}
		bad = bad || J->hm->hfunc_init(&hc);
		if (!bad && J->dom) {
			bad = dom_prefix(J->hm, &hc, J->is448, J->ph, ad, adl);
		}
		bad = bad || J->hm->hfunc_update(&hc, J->s[i], J->klen) || J->hm->hfunc_update(&hc, kenc, J->klen);
		if (!bad && J->ph) {
			bad = J->hm->hfunc_init(&hp) || J->hm->hfunc_update(&hp, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hp, dig) ||
			      J->hm->hfunc_update(&hc, dig, J->ph_len);
		} else if (!bad) {
			bad = J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]);
		}
		bad = bad || J->hm->hfunc_finalize(&hc, J->dg + (size_t)j * J->hlen);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(kenc, 0xff, J->klen);   /* y >= p: rejected by the decoder */
			memset(J->sg + (size_t)j * J->siglen, 0xff, J->siglen);
			memset(J->dg + (size_t)j * J->hlen, 0, J->hlen);
		} else {
			memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
		}
	}
}

/* EDDSA25519 with the hashing on the device, round 4: ONE pass.  The key leaves as the projective point the ec_pub_key holds; the device
 * imports it, encodes it the way eddsa_export_pub_key does and writes the 32 octets into the hash input R || A || M itself
 * (ec_eddsa_verify_msg_prj_batch), so the separate encoding call of the whole group -- and with it the only reason the packing could
 * not start before a GPU round trip -- is gone.  $ECAMD_COMPAT_ED_TWO_PASS keeps the two calls. */
static void eddsa_pack_prj(u32 lo, u32 hi, void *arg)
{
	static const u8 blank[57 + 64] = {0};   /* the blank for A (32 / 57 octets), or for A || PH(M) */
	ver_job *J = (ver_job *)arg;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_pub_key *pk = J->pub_keys[i];
		u8 *kdst = J->kprj + (size_t)j * 3 * J->clen;
		/* _eddsa_verify_init (sig/eddsa.c:1880-1960), as in eddsa_pack */
		int bad = (key_prefetch(J, j, hi), key_unusable(J, pk)) || !J->s[i] ||
			  J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
		bad = bad || prj_to_be(kdst, J->clen, &pk->y, &(J->params->ec_curve));
		if (!bad && J->dom_len) {
			/* EDDSA25519CTX / PH, EDDSA448 / PH: dom2 / dom4(phflag, context) in front of R (sig/eddsa.c:56-84); the group's contexts have
			 * one length (eddsa_group); EDDSA25519CTX wants a context, the others take one or none */
			const u8 *ad = J->adata ? J->adata[i] : NULL;
			const u32 pl = J->is448 ? 8u : 32u, adl = J->dom_len - pl - 2u;
			u8 head[32 + 2 + 255 + 57];
			bad = (u32)(J->adata_len ? J->adata_len[i] : 0) != adl || ((J->ph || J->is448) ? (adl && !ad) : !ad);
			if (!bad) {
				memcpy(head, J->is448 ? "SigEd448" : "SigEd25519 no Ed25519 collisions", pl);
				head[pl] = (u8)(J->ph ? 1 : 0);
				head[pl + 1] = (u8)adl;
				if (adl) {
					memcpy(head + pl + 2, ad, adl);
				}
				memcpy(head + J->dom_len, J->s[i], J->klen);
				if (J->ph) {
					slot_put(J->dg + (size_t)j * J->slot, J->slot, head, J->dom_len + J->klen, blank, J->klen + 64u, NULL, 0);
					slot_put(J->ms + (size_t)j * J->mslot, J->mslot, NULL, 0, NULL, 0, J->m[i], J->m_len[i]);
				} else {
					slot_put(J->dg + (size_t)j * J->slot, J->slot, head, J->dom_len + J->klen, blank, J->klen, J->m[i], J->m_len[i]);
				}
				memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
			}
		} else if (!bad) {
			slot_put(J->dg + (size_t)j * J->slot, J->slot, J->s[i], J->klen, blank, J->klen, J->m[i], J->m_len[i]);
			memcpy(J->sg + (size_t)j * J->siglen, J->s[i], J->siglen);
		}
		if (bad) {
			memset(kdst, 0xff, (size_t)3 * J->clen);   /* coordinates >= p: an import error on the device */
			memset(J->sg + (size_t)j * J->siglen, 0xff, J->siglen);
			memset(J->dg + (size_t)j * J->slot, 0, J->slot);
			if (J->ph) {
				memset(J->ms + (size_t)j * J->mslot, 0, J->mslot);
			}
		}
		J->pre[j] = bad ? 1 : 0;
	}
}

static int eddsa_ver_gpu_prj(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	if (J->ph) {
		if (ecamd_multi_eddsa_verify_ph_prj_batch(g_multi, J->e->mc, hi - lo, J->kprj + (size_t)lo * 3 * J->clen, J->sg + (size_t)lo * J->siglen,
							  J->dg + (size_t)lo * J->slot, J->slot, J->a_off, J->ms + (size_t)lo * J->mslot, J->mslot, J->res + lo)) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			return -1;
		}
		return 0;
	}
	if (ecamd_multi_eddsa_verify_msg_prj_batch(g_multi, J->e->mc, hi - lo, J->kprj + (size_t)lo * 3 * J->clen, J->sg + (size_t)lo * J->siglen,
						   J->dg + (size_t)lo * J->slot, J->slot, J->a_off, J->res + lo)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

/* Round 6: only the batch bit is wanted and the batch is large -- the same packed arrays as ONE streamed call whose front end files
 * every chunk and whose end is the reference's batch equation over the whole batch (ec_eddsa_verify_msg_prj_all_batch: by buckets).  The
 * z_i are keyed through the application's get_random, as in eddsa_ver_gpu_all.  An item that fails a host-side check makes the attempt
 * void (J->all_ok = 0): the caller then verifies item by item. */
static unsigned long g_ed_msm_calls;
unsigned long ecamd_compat_ed_msm_calls(void) { return AT_LOAD(&g_ed_msm_calls); }
static int eddsa_ver_gpu_prj_all(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	int all = 0;
	u8 seed[32];
	int r;
	if (AT_LOAD(&g_rand_concurrent)) {
		r = get_random(seed, sizeof(seed));
	} else {
		pthread_mutex_lock(&g_rand_mu);
		r = get_random(seed, sizeof(seed));
		pthread_mutex_unlock(&g_rand_mu);
	}
	r = r || ecamd_multi_set_msm_seed(g_multi, seed);
	wipe(seed, sizeof(seed));
	if (r) {
		return -1;
	}
	if (ecamd_multi_eddsa_verify_msg_prj_all_batch(g_multi, J->e->mc, hi - lo, J->kprj + (size_t)lo * 3 * J->clen, J->sg + (size_t)lo * J->siglen,
						       J->dg + (size_t)lo * J->slot, J->slot, J->a_off, &all)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	AT_ADD(&g_ed_msm_calls, 1);
	if (!all) {
		J->all_ok = 0;
	}
	return 0;
}
/* (the pre-checks of the items: any failure voids the whole-batch attempt) */
static void eddsa_any_pre(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	u32 j, bad = 0;
	for (j = lo; j < hi; j++) {
		bad |= J->pre[j];
	}
	if (bad) {
		AT_STORE(&J->any_fail, 1);
	}
}
static int ed_msm_wanted(u32 cnt)
{
	unsigned long min_items = 1ul << 18;
	const char *e = getenv("ECAMD_COMPAT_ED_MSM_MIN");
	const int ranks = ecamd_multi_size(g_multi);
	if (e) {
		min_items = strtoul(e, NULL, 10);
		if (min_items == 0) {
			return 0;
		}
	}
	return ranks > 0 && (unsigned long)cnt / (unsigned long)ranks >= min_items;
}

static int eddsa_ver_gpu(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	const int r = J->dev_hash ? ecamd_multi_eddsa_verify_msg_batch(g_multi, J->e->mc, hi - lo, J->pk + (size_t)lo * J->klen,
								       J->sg + (size_t)lo * J->siglen, J->dg + (size_t)lo * J->slot, J->slot, J->res + lo)
				  : ecamd_multi_eddsa_verify_batch(g_multi, J->e->mc, hi - lo, J->pk + (size_t)lo * J->klen, J->sg + (size_t)lo * J->siglen,
								   J->dg + (size_t)lo * J->hlen, J->hlen, J->res + lo);
	if (r) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	return 0;
}

/* the caller only wants ec_verify_batch's one bit: a chunk is decided by the device's multi-scalar multiplication (the
 * reference's own random linear combination, ec_eddsa_verify_all_batch) after the host-side checks of its items */
static int eddsa_ver_gpu_all(u32 lo, u32 hi, void *arg)
{
	ver_job *J = (ver_job *)arg;
	int all = 0;
	u32 j;
	if (!J->all_ok) {
		return 0;   /* already rejected: the remaining chunks cannot change the answer */
	}
	for (j = lo; j < hi; j++) {
		if (J->pre[j]) {
			J->all_ok = 0;
			return 0;
		}
	}
	{
		/* the z_i of the combination are keyed by the application's own randomness source, the import libecc draws them from
		 * (sig/eddsa.c:2388; SURVEY.md 8b: get_random stays the application's) */
		u8 seed[32];
		int r = 0;
		if (!J->is448) {   /* only the Ed25519 combination consumes a seed; the engine discards an unused one when the call returns */
			if (AT_LOAD(&g_rand_concurrent)) {
				r = get_random(seed, sizeof(seed));
			} else {
				pthread_mutex_lock(&g_rand_mu);
				r = get_random(seed, sizeof(seed));
				pthread_mutex_unlock(&g_rand_mu);
			}
			r = r || ecamd_multi_set_msm_seed(g_multi, seed);
			wipe(seed, sizeof(seed));
			if (r) {
				return -1;
			}
		}
	}
	if (ecamd_multi_eddsa_verify_all_batch(g_multi, J->e->mc, hi - lo, J->pk + (size_t)lo * J->klen, J->sg + (size_t)lo * J->siglen,
					       J->dg + (size_t)lo * J->hlen, J->hlen, &all, NULL)) {
		fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
		return -1;
	}
	if (!all) {
		J->all_ok = 0;
	}
	return 0;
}

/* do the items of an EDDSA25519CTX group carry contexts of one length?  (items without one fail in the pack step whatever the answer) */
typedef struct {
	const ver_job *J;
	u32 first, mixed;
} adl_job;
static void adl_scan(u32 lo, u32 hi, void *arg)
{
	adl_job *A = (adl_job *)arg;
	u32 j, mixed = 0;
	for (j = lo; j < hi; j++) {
		const u32 i = A->J->idx[j];
		const u32 adl = A->J->adata_len ? A->J->adata_len[i] : 0;
		const u8 *ad = A->J->adata ? A->J->adata[i] : NULL;
		if ((A->J->ph || A->J->is448) ? (adl != A->first || (adl && !ad)) : (ad && adl != A->first)) {
			mixed = 1;   /* (pre-hashed variants and Ed448: the context is optional, its length octet is hashed either way) */
		}
	}
	if (mixed) {
		AT_STORE(&A->mixed, 1);
	}
}

static int eddsa_group(ver_job *J, u32 cnt, int *results)
{
	u32 j;
	int one_pass = 0, ctx_uniform = 0, ed_hash_on_device = 0;
	J->clen = J->e->clen;
	J->hlen = J->hm->digest_size;      /* 64 (SHA-512) / 114 (SHAKE256 as libecc configures it) */
	J->klen = J->hlen / 2;             /* EDDSA_R_LEN: 32 / 57 */
	J->siglen = J->hlen;               /* EDDSA_SIGLEN */
	if (J->klen != (J->is448 ? 57u : 32u) || J->clen != (J->is448 ? 56u : 32u)) {
		return -1;
	}
	/* hashing on the device: the plain Ed25519 variant (no dom2 prefix, no pre-hash), item-by-item results */
	J->dev_hash = 0;
	J->slot = 0;
#if defined(WITH_SIG_EDDSA25519)
	J->dom_len = 0;
	if (J->dom && cnt) {   /* EDDSA25519CTX / PH, EDDSA448 / PH */
		/* the contexts of a batch normally have one length: then dom2 || R || A sits at one offset in every hash input */
		adl_job A;
		A.J = J;
		A.first = J->adata_len ? J->adata_len[J->idx[0]] : 0;
		A.mixed = A.first > 255 ? 1 : 0;
		if (!A.mixed) {
			parallel_for(cnt, adl_scan, &A);
		}
		ctx_uniform = !AT_LOAD(&A.mixed);
		J->dom_len = ctx_uniform ? (J->is448 ? 10u : 34u) + A.first : 0;
	}
	/* the variant's own hash on the device: SHA-512 (k_sha2_slots) for Ed25519, SHAKE256 (k_shake256_slots) for Ed448 */
	ed_hash_on_device = !getenv("ECAMD_COMPAT_HOST_HASH") && (J->is448 ? J->hm->type == SHAKE256 : J->hm->type == SHA512);
	if (((!J->dom && !J->ph) || ctx_uniform) && ed_hash_on_device) {
		/* (also when only the conjunction is wanted: since the half-length scalars of round 4 the item-by-item verification of 2^20
		 * signatures takes the 12 ms the multi-scalar combination takes, needs no z_i, and does not wait for the host to hash) */
		if (J->ph) {
			/* pre-hashed: the hash input is dom2 || R || A || PH(M) -- a fixed length --, the messages travel in slots of their own */
			J->mslot = dev_hash_slot(J, cnt, 0);
			J->slot = J->mslot ? ((4u + J->dom_len + 2 * J->klen + 64u + 3u) & ~3u) : 0;
			if (J->slot > DEV_HASH_MAX_SLOT) {
				J->slot = 0;
			}
		} else {
			J->slot = dev_hash_slot(J, cnt, J->dom_len + 2 * J->klen);
		}
		J->a_off = J->dom_len + J->klen;
		one_pass = J->slot && !getenv("ECAMD_COMPAT_ED_TWO_PASS");
		/* (the two-pass path builds its hash inputs for the plain variant only: CTX / PH hash on the host there) */
		J->dev_hash = (J->slot && (one_pass || (!J->all_only && !J->dom_len && !J->ph))) ? 4 : 0;
		if (!J->dev_hash) {
			J->slot = 0;
		}
	}
#endif
	J->pk = buf_get(0, (size_t)cnt * J->klen);
	J->sg = buf_get(1, (size_t)cnt * J->siglen);
	J->dg = buf_get(2, (size_t)cnt * (J->dev_hash ? J->slot : J->hlen));
	J->pre = buf_get(3, cnt);
	J->res = buf_get(4, cnt);
	J->kprj = buf_get(5, (size_t)cnt * 3 * J->clen);
	J->ms = (one_pass && J->ph) ? buf_get(6, (size_t)cnt * J->mslot) : NULL;
	if (!J->pk || !J->sg || !J->dg || !J->pre || !J->res || !J->kprj || (one_pass && J->ph && !J->ms)) {
		return -1;
	}
	if (getenv("ECAMD_COMPAT_TIMING")) {
		fprintf(stderr, "libecc_amd compat timing: EdDSA group of %u items: %s, dom2 prefix %u octets\n", cnt, one_pass ? "one device call" : "two passes", J->dom_len);
	}
	if (one_pass && J->all_only && !J->ph && ed_msm_wanted(cnt)) {
		/* ec_verify_batch of a large Ed25519 / Ed25519ctx / Ed448 batch: the batch equation first (round 6; Ed448: on the Weierstrass model with
		 * the cofactored final test); it vouches for VALID batches only */
		J->all_ok = 1;
		J->any_fail = 0;
		if (verify_pipeline(cnt, eddsa_pack_prj, eddsa_ver_gpu_prj_all, NULL, J)) {
			fprintf(stderr, "libecc_amd compat: the whole-batch form failed (%s); verifying item by item\n", ecamd_last_error());
		} else {
			parallel_for(cnt, eddsa_any_pre, J);
			if (J->all_ok && !AT_LOAD(&J->any_fail)) {
				note_items(cnt);
				for (j = 0; j < cnt; j++) {
					results[J->idx[j]] = 0;
				}
				J->fail_tracked = 1;   /* (and nothing failed) */
				return 0;
			}
		}
		J->any_fail = 0;
	}
	if (one_pass) {
		J->results = results;
		J->fail_tracked = 1;
		if (verify_pipeline(cnt, eddsa_pack_prj, eddsa_ver_gpu_prj, ver_unpack, J)) {
			return -1;
		}
		note_items(cnt);
		return 0;
	}
	/* the keys as the reference hashes them: exported as points here, encoded on the device (pre[j] != 0: no encoding) */
	parallel_for(cnt, eddsa_export_keys, J);
	{
		int r;
		pthread_mutex_lock(&g_gpu_mu);   /* (a C-ABI call outside a pipeline: see g_gpu_mu) */
		r = ecamd_multi_eddsa_encode_point_batch(g_multi, J->e->mc, cnt, J->kprj, J->pk, J->pre);
		pthread_mutex_unlock(&g_gpu_mu);
		if (r) {
			fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
			return -1;
		}
	}
	if (J->all_only) {
		J->all_ok = 1;
		if (pipeline_run(cnt, chunk_items_for(g_chunk_all, cnt), eddsa_pack, eddsa_ver_gpu_all, NULL, J)) {
			return -1;
		}
		note_items(cnt);
		for (j = 0; j < cnt; j++) {
			results[J->idx[j]] = J->all_ok ? 0 : -1;   /* only their conjunction is looked at */
		}
		return 0;
	}
	J->results = results;
	J->fail_tracked = 1;
	if (verify_pipeline(cnt, eddsa_pack, eddsa_ver_gpu, ver_unpack, J)) {
		return -1;
	}
	note_items(cnt);
	return 0;
}


/* ---- BIP0340 (Schnorr signatures as libecc implements them on any curve, sig/bip0340.c:383-560) and ECFSDSA (sig/ecfsdsa.c:404-580) ----
 * per item: Y = the key's unique representative with an even y (lift_x), r < p, s < q, e = H_tag(r || Y.x || m) mod q, then
 * R = [s]G + [q - e]Y must be finite, have an even y and x = r.  The normalisation of the keys, both scalar multiplications and
 * the addition run on the GPU(s) as whole-batch calls; the hashes and the byte checks on the host threads between them.
 * Buffers: 0 keys X||Y||Z, 1 keys affine -> Y, 2 their status, 3 s, 4 q - e, 5 [s]G, 6 its status, 7 [q - e]Y, 8 its status,
 * 9 the sum, 10 its status, 11 pre-check. */
typedef struct {
	ver_job v;
	int fs;                  /* 0: BIP0340, 1: ECFSDSA */
	u8 *kaff, *kst, *sc_s, *sc_e, *pA, *stA, *pB, *stB, *sum, *stS;
	u8 *wpt, *wst, *kinf;    /* ECFSDSA: the signatures' points W (validated on the device), their status; keys at infinity */
	u8 *rx;                  /* BIP0340: the signatures' r, n x clen (the abscissae of the multi-scalar form) */
	u8 p_be[80], q_be[80], tagd[MAX_DIGEST_SIZE];
} bip_job;

static int be_lt(const u8 *a, const u8 *b, u32 len)   /* a < b, big-endian */
{
	u32 i;
	for (i = 0; i < len; i++) {
		if (a[i] != b[i]) {
			return a[i] < b[i];
		}
	}
	return 0;
}

static void bip_export_keys(u32 lo, u32 hi, void *arg)
{
	bip_job *B = (bip_job *)arg;
	ver_job *J = &B->v;
	u32 j;
	for (j = lo; j < hi; j++) {
		const ec_pub_key *pk = J->pub_keys[J->idx[j]];
		u8 *dst = J->kprj + (size_t)j * 3 * J->clen;
		J->pre[j] = (key_unusable(J, pk) ||
			     prj_to_be(dst, J->clen, &pk->y, &(J->params->ec_curve))) ? 1 : 0;
		if (J->pre[j]) {
			memset(dst, 0xff, (size_t)3 * J->clen);
		}
	}
}

static void bip_pack(u32 lo, u32 hi, void *arg)
{
	bip_job *B = (bip_job *)arg;
	ver_job *J = &B->v;
	const u32 cl = J->clen, ql = J->qlen, rl = B->fs ? 2 * cl : cl;
	u32 j, k;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const u8 *sig = J->s[i];
		u8 *Y = B->kaff + (size_t)j * 2 * cl, dig[MAX_DIGEST_SIZE];
		hash_context hc;
		nn e;
		int bad;
		e.magic = WORD(0);
		bad = J->pre[j] || !sig || J->s_len[i] != J->siglen || (!J->m[i] && J->m_len[i]);
		if (!B->fs) {
			/* _bip0340_verify_init (sig/bip0340.c:383-462): the key's unique representative (a key at infinity fails in
			 * prj_pt_unique), r < p (fp_import_from_buf), s < q; e = H(H(tag) || H(tag) || r || Y.x || m) mod q (:45-69,
			 * :437-441, :470-494) */
			bad = bad || B->kst[j] != ECAMD_OK || !be_lt(sig, B->p_be, cl) || !be_lt(sig + cl, B->q_be, ql);
			bad = bad || J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, B->tagd, J->hm->digest_size) ||
			      J->hm->hfunc_update(&hc, B->tagd, J->hm->digest_size) || J->hm->hfunc_update(&hc, sig, cl) || J->hm->hfunc_update(&hc, Y, cl);
		} else {
			/* _ecfsdsa_verify_init (sig/ecfsdsa.c:404-470): r = W.x || W.y, both < p and on the curve (checked on the device:
			 * wst), s in [1, q - 1]; e = H(r || m) mod q.  The key is used as it is: at infinity it contributes nothing. */
			int snz = 0;
			for (k = 0; k < ql && sig; k++) {
				snz |= sig[2 * cl + k];
			}
			B->kinf[j] = (B->kst[j] == ECAMD_INF) ? 1 : 0;
			bad = bad || B->kst[j] == ECAMD_ERR || B->wst[j] != ECAMD_OK || !snz || !be_lt(sig + 2 * cl, B->q_be, ql);
			bad = bad || J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, sig, 2 * cl);
		}
		bad = bad || J->hm->hfunc_update(&hc, J->m[i], J->m_len[i]) || J->hm->hfunc_finalize(&hc, dig);
		/* ... then q - e (bip0340.c:531, ecfsdsa.c:561) */
		bad = bad || nn_init_from_buf(&e, dig, J->hm->digest_size) || nn_mod(&e, &e, &(J->params->ec_gen_order)) ||
		      nn_mod_neg(&e, &e, &(J->params->ec_gen_order)) || nn_to_be(B->sc_e + (size_t)j * ql, ql, &e);
		nn_uninit(&e);
		J->pre[j] = bad ? 1 : 0;
		if (bad) {
			memset(B->sc_s + (size_t)j * ql, 0, ql);
			memset(B->sc_e + (size_t)j * ql, 0, ql);
			memset(Y, 0xff, (size_t)2 * cl);
			if (!B->fs) {
				memset(B->rx + (size_t)j * cl, 0xff, cl);
			}
			continue;
		}
		memcpy(B->sc_s + (size_t)j * ql, sig + rl, ql);
		if (!B->fs) {
			memcpy(B->rx + (size_t)j * cl, sig, cl);
		}
		if (B->fs && B->kinf[j]) {
			memcpy(Y, B->wpt + (size_t)j * 2 * cl, (size_t)2 * cl);   /* any valid point: the product is not used */
		}
		/* BIP0340 lift_x: the representative with an even y (:532-535): y <- p - y when y is odd */
		if (!B->fs && (Y[2 * cl - 1] & 1)) {
			int borrow = 0;
			for (k = cl; k-- > 0;) {
				const int d = (int)B->p_be[k] - (int)Y[cl + k] - borrow;
				Y[cl + k] = (u8)(d & 0xff);
				borrow = d < 0;
			}
		}
	}
}

static void bip_final(u32 lo, u32 hi, void *arg)
{
	bip_job *B = (bip_job *)arg;
	ver_job *J = &B->v;
	const u32 cl = J->clen;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u8 *W = NULL;
		int ok = 0;
		if (!J->pre[j] && B->stA[j] != ECAMD_ERR && B->stB[j] != ECAMD_ERR) {
			const u8 b_inf = (B->stB[j] == ECAMD_INF) || (B->fs && B->kinf[j]);
			/* prj_pt_add of the two products, then prj_pt_unique / prj_pt_iszero (bip0340.c:536-541, ecfsdsa.c:563-567): an operand
			 * at infinity leaves the other one; the sum itself at infinity is rejected */
			if (B->stA[j] == ECAMD_INF && b_inf) {
				W = NULL;
			} else if (B->stA[j] == ECAMD_INF) {
				W = B->pB + (size_t)j * 2 * cl;
			} else if (b_inf) {
				W = B->pA + (size_t)j * 2 * cl;
			} else if (B->stS[j] == ECAMD_OK) {
				W = B->sum + (size_t)j * 2 * cl;
			}
			if (B->fs) {
				ok = W && !memcmp(W, J->s[J->idx[j]], (size_t)2 * cl);                    /* W' = W, both coordinates (ecfsdsa.c:569-576) */
			} else {
				ok = W && !(W[2 * cl - 1] & 1) && !memcmp(W, J->s[J->idx[j]], cl);        /* y even and x = r (bip0340.c:542-547) */
			}
		}
		J->res[j] = ok ? 0 : 1;
	}
}

static void fs_export_w(u32 lo, u32 hi, void *arg)
{
	bip_job *B = (bip_job *)arg;
	ver_job *J = &B->v;
	u32 j;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		u8 *dst = B->wpt + (size_t)j * 2 * J->clen;
		if (J->s[i] && J->s_len[i] == J->siglen) {
			memcpy(dst, J->s[i], (size_t)2 * J->clen);
		} else {
			memset(dst, 0xff, (size_t)2 * J->clen);
		}
	}
}

/* Round 6: the whole Schnorr-type batch as ONE streamed device call (include/libecc_amd.h: ec_schnorr_verify_msg_all_batch).  The pool packs
 * what ec_verify_init would look at -- the key's live limbs (X || Y || Z, or X || Y when every key has Z = 1), the signature, and the
 * scheme's hash input with a blank where the key's x goes (BIP0340: H(tag) || H(tag) || r || <blank> || m, sig/bip0340.c:437-494; ECFSDSA:
 * W || m, sig/ecfsdsa.c:520-540) -- while the device imports, hashes, reduces and evaluates the batch equation: no libecc hash, no nn_mod, no
 * separate normalisation call on the way.  Any item that fails a host-side check (key type, lengths, r >= p, s = 0 or >= q) makes the
 * attempt void (`dirty`): the caller then takes the item-by-item path, as for a batch the device does not vouch for. */
typedef struct {
	ver_job *J;
	int fs;
	u8 *keys, *sigs, *slots;
	u32 kw, stride, xoff, pre_len;
	u8 pre[2 * MAX_DIGEST_SIZE];
	u8 p_be[80], q_be[80];
	u32 dirty;
	int all;
} schfast_job;
static void schfast_pack(u32 lo, u32 hi, void *arg)
{
	schfast_job *F = (schfast_job *)arg;
	ver_job *J = F->J;
	const u32 cl = J->clen, ql = J->qlen, rl = F->fs ? 2 * cl : cl;
	u32 j, k;
	for (j = lo; j < hi; j++) {
		const u32 i = J->idx[j];
		const ec_pub_key *pk = J->pub_keys[i];
		const u8 *sig = J->s[i];
		u8 *slot = F->slots + (size_t)j * F->stride, tmp[3 * 80];
		int bad = (key_prefetch(J, j, hi), key_unusable(J, pk)) || !sig || J->s_len[i] != J->siglen ||
			  (!J->m[i] && J->m_len[i]) || prj_to_be(tmp, cl, &pk->y, &(J->params->ec_curve));
		if (!bad) {
			int snz = 0;
			for (k = 0; k < ql; k++) {
				snz |= sig[rl + k];
			}
			bad = !be_lt(sig + rl, F->q_be, ql) || (F->fs ? !snz : !be_lt(sig, F->p_be, cl));
		}
		if (bad) {
			AT_STORE(&F->dirty, 1);
			memset(F->keys + (size_t)j * F->kw, 0xff, F->kw);
			memset(F->sigs + (size_t)j * J->siglen, 0, J->siglen);
			memset(slot, 0, F->stride);
			continue;
		}
		if (F->kw == 2 * cl) {
			affine_guess_check(J, tmp, cl);
		}
		memcpy(F->keys + (size_t)j * F->kw, tmp, F->kw);
		memcpy(F->sigs + (size_t)j * J->siglen, sig, J->siglen);
		if (F->fs) {
			slot_put(slot, F->stride, sig, 2 * cl, NULL, 0, J->m[i], J->m_len[i]);
		} else {
			/* H(tag) || H(tag) || r || blank || m: the blank (clen zero octets here) is filled with the key's x on the device */
			const u32 len = F->pre_len + 2 * cl + J->m_len[i];
			slot[0] = (u8)len; slot[1] = (u8)(len >> 8); slot[2] = (u8)(len >> 16); slot[3] = (u8)(len >> 24);
			memcpy(slot + 4, F->pre, F->pre_len);
			memcpy(slot + 4 + F->pre_len, sig, cl);
			memset(slot + 4 + F->pre_len + cl, 0, cl);
			if (J->m_len[i]) {
				memcpy(slot + 4 + F->pre_len + 2 * cl, J->m[i], J->m_len[i]);
			}
			memset(slot + 4 + len, 0, F->stride - 4 - len);
		}
	}
}
static int schfast_gpu(u32 lo, u32 hi, void *arg)
{
	schfast_job *F = (schfast_job *)arg;
	ver_job *J = F->J;
	int all = 0;
	if (ecamd_multi_schnorr_verify_msg_all_batch(g_multi, J->e->mc, hi - lo, F->keys + (size_t)lo * F->kw, F->kw == 2 * J->clen ? ECAMD_PT_AFFINE : ECAMD_PT_PROJECTIVE,
						      F->sigs + (size_t)lo * J->siglen, F->fs ? 0 : 1, J->dev_hash, F->slots + (size_t)lo * F->stride, F->stride,
						      F->xoff, &all)) {
		return -1;
	}
	F->all = F->all && all;
	return 0;
}

/* the multi-scalar form pays from about 2^17 items per device on (profiles/r5b_schnorr_msm.md); $ECAMD_COMPAT_SCHNORR_MSM_MIN moves the
 * threshold (items per device; 0 = never) */
static unsigned long g_schnorr_msm_calls;
unsigned long ecamd_compat_schnorr_msm_calls(void) { return AT_LOAD(&g_schnorr_msm_calls); }
static int schnorr_msm_wanted(u32 cnt)
{
	unsigned long min_items = 1ul << 17;
	const char *e = getenv("ECAMD_COMPAT_SCHNORR_MSM_MIN");
	const int ranks = ecamd_multi_size(g_multi);
	if (e) {
		min_items = strtoul(e, NULL, 10);
		if (min_items == 0) {
			return 0;
		}
	}
	return ranks > 0 && (unsigned long)cnt / (unsigned long)ranks >= min_items;
}

static int schnorr_group(ver_job *J0, u32 cnt, int *results, int fs)
{
	bip_job B;
	ver_job *J = &B.v;
	hash_context hc;
	u32 j;
	int ret = -1, was_secret = g_secret, msm_declined = 0;
	memset(&B, 0, sizeof(B));
	B.v = *J0;
	B.fs = fs;
	J->clen = J->e->clen;
	J->qlen = J->e->qlen;
	J->siglen = (fs ? 2 * J->clen : J->clen) + J->qlen;   /* ECFSDSA_SIGLEN / BIP0340_SIGLEN */
	if (J->clen > 80 || J->qlen > 80 || nn_to_be(B.p_be, J->clen, &(J->params->ec_fp.p)) || nn_to_be(B.q_be, J->qlen, &(J->params->ec_gen_order)) ||
	    J->hm->hfunc_init(&hc) || J->hm->hfunc_update(&hc, (const u8 *)"BIP0340/challenge", 17) || J->hm->hfunc_finalize(&hc, B.tagd)) {
		return -1;
	}
	/* round 6: one streamed device call from keys, signatures and hash inputs first (schfast_job above) */
	if (schnorr_msm_wanted(cnt) && ec_schnorr_verify_all_available(ecamd_multi_curve_handle(J->e->mc, 0), fs ? 0 : 1) && !getenv("ECAMD_COMPAT_SCHNORR_HOST_PATH")) {
		schfast_job F;
		memset(&F, 0, sizeof(F));
		F.J = J;
		F.fs = fs;
		F.all = 1;
		J->dev_hash = dev_hash_type(J->hm);
		F.pre_len = fs ? 0 : 2u * J->hm->digest_size;
		F.stride = J->dev_hash ? dev_hash_slot(J, cnt, fs ? 2 * J->clen : F.pre_len + 2 * J->clen) : 0;
		F.xoff = fs ? 0xffffffffu : F.pre_len + J->clen;
		F.kw = (J->pre_scanned && !J->pre_not_affine && !getenv("ECAMD_COMPAT_PRJ_KEYS")) ? 2 * J->clen : 3 * J->clen;
		memcpy(F.p_be, B.p_be, sizeof(F.p_be));
		memcpy(F.q_be, B.q_be, sizeof(F.q_be));
		if (!fs) {
			memcpy(F.pre, B.tagd, J->hm->digest_size);
			memcpy(F.pre + J->hm->digest_size, B.tagd, J->hm->digest_size);
		}
		if (F.stride) {
			int tried = 1;
			F.keys = buf_get(0, (size_t)cnt * F.kw);
			F.sigs = buf_get(1, (size_t)cnt * J->siglen);
			F.slots = buf_get(2, (size_t)cnt * F.stride);
			if (!F.keys || !F.sigs || !F.slots) {
				return -1;
			}
			if (fs) {
				/* (ECFSDSA keys the combination through the application's get_random, as below) */
				u8 seed[32];
				int r;
				if (AT_LOAD(&g_rand_concurrent)) {
					r = get_random(seed, sizeof(seed));
				} else {
					pthread_mutex_lock(&g_rand_mu);
					r = get_random(seed, sizeof(seed));
					pthread_mutex_unlock(&g_rand_mu);
				}
				r = r || ecamd_multi_set_msm_seed(g_multi, seed);
				wipe(seed, sizeof(seed));
				tried = !r;
			}
			if (tried) {
				if (verify_pipeline(cnt, schfast_pack, schfast_gpu, NULL, &F)) {
					fprintf(stderr, "libecc_amd compat: the one-call Schnorr form failed (%s); verifying step by step\n", ecamd_last_error());
				} else {
					AT_ADD(&g_schnorr_msm_calls, 1);
					if (F.all && !AT_LOAD(&F.dirty)) {
						note_items(cnt);
						for (j = 0; j < cnt; j++) {
							results[J->idx[j]] = 0;
						}
						return 0;
					}
					msm_declined = 1;   /* the combination was evaluated and does not vouch: straight to the item-by-item pass below */
				}
			}
		}
	}
	J->kprj = buf_get(0, (size_t)cnt * 3 * J->clen);
	B.kaff = buf_get(1, (size_t)cnt * 2 * J->clen);
	B.kst = buf_get(2, cnt);
	B.sc_s = buf_get(3, (size_t)cnt * J->qlen);
	B.sc_e = buf_get(4, (size_t)cnt * J->qlen);
	B.pA = buf_get(5, (size_t)cnt * 2 * J->clen);
	B.stA = buf_get(6, cnt);
	B.pB = buf_get(7, (size_t)cnt * 2 * J->clen);
	B.stB = buf_get(8, cnt);
	B.sum = buf_get(9, (size_t)cnt * 2 * J->clen);
	B.stS = buf_get(10, cnt);
	J->pre = buf_get(11, cnt);
	B.wpt = buf_get(12, fs ? (size_t)cnt * 2 * J->clen : 1);
	B.wst = buf_get(13, cnt);
	B.kinf = buf_get(14, cnt);
	B.rx = buf_get(15, fs ? 1 : (size_t)cnt * J->clen);
	J->res = B.stS;   /* the verdicts overwrite the sum's status, read just before */
	if (!J->kprj || !B.kaff || !B.kst || !B.sc_s || !B.sc_e || !B.pA || !B.stA || !B.pB || !B.stB || !B.sum || !B.stS || !J->pre || !B.wpt ||
	    !B.wst || !B.kinf || !B.rx) {
		return -1;
	}
	/* everything a verification multiplies by is public: digit-indexed look-ups and the generator's comb table */
	if (was_secret && ecamd_multi_set_secret_scalars(g_multi, 0)) {
		return -1;
	}
	parallel_for(cnt, bip_export_keys, &B);
	if (ecamd_multi_prj_pt_unique_batch(g_multi, J->e->mc, cnt, J->kprj, ECAMD_PT_PROJECTIVE, B.kaff, ECAMD_PT_AFFINE, B.kst)) {
		goto gpu_err;
	}
	if (fs) {
		/* the points the signatures carry: coordinates < p and on the curve (is_on_shortw_curve, sig/ecfsdsa.c:452-455) */
		parallel_for(cnt, fs_export_w, &B);
		if (ecamd_multi_prj_pt_unique_batch(g_multi, J->e->mc, cnt, B.wpt, ECAMD_PT_AFFINE, B.sum, ECAMD_PT_AFFINE, B.wst)) {
			goto gpu_err;
		}
	}
	parallel_for(cnt, bip_pack, &B);
	/* The whole batch as ONE multi-scalar multiplication first (sig/bip0340.c:808-1025, sig/ecfsdsa.c:657-837: the reference's own batch
	 * equation; include/libecc_amd.h: ec_schnorr_verify_all_batch) when every item passed its pre-checks and the shards are large enough
	 * for it to pay (profiles/r5b_schnorr_msm.md): it vouches for a VALID batch; anything else -- a bad signature, an abscissa without
	 * a point, an exceptional addition -- comes back "not decided" and the item-by-item pass below gives the verdict, as the reference's
	 * own fall-back does.  ECFSDSA keys the combination through the application's get_random, the import the reference draws its a_i
	 * from (nn_get_random_mod, sig/ecfsdsa.c:745, :960); BIP0340's reference takes no randomness (a ChaCha20 stream keyed by a hash of the
	 * batch, sig/bip0340.c:758-860): there the engine keys its z_i with getrandom. */
	if (!msm_declined && schnorr_msm_wanted(cnt) && ec_schnorr_verify_all_available(ecamd_multi_curve_handle(J->e->mc, 0), fs ? 0 : 1)) {
		/* (the handle is asked first: nothing is drawn from the application's get_random for a curve the form does not serve --
		 * a curve with a cofactor, a field without a radix-2^29 unit) */
		int all = 0, clean = 1, tried = 1;
		for (j = 0; j < cnt && clean; j++) {
			clean = !J->pre[j] && !(fs && B.kinf[j]);
		}
		if (clean && fs) {
			u8 seed[32];
			int r;
			if (AT_LOAD(&g_rand_concurrent)) {
				r = get_random(seed, sizeof(seed));
			} else {
				pthread_mutex_lock(&g_rand_mu);
				r = get_random(seed, sizeof(seed));
				pthread_mutex_unlock(&g_rand_mu);
			}
			r = r || ecamd_multi_set_msm_seed(g_multi, seed);
			wipe(seed, sizeof(seed));
			if (r) {
				tried = 0;   /* no seed: the item-by-item pass below needs none */
			}
		}
		if (clean && tried) {
			/* an accelerator in front of the item-by-item pass, never the arbiter: a failure here (its multi-gigabyte tables at 2^20
			 * items, say) is reported and the batch is verified item by item */
			if (ecamd_multi_schnorr_verify_all_batch(g_multi, J->e->mc, cnt, B.sc_s, B.sc_e, B.kaff, fs ? B.wpt : B.rx, fs ? 0 : 1, &all)) {
				fprintf(stderr, "libecc_amd compat: the multi-scalar form failed (%s); verifying item by item\n", ecamd_last_error());
				all = 0;
			} else {
				AT_ADD(&g_schnorr_msm_calls, 1);
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
	}
	if (ecamd_multi_prj_pt_mul_batch(g_multi, J->e->mc, cnt, B.sc_s, J->qlen, NULL, B.pA, B.stA) ||
	    ecamd_multi_prj_pt_mul_batch(g_multi, J->e->mc, cnt, B.sc_e, J->qlen, B.kaff, B.pB, B.stB) ||
	    ecamd_multi_prj_pt_add_batch(g_multi, J->e->mc, cnt, B.pA, B.pB, B.sum, B.stS)) {
		goto gpu_err;
	}
	parallel_for(cnt, bip_final, &B);
	note_items(cnt);
	for (j = 0; j < cnt; j++) {
		results[J->idx[j]] = J->res[j] ? -1 : 0;
	}
	ret = 0;
	goto done;
gpu_err:
	fprintf(stderr, "libecc_amd compat: %s\n", ecamd_last_error());
done:
	if (was_secret && ecamd_multi_set_secret_scalars(g_multi, 1)) {
		ret = -1;
	}
	return ret;
}

static int is_bip0340(ec_alg_type t)
{
#if defined(WITH_SIG_BIP0340)
	return t == BIP0340;
#else
	(void)t;
	return 0;
#endif
}

static int is_ecfsdsa(ec_alg_type t)
{
#if defined(WITH_SIG_ECFSDSA)
	return t == ECFSDSA;
#else
	(void)t;
	return 0;
#endif
}

/* Index / result arrays of a verification call (4 bytes per item each): kept across calls.  A fresh 4 MB malloc is an mmap whose
 * pages fault in one by one under the first loop that touches them -- half a millisecond per array and call on the GPU host, the
 * price of a whole 2^20-item pack under a sandboxed kernel.  Up to four blocks rest here between calls; concurrent callers that find
 * none simply allocate. */
static pthread_mutex_t g_words_mu = PTHREAD_MUTEX_INITIALIZER;
static void *g_words[4];
static size_t g_words_cap[4];
static void *words_take(size_t bytes)
{
	void *p = NULL;
	int k;
	pthread_mutex_lock(&g_words_mu);
	for (k = 0; k < 4 && !p; k++) {
		if (g_words[k] && g_words_cap[k] >= bytes) {
			p = g_words[k];
			g_words[k] = NULL;
		}
	}
	pthread_mutex_unlock(&g_words_mu);
	if (!p) {
		size_t cap = 1u << 16;
		while (cap < bytes) {
			cap <<= 1;
		}
		p = malloc(cap + sizeof(size_t) * 2);
		if (!p) {
			return NULL;
		}
		*(size_t *)p = cap;
		return (u8 *)p + sizeof(size_t) * 2;
	}
	return p;
}
static void words_give(void *q)
{
	int k;
	if (!q) {
		return;
	}
	pthread_mutex_lock(&g_words_mu);
	for (k = 0; k < 4; k++) {
		if (!g_words[k]) {
			g_words[k] = q;
			g_words_cap[k] = *(size_t *)((u8 *)q - sizeof(size_t) * 2);
			q = NULL;
			break;
		}
	}
	pthread_mutex_unlock(&g_words_mu);
	if (q) {
		free((u8 *)q - sizeof(size_t) * 2);
	}
}
static void words_free_all(void)
{
	int k;
	pthread_mutex_lock(&g_words_mu);
	for (k = 0; k < 4; k++) {
		if (g_words[k]) {
			free((u8 *)g_words[k] - sizeof(size_t) * 2);
			g_words[k] = NULL;
		}
	}
	pthread_mutex_unlock(&g_words_mu);
}

/* the common shape of a batch -- every key present, initialised and of ONE set of parameters -- established on the pool, together
 * with the initial -1 of every result and the identity index map (a serial pass over 2^20 key pointers costs the caller more
 * than a millisecond); any other batch goes through the serial grouping below */
typedef struct {
	const ec_pub_key **pub_keys;
	const ec_params *params;
	const u32 *m_len;
	ec_alg_type sig_type;
	u32 *idx;
	int *results;
	u32 mixed, max_mlen, not_affine;   /* (not_affine: ecdsa_keys_affine's answer, from the same pass) */
} scan_job;
static void scan_keys(u32 lo, u32 hi, void *arg)
{
	scan_job *S = (scan_job *)arg;
	u32 i, mixed = 0, mx = 0, naff = 0, cur;
	for (i = lo; i < hi; i++) {
		const ec_pub_key *pk = S->pub_keys[i];
		int one = 0;
		S->results[i] = -1;
		S->idx[i] = i;
		mx = S->m_len[i] > mx ? S->m_len[i] : mx;
		if (!pk || pk->magic != PUB_KEY_MAGIC || pk->params != S->params) {
			mixed = 1;
			continue;
		}
		if (naff || pub_key_check_initialized_and_type(pk, S->sig_type) || prj_pt_check_initialized(&pk->y) || fp_check_initialized(&pk->y.Z)) {
			continue;
		}
		if (nn_isone(&pk->y.Z.fp_val, &one) || !one) {
			naff = 1;
		}
	}
	if (mixed) {
		AT_STORE(&S->mixed, 1);
	}
	if (naff) {
		AT_STORE(&S->not_affine, 1);
	}
	cur = AT_LOAD(&S->max_mlen);
	while (mx > cur && !__atomic_compare_exchange_n(&S->max_mlen, &cur, mx, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
	}
}

/* The pass WITHOUT the keys (round 6): a batch's keys are 0.7 KB structures each -- reading a million of them for their parameters pointer and
 * their Z is 3 ms of DRAM traffic that the packing steps repeat anyway.  So: results and indices initialised, the longest message found, one
 * set of parameters taken for granted and "Z = 1 everywhere" guessed from 64 keys; a packing step that meets a key under other parameters, no
 * key at all, or a Z != 1 under the affine guess sets the job's `broken` word and verify_results starts over with scan_keys. */
static unsigned long g_verify_restarts;
unsigned long ecamd_compat_verify_restarts(void) { return AT_LOAD(&g_verify_restarts); }
static void scan_light(u32 lo, u32 hi, void *arg)
{
	scan_job *S = (scan_job *)arg;
	u32 i, mx = 0, cur;
	for (i = lo; i < hi; i++) {
		S->results[i] = -1;
		S->idx[i] = i;
		mx = S->m_len[i] > mx ? S->m_len[i] : mx;
	}
	cur = AT_LOAD(&S->max_mlen);
	while (mx > cur && !__atomic_compare_exchange_n(&S->max_mlen, &cur, mx, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
	}
}
static u32 sample_not_affine(const scan_job *S, u32 num)
{
	const u32 step = num > 64 ? num / 64 : 1;
	u32 i;
	for (i = 0; i < num; i += step) {
		const ec_pub_key *pk = S->pub_keys[i];
		int one = 0;
		if (!pk || pk->magic != PUB_KEY_MAGIC || pk->params != S->params || pub_key_check_initialized_and_type(pk, S->sig_type) ||
		    prj_pt_check_initialized(&pk->y) || fp_check_initialized(&pk->y.Z)) {
			continue;
		}
		if (nn_isone(&pk->y.Z.fp_val, &one) || !one) {
			return 1;
		}
	}
	return 0;
}

static int verify_results(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			  ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int *results, int all_only,
			  int *fail_known, int *fail, int one_params)
{
	const hash_mapping *hm;
	hash_alg_type eh = UNKNOWN_HASH_ALG;
	ec_curve_type ec = UNKNOWN_CURVE;
	int ph = 0, dom = 0, is448 = 0, ed, ret = -1, one_group = 0;
	scan_job S;
	u32 *idx = NULL, i, done = 0, broken = 0;
	u8 *seen = NULL;
	int optimistic;
	if (!s || !s_len || !pub_keys || !m || !m_len || !results) {
		return -1;
	}
	ed = !eddsa_variant(sig_type, &eh, &ec, &ph, &dom, &is448);
	if (!ed && !is_ecdsa(sig_type) && !is_bip0340(sig_type) && !is_ecfsdsa(sig_type)) {
		return -1;
	}
	if (num == 0) {
		return 0;
	}
	hm = find_hash(hash_type);
	if (!hm || (ed && hash_type != eh) || ecamd_compat_init(NULL, 0, 0)) {
		for (i = 0; i < num; i++) {
			results[i] = -1;
		}
		return (!hm || (ed && hash_type != eh)) ? 0 : -1;   /* (every ec_verify fails in ec_verify_init / _eddsa_verify_init) */
	}
	idx = (u32 *)words_take((size_t)num * sizeof(u32));
	if (!idx) {
		goto out;
	}
	optimistic = getenv("ECAMD_COMPAT_FULL_SCAN") == NULL;
again:
	{
		S.m_len = m_len;
		S.sig_type = sig_type;
		S.max_mlen = S.not_affine = 0;
		S.pub_keys = pub_keys;
		S.params = (pub_keys[0] && pub_keys[0]->magic == PUB_KEY_MAGIC) ? pub_keys[0]->params : NULL;
		S.idx = idx;
		S.results = results;
		S.mixed = S.params ? 0 : 1;
		optimistic = optimistic && S.params != NULL;
		done = 0;
		AT_STORE(&broken, 0);
		call_enter(1);
		if (optimistic) {
			parallel_for(num, scan_light, &S);
			S.not_affine = sample_not_affine(&S, num);
		} else {
			parallel_for(num, scan_keys, &S);
		}
		call_leave();
		one_group = !AT_LOAD(&S.mixed);
	}
	if (one_params && !one_group) {
		/* eddsa_verify_batch / bip0340_verify_batch: "all our public keys have the same parameters" (sig/eddsa.c:2358, sig/bip0340.c:843-845) --
		 * a missing key, another set of parameters: -1 for the batch; the scan above has just looked at every key on the pool */
		goto out;
	}
	if (!one_group) {
		seen = (u8 *)calloc(num, 1);
		if (!seen) {
			goto out;
		}
	}
	/* groups of items that share their ec_params (one GPU batch each; normally there is one group) */
	while (done < num) {
		const ec_params *params = NULL;
		ver_job J;
		u32 cnt = 0;
		int r;
		if (one_group) {
			params = pub_keys[0]->params;
			cnt = done = num;
		}
		for (i = 0; i < num && !one_group; i++) {
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
		if (ed && params->curve_type != ec) {
			continue;   /* eddsa_key_type_check_curve fails: -1 for the group */
		}
		memset(&J, 0, sizeof(J));
		J.s = s; J.s_len = s_len; J.m = m; J.m_len = m_len; J.adata = adata; J.adata_len = adata_len;
		J.pub_keys = pub_keys; J.sig_type = sig_type; J.hm = hm; J.idx = idx; J.params = params;
		J.ph = ph; J.dom = dom; J.is448 = is448;
		J.ph_len = is448 ? 64 : hm->digest_size;   /* EDDSA448PH: SHAKE256 with 64 bytes (sig/eddsa.c:2343-2346) */
		J.all_only = all_only && ed;
		J.e = curve_from_params(params);
		if (!J.e) {
			goto out;
		}
		if (one_group) {
			J.pre_scanned = 1;
			J.pre_max_mlen = S.max_mlen;
			J.pre_not_affine = S.not_affine;
		}
		if (optimistic) {
			J.broken = &broken;
			J.assumed_affine = !S.not_affine;
		}
		/* ECDSA and EdDSA verification: a pipeline around ONE kind of C-ABI call on public data -- up to NARENA application threads at a
		 * time, each in its own staging, the GPU calls one after the other (g_gpu_mu); the Schnorr-type path makes dependent calls and
		 * switches the devices' scalar mode: exclusive */
		call_enter((is_bip0340(sig_type) || is_ecfsdsa(sig_type)) ? 0 : 1);
		r = ed ? eddsa_group(&J, cnt, results) : ((is_bip0340(sig_type) || is_ecfsdsa(sig_type)) ? schnorr_group(&J, cnt, results, is_ecfsdsa(sig_type)) : ecdsa_group(&J, cnt, results));
		call_leave();
		if (optimistic && AT_LOAD(&broken)) {
			AT_ADD(&g_verify_restarts, 1);
			/* a key under other parameters, a missing key, a Z != 1 under the affine guess: the whole call again, behind the pass over the keys
			 * (whatever the attempt wrote into `results` is initialised again there) */
			optimistic = 0;
			goto again;
		}
		if (r) {
			goto out;
		}
		if (one_group && J.fail_tracked && fail_known) {
			*fail_known = 1;
			*fail = AT_LOAD(&J.any_fail) ? 1 : 0;
		}
	}
	ret = 0;
out:
	words_give(idx);
	free(seen);
	return ret;
}

int ec_verify_batch_results(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			    ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int *results)
{
	return verify_results(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, results, 0, NULL, NULL, 0);
}

typedef struct {
	const int *res;
	u32 fail;
} anyfail_job;
static void any_failure(u32 lo, u32 hi, void *arg)
{
	anyfail_job *A = (anyfail_job *)arg;
	u32 i, f = 0;
	for (i = lo; i < hi; i++) {
		f |= (u32)(A->res[i] != 0);
	}
	if (f) {
		AT_STORE(&A->fail, 1);
	}
}

static int all_accepted(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len, int all_only, int one_params)
{
	int *res, ret = -1, known = 0, fail = 0;
	double t0 = 0;
	if (num == 0) {
		return -1;   /* "We need at least one element in our batch data bags" (sig/eddsa.c:2312) */
	}
	res = (int *)words_take((size_t)num * sizeof(int));
	if (!res) {
		return -1;
	}
	if (getenv("ECAMD_COMPAT_TIMING")) {
		t0 = now_ms();
	}
	if (!verify_results(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, res, all_only, &known, &fail, one_params)) {
		if (known) {
			ret = fail ? -1 : 0;
		} else {
			anyfail_job A;
			A.res = res;
			A.fail = 0;
			call_enter(1);
			parallel_for(num, any_failure, &A);
			call_leave();
			ret = AT_LOAD(&A.fail) ? -1 : 0;
		}
	}
	words_give(res);
	if (t0 != 0) {
		fprintf(stderr, "libecc_amd compat timing: verify call of %u items: %.2f ms in all\n", num, now_ms() - t0);
	}
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
	return all_accepted(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, 0, 0);
}

int eddsa_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			   ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			   verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len)
{
	hash_alg_type eh = UNKNOWN_HASH_ALG;
	ec_curve_type ec = UNKNOWN_CURVE;
	int ph, dom, is448;
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
	/* "all our public keys have the same parameters": verify_results' scan of the keys (one_params) */
	return all_accepted(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, 1, 1);
}

int bip0340_verify_batch_gpu(const u8 **s, const u8 *s_len, const ec_pub_key **pub_keys, const u8 **m, const u32 *m_len, u32 num,
			     ec_alg_type sig_type, hash_alg_type hash_type, const u8 **adata, const u16 *adata_len,
			     verify_batch_scratch_pad *scratch_pad_area, u32 *scratch_pad_area_len)
{
	if (!is_bip0340(sig_type) && !is_ecfsdsa(sig_type)) {
		return -1;
	}
	/* argument checks of bip0340_verify_batch / _bip0340_verify_batch[_no_memory] (sig/bip0340.c:1296-1318, :843-845, :1065-1107):
	 * arrays present, at least one item, one set of parameters, the scratch pad long enough when one is given */
	if (!s || !pub_keys || !m) {
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
	return all_accepted(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, 0, 1);
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
	if (is_bip0340(sig_type) || is_ecfsdsa(sig_type)) {
		return bip0340_verify_batch_gpu(s, s_len, pub_keys, m, m_len, num, sig_type, hash_type, adata, adata_len, scratch_pad_area,
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
