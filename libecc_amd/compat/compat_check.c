/*
 * libecc_amd/compat/compat_check.c -- check program of the libecc-typed boundary (include/libecc_amd_compat.h).
 *
 * Written as a libecc application: it holds ec_params, ec_key_pair, prj_pt, nn and calls the batch entry points of
 * libsign_amd.so with the same array shapes tests/ec_self_tests_core.c uses for ec_verify_batch (:373-383, :556-616);
 * every batch result is compared with libecc's own scalar function (prj_pt_mul, ecccdh_derive_secret, ec_verify -- the
 * CPU code of the very libecc the library was linked from) on the same inputs.  Needs an MI355X.
 *   usage: compat_check [items per case, default 256] | compat_check quick <items> | compat_check bench <log2 items> | compat_check bench_schnorr <log2 items>
 * Exit status 0 iff everything matched; prints one line per case.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/random.h>
#include "libecc_amd_compat.h"
#include <pthread.h>
#include <unistd.h>
#include "external_deps/rand.h"   /* get_random: supplied by the application, as for libecc's own libsign */

/* The application's randomness source (libsign leaves get_random undefined: external_deps/rand.c is libecc's example of
 * it and opens /dev/urandom on every call -- two opens per nonce, which caps a 16-thread host at about 1 M nonces/s).
 * This application reads the kernel's generator through getrandom(2) in blocks of 4 KiB per thread. */
static volatile int g_rand_inside;       /* a caller is inside get_random */
static volatile int g_rand_overlaps;     /* entries that found another caller inside */
static int g_rand_expect_serial = 1;     /* the default contract: the library calls get_random from one thread at a time */
int get_random(unsigned char *buf, u16 len)
{
	static __thread unsigned char pool[4096];
	static __thread unsigned int pos = sizeof(pool);
	u16 done = 0;
	/* libecc_amd_compat.h: unless the application lifts it (ecamd_compat_set_concurrent_random), the batch forms serialise
	 * their calls into this function; count the entries that overlap another one (checked at the end of the run) */
	if (g_rand_expect_serial && __atomic_exchange_n(&g_rand_inside, 1, __ATOMIC_ACQ_REL)) {
		__atomic_add_fetch(&g_rand_overlaps, 1, __ATOMIC_RELAXED);   /* (not in bench mode: the shared flag would be the bottleneck there) */
	}
#define RAND_LEAVE() do { if (g_rand_expect_serial) __atomic_store_n(&g_rand_inside, 0, __ATOMIC_RELEASE); } while (0)
	while (done < len) {
		unsigned int take;
		if (pos == sizeof(pool)) {
			size_t got = 0;
			while (got < sizeof(pool)) {
				const ssize_t r = getrandom(pool + got, sizeof(pool) - got, 0);
				if (r <= 0) {
					RAND_LEAVE();
					return -1;
				}
				got += (size_t)r;
			}
			pos = 0;
		}
		take = (unsigned int)(len - done);
		if (take > sizeof(pool) - pos) {
			take = (unsigned int)(sizeof(pool) - pos);
		}
		memcpy(buf + done, pool + pos, take);
		memset(pool + pos, 0, take);
		pos += take;
		done = (u16)(done + take);
	}
	RAND_LEAVE();
	return 0;
}

/* the other symbol libsign imports: the non-cryptographic generator of libecc's self tests (here simply the same source) */
int get_unsafe_random(unsigned char *buf, u16 len)
{
	return get_random(buf, len);
}

static int failures;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("  MISMATCH " __VA_ARGS__); printf("\n"); } } while (0)

static int load_params(const char *name, ec_params *params)
{
	const ec_str_params *sp = NULL;
	if (ec_get_curve_params_by_name((const u8 *)name, (u8)(strlen(name) + 1), &sp) || !sp) {
		return -1;
	}
	return import_params(params, sp);
}

/* ---- prj_pt_mul_batch against prj_pt_mul ---- */
static void check_mul(const char *curve, u32 n, int blind)
{
	ec_params params;
	prj_pt *in = calloc(n, sizeof(prj_pt)), *out = calloc(n, sizeof(prj_pt));
	nn *m = calloc(n, sizeof(nn));
	int *rets = calloc(n, sizeof(int));
	u32 i, bad = 0, ninf = 0, nerr = 0;
	const u32 before = failures;
	if (load_params(curve, &params) || !in || !out || !m || !rets) {
		CHECK(0, "%s: setup", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		nn t;
		int r = nn_get_random_mod(&m[i], &params.ec_gen_order) || nn_get_random_mod(&t, &params.ec_gen_order) ||
			prj_pt_mul(&in[i], &t, &params.ec_gen);   /* whatever projective representative prj_pt_mul leaves */
		if (r) {
			CHECK(0, "%s: input generation", curve);
			return;
		}
	}
	/* edge items: scalars 0, 1, q - 1, q, q + 1, a scalar of twice the order's length; the generator; the point at infinity */
	if (n >= 16) {
		nn one;
		nn_init(&one, 0); nn_one(&one);
		nn_zero(&m[0]);
		nn_one(&m[1]);
		nn_sub(&m[2], &params.ec_gen_order, &one);
		nn_copy(&m[3], &params.ec_gen_order);
		nn_add(&m[4], &params.ec_gen_order, &one);
		nn_mul(&m[5], &m[5], &m[6]);                       /* ~2 |q| bits */
		prj_pt_copy(&in[7], &params.ec_gen);
		nn_copy(&m[8], &params.ec_gen_order);
		prj_pt_copy(&in[8], &params.ec_gen);               /* [q]G = infinity */
		prj_pt_zero(&in[9]);                               /* infinity in */
		prj_pt_copy(&in[10], &params.ec_gen);
		nn_sub(&m[10], &params.ec_gen_order, &one);        /* [q - 1]G = -G */
		/* a point that is not on the curve: Y + 1 */
		fp_inc(&in[11].Y, &in[11].Y);
	}
	if (blind ? prj_pt_mul_blind_batch(out, m, in, n, rets) : prj_pt_mul_batch(out, m, in, n, rets)) {
		CHECK(0, "%s: prj_pt_mul%s_batch failed", curve, blind ? "_blind" : "");
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int r = prj_pt_mul(&ref, &m[i], &in[i]), cmp = 1, z1 = 0, z2 = 0;
		if (r != rets[i]) {
			bad++;
			CHECK(0, "%s: item %u returns %d, prj_pt_mul %d", curve, i, rets[i], r);
			continue;
		}
		if (r) {
			nerr++;
			continue;
		}
		if (prj_pt_iszero(&ref, &z1) || prj_pt_iszero(&out[i], &z2) || z1 != z2) {
			bad++;
			CHECK(0, "%s: item %u infinity flag %d vs %d", curve, i, z2, z1);
			continue;
		}
		if (z1) {
			ninf++;
			continue;
		}
		if (prj_pt_cmp(&ref, &out[i], &cmp) || cmp) {
			bad++;
			CHECK(0, "%s: item %u differs from prj_pt_mul", curve, i);
		}
	}
	printf("prj_pt_mul%s_batch %-16s %u items, %u errors, %u at infinity: %s\n", blind ? "_blind" : "", curve, n, nerr, ninf, failures == before ? "ok" : "FAILED");
	(void)bad;
	free(in); free(out); free(m); free(rets);
}

/* ---- round 4: prj_pt_add / dbl / unique / is_on_curve, _prj_pt_unprotected_mult, check_prj_pt_order and
 *      ec_pub_key_import_from_aff_buf in batch form against the scalar functions ---- */
static int same_point(prj_pt_src_t a, prj_pt_src_t b)
{
	int z1 = 0, z2 = 0, cmp = 1;
	if (prj_pt_iszero(a, &z1) || prj_pt_iszero(b, &z2) || z1 != z2) {
		return 0;
	}
	return z1 ? 1 : (!prj_pt_cmp(a, b, &cmp) && !cmp);
}

static void check_group_law(const char *curve, u32 n)
{
	ec_params params;
	prj_pt *a = calloc(n, sizeof(prj_pt)), *b = calloc(n, sizeof(prj_pt)), *out = calloc(n, sizeof(prj_pt));
	nn *m = calloc(n, sizeof(nn));
	int *rets = calloc(n, sizeof(int)), *flags = calloc(n, sizeof(int));
	ec_pub_key *pubs = calloc(n, sizeof(ec_pub_key));
	u8 *bufs = calloc(n, 2 * 72);
	const u8 **bp = calloc(n, sizeof(u8 *));
	u32 i, clen, nerr = 0, ninf = 0;
	const u32 before = failures;
	int isone = 0;
	if (load_params(curve, &params) || !a || !b || !out || !m || !rets || !flags || !pubs || !bufs || !bp) {
		CHECK(0, "%s: setup", curve);
		return;
	}
	clen = (u32)BYTECEIL(params.ec_fp.p_bitlen);
	nn_isone(&params.ec_gen_cofactor, &isone);
	for (i = 0; i < n; i++) {
		nn t;
		if (nn_get_random_mod(&t, &params.ec_gen_order) || prj_pt_mul(&a[i], &t, &params.ec_gen) ||
		    nn_get_random_mod(&t, &params.ec_gen_order) || prj_pt_mul(&b[i], &t, &params.ec_gen) ||
		    nn_get_random_mod(&m[i], &params.ec_gen_order)) {
			CHECK(0, "%s: input generation", curve);
			return;
		}
		if (i % 5 == 4) {
			u8 w[2];
			get_random(w, 2);
			nn_init(&m[i], 0);
			m[i].val[0] = (word_t)(w[0] | (w[1] << 8));   /* short public scalars (cofactors and the like) */
			m[i].wlen = 1;
		}
	}
	if (n >= 16) {
		prj_pt_copy(&b[0], &a[0]);                              /* P + P through the addition */
		prj_pt_neg(&b[1], &a[1]);                               /* P + (-P) = infinity */
		prj_pt_zero(&b[2]);                                     /* P + infinity */
		prj_pt_zero(&a[3]);                                     /* infinity + Q */
		prj_pt_zero(&a[4]); prj_pt_zero(&b[4]);                 /* infinity + infinity */
		fp_inc(&a[5].Y, &a[5].Y);                               /* off the curve: an error in the batch forms */
		nn_zero(&m[6]);                                         /* [0]P */
		nn_copy(&m[7], &params.ec_gen_order);                   /* [q]P = infinity */
		nn_one(&m[8]);
		if (!isone) {
			/* a cofactor curve: the point of order two (x, 0) exists -- on libecc's 25519 / 448 models x = A/3 -- as does a
			 * point of order 2q; the addition's exceptional pair and the double-and-add's -1 live there */
			int found = 0, tries;
			for (tries = 0; tries < 200 && !found; tries++) {
				/* a curve point outside the subgroup: x = a small value, y from the curve equation; [q] of it has an order
				 * that divides the cofactor, and doubling that until the next doubling is infinity leaves the point of order two */
				aff_pt ap;
				prj_pt T, Q2, D;
				fp xs, y1, yb;
				int z = 0, zz = 0, guard = 0;
				if (fp_init(&xs, &params.ec_fp) || fp_init(&y1, &params.ec_fp) || fp_init(&yb, &params.ec_fp) ||
				    fp_set_word_value(&xs, (word_t)(2 + tries))) {
					break;
				}
				if (aff_pt_y_from_x(&y1, &yb, &xs, &params.ec_curve) || aff_pt_init_from_coords(&ap, &params.ec_curve, &xs, &y1) ||
				    ec_shortw_aff_to_prj(&T, &ap) || prj_pt_mul(&Q2, &params.ec_gen_order, &T) || prj_pt_iszero(&Q2, &z) || z) {
					continue;
				}
				while (guard++ < 8 && !prj_pt_dbl(&D, &Q2) && !prj_pt_iszero(&D, &zz) && !zz) {
					prj_pt_copy(&Q2, &D);
				}
				if (zz) {
					/* Q2 has order two: a[9] + (a[9] + Q2) is an exceptional pair; a[10] = Q2 itself; a[11] has order 2q */
					if (!prj_pt_add(&b[9], &a[9], &Q2) && !prj_pt_copy(&a[10], &Q2) && !nn_copy(&m[10], &params.ec_gen_order) &&
					    !prj_pt_add(&a[11], &a[11], &Q2) && !nn_copy(&m[11], &params.ec_gen_order)) {
						found = 1;
					}
				}
			}
			CHECK(found, "%s: no point of order two found for the exceptional-pair items", curve);
		}
	}
	/* prj_pt_add_batch / prj_pt_dbl_batch / prj_pt_unique_batch / prj_pt_is_on_curve_batch */
	if (prj_pt_add_batch(out, a, b, n, rets)) {
		CHECK(0, "%s: prj_pt_add_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int on1 = 0, on2 = 0;
		int r = prj_pt_add(&ref, &a[i], &b[i]);
		prj_pt_is_on_curve(&a[i], &on1); prj_pt_is_on_curve(&b[i], &on2);
		if (!on1 || !on2) {
			CHECK(rets[i] == -1, "%s: add item %u off the curve must be an error", curve, i);
			nerr++;
			continue;
		}
		CHECK(r == rets[i], "%s: add item %u returns %d, prj_pt_add %d", curve, i, rets[i], r);
		if (r) { nerr++; continue; }
		CHECK(same_point(&ref, &out[i]), "%s: add item %u differs from prj_pt_add", curve, i);
	}
	if (prj_pt_dbl_batch(out, a, n, rets)) {
		CHECK(0, "%s: prj_pt_dbl_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int on1 = 0, z = 0;
		prj_pt_is_on_curve(&a[i], &on1);
		if (!on1) { CHECK(rets[i] == -1, "%s: dbl item %u off the curve must be an error", curve, i); continue; }
		CHECK(!prj_pt_dbl(&ref, &a[i]) && rets[i] == 0 && same_point(&ref, &out[i]), "%s: dbl item %u differs from prj_pt_dbl", curve, i);
		if (!prj_pt_iszero(&out[i], &z) && z) ninf++;
	}
	/* prj_pt_neg_batch / prj_pt_cmp_batch / prj_pt_eq_or_opp_batch (round 6): against the scalar functions on the same pairs -- equal,
	 * opposite, infinity on either side, different -- and on each point against itself, its opposite and its double */
	if (prj_pt_neg_batch(out, a, n, rets)) {
		CHECK(0, "%s: prj_pt_neg_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int on1 = 0;
		prj_pt_is_on_curve(&a[i], &on1);
		if (!on1) { CHECK(rets[i] == -1, "%s: neg item %u off the curve must be an error", curve, i); continue; }
		CHECK(!prj_pt_neg(&ref, &a[i]) && rets[i] == 0 && same_point(&ref, &out[i]), "%s: neg item %u differs from prj_pt_neg", curve, i);
	}
	{
		int pass, neq = 0, nopp = 0, ndiff = 0;
		for (pass = 0; pass < 3; pass++) {
			/* pass 0: the pairs of the addition test; pass 1: each point against its opposite as the device returned it (another Z);
			 * pass 2: each point against the next one */
			const prj_pt *second = pass == 0 ? b : (pass == 1 ? out : NULL);
			prj_pt *rot = NULL;
			if (pass == 2) {
				rot = (prj_pt *)malloc((size_t)n * sizeof(prj_pt));
				if (!rot) { CHECK(0, "%s: out of memory", curve); return; }
				for (i = 0; i < n; i++) rot[i] = a[(i + 1) % n];
				second = rot;
			}
			if (prj_pt_cmp_batch(a, second, n, flags, rets)) {
				CHECK(0, "%s: prj_pt_cmp_batch failed", curve);
				free(rot);
				return;
			}
			for (i = 0; i < n; i++) {
				int on1 = 0, on2 = 0, c = 7;
				prj_pt_is_on_curve(&a[i], &on1); prj_pt_is_on_curve(&second[i], &on2);
				if (!on1 || !on2) { CHECK(rets[i] == -1, "%s: cmp item %u off the curve must be an error", curve, i); continue; }
				CHECK(!prj_pt_cmp(&a[i], &second[i], &c) && rets[i] == 0 && (flags[i] != 0) == (c != 0), "%s: cmp pass %d item %u: %d, prj_pt_cmp %d", curve,
				      pass, i, flags[i], c);
				if (!c) neq++; else ndiff++;
			}
			if (prj_pt_eq_or_opp_batch(a, second, n, flags, rets)) {
				CHECK(0, "%s: prj_pt_eq_or_opp_batch failed", curve);
				free(rot);
				return;
			}
			for (i = 0; i < n; i++) {
				int on1 = 0, on2 = 0, c = 7;
				prj_pt_is_on_curve(&a[i], &on1); prj_pt_is_on_curve(&second[i], &on2);
				if (!on1 || !on2) { CHECK(rets[i] == -1, "%s: eq_or_opp item %u off the curve must be an error", curve, i); continue; }
				CHECK(!prj_pt_eq_or_opp(&a[i], &second[i], &c) && rets[i] == 0 && flags[i] == c, "%s: eq_or_opp pass %d item %u: %d, prj_pt_eq_or_opp %d",
				      curve, pass, i, flags[i], c);
				if (c) nopp++;
			}
			free(rot);
		}
		CHECK(n < 16 || (neq > 0 && ndiff > 0 && nopp > neq), "%s: the comparison items do not cover equal / opposite / different (%d %d %d)", curve, neq,
		      ndiff, nopp);
	}
	if (prj_pt_unique_batch(out, a, n, rets) || prj_pt_is_on_curve_batch(a, n, flags, NULL)) {
		CHECK(0, "%s: prj_pt_unique_batch / prj_pt_is_on_curve_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int on1 = 0, cmp = 1, isone_z = 0;
		fp one_fp;
		int r;
		prj_pt_is_on_curve(&a[i], &on1);
		CHECK(flags[i] == on1, "%s: on-curve flag of item %u is %d, prj_pt_is_on_curve %d", curve, i, flags[i], on1);
		if (!on1) { CHECK(rets[i] == -1, "%s: unique item %u off the curve must be an error", curve, i); continue; }
		r = prj_pt_unique(&ref, &a[i]);
		CHECK(r == rets[i], "%s: unique item %u returns %d, prj_pt_unique %d", curve, i, rets[i], r);
		if (r) continue;
		CHECK(!prj_pt_cmp(&ref, &out[i], &cmp) && !cmp && !fp_init(&one_fp, &params.ec_fp) && !fp_one(&one_fp) && !fp_cmp(&out[i].Z, &one_fp, &isone_z) && !isone_z, "%s: unique item %u differs", curve, i);
	}
	/* _prj_pt_unprotected_mult_batch and check_prj_pt_order_batch (PUBLIC_PT and sensitive) */
	if (_prj_pt_unprotected_mult_batch(out, m, a, n, rets)) {
		CHECK(0, "%s: _prj_pt_unprotected_mult_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		const int r = _prj_pt_unprotected_mult(&ref, &m[i], &a[i]);
		CHECK(r == rets[i], "%s: unprotected mult item %u returns %d, the scalar function %d", curve, i, rets[i], r);
		if (r) continue;
		CHECK(same_point(&ref, &out[i]), "%s: unprotected mult item %u differs", curve, i);
	}
	{
		int s;
		for (s = 0; s < 2; s++) {
			const prj_pt_sensitivity sens = s ? PRIVATE_PT : PUBLIC_PT;
			const u32 cnt = s ? (n < 64 ? n : 64) : n;
			if (check_prj_pt_order_batch(a, &params.ec_gen_order, sens, cnt, flags, rets)) {
				CHECK(0, "%s: check_prj_pt_order_batch failed", curve);
				return;
			}
			for (i = 0; i < cnt; i++) {
				int chk = 0;
				const int r = check_prj_pt_order(&a[i], &params.ec_gen_order, sens, &chk);
				CHECK(r == rets[i] && (r || chk == flags[i]), "%s: order check (%s) item %u: %d/%d, the scalar function %d/%d", curve,
				      s ? "sensitive" : "public", i, rets[i], flags[i], r, chk);
			}
		}
	}
	/* ec_pub_key_import_from_aff_buf_batch: the affine octets of the points above (and what does not decode) */
	for (i = 0; i < n; i++) {
		int z = 0, on = 0;
		bp[i] = bufs + (size_t)i * 2 * 72;
		memset(bufs + (size_t)i * 2 * 72, 0xff, 2 * clen);
		prj_pt_is_on_curve(&a[i], &on);
		if (on && !prj_pt_iszero(&a[i], &z) && !z) {
			prj_pt u;
			if (!prj_pt_unique(&u, &a[i])) {
				prj_pt_export_to_aff_buf(&u, bufs + (size_t)i * 2 * 72, 2 * clen);
			}
		}
	}
	if (ec_pub_key_import_from_aff_buf_batch(pubs, &params, bp, (u8)(2 * clen), ECDSA, n, rets)) {
		CHECK(0, "%s: ec_pub_key_import_from_aff_buf_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		ec_pub_key ref;
		const int r = ec_pub_key_import_from_aff_buf(&ref, &params, bp[i], (u8)(2 * clen), ECDSA);
		CHECK(r == rets[i], "%s: key import item %u returns %d, the scalar function %d", curve, i, rets[i], r);
		if (r) continue;
		CHECK(same_point(&ref.y, &pubs[i].y) && pubs[i].key_type == ECDSA && pubs[i].params == &params && pubs[i].magic == ref.magic,
		      "%s: key import item %u differs", curve, i);
	}
	printf("group law / public-scalar batch forms %-16s %u items, %u addition errors, %u doublings at infinity: %s\n", curve, n, nerr, ninf,
	       failures == before ? "ok" : "FAILED");
	free(a); free(b); free(out); free(m); free(rets); free(flags); free(pubs); free(bufs); free((void *)bp);
}

/* ---- ecccdh_derive_secret_batch against ecccdh_derive_secret ---- */
static void check_cdh(const char *curve, u32 n)
{
	ec_params params;
	ec_key_pair *ours = calloc(n, sizeof(ec_key_pair));
	const ec_priv_key **privs = calloc(n, sizeof(*privs));
	u8 *peerbuf, *secbuf, plen = 0, slen = 0;
	const u8 **peers = calloc(n, sizeof(*peers));
	u8 **secs = calloc(n, sizeof(*secs));
	int *rets = calloc(n, sizeof(int));
	u32 i, nerr = 0;
	const u32 before = failures;
	if (load_params(curve, &params) || ecccdh_serialized_pub_key_size(&params, &plen) || ecccdh_shared_secret_size(&params, &slen)) {
		CHECK(0, "%s: setup", curve);
		return;
	}
	peerbuf = calloc(n, plen);
	secbuf = calloc(n, slen);
	for (i = 0; i < n; i++) {
		ec_key_pair peer;
		if (ecccdh_gen_key_pair(&ours[i], &params) || ecccdh_gen_key_pair(&peer, &params) ||
		    ecccdh_serialize_pub_key(&peer.pub_key, peerbuf + (size_t)i * plen, plen)) {
			CHECK(0, "%s: key generation", curve);
			return;
		}
		privs[i] = &ours[i].priv_key;
		peers[i] = peerbuf + (size_t)i * plen;
		secs[i] = secbuf + (size_t)i * slen;
	}
	if (n >= 8) {
		peerbuf[(size_t)3 * plen + plen - 1] ^= 1;        /* off the curve */
		memset(peerbuf + (size_t)4 * plen, 0xff, plen);   /* coordinates >= p */
		memset(peerbuf + (size_t)5 * plen, 0, plen);      /* (0, 0) */
	}
	if (ecccdh_derive_secret_batch(privs, peers, plen, secs, slen, n, rets)) {
		CHECK(0, "%s: ecccdh_derive_secret_batch failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		u8 ref[80];
		int r = ecccdh_derive_secret(privs[i], peers[i], plen, ref, slen);
		if (r != rets[i]) {
			CHECK(0, "%s: item %u returns %d, ecccdh_derive_secret %d", curve, i, rets[i], r);
		} else if (!r && memcmp(ref, secs[i], slen)) {
			CHECK(0, "%s: item %u secret differs", curve, i);
		}
		nerr += r ? 1 : 0;
	}
	printf("ecccdh_derive_secret_batch %-10s %u items, %u rejected: %s\n", curve, n, nerr, failures == before ? "ok" : "FAILED");
	free(ours); free(privs); free(peers); free(secs); free(rets); free(peerbuf); free(secbuf);
}

/* ---- ec_verify_batch / ec_verify_batch_results against ec_verify ---- */
static void check_verify(const char *curve, ec_alg_type sig_type, hash_alg_type hash_type, const char *label, u32 n, int on_gpu)
{
	ec_params params;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair));
	const ec_pub_key **pubs = calloc(n, sizeof(*pubs));
	const u8 **sigs = calloc(n, sizeof(*sigs)), **msgs = calloc(n, sizeof(*msgs)), **adatas = calloc(n, sizeof(*adatas));
	u8 *siglens = calloc(n, 1), *sigbuf, *msgbuf = calloc(n, 64), siglen = 0;
	u32 *msglens = calloc(n, sizeof(u32));
	u16 *adlens = calloc(n, sizeof(u16));
	int *res = calloc(n, sizeof(int));
	static const u8 ctx[] = "libecc_amd compat check";
	const int needs_ctx =
#if defined(WITH_SIG_EDDSA25519)
		(sig_type == EDDSA25519CTX) || (sig_type == EDDSA25519PH) ||
#endif
#if defined(WITH_SIG_EDDSA448)
		(sig_type == EDDSA448) || (sig_type == EDDSA448PH) ||
#endif
		0;
	u32 i, nbad = 0, scratch_len = 0;
	int r, check = 0;
	const u32 before = failures;
	verify_batch_scratch_pad *pad = NULL;
	if (load_params(curve, &params) || ec_get_sig_len(&params, sig_type, hash_type, &siglen)) {
		CHECK(0, "%s: setup", label);
		return;
	}
	sigbuf = calloc(n, siglen);
	for (i = 0; i < n; i++) {
		u32 ml;
		if (ec_key_pair_gen(&kps[i], &params, sig_type) || get_random((u8 *)&ml, sizeof(ml)) ) {
			CHECK(0, "%s: key generation", label);
			return;
		}
		ml %= 64;
		msglens[i] = ml;
		msgs[i] = msgbuf + (size_t)i * 64;
		if (ml && get_random(msgbuf + (size_t)i * 64, ml)) {
			return;
		}
		pubs[i] = &kps[i].pub_key;
		sigs[i] = sigbuf + (size_t)i * siglen;
		siglens[i] = siglen;
		adatas[i] = needs_ctx ? ctx : NULL;
		adlens[i] = needs_ctx ? (u16)(sizeof(ctx) - 1) : 0;
		if (ec_sign(sigbuf + (size_t)i * siglen, siglen, &kps[i], msgs[i], msglens[i], sig_type, hash_type, adatas[i], adlens[i])) {
			CHECK(0, "%s: ec_sign", label);
			return;
		}
	}
	r = is_verify_batch_mode_supported(sig_type, &check);
	if (on_gpu < 0) {
		/* an algorithm without any batch form (libecc: unsupported_verify_batch): the replaced symbols must keep saying so */
		CHECK(!r && !check, "%s: is_verify_batch_mode_supported says yes", label);
		r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		CHECK(r == -1, "%s: ec_verify_batch accepted a batch of an algorithm without a batch form", label);
		r = ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, res);
		CHECK(r == -1, "%s: ec_verify_batch_results took an algorithm it does not implement", label);
		printf("ec_verify_batch %-28s %u items: %s\n", label, n, failures == before ? "ok" : "FAILED");
		return;
	}
	CHECK(!r && check, "%s: is_verify_batch_mode_supported says no", label);
	/* 1. the three calls of ec_self_tests_core.c: no scratch pad, length query, with scratch pad */
	r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
	CHECK(r == 0, "%s: valid batch rejected (no scratch pad): %d", label, r);
	r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, &scratch_len);
	CHECK(r == 0, "%s: valid batch rejected (length query): %d", label, r);
	scratch_len = (u32)(((2 * (size_t)n) + 1) * sizeof(verify_batch_scratch_pad));
	pad = malloc(scratch_len);
	r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, pad, &scratch_len);
	CHECK(r == 0, "%s: valid batch rejected (scratch pad): %d", label, r);
	free(pad);
	/* 1b. one key under a second copy of the parameters, then a missing key: the schemes libecc has a batch form for answer -1 ("all our public
	 * keys have the same parameters", sig/eddsa.c:2358, sig/bip0340.c:843-845) and so must the replaced symbol; our ECDSA form groups by parameters */
	if (n >= 3) {
		static ec_params params2;
		ec_pub_key alt;
		u8 kbuf[3 * 80];
		const u8 klen = (u8)(3 * BYTECEIL(params.ec_fp.p_bitlen));
		int lib_check = 0;
		if (load_params(curve, &params2) || ec_pub_key_export_to_buf(&kps[1].pub_key, kbuf, klen) ||
		    ec_pub_key_import_from_buf(&alt, &params2, kbuf, klen, sig_type)) {
			CHECK(0, "%s: the key under a second copy of the parameters", label);
			return;
		}
		pubs[1] = &alt;
		r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		if (!libecc_cpu_is_verify_batch_mode_supported(sig_type, &lib_check) && lib_check) {
			const int ref = libecc_cpu_ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
			CHECK(r == ref && r == -1, "%s: a key under other parameters: %d, libecc's ec_verify_batch %d", label, r, ref);
		} else {
			CHECK(r == 0, "%s: a key under a copy of the parameters rejected: %d", label, r);
		}
		pubs[1] = NULL;
		r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		CHECK(r == -1, "%s: a batch with a missing key accepted", label);
		pubs[1] = &kps[1].pub_key;
	}
	/* 1c. keys as an application imports them (Z = 1) with ONE key as a multiplication left it (Z != 1), at an index the layer's 64-key sample
	 * does not look at: the packing steps must notice and the call start over (libecc_amd_compat.c: affine_guess_check) */
	if (n >= 130 && on_gpu > 0) {
		ec_pub_key *aff = calloc(n, sizeof(ec_pub_key));
		int one = 0;
		for (i = 0; i < n; i++) {
			if (i == 1) {
				continue;
			}
			aff[i] = kps[i].pub_key;   /* (the copy shares the parameters; its point is normalised as ec_pub_key_import_from_aff_buf leaves one) */
			if (prj_pt_unique(&aff[i].y, &aff[i].y)) {
				CHECK(0, "%s: normalisation of key %u", label, i);
				return;
			}
			pubs[i] = &aff[i];
		}
		const unsigned long restarts0 = ecamd_compat_verify_restarts();
		int z1 = 0;
		CHECK(!nn_isone(&(aff[0].y.Z.fp_val), &one) && one, "%s: an imported key without Z = 1", label);
		r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		CHECK(r == 0, "%s: valid batch of imported keys and one projective key rejected: %d", label, r);
		/* ECDSA and the one-call Schnorr form send affine keys as X || Y: there the guess must have been caught (EdDSA always sends X || Y || Z) */
		if (!nn_isone(&(kps[1].pub_key.y.Z.fp_val), &z1) && !z1 && !getenv("ECAMD_COMPAT_FULL_SCAN") && !getenv("ECAMD_COMPAT_PRJ_KEYS") &&
		    (sig_type == ECDSA || sig_type == DECDSA)) {
			CHECK(ecamd_compat_verify_restarts() > restarts0, "%s: a projective key among affine ones did not restart the call", label);
		}
		r = ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, res);
		CHECK(r == 0, "%s: ec_verify_batch_results (imported keys) failed", label);
		for (i = 0; i < n && !r; i++) {
			CHECK(res[i] == 0, "%s: item %u of the imported-key batch: %d", label, i, res[i]);
		}
		for (i = 0; i < n; i++) {
			pubs[i] = &kps[i].pub_key;
		}
		free(aff);
	}
	/* 2. spoil some items: a flipped signature bit, another message, another key's signature, a short signature */
	for (i = 0; i < n; i += 7) {
		switch ((i / 7) % 4) {
		case 0: sigbuf[(size_t)i * siglen + (i % siglen)] ^= 0x10; break;
		case 1: msgbuf[(size_t)i * 64] ^= 1; if (!msglens[i]) { msglens[i] = 1; } break;
		case 2: sigs[i] = sigbuf + (size_t)((i + 1) % n) * siglen; break;
		default: siglens[i] = (u8)(siglen - 1); break;
		}
	}
	r = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
	CHECK(r == -1, "%s: spoiled batch accepted", label);
	r = ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, res);
	if (!on_gpu) {
		CHECK(r == -1, "%s: ec_verify_batch_results took an algorithm it does not implement", label);
		printf("ec_verify_batch %-28s %u items: %s\n", label, n, failures == before ? "ok" : "FAILED");
		return;
	}
	CHECK(r == 0, "%s: ec_verify_batch_results failed", label);
	for (i = 0; i < n; i++) {
		const int ref = ec_verify(sigs[i], siglens[i], pubs[i], msgs[i], msglens[i], sig_type, hash_type, adatas[i], adlens[i]);
		if (ref != res[i]) {
			CHECK(0, "%s: item %u GPU %d, ec_verify %d", label, i, res[i], ref);
		}
		nbad += ref ? 1 : 0;
	}
	printf("ec_verify_batch %-28s %u items, %u rejected by ec_verify: %s\n", label, n, nbad, failures == before ? "ok" : "FAILED");
	free(kps); free(pubs); free(sigs); free(msgs); free(adatas); free(siglens); free(sigbuf); free(msgbuf); free(msglens); free(adlens); free(res);
}

/* ---- a public key that is the point at infinity: libecc imports (0 : 1 : 0) and verifies against it (W' = uG) ---- */
static void check_inf_key(const char *curve, hash_alg_type hash_type, u32 n)
{
	ec_params params;
	ec_pub_key *keys = calloc(n, sizeof(ec_pub_key));
	const ec_pub_key **pubs = calloc(n, sizeof(*pubs));
	const u8 **sigs = calloc(n, sizeof(*sigs)), **msgs = calloc(n, sizeof(*msgs));
	u8 *siglens = calloc(n, 1), *sigbuf, *msgbuf = calloc(n, 32), siglen = 0, kb[3 * 72];
	u32 *msglens = calloc(n, sizeof(u32)), i, clen, qlen, nacc = 0;
	int *res = calloc(n, sizeof(int));
	const hash_mapping *hm = NULL;
	const u32 before = failures;
	if (load_params(curve, &params) || ec_get_sig_len(&params, ECDSA, hash_type, &siglen) || get_hash_by_type(hash_type, &hm) || !hm) {
		CHECK(0, "inf key %s: setup", curve);
		return;
	}
	clen = (u32)BYTECEIL(params.ec_fp.p_bitlen);
	qlen = (u32)BYTECEIL(params.ec_gen_order_bitlen);
	sigbuf = calloc(n, siglen);
	memset(kb, 0, sizeof(kb));
	kb[2 * clen - 1] = 1;   /* (0 : 1 : 0) */
	for (i = 0; i < n; i++) {
		hash_context hc;
		u8 dig[MAX_DIGEST_SIZE];
		nn e, s, sinv, u, r;
		prj_pt W;
		aff_pt Wa;
		bitcnt_t rshift = 0;
		if (ec_pub_key_import_from_buf(&keys[i], &params, kb, (u8)(3 * clen), ECDSA)) {
			CHECK(0, "inf key %s: libecc does not import the point at infinity as a key", curve);
			return;
		}
		pubs[i] = &keys[i];
		msgs[i] = msgbuf + (size_t)i * 32;
		msglens[i] = 32;
		sigs[i] = sigbuf + (size_t)i * siglen;
		siglens[i] = siglen;
		get_random(msgbuf + (size_t)i * 32, 32);
		/* r = x([e / s]G) mod q for a random s: steps 2-6, 9 of __ecdsa_verify_finalize done by the application */
		hm->hfunc_init(&hc); hm->hfunc_update(&hc, msgs[i], 32); hm->hfunc_finalize(&hc, dig);
		if ((hm->digest_size * 8) > params.ec_gen_order_bitlen) {
			rshift = (bitcnt_t)((hm->digest_size * 8) - params.ec_gen_order_bitlen);
		}
		nn_init_from_buf(&e, dig, hm->digest_size);
		if (rshift) {
			nn_rshift_fixedlen(&e, &e, rshift);
		}
		nn_mod(&e, &e, &params.ec_gen_order);
		nn_get_random_mod(&s, &params.ec_gen_order);
		nn_modinv(&sinv, &s, &params.ec_gen_order);
		nn_mod_mul(&u, &e, &sinv, &params.ec_gen_order);
		prj_pt_mul(&W, &u, &params.ec_gen);
		prj_pt_to_aff(&Wa, &W);
		nn_mod(&r, &(Wa.x.fp_val), &params.ec_gen_order);
		if (i % 3 == 2) {
			nn_inc(&r, &r);   /* another r: rejected */
		}
		nn_export_to_buf(sigbuf + (size_t)i * siglen, (u16)qlen, &r);
		nn_export_to_buf(sigbuf + (size_t)i * siglen + qlen, (u16)qlen, &s);
	}
	if (ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, n, ECDSA, hash_type, NULL, NULL, res)) {
		CHECK(0, "inf key %s: ec_verify_batch_results failed", curve);
		return;
	}
	for (i = 0; i < n; i++) {
		const int ref = ec_verify(sigs[i], siglens[i], pubs[i], msgs[i], msglens[i], ECDSA, hash_type, NULL, 0);
		CHECK(ref == res[i], "inf key %s: item %u GPU %d, ec_verify %d", curve, i, res[i], ref);
		nacc += ref ? 0 : 1;
	}
	CHECK(nacc > 0 && nacc < n, "inf key %s: expected a mix of verdicts, libecc accepted %u of %u", curve, nacc, n);
	printf("ec_verify_batch key at infinity %-12s %u items, %u accepted by ec_verify: %s\n", curve, n, nacc, failures == before ? "ok" : "FAILED");
}


/* ---- ec_sign_batch against _ec_sign / ec_sign (same nonce source), and the signatures against ec_verify ---- */
static u64 g_det_ctr;
static u64 splitmix(u64 *x)
{
	u64 z = (*x += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}
/* a deterministic nonce source with the prototype of nn_get_random_mod: value in [1, q - 1], with k = 1, k = q - 1 and k = 2 mixed in */
static int det_rand(nn_t out, nn_src_t q)
{
	u8 buf[160];
	nn t, qm1;
	bitcnt_t qb = 0;
	u32 i, len;
	const u64 c = g_det_ctr++;
	u64 st = c * 0x1234567ULL + 99;
	int ret;
	t.magic = qm1.magic = WORD(0);
	ret = nn_bitlen(q, &qb); EG(ret, err);
	len = 2 * (u32)BYTECEIL(qb);
	for (i = 0; i < len; i++) {
		buf[i] = (u8)splitmix(&st);
	}
	ret = nn_init(&qm1, 0); EG(ret, err);
	ret = nn_dec(&qm1, q); EG(ret, err);
	if (c % 61 == 7) {
		ret = nn_init(out, 0); EG(ret, err);
		ret = nn_one(out);
	} else if (c % 61 == 8) {
		ret = nn_copy(out, &qm1);
	} else {
		ret = nn_init_from_buf(&t, buf, (u16)len); EG(ret, err);
		ret = nn_mod(out, &t, &qm1); EG(ret, err);
		ret = nn_inc(out, out);
	}
err:
	nn_uninit(&t);
	nn_uninit(&qm1);
	return ret;
}

static void check_sign(const char *curve, ec_alg_type sig_type, hash_alg_type hash_type, const char *label, u32 n, int deterministic)
{
	ec_params params;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair));
	const ec_key_pair **kpp = calloc(n, sizeof(*kpp));
	u8 **sigs = calloc(n, sizeof(*sigs)), *sigbuf, *refbuf, *msgbuf = calloc(n, 64), siglen = 0;
	const u8 **msgs = calloc(n, sizeof(*msgs)), **adatas = calloc(n, sizeof(*adatas));
	u32 *msglens = calloc(n, sizeof(u32)), i, nfail = 0, pos = 0;
	u16 *adlens = calloc(n, sizeof(u16));
	int *rets = calloc(n, sizeof(int));
	static const u8 ctx[] = "libecc_amd compat check";
	const int is_ed =
#if defined(WITH_SIG_EDDSA25519)
		(sig_type == EDDSA25519) || (sig_type == EDDSA25519CTX) || (sig_type == EDDSA25519PH) ||
#endif
#if defined(WITH_SIG_EDDSA448)
		(sig_type == EDDSA448) || (sig_type == EDDSA448PH) ||
#endif
		0;
	const int needs_ctx = is_ed && sig_type != EDDSA25519;
	const u32 before = failures;
	if (load_params(curve, &params) || ec_get_sig_len(&params, sig_type, hash_type, &siglen)) {
		CHECK(0, "sign %s: setup", label);
		return;
	}
	sigbuf = calloc(n, siglen);
	refbuf = calloc(n, siglen);
	for (i = 0; i < n; i++) {
		u32 ml;
		if (ec_key_pair_gen(&kps[i], &params, sig_type) || get_random((u8 *)&ml, sizeof(ml))) {
			CHECK(0, "sign %s: key generation", label);
			return;
		}
		ml %= 64;
		msglens[i] = ml;
		msgs[i] = msgbuf + (size_t)i * 64;
		if (ml && get_random(msgbuf + (size_t)i * 64, ml)) {
			return;
		}
		kpp[i] = &kps[i];
		sigs[i] = sigbuf + (size_t)i * siglen;
		adatas[i] = needs_ctx ? ctx : NULL;
		adlens[i] = needs_ctx ? (u16)(sizeof(ctx) - 1) : 0;
	}
	/* edge items: private keys 1 and q - 1 (ECDSA family), a key of another algorithm, a NULL key pair, a NULL message of
	 * non-zero length, an empty message */
	if (n >= 16) {
		if (sig_type == ECDSA || sig_type == DECDSA) {
			nn one;
			nn_init(&one, 0); nn_one(&one);
			nn_one(&kps[1].priv_key.x);
			init_pubkey_from_privkey(&kps[1].pub_key, &kps[1].priv_key);
			nn_sub(&kps[2].priv_key.x, &params.ec_gen_order, &one);
			init_pubkey_from_privkey(&kps[2].pub_key, &kps[2].priv_key);
			nn_copy(&kps[3].priv_key.x, &params.ec_gen_order);   /* x = q: "private key is not compliant" */
		}
		kps[4].priv_key.key_type = (sig_type == ECDSA) ? ECKCDSA : ECDSA;
		kpp[5] = NULL;
		msgs[6] = NULL; msglens[6] = 3;
		msglens[7] = 0;
	}
	g_det_ctr = 1000;
	if (ec_sign_batch(sigs, siglen, kpp, msgs, msglens, n, deterministic == 2 ? det_rand : NULL, sig_type, hash_type, adatas, adlens, rets)) {
		CHECK(0, "sign %s: ec_sign_batch failed", label);
		return;
	}
	g_det_ctr = 1000;
	for (i = 0; i < n; i++) {
		u8 *ref = refbuf + (size_t)i * siglen;
		int r = -1;
		if (deterministic) {
			/* The batch form calls the nonce hook once per item of the group, in index order (items without a key pair
			 * belong to no group); the scalar function is given the same nonce by positioning the hook's counter. */
			g_det_ctr = 1000 + pos;
			pos += kpp[i] ? 1 : 0;
			r = (kpp[i] && (msgs[i] || !msglens[i])) ?
				_ec_sign(ref, siglen, kpp[i], msgs[i], msglens[i], deterministic == 2 ? det_rand : NULL, sig_type, hash_type, adatas[i], adlens[i]) : -1;
			if (r != rets[i]) {
				CHECK(0, "sign %s: item %u returns %d, _ec_sign %d", label, i, rets[i], r);
			} else if (!r && memcmp(ref, sigs[i], siglen)) {
				CHECK(0, "sign %s: item %u signature differs from _ec_sign", label, i);
			}
		} else {
			/* random nonces: the signature must verify with libecc's ec_verify; failures must be libecc's failures */
			r = (kpp[i] && (msgs[i] || !msglens[i])) ? ec_sign(ref, siglen, kpp[i], msgs[i], msglens[i], sig_type, hash_type, adatas[i], adlens[i]) : -1;
			if (r != rets[i]) {
				CHECK(0, "sign %s: item %u returns %d, ec_sign %d", label, i, rets[i], r);
			} else if (!r && ec_verify(sigs[i], siglen, &kpp[i]->pub_key, msgs[i], msglens[i], sig_type, hash_type, adatas[i], adlens[i])) {
				CHECK(0, "sign %s: item %u signature rejected by ec_verify", label, i);
			}
		}
		nfail += r ? 1 : 0;
	}
	printf("ec_sign_batch %-30s %u items (%s), %u fail in libecc too: %s\n", label, n,
	       deterministic == 2 ? "nonce hook, bytes equal" : deterministic ? "deterministic, bytes equal" : "random nonces, ec_verify", nfail,
	       failures == before ? "ok" : "FAILED");
	free(kps); free(kpp); free(sigs); free(sigbuf); free(refbuf); free(msgbuf); free(msgs); free(adatas); free(msglens); free(adlens); free(rets);
}

/* ---- key pairs: ec_key_pair_gen_batch / ec_key_pair_import_from_priv_key_buf_batch against the scalar functions ---- */
static int pub_equal(const ec_pub_key *a, const ec_pub_key *b)
{
	int cmp = 1, z1 = 0, z2 = 0;
	if (a->magic != b->magic || a->key_type != b->key_type || a->params != b->params) {
		return 0;
	}
	if (prj_pt_iszero(&a->y, &z1) || prj_pt_iszero(&b->y, &z2) || z1 != z2) {
		return 0;
	}
	return z1 || (!prj_pt_cmp(&a->y, &b->y, &cmp) && !cmp);
}

static void check_keys(const char *curve, ec_alg_type alg, const char *label, u32 n)
{
	ec_params params;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair)), *imp = calloc(n, sizeof(ec_key_pair));
	ec_pub_key *pubs = calloc(n, sizeof(ec_pub_key));
	const ec_priv_key **privs = calloc(n, sizeof(*privs));
	u8 *bufs = calloc(n, 80), qlen;
	const u8 **bufp = calloc(n, sizeof(*bufp));
	int *rets = calloc(n, sizeof(int));
	u32 i, nfail = 0;
	const u32 before = failures;
	if (load_params(curve, &params)) {
		CHECK(0, "keys %s: setup", label);
		return;
	}
	qlen = (u8)BYTECEIL(params.ec_gen_order_bitlen);
	/* 1. generation: every key pair is consistent (the public key is what libecc derives from the private key) */
	if (ec_key_pair_gen_batch(kps, &params, alg, n, rets)) {
		CHECK(0, "keys %s: ec_key_pair_gen_batch failed", label);
		return;
	}
	for (i = 0; i < n; i++) {
		ec_pub_key ref;
		int r;
		if (rets[i]) {
			CHECK(0, "keys %s: generation of item %u failed", label, i);
			continue;
		}
#if defined(WITH_ECCCDH)
		r = (alg == ECCCDH) ? ecccdh_init_pub_key(&ref, &kps[i].priv_key) : init_pubkey_from_privkey(&ref, &kps[i].priv_key);
#else
		r = init_pubkey_from_privkey(&ref, &kps[i].priv_key);
#endif
		CHECK(!r && key_pair_check_initialized_and_type(&kps[i], alg) == 0 && pub_equal(&ref, &kps[i].pub_key), "keys %s: generated pair %u is inconsistent", label, i);
		if (i && !nn_cmp(&kps[i].priv_key.x, &kps[i - 1].priv_key.x, &r) && !r) {
			CHECK(0, "keys %s: two equal private keys in a row", label);
		}
		privs[i] = &kps[i].priv_key;
	}
	/* 2. init_pubkey_from_privkey_batch on the same private keys */
	if (init_pubkey_from_privkey_batch(pubs, privs, n, rets)) {
		CHECK(0, "keys %s: init_pubkey_from_privkey_batch failed", label);
	} else {
		for (i = 0; i < n; i++) {
			CHECK(!rets[i] && pub_equal(&pubs[i], &kps[i].pub_key), "keys %s: init_pubkey_from_privkey_batch item %u", label, i);
		}
	}
	/* 3. import from buffers, edge values included: 0, 1, q - 1, q, q + 1, all ones, short */
#if defined(WITH_ECCCDH)
	if (alg != ECCCDH)
#endif
	{
		for (i = 0; i < n; i++) {
			get_random(bufs + (size_t)i * 80, qlen);
			bufs[(size_t)i * 80] &= 0x3f;
			bufp[i] = bufs + (size_t)i * 80;
		}
		if (n >= 16) {
			nn t, one;
			nn_init(&one, 0); nn_one(&one);
			memset(bufs + 0 * 80, 0, qlen);
			memset(bufs + 1 * 80, 0, qlen); bufs[1 * 80 + qlen - 1] = 1;
			nn_sub(&t, &params.ec_gen_order, &one); nn_export_to_buf(bufs + 2 * 80, qlen, &t);
			nn_export_to_buf(bufs + 3 * 80, qlen, &params.ec_gen_order);
			nn_add(&t, &params.ec_gen_order, &one); nn_export_to_buf(bufs + 4 * 80, qlen, &t);
			memset(bufs + 5 * 80, 0xff, qlen);
			nn_sub(&t, &params.ec_gen_order, &one); nn_sub(&t, &t, &one); nn_export_to_buf(bufs + 6 * 80, qlen, &t);   /* q - 2 (SM2's bound) */
			bufp[7] = NULL;
		}
		if (ec_key_pair_import_from_priv_key_buf_batch(imp, &params, bufp, qlen, alg, n, rets)) {
			CHECK(0, "keys %s: ec_key_pair_import_from_priv_key_buf_batch failed", label);
		} else {
			for (i = 0; i < n; i++) {
				ec_key_pair ref;
				int r = bufp[i] ? ec_key_pair_import_from_priv_key_buf(&ref, &params, bufp[i], qlen, alg) : -1, c = 1;
				if (r != rets[i]) {
					CHECK(0, "keys %s: import item %u returns %d, libecc %d", label, i, rets[i], r);
				} else if (!r) {
					CHECK(!nn_cmp(&ref.priv_key.x, &imp[i].priv_key.x, &c) && !c && pub_equal(&ref.pub_key, &imp[i].pub_key), "keys %s: imported pair %u differs", label, i);
				}
				nfail += r ? 1 : 0;
			}
		}
	}
	printf("ec_key_pair_{gen,import}_batch %-22s %u items, %u imports fail in libecc too: %s\n", label, n, nfail, failures == before ? "ok" : "FAILED");
	free(kps); free(imp); free(pubs); free(privs); free(bufs); free(bufp); free(rets);
}

#if defined(WITH_SIG_EDDSA25519)
static void check_eddsa_import(u32 n)
{
	ec_params params;
	ec_key_pair *imp = calloc(n, sizeof(ec_key_pair));
	u8 *bufs = calloc(n, 32);
	const u8 **bufp = calloc(n, sizeof(*bufp));
	int *rets = calloc(n, sizeof(int));
	u32 i;
	const u32 before = failures;
	if (load_params("WEI25519", &params)) {
		return;
	}
	for (i = 0; i < n; i++) {
		get_random(bufs + (size_t)i * 32, 32);
		bufp[i] = bufs + (size_t)i * 32;
	}
	/* RFC 8032 section 7.1 TEST 1 secret key: its public key is d75a9801... */
	if (n >= 2) {
		static const u8 sk1[32] = {0x9d, 0x61, 0xb1, 0x9d, 0xef, 0xfd, 0x5a, 0x60, 0xba, 0x84, 0x4a, 0xf4, 0x92, 0xec, 0x2c, 0xc4,
					   0x44, 0x49, 0xc5, 0x69, 0x7b, 0x32, 0x69, 0x19, 0x70, 0x3b, 0xac, 0x03, 0x1c, 0xae, 0x7f, 0x60};
		memcpy(bufs, sk1, 32);
	}
	if (eddsa_import_key_pair_from_priv_key_buf_batch(imp, bufp, 32, &params, EDDSA25519, n, rets)) {
		CHECK(0, "eddsa import: batch call failed");
		return;
	}
	for (i = 0; i < n; i++) {
		ec_key_pair ref;
		u8 e1[32], e2[32];
		int r = eddsa_import_key_pair_from_priv_key_buf(&ref, bufp[i], 32, &params, EDDSA25519), c = 1;
		if (r != rets[i]) {
			CHECK(0, "eddsa import: item %u returns %d, libecc %d", i, rets[i], r);
		} else if (!r) {
			CHECK(!nn_cmp(&ref.priv_key.x, &imp[i].priv_key.x, &c) && !c && pub_equal(&ref.pub_key, &imp[i].pub_key), "eddsa import: pair %u differs", i);
			CHECK(!eddsa_export_pub_key(&ref.pub_key, e1, 32) && !eddsa_export_pub_key(&imp[i].pub_key, e2, 32) && !memcmp(e1, e2, 32), "eddsa import: encoded key %u differs", i);
			if (i == 0 && n >= 2) {
				CHECK(e2[0] == 0xd7 && e2[1] == 0x5a && e2[2] == 0x98 && e2[3] == 0x01 && e2[31] == 0x1a, "eddsa import: RFC 8032 TEST 1 public key");
			}
		}
	}
	printf("eddsa_import_key_pair_from_priv_key_buf_batch   %u items: %s\n", n, failures == before ? "ok" : "FAILED");
	free(imp); free(bufs); free(bufp); free(rets);
}
#endif

/* ---- x25519_batch / x448_batch against x25519 / x448 ---- */
static void check_xdh(u32 len, u32 n)
{
	u8 *kb = calloc(n, len), *ub = calloc(n, len), *rb = calloc(n, len), *pb = calloc(n, len), ref[56];
	const u8 **kp = calloc(n, sizeof(*kp)), **up = calloc(n, sizeof(*up));
	u8 **rp = calloc(n, sizeof(*rp)), **pp = calloc(n, sizeof(*pp));
	int *rets = calloc(n, sizeof(int));
	u32 i, nrej = 0;
	const u32 before = failures;
	for (i = 0; i < n; i++) {
		get_random(kb + (size_t)i * len, (u16)len);
		kp[i] = kb + (size_t)i * len;
		rp[i] = rb + (size_t)i * len;
		pp[i] = pb + (size_t)i * len;
	}
	/* public keys first (base point), then u = those public keys (on the curve), with edge u values mixed in */
	if (len == 32 ? x25519_init_pub_key_batch(kp, pp, n, rets) : x448_init_pub_key_batch(kp, pp, n, rets)) {
		CHECK(0, "x%u: init_pub_key_batch failed", len == 32 ? 25519 : 448);
		return;
	}
	for (i = 0; i < n; i++) {
		const int r = len == 32 ? x25519_init_pub_key(kp[i], ref) : x448_init_pub_key(kp[i], ref);
		CHECK(r == rets[i] && (r || !memcmp(ref, pp[i], len)), "x%u: public key %u (ret %d / %d)", len == 32 ? 25519 : 448, i, rets[i], r);
		memcpy(ub + (size_t)i * len, pp[(i * 7 + 1) % n], len);
		up[i] = ub + (size_t)i * len;
	}
	if (n >= 16) {
		memset(ub + 0 * len, 0, len);                                   /* u = 0: small order */
		memset(ub + 1 * len, 0, len); ub[1 * len] = 1;                   /* u = 1 */
		memset(ub + 2 * len, 0xff, len);                                /* non-canonical */
		get_random(ub + 3 * len, (u16)len);                             /* random: on the twist with probability 1/2 */
		get_random(ub + 4 * len, (u16)len);
		get_random(ub + 5 * len, (u16)len);
		memset(kb + 6 * len, 0, len);                                   /* k = 0 before clamping */
		memset(kb + 7 * len, 0xff, len);
		memset(ub + 8 * len, 0, len); ub[8 * len] = (len == 32) ? 9 : 5; /* the base point */
	}
	if (len == 32 ? x25519_batch(kp, up, rp, n, rets) : x448_batch(kp, up, rp, n, rets)) {
		CHECK(0, "x%u: batch call failed", len == 32 ? 25519 : 448);
		return;
	}
	for (i = 0; i < n; i++) {
		const int r = len == 32 ? x25519(kp[i], up[i], ref) : x448(kp[i], up[i], ref);
		CHECK(r == rets[i] && (r || !memcmp(ref, rp[i], len)), "x%u: item %u (ret %d / %d)", len == 32 ? 25519 : 448, i, rets[i], r);
		nrej += r ? 1 : 0;
	}
	printf("x%u_batch / x%u_init_pub_key_batch   %u items, %u rejected by libecc too: %s\n", len == 32 ? 25519 : 448, len == 32 ? 25519 : 448, n, nrej,
	       failures == before ? "ok" : "FAILED");
	free(kb); free(ub); free(rb); free(pb); free(kp); free(up); free(rp); free(pp); free(rets);
}

/* ---- a group that is NOT the built-in curve of its name: same curve and name, another generator ---- */
static void check_foreign_generator(u32 n)
{
	ec_params base, mine;
	nn two;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair));
	int *rets = calloc(n, sizeof(int));
	u32 i;
	const u32 before = failures;
	if (load_params("SECP256R1", &base)) {
		return;
	}
	mine = base;
	/* G' = [2]G: same order, same curve, same curve_name */
	nn_init(&two, 0); nn_one(&two); nn_inc(&two, &two);
	if (prj_pt_mul(&mine.ec_gen, &two, &base.ec_gen)) {
		CHECK(0, "foreign generator: setup");
		return;
	}
	/* the point structures of `mine` must refer to mine's own curve object */
	mine.ec_gen.crv = &mine.ec_curve;
	mine.ec_gen.X.ctx = mine.ec_gen.Y.ctx = mine.ec_gen.Z.ctx = &mine.ec_fp;
	mine.ec_curve.a.ctx = mine.ec_curve.b.ctx = mine.ec_curve.a_monty.ctx = mine.ec_curve.b3.ctx = mine.ec_curve.b_monty.ctx = mine.ec_curve.b3_monty.ctx = &mine.ec_fp;
	if (ec_key_pair_gen_batch(kps, &base, ECDSA, n, rets) || ec_key_pair_gen_batch(kps, &mine, ECDSA, n, rets)) {
		CHECK(0, "foreign generator: ec_key_pair_gen_batch failed");
		return;
	}
	for (i = 0; i < n; i++) {
		ec_pub_key ref;
		CHECK(!rets[i] && !init_pubkey_from_privkey(&ref, &kps[i].priv_key) && pub_equal(&ref, &kps[i].pub_key),
		      "foreign generator: item %u was computed with another generator", i);
	}
	printf("ec_params with a foreign generator under a built-in name   %u items: %s\n", n, failures == before ? "ok" : "FAILED");
	free(kps); free(rets);
}

/* ---- "compat_check bench <log2 n>": end-to-end rate of ec_verify_batch as a libecc application sees it -- libecc structures
 * in, one int out, the marshalling and the message hashing on the host threads included.  `base` distinct (key, message,
 * signature) triples made with libecc's ec_sign, repeated to n pointers (the batch arrays are arrays of pointers). ---- */
/* ---- a commitment shifted by a point of small order on a curve with a cofactor.  ECFSDSA on WEI25519: W' = W + D,
 * D = [q]P != O of order 2, 4 or 8.  ec_verify rejects the item (W' != [s]G - [e]Y); a random linear combination of the batch
 * accepts it whenever z_i D = O -- probability 1 / ord(D) -- so ec_verify_batch would differ from a loop of ec_verify if the
 * multi-scalar form served this curve.  One shifted item per round, `rounds` rounds with fresh randomness each. ---- */
static void check_torsion_shift(u32 rounds)
{
	enum { N = 24 };
	ec_params params;
	ec_key_pair kps[N];
	const ec_pub_key *pubs[N];
	const u8 *sigs[N], *msgs[N];
	u8 sigbuf[N][3 * 32 + 8], msgbuf[N][16], siglens[N], siglen = 0, saved[64];
	u32 msglens[N], i, r;
	int res[N];
	prj_pt P, D, W;
	aff_pt A;
	fp x, y1, y2;
	int iszero = 1, tries = 0;
	const u32 before = failures;
	if (load_params("WEI25519", &params) || ec_get_sig_len(&params, ECFSDSA, SHA512, &siglen) || siglen != 96) {
		CHECK(0, "torsion shift: setup");
		return;
	}
	/* D: the cofactor-clearing complement of a random curve point */
	while (iszero && tries++ < 64) {
		u8 xb[32];
		if (get_random(xb, sizeof(xb))) {
			return;
		}
		xb[0] &= 0x3f;
		if (fp_init(&x, &(params.ec_fp)) || fp_init(&y1, &(params.ec_fp)) || fp_init(&y2, &(params.ec_fp)) || fp_import_from_buf(&x, xb, 32)) {
			continue;
		}
		if (aff_pt_y_from_x(&y1, &y2, &x, &(params.ec_curve))) {
			continue;   /* x is no abscissa */
		}
		if (aff_pt_init_from_coords(&A, &(params.ec_curve), &x, &y1) || ec_shortw_aff_to_prj(&P, &A) ||
		    _prj_pt_unprotected_mult(&D, &(params.ec_gen_order), &P) || prj_pt_iszero(&D, &iszero)) {
			CHECK(0, "torsion shift: building D");
			return;
		}
	}
	CHECK(!iszero, "torsion shift: no point of small order found");
	for (i = 0; i < N; i++) {
		msglens[i] = 16;
		msgs[i] = msgbuf[i];
		pubs[i] = &kps[i].pub_key;
		sigs[i] = sigbuf[i];
		siglens[i] = siglen;
		if (ec_key_pair_gen(&kps[i], &params, ECFSDSA) || get_random(msgbuf[i], 16) ||
		    ec_sign(sigbuf[i], siglen, &kps[i], msgs[i], 16, ECFSDSA, SHA512, NULL, 0)) {
			CHECK(0, "torsion shift: signing");
			return;
		}
	}
	CHECK(ec_verify_batch(sigs, siglens, pubs, msgs, msglens, N, ECFSDSA, SHA512, NULL, NULL, NULL, NULL) == 0, "torsion shift: valid batch rejected");
	for (r = 0; r < rounds; r++) {
		const u32 k = r % N;
		int rb;
		memcpy(saved, sigbuf[k], 64);
		if (aff_pt_import_from_buf(&A, sigbuf[k], 64, &(params.ec_curve)) || ec_shortw_aff_to_prj(&W, &A) || prj_pt_add(&W, &W, &D) ||
		    prj_pt_to_aff(&A, &W) || aff_pt_export_to_buf(&A, sigbuf[k], 64)) {
			CHECK(0, "torsion shift: W + D");
			return;
		}
		CHECK(ec_verify(sigs[k], siglen, pubs[k], msgs[k], 16, ECFSDSA, SHA512, NULL, 0) != 0, "torsion shift: ec_verify accepts W + D?");
		rb = ec_verify_batch(sigs, siglens, pubs, msgs, msglens, N, ECFSDSA, SHA512, NULL, NULL, NULL, NULL);
		CHECK(rb == -1, "torsion shift: round %u: ec_verify_batch accepted a batch whose item %u a loop of ec_verify rejects", r, k);
		CHECK(ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, N, ECFSDSA, SHA512, NULL, NULL, res) == 0, "torsion shift: results call");
		for (i = 0; i < N; i++) {
			CHECK(res[i] == (i == k ? -1 : 0), "torsion shift: round %u item %u: %d", r, i, res[i]);
		}
		memcpy(sigbuf[k], saved, 64);
	}
	printf("ec_verify_batch ECFSDSA/WEI25519, W + D with D of small order, %u rounds: %s\n", rounds, failures == before ? "ok" : "FAILED");
}

/* ---- round 6: several application threads inside ec_verify_batch at once (libecc is re-entrant; the batch layer gives every
 * call its own staging and lets the pool pack one call while another has the GPU).  Three threads -- ECDSA/SECP256R1, EDDSA25519 and
 * ECDSA/SECP384R1, each with its own spoiled items -- verify at the same time, several rounds; every item's result must be ec_verify's. ---- */
typedef struct {
	const char *curve, *label;
	ec_alg_type sig_type;
	hash_alg_type hash_type;
	u32 n, rounds;
	int bad;
} conc_arg;
static void *conc_verify_thread(void *a)
{
	conc_arg *C = (conc_arg *)a;
	enum { ML = 24 };
	ec_params params;
	const u32 n = C->n;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair));
	const ec_pub_key **pubs = calloc(n, sizeof(*pubs));
	const u8 **sigs = calloc(n, sizeof(*sigs)), **msgs = calloc(n, sizeof(*msgs)), **adatas = calloc(n, sizeof(*adatas));
	u8 *siglens = calloc(n, 1), *sigbuf, *msgbuf = calloc(n, ML), siglen = 0;
	u32 *msglens = calloc(n, sizeof(u32)), i, r;
	u16 *adlens = calloc(n, sizeof(u16));
	int *res = calloc(n, sizeof(int)), *want = calloc(n, sizeof(int));
	C->bad = 1;
	if (!kps || load_params(C->curve, &params) || ec_get_sig_len(&params, C->sig_type, C->hash_type, &siglen)) {
		return NULL;
	}
	sigbuf = calloc(n, siglen);
	for (i = 0; i < n; i++) {
		if (ec_key_pair_gen(&kps[i], &params, C->sig_type) || get_random(msgbuf + (size_t)i * ML, ML) ||
		    ec_sign(sigbuf + (size_t)i * siglen, siglen, &kps[i], msgbuf + (size_t)i * ML, ML, C->sig_type, C->hash_type, NULL, 0)) {
			return NULL;
		}
		pubs[i] = &kps[i].pub_key; sigs[i] = sigbuf + (size_t)i * siglen; msgs[i] = msgbuf + (size_t)i * ML; siglens[i] = siglen; msglens[i] = ML;
	}
	for (i = 3; i < n; i += 11) {
		sigbuf[(size_t)i * siglen + (i % siglen)] ^= 0x08;
	}
	for (i = 0; i < n; i++) {
		want[i] = ec_verify(sigs[i], siglens[i], pubs[i], msgs[i], msglens[i], C->sig_type, C->hash_type, NULL, 0);
	}
	C->bad = 0;
	for (r = 0; r < C->rounds; r++) {
		memset(res, 0x55, n * sizeof(int));
		if (ec_verify_batch_results(sigs, siglens, pubs, msgs, msglens, n, C->sig_type, C->hash_type, adatas, adlens, res)) {
			C->bad = 1;
		}
		for (i = 0; i < n; i++) {
			C->bad |= (res[i] != want[i]);
		}
		C->bad |= (ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, C->sig_type, C->hash_type, adatas, adlens, NULL, NULL) != -1);
	}
	free(kps); free(pubs); free(sigs); free(msgs); free(adatas); free(siglens); free(sigbuf); free(msgbuf); free(msglens); free(adlens); free(res); free(want);
	return NULL;
}
static void check_concurrent_verify(u32 n)
{
	conc_arg C[3] = {{"SECP256R1", "ECDSA/SECP256R1/SHA256", ECDSA, SHA256, n, 4, 0},
			 {"WEI25519", "EDDSA25519", EDDSA25519, SHA512, n, 4, 0},
			 {"SECP384R1", "ECDSA/SECP384R1/SHA384", ECDSA, SHA384, n > 96 ? n / 2 : n, 4, 0}};
	pthread_t th[3];
	const u32 before = failures;
	const int was_serial = g_rand_expect_serial;
	int t;
	g_rand_expect_serial = 0;   /* (three application threads draw their own keys at once: that is the application's business) */
	for (t = 0; t < 3; t++) {
		pthread_create(&th[t], NULL, conc_verify_thread, &C[t]);
	}
	for (t = 0; t < 3; t++) {
		pthread_join(th[t], NULL);
		CHECK(!C[t].bad, "concurrent callers: %s: a result differs from ec_verify", C[t].label);
	}
	g_rand_expect_serial = was_serial;
	printf("ec_verify_batch from three application threads at once, %u items each: %s\n", n, failures == before ? "ok" : "FAILED");
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void bench_verify(const char *curve, ec_alg_type sig_type, hash_alg_type hash_type, const char *label, u32 n)
{
	enum { base = 512, ML = 48 };
	ec_params params;
	static ec_key_pair kps[base];
	static u8 sigbuf[base][2 * 72], msgbuf[base][ML];
	const ec_pub_key **pubs = calloc(n, sizeof(*pubs));
	const u8 **sigs = calloc(n, sizeof(*sigs)), **msgs = calloc(n, sizeof(*msgs)), **adatas = calloc(n, sizeof(*adatas));
	u8 *siglens = calloc(n, 1), siglen = 0;
	u32 *msglens = calloc(n, sizeof(u32)), i;
	u16 *adlens = calloc(n, sizeof(u16));
	double t0, best = 1e30;
	int r = 0, rep;
	if (load_params(curve, &params) || ec_get_sig_len(&params, sig_type, hash_type, &siglen) || siglen > sizeof(sigbuf[0])) {
		printf("bench %s: setup failed\n", label);
		return;
	}
	for (i = 0; i < base; i++) {
		if (ec_key_pair_gen(&kps[i], &params, sig_type) || get_random(msgbuf[i], ML) ||
		    ec_sign(sigbuf[i], siglen, &kps[i], msgbuf[i], ML, sig_type, hash_type, NULL, 0)) {
			printf("bench %s: signing failed\n", label);
			return;
		}
	}
	for (i = 0; i < n; i++) {
		pubs[i] = &kps[i % base].pub_key;
		sigs[i] = sigbuf[i % base];
		msgs[i] = msgbuf[i % base];
		siglens[i] = siglen;
		msglens[i] = ML;
	}
	for (rep = 0; rep < 4; rep++) {   /* the first call also creates the curve handle and its tables */
		t0 = now_s();
		r |= ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		if (rep && now_s() - t0 < best) {
			best = now_s() - t0;
		}
	}
	printf("bench ec_verify_batch %-24s n = %u: %s, %.1f ms, %.2f M verifications/s end to end (libecc structures in, host hashing and marshalling included)\n",
	       label, n, r ? "REJECTED" : "accepted", best * 1e3, (double)n / best / 1e6);
	free(pubs); free(sigs); free(msgs); free(adatas); free(siglens); free(msglens); free(adlens);
}


/* ec_verify_batch of a Schnorr-type algorithm on n DISTINCT signatures (keys from ec_key_pair_gen_batch, signatures from ec_sign_batch --
 * libecc's own _ec_sign on the pool threads for these algorithms): with the multi-scalar form (default threshold) and without it
 * ($ECAMD_COMPAT_SCHNORR_MSM_MIN=0 in the environment of a second run) */
static void bench_schnorr(const char *curve, ec_alg_type sig_type, hash_alg_type hash_type, const char *label, u32 n)
{
	enum { ML = 32 };
	ec_params params;
	ec_key_pair *kps = calloc(n, sizeof(ec_key_pair));
	const ec_key_pair **kpp = calloc(n, sizeof(*kpp));
	const ec_pub_key **pubs = calloc(n, sizeof(*pubs));
	u8 **sigw = calloc(n, sizeof(*sigw));
	const u8 **sigs = calloc(n, sizeof(*sigs)), **msgs = calloc(n, sizeof(*msgs)), **adatas = calloc(n, sizeof(*adatas));
	u8 *siglens = calloc(n, 1), *sigbuf, *msgbuf = calloc(n, ML), siglen = 0;
	u32 *msglens = calloc(n, sizeof(u32)), i;
	u16 *adlens = calloc(n, sizeof(u16));
	int *rets = calloc(n, sizeof(int)), r = 0, rep;
	double t0, best = 1e30;
	const unsigned long calls0 = ecamd_compat_schnorr_msm_calls();
	if (!kps || load_params(curve, &params) || ec_get_sig_len(&params, sig_type, hash_type, &siglen)) {
		printf("bench %s: setup failed\n", label);
		return;
	}
	sigbuf = calloc(n, siglen);
	if (ec_key_pair_gen_batch(kps, &params, sig_type, n, rets) || get_random(msgbuf, ML)) {
		printf("bench %s: key generation failed\n", label);
		return;
	}
	for (i = 0; i < n; i++) {
		memcpy(msgbuf + (size_t)i * ML, msgbuf, ML);
		memcpy(msgbuf + (size_t)i * ML, &i, sizeof(i));
		kpp[i] = &kps[i];
		pubs[i] = &kps[i].pub_key;
		sigw[i] = sigbuf + (size_t)i * siglen;
		sigs[i] = sigw[i];
		msgs[i] = msgbuf + (size_t)i * ML;
		siglens[i] = siglen;
		msglens[i] = ML;
		r |= rets[i];
	}
	t0 = now_s();
	r |= ec_sign_batch(sigw, siglen, kpp, msgs, msglens, n, NULL, sig_type, hash_type, NULL, NULL, rets);
	printf("bench %s: %u key pairs, %u signatures by libecc's _ec_sign on the pool in %.1f s%s\n", label, n, n, now_s() - t0, r ? " (FAILED)" : "");
	for (rep = 0; rep < 4; rep++) {
		t0 = now_s();
		r |= ec_verify_batch(sigs, siglens, pubs, msgs, msglens, n, sig_type, hash_type, adatas, adlens, NULL, NULL);
		if (rep && now_s() - t0 < best) {
			best = now_s() - t0;
		}
	}
	printf("bench ec_verify_batch %-24s n = %u: %s, %.1f ms, %.2f M verifications/s end to end, %lu multi-scalar calls\n", label, n, r ? "REJECTED" : "accepted",
	       best * 1e3, (double)n / best / 1e6, ecamd_compat_schnorr_msm_calls() - calls0);
	free(kps); free(kpp); free(pubs); free(sigw); free(sigs); free(msgs); free(adatas); free(siglens); free(sigbuf); free(msgbuf); free(msglens); free(adlens); free(rets);
}

/* end-to-end rates of the secret-key entry points: ec_sign_batch, ec_key_pair_gen_batch, x25519_batch (libecc structures /
 * pointer arrays in and out; hashing, nonce generation and marshalling on the host threads included) */
static void bench_secret_half(u32 n)
{
	enum { base = 512, ML = 48 };
	ec_params params;
	static ec_key_pair kps[base];
	static u8 msgbuf[base][ML];
	const ec_key_pair **kpp = calloc(n, sizeof(*kpp));
	const u8 **msgs = calloc(n, sizeof(*msgs));
	u8 **sigs = calloc(n, sizeof(*sigs)), *sigbuf = calloc(n, 64), *kb = calloc(n, 32), *rb = calloc(n, 32);
	const u8 **kp = calloc(n, sizeof(*kp));
	u8 **rp = calloc(n, sizeof(*rp));
	u32 *msglens = calloc(n, sizeof(u32)), i;
	int *rets = calloc(n, sizeof(int)), rep, r = 0;
	ec_key_pair *gen = calloc(n, sizeof(ec_key_pair));
	double t0, best;
	if (load_params("SECP256R1", &params) || !gen) {
		printf("bench: setup failed\n");
		return;
	}
	for (i = 0; i < base; i++) {
		if (ec_key_pair_gen(&kps[i], &params, ECDSA) || get_random(msgbuf[i], ML)) {
			return;
		}
	}
	for (i = 0; i < n; i++) {
		kpp[i] = &kps[i % base];
		msgs[i] = msgbuf[i % base];
		msglens[i] = ML;
		sigs[i] = sigbuf + (size_t)i * 64;
		kp[i] = kb + (size_t)i * 32;
		rp[i] = rb + (size_t)i * 32;
	}
	get_random(kb, 32);
	for (i = 32; i < n * 32u; i++) {
		kb[i] = (u8)(kb[i - 32] * 5 + i);
	}
	for (best = 1e30, rep = 0; rep < 3; rep++) {
		t0 = now_s();
		r |= ec_sign_batch(sigs, 64, kpp, msgs, msglens, n, NULL, ECDSA, SHA256, NULL, NULL, rets);
		if (rep && now_s() - t0 < best) { best = now_s() - t0; }
	}
	printf("bench ec_sign_batch ECDSA/SECP256R1/SHA256 n = %u: rc %d, %.1f ms, %.2f M signatures/s (nn_get_random_mod nonces, hashing, marshalling included)\n", n, r, best * 1e3, n / best / 1e6);
	for (best = 1e30, rep = 0; rep < 3; rep++) {
		t0 = now_s();
		r |= ec_sign_batch(sigs, 64, kpp, msgs, msglens, n, NULL, DECDSA, SHA256, NULL, NULL, rets);
		if (rep && now_s() - t0 < best) { best = now_s() - t0; }
	}
	printf("bench ec_sign_batch DECDSA (key type mismatch: every item fails in libecc's checks) n = %u: %.1f ms\n", n, best * 1e3);
	for (best = 1e30, rep = 0; rep < 3; rep++) {
		t0 = now_s();
		r |= ec_key_pair_gen_batch(gen, &params, ECDSA, n, rets);
		if (rep && now_s() - t0 < best) { best = now_s() - t0; }
	}
	printf("bench ec_key_pair_gen_batch ECDSA/SECP256R1 n = %u: rc %d, %.1f ms, %.2f M key pairs/s\n", n, r, best * 1e3, n / best / 1e6);
	for (best = 1e30, rep = 0; rep < 3; rep++) {
		t0 = now_s();
		r |= x25519_init_pub_key_batch(kp, rp, n, rets);
		if (rep && now_s() - t0 < best) { best = now_s() - t0; }
	}
	printf("bench x25519_init_pub_key_batch n = %u: rc %d, %.1f ms, %.2f M/s\n", n, r, best * 1e3, n / best / 1e6);
	free(kpp); free(msgs); free(sigs); free(sigbuf); free(kb); free(rb); free(kp); free(rp); free(msglens); free(rets); free(gen);
}

/* ------------------------------------------------------------------------------------------------
 * benchj <log2 n>: the typed boundary as ONE JSON object (bench.py's "typed_boundary" record) -- end-to-end rates of the libecc-typed batch
 * entry points on 2^n items, each beside libecc's OWN function for the same job timed in this process on every host thread the
 * library uses (a loop of ec_verify / ec_sign over a bounded sample of the same structures; for the algorithms libecc verifies in
 * batches also its ec_verify_batch on pieces of 256 items per thread).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
	const u8 **sigs, **msgs;
	const u8 *siglens;
	const u32 *msglens;
	const ec_pub_key **pubs;
	const ec_key_pair **kps;
	ec_alg_type sig_type;
	hash_alg_type hash_type;
	u32 lo, hi;
	int mode;        /* 0: loop of ec_verify, 1: libecc's ec_verify_batch on pieces of 256, 2: loop of ec_sign */
	int bad;
} cpu_job;
static void *cpu_worker(void *arg)
{
	cpu_job *J = (cpu_job *)arg;
	u32 i;
	if (J->mode == 1) {
		for (i = J->lo; i < J->hi; i += 256) {
			const u32 m = (J->hi - i) < 256 ? (J->hi - i) : 256;
			u16 adl[256];
			const u8 *ad[256];
			memset(adl, 0, sizeof(adl));
			memset(ad, 0, sizeof(ad));
			J->bad |= libecc_cpu_ec_verify_batch(J->sigs + i, J->siglens + i, J->pubs + i, J->msgs + i, J->msglens + i, m, J->sig_type,
							     J->hash_type, ad, adl, NULL, NULL) ? 1 : 0;
		}
		return NULL;
	}
	for (i = J->lo; i < J->hi; i++) {
		if (J->mode == 0) {
			J->bad |= ec_verify(J->sigs[i], J->siglens[i], J->pubs[i], J->msgs[i], J->msglens[i], J->sig_type, J->hash_type, NULL, 0) ? 1 : 0;
		} else {
			u8 sg[2 * 72];
			J->bad |= ec_sign(sg, J->siglens[i], J->kps[i], J->msgs[i], J->msglens[i], J->sig_type, J->hash_type, NULL, 0) ? 1 : 0;
		}
	}
	return NULL;
}
static int host_threads(void)
{
	int n = (int)sysconf(_SC_NPROCESSORS_ONLN);
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		char quota[32];
		long period = 0;
		if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") && period > 0) {
			const int c = (int)((atol(quota) + period - 1) / period);
			if (c >= 1 && c < n) { n = c; }
		}
		fclose(f);
	}
	if (getenv("ECAMD_COMPAT_THREADS") && atoi(getenv("ECAMD_COMPAT_THREADS")) > 0) {
		n = atoi(getenv("ECAMD_COMPAT_THREADS"));
	}
	return n < 1 ? 1 : (n > 256 ? 256 : n);
}
/* items/s of libecc's own function over the first `sample` items on every host thread; prints the "cpu" member */
static void cpu_rate(const char *what, cpu_job *proto, u32 sample, int mode)
{
	pthread_t th[256];
	cpu_job J[256];
	const int T = host_threads();
	int t, bad = 0;
	const double t0 = now_s();
	double el;
	for (t = 0; t < T; t++) {
		J[t] = *proto;
		J[t].mode = mode;
		J[t].bad = 0;
		J[t].lo = (u32)((u64)sample * (u64)t / (u64)T);
		J[t].hi = (u32)((u64)sample * (u64)(t + 1) / (u64)T);
		if (mode == 1) {   /* whole pieces */
			J[t].lo = (J[t].lo / 256) * 256;
			J[t].hi = (t + 1 == T) ? sample : (J[t].hi / 256) * 256;
		}
		pthread_create(&th[t], NULL, cpu_worker, &J[t]);
	}
	for (t = 0; t < T; t++) {
		pthread_join(th[t], NULL);
		bad |= J[t].bad;
	}
	el = now_s() - t0;
	printf("\"%s\": {\"what\": \"%s\", \"items\": %u, \"threads\": %d, \"seconds\": %.3f, \"rate\": %.1f, \"all_accepted\": %s}",
	       mode == 1 ? "cpu_batch" : "cpu", what, sample, T, el, (double)sample / el, bad ? "false" : "true");
}

static double best_of(int (*fn)(void *), void *arg, int reps, int *rc)
{
	double best = 1e30;
	int rep;
	for (rep = 0; rep < reps; rep++) {
		const double t0 = now_s();
		*rc |= fn(arg);
		if (rep && now_s() - t0 < best) {
			best = now_s() - t0;
		}
	}
	return best;
}
typedef struct {
	const u8 **sigs, **msgs, **adatas;
	u8 **sigw;
	const u8 *siglens;
	const u32 *msglens;
	const u16 *adlens;
	const ec_pub_key **pubs;
	const ec_key_pair **kps;
	ec_alg_type sig_type;
	hash_alg_type hash_type;
	u32 n;
	u8 siglen;
	int *rets;
} typed_call;
static int call_verify(void *a)
{
	typed_call *c = (typed_call *)a;
	return ec_verify_batch(c->sigs, c->siglens, c->pubs, c->msgs, c->msglens, c->n, c->sig_type, c->hash_type, c->adatas, c->adlens, NULL, NULL);
}
static int call_sign(void *a)
{
	typed_call *c = (typed_call *)a;
	return ec_sign_batch(c->sigw, c->siglen, c->kps, c->msgs, c->msglens, c->n, NULL, c->sig_type, c->hash_type, NULL, NULL, c->rets);
}

/* n DISTINCT valid BIP0340 signatures without libecc's one-at-a-time signer (86 s per 2^20 on 16 threads): keys from
 * ec_key_pair_gen_batch, the nonce points R = [k]G from prj_pt_mul_batch, then sig/bip0340.c:135-330 item by item on libecc's own
 * arithmetic -- d = x or q - x for an even P.y, k negated for an even R.y, e = H(H(tag) || H(tag) || R.x || P.x || m) mod q,
 * s = k + e d mod q.  A sample is checked with libecc's ec_verify by the caller. */
typedef struct {
	ec_key_pair *kps;
	nn *ks;
	prj_pt *Rs;
	u8 *sigbuf, *msgbuf;
	const ec_params *params;
	u32 lo, hi, ml;
	int bad;
} bipsign_job;
static void *bipsign_worker(void *arg)
{
	bipsign_job *J = (bipsign_job *)arg;
	const ec_params *P = J->params;
	const u32 cl = (u32)BYTECEIL(P->ec_fp.p_bitlen), ql = (u32)BYTECEIL(P->ec_gen_order_bitlen);
	u8 tagd[SHA256_DIGEST_SIZE], dig[SHA256_DIGEST_SIZE], buf[2 * 72];
	sha256_context hc;
	u32 i;
	if (sha256_init(&hc) || sha256_update(&hc, (const u8 *)"BIP0340/challenge", 17) || sha256_final(&hc, tagd)) {
		J->bad = 1;
		return NULL;
	}
	for (i = J->lo; i < J->hi; i++) {
		aff_pt Ra, Pa;
		nn d, k, e, s;
		int odd = 0;
		u8 *sig = J->sigbuf + (size_t)i * (cl + ql);
		d.magic = k.magic = e.magic = s.magic = WORD(0);
		if (prj_pt_to_aff(&Ra, &J->Rs[i]) || prj_pt_to_aff(&Pa, &J->kps[i].pub_key.y) || nn_copy(&d, &J->kps[i].priv_key.x) || nn_copy(&k, &J->ks[i])) {
			J->bad = 1;
			continue;
		}
		if (nn_isodd(&(Pa.y.fp_val), &odd) || (odd && nn_mod_neg(&d, &d, &(P->ec_gen_order)))) { J->bad = 1; }
		if (nn_isodd(&(Ra.y.fp_val), &odd) || (odd && nn_mod_neg(&k, &k, &(P->ec_gen_order)))) { J->bad = 1; }
		if (fp_export_to_buf(sig, (u16)cl, &Ra.x) || fp_export_to_buf(buf, (u16)cl, &Pa.x) || sha256_init(&hc) ||
		    sha256_update(&hc, tagd, sizeof(tagd)) || sha256_update(&hc, tagd, sizeof(tagd)) || sha256_update(&hc, sig, cl) ||
		    sha256_update(&hc, buf, cl) || sha256_update(&hc, J->msgbuf + (size_t)i * J->ml, J->ml) || sha256_final(&hc, dig) ||
		    nn_init_from_buf(&e, dig, sizeof(dig)) || nn_mod(&e, &e, &(P->ec_gen_order)) || nn_mod_mul(&s, &e, &d, &(P->ec_gen_order)) ||
		    nn_mod_add(&s, &s, &k, &(P->ec_gen_order)) || nn_export_to_buf(sig + cl, (u16)ql, &s)) {
			J->bad = 1;
		}
		nn_uninit(&d); nn_uninit(&k); nn_uninit(&e); nn_uninit(&s);
	}
	return NULL;
}
static int bip0340_sign_distinct(const ec_params *params, ec_key_pair *kps, u8 *sigbuf, u8 *msgbuf, u32 ml, u32 n)
{
	nn *ks = calloc(n, sizeof(nn));
	prj_pt *Rs = calloc(n, sizeof(prj_pt)), *Gs = calloc(n, sizeof(prj_pt));
	int *rets = calloc(n, sizeof(int)), bad = 0, t;
	const int T = host_threads();
	pthread_t th[256];
	bipsign_job J[256];
	u32 i;
	if (!ks || !Rs || !Gs || !rets || ec_key_pair_gen_batch(kps, params, BIP0340, n, rets)) {
		return -1;
	}
	for (i = 0; i < n && !bad; i++) {
		/* (nonces: private scalars of a second batch of key pairs would do as well; here 32 random octets reduced by libecc) */
		u8 rb[40];
		bad = rets[i] || get_random(rb, sizeof(rb)) || nn_init_from_buf(&ks[i], rb, sizeof(rb)) || nn_mod(&ks[i], &ks[i], &(params->ec_gen_order)) ||
		      prj_pt_copy(&Gs[i], &(params->ec_gen));
	}
	bad = bad || prj_pt_mul_batch(Rs, ks, Gs, n, rets);
	for (i = 0; i < n && !bad; i++) {
		bad = rets[i];
	}
	for (t = 0; t < T && !bad; t++) {
		J[t].kps = kps; J[t].ks = ks; J[t].Rs = Rs; J[t].sigbuf = sigbuf; J[t].msgbuf = msgbuf; J[t].params = params; J[t].ml = ml; J[t].bad = 0;
		J[t].lo = (u32)((u64)n * (u64)t / (u64)T);
		J[t].hi = (u32)((u64)n * (u64)(t + 1) / (u64)T);
		pthread_create(&th[t], NULL, bipsign_worker, &J[t]);
	}
	for (t = 0; t < T && !bad; t++) {
		pthread_join(th[t], NULL);
	}
	for (t = 0; t < T; t++) {
		bad |= J[t].bad;
	}
	free(ks); free(Rs); free(Gs); free(rets);
	return bad ? -1 : 0;
}

static void benchj_family(const char *curve, ec_alg_type sig_type, hash_alg_type hash_type, const char *label, u32 n, int distinct, int with_sign,
			  int cpu_batch)
{
	enum { base = 512, ML = 48 };
	ec_params params;
	const u32 nk = distinct ? n : base;
	ec_key_pair *kps = calloc(nk, sizeof(ec_key_pair));
	u8 *sigbuf, *msgbuf = calloc(nk, ML), siglen = 0;
	typed_call C;
	cpu_job proto;
	u32 i, sample;
	int rc = 0;
	double el;
	memset(&C, 0, sizeof(C));
	memset(&proto, 0, sizeof(proto));
	if (!kps || load_params(curve, &params) || ec_get_sig_len(&params, sig_type, hash_type, &siglen)) {
		printf("{\"call\": \"%s\", \"error\": \"setup\"}", label);
		return;
	}
	sigbuf = calloc(nk, siglen);
	C.pubs = calloc(n, sizeof(*C.pubs)); C.kps = calloc(n, sizeof(*C.kps));
	C.sigs = calloc(n, sizeof(*C.sigs)); C.sigw = calloc(n, sizeof(*C.sigw)); C.msgs = calloc(n, sizeof(*C.msgs)); C.adatas = calloc(n, sizeof(*C.adatas));
	C.siglens = calloc(n, 1); C.msglens = calloc(n, sizeof(u32)); C.adlens = calloc(n, sizeof(u16)); C.rets = calloc(n, sizeof(int));
	C.sig_type = sig_type; C.hash_type = hash_type; C.n = n; C.siglen = siglen;
	if (get_random(msgbuf, ML)) { rc = 1; }
	for (i = 1; i < nk; i++) {
		memcpy(msgbuf + (size_t)i * ML, msgbuf, ML);
		memcpy(msgbuf + (size_t)i * ML, &i, sizeof(i));
	}
	if (distinct) {
		rc |= bip0340_sign_distinct(&params, kps, sigbuf, msgbuf, ML, n);
	} else {
		for (i = 0; i < base && !rc; i++) {
			rc = ec_key_pair_gen(&kps[i], &params, sig_type) || ec_sign(sigbuf + (size_t)i * siglen, siglen, &kps[i], msgbuf + (size_t)i * ML, ML, sig_type, hash_type, NULL, 0);
		}
	}
	for (i = 0; i < n; i++) {
		C.pubs[i] = &kps[i % nk].pub_key;
		C.kps[i] = &kps[i % nk];
		C.sigs[i] = sigbuf + (size_t)(i % nk) * siglen;
		C.msgs[i] = msgbuf + (size_t)(i % nk) * ML;
		((u8 *)C.siglens)[i] = siglen;
		((u32 *)C.msglens)[i] = ML;
	}
	if (rc) {
		printf("{\"call\": \"ec_verify_batch %s\", \"error\": \"signing\"}", label);
		return;
	}
	{
		const unsigned long calls0 = ecamd_compat_schnorr_msm_calls();
		el = best_of(call_verify, &C, 4, &rc);
		printf("{\"call\": \"ec_verify_batch %s\", \"n\": %u, \"ms\": %.3f, \"rate\": %.1f, \"accepted\": %s, \"distinct_items\": %u, \"multi_scalar_calls\": %lu, ",
		       label, n, el * 1e3, (double)n / el, rc ? "false" : "true", nk, ecamd_compat_schnorr_msm_calls() - calls0);
	}
	proto.sigs = C.sigs; proto.msgs = C.msgs; proto.siglens = C.siglens; proto.msglens = C.msglens; proto.pubs = C.pubs; proto.kps = C.kps;
	proto.sig_type = sig_type; proto.hash_type = hash_type;
	sample = (u32)(4096 * (host_threads() >= 8 ? 2 : 1));
	sample = sample < n ? sample : n;
	cpu_rate("a loop of libecc's ec_verify over the first items of the same arrays", &proto, sample, 0);
	if (cpu_batch) {
		printf(", ");
		cpu_rate("libecc's ec_verify_batch (no scratch pad) on pieces of 256 of the same items", &proto, sample, 1);
	}
	printf("}");
	if (with_sign) {
		u8 *out = calloc(n, siglen);
		for (i = 0; i < n; i++) {
			C.sigw[i] = out + (size_t)i * siglen;
		}
		rc = 0;
		el = best_of(call_sign, &C, 3, &rc);
		printf(",\n {\"call\": \"ec_sign_batch %s\", \"n\": %u, \"ms\": %.3f, \"rate\": %.1f, \"rc\": %d, ", label, n, el * 1e3, (double)n / el, rc);
		cpu_rate("a loop of libecc's ec_sign over the first items of the same arrays", &proto, sample, 2);
		printf("}");
		free(out);
	}
	free(kps); free(sigbuf); free(msgbuf); free(C.pubs); free(C.kps); free(C.sigs); free(C.sigw); free(C.msgs); free(C.adatas);
	free((void *)C.siglens); free((void *)C.msglens); free((void *)C.adlens); free(C.rets);
}

/* two application threads inside ec_verify_batch at once (libecc is re-entrant): wall time of two concurrent calls of n / 2
 * items against the same two calls one after the other */
static void *conc_worker(void *a)
{
	typed_call *c = (typed_call *)a;
	c->rets[0] = call_verify(c);
	return NULL;
}
static void benchj_concurrent(u32 n)
{
	enum { base = 512, ML = 48 };
	ec_params params;
	static ec_key_pair kps[base];
	static u8 sigbuf[base][64], msgbuf[base][ML];
	typed_call C[2];
	pthread_t th[2];
	u32 i, h = n / 2;
	int rets[2] = {0, 0}, k, rep, rc = 0;
	double t_seq = 1e30, t_par = 1e30, t0;
	if (load_params("SECP256R1", &params)) {
		return;
	}
	for (i = 0; i < base; i++) {
		rc |= ec_key_pair_gen(&kps[i], &params, ECDSA) || get_random(msgbuf[i], ML) || ec_sign(sigbuf[i], 64, &kps[i], msgbuf[i], ML, ECDSA, SHA256, NULL, 0);
	}
	for (k = 0; k < 2; k++) {
		memset(&C[k], 0, sizeof(C[k]));
		C[k].pubs = calloc(h, sizeof(*C[k].pubs)); C[k].sigs = calloc(h, sizeof(*C[k].sigs)); C[k].msgs = calloc(h, sizeof(*C[k].msgs));
		C[k].adatas = calloc(h, sizeof(*C[k].adatas)); C[k].siglens = calloc(h, 1); C[k].msglens = calloc(h, sizeof(u32)); C[k].adlens = calloc(h, sizeof(u16));
		C[k].rets = &rets[k]; C[k].sig_type = ECDSA; C[k].hash_type = SHA256; C[k].n = h;
		for (i = 0; i < h; i++) {
			C[k].pubs[i] = &kps[(i + 7 * k) % base].pub_key; C[k].sigs[i] = sigbuf[(i + 7 * k) % base]; C[k].msgs[i] = msgbuf[(i + 7 * k) % base];
			((u8 *)C[k].siglens)[i] = 64; ((u32 *)C[k].msglens)[i] = ML;
		}
	}
	for (rep = 0; rep < 4; rep++) {
		t0 = now_s();
		rc |= call_verify(&C[0]) | call_verify(&C[1]);
		if (rep && now_s() - t0 < t_seq) { t_seq = now_s() - t0; }
	}
	for (rep = 0; rep < 4; rep++) {
		t0 = now_s();
		pthread_create(&th[0], NULL, conc_worker, &C[0]);
		pthread_create(&th[1], NULL, conc_worker, &C[1]);
		pthread_join(th[0], NULL);
		pthread_join(th[1], NULL);
		rc |= rets[0] | rets[1];
		if (rep && now_s() - t0 < t_par) { t_par = now_s() - t0; }
	}
	printf("{\"call\": \"two application threads, ec_verify_batch ECDSA/SECP256R1/SHA256 of %u items each\", \"one_after_the_other_ms\": %.3f, "
	       "\"at_once_ms\": %.3f, \"overlap_gain\": %.3f, \"accepted\": %s}", h, t_seq * 1e3, t_par * 1e3, t_seq / t_par, rc ? "false" : "true");
}

static void benchj(u32 n)
{
	printf("{\"items\": %u, \"host_threads\": %d, \"records\": [\n ", n, host_threads());
	benchj_family("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", n, 0, 1, 0);
	printf(",\n ");
	benchj_family("WEI25519", EDDSA25519, SHA512, "EDDSA25519", n, 0, 0, 1);
	printf(",\n ");
	benchj_family("SECP256K1", BIP0340, SHA256, "BIP0340/SECP256K1/SHA256", n, 1, 0, 1);
	printf(",\n ");
	benchj_concurrent(n);
	printf("\n]}\n");
}

int main(int argc, char **argv)
{
	const u32 n = (argc > 1) ? (u32)atoi(argv[1]) : 256;
	if (argc > 2 && !strcmp(argv[1], "benchv")) {
		/* only the ECDSA secp256r1 (or, with a third argument "384", secp384r1) verification of `bench` (for profiling runs) */
		const u32 bn = 1u << (u32)atoi(argv[2]);
		if (ecamd_compat_init(NULL, 0, 0)) {
			printf("no GPU path\n");
			return 3;
		}
		if (argc > 3 && !strcmp(argv[3], "384")) {
			bench_verify("SECP384R1", ECDSA, SHA384, "ECDSA/SECP384R1/SHA384", bn);
		} else if (argc > 3 && !strcmp(argv[3], "ed25519")) {
			/* benchj's EDDSA25519 record alone: ec_verify_batch of 2^n signatures, the batch bit (profiling runs) */
			printf("{\"records\": [\n");
			benchj_family("WEI25519", EDDSA25519, SHA512, "EDDSA25519", bn, 0, 0, 1);
			printf("]}\n");
		} else if (argc > 3 && !strcmp(argv[3], "bip0340")) {
			printf("{\"records\": [\n");
			benchj_family("SECP256K1", BIP0340, SHA256, "BIP0340/SECP256K1/SHA256", bn, 1, 0, 1);
			printf("]}\n");
		} else {
			bench_verify("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", bn);
		}
		ecamd_compat_shutdown();
		return 0;
	}
	if (argc > 2 && !strcmp(argv[1], "benchj")) {
		const u32 bn = 1u << (u32)atoi(argv[2]);
		if (ecamd_compat_init(NULL, 0, 0)) {
			printf("no GPU path\n");
			return 3;
		}
		ecamd_compat_set_concurrent_random(1);   /* this application's get_random is thread-safe (per-thread pools) */
		g_rand_expect_serial = 0;
		benchj(bn);
		ecamd_compat_shutdown();
		return 0;
	}
	if (argc > 2 && !strcmp(argv[1], "bench_schnorr")) {
		const u32 bn = 1u << (u32)atoi(argv[2]);
		if (ecamd_compat_init(NULL, 0, 0)) {
			printf("no GPU path\n");
			return 3;
		}
		ecamd_compat_set_concurrent_random(1);
		g_rand_expect_serial = 0;
		bench_schnorr("SECP256K1", BIP0340, SHA256, "BIP0340/SECP256K1/SHA256", bn);
		bench_schnorr("SECP256R1", ECFSDSA, SHA256, "ECFSDSA/SECP256R1/SHA256", bn);
		ecamd_compat_shutdown();
		return 0;
	}
	if (argc > 2 && !strcmp(argv[1], "bench")) {
		const u32 bn = 1u << (u32)atoi(argv[2]);
		if (ecamd_compat_init(NULL, 0, 0)) {
			printf("no GPU path\n");
			return 3;
		}
		ecamd_compat_set_concurrent_random(1);   /* this application's get_random is thread-safe (per-thread pools) */
		g_rand_expect_serial = 0;
		bench_verify("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", bn);
		bench_verify("SECP384R1", ECDSA, SHA384, "ECDSA/SECP384R1/SHA384", bn);
		bench_verify("WEI25519", EDDSA25519, SHA512, "EDDSA25519", bn);
		bench_secret_half(bn);
		ecamd_compat_shutdown();
		return 0;
	}
	if (ecamd_compat_init(NULL, 0, 0)) {
		printf("no GPU path\n");
		return 3;
	}
	if (argc > 1 && !strcmp(argv[1], "cfg1")) {
		/* BASELINE.json configs[0] / SURVEY.md 8d-1, the plumbing case: 1024 secp256r1 scalar multiplications (scalars uniform in
		 * [1, q - 1], bases [t]G and G) through the library API -- every item of prj_pt_mul_batch against a loop of libecc's own
		 * prj_pt_mul on the same structures */
		check_mul("SECP256R1", 1024, 0);
		ecamd_compat_shutdown();
		printf(failures ? "compat_check cfg1: %d FAILURES\n" : "compat_check cfg1: all ok (%d failures)\n", failures);
		return failures ? 1 : 0;
	}
	if (argc > 2 && !strcmp(argv[1], "quick")) {
		/* one case per entry-point family at a size that goes through the thread pool and several pipeline chunks */
		const u32 qn = (u32)atoi(argv[2]);
		check_mul("SECP256R1", qn, 0);
		check_group_law("WEI25519", qn < 96 ? qn : 96);
		check_cdh("SECP256R1", qn);
		check_verify("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", qn, 1);
		check_verify("WEI25519", EDDSA25519, SHA512, "EDDSA25519", qn, 1);
		check_verify("WEI25519", EDDSA25519CTX, SHA512, "EDDSA25519CTX", qn < 128 ? qn : 128, 1);
		check_verify("WEI25519", EDDSA25519PH, SHA512, "EDDSA25519PH", qn < 128 ? qn : 128, 1);
		check_verify("WEI448", EDDSA448, SHAKE256, "EDDSA448", qn < 48 ? qn : 48, 1);
		check_verify("WEI448", EDDSA448PH, SHAKE256, "EDDSA448PH", qn < 48 ? qn : 48, 1);
		check_sign("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", qn, 2);
		check_sign("WEI25519", EDDSA25519, SHA512, "EDDSA25519", qn, 1);
		check_keys("SECP256R1", ECDSA, "ECDSA/SECP256R1", qn);
		check_xdh(32, qn);
		check_verify("SECP256K1", BIP0340, SHA256, "BIP0340/SECP256K1/SHA256", qn < 160 ? qn : 160, 1);
		check_verify("SECP256R1", ECFSDSA, SHA256, "ECFSDSA/SECP256R1/SHA256", qn < 96 ? qn : 96, 1);
		check_verify("BRAINPOOLP384R1", ECFSDSA, SHA384, "ECFSDSA/BRAINPOOLP384R1/SHA384", qn < 32 ? qn : 32, 1);
		check_torsion_shift(8);
		check_concurrent_verify(qn);
		printf("schnorr multi-scalar calls: %lu\n", ecamd_compat_schnorr_msm_calls());
		printf("ed25519 whole-batch calls: %lu\n", ecamd_compat_ed_msm_calls());
		ecamd_compat_shutdown();
		CHECK(!g_rand_expect_serial || g_rand_overlaps == 0, "get_random entered concurrently %d times", g_rand_overlaps);
		printf(failures ? "compat_check: %d FAILURES\n" : "compat_check: all ok (%d failures)\n", failures);
		return failures ? 1 : 0;
	}
	check_mul("SECP256R1", n, 0);
	check_mul("SECP384R1", n, 0);
	check_mul("SECP521R1", n, 0);
	check_mul("BRAINPOOLP256R1", n, 0);
	check_mul("WEI25519", n, 0);
	check_mul("SECP256R1", n, 1);
	check_group_law("SECP256R1", n);
	check_group_law("WEI25519", n < 256 ? n : 256);
	check_group_law("SECP384R1", n < 128 ? n : 128);
	check_mul("SECP384R1", n < 128 ? n : 128, 1);
	check_mul("WEI25519", n < 128 ? n : 128, 1);
	check_cdh("SECP256R1", n);
	check_cdh("SECP384R1", n);
	check_cdh("WEI25519", n);
	check_verify("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", n, 1);
	check_verify("SECP256R1", ECDSA, SHA512, "ECDSA/SECP256R1/SHA512", n, 1);
	check_verify("SECP384R1", ECDSA, SHA384, "ECDSA/SECP384R1/SHA384", n, 1);
	check_verify("SECP521R1", ECDSA, SHA3_512, "ECDSA/SECP521R1/SHA3_512", n, 1);
	check_verify("BRAINPOOLP256R1", DECDSA, SHA256, "DECDSA/BRAINPOOLP256R1/SHA256", n, 1);
	check_verify("WEI25519", EDDSA25519, SHA512, "EDDSA25519", n, 1);
	check_verify("WEI25519", EDDSA25519CTX, SHA512, "EDDSA25519CTX", n, 1);
	check_verify("WEI25519", EDDSA25519PH, SHA512, "EDDSA25519PH", n, 1);
	check_verify("WEI448", EDDSA448, SHAKE256, "EDDSA448", n, 1);
	check_verify("WEI448", EDDSA448PH, SHAKE256, "EDDSA448PH", n, 1);
	if (!getenv("COMPAT_CHECK_MOCK")) {   /* (the CPU stand-in of tests/ does not model a key at infinity) */
		check_inf_key("SECP256R1", SHA256, n < 48 ? n : 48);
		check_inf_key("SECP384R1", SHA512, n < 48 ? n : 48);
	}
	/* the secret-key half: signing, key pairs, X25519 / X448 */
	check_sign("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", n, 2);
	check_sign("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", n, 0);
	check_sign("SECP384R1", ECDSA, SHA384, "ECDSA/SECP384R1/SHA384", n, 2);
	check_sign("SECP521R1", ECDSA, SHA512, "ECDSA/SECP521R1/SHA512", n < 256 ? n : 256, 2);
	check_sign("SECP256R1", DECDSA, SHA256, "DECDSA/SECP256R1/SHA256 (RFC 6979)", n, 1);
	check_sign("BRAINPOOLP256R1", DECDSA, SHA512, "DECDSA/BRAINPOOLP256R1/SHA512", n < 256 ? n : 256, 1);
	check_sign("SECP224R1", DECDSA, SHA256, "DECDSA/SECP224R1/SHA256", n < 256 ? n : 256, 1);
	check_sign("WEI25519", EDDSA25519, SHA512, "EDDSA25519", n, 1);
	check_sign("WEI25519", EDDSA25519CTX, SHA512, "EDDSA25519CTX", n < 256 ? n : 256, 1);
	check_sign("WEI25519", EDDSA25519PH, SHA512, "EDDSA25519PH", n < 256 ? n : 256, 1);
	if (!getenv("COMPAT_CHECK_MOCK")) {
		check_sign("WEI448", EDDSA448, SHAKE256, "EDDSA448", n < 256 ? n : 256, 1);
		check_sign("WEI448", EDDSA448PH, SHAKE256, "EDDSA448PH", n < 128 ? n : 128, 1);
	}
	check_sign("SECP256R1", ECSDSA, SHA256, "ECSDSA (libecc's CPU path)", n < 16 ? n : 16, 0);
	check_keys("SECP256R1", ECDSA, "ECDSA/SECP256R1", n);
	check_keys("SECP384R1", ECDSA, "ECDSA/SECP384R1", n < 256 ? n : 256);
	check_keys("SECP256R1", ECKCDSA, "ECKCDSA/SECP256R1 (x^-1)", n < 128 ? n : 128);
	check_keys("SECP256R1", SM2, "SM2/SECP256R1 (x < q - 1)", n < 128 ? n : 128);
	check_keys("SECP256K1", BIP0340, "BIP0340/SECP256K1 (any x)", n < 128 ? n : 128);
	check_keys("SECP256R1", ECCCDH, "ECCCDH/SECP256R1", n < 256 ? n : 256);
	check_keys("WEI25519", EDDSA25519, "EDDSA25519", n < 256 ? n : 256);
	check_keys("WEI448", EDDSA448, "EDDSA448", n < 128 ? n : 128);
	check_eddsa_import(n < 256 ? n : 256);
	check_xdh(32, n);
	check_xdh(56, n < 256 ? n : 256);
	check_foreign_generator(n < 64 ? n : 64);
	check_verify("SECP256K1", BIP0340, SHA256, "BIP0340/SECP256K1/SHA256", n, 1);
	check_verify("SECP256R1", BIP0340, SHA512, "BIP0340/SECP256R1/SHA512", n < 128 ? n : 128, 1);
	check_verify("SECP256R1", ECFSDSA, SHA256, "ECFSDSA/SECP256R1/SHA256", n, 1);
	check_verify("BRAINPOOLP384R1", ECFSDSA, SHA384, "ECFSDSA/BRAINPOOLP384R1/SHA384", n < 128 ? n : 128, 1);
	check_torsion_shift(8);
	check_concurrent_verify(n);
	/* an algorithm the GPU does not take goes to libecc's own verifier (no batch form there: -1) */
	check_verify("SECP256R1", ECKCDSA, SHA256, "ECKCDSA (no batch form)", n < 16 ? n : 16, -1);
	printf("items sent to the GPU: %llu\n", ecamd_compat_gpu_items());
	printf("schnorr multi-scalar calls: %lu\n", ecamd_compat_schnorr_msm_calls());
	if (!ecamd_compat_gpu_items()) {
		failures++;
	}
	ecamd_compat_shutdown();
	CHECK(!g_rand_expect_serial || g_rand_overlaps == 0, "get_random entered concurrently %d times", g_rand_overlaps);
	printf(failures ? "compat_check: %d FAILURES\n" : "compat_check: all ok (%d failures)\n", failures);
	return failures ? 1 : 0;
}
