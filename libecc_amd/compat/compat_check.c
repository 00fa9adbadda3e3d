/*
 * libecc_amd/compat/compat_check.c -- check program of the libecc-typed boundary (include/libecc_amd_compat.h).
 *
 * Written as a libecc application: it holds ec_params, ec_key_pair, prj_pt, nn and calls the batch entry points of
 * libsign_amd.so with the same array shapes tests/ec_self_tests_core.c uses for ec_verify_batch (:373-383, :556-616);
 * every batch result is compared with libecc's own scalar function (prj_pt_mul, ecccdh_derive_secret, ec_verify -- the
 * CPU code of the very libecc the library was linked from) on the same inputs.  Needs an MI355X.
 *   usage: compat_check [items per case, default 256]
 * Exit status 0 iff everything matched; prints one line per case.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "libecc_amd_compat.h"
#include "external_deps/rand.h"   /* get_random: supplied by the application, as for libecc's own libsign */

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

/* ---- "compat_check bench <log2 n>": end-to-end rate of ec_verify_batch as a libecc application sees it -- libecc structures
 * in, one int out, the marshalling and the message hashing on the host threads included.  `base` distinct (key, message,
 * signature) triples made with libecc's ec_sign, repeated to n pointers (the batch arrays are arrays of pointers). ---- */
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

int main(int argc, char **argv)
{
	const u32 n = (argc > 1) ? (u32)atoi(argv[1]) : 256;
	if (argc > 2 && !strcmp(argv[1], "bench")) {
		const u32 bn = 1u << (u32)atoi(argv[2]);
		if (ecamd_compat_init(NULL, 0, 0)) {
			printf("no GPU path\n");
			return 3;
		}
		bench_verify("SECP256R1", ECDSA, SHA256, "ECDSA/SECP256R1/SHA256", bn);
		bench_verify("SECP384R1", ECDSA, SHA384, "ECDSA/SECP384R1/SHA384", bn);
		bench_verify("WEI25519", EDDSA25519, SHA512, "EDDSA25519", bn);
		ecamd_compat_shutdown();
		return 0;
	}
	if (ecamd_compat_init(NULL, 0, 0)) {
		printf("no GPU path\n");
		return 3;
	}
	check_mul("SECP256R1", n, 0);
	check_mul("SECP384R1", n, 0);
	check_mul("SECP521R1", n, 0);
	check_mul("BRAINPOOLP256R1", n, 0);
	check_mul("WEI25519", n, 0);
	check_mul("SECP256R1", n, 1);
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
	check_inf_key("SECP256R1", SHA256, n < 48 ? n : 48);
	check_inf_key("SECP384R1", SHA512, n < 48 ? n : 48);
	/* an algorithm the GPU does not take goes to libecc's own verifier */
	check_verify("SECP256K1", BIP0340, SHA256, "BIP0340 (libecc's CPU path)", n < 16 ? n : 16, 0);
	printf("items sent to the GPU: %llu\n", ecamd_compat_gpu_items());
	if (!ecamd_compat_gpu_items()) {
		failures++;
	}
	ecamd_compat_shutdown();
	printf(failures ? "compat_check: %d FAILURES\n" : "compat_check: all ok (%d failures)\n", failures);
	return failures ? 1 : 0;
}
