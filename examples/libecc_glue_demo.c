/*
 * examples/libecc_glue_demo.c -- what a libecc application adds to use libecc_amd,
 * as a program: struct-array adaptors written against libecc's OWN types and exporters, then a check that
 * the GPU batch agrees with libecc's scalar prj_pt_mul on the same inputs.
 *
 *   prj_pt_mul_batch(out[], m[], in[], n)   nn[] / prj_pt[] in, prj_pt[] out   (curves/prj_pt.h:76)
 *   ecdsa_verify_batch(...)                 ec_pub_key[] + raw signatures + digests -> accept bits
 *
 * Build (authoring container, where the libecc sources are mounted):
 *   make -C oracle glue_demo          -> oracle/_ref/glue_demo (links libecc_ref.so = unmodified libecc)
 * Run on an MI355X:  oracle/_ref/glue_demo [n]
 * This file contains no libecc code: it only calls its public API (libsig.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libsig.h"
#include "libecc_amd.h"

static ecamd_ctx *g_ctx;
static ecamd_curve *g_crv;

static int glue_init(const ec_params *params)
{
	if (ecamd_ctx_create(&g_ctx, 0)) {
		fprintf(stderr, "libecc_amd: %s\n", ecamd_last_error());
		return -1; /* no GPU: the caller keeps libecc's CPU path */
	}
	return ecamd_curve_by_name(g_ctx, (const char *)params->curve_name, &g_crv);
}

/* out[i] = [m[i]] in[i] for i < n; ok[i] = 0 / -1 exactly as prj_pt_mul + prj_pt_unique would return.
 * Points travel in the projective wire format of prj_pt_export_to_buf, so no CPU-side inversion. */
static int prj_pt_mul_batch(prj_pt *out, int *ok, const nn *m, const prj_pt *in, u32 n, const ec_params *params)
{
	const u32 clen = (u32)ecamd_curve_coord_len(g_crv), slen = (u32)ecamd_curve_order_len(g_crv);
	u8 *sc = malloc((size_t)n * slen), *pin = malloc((size_t)n * 3 * clen), *pout = malloc((size_t)n * 3 * clen), *st = malloc(n);
	u32 i;
	int ret = -1;
	if (!sc || !pin || !pout || !st) {
		goto done;
	}
	for (i = 0; i < n; i++) {
		if (nn_export_to_buf(sc + (size_t)i * slen, (u16)slen, &m[i]) ||
		    prj_pt_export_to_buf(&in[i], pin + (size_t)i * 3 * clen, 3 * clen)) {
			goto done;
		}
	}
	if (ec_prj_pt_mul_batch_fmt(g_ctx, g_crv, n, sc, slen, pin, ECAMD_PT_PROJECTIVE, pout, ECAMD_PT_PROJECTIVE, st)) {
		fprintf(stderr, "libecc_amd: %s\n", ecamd_last_error());
		goto done;
	}
	for (i = 0; i < n; i++) {
		ok[i] = -1;
		if (st[i] == ECAMD_OK) {
			ok[i] = prj_pt_import_from_buf(&out[i], pout + (size_t)i * 3 * clen, (u16)(3 * clen), &params->ec_curve);
		} else if (st[i] == ECAMD_INF) {
			ok[i] = (prj_pt_init(&out[i], &params->ec_curve) || prj_pt_zero(&out[i])) ? -1 : 1; /* 1: the point at infinity */
		}
	}
	ret = 0;
done:
	free(sc); free(pin); free(pout); free(st);
	return ret;
}

/* accept[i] = 0 iff ec_verify(sig_i, pub_i, H(m_i)) would return 0; digests are computed by the caller */
static int ecdsa_verify_batch(int *accept, const ec_pub_key *pubs, const u8 *sigs, const u8 *digests, u32 hlen, u32 n)
{
	const u32 clen = (u32)ecamd_curve_coord_len(g_crv);
	u8 *pk = malloc((size_t)n * 2 * clen), *res = malloc(n);
	u32 i;
	int ret = -1;
	if (!pk || !res) {
		goto done;
	}
	for (i = 0; i < n; i++) {
		if (ec_pub_key_export_to_aff_buf(&pubs[i], pk + (size_t)i * 2 * clen, (u8)(2 * clen))) {
			goto done;
		}
	}
	if (ec_ecdsa_verify_batch(g_ctx, g_crv, n, pk, sigs, digests, hlen, res)) {
		fprintf(stderr, "libecc_amd: %s\n", ecamd_last_error());
		goto done;
	}
	for (i = 0; i < n; i++) {
		accept[i] = res[i] ? -1 : 0;
	}
	ret = 0;
done:
	free(pk); free(res);
	return ret;
}

int main(int argc, char **argv)
{
	const u32 n = (argc > 1) ? (u32)atoi(argv[1]) : 256;
	const char *name = "SECP256R1";
	const ec_str_params *sp = NULL;
	ec_params params;
	prj_pt *in = calloc(n, sizeof(prj_pt)), *out = calloc(n, sizeof(prj_pt));
	nn *m = calloc(n, sizeof(nn));
	int *ok = calloc(n, sizeof(int));
	u32 i, bad = 0;
	if (ec_get_curve_params_by_name((const u8 *)name, (u8)(strlen(name) + 1), &sp) || !sp || import_params(&params, sp)) {
		return 2;
	}
	if (glue_init(&params)) {
		fprintf(stderr, "no GPU path: %s\n", ecamd_last_error());
		return 3;
	}
	/* inputs made by libecc itself: random scalars, points [t]G left in whatever projective form prj_pt_mul returns */
	for (i = 0; i < n; i++) {
		nn t;
		if (nn_get_random_mod(&m[i], &params.ec_gen_order) || nn_get_random_mod(&t, &params.ec_gen_order) ||
		    prj_pt_mul(&in[i], &t, &params.ec_gen)) {
			return 4;
		}
	}
	if (i > 2) {
		(void)nn_zero(&m[1]);                                   /* [0]P = infinity */
		(void)nn_copy(&m[2], &params.ec_gen_order);            /* [q]P = infinity */
	}
	if (prj_pt_mul_batch(out, ok, m, in, n, &params)) {
		return 5;
	}
	for (i = 0; i < n; i++) {
		prj_pt ref;
		int cmp = 1, iszero = 0;
		if (prj_pt_mul(&ref, &m[i], &in[i]) || prj_pt_iszero(&ref, &iszero)) {
			return 6;
		}
		if (iszero) {
			bad += (ok[i] != 1);
		} else {
			bad += (ok[i] != 0) || prj_pt_cmp(&ref, &out[i], &cmp) || (cmp != 0);
		}
	}
	printf("prj_pt_mul_batch: %u items, %u mismatches against libecc's prj_pt_mul\n", n, bad);
	{
		/* ECDSA: keys and signatures by libecc (ec_sign), verification on the GPU with SHA-256 digests by libecc */
		const u32 nv = n < 64 ? n : 64;
		ec_key_pair *kp = calloc(nv, sizeof(ec_key_pair));
		ec_pub_key *pubs = calloc(nv, sizeof(ec_pub_key));
		u8 *sigs = malloc((size_t)nv * 64), *dg = malloc((size_t)nv * 32);
		int *acc = calloc(nv, sizeof(int));
		u32 vbad = 0;
		for (i = 0; i < nv; i++) {
			u8 msg[16];
			const u8 *chunks[2] = {msg, NULL};
			u32 lens[1] = {sizeof(msg)};
			memset(msg, (int)i, sizeof(msg));
			if (ec_key_pair_gen(&kp[i], &params, ECDSA) || ec_sign(sigs + (size_t)i * 64, 64, &kp[i], msg, sizeof(msg), ECDSA, SHA256, NULL, 0) ||
			    sha256_scattered(chunks, lens, dg + (size_t)i * 32)) {
				return 7;
			}
			pubs[i] = kp[i].pub_key;
			if (i % 5 == 4) {
				sigs[(size_t)i * 64 + 7] ^= 0x20; /* forged */
			}
		}
		if (ecdsa_verify_batch(acc, pubs, sigs, dg, 32, nv)) {
			return 8;
		}
		for (i = 0; i < nv; i++) {
			u8 msg[16];
			memset(msg, (int)i, sizeof(msg));
			const int r = ec_verify(sigs + (size_t)i * 64, 64, &pubs[i], msg, sizeof(msg), ECDSA, SHA256, NULL, 0);
			vbad += ((r == 0) != (acc[i] == 0));
		}
		printf("ecdsa_verify_batch: %u items, %u mismatches against libecc's ec_verify\n", nv, vbad);
		bad += vbad;
	}
	ecamd_curve_free(g_crv);
	ecamd_ctx_destroy(g_ctx);
	return bad ? 1 : 0;
}
