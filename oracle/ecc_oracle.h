/*
 * oracle/ecc_oracle.h -- TEST INFRASTRUCTURE ONLY (see ecc_oracle.c).
 */
#ifndef ECC_ORACLE_H
#define ECC_ORACLE_H
#include <stdint.h>

#define ORC_MAXW 24 /* 64-bit limbs: room for q^2 (2*9) plus slack */

typedef struct {
	int n;                 /* limbs of p (wlen of ctx->p in the reference) */
	int pbits;             /* p_bitlen */
	uint64_t p[ORC_MAXW];
	uint64_t mpinv;        /* -p^-1 mod 2^64 */
	uint64_t r[ORC_MAXW];  /* 2^(64n) mod p */
	uint64_t r2[ORC_MAXW]; /* 2^(128n) mod p */
} orc_fp_ctx;

typedef struct {
	orc_fp_ctx fp;         /* field of definition */
	orc_fp_ctx fq;         /* Montgomery context for the generator order q (mod-q algebra) */
	uint64_t a[ORC_MAXW], b[ORC_MAXW];          /* plain */
	uint64_t a_m[ORC_MAXW], b3_m[ORC_MAXW];     /* Montgomery form: a*R, 3b*R */
	uint64_t order[ORC_MAXW]; int order_n;      /* CURVE order #E */
	uint64_t q[ORC_MAXW]; int q_n; int qbits;   /* generator order */
	uint64_t gx[ORC_MAXW], gy[ORC_MAXW];
	int clen;              /* BYTECEIL(pbits) */
	int qlen;              /* BYTECEIL(qbits) */
} orc_curve;

int orc_curve_init(orc_curve *c, const uint8_t *p, int plen, const uint8_t *a, int alen,
		   const uint8_t *b, int blen, const uint8_t *order, int olen,
		   const uint8_t *gx, int gxlen, const uint8_t *gy, int gylen,
		   const uint8_t *q, int qlen);
int orc_sizeof_curve(void);

/* op: 0 mul_monty, 1 add, 2 sub, 3 mul plain, 4 inv */
int orc_fp_op_batch(const orc_curve *c, int op, uint32_t n, const uint64_t *a, const uint64_t *b,
		    uint64_t *out);
int orc_scalar_mult_batch(const orc_curve *c, uint32_t n, const uint8_t *scalars, uint32_t slen,
			  const uint8_t *points, uint8_t *out, uint8_t *status);
int orc_pt_add_batch(const orc_curve *c, uint32_t n, const uint8_t *p1, const uint8_t *p2,
		     uint8_t *out, uint8_t *status, int dbl);
int orc_ecdsa_verify_batch(const orc_curve *c, uint32_t n, const uint8_t *pubs_aff,
			   const uint8_t *sigs, const uint8_t *digests, uint32_t hsize,
			   uint8_t *result);
int orc_ecdsa_sign_batch(const orc_curve *c, uint32_t n, const uint8_t *privs,
			 const uint8_t *nonces, const uint8_t *digests, uint32_t hsize,
			 uint8_t *sigs, uint8_t *status);
/* nn_get_random_mod given its 2 * qlen random bytes per item (little-endian integer mod (q - 1), plus one) */
int orc_random_mod_batch(const orc_curve *c, uint32_t n, const uint8_t *raw, uint8_t *out);
int orc_ecccdh_batch(const orc_curve *c, uint32_t n, const uint8_t *privs, const uint8_t *peers_aff,
		     uint8_t *secrets, uint8_t *status);
int orc_xdh_batch(const orc_curve *c, uint32_t len, uint32_t n, const uint8_t *k, const uint8_t *u,
		  uint8_t *out, uint8_t *status);
int orc_eddsa25519_verify_batch(const orc_curve *c, uint32_t n, const uint8_t *pubs, const uint8_t *sigs,
				const uint8_t *hram, uint32_t hlen, uint8_t *result);
int orc_eddsa25519_sign_R_batch(const orc_curve *c, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc, uint8_t *status);
int orc_eddsa25519_sign_S_batch(const orc_curve *c, uint32_t n, const uint8_t *r_hash, const uint8_t *hram,
				const uint8_t *a_scalars, uint8_t *S_out);
int orc_prj_batch(const orc_curve *c, uint32_t n, const uint8_t *scalars, uint32_t slen, const uint8_t *points,
		  uint8_t *out, uint8_t *status);
int orc_eddsa448_verify_batch(const orc_curve *c, uint32_t n, const uint8_t *pubs, const uint8_t *sigs,
			      const uint8_t *hram, uint32_t hlen, uint8_t *result);
int orc_pt_op_batch_fmt(const orc_curve *c, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2, int in_fmt,
			uint8_t *out, int out_fmt, uint8_t *status);
int orc_unprotected_mult_batch(const orc_curve *c, uint32_t n, const uint8_t *scalars, uint32_t slen, uint32_t sstride,
			       const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status);
int orc_y_from_x_batch(const orc_curve *c, uint32_t n, const uint8_t *xs, uint8_t *y1, uint8_t *y2, uint8_t *status);
#endif
