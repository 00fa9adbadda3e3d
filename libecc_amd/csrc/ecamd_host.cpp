// libecc_amd/csrc/ecamd_host.cpp -- host side of the C ABI declared in include/libecc_amd.h.
//
// What lives here (and mirrors in the reference, paths relative to /root/reference/src):
//   * the built-in curve table and import_params (curves/ec_params.c:24-194): domain parameters
//     -> Montgomery constants R, R^2, -p^-1, a*R, b*R, 3b*R, p-2 (derivations as in
//     scripts/expand_libecc.py:62-71, but for our radix 2^(32*NW)), uploaded to __constant__;
//   * batch plumbing: staging buffers, per-lane window-table scratch, chunking, streams.
// No arithmetic of the hot path runs on the host and there is no CPU fallback: every entry
// point needs a live HIP device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <strings.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <functional>
#include <mutex>
#include <sys/random.h>

#include "../../include/libecc_amd.h"
#include "ecamd_internal.h"

// ------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string &m)
{
	g_err = m;
	return -1;
}
#define HIPCHK(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) { \
			return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
		} \
	} while (0)

extern "C" const char *ecamd_last_error(void) { return g_err.c_str(); }
// for the other translation units of the library (ecamd_multi.cpp): not part of the public ABI
__attribute__((visibility("default"))) void ecamd_set_error(const char *msg) { g_err = msg ? msg : ""; }

// ------------------------------------------------------------------------------------------
// small host big integers (little-endian 32-bit words); only used to derive constants
// ------------------------------------------------------------------------------------------
typedef std::vector<uint32_t> Big;

static void big_trim(Big &a)
{
	while (a.size() > 1 && a.back() == 0) {
		a.pop_back();
	}
	if (a.empty()) {
		a.push_back(0);
	}
}
static Big big_from_be(const uint8_t *b, size_t len)
{
	Big r((len + 3) / 4 + 1, 0);
	for (size_t i = 0; i < len; i++) {
		size_t pos = len - 1 - i;
		r[pos / 4] |= (uint32_t)b[i] << (8 * (pos % 4));
	}
	big_trim(r);
	return r;
}
static Big big_from_hex(const char *h)
{
	size_t n = strlen(h);
	Big r(n / 8 + 2, 0);
	for (size_t i = 0; i < n; i++) {
		char c = h[n - 1 - i];
		uint32_t v = (c >= '0' && c <= '9') ? (uint32_t)(c - '0')
			     : (c >= 'a' && c <= 'f') ? (uint32_t)(c - 'a' + 10) : (uint32_t)(c - 'A' + 10);
		r[i / 8] |= v << (4 * (i % 8));
	}
	big_trim(r);
	return r;
}
static int big_cmp(const Big &a, const Big &b)
{
	size_t n = a.size() > b.size() ? a.size() : b.size();
	for (size_t i = n; i-- > 0;) {
		uint32_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
		if (x != y) {
			return x < y ? -1 : 1;
		}
	}
	return 0;
}
static int big_bitlen(const Big &a)
{
	for (size_t i = a.size(); i-- > 0;) {
		if (a[i]) {
			int b = 31;
			while (!((a[i] >> b) & 1)) {
				b--;
			}
			return (int)i * 32 + b + 1;
		}
	}
	return 0;
}
static Big big_add(const Big &a, const Big &b)
{
	size_t n = (a.size() > b.size() ? a.size() : b.size()) + 1;
	Big r(n, 0);
	uint64_t c = 0;
	for (size_t i = 0; i < n; i++) {
		c += (uint64_t)(i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0);
		r[i] = (uint32_t)c;
		c >>= 32;
	}
	big_trim(r);
	return r;
}
static Big big_sub(const Big &a, const Big &b)  // a >= b
{
	Big r(a.size(), 0);
	int64_t c = 0;
	for (size_t i = 0; i < a.size(); i++) {
		int64_t x = (int64_t)a[i] - (i < b.size() ? b[i] : 0) + c;
		r[i] = (uint32_t)x;
		c = x >> 32;
	}
	big_trim(r);
	return r;
}
static Big big_mul(const Big &a, const Big &b)
{
	Big r(a.size() + b.size() + 1, 0);
	for (size_t i = 0; i < a.size(); i++) {
		uint64_t c = 0;
		for (size_t j = 0; j < b.size(); j++) {
			c += (uint64_t)a[i] * b[j] + r[i + j];
			r[i + j] = (uint32_t)c;
			c >>= 32;
		}
		r[i + b.size()] += (uint32_t)c;
	}
	big_trim(r);
	return r;
}
static Big big_mod(const Big &a, const Big &m)  // bitwise shift-subtract; sizes are tiny
{
	Big r(1, 0);
	for (int bit = big_bitlen(a) - 1; bit >= 0; bit--) {
		// r = 2r + bit
		Big t(r.size() + 1, 0);
		for (size_t i = 0; i < r.size(); i++) {
			t[i] |= r[i] << 1;
			t[i + 1] |= r[i] >> 31;
		}
		t[0] |= (a[(size_t)bit / 32] >> (bit % 32)) & 1u;
		big_trim(t);
		r = (big_cmp(t, m) >= 0) ? big_sub(t, m) : t;
	}
	return r;
}
static Big big_pow2(int e)
{
	Big r((size_t)e / 32 + 1, 0);
	r[(size_t)e / 32] = 1u << (e % 32);
	return r;
}
static Big big_mulmod(const Big &a, const Big &b, const Big &m) { return big_mod(big_mul(a, b), m); }
static Big big_powmod(const Big &a, const Big &e, const Big &m)
{
	Big r(1, 1), base = big_mod(a, m);
	for (int i = big_bitlen(e) - 1; i >= 0; i--) {
		r = big_mulmod(r, r, m);
		if ((e[(size_t)i / 32] >> (i % 32)) & 1u) {
			r = big_mulmod(r, base, m);
		}
	}
	return r;
}
static void big_store(uint32_t *dst, int nw, const Big &a)
{
	for (int i = 0; i < nw; i++) {
		dst[i] = (size_t)i < a.size() ? a[(size_t)i] : 0;
	}
}
static void big_to_be(uint8_t *dst, int len, const Big &a)
{
	for (int i = 0; i < len; i++) {
		int pos = len - 1 - i;
		dst[i] = ((size_t)pos / 4 < a.size()) ? (uint8_t)(a[(size_t)pos / 4] >> (8 * (pos % 4))) : 0;
	}
}

// ------------------------------------------------------------------------------------------
// curve table (domain parameters only; generated by tools/gen_curve_table.py)
// ------------------------------------------------------------------------------------------
struct CurveRow {
	const char *name;
	int type;
	const char *p, *a, *b, *order, *gx, *gy, *q, *h;
};
static const CurveRow g_curve_rows[] = {
#include "ecamd_curve_table.inc"
};

// ------------------------------------------------------------------------------------------
// objects behind the opaque handles
// ------------------------------------------------------------------------------------------
#define ECAMD_NSTAGE 30   /* 24 .. 28: the batch-wide arrays of ec_schnorr_verify_msg_all_batch */
struct ecamd_ctx {
	int device;
	hipStream_t stream;
	uint32_t max_chunk;
	// grow-only scratch
	uint32_t *tbl;      // complete-formula kernel: 16 x 3 x NW words per item, word-major
	size_t tbl_bytes;
	uint32_t *tbl_fast; // fast paths: per-item window tables (secp256r1: 512 B affine table + 1120 B staging per item)
	size_t tbl_fast_bytes;
	uint8_t *prep_scratch;     // k_ecdsa_prep_g: prefix products, n x NL words
	size_t prep_scratch_bytes;
	uint8_t *stage[ECAMD_NSTAGE];
	size_t stage_bytes[ECAMD_NSTAGE];
	// the scratch above is shared by every call: a call enqueued on another stream than the previous one first waits
	// for `busy`, recorded behind the previous call's last kernel (StreamScope)
	hipEvent_t busy;
	hipStream_t last_stream;
	bool inflight;
	// host-pointer entry points: chunks of host_chunk items, the copy of chunk c+1 overlaps the kernels of chunk c
	hipStream_t copy_stream;
	hipEvent_t in_ready[2];
	// ECDSA verification on secp256r1: k_ecdsa_prep (s^-1, u1, u2 mod q: two waves per SIMD, compute only) runs on this stream
	// beside the bandwidth-bound k_p256_table / k_p256_affine of the same chunk; the interleaved window loop -- the first reader
	// of u1, u2 and the flags -- waits for side_done ($ECAMD_NO_SIDE_STREAM: off).  66.0 -> 67.0 M verifications/s.
	hipStream_t side_stream;
	hipEvent_t side_fork, side_done, side_mid, side_aux;   // (side_mid / side_aux: the bucket evaluation's second and third joins)
	hipEvent_t side_hi, side_red;   // the bucket evaluation: the key-only windows are summed (on the caller's stream) / reduced (on the side stream)
	hipStream_t side2_stream;       // the bucket evaluation: the windows below 2^128 are filed here while the key-only windows are summed
	hipEvent_t side2_fork, side2_done;
	bool side2_ok;
	bool side_ok;
	uint32_t host_chunk;
	uint32_t host_first_min;   // smallest first chunk of a multi-chunk call (ECAMD_HOST_RAMP_MIN, default 2^16; host_pipeline)
	// producer hook of the host-pointer entry points (ecamd_ctx_set_host_ready_hook): called before a range of the caller's input arrays is read
	ecamd_host_ready_fn ready_fn = nullptr;
	void *ready_arg = nullptr;
	uint8_t *hbuf[2][6];       // double-buffered device staging of the caller's arrays (inputs and outputs)
	size_t hbuf_bytes[2][6];
	uint32_t comb_min_batch;   // fixed-base batches of at least this many items build / use the generator's comb table (0: never)
	bool secret_scalars;       // ecamd_ctx_set_secret_scalars: constant-address look-ups, no data-dependent kernel choice
	int eddsa_msm;             // ecamd_ctx_set_eddsa_msm: 0 never, 1 batches of at least msm_min items, 2 always
	uint32_t msm_min, msm_k;   // msm_k: items per lane of the Straus evaluation (0: chosen from the batch size)
	uint8_t *msm;              // scratch of the multi-scalar multiplication
	size_t msm_bytes;
	uint8_t msm_seed_bytes[32]; // ecamd_ctx_set_msm_seed: key of the next whole-batch combination's z_i (used once)
	bool msm_seed_valid;
	bool timing;               // record HIP events around the kernels of the scalar-mult pipeline
	hipEvent_t ev[ECAMD_NTIMED + 1];
	bool ev_valid;
	hipEvent_t ev_dom[2];      // around the dominant kernel of the last protocol call (verify loop, ladder, Edwards window loop)
	bool ev_dom_valid;
	std::mutex mu;
};

struct ecamd_curve {
	ecamd_ctx *ctx;
	int nw;     // 32-bit words per element
	int slot;   // __constant__ slot: field of definition
	int qslot;  // __constant__ slot: generator order q as a modulus (-1: protocol ops unavailable)
	int curve_type;  // libecc's ec_curve_type of a built-in curve (structured keys carry it), 0 for user curves
	int qnw;    // words of the mod-q kernels: nw, or more when q is longer than p (secp224k1: |q| = 225 > 224)
	int clen;   // BYTECEIL(pbits)
	int qlen;   // BYTECEIL(qbits)
	int pbits, qbits;
	Big p, a, b, order, gx, gy, q;
	uint8_t *d_gen;  // generator, affine X||Y big-endian, in HBM; followed by q (qlen bytes) and the cofactor byte
	uint32_t cofactor; // order / q when that is 1..255, else 0
	// per-curve constants of the Ed25519 / X25519 / X448 entry points, computed on first use (ctx->mu held)
	int ed_state;    // 0 not yet, 1 ready, -1 this handle is not the Ed25519 model (ed_err says why)
	const char *ed_err;
	EcamdEdDecodeArgs ed_tmpl;
	uint32_t ed_cof_dbl;
	int ed448_state;     // Ed448 on the WEI448 handle: 0 not yet, 1 ready, -1 not that curve
	const char *ed448_err;
	EcamdEd448DecodeArgs ed448_tmpl;
	uint8_t ed448_c4[56]; // 4^-1 mod q, big-endian (eddsa_import_pub_key multiplies the key by it)
	uint32_t ed_2d[9];   // 2 d mod p, plain radix-2^29 digits (Edwards arithmetic of the 2^255 - 19 unit)
	uint32_t ed_Bx[9], ed_By[9];  // the Ed25519 base point, the same digits
	int xdh_state;
	const char *xdh_err;
	EcamdXdhPrepArgs xdh_tmpl;
	uint32_t xdh_A3[17];
	uint8_t xdh_cof;
	int sqrt_state;  // Tonelli-Shanks constants of fp_sqrt for this field: 0 not yet, 1 ready (ctx->mu held)
	EcamdYfromXArgs sqrt_tmpl;
	bool is_p256;    // exactly secp256r1: hand-specialised radix-2^29 Jacobian kernel
	int gslot;       // constant slot of the generic radix-2^29 Jacobian kernel (-1: none)
	int gpslot;      // secp256r1 handles only (their scalar multiplication has its own kernels): slot of the dense 256-bit radix-2^29 unit holding
			 // the curve, for k_prj_import_g (-1: none, k_prj_import<8> serves)
	int gqslot;      // constant slot of the dense radix-2^29 unit of the ORDER's size holding q as its modulus (k_ecdsa_prep_g; -1: none)
	int gflavour;    // 0 dense reduction, 1 secp521r1 (p = 2^521 - 1), 2 p = 2^255 - 19, 3 secp384r1's prime, 4 secp256k1's prime, 5 p = 2^448 - 2^224 - 1,
	                 // 6 secp224r1's prime, 7 secp192r1's prime (3, 6, 7: signed sparse Montgomery reduction)
	uint32_t *d_comb4;  // the 4-bit comb of the generator for SECRET scalars: secp256r1 65 x 8 entries of 40 words (k_p256_comb4m, built with the
			    // handle), radix-2^29 units (8 NW + 1) x 8 comb entries (k_comb_g<.., SCAN4>, built on first use); NULL: none
	bool comb4_off;     // ... its construction was tried
	uint32_t *d_gtab; // secp256r1: affine window table [1..8]G, radix-2^29 Montgomery digits, 8 x 40 words
	uint32_t *d_comb; // fast paths: 16-bit comb table of G, built on the first large fixed-base batch (NULL before / disabled)
	bool comb_off;    // construction failed or is in progress: do not try (again)
	uint32_t *d_edcomb; // WEI25519 on the 2^255 - 19 unit: 16-bit comb table of the Ed25519 base point ON THE EDWARDS CURVE
	                    // (k_ed_tail_c25519), built on the first verification batch of comb_min_batch items; NULL before / disabled
	bool edcomb_off;
	uint32_t qdig[9]; // secp256r1: digits of the group order
};

static const int k_widths[] = {6, 7, 8, 10, 12, 14, 16, 17};

static int pick_nw(int pbits)
{
	for (int w : k_widths) {
		if (32 * w >= pbits) {
			return w;
		}
	}
	return 0;
}

extern "C" int ecamd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		return 0;
	}
	return n;
}

extern "C" int ecamd_ctx_create(ecamd_ctx **out, int device)
{
	if (!out) {
		return fail("ecamd_ctx_create: NULL out pointer");
	}
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		return fail("ecamd_ctx_create: no HIP device available (this library has no CPU fallback)");
	}
	if (device < 0 || device >= n) {
		return fail("ecamd_ctx_create: device index out of range");
	}
	HIPCHK(hipSetDevice(device));
	ecamd_ctx *c = new ecamd_ctx();
	c->device = device;
	c->max_chunk = 1u << 20;
	{
		// fixed-base comb tables: built on the first fixed-base batch of at least this many items
		// (ECAMD_COMB_MIN_BATCH, default 4096; ECAMD_NO_COMB disables them)
		const char *e = getenv("ECAMD_COMB_MIN_BATCH");
		c->comb_min_batch = e ? (uint32_t)strtoul(e, nullptr, 10) : 4096u;
		if (getenv("ECAMD_NO_COMB") != nullptr) {
			c->comb_min_batch = 0;
		}
	}
	c->tbl = nullptr;
	c->tbl_bytes = 0;
	c->tbl_fast = nullptr;
	c->tbl_fast_bytes = 0;
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		c->stage[i] = nullptr;
		c->stage_bytes[i] = 0;
	}
	c->prep_scratch = nullptr;
	c->prep_scratch_bytes = 0;
	c->last_stream = nullptr;
	c->inflight = false;
	if (hipEventCreateWithFlags(&c->busy, hipEventDisableTiming) != hipSuccess) {
		delete c;
		return fail("ecamd_ctx_create: hipEventCreate failed");
	}
	c->timing = false;
	c->secret_scalars = false;
	c->eddsa_msm = 1;
	c->msm_min = 1u << 17;   // measured break-even: 1.16x at 2^17, 0.92x at 2^16 (profiles/r2l_eddsa_msm.json)
	c->msm_k = 0;
	{
		const char *e = getenv("ECAMD_MSM_MIN");
		if (e) {
			c->msm_min = (uint32_t)strtoul(e, nullptr, 10);
		}
		e = getenv("ECAMD_MSM_K");
		if (e) {
			c->msm_k = (uint32_t)strtoul(e, nullptr, 10);
		}
	}
	c->msm = nullptr;
	c->msm_bytes = 0;
	c->msm_seed_valid = false;
	c->ev_valid = false;
	c->ev_dom_valid = false;
	for (int i = 0; i <= ECAMD_NTIMED; i++) {
		if (hipEventCreate(&c->ev[i]) != hipSuccess) {
			delete c;
			return fail("ecamd_ctx_create: hipEventCreate failed");
		}
	}
	if (hipEventCreate(&c->ev_dom[0]) != hipSuccess || hipEventCreate(&c->ev_dom[1]) != hipSuccess) {
		delete c;
		return fail("ecamd_ctx_create: hipEventCreate failed");
	}
	memset(c->hbuf, 0, sizeof(c->hbuf));
	memset(c->hbuf_bytes, 0, sizeof(c->hbuf_bytes));
	{
		const char *e = getenv("ECAMD_HOST_CHUNK");
		// (round 4: 2^19 -- a 2^18-item launch runs the window kernels at 0.92 of their 2^20 rate, a 2^19-item one at 0.97,
		// and two pieces of a 2^20 batch still overlap the second's copy with the first's kernels; profiles/r4_batch_sweep.md)
		c->host_chunk = e ? (uint32_t)strtoul(e, nullptr, 10) : (1u << 19);
		if (c->host_chunk == 0) {
			c->host_chunk = 1u << 19;
		}
		e = getenv("ECAMD_HOST_RAMP_MIN");
		c->host_first_min = e ? (uint32_t)strtoul(e, nullptr, 10) : (1u << 16);
		if (c->host_first_min == 0) {
			c->host_first_min = 1u << 16;
		}
	}
	if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&c->in_ready[0], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->in_ready[1], hipEventDisableTiming) != hipSuccess) {
		delete c;
		return fail("ecamd_ctx_create: copy stream creation failed");
	}
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
		delete c;
		return fail("ecamd_ctx_create: hipStreamCreate failed");
	}
	if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_mid, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_aux, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_hi, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_red, hipEventDisableTiming) != hipSuccess ||
	    hipStreamCreateWithFlags(&c->side2_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side2_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side2_done, hipEventDisableTiming) != hipSuccess) {
		delete c;
		return fail("ecamd_ctx_create: side stream creation failed");
	}
	c->side_ok = getenv("ECAMD_NO_SIDE_STREAM") == nullptr;
	c->side2_ok = c->side_ok;
	*out = c;
	return 0;
}

extern "C" void ecamd_ctx_destroy(ecamd_ctx *c)
{
	if (!c) {
		return;
	}
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	if (c->tbl) {
		(void)hipFree(c->tbl);
	}
	if (c->tbl_fast) {
		(void)hipFree(c->tbl_fast);
	}
	if (c->msm) {
		(void)hipFree(c->msm);
	}
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		if (c->stage[i]) {
			(void)hipFree(c->stage[i]);
		}
	}
	if (c->prep_scratch) {
		(void)hipFree(c->prep_scratch);
	}
	for (int i = 0; i <= ECAMD_NTIMED; i++) {
		(void)hipEventDestroy(c->ev[i]);
	}
	(void)hipEventDestroy(c->ev_dom[0]);
	(void)hipEventDestroy(c->ev_dom[1]);
	(void)hipStreamDestroy(c->stream);
	(void)hipStreamDestroy(c->copy_stream);
	(void)hipStreamSynchronize(c->side_stream);
	(void)hipStreamDestroy(c->side_stream);
	(void)hipEventDestroy(c->side_fork);
	(void)hipEventDestroy(c->side_done);
	(void)hipEventDestroy(c->side_mid);
	(void)hipEventDestroy(c->side_aux);
	(void)hipEventDestroy(c->side_hi);
	(void)hipEventDestroy(c->side_red);
	(void)hipStreamSynchronize(c->side2_stream);
	(void)hipStreamDestroy(c->side2_stream);
	(void)hipEventDestroy(c->side2_fork);
	(void)hipEventDestroy(c->side2_done);
	(void)hipEventDestroy(c->in_ready[0]);
	(void)hipEventDestroy(c->in_ready[1]);
	(void)hipEventDestroy(c->busy);
	for (int b = 0; b < 2; b++) {
		for (int k = 0; k < 6; k++) {
			if (c->hbuf[b][k]) {
				(void)hipFree(c->hbuf[b][k]);
			}
		}
	}
	delete c;
}

extern "C" int ecamd_ctx_set_max_chunk(ecamd_ctx *c, uint32_t max_items)
{
	if (!c || max_items == 0) {
		return fail("ecamd_ctx_set_max_chunk: bad argument");
	}
	c->max_chunk = max_items;
	return 0;
}

extern "C" int ecamd_ctx_set_eddsa_msm(ecamd_ctx *c, int mode, uint32_t min_items, uint32_t items_per_lane)
{
	if (!c || mode < 0 || mode > 2 || items_per_lane > 64) {
		return fail("ecamd_ctx_set_eddsa_msm: bad argument");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	c->eddsa_msm = mode;
	if (min_items) {
		c->msm_min = min_items;
	}
	c->msm_k = items_per_lane;
	return 0;
}

// The z_i of the next ec_eddsa_verify_all_batch[_dev] call on this context are ChaCha20(seed, item index) instead of being keyed
// by getrandom: an application that owns the randomness source of its libecc (get_random) keys the batch equation with it.
// Used once; the bytes are wiped when consumed.
extern "C" int ecamd_ctx_set_msm_seed(ecamd_ctx *c, const uint8_t seed[32])
{
	if (!c || !seed) {
		return fail("ecamd_ctx_set_msm_seed: bad argument");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	memcpy(c->msm_seed_bytes, seed, 32);
	c->msm_seed_valid = true;
	return 0;
}

// Drops a seed that is still pending (ecamd_multi_*_verify_all_batch: a rank whose shard was empty never consumed its copy).
extern "C" int ecamd_ctx_discard_msm_seed(ecamd_ctx *c)
{
	if (!c) {
		return fail("ecamd_ctx_discard_msm_seed: NULL context");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	if (c->msm_seed_valid) {
		memset(c->msm_seed_bytes, 0, sizeof(c->msm_seed_bytes));
		c->msm_seed_valid = false;
	}
	return 0;
}

extern "C" int ecamd_ctx_set_secret_scalars(ecamd_ctx *c, int on)
{
	if (!c) {
		return fail("ecamd_ctx_set_secret_scalars: NULL context");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	c->secret_scalars = on != 0;
	return 0;
}

extern "C" int ecamd_ctx_enable_kernel_timing(ecamd_ctx *c, int on)
{
	if (!c) {
		return fail("ecamd_ctx_enable_kernel_timing: NULL context");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	c->timing = on != 0;
	c->ev_valid = false;
	c->ev_dom_valid = false;
	return 0;
}

extern "C" int ecamd_ctx_kernel_times(ecamd_ctx *c, double *ms, int n)
{
	if (!c || !ms || n < ECAMD_NTIMED - 1) {
		return fail("ecamd_ctx_kernel_times: bad argument (need room for 4 values)");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	if (!c->timing || !c->ev_valid) {
		return fail("ecamd_ctx_kernel_times: timing not enabled or no fast-path batch ran since");
	}
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipEventSynchronize(c->ev[ECAMD_NTIMED - 1]));
	for (int i = 0; i < ECAMD_NTIMED - 1; i++) {
		float f = 0.0f;
		HIPCHK(hipEventElapsedTime(&f, c->ev[i], c->ev[i + 1]));
		ms[i] = (double)f;
	}
	return 0;
}

// duration of the dominant kernel of the last protocol call made with timing enabled: k_p256_verify_loop (ECDSA verification
// on secp256r1), k_x25519_ladder / k_x448_ladder (X25519 / X448), k_ed_smul_c25519<1> (the [h]A window loop of Ed25519
// verification); the first chunk of the call
extern "C" int ecamd_ctx_dominant_kernel_ms(ecamd_ctx *c, double *ms)
{
	if (!c || !ms) {
		return fail("ecamd_ctx_dominant_kernel_ms: bad argument");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	if (!c->timing || !c->ev_dom_valid) {
		return fail("ecamd_ctx_dominant_kernel_ms: timing not enabled or no protocol call with a timed kernel ran since");
	}
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipEventSynchronize(c->ev_dom[1]));
	float f = 0.0f;
	HIPCHK(hipEventElapsedTime(&f, c->ev_dom[0], c->ev_dom[1]));
	*ms = (double)f;
	return 0;
}

extern "C" void *ecamd_ctx_stream(ecamd_ctx *c) { return c ? (void *)c->stream : nullptr; }

// Page-locked host memory for the arrays handed to the host-pointer entry points: the copies to and from the device then run
// as asynchronous DMA at PCIe rate instead of going through the runtime's pageable staging.  Portable: valid for every device.
extern "C" void *ecamd_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
		(void)hipGetLastError();
		fail("ecamd_host_alloc: hipHostMalloc failed");
		return nullptr;
	}
	return p;
}

extern "C" void ecamd_host_free(void *p)
{
	if (p) {
		(void)hipHostFree(p);
	}
}

// Zero every scratch buffer of the context (window tables, recoded scalars, staged inputs and outputs): after calls that
// worked on secret scalars nothing derived from them stays in HBM.  Enqueued behind the context's last call; synchronous.
extern "C" int ecamd_ctx_wipe_scratch(ecamd_ctx *c)
{
	if (!c) {
		return fail("ecamd_ctx_wipe_scratch: NULL context");
	}
	std::lock_guard<std::mutex> lk(c->mu);
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	if (c->inflight && c->last_stream != s) {
		HIPCHK(hipStreamWaitEvent(s, c->busy, 0));
	}
	HIPCHK(hipStreamSynchronize(c->copy_stream));
	if (c->tbl) {
		HIPCHK(hipMemsetAsync(c->tbl, 0, c->tbl_bytes, s));
	}
	if (c->tbl_fast) {
		HIPCHK(hipMemsetAsync(c->tbl_fast, 0, c->tbl_fast_bytes, s));
	}
	if (c->msm) {
		HIPCHK(hipMemsetAsync(c->msm, 0, c->msm_bytes, s));
	}
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		if (c->stage[i]) {
			HIPCHK(hipMemsetAsync(c->stage[i], 0, c->stage_bytes[i], s));
		}
	}
	if (c->prep_scratch) {
		HIPCHK(hipMemsetAsync(c->prep_scratch, 0, c->prep_scratch_bytes, s));
	}
	for (int b = 0; b < 2; b++) {
		for (int k = 0; k < 6; k++) {
			if (c->hbuf[b][k]) {
				HIPCHK(hipMemsetAsync(c->hbuf[b][k], 0, c->hbuf_bytes[b][k], s));
			}
		}
	}
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

extern "C" int ecamd_ctx_synchronize(ecamd_ctx *c)
{
	if (!c) {
		return fail("ecamd_ctx_synchronize: NULL context");
	}
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	return 0;
}

// Every call works in the context's shared scratch (window tables, stage[] buffers).  A call enqueued on another
// stream than the previous one therefore first makes its stream wait for the previous call's last kernel, and
// leaves an event behind its own last kernel (ctx->mu held for the lifetime of the scope).
// (the stream of the innermost scope on this thread: what ensure() orders the wipe of a growing scratch buffer behind)
static thread_local hipStream_t t_scope_stream = nullptr;
static thread_local bool t_scope_active = false;
struct StreamScope {
	ecamd_ctx *c;
	hipStream_t s, outer;
	bool outer_active;
	StreamScope(ecamd_ctx *ctx, hipStream_t stream) : c(ctx), s(stream), outer(t_scope_stream), outer_active(t_scope_active)
	{
		if (c->inflight && c->last_stream != s) {
			(void)hipStreamWaitEvent(s, c->busy, 0);
		}
		t_scope_stream = s;
		t_scope_active = true;
	}
	~StreamScope()
	{
		if (hipEventRecord(c->busy, s) == hipSuccess) {
			c->last_stream = s;
			c->inflight = true;
		}
		t_scope_stream = outer;
		t_scope_active = outer_active;
	}
};

// Scalars that are public by construction (u1, u2 of a verification, h and S of EdDSA, the group order and cofactor of a
// subgroup check) keep the digit-indexed look-ups and the comb table even when the context is in secret-scalar mode
// (ecamd_ctx_set_secret_scalars is about private keys, nonces and blinded scalars).  ctx->mu held.
struct PublicScalars {
	ecamd_ctx *c;
	bool saved;
	explicit PublicScalars(ecamd_ctx *ctx) : c(ctx), saved(ctx->secret_scalars) { c->secret_scalars = false; }
	~PublicScalars() { c->secret_scalars = saved; }
};

static int ensure(uint8_t **buf, size_t *have, size_t need)
{
	if (*have >= need) {
		return 0;
	}
	if (*buf) {
		// a scratch buffer that has to grow may hold values derived from secret scalars (window tables, recoded or staged
		// scalars of an earlier group of the same call): it goes back to the allocator zeroed.  The wipe is ordered
		// behind the call's own stream -- everything that used the buffer was enqueued on it or is waited for by it (StreamScope, the
		// side stream's join) -- and only that stream is drained: other contexts on the device keep running (a
		// device-wide synchronisation stalled every rank of a many-contexts-per-device run on each growth).
		if (t_scope_active) {
			HIPCHK(hipMemsetAsync(*buf, 0, *have, t_scope_stream));
			HIPCHK(hipStreamSynchronize(t_scope_stream));
		} else {
			HIPCHK(hipMemset(*buf, 0, *have));
			HIPCHK(hipDeviceSynchronize());
		}
		HIPCHK(hipFree(*buf));
		*buf = nullptr;
		*have = 0;
	}
	// grow with an eighth of headroom: a sequence of slowly growing batches reallocates (and wipes, and waits) a few times, not every call
	const size_t want = need + (need >> 3);
	if (hipMalloc((void **)buf, want) == hipSuccess) {
		*have = want;
		return 0;
	}
	(void)hipGetLastError();
	HIPCHK(hipMalloc((void **)buf, need));
	*have = need;
	return 0;
}


// ------------------------------------------------------------------------------------------
// __constant__ curve slots.  The tables the kernels index (g_curves_<NW>[slot], g_g29_<unit>[slot]) exist once per
// DEVICE and per process, whatever the number of contexts opened on that device, so their bookkeeping is
// process-global: one registry per device under one mutex, reference-counted, and handles whose constant image is
// byte-identical share a slot (two contexts that both load SECP256R1 use the same constants; two contexts that load
// different curves of one width get different slots instead of overwriting each other).  A slot is only written
// while no live handle refers to it, so an upload never races with a kernel that reads it.
// ------------------------------------------------------------------------------------------
#define ECAMD_MAX_DEVICES 64
#define ECAMD_G29_KEYS 640  /* pbits + flavour */
struct SlotEnt {
	int ref = 0;
	std::vector<uint32_t> img;  // what the slot holds (kept after the last release: a re-acquire needs no upload)
};
struct SlotRegistry {
	SlotEnt s[18][ECAMD_MAX_SLOTS_HOST];  // [nw][slot]: saturated-word units
	SlotEnt g[ECAMD_G29_KEYS][8];         // [pbits + flavour][slot]: radix-2^29 units
};
static std::mutex g_slot_mu;
static SlotRegistry *g_slots[ECAMD_MAX_DEVICES];

// find (or fill) a slot of `row` holding `img`; returns the slot or -1 (none free) / -2 (upload failed)
template <class Upload> static int slot_acquire(SlotEnt *row, int nslots, const std::vector<uint32_t> &img, Upload upload)
{
	for (int i = 0; i < nslots; i++) {
		if (row[i].ref > 0 && row[i].img == img) {
			row[i].ref++;
			return i;
		}
	}
	int pick = -1;
	for (int i = 0; i < nslots && pick < 0; i++) {
		if (row[i].ref == 0 && row[i].img == img) {
			pick = i;  // still resident from an earlier handle
		}
	}
	bool need_upload = pick < 0;
	for (int i = 0; i < nslots && pick < 0; i++) {
		if (row[i].ref == 0 && row[i].img.empty()) {
			pick = i;
		}
	}
	for (int i = 0; i < nslots && pick < 0; i++) {
		if (row[i].ref == 0) {
			pick = i;
		}
	}
	if (pick < 0) {
		return -1;
	}
	if (need_upload) {
		row[pick].img.clear();
		if (upload(pick) != hipSuccess) {
			return -2;
		}
		row[pick].img = img;
	}
	row[pick].ref = 1;
	return pick;
}
static SlotRegistry *slot_registry(int device)
{
	if (device < 0 || device >= ECAMD_MAX_DEVICES) {
		return nullptr;
	}
	if (!g_slots[device]) {
		g_slots[device] = new SlotRegistry();
	}
	return g_slots[device];
}

// CurveK<NW> as a flat word image: p r2 one pm2 a b b3 fix64 (NW words each), then
// mpinv pbits a_is_m3 fix_is_id -- must match ecamd_field.h.  Returns the slot holding it (g_slot_mu held).
static int acquire_modulus(int device, int nw, const Big &p, const Big &a_in, const Big &b_in, int *slot_out)
{
	Big a_red = big_mod(a_in, p), b_red = big_mod(b_in, p);
	struct {
		const Big &a, &b;
		int pbits;
	} cvv = {a_red, b_red, big_bitlen(p)};
	auto *cv = &cvv;
	const Big R = big_mod(big_pow2(32 * nw), p);
	const Big R2 = big_mulmod(R, R, p);
	Big two(1, 2), three(1, 3);
	const Big pm2 = big_sub(p, two);
	const Big aR = big_mulmod(cv->a, R, p);
	const Big bR = big_mulmod(cv->b, R, p);
	const Big b3R = big_mulmod(big_mulmod(cv->b, three, p), R, p);
	// fix64 = R^2 * 2^(-64*ceil(nw/2)) mod p
	const int n64 = (nw + 1) / 2;
	const Big R64 = big_mod(big_pow2(64 * n64), p);
	const Big R64inv = big_powmod(R64, pm2, p);
	const Big fix = big_mulmod(R2, R64inv, p);
	// -p^-1 mod 2^32 by Newton iteration
	uint32_t p0 = p[0], x = 1;
	for (int i = 0; i < 5; i++) {
		x *= 2u - p0 * x;
	}
	const uint32_t mpinv = 0u - x;
	std::vector<uint32_t> img((size_t)8 * nw + 4, 0);
	big_store(&img[0 * nw], nw, p);
	big_store(&img[1 * nw], nw, R2);
	big_store(&img[2 * nw], nw, R);
	big_store(&img[3 * nw], nw, pm2);
	big_store(&img[4 * nw], nw, aR);
	big_store(&img[5 * nw], nw, bR);
	big_store(&img[6 * nw], nw, b3R);
	big_store(&img[7 * nw], nw, fix);
	img[8 * nw + 0] = mpinv;
	img[8 * nw + 1] = (uint32_t)cv->pbits;
	img[8 * nw + 2] = (big_cmp(big_add(cv->a, three), p) == 0) ? 1u : 0u;
	img[8 * nw + 3] = (big_cmp(fix, R) == 0) ? 1u : 0u;
	if (img.size() * 4 != ecamd_curvek_bytes(nw)) {
		return fail("internal: CurveK image size mismatch");
	}
	SlotRegistry *reg = slot_registry(device);
	if (!reg || nw < 0 || nw >= 18) {
		return fail("internal: no slot registry for this device / width");
	}
	const int slot = slot_acquire(reg->s[nw], ECAMD_MAX_SLOTS_HOST, img,
				      [&](int sl) { return ecamd_upload_curve(nw, sl, img.data(), img.size() * 4); });
	if (slot == -1) {
		return fail("curve: all constant-memory curve slots of this width are in use on this device (free a curve first)");
	}
	if (slot < 0) {
		return fail("curve: upload of the curve constants failed");
	}
	*slot_out = slot;
	return 0;
}
static void release_modulus(int device, int nw, int slot)
{
	SlotRegistry *reg = slot_registry(device);
	if (reg && nw >= 0 && nw < 18 && slot >= 0 && slot < ECAMD_MAX_SLOTS_HOST && reg->s[nw][slot].ref > 0) {
		reg->s[nw][slot].ref--;
	}
}

static int smul_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_scalars, uint32_t slen,
			   const uint8_t *d_points, uint8_t *d_out, uint8_t *d_status, hipStream_t s,
			   uint32_t sstride = 0xffffffffu, bool redo_only = false, const uint8_t *d_scalars2 = nullptr);
static int prep_scratch_ok(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n);
static hipError_t launch_ecdsa_prep(ecamd_ctx *ctx, const ecamd_curve *cv, const EcamdEcdsaPrepArgs &P, hipStream_t s);

// w-bit digits of a (nl of them, the last one takes whatever is left): 29 bits on every radix-2^29 unit but the Goldilocks one
// (flavour 5), which runs on 28-bit limbs so that 2^224 falls on a limb boundary (ecamd_u29g.h)
static void big_digits29(uint32_t *dst, int nl, const Big &a, int w = 29)
{
	const int bits = big_bitlen(a);
	for (int i = 0; i < nl; i++) {
		uint64_t d = 0;
		const int lo = w * i, hi = (i == nl - 1) ? (bits > lo + w ? bits : lo + w) : lo + w;
		for (int b = lo; b < hi && b < bits; b++) {
			if ((a[(size_t)b / 32] >> (b % 32)) & 1u) {
				d |= 1ull << (b - lo);
			}
		}
		dst[i] = (uint32_t)d;
	}
}
static Big big_shl(const Big &a, int e)
{
	return big_mul(a, big_pow2(e));
}

// CurveG<NL> image of ecamd_u29g.h: p r2 one a b pm2 (NL digits each), 16 bias tables, ix iy ex ey, mpinv pbits
// a_is_m3 pad -- mirrored by tools/g29_consts.py, which the CPU tests use against Python integers
// (p, a, b, pbits, flavour): the curve's own field (upload_g29), or the group order q with a = b = 0 on the dense unit of its
// size (upload_g29_order: the mod-q algebra of ECDSA verification runs on the same field code, k_ecdsa_prep_g)
static int upload_g29_mod(ecamd_curve *cv, const Big &p, const Big &a_in, const Big &b_in, int pbits, int gflavour, int *slot_out)
{
	const int nl = ecamd_g29_nl(pbits, gflavour);
	// flavours 2 (p = 2^255 - 19), 4 (secp256k1's prime) and 5 (p = 2^448 - 2^224 - 1) keep plain residues: R = 1
	const int w = (gflavour == 5) ? 28 : 29;   // limb width of the unit (g29::W of its translation unit)
	// flavours 1 (p = 2^521 - 1 on 18 limbs), 2, 4 and 5 keep plain residues: R = 1
	const bool plain = gflavour == 1 || gflavour == 2 || gflavour == 4 || gflavour == 5;
	const Big R = plain ? Big(1, 1) : big_mod(big_pow2(w * nl), p);
	Big two(1, 2), three(1, 3);
	static const int step29[16] = {2, 4, 6, 8, 10, 12, 14, 16, 2, 4, 6, 8, 10, 12, 14, 16};
	static const int step28[16] = {1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8};    // g29::bias_step of the no-headroom flavours
	const int *step = (gflavour == 5 || gflavour == 1) ? step28 : step29;
	static const int sv[16] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2};
	const int topsh = pbits - w * (nl - 1);
	const int off = (1 - topsh) > 0 ? (1 - topsh) : 0;
	// Isomorphism onto a curve with a = -3 (cheaper doubling): (x, y) -> (u^2 x, u^3 y) with u^4 a = -3, when such a u
	// exists.  Tried for p = 3 mod 4, where square roots are one exponentiation and exactly one of +-s is a square
	// (the brainpool r1 curves -- their t1 twins are these images --, two GOST 512-bit sets); ECAMD_NO_ISO disables it.
	Big u(1, 1);
	Big a_img = a_in, b_img = b_in;
	if ((gflavour == 0 || gflavour == 3 || gflavour == 5) && (p[0] & 3u) == 3u && big_bitlen(a_in) > 0 && big_cmp(big_add(a_in, three), p) != 0 &&
	    getenv("ECAMD_NO_ISO") == nullptr) {
		Big e = big_add(p, Big(1, 1));  // (p + 1) / 4
		Big q4(e.size(), 0);
		for (size_t i = 0; i < e.size(); i++) {
			q4[i] = (e[i] >> 2) | ((i + 1 < e.size()) ? (e[i + 1] << 30) : 0u);
		}
		big_trim(q4);
		const Big t = big_mulmod(big_sub(p, three), big_powmod(a_in, big_sub(p, two), p), p);  // -3 / a
		Big s1 = big_powmod(t, q4, p);
		if (big_cmp(big_mulmod(s1, s1, p), t) == 0) {
			Big r = big_powmod(s1, q4, p);
			if (big_cmp(big_mulmod(r, r, p), s1) != 0) {
				s1 = big_sub(p, s1);
				r = big_powmod(s1, q4, p);
			}
			if (big_cmp(big_mulmod(r, r, p), s1) == 0) {
				u = r;
				const Big u2 = big_mulmod(u, u, p), u4 = big_mulmod(u2, u2, p);
				a_img = big_mulmod(a_in, u4, p);
				b_img = big_mulmod(b_in, big_mulmod(u4, u2, p), p);
				if (big_cmp(big_add(a_img, three), p) != 0) {
					return fail("internal: isomorphism onto a = -3 failed");
				}
			}
		}
	}
	const Big u2 = big_mulmod(u, u, p), u3 = big_mulmod(u2, u, p);
	const Big RR = big_mulmod(R, R, p);
	std::vector<uint32_t> img((size_t)(10 + 16) * nl + 4, 0);
	big_digits29(&img[0 * nl], nl, p, w);
	big_digits29(&img[1 * nl], nl, RR, w);
	big_digits29(&img[2 * nl], nl, R, w);
	big_digits29(&img[3 * nl], nl, big_mulmod(a_img, R, p), w);
	big_digits29(&img[4 * nl], nl, big_mulmod(b_img, R, p), w);
	big_digits29(&img[5 * nl], nl, big_sub(p, two), w);
	big_digits29(&img[22 * nl], nl, big_mulmod(u2, RR, p), w);                             // ix
	big_digits29(&img[23 * nl], nl, big_mulmod(u3, RR, p), w);                             // iy
	big_digits29(&img[24 * nl], nl, big_powmod(u2, big_sub(p, two), p), w);                // ex = u^-2
	big_digits29(&img[25 * nl], nl, big_powmod(u3, big_sub(p, two), p), w);                // ey = u^-3
	for (int t = 0; t < 16; t++) {
		uint32_t *l = &img[(size_t)(6 + t) * nl];
		if (big_bitlen(p) + step[t] + off > w * (nl - 1) + 32) {
			continue;  // this multiple does not fit the limbs (only without a headroom limb); never selected
		}
		big_digits29(l, nl, big_shl(p, step[t] + off), w);
		const uint32_t M = 1u << (w + sv[t]), BW = 1u << sv[t];
		if (l[nl - 1] < BW) {
			if (gflavour == 5 || gflavour == 1) {
				memset(l, 0, sizeof(uint32_t) * (size_t)nl);
				continue;  // a multiple too small to lend the borrow; never selected (BiasB's static_asserts)
			}
			return fail("internal: bias table underflow");
		}
		l[0] += M;
		for (int j = 1; j < nl - 1; j++) {
			l[j] += M - BW;
		}
		l[nl - 1] -= BW;
	}
	uint32_t p0 = p[0], x = 1;
	for (int i = 0; i < 5; i++) {
		x *= 2u - p0 * x;
	}
	img[(size_t)26 * nl + 0] = (0u - x) & ((1u << w) - 1u);
	img[(size_t)26 * nl + 1] = (uint32_t)pbits;
	img[(size_t)26 * nl + 2] = (big_cmp(big_add(a_img, three), p) == 0) ? 1u : 0u;
	img[(size_t)26 * nl + 3] = (big_bitlen(a_img) == 0) ? 1u : 0u;
	if (img.size() * 4 != ecamd_g29_image_bytes(pbits, gflavour)) {
		return fail("internal: CurveG image size mismatch");
	}
	SlotRegistry *reg = slot_registry(cv->ctx->device);
	const int key = pbits + gflavour;
	if (!reg || key >= ECAMD_G29_KEYS) {
		return fail("internal: no slot registry for this device / field size");
	}
	const int flavour = gflavour;
	const int slot = slot_acquire(reg->g[key], ecamd_g29_slots() < 8 ? ecamd_g29_slots() : 8, img,
				      [&](int sl) { return ecamd_g29_upload(pbits, sl, img.data(), img.size() * 4, flavour); });
	if (slot == -2) {
		return fail("curve: upload of the radix-2^29 curve constants failed");
	}
	*slot_out = slot < 0 ? -1 : slot;  // no free slot: the saturated-word kernels serve this handle
	return 0;
}

static int upload_g29(ecamd_curve *cv)
{
	return upload_g29_mod(cv, cv->p, cv->a, cv->b, cv->pbits, cv->gflavour, &cv->gslot);
}

// prj_pt_import_from_buf + prj_pt_unique of a batch: the handle's radix-2^29 unit when it has one (one inversion per eight triples),
// else one Fermat inversion per triple on saturated words.  ECAMD_NO_PRJ_IMPORT_G29 keeps the latter (A/B hook).
static hipError_t launch_prj_import(const ecamd_curve *cv, const EcamdPrjInArgs &I, hipStream_t s)
{
	static const bool off = getenv("ECAMD_NO_PRJ_IMPORT_G29") != nullptr;
	if (!off && cv->gslot >= 0) {
		return ecamd_g29_prj_import(cv->pbits, cv->gslot, I, s, cv->gflavour);
	}
	if (!off && cv->is_p256 && cv->gpslot >= 0) {
		return ecamd_g29_prj_import(256, cv->gpslot, I, s, 0);
	}
	return ecamd_launch_prj_import(cv->nw, I, s);
}

// the group order as the modulus of the dense unit of its size (only sizes that unit exists for; failure is not an error:
// the saturated-word k_ecdsa_prep keeps serving)
static void upload_g29_order(ecamd_curve *cv)
{
	cv->gqslot = -1;
	const int qbits = big_bitlen(cv->q);
	if (!ecamd_g29_supported(qbits) || qbits == 255 || qbits == 448 || qbits >= ECAMD_G29_KEYS || getenv("ECAMD_NO_PREP_G29") != nullptr ||
	    getenv("ECAMD_NO_FAST_PATH") != nullptr) {
		return;
	}
	int slot = -1;
	if (upload_g29_mod(cv, cv->q, Big(), Big(), qbits, 0, &slot) == 0) {
		cv->gqslot = slot;
	}
}


// device allocations of a curve handle (also used on the failure paths of its construction)
static void curve_free_device(ecamd_curve *cv)
{
	if (cv->d_gen) {
		(void)hipFree(cv->d_gen);
		cv->d_gen = nullptr;
	}
	if (cv->d_comb) {
		(void)hipFree(cv->d_comb);
		cv->d_comb = nullptr;
	}
	if (cv->d_edcomb) {
		(void)hipFree(cv->d_edcomb);
		cv->d_edcomb = nullptr;
	}
	if (cv->d_gtab) {
		(void)hipFree(cv->d_gtab);
		cv->d_gtab = nullptr;
	}
	if (cv->d_comb4) {
		(void)hipFree(cv->d_comb4);
		cv->d_comb4 = nullptr;
	}
}

// constant slots of a handle back to the device's registry (g_slot_mu NOT held by the caller)
static void curve_release_slots(ecamd_curve *cv)
{
	if (!cv->ctx) {
		return;
	}
	std::lock_guard<std::mutex> lk(g_slot_mu);
	const int dev = cv->ctx->device;
	release_modulus(dev, cv->nw, cv->slot);
	release_modulus(dev, cv->qnw, cv->qslot);
	SlotRegistry *reg = slot_registry(dev);
	const int key = cv->pbits + cv->gflavour;
	{
		const int qkey = big_bitlen(cv->q);
		if (reg && cv->gqslot >= 0 && cv->gqslot < 8 && qkey < ECAMD_G29_KEYS && reg->g[qkey][cv->gqslot].ref > 0) {
			reg->g[qkey][cv->gqslot].ref--;
		}
		cv->gqslot = -1;
	}
	if (reg && cv->gslot >= 0 && cv->gslot < 8 && key < ECAMD_G29_KEYS && reg->g[key][cv->gslot].ref > 0) {
		reg->g[key][cv->gslot].ref--;
	}
	if (reg && cv->gpslot >= 0 && cv->gpslot < 8 && reg->g[256][cv->gpslot].ref > 0) {
		reg->g[256][cv->gpslot].ref--;
	}
	cv->slot = cv->qslot = cv->gslot = cv->gqslot = cv->gpslot = -1;
}

// failure exit of curve construction: releases the handle; msg == NULL keeps the error already recorded
static int curve_abort(ecamd_curve *cv, const char *msg)
{
	curve_free_device(cv);
	curve_release_slots(cv);
	delete cv;
	return msg ? fail(msg) : -1;
}

static int curve_finish(ecamd_ctx *ctx, ecamd_curve *cv, ecamd_curve **out)
{
	cv->ctx = nullptr;
	cv->slot = cv->qslot = cv->gslot = cv->gqslot = cv->gpslot = -1;
	cv->d_gen = nullptr;
	cv->d_comb = nullptr;
	cv->d_edcomb = nullptr;
	cv->edcomb_off = false;
	cv->d_gtab = nullptr;
	cv->d_comb4 = nullptr;
	cv->comb4_off = false;
	if (!(cv->p[0] & 1) || big_bitlen(cv->p) < 160) {
		return curve_abort(cv, "curve: p must be odd and at least 160 bits");
	}
	cv->pbits = big_bitlen(cv->p);
	cv->qbits = big_bitlen(cv->q);
	cv->nw = pick_nw(cv->pbits);
	if (!cv->nw || !ecamd_nw_supported(cv->nw)) {
		return curve_abort(cv, "curve: field size not supported (max 544 bits)");
	}
	if (big_cmp(cv->a, cv->p) >= 0 || big_cmp(cv->b, cv->p) >= 0 || big_cmp(cv->gx, cv->p) >= 0 ||
	    big_cmp(cv->gy, cv->p) >= 0) {
		return curve_abort(cv, "curve: a, b, gx, gy must be < p");
	}
	cv->clen = (cv->pbits + 7) / 8;
	cv->qlen = (cv->qbits + 7) / 8;
	cv->is_p256 =
	    big_cmp(cv->p, big_from_hex("ffffffff00000001000000000000000000000000ffffffffffffffffffffffff")) == 0 &&
	    big_cmp(cv->a, big_from_hex("ffffffff00000001000000000000000000000000fffffffffffffffffffffffc")) == 0 &&
	    big_cmp(cv->b, big_from_hex("5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b")) == 0 &&
	    big_cmp(cv->order, cv->q) == 0 && getenv("ECAMD_NO_FAST_PATH") == nullptr;
	cv->ctx = ctx;
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (hipSetDevice(ctx->device) != hipSuccess) {
		return curve_abort(cv, "curve: hipSetDevice failed");
	}
	cv->qnw = big_bitlen(cv->q) <= 32 * cv->nw ? cv->nw : pick_nw(big_bitlen(cv->q));
	{
		// slot: the field of definition (modulus p, curve coefficients);  qslot: Montgomery context of the generator
		// order q for the mod-q algebra of the protocol layer (only when q is odd and fits a supported width)
		std::lock_guard<std::mutex> sl(g_slot_mu);
		if (acquire_modulus(ctx->device, cv->nw, cv->p, cv->a, cv->b, &cv->slot)) {
			cv->slot = -1;
		} else if ((cv->q[0] & 1) && cv->qnw && ecamd_nw_supported(cv->qnw)) {
			Big zero(1, 0);
			if (acquire_modulus(ctx->device, cv->qnw, cv->q, zero, zero, &cv->qslot)) {
				cv->qslot = -1;  // protocol entry points unavailable on this handle
			}
		}
	}
	if (cv->slot < 0) {
		return curve_abort(cv, nullptr);
	}
	cv->gflavour = (cv->pbits == 521 && big_cmp(big_add(cv->p, Big(1, 1)), big_pow2(521)) == 0) ? 1 : 0;
	if (cv->pbits == 255 && big_cmp(big_add(cv->p, Big(1, 19)), big_pow2(255)) == 0 && getenv("ECAMD_NO_P25519") == nullptr) {
		cv->gflavour = 2;
	}
	if (cv->pbits == 256 && getenv("ECAMD_NO_K256") == nullptr &&
	    big_cmp(cv->p, big_from_hex("fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f")) == 0) {
		cv->gflavour = 4;  // p = 2^256 - 2^32 - 977 (secp256k1): plain residues, pseudo-Mersenne folds
	}
	if (cv->pbits == 448 && getenv("ECAMD_NO_P448") == nullptr &&
	    big_cmp(cv->p, big_sub(big_sub(big_pow2(448), big_pow2(224)), Big(1, 1))) == 0) {
		cv->gflavour = 5;  // p = 2^448 - 2^224 - 1 (WEI448): plain residues, Goldilocks folds
	}
	if (getenv("ECAMD_NO_MPINV1") == nullptr) {
		// the other two NIST primes with a two-term p -+ 1: secp224r1 (2^224 - 2^96 + 1) and secp192r1 (2^192 - 2^64 - 1)
		if (cv->pbits == 224 && big_cmp(big_add(cv->p, big_pow2(96)), big_add(big_pow2(224), Big(1, 1))) == 0) {
			cv->gflavour = 6;
		}
		if (cv->pbits == 192 && big_cmp(big_add(cv->p, big_add(big_pow2(64), Big(1, 1))), big_pow2(192)) == 0) {
			cv->gflavour = 7;
		}
	}
	if (cv->pbits == 384 && getenv("ECAMD_NO_MPINV1") == nullptr &&
	    big_cmp(big_add(cv->p, big_add(big_pow2(128), big_pow2(96))), big_add(big_pow2(384), big_sub(big_pow2(32), Big(1, 1)))) == 0) {
		cv->gflavour = 3;  // secp384r1's prime: Montgomery reduction on the four signed digits of p + 1 (ecamd_u29g.h, -DG29_P384S)
	}
	if (!cv->is_p256 && ecamd_g29_supported(cv->pbits) && cv->pbits + 8 < ECAMD_G29_KEYS && getenv("ECAMD_NO_FAST_PATH") == nullptr) {
		int rc;
		{
			std::lock_guard<std::mutex> sl(g_slot_mu);
			rc = upload_g29(cv);
		}
		if (rc) {
			return curve_abort(cv, nullptr);
		}
	}
	{
		std::lock_guard<std::mutex> sl(g_slot_mu);
		upload_g29_order(cv);
		if (cv->is_p256 && getenv("ECAMD_NO_FAST_PATH") == nullptr) {
			int slot = -1;
			if (upload_g29_mod(cv, cv->p, cv->a, cv->b, 256, 0, &slot) == 0) {
				cv->gpslot = slot;   // failure is not an error: the saturated-word import keeps serving
			}
		}
	}
	// generator X || Y, then the two broadcast scalars of the subgroup / cofactor passes: q (qlen bytes) and h (1 byte)
	std::vector<uint8_t> g((size_t)2 * cv->clen + cv->qlen + 1, 0);
	big_to_be(g.data(), cv->clen, cv->gx);
	big_to_be(g.data() + cv->clen, cv->clen, cv->gy);
	big_to_be(g.data() + 2 * cv->clen, cv->qlen, cv->q);
	{
		Big t = cv->q;
		for (uint32_t c = 1; c <= 255; c++) {
			if (big_cmp(t, cv->order) == 0) {
				g[(size_t)2 * cv->clen + cv->qlen] = (uint8_t)c;   // stays 0 when order is not a small multiple of q
				cv->cofactor = c;
				break;
			}
			t = big_add(t, cv->q);
		}
	}
	if (hipMalloc((void **)&cv->d_gen, g.size()) != hipSuccess ||
	    hipMemcpy(cv->d_gen, g.data(), g.size(), hipMemcpyHostToDevice) != hipSuccess) {
		return curve_abort(cv, "curve: generator upload failed");
	}
	cv->d_gtab = nullptr;
	cv->d_comb4 = nullptr;
	cv->comb4_off = false;
	cv->d_comb = nullptr;
	cv->comb_off = false;
	cv->d_edcomb = nullptr;
	cv->edcomb_off = false;
	cv->sqrt_state = 0;
	cv->ed_state = 0;
	cv->ed448_state = 0;
	cv->ed448_err = nullptr;
	cv->xdh_state = 0;
	cv->ed_err = cv->xdh_err = nullptr;
	if (cv->is_p256) {
		// affine window table of the generator for the interleaved ECDSA verification loop:
		// [1..8]G through our own kernels, then x R mod p, y R mod p (R = 2^261) as 29-bit digits
		uint8_t sc[8 * 32], pts[8 * 64], st[8];
		memset(sc, 0, sizeof(sc));
		for (int i = 0; i < 8; i++) {
			sc[i * 32 + 31] = (uint8_t)(i + 1);
		}
		uint8_t *d = nullptr;
		if (hipMalloc((void **)&d, sizeof(sc) + sizeof(pts) + sizeof(st)) != hipSuccess) {
			return curve_abort(cv, "curve: generator table allocation failed");
		}
		if (hipMemcpy(d, sc, sizeof(sc), hipMemcpyHostToDevice) != hipSuccess) {
			(void)hipFree(d);
			return curve_abort(cv, "curve: generator table upload failed");
		}
		int rc = smul_dev_locked(ctx, cv, 8, d, 32, nullptr, d + sizeof(sc), d + sizeof(sc) + sizeof(pts), ctx->stream);
		if (!rc && (hipStreamSynchronize(ctx->stream) != hipSuccess ||
			    hipMemcpy(pts, d + sizeof(sc), sizeof(pts), hipMemcpyDeviceToHost) != hipSuccess ||
			    hipMemcpy(st, d + sizeof(sc) + sizeof(pts), sizeof(st), hipMemcpyDeviceToHost) != hipSuccess)) {
			rc = -1;
		}
		(void)hipFree(d);
		std::vector<uint32_t> tab(8 * 40, 0);
		const Big R261 = big_mod(big_pow2(261), cv->p);
		for (int i = 0; i < 8 && !rc; i++) {
			if (st[i] != 0) {
				rc = -1;
				break;
			}
			big_digits29(&tab[(size_t)i * 40], 9, big_mulmod(big_from_be(pts + i * 64, 32), R261, cv->p));
			big_digits29(&tab[(size_t)i * 40 + 9], 9, big_mulmod(big_from_be(pts + i * 64 + 32, 32), R261, cv->p));
		}
		if (rc || hipMalloc((void **)&cv->d_gtab, tab.size() * 4) != hipSuccess ||
		    hipMemcpy(cv->d_gtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
			return curve_abort(cv, "curve: generator table construction failed");
		}
		big_digits29(cv->qdig, 9, cv->q);
		// the masked 4-bit comb of the generator for secret scalars (k_p256_comb4m): [m 16^j]G, m = 1..8, j = 0..63, and [2^256]G, again
		// through our own kernels (public scalars: constants of the curve).  Not fatal when it fails: the masked window loop serves.
		if (getenv("ECAMD_NO_SECRET_COMB") == nullptr) {
			const uint32_t ne = 64 * 8 + 1;
			std::vector<uint8_t> hs((size_t)ne * 32, 0), hp((size_t)ne * 64), hst(ne, 1);
			for (uint32_t j = 0; j < 64; j++) {
				for (uint32_t m = 1; m <= 8; m++) {
					hs[((size_t)j * 8 + (m - 1)) * 32 + 31 - j / 2] = (uint8_t)(m << (4 * (j & 1)));
				}
			}
			big_to_be(&hs[(size_t)64 * 8 * 32], 32, big_mod(big_pow2(256), cv->q));
			uint8_t *dd = nullptr;
			bool ok = hipMalloc((void **)&dd, hs.size() + hp.size() + hst.size()) == hipSuccess &&
				  hipMemcpy(dd, hs.data(), hs.size(), hipMemcpyHostToDevice) == hipSuccess;
			if (ok) {
				PublicScalars pub_scope(ctx);
				ok = smul_dev_locked(ctx, cv, ne, dd, 32, nullptr, dd + hs.size(), dd + hs.size() + hp.size(), ctx->stream) == 0 &&
				     hipStreamSynchronize(ctx->stream) == hipSuccess &&
				     hipMemcpy(hp.data(), dd + hs.size(), hp.size(), hipMemcpyDeviceToHost) == hipSuccess &&
				     hipMemcpy(hst.data(), dd + hs.size() + hp.size(), hst.size(), hipMemcpyDeviceToHost) == hipSuccess;
			}
			if (dd) {
				(void)hipFree(dd);
			}
			std::vector<uint32_t> tab4((size_t)65 * 8 * 40, 0);
			for (uint32_t e = 0; ok && e < ne; e++) {
				ok = hst[e] == 0;
				big_digits29(&tab4[(size_t)e * 40], 9, big_mulmod(big_from_be(&hp[(size_t)e * 64], 32), R261, cv->p));
				big_digits29(&tab4[(size_t)e * 40 + 9], 9, big_mulmod(big_from_be(&hp[(size_t)e * 64 + 32], 32), R261, cv->p));
			}
			if (ok && hipMalloc((void **)&cv->d_comb4, tab4.size() * 4) == hipSuccess &&
			    hipMemcpy(cv->d_comb4, tab4.data(), tab4.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
				(void)hipFree(cv->d_comb4);
				cv->d_comb4 = nullptr;
			}
			(void)hipGetLastError();
		}
	}
	*out = cv;
	return 0;
}

extern "C" int ecamd_curve_by_name(ecamd_ctx *ctx, const char *name, ecamd_curve **out)
{
	if (!ctx || !name || !out) {
		return fail("ecamd_curve_by_name: NULL argument");
	}
	*out = nullptr;
	for (const CurveRow &r : g_curve_rows) {
		if (strcasecmp(r.name, name) == 0) {
			ecamd_curve *cv = new ecamd_curve();
			cv->curve_type = r.type;
			cv->p = big_from_hex(r.p);
			cv->a = big_from_hex(r.a);
			cv->b = big_from_hex(r.b);
			cv->order = big_from_hex(r.order);
			cv->gx = big_from_hex(r.gx);
			cv->gy = big_from_hex(r.gy);
			cv->q = big_from_hex(r.q);
			return curve_finish(ctx, cv, out);
		}
	}
	return fail(std::string("ecamd_curve_by_name: unknown curve ") + name);
}

extern "C" int ecamd_curve_from_params(ecamd_ctx *ctx, const uint8_t *p, uint32_t p_len, const uint8_t *a,
				       uint32_t a_len, const uint8_t *b, uint32_t b_len,
				       const uint8_t *curve_order, uint32_t curve_order_len,
				       const uint8_t *gx, uint32_t gx_len, const uint8_t *gy, uint32_t gy_len,
				       const uint8_t *gen_order, uint32_t gen_order_len, ecamd_curve **out)
{
	if (!ctx || !p || !a || !b || !curve_order || !gx || !gy || !gen_order || !out) {
		return fail("ecamd_curve_from_params: NULL argument");
	}
	*out = nullptr;
	ecamd_curve *cv = new ecamd_curve();
	cv->p = big_from_be(p, p_len);
	cv->a = big_from_be(a, a_len);
	cv->b = big_from_be(b, b_len);
	cv->order = big_from_be(curve_order, curve_order_len);
	cv->gx = big_from_be(gx, gx_len);
	cv->gy = big_from_be(gy, gy_len);
	cv->q = big_from_be(gen_order, gen_order_len);
	return curve_finish(ctx, cv, out);
}

extern "C" void ecamd_curve_free(ecamd_curve *cv)
{
	if (!cv) {
		return;
	}
	{
		std::lock_guard<std::mutex> lk(cv->ctx->mu);
		(void)hipSetDevice(cv->ctx->device);
		// kernels of this handle may still be in flight on the context's stream: let them finish before its constant
		// slots can be handed to another curve
		(void)hipStreamSynchronize(cv->ctx->stream);
		curve_free_device(cv);
	}
	curve_release_slots(cv);
	delete cv;
}

extern "C" int ecamd_curve_coord_len(const ecamd_curve *cv) { return cv ? cv->clen : -1; }
extern "C" int ecamd_curve_order_len(const ecamd_curve *cv) { return cv ? cv->qlen : -1; }
extern "C" int ecamd_curve_words(const ecamd_curve *cv) { return cv ? cv->nw : -1; }

// ------------------------------------------------------------------------------------------
// batched prj_pt_mul
// ------------------------------------------------------------------------------------------
// secp256r1 fast path scratch (ecamd_p256_kernel.hip): 8 x 64-byte affine records + 70 x 16-byte staging quads per item
#define P256_TAB_BYTES 512u
#define P256_SCRATCH_PER_ITEM ((size_t)P256_TAB_BYTES + 70u * 16u)

static size_t tbl_bytes_for(const ecamd_curve *cv, uint32_t stride)
{
	return (size_t)ECAMD_TBL_ENTRIES * 3 * (size_t)cv->nw * 4 * (size_t)stride;
}

// 16-bit comb table of the generator: T[j][m-1] = [m 2^(16 j)]G (m = 1..32768, j over every 16-bit window of
// the longest scalar the fast path takes) plus [2^(8 slen)]G, computed by this engine itself through the
// window path and converted to the kernels' field representation on the device.  42 MB (256-bit curves) to
// 183 MB (521 bits) of the 288 GB per curve handle; turns [k]G into 2 NW + 1 additions.  ctx->mu held.
static void maybe_build_comb(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n)
{
	if (cv->d_comb || cv->comb_off || ctx->comb_min_batch == 0 || n < ctx->comb_min_batch) {
		return;
	}
	const bool p256 = cv->is_p256;
	if (!p256 && cv->gslot < 0) {
		return;
	}
	cv->comb_off = true;  // no recursion while building; stays set if anything fails
	// the build reuses the context's scratch tables on the context's stream: let kernels the caller has in
	// flight on other streams (earlier stages of the same entry point) finish first -- a one-time event
	if (hipDeviceSynchronize() != hipSuccess) {
		return;
	}
	const uint32_t nwords = (uint32_t)((cv->pbits + 31) / 32);
	const uint32_t slen = 4 * nwords, nwin = 2 * nwords;
	const uint32_t ne = nwin * 32768u + 1u;
	const uint32_t ew = p256 ? 20u : ecamd_g29_comb_entry_words(cv->pbits, cv->gflavour);
	if (!p256 && ne != ecamd_g29_comb_entries(cv->pbits)) {
		return;
	}
	std::vector<uint8_t> hs((size_t)ne * slen, 0);
	for (uint32_t j = 0; j < nwin; j++) {
		for (uint32_t m = 1; m <= 32768; m++) {
			uint8_t *e = &hs[((size_t)j * 32768 + (m - 1)) * slen];
			e[slen - 1 - 2 * j] = (uint8_t)(m & 0xff);
			e[slen - 2 - 2 * j] = (uint8_t)(m >> 8);
		}
	}
	big_to_be(&hs[(size_t)nwin * 32768 * slen], (int)slen, big_mod(big_pow2(8 * (int)slen), cv->q));
	const size_t plen = (size_t)2 * cv->clen;
	uint8_t *dsc = nullptr, *dpt = nullptr, *dst = nullptr;
	uint32_t *table = nullptr;
	std::vector<uint8_t> st(ne);
	hipStream_t s = ctx->stream;
	bool ok = hipMalloc((void **)&dsc, hs.size()) == hipSuccess && hipMalloc((void **)&dpt, (size_t)ne * plen) == hipSuccess &&
		  hipMalloc((void **)&dst, ne) == hipSuccess && hipMalloc((void **)&table, (size_t)ne * ew * 4) == hipSuccess &&
		  hipMemcpy(dsc, hs.data(), hs.size(), hipMemcpyHostToDevice) == hipSuccess;
	{
		// the build is one big batch of its own: do not let a small user chunk turn it into thousands of launches
		const uint32_t user_chunk = ctx->max_chunk;
		ctx->max_chunk = user_chunk < (1u << 20) ? (1u << 20) : user_chunk;
		ok = ok && smul_dev_locked(ctx, cv, ne, dsc, slen, nullptr, dpt, dst, s) == 0;
		ctx->max_chunk = user_chunk;
	}
	if (ok) {
		const hipError_t e = p256 ? ecamd_launch_comb_build_p256(dpt, ne, table, s)
					  : ecamd_g29_comb_build(cv->pbits, cv->gslot, dpt, ne, (uint32_t)cv->clen, table, s, cv->gflavour);
		ok = e == hipSuccess && hipMemcpyAsync(st.data(), dst, ne, hipMemcpyDeviceToHost, s) == hipSuccess &&
		     hipStreamSynchronize(s) == hipSuccess;
	}
	for (uint32_t i = 0; ok && i < ne; i++) {
		ok = (st[i] == 0);
	}
	(void)hipFree(dsc);
	(void)hipFree(dpt);
	(void)hipFree(dst);
	if (!ok) {
		(void)hipFree(table);
		(void)hipGetLastError();
		return;  // the window path keeps serving fixed-base calls
	}
	cv->d_comb = table;
	cv->comb_off = false;
}

// The generator's 4-bit comb for SECRET scalars on a radix-2^29 unit (k_comb_g<.., SCAN4>): [m 16^j]G, m = 1..8, j = 0 .. 8 NW - 1, and
// [2^(32 NW)]G, (8 NW + 1) x 8 entries of the unit's comb-entry format, built on the first secret fixed-base call of the handle through
// the library's own (public-scalar) multiplication.  Failure is not an error: the masked window loop keeps serving.
static void maybe_build_comb4(ecamd_ctx *ctx, ecamd_curve *cv)
{
	static const bool off = getenv("ECAMD_NO_SECRET_COMB") != nullptr;
	if (off || cv->d_comb4 || cv->comb4_off || cv->is_p256 || cv->gslot < 0) {
		return;
	}
	cv->comb4_off = true;   // one attempt
	if (hipDeviceSynchronize() != hipSuccess) {
		return;
	}
	const uint32_t nwords = (uint32_t)((cv->pbits + 31) / 32);
	const uint32_t slen = 4 * nwords, nwin = 8 * nwords, ne = nwin * 8u + 1u;
	const uint32_t ew = ecamd_g29_comb_entry_words(cv->pbits, cv->gflavour);
	if (slen > ecamd_g29_comb_max_slen(cv->pbits)) {
		return;
	}
	std::vector<uint8_t> hs((size_t)ne * slen, 0), st(ne, 1);
	for (uint32_t j = 0; j < nwin; j++) {
		for (uint32_t m = 1; m <= 8; m++) {
			hs[((size_t)j * 8 + (m - 1)) * slen + slen - 1 - j / 2] = (uint8_t)(m << (4 * (j & 1)));
		}
	}
	big_to_be(&hs[(size_t)nwin * 8 * slen], (int)slen, big_mod(big_pow2(8 * (int)slen), cv->q));
	const size_t plen = (size_t)2 * cv->clen;
	uint8_t *dsc = nullptr, *dpt = nullptr, *dst = nullptr;
	uint32_t *table = nullptr;
	hipStream_t s = ctx->stream;
	bool ok = hipMalloc((void **)&dsc, hs.size()) == hipSuccess && hipMalloc((void **)&dpt, (size_t)ne * plen) == hipSuccess &&
		  hipMalloc((void **)&dst, ne) == hipSuccess && hipMalloc((void **)&table, (size_t)(nwin + 1) * 8 * ew * 4) == hipSuccess &&
		  hipMemset(table, 0, (size_t)(nwin + 1) * 8 * ew * 4) == hipSuccess &&
		  hipMemcpy(dsc, hs.data(), hs.size(), hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		PublicScalars pub_scope(ctx);   // constants of the curve
		ok = smul_dev_locked(ctx, cv, ne, dsc, slen, nullptr, dpt, dst, s) == 0 &&
		     ecamd_g29_comb_build(cv->pbits, cv->gslot, dpt, ne, (uint32_t)cv->clen, table, s, cv->gflavour) == hipSuccess &&
		     hipMemcpyAsync(st.data(), dst, ne, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
	}
	for (uint32_t i = 0; ok && i < ne; i++) {
		ok = (st[i] == 0);
	}
	(void)hipFree(dsc);
	(void)hipFree(dpt);
	(void)hipFree(dst);
	if (!ok) {
		(void)hipFree(table);
		(void)hipGetLastError();
		return;
	}
	cv->d_comb4 = table;
}

// sstride = slen normally; 0 broadcasts one scalar to every item (subgroup / cofactor passes)
static int smul_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_scalars,
			   uint32_t slen, const uint8_t *d_points, uint8_t *d_out, uint8_t *d_status,
			   hipStream_t s, uint32_t sstride, bool redo_only, const uint8_t *d_scalars2)
{
	// d_scalars2 (generic radix-2^29 units with a comb table, see fused_verify_ok): out = [scalars]P + [scalars2]G by the fused
	// window loop; items that met an exceptional pair keep ECAMD_STATUS_REDO in d_status for the caller (no complete-formula pass here)
	// redo_only: d_status is given; only the items marked ECAMD_STATUS_REDO in it are computed (complete-formula kernel)
	if (sstride == 0xffffffffu) {
		sstride = slen;
	}
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	const uint32_t stride = (chunk + 63u) & ~63u;
	// secp256r1: the window loop takes scalars of up to 68 bytes (blinded scalars m + b #E of about 2 |q| bits stay on the fast
	// kernel); the fixed-base comb covers the 32-byte ones
	// secret-scalar mode: every item goes through the complete-formula kernel with masked full-table look-ups -- no digit-indexed
	// address (window or comb table), no exceptional-pair detour whose occurrence depends on the scalar
	// (secp256r1 keeps its radix-2^29 pipeline in this mode: k_p256_loop<KW, MASKED> scans the eight-entry tables, the comb is off)
	const bool secret = ctx->secret_scalars;
	const bool fast256 = cv->is_p256 && slen <= 68;
	const bool fastg = !fast256 && cv->gslot >= 0 && slen <= ecamd_g29_max_slen(cv->pbits);
	const bool comb_ok = !secret && (cv->is_p256 ? slen <= 32 : slen <= ecamd_g29_comb_max_slen(cv->pbits));
	const bool fast = !redo_only && (fast256 || fastg);
	if (fast && !d_points && comb_ok) {
		maybe_build_comb(ctx, const_cast<ecamd_curve *>(cv), n);
	}
	if (fast && fastg && !d_points && secret && n >= 64) {
		maybe_build_comb4(ctx, const_cast<ecamd_curve *>(cv));
	}
	{
		uint8_t *t = (uint8_t *)ctx->tbl;
		const int rc = ensure(&t, &ctx->tbl_bytes, tbl_bytes_for(cv, stride));
		ctx->tbl = (uint32_t *)t;
		if (rc) {
			return -1;
		}
	}
	if (fast) {
		uint8_t *t = (uint8_t *)ctx->tbl_fast;
		const size_t per_item = fast256 ? P256_SCRATCH_PER_ITEM
						: ((size_t)ecamd_g29_table_words(cv->pbits, cv->gflavour) + ecamd_g29_affine_words(cv->pbits, cv->gflavour)) * 4;
		const int rc = ensure(&t, &ctx->tbl_fast_bytes, (size_t)stride * per_item);
		ctx->tbl_fast = (uint32_t *)t;
		if (rc) {
			return -1;
		}
	}
	for (uint32_t off = 0; off < n; off += chunk) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		EcamdSmulArgs A;
		A.scalars = d_scalars + (size_t)off * sstride;
		A.sstride = sstride;
		A.pstride = d_points ? 2u * (uint32_t)cv->clen : 0u;
		A.points = d_points ? d_points + (size_t)off * 2 * cv->clen : cv->d_gen;
		A.out = d_out + (size_t)off * 2 * cv->clen;
		A.status = d_status + off;
		A.tbl = ctx->tbl;
		A.n = m;
		A.slen = slen;
		A.clen = (uint32_t)cv->clen;
		A.stride = stride;
		A.slot = cv->slot;
		A.only_redo = redo_only ? 1 : 0;
		A.lut = nullptr;
		A.lut_kind = 0;
		A.stg = nullptr;
		A.masked = secret ? 1 : 0;   // (copied into Fa below: the secp256r1 loop honours it too)
		A.scalars2 = nullptr;
		A.s2len = 0;
		if (d_scalars2) {
			if (!fastg || !d_points || !cv->d_comb) {
				return fail("internal: fused double-scalar loop requested without its preconditions");
			}
			EcamdSmulArgs Fa = A;
			Fa.tbl = ctx->tbl_fast;
			Fa.stg = ctx->tbl_fast + (size_t)stride * (size_t)ecamd_g29_table_words(cv->pbits, cv->gflavour);
			Fa.lut = cv->d_comb;
			Fa.lut_kind = 2u;
			Fa.scalars2 = d_scalars2 + (size_t)off * slen;
			Fa.s2len = slen;
			Fa.masked = 0;
			hipEvent_t *ev = (ctx->timing && off == 0) ? ctx->ev : nullptr;
			HIPCHK(ecamd_launch_smul_g29(cv->pbits, cv->gslot, Fa, s, ev, cv->gflavour));
			ctx->ev_valid = ctx->ev_valid || (ev != nullptr);
			continue;
		}
		if (fast) {
			// Jacobian fast path; lanes that met an exceptional pair come back as ECAMD_STATUS_REDO and
			// are recomputed by the complete-formula kernel (all other lanes exit at once)
			EcamdSmulArgs Fa = A;
			Fa.tbl = ctx->tbl_fast;
			Fa.stg = ctx->tbl_fast + (size_t)stride * (fast256 ? (size_t)(P256_TAB_BYTES / 4) : (size_t)ecamd_g29_table_words(cv->pbits, cv->gflavour));
			// fixed base: a constant table of the generator replaces the per-item table kernels
			if (!d_points) {
				const bool use_comb = cv->d_comb && comb_ok;
				Fa.lut = use_comb ? cv->d_comb : (fast256 ? cv->d_gtab : nullptr);
				Fa.lut_kind = use_comb ? 1u : 0u;
				if (secret && cv->d_comb4 && (fast256 ? slen <= 32 : slen <= 4 * (uint32_t)((cv->pbits + 31) / 32))) {
					Fa.lut = cv->d_comb4;   // secret scalars: the scanned 4-bit comb (8 NW + 1 additions) instead of the scanned window loop
					Fa.lut_kind = 3u;
				}
			}
			hipEvent_t *ev = (ctx->timing && off == 0) ? ctx->ev : nullptr;  // first chunk of the call
			if (fast256) {
				HIPCHK(ecamd_launch_smul_p256(Fa, s, ev));
			} else {
				HIPCHK(ecamd_launch_smul_g29(cv->pbits, cv->gslot, Fa, s, ev, cv->gflavour));
			}
			ctx->ev_valid = ctx->ev_valid || (ev != nullptr);
			A.only_redo = 1;
		}
		HIPCHK(ecamd_launch_smul(cv->nw, A, s));
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// Host-pointer entry points = the device-pointer cores behind copies of the caller's arrays.  The batch goes
// through in chunks of ctx->host_chunk items with double-buffered device staging: while the kernels of chunk
// c run on the compute stream, the host thread is inside the (pageable, hence host-blocking) copy of chunk
// c+1 on the copy stream, so PCIe and the GPU work concurrently.  `ins` / `outs`: the caller's arrays with
// their per-item strides (a NULL input is passed on as NULL); core(m, in_ptrs, out_ptrs, stream) enqueues one
// chunk and may synchronise (after calling `between`).  ctx->mu held.
// ------------------------------------------------------------------------------------------
struct HostArr {
	const uint8_t *in;   // input array (or NULL)
	uint8_t *out;        // output array (or NULL)
	size_t stride;       // bytes per item
};

template <class Core>
static int host_pipeline(ecamd_ctx *ctx, int pbits, uint32_t n, const std::vector<HostArr> &arrs, Core core, uint32_t even_chunk = 0)
{
	// even_chunk != 0 (the whole-batch forms that file a batch chunk by chunk: their per-chunk kernels are light, what matters is that the LAST
	// chunk -- whose copy nothing hides -- is small): a short first chunk, then equal chunks of that size
	const uint32_t chunk = even_chunk && even_chunk < ctx->host_chunk ? (n < even_chunk ? n : even_chunk) : (n < ctx->host_chunk ? n : ctx->host_chunk);
	const size_t na = arrs.size();
	if (na > 6) {
		return fail("internal: too many host arrays");
	}
	// The chunk schedule.  Nothing overlaps the copy of the FIRST chunk (and, with a producer hook, its packing), and the copy of chunk
	// c + 1 hides behind the kernels of chunk c only when it is not much larger than c: so a batch of several chunks starts short -- 2^16
	// items ($ECAMD_HOST_RAMP_MIN) -- and DOUBLES up to the chunk size (round 6: with an eighth of a chunk followed at once by a full
	// chunk the device sat idle for 1.7 ms of a 2^20-item verification, waiting for 111 MB to arrive behind 1.4 ms of kernels;
	// profiles/r6c_typed_boundary.md), and a remainder of up to a chunk and a quarter goes as ONE launch (2^20 items: 2^16, 2^17, 2^18,
	// 589 824 = three full rounds of the window kernels' 3 072 resident waves).  $ECAMD_HOST_SCHEDULE=a,b,c: explicit leading chunk
	// sizes (tests, measurements); $ECAMD_NO_HOST_RAMP: equal chunks.
	static const bool no_ramp = getenv("ECAMD_NO_HOST_RAMP") != nullptr;
	std::vector<uint32_t> sched;
	{
		uint32_t left = n;
		if (const char *e = getenv("ECAMD_HOST_SCHEDULE")) {
			while (*e && left) {
				uint32_t v = (uint32_t)strtoul(e, nullptr, 10);
				v = v < 1 ? 1 : (v > left ? left : v);
				sched.push_back(v);
				left -= v;
				e = strchr(e, ',');
				if (!e) {
					break;
				}
				e++;
			}
		} else if (even_chunk && n > chunk) {
			const uint32_t m = ctx->host_first_min < chunk ? ctx->host_first_min : chunk;
			if (m < chunk && left > m) {
				sched.push_back(m);
				left -= m;
			}
		} else if (n > chunk && !no_ramp && pbits <= 256) {
			for (uint32_t m = ctx->host_first_min < chunk ? ctx->host_first_min : chunk; m < chunk && left > m + m; m += m) {
				sched.push_back(m);
				left -= m;
			}
		} else if (n > chunk && !no_ramp) {
			// fields above 256 bits: the kernels of a 2^16-item chunk outlast the copy of a full one (secp384r1: 4 ms against 2.7), and
			// their window loops lose more to short launches than the doubling wins (measured: 60.6 against 63.6 ms per 2^20
			// verifications) -- one short chunk, then full ones
			const uint32_t m = chunk / 8 < ctx->host_first_min ? ctx->host_first_min : chunk / 8;
			if (m < chunk) {
				sched.push_back(m);
				left -= m;
			}
		}
		while (left) {
			const uint32_t m = (left <= chunk + chunk / 4) ? left : chunk;
			sched.push_back(m);
			left -= m;
		}
	}
	uint32_t largest = 0;
	for (uint32_t m : sched) {
		largest = m > largest ? m : largest;
	}
	const int nbuf = sched.size() > 1 ? 2 : 1;
	for (int b = 0; b < nbuf; b++) {
		for (size_t k = 0; k < na; k++) {
			if (ensure(&ctx->hbuf[b][k], &ctx->hbuf_bytes[b][k], (size_t)largest * arrs[k].stride)) {
				return -1;
			}
		}
	}
	hipStream_t cs = ctx->copy_stream, s = ctx->stream;
	StreamScope scope(ctx, s);
	auto copy_in = [&](uint32_t off, uint32_t m, int b) -> int {
		if (ctx->ready_fn) {
			ctx->ready_fn(ctx->ready_arg, off, m);   // the producer of the arrays completes [off, off + m) first
		}
		for (size_t k = 0; k < na; k++) {
			if (arrs[k].in) {
				HIPCHK(hipMemcpyAsync(ctx->hbuf[b][k], arrs[k].in + (size_t)off * arrs[k].stride, (size_t)m * arrs[k].stride,
						      hipMemcpyHostToDevice, cs));
			}
		}
		HIPCHK(hipEventRecord(ctx->in_ready[b], cs));
		return 0;
	};
	if (copy_in(0, sched[0], 0)) {
		return -1;
	}
	int b = 0;
	uint32_t off = 0;
	for (size_t ci = 0; ci < sched.size(); off += sched[ci], ci++, b ^= (nbuf - 1)) {
		const uint32_t m = sched[ci];
		std::vector<const uint8_t *> ip(na, nullptr);
		std::vector<uint8_t *> op(na, nullptr);
		for (size_t k = 0; k < na; k++) {
			ip[k] = arrs[k].in ? ctx->hbuf[b][k] : nullptr;
			op[k] = arrs[k].out ? ctx->hbuf[b][k] : nullptr;
		}
		HIPCHK(hipStreamWaitEvent(s, ctx->in_ready[b], 0));
		// `between`: called once this chunk's kernels are enqueued and before anything waits for them -- by the
		// core itself if it has to synchronise (ECDSA re-checks exceptional items), otherwise right after it
		bool next_issued = false;
		const uint32_t noff = off + m;
		const std::function<int()> between = [&]() -> int {
			if (next_issued || ci + 1 >= sched.size()) {
				return 0;
			}
			next_issued = true;
			// the other staging set is free: its chunk was drained at the end of the previous iteration
			return copy_in(noff, sched[ci + 1], b ^ 1);
		};
		if (core(m, ip, op, s, between) || between()) {
			// leave nothing in flight on the staging buffers of a failed call
			(void)hipStreamSynchronize(cs);
			(void)hipStreamSynchronize(s);
			return -1;
		}
		for (size_t k = 0; k < na; k++) {
			if (arrs[k].out) {
				HIPCHK(hipMemcpyAsync(arrs[k].out + (size_t)off * arrs[k].stride, ctx->hbuf[b][k], (size_t)m * arrs[k].stride,
						      hipMemcpyDeviceToHost, s));
			}
		}
		HIPCHK(hipStreamSynchronize(s));
	}
	return 0;
}

extern "C" int ecamd_ctx_set_host_ready_hook(ecamd_ctx *ctx, ecamd_host_ready_fn fn, void *arg)
{
	if (!ctx) {
		return fail("ecamd_ctx_set_host_ready_hook: NULL context");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	ctx->ready_fn = fn;
	ctx->ready_arg = fn ? arg : nullptr;
	return 0;
}

extern "C" int ec_prj_pt_mul_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n,
				       const void *d_scalars, uint32_t slen, const void *d_points,
				       void *d_out, void *d_status, void *hip_stream)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!d_scalars || !d_out || !d_status))) {
		return fail("ec_prj_pt_mul_batch_dev: bad argument");
	}
	if (slen == 0 || slen > 1024) {
		return fail("ec_prj_pt_mul_batch_dev: scalar_len must be in 1..1024");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	return smul_dev_locked(ctx, cv, n, (const uint8_t *)d_scalars, slen, (const uint8_t *)d_points,
			       (uint8_t *)d_out, (uint8_t *)d_status, s);
}

extern "C" int ec_prj_pt_mul_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *scalars,
				   uint32_t slen, const uint8_t *points, uint8_t *out, uint8_t *status)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!scalars || !out || !status))) {
		return fail("ec_prj_pt_mul_batch: bad argument");
	}
	if (slen == 0 || slen > 1024) {
		return fail("ec_prj_pt_mul_batch: scalar_len must be in 1..1024");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t plen = (size_t)2 * cv->clen;
	const std::vector<HostArr> arrs = {{scalars, nullptr, slen}, {points, nullptr, plen}, {nullptr, out, plen}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return smul_dev_locked(ctx, cv, m, ip[0], slen, ip[1], op[2], op[3], s);
	});
}

// prj_pt_mul_blind in batch: the scalar multiplied is m + b #E (curves/prj_pt.c:1782-1822), b supplied by the caller
extern "C" int ec_prj_pt_mul_blind_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *scalars, uint32_t slen,
					 const uint8_t *blinds, uint32_t blen, const uint8_t *points, uint8_t *out, uint8_t *status)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!scalars || !blinds || !out || !status))) {
		return fail("ec_prj_pt_mul_blind_batch: bad argument");
	}
	if (slen == 0 || slen > 128 || blen == 0 || blen > 72) {
		return fail("ec_prj_pt_mul_blind_batch: scalar_len must be in 1..128, blind_len in 1..72");
	}
	if (n == 0) {
		return 0;
	}
	const uint32_t ow = (uint32_t)((big_bitlen(cv->order) + 31) / 32);
	uint32_t outlen = blen + 4 * ow;
	outlen = (outlen > slen ? outlen : slen) + 1;
	if (ow > 18 || (outlen + 3) / 4 > 72) {
		return fail("ec_prj_pt_mul_blind_batch: blinded scalar too long");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t plen = (size_t)2 * cv->clen;
	// stage: 12 blinded scalars, 13 "bad blind" flags (beside the host pipeline's own buffers)
	const std::vector<HostArr> arrs = {{scalars, nullptr, slen}, {blinds, nullptr, blen}, {points, nullptr, plen}, {nullptr, out, plen}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op, hipStream_t s,
					       const std::function<int()> &) {
		if (ensure(&ctx->stage[12], &ctx->stage_bytes[12], (size_t)m * outlen) || ensure(&ctx->stage[13], &ctx->stage_bytes[13], m)) {
			return -1;
		}
		EcamdBlindArgs B;
		B.m = ip[0];
		B.b = ip[1];
		B.out = ctx->stage[12];
		B.bad = ctx->stage[13];
		B.n = m;
		B.mlen = slen;
		B.blen = blen;
		B.outlen = outlen;
		B.owords = ow;
		for (uint32_t w = 0; w < 18; w++) {
			B.order[w] = w < cv->order.size() ? cv->order[w] : 0u;
		}
		HIPCHK(ecamd_launch_blind_scalar(B, s));
		if (smul_dev_locked(ctx, cv, m, ctx->stage[12], outlen, ip[2], op[3], op[4], s)) {
			return -1;
		}
		// a blind outside [1, #E) is not something the reference would have drawn: report it as an error of the item
		HIPCHK(ecamd_launch_status_or(op[4], ctx->stage[13], op[3], (uint32_t)plen, m, s));
		return 0;
	});
}

// ------------------------------------------------------------------------------------------
// batched prj_pt_add / prj_pt_dbl
// ------------------------------------------------------------------------------------------
static int pt_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *p1, const uint8_t *p2,
		    uint8_t *out, uint8_t *status, int dbl)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!p1 || (!dbl && !p2) || !out || !status))) {
		return fail("ec_prj_pt_add/dbl_batch: bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t plen = (size_t)2 * cv->clen;
	if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)n * plen) ||
	    ensure(&ctx->stage[1], &ctx->stage_bytes[1], (size_t)n * plen) ||
	    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)n * plen) ||
	    ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)n)) {
		return -1;
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	HIPCHK(hipMemcpyAsync(ctx->stage[0], p1, (size_t)n * plen, hipMemcpyHostToDevice, s));
	if (!dbl) {
		HIPCHK(hipMemcpyAsync(ctx->stage[1], p2, (size_t)n * plen, hipMemcpyHostToDevice, s));
	}
	EcamdPtArgs A;
	A.p1 = ctx->stage[0];
	A.p2 = ctx->stage[1];
	A.out = ctx->stage[2];
	A.status = ctx->stage[3];
	A.n = n;
	A.clen = (uint32_t)cv->clen;
	A.dbl = dbl;
	A.slot = cv->slot;
	HIPCHK(ecamd_launch_pt(cv->nw, A, s));
	HIPCHK(hipMemcpyAsync(out, ctx->stage[2], (size_t)n * plen, hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(status, ctx->stage[3], (size_t)n, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

extern "C" int ec_prj_pt_add_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *p1,
				   const uint8_t *p2, uint8_t *out, uint8_t *status)
{
	return pt_batch(ctx, cv, n, p1, p2, out, status, 0);
}

extern "C" int ec_prj_pt_dbl_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *p,
				   uint8_t *out, uint8_t *status)
{
	return pt_batch(ctx, cv, n, p, nullptr, out, status, 1);
}

// ------------------------------------------------------------------------------------------
// Round 4: prj_pt_add / prj_pt_dbl / prj_pt_is_on_curve in either wire format (k_ptf), and _prj_pt_unprotected_mult /
// check_prj_pt_order statement for statement (k_unprot) -- SURVEY.md 8a rows a17, a18, a21, a22 as callable batch operations
// ------------------------------------------------------------------------------------------
extern "C" int ec_prj_pt_op_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *cv, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2,
				      int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	const bool two = op == ECAMD_PT_OP_ADD || op == ECAMD_PT_OP_CMP || op == ECAMD_PT_OP_EQ_OR_OPP;
	if (!ctx || !cv || cv->ctx != ctx || op < 0 || op > ECAMD_PT_OP_EQ_OR_OPP || (in_fmt != 0 && in_fmt != 1) ||
	    (out_fmt != 0 && out_fmt != 1) || (n && (!p1 || (two && !p2) || (op != ECAMD_PT_OP_ON_CURVE && !out) || !status))) {
		return fail("ec_prj_pt_op_batch_fmt: bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	// the two predicates hand back one byte per item, the other operations a point
	const size_t iw = (size_t)(in_fmt ? 3 : 2) * cv->clen, ow = op >= ECAMD_PT_OP_CMP ? 1 : (size_t)(out_fmt ? 3 : 2) * cv->clen;
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
		const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
		if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)m * iw) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], (size_t)m * iw) ||
		    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)m * ow) || ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)m)) {
			return -1;
		}
		HIPCHK(hipMemcpyAsync(ctx->stage[0], p1 + (size_t)off * iw, (size_t)m * iw, hipMemcpyHostToDevice, s));
		if (two) {
			HIPCHK(hipMemcpyAsync(ctx->stage[1], p2 + (size_t)off * iw, (size_t)m * iw, hipMemcpyHostToDevice, s));
		}
		EcamdPtfArgs A;
		memset(&A, 0, sizeof(A));
		A.p1 = ctx->stage[0];
		A.p2 = ctx->stage[1];
		A.out = ctx->stage[2];
		A.status = ctx->stage[3];
		A.n = m;
		A.clen = (uint32_t)cv->clen;
		A.op = op;
		A.in_fmt = in_fmt;
		A.out_fmt = out_fmt;
		A.slot = cv->slot;
		HIPCHK(ecamd_launch_ptf(cv->nw, A, s));
		if (op != 2) {
			HIPCHK(hipMemcpyAsync(out + (size_t)off * ow, ctx->stage[2], (size_t)m * ow, hipMemcpyDeviceToHost, s));
		}
		HIPCHK(hipMemcpyAsync(status + off, ctx->stage[3], (size_t)m, hipMemcpyDeviceToHost, s));
		HIPCHK(hipStreamSynchronize(s));
	}
	return 0;
}

extern "C" int ec_prj_pt_unprotected_mult_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *scalars,
						uint32_t scalar_len, uint32_t scalar_stride, const uint8_t *points, int in_fmt, uint8_t *out,
						int out_fmt, uint8_t *status)
{
	if (!ctx || !cv || cv->ctx != ctx || (in_fmt != 0 && in_fmt != 1) || (out_fmt != 0 && out_fmt != 1) || scalar_len == 0 ||
	    scalar_len > 4096 || (scalar_stride != 0 && scalar_stride != scalar_len) || (n && (!scalars || !points || !out || !status))) {
		return fail("ec_prj_pt_unprotected_mult_batch: bad argument (scalar_stride is scalar_len, or 0 for one scalar for every item)");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t iw = (size_t)(in_fmt ? 3 : 2) * cv->clen, ow = (size_t)(out_fmt ? 3 : 2) * cv->clen;
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
		const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
		const size_t sbytes = scalar_stride ? (size_t)m * scalar_len : (size_t)scalar_len;
		if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)m * iw) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], sbytes) ||
		    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)m * ow) || ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)m)) {
			return -1;
		}
		HIPCHK(hipMemcpyAsync(ctx->stage[0], points + (size_t)off * iw, (size_t)m * iw, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[1], scalars + (size_t)off * scalar_stride, sbytes, hipMemcpyHostToDevice, s));
		EcamdUnprotArgs A;
		memset(&A, 0, sizeof(A));
		A.points = ctx->stage[0];
		A.scalars = ctx->stage[1];
		A.out = ctx->stage[2];
		A.status = ctx->stage[3];
		A.n = m;
		A.clen = (uint32_t)cv->clen;
		A.slen = scalar_len;
		A.sstride = scalar_stride;
		A.in_fmt = in_fmt;
		A.out_fmt = out_fmt;
		A.slot = cv->slot;
		HIPCHK(ecamd_launch_unprot(cv->nw, A, s));
		HIPCHK(hipMemcpyAsync(out + (size_t)off * ow, ctx->stage[2], (size_t)m * ow, hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(status + off, ctx->stage[3], (size_t)m, hipMemcpyDeviceToHost, s));
		HIPCHK(hipStreamSynchronize(s));
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// batched field ops on libecc's 64-bit limb layout
// ------------------------------------------------------------------------------------------
extern "C" int ec_fp_op_batch(ecamd_ctx *ctx, const ecamd_curve *cv, int op, uint32_t n, const uint64_t *a,
			      const uint64_t *b, uint64_t *out)
{
	if (!ctx || !cv || cv->ctx != ctx || op < 0 || op > 4 || (n && (!a || !b || !out))) {
		return fail("ec_fp_op_batch: bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const uint32_t nl = (uint32_t)(cv->pbits + 63) / 64;
	const size_t bytes = (size_t)n * nl * 8;
	if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], bytes) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], bytes) ||
	    ensure(&ctx->stage[2], &ctx->stage_bytes[2], bytes)) {
		return -1;
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	HIPCHK(hipMemcpyAsync(ctx->stage[0], a, bytes, hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(ctx->stage[1], b, bytes, hipMemcpyHostToDevice, s));
	EcamdFpArgs A;
	A.a = (const uint32_t *)ctx->stage[0];
	A.b = (const uint32_t *)ctx->stage[1];
	A.out = (uint32_t *)ctx->stage[2];
	A.n = n;
	A.wstride = 2 * nl;
	A.op = op;
	A.slot = cv->slot;
	HIPCHK(ecamd_launch_fp(cv->nw, A, s));
	HIPCHK(hipMemcpyAsync(out, ctx->stage[2], bytes, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

// ------------------------------------------------------------------------------------------
// batched ECDSA verification (digest supplied by the caller)
// ------------------------------------------------------------------------------------------
// The reference's structure: two independent scalar multiplications and one addition per item.
// All pointers are device pointers; intermediates live in stage[3..11].
// d_pub == NULL: every public key is the point at infinity (libecc imports (0 : 1 : 0) as a key, and its verification
// then computes W' = uG + v*infinity = uG, sig/ecdsa_common.c:788-800): [u2]Y is not computed and W' = [u1]G.
// d_only (may be d_res itself): a status array; only the items marked ECAMD_STATUS_REDO in it are verified (by the complete-formula
// kernel) and get a result -- the redo pass of the interleaved secp256r1 loop, entirely on the device.  Only enqueues.
static int ecdsa_two_smul_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_pub,
			      const uint8_t *d_sig, const uint8_t *d_dig, uint32_t hlen, uint8_t *d_res, hipStream_t s,
			      const uint8_t *d_only = nullptr)
{
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (ecdsa_two_smul_dev(ctx, cv, m, d_pub ? d_pub + (size_t)off * 2 * cv->clen : nullptr, d_sig + (size_t)off * 2 * cv->qlen,
					       d_dig + (size_t)off * hlen, hlen, d_res + off, s, d_only ? d_only + off : nullptr)) {
				return -1;
			}
		}
		return 0;
	}
	const size_t plen = (size_t)2 * cv->clen, ql = (size_t)cv->qlen;
	PublicScalars pub_scope(ctx);   // u1, u2 and the group order are public
	// stage: 3 u1, 4 u2, 5 A, 6 B, 7 stA, 8 stB, 9 flags, 10 subgroup status, 11 q scalar / tmp
	const size_t need[12] = {0, 0, 0, n * ql, n * ql, n * plen, n * plen, n, n, n, n, n * plen + 256};
	for (int i = 3; i < 12; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	EcamdEcdsaPrepArgs P;
	P.sigs = d_sig;
	P.digests = d_dig;
	P.u1 = S[3];
	P.u2 = S[4];
	P.flags = S[9];
	P.n = n;
	P.qlen = (uint32_t)cv->qlen;
	P.hlen = hlen;
	P.qbits = (uint32_t)cv->qbits;
	P.qslot = cv->qslot;
	P.only = d_only;
	if (prep_scratch_ok(ctx, cv, n)) {
		return -1;
	}
	HIPCHK(launch_ecdsa_prep(ctx, cv, P, s));
	const bool redo = d_only != nullptr;
	if (redo) {
		// the marks select the lanes of the complete-formula kernel (A.only_redo); the other items keep what they have
		HIPCHK(hipMemcpyAsync(S[7], d_only, n, hipMemcpyDeviceToDevice, s));
		HIPCHK(hipMemcpyAsync(S[8], d_only, n, hipMemcpyDeviceToDevice, s));
	}
	// uG and vY: two independent prj_pt_mul, as in the reference (sig/ecdsa_common.c:788,793)
	if (smul_dev_locked(ctx, cv, n, S[3], (uint32_t)cv->qlen, nullptr, S[5], S[7], s, 0xffffffffu, redo)) {
		return -1;
	}
	if (!d_pub) {
		HIPCHK(hipMemsetAsync(S[6], 0, n * plen, s));
		HIPCHK(hipMemsetAsync(S[8], 2, n, s));   // [u2]Y = infinity
	} else if (smul_dev_locked(ctx, cv, n, S[4], (uint32_t)cv->qlen, d_pub, S[6], S[8], s, 0xffffffffu, redo)) {
		return -1;
	}
	if (d_pub && big_cmp(cv->order, cv->q) != 0) {
		// cofactor != 1: ec_pub_key_import_from_aff_buf also requires [q]Y == infinity
		// (sig/ec_key.c:199-205).  One more pass with the broadcast scalar q (it sits behind the generator in HBM); a key
		// outside the subgroup is turned into an import error (status 1) for the final stage, on the device.
		const uint8_t *qs = cv->d_gen + plen;
		if (redo) {
			HIPCHK(hipMemcpyAsync(S[10], d_only, n, hipMemcpyDeviceToDevice, s));
		}
		if (smul_dev_locked(ctx, cv, n, qs, (uint32_t)ql, d_pub, S[11], S[10], s, 0, redo)) {
			return -1;
		}
		// S[10] holds 2 (infinity) for keys in the subgroup; anything else rejects: fold into stB
		// (redo pass: items that were not marked keep their old result byte there, and the final stage skips them)
		HIPCHK(ecamd_launch_status_require(S[8], S[10], 2, n, s));
	}
	EcamdEcdsaFinArgs Fn;
	Fn.A = S[5];
	Fn.stA = S[7];
	Fn.B = S[6];
	Fn.stB = S[8];
	Fn.sigs = d_sig;
	Fn.flags = S[9];
	Fn.result = d_res;
	Fn.n = n;
	Fn.clen = (uint32_t)cv->clen;
	Fn.qlen = (uint32_t)cv->qlen;
	{
		// jmax = floor((p - 1) / q), tiny (1 for cofactor-1 curves with p > q, up to 8 for WEI25519)
		uint32_t j = 0;
		Big t = cv->q;
		while (big_cmp(t, cv->p) < 0 && j < 64) {
			t = big_add(t, cv->q);
			j++;
		}
		Fn.jmax = j;
	}
	for (int w = 0; w < 17; w++) {
		Fn.q[w] = (size_t)w < cv->q.size() ? cv->q[(size_t)w] : 0;
	}
	Fn.slot = cv->slot;
	Fn.only = d_only;
	HIPCHK(ecamd_launch_ecdsa_fin(cv->nw, Fn, s));
	return 0;
}

// The mod-q algebra in front of a verification: on the dense radix-2^29 unit of the order's size when the handle has q in a slot
// there (round 4: k_ecdsa_prep_g, sixteen items per inversion from 320 bits on, eight below), else on the saturated words.
// The caller has sized ctx->prep_scratch (prep_scratch_ok).
static bool prep_g29(const ecamd_curve *cv)
{
	return cv->gqslot >= 0 && getenv("ECAMD_NO_PREP_G29") == nullptr;
}
static int prep_scratch_ok(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n)
{
	if (!prep_g29(cv)) {
		return 0;
	}
	return ensure(&ctx->prep_scratch, &ctx->prep_scratch_bytes, (size_t)n * (size_t)ecamd_g29_nl(big_bitlen(cv->q), 0) * 4);
}
static hipError_t launch_ecdsa_prep(ecamd_ctx *ctx, const ecamd_curve *cv, const EcamdEcdsaPrepArgs &P, hipStream_t s)
{
	if (prep_g29(cv) && ctx->prep_scratch) {
		const int qbits = big_bitlen(cv->q);
		static const int kp_env = getenv("ECAMD_PREP_KP") ? atoi(getenv("ECAMD_PREP_KP")) : 0;   // A/B hook
		const int kp = (kp_env >= 1 && kp_env <= 64) ? kp_env : (qbits > 320 ? 16 : 8);
		return ecamd_g29_ecdsa_prep(qbits, cv->gqslot, P, (uint32_t *)ctx->prep_scratch, kp, s);
	}
	return ecamd_launch_ecdsa_prep(cv->qnw, P, s);
}

// k_ecdsa_prep of a chunk on the side stream: it starts when everything enqueued on s so far is done (the stage buffers it
// writes are read by the previous chunk's loop) and the caller makes the consumer of u1 / u2 wait for side_done.
// Returns the event to wait for, or nullptr when the kernel was enqueued on s itself.
static hipEvent_t ecdsa_prep_beside(ecamd_ctx *ctx, const ecamd_curve *cv, const EcamdEcdsaPrepArgs &P, hipStream_t s, hipError_t *err)
{
	if (!ctx->side_ok) {
		*err = launch_ecdsa_prep(ctx, cv, P, s);
		return nullptr;
	}
	*err = hipEventRecord(ctx->side_fork, s);
	if (*err == hipSuccess) {
		*err = hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0);
	}
	if (*err == hipSuccess) {
		*err = launch_ecdsa_prep(ctx, cv, P, ctx->side_stream);
	}
	if (*err == hipSuccess) {
		*err = hipEventRecord(ctx->side_done, ctx->side_stream);
	}
	return ctx->side_done;
}

// device pointers in and out; only enqueues on s
// The fused double-scalar loop of the generic radix-2^29 units (k_loop_g then k_comb_add_g): curves on the affine-table pipeline (every
// flavour but the two nine-limb ones), prime-order groups (no subgroup check of the key to run beside it), u1 within the comb
// table's reach, and a comb table of the generator (built on the first batch of at least comb_min_batch items).
static bool fused_verify_ok(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n)
{
	if (cv->is_p256 || cv->gslot < 0 || getenv("ECAMD_NO_FUSED_VERIFY") != nullptr) {
		return false;
	}
	if (!(cv->gflavour == 0 || cv->gflavour == 1 || cv->gflavour == 3 || cv->gflavour == 5 || cv->gflavour == 6 || cv->gflavour == 7)) {
		return false;
	}
	if (big_cmp(cv->order, cv->q) != 0 || (uint32_t)cv->qlen > ecamd_g29_comb_max_slen(cv->pbits)) {
		return false;
	}
	{
		PublicScalars pub_scope(ctx);
		maybe_build_comb(ctx, const_cast<ecamd_curve *>(cv), n);
	}
	return cv->d_comb != nullptr;
}

// W' = [u1]G + [u2]Q by ONE window loop over Q's affine table plus 2 NW + 1 comb additions for G, for every field size (the
// reference: two prj_pt_mul and a prj_pt_add, sig/ecdsa_common.c:786-796); the affine W' then meets the same final stage as the
// two-multiplication path (x mod q == r over the candidates r + j q < p).  Items whose loop met an exceptional pair come back
// marked and are verified again by ecdsa_two_smul_dev (complete formulas), entirely on the device.
static int ecdsa_fused_g_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig,
			     const uint8_t *d_dig, uint32_t hlen, uint8_t *d_res, hipStream_t s)
{
	const size_t plen = (size_t)2 * cv->clen, ql = (size_t)cv->qlen;
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	PublicScalars pub_scope(ctx);
	// stage: 3 u1, 4 u2, 5 W' affine, 7 its status, 8 "the other point is infinity", 9 flags
	const size_t need[10] = {0, 0, 0, chunk * ql, chunk * ql, chunk * plen, 0, chunk, chunk, chunk};
	for (int i = 3; i < 10; i++) {
		if (need[i] && ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	uint32_t jmax = 0;
	{
		Big t = cv->q;
		while (big_cmp(t, cv->p) < 0 && jmax < 64) {
			t = big_add(t, cv->q);
			jmax++;
		}
	}
	for (uint32_t off = 0; off < n; off += chunk) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		EcamdEcdsaPrepArgs P;
		P.sigs = d_sig + (size_t)off * 2 * ql;
		P.digests = d_dig + (size_t)off * hlen;
		P.u1 = S[3];
		P.u2 = S[4];
		P.flags = S[9];
		P.n = m;
		P.qlen = (uint32_t)cv->qlen;
		P.hlen = hlen;
		P.qbits = (uint32_t)cv->qbits;
		P.qslot = cv->qslot;
		P.only = nullptr;
		// (on s itself: k_table_g already reads the scalars -- it stores their recoding beside the window table -- so there is
		// nothing for this kernel to run beside; the secp256r1 path below does overlap it)
		if (prep_scratch_ok(ctx, cv, m)) {
			return -1;
		}
		HIPCHK(launch_ecdsa_prep(ctx, cv, P, s));
		if (smul_dev_locked(ctx, cv, m, S[4], (uint32_t)ql, d_pub + (size_t)off * plen, S[5], S[7], s, 0xffffffffu, false, S[3])) {
			return -1;
		}
		HIPCHK(hipMemsetAsync(S[8], 2, m, s));
		EcamdEcdsaFinArgs Fn;
		Fn.A = S[5];
		Fn.stA = S[7];
		Fn.B = S[5];            // never read: its status says infinity
		Fn.stB = S[8];
		Fn.sigs = d_sig + (size_t)off * 2 * ql;
		Fn.flags = S[9];
		Fn.result = d_res + off;
		Fn.n = m;
		Fn.clen = (uint32_t)cv->clen;
		Fn.qlen = (uint32_t)cv->qlen;
		Fn.jmax = jmax;
		for (int w = 0; w < 17; w++) {
			Fn.q[w] = (size_t)w < cv->q.size() ? cv->q[(size_t)w] : 0;
		}
		Fn.slot = cv->slot;
		Fn.only = nullptr;
		HIPCHK(ecamd_launch_ecdsa_fin(cv->nw, Fn, s));
	}
	// the redo pass: only the marked items (with nothing marked: near-empty launches)
	return ecdsa_two_smul_dev(ctx, cv, n, d_pub, d_sig, d_dig, hlen, d_res, s, d_res);
}

static int ecdsa_verify_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_pub,
				   const uint8_t *d_sig, const uint8_t *d_dig, uint32_t hlen, uint8_t *d_res, hipStream_t s,
				   const std::function<int()> *between = nullptr)
{
	if (!cv->is_p256 || !cv->d_gtab) {
		if (fused_verify_ok(ctx, cv, n) ? ecdsa_fused_g_dev(ctx, cv, n, d_pub, d_sig, d_dig, hlen, d_res, s)
						: ecdsa_two_smul_dev(ctx, cv, n, d_pub, d_sig, d_dig, hlen, d_res, s)) {
			return -1;
		}
		if (between && (*between)()) {
			return -1;
		}
		return 0;
	}
	// secp256r1: interleaved [u1]G + [u2]Q loop (ecamd_launch_verify_p256), in chunks
	PublicScalars pub_scope(ctx);   // everything a verification multiplies by is public
	maybe_build_comb(ctx, const_cast<ecamd_curve *>(cv), n);
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	{
		uint8_t *t = (uint8_t *)ctx->tbl_fast;
		const int rc = ensure(&t, &ctx->tbl_fast_bytes, (size_t)((chunk + 63u) & ~63u) * P256_SCRATCH_PER_ITEM);
		ctx->tbl_fast = (uint32_t *)t;
		if (rc) {
			return -1;
		}
	}
	// stage: 3 u1, 4 u2, 5 flags, 7 zeroed points of rejected keys + key status
	const size_t need[8] = {0, 0, 0, (size_t)chunk * 32, (size_t)chunk * 32, chunk, 0, (size_t)chunk * 64 + chunk};
	for (int i = 3; i < 8; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	for (uint32_t off = 0; off < n; off += chunk) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		EcamdEcdsaPrepArgs P;
		P.sigs = d_sig + (size_t)off * 64;
		P.digests = d_dig + (size_t)off * hlen;
		P.u1 = S[3];
		P.u2 = S[4];
		P.flags = S[5];
		P.n = m;
		P.qlen = 32;
		P.hlen = hlen;
		P.qbits = (uint32_t)cv->qbits;
		P.qslot = cv->qslot;
		P.only = nullptr;
		hipError_t perr = hipSuccess;
		if (prep_scratch_ok(ctx, cv, m)) {
			return -1;
		}
		const hipEvent_t prep_done = ecdsa_prep_beside(ctx, cv, P, s, &perr);
		HIPCHK(perr);
		EcamdSmulArgs K;
		memset(&K, 0, sizeof(K));
		K.points = d_pub + (size_t)off * 64;
		K.pstride = 64;
		K.out = S[7];           // zeroed for rejected keys; otherwise unused
		K.status = S[7] + (size_t)m * 64;
		K.tbl = ctx->tbl_fast;
		K.stg = ctx->tbl_fast + (size_t)((chunk + 63u) & ~63u) * (P256_TAB_BYTES / 4);
		K.n = m;
		K.clen = 32;
		K.slot = cv->slot;
		hipEvent_t *dom = (ctx->timing && off == 0) ? ctx->ev_dom : nullptr;
		HIPCHK(ecamd_launch_verify_p256(K, S[3], S[4], d_sig + (size_t)off * 64, S[5],
						cv->d_comb ? cv->d_comb : cv->d_gtab, cv->d_comb ? 1 : 0, cv->qdig, d_res + off, s, dom, prep_done));
		ctx->ev_dom_valid = ctx->ev_dom_valid || (dom != nullptr);
	}
	// exceptional pairs inside the interleaved loop (never for honest signatures) come back as ECAMD_STATUS_REDO: those
	// items are verified again the reference's way -- two complete-formula multiplications -- by kernels whose other
	// lanes exit at once (with nothing marked: five near-empty launches)
	if (ecdsa_two_smul_dev(ctx, cv, n, d_pub, d_sig, d_dig, hlen, d_res, s, d_res)) {
		return -1;
	}
	if (between && (*between)()) {
		return -1;
	}
	return 0;
}

static int ecdsa_verify_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *a,
				const void *b, const void *c, const void *d, uint32_t hlen)
{
	static thread_local char msg[160];
	if (!ctx || !cv || cv->ctx != ctx || (n && (!a || !b || !c || !d))) {
		snprintf(msg, sizeof(msg), "%s: bad argument", fn);
		return fail(msg);
	}
	if (cv->qslot < 0) {
		snprintf(msg, sizeof(msg), "%s: generator order not supported for this curve", fn);
		return fail(msg);
	}
	if (hlen == 0 || hlen > 128) {
		snprintf(msg, sizeof(msg), "%s: digest length must be in 1..128", fn);
		return fail(msg);
	}
	return 0;
}

extern "C" int ec_ecdsa_verify_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *d_pubkeys,
					 const void *d_sigs, const void *d_digests, uint32_t hlen, void *d_result,
					 void *hip_stream)
{
	if (ecdsa_verify_args_ok("ec_ecdsa_verify_batch_dev", ctx, cv, n, d_pubkeys, d_sigs, d_digests, d_result, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	return ecdsa_verify_dev_locked(ctx, cv, n, (const uint8_t *)d_pubkeys, (const uint8_t *)d_sigs,
				       (const uint8_t *)d_digests, hlen, (uint8_t *)d_result, s);
}

// Messages instead of digests (round 4): `data` then holds one fixed-stride SLOT per item -- a little-endian u32 length and the
// message bytes (ecamd_hash.hip) -- and the digests H(m) are computed on the device (hash_type: libecc's hash_alg_type numbers,
// SHA224 = 1 ... SHA512 = 4) into stage 17 before the verification core runs; dig_out (host, n x digest length, may be NULL)
// receives them for the callers that need a digest on the host again.
static int ecdsa_hash_stage(ecamd_ctx *ctx, int hash_type, uint32_t m, const uint8_t *d_slots, uint32_t stride, uint32_t dlen, hipStream_t s)
{
	if (ensure(&ctx->stage[17], &ctx->stage_bytes[17], (size_t)m * dlen)) {
		return -1;
	}
	if (hash_type == 5) {
		HIPCHK(ecamd_launch_shake256_slots(d_slots, stride, m, ctx->stage[17], dlen, dlen, s));   // SHAKE256, dlen octets of output
	} else {
		HIPCHK(ecamd_launch_sha2_slots(hash_type, d_slots, stride, m, ctx->stage[17], dlen, s));
	}
	return 0;
}

static int ecdsa_verify_host_aff(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				 const uint8_t *data, uint32_t data_stride, int hash_type, uint32_t hlen, uint8_t *result, const char *fn)
{
	if (ecdsa_verify_args_ok(fn, ctx, cv, n, pubkeys, sigs, data, result, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t plen = (size_t)2 * cv->clen, slen2 = (size_t)2 * cv->qlen;
	const std::vector<HostArr> arrs = {{pubkeys, nullptr, plen}, {sigs, nullptr, slen2}, {data, nullptr, data_stride}, {nullptr, result, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &between) {
		const uint8_t *d_dig = ip[2];
		if (hash_type) {
			if (ecdsa_hash_stage(ctx, hash_type, m, ip[2], data_stride, hlen, s)) {
				return -1;
			}
			d_dig = ctx->stage[17];
		}
		return ecdsa_verify_dev_locked(ctx, cv, m, ip[0], ip[1], d_dig, hlen, op[3], s, &between);
	});
}

extern "C" int ec_ecdsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys,
				     const uint8_t *sigs, const uint8_t *digests, uint32_t hlen, uint8_t *result)
{
	return ecdsa_verify_host_aff(ctx, cv, n, pubkeys, sigs, digests, hlen, 0, hlen, result, "ec_ecdsa_verify_batch");
}

// ECDSA verification with the public keys in either point wire format.  ECAMD_PT_PROJECTIVE is what an ec_pub_key holds
// (ec_pub_key_export_to_buf: X || Y || Z of pub_key->y): imported as prj_pt_import_from_buf does, normalised on the device,
// and a key that is the point at infinity -- which libecc imports and verifies against, W' = uG -- is handled as the
// reference does.
static int ecdsa_verify_host_fmt(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, int pub_fmt, const uint8_t *sigs,
				 const uint8_t *data, uint32_t data_stride, int hash_type, uint32_t hlen, uint8_t *result, const char *fn)
{
	if (pub_fmt == ECAMD_PT_AFFINE) {
		return ecdsa_verify_host_aff(ctx, cv, n, pubkeys, sigs, data, data_stride, hash_type, hlen, result, fn);
	}
	if (pub_fmt != ECAMD_PT_PROJECTIVE) {
		return fail("ec_ecdsa_verify_batch_fmt: point format must be ECAMD_PT_AFFINE or ECAMD_PT_PROJECTIVE");
	}
	if (ecdsa_verify_args_ok(fn, ctx, cv, n, pubkeys, sigs, data, result, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	const size_t alen = (size_t)2 * cv->clen, sl = (size_t)2 * cv->qlen;
	std::vector<uint8_t> st(n);
	{
		// keys, signatures and digests go to the device once, chunk by chunk behind the double-buffered staging; the keys are
		// imported and normalised there (k_prj_import) and the verification core consumes the affine form in HBM
		std::lock_guard<std::mutex> lk(ctx->mu);
		HIPCHK(hipSetDevice(ctx->device));
		std::vector<HostArr> arrs = {{pubkeys, nullptr, 3 * (size_t)cv->clen}, {sigs, nullptr, sl}, {data, nullptr, data_stride},
					     {nullptr, result, 1}, {nullptr, st.data(), 1}};
		if (host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
						     hipStream_t s, const std::function<int()> &between) {
			    if (ensure(&ctx->stage[12], &ctx->stage_bytes[12], (size_t)m * alen)) {
				    return -1;
			    }
			    const uint8_t *d_dig = ip[2];
			    if (hash_type) {
				    if (ecdsa_hash_stage(ctx, hash_type, m, ip[2], data_stride, hlen, s)) {
					    return -1;
				    }
				    d_dig = ctx->stage[17];
			    }
			    EcamdPrjInArgs I;
			    I.in = ip[0];
			    I.aff = ctx->stage[12];
			    I.pre = op[4];
			    I.n = m;
			    I.clen = (uint32_t)cv->clen;
			    I.for_mul = 0;
			    I.slot = cv->slot;
			    HIPCHK(launch_prj_import(cv, I, s));
			    return ecdsa_verify_dev_locked(ctx, cv, m, ctx->stage[12], ip[1], d_dig, hlen, op[3], s, &between);
		    })) {
			return -1;
		}
	}
	std::vector<uint32_t> inf;
	for (uint32_t i = 0; i < n; i++) {
		if (st[i] == ECAMD_ERR) {
			result[i] = 1;
		} else if (st[i] == ECAMD_INF) {
			// (0 : 0 : 0) passes prj_pt_import_from_buf too; the reference's scalar multiplication then fails on it
			bool allzero = true;
			for (size_t b = 0; b < 3 * (size_t)cv->clen && allzero; b++) {
				allzero = pubkeys[(size_t)i * 3 * cv->clen + b] == 0;
			}
			result[i] = 1;
			if (!allzero) {
				inf.push_back(i);
			}
		}
	}
	if (inf.empty()) {
		return 0;
	}
	// keys at infinity: W' = [u1]G.  Their signatures and digests (or message slots, hashed on the device again: the digests of
	// the batch never travel back) are gathered on the host -- a handful of items at most.
	const uint32_t r = (uint32_t)inf.size();
	const size_t dl = hash_type ? (size_t)data_stride : (size_t)hlen;
	std::vector<uint8_t> hs((size_t)r * sl), hd((size_t)r * dl), hr(r);
	for (uint32_t j = 0; j < r; j++) {
		memcpy(&hs[(size_t)j * sl], sigs + (size_t)inf[j] * sl, sl);
		memcpy(&hd[(size_t)j * dl], data + (size_t)inf[j] * dl, dl);
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t gneed[3] = {(size_t)r * sl, (size_t)r * dl, r};
	for (int i = 0; i < 3; i++) {
		if (ensure(&ctx->stage[14 + i], &ctx->stage_bytes[14 + i], gneed[i])) {
			return -1;
		}
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	uint8_t **S = ctx->stage;
	HIPCHK(hipMemcpyAsync(S[14], hs.data(), hs.size(), hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(S[15], hd.data(), hd.size(), hipMemcpyHostToDevice, s));
	const uint8_t *d_dig = S[15];
	if (hash_type) {
		if (ecdsa_hash_stage(ctx, hash_type, r, S[15], data_stride, hlen, s)) {
			return -1;
		}
		d_dig = S[17];
	}
	if (ecdsa_two_smul_dev(ctx, cv, r, nullptr, S[14], d_dig, hlen, S[16], s)) {
		return -1;
	}
	HIPCHK(hipMemcpyAsync(hr.data(), S[16], r, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	for (uint32_t j = 0; j < r; j++) {
		result[inf[j]] = hr[j];
	}
	return 0;
}

extern "C" int ec_ecdsa_verify_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
					 const uint8_t *sigs, const uint8_t *digests, uint32_t hlen, uint8_t *result)
{
	return ecdsa_verify_host_fmt(ctx, cv, n, pubkeys, pub_fmt, sigs, digests, hlen, 0, hlen, result, "ec_ecdsa_verify_batch_fmt");
}

// ECDSA verification from MESSAGES: item i's message sits in a slot of msg_stride bytes (a multiple of 4, at most 4096): a
// little-endian u32 length, then the bytes (4 + length <= msg_stride); H(m) is SHA-224 / 256 / 384 / 512 (hash_type 1 .. 4,
// libecc's hash_alg_type numbers) computed on the device, the rest is ec_ecdsa_verify_batch_fmt.
extern "C" int ec_ecdsa_verify_msg_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
					     const uint8_t *sigs, int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result)
{
	const int dl = ecamd_sha2_digest_len(hash_type);
	if (dl == 0 || msg_stride < 4 || (msg_stride & 3u) || msg_stride > 4096) {
		return fail("ec_ecdsa_verify_msg_batch_fmt: hash_type must be 1 .. 4 (SHA-224 / 256 / 384 / 512), msg_stride a multiple of 4 in 4 .. 4096");
	}
	return ecdsa_verify_host_fmt(ctx, cv, n, pubkeys, pub_fmt, sigs, msg_slots, msg_stride, hash_type, (uint32_t)dl, result,
				     "ec_ecdsa_verify_msg_batch_fmt");
}

// ------------------------------------------------------------------------------------------
// batched ECDSA signing with caller-supplied nonces
// ------------------------------------------------------------------------------------------
// Device core of ECDSA signing: every pointer a device pointer; kG and its status live in stage[3], stage[4].
static int ecdsa_sign_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_privs,
				 const uint8_t *d_nonces, const uint8_t *d_digests, uint32_t hlen, uint8_t *d_sigs,
				 uint8_t *d_status, hipStream_t s)
{
	const size_t plen = (size_t)2 * cv->clen, ql = (size_t)cv->qlen;
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (ecdsa_sign_dev_locked(ctx, cv, m, d_privs + off * ql, d_nonces + off * ql, d_digests + (size_t)off * hlen,
						  hlen, d_sigs + off * 2 * ql, d_status + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	if (ensure(&ctx->stage[3], &ctx->stage_bytes[3], n * plen) || ensure(&ctx->stage[4], &ctx->stage_bytes[4], n)) {
		return -1;
	}
	uint8_t **S = ctx->stage;
	if (smul_dev_locked(ctx, cv, n, d_nonces, (uint32_t)ql, nullptr, S[3], S[4], s)) {  // kG (:479)
		return -1;
	}
	EcamdEcdsaSignArgs A;
	A.privs = d_privs;
	A.nonces = d_nonces;
	A.digests = d_digests;
	A.kG = S[3];
	A.stkG = S[4];
	A.sigs = d_sigs;
	A.status = d_status;
	A.n = n;
	A.clen = (uint32_t)cv->clen;
	A.qlen = (uint32_t)cv->qlen;
	A.hlen = hlen;
	A.qbits = (uint32_t)cv->qbits;
	{
		uint32_t j = 0;
		Big t = cv->q;
		while (big_cmp(t, cv->p) < 0 && j < 64) {
			t = big_add(t, cv->q);
			j++;
		}
		A.jmax = j;
	}
	A.qslot = cv->qslot;
	HIPCHK(ecamd_launch_ecdsa_sign(cv->qnw, A, s));
	return 0;
}

static int ecdsa_sign_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *a,
			      const void *b, const void *c, const void *d, const void *e, uint32_t hlen)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!a || !b || !c || !d || !e))) {
		return fail(std::string(fn) + ": bad argument");
	}
	if (cv->qslot < 0) {
		return fail(std::string(fn) + ": generator order not supported for this curve");
	}
	if (hlen == 0 || hlen > 128) {
		return fail(std::string(fn) + ": digest length must be in 1..128");
	}
	return 0;
}

extern "C" int ec_ecdsa_sign_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *d_privs,
				       const void *d_nonces, const void *d_digests, uint32_t hlen, void *d_sigs,
				       void *d_status, void *hip_stream)
{
	if (ecdsa_sign_args_ok("ec_ecdsa_sign_batch_dev", ctx, cv, n, d_privs, d_nonces, d_digests, d_sigs, d_status, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	return ecdsa_sign_dev_locked(ctx, cv, n, (const uint8_t *)d_privs, (const uint8_t *)d_nonces, (const uint8_t *)d_digests,
				     hlen, (uint8_t *)d_sigs, (uint8_t *)d_status, s);
}

extern "C" int ec_ecdsa_sign_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *privs,
				   const uint8_t *nonces, const uint8_t *digests, uint32_t hlen, uint8_t *sigs,
				   uint8_t *status)
{
	if (ecdsa_sign_args_ok("ec_ecdsa_sign_batch", ctx, cv, n, privs, nonces, digests, sigs, status, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t ql = (size_t)cv->qlen;
	const std::vector<HostArr> arrs = {{privs, nullptr, ql}, {nonces, nullptr, ql}, {digests, nullptr, hlen},
					   {nullptr, sigs, 2 * ql}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return ecdsa_sign_dev_locked(ctx, cv, m, ip[0], ip[1], ip[2], hlen, op[3], op[4], s);
	});
}

// ------------------------------------------------------------------------------------------
// Round 4: the secret-key half without libecc's division on the host.  nn_get_random_mod (nn/nn_rand.c:92-150) is get_random for
// 2 * qlen bytes, then a constant-time reduction modulo q - 1 that costs a host thread about a microsecond: more than everything else
// a signature or a key pair needs from the host.  The random bytes stay the application's; the reduction moves to the
// device (k_rand_mod, ecamd_randmod.h), and so does H(m) (k_sha2_slots).
// ------------------------------------------------------------------------------------------
static int rand_mod_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_raw, uint8_t *d_out, hipStream_t s)
{
	(void)ctx;
	EcamdRandModArgs R;
	R.raw = d_raw;
	R.out = d_out;
	R.n = n;
	R.qlen = (uint32_t)cv->qlen;
	R.rawlen = 2 * (uint32_t)cv->qlen;
	for (int w = 0; w < 18; w++) {
		R.q[w] = (size_t)w < cv->q.size() ? cv->q[(size_t)w] : 0;
	}
	HIPCHK(ecamd_launch_rand_mod(cv->qnw, R, s));
	return 0;
}
static int rand_mod_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv)
{
	if (!ctx || !cv || cv->ctx != ctx) {
		return fail(std::string(fn) + ": bad argument");
	}
	if (!(cv->q[0] & 1) || big_bitlen(cv->q) < 2 || !cv->qnw || !ecamd_nw_supported(cv->qnw) || cv->q.size() > 18) {
		return fail(std::string(fn) + ": generator order not supported (odd, at most 544 bits)");
	}
	return 0;
}

extern "C" int ec_nn_random_mod_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *raw, uint8_t *out)
{
	if (rand_mod_args_ok("ec_nn_random_mod_batch", ctx, cv)) {
		return -1;
	}
	if (n && (!raw || !out)) {
		return fail("ec_nn_random_mod_batch: bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t ql = (size_t)cv->qlen;
	const std::vector<HostArr> arrs = {{raw, nullptr, 2 * ql}, {nullptr, out, ql}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return rand_mod_dev_locked(ctx, cv, m, ip[0], op[1], s);
	});
}

// ECDSA signatures from messages and RAW nonce material: k = nn_get_random_mod's value for the 2 * qlen random bytes of the item,
// h = SHA-2(m) of its message slot (hash_type 1 .. 4; 0: the slots ARE the digests, msg_stride bytes each), then ec_ecdsa_sign_batch.
extern "C" int ec_ecdsa_sign_msg_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw,
				       int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status)
{
	const uint32_t hlen = hash_type ? (uint32_t)ecamd_sha2_digest_len(hash_type) : msg_stride;
	if (hash_type && (hlen == 0 || msg_stride < 4 || (msg_stride & 3u) || msg_stride > 4096)) {
		return fail("ec_ecdsa_sign_msg_batch: hash_type must be 0 (digests) or 1 .. 4 (SHA-224 / 256 / 384 / 512) with a slot stride that is a multiple of 4 in 4 .. 4096");
	}
	if (rand_mod_args_ok("ec_ecdsa_sign_msg_batch", ctx, cv) ||
	    ecdsa_sign_args_ok("ec_ecdsa_sign_msg_batch", ctx, cv, n, privs, nonce_raw, msg_slots, sigs, status, hlen)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t ql = (size_t)cv->qlen;
	const std::vector<HostArr> arrs = {{privs, nullptr, ql}, {nonce_raw, nullptr, 2 * ql}, {msg_slots, nullptr, msg_stride},
					   {nullptr, sigs, 2 * ql}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		// stage 20: the nonces (secret: ensure() wipes what it frees, ecamd_ctx_wipe_scratch the rest); 17: the digests
		if (ensure(&ctx->stage[20], &ctx->stage_bytes[20], (size_t)m * ql) || rand_mod_dev_locked(ctx, cv, m, ip[1], ctx->stage[20], s)) {
			return -1;
		}
		const uint8_t *d_dig = ip[2];
		if (hash_type) {
			if (ecdsa_hash_stage(ctx, hash_type, m, ip[2], msg_stride, hlen, s)) {
				return -1;
			}
			d_dig = ctx->stage[17];
		}
		return ecdsa_sign_dev_locked(ctx, cv, m, ip[0], ctx->stage[20], d_dig, hlen, op[3], op[4], s);
	});
}

// Key pairs from RAW random material: x = nn_get_random_mod's value for the item's 2 * qlen random bytes (ec_key_pair_gen ->
// generic_gen_priv_key, sig/ec_key.c:602), Y = [x]G.  priv_out: n x qlen big-endian; pub_out: n x 2 clen affine X || Y.
extern "C" int ec_key_pair_gen_raw_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *raw, uint8_t *priv_out,
					 uint8_t *pub_out, uint8_t *status)
{
	if (rand_mod_args_ok("ec_key_pair_gen_raw_batch", ctx, cv)) {
		return -1;
	}
	if (n && (!raw || !priv_out || !pub_out || !status)) {
		return fail("ec_key_pair_gen_raw_batch: bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t ql = (size_t)cv->qlen, plen = (size_t)2 * cv->clen;
	const std::vector<HostArr> arrs = {{raw, nullptr, 2 * ql}, {nullptr, priv_out, ql}, {nullptr, pub_out, plen}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		if (rand_mod_dev_locked(ctx, cv, m, ip[0], op[1], s)) {
			return -1;
		}
		return smul_dev_locked(ctx, cv, m, op[1], (uint32_t)ql, nullptr, op[2], op[3], s);
	});
}

// ------------------------------------------------------------------------------------------
// batched ECC-CDH (ecccdh_derive_secret, ecdh/ecccdh.c:167-233)
// ------------------------------------------------------------------------------------------
// Device core: stage 3 [d]Q', 4 its status, 5 [h]Q, 6 status of [h]Q, 7 [q]Q (discarded), 8 its status.
static int ecccdh_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_privs,
			     const uint8_t *d_peers, uint8_t *d_secrets, uint8_t *d_status, hipStream_t s)
{
	const size_t plen = (size_t)2 * cv->clen, ql = (size_t)cv->qlen;
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (ecccdh_dev_locked(ctx, cv, m, d_privs + off * ql, d_peers + off * plen,
					      d_secrets + (size_t)off * cv->clen, d_status + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	const bool cof = cv->cofactor != 1;
	const size_t need[9] = {0, 0, 0, n * plen, n, cof ? n * plen : 0, cof ? n : 0, cof ? n * plen : 0, cof ? n : 0};
	for (int i = 3; i < 9; i++) {
		if (need[i] && ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	EcamdCdhArgs A;
	memset(&A, 0, sizeof(A));
	A.n = n;
	A.clen = (uint32_t)cv->clen;
	const uint8_t *pts_in = d_peers;
	if (cof) {
		// cofactor h != 1 (ecdh/ecccdh.c:187-217): the peer key must lie in the subgroup ([q]Q = infinity,
		// sig/ec_key.c:199-205), then Q' = [h]Q must not be infinity, then [d]Q'
		const uint8_t *d_q = cv->d_gen + plen, *d_h = d_q + ql;
		PublicScalars pub_scope(ctx);   // the order and the cofactor are public; the private key below is not
		if (smul_dev_locked(ctx, cv, n, d_q, (uint32_t)ql, d_peers, S[7], S[8], s, 0) ||
		    smul_dev_locked(ctx, cv, n, d_h, 1, d_peers, S[5], S[6], s, 0)) {
			return -1;
		}
		A.st_sub = S[8];
		A.st_h = S[6];
		A.hq = S[5];
		HIPCHK(ecamd_launch_cdh_gate(A, s));
		pts_in = S[5];
	}
	// cofactor 1: import + prj_pt_mul + prj_pt_unique in one pass
	if (smul_dev_locked(ctx, cv, n, d_privs, (uint32_t)ql, pts_in, S[3], S[4], s)) {
		return -1;
	}
	A.pts = S[3];
	A.st = S[4];
	A.secrets = d_secrets;
	A.status = d_status;
	HIPCHK(ecamd_launch_cdh_fin(A, s));
	return 0;
}

static int ecccdh_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *a, const void *b,
			  const void *c, const void *d)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!a || !b || !c || !d))) {
		return fail(std::string(fn) + ": bad argument");
	}
	if (cv->cofactor == 0) {
		return fail(std::string(fn) + ": unexpected cofactor");
	}
	return 0;
}

extern "C" int ec_ecccdh_derive_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *d_privs,
					  const void *d_peers, void *d_secrets, void *d_status, void *hip_stream)
{
	if (ecccdh_args_ok("ec_ecccdh_derive_batch_dev", ctx, cv, n, d_privs, d_peers, d_secrets, d_status)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	return ecccdh_dev_locked(ctx, cv, n, (const uint8_t *)d_privs, (const uint8_t *)d_peers, (uint8_t *)d_secrets,
				 (uint8_t *)d_status, s);
}

extern "C" int ec_ecccdh_derive_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *privs,
				      const uint8_t *peers, uint8_t *secrets, uint8_t *status)
{
	if (ecccdh_args_ok("ec_ecccdh_derive_batch", ctx, cv, n, privs, peers, secrets, status)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t plen = (size_t)2 * cv->clen, ql = (size_t)cv->qlen;
	const std::vector<HostArr> arrs = {{privs, nullptr, ql}, {peers, nullptr, plen}, {nullptr, secrets, (size_t)cv->clen},
					   {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return ecccdh_dev_locked(ctx, cv, m, ip[0], ip[1], op[2], op[3], s);
	});
}

// ------------------------------------------------------------------------------------------
// batched X25519 / X448 (x25519() / x448(), ecdh/x25519_448.c:380-425 -> x25519_448_core :146)
// ------------------------------------------------------------------------------------------
static Big big_sqrt_m1(const Big &p)  // 2^((p-1)/4) mod p, a square root of -1 when p = 5 mod 8
{
	Big one(1, 1), two(1, 2);
	Big e = big_sub(p, one);
	// divide by 4
	Big q(e.size(), 0);
	for (size_t i = 0; i < e.size(); i++) {
		q[i] = (e[i] >> 2) | ((i + 1 < e.size()) ? (e[i + 1] << 30) : 0u);
	}
	big_trim(q);
	return big_powmod(two, q, p);
}

// constants of the X25519 / X448 path for this handle; ctx->mu held
static void xdh_setup(ecamd_curve *cv)
{
	cv->xdh_state = -1;
	// the Weierstrass models of Curve25519 / Curve448: A = 486662 / 156326, B = 1
	uint32_t Aval = 0;
	if (cv->pbits == 255 && cv->clen == 32 && big_cmp(cv->p, big_sub(big_pow2(255), Big(1, 19))) == 0) {
		Aval = 486662;
	} else if (cv->pbits == 448 && cv->clen == 56 &&
		   big_cmp(cv->p, big_sub(big_sub(big_pow2(448), big_pow2(224)), Big(1, 1))) == 0) {
		Aval = 156326;
	} else {
		cv->xdh_err = "ec_xdh_batch: the curve is neither WEI25519 nor WEI448";
		return;
	}
	const Big &p = cv->p;
	const int nw = cv->nw;
	const Big A(1, Aval), one(1, 1), two(1, 2), three(1, 3), nine(1, 9), tw7(1, 27);
	const Big pm2 = big_sub(p, two);
	{
		// make sure the handle really is the birationally equivalent Weierstrass curve:
		// a = (3 - A^2) / 3, b = (2 A^3 - 9 A) / 27
		Big A2 = big_mulmod(A, A, p);
		Big a_exp = big_mulmod(big_mod(big_add(big_sub(p, A2), three), p), big_powmod(three, pm2, p), p);
		Big A3c = big_mulmod(A2, A, p);
		Big num = big_mod(big_add(big_mulmod(two, A3c, p), big_sub(p, big_mulmod(nine, A, p))), p);
		Big b_exp = big_mulmod(num, big_powmod(tw7, pm2, p), p);
		if (big_cmp(a_exp, cv->a) != 0 || big_cmp(b_exp, cv->b) != 0) {
			cv->xdh_err = "ec_xdh_batch: curve coefficients do not match the Montgomery curve";
			return;
		}
	}
	const Big R = big_mod(big_pow2(32 * nw), p);
	const Big A3 = big_mulmod(A, big_powmod(three, pm2, p), p);
	EcamdXdhPrepArgs &P = cv->xdh_tmpl;
	memset(&P, 0, sizeof(P));
	P.len = (uint32_t)cv->clen;
	Big e;
	if ((p[0] & 7u) == 5u) {
		P.mode = 0;
		e = big_add(p, three);  // (p + 3) / 8
		Big q(e.size(), 0);
		for (size_t i = 0; i < e.size(); i++) {
			q[i] = (e[i] >> 3) | ((i + 1 < e.size()) ? (e[i + 1] << 29) : 0u);
		}
		e = q;
		big_store(P.sm1, nw, big_mulmod(big_sqrt_m1(p), R, p));
	} else {
		P.mode = 1;
		e = big_add(p, one);  // (p + 1) / 4
		Big q(e.size(), 0);
		for (size_t i = 0; i < e.size(); i++) {
			q[i] = (e[i] >> 2) | ((i + 1 < e.size()) ? (e[i + 1] << 30) : 0u);
		}
		e = q;
	}
	big_trim(e);
	P.ebits = (uint32_t)big_bitlen(e);
	big_store(P.e, 17, e);
	big_store(P.A, nw, big_mulmod(A, R, p));
	big_store(P.A3, nw, big_mulmod(A3, R, p));
	if (Aval == 486662) {
		big_digits29(P.g_A, 9, A);
		big_digits29(P.g_A3, 9, A3);
		big_digits29(P.g_sm1, 9, big_sqrt_m1(p));
	} else if (cv->pbits == 448) {
		big_digits29(P.g_A, 16, A, 28);      // the Goldilocks unit's 28-bit limbs
		big_digits29(P.g_A3, 16, A3, 28);
	}
	P.slot = cv->slot;
	big_store(cv->xdh_A3, 17, A3);
	// cofactor h: order = h q
	uint32_t hval = 0;
	Big t = cv->q;
	for (uint32_t c = 1; c <= 16; c++) {
		if (big_cmp(t, cv->order) == 0) {
			hval = c;
			break;
		}
		t = big_add(t, cv->q);
	}
	if (hval != 4 && hval != 8) {
		cv->xdh_err = "ec_xdh_batch: unexpected cofactor";
		return;
	}
	cv->xdh_cof = (uint8_t)hval;
	P.cof_dbl = hval == 8 ? 3u : 2u;
	cv->xdh_state = 1;
}

// device pointers in and out; only enqueues on s
static int xdh_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_k, const uint8_t *d_u,
			  uint8_t *d_out, uint8_t *d_status, hipStream_t s)
{
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			const size_t l = (size_t)cv->clen;
			if (xdh_dev_locked(ctx, cv, m, d_k + off * l, d_u + off * l, d_out + off * l, d_status + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	const size_t len = (size_t)cv->clen, plen = 2 * len;
	// stage: 2 scalars BE, 3 points, 4 flags, 7 [k]Q, 8 stk   (0, 1, 9, 10 belong to the host-pointer wrapper)
	const size_t need[ECAMD_NSTAGE] = {0, 0, n * len, n * plen, n, 0, 0, n * plen, n};
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	const int nw = cv->nw;
	EcamdXdhPrepArgs P = cv->xdh_tmpl;
	P.k = d_k;
	P.u = d_u;
	P.scalars = S[2];
	P.points = S[3];
	P.flags = S[4];
	P.n = n;
	if (cv->gflavour == 2 && cv->gslot >= 0) {
		// X25519 on the radix-2^29 field of the 2^255 - 19 unit: validation + clamping, then the x-only
		// Montgomery ladder with a shared inversion (the u coordinate is all the reference exposes)
		HIPCHK(ecamd_launch_xdh_prep_c25519(P, cv->gslot, s));
		if (getenv("ECAMD_NO_X25519_LADDER") == nullptr) {
			if (ensure(&ctx->stage[5], &ctx->stage_bytes[5], (size_t)n * ECAMD_XDH_REC_WORDS * 4)) {
				return -1;
			}
			EcamdXdhLadderArgs L;
			L.u = d_u;
			L.scalars = S[2];
			L.flags = S[4];
			L.rec = (uint32_t *)S[5];
			L.out = d_out;
			L.status = d_status;
			L.n = n;
			HIPCHK(ecamd_launch_x25519_ladder(L, cv->gslot, s, ctx->timing ? ctx->ev_dom : nullptr));
			ctx->ev_dom_valid = ctx->ev_dom_valid || ctx->timing;
			return 0;
		}
	} else if (cv->gflavour == 5 && cv->gslot >= 0 && P.mode == 1 && getenv("ECAMD_NO_G448_DECODE") == nullptr) {
		HIPCHK(ecamd_launch_xdh_prep_c448(P, cv->gslot, s));   // X448: the same front end on the Goldilocks radix-2^29 field
		if (getenv("ECAMD_NO_X448_LADDER") == nullptr) {
			// and the x-only Montgomery ladder with a shared inversion, as for X25519
			if (ensure(&ctx->stage[5], &ctx->stage_bytes[5], (size_t)n * ECAMD_X448_REC_WORDS * 4)) {
				return -1;
			}
			EcamdXdhLadderArgs L;
			L.u = d_u;
			L.scalars = S[2];
			L.flags = S[4];
			L.rec = (uint32_t *)S[5];
			L.out = d_out;
			L.status = d_status;
			L.n = n;
			HIPCHK(ecamd_launch_x448_ladder(L, cv->gslot, s, ctx->timing ? ctx->ev_dom : nullptr));
			ctx->ev_dom_valid = ctx->ev_dom_valid || ctx->timing;
			return 0;
		}
	} else {
		HIPCHK(ecamd_launch_xdh_prep(nw, P, s));
	}
	// [k]Q  ([h]Q != infinity was checked by the prep kernel)
	if (smul_dev_locked(ctx, cv, n, S[2], (uint32_t)len, S[3], S[7], S[8], s)) {
		return -1;
	}
	EcamdXdhFinArgs Fn;
	memset(&Fn, 0, sizeof(Fn));
	Fn.pts = S[7];
	Fn.stk = S[8];
	Fn.flags = S[4];
	Fn.out = d_out;
	Fn.status = d_status;
	Fn.n = n;
	Fn.len = (uint32_t)len;
	memcpy(Fn.A3, cv->xdh_A3, sizeof(Fn.A3));
	Fn.slot = cv->slot;
	HIPCHK(ecamd_launch_xdh_fin(nw, Fn, s));
	return 0;
}

static int xdh_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *a, const void *b,
		       const void *c, const void *d)
{
	if (!ctx || !cv_in || cv_in->ctx != ctx || (n && (!a || !b || !c || !d))) {
		return fail(std::string(fn) + ": bad argument");
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (cv->xdh_state == 0) {
		xdh_setup(cv);
	}
	if (cv->xdh_state < 0) {
		return fail(cv->xdh_err);
	}
	return 0;
}

extern "C" int ec_xdh_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *d_k, const void *d_u,
				void *d_out, void *d_status, void *hip_stream)
{
	if (!ctx) {
		return fail("ec_xdh_batch_dev: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (xdh_args_ok("ec_xdh_batch_dev", ctx, cv_in, n, d_k, d_u, d_out, d_status)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	return xdh_dev_locked(ctx, const_cast<ecamd_curve *>(cv_in), n, (const uint8_t *)d_k, (const uint8_t *)d_u,
			      (uint8_t *)d_out, (uint8_t *)d_status, s);
}

extern "C" int ec_xdh_batch(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *k, const uint8_t *u,
			    uint8_t *out, uint8_t *status)
{
	if (!ctx) {
		return fail("ec_xdh_batch: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (xdh_args_ok("ec_xdh_batch", ctx, cv_in, n, k, u, out, status)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	const size_t len = (size_t)cv->clen;
	const std::vector<HostArr> arrs = {{k, nullptr, len}, {u, nullptr, len}, {nullptr, out, len}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return xdh_dev_locked(ctx, cv, m, ip[0], ip[1], op[2], op[3], s);
	});
}

// ------------------------------------------------------------------------------------------
// Ed25519 verification: eddsa_import_pub_key (sig/eddsa.c:862) + _eddsa_verify_init (:1846) +
// _eddsa_verify_finalize (:2130) with the hash H(dom || R || A || M) supplied by the caller.
// ------------------------------------------------------------------------------------------
static Big big_inv_p(const Big &a, const Big &p) { return big_powmod(a, big_sub(p, Big(1, 2)), p); }
static Big big_negmod(const Big &a, const Big &p) { return big_mod(big_sub(p, big_mod(a, p)), p); }

// square root for p = 5 mod 8; returns false when n is a non-residue
static bool big_sqrt_5mod8(const Big &n, const Big &p, Big *out)
{
	Big e = big_add(p, Big(1, 3));
	Big q(e.size(), 0);
	for (size_t i = 0; i < e.size(); i++) {
		q[i] = (e[i] >> 3) | ((i + 1 < e.size()) ? (e[i + 1] << 29) : 0u);
	}
	big_trim(q);
	Big c = big_powmod(n, q, p);
	if (big_cmp(big_mulmod(c, c, p), big_mod(n, p)) != 0) {
		c = big_mulmod(c, big_sqrt_m1(p), p);
	}
	*out = c;
	return big_cmp(big_mulmod(c, c, p), big_mod(n, p)) == 0;
}

// constants of the Ed25519 path for this handle; ctx->mu held
static void ed_setup(ecamd_curve *cv)
{
	cv->ed_state = -1;
	if (!(cv->pbits == 255 && cv->clen == 32 && cv->nw == 8 && cv->qslot >= 0 &&
	      big_cmp(cv->p, big_sub(big_pow2(255), Big(1, 19))) == 0)) {
		cv->ed_err = "ec_eddsa_verify_batch: only Ed25519 (the WEI25519 curve handle) is supported";
		return;
	}
	const Big &p = cv->p;
	const Big one(1, 1), two(1, 2), three(1, 3);
	const Big A(1, 486662);
	const Big A3 = big_mulmod(A, big_inv_p(three, p), p);
	// edwards25519: a = -1, d = -121665/121666; alpha_edwards^2 = -(A + 2) (B = 1 Montgomery model)
	const Big a_ed = big_sub(p, one);
	const Big d_ed = big_negmod(big_mulmod(Big(1, 121665), big_inv_p(Big(1, 121666), p), p), p);
	Big alpha;
	if (!big_sqrt_5mod8(big_negmod(big_add(A, two), p), p, &alpha)) {
		cv->ed_err = "ec_eddsa_verify_batch: internal: alpha";
		return;
	}
	{
		// the handle must be the Weierstrass model whose generator is the image of the Ed25519 base
		// point (y = 4/5, x even); this also fixes the sign of alpha_edwards
		const Big yb = big_mulmod(Big(1, 4), big_inv_p(Big(1, 5), p), p);
		const Big y2 = big_mulmod(yb, yb, p);
		const Big num = big_mod(big_add(one, big_sub(p, y2)), p);
		const Big den = big_mod(big_add(a_ed, big_sub(p, big_mulmod(d_ed, y2, p))), p);
		Big xb;
		if (!big_sqrt_5mod8(big_mulmod(num, big_inv_p(den, p), p), p, &xb)) {
			cv->ed_err = "ec_eddsa_verify_batch: internal: base point";
			return;
		}
		if (xb[0] & 1u) {
			xb = big_sub(p, xb);
		}
		const Big um = big_mulmod(big_mod(big_add(one, yb), p), big_inv_p(big_mod(big_add(one, big_sub(p, yb)), p), p), p);
		const Big X = big_mod(big_add(um, A3), p);
		Big vm = big_mulmod(big_mulmod(alpha, um, p), big_inv_p(xb, p), p);
		if (big_cmp(vm, cv->gy) != 0) {
			alpha = big_sub(p, alpha);
			vm = big_sub(p, vm);
		}
		if (big_cmp(X, cv->gx) != 0 || big_cmp(vm, cv->gy) != 0) {
			cv->ed_err = "ec_eddsa_verify_batch: the curve generator is not the image of the Ed25519 base point";
			return;
		}
		big_digits29(cv->ed_Bx, 9, xb);
		big_digits29(cv->ed_By, 9, yb);
	}
	cv->ed_cof_dbl = 0xffffffffu;
	{
		Big t = cv->q;
		for (uint32_t c = 0; c <= 4; c++) {
			if (big_cmp(t, cv->order) == 0) {
				cv->ed_cof_dbl = c;
				break;
			}
			t = big_add(t, t);
		}
	}
	if (cv->ed_cof_dbl == 0xffffffffu) {
		cv->ed_err = "ec_eddsa_verify_batch: unexpected cofactor";
		return;
	}
	const int nw = cv->nw;
	const Big R = big_mod(big_pow2(32 * nw), p);
	EcamdEdDecodeArgs &D = cv->ed_tmpl;
	memset(&D, 0, sizeof(D));
	D.len = 32;
	D.slot = cv->slot;
	big_store(D.a, nw, big_mulmod(a_ed, R, p));
	big_store(D.d, nw, big_mulmod(d_ed, R, p));
	big_store(D.sm1, nw, big_mulmod(big_sqrt_m1(p), R, p));
	big_store(D.alpha, nw, big_mulmod(alpha, R, p));
	big_store(D.A3, nw, big_mulmod(A3, R, p));
	big_digits29(D.g_d, 9, d_ed);
	big_digits29(D.g_sm1, 9, big_sqrt_m1(p));
	big_digits29(D.g_alpha, 9, alpha);
	big_digits29(D.g_A3, 9, A3);
	big_digits29(cv->ed_2d, 9, big_mod(big_add(d_ed, d_ed), p));
	cv->ed_state = 1;
}

// Round 4: the 16-bit comb table of the Ed25519 base point on the Edwards curve, T[j][m-1] = [m 2^(16 j)]B (m = 1..32768, j < 16)
// and [2^256]B, as precomputed entries (y - x, y + x, 2d x y), 128 bytes each (67 MB of the 288 GB).  The multiples come from this
// engine's own scalar multiplication on the Weierstrass model -- the batch maybe_build_comb uses -- and k_edcomb_build_c25519 maps
// them to the Edwards curve.  With it k_ed_tail_c25519 finishes a verification without leaving the Edwards curve.  ctx->mu held.
static void maybe_build_edcomb(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n)
{
	if (cv->d_edcomb || cv->edcomb_off || ctx->comb_min_batch == 0 || n < ctx->comb_min_batch || cv->gflavour != 2 || cv->gslot < 0) {
		return;
	}
	cv->edcomb_off = true;   // stays set if anything fails
	if (hipDeviceSynchronize() != hipSuccess) {   // the build borrows the context's scratch (see maybe_build_comb)
		return;
	}
	const uint32_t slen = 32, nwin = 16, ne = ECAMD_EDC_ENTRIES;
	std::vector<uint8_t> hs((size_t)ne * slen, 0);
	for (uint32_t j = 0; j < nwin; j++) {
		for (uint32_t m = 1; m <= 32768; m++) {
			uint8_t *e = &hs[((size_t)j * 32768 + (m - 1)) * slen];
			e[slen - 1 - 2 * j] = (uint8_t)(m & 0xff);
			e[slen - 2 - 2 * j] = (uint8_t)(m >> 8);
		}
	}
	big_to_be(&hs[(size_t)nwin * 32768 * slen], (int)slen, big_mod(big_pow2(256), cv->q));
	uint8_t *dsc = nullptr, *dpt = nullptr, *dst = nullptr;
	uint32_t *table = nullptr;
	std::vector<uint8_t> st(ne);
	hipStream_t s = ctx->stream;
	bool ok = hipMalloc((void **)&dsc, hs.size()) == hipSuccess && hipMalloc((void **)&dpt, (size_t)ne * 64) == hipSuccess &&
		  hipMalloc((void **)&dst, ne) == hipSuccess && hipMalloc((void **)&table, (size_t)ne * ECAMD_EDC_ENT_WORDS * 4) == hipSuccess &&
		  hipMemcpy(dsc, hs.data(), hs.size(), hipMemcpyHostToDevice) == hipSuccess;
	{
		const uint32_t user_chunk = ctx->max_chunk;
		ctx->max_chunk = user_chunk < (1u << 20) ? (1u << 20) : user_chunk;
		ok = ok && smul_dev_locked(ctx, cv, ne, dsc, slen, nullptr, dpt, dst, s) == 0;
		ctx->max_chunk = user_chunk;
	}
	if (ok) {
		EcamdEdTailConsts C;
		memcpy(C.g_2d, cv->ed_2d, sizeof(C.g_2d));
		memcpy(C.g_alpha, cv->ed_tmpl.g_alpha, sizeof(C.g_alpha));
		memcpy(C.g_A3, cv->ed_tmpl.g_A3, sizeof(C.g_A3));
		ok = ecamd_launch_edcomb_build_c25519(dpt, ne, table, C, cv->gslot, s) == hipSuccess &&
		     hipMemcpyAsync(st.data(), dst, ne, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
	}
	for (uint32_t i = 0; ok && i < ne; i++) {
		ok = (st[i] == 0);
	}
	(void)hipFree(dsc);
	(void)hipFree(dpt);
	(void)hipFree(dst);
	if (!ok) {
		(void)hipFree(table);
		(void)hipGetLastError();
		return;   // the Weierstrass tail keeps serving
	}
	cv->d_edcomb = table;
	cv->edcomb_off = false;
}

// device pointers in and out; only enqueues on s
static int eddsa_verify_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig,
				   const uint8_t *d_hram, uint32_t hram_len, uint8_t *d_res, hipStream_t s)
{
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (eddsa_verify_dev_locked(ctx, cv, m, d_pub + (size_t)off * 32, d_sig + (size_t)off * 64,
						    d_hram + (size_t)off * hram_len, hram_len, d_res + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	const size_t len = 32, plen = 64;
	const uint32_t cof_dbl = cv->ed_cof_dbl;
	// stage: 3 A (Weierstrass), 4 R, 5 flagsA, 6 flagsR, 7 flagsS, 8 S, 9 h, 12 [h]A, 13 sthA, 14 [S]G, 15 stSG
	//        (0..2 and 16 belong to the host-pointer wrapper)
	const size_t need[ECAMD_NSTAGE] = {0, 0, 0, n * plen, n * plen, n, n, n, n * len, n * len,
					   0, 0, n * plen, n, n * plen, n, 0, 0};
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	const int nw = cv->nw;
	EcamdEdDecodeArgs D = cv->ed_tmpl;
	D.n = n;
	D.encA = d_pub;
	D.strideA = (uint32_t)len;
	D.encR = d_sig;
	D.strideR = (uint32_t)plen;
	D.pointsA = S[3];
	D.pointsR = S[4];
	D.flagsA = S[5];
	D.flagsR = S[6];
	D.cof_dbl = cof_dbl;
	const bool unit25519 = cv->gflavour == 2 && cv->gslot >= 0;
	const bool edwards_hA = unit25519 && getenv("ECAMD_NO_EDWARDS_SMUL") == nullptr;
	D.edA = nullptr;
	if (edwards_hA) {
		// stage 10: A on the Edwards curve, 11: extended result records, 18: window tables
		if (ensure(&ctx->stage[10], &ctx->stage_bytes[10], (size_t)n * 20 * 4) ||
		    ensure(&ctx->stage[11], &ctx->stage_bytes[11], (size_t)n * ECAMD_EDR_REC_WORDS * 4) ||
		    ensure(&ctx->stage[18], &ctx->stage_bytes[18], (size_t)n * ECAMD_EDT_ITEM_WORDS * 4)) {
			return -1;
		}
		D.edA = (uint32_t *)S[10];
	}
	// Edwards path: R's map to the Weierstrass model rides on the shared inversion of k_ed_hA_fin instead of costing the
	// decode kernel an inversion per item (stage 19: R on the Edwards curve)
	const bool late_map = edwards_hA && getenv("ECAMD_NO_ED_LATE_MAP") == nullptr;
	// round 4: the whole tail on the Edwards curve ([S]B from the Edwards comb table; no map, no inversion, no Weierstrass pass)
	bool ed_tail = false;
	if (late_map && getenv("ECAMD_NO_ED_TAIL") == nullptr) {
		maybe_build_edcomb(ctx, cv, n);
		ed_tail = cv->d_edcomb != nullptr;
	}
	// round 4, second step: half-length scalars (k_ed_lat) and one window loop over the tables of A and R (k_ed_smul2_c25519)
	const bool lattice = ed_tail && len == 32 && getenv("ECAMD_NO_ED_LATTICE") == nullptr;
	if (lattice && ensure(&ctx->stage[18], &ctx->stage_bytes[18], (size_t)n * 2 * ECAMD_EDT_ITEM_WORDS * 4)) {
		return -1;
	}
	// k_ed_lat (S < q, h mod q, the truncated Euclid: integer work with data-dependent trip counts, 0.43 VALU busy) beside the decode
	// kernel (two square-root chains, MAD bound) on the side stream: both read only the caller's arrays.  $ECAMD_NO_ED_LAT_BESIDE: in line.
	bool lat_beside = false;
	EcamdEdLatArgs L;   // stage 3 (n x 64, free on this path): v and |u|, 12 words per item; 9: s' = u S mod q; 13: meta
	memset(&L, 0, sizeof(L));
	if (lattice) {
		L.sigs = d_sig;
		L.hram = d_hram;
		L.S_be = S[8];
		L.sp_be = S[9];
		L.uv = (uint32_t *)S[3];
		L.meta = S[13];
		L.flags = S[7];
		L.n = n;
		L.hlen = hram_len;
		L.qslot = cv->qslot;
		if (ctx->side_ok && getenv("ECAMD_NO_ED_LAT_BESIDE") == nullptr) {
			HIPCHK(hipEventRecord(ctx->side_fork, s));
			HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
			HIPCHK(ecamd_launch_ed_lat(L, ctx->side_stream));
			HIPCHK(hipEventRecord(ctx->side_done, ctx->side_stream));
			lat_beside = true;
		}
	}
	if (late_map) {
		if (ensure(&ctx->stage[19], &ctx->stage_bytes[19], (size_t)n * 20 * 4)) {
			return -1;
		}
		D.edR = (uint32_t *)ctx->stage[19];
		HIPCHK(ecamd_launch_ed_decode_ed_c25519(D, cv->gslot, s));
	} else if (unit25519) {
		HIPCHK(ecamd_launch_ed_decode_c25519(D, cv->gslot, s));  // the same decoding on the radix-2^29 field
	} else {
		HIPCHK(ecamd_launch_ed_decode(nw, D, s));
	}
	if (lattice) {
		if (lat_beside) {
			HIPCHK(hipStreamWaitEvent(s, ctx->side_done, 0));
		} else {
			HIPCHK(ecamd_launch_ed_lat(L, s));
		}
		EcamdEdSmul2Args E;
		memset(&E, 0, sizeof(E));
		E.edA = (const uint32_t *)S[10];
		E.edR = (const uint32_t *)ctx->stage[19];
		E.flagsA = S[5];
		E.flagsR = S[6];
		E.flagsS = S[7];
		E.uv = (const uint32_t *)S[3];
		E.meta = S[13];
		E.tbl = (uint32_t *)ctx->stage[18];
		E.rec = (uint32_t *)S[11];
		E.n = n;
		memcpy(E.g_2d, cv->ed_2d, sizeof(E.g_2d));
		HIPCHK(ecamd_launch_ed_smul2_c25519(E, cv->gslot, s, ctx->timing ? ctx->ev_dom : nullptr));
		ctx->ev_dom_valid = ctx->ev_dom_valid || ctx->timing;
		EcamdEdTailArgs T;
		memset(&T, 0, sizeof(T));
		T.rec = (const uint32_t *)S[11];
		T.edR = (const uint32_t *)ctx->stage[19];
		T.flagsA = S[5];
		T.flagsR = S[6];
		T.flagsS = S[7];
		T.S_be = S[8];
		T.sp_be = S[9];
		T.meta = S[13];
		T.comb = cv->d_edcomb;
		T.result = d_res;
		T.n = n;
		T.cof_dbl = cof_dbl;
		memcpy(T.C.g_2d, cv->ed_2d, sizeof(T.C.g_2d));
		HIPCHK(ecamd_launch_ed_tail2_c25519(T, cv->gslot, s));
		return 0;
	}
	EcamdEdScalArgs C;
	memset(&C, 0, sizeof(C));
	C.sigs = d_sig;
	C.hram = d_hram;
	C.S_be = S[8];
	C.h_be = S[9];
	C.flags = S[7];
	C.n = n;
	C.len = (uint32_t)len;
	C.hlen = hram_len;
	C.qslot = cv->qslot;
	HIPCHK(ecamd_launch_ed_scal(nw, C, s));
	// [h]A, [S]G  ([8]A != infinity was checked by the decode kernel)
	if (edwards_hA) {
		// [h]A on the Edwards curve itself (complete extended-coordinate formulas), mapped to the Weierstrass model
		EcamdEdSmulArgs E;
		memset(&E, 0, sizeof(E));
		E.edA = (const uint32_t *)S[10];
		E.scalars = S[9];
		E.flags = S[5];
		E.tbl = (uint32_t *)S[18];
		E.rec = (uint32_t *)S[11];
		E.out = S[12];
		E.status = S[13];
		E.n = n;
		memcpy(E.g_2d, cv->ed_2d, sizeof(E.g_2d));
		memcpy(E.g_alpha, cv->ed_tmpl.g_alpha, sizeof(E.g_alpha));
		memcpy(E.g_A3, cv->ed_tmpl.g_A3, sizeof(E.g_A3));
		if (late_map) {
			E.edR = (const uint32_t *)ctx->stage[19];
			E.flagsR = S[6];
			E.outR = S[4];
		}
		if (ed_tail) {
			E.out = nullptr;   // no k_ed_hA_fin: [h]A stays in E.rec
		}
		HIPCHK(ecamd_launch_ed_smul_c25519(E, cv->gslot, s, ctx->timing ? ctx->ev_dom : nullptr));
		ctx->ev_dom_valid = ctx->ev_dom_valid || ctx->timing;
	}
	if (ed_tail) {
		EcamdEdTailArgs T;
		memset(&T, 0, sizeof(T));
		T.rec = (const uint32_t *)S[11];
		T.edR = (const uint32_t *)ctx->stage[19];
		T.flagsA = S[5];
		T.flagsR = S[6];
		T.flagsS = S[7];
		T.S_be = S[8];
		T.comb = cv->d_edcomb;
		T.result = d_res;
		T.n = n;
		T.cof_dbl = cof_dbl;
		memcpy(T.C.g_2d, cv->ed_2d, sizeof(T.C.g_2d));
		HIPCHK(ecamd_launch_ed_tail_c25519(T, cv->gslot, s));
		return 0;
	}
	{
		PublicScalars pub_scope(ctx);   // h and S of a signature are public
		if ((!edwards_hA && smul_dev_locked(ctx, cv, n, S[9], (uint32_t)len, S[3], S[12], S[13], s)) ||
		    smul_dev_locked(ctx, cv, n, S[8], (uint32_t)len, nullptr, S[14], S[15], s)) {
			return -1;
		}
	}
	EcamdEdFinArgs F;
	memset(&F, 0, sizeof(F));
	F.SG = S[14];
	F.stSG = S[15];
	F.hA = S[12];
	F.sthA = S[13];
	F.R = S[4];
	F.flagsA = S[5];
	F.flagsR = S[6];
	F.flagsS = S[7];
	F.result = d_res;
	F.n = n;
	F.clen = (uint32_t)len;
	F.cof_dbl = cof_dbl;
	F.slot = cv->slot;
	if (cv->gflavour == 2 && cv->gslot >= 0 && getenv("ECAMD_NO_ED_FIN_G") == nullptr) {
		HIPCHK(ecamd_launch_ed_fin_g29(2, cv->gslot, F, s));   // the same tail on the 2^255 - 19 unit (k_ed_fin_g)
	} else {
		HIPCHK(ecamd_launch_ed_fin(nw, F, s));
	}
	return 0;
}

// ---- Ed448 (EDDSA448 branches of sig/eddsa.c) on the WEI448 handle ----
static Big big_div4(const Big &x)
{
	Big q(x.size(), 0);
	for (size_t i = 0; i < x.size(); i++) {
		q[i] = (x[i] >> 2) | ((i + 1 < x.size()) ? (x[i + 1] << 30) : 0u);
	}
	big_trim(q);
	return q;
}

static void ed448_setup(ecamd_curve *cv)
{
	cv->ed448_state = -1;
	const Big p448 = big_sub(big_sub(big_pow2(448), big_pow2(224)), Big(1, 1));
	if (!(cv->pbits == 448 && cv->clen == 56 && cv->nw == 14 && cv->qslot >= 0 && cv->qnw == 14 && cv->qlen == 56 &&
	      big_cmp(cv->p, p448) == 0)) {
		cv->ed448_err = "ec_eddsa_verify_batch: Ed448 needs the WEI448 curve handle";
		return;
	}
	const Big &p = cv->p;
	const Big one(1, 1), two(1, 2), three(1, 3);
	const Big A(1, 156326);
	const Big A3 = big_mulmod(A, big_inv_p(three, p), p);
	const Big d448 = big_sub(p, Big(1, 39081));
	const Big e4 = big_div4(big_add(p, one));                    // (p + 1) / 4: square roots for p = 3 mod 4
	Big alpha = big_powmod(Big(1, 156324), e4, p);               // alpha^2 = A - 2
	if (big_cmp(big_mulmod(alpha, alpha, p), Big(1, 156324)) != 0) {
		cv->ed448_err = "ec_eddsa_verify_batch: internal: alpha";
		return;
	}
	const Big diso = big_mulmod(Big(1, 156328), big_inv_p(Big(1, 156324), p), p);
	{
		// the Ed448 base point (RFC 8032) must map to the generator of the handle; this fixes the sign of alpha
		static const uint8_t yb_le[56] = {
			0x14, 0xfa, 0x30, 0xf2, 0x5b, 0x79, 0x08, 0x98, 0xad, 0xc8, 0xd7, 0x4e, 0x2c, 0x13, 0xbd, 0xfd, 0xc4, 0x39, 0x7c,
			0xe6, 0x1c, 0xff, 0xd3, 0x3a, 0xd7, 0xc2, 0xa0, 0x05, 0x1e, 0x9c, 0x78, 0x87, 0x40, 0x98, 0xa3, 0x6c, 0x73, 0x73,
			0xea, 0x4b, 0x62, 0xc7, 0xc9, 0x56, 0x37, 0x20, 0x76, 0x88, 0x24, 0xbc, 0xb6, 0x6e, 0x71, 0x46, 0x3f, 0x69};
		uint8_t yb_be[56];
		for (int i = 0; i < 56; i++) {
			yb_be[i] = yb_le[55 - i];
		}
		const Big y = big_from_be(yb_be, 56);
		const Big yy = big_mulmod(y, y, p);
		auto sub = [&](const Big &a, const Big &b) { return big_mod(big_add(a, big_sub(p, big_mod(b, p))), p); };
		const Big t = big_mulmod(sub(one, yy), big_inv_p(sub(one, big_mulmod(d448, yy, p)), p), p);
		Big x = big_powmod(t, e4, p);
		if (big_cmp(big_mulmod(x, x, p), t) != 0) {
			cv->ed448_err = "ec_eddsa_verify_batch: internal: base point";
			return;
		}
		if (x[0] & 1u) {
			x = big_sub(p, x);                                   // the encoding's sign bit is 0
		}
		const Big xx = big_mulmod(x, x, p);
		const Big X = big_mulmod(big_mulmod(alpha, big_mulmod(x, y, p), p), big_inv_p(sub(sub(two, xx), yy), p), p);
		const Big Y = big_mulmod(big_mod(big_add(xx, yy), p), big_inv_p(sub(yy, xx), p), p);
		const Big u = big_mulmod(big_mod(big_add(one, Y), p), big_inv_p(sub(one, Y), p), p);
		Big v = big_mulmod(big_mulmod(alpha, u, p), big_inv_p(X, p), p);
		const Big Xw = sub(A3, u);
		Big Yw = sub(Big(1, 0), v);
		if (big_cmp(Yw, cv->gy) != 0) {
			alpha = big_sub(p, alpha);
			Yw = sub(Big(1, 0), Yw);
		}
		if (big_cmp(Xw, cv->gx) != 0 || big_cmp(Yw, cv->gy) != 0) {
			cv->ed448_err = "ec_eddsa_verify_batch: the curve generator is not the image of the Ed448 base point";
			return;
		}
	}
	{
		Big t = big_add(cv->q, cv->q);
		t = big_add(t, t);
		if (big_cmp(t, cv->order) != 0) {
			cv->ed448_err = "ec_eddsa_verify_batch: unexpected cofactor";
			return;
		}
	}
	Big c4;
	for (uint32_t j = 1; j <= 3; j++) {
		Big jq = big_add(big_mul(cv->q, Big(1, j)), one);
		if ((jq[0] & 3u) == 0) {
			c4 = big_div4(jq);
			break;
		}
	}
	big_to_be(cv->ed448_c4, 56, c4);
	const int nw = cv->nw;
	const Big R = big_mod(big_pow2(32 * nw), p);
	EcamdEd448DecodeArgs &D = cv->ed448_tmpl;
	memset(&D, 0, sizeof(D));
	D.slot = cv->slot;
	big_store(D.d448, nw, big_mulmod(d448, R, p));
	big_store(D.diso, nw, big_mulmod(diso, R, p));
	big_store(D.alpha, nw, big_mulmod(alpha, R, p));
	big_store(D.A3, nw, big_mulmod(A3, R, p));
	big_digits29(D.g_d448, 16, d448, 28);     // the Goldilocks unit's 28-bit limbs
	big_digits29(D.g_diso, 16, diso, 28);
	big_digits29(D.g_alpha, 16, alpha, 28);
	big_digits29(D.g_A3, 16, A3, 28);
	cv->ed448_state = 1;
}

// device pointers in and out; pubs n x 57, sigs n x 114, hram n x 114
static int eddsa448_verify_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig,
				      const uint8_t *d_hram, uint8_t *d_res, hipStream_t s)
{
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (eddsa448_verify_dev_locked(ctx, cv, m, d_pub + (size_t)off * 57, d_sig + (size_t)off * 114,
						       d_hram + (size_t)off * 114, d_res + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	const size_t len = 56, plen = 112;
	// stage: 3 A (Weierstrass), 4 R, 5 flagsA, 6 flagsR, 7 flagsS, 8 S, 9 k = 4h 4^-1 mod 4q, 12 [k]A, 13 its status,
	//        14 [S]G, 15 stSG
	const size_t need[ECAMD_NSTAGE] = {0, 0, 0, n * plen, n * plen, n, n, n, n * len, n * len,
					   0, 0, n * plen, n, n * plen, n};
	for (int i = 0; i < ECAMD_NSTAGE; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	EcamdEd448DecodeArgs D = cv->ed448_tmpl;
	D.n = n;
	D.encA = d_pub;
	D.strideA = 57;
	D.encR = d_sig;
	D.strideR = 114;
	D.pointsA = S[3];
	D.pointsR = S[4];
	D.flagsA = S[5];
	D.flagsR = S[6];
	if (cv->gflavour == 5 && cv->gslot >= 0 && getenv("ECAMD_NO_G448_DECODE") == nullptr) {
		HIPCHK(ecamd_launch_ed448_decode_g(D, cv->gslot, s));   // the same decoding on the Goldilocks radix-2^29 field
	} else {
		HIPCHK(ecamd_launch_ed448_decode(D, s));
	}
	EcamdEdScalArgs C;
	memset(&C, 0, sizeof(C));
	C.sigs = d_sig;
	C.hram = d_hram;
	C.S_be = S[8];
	C.h_be = S[9];
	C.flags = S[7];
	C.n = n;
	C.len = 57;
	C.hlen = 114;
	C.qslot = cv->qslot;
	C.c4_mod4 = cv->ed448_c4[55] & 3u;
	HIPCHK(ecamd_launch_ed448_scal(C, s));
	// the reference's [4h mod q]([4^-1 mod q]A) as one multiplication of A (k_ed448_scal), and [S]G
	{
		PublicScalars pub_scope(ctx);   // h and S of a signature are public
		if (smul_dev_locked(ctx, cv, n, S[9], (uint32_t)len, S[3], S[12], S[13], s) ||
		    smul_dev_locked(ctx, cv, n, S[8], (uint32_t)len, nullptr, S[14], S[15], s)) {
			return -1;
		}
	}
	EcamdEdFinArgs F;
	memset(&F, 0, sizeof(F));
	F.SG = S[14];
	F.stSG = S[15];
	F.hA = S[12];
	F.sthA = S[13];
	F.R = S[4];
	F.flagsA = S[5];
	F.flagsR = S[6];
	F.flagsS = S[7];
	F.result = d_res;
	F.n = n;
	F.clen = (uint32_t)len;
	F.cof_dbl = 2;
	F.Akey = S[3];      // [4]A = infinity <=> the stored key [4^-1 mod q]A has small order
	F.stA = nullptr;
	F.slot = cv->slot;
	if (cv->gflavour == 5 && cv->gslot >= 0 && getenv("ECAMD_NO_ED_FIN_G") == nullptr) {
		HIPCHK(ecamd_launch_ed_fin_g29(5, cv->gslot, F, s));   // the same tail on the Goldilocks unit (k_ed_fin_g)
	} else {
		HIPCHK(ecamd_launch_ed_fin(cv->nw, F, s));
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// Round 6: Ed448 whole-batch verification (SURVEY.md 8 row f4 for EDDSA448 / EDDSA448PH; _eddsa_verify_batch, sig/eddsa.c:2580-2860, serves both
// EdDSA curves).  The reference's combination  sum [cof z_i]R_i + sum [z_i 4 h_i][cof]A'_i + [-sum z_i S_i][cof]G = infinity  on the Weierstrass
// model is, on the prime-order components, the Schnorr-type equation  [sum z_i S_i]G + sum [z_i (q - h_i)]A_i - sum [z_i]R_i  of
// schnorr_msm_dev_locked with its final test cofactored: the decoding of eddsa448_verify_dev_locked (A_i, R_i as affine points of WEI448), S_i
// and (q - h_i) mod q from k_ed448_scal, the reference's per-item rejections as one gate word (k_ed_msm_gate), then the multi-scalar
// multiplication on the Goldilocks unit -- by buckets from 2^17 items on, the Straus loop below -- ending in [4](sum + [c]G) = infinity.
// Soundness as for the item form: [4] kills every torsion component, so scalars taken mod q and the decoded (not the stored [4^-1]A) key are
// exact; a batch with an item that fails its cofactored equation passes with probability ~2^-128.  d_verdict[0] is SET to 1 for "not
// decided here" and never cleared (the pieces of one call share it).  Only enqueues.
// ------------------------------------------------------------------------------------------
static int msm_seed(ecamd_ctx *ctx, uint8_t seed[32]);
// the bucket evaluation FILED CHUNK BY CHUNK (ec_schnorr_verify_msg_all_batch): which part of schnorr_msm_dev_locked a call runs
struct SchnorrStreamStep {
	int mode;                // 1: scratch sized, flag word and counters cleared; 2: the items [first, first + count) imported, their scalars
	                         // formed, filed; 3: c and [c]G, the ranking, the buckets summed, reduced and compared
	uint32_t first, count;
};
static bool schnorr_msm_unit(const ecamd_curve *cv, int *pbits, int *flavour, int *slot);
static int schnorr_msm_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_s, const uint8_t *d_ne, const uint8_t *d_keys,
				  const uint8_t *d_r, int r_fmt, const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, uint8_t *d_z_dump,
				  uint32_t *d_sum_dump, hipStream_t s, uint32_t cof_dbl = 0, const struct SchnorrStreamStep *step = nullptr);
static bool eddsa448_msm_available(const ecamd_curve *cv)
{
	int pb, fl, sl;
	return cv->pbits == 448 && cv->ed448_state == 1 && cv->nw == 14 && cv->cofactor == 4 && cv->qslot >= 0 && cv->qbits >= 160 &&
	       schnorr_msm_unit(cv, &pb, &fl, &sl);
}

static int eddsa448_msm_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
				   const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, hipStream_t s)
{
	if (n == 0 || n > ctx->max_chunk) {
		return fail("internal: eddsa448_msm_dev_locked serves one piece of at most max_chunk items");
	}
	const size_t len = 56, plen = 112;
	// stage: 3 A (Weierstrass, affine), 4 R, 5 flagsA, 6 flagsR, 7 flagsS, 8 S, 9 k (unused here), 12 (q - h) mod q, 13 gate word + piece verdict
	size_t need[ECAMD_NSTAGE] = {0};
	need[3] = need[4] = n * plen;
	need[5] = need[6] = need[7] = n;
	need[8] = need[9] = need[12] = n * len;
	need[13] = 256;
	for (int i = 0; i < 16; i++) {
		if (need[i] && ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	EcamdEd448DecodeArgs D = cv->ed448_tmpl;
	D.n = n;
	D.encA = d_pub;
	D.strideA = 57;
	D.encR = d_sig;
	D.strideR = 114;
	D.pointsA = S[3];
	D.pointsR = S[4];
	D.flagsA = S[5];
	D.flagsR = S[6];
	if (cv->gflavour == 5 && cv->gslot >= 0 && getenv("ECAMD_NO_G448_DECODE") == nullptr) {
		HIPCHK(ecamd_launch_ed448_decode_g(D, cv->gslot, s));
	} else {
		HIPCHK(ecamd_launch_ed448_decode(D, s));
	}
	EcamdEdScalArgs C;
	memset(&C, 0, sizeof(C));
	C.sigs = d_sig;
	C.hram = d_hram;
	C.S_be = S[8];
	C.h_be = S[9];
	C.ne_be = S[12];
	C.flags = S[7];
	C.n = n;
	C.len = 57;
	C.hlen = 114;
	C.qslot = cv->qslot;
	C.c4_mod4 = cv->ed448_c4[55] & 3u;
	HIPCHK(ecamd_launch_ed448_scal(C, s));
	uint32_t *d_gate = (uint32_t *)S[13];
	uint8_t *d_piece = S[13] + 16;
	HIPCHK(hipMemsetAsync(S[13], 0, 16, s));
	HIPCHK(hipMemsetAsync(d_piece, 1, 1, s));
	// (the keys of small order, [4]A = infinity: seen by the combination's own import of the keys on the fast unit -- cof_dbl of EcamdMsmArgs)
	HIPCHK(ecamd_launch_ed_msm_gate(cv->nw, S[3], S[5], S[6], S[7], n, (uint32_t)len, 0, cv->slot, d_gate, s));
	{
		PublicScalars pub_scope(ctx);   // everything a verification multiplies by is public
		if (schnorr_msm_dev_locked(ctx, cv, n, S[8], S[12], S[3], S[4], 0, seed, piece, d_piece, nullptr, nullptr, s, 2)) {
			return -1;
		}
	}
	HIPCHK(ecamd_launch_verdict_or(d_verdict, d_piece, d_gate, s));
	return 0;
}

// host arrays -> one verdict: pieces of max_chunk items, each with its own combination.  *accept = 1 when every piece accepts.
static int eddsa448_msm_host_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs, const uint8_t *hram,
				    int *accept)
{
	uint8_t seed[32];
	if (msm_seed(ctx, seed)) {
		return -1;
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)chunk * 57) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], (size_t)chunk * 114) ||
	    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)chunk * 114) || ensure(&ctx->stage[14], &ctx->stage_bytes[14], 256)) {
		return -1;
	}
	uint8_t *d_verdict = ctx->stage[14];
	HIPCHK(hipMemsetAsync(d_verdict, 0, 1, s));
	for (uint32_t off = 0, pc = 0; off < n; off += chunk, pc++) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		HIPCHK(hipMemcpyAsync(ctx->stage[0], pubkeys + (size_t)off * 57, (size_t)m * 57, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[1], sigs + (size_t)off * 114, (size_t)m * 114, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[2], hram + (size_t)off * 114, (size_t)m * 114, hipMemcpyHostToDevice, s));
		if (eddsa448_msm_dev_locked(ctx, cv, m, ctx->stage[0], ctx->stage[1], ctx->stage[2], seed, pc, d_verdict, s)) {
			(void)hipStreamSynchronize(s);
			return -1;
		}
	}
	uint8_t v = 1;
	HIPCHK(hipMemcpyAsync(&v, d_verdict, 1, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	memset(seed, 0, sizeof(seed));
	*accept = v == 0;
	return 0;
}

static int eddsa_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *a, const void *b,
			 const void *c, const void *d, uint32_t hram_len)
{
	if (!ctx || !cv_in || cv_in->ctx != ctx || (n && (!a || !b || !c || !d))) {
		return fail(std::string(fn) + ": bad argument");
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (cv->pbits == 448) {
		if (cv->ed448_state == 0) {
			ed448_setup(cv);
		}
		if (cv->ed448_state < 0) {
			return fail(cv->ed448_err);
		}
		if (hram_len != 114) {
			return fail(std::string(fn) + ": Ed448 hashes with SHAKE256 (114 bytes): hram_len must be 114");
		}
		return 0;
	}
	if (cv->ed_state == 0) {
		ed_setup(cv);
	}
	if (cv->ed_state < 0) {
		return fail(cv->ed_err);
	}
	if (hram_len != 64) {
		return fail(std::string(fn) + ": Ed25519 hashes with SHA-512: hram_len must be 64");
	}
	return 0;
}

extern "C" int ec_eddsa_verify_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *d_pubkeys,
					 const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_result,
					 void *hip_stream)
{
	if (!ctx) {
		return fail("ec_eddsa_verify_batch_dev: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (eddsa_args_ok("ec_eddsa_verify_batch_dev", ctx, cv_in, n, d_pubkeys, d_sigs, d_hram, d_result, hram_len)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	if (cv_in->pbits == 448) {
		return eddsa448_verify_dev_locked(ctx, const_cast<ecamd_curve *>(cv_in), n, (const uint8_t *)d_pubkeys,
						  (const uint8_t *)d_sigs, (const uint8_t *)d_hram, (uint8_t *)d_result, s);
	}
	return eddsa_verify_dev_locked(ctx, const_cast<ecamd_curve *>(cv_in), n, (const uint8_t *)d_pubkeys,
				       (const uint8_t *)d_sigs, (const uint8_t *)d_hram, hram_len, (uint8_t *)d_result, s);
}

extern "C" int ec_eddsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *pubkeys,
				     const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, uint8_t *result)
{
	if (!ctx) {
		return fail("ec_eddsa_verify_batch: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (eddsa_args_ok("ec_eddsa_verify_batch", ctx, cv_in, n, pubkeys, sigs, hram, result, hram_len)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	const bool e448 = cv->pbits == 448;
	const std::vector<HostArr> arrs = {{pubkeys, nullptr, e448 ? 57u : 32u}, {sigs, nullptr, e448 ? 114u : 64u},
					   {hram, nullptr, hram_len}, {nullptr, result, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return e448 ? eddsa448_verify_dev_locked(ctx, cv, m, ip[0], ip[1], ip[2], op[3], s)
			    : eddsa_verify_dev_locked(ctx, cv, m, ip[0], ip[1], ip[2], hram_len, op[3], s);
	});
}

// Ed25519 verification with the hash INPUTS instead of the hashes: slot i (stride bytes: u32 length, then the bytes) holds
// dom2 || R || A || PH(M) as the verifier would feed them to SHA-512 (sig/eddsa.c:1995-2045, :2180-2200); hram is computed on the device.
extern "C" int ec_eddsa_verify_msg_batch(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
					 const uint8_t *hash_slots, uint32_t stride, uint8_t *result)
{
	if (!ctx || stride < 4 || (stride & 3u) || stride > 4096) {
		return fail("ec_eddsa_verify_msg_batch: bad argument (stride: a multiple of 4 in 4 .. 4096)");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (eddsa_args_ok("ec_eddsa_verify_msg_batch", ctx, cv_in, n, pubkeys, sigs, hash_slots, result, 64)) {
		return -1;
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (cv->pbits != 255) {
		return fail("ec_eddsa_verify_msg_batch: Ed25519 (the WEI25519 handle) only: Ed448 hashes with SHAKE256");
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const std::vector<HostArr> arrs = {{pubkeys, nullptr, 32u}, {sigs, nullptr, 64u}, {hash_slots, nullptr, stride}, {nullptr, result, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		if (ecdsa_hash_stage(ctx, 4, m, ip[2], stride, 64, s)) {
			return -1;
		}
		return eddsa_verify_dev_locked(ctx, cv, m, ip[0], ip[1], ctx->stage[17], 64, op[3], s);
	});
}

static int eddsa_sign_setup(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *a, const void *b,
			    const void *c, EcamdEdSignArgs *T);
// The same from the PROJECTIVE key an ec_pub_key holds (X || Y || Z on WEI25519): what a verifier that starts from libecc structures
// needs -- the reference hashes the key's Ed25519 ENCODING (eddsa_export_pub_key, sig/eddsa.c:795: Weierstrass -> Edwards -> octets), which
// ec_eddsa_encode_point_batch computes; here that encoding goes straight into the item's hash input on the device (bytes a_offset ..
// a_offset + 32 of the slot's message, which the caller leaves blank; the caller's array itself is not written), so one call replaces encode / copy back / build the inputs /
// verify.  A key that does not import (coordinates >= p, not on the curve) or is the point at infinity rejects its item.
// the encode step of EcamdEdSignArgs (Rw, stR -> out, status): on the 2^255 - 19 unit when the handle has it (k_ed_enc_c25519; round 6), else the
// saturated-word kernel ($ECAMD_NO_ED_ENC_G: always the latter)
static hipError_t launch_ed_sign_enc(const ecamd_curve *cv, const EcamdEdSignArgs &A, hipStream_t s)
{
	if (!A.is448 && cv->pbits == 255 && cv->ed_state == 1 && cv->gflavour == 2 && cv->gslot >= 0 && getenv("ECAMD_NO_ED_ENC_G") == nullptr) {
		EcamdEdTailConsts C;
		memcpy(C.g_2d, cv->ed_2d, sizeof(C.g_2d));
		memcpy(C.g_alpha, cv->ed_tmpl.g_alpha, sizeof(C.g_alpha));
		memcpy(C.g_A3, cv->ed_tmpl.g_A3, sizeof(C.g_A3));
		return ecamd_launch_ed_enc_c25519(A.Rw, A.stR, A.out, A.status, A.n, C, cv->gslot, s);
	}
	return ecamd_launch_ed_sign_enc(A, s);
}
struct EdBktRun {
	EcamdEdMsmArgs A;
	EcamdEdBktArgs B;
	EcamdEdMsmScalArgs C;
	EcamdEdMsmLaneArgs N;
	uint32_t *flagword;
	uint32_t n;
};
static bool eddsa_msm_use_buckets(uint32_t n);
static int eddsa_bkt_stream_begin(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
				  const uint8_t seed[32], EdBktRun &R, hipStream_t s);
static int eddsa_bkt_stream_chunk(ecamd_curve *cv, EdBktRun &R, uint32_t first, uint32_t count, hipStream_t s);
static int eddsa_bkt_stream_end(ecamd_ctx *ctx, ecamd_curve *cv, EdBktRun &R, uint8_t *d_verdict, hipStream_t s);
static bool eddsa_msm_available(const ecamd_curve *cv);
static bool eddsa448_msm_available(const ecamd_curve *cv);
static int eddsa448_msm_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
				   const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, hipStream_t s);
static int msm_seed(ecamd_ctx *ctx, uint8_t seed[32]);
static void msm_seed_discard(ecamd_ctx *ctx);
static int eddsa_msm_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig,
				const uint8_t *d_hram, const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, uint8_t *d_z_dump,
				uint32_t *d_sum_dump, hipStream_t s);
// all_valid != NULL (round 6, Ed25519 only): ONE accept bit instead of per-item results -- the front end of every chunk (import, encoding, hashes)
// files the encoded keys, the signatures and the hashes in batch-wide arrays (stage 24 - 26; 27: per-item "the key has no encoding"), and when
// the last chunk is in the batch equation is evaluated once per max_chunk items (eddsa_msm_dev_locked: by buckets from 2^18 items on).
// *all_valid = 0: not decided here (ec_eddsa_verify_all_batch's contract without its item-by-item pass: the caller has one).
static int eddsa_verify_msg_prj_impl(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				     const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
				     uint8_t *result, int *all_valid = nullptr)
{
	const std::string f(fn);
	if (!ctx || !cv_in) {
		return fail(f + ": bad argument");
	}
	// Ed25519 (WEI25519 handle): 32-octet keys, SHA-512; Ed448 (WEI448 handle): 57-octet keys, SHAKE256 with 114 octets (pre-hash: 64)
	const bool e448 = cv_in->pbits == 448;
	const uint32_t kl = e448 ? 57u : 32u, sl = 2 * kl, hl = sl;
	const int hash_type = e448 ? 5 : 4;
	const uint32_t blank = msg_slots ? kl + 64u : kl;   // A, or A || PH(M)
	if (stride < 4 || (stride & 3u) || stride > 4096 || (uint64_t)a_offset + 4 + blank > stride) {
		return fail(f + ": bad argument (stride: a multiple of 4 in 4 .. 4096; the blank for A [and PH(M)] must lie inside the slot)");
	}
	if (msg_slots && (msg_stride < 4 || (msg_stride & 3u) || msg_stride > 4096)) {
		return fail(f + ": bad argument (msg_stride: a multiple of 4 in 4 .. 4096)");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	EcamdEdSignArgs T;
	uint8_t dummy = 0;
	if (eddsa_sign_setup(fn, ctx, cv_in, n, keys_prj, sigs, hash_slots, &T) ||
	    eddsa_args_ok(fn, ctx, cv_in, n, keys_prj, sigs, hash_slots, all_valid ? &dummy : result, hl)) {
		return -1;
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const size_t cl = (size_t)cv->clen;
	uint8_t seed[32] = {0};
	uint32_t done = 0;
	if (all_valid) {
		*all_valid = 0;
		if (e448 ? !eddsa448_msm_available(cv) : !eddsa_msm_available(cv)) {
			return 0;   // not decided here
		}
		if (msm_seed(ctx, seed) || ensure(&ctx->stage[24], &ctx->stage_bytes[24], (size_t)n * kl) || ensure(&ctx->stage[25], &ctx->stage_bytes[25], (size_t)n * sl) ||
		    ensure(&ctx->stage[26], &ctx->stage_bytes[26], (size_t)n * hl) || ensure(&ctx->stage[27], &ctx->stage_bytes[27], (size_t)n + 256)) {
			return -1;
		}
	}
	std::vector<HostArr> arrs = {{keys_prj, nullptr, 3 * cl}, {sigs, nullptr, sl}, {hash_slots, nullptr, stride}, {nullptr, all_valid ? nullptr : result, 1}};
	if (msg_slots) {
		arrs.push_back({msg_slots, nullptr, msg_stride});
	}
	// Ed25519, one piece, by buckets: the combination's per-item stages on every chunk as it lands (eddsa_bkt_stream_*; $ECAMD_NO_ED_STREAM: the
	// whole combination after the last chunk, as for Ed448 and for batches of several pieces)
	EdBktRun run;
	const bool streamed = all_valid && !e448 && n <= ctx->max_chunk && eddsa_msm_use_buckets(n) && getenv("ECAMD_NO_ED_STREAM") == nullptr;
	if (streamed) {
		StreamScope scope(ctx, ctx->stream);
		if (eddsa_bkt_stream_begin(ctx, cv, n, ctx->stage[24], ctx->stage[25], ctx->stage[26], seed, run, ctx->stream)) {
			return -1;
		}
	}
	const int prc = host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		// stage: 20 affine keys, 21 import status, 22 encodings, 23 their status (the verification core owns 3 .. 19)
		if (ensure(&ctx->stage[20], &ctx->stage_bytes[20], (size_t)m * 2 * cl) || ensure(&ctx->stage[21], &ctx->stage_bytes[21], m) ||
		    ensure(&ctx->stage[22], &ctx->stage_bytes[22], (size_t)m * kl) || ensure(&ctx->stage[23], &ctx->stage_bytes[23], m)) {
			return -1;
		}
		uint8_t *slots = const_cast<uint8_t *>(ip[2]);   // the staged copy of the caller's slots
		if (msg_slots) {
			// PH(M): SHA-512(M), or the first 64 octets of SHAKE256(M), of every message, written behind the blank for A
			// (eddsa_compute_pre_hash, sig/eddsa.c:1049-1080, :1650-1657)
			if (ecdsa_hash_stage(ctx, hash_type, m, ip[4], msg_stride, 64, s)) {
				return -1;
			}
			HIPCHK(ecamd_launch_slot_patch(slots, stride, a_offset + kl, ctx->stage[17], 64, nullptr, m, s));
		}
		EcamdPrjInArgs I;
		I.in = ip[0];
		I.aff = ctx->stage[20];
		I.pre = ctx->stage[21];
		I.n = m;
		I.clen = (uint32_t)cl;
		I.for_mul = 0;
		I.slot = cv->slot;
		HIPCHK(launch_prj_import(cv, I, s));
		EcamdEdSignArgs A = T;
		A.n = m;
		A.Rw = ctx->stage[20];
		A.stR = ctx->stage[21];
		// (the whole-batch form files the encodings and their "no encoding" marks in batch-wide arrays: Ed25519's 32-octet encodings are written
		// there directly; Ed448's 57-octet ones keep their aligned staging and a copy)
		const bool direct = all_valid && !e448;
		A.out = direct ? ctx->stage[24] + (size_t)done * kl : ctx->stage[22];
		A.status = direct ? ctx->stage[27] + (size_t)done : ctx->stage[23];
		HIPCHK(launch_ed_sign_enc(cv, A, s));
		HIPCHK(ecamd_launch_slot_patch(slots, stride, a_offset, A.out, kl, A.status, m, s));
		if (all_valid) {
			// the whole-batch form: file this chunk's encoded keys, signatures, hashes and "no encoding" marks; the equation comes at the end
			if (ecdsa_hash_stage(ctx, hash_type, m, slots, stride, hl, s)) {
				return -1;
			}
			if (!direct) {
				HIPCHK(hipMemcpyAsync(ctx->stage[24] + (size_t)done * kl, ctx->stage[22], (size_t)m * kl, hipMemcpyDeviceToDevice, s));
				HIPCHK(hipMemcpyAsync(ctx->stage[27] + (size_t)done, ctx->stage[23], m, hipMemcpyDeviceToDevice, s));
			}
			HIPCHK(hipMemcpyAsync(ctx->stage[25] + (size_t)done * sl, ip[1], (size_t)m * sl, hipMemcpyDeviceToDevice, s));
			HIPCHK(hipMemcpyAsync(ctx->stage[26] + (size_t)done * hl, ctx->stage[17], (size_t)m * hl, hipMemcpyDeviceToDevice, s));
			if (streamed && eddsa_bkt_stream_chunk(cv, run, done, m, s)) {
				return -1;
			}
			done += m;
			return 0;
		}
		if (ecdsa_hash_stage(ctx, hash_type, m, slots, stride, hl, s) ||
		    (e448 ? eddsa448_verify_dev_locked(ctx, cv, m, ctx->stage[22], ip[1], ctx->stage[17], op[3], s)
			  : eddsa_verify_dev_locked(ctx, cv, m, ctx->stage[22], ip[1], ctx->stage[17], 64, op[3], s))) {
			return -1;
		}
		HIPCHK(ecamd_launch_reject_where(op[3], ctx->stage[23], m, s));
		return 0;
	}, streamed ? (1u << 17) : 0u);
	if (prc || !all_valid) {
		return prc;
	}
	// the batch equation over the filed arrays, a piece of max_chunk items at a time; the verdict byte rests behind the marks
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	uint8_t *d_verdict = ctx->stage[27] + n;
	HIPCHK(hipMemsetAsync(d_verdict, 0, 1, s));
	uint32_t pc = 0;
	if (streamed && eddsa_bkt_stream_end(ctx, cv, run, d_verdict, s)) {
		(void)hipStreamSynchronize(s);
		return -1;
	}
	for (uint32_t off = 0; off < n && !streamed; off += ctx->max_chunk, pc++) {
		const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
		if (e448 ? eddsa448_msm_dev_locked(ctx, cv, m, ctx->stage[24] + (size_t)off * kl, ctx->stage[25] + (size_t)off * sl, ctx->stage[26] + (size_t)off * hl,
						   seed, pc, d_verdict, s)
			 : eddsa_msm_dev_locked(ctx, cv, m, ctx->stage[24] + (size_t)off * 32, ctx->stage[25] + (size_t)off * 64, ctx->stage[26] + (size_t)off * 64, seed, pc,
						d_verdict, nullptr, nullptr, s)) {
			(void)hipStreamSynchronize(s);
			return -1;
		}
	}
	std::vector<uint8_t> marks((size_t)n + 1, 1);
	HIPCHK(hipMemcpyAsync(marks.data(), ctx->stage[27], (size_t)n + 1, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	memset(seed, 0, sizeof(seed));
	int ok = marks[n] == 0;
	for (uint32_t i = 0; i < n && ok; i++) {
		ok = marks[i] == 0;   // a key without an encoding (not on the curve, at infinity): the reference rejects the item
	}
	*all_valid = ok;
	return 0;
}

// ec_verify_batch's one bit for plain Ed25519 FROM the projective keys, signatures and hash inputs of ec_eddsa_verify_msg_prj_batch (round 6):
// the same front end per staging chunk, then the batch equation over the whole batch as one multi-scalar multiplication per max_chunk
// items.  *all_valid = 0: not decided here -- a bad signature, a key that does not import or has no encoding, or a handle without the
// form (Ed448): the caller verifies item by item.
extern "C" int ec_eddsa_verify_msg_prj_all_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
						 const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, int *all_valid)
{
	if (!all_valid || n == 0) {
		return fail("ec_eddsa_verify_msg_prj_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	const int r = eddsa_verify_msg_prj_impl("ec_eddsa_verify_msg_prj_all_batch", ctx, cv, n, keys_prj, sigs, hash_slots, stride, a_offset, nullptr, 0, nullptr,
						all_valid);
	msm_seed_discard(ctx);
	return r;
}

extern "C" int ec_eddsa_verify_msg_prj_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					     const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result)
{
	return eddsa_verify_msg_prj_impl("ec_eddsa_verify_msg_prj_batch", ctx, cv, n, keys_prj, sigs, hash_slots, stride, a_offset, nullptr, 0, result);
}

// The pre-hashed variant (EDDSA25519PH): the hash input is dom2(1, ctx) || R || A || PH(M) with PH(M) = SHA-512(M); the caller leaves 96
// blank octets at a_offset (A, then PH(M)) and hands the messages over in slots of their own, hashed on the device as well.
extern "C" int ec_eddsa_verify_ph_prj_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
					    const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
					    uint8_t *result)
{
	if (n && !msg_slots) {
		return fail("ec_eddsa_verify_ph_prj_batch: bad argument");
	}
	static const uint8_t none = 0;
	return eddsa_verify_msg_prj_impl("ec_eddsa_verify_ph_prj_batch", ctx, cv, n, keys_prj, sigs, hash_slots, stride, a_offset, msg_slots ? msg_slots : &none,
					 msg_stride, result);
}

// ------------------------------------------------------------------------------------------
// Ed25519 whole-batch verification as one multi-scalar multiplication (the equation of _eddsa_verify_batch_no_memory,
// sig/eddsa.c:2278-2545, on the Edwards curve; kernels k_edmsm_* of ecamd_g29_kernel.hip / ecamd_kernels.hip).
// verdict byte: 0 = the combination vanishes and no item was rejected beforehand, 1 otherwise.  Only enqueues.
// ------------------------------------------------------------------------------------------
static size_t msm_align(size_t x) { return (x + 255) & ~(size_t)255; }

static bool eddsa_msm_available(const ecamd_curve *cv)
{
	return cv->pbits == 255 && cv->ed_state == 1 && cv->gflavour == 2 && cv->gslot >= 0;
}

static uint32_t eddsa_msm_pick_k(const ecamd_ctx *ctx, uint32_t n)
{
	if (ctx->msm_k) {
		return ctx->msm_k;
	}
	// one wave per SIMD needs 65536 lanes; two or more hide the table look-ups better
	uint32_t k = n >> 16;   // measured: 2^17 -> 2, 2^18 -> 4, 2^19 and up -> 8 (profiles/r2l_eddsa_msm.json)
	if (k < 1) {
		k = 1;
	}
	return k > 8 ? 8 : k;
}

// Round 6: the Ed25519 combination by buckets (k_edbkt_*; the Schnorr-type form: schnorr_msm_dev_locked) from 2^18 items on --
// $ECAMD_ED_MSM_ALGO=straus|bucket overrides the size rule.  The base point's term [q - sum z_i S_i]B enters as one copy of B per 64
// items with that group's share of the scalar: no global sum, no separate multiplication.
// scratch and kernel arguments of one combination over n items (ctx->msm; the flag word cleared)
static int eddsa_bkt_setup(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
			   const uint8_t seed[32], uint32_t piece, uint8_t *d_z_dump, EdBktRun &R, hipStream_t s)
{
	const uint32_t LB = (n + 63) / 64;
	const size_t counters = (size_t)16 << 16, recw = ECAMD_EDM_REC_WORDS;
	const uint32_t cap = 32u + 2u * (uint32_t)(((size_t)2 * n + LB + 65535) >> 16);
	// window 15: z_i h_i mod q < q = 2^252 + ..., so its digits take 4 097 values only and the n + LB scalars of that window crowd into as many buckets
	const uint32_t cap_top = 32u + 2u * (uint32_t)(((size_t)n + LB + 4096) / 4097);
	const size_t red_words = 2 * (2 * (size_t)16 * (65536 / ecamd_bkt_fold()) * recw + 18 * recw);   // (k_edbkt_reduce: two halves, T and U of the first level)
	size_t off = 0;
	auto carve = [&](size_t bytes) {
		const size_t o = off;
		off += msm_align(bytes);
		return o;
	};
	const size_t o_pts = carve(((size_t)2 * n + LB) * ECAMD_EDB_PT_STRIDE * 4);
	const size_t o_cA = carve((size_t)n * 32), o_zR = carve((size_t)n * 20), o_zs = carve((size_t)n * 32);
	const size_t o_rawC = carve((size_t)n * 32), o_rawZ = carve((size_t)n * 16), o_rawB = carve((size_t)LB * 32), o_sB = carve((size_t)LB * 32);
	const size_t o_flags = carve(n), o_flagsS = carve(n);
	const size_t o_cnt = carve(2 * counters * 4), o_ord = carve(((size_t)15 << 16) * cap * 4 + ((size_t)1 << 16) * cap_top * 4);
	const size_t o_bsum = carve(counters * recw * 4), o_red = carve(red_words * 4);
	const size_t o_word = carve(4);
	if (ensure(&ctx->msm, &ctx->msm_bytes, off)) {
		return -1;
	}
	uint8_t *M = ctx->msm;
	HIPCHK(hipMemsetAsync(M + o_word, 0, 4, s));
	R.n = n;
	R.flagword = (uint32_t *)(M + o_word);
	EcamdEdMsmArgs &A = R.A;
	memset(&A, 0, sizeof(A));
	A.encA = d_pub;
	A.strideA = 32;
	A.encR = d_sig;
	A.strideR = 64;
	A.flags = M + o_flags;
	A.n = n;
	A.cof_dbl = cv->ed_cof_dbl;
	memcpy(A.g_d, cv->ed_tmpl.g_d, sizeof(A.g_d));
	memcpy(A.g_sm1, cv->ed_tmpl.g_sm1, sizeof(A.g_sm1));
	memcpy(A.g_2d, cv->ed_2d, sizeof(A.g_2d));
	memcpy(A.g_Bx, cv->ed_Bx, sizeof(A.g_Bx));
	memcpy(A.g_By, cv->ed_By, sizeof(A.g_By));
	EcamdEdBktArgs &B = R.B;
	memset(&B, 0, sizeof(B));
	B.rawC = (const uint32_t *)(M + o_rawC);
	B.rawB = (const uint32_t *)(M + o_rawB);
	B.rawZ = (const uint32_t *)(M + o_rawZ);
	B.pts = (uint32_t *)(M + o_pts);
	B.count = (uint32_t *)(M + o_cnt);
	B.perm = B.count + counters;
	B.order = (uint32_t *)(M + o_ord);
	B.bsum = (uint32_t *)(M + o_bsum);
	B.red = (uint32_t *)(M + o_red);
	B.red_words = red_words;
	B.flagword = R.flagword;
	B.n = n;
	B.LB = LB;
	B.cap = cap;
	B.cap_top = cap_top;
	EcamdEdMsmScalArgs &C = R.C;
	memset(&C, 0, sizeof(C));
	C.sigs = d_sig;
	C.hram = d_hram;
	C.cA = (uint32_t *)(M + o_cA);
	C.zR = (uint32_t *)(M + o_zR);
	C.zs = (uint32_t *)(M + o_zs);
	C.rawC = (uint32_t *)(M + o_rawC);
	C.rawZ = (uint32_t *)(M + o_rawZ);
	C.flagsS = M + o_flagsS;
	C.z_dump = d_z_dump;
	memcpy(C.seed, seed, 32);
	C.nonce[0] = piece;
	C.n = n;
	C.qslot = cv->qslot;
	EcamdEdMsmLaneArgs &N = R.N;
	memset(&N, 0, sizeof(N));
	N.zs = (const uint32_t *)(M + o_zs);
	N.flags = M + o_flags;
	N.flagsS = M + o_flagsS;
	N.sB = (uint32_t *)(M + o_sB);
	N.rawB = (uint32_t *)(M + o_rawB);
	N.flagword = R.flagword;
	N.n = n;
	N.K = 64;
	N.L = LB;
	N.qslot = cv->qslot;
	return 0;
}
// the buckets summed, reduced and compared (everything is filed)
static int eddsa_bkt_tail(ecamd_ctx *ctx, ecamd_curve *cv, EdBktRun &R, uint8_t *d_verdict, uint32_t *d_sum_dump, hipStream_t s)
{
	if (ctx->timing) {
		HIPCHK(hipEventRecord(ctx->ev_dom[0], s));   // the dominant kernel: k_edbkt_accum, both launches (ecamd_ctx_dominant_kernel_ms)
	}
	if (ctx->side_ok && getenv("ECAMD_NO_EDBKT_SPLIT") == nullptr) {
		// the key-only windows (8 .. 15: z_i has 128 bits) first; their reduction and their doubling chains -- 16 w doublings in one lane, the long
		// ones -- on the side stream (idle by now) while the windows that hold the commitments too are summed (the Schnorr-type form does the same)
		HIPCHK(ecamd_launch_edbkt(R.A, R.B, 10, nullptr, nullptr, nullptr, cv->gslot, s));
		HIPCHK(hipEventRecord(ctx->side_hi, s));
		HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_hi, 0));
		HIPCHK(ecamd_launch_edbkt(R.A, R.B, 12, nullptr, nullptr, nullptr, cv->gslot, ctx->side_stream));
		HIPCHK(hipEventRecord(ctx->side_red, ctx->side_stream));
		HIPCHK(ecamd_launch_edbkt(R.A, R.B, 11, nullptr, nullptr, nullptr, cv->gslot, s));
		if (ctx->timing) {
			HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
			ctx->ev_dom_valid = true;
		}
		HIPCHK(ecamd_launch_edbkt(R.A, R.B, 13, nullptr, nullptr, nullptr, cv->gslot, s));
		HIPCHK(hipStreamWaitEvent(s, ctx->side_red, 0));
		HIPCHK(ecamd_launch_edbkt(R.A, R.B, 14, R.flagword, d_verdict, d_sum_dump, cv->gslot, s));
		return 0;
	}
	HIPCHK(ecamd_launch_edbkt(R.A, R.B, 1, nullptr, nullptr, nullptr, cv->gslot, s));
	if (ctx->timing) {
		HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
		ctx->ev_dom_valid = true;
	}
	HIPCHK(ecamd_launch_edbkt(R.A, R.B, 2, R.flagword, d_verdict, d_sum_dump, cv->gslot, s));
	return 0;
}

static int eddsa_bkt_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
				const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, uint8_t *d_z_dump, uint32_t *d_sum_dump, hipStream_t s)
{
	EdBktRun R;
	if (eddsa_bkt_setup(ctx, cv, n, d_pub, d_sig, d_hram, seed, piece, d_z_dump, R, s)) {
		return -1;
	}
	// the decoding (two square roots per item: VALU work on the caller's arrays alone) on the side stream, beside the scalars and the filing
	const bool beside = ctx->side_ok && getenv("ECAMD_NO_BKT_BESIDE") == nullptr;
	hipStream_t ps = s;
	if (beside) {
		HIPCHK(hipEventRecord(ctx->side_fork, s));
		HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
		ps = ctx->side_stream;
	}
	HIPCHK(ecamd_launch_edbkt(R.A, R.B, 0, nullptr, nullptr, nullptr, cv->gslot, ps));
	if (beside) {
		HIPCHK(hipEventRecord(ctx->side_done, ctx->side_stream));
	}
	HIPCHK(ecamd_launch_edmsm_scal(R.C, s));
	if (beside) {
		HIPCHK(hipStreamWaitEvent(s, ctx->side_done, 0));   // (the lane kernel reads the decoding's flags)
	}
	HIPCHK(ecamd_launch_edmsm_lane(R.N, s));
	HIPCHK(ecamd_launch_edbkt_file(R.B, s));
	return eddsa_bkt_tail(ctx, cv, R, d_verdict, d_sum_dump, s);
}

// The same combination FILED CHUNK BY CHUNK (round 6, ec_eddsa_verify_msg_prj_all_batch): a batch that arrives through the host pipeline is
// copy-bound while it arrives -- 285 MB per 2^20 items at PCIe rate, the device nearly idle -- and the combination used to start when the last
// chunk was in (7 ms).  Its per-item stages (the two decodings, the scalars, the filing: 4.4 of the 7 ms) now run on each chunk as it lands;
// what is left for the end is the base point's copies, the ranking, the additions and the reduction.  z_i is keyed by the item's index in the
// batch, so the verdict is that of eddsa_bkt_dev_locked with the same seed.
static int eddsa_bkt_stream_begin(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig, const uint8_t *d_hram,
				  const uint8_t seed[32], EdBktRun &R, hipStream_t s)
{
	if (eddsa_bkt_setup(ctx, cv, n, d_pub, d_sig, d_hram, seed, 0, nullptr, R, s)) {
		return -1;
	}
	HIPCHK(hipMemsetAsync(R.B.count, 0, ((size_t)16 << 16) * 4, s));
	return 0;
}
static int eddsa_bkt_stream_chunk(ecamd_curve *cv, EdBktRun &R, uint32_t first, uint32_t count, hipStream_t s)
{
	if (count == 0) {
		return 0;
	}
	R.A.first = R.C.first = R.B.first = first;
	R.A.count = R.C.count = R.B.count_items = count;
	R.B.part = 1;
	HIPCHK(ecamd_launch_edbkt(R.A, R.B, 0, nullptr, nullptr, nullptr, cv->gslot, s));
	HIPCHK(ecamd_launch_edmsm_scal(R.C, s));
	HIPCHK(ecamd_launch_edbkt_file(R.B, s));
	return 0;
}
static int eddsa_bkt_stream_end(ecamd_ctx *ctx, ecamd_curve *cv, EdBktRun &R, uint8_t *d_verdict, hipStream_t s)
{
	R.A.first = R.A.count = R.C.first = R.C.count = 0;
	HIPCHK(ecamd_launch_edmsm_lane(R.N, s));
	R.B.part = 2;
	HIPCHK(ecamd_launch_edbkt_file(R.B, s));
	return eddsa_bkt_tail(ctx, cv, R, d_verdict, nullptr, s);
}

// buckets or the Straus loop for a piece of n items ($ECAMD_ED_MSM_ALGO=straus|bucket overrides the size rule)
static bool eddsa_msm_use_buckets(uint32_t n)
{
	const char *e = getenv("ECAMD_ED_MSM_ALGO");
	const bool force_b = e && !strcmp(e, "bucket"), force_s = e && !strcmp(e, "straus");
	return force_b || (!force_s && n >= (1u << 18));   // measured: 2^17 items 2.18 ms by buckets, 2.06 by Straus; 2^18: 2.80 / 3.22; 2^20: 6.85 / 10.8
}

static int eddsa_msm_dev_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *d_pub, const uint8_t *d_sig,
				const uint8_t *d_hram, const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, uint8_t *d_z_dump,
				uint32_t *d_sum_dump, hipStream_t s)
{
	if (eddsa_msm_use_buckets(n)) {
		return eddsa_bkt_dev_locked(ctx, cv, n, d_pub, d_sig, d_hram, seed, piece, d_verdict, d_z_dump, d_sum_dump, s);
	}
	const uint32_t K = eddsa_msm_pick_k(ctx, n);
	const uint32_t L = (n + K - 1) / K;
	size_t off = 0;
	auto carve = [&](size_t bytes) {
		const size_t o = off;
		off += msm_align(bytes);
		return o;
	};
	const size_t o_tbl = carve((size_t)n * 2 * ECAMD_EDT_ITEM_WORDS * 4);
	const size_t o_cA = carve((size_t)n * 32), o_zR = carve((size_t)n * 20), o_zs = carve((size_t)n * 32);
	const size_t o_flags = carve(n), o_flagsS = carve(n);
	const size_t o_sB = carve((size_t)L * 32), o_rec = carve((size_t)L * ECAMD_EDM_REC_WORDS * 4);
	const size_t o_tmp = carve(((size_t)L / 16 + 1) * ECAMD_EDM_REC_WORDS * 4);
	const size_t o_tblB = carve((size_t)ECAMD_EDT_ITEM_WORDS * 4), o_word = carve(4);
	if (ensure(&ctx->msm, &ctx->msm_bytes, off)) {
		return -1;
	}
	uint8_t *M = ctx->msm;
	HIPCHK(hipMemsetAsync(M + o_word, 0, 4, s));
	EcamdEdMsmArgs A;
	memset(&A, 0, sizeof(A));
	A.encA = d_pub;
	A.strideA = 32;
	A.encR = d_sig;
	A.strideR = 64;
	A.tbl = (uint32_t *)(M + o_tbl);
	A.flags = M + o_flags;
	A.cA = (const uint32_t *)(M + o_cA);
	A.zR = (const uint32_t *)(M + o_zR);
	A.sB = (const uint32_t *)(M + o_sB);
	A.tblB = (const uint32_t *)(M + o_tblB);
	A.rec = (uint32_t *)(M + o_rec);
	A.n = n;
	A.K = K;
	A.L = L;
	A.cof_dbl = cv->ed_cof_dbl;
	memcpy(A.g_d, cv->ed_tmpl.g_d, sizeof(A.g_d));
	memcpy(A.g_sm1, cv->ed_tmpl.g_sm1, sizeof(A.g_sm1));
	memcpy(A.g_2d, cv->ed_2d, sizeof(A.g_2d));
	memcpy(A.g_Bx, cv->ed_Bx, sizeof(A.g_Bx));
	memcpy(A.g_By, cv->ed_By, sizeof(A.g_By));
	HIPCHK(ecamd_launch_edmsm_btable(A, (uint32_t *)(M + o_tblB), cv->gslot, s));
	HIPCHK(ecamd_launch_edmsm_prep(A, cv->gslot, s));
	EcamdEdMsmScalArgs C;
	memset(&C, 0, sizeof(C));
	C.sigs = d_sig;
	C.hram = d_hram;
	C.cA = (uint32_t *)(M + o_cA);
	C.zR = (uint32_t *)(M + o_zR);
	C.zs = (uint32_t *)(M + o_zs);
	C.flagsS = M + o_flagsS;
	C.z_dump = d_z_dump;
	memcpy(C.seed, seed, 32);
	C.nonce[0] = piece;
	C.n = n;
	C.qslot = cv->qslot;
	HIPCHK(ecamd_launch_edmsm_scal(C, s));
	EcamdEdMsmLaneArgs N;
	memset(&N, 0, sizeof(N));
	N.zs = (const uint32_t *)(M + o_zs);
	N.flags = M + o_flags;
	N.flagsS = M + o_flagsS;
	N.sB = (uint32_t *)(M + o_sB);
	N.flagword = (uint32_t *)(M + o_word);
	N.n = n;
	N.K = K;
	N.L = L;
	N.qslot = cv->qslot;
	HIPCHK(ecamd_launch_edmsm_lane(N, s));
	if (ctx->timing) {
		HIPCHK(hipEventRecord(ctx->ev_dom[0], s));   // the dominant kernel: k_edmsm_loop (ecamd_ctx_dominant_kernel_ms)
	}
	HIPCHK(ecamd_launch_edmsm_loop(A, cv->gslot, s));
	if (ctx->timing) {
		HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
		ctx->ev_dom_valid = true;
	}
	HIPCHK(ecamd_launch_edmsm_reduce(A, (uint32_t *)(M + o_tmp), (const uint32_t *)(M + o_word), d_verdict, d_sum_dump, cv->gslot, s));
	return 0;
}

// key of the z_i of the batch equation: the caller's 32 bytes when ecamd_ctx_set_msm_seed left some (used once), else getrandom
static int msm_seed(ecamd_ctx *ctx, uint8_t seed[32])
{
	if (ctx->msm_seed_valid) {
		memcpy(seed, ctx->msm_seed_bytes, 32);
		memset(ctx->msm_seed_bytes, 0, 32);
		ctx->msm_seed_valid = false;
		return 0;
	}
	size_t got = 0;
	while (got < 32) {
		const ssize_t r = getrandom(seed + got, 32 - got, 0);
		if (r <= 0) {
			return fail("ec_eddsa_verify_all_batch: getrandom failed (the z_i of the batch equation must be unpredictable)");
		}
		got += (size_t)r;
	}
	return 0;
}

// host arrays -> one verdict: pieces of max_chunk items, each with its own combination.  *accept = 1 when every piece accepts.
static int eddsa_msm_host_locked(ecamd_ctx *ctx, ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
				 const uint8_t *hram, const uint8_t *fixed_seed, int *accept, uint8_t *z_out, uint32_t *sum_out)
{
	uint8_t seed[32];
	if (fixed_seed) {
		memcpy(seed, fixed_seed, 32);
	} else if (msm_seed(ctx, seed)) {
		return -1;
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	const uint32_t pieces = (n + chunk - 1) / chunk;
	if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)chunk * 32) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], (size_t)chunk * 64) ||
	    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)chunk * 64) ||
	    ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)pieces + (z_out ? (size_t)chunk * 16 : 0) + 256 + ECAMD_EDM_REC_WORDS * 4)) {
		return -1;
	}
	uint8_t *d_verdicts = ctx->stage[3];
	HIPCHK(hipMemsetAsync(d_verdicts, 0, pieces, s));
	uint32_t *d_sum = (uint32_t *)(ctx->stage[3] + ((pieces + 255) & ~(size_t)255));
	uint8_t *d_z = z_out ? (uint8_t *)(d_sum + ECAMD_EDM_REC_WORDS) : nullptr;
	for (uint32_t off = 0, pc = 0; off < n; off += chunk, pc++) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		HIPCHK(hipMemcpyAsync(ctx->stage[0], pubkeys + (size_t)off * 32, (size_t)m * 32, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[1], sigs + (size_t)off * 64, (size_t)m * 64, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[2], hram + (size_t)off * 64, (size_t)m * 64, hipMemcpyHostToDevice, s));
		if (eddsa_msm_dev_locked(ctx, cv, m, ctx->stage[0], ctx->stage[1], ctx->stage[2], seed, pc, d_verdicts + pc, d_z,
					 sum_out ? d_sum : nullptr, s)) {
			(void)hipStreamSynchronize(s);
			return -1;
		}
		if (z_out) {
			HIPCHK(hipMemcpyAsync(z_out + (size_t)off * 16, d_z, (size_t)m * 16, hipMemcpyDeviceToHost, s));
		}
	}
	std::vector<uint8_t> v(pieces, 1);
	HIPCHK(hipMemcpyAsync(v.data(), d_verdicts, pieces, hipMemcpyDeviceToHost, s));
	if (sum_out) {
		HIPCHK(hipMemcpyAsync(sum_out, d_sum, ECAMD_EDM_REC_WORDS * 4, hipMemcpyDeviceToHost, s));
	}
	HIPCHK(hipStreamSynchronize(s));
	*accept = 1;
	for (uint32_t pc = 0; pc < pieces; pc++) {
		if (v[pc] != 0) {
			*accept = 0;
		}
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// Schnorr-type whole-batch verification on a short-Weierstrass curve as one multi-scalar multiplication (EcamdMsmArgs in
// ecamd_internal.h; BIP0340's and ECFSDSA's batch equation, sig/bip0340.c:808-1025, sig/ecfsdsa.c:657-837): per piece of at most
// max_chunk items   k_msm_scal (z_i, z_i (q - e_i), z_i s_i mod q)  ->  k_msm_vsum (c = sum z_i s_i)  ->  [c]G by the handle's
// fixed-base path  ->  k_msm_table_g (2n window tables)  ->  k_msm_loop_g (Straus, K items per lane)  ->  k_msm_sum_g / final.
// ------------------------------------------------------------------------------------------
static bool schnorr_msm_unit(const ecamd_curve *cv, int *pbits, int *flavour, int *slot)
{
	if (cv->gslot >= 0) {
		*pbits = cv->pbits;
		*flavour = cv->gflavour;
		*slot = cv->gslot;
		return true;
	}
	if (cv->is_p256 && cv->gpslot >= 0) {   // secp256r1: the dense 256-bit unit that serves its projective import
		*pbits = 256;
		*flavour = 0;
		*slot = cv->gpslot;
		return true;
	}
	return false;
}

static uint32_t schnorr_msm_pick_k(uint32_t n)
{
	if (const char *e = getenv("ECAMD_SCHNORR_MSM_K")) {
		const uint32_t k = (uint32_t)strtoul(e, nullptr, 10);
		if (k >= 1 && k <= 64) {
			return k;
		}
	}
	// measured on secp256k1 (profiles/r5h_schnorr_msm.md): the Straus loop wants about 2^18 lanes (four waves per SIMD: its additions are
	// dependent chains) before sharing doublings pays -- 2^20 items: K = 4 (19.3 ms) beats 8 (23.1) and 2 (20.6); 2^18 items: K = 1
	uint32_t k = n >> 18;
	if (k < 1) {
		k = 1;
	}
	return k > 8 ? 8 : k;
}

// Straus (round 5) or buckets (round 6) for a piece of n items.  Buckets: windows of 16 bits -- z_i has 128 = 8 x 16 of them, and the
// orders of the usual curves a multiple of 16 -- from 2^17 items on (below, the fixed costs of the sort and of the reduction over 2^16
// buckets per window outweigh the additions they save: profiles/r6h_schnorr_bucket.md), and only when the order's TOP window still has
// at least 12 bits: a top window of t bits files n points into 2^t buckets, each added up by one lane (secp521r1: 9 bits, secp224k1:
// 1 bit -- those keep the Straus loop).  $ECAMD_SCHNORR_MSM_ALGO=straus|bucket overrides the size rule (tests, measurements).
static bool schnorr_msm_use_buckets(const ecamd_curve *cv, uint32_t n)
{
	const uint32_t top_bits = (uint32_t)cv->qbits - 16u * (((uint32_t)cv->qbits - 1u) / 16u);
	if (top_bits < 12u) {
		return false;
	}
	if (const char *e = getenv("ECAMD_SCHNORR_MSM_ALGO")) {
		if (!strcmp(e, "straus")) {
			return false;
		}
		if (!strcmp(e, "bucket")) {
			return true;
		}
	}
	return n >= (1u << 17);
}
static uint32_t schnorr_bkt_window(uint32_t) { return 16u; }
// the streamed form (SchnorrStreamStep) files into fixed-capacity buckets and ends on the side stream's joins
static bool schnorr_msm_streams(const ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n)
{
	return n <= ctx->max_chunk && schnorr_msm_use_buckets(cv, n) && ctx->side_ok && getenv("ECAMD_NO_BKT_BESIDE") == nullptr &&
	       getenv("ECAMD_BKT_EXACT_SORT") == nullptr && getenv("ECAMD_NO_SCHNORR_STREAM") == nullptr;
}

static int schnorr_msm_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *d_s, const uint8_t *d_ne, const uint8_t *d_keys,
				  const uint8_t *d_r, int r_fmt, const uint8_t seed[32], uint32_t piece, uint8_t *d_verdict, uint8_t *d_z_dump,
				  uint32_t *d_sum_dump, hipStream_t s, uint32_t cof_dbl, const SchnorrStreamStep *step)
{
	const int mode = step ? step->mode : 0;
	// cof_dbl > 0 (eddsa448_msm_dev_locked only): the final test is [2^cof_dbl](sum + [c]G) = infinity, EdDSA's cofactored equation
	int pbits = 0, flav = 0, gslot = -1;
	if (!schnorr_msm_unit(cv, &pbits, &flav, &gslot) || cv->qslot < 0) {
		return fail("internal: Schnorr multi-scalar multiplication without a radix-2^29 unit");
	}
	const bool buckets = schnorr_msm_use_buckets(cv, n);
	const uint32_t K = schnorr_msm_pick_k(n);
	const uint32_t L = (n + K - 1) / K;
	const size_t itemw = ecamd_g29_table_words(pbits, flav), recw = ecamd_g29_msm_rec_words(pbits, flav);
	const size_t cl = (size_t)cv->clen, ql = (size_t)cv->qlen, qnw = (size_t)cv->qnw;
	// bucket evaluation: window bits, windows of the full-length scalars / of the 128-bit z_i, counters, reduction scratch
	const uint32_t bc = schnorr_bkt_window(n), bnwin = (uint32_t)((8 * ql + bc - 1) / bc), bnwinZ = (128u + bc - 1) / bc;
	const size_t bcounters = (size_t)bnwin << bc, bpw = ecamd_g29_bkt_point_words(pbits, flav);
	const size_t bfold = ecamd_bkt_fold();
	const size_t bred_words = 2 * (2 * (size_t)bnwin * ((((size_t)1 << bc) + bfold - 1) / bfold) * recw + ((size_t)bnwin + 2) * recw);
	size_t off = 0;
	auto carve = [&](size_t bytes) {
		const size_t o = off;
		off += msm_align(bytes);
		return o;
	};
	const size_t o_tbl = carve(buckets ? (size_t)2 * n * bpw * 4 : (size_t)2 * n * itemw * 4);
	const size_t o_rec = carve(buckets ? bcounters * recw * 4 : (size_t)L * recw * 4);
	const size_t o_tmp = carve(buckets ? bred_words * 4 : ((size_t)L / 16 + 2) * recw * 4);
	// fixed-capacity filing unless $ECAMD_BKT_EXACT_SORT: 32 + 2 lambda slots per bucket, lambda = 2n / 2^16 the mean of the fullest windows
	const uint32_t bcap = getenv("ECAMD_BKT_EXACT_SORT") ? 0u : 32u + 2u * (uint32_t)(((size_t)2 * n + 65535) >> 16);
	// ... and the order's TOP window: z_i (q - e_i) < q, so its digits take (q >> 16 top_win) + 1 values only and the n keys crowd into that many
	// buckets (brainpoolP256r1: 43 516 of 65 536; an order just above a power of 2^16: a few) -- a capacity of its own
	const uint32_t btop = ((uint32_t)cv->qbits - 1u) / 16u;
	uint32_t btopvals = 1;
	{
		const uint32_t bit = 16u * btop;   // the digit of q itself in that window, + 1
		const size_t wi = bit / 32u;
		const uint32_t word = wi < cv->q.size() ? cv->q[wi] : 0u;
		btopvals = ((word >> (bit % 32u)) & 0xffffu) + 1u;
	}
	const uint32_t bcap_top = bcap ? 32u + 2u * (uint32_t)(((size_t)n + btopvals - 1) / btopvals) : 0u;
	const size_t o_cnt = carve(buckets ? 4 * bcounters * 4 : 0);
	const size_t o_ord = carve(buckets ? (bcap ? ((size_t)btop << 16) * bcap * 4 + ((size_t)(bnwin - btop) << 16) * bcap_top * 4 : (size_t)bnwin * 2 * n * 4) : 0);
	const size_t o_w = carve((size_t)n * ql), o_z = carve((size_t)n * 16), o_v = carve((size_t)n * qnw * 4);
	const size_t o_v1 = carve(((size_t)n / 64 + 2) * qnw * 4), o_v2 = carve(((size_t)n / 4096 + 2) * qnw * 4);
	const size_t o_c = carve(ql), o_gen = carve(2 * cl), o_gst = carve(4), o_word = carve(4);
	if (ensure(&ctx->msm, &ctx->msm_bytes, off)) {
		return -1;
	}
	uint8_t *M = ctx->msm;
	if (mode <= 1) {
		HIPCHK(hipMemsetAsync(M + o_word, 0, 4, s));
	}
	if (mode == 1) {
		HIPCHK(hipMemsetAsync(M + o_cnt, 0, bcounters * 4, s));   // (the filing's counters)
		return 0;
	}
	if (mode == 2) {
		// one staging chunk of the streamed form, everything on the caller's stream: the keys' and the commitments' import, the scalars, the filing
		EcamdMsmArgs P;
		memset(&P, 0, sizeof(P));
		P.ptsY = d_keys;
		P.ptsR = d_r;
		P.flagword = (uint32_t *)(M + o_word);
		P.pts = (uint32_t *)(M + o_tbl);
		P.n = n;
		P.clen = (uint32_t)cl;
		P.r_fmt = (uint32_t)r_fmt;
		P.cof_dbl = cof_dbl;
		P.pt_first = step->first;
		P.pt_count = step->count;
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 10, P, nullptr, nullptr, nullptr, nullptr, nullptr, s));
		P.pt_first = n + step->first;
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 10, P, nullptr, nullptr, nullptr, nullptr, nullptr, s));
	}
	bool points_beside = mode == 3;   // (the streamed form asked schnorr_msm_streams first: buckets, a side stream)
	if (mode == 0 && buckets && ctx->side_ok && getenv("ECAMD_NO_BKT_BESIDE") == nullptr) {
		// the import of the 2n points (on-curve checks, BIP0340's square roots: VALU work that reads only the caller's arrays) runs on the side
		// stream beside the scalar kernels and the counting sort (atomics and scattered stores); the bucket additions wait for both
		EcamdMsmArgs P;
		memset(&P, 0, sizeof(P));
		P.ptsY = d_keys;
		P.ptsR = d_r;
		P.flagword = (uint32_t *)(M + o_word);
		P.pts = (uint32_t *)(M + o_tbl);
		P.n = n;
		P.clen = (uint32_t)cl;
		P.r_fmt = (uint32_t)r_fmt;
		P.cof_dbl = cof_dbl;
		HIPCHK(hipEventRecord(ctx->side_fork, s));
		HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
		// the keys first (an on-curve check each): the windows above z_i's 128 bits hold keys only and can be added up while the
		// commitments are still being lifted (a square root each)
		P.pt_first = 0;
		P.pt_count = n;
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 10, P, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->side_stream));
		HIPCHK(hipEventRecord(ctx->side_mid, ctx->side_stream));
		P.pt_first = n;
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 10, P, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->side_stream));
		HIPCHK(hipEventRecord(ctx->side_done, ctx->side_stream));
		points_beside = true;
	}
	EcamdMsmScalArgs C;
	memset(&C, 0, sizeof(C));
	C.s = d_s;
	C.ne = d_ne;
	C.scW = M + o_w;
	C.scZ = M + o_z;
	C.v = (uint32_t *)(M + o_v);
	C.flagword = (uint32_t *)(M + o_word);
	C.z_dump = d_z_dump;
	memcpy(C.seed, seed, 32);
	C.nonce[0] = piece;
	C.nonce[1] = 0x5343484eu;   // "SCHN": another stream than the Ed25519 combination's under the same seed
	C.n = n;
	C.qlen = (uint32_t)ql;
	C.qslot = cv->qslot;
	if (mode == 2) {
		C.first = step->first;
		C.count = step->count;
	}
	if (mode != 3) {
		HIPCHK(ecamd_launch_msm_scal(cv->qnw, C, s));
	}
	if (mode == 2) {
		uint32_t *cnt = (uint32_t *)(M + o_cnt);
		EcamdBktSortArgs B;
		memset(&B, 0, sizeof(B));
		B.scW = M + o_w;
		B.scZ = M + o_z;
		B.hist = cnt;
		B.cap = bcap;
		B.cap_top = bcap_top;
		B.top_win = btop;
		B.flag = (uint32_t *)(M + o_word);
		B.order = (uint32_t *)(M + o_ord);
		B.n = n;
		B.wlen = (uint32_t)ql;
		B.zlen = 16;
		B.c = bc;
		B.nwin = bnwin;
		B.nwinZ = bnwinZ < bnwin ? bnwinZ : bnwin;
		B.part = 1;
		B.item_first = step->first;
		B.item_count = step->count;
		HIPCHK(ecamd_launch_bkt_sort(B, s));
		return 0;
	}
	// c = sum z_i s_i and [c]G: four tiny reduction launches and a one-item fixed-base multiplication, a latency-bound chain of 0.7 ms that
	// only the final comparison needs -- on the side stream behind the points when there is one, beside the counting sort
	hipStream_t vs = s;
	if (points_beside) {
		HIPCHK(hipEventRecord(ctx->side_fork, s));
		HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
		vs = ctx->side_stream;
	}
	{
		// c = sum v_i mod q: levels of fan-in 64, the last one writes the big-endian bytes
		EcamdMsmVsumArgs V;
		memset(&V, 0, sizeof(V));
		V.qlen = (uint32_t)ql;
		V.qslot = cv->qslot;
		const uint32_t *src = (const uint32_t *)(M + o_v);
		uint32_t *bufs[2] = {(uint32_t *)(M + o_v1), (uint32_t *)(M + o_v2)};
		uint32_t count = n;
		int b = 0;
		do {
			const uint32_t outc = (count + 63) / 64;
			V.in = src;
			V.out = bufs[b];
			V.count = count;
			V.c_be = outc == 1 ? M + o_c : nullptr;
			HIPCHK(ecamd_launch_msm_vsum(cv->qnw, V, vs));
			src = bufs[b];
			b ^= 1;
			count = outc;
		} while (count > 1);
	}
	// [c]G from the handle's own fixed-base path, one item, device pointers (the comb table is built for a batch of this size: without
	// it the single item would walk the whole window loop alone, a millisecond of latency)
	maybe_build_comb(ctx, const_cast<ecamd_curve *>(cv), n);
	if (smul_dev_locked(ctx, cv, 1, M + o_c, (uint32_t)ql, nullptr, M + o_gen, M + o_gst, vs, 0xffffffffu, false, nullptr)) {
		return -1;
	}
	if (points_beside) {
		HIPCHK(hipEventRecord(ctx->side_aux, ctx->side_stream));
	}
	EcamdMsmArgs A;
	memset(&A, 0, sizeof(A));
	A.ptsY = d_keys;
	A.ptsR = d_r;
	A.scW = M + o_w;
	A.scZ = M + o_z;
	A.tbl = (uint32_t *)(M + o_tbl);
	A.rec = (uint32_t *)(M + o_rec);
	A.flagword = (uint32_t *)(M + o_word);
	A.n = n;
	A.K = K;
	A.L = L;
	A.clen = (uint32_t)cl;
	A.wlen = (uint32_t)ql;
	A.zlen = 16;
	A.r_fmt = (uint32_t)r_fmt;
	A.cof_dbl = cof_dbl;
	if (buckets) {
		A.pts = (uint32_t *)(M + o_tbl);
		A.bsum = (uint32_t *)(M + o_rec);
		A.red = (uint32_t *)(M + o_tmp);
		A.red_words = bred_words;
		A.c = bc;
		A.nwin = bnwin;
		uint32_t *cnt = (uint32_t *)(M + o_cnt);
		A.bcount = cnt;
		A.bstart = cnt + bcounters;
		A.order = (const uint32_t *)(M + o_ord);
		EcamdBktSortArgs B;
		memset(&B, 0, sizeof(B));
		B.scW = M + o_w;
		B.scZ = M + o_z;
		B.hist = cnt;
		B.start = cnt + bcounters;
		B.cursor = cnt + 2 * bcounters;
		B.perm = getenv("ECAMD_NO_BKT_RANK") ? nullptr : cnt + 3 * bcounters;
		A.perm = B.perm;
		B.cap = A.cap = bcap;
		B.cap_top = A.cap_top = bcap_top;
		B.top_win = A.top_win = btop;
		B.flag = (uint32_t *)(M + o_word);
		B.order = (uint32_t *)(M + o_ord);
		B.n = n;
		B.wlen = (uint32_t)ql;
		B.zlen = 16;
		B.c = bc;
		B.nwin = bnwin;
		B.nwinZ = bnwinZ < bnwin ? bnwinZ : bnwin;
		// $ECAMD_BKT_FILE_SPLIT (off by default): the key-only windows filed first, on the caller's stream, and summed while the windows below
		// 2^128 are filed on a stream of their own.  Measured (profiles/r6_f4_kernels.md): secp256k1 2^20 items 6.91 -> 7.35 ms, Ed448 25.8 -> 25.7 --
		// the filing's atomics and scattered stores slow the gathers of the accumulation they run beside by more than the wait they save
		const bool file_split = mode == 0 && points_beside && B.nwinZ < bnwin && bcap != 0u && ctx->side2_ok && getenv("ECAMD_BKT_FILE_SPLIT") != nullptr;
		if (file_split) {
			HIPCHK(hipMemsetAsync(B.hist, 0, bcounters * 4, s));
			B.win_first = B.nwinZ;
			B.win_count = bnwin - B.nwinZ;
			HIPCHK(ecamd_launch_bkt_sort(B, s));
			HIPCHK(hipEventRecord(ctx->side2_fork, s));
			HIPCHK(hipStreamWaitEvent(ctx->side2_stream, ctx->side2_fork, 0));
			B.win_first = 0;
			B.win_count = B.nwinZ;
			HIPCHK(ecamd_launch_bkt_sort(B, ctx->side2_stream));
			HIPCHK(hipEventRecord(ctx->side2_done, ctx->side2_stream));
		} else {
			B.part = mode == 3 ? 2u : 0u;   // (the streamed form has filed everything: the ranking alone)
			HIPCHK(ecamd_launch_bkt_sort(B, s));
		}
		uint32_t *d_total = (uint32_t *)(M + o_tmp) + bred_words - recw;
		if (!points_beside) {
			HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 10, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
		}
		if (ctx->timing) {   // the dominant kernel: k_bkt_accum_g, both launches (ecamd_ctx_dominant_kernel_ms)
			HIPCHK(hipEventRecord(ctx->ev_dom[0], s));
		}
		bool reduced = false;
		if (points_beside && B.nwinZ < bnwin) {
			// the windows that hold keys only, as soon as the keys are in; then the rest once the commitments are
			if (mode == 0) {
				HIPCHK(hipStreamWaitEvent(s, ctx->side_mid, 0));
			}
			A.win_first = B.nwinZ;
			A.win_count = bnwin - B.nwinZ;
			HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 11, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
			const bool red_beside = getenv("ECAMD_NO_BKT_RED_BESIDE") == nullptr;
			if (red_beside) {
				// ... and their reduction on the side stream (idle by now: the commitments and [c]G are short) while the other windows are
				// still being summed: the doubling chain of window w is 16 w long -- one lane per window, latency-bound -- and these are the
				// long ones; the windows below 2^128 then end in chains of at most 112 doublings
				HIPCHK(hipEventRecord(ctx->side_hi, s));
				HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_hi, 0));
				HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 12, A, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->side_stream));
				HIPCHK(hipEventRecord(ctx->side_red, ctx->side_stream));
			}
			if (mode == 0) {
				HIPCHK(hipStreamWaitEvent(s, ctx->side_done, 0));
			}
			if (file_split) {
				HIPCHK(hipStreamWaitEvent(s, ctx->side2_done, 0));
			}
			A.win_first = 0;
			A.win_count = B.nwinZ;
			HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 11, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
			if (red_beside) {
				if (ctx->timing) {
					HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
					ctx->ev_dom_valid = true;
				}
				HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 12, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
				HIPCHK(hipStreamWaitEvent(s, ctx->side_red, 0));
				A.win_count = 0;
				HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 14, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
				reduced = true;
			}
			A.win_count = 0;
		} else {
			if (points_beside && mode == 0) {
				HIPCHK(hipStreamWaitEvent(s, ctx->side_done, 0));
			}
			HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 11, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
		}
		if (!reduced) {
			if (ctx->timing) {
				HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
				ctx->ev_dom_valid = true;
			}
			HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 12, A, nullptr, nullptr, nullptr, nullptr, nullptr, s));
		}
		if (points_beside) {
			HIPCHK(hipStreamWaitEvent(s, ctx->side_aux, 0));   // [c]G
		}
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, 13, A, d_total, M + o_gen, M + o_gst, d_verdict, d_sum_dump, s));
		return 0;
	}
	for (int phase = 0; phase < 3; phase++) {
		const bool timed = ctx->timing && phase == 1;   // the dominant kernel: k_msm_loop_g (ecamd_ctx_dominant_kernel_ms)
		if (timed) {
			HIPCHK(hipEventRecord(ctx->ev_dom[0], s));
		}
		HIPCHK(ecamd_launch_msm_g29(pbits, gslot, flav, phase, A, (uint32_t *)(M + o_tmp), M + o_gen, M + o_gst, d_verdict, d_sum_dump, s));
		if (timed) {
			HIPCHK(hipEventRecord(ctx->ev_dom[1], s));
			ctx->ev_dom_valid = true;
		}
	}
	return 0;
}

static int schnorr_args_ok(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *a, const void *b, const void *c,
			   const void *d, int r_fmt, const void *out)
{
	if (!ctx || !cv || cv->ctx != ctx || !out || n == 0 || !a || !b || !c || !d || (r_fmt != 0 && r_fmt != 1)) {
		return fail(std::string(fn) + ": bad argument (the reference rejects num = 0 too)");
	}
	return 0;
}

// 1: the multi-scalar form can serve this handle (a radix-2^29 unit; q usable as a modulus of at least 160 bits; for r_fmt 1, p = 3 mod 4)
static bool schnorr_msm_available(const ecamd_curve *cv, int r_fmt)
{
	int pb, fl, sl;
	if (!schnorr_msm_unit(cv, &pb, &fl, &sl) || cv->qslot < 0 || cv->qbits < 160 || (uint32_t)cv->qlen < 16u) {
		return false;
	}
	// Prime-order groups only.  On a curve with a cofactor a commitment R + D, D of small order, is rejected by the item
	// form but passes the random combination whenever z_i D = O -- with probability 1 / ord(D), up to 1/2 -- so the batch form would accept
	// what a loop of ec_verify rejects (the reference's batch has that weakness; this one must not).  Such curves take the item form.
	if (cv->cofactor != 1) {
		return false;
	}
	if (r_fmt == 1 && (cv->p[0] & 3u) != 3u) {
		return false;
	}
	return true;
}

static int schnorr_msm_host_locked(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *sv, const uint8_t *ne, const uint8_t *keys,
				   const uint8_t *r, int r_fmt, const uint8_t *fixed_seed, int *accept, uint8_t *z_out, uint32_t *sum_out)
{
	uint8_t seed[32];
	if (fixed_seed) {
		memcpy(seed, fixed_seed, 32);
	} else if (msm_seed(ctx, seed)) {
		return -1;
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	PublicScalars pub_scope(ctx);   // everything a verification multiplies by is public
	const size_t cl = (size_t)cv->clen, ql = (size_t)cv->qlen, rl = r_fmt ? cl : 2 * cl;
	const uint32_t chunk = n < ctx->max_chunk ? n : ctx->max_chunk;
	const uint32_t pieces = (n + chunk - 1) / chunk;
	int pb, fl, sl;
	(void)schnorr_msm_unit(cv, &pb, &fl, &sl);
	const size_t recw = ecamd_g29_msm_rec_words(pb, fl);
	if (ensure(&ctx->stage[0], &ctx->stage_bytes[0], (size_t)chunk * ql) || ensure(&ctx->stage[1], &ctx->stage_bytes[1], (size_t)chunk * ql) ||
	    ensure(&ctx->stage[2], &ctx->stage_bytes[2], (size_t)chunk * 2 * cl) || ensure(&ctx->stage[4], &ctx->stage_bytes[4], (size_t)chunk * rl) ||
	    ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)pieces + (z_out ? (size_t)chunk * 16 : 0) + 512 + recw * 4)) {
		return -1;
	}
	uint8_t *d_verdicts = ctx->stage[3];
	HIPCHK(hipMemsetAsync(d_verdicts, 1, pieces, s));
	uint32_t *d_sum = (uint32_t *)(ctx->stage[3] + ((pieces + 255) & ~(size_t)255));
	uint8_t *d_z = z_out ? (uint8_t *)(d_sum + recw) : nullptr;
	for (uint32_t off = 0, pc = 0; off < n; off += chunk, pc++) {
		const uint32_t m = (n - off) < chunk ? (n - off) : chunk;
		HIPCHK(hipMemcpyAsync(ctx->stage[0], sv + (size_t)off * ql, (size_t)m * ql, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[1], ne + (size_t)off * ql, (size_t)m * ql, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[2], keys + (size_t)off * 2 * cl, (size_t)m * 2 * cl, hipMemcpyHostToDevice, s));
		HIPCHK(hipMemcpyAsync(ctx->stage[4], r + (size_t)off * rl, (size_t)m * rl, hipMemcpyHostToDevice, s));
		if (schnorr_msm_dev_locked(ctx, cv, m, ctx->stage[0], ctx->stage[1], ctx->stage[2], ctx->stage[4], r_fmt, seed, pc, d_verdicts + pc, d_z,
					   sum_out ? d_sum : nullptr, s)) {
			(void)hipStreamSynchronize(s);
			return -1;
		}
		if (z_out) {
			HIPCHK(hipMemcpyAsync(z_out + (size_t)off * 16, d_z, (size_t)m * 16, hipMemcpyDeviceToHost, s));
		}
	}
	std::vector<uint8_t> v(pieces, 1);
	HIPCHK(hipMemcpyAsync(v.data(), d_verdicts, pieces, hipMemcpyDeviceToHost, s));
	if (sum_out) {
		HIPCHK(hipMemcpyAsync(sum_out, d_sum, recw * 4, hipMemcpyDeviceToHost, s));
	}
	HIPCHK(hipStreamSynchronize(s));
	memset(seed, 0, sizeof(seed));
	*accept = 1;
	for (uint32_t pc = 0; pc < pieces; pc++) {
		if (v[pc] != 0) {
			*accept = 0;
		}
	}
	return 0;
}

static void msm_seed_discard(ecamd_ctx *ctx);
extern "C" int ec_schnorr_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *s, const uint8_t *ne,
					   const uint8_t *keys_aff, const uint8_t *r, int r_fmt, int *all_valid)
{
	if (schnorr_args_ok("ec_schnorr_verify_all_batch", ctx, cv, n, s, ne, keys_aff, r, r_fmt, all_valid)) {
		return -1;
	}
	*all_valid = 0;
	int ret = 0;
	{
		std::lock_guard<std::mutex> lk(ctx->mu);
		HIPCHK(hipSetDevice(ctx->device));
		if (schnorr_msm_available(cv, r_fmt)) {
			ret = schnorr_msm_host_locked(ctx, cv, n, s, ne, keys_aff, r, r_fmt, nullptr, all_valid, nullptr, nullptr);
		}   // else: not decided here, the caller verifies item by item
	}
	// the seed of ecamd_ctx_set_msm_seed keys THIS call only: whichever way it ended ("not available", an error before the seed was
	// read), nothing stays pending for an unrelated whole-batch call
	msm_seed_discard(ctx);
	return ret;
}

// The combination alone, device pointers, one piece (n <= the context's max_chunk): d_verdict[0] = 0 when the batch equation holds and no
// exceptional event was met, 1 otherwise ("not decided here").  Only enqueues on the stream.
extern "C" int ec_schnorr_verify_all_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const void *d_s, const void *d_ne,
						   const void *d_keys_aff, const void *d_r, int r_fmt, void *d_verdict, void *hip_stream)
{
	if (schnorr_args_ok("ec_schnorr_verify_all_batch_dev", ctx, cv, n, d_s, d_ne, d_keys_aff, d_r, r_fmt, d_verdict)) {
		return -1;
	}
	int ret = -1;
	{
		std::lock_guard<std::mutex> lk(ctx->mu);
		HIPCHK(hipSetDevice(ctx->device));
		if (n > ctx->max_chunk || !schnorr_msm_available(cv, r_fmt)) {
			ret = fail("ec_schnorr_verify_all_batch_dev: needs a prime-order curve with a radix-2^29 unit (ec_schnorr_verify_all_available) and n <= max_chunk");
		} else {
			uint8_t seed[32];
			if (!msm_seed(ctx, seed)) {
				hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
				StreamScope scope(ctx, s);
				PublicScalars pub_scope(ctx);
				if (hipMemsetAsync(d_verdict, 1, 1, s) == hipSuccess) {
					ret = schnorr_msm_dev_locked(ctx, cv, n, (const uint8_t *)d_s, (const uint8_t *)d_ne, (const uint8_t *)d_keys_aff,
								     (const uint8_t *)d_r, r_fmt, seed, 0, (uint8_t *)d_verdict, nullptr, nullptr, s);
				}
				memset(seed, 0, sizeof(seed));
			}
		}
	}
	msm_seed_discard(ctx);
	return ret;
}

// BIP0340 / ECFSDSA whole-batch verification FROM keys, signatures and hash inputs (round 6): what libsign_amd.so's ec_verify_batch needs
// so that nothing but marshalling stays on the host.  Per chunk of the double-buffered staging: the keys are imported and normalised on
// the device (k_prj_import_g; affine keys go as they are), k_schnorr_prep writes the key's x into the blank of the item's hash input
// (BIP0340: e = H(H(tag) || H(tag) || r || Y.x || m), sig/bip0340.c:437-494 -- the caller supplies everything but Y.x) and files the key as
// the equation uses it (the even-y representative), s and the commitment in batch-wide arrays; the hash inputs are hashed
// (k_sha2_slots) and k_schnorr_ne leaves q - e.  When the last chunk is in, the batch is ONE multi-scalar multiplication per max_chunk
// items (schnorr_msm_dev_locked): the copies of the 200 MB a 2^20-item batch brings hide behind the front-end kernels, and the
// Straus loop runs at its full-batch rate.  *all_valid = 0: not decided here (see ec_schnorr_verify_all_batch).
extern "C" int ec_schnorr_verify_msg_all_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *keys, int key_fmt,
						   const uint8_t *sigs, int r_fmt, int hash_type, const uint8_t *hash_slots, uint32_t stride,
						   uint32_t x_offset, int *all_valid)
{
	if (schnorr_args_ok("ec_schnorr_verify_msg_all_batch", ctx, cv, n, keys, sigs, hash_slots, hash_slots, r_fmt, all_valid)) {
		return -1;
	}
	*all_valid = 0;
	const int hl = ecamd_sha2_digest_len(hash_type);
	const size_t cl = (size_t)cv->clen, ql = (size_t)cv->qlen, rl = r_fmt ? cl : 2 * cl, sl = rl + ql;
	if ((key_fmt != ECAMD_PT_AFFINE && key_fmt != ECAMD_PT_PROJECTIVE) || hl == 0 || stride < 4 || (stride & 3u) || stride > 4096 ||
	    (x_offset != 0xffffffffu && (uint64_t)x_offset + 4 + cl > stride) || cl > 72) {
		return fail("ec_schnorr_verify_msg_all_batch: bad argument (key_fmt ECAMD_PT_AFFINE / _PROJECTIVE; hash_type 1 .. 4 (SHA-224 / 256 / 384 / 512); stride a "
			    "multiple of 4 in 4 .. 4096; the blank for the key's x, if any, inside the slot)");
	}
	int ret = 0;
	{
		std::lock_guard<std::mutex> lk(ctx->mu);
		HIPCHK(hipSetDevice(ctx->device));
		if (schnorr_msm_available(cv, r_fmt)) {
			ret = -1;
			uint8_t seed[32];
			const size_t kw = (key_fmt == ECAMD_PT_PROJECTIVE ? 3 : 2) * cl;
			// batch-wide arrays: 24 s, 25 q - e, 26 keys, 27 commitments, 28 the front end's flag word and the verdicts
			const uint32_t pieces = (n + ctx->max_chunk - 1) / ctx->max_chunk;
			if (!msm_seed(ctx, seed) && !ensure(&ctx->stage[24], &ctx->stage_bytes[24], (size_t)n * ql) &&
			    !ensure(&ctx->stage[25], &ctx->stage_bytes[25], (size_t)n * ql) && !ensure(&ctx->stage[26], &ctx->stage_bytes[26], (size_t)n * 2 * cl) &&
			    !ensure(&ctx->stage[27], &ctx->stage_bytes[27], (size_t)n * rl) && !ensure(&ctx->stage[28], &ctx->stage_bytes[28], 256 + (size_t)pieces)) {
				uint8_t **S = ctx->stage;
				uint32_t *d_flag = (uint32_t *)S[28];
				uint8_t *d_verdicts = S[28] + 256;
				uint32_t done = 0;
				PublicScalars pub_scope(ctx);   // everything a verification multiplies by is public
				EcamdSchnorrPrepArgs P;
				memset(&P, 0, sizeof(P));
				big_to_be(P.p_be, (int)cl, cv->p);
				EcamdSchnorrNeArgs N;
				memset(&N, 0, sizeof(N));
				for (int w = 0; w < 18; w++) {
					N.q[w] = (size_t)w < cv->q.size() ? cv->q[(size_t)w] : 0;
				}
				bool first = true;
				const std::vector<HostArr> arrs = {{keys, nullptr, kw}, {sigs, nullptr, sl}, {hash_slots, nullptr, stride}};
				// One piece, by buckets: the combination's per-item stages (the 2n points' import, the scalars, the filing) on every chunk as it
				// lands, in even chunks of 2^17 items -- what waits for the last chunk is the ranking, the additions and the reduction
				// ($ECAMD_NO_SCHNORR_STREAM: the whole combination after the last chunk, as for batches of several pieces)
				const bool streamed = schnorr_msm_streams(ctx, cv, n);
				SchnorrStreamStep step = {1, 0, 0};
				int begun = 0;
				if (streamed) {
					StreamScope scope(ctx, ctx->stream);
					begun = schnorr_msm_dev_locked(ctx, cv, n, S[24], S[25], S[26], S[27], r_fmt, seed, 0, d_verdicts, nullptr, nullptr, ctx->stream, 0, &step);
				}
				int rc = begun ? -1 : host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &,
										     hipStream_t s, const std::function<int()> &) {
					if (first) {
						HIPCHK(hipMemsetAsync(d_flag, 0, 4, s));
						HIPCHK(hipMemsetAsync(d_verdicts, 1, pieces, s));
						first = false;
					}
					const uint8_t *d_aff = ip[0], *d_kst = nullptr;
					if (key_fmt == ECAMD_PT_PROJECTIVE) {
						if (ensure(&ctx->stage[20], &ctx->stage_bytes[20], (size_t)m * 2 * cl) || ensure(&ctx->stage[21], &ctx->stage_bytes[21], m)) {
							return -1;
						}
						EcamdPrjInArgs I;
						I.in = ip[0];
						I.aff = S[20];
						I.pre = S[21];
						I.n = m;
						I.clen = (uint32_t)cl;
						I.for_mul = 0;
						I.slot = cv->slot;
						HIPCHK(launch_prj_import(cv, I, s));
						d_aff = S[20];
						d_kst = S[21];
					}
					P.keys_aff = d_aff;
					P.kst = d_kst;
					P.sigs = ip[1];
					P.slots = const_cast<uint8_t *>(ip[2]);   // the staged copy of the caller's slots
					P.keys_out = S[26] + (size_t)done * 2 * cl;
					P.s_out = S[24] + (size_t)done * ql;
					P.r_out = S[27] + (size_t)done * rl;
					P.flag = d_flag;
					P.n = m;
					P.clen = (uint32_t)cl;
					P.qlen = (uint32_t)ql;
					P.rlen = (uint32_t)rl;
					P.stride = stride;
					P.x_off = x_offset;
					P.even_y = r_fmt ? 1u : 0u;
					HIPCHK(ecamd_launch_schnorr_prep(P, s));
					if (ecdsa_hash_stage(ctx, hash_type, m, ip[2], stride, (uint32_t)hl, s)) {
						return -1;
					}
					N.dig = S[17];
					N.ne = S[25] + (size_t)done * ql;
					N.n = m;
					N.hlen = (uint32_t)hl;
					N.qlen = (uint32_t)ql;
					HIPCHK(ecamd_launch_schnorr_ne(cv->qnw, N, s));
					if (streamed) {
						step.mode = 2;
						step.first = done;
						step.count = m;
						if (schnorr_msm_dev_locked(ctx, cv, n, S[24], S[25], S[26], S[27], r_fmt, seed, 0, d_verdicts, nullptr, nullptr, s, 0, &step)) {
							return -1;
						}
					}
					done += m;
					return 0;
				}, streamed ? (1u << 17) : 0u);
				if (!rc) {
					hipStream_t s = ctx->stream;
					StreamScope scope(ctx, s);
					uint32_t pc = 0;
					if (streamed) {
						step.mode = 3;
						rc = schnorr_msm_dev_locked(ctx, cv, n, S[24], S[25], S[26], S[27], r_fmt, seed, 0, d_verdicts, nullptr, nullptr, s, 0, &step);
					}
					for (uint32_t off = 0; off < n && !rc && !streamed; off += ctx->max_chunk, pc++) {
						const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
						rc = schnorr_msm_dev_locked(ctx, cv, m, S[24] + (size_t)off * ql, S[25] + (size_t)off * ql, S[26] + (size_t)off * 2 * cl,
									    S[27] + (size_t)off * rl, r_fmt, seed, pc, d_verdicts + pc, nullptr, nullptr, s);
					}
					std::vector<uint8_t> v(pieces, 1);
					uint32_t flag = 1;
					if (!rc && hipMemcpyAsync(v.data(), d_verdicts, pieces, hipMemcpyDeviceToHost, s) == hipSuccess &&
					    hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
						int ok = flag == 0;
						for (uint32_t k = 0; k < pieces; k++) {
							ok = ok && v[k] == 0;
						}
						*all_valid = ok;
						ret = 0;
					} else {
						(void)hipStreamSynchronize(s);
						if (!rc) {
							ret = fail("ec_schnorr_verify_msg_all_batch: reading the verdict back failed");
						}
					}
				}
			}
			memset(seed, 0, sizeof(seed));
		}   // else: not decided here
	}
	msm_seed_discard(ctx);
	return ret;
}

extern "C" uint32_t ecamd_debug_schnorr_msm_words(const ecamd_curve *cv)
{
	int pb, fl, sl;
	return (cv && schnorr_msm_unit(cv, &pb, &fl, &sl)) ? ecamd_g29_msm_rec_words(pb, fl) : 0u;
}

extern "C" int ec_schnorr_verify_all_available(const ecamd_curve *cv, int r_fmt) { return (cv && schnorr_msm_available(cv, r_fmt)) ? 1 : 0; }

// test hook: the combination with a caller-chosen seed; z_out (n x 16 little-endian) and sum_out (the lanes' Jacobian sum: X, Y, Z digits
// of the unit and an "is infinity" word, ecamd_debug_schnorr_msm_words() words) may be NULL
extern "C" int ecamd_debug_schnorr_msm(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *s, const uint8_t *ne, const uint8_t *keys_aff,
				       const uint8_t *r, int r_fmt, const uint8_t seed[32], int *accept, uint8_t *z_out, uint32_t *sum_out)
{
	if (schnorr_args_ok("ecamd_debug_schnorr_msm", ctx, cv, n, s, ne, keys_aff, r, r_fmt, accept) || !seed) {
		return -1;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	if (n > ctx->max_chunk || !schnorr_msm_available(cv, r_fmt)) {
		return fail("ecamd_debug_schnorr_msm: needs a radix-2^29 unit for the curve and 0 < n <= max_chunk");
	}
	return schnorr_msm_host_locked(ctx, cv, n, s, ne, keys_aff, r, r_fmt, seed, accept, z_out, sum_out);
}

static void msm_seed_discard(ecamd_ctx *ctx);
static int eddsa_verify_all_batch_dev_impl(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *d_pubkeys,
					     const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_verdict,
					     void *hip_stream);
extern "C" int ec_eddsa_verify_all_batch_dev(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *d_pubkeys,
					     const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_verdict,
					     void *hip_stream)
{
	const int r = eddsa_verify_all_batch_dev_impl(ctx, cv_in, n, d_pubkeys, d_sigs, d_hram, hram_len, d_verdict, hip_stream);
	msm_seed_discard(ctx);
	return r;
}
static int eddsa_verify_all_batch_dev_impl(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *d_pubkeys,
					     const void *d_sigs, const void *d_hram, uint32_t hram_len, void *d_verdict,
					     void *hip_stream)
{
	if (!ctx) {
		return fail("ec_eddsa_verify_all_batch_dev: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	if (eddsa_args_ok("ec_eddsa_verify_all_batch_dev", ctx, cv_in, n, d_pubkeys, d_sigs, d_hram, d_verdict, hram_len)) {
		return -1;
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	const bool e448 = cv->pbits == 448;
	if (n == 0 || (e448 ? !eddsa448_msm_available(cv) : !eddsa_msm_available(cv))) {
		return fail("ec_eddsa_verify_all_batch_dev: needs n > 0 and Ed25519 (the WEI25519 handle on the 2^255 - 19 unit) or Ed448 (WEI448 on the Goldilocks unit)");
	}
	uint8_t seed[32];
	if (msm_seed(ctx, seed)) {
		return -1;
	}
	HIPCHK(hipSetDevice(ctx->device));
	hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
	StreamScope scope(ctx, s);
	HIPCHK(hipMemsetAsync(d_verdict, 0, 1, s));
	uint32_t pc = 0;
	for (uint32_t off = 0; off < n; off += ctx->max_chunk, pc++) {
		const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
		if (e448) {
			if (eddsa448_msm_dev_locked(ctx, cv, m, (const uint8_t *)d_pubkeys + (size_t)off * 57, (const uint8_t *)d_sigs + (size_t)off * 114,
						    (const uint8_t *)d_hram + (size_t)off * 114, seed, pc, (uint8_t *)d_verdict, s)) {
				return -1;
			}
			continue;
		}
		if (eddsa_msm_dev_locked(ctx, cv, m, (const uint8_t *)d_pubkeys + (size_t)off * 32, (const uint8_t *)d_sigs + (size_t)off * 64,
					 (const uint8_t *)d_hram + (size_t)off * 64, seed, pc, (uint8_t *)d_verdict, nullptr, nullptr, s)) {
			return -1;
		}
	}
	return 0;
}

// test hook: the combination with a caller-chosen seed; z_out (n x 16 little-endian z_i) and sum_out (36 words: X, Y, Z, T of
// the sum before the cofactor, radix-2^29 digits of lazily reduced residues) may be NULL.  n <= the context's max_chunk.
extern "C" int ecamd_debug_eddsa_msm(ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *pubkeys,
				     const uint8_t *sigs, const uint8_t *hram, const uint8_t seed[32], int *accept, uint8_t *z_out,
				     uint32_t *sum_out)
{
	if (!ctx || !accept || !seed) {
		return fail("ecamd_debug_eddsa_msm: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	uint8_t dummy = 0;
	if (eddsa_args_ok("ecamd_debug_eddsa_msm", ctx, cv_in, n, pubkeys, sigs, hram, &dummy, 64)) {
		return -1;
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (n == 0 || n > ctx->max_chunk || !eddsa_msm_available(cv)) {
		return fail("ecamd_debug_eddsa_msm: needs Ed25519 on the 2^255 - 19 unit and 0 < n <= max_chunk");
	}
	HIPCHK(hipSetDevice(ctx->device));
	return eddsa_msm_host_locked(ctx, cv, n, pubkeys, sigs, hram, seed, accept, z_out, sum_out);
}

// Whole-batch predicate of ec_verify_batch (sig/sig_algs.c:675, eddsa_verify_batch sig/eddsa.c:2904): libecc accepts the
// batch when one random linear combination of the cofactored equations vanishes, which holds when every signature verifies
// and fails otherwise except with probability ~2^-128 over its random z_i.  Here every item is verified (the batch is the
// parallel dimension already), so the bit is the exact conjunction and the first rejected index comes for free.
// the seed of ecamd_ctx_set_msm_seed keys ONE whole-batch call: whichever path that call took (Ed448, a batch below msm_min,
// mode 0, an error), it is gone afterwards
static void msm_seed_discard(ecamd_ctx *ctx)
{
	if (ctx) {
		std::lock_guard<std::mutex> lk(ctx->mu);
		if (ctx->msm_seed_valid) {
			memset(ctx->msm_seed_bytes, 0, sizeof(ctx->msm_seed_bytes));
			ctx->msm_seed_valid = false;
		}
	}
}
static int eddsa_verify_all_batch_impl(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys,
				       const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
				       uint32_t *first_rejected);
extern "C" int ec_eddsa_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys,
					 const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
					 uint32_t *first_rejected)
{
	const int r = eddsa_verify_all_batch_impl(ctx, cv, n, pubkeys, sigs, hram, hram_len, all_valid, first_rejected);
	msm_seed_discard(ctx);
	return r;
}
static int eddsa_verify_all_batch_impl(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *pubkeys,
				       const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
				       uint32_t *first_rejected)
{
	if (!all_valid || n == 0) {
		return fail("ec_eddsa_verify_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	*all_valid = 0;
	if (first_rejected) {
		*first_rejected = n;
	}
	// large Ed25519 batches: the reference's own batch equation as one multi-scalar multiplication first; only a batch it
	// rejects is verified item by item (for the first rejected index, and because a rejection of the combination proves
	// that some item fails while the item-by-item results say which)
	if (ctx && cv && cv->ctx == ctx && pubkeys && sigs && hram && hram_len == 64 && cv->pbits == 255) {
		std::lock_guard<std::mutex> lk(ctx->mu);
		ecamd_curve *cw = const_cast<ecamd_curve *>(cv);
		if (ctx->eddsa_msm != 0 && (ctx->eddsa_msm == 2 || n >= ctx->msm_min)) {
			if (cw->ed_state == 0) {
				ed_setup(cw);
			}
			if (eddsa_msm_available(cw)) {
				HIPCHK(hipSetDevice(ctx->device));
				int accept = 0;
				if (eddsa_msm_host_locked(ctx, cw, n, pubkeys, sigs, hram, nullptr, &accept, nullptr, nullptr)) {
					return -1;
				}
				if (accept) {
					*all_valid = 1;
					return 0;
				}
			}
		}
	}
	// Ed448 (round 6): the same, on the Weierstrass model with the cofactored final test (eddsa448_msm_dev_locked)
	if (ctx && cv && cv->ctx == ctx && pubkeys && sigs && hram && hram_len == 114 && cv->pbits == 448) {
		std::lock_guard<std::mutex> lk(ctx->mu);
		ecamd_curve *cw = const_cast<ecamd_curve *>(cv);
		if (ctx->eddsa_msm != 0 && (ctx->eddsa_msm == 2 || n >= ctx->msm_min)) {
			if (cw->ed448_state == 0) {
				ed448_setup(cw);
			}
			if (eddsa448_msm_available(cw)) {
				HIPCHK(hipSetDevice(ctx->device));
				int accept = 0;
				if (eddsa448_msm_host_locked(ctx, cw, n, pubkeys, sigs, hram, &accept)) {
					return -1;
				}
				if (accept) {
					*all_valid = 1;
					return 0;
				}
			}
		}
	}
	std::vector<uint8_t> res(n, 1);
	if (ec_eddsa_verify_batch(ctx, cv, n, pubkeys, sigs, hram, hram_len, res.data())) {
		return -1;
	}
	uint32_t i = 0;
	while (i < n && res[i] == 0) {   // any non-zero byte is a rejection
		i++;
	}
	*all_valid = (i == n) ? 1 : 0;
	if (first_rejected) {
		*first_rejected = i;
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// Ed25519 signing: the two device-side steps of _eddsa_sign (sig/eddsa.c:1554-1870) around the caller's hashes
// ------------------------------------------------------------------------------------------
static int eddsa_sign_setup(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const void *a, const void *b,
			    const void *c, EcamdEdSignArgs *T)
{
	if (!ctx || !cv_in || cv_in->ctx != ctx || (n && (!a || !b || !c))) {
		return fail(std::string(fn) + ": bad argument");
	}
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	memset(T, 0, sizeof(*T));
	if (cv->pbits == 448) {
		if (cv->ed448_state == 0) {
			ed448_setup(cv);
		}
		if (cv->ed448_state < 0) {
			return fail(std::string(fn) + ": Ed448 signing needs the WEI448 curve handle");
		}
		memcpy(T->alpha, cv->ed448_tmpl.alpha, sizeof(T->alpha));
		memcpy(T->A3, cv->ed448_tmpl.A3, sizeof(T->A3));
		big_store(T->c4, 17, big_from_be(cv->ed448_c4, 56));
		T->is448 = 1;
	} else {
		if (cv->ed_state == 0) {
			ed_setup(cv);
		}
		if (cv->ed_state < 0) {
			return fail(std::string(fn) + ": EdDSA signing needs the WEI25519 or the WEI448 curve handle");
		}
		memcpy(T->alpha, cv->ed_tmpl.alpha, sizeof(T->alpha));
		memcpy(T->A3, cv->ed_tmpl.A3, sizeof(T->A3));
	}
	T->slot = cv->slot;
	T->qslot = cv->qslot;
	return 0;
}

// stage: 3 r big-endian, 4 [r]G, 5 its status
static int eddsa_sign_R_dev_locked(ecamd_ctx *ctx, const ecamd_curve *cv, const EcamdEdSignArgs &T, uint32_t n,
				   const uint8_t *d_rhash, uint8_t *d_Renc, uint8_t *d_status, hipStream_t s)
{
	if (n > ctx->max_chunk) {  // bound the scratch: pieces of max_chunk items, in order on the stream
		for (uint32_t off = 0; off < n; off += ctx->max_chunk) {
			const uint32_t m = (n - off) < ctx->max_chunk ? (n - off) : ctx->max_chunk;
			if (eddsa_sign_R_dev_locked(ctx, cv, T, m, d_rhash + (size_t)off * (T.is448 ? 114 : 64), d_Renc + (size_t)off * (T.is448 ? 57 : 32),
						    d_status + off, s)) {
				return -1;
			}
		}
		return 0;
	}
	const uint32_t cl = (uint32_t)cv->clen;   // 32 / 56
	if (ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)n * cl) || ensure(&ctx->stage[4], &ctx->stage_bytes[4], (size_t)n * 2 * cl) ||
	    ensure(&ctx->stage[5], &ctx->stage_bytes[5], n)) {
		return -1;
	}
	uint8_t **S = ctx->stage;
	EcamdEdSignArgs A = T;
	A.n = n;
	A.r_hash = d_rhash;
	A.r_be = S[3];
	HIPCHK(ecamd_launch_ed_sign_r(A, s));
	if (smul_dev_locked(ctx, cv, n, S[3], cl, nullptr, S[4], S[5], s)) {   // prj_pt_mul(r, G) (:1776; Ed448: r / 4, :1746)
		return -1;
	}
	A.Rw = S[4];
	A.stR = S[5];
	A.out = d_Renc;
	A.status = d_status;
	HIPCHK(launch_ed_sign_enc(cv, A, s));
	return 0;
}

extern "C" int ec_eddsa_sign_R_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc,
				     uint8_t *status)
{
	if (!ctx) {
		return fail("ec_eddsa_sign_R_batch: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	EcamdEdSignArgs T;
	if (eddsa_sign_setup("ec_eddsa_sign_R_batch", ctx, cv, n, r_hash, R_enc, status, &T)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const std::vector<HostArr> arrs = {{r_hash, nullptr, (size_t)(T.is448 ? 114 : 64)}, {nullptr, R_enc, (size_t)(T.is448 ? 57 : 32)}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		return eddsa_sign_R_dev_locked(ctx, cv, T, m, ip[0], op[1], op[2], s);
	});
}

// eddsa_export_pub_key in batch (sig/eddsa.c:795-860): the projective Weierstrass point an ec_pub_key holds ->
// prj_pt_shortw_to_aff_pt_edwards -> eddsa_encode_point, i.e. the octets the verifier hashes as "A".  (libecc spends about
// 1.5 ms of CPU per call on it -- it rebuilds the curve maps every time --, which is what made ec_verify_batch through
// libsign_amd.so host-bound by three orders of magnitude before this entry point existed.)
extern "C" int ec_eddsa_encode_point_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *points_prj,
					   uint8_t *enc, uint8_t *status)
{
	if (!ctx) {
		return fail("ec_eddsa_encode_point_batch: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	EcamdEdSignArgs T;
	if (eddsa_sign_setup("ec_eddsa_encode_point_batch", ctx, cv, n, points_prj, enc, status, &T)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const size_t cl = (size_t)cv->clen, kl = T.is448 ? 57 : 32;
	const std::vector<HostArr> arrs = {{points_prj, nullptr, 3 * cl}, {nullptr, enc, kl}, {nullptr, status, 1}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		// stage: 3 affine points, 4 import status (0 / 1 error / 2 infinity: the encode kernel's convention)
		if (ensure(&ctx->stage[3], &ctx->stage_bytes[3], (size_t)m * 2 * cl) || ensure(&ctx->stage[4], &ctx->stage_bytes[4], m)) {
			return -1;
		}
		EcamdPrjInArgs I;
		I.in = ip[0];
		I.aff = ctx->stage[3];
		I.pre = ctx->stage[4];
		I.n = m;
		I.clen = (uint32_t)cl;
		I.for_mul = 0;
		I.slot = cv->slot;
		HIPCHK(launch_prj_import(cv, I, s));
		EcamdEdSignArgs A = T;
		A.n = m;
		A.Rw = ctx->stage[3];
		A.stR = ctx->stage[4];
		A.out = op[1];
		A.status = op[2];
		HIPCHK(launch_ed_sign_enc(cv, A, s));
		return 0;
	});
}

extern "C" int ec_eddsa_sign_S_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *r_hash,
				     const uint8_t *hram, const uint8_t *a_scalars, uint8_t *S_out)
{
	if (!ctx || (n && !S_out)) {
		return fail("ec_eddsa_sign_S_batch: bad argument");
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	EcamdEdSignArgs T;
	if (eddsa_sign_setup("ec_eddsa_sign_S_batch", ctx, cv, n, r_hash, hram, a_scalars, &T)) {
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	HIPCHK(hipSetDevice(ctx->device));
	const size_t hl = T.is448 ? 114 : 64, kl = T.is448 ? 57 : 32;
	const std::vector<HostArr> arrs = {{r_hash, nullptr, hl}, {hram, nullptr, hl}, {a_scalars, nullptr, kl}, {nullptr, S_out, kl}};
	return host_pipeline(ctx, cv->pbits, n, arrs, [&](uint32_t m, const std::vector<const uint8_t *> &ip, const std::vector<uint8_t *> &op,
					       hipStream_t s, const std::function<int()> &) {
		EcamdEdSignArgs A = T;
		A.n = m;
		A.r_hash = ip[0];
		A.hram = ip[1];
		A.a = ip[2];
		A.out = op[3];
		HIPCHK(ecamd_launch_ed_sign_S(A, s));
		return 0;
	});
}

// ------------------------------------------------------------------------------------------
// scalar multiplication / normalisation with a choice of point wire formats
// ------------------------------------------------------------------------------------------
static int pt_fmt_batch(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *scalars,
			uint32_t slen, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status,
			bool mul)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!out || !status || (mul && !scalars) || (!mul && !points)))) {
		return fail(std::string(fn) + ": bad argument");
	}
	if ((in_fmt != 0 && in_fmt != 1) || (out_fmt != 0 && out_fmt != 1)) {
		return fail(std::string(fn) + ": point format must be ECAMD_PT_AFFINE or ECAMD_PT_PROJECTIVE");
	}
	if (mul && (slen == 0 || slen > 1024)) {
		return fail(std::string(fn) + ": scalar_len must be in 1..1024");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t clen = (size_t)cv->clen, alen = 2 * clen;
	const size_t ilen = (in_fmt ? 3 : 2) * clen, olen = (out_fmt ? 3 : 2) * clen;
	// stage: 0 scalars, 1 points as given, 2 affine inputs, 3 import status, 4 affine results, 5 their status,
	//        6 output, 7 output status
	const size_t need[8] = {mul ? (size_t)n * slen : 0, points ? n * ilen : 0, n * alen, n, n * alen, n, n * olen, n};
	for (int i = 0; i < 8; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	uint8_t **S = ctx->stage;
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	if (mul) {
		HIPCHK(hipMemcpyAsync(S[0], scalars, (size_t)n * slen, hipMemcpyHostToDevice, s));
	}
	if (points) {
		HIPCHK(hipMemcpyAsync(S[1], points, n * ilen, hipMemcpyHostToDevice, s));
	}
	const uint8_t *d_aff = points ? S[1] : nullptr;  // affine inputs of the scalar multiplication (NULL: generator)
	const uint8_t *d_pre = nullptr;
	if (points && in_fmt == 1) {
		EcamdPrjInArgs I;
		I.in = S[1];
		I.aff = S[2];
		I.pre = S[3];
		I.n = n;
		I.clen = (uint32_t)clen;
		I.for_mul = mul ? 1 : 0;
		I.slot = cv->slot;
		HIPCHK(launch_prj_import(cv, I, s));
		d_aff = S[2];
		d_pre = S[3];
	}
	const uint8_t *d_res = d_aff, *d_st = nullptr;
	if (mul) {
		if (smul_dev_locked(ctx, cv, n, S[0], slen, d_aff, S[4], S[5], s)) {
			return -1;
		}
		d_res = S[4];
		d_st = S[5];
	} else if (in_fmt == 0) {
		// affine in, no multiplication: validation only, through the group-law kernel's import checks
		// (P + P is computed and dropped; its status carries the import result)
		EcamdPtArgs A;
		A.p1 = S[1];
		A.p2 = nullptr;
		A.out = S[4];
		A.status = S[5];
		A.n = n;
		A.clen = (uint32_t)clen;
		A.dbl = 1;
		A.slot = cv->slot;
		HIPCHK(ecamd_launch_pt(cv->nw, A, s));
		// a valid point may double to infinity (order 2): only the import error matters here
		std::vector<uint8_t> st(n);
		HIPCHK(hipMemcpyAsync(st.data(), S[5], n, hipMemcpyDeviceToHost, s));
		HIPCHK(hipStreamSynchronize(s));
		for (uint32_t i = 0; i < n; i++) {
			st[i] = st[i] == 1 ? 1 : 0;
		}
		HIPCHK(hipMemcpyAsync(S[5], st.data(), n, hipMemcpyHostToDevice, s));
		HIPCHK(hipStreamSynchronize(s));
		d_st = S[5];
	}
	EcamdPrjOutArgs O;
	O.aff = d_res;
	O.st = d_st;
	O.pre = d_pre;
	O.out = S[6];
	O.status = S[7];
	O.n = n;
	O.clen = (uint32_t)clen;
	O.out_prj = out_fmt;
	HIPCHK(ecamd_launch_prj_export(O, s));
	HIPCHK(hipMemcpyAsync(out, S[6], n * olen, hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(status, S[7], n, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

extern "C" int ec_prj_pt_mul_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *scalars,
				       uint32_t slen, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
				       uint8_t *status)
{
	return pt_fmt_batch("ec_prj_pt_mul_batch_fmt", ctx, cv, n, scalars, slen, points, in_fmt, out, out_fmt, status, true);
}

extern "C" int ec_prj_pt_unique_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *points, int in_fmt,
				      uint8_t *out, int out_fmt, uint8_t *status)
{
	return pt_fmt_batch("ec_prj_pt_unique_batch", ctx, cv, n, nullptr, 0, points, in_fmt, out, out_fmt, status, false);
}

// ------------------------------------------------------------------------------------------
// point decompression: aff_pt_y_from_x (curves/aff_pt.c:102) / fp_sqrt (fp/fp_sqrt.c:107)
// ------------------------------------------------------------------------------------------
static int sqrt_setup(ecamd_curve *cv)
{
	if (cv->sqrt_state == 1) {
		return 0;
	}
	const Big &p = cv->p;
	const Big one(1, 1);
	Big q = big_sub(p, one);
	uint32_t s = 0;
	while (!(q[0] & 1u)) {           // p - 1 = q 2^s
		Big t(q.size(), 0);
		for (size_t i = 0; i < q.size(); i++) {
			t[i] = (q[i] >> 1) | ((i + 1 < q.size()) ? (q[i + 1] << 31) : 0u);
		}
		big_trim(t);
		q = t;
		s++;
	}
	// z: the smallest quadratic non-residue, found as fp_sqrt finds it (counting up from 0, fp/fp_sqrt.c:197-200)
	Big half = big_sub(p, one);
	{
		Big t(half.size(), 0);
		for (size_t i = 0; i < half.size(); i++) {
			t[i] = (half[i] >> 1) | ((i + 1 < half.size()) ? (half[i + 1] << 31) : 0u);
		}
		big_trim(t);
		half = t;
	}
	Big z(1, 2);
	const Big pm1 = big_sub(p, one);
	for (uint32_t zz = 2; zz < 100000; zz++) {
		z = Big(1, zz);
		if (big_cmp(big_powmod(z, half, p), pm1) == 0) {
			break;
		}
	}
	if (big_cmp(big_powmod(z, half, p), pm1) != 0) {
		return fail("ec_aff_pt_y_from_x_batch: no small quadratic non-residue (is p prime?)");
	}
	EcamdYfromXArgs &T = cv->sqrt_tmpl;
	memset(&T, 0, sizeof(T));
	T.s = s;
	Big e = big_sub(q, one);         // (q - 1) / 2
	{
		Big t(e.size(), 0);
		for (size_t i = 0; i < e.size(); i++) {
			t[i] = (e[i] >> 1) | ((i + 1 < e.size()) ? (e[i + 1] << 31) : 0u);
		}
		big_trim(t);
		e = t;
	}
	T.ebits = (uint32_t)big_bitlen(e);
	big_store(T.e, 17, e);
	const Big R = big_mod(big_pow2(32 * cv->nw), p);
	big_store(T.c, 17, big_mulmod(big_powmod(z, q, p), R, p));
	T.clen = (uint32_t)cv->clen;
	T.slot = cv->slot;
	cv->sqrt_state = 1;
	return 0;
}

static int y_from_x_common(const char *fn, ecamd_ctx *ctx, const ecamd_curve *cv_in, uint32_t n, const uint8_t *in, uint32_t in_stride,
			   uint8_t *o1, uint8_t *o2, uint8_t *status, uint32_t mode)
{
	ecamd_curve *cv = const_cast<ecamd_curve *>(cv_in);
	if (!ctx || !cv || cv->ctx != ctx || (n && (!in || !o1 || !status || (mode == 0 && !o2)))) {
		return fail(std::string(fn) + ": bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(ctx->mu);
	HIPCHK(hipSetDevice(ctx->device));
	if (sqrt_setup(cv)) {
		return -1;
	}
	const size_t cl = (size_t)cv->clen;
	const size_t need[4] = {(size_t)n * in_stride, (size_t)n * cl * (mode ? 2 : 1), (size_t)n * cl, n};
	for (int i = 0; i < 4; i++) {
		if (ensure(&ctx->stage[i], &ctx->stage_bytes[i], need[i])) {
			return -1;
		}
	}
	hipStream_t s = ctx->stream;
	StreamScope scope(ctx, s);
	uint8_t **S = ctx->stage;
	HIPCHK(hipMemcpyAsync(S[0], in, need[0], hipMemcpyHostToDevice, s));
	EcamdYfromXArgs A = cv->sqrt_tmpl;
	A.x = S[0];
	A.xstride = in_stride;
	A.y1 = S[1];
	A.y2 = S[2];
	A.aff = S[1];
	A.status = S[3];
	A.n = n;
	A.mode = mode;
	HIPCHK(ecamd_launch_y_from_x(cv->nw, A, s));
	HIPCHK(hipMemcpyAsync(o1, S[1], need[1], hipMemcpyDeviceToHost, s));
	if (mode == 0) {
		HIPCHK(hipMemcpyAsync(o2, S[2], need[2], hipMemcpyDeviceToHost, s));
	}
	HIPCHK(hipMemcpyAsync(status, S[3], n, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	return 0;
}

extern "C" int ec_aff_pt_y_from_x_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *x, uint8_t *y1, uint8_t *y2,
					uint8_t *status)
{
	return y_from_x_common("ec_aff_pt_y_from_x_batch", ctx, cv, n, x, cv ? (uint32_t)cv->clen : 0, y1, y2, status, 0);
}

extern "C" int ec_point_decompress_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *compressed, uint8_t *out_aff,
					 uint8_t *status)
{
	return y_from_x_common("ec_point_decompress_batch", ctx, cv, n, compressed, cv ? (uint32_t)cv->clen + 1 : 0, out_aff, nullptr, status, 1);
}

// ------------------------------------------------------------------------------------------
// structured signatures and private keys: 3 header bytes in front of the raw bytes (sig/sig_algs.c:702-780, sig/ec_key.c:312-360)
// ------------------------------------------------------------------------------------------
extern "C" int ec_structured_sig_import_batch(const ecamd_curve *cv, uint32_t n, const uint8_t *structured, uint32_t structured_len,
					      int alg_type, int hash_type, uint8_t *raw_sigs, uint8_t *status)
{
	if (!cv || (n && (!structured || !raw_sigs || !status)) || structured_len <= 3 || structured_len > 3 + 255) {
		return fail("ec_structured_sig_import_batch: bad argument");
	}
	if (cv->curve_type <= 0) {
		return fail("ec_structured_sig_import_batch: the curve handle is not one of libecc's built-in curves");
	}
	const size_t raw = structured_len - 3;
	for (uint32_t i = 0; i < n; i++) {
		const uint8_t *k = structured + (size_t)i * structured_len;
		// ec_structured_sig_import_from_buf hands the three bytes back; a caller then checks them against what it expects
		// (tests/ec_utils.c:verify_bin_file): algorithm, hash, curve
		const bool ok = k[0] == (uint8_t)alg_type && k[1] == (uint8_t)hash_type && k[2] == (uint8_t)cv->curve_type;
		status[i] = ok ? 0 : 1;
		if (ok) {
			memcpy(raw_sigs + (size_t)i * raw, k + 3, raw);
		} else {
			memset(raw_sigs + (size_t)i * raw, 0, raw);
		}
	}
	return 0;
}

// ec_structured_key_pair_import_from_priv_key_buf (sig/ec_key.c:443-470) for the algorithms whose public key is Y = xG
// (ECDSA = 1, DECDSA = 14 in lib_ecc_types.h; __ecdsa_init_pub_key, sig/ecdsa_common.c:172-200): header EC_PRIVKEY (1), algorithm, curve;
// x < q or error; Y = [x]G on the device (x = 0 gives the key at infinity, as the reference's prj_pt_mul_blind does).
extern "C" int ec_structured_key_pair_import_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *priv_keys,
						   uint32_t key_len, int alg_type, uint8_t *priv_out, uint8_t *pub_prj_out, uint8_t *status)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!priv_keys || !pub_prj_out || !status))) {
		return fail("ec_structured_key_pair_import_batch: bad argument");
	}
	if (cv->curve_type <= 0) {
		return fail("ec_structured_key_pair_import_batch: the curve handle is not one of libecc's built-in curves");
	}
	const size_t ql = (size_t)cv->qlen, cl = (size_t)cv->clen;
	if (key_len <= 3 || key_len > 3 + 255) {
		return fail("ec_structured_key_pair_import_batch: key_len must be 3 + the private key length");
	}
	if (n == 0) {
		return 0;
	}
	const size_t xl = key_len - 3;
	std::vector<uint8_t> x((size_t)n * ql, 0), bad(n, 0), pts((size_t)n * 3 * cl), st(n);
	for (uint32_t i = 0; i < n; i++) {
		const uint8_t *k = priv_keys + (size_t)i * key_len;
		bool ok = k[0] == 1 && k[1] == (uint8_t)alg_type && k[2] == (uint8_t)cv->curve_type;
		const Big xv = big_from_be(k + 3, xl);       // nn_init_from_buf takes any length
		ok = ok && big_cmp(xv, cv->q) < 0;           // "Sanity check on key compliance"
		bad[i] = ok ? 0 : 1;
		if (ok) {
			big_to_be(&x[(size_t)i * ql], (int)ql, xv);
		}
		if (priv_out) {
			if (ok) {
				memcpy(priv_out + (size_t)i * ql, &x[(size_t)i * ql], ql);
			} else {
				memset(priv_out + (size_t)i * ql, 0, ql);
			}
		}
	}
	const int rc = ec_prj_pt_mul_batch_fmt(ctx, cv, n, x.data(), (uint32_t)ql, nullptr, ECAMD_PT_AFFINE, pts.data(), ECAMD_PT_PROJECTIVE, st.data());
	std::fill(x.begin(), x.end(), 0);   // private scalars
	if (rc) {
		return -1;
	}
	for (uint32_t i = 0; i < n; i++) {
		uint8_t *o = pub_prj_out + (size_t)i * 3 * cl;
		if (bad[i] || st[i] == ECAMD_ERR) {
			status[i] = ECAMD_ERR;
			memset(o, 0, 3 * cl);
		} else if (st[i] == ECAMD_INF) {
			status[i] = ECAMD_INF;          // x = 0: the key is the point at infinity (0 : 1 : 0)
			memset(o, 0, 3 * cl);
			o[2 * cl - 1] = 1;
		} else {
			status[i] = ECAMD_OK;
			memcpy(o, &pts[(size_t)i * 3 * cl], 3 * cl);
		}
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// structured public keys: ec_structured_pub_key_import_from_buf (sig/ec_key.c:312-345)
// ------------------------------------------------------------------------------------------
extern "C" int ec_structured_pub_key_import_batch(ecamd_ctx *ctx, const ecamd_curve *cv, uint32_t n, const uint8_t *keys,
						  uint32_t key_len, int alg_type, uint8_t *out_aff, uint8_t *status)
{
	if (!ctx || !cv || cv->ctx != ctx || (n && (!keys || !out_aff || !status))) {
		return fail("ec_structured_pub_key_import_batch: bad argument");
	}
	const size_t clen = (size_t)cv->clen, plen = 3 * clen, alen = 2 * clen;
	if (key_len != 3 + plen) {
		return fail("ec_structured_pub_key_import_batch: key_len must be 3 + 3 * coordinate length");
	}
	if (cv->curve_type <= 0) {
		return fail("ec_structured_pub_key_import_batch: the curve handle is not one of libecc's built-in curves");
	}
	if (n == 0) {
		return 0;
	}
	// header: EC_PUBKEY (0), the signature / ECDH algorithm, the curve type (sig/ec_key.h:31, lib_ecc_types.h)
	std::vector<uint8_t> hdr_bad(n), packed((size_t)n * plen);
	for (uint32_t i = 0; i < n; i++) {
		const uint8_t *k = keys + (size_t)i * key_len;
		hdr_bad[i] = (k[0] != 0 || k[1] != (uint8_t)alg_type || k[2] != (uint8_t)cv->curve_type) ? 1 : 0;
		memcpy(&packed[(size_t)i * plen], k + 3, plen);
	}
	// ec_pub_key_import_from_buf (:216-245): prj_pt_import_from_buf, then [q]Y = infinity when the cofactor is not 1
	if (pt_fmt_batch("ec_structured_pub_key_import_batch", ctx, cv, n, nullptr, 0, packed.data(), 1, out_aff, 0, status, false)) {
		return -1;
	}
	if (big_cmp(cv->order, cv->q) != 0) {
		std::vector<uint8_t> qb((size_t)n * cv->qlen), tmp((size_t)n * alen), st(n);
		for (uint32_t i = 0; i < n; i++) {
			big_to_be(&qb[(size_t)i * cv->qlen], cv->qlen, cv->q);
		}
		if (ec_prj_pt_mul_batch(ctx, cv, n, qb.data(), (uint32_t)cv->qlen, out_aff, tmp.data(), st.data())) {
			return -1;
		}
		for (uint32_t i = 0; i < n; i++) {
			if (status[i] == 0 && st[i] != 2) {
				status[i] = 1;  // not in the subgroup of the generator
			}
			if (status[i] == 2) {
				// infinity (0 : Y : 0) passes check_prj_pt_order; the degenerate (0 : 0 : 0) fails in its additions
				bool allzero = true;
				for (size_t b = 0; b < plen && allzero; b++) {
					allzero = packed[(size_t)i * plen + b] == 0;
				}
				if (allzero) {
					status[i] = 1;
				}
			}
		}
	}
	for (uint32_t i = 0; i < n; i++) {
		if (hdr_bad[i]) {
			status[i] = 1;
		}
		if (status[i] != 0) {
			memset(out_aff + (size_t)i * alen, 0, alen);
		}
	}
	return 0;
}
