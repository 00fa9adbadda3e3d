// tests/multi_host_shim.cpp -- test infrastructure: libecc_amd/csrc/ecamd_multi.cpp ITSELF compiled for the host (g++, the real HIP /
// RCCL headers for the types only, nothing linked), with every single-device entry point of include/libecc_amd.h it calls replaced
// by a recorder.  A recorder stores (function, rank, n, every pointer argument, every integer argument); tests/test_multi_host.py
// hands in fake base addresses that are never dereferenced and checks that rank r's pointers are base + lo(r) * <item size>, the item
// sizes restated in the test from the header's comments -- so a shard offset that does not follow the curve's lengths (the
// OFF(sigs, 64) of round 4 on 114-octet Ed448 signatures) fails on the CPU box.
#include "../libecc_amd/csrc/ecamd_multi.cpp"

#include <map>

struct ecamd_ctx {
	int device;
	int rank;  // creation order inside the multi-context
	ecamd_host_ready_fn ready;
	void *ready_arg;
};
struct ecamd_curve {
	int cl, ql;
};

struct Rec {
	std::string fn;
	int rank;
	uint32_t n;
	std::vector<uintptr_t> ptrs;
	std::vector<int64_t> ints;
};
static std::mutex g_mu;
static std::vector<Rec> g_rec;
static int g_next_rank = 0;
static thread_local std::string g_err;
static int g_fail_rank = -1;  // the rank whose next batch call fails (error propagation test)

static int rec(const char *fn, ecamd_ctx *ctx, uint32_t n, std::vector<const void *> ptrs, std::vector<int64_t> ints = {})
{
	std::lock_guard<std::mutex> lk(g_mu);
	Rec r;
	r.fn = fn;
	r.rank = ctx->rank;
	r.n = n;
	for (const void *p : ptrs) {
		r.ptrs.push_back((uintptr_t)p);
	}
	r.ints = ints;
	g_rec.push_back(r);
	if (ctx->rank == g_fail_rank) {
		g_err = "stub failure";
		return -1;
	}
	return 0;
}

void ecamd_set_error(const char *msg) { g_err = msg ? msg : ""; }

extern "C" {
const char *ecamd_last_error(void) { return g_err.c_str(); }
int ecamd_device_count(void) { return 8; }
int ecamd_ctx_create(ecamd_ctx **ctx, int device)
{
	*ctx = new ecamd_ctx{device, g_next_rank++, nullptr, nullptr};
	return 0;
}
void ecamd_ctx_destroy(ecamd_ctx *ctx) { delete ctx; }
int ecamd_ctx_set_host_ready_hook(ecamd_ctx *ctx, ecamd_host_ready_fn fn, void *arg)
{
	ctx->ready = fn;
	ctx->ready_arg = arg;
	return 0;
}
int ecamd_ctx_discard_msm_seed(ecamd_ctx *ctx) { return rec("ecamd_ctx_discard_msm_seed", ctx, 0, {}, {}); }
int ecamd_ctx_set_msm_seed(ecamd_ctx *ctx, const uint8_t seed[32]) { return rec("ecamd_ctx_set_msm_seed", ctx, 0, {}, {seed[0], seed[1]}); }
int ecamd_ctx_set_secret_scalars(ecamd_ctx *ctx, int on) { return rec("ecamd_ctx_set_secret_scalars", ctx, 0, {}, {on}); }
void *ecamd_ctx_stream(ecamd_ctx *) { return nullptr; }
int ecamd_ctx_wipe_scratch(ecamd_ctx *ctx) { return rec("ecamd_ctx_wipe_scratch", ctx, 0, {}); }
int ecamd_curve_by_name(ecamd_ctx *, const char *name, ecamd_curve **curve)
{
	// coordinate / order lengths in octets of the curves the test walks (lib_ecc_config.h's curve list; SECP224K1's order is one
	// octet longer than its field, WEI448's EdDSA encodings are 57 octets on a 56-octet field)
	static const std::map<std::string, std::pair<int, int>> T = {
		{"SECP192R1", {24, 24}}, {"SECP224K1", {28, 29}}, {"SECP256R1", {32, 32}}, {"SECP384R1", {48, 48}},
		{"SECP521R1", {66, 66}}, {"WEI25519", {32, 32}},  {"WEI448", {56, 56}},    {"BRAINPOOLP512R1", {64, 64}},
	};
	auto it = T.find(name);
	if (it == T.end()) {
		g_err = "unknown curve";
		return -1;
	}
	*curve = new ecamd_curve{it->second.first, it->second.second};
	return 0;
}
int ecamd_curve_from_params(ecamd_ctx *, const uint8_t *, uint32_t p_len, const uint8_t *, uint32_t, const uint8_t *, uint32_t, const uint8_t *,
			    uint32_t, const uint8_t *, uint32_t, const uint8_t *, uint32_t, const uint8_t *, uint32_t gen_order_len, ecamd_curve **curve)
{
	*curve = new ecamd_curve{(int)p_len, (int)gen_order_len};
	return 0;
}
int ecamd_curve_coord_len(const ecamd_curve *c) { return c->cl; }
int ecamd_curve_order_len(const ecamd_curve *c) { return c->ql; }
void ecamd_curve_free(ecamd_curve *c) { delete c; }

// the producer hook is fired once per call with the shard-local range, as ecamd_host.cpp's pipeline does chunk by chunk
static void fire_ready(ecamd_ctx *ctx, uint32_t n)
{
	if (ctx->ready) {
		ctx->ready(ctx->ready_arg, 0, n);
	}
}

int ec_ecccdh_derive_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *privs, const uint8_t *peers_aff, uint8_t *secrets,
			   uint8_t *status)
{
	return rec(__func__, ctx, n, {privs, peers_aff, secrets, status});
}
int ec_ecdsa_sign_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *privs, const uint8_t *nonces, const uint8_t *digests,
			uint32_t digest_len, uint8_t *sigs, uint8_t *status)
{
	return rec(__func__, ctx, n, {privs, nonces, digests, sigs, status}, {digest_len});
}
int ec_ecdsa_sign_msg_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw, int hash_type,
			    const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status)
{
	return rec(__func__, ctx, n, {privs, nonce_raw, msg_slots, sigs, status}, {hash_type, msg_stride});
}
int ec_ecdsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys_aff, const uint8_t *sigs, const uint8_t *digests,
			  uint32_t digest_len, uint8_t *result)
{
	return rec(__func__, ctx, n, {pubkeys_aff, sigs, digests, result}, {digest_len});
}
int ec_ecdsa_verify_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys, int pub_fmt, const uint8_t *sigs,
			      const uint8_t *digests, uint32_t digest_len, uint8_t *result)
{
	return rec(__func__, ctx, n, {pubkeys, sigs, digests, result}, {pub_fmt, digest_len});
}
int ec_ecdsa_verify_msg_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys, int pub_fmt, const uint8_t *sigs,
				  int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result)
{
	fire_ready(ctx, n);
	return rec(__func__, ctx, n, {pubkeys, sigs, msg_slots, result}, {pub_fmt, hash_type, msg_stride});
}
int ec_eddsa_encode_point_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *points_prj, uint8_t *enc, uint8_t *status)
{
	return rec(__func__, ctx, n, {points_prj, enc, status});
}
int ec_eddsa_sign_R_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc, uint8_t *status)
{
	return rec(__func__, ctx, n, {r_hash, R_enc, status});
}
int ec_eddsa_sign_S_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *r_hash, const uint8_t *hram, const uint8_t *a_scalars,
			  uint8_t *S_out)
{
	return rec(__func__, ctx, n, {r_hash, hram, a_scalars, S_out});
}
// the shard verdict: bit r of g_bad_mask set -> rank r reports its item g_bad_item (shard-local) as the first rejected one
static uint32_t g_bad_mask = 0, g_bad_item = 0;
int ec_eddsa_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs, const uint8_t *hram,
			      uint32_t hram_len, int *all_valid, uint32_t *first_rejected)
{
	const bool bad = (g_bad_mask >> ctx->rank) & 1u;
	*all_valid = bad ? 0 : 1;
	*first_rejected = bad ? (g_bad_item < n ? g_bad_item : n - 1) : n;
	return rec(__func__, ctx, n, {pubkeys, sigs, hram}, {hram_len});
}
int ec_schnorr_verify_all_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *s, const uint8_t *ne, const uint8_t *keys_aff,
				const uint8_t *r, int r_fmt, int *all_valid)
{
	*all_valid = ((g_bad_mask >> ctx->rank) & 1u) ? 0 : 1;
	return rec(__func__, ctx, n, {s, ne, keys_aff, r}, {r_fmt});
}
int ec_eddsa_verify_msg_prj_all_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs, const uint8_t *hash_slots,
				      uint32_t stride, uint32_t a_offset, int *all_valid)
{
	*all_valid = ((g_bad_mask >> ctx->rank) & 1u) ? 0 : 1;
	return rec(__func__, ctx, n, {keys_prj, sigs, hash_slots}, {stride, a_offset});
}
int ec_schnorr_verify_msg_all_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *keys, int key_fmt, const uint8_t *sigs, int r_fmt,
				    int hash_type, const uint8_t *hash_slots, uint32_t stride, uint32_t x_offset, int *all_valid)
{
	*all_valid = ((g_bad_mask >> ctx->rank) & 1u) ? 0 : 1;
	return rec(__func__, ctx, n, {keys, sigs, hash_slots}, {key_fmt, r_fmt, hash_type, stride, x_offset});
}
int ec_eddsa_verify_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs, const uint8_t *hram,
			  uint32_t hram_len, uint8_t *result)
{
	return rec(__func__, ctx, n, {pubkeys, sigs, hram, result}, {hram_len});
}
int ec_eddsa_verify_msg_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs, const uint8_t *hash_slots,
			      uint32_t stride, uint8_t *result)
{
	fire_ready(ctx, n);
	return rec(__func__, ctx, n, {pubkeys, sigs, hash_slots, result}, {stride});
}
int ec_eddsa_verify_msg_prj_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				  const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result)
{
	fire_ready(ctx, n);
	return rec(__func__, ctx, n, {keys_prj, sigs, hash_slots, result}, {stride, a_offset});
}
int ec_eddsa_verify_ph_prj_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
				 const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots, uint32_t msg_stride,
				 uint8_t *result)
{
	fire_ready(ctx, n);
	return rec(__func__, ctx, n, {keys_prj, sigs, hash_slots, msg_slots, result}, {stride, a_offset, msg_stride});
}
int ec_key_pair_gen_raw_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *raw, uint8_t *priv_out, uint8_t *pub_out_aff,
			      uint8_t *status)
{
	return rec(__func__, ctx, n, {raw, priv_out, pub_out_aff, status});
}
int ec_prj_pt_add_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *p1_aff, const uint8_t *p2_aff, uint8_t *out_aff,
			uint8_t *status)
{
	return rec(__func__, ctx, n, {p1_aff, p2_aff, out_aff, status});
}
int ec_prj_pt_mul_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *scalars, uint32_t scalar_len, const uint8_t *points_aff,
			uint8_t *out_aff, uint8_t *status)
{
	return rec(__func__, ctx, n, {scalars, points_aff, out_aff, status}, {scalar_len});
}
int ec_prj_pt_mul_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *scalars, uint32_t scalar_len, const uint8_t *points,
			    int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	return rec(__func__, ctx, n, {scalars, points, out, status}, {scalar_len, in_fmt, out_fmt});
}
int ec_prj_pt_op_batch_fmt(ecamd_ctx *ctx, const ecamd_curve *, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2, int in_fmt, uint8_t *out,
			   int out_fmt, uint8_t *status)
{
	return rec(__func__, ctx, n, {p1, p2, out, status}, {op, in_fmt, out_fmt});
}
int ec_prj_pt_unique_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
			   uint8_t *status)
{
	return rec(__func__, ctx, n, {points, out, status}, {in_fmt, out_fmt});
}
int ec_prj_pt_unprotected_mult_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *scalars, uint32_t scalar_len,
				     uint32_t scalar_stride, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	return rec(__func__, ctx, n, {scalars, points, out, status}, {scalar_len, scalar_stride, in_fmt, out_fmt});
}
int ec_xdh_batch(ecamd_ctx *ctx, const ecamd_curve *, uint32_t n, const uint8_t *k, const uint8_t *u, uint8_t *out, uint8_t *status)
{
	return rec(__func__, ctx, n, {k, u, out, status});
}

// ---- HIP runtime symbols ecamd_multi.cpp references (the all-gather's plumbing; never reached by the shard tests) ----
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
	memcpy(dst, src, n);
	return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

// ---- the test's view of the recorder ----
void mh_reset(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_rec.clear();
	g_next_rank = 0;
	g_fail_rank = -1;
	g_bad_mask = 0;
	g_bad_item = 0;
}
void mh_set_fail_rank(int r) { g_fail_rank = r; }
void mh_set_bad(uint32_t mask, uint32_t item) { g_bad_mask = mask; g_bad_item = item; }
int mh_count(void) { return (int)g_rec.size(); }
// record i -> name (returned), rank, n, number of pointers / integers; mh_ptr / mh_int read them
const char *mh_record(int i, int *rank, uint32_t *n, int *nptr, int *nint)
{
	const Rec &r = g_rec[(size_t)i];
	*rank = r.rank;
	*n = r.n;
	*nptr = (int)r.ptrs.size();
	*nint = (int)r.ints.size();
	return r.fn.c_str();
}
uint64_t mh_ptr(int i, int k) { return (uint64_t)g_rec[(size_t)i].ptrs[(size_t)k]; }
int64_t mh_int(int i, int k) { return g_rec[(size_t)i].ints[(size_t)k]; }
}
