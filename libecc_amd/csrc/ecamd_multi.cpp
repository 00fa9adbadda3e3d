// libecc_amd/csrc/ecamd_multi.cpp -- the multi-GPU form of the C ABI: one context and
// one host thread per device, the batch cut into contiguous shards (rank r of N owns items
// [r*n/N, (r+1)*n/N) -- work per item is constant, so equal counts are balanced), no collective on the
// compute path.  Host-pointer results land in the caller's arrays straight from each device (N independent
// D2H copies); ecamd_multi_allgather is the one RCCL all-gather over xGMI for callers that keep the outputs
// device-resident on every GPU (north_star).  Built only on the public single-device entry points of
// include/libecc_amd.h, so every shard runs exactly the code the single-GPU tests pin.
//
// A device may be listed several times (one context and one shard each): the shards then share that GPU.
// That is how the 1-GPU development box exercises this file (tests/test_gpu_parity.py::test_multi_*).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enumerators only (ncclComm_t, ncclUint8, ncclResult_t): librccl itself is dlopen'ed on first use
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/libecc_amd.h"

void ecamd_set_error(const char *msg);  // ecamd_host.cpp: the thread-local string behind ecamd_last_error()

struct ecamd_multi {
	std::vector<int> devices;
	std::vector<ecamd_ctx *> ctx;
	std::mutex mu;              // one multi-call at a time (the per-device contexts serialise anyway)
	// producer hook (ecamd_multi_set_host_ready_hook): per rank a trampoline that adds the shard's first item
	ecamd_host_ready_fn ready_fn = nullptr;
	void *ready_arg = nullptr;
	struct ReadyTramp {
		ecamd_multi *m;
		uint32_t base;
	};
	std::vector<ReadyTramp> tramp;
	// RCCL, loaded on the first all-gather
	void *rccl = nullptr;
	std::vector<ncclComm_t> comms;  // one communicator per rank
	std::vector<hipStream_t> cstreams;
	std::vector<hipEvent_t> cready;  // recorded on a rank's compute stream, waited for by its gather stream
};

struct ecamd_mcurve {
	ecamd_multi *m;
	std::vector<ecamd_curve *> cv;  // one handle per rank
};

static int mfail(const std::string &s)
{
	ecamd_set_error(s.c_str());
	return -1;
}

static inline uint32_t shard_lo(uint32_t n, int r, int N) { return (uint32_t)(((uint64_t)n * (uint64_t)r) / (uint64_t)N); }

extern "C" void ecamd_multi_shard_range(uint32_t n, int rank, int nranks, uint32_t *lo, uint32_t *hi)
{
	if (nranks <= 0 || rank < 0 || rank >= nranks) {
		*lo = *hi = 0;
		return;
	}
	*lo = shard_lo(n, rank, nranks);
	*hi = shard_lo(n, rank + 1, nranks);
}

extern "C" int ecamd_multi_create(ecamd_multi **out, const int *devices, int ndev)
{
	if (!out) {
		return mfail("ecamd_multi_create: NULL out pointer");
	}
	*out = nullptr;
	std::vector<int> devs;
	if (!devices || ndev <= 0) {
		const int n = ecamd_device_count();
		for (int i = 0; i < n; i++) {
			devs.push_back(i);
		}
	} else {
		devs.assign(devices, devices + ndev);
	}
	if (devs.empty()) {
		return mfail("ecamd_multi_create: no HIP device available (this library has no CPU fallback)");
	}
	ecamd_multi *m = new ecamd_multi();
	m->devices = devs;
	for (int d : devs) {
		ecamd_ctx *c = nullptr;
		if (ecamd_ctx_create(&c, d)) {
			for (ecamd_ctx *x : m->ctx) {
				ecamd_ctx_destroy(x);
			}
			delete m;
			return -1;  // ecamd_last_error() already says why
		}
		m->ctx.push_back(c);
	}
	*out = m;
	return 0;
}

typedef ncclResult_t (*nccl_destroy_fn)(ncclComm_t);

// communicators, gather streams and events of the all-gather, and the library handle (also the clean-up of a partial set-up)
static void rccl_teardown(ecamd_multi *m, void *h)
{
	nccl_destroy_fn destroy = h ? (nccl_destroy_fn)dlsym(h, "ncclCommDestroy") : nullptr;
	for (size_t r = 0; r < m->devices.size(); r++) {
		(void)hipSetDevice(m->devices[r]);
		if (destroy && r < m->comms.size() && m->comms[r]) {
			(void)destroy(m->comms[r]);
		}
		if (r < m->cstreams.size() && m->cstreams[r]) {
			(void)hipStreamDestroy(m->cstreams[r]);
		}
		if (r < m->cready.size() && m->cready[r]) {
			(void)hipEventDestroy(m->cready[r]);
		}
	}
	m->comms.clear();
	m->cstreams.clear();
	m->cready.clear();
	if (h) {
		dlclose(h);
	}
	m->rccl = nullptr;
}

extern "C" void ecamd_multi_destroy(ecamd_multi *m)
{
	if (!m) {
		return;
	}
	if (m->rccl) {
		rccl_teardown(m, m->rccl);
	}
	for (ecamd_ctx *c : m->ctx) {
		ecamd_ctx_destroy(c);
	}
	delete m;
}

extern "C" int ecamd_multi_size(const ecamd_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" int ecamd_multi_device(const ecamd_multi *m, int rank)
{
	return (m && rank >= 0 && rank < (int)m->devices.size()) ? m->devices[(size_t)rank] : -1;
}
extern "C" ecamd_ctx *ecamd_multi_ctx(ecamd_multi *m, int rank)
{
	return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[(size_t)rank] : nullptr;
}

extern "C" void ecamd_multi_curve_free(ecamd_mcurve *c)
{
	if (!c) {
		return;
	}
	for (ecamd_curve *cv : c->cv) {
		ecamd_curve_free(cv);
	}
	delete c;
}

static int mcurve_make(ecamd_multi *m, ecamd_mcurve **out, const std::function<int(ecamd_ctx *, ecamd_curve **)> &make)
{
	if (!m || !out) {
		return mfail("ecamd_multi_curve: NULL argument");
	}
	*out = nullptr;
	ecamd_mcurve *c = new ecamd_mcurve();
	c->m = m;
	for (ecamd_ctx *x : m->ctx) {
		ecamd_curve *cv = nullptr;
		if (make(x, &cv)) {
			ecamd_multi_curve_free(c);
			return -1;
		}
		c->cv.push_back(cv);
	}
	*out = c;
	return 0;
}

extern "C" int ecamd_multi_curve_by_name(ecamd_multi *m, const char *name, ecamd_mcurve **out)
{
	return mcurve_make(m, out, [&](ecamd_ctx *x, ecamd_curve **cv) { return ecamd_curve_by_name(x, name, cv); });
}

extern "C" int ecamd_multi_curve_from_params(ecamd_multi *m, const uint8_t *p, uint32_t p_len, const uint8_t *a, uint32_t a_len,
					     const uint8_t *b, uint32_t b_len, const uint8_t *curve_order, uint32_t curve_order_len,
					     const uint8_t *gx, uint32_t gx_len, const uint8_t *gy, uint32_t gy_len,
					     const uint8_t *gen_order, uint32_t gen_order_len, ecamd_mcurve **out)
{
	return mcurve_make(m, out, [&](ecamd_ctx *x, ecamd_curve **cv) {
		return ecamd_curve_from_params(x, p, p_len, a, a_len, b, b_len, curve_order, curve_order_len, gx, gx_len, gy, gy_len,
					       gen_order, gen_order_len, cv);
	});
}

extern "C" const ecamd_curve *ecamd_multi_curve_handle(const ecamd_mcurve *c, int rank)
{
	return (c && rank >= 0 && rank < (int)c->cv.size()) ? c->cv[(size_t)rank] : nullptr;
}
extern "C" int ecamd_multi_curve_coord_len(const ecamd_mcurve *c) { return (c && !c->cv.empty()) ? ecamd_curve_coord_len(c->cv[0]) : -1; }
extern "C" int ecamd_multi_curve_order_len(const ecamd_mcurve *c) { return (c && !c->cv.empty()) ? ecamd_curve_order_len(c->cv[0]) : -1; }

// run shard(rank, lo, hi) on one host thread per rank; the first failing rank's message becomes the caller's error
static int run_sharded(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const char *fn,
		       const std::function<int(int, uint32_t, uint32_t)> &shard, bool consumes_msm_seed = false)
{
	if (!m || !c || c->m != m) {
		return mfail(std::string(fn) + ": bad argument");
	}
	if (n == 0) {
		return 0;
	}
	std::lock_guard<std::mutex> lk(m->mu);
	const int N = (int)m->ctx.size();
	std::vector<int> rc((size_t)N, 0);
	std::vector<std::string> err((size_t)N);
	auto body = [&](int r) {
		const uint32_t lo = shard_lo(n, r, N), hi = shard_lo(n, r + 1, N);
		if (hi > lo) {
			if (m->ready_fn) {
				m->tramp[(size_t)r].base = lo;
			}
			rc[(size_t)r] = shard(r, lo, hi);
			if (rc[(size_t)r]) {
				err[(size_t)r] = ecamd_last_error();  // thread-local of the worker
			}
		} else if (consumes_msm_seed) {
			// a whole-batch call: this rank has no shard, so nothing consumed its copy of the one-shot seed
			(void)ecamd_ctx_discard_msm_seed(m->ctx[(size_t)r]);
		}
	};
	if (N == 1) {
		body(0);
	} else {
		std::vector<std::thread> th;
		for (int r = 0; r < N; r++) {
			th.emplace_back(body, r);
		}
		for (std::thread &t : th) {
			t.join();
		}
	}
	for (int r = 0; r < N; r++) {
		if (rc[(size_t)r]) {
			char pre[64];
			snprintf(pre, sizeof(pre), "%s: rank %d (device %d): ", fn, r, m->devices[(size_t)r]);
			return mfail(pre + err[(size_t)r]);
		}
	}
	return 0;
}

#define OFF(p, stride) ((p) ? (p) + (size_t)lo * (stride) : nullptr)

// octets of an EdDSA point encoding on the curve behind `c` (sig/eddsa.c:93-120, EDDSA_R_LEN): 32 on WEI25519, 57 on WEI448 (coordinate
// length 56 plus the sign octet); a signature is two of them.  Every EdDSA array below is sharded by these, never by a literal.
static inline size_t eddsa_enc_len(const ecamd_mcurve *c)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c);
	return (cl == 56) ? 57 : cl;
}

extern "C" int ecamd_multi_prj_pt_mul_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars,
					    uint32_t scalar_len, const uint8_t *points_aff, uint8_t *out_aff, uint8_t *status)
{
	const size_t plen = 2 * (size_t)ecamd_multi_curve_coord_len(c);
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_mul_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_mul_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(scalars, scalar_len), scalar_len, OFF(points_aff, plen),
					   OFF(out_aff, plen), OFF(status, 1));
	});
}

extern "C" int ecamd_multi_prj_pt_mul_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars,
						uint32_t scalar_len, const uint8_t *points, int in_fmt, uint8_t *out, int out_fmt,
						uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), il = (in_fmt ? 3 : 2) * cl, ol = (out_fmt ? 3 : 2) * cl;
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_mul_batch_fmt", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_mul_batch_fmt(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(scalars, scalar_len), scalar_len, OFF(points, il),
					       in_fmt, OFF(out, ol), out_fmt, OFF(status, 1));
	});
}

extern "C" int ecamd_multi_ecdsa_verify_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys_aff,
					      const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result)
{
	const size_t plen = 2 * (size_t)ecamd_multi_curve_coord_len(c), sl = 2 * (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecdsa_verify_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecdsa_verify_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys_aff, plen), OFF(sigs, sl),
					     OFF(digests, digest_len), digest_len, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_ecdsa_verify_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
						  const uint8_t *sigs, const uint8_t *digests, uint32_t digest_len, uint8_t *result)
{
	const size_t plen = (pub_fmt ? 3 : 2) * (size_t)ecamd_multi_curve_coord_len(c), sl = 2 * (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecdsa_verify_batch_fmt", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecdsa_verify_batch_fmt(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys, plen), pub_fmt, OFF(sigs, sl),
						 OFF(digests, digest_len), digest_len, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_ecdsa_verify_msg_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, int pub_fmt,
						      const uint8_t *sigs, int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *result)
{
	const size_t plen = (pub_fmt ? 3 : 2) * (size_t)ecamd_multi_curve_coord_len(c), sl = 2 * (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecdsa_verify_msg_batch_fmt", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecdsa_verify_msg_batch_fmt(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys, plen), pub_fmt, OFF(sigs, sl), hash_type,
						     OFF(msg_slots, (size_t)msg_stride), msg_stride, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_eddsa_verify_msg_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys, const uint8_t *sigs,
						  const uint8_t *hash_slots, uint32_t stride, uint8_t *result)
{
	// Ed25519: 32-byte keys, 64-byte signatures; Ed448: 57 / 114 (coordinate length 56)
	const size_t kl = eddsa_enc_len(c);
	return run_sharded(m, c, n, "ecamd_multi_eddsa_verify_msg_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_verify_msg_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys, kl), OFF(sigs, 2 * kl), OFF(hash_slots, (size_t)stride),
						 stride, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_eddsa_verify_msg_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
						      const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, uint8_t *result)
{
	const size_t kl = 3 * (size_t)ecamd_multi_curve_coord_len(c), sl = 2 * eddsa_enc_len(c);  // signatures: 64 octets (Ed25519), 114 (Ed448)
	return run_sharded(m, c, n, "ecamd_multi_eddsa_verify_msg_prj_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_verify_msg_prj_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(keys_prj, kl), OFF(sigs, sl), OFF(hash_slots, (size_t)stride),
						     stride, a_offset, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_ecdsa_sign_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs,
					    const uint8_t *nonces, const uint8_t *digests, uint32_t digest_len, uint8_t *sigs,
					    uint8_t *status)
{
	const size_t ql = (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecdsa_sign_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecdsa_sign_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(privs, ql), OFF(nonces, ql), OFF(digests, digest_len),
					   digest_len, OFF(sigs, 2 * ql), OFF(status, 1));
	});
}

extern "C" int ecamd_multi_eddsa_verify_ph_prj_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
						     const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, const uint8_t *msg_slots,
						     uint32_t msg_stride, uint8_t *result)
{
	const size_t kl = 3 * (size_t)ecamd_multi_curve_coord_len(c), sl = 2 * eddsa_enc_len(c);
	return run_sharded(m, c, n, "ecamd_multi_eddsa_verify_ph_prj_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_verify_ph_prj_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(keys_prj, kl), OFF(sigs, sl), OFF(hash_slots, (size_t)stride),
						    stride, a_offset, OFF(msg_slots, (size_t)msg_stride), msg_stride, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_ecdsa_sign_msg_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs, const uint8_t *nonce_raw,
						int hash_type, const uint8_t *msg_slots, uint32_t msg_stride, uint8_t *sigs, uint8_t *status)
{
	const size_t ql = (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecdsa_sign_msg_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecdsa_sign_msg_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(privs, ql), OFF(nonce_raw, 2 * ql), hash_type,
					       OFF(msg_slots, (size_t)msg_stride), msg_stride, OFF(sigs, 2 * ql), OFF(status, 1));
	});
}

extern "C" int ecamd_multi_key_pair_gen_raw_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *raw, uint8_t *priv_out,
						  uint8_t *pub_out, uint8_t *status)
{
	const size_t ql = (size_t)ecamd_multi_curve_order_len(c), cl = (size_t)ecamd_multi_curve_coord_len(c);
	return run_sharded(m, c, n, "ecamd_multi_key_pair_gen_raw_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_key_pair_gen_raw_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(raw, 2 * ql), OFF(priv_out, ql), OFF(pub_out, 2 * cl),
						 OFF(status, 1));
	});
}

extern "C" int ecamd_multi_ecccdh_derive_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *privs,
					       const uint8_t *peers_aff, uint8_t *secrets, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), ql = (size_t)ecamd_multi_curve_order_len(c);
	return run_sharded(m, c, n, "ecamd_multi_ecccdh_derive_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_ecccdh_derive_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(privs, ql), OFF(peers_aff, 2 * cl), OFF(secrets, cl),
					      OFF(status, 1));
	});
}

extern "C" int ecamd_multi_xdh_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *k, const uint8_t *u,
				     uint8_t *out, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c);
	return run_sharded(m, c, n, "ecamd_multi_xdh_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_xdh_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(k, cl), OFF(u, cl), OFF(out, cl), OFF(status, 1));
	});
}

extern "C" int ecamd_multi_eddsa_verify_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys,
					      const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, uint8_t *result)
{
	// Ed25519: 32-byte keys, 64-byte signatures; Ed448: 57 / 114 (coordinate length 56)
	const size_t kl = eddsa_enc_len(c);
	return run_sharded(m, c, n, "ecamd_multi_eddsa_verify_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_verify_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys, kl), OFF(sigs, 2 * kl), OFF(hram, hram_len),
					     hram_len, OFF(result, 1));
	});
}

extern "C" int ecamd_multi_eddsa_encode_point_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *points_prj,
						    uint8_t *enc, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), kl = (cl == 56) ? 57 : cl;
	return run_sharded(m, c, n, "ecamd_multi_eddsa_encode_point_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_encode_point_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(points_prj, 3 * cl), OFF(enc, kl), OFF(status, 1));
	});
}

// ec_verify_batch's one bit for EdDSA, sharded: every device decides its shard (Ed25519 shards of at least 2^17 items with the
// multi-scalar multiplication, ec_eddsa_verify_all_batch); the batch is valid when every shard is
extern "C" int ecamd_multi_eddsa_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *pubkeys,
						  const uint8_t *sigs, const uint8_t *hram, uint32_t hram_len, int *all_valid,
						  uint32_t *first_rejected)
{
	if (!all_valid || n == 0) {
		return mfail("ecamd_multi_eddsa_verify_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	*all_valid = 0;
	if (first_rejected) {
		*first_rejected = n;
	}
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), kl = (cl == 56) ? 57 : cl;
	const int N = m ? (int)m->ctx.size() : 0;
	std::vector<int> ok((size_t)(N > 0 ? N : 1), 1);
	std::vector<uint32_t> first((size_t)(N > 0 ? N : 1), 0xffffffffu);
	if (run_sharded(m, c, n, "ecamd_multi_eddsa_verify_all_batch", [&](int r, uint32_t lo, uint32_t hi) {
		    uint32_t f = hi - lo;
		    const int rc = ec_eddsa_verify_all_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(pubkeys, kl), OFF(sigs, 2 * kl),
							     OFF(hram, hram_len), hram_len, &ok[(size_t)r], &f);
		    first[(size_t)r] = (rc == 0 && !ok[(size_t)r]) ? lo + f : 0xffffffffu;
		    return rc;
	    }, true)) {
		return -1;
	}
	int all = 1;
	uint32_t fr = n;
	for (int r = 0; r < N; r++) {
		all = all && ok[(size_t)r];
		if (first[(size_t)r] < fr) {
			fr = first[(size_t)r];
		}
	}
	*all_valid = all;
	if (first_rejected) {
		*first_rejected = all ? n : fr;
	}
	return 0;
}

// ec_verify_batch's one bit for the Schnorr-type algorithms, sharded: every device decides its shard with a combination of its own
// (rank r keys it with seed ^ r, ecamd_multi_set_msm_seed); the batch is valid when every shard is
extern "C" int ecamd_multi_schnorr_verify_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *s, const uint8_t *ne,
						    const uint8_t *keys_aff, const uint8_t *r, int r_fmt, int *all_valid)
{
	if (!all_valid || n == 0) {
		return mfail("ecamd_multi_schnorr_verify_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	*all_valid = 0;
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), ql = (size_t)ecamd_multi_curve_order_len(c), rl = r_fmt ? cl : 2 * cl;
	const int N = m ? (int)m->ctx.size() : 0;
	std::vector<int> ok((size_t)(N > 0 ? N : 1), 1);
	if (run_sharded(m, c, n, "ecamd_multi_schnorr_verify_all_batch", [&](int rk, uint32_t lo, uint32_t hi) {
		    return ec_schnorr_verify_all_batch(m->ctx[(size_t)rk], c->cv[(size_t)rk], hi - lo, OFF(s, ql), OFF(ne, ql), OFF(keys_aff, 2 * cl), OFF(r, rl), r_fmt,
						       &ok[(size_t)rk]);
	    }, true)) {
		return -1;
	}
	int all = 1;
	for (int rk = 0; rk < N; rk++) {
		all = all && ok[(size_t)rk];
	}
	*all_valid = all;
	return 0;
}

// ec_verify_batch's one bit for plain Ed25519 from projective keys, signatures and hash inputs (ec_eddsa_verify_msg_prj_all_batch): every
// device runs the front end and decides its shard with a combination of its own
extern "C" int ecamd_multi_eddsa_verify_msg_prj_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys_prj, const uint8_t *sigs,
							  const uint8_t *hash_slots, uint32_t stride, uint32_t a_offset, int *all_valid)
{
	if (!all_valid || n == 0) {
		return mfail("ecamd_multi_eddsa_verify_msg_prj_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	*all_valid = 0;
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), kl = eddsa_enc_len(c);
	const int N = m ? (int)m->ctx.size() : 0;
	std::vector<int> ok((size_t)(N > 0 ? N : 1), 1);
	if (run_sharded(m, c, n, "ecamd_multi_eddsa_verify_msg_prj_all_batch", [&](int rk, uint32_t lo, uint32_t hi) {
		    return ec_eddsa_verify_msg_prj_all_batch(m->ctx[(size_t)rk], c->cv[(size_t)rk], hi - lo, OFF(keys_prj, 3 * cl), OFF(sigs, 2 * kl), OFF(hash_slots, stride),
							     stride, a_offset, &ok[(size_t)rk]);
	    }, true)) {
		return -1;
	}
	int all = 1;
	for (int rk = 0; rk < N; rk++) {
		all = all && ok[(size_t)rk];
	}
	*all_valid = all;
	return 0;
}

// the same from keys, signatures and hash inputs (ec_schnorr_verify_msg_all_batch): every device imports, hashes and decides its shard
extern "C" int ecamd_multi_schnorr_verify_msg_all_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *keys, int key_fmt,
							const uint8_t *sigs, int r_fmt, int hash_type, const uint8_t *hash_slots, uint32_t stride,
							uint32_t x_offset, int *all_valid)
{
	if (!all_valid || n == 0) {
		return mfail("ecamd_multi_schnorr_verify_msg_all_batch: bad argument (the reference rejects num = 0 too)");
	}
	*all_valid = 0;
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), ql = (size_t)ecamd_multi_curve_order_len(c), rl = r_fmt ? cl : 2 * cl;
	const size_t kw = (key_fmt == ECAMD_PT_PROJECTIVE ? 3 : 2) * cl;
	const int N = m ? (int)m->ctx.size() : 0;
	std::vector<int> ok((size_t)(N > 0 ? N : 1), 1);
	if (run_sharded(m, c, n, "ecamd_multi_schnorr_verify_msg_all_batch", [&](int rk, uint32_t lo, uint32_t hi) {
		    return ec_schnorr_verify_msg_all_batch(m->ctx[(size_t)rk], c->cv[(size_t)rk], hi - lo, OFF(keys, kw), key_fmt, OFF(sigs, rl + ql), r_fmt, hash_type,
							   OFF(hash_slots, stride), stride, x_offset, &ok[(size_t)rk]);
	    }, true)) {
		return -1;
	}
	int all = 1;
	for (int rk = 0; rk < N; rk++) {
		all = all && ok[(size_t)rk];
	}
	*all_valid = all;
	return 0;
}

extern "C" int ecamd_multi_prj_pt_add_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *p1_aff, const uint8_t *p2_aff,
					    uint8_t *out_aff, uint8_t *status)
{
	const size_t plen = 2 * (size_t)ecamd_multi_curve_coord_len(c);
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_add_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_add_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(p1_aff, plen), OFF(p2_aff, plen), OFF(out_aff, plen),
					   OFF(status, 1));
	});
}

extern "C" int ecamd_multi_prj_pt_unique_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *points, int in_fmt,
					       uint8_t *out, int out_fmt, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), il = (in_fmt ? 3 : 2) * cl, ol = (out_fmt ? 3 : 2) * cl;
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_unique_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_unique_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(points, il), in_fmt, OFF(out, ol), out_fmt,
					      OFF(status, 1));
	});
}

extern "C" int ecamd_multi_prj_pt_op_batch_fmt(ecamd_multi *m, const ecamd_mcurve *c, int op, uint32_t n, const uint8_t *p1, const uint8_t *p2,
					       int in_fmt, uint8_t *out, int out_fmt, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), il = (in_fmt ? 3 : 2) * cl;
	const size_t ol = op >= ECAMD_PT_OP_CMP ? 1 : (out_fmt ? 3 : 2) * cl;   // the predicates: one byte per item
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_op_batch_fmt", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_op_batch_fmt(m->ctx[(size_t)r], c->cv[(size_t)r], op, hi - lo, OFF(p1, il), p2 ? OFF(p2, il) : nullptr, in_fmt,
					      out ? OFF(out, ol) : nullptr, out_fmt, OFF(status, 1));
	});
}

extern "C" int ecamd_multi_prj_pt_unprotected_mult_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *scalars,
							 uint32_t scalar_len, uint32_t scalar_stride, const uint8_t *points, int in_fmt,
							 uint8_t *out, int out_fmt, uint8_t *status)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), il = (in_fmt ? 3 : 2) * cl, ol = (out_fmt ? 3 : 2) * cl;
	return run_sharded(m, c, n, "ecamd_multi_prj_pt_unprotected_mult_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_prj_pt_unprotected_mult_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(scalars, (size_t)scalar_stride), scalar_len,
							scalar_stride, OFF(points, il), in_fmt, OFF(out, ol), out_fmt, OFF(status, 1));
	});
}

// ---- settings applied to every rank's context ----
extern "C" int ecamd_multi_set_secret_scalars(ecamd_multi *m, int on)
{
	if (!m) {
		return mfail("ecamd_multi_set_secret_scalars: NULL argument");
	}
	std::lock_guard<std::mutex> lk(m->mu);
	for (ecamd_ctx *c : m->ctx) {
		if (ecamd_ctx_set_secret_scalars(c, on)) {
			return -1;
		}
	}
	return 0;
}

// one 32-byte seed for the next whole-batch EdDSA verification: rank r keys its shard's combination with seed ^ r (first byte)
static void ready_trampoline(void *arg, uint32_t first, uint32_t count)
{
	const ecamd_multi::ReadyTramp *t = (const ecamd_multi::ReadyTramp *)arg;
	if (t->m->ready_fn) {
		t->m->ready_fn(t->m->ready_arg, t->base + first, count);
	}
}

extern "C" int ecamd_multi_set_host_ready_hook(ecamd_multi *m, ecamd_host_ready_fn fn, void *arg)
{
	if (!m) {
		return mfail("ecamd_multi_set_host_ready_hook: NULL argument");
	}
	std::lock_guard<std::mutex> lk(m->mu);
	m->ready_fn = fn;
	m->ready_arg = fn ? arg : nullptr;
	m->tramp.resize(m->ctx.size());
	for (size_t r = 0; r < m->ctx.size(); r++) {
		m->tramp[r].m = m;
		m->tramp[r].base = 0;
		if (ecamd_ctx_set_host_ready_hook(m->ctx[r], fn ? ready_trampoline : nullptr, &m->tramp[r])) {
			return mfail(std::string("ecamd_multi_set_host_ready_hook: ") + ecamd_last_error());
		}
	}
	return 0;
}

extern "C" int ecamd_multi_set_msm_seed(ecamd_multi *m, const uint8_t seed[32])
{
	if (!m || !seed) {
		return mfail("ecamd_multi_set_msm_seed: NULL argument");
	}
	std::lock_guard<std::mutex> lk(m->mu);
	for (size_t r = 0; r < m->ctx.size(); r++) {
		uint8_t sr[32];
		memcpy(sr, seed, 32);
		sr[0] ^= (uint8_t)r;
		sr[1] ^= (uint8_t)(r >> 8);
		const int rc = ecamd_ctx_set_msm_seed(m->ctx[r], sr);
		memset(sr, 0, sizeof(sr));
		if (rc) {
			return -1;
		}
	}
	return 0;
}

extern "C" int ecamd_multi_wipe_scratch(ecamd_multi *m)
{
	if (!m) {
		return mfail("ecamd_multi_wipe_scratch: NULL argument");
	}
	std::lock_guard<std::mutex> lk(m->mu);
	for (ecamd_ctx *c : m->ctx) {
		if (ecamd_ctx_wipe_scratch(c)) {
			return -1;
		}
	}
	return 0;
}

extern "C" int ecamd_multi_eddsa_sign_R_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *r_hash, uint8_t *R_enc,
					      uint8_t *status)
{
	// Ed25519: 64-byte hashes, 32-byte encodings; Ed448: 114 / 57 (coordinate length 56)
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), kl = (cl == 56) ? 57 : cl;
	return run_sharded(m, c, n, "ecamd_multi_eddsa_sign_R_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_sign_R_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(r_hash, 2 * kl), OFF(R_enc, kl), OFF(status, 1));
	});
}

extern "C" int ecamd_multi_eddsa_sign_S_batch(ecamd_multi *m, const ecamd_mcurve *c, uint32_t n, const uint8_t *r_hash, const uint8_t *hram,
					      const uint8_t *a_scalars, uint8_t *S_out)
{
	const size_t cl = (size_t)ecamd_multi_curve_coord_len(c), kl = (cl == 56) ? 57 : cl;
	return run_sharded(m, c, n, "ecamd_multi_eddsa_sign_S_batch", [&](int r, uint32_t lo, uint32_t hi) {
		return ec_eddsa_sign_S_batch(m->ctx[(size_t)r], c->cv[(size_t)r], hi - lo, OFF(r_hash, 2 * kl), OFF(hram, 2 * kl), OFF(a_scalars, kl),
					     OFF(S_out, kl));
	});
}

// ------------------------------------------------------------------------------------------
// the one collective: RCCL all-gather of equal-size device-resident shards (north_star: "RCCL gather over xGMI
// for the output points").  d_send[r]: bytes_per_rank bytes on device r; d_recv[r]: nranks * bytes_per_rank bytes
// on device r.  librccl is loaded on first use (dlopen), one communicator per rank of this process
// (ncclCommInitAll), the gathers of all ranks issued inside one group call.
// ------------------------------------------------------------------------------------------
typedef ncclResult_t (*nccl_init_all_fn)(ncclComm_t *, int, const int *);
typedef ncclResult_t (*nccl_group_fn)(void);
typedef ncclResult_t (*nccl_allgather_fn)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef const char *(*nccl_err_fn)(ncclResult_t);

// The gathers run on private streams (one per rank) so that they overlap whatever the contexts enqueue next; each gather
// stream first waits for an event recorded on its rank's compute stream, so shards produced by the *_dev entry points on
// ecamd_multi_ctx(m, r)'s stream need no host synchronisation in between (producers on other streams: pass them in
// `producer_streams`, one hipStream_t per rank, or synchronise them first).
extern "C" int ecamd_multi_allgather_streams(ecamd_multi *m, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank,
					     void *const *producer_streams)
{
	if (!m || !d_send || !d_recv) {
		return mfail("ecamd_multi_allgather: NULL argument");
	}
	std::lock_guard<std::mutex> lk(m->mu);
	const int N = (int)m->ctx.size();
	if (N == 1) {
		// the producer is drained first: the copy below runs on the null stream, the contexts' streams are non-blocking ones
		hipStream_t prod = (producer_streams && producer_streams[0]) ? (hipStream_t)producer_streams[0] : (hipStream_t)ecamd_ctx_stream(m->ctx[0]);
		if (hipSetDevice(m->devices[0]) != hipSuccess || hipStreamSynchronize(prod) != hipSuccess ||
		    hipMemcpy(d_recv[0], d_send[0], bytes_per_rank, hipMemcpyDeviceToDevice) != hipSuccess) {
			return mfail("ecamd_multi_allgather: device copy failed");
		}
		return 0;
	}
	for (int r = 0; r < N; r++) {
		for (int q = 0; q < r; q++) {
			if (m->devices[(size_t)r] == m->devices[(size_t)q]) {
				return mfail("ecamd_multi_allgather: RCCL needs distinct devices (a device is listed twice in this multi-context)");
			}
		}
	}
	if (!m->rccl) {
		void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!h) {
			h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		}
		if (!h) {
			return mfail(std::string("ecamd_multi_allgather: cannot load librccl: ") + dlerror());
		}
		nccl_init_all_fn init_all = (nccl_init_all_fn)dlsym(h, "ncclCommInitAll");
		if (!init_all) {
			dlclose(h);
			return mfail("ecamd_multi_allgather: librccl has no ncclCommInitAll");
		}
		m->comms.assign((size_t)N, nullptr);
		const ncclResult_t e = init_all(m->comms.data(), N, m->devices.data());
		if (e != ncclSuccess) {
			nccl_err_fn es = (nccl_err_fn)dlsym(h, "ncclGetErrorString");
			const std::string msg = std::string("ecamd_multi_allgather: ncclCommInitAll: ") + (es ? es(e) : "error");
			m->comms.clear();
			dlclose(h);
			return mfail(msg);
		}
		m->cstreams.assign((size_t)N, nullptr);
		m->cready.assign((size_t)N, nullptr);
		for (int r = 0; r < N; r++) {
			if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess ||
			    hipStreamCreateWithFlags(&m->cstreams[(size_t)r], hipStreamNonBlocking) != hipSuccess ||
			    hipEventCreateWithFlags(&m->cready[(size_t)r], hipEventDisableTiming) != hipSuccess) {
				rccl_teardown(m, h);  // communicators, the streams and events made so far, the library handle: a retry starts afresh
				return mfail("ecamd_multi_allgather: stream / event creation failed");
			}
		}
		m->rccl = h;
	}
	nccl_group_fn gstart = (nccl_group_fn)dlsym(m->rccl, "ncclGroupStart"), gend = (nccl_group_fn)dlsym(m->rccl, "ncclGroupEnd");
	nccl_allgather_fn ag = (nccl_allgather_fn)dlsym(m->rccl, "ncclAllGather");
	nccl_err_fn es = (nccl_err_fn)dlsym(m->rccl, "ncclGetErrorString");
	if (!gstart || !gend || !ag) {
		return mfail("ecamd_multi_allgather: librccl misses ncclGroupStart / ncclGroupEnd / ncclAllGather");
	}
	// every gather stream waits for what its rank's producer stream holds at this point
	for (int r = 0; r < N; r++) {
		hipStream_t prod = (producer_streams && producer_streams[r]) ? (hipStream_t)producer_streams[r] : (hipStream_t)ecamd_ctx_stream(m->ctx[(size_t)r]);
		if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess || hipEventRecord(m->cready[(size_t)r], prod) != hipSuccess ||
		    hipStreamWaitEvent(m->cstreams[(size_t)r], m->cready[(size_t)r], 0) != hipSuccess) {
			return mfail("ecamd_multi_allgather: cannot order the gather behind the producer stream");
		}
	}
	ncclResult_t e = gstart();
	for (int r = 0; r < N && e == ncclSuccess; r++) {
		e = ag(d_send[r], d_recv[r], bytes_per_rank, ncclUint8, m->comms[(size_t)r], m->cstreams[(size_t)r]);
	}
	const ncclResult_t e2 = gend();
	if (e == ncclSuccess) {
		e = e2;
	}
	if (e != ncclSuccess) {
		return mfail(std::string("ecamd_multi_allgather: ") + (es ? es(e) : "RCCL error"));
	}
	for (int r = 0; r < N; r++) {
		if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess || hipStreamSynchronize(m->cstreams[(size_t)r]) != hipSuccess) {
			return mfail("ecamd_multi_allgather: stream synchronisation failed");
		}
	}
	return 0;
}

extern "C" int ecamd_multi_allgather(ecamd_multi *m, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank)
{
	return ecamd_multi_allgather_streams(m, d_send, d_recv, bytes_per_rank, nullptr);
}
