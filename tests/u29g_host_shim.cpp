// tests/u29g_host_shim.cpp -- TEST INFRASTRUCTURE: host build of the generic radix-2^29 headers
// (ecamd_u29g.h, ecamd_jacg.h) for tests/test_u29g_host.py.  One set of entry points per
// field size; the curve constants (CurveG image) are supplied by the test as a flat u32 array.
#include <cstring>
#include <cstdint>
#ifdef ECAMD_COUNT_MADS
extern "C" { uint64_t ecamd_mad_count = 0; }
#endif
#include "../libecc_amd/csrc/ecamd_jacg.h"
#if defined(G29_P25519) || defined(G29_P448)
#include "../libecc_amd/csrc/ecamd_rcbg.h"   /* the two units whose EdDSA tail uses it */
#define SHIM_RCB 1
#endif
#ifdef ECAMD_COUNT_MADS
extern "C" void g_madcount_reset(void) { ecamd_mad_count = 0; }
extern "C" uint64_t g_madcount_get(void) { return ecamd_mad_count; }
#endif

using namespace jacg;

template <int PB> struct Shim {
	typedef Cfg<PB> C;
	static constexpr int NL = C::NL;
	typedef CurveG<NL> CK;
	static void mul_(const uint32_t *k, const uint32_t *a, const uint32_t *b, uint32_t *out, int sq)
	{
		const CK &K = *(const CK *)k;
		typename Cls<PB>::FA x, y;
		memcpy(x.l, a, 4 * NL);
		memcpy(y.l, b, 4 * NL);
		if (sq) {
			auto r = sqr(x, K);
			memcpy(out, r.l, 4 * NL);
		} else {
			auto r = mul(x, y, K);
			memcpy(out, r.l, 4 * NL);
		}
	}
	// a * 121665 (2^255 - 19 flavour) / a * 39081 (Goldilocks flavour): the a24 of the x-only ladders as one word (mul_word)
	static void mulword_(const uint32_t *a, uint32_t *out)
	{
#if defined(G29_P25519) || defined(G29_P448)
		// the loosest operand class the ladders hand over: a carried difference
		E<PB, (1ull << 32) - 1, g29::P448 ? ((1ull << 29) - 1) : ((1ull << 32) - 1), 16> x;
		memcpy(x.l, a, 4 * NL);
		auto r = mul_word<g29::P448 ? 39081u : 121665u>(x);
		memcpy(out, r.l, 4 * NL);
#else
		(void)a;
		(void)out;
#endif
	}
	static void dbl_(const uint32_t *k, const uint32_t *p, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		Jac<PB> P;
		memcpy(P.X.l, p, 4 * NL);
		memcpy(P.Y.l, p + NL, 4 * NL);
		memcpy(P.Z.l, p + 2 * NL, 4 * NL);
		Jac<PB> R = dbl(P, K);
		memcpy(out, R.X.l, 4 * NL);
		memcpy(out + NL, R.Y.l, 4 * NL);
		memcpy(out + 2 * NL, R.Z.l, 4 * NL);
	}
	static int add_(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		Jac<PB> P;
		typename Cls<PB>::FA X2, Y2, Z2;
		memcpy(P.X.l, p, 4 * NL);
		memcpy(P.Y.l, p + NL, 4 * NL);
		memcpy(P.Z.l, p + 2 * NL, 4 * NL);
		memcpy(X2.l, q, 4 * NL);
		memcpy(Y2.l, q + NL, 4 * NL);
		memcpy(Z2.l, q + 2 * NL, 4 * NL);
		bool hz;
		Jac<PB> R = add_jac(P, X2, Y2, Z2, hz, K);
		memcpy(out, R.X.l, 4 * NL);
		memcpy(out + NL, R.Y.l, 4 * NL);
		memcpy(out + 2 * NL, R.Z.l, 4 * NL);
		return hz ? 1 : 0;
	}
#if !defined(G29_K256)
	// mixed addition / doubling on the tight accumulator class JacT (the window loop of the affine-table kernels)
	static int madd_(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		JacT<PB> P;
		typename Cls<PB>::FA X2, Y2;
		memcpy(P.X.l, p, 4 * NL);
		memcpy(P.Y.l, p + NL, 4 * NL);
		memcpy(P.Z.l, p + 2 * NL, 4 * NL);
		memcpy(X2.l, q, 4 * NL);
		memcpy(Y2.l, q + NL, 4 * NL);
		JacT<PB> R = madd_jac(P, X2, Y2, K);
		memcpy(out, R.X.l, 4 * NL);
		memcpy(out + NL, R.Y.l, 4 * NL);
		memcpy(out + 2 * NL, R.Z.l, 4 * NL);
		return 0;
	}
	static void dblt_(const uint32_t *k, const uint32_t *p, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		JacT<PB> P;
		memcpy(P.X.l, p, 4 * NL);
		memcpy(P.Y.l, p + NL, 4 * NL);
		memcpy(P.Z.l, p + 2 * NL, 4 * NL);
		JacT<PB> R = dbl(P, K);
		memcpy(out, R.X.l, 4 * NL);
		memcpy(out + NL, R.Y.l, 4 * NL);
		memcpy(out + 2 * NL, R.Z.l, 4 * NL);
	}
	static void infot_(uint32_t *out)
	{
		out[0] = (uint32_t)ClsT<PB>::VT;
		out[1] = (uint32_t)ClsT<PB>::FT::LB;
		out[2] = (uint32_t)ClsT<PB>::FT::TB;
	}
#else
	// secp256k1's flavour keeps the Jacobian-table kernel (its products leave no room for the bias of a subtraction from
	// the accumulator itself)
	static int madd_(const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *) { return -1; }
	static void dblt_(const uint32_t *, const uint32_t *, uint32_t *) {}
	static void infot_(uint32_t *out) { out[0] = 0; }
#endif
	static void neg_(const uint32_t *k, const uint32_t *a, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		typename Cls<PB>::FM x;
		memcpy(x.l, a, 4 * NL);
		auto r = neg<PB>(x, K);
		memcpy(out, r.l, 4 * NL);
	}
	static void inv_(const uint32_t *k, const uint32_t *a, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		typename Cls<PB>::FM x;
		memcpy(x.l, a, 4 * NL);
		auto r = inv<PB>(x, K);
		memcpy(out, r.l, 4 * NL);
	}
	static void canon_(const uint32_t *k, const uint32_t *a, uint32_t *digits)
	{
		const CK &K = *(const CK *)k;
		typename Cls<PB>::FM x;
		memcpy(x.l, a, 4 * NL);
		canonical_digits(digits, x, K);
	}
#ifdef SHIM_RCB
	// complete (RCB) addition / doubling of ecamd_rcbg.h: p, q, out = X || Y || Z (FA class); dbl: q ignored.  Returns Y == Z == 0
	static int rcb_(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *out, int dbl)
	{
		const CK &K = *(const CK *)k;
		rcbg::PtG<PB> P, Q;
		memcpy(P.X.l, p, 4 * NL);
		memcpy(P.Y.l, p + NL, 4 * NL);
		memcpy(P.Z.l, p + 2 * NL, 4 * NL);
		memcpy(Q.X.l, q, 4 * NL);
		memcpy(Q.Y.l, q + NL, 4 * NL);
		memcpy(Q.Z.l, q + 2 * NL, 4 * NL);
		const rcbg::PtG<PB> R = dbl ? rcbg::dbl_rcb<PB>(P, K) : rcbg::add_rcb<PB>(P, Q, K);
		memcpy(out, R.X.l, 4 * NL);
		memcpy(out + NL, R.Y.l, 4 * NL);
		memcpy(out + 2 * NL, R.Z.l, 4 * NL);
		return (rcbg::coord_is_zero<PB>(R.Y, K) ? 1 : 0) | (rcbg::coord_is_zero<PB>(R.Z, K) ? 2 : 0);
	}
#else
	static int rcb_(const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *, int) { return -1; }
#endif
	// lift_x_even: xw = the abscissa as NW saturated little-endian words; out = xo || yo (multiplication-result class); returns ok
	static int liftx_(const uint32_t *k, const uint32_t *xw, uint32_t *out)
	{
		const CK &K = *(const CK *)k;
		constexpr int NW = (PB + 31) / 32;
		const auto xd = from_words<PB, NW>(xw);
		typename Cls<PB>::FM xo, yo;
		const bool ok = lift_x_even<PB>(xd, K, xo, yo);
		memcpy(out, xo.l, 4 * NL);
		memcpy(out + NL, yo.l, 4 * NL);
		return ok ? 1 : 0;
	}
	static void info_(uint32_t *out)
	{
		out[0] = NL;
		out[1] = (uint32_t)C::HEAD;
		out[2] = (uint32_t)(int32_t)C::TOPSH;
		out[3] = (uint32_t)Cls<PB>::LC;
		out[4] = (uint32_t)(Cls<PB>::VA & 0xffffffffu);
		out[5] = (uint32_t)sizeof(CK);
		out[6] = (uint32_t)Cls<PB>::FA::LB;
		out[7] = (uint32_t)Cls<PB>::FA::TB;
	}
};

#define SHIM(PB) \
	extern "C" { \
	void g_mul_##PB(const uint32_t *k, const uint32_t *a, const uint32_t *b, uint32_t *o, int sq) { Shim<PB>::mul_(k, a, b, o, sq); } \
	void g_dbl_##PB(const uint32_t *k, const uint32_t *p, uint32_t *o) { Shim<PB>::dbl_(k, p, o); } \
	void g_mulword_##PB(const uint32_t *a, uint32_t *o) { Shim<PB>::mulword_(a, o); } \
	int g_add_##PB(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *o) { return Shim<PB>::add_(k, p, q, o); } \
	int g_madd_##PB(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *o) { return Shim<PB>::madd_(k, p, q, o); } \
	void g_dblt_##PB(const uint32_t *k, const uint32_t *p, uint32_t *o) { Shim<PB>::dblt_(k, p, o); } \
	void g_infot_##PB(uint32_t *o) { Shim<PB>::infot_(o); } \
	void g_neg_##PB(const uint32_t *k, const uint32_t *a, uint32_t *o) { Shim<PB>::neg_(k, a, o); } \
	void g_inv_##PB(const uint32_t *k, const uint32_t *a, uint32_t *o) { Shim<PB>::inv_(k, a, o); } \
	void g_canon_##PB(const uint32_t *k, const uint32_t *a, uint32_t *o) { Shim<PB>::canon_(k, a, o); } \
	void g_info_##PB(uint32_t *o) { Shim<PB>::info_(o); } \
	int g_liftx_##PB(const uint32_t *k, const uint32_t *xw, uint32_t *o) { return Shim<PB>::liftx_(k, xw, o); } \
	int g_rcb_##PB(const uint32_t *k, const uint32_t *p, const uint32_t *q, uint32_t *o, int dbl) { return Shim<PB>::rcb_(k, p, q, o, dbl); } \
	}
#if defined(SHIM_ONLY_255)
SHIM(255)
#elif defined(SHIM_ONLY_384)
SHIM(384)
#elif defined(SHIM_ONLY_256)
SHIM(256)
#elif defined(SHIM_ONLY_448)
SHIM(448)
#elif defined(SHIM_ONLY_224)
SHIM(224)
#elif defined(SHIM_ONLY_192)
SHIM(192)
#elif !defined(SHIM_ONLY_521)
SHIM(192)
SHIM(224)
SHIM(255)
SHIM(256)
SHIM(320)
SHIM(384)
SHIM(448)
SHIM(511)
SHIM(512)
#endif
#if !defined(SHIM_ONLY_255) && !defined(SHIM_ONLY_384) && !defined(SHIM_ONLY_256) && !defined(SHIM_ONLY_448) && !defined(SHIM_ONLY_224) && \
	!defined(SHIM_ONLY_192)
SHIM(521)
#endif
