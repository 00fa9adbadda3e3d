// tests/u29_host_shim.cpp -- TEST INFRASTRUCTURE: compiles the product's radix-2^29 headers for
// the host (plain C++17, no HIP) so tests/test_u29_host.py can drive the exact same template code
// (including its compile-time bound checks) against Python integers without a GPU.
#include <cstring>
#define U29_INLINE_MUL 1
#include "../libecc_amd/csrc/ecamd_p256.h"

using namespace p256;

extern "C" {
// a, b: 9 limbs each (must respect Fmul bounds); out: 9 limbs
void t_mul(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	Fmul x, y;
	memcpy(x.l, a, 36);
	memcpy(y.l, b, 36);
	auto r = mul(x, y);
	memcpy(out, r.l, 36);
}
void t_sqr(const uint32_t *a, uint32_t *out)
{
	Fmul x;
	memcpy(x.l, a, 36);
	auto r = sqr(x);
	memcpy(out, r.l, 36);
}
// loose operands: limbs up to 2^30.3, value up to ~6p (class used for worst-case tests)
void t_mul_loose(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	F<(1ull << 30), (6ull << 24), 96> x, y;
	memcpy(x.l, a, 36);
	memcpy(y.l, b, 36);
	auto r = mul(x, y);
	memcpy(out, r.l, 36);
}
void t_fold(const uint32_t *a, uint32_t *out)
{
	F<0xfffffffeull, (6ull << 24), 100> x;
	memcpy(x.l, a, 36);
	auto r = fold(x);
	memcpy(out, r.l, 36);
}
void t_inv(const uint32_t *a, uint32_t *out)
{
	Fmul x;
	memcpy(x.l, a, 36);
	auto r = inv(x);
	memcpy(out, r.l, 36);
}
void t_canonical_words(const uint32_t *a, uint32_t *words)
{
	Fmul x;
	memcpy(x.l, a, 36);
	Fcanon c = canonical(x);
	to_words(words, c);
}
void t_from_words(const uint32_t *words, uint32_t *out)
{
	Fcanon c = from_words(words);
	memcpy(out, c.l, 36);
}
int t_is_zero(const uint32_t *a)
{
	F<MASK, (4ull << 24), 64> x;
	memcpy(x.l, a, 36);
	return is_zero_mulout(x) ? 1 : 0;
}
// Jacobian ops on loop-carried classes: in/out 27 limbs (X, Y, Z)
void t_dbl(const uint32_t *p, uint32_t *out)
{
	Jac P;
	memcpy(P.X.l, p, 36);
	memcpy(P.Y.l, p + 9, 36);
	memcpy(P.Z.l, p + 18, 36);
	Jac R = dbl(P);
	memcpy(out, R.X.l, 36);
	memcpy(out + 9, R.Y.l, 36);
	memcpy(out + 18, R.Z.l, 36);
}
int t_add(const uint32_t *p, const uint32_t *q, uint32_t *out)
{
	Jac P;
	memcpy(P.X.l, p, 36);
	memcpy(P.Y.l, p + 9, 36);
	memcpy(P.Z.l, p + 18, 36);
	FX X2;
	FZ Z2;
	FYsel Y2;
	memcpy(X2.l, q, 36);
	memcpy(Y2.l, q + 9, 36);
	memcpy(Z2.l, q + 18, 36);
	bool hz;
	Jac R = add_jac(P, X2, Y2, Z2, hz);
	memcpy(out, R.X.l, 36);
	memcpy(out + 9, R.Y.l, 36);
	memcpy(out + 18, R.Z.l, 36);
	return hz ? 1 : 0;
}
int t_madd(const uint32_t *p, const uint32_t *q, uint32_t *out, int negate)
{
	Jac P;
	memcpy(P.X.l, p, 36);
	memcpy(P.Y.l, p + 9, 36);
	memcpy(P.Z.l, p + 18, 36);
	Fcanon X2, Y2;
	memcpy(X2.l, q, 36);
	memcpy(Y2.l, q + 9, 36);
	bool hz;
	Jac R = negate ? madd(P, X2, neg_aff(Y2), hz) : madd(P, X2, weaken<FYaff>(Y2), hz);
	memcpy(out, R.X.l, 36);
	memcpy(out + 9, R.Y.l, 36);
	memcpy(out + 18, R.Z.l, 36);
	return hz ? 1 : 0;
}
void t_consts(uint32_t *out)  // P, D, R2, ONE, BM (9 each), Q3 Q6 Q7 Q8
{
	memcpy(out, u29::P256::P, 36);
	memcpy(out + 9, u29::P256::D, 36);
	memcpy(out + 18, K::R2, 36);
	memcpy(out + 27, K::ONE, 36);
	memcpy(out + 36, K::BM, 36);
	out[45] = u29::P256::Q3;
	out[46] = u29::P256::Q6;
	out[47] = u29::P256::Q7;
	out[48] = u29::P256::Q8;
}
}
