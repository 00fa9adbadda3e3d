// libecc_amd/csrc/ecamd_madchain.h -- chains of v_mad_u64_u32 as single inline-asm statements, shared by
// the secp256r1 field (ecamd_u29.h) and the generic radix-2^29 field (ecamd_u29g.h).
//
// hipcc pads every inline-asm statement that defines registers with an s_nop; with one statement per
// product that is one s_nop per v_mad_u64_u32, which a kernel running one wave per SIMD (the large
// fields: 256 VGPRs) cannot hide -- measured ~30 % of its time there.  ecamd_mad_chain<N, DUAL, YS> adds N
// products x[i] * y[i] to the 64-bit accumulator(s) in statements of up to twelve instructions (the operand limit of an asm statement is 30).
//   DUAL: products alternate between acc and acc2 (10 cycles result latency against 5.3 cycles issue: two
//         independent chains keep a lone wave issuing); the caller adds the two at the end of a column;
//   YS:   the second factors are wave-uniform and go in SGPRs (digits of p, reduction constants).
// The carry-out SGPR pair of the instruction is dead; it is an early-clobber output because a statement
// holds several instructions and the later ones still read their scalar inputs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ECAMD_CHAIN_FN __device__ __forceinline__
#else
#define ECAMD_CHAIN_FN inline
#endif

#if defined(__HIPCC__) && defined(U29_ASM_MAD)
// Z2: acc2 holds nothing yet -- its first product is written with a zero addend instead of being accumulated
template <int N, bool DUAL, bool YS, bool Z2 = false>
ECAMD_CHAIN_FN void ecamd_mad_chain(uint64_t &acc, uint64_t &acc2, const uint32_t *x, const uint32_t *y)
{
	uint64_t dead_;
	(void)acc2;
	if constexpr (N >= 12) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_mad_u64_u32 %1, %2, %25, %26, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_mad_u64_u32 %1, %2, %25, %26, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_mad_u64_u32 %1, %2, %25, %26, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0\n\tv_mad_u64_u32 %1, %2, %25, %26, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0\n\tv_mad_u64_u32 %0, %1, %24, %25, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]));
		}
		ecamd_mad_chain<N - 12, DUAL, YS, false>(acc, acc2, x + 12, y + 12);
	} else if constexpr (N == 11) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1\n\tv_mad_u64_u32 %0, %2, %23, %24, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]));
		}
	} else if constexpr (N == 10) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0\n\tv_mad_u64_u32 %1, %2, %21, %22, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]));
		}
	} else if constexpr (N == 9) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1\n\tv_mad_u64_u32 %0, %2, %19, %20, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]));
		}
	} else if constexpr (N == 8) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0\n\tv_mad_u64_u32 %1, %2, %17, %18, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]));
		}
	} else if constexpr (N == 7) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1\n\tv_mad_u64_u32 %0, %2, %15, %16, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]));
		}
	} else if constexpr (N == 6) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0\n\tv_mad_u64_u32 %1, %2, %13, %14, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]));
		}
	} else if constexpr (N == 5) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1\n\tv_mad_u64_u32 %0, %2, %11, %12, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]));
		}
	} else if constexpr (N == 4) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0\n\tv_mad_u64_u32 %1, %2, %9, %10, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]));
		}
	} else if constexpr (N == 3) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0\n\tv_mad_u64_u32 %0, %2, %7, %8, %0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1\n\tv_mad_u64_u32 %0, %2, %7, %8, %0"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]));
		}
	} else if constexpr (N == 2) {
		if constexpr (DUAL && YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]));
		} else if constexpr (DUAL && YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]));
		} else if constexpr (DUAL && !YS && Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, 0"
			    : "+v"(acc), "=&v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]));
		} else if constexpr (DUAL && !YS && !Z2) {
			asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_mad_u64_u32 %1, %2, %5, %6, %1"
			    : "+v"(acc), "+v"(acc2), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]));
		} else if constexpr (!DUAL && YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]));
		} else if constexpr (!DUAL && !YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]));
		}
	} else if constexpr (N == 1) {
		if constexpr (YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "s"(y[0]));
		} else if constexpr (!YS) {
			asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
			    : "+v"(acc), "=&s"(dead_)
			    : "v"(x[0]), "v"(y[0]));
		}
	}
}
// a chain of N <= 8 VGPR x VGPR products that STARTS an accumulator (first addend 0): no v_mov to clear the register pair
template <int N> ECAMD_CHAIN_FN void ecamd_mad_chain_z(uint64_t &acc, const uint32_t *x, const uint32_t *y)
{
	uint64_t dead_;
	static_assert(N >= 1 && N <= 8, "zero-start chains of one to eight products");
	if constexpr (N == 1) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]));
	} 	else if constexpr (N == 2) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]));
	} 	else if constexpr (N == 3) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]));
	} 	else if constexpr (N == 4) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]));
	} 	else if constexpr (N == 5) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]));
	} 	else if constexpr (N == 6) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]));
	} 	else if constexpr (N == 7) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]));
	} 	else if constexpr (N == 8) {
		asm("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
		    : "=&v"(acc), "=&s"(dead_)
		    : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]));
	}
}
#else
// host build (tests): -DECAMD_COUNT_MADS makes every MAD the field code would issue increment a counter, so that the work models of
// bench.py can be checked against the code (tests/test_u29g_host.py::test_mad_counts_match_the_work_model)
#ifdef ECAMD_COUNT_MADS
extern "C" uint64_t ecamd_mad_count;
#define ECAMD_COUNT_MAD(n) (ecamd_mad_count += (uint64_t)(n))
#else
#define ECAMD_COUNT_MAD(n) ((void)0)
#endif
template <int N, bool DUAL, bool YS, bool Z2 = false>
ECAMD_CHAIN_FN void ecamd_mad_chain(uint64_t &acc, uint64_t &acc2, const uint32_t *x, const uint32_t *y)
{
	ECAMD_COUNT_MAD(N);
	if (Z2 && DUAL && N > 1) {
		acc2 = 0;
	}
	for (int i = 0; i < N; i++) {
		if (DUAL && (i & 1)) {
			acc2 += (uint64_t)x[i] * y[i];
		} else {
			acc += (uint64_t)x[i] * y[i];
		}
	}
}
template <int N> ECAMD_CHAIN_FN void ecamd_mad_chain_z(uint64_t &acc, const uint32_t *x, const uint32_t *y)
{
	ECAMD_COUNT_MAD(N);
	acc = 0;
	for (int i = 0; i < N; i++) {
		acc += (uint64_t)x[i] * y[i];
	}
}
#endif

