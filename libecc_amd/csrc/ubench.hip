// libecc_amd/csrc/ubench.hip -- instruction-rate micro-benchmark for the integer-MAD roofline.
//
// Measures, on the actual device, the issue rate of the instructions the scalar-multiplication
// path is made of ("Peak = measured on the device by a dependency-free
// v_mad_u64_u32 stream micro-benchmark").  Prints one JSON object.
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 8 independent accumulators, each iteration issues 8*4 = 32 instructions of the kind
template <int KIND> __global__ __launch_bounds__(256) void k_rate(u32 *out, u32 seed, int iters)
{
	u32 a = seed ^ threadIdx.x, b = seed * 2654435761u + blockIdx.x;
	u64 acc[8];
	u32 w[8];
	u64 mask = 0x5555aaaa5555aaaaull ^ seed;
	mask = ((u64)__builtin_amdgcn_readfirstlane((u32)(mask >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)mask);
	u64 cs[4] = {mask, ~mask, mask >> 1, mask << 1};
#pragma unroll
	for (int i = 0; i < 4; i++) {
		cs[i] = ((u64)__builtin_amdgcn_readfirstlane((u32)(cs[i] >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)cs[i]);
	}
#pragma unroll
	for (int i = 0; i < 8; i++) {
		acc[i] = (u64)(a + i) << 20;
		w[i] = b + i;
	}
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 0) {  // v_mad_u64_u32, independent
					asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(w[i]) : "vcc");
				} else if (KIND == 1) {  // v_add_u32 (full-rate baseline)
					asm volatile("v_add_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
				} else if (KIND == 2) {  // v_mul_lo_u32
					asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
				} else if (KIND == 3) {  // v_mul_hi_u32
					asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
				} else if (KIND == 4) {  // v_lshl_add_u64
					asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
				} else if (KIND == 5) {  // v_mad_u32_u24
					asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
				} else if (KIND == 6) {  // v_addc_co_u32 with SGPR-pair carry (VOP3)
					asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(w[i]) : : "vcc");
				} else if (KIND == 7) {  // v_cndmask_b32
					asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(a) : "vcc");
				} else if (KIND == 8) {  // v_mad_u64_u32 + v_addc pair (the MAC idiom), hazard-free spacing by construction
					asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(w[(i + 3) & 7]) : "vcc");
				} else if (KIND == 9) {  // v_alignbit_b32
					asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(w[i]) : "v"(a));
				} else if (KIND == 10) {  // v_cndmask_b32 with an SGPR-pair condition that nobody rewrites
					asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(a), "s"(mask));
				} else if (KIND == 11) {  // v_add_co_u32 (writes vcc), independent registers
					asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(w[i]) : "v"(a) : "vcc");
				} else if (KIND == 12) {  // v_addc_co_u32 e64 with distinct SGPR carry pairs (no vcc chain)
					asm volatile("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(w[i]), "+s"(cs[i & 3]));
				} else if (KIND == 13) {  // v_xor_b32 VOP2 e32
					asm volatile("v_xor_b32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
				} else if (KIND == 14) {  // v_add3_u32 (VOP3, 8-byte encoding)
					asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[i]) : "v"(a), "v"(b));
				}
			}
		}
	}
	u32 r = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		r ^= (u32)acc[i] ^ (u32)(acc[i] >> 32) ^ w[i];
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// one dependent chain: latency of a v_mad_u64_u32 feeding the next one's addend
__global__ __launch_bounds__(64) void k_dep(u32 *out, u32 seed, int iters)
{
	u32 a = seed ^ threadIdx.x, b = seed * 2654435761u;
	u64 acc = a;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 32; r++) {
			asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);
}

template <int KIND> static double run_rate(u32 *d_out, int blocks, int iters)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double lane_ops = (double)blocks * 256 * iters * 32.0;
	return lane_ops / (ms * 1e-3);  // lane-ops per second
}

int main(int argc, char **argv)
{
	int dev = 0;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
		fprintf(stderr, "no HIP device\n");
		return 1;
	}
	const int cus = prop.multiProcessorCount;
	const int blocks = cus * 8;  // 8 x 256 threads = 32 waves per CU (full occupancy)
	const int iters = (argc > 1) ? atoi(argv[1]) : 4000;
	u32 *d_out;
	hipMalloc(&d_out, (size_t)blocks * 256 * 4);
	const int NK = 15;
	const char *names[NK] = {"v_mad_u64_u32", "v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64",
				 "v_mad_u32_u24", "v_addc_co_u32_vccchain", "v_cndmask_b32_vcc", "v_mad_u64_u32_b", "v_alignbit_b32",
				 "v_cndmask_b32_sgpr", "v_add_co_u32", "v_addc_co_u32_sgprpairs", "v_xor_b32", "v_add3_u32"};
	double r[NK];
	r[0] = run_rate<0>(d_out, blocks, iters);
	r[1] = run_rate<1>(d_out, blocks, iters);
	r[2] = run_rate<2>(d_out, blocks, iters);
	r[3] = run_rate<3>(d_out, blocks, iters);
	r[4] = run_rate<4>(d_out, blocks, iters);
	r[5] = run_rate<5>(d_out, blocks, iters);
	r[6] = run_rate<6>(d_out, blocks, iters);
	r[7] = run_rate<7>(d_out, blocks, iters);
	r[8] = run_rate<8>(d_out, blocks, iters);
	r[9] = run_rate<9>(d_out, blocks, iters);
	r[10] = run_rate<10>(d_out, blocks, iters);
	r[11] = run_rate<11>(d_out, blocks, iters);
	r[12] = run_rate<12>(d_out, blocks, iters);
	r[13] = run_rate<13>(d_out, blocks, iters);
	r[14] = run_rate<14>(d_out, blocks, iters);
	// dependent chain, one wave per SIMD
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k_dep, dim3(cus * 4), dim3(64), 0, 0, d_out, 7u, 16);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_dep, dim3(cus * 4), dim3(64), 0, 0, d_out, 7u, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double clk_hz = (double)prop.clockRate * 1e3;
	const double dep_cycles = (ms * 1e-3) * clk_hz / ((double)iters * 32.0);
	printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"iters\": %d,\n", prop.name,
	       prop.gcnArchName, cus, clk_hz / 1e6, iters);
	for (int i = 0; i < NK; i++) {
		// cycles per wave64 instruction per SIMD at the nominal clock
		const double per_simd = r[i] / ((double)cus * 4.0);          // lane-ops/s per SIMD
		const double cyc = 64.0 * clk_hz / per_simd;
		printf(" \"%s\": {\"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd\": %.3f},\n", names[i], r[i], cyc);
	}
	printf(" \"v_mad_u64_u32_dependent_latency_cycles\": %.2f}\n", dep_cycles);
	hipFree(d_out);
	return 0;
}
