// tools/msm_bucket_proto.hip -- MEASUREMENT PROTOTYPE (not part of the product): the data-movement phases of a bucket (Pippenger)
// evaluation of the Ed25519 batch equation, to decide whether building it pays (VERDICT round 2 item 6:
// "build it or commit the measured prototype that shows the sort / gather kills it").
//
// The equation has, per signature, the point A_i with a 253-bit scalar (16 windows of 16 bits) and R_i with a 128-bit one (8 windows):
// 24 n (bucket, point) pairs for n signatures.  A bucket evaluation needs, besides field arithmetic that can be counted exactly
// (24 n mixed additions of 7 M for the accumulation, 2 x 2^16 additions per window, 16 windows, for the bucket reduction):
//   1. the digits and a histogram of bucket sizes per window,
//   2. an exclusive scan per window,
//   3. the scatter of point indices into bucket order (a counting sort),
//   4. the GATHER of the points in that order -- one lane per bucket walking its segment, a 64-byte (affine x, y) or 128-byte
//      (Y-X, Y+X, 2dT with padding) record per pair, from a table of 2 n records.
// This program times 1-4 on uniformly random digits (the scalars z_i h_i mod q and z_i are uniform) and reports, per phase, the
// average of `reps` runs.  Phase 4 sums the limbs it loads (so that the loads are real) instead of adding points.
//   hipcc --offload-arch=gfx950 -O3 tools/msm_bucket_proto.hip -o gpurun_out/msm_bucket_proto && gpurun_out/msm_bucket_proto 20
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static const int WIN_A = 16, WIN_R = 8, NWIN = WIN_A, NB = 1 << 16;   // R's 8 windows share the buckets of A's low 8 windows

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
// digit of point `pt` (0..2n-1: even = A_i, odd = R_i) in window w; R has only WIN_R windows
__device__ __forceinline__ uint32_t digit_of(uint32_t pt, uint32_t w) { return mix(pt * 31u + w * 0x9e3779b9u) & 0xffffu; }

// pairs are enumerated window-major: pair p = (w, j) with j over the points that have window w
__global__ void k_hist(uint32_t *hist, uint32_t n)
{
	const uint32_t w = blockIdx.y;
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t cnt = (w < WIN_R) ? 2 * n : n;       // windows 0..7: A and R; 8..15: A only
	if (j >= cnt) {
		return;
	}
	const uint32_t pt = (w < WIN_R) ? j : 2 * j;
	atomicAdd(&hist[(size_t)w * NB + digit_of(pt, w)], 1u);
}

// exclusive scan of the 2^16 counters of one window by one block of 1024 threads (64 counters each)
__global__ void k_scan(const uint32_t *hist, uint32_t *start)
{
	__shared__ uint32_t part[1024];
	const uint32_t w = blockIdx.x, t = threadIdx.x;
	const uint32_t *h = hist + (size_t)w * NB + (size_t)t * 64;
	uint32_t s = 0;
	for (int k = 0; k < 64; k++) {
		s += h[k];
	}
	part[t] = s;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const uint32_t v = (t >= (uint32_t)d) ? part[t - d] : 0u;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint32_t run = part[t] - s;
	uint32_t *o = start + (size_t)w * NB + (size_t)t * 64;
	for (int k = 0; k < 64; k++) {
		o[k] = run;
		run += h[k];
	}
}

__global__ void k_scatter(const uint32_t *start, uint32_t *cursor, uint32_t *order, uint32_t n, size_t win_stride)
{
	const uint32_t w = blockIdx.y;
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t cnt = (w < WIN_R) ? 2 * n : n;
	if (j >= cnt) {
		return;
	}
	const uint32_t pt = (w < WIN_R) ? j : 2 * j;
	const uint32_t b = digit_of(pt, w);
	const uint32_t pos = start[(size_t)w * NB + b] + atomicAdd(&cursor[(size_t)w * NB + b], 1u);
	order[(size_t)w * win_stride + pos] = pt;
}

// one lane per bucket: walk the segment, load every point's record (REC16 x 16 bytes), fold it into 16 words
template <int REC16> __global__ void k_gather(const uint32_t *start, const uint32_t *hist, const uint32_t *order, const uint4 *table, uint4 *sums,
					      size_t win_stride)
{
	const uint32_t w = blockIdx.y;
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lo = start[(size_t)w * NB + b], cnt = hist[(size_t)w * NB + b];
	uint4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	for (uint32_t k = 0; k < cnt; k++) {
		const uint32_t pt = order[(size_t)w * win_stride + lo + k];
		const uint4 *rec = table + (size_t)pt * REC16;
#pragma unroll
		for (int q = 0; q < REC16; q++) {
			const uint4 v = rec[q];
			acc[q & 3].x += v.x; acc[q & 3].y ^= v.y; acc[q & 3].z += v.z; acc[q & 3].w ^= v.w;
		}
	}
	uint4 *o = sums + ((size_t)w * NB + b) * 4;
	for (int q = 0; q < 4; q++) {
		o[q] = acc[q];
	}
}

int main(int argc, char **argv)
{
	const int lg = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 5;
	const uint32_t n = 1u << lg;
	const size_t win_stride = (size_t)2 * n;
	uint32_t *hist, *start, *cursor, *order;
	uint4 *table, *sums;
	CHK(hipMalloc(&hist, (size_t)NWIN * NB * 4));
	CHK(hipMalloc(&start, (size_t)NWIN * NB * 4));
	CHK(hipMalloc(&cursor, (size_t)NWIN * NB * 4));
	CHK(hipMalloc(&order, (size_t)NWIN * win_stride * 4));
	CHK(hipMalloc(&table, (size_t)2 * n * 128));
	CHK(hipMalloc(&sums, (size_t)NWIN * NB * 64));
	CHK(hipMemset(table, 0x5a, (size_t)2 * n * 128));
	hipEvent_t ev[6];
	for (int i = 0; i < 6; i++) {
		CHK(hipEventCreate(&ev[i]));
	}
	double ms[5] = {0, 0, 0, 0, 0};
	const dim3 gp((2 * n + 255) / 256, NWIN), gb(NB / 256, NWIN);
	for (int r = 0; r <= reps; r++) {     // run 0 warms up
		CHK(hipMemset(hist, 0, (size_t)NWIN * NB * 4));
		CHK(hipMemset(cursor, 0, (size_t)NWIN * NB * 4));
		CHK(hipDeviceSynchronize());
		CHK(hipEventRecord(ev[0]));
		hipLaunchKernelGGL(k_hist, gp, dim3(256), 0, 0, hist, n);
		CHK(hipEventRecord(ev[1]));
		hipLaunchKernelGGL(k_scan, dim3(NWIN), dim3(1024), 0, 0, hist, start);
		CHK(hipEventRecord(ev[2]));
		hipLaunchKernelGGL(k_scatter, gp, dim3(256), 0, 0, start, cursor, order, n, win_stride);
		CHK(hipEventRecord(ev[3]));
		hipLaunchKernelGGL(k_gather<4>, gb, dim3(256), 0, 0, start, hist, order, table, sums, win_stride);
		CHK(hipEventRecord(ev[4]));
		hipLaunchKernelGGL(k_gather<8>, gb, dim3(256), 0, 0, start, hist, order, table, sums, win_stride);
		CHK(hipEventRecord(ev[5]));
		CHK(hipEventSynchronize(ev[5]));
		if (r) {
			for (int i = 0; i < 5; i++) {
				float f = 0;
				CHK(hipEventElapsedTime(&f, ev[i], ev[i + 1]));
				ms[i] += f / reps;
			}
		}
	}
	const double pairs = (double)n * (2.0 * WIN_R + (WIN_A - WIN_R));
	printf("{\"signatures\": %u, \"pairs\": %.0f, \"buckets_per_window\": %d, \"windows\": %d, \"mean_bucket_size\": %.1f,\n", n, pairs, NB, NWIN,
	       pairs / ((double)NWIN * NB));
	printf(" \"ms\": {\"histogram\": %.3f, \"scan\": %.3f, \"scatter\": %.3f, \"gather_64B_records\": %.3f, \"gather_128B_records\": %.3f},\n", ms[0], ms[1], ms[2],
	       ms[3], ms[4]);
	printf(" \"sort_plus_gather_ms\": {\"64B\": %.3f, \"128B\": %.3f},\n", ms[0] + ms[1] + ms[2] + ms[3], ms[0] + ms[1] + ms[2] + ms[4]);
	printf(" \"gather_GBps\": {\"64B\": %.1f, \"128B\": %.1f}}\n", pairs * 64.0 / (ms[3] * 1e6), pairs * 128.0 / (ms[4] * 1e6));
	return 0;
}
