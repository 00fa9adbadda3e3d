// libecc_amd/csrc/ecamd_hash.hip -- SHA-224 / SHA-256 / SHA-384 / SHA-512 (FIPS 180-4) of a batch of SHORT messages, one message
// per lane, for the protocol entry points that take messages instead of digests (ec_ecdsa_verify_msg_batch_fmt,
// ec_eddsa_verify_msg_batch: round 4).
//
// Why it is here although hashes are not on the data-parallel path of SURVEY.md section 8: measured end to end (profiles/
// r3c_compat_end_to_end.md), the libecc-typed ec_verify_batch spends more host time hashing the messages through libecc's
// portable hash_maps[] code (326 ns per 32-byte message with SHA-256, about 500 ns with SHA-384 / SHA-512, on one host thread) than
// on everything else it does per item, and that is what separates the 30 M verifications/s the layer delivers from the 67 M/s of
// the device-resident kernels.  On the GPU a short message is one or two compression functions in one lane: a million of them cost
// less than 0.1 ms.  The digests are the standard ones (tests/test_gpu_hash.py against hashlib on every length around the block
// boundaries); libecc's hfunc_* produce the same bytes for these four functions, so what the verification sees does not change.
//
// Layout: message i sits in a SLOT of `stride` bytes (a multiple of 4): a little-endian u32 length followed by the bytes.  The
// slots have one stride per call, so the host staging pipeline of ecamd_host.cpp moves them like any other fixed-stride array;
// the caller (libecc_amd_compat.c) takes this path when every message of a batch fits 4 + len <= stride <= 256 and hashes on the
// host otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ecamd_internal.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

__constant__ u32 c_k256[64] = {
	0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
	0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
	0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
	0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
	0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
	0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__constant__ u64 c_k512[80] = {
	0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
	0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
	0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
	0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
	0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
	0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
	0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
	0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
	0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
	0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
	0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
	0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
	0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
	0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

static __device__ __forceinline__ u32 ror32(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, (u32)r); }
static __device__ __forceinline__ u64 ror64(u64 x, int r) { return (x >> r) | (x << (64 - r)); }

// big-endian word j of the padded message: the bytes, then 0x80, zeros, and the bit length in the last 8 (16) bytes of the last block
static __device__ __forceinline__ u32 padded_word(const u32 *msg, u32 len, u32 j, u32 last_word)
{
	const u32 pos = 4 * j;
	u32 w = 0;
	if (pos < len) {
		w = __builtin_bswap32(msg[j]);
		const u32 rem = len - pos;              // bytes of this word that belong to the message
		if (rem < 4) {
			const u32 keep = 0xffffffffu << (8 * (4 - rem));
			w = (w & keep) | (0x80u << (8 * (3 - rem)));
		}
	} else if (pos == len) {
		w = 0x80000000u;
	}
	if (j == last_word) {
		w = len << 3;                           // (messages of a slot are far below 2^29 bytes: the length fits the last word)
	}
	return w;
}

// ALG: 224, 256, 384, 512
template <int ALG> __global__ __launch_bounds__(64) void k_sha2_slots(const u8 *slots, u32 stride, u32 n, u8 *out, u32 out_stride)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	const u32 *slot = (const u32 *)(slots + (size_t)i * stride);
	u32 len = slot[0];
	len = len > stride - 4 ? stride - 4 : len;      // a length the slot cannot hold is the caller's error; stay inside the buffer
	const u32 *msg = slot + 1;
	u8 *dst = out + (size_t)i * out_stride;
	if constexpr (ALG == 224 || ALG == 256) {
		u32 h[8];
		if (ALG == 256) {
			h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
		} else {
			h[0] = 0xc1059ed8; h[1] = 0x367cd507; h[2] = 0x3070dd17; h[3] = 0xf70e5939; h[4] = 0xffc00b31; h[5] = 0x68581511; h[6] = 0x64f98fa7; h[7] = 0xbefa4fa4;
		}
		const u32 nblocks = (len + 9 + 63) / 64;
		const u32 last_word = 16 * nblocks - 1;
#pragma unroll 1
		for (u32 b = 0; b < nblocks; b++) {
			u32 w[16];
#pragma unroll
			for (int t = 0; t < 16; t++) {
				w[t] = padded_word(msg, len, 16 * b + t, last_word);
			}
			u32 a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
			for (int r = 0; r < 64; r += 16) {
#pragma unroll
				for (int t = 0; t < 16; t++) {
					if (r > 0) {
						const u32 w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
						const u32 s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
						const u32 s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
						w[t] = w[t] + s0 + w[(t + 9) & 15] + s1;
					}
					const u32 S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
					const u32 ch = (e & f) ^ (~e & g);
					const u32 t1 = hh + S1 + ch + c_k256[r + t] + w[t];
					const u32 S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
					const u32 mj = (a & bb) ^ (a & c) ^ (bb & c);
					const u32 t2 = S0 + mj;
					hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
				}
			}
			h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
		}
		constexpr int OUTW = (ALG == 256) ? 8 : 7;
#pragma unroll
		for (int k = 0; k < OUTW; k++) {
			dst[4 * k] = (u8)(h[k] >> 24);
			dst[4 * k + 1] = (u8)(h[k] >> 16);
			dst[4 * k + 2] = (u8)(h[k] >> 8);
			dst[4 * k + 3] = (u8)h[k];
		}
	} else {
		u64 h[8];
		if (ALG == 512) {
			h[0] = 0x6a09e667f3bcc908ull; h[1] = 0xbb67ae8584caa73bull; h[2] = 0x3c6ef372fe94f82bull; h[3] = 0xa54ff53a5f1d36f1ull;
			h[4] = 0x510e527fade682d1ull; h[5] = 0x9b05688c2b3e6c1full; h[6] = 0x1f83d9abfb41bd6bull; h[7] = 0x5be0cd19137e2179ull;
		} else {
			h[0] = 0xcbbb9d5dc1059ed8ull; h[1] = 0x629a292a367cd507ull; h[2] = 0x9159015a3070dd17ull; h[3] = 0x152fecd8f70e5939ull;
			h[4] = 0x67332667ffc00b31ull; h[5] = 0x8eb44a8768581511ull; h[6] = 0xdb0c2e0d64f98fa7ull; h[7] = 0x47b5481dbefa4fa4ull;
		}
		const u32 nblocks = (len + 17 + 127) / 128;
		const u32 last_word = 32 * nblocks - 1;
#pragma unroll 1
		for (u32 b = 0; b < nblocks; b++) {
			u64 w[16];
#pragma unroll
			for (int t = 0; t < 16; t++) {
				w[t] = ((u64)padded_word(msg, len, 32 * b + 2 * t, last_word) << 32) | padded_word(msg, len, 32 * b + 2 * t + 1, last_word);
			}
			u64 a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
			for (int r = 0; r < 80; r += 16) {
#pragma unroll
				for (int t = 0; t < 16; t++) {
					if (r > 0) {
						const u64 w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
						const u64 s0 = ror64(w15, 1) ^ ror64(w15, 8) ^ (w15 >> 7);
						const u64 s1 = ror64(w2, 19) ^ ror64(w2, 61) ^ (w2 >> 6);
						w[t] = w[t] + s0 + w[(t + 9) & 15] + s1;
					}
					const u64 S1 = ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41);
					const u64 ch = (e & f) ^ (~e & g);
					const u64 t1 = hh + S1 + ch + c_k512[r + t] + w[t];
					const u64 S0 = ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39);
					const u64 mj = (a & bb) ^ (a & c) ^ (bb & c);
					const u64 t2 = S0 + mj;
					hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
				}
			}
			h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
		}
		constexpr int OUTW = (ALG == 512) ? 8 : 6;
#pragma unroll
		for (int k = 0; k < OUTW; k++) {
#pragma unroll
			for (int bq = 0; bq < 8; bq++) {
				dst[8 * k + bq] = (u8)(h[k] >> (56 - 8 * bq));
			}
		}
	}
}

// hash_type: libecc's hash_alg_type numbering (hash/hash_algs.h): SHA224 = 1, SHA256 = 2, SHA384 = 3, SHA512 = 4
int ecamd_sha2_digest_len(int hash_type)
{
	switch (hash_type) {
	case 1: return 28;
	case 2: return 32;
	case 3: return 48;
	case 4: return 64;
	default: return 0;
	}
}

hipError_t ecamd_launch_sha2_slots(int hash_type, const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	if (stride < 4 || (stride & 3u)) {
		return hipErrorInvalidValue;
	}
	const dim3 grid((n + 63) / 64), block(64);
	switch (hash_type) {
	case 1: hipLaunchKernelGGL(k_sha2_slots<224>, grid, block, 0, s, slots, stride, n, out, out_stride); break;
	case 2: hipLaunchKernelGGL(k_sha2_slots<256>, grid, block, 0, s, slots, stride, n, out, out_stride); break;
	case 3: hipLaunchKernelGGL(k_sha2_slots<384>, grid, block, 0, s, slots, stride, n, out, out_stride); break;
	case 4: hipLaunchKernelGGL(k_sha2_slots<512>, grid, block, 0, s, slots, stride, n, out, out_stride); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ---- SHAKE256 (FIPS 202) of the same slots, for the Ed448 forms of the message-taking verification (round 4): Keccak-f[1600] on 25
//      64-bit lanes per thread, rate 136 octets, domain separation 0x1F, the first outlen <= 136 octets of the output (libecc's hash
//      type SHAKE256 is that function with 114 octets; EDDSA448PH's pre-hash takes its first 64, sig/eddsa.c:1650-1657) ----
__constant__ u64 c_keccak_rc[24] = {
	0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
	0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
	0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
	0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

static __device__ __forceinline__ u64 rol64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

static __device__ __forceinline__ void keccak_f1600(u64 *st)
{
	constexpr int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
	constexpr int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#pragma unroll 1
	for (int round = 0; round < 24; round++) {
		u64 c[5];
#pragma unroll
		for (int x = 0; x < 5; x++) {
			c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
		}
#pragma unroll
		for (int x = 0; x < 5; x++) {
			const u64 d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
			for (int y = 0; y < 25; y += 5) {
				st[y + x] ^= d;
			}
		}
		u64 cur = st[1];
#pragma unroll
		for (int t = 0; t < 24; t++) {
			const u64 tmp = st[piln[t]];
			st[piln[t]] = rol64(cur, rotc[t]);
			cur = tmp;
		}
#pragma unroll
		for (int y = 0; y < 25; y += 5) {
			u64 row[5];
#pragma unroll
			for (int x = 0; x < 5; x++) {
				row[x] = st[y + x];
			}
#pragma unroll
			for (int x = 0; x < 5; x++) {
				st[y + x] = row[x] ^ (~row[(x + 1) % 5] & row[(x + 2) % 5]);
			}
		}
		st[0] ^= c_keccak_rc[round];
	}
}

__global__ __launch_bounds__(64) void k_shake256_slots(const u8 *slots, u32 stride, u32 n, u8 *out, u32 out_stride, u32 outlen)
{
	const u32 i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) {
		return;
	}
	const u8 *slot = slots + (size_t)i * stride;
	u32 len = *(const u32 *)slot;
	if (len > stride - 4) {
		len = stride - 4;   // (a malformed length cannot read outside the slot)
	}
	const u8 *msg = slot + 4;
	constexpr u32 RATE = 136;
	const u32 nblocks = len / RATE + 1, total = nblocks * RATE;
	u64 st[25];
#pragma unroll
	for (int k = 0; k < 25; k++) {
		st[k] = 0;
	}
#pragma unroll 1
	for (u32 b = 0; b < nblocks; b++) {
#pragma unroll
		for (u32 w = 0; w < RATE / 8; w++) {
			u64 v = 0;
#pragma unroll
			for (u32 k = 0; k < 8; k++) {
				const u32 pos = b * RATE + 8 * w + k;
				u32 byte = pos < len ? (u32)msg[pos] : (pos == len ? 0x1fu : 0u);
				byte ^= (pos == total - 1) ? 0x80u : 0u;
				v |= (u64)byte << (8 * k);
			}
			st[w] ^= v;
		}
		keccak_f1600(st);
	}
	u8 *dst = out + (size_t)i * out_stride;
#pragma unroll
	for (u32 w = 0; w < RATE / 8; w++) {
#pragma unroll
		for (u32 k = 0; k < 8; k++) {
			if (8 * w + k < outlen) {
				dst[8 * w + k] = (u8)(st[w] >> (8 * k));
			}
		}
	}
}

hipError_t ecamd_launch_shake256_slots(const uint8_t *slots, uint32_t stride, uint32_t n, uint8_t *out, uint32_t out_stride, uint32_t outlen, hipStream_t s)
{
	if (n == 0) {
		return hipSuccess;
	}
	if (stride < 4 || (stride & 3u) || outlen == 0 || outlen > 136 || out_stride < outlen) {
		return hipErrorInvalidValue;
	}
	hipLaunchKernelGGL(k_shake256_slots, dim3((n + 63) / 64), dim3(64), 0, s, slots, stride, n, out, out_stride, outlen);
	return hipGetLastError();
}

// ---- two byte movers of ec_eddsa_verify_msg_prj_batch: the key's encoding into its place in the hash input, and "an item whose key did not
//      import is rejected" ----
__global__ __launch_bounds__(256) void k_slot_patch(u8 *slots, u32 stride, u32 off, const u8 *src, u32 len, const u8 *skip, u32 n)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n || (skip && skip[i])) {
		return;
	}
	u8 *d = slots + (size_t)i * stride + 4 + off;
	const u8 *v = src + (size_t)i * len;
	for (u32 b = 0; b < len; b++) {
		d[b] = v[b];
	}
}
__global__ __launch_bounds__(256) void k_reject_where(u8 *result, const u8 *status, u32 n)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i < n && status[i]) {
		result[i] = 1;
	}
}
hipError_t ecamd_launch_slot_patch(uint8_t *slots, uint32_t stride, uint32_t off, const uint8_t *src, uint32_t len, const uint8_t *skip, uint32_t n,
				   hipStream_t s)
{
	if (n) {
		hipLaunchKernelGGL(k_slot_patch, dim3((n + 255) / 256), dim3(256), 0, s, slots, stride, off, src, len, skip, n);
	}
	return hipGetLastError();
}
// ---- the byte mover in front of the Schnorr-type multi-scalar form (EcamdSchnorrPrepArgs) ----
__global__ __launch_bounds__(256) void k_schnorr_prep(EcamdSchnorrPrepArgs A)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= A.n) {
		return;
	}
	const u32 cl = A.clen, ql = A.qlen, rl = A.rlen;
	const u8 *key = A.keys_aff + (size_t)i * 2 * cl, *sig = A.sigs + (size_t)i * (rl + ql);
	u8 *ko = A.keys_out + (size_t)i * 2 * cl;
	if (A.kst && A.kst[i] != 0) {
		atomicOr(A.flag, 1u);   // the reference fails on such a key (prj_pt_unique of a point at infinity, an import error): not decided here
	}
	if (A.x_off != 0xffffffffu) {
		u8 *d = A.slots + (size_t)i * A.stride + 4 + A.x_off;
		for (u32 b = 0; b < cl; b++) {
			d[b] = key[b];
		}
	}
	for (u32 b = 0; b < cl; b++) {
		ko[b] = key[b];
	}
	if (A.even_y && (key[2 * cl - 1] & 1)) {
		int borrow = 0;
		for (u32 b = cl; b-- > 0;) {
			const int d = (int)A.p_be[b] - (int)key[cl + b] - borrow;
			ko[cl + b] = (u8)(d & 0xff);
			borrow = d < 0;
		}
	} else {
		for (u32 b = 0; b < cl; b++) {
			ko[cl + b] = key[cl + b];
		}
	}
	u8 *ro = A.r_out + (size_t)i * rl, *so = A.s_out + (size_t)i * ql;
	for (u32 b = 0; b < rl; b++) {
		ro[b] = sig[b];
	}
	for (u32 b = 0; b < ql; b++) {
		so[b] = sig[rl + b];
	}
}
hipError_t ecamd_launch_schnorr_prep(const EcamdSchnorrPrepArgs &a, hipStream_t s)
{
	if (a.n) {
		hipLaunchKernelGGL(k_schnorr_prep, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
	}
	return hipGetLastError();
}
hipError_t ecamd_launch_reject_where(uint8_t *result, const uint8_t *status, uint32_t n, hipStream_t s)
{
	if (n) {
		hipLaunchKernelGGL(k_reject_where, dim3((n + 255) / 256), dim3(256), 0, s, result, status, n);
	}
	return hipGetLastError();
}
