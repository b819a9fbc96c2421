// Q-learning guided sampling of the indirect bounce (BASELINE config 5 "Q-learning sampler on").
// The reference snapshot holds NO code for it -- only README.md:36-42: "use QLearning to influence the sampling
// direction (Dahm & Keller 2017, Learning Light Transport the Reinforced Way) ... initialize sampling positions; pick sampling
// direction according to the QValue of neighboring points; store and update directions with a corresponding probability per
// sampling point".  What follows is therefore this repository's own statement of that paper's scheme (parity unpinned; the
// CPU checker of the test tier states the same definition, and the two are held bit against bit on the table and within 1e-4
// on the frames):
//   sampling positions   the cells of a G x G x G grid over a caller-given box ("voxel" of a hit point)
//   directions           64 equal-area patches of the sphere per cell: 8 bands in z x 8 sectors in phi (pi / 16 sr each)
//   Q[cell][patch]       a scalar (luminance) estimate of the radiance arriving at the cell from the patch
//   pick                 at a DIFFUSE hit the scattered direction's patch is drawn with P = (1 - eps) Q / sum(Q) + eps / 64
//                        (every direction keeps a positive density: unbiased), the direction uniformly inside the patch; the
//                        estimator's factor 2 = 1 / (pi pdf) of the uniform hemisphere becomes 1 / (16 P); a direction
//                        below the surface contributes nothing and teaches the patch a reward of 0
//   update (eq. 8)       when the scattered ray's hit y is shaded: reward = what y emits towards x (sky, light) or, for a
//                        surface, the expected reflected Q at y: rho / 16 * V[cell(y)][patch(n_y)] (diffuse) with
//                        V[cell][m] = sum_p Q[cell][p] max(0, d_m . d_p) over the 64 patch centres -- the integral of eq. 8 with the
//                        normal quantised to the patch it points into; luminance(col) * mean Q[cell(y)] (specular).  V is
//                        recomputed whenever Q changes (k_q_init, k_q_apply: a 64 x 64 product per cell, microseconds), so a hit
//                        reads ONE value where round 3 read the cell's 64 and took 64 dot products (config 5: shade 1.16 s of
//                        2.20 s).  Rewards are summed as 48.16 fixed-point INTEGERS with a count per (cell, patch) (one packed
//                        64-bit atomic per reward, q_reward);
//                        rt_qlearn_apply folds them into Q <- (1 - alpha) Q + alpha mean between batches.  Within a batch Q
//                        and V are read-only.
// Why integers: sums of integers do not depend on the order in which lanes, waves or GPUs add them, so a frame is
// reproducible, equals the CPU statement's, and -- with the sums all-reduced between ranks before the apply (bench.py) -- does
// not depend on how the rows were sharded.  The rule this picks for "per-pixel streams must stay independent of sharding":
// learning happens BETWEEN batches, never inside one.
#pragma once
#include "rt_kernels.h"

namespace rtd {

#define RT_Q_PATCHES 64
#define RT_Q_ROW 72 // floats per cell: 8 band sums, then 64 values (band-major)
struct QTable {
	float* q;            // [cells][RT_Q_ROW]
	float* v;            // [cells][64] V[cell][m] = sum_p Q[cell][p] * wgt[m][p]: the expected reflected Q for a normal in patch m
	const float* wgt;    // [64][64] max(0, d_m . d_p) of the patch centres
	long long* sum;      // [cells][64] rewards of the current batch, 48.16 fixed point   } the exchange format (rt_qlearn_get_sums / set_sums);
	uint* cnt;           // [cells][64]                                                   } k_q_fold moves acc[] here
	unsigned long long* acc; // [cells][64] what the shading kernel adds to: count << 44 | sum, ONE atomic per reward (see q_reward)
	const float4* centre; // [64] patch centre directions
	int grid, on;
	float lo[3], inv[3]; // cell = (int)((x - lo) * inv), clamped
	float eps, alpha, qMin;
	uint learnMask;      // a sample pays rewards iff (its stream's state after the pixel jitter) & learnMask == 0
	int* ovf;            // sticky: a count field of acc[] got past RT_Q_ACC_LIMIT (q_reward sees it in the word it adds to, k_q_fold in the word it folds)
};
#define RT_Q_LEARNER 0x80000000u // bit of an entry's key word (W.w): this sample pays rewards

__device__ __forceinline__ float q_lum(const f3& c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
__device__ __forceinline__ int q_cell(const QTable& Q, const f3& x)
{
	int i[3];
	const float f[3] = { (x.x - Q.lo[0]) * Q.inv[0], (x.y - Q.lo[1]) * Q.inv[1], (x.z - Q.lo[2]) * Q.inv[2] };
	for (int a = 0; a < 3; a++) {
		int k = f[a] > 0 ? (f[a] < (float)Q.grid ? (int)f[a] : Q.grid - 1) : 0; // NaN -> 0
		i[a] = k < Q.grid ? k : Q.grid - 1;
	}
	return (i[2] * Q.grid + i[1]) * Q.grid + i[0];
}
// direction of patch (band i, sector j) at (u1, u2) in [0, 1)^2
__device__ __forceinline__ f3 q_direction(int i, int j, float u1, float u2)
{
	const float z = -1 + ((float)i + u1) * 0.25f;
	const float phi = ((float)j + u2) * (RT_TWOPI * 0.125f);
	const float s = sqrtf(t_fmaxf(0.f, 1 - z * z));
	return f3(s * x_cosf(phi), s * x_sinf(phi), z);
}
// the patch a unit vector points into: band from z, sector from the signs and the larger of |x|, |y| (the sector boundaries are
// the multiples of 45 degrees: no arc tangent needed)
__device__ __forceinline__ int q_patch_of(const f3& n)
{
	const float fz = (n.z + 1) * 4;
	const int i = fz > 0 ? (fz < 8 ? (int)fz : 7) : 0; // NaN -> 0
	const float ax = fabsf(n.x), ay = fabsf(n.y);
	int j;
	if (n.y >= 0) j = n.x > 0 ? (ay < ax ? 0 : 1) : (ax < ay ? 2 : 3);
	else j = n.x < 0 ? (ay < ax ? 4 : 5) : (ax < ay ? 6 : 7);
	if (n.x == 0 && n.y == 0) j = 0;
	return 8 * i + j;
}
// the band sums and the V row of cell t from its 64 values, in patch order
__device__ __forceinline__ void q_derive(const QTable& Q, int t)
{
	float* row = Q.q + (size_t)t * RT_Q_ROW;
	for (int i = 0; i < 8; i++) {
		float b = 0;
		for (int j = 0; j < 8; j++) b = b + row[8 + 8 * i + j];
		row[i] = b;
	}
	for (int m = 0; m < RT_Q_PATCHES; m++) {
		float s = 0;
		for (int p = 0; p < RT_Q_PATCHES; p++) s = s + row[8 + p] * Q.wgt[m * RT_Q_PATCHES + p];
		Q.v[(size_t)t * RT_Q_PATCHES + m] = s;
	}
}
// patch centres and their pairwise weights max(0, d_m . d_p) (before k_q_init: one launch of 64 x 64 threads)
__global__ void k_q_weights(float4* centre, float* wgt)
{
	const int m = blockIdx.x, p = threadIdx.x;
	const f3 dm = q_direction(m >> 3, m & 7, 0.5f, 0.5f), dp = q_direction(p >> 3, p & 7, 0.5f, 0.5f);
	if (m == 0) centre[p] = mk4(dp, 0.0f);
	wgt[m * RT_Q_PATCHES + p] = t_fmaxf(0.f, dot(dm, dp));
}
__global__ void k_q_init(QTable Q, float qInit)
{
	const int cells = Q.grid * Q.grid * Q.grid;
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= cells) return;
	float* row = Q.q + (size_t)t * RT_Q_ROW;
	for (int p = 0; p < RT_Q_PATCHES; p++) row[8 + p] = qInit, Q.sum[(size_t)t * RT_Q_PATCHES + p] = 0, Q.cnt[(size_t)t * RT_Q_PATCHES + p] = 0, Q.acc[(size_t)t * RT_Q_PATCHES + p] = 0;
	q_derive(Q, t);
}
// fold the batch's rewards into the table: one thread per cell
__global__ void k_q_apply(QTable Q)
{
	const int cells = Q.grid * Q.grid * Q.grid;
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= cells) return;
	float* row = Q.q + (size_t)t * RT_Q_ROW;
	for (int p = 0; p < RT_Q_PATCHES; p++) {
		const size_t k = (size_t)t * RT_Q_PATCHES + p;
		const uint n = Q.cnt[k];
		if (n) {
			const float mean = (float)((double)Q.sum[k] / ((double)n * 65536.0));
			row[8 + p] = t_fmaxf((1 - Q.alpha) * row[8 + p] + Q.alpha * mean, Q.qMin);
			Q.sum[k] = 0, Q.cnt[k] = 0;
		}
	}
	q_derive(Q, t);
}

// the reward a scattered ray brings back to (cell, patch) = key - 1.
// Scattered global atomics execute at the memory side on this part, one 64-byte request per lane, ~16-20 G per second for the whole
// chip (MI355X_MICROARCH.md "Global float atomics": 64 lanes in 64 rows 17x below the contiguous rate; measured here: config 5's
// shade launches 0.79 s with 0.1 % of the samples paying rewards, 1.02 s with 25 %, 2.06 s with all -- profiles/r05_ab_qlearn_atomics.txt).
// A reward was two of them (sum, count); it is ONE: the count rides in the top 20 bits of the 64-bit word, the sum (<= 2^22 per
// reward) in the low 44.  k_q_fold moves the words into the wide sums at the end of every batch of frames (stream-ordered, no host
// wait) and reports a count field that got past half its range (RT_Q_ACC_LIMIT rewards for one (cell, patch) within one batch: 7x
// what config 5 reaches with every sample paying) as RT_E_OVERFLOW.  The field cannot wrap unseen either: the atomic returns the
// word it added to, and a reward that lands on a count in [2^19, 2^20) -- bit 63 of the old word -- is reported by the lane that
// paid it (the 2^19 rewards a word must take before its count wraps all see that bit).
#define RT_Q_ACC_SHIFT 44
#define RT_Q_ACC_LIMIT (1u << 19)
// returns the top half of the word the reward was added to (bit 31: the count was already past RT_Q_ACC_LIMIT)
__device__ __forceinline__ uint q_reward(const QTable& Q, uint key, float R)
{
	R = (R >= 0) ? (R < 64.0f ? R : 64.0f) : 0.0f; // a directly seen light is +inf in the reference (Q7); NaN teaches nothing
	const unsigned long long old = atomicAdd(&Q.acc[key - 1], (1ull << RT_Q_ACC_SHIFT) | (unsigned long long)__float2ll_rn(R * 65536.0f));
	return (uint)(old >> 32);
}
// acc[] -> sum[], cnt[] (one thread per (cell, patch)); *Q.ovf = 1 when a count field is past RT_Q_ACC_LIMIT
__global__ void k_q_fold(QTable Q)
{
	const size_t n = (size_t)Q.grid * Q.grid * Q.grid * RT_Q_PATCHES;
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const unsigned long long a = Q.acc[k];
	if (a == 0) return;
	const uint c = (uint)(a >> RT_Q_ACC_SHIFT);
	Q.sum[k] += (long long)(a & ((1ull << RT_Q_ACC_SHIFT) - 1)), Q.cnt[k] += c;
	Q.acc[k] = 0;
	if (c >= RT_Q_ACC_LIMIT) *Q.ovf = 1;
}
// expected reflected Q at a surface hit (the integral of eq. 8 over the 64 patches): one value of the cell's V row for a diffuse
// surface, the mean of the cell's Q for a specular one
__device__ __forceinline__ float q_expected(const QTable& Q, int cell, const f3& normal, float rho, bool diffuse)
{
	if (!diffuse) {
		const float4* row4 = (const float4*)(Q.q + (size_t)cell * RT_Q_ROW);
		const float4 b0 = row4[0], b1 = row4[1];
		float T = 0;
		T = T + b0.x, T = T + b0.y, T = T + b0.z, T = T + b0.w, T = T + b1.x, T = T + b1.y, T = T + b1.z, T = T + b1.w;
		return rho * (T * (1.0f / 64));
	}
	return rho * (Q.v[(size_t)cell * RT_Q_PATCHES + q_patch_of(normal)] * (1.0f / 16));
}
// the guided pick at a diffuse hit: four draws (mixture, patch, two inside the patch)
__device__ __forceinline__ f3 q_sample(const QTable& Q, int cell, uint& seed, float& P, int& patch)
{
	const float* row = Q.q + (size_t)cell * RT_Q_ROW;
	const float4* row4 = (const float4*)row;
	const float4 b0 = row4[0], b1 = row4[1];
	const float b[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
	float T = 0;
	for (int i = 0; i < 8; i++) T = T + b[i];
	const float uSel = RandomFloat(seed), uPick = RandomFloat(seed), u1 = RandomFloat(seed), u2 = RandomFloat(seed);
	int i = 0, j = 0;
	float qp;
	if (uSel < Q.eps || !(T > 0)) {
		patch = (int)(uPick * 64);
		if (patch > 63) patch = 63;
		i = patch >> 3, j = patch & 7;
		qp = row[8 + patch];
	} else {
		const float x = uPick * T;
		float acc = 0;
		for (i = 0; i < 7; i++) {
			if (x < acc + b[i]) break;
			acc = acc + b[i];
		}
		const float x2 = x - acc;
		const float4 v0 = row4[2 + 2 * i], v1 = row4[3 + 2 * i]; // the band's eight values
		const float v[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
		float acc2 = 0;
		for (j = 0; j < 7; j++) {
			if (x2 < acc2 + v[j]) break;
			acc2 = acc2 + v[j];
		}
		patch = 8 * i + j;
		qp = v[j];
	}
	P = T > 0 ? (1 - Q.eps) * (qp / T) + Q.eps * (1.0f / 64) : 1.0f / 64;
	return q_direction(i, j, u1, u2);
}

} // namespace rtd
