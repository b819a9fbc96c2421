// ORACLE (test infrastructure, NOT product code) -- PARITY UNPINNED: the reference snapshot holds no code for this
// (SURVEY.md F2).  CPU statement of the Q-learning guided sampler the product implements in
// ray-and-pathtracer_amd/csrc/rt_qlearn.h -- this repository's own reading of /root/reference/README.md:36-42
// ("use QLearning to influence the sampling direction (Dahm & Keller 2017) ... initialize sampling positions; pick sampling
// direction according to the QValue of neighboring points; store and update directions with a corresponding probability per
// sampling point").  Same definitions, same operand order: cells of a grid^3 box, 64 equal-area direction patches per cell
// (8 bands in z x 8 sectors in phi), P(patch) = (1 - eps) Q / sum Q + eps / 64, rewards summed as 48.16 fixed-point
// integers per (cell, patch) and folded into Q <- (1 - alpha) Q + alpha mean between batches.  The reward a diffuse surface
// hands back is rho / 16 * V[cell][patch(normal)], V[cell][m] = sum_p Q[cell][p] max(0, d_m . d_p) over the patch centres
// (the normal quantised to the patch it points into; V follows every change of Q).
#pragma once
#include "orc_math.h"
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc {

struct QLearn {
	bool on = false;
	int grid = 0;
	float lo[3] = { 0, 0, 0 }, inv[3] = { 0, 0, 0 };
	float eps = 0, alpha = 0, qMin = 1e-4f;
	uint learnMask = 0;          // a sample pays rewards iff (its stream's state after the pixel jitter) & learnMask == 0
	std::vector<float> q;        // [cells][72]: 8 band sums, then 64 values
	std::vector<float> v;        // [cells][64]: V[cell][m] = sum_p Q[cell][p] * wgt[m][p]
	float wgt[64 * 64];          // max(0, d_m . d_p) of the patch centres
	std::vector<long long> sum;  // [cells][64]
	std::vector<uint> cnt;       // [cells][64]
	float3 centre[64];

	static float lum(const float3& c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
	static float3 direction(int i, int j, float u1, float u2)
	{
		const float z = -1 + ((float)i + u1) * 0.25f;
		const float phi = ((float)j + u2) * (TWOPI * 0.125f);
		const float s = sqrtf(t_fmaxf(0.f, 1 - z * z));
		return float3(s * x_cosf(phi), s * x_sinf(phi), z);
	}
	// the patch a unit vector points into: band from z, sector from the signs and the larger of |x|, |y|
	static int patch_of(const float3& n)
	{
		const float fz = (n.z + 1) * 4;
		const int i = fz > 0 ? (fz < 8 ? (int)fz : 7) : 0;
		const float ax = fabsf(n.x), ay = fabsf(n.y);
		int j;
		if (n.y >= 0) j = n.x > 0 ? (ay < ax ? 0 : 1) : (ax < ay ? 2 : 3);
		else j = n.x < 0 ? (ay < ax ? 4 : 5) : (ax < ay ? 6 : 7);
		if (n.x == 0 && n.y == 0) j = 0;
		return 8 * i + j;
	}
	// band sums and V row of a cell from its 64 values
	void rebuild_bands(size_t cell)
	{
		float* row = &q[cell * 72];
		for (int i = 0; i < 8; i++) {
			float b = 0;
			for (int j = 0; j < 8; j++) b = b + row[8 + 8 * i + j];
			row[i] = b;
		}
		for (int m = 0; m < 64; m++) {
			float s = 0;
			for (int p = 0; p < 64; p++) s = s + row[8 + p] * wgt[m * 64 + p];
			v[cell * 64 + m] = s;
		}
	}
	// load a table (e.g. one the device learned): cells * 64 values
	void set_table(const float* table)
	{
		const size_t cells = (size_t)grid * grid * grid;
		for (size_t c = 0; c < cells; c++) {
			for (int p = 0; p < 64; p++) q[c * 72 + 8 + p] = table[c * 64 + p];
			rebuild_bands(c);
		}
	}
	void enable(int g, const float* l, const float* h, float a, float e, float qInit, uint mask = 0)
	{
		on = true, grid = g, alpha = a, eps = e, learnMask = mask;
		for (int k = 0; k < 3; k++) lo[k] = l[k], inv[k] = (float)g / (h[k] - l[k]);
		const size_t cells = (size_t)g * g * g;
		q.assign(cells * 72, 0.0f), v.assign(cells * 64, 0.0f), sum.assign(cells * 64, 0), cnt.assign(cells * 64, 0);
		for (int p = 0; p < 64; p++) centre[p] = direction(p >> 3, p & 7, 0.5f, 0.5f);
		for (int m = 0; m < 64; m++)
			for (int p = 0; p < 64; p++) wgt[m * 64 + p] = t_fmaxf(0.f, dot(direction(m >> 3, m & 7, 0.5f, 0.5f), direction(p >> 3, p & 7, 0.5f, 0.5f)));
		for (size_t c = 0; c < cells; c++) {
			for (int p = 0; p < 64; p++) q[c * 72 + 8 + p] = qInit;
			rebuild_bands(c);
		}
	}
	int cell(const float3& x) const
	{
		int i[3];
		const float f[3] = { (x.x - lo[0]) * inv[0], (x.y - lo[1]) * inv[1], (x.z - lo[2]) * inv[2] };
		for (int a = 0; a < 3; a++) {
			int k = f[a] > 0 ? (f[a] < (float)grid ? (int)f[a] : grid - 1) : 0;
			i[a] = k < grid ? k : grid - 1;
		}
		return (i[2] * grid + i[1]) * grid + i[0];
	}
	void reward(uint key, float R)
	{
		R = (R >= 0) ? (R < 64.0f ? R : 64.0f) : 0.0f;
		const long long fixed = llrintf(R * 65536.0f);
		__atomic_fetch_add(&sum[key - 1], fixed, __ATOMIC_RELAXED);
		__atomic_fetch_add(&cnt[key - 1], 1u, __ATOMIC_RELAXED);
	}
	float expected(int c, const float3& normal, float rho, bool diffuse) const
	{
		const float* row = &q[(size_t)c * 72];
		if (!diffuse) {
			float T = 0;
			for (int i = 0; i < 8; i++) T = T + row[i];
			return rho * (T * (1.0f / 64));
		}
		return rho * (v[(size_t)c * 64 + patch_of(normal)] * (1.0f / 16));
	}
	float3 sample(int c, uint& seed, float& P, int& patch) const
	{
		const float* row = &q[(size_t)c * 72];
		float b[8];
		float T = 0;
		for (int i = 0; i < 8; i++) b[i] = row[i], T = T + b[i];
		const float uSel = RandomFloat(seed), uPick = RandomFloat(seed), u1 = RandomFloat(seed), u2 = RandomFloat(seed);
		int i = 0, j = 0;
		float qp;
		if (uSel < eps || !(T > 0)) {
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
			float acc2 = 0;
			for (j = 0; j < 7; j++) {
				if (x2 < acc2 + row[8 + 8 * i + j]) break;
				acc2 = acc2 + row[8 + 8 * i + j];
			}
			patch = 8 * i + j;
			qp = row[8 + patch];
		}
		P = T > 0 ? (1 - eps) * (qp / T) + eps * (1.0f / 64) : 1.0f / 64;
		return direction(i, j, u1, u2);
	}
	void apply()
	{
		const size_t cells = (size_t)grid * grid * grid;
		for (size_t c = 0; c < cells; c++) {
			float* row = &q[c * 72];
			for (int p = 0; p < 64; p++) {
				const size_t k = c * 64 + p;
				const uint n = cnt[k];
				if (n) {
					const float mean = (float)((double)sum[k] / ((double)n * 65536.0));
					row[8 + p] = t_fmaxf((1 - alpha) * row[8 + p] + alpha * mean, qMin);
					sum[k] = 0, cnt[k] = 0;
				}
			}
			rebuild_bands(c);
		}
	}
};

} // namespace orc
