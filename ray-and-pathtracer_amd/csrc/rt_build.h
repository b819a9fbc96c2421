// bvh::Build (bvh.cpp:18-56, 67-114, 116-200, 223-333, 514-554) on the device, all four split methods of bvh.h:38-43
// (BINNEDSAH: 8-bin SAH, the default; SAMESIZE: median of the longest axis; LONGESTAXIS: spatial middle of the
// longest axis; SAH: every centroid of every axis tried as the plane, EvaluateSAH, O(n^2) like the reference): the step
// immediately before the trace loop.  The tree must come out IDENTICAL to the
// reference's -- node numbering, boxes, primitiveIdx order -- because the stored boxes and the leaf order
// decide which primitive a ray reports, so this is a restatement of the same arithmetic, reorganised:
//
//   * level by level instead of depth first: every open node of a level is subdivided by one block; the
//     depth-first numbering (children pairs are allocated in the order Subdivide reaches their parents) is
//     restored at the end from the tree's shape;
//   * centroid bounds, bin counts and bin boxes are min / max / integer sums: any order gives the same bits
//     (inputs are checked to be finite: NaN would make the reference's ternary min / max order dependent);
//   * the 7-plane sweep and the '<' comparisons run on one lane in the reference's order;
//   * the in-place partition loop (bvh.cpp:296-313) moves elements in a fixed pattern that has a closed
//     form: with L = "centroid < splitPos", nL = #L, holes h_1 < h_2 < ... = positions below nL holding an R,
//     fillers f_1 > f_2 > ... = positions from nL up holding an L (f_0 = count, K of each):
//         L below nL          stays
//         hole h_k            -> f_(k-1) - 1
//         filler f_k          -> h_k
//         R at q > f_K        -> q - 1
//         R at q < f_K        -> q - 1, except q = nL -> f_K - 1
//     (tests/test_host_cpu.py checks the closed form against the loop on random arrays.)
#pragma once
#include "rt_dmath.h"

namespace rtd {

#define RT_BUILD_BINS 8 // bvh.cpp:118

struct TNode { // a node in creation order (not yet the reference's numbering)
	float lo[3]; uint first;
	float hi[3]; uint count;
	int left, right; // TNode indices of the children, -1: leaf
};

struct BuildArrays {
	float4* cen;  // centroid (triangle: (v0 + v1 + v2) * 0.333f, template/scene.h:186; sphere: pos)
	float4* nlo;  // what UpdateNodeBounds grows a node by (triangle: its vertices; sphere: pos -+ r)
	float4* nhi;
	float4* blo;  // what FindBestSplitPlane grows a bin by (sphere: pos -+ 2r, Q3)
	float4* bhi;
	uint* idx;    // primitiveIdx
	uint* tmp;    // partition target
	uint* hpos;   // holes / fillers of the node being partitioned, at the node's own range
	uint* fpos;
	TNode* nodes;
	int* counters; // [0] nodes used, [2] error flag, [4] / [5] open nodes of the even / odd levels
};

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_sum(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int wave_excl(int v, uint lane) // exclusive prefix sum over the wave
{
	int incl = v;
	for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
	return incl - v;
}
#define RT_BUILD_THREADS 256 // threads that subdivide one node
#define RT_BUILD_WAVES (RT_BUILD_THREADS / 64)
// block-wide reductions over RT_BUILD_THREADS threads ('sh': RT_BUILD_WAVES words of LDS per call site)
__device__ __forceinline__ float block_min(float v, float* sh)
{
	v = wave_min(v);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	float r = sh[0];
	for (int w = 1; w < RT_BUILD_WAVES; w++) r = fminf(r, sh[w]);
	__syncthreads();
	return r;
}
__device__ __forceinline__ float block_max(float v, float* sh)
{
	v = wave_max(v);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	float r = sh[0];
	for (int w = 1; w < RT_BUILD_WAVES; w++) r = fmaxf(r, sh[w]);
	__syncthreads();
	return r;
}
// exclusive prefix sum over the block and the block's total
__device__ __forceinline__ int block_excl(int v, int* sh, int& total)
{
	const uint lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int inWave = wave_excl(v, lane);
	const int ws = wave_sum(v);
	if (lane == 0) sh[wave] = ws;
	__syncthreads();
	int before = 0;
	total = 0;
	for (int w = 0; w < RT_BUILD_WAVES; w++) { if (w < (int)wave) before += sh[w]; total += sh[w]; }
	__syncthreads();
	return before + inWave;
}
__device__ __forceinline__ bool finite3(const float* v) { return fabsf(v[0]) < 1e30f && fabsf(v[1]) < 1e30f && fabsf(v[2]) < 1e30f; }

// per-primitive quantities; primitives [0, nTri) are triangles, [nTri, nTri + nSph) spheres
__global__ void k_build_prep(const float* tris, int triStride, int nTri, const float* sph, int sphStride, int nSph, BuildArrays B)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nTri + nSph) return;
	float c[3], lo[3], hi[3], l2[3], h2[3];
	bool ok;
	if (i < nTri) {
		const float* t = tris + (size_t)i * triStride; // v0[3] v1[3] v2[3]
		ok = finite3(t) && finite3(t + 3) && finite3(t + 6);
		for (int k = 0; k < 3; k++) {
			c[k] = ((t[k] + t[3 + k]) + t[6 + k]) * 0.333f;
			lo[k] = fminf(fminf(t[k], t[3 + k]), t[6 + k]), hi[k] = fmaxf(fmaxf(t[k], t[3 + k]), t[6 + k]);
			l2[k] = lo[k], h2[k] = hi[k];
		}
	} else {
		const float* s = sph + (size_t)(i - nTri) * sphStride; // pos[3] r2 invr r
		const float r = s[5];
		ok = finite3(s) && fabsf(r) < 1e30f;
		for (int k = 0; k < 3; k++) {
			c[k] = s[k];
			lo[k] = s[k] - r, hi[k] = s[k] + r;
			l2[k] = s[k] - 2 * r, h2[k] = s[k] + 2 * r;
		}
	}
	if (!ok) B.counters[2] = 1;
	// w: the sphere's radius / the sphere flag (EvaluateSAH grows a side by pos[axis] -+ r broadcast to all three axes, bvh.cpp:533-544)
	B.cen[i] = make_float4(c[0], c[1], c[2], i < nTri ? 0.0f : sph[(size_t)(i - nTri) * sphStride + 5]);
	B.nlo[i] = make_float4(lo[0], lo[1], lo[2], i < nTri ? 0.0f : 1.0f), B.nhi[i] = make_float4(hi[0], hi[1], hi[2], 0);
	B.blo[i] = make_float4(l2[0], l2[1], l2[2], 0), B.bhi[i] = make_float4(h2[0], h2[1], h2[2], 0);
	B.idx[i] = (uint)i;
}

// UpdateNodeBounds over a range of primitiveIdx (bvh.cpp:67-114), one block
__device__ __forceinline__ void range_bounds(const BuildArrays& B, uint first, uint count, float* sh, float* lo, float* hi)
{
	float l[3] = { 1e30f, 1e30f, 1e30f }, h[3] = { -1e30f, -1e30f, -1e30f };
	for (uint i = threadIdx.x; i < count; i += RT_BUILD_THREADS) {
		const uint p = B.idx[first + i];
		const float4 a = B.nlo[p], b = B.nhi[p];
		l[0] = fminf(l[0], a.x), l[1] = fminf(l[1], a.y), l[2] = fminf(l[2], a.z);
		h[0] = fmaxf(h[0], b.x), h[1] = fmaxf(h[1], b.y), h[2] = fmaxf(h[2], b.z);
	}
	for (int k = 0; k < 3; k++) lo[k] = block_min(l[k], sh), hi[k] = block_max(h[k], sh);
}

__global__ void __launch_bounds__(RT_BUILD_THREADS) k_build_root(BuildArrays B, uint count)
{
	__shared__ float shf[RT_BUILD_WAVES];
	float lo[3], hi[3];
	range_bounds(B, 0, count, shf, lo, hi);
	if (threadIdx.x == 0) {
		TNode& n = B.nodes[0];
		for (int k = 0; k < 3; k++) n.lo[k] = lo[k], n.hi[k] = hi[k];
		n.first = 0, n.count = count, n.left = n.right = -1;
		B.counters[0] = 1;
	}
}

struct SweepBox { // aabb of template/precomp.h as FindBestSplitPlane uses it
	float lo[3], hi[3];
	__device__ __forceinline__ void reset() { for (int k = 0; k < 3; k++) lo[k] = 1e30f, hi[k] = -1e30f; }
	__device__ __forceinline__ void grow(const float* blo, const float* bhi) // aabb::grow(const aabb&): skipped for an untouched box
	{
		if (blo[0] != 1e30f) {
			for (int k = 0; k < 3; k++) { lo[k] = t_fminf(lo[k], blo[k]); hi[k] = t_fmaxf(hi[k], blo[k]); }
			for (int k = 0; k < 3; k++) { lo[k] = t_fminf(lo[k], bhi[k]); hi[k] = t_fmaxf(hi[k], bhi[k]); }
		}
	}
	__device__ __forceinline__ float area() const { const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2]; return ex * ey + ey * ez + ez * ex; }
};

// Subdivide (bvh.cpp:223-333) for every open node of one level: one block per node
#define RT_SPLIT_BINNEDSAH 0
#define RT_SPLIT_SAMESIZE 1
#define RT_SPLIT_LONGESTAXIS 2
#define RT_SPLIT_SAH 3
// block-wide "first minimum": the smallest key wins, equal keys by the smaller index (the reference's loops keep the
// first candidate under a strict '<'); a NaN key never wins
__device__ __forceinline__ void block_argmin(float& key, int& idx, float* shk, int* shi2)
{
	if (!(key == key)) key = __builtin_inff(), idx = 0x7fffffff;
	for (int o = 32; o > 0; o >>= 1) {
		const float k2 = __shfl_xor(key, o);
		const int i2 = __shfl_xor(idx, o);
		if (k2 < key || (k2 == key && i2 < idx)) key = k2, idx = i2;
	}
	if ((threadIdx.x & 63) == 0) shk[threadIdx.x >> 6] = key, shi2[threadIdx.x >> 6] = idx;
	__syncthreads();
	key = shk[0], idx = shi2[0];
	for (int w = 1; w < RT_BUILD_WAVES; w++) if (shk[w] < key || (shk[w] == key && shi2[w] < idx)) key = shk[w], idx = shi2[w];
	__syncthreads();
}
__device__ __forceinline__ float comp3(const float4& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

__global__ void __launch_bounds__(RT_BUILD_THREADS) k_build_level(BuildArrays B, const int* open, int* next, int level, int method)
{
	__shared__ float binLo[RT_BUILD_BINS * 3][RT_BUILD_THREADS], binHi[RT_BUILD_BINS * 3][RT_BUILD_THREADS]; // [bin * 3 + axis of the box][thread]: private columns
	__shared__ int binCnt[RT_BUILD_BINS][RT_BUILD_THREADS];
	__shared__ float redLo[RT_BUILD_BINS * 3], redHi[RT_BUILD_BINS * 3];
	__shared__ int redCnt[RT_BUILD_BINS];
	__shared__ float shf[RT_BUILD_WAVES];
	__shared__ int shi[RT_BUILD_WAVES];
	__shared__ float bestS[2];
	__shared__ int axisS;
	const uint tid = threadIdx.x;
	// the open list of a level: counters[4 + (level & 1)] entries, written by the level before
	if ((int)blockIdx.x >= B.counters[4 + (level & 1)]) return;
	const int id = open[blockIdx.x];
	const TNode node = B.nodes[id];
	const uint first = node.first, count = node.count;

	if (method != RT_SPLIT_BINNEDSAH) {
		// ---- the other three split methods (bvh.cpp:226-293) ----
		const float ext[3] = { node.hi[0] - node.lo[0], node.hi[1] - node.lo[1], node.hi[2] - node.lo[2] };
		int la = 0; // the longest axis as the reference picks it (:228-231, :236-239)
		if (ext[1] > ext[0]) la = 1;
		if (ext[2] > ext[la]) la = 2;
		if (method == RT_SPLIT_LONGESTAXIS) {
			if (tid == 0) axisS = la, bestS[1] = node.lo[la] + ext[la] * 0.5f, bestS[0] = -1.0f;
		} else if (method == RT_SPLIT_SAMESIZE) {
			// std::sort of (key, primIdx) tuples, splitPos = key of element count / 2 (:240-252): the key k with
			// #{keys < k} <= m < #{keys <= k}
			const int m = (int)(count / 2);
			if (tid == 0) axisS = la, bestS[0] = -1.0f, bestS[1] = 0;
			__syncthreads();
			for (uint i = tid; i < count; i += RT_BUILD_THREADS) {
				const float k = comp3(B.cen[B.idx[first + i]], la);
				int less = 0, leq = 0;
				for (uint j = 0; j < count; j++) { const float kj = comp3(B.cen[B.idx[first + j]], la); less += kj < k ? 1 : 0, leq += kj <= k ? 1 : 0; }
				if (less <= m && m < leq) bestS[1] = k; // every thread that gets here writes the same value
			}
		} else {
			// SAH (:275-293) with EvaluateSAH (:514-554): candidate q = axis * count + i, first minimum wins
			float bestC = 1e30f;
			int bestQ = 0x7fffffff;
			for (uint q = tid; q < 3 * count; q += RT_BUILD_THREADS) {
				const int a = (int)(q / count);
				const float pos = comp3(B.cen[B.idx[first + q % count]], a);
				SweepBox lb, rb;
				lb.reset(), rb.reset();
				int lc = 0, rc = 0;
				for (uint j = 0; j < count; j++) {
					const uint p = B.idx[first + j];
					const float4 c4 = B.cen[p], lo4 = B.nlo[p], hi4 = B.nhi[p];
					const bool left = comp3(c4, a) < pos;
					float glo[3] = { lo4.x, lo4.y, lo4.z }, ghi[3] = { hi4.x, hi4.y, hi4.z };
					if (lo4.w != 0.0f) { // sphere: the scalar pos[axis] -+ r on all three axes
						const float s0 = comp3(c4, a) - c4.w, s1 = comp3(c4, a) + c4.w;
						glo[0] = glo[1] = glo[2] = s0, ghi[0] = ghi[1] = ghi[2] = s1;
					}
					SweepBox& bx = left ? lb : rb;
					for (int k = 0; k < 3; k++) { bx.lo[k] = t_fminf(bx.lo[k], glo[k]); bx.hi[k] = t_fmaxf(bx.hi[k], glo[k]); }
					for (int k = 0; k < 3; k++) { bx.lo[k] = t_fminf(bx.lo[k], ghi[k]); bx.hi[k] = t_fmaxf(bx.hi[k], ghi[k]); }
					lc += left ? 1 : 0, rc += left ? 0 : 1;
				}
				float cost = lc * lb.area() + rc * rb.area();
				cost = cost > 0 ? cost : 1e30f;
				if (cost < bestC) bestC = cost, bestQ = (int)q; // q ascends within a thread: its first minimum
			}
			block_argmin(bestC, bestQ, shf, shi);
			if (tid == 0) {
				const bool found = bestC < 1e30f && bestQ != 0x7fffffff; // the reference would index centroid[-1] otherwise: "no split"
				bestS[0] = found ? -1.0f : 1e30f;
				if (found) axisS = bestQ / (int)count, bestS[1] = comp3(B.cen[B.idx[first + (uint)bestQ % count]], bestQ / (int)count);
			}
		}
		__syncthreads();
	}
	// ---- FindBestSplitPlane (bvh.cpp:116-193) ----
	if (method == RT_SPLIT_BINNEDSAH && tid == 0) bestS[0] = 1e30f, bestS[1] = 0, axisS = 0;
	for (int a = 0; a < 3 && method == RT_SPLIT_BINNEDSAH; a++) {
		float mn = 1e30f, mx = -1e30f;
		for (uint i = tid; i < count; i += RT_BUILD_THREADS) {
			const float4 c4 = B.cen[B.idx[first + i]];
			const float c = a == 0 ? c4.x : (a == 1 ? c4.y : c4.z);
			mn = fminf(mn, c), mx = fmaxf(mx, c);
		}
		mn = block_min(mn, shf), mx = block_max(mx, shf);
		if (mn == mx) continue;
		float scale = RT_BUILD_BINS / (mx - mn);
		for (int b = 0; b < RT_BUILD_BINS; b++) {
			binCnt[b][tid] = 0;
			for (int k = 0; k < 3; k++) binLo[b * 3 + k][tid] = 1e30f, binHi[b * 3 + k][tid] = -1e30f;
		}
		for (uint i = tid; i < count; i += RT_BUILD_THREADS) {
			const uint p = B.idx[first + i];
			const float4 c4 = B.cen[p];
			const float c = a == 0 ? c4.x : (a == 1 ? c4.y : c4.z);
			int b = f2i((c - mn) * scale);
			b = b < RT_BUILD_BINS - 1 ? b : RT_BUILD_BINS - 1; // std::min(BINS - 1, ...)
			if (b < 0) { B.counters[2] = 1; b = 0; }
			const float4 l = B.blo[p], h = B.bhi[p];
			binCnt[b][tid]++;
			binLo[b * 3 + 0][tid] = fminf(binLo[b * 3 + 0][tid], l.x), binHi[b * 3 + 0][tid] = fmaxf(binHi[b * 3 + 0][tid], h.x);
			binLo[b * 3 + 1][tid] = fminf(binLo[b * 3 + 1][tid], l.y), binHi[b * 3 + 1][tid] = fmaxf(binHi[b * 3 + 1][tid], h.y);
			binLo[b * 3 + 2][tid] = fminf(binLo[b * 3 + 2][tid], l.z), binHi[b * 3 + 2][tid] = fmaxf(binHi[b * 3 + 2][tid], h.z);
		}
		__syncthreads();
		// columns -> one value per bin: wave w of the block folds rows w, w + WAVES, ...
		for (int row = (int)(tid >> 6); row < RT_BUILD_BINS * 3; row += RT_BUILD_WAVES) {
			float l = 1e30f, h = -1e30f;
			for (int t = (int)(tid & 63); t < RT_BUILD_THREADS; t += 64) l = fminf(l, binLo[row][t]), h = fmaxf(h, binHi[row][t]);
			l = wave_min(l), h = wave_max(h);
			if ((tid & 63) == 0) redLo[row] = l, redHi[row] = h;
		}
		for (int row = (int)(tid >> 6); row < RT_BUILD_BINS; row += RT_BUILD_WAVES) {
			int sN = 0;
			for (int t = (int)(tid & 63); t < RT_BUILD_THREADS; t += 64) sN += binCnt[row][t];
			sN = wave_sum(sN);
			if ((tid & 63) == 0) redCnt[row] = sN;
		}
		__syncthreads();
		if (tid == 0) {
			float bestCost = bestS[0], splitPos = bestS[1];
			int axis = axisS;
			float leftArea[RT_BUILD_BINS - 1], rightArea[RT_BUILD_BINS - 1];
			int leftCount[RT_BUILD_BINS - 1], rightCount[RT_BUILD_BINS - 1];
			SweepBox leftBox, rightBox;
			leftBox.reset(), rightBox.reset();
			int leftSum = 0, rightSum = 0;
			for (int i = 0; i < RT_BUILD_BINS - 1; i++) {
				leftSum += redCnt[i];
				leftCount[i] = leftSum;
				leftBox.grow(&redLo[i * 3], &redHi[i * 3]);
				leftArea[i] = leftBox.area();
				rightSum += redCnt[RT_BUILD_BINS - 1 - i];
				rightCount[RT_BUILD_BINS - 2 - i] = rightSum;
				rightBox.grow(&redLo[(RT_BUILD_BINS - 1 - i) * 3], &redHi[(RT_BUILD_BINS - 1 - i) * 3]);
				rightArea[RT_BUILD_BINS - 2 - i] = rightBox.area();
			}
			scale = (mx - mn) / RT_BUILD_BINS;
			for (int i = 0; i < RT_BUILD_BINS - 1; i++) {
				const float planeCost = leftCount[i] * leftArea[i] + rightCount[i] * rightArea[i];
				if (planeCost < bestCost) axis = a, splitPos = mn + scale * (i + 1), bestCost = planeCost;
			}
			bestS[0] = bestCost, bestS[1] = splitPos, axisS = axis;
		}
		__syncthreads();
	}
	__syncthreads();
	const float bestCost = bestS[0], splitPos = bestS[1];
	const int axis = axisS;
	// CalculateNodeCost (bvh.cpp:196-200)
	const float ex = node.hi[0] - node.lo[0], ey = node.hi[1] - node.lo[1], ez = node.hi[2] - node.lo[2];
	const float nosplitCost = count * (ex * ey + ey * ez + ez * ex);
	if (method == RT_SPLIT_BINNEDSAH ? bestCost >= nosplitCost : bestCost > 0.0f) return; // stays a leaf, primitiveIdx untouched (the other methods split unconditionally, or found no plane)

	// ---- the partition loop (bvh.cpp:296-313) in closed form ----
	auto isLeft = [&](uint p) { const float4 c4 = B.cen[p]; return (axis == 0 ? c4.x : (axis == 1 ? c4.y : c4.z)) < splitPos; };
	int nL = 0, part;
	for (uint i0 = 0; i0 < count; i0 += RT_BUILD_THREADS) {
		const uint i = i0 + tid;
		block_excl(i < count && isLeft(B.idx[first + i]) ? 1 : 0, shi, part);
		nL += part;
	}
	// holes, front to back
	int K = 0;
	for (uint i0 = 0; i0 < (uint)nL; i0 += RT_BUILD_THREADS) {
		const uint i = i0 + tid;
		const int hole = i < (uint)nL && !isLeft(B.idx[first + i]) ? 1 : 0;
		const int k = K + block_excl(hole, shi, part);
		if (hole) B.hpos[first + k] = i;
		K += part;
	}
	// fillers, back to front
	int F = 0;
	for (uint j0 = 0; j0 < count - (uint)nL; j0 += RT_BUILD_THREADS) {
		const uint j = j0 + tid; // distance from the right end
		const bool in = j < count - (uint)nL;
		const uint q = count - 1 - j;
		const int fill = in && isLeft(B.idx[first + q]) ? 1 : 0;
		const int k = F + block_excl(fill, shi, part);
		if (fill) B.fpos[first + k] = q;
		F += part;
	}
	__threadfence_block();
	__syncthreads(); // the hpos / fpos writes above are read by other threads below
	const uint fK = K > 0 ? B.fpos[first + K - 1] : count;
	int holesSeen = 0, fillersSeen = 0;
	for (uint i0 = 0; i0 < count; i0 += RT_BUILD_THREADS) {
		const uint i = i0 + tid;
		const bool in = i < count;
		const uint p = in ? B.idx[first + i] : 0;
		const bool L = in && isLeft(p);
		const int hole = in && i < (uint)nL && !L ? 1 : 0;
		const int fill = in && i >= (uint)nL && L ? 1 : 0;
		int holesHere, fillersHere;
		const int hk = holesSeen + block_excl(hole, shi, holesHere);
		const int fk = K - 1 - (fillersSeen + block_excl(fill, shi, fillersHere));
		if (in) {
			uint dest;
			if (i < (uint)nL) dest = L ? i : (hk == 0 ? count : B.fpos[first + hk - 1]) - 1;
			else if (L) dest = B.hpos[first + fk];
			else dest = i > fK ? i - 1 : (i == (uint)nL ? fK - 1 : i - 1);
			B.tmp[first + dest] = p;
		}
		holesSeen += holesHere, fillersSeen += fillersHere;
	}
	__threadfence_block();
	__syncthreads();
	for (uint i = tid; i < count; i += RT_BUILD_THREADS) B.idx[first + i] = B.tmp[first + i];
	__threadfence_block();
	__syncthreads();
	if (nL == 0 || nL == (int)count) return; // a leaf after all, with its primitiveIdx range permuted (bvh.cpp:315)

	// ---- children (bvh.cpp:317-330) ----
	float alo[3], ahi[3], blo[3], bhi[3];
	range_bounds(B, first, (uint)nL, shf, alo, ahi);
	range_bounds(B, first + (uint)nL, count - (uint)nL, shf, blo, bhi);
	if (tid == 0) {
		const int c = atomicAdd(&B.counters[0], 2);
		TNode& l = B.nodes[c];
		TNode& r = B.nodes[c + 1];
		for (int k = 0; k < 3; k++) l.lo[k] = alo[k], l.hi[k] = ahi[k], r.lo[k] = blo[k], r.hi[k] = bhi[k];
		l.first = first, l.count = (uint)nL, l.left = l.right = -1;
		r.first = first + (uint)nL, r.count = count - (uint)nL, r.left = r.right = -1;
		B.nodes[id].left = c, B.nodes[id].right = c + 1;
		const int o = atomicAdd(&B.counters[4 + ((level + 1) & 1)], 2);
		next[o] = c, next[o + 1] = c + 1;
	}
}

// ---- tlas::build (tlas.cpp:13-48) with tlas::FindBestMatch (:50-63): agglomerative clustering ------------------------
// At most 256 instances (the reference's nodeIdx[256]), and every step depends on the one before: ONE wave runs the
// reference's loop as it stands -- the scalar bookkeeping (A, B, C, nodeIdx[], nodesUsed) uniformly in every lane,
// FindBestMatch as a wave-wide "first minimum" over the up to 256 candidates (four per lane) -- so the nodes come
// out in the reference's order with the reference's boxes.
#define RT_TLAS_MAX 256
struct TlasNodeDev { float lo[3]; uint leftRight; float hi[3]; uint blas; };
__device__ __forceinline__ int tlas_best_match(const TlasNodeDev* node, const int* list, int n, int A)
{
	const uint lane = threadIdx.x & 63;
	const TlasNodeDev a = node[list[A]];
	float best = 1e30f;
	int bestB = 0x7fffffff; // "no candidate": smallest < 1e30f never held (the reference returns -1)
	for (int Bc = (int)lane; Bc < n; Bc += 64) {
		if (Bc == A) continue;
		const TlasNodeDev b = node[list[Bc]];
		const float ex = t_fmaxf(a.hi[0], b.hi[0]) - t_fminf(a.lo[0], b.lo[0]);
		const float ey = t_fmaxf(a.hi[1], b.hi[1]) - t_fminf(a.lo[1], b.lo[1]);
		const float ez = t_fmaxf(a.hi[2], b.hi[2]) - t_fminf(a.lo[2], b.lo[2]);
		const float area = ex * ey + ey * ez + ez * ex;
		if (area < best) best = area, bestB = Bc; // ascending within a lane: its first minimum
	}
	for (int o = 32; o > 0; o >>= 1) {
		const float k2 = __shfl_xor(best, o);
		const int i2 = __shfl_xor(bestB, o);
		if (k2 < best || (k2 == best && i2 < bestB)) best = k2, bestB = i2;
	}
	return bestB == 0x7fffffff ? -1 : bestB;
}
__global__ void __launch_bounds__(64) k_build_tlas(const float* bounds6, int n, TlasNodeDev* node, int* nodesUsedOut)
{
	__shared__ int nodeIdx[RT_TLAS_MAX];
	const uint lane = threadIdx.x;
	int nodesUsed = 1;
	for (int i = (int)lane; i < n; i += 64) {
		TlasNodeDev leaf;
		for (int k = 0; k < 3; k++) leaf.lo[k] = bounds6[6 * i + k], leaf.hi[k] = bounds6[6 * i + 3 + k];
		leaf.blas = (uint)i, leaf.leftRight = 0;
		node[1 + i] = leaf;
		nodeIdx[i] = 1 + i;
	}
	nodesUsed += n;
	__threadfence_block();
	__builtin_amdgcn_wave_barrier();
	int live = n, A = 0, Bi = tlas_best_match(node, nodeIdx, live, A);
	int guard = 0;
	while (live > 1 && Bi >= 0 && guard++ < 4 * RT_TLAS_MAX * RT_TLAS_MAX) {
		const int C = tlas_best_match(node, nodeIdx, live, Bi);
		if (C < 0) break;
		if (A != C) { A = Bi, Bi = C; continue; }
		const int ia = nodeIdx[A], ib = nodeIdx[Bi];
		if (lane == 0) {
			const TlasNodeDev na = node[ia], nb = node[ib];
			TlasNodeDev p;
			p.leftRight = (uint)ia + ((uint)ib << 16);
			p.blas = 0;
			for (int k = 0; k < 3; k++) p.lo[k] = t_fminf(na.lo[k], nb.lo[k]), p.hi[k] = t_fmaxf(na.hi[k], nb.hi[k]);
			node[nodesUsed] = p;
			nodeIdx[A] = nodesUsed;
			nodeIdx[Bi] = nodeIdx[live - 1];
		}
		nodesUsed++, live--;
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
		Bi = tlas_best_match(node, nodeIdx, live, A);
	}
	if (lane == 0) {
		node[0] = node[nodeIdx[A]];
		*nodesUsedOut = live == 1 ? nodesUsed : -1; // -1: no finite union area left to pick (the reference would index nodeIdx[-1])
	}
}

// between two levels: the list just consumed becomes the empty list the level after next appends to
__global__ void k_build_advance(BuildArrays B, int level) { B.counters[4 + (level & 1)] = 0; }

} // namespace rtd
