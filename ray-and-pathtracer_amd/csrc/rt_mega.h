// Renderer::Trace (Whitted, renderer.cpp:21-126) as ONE persistent launch per frame: the reference's interactive mode is one
// frame per Tick, and a frame of the wavefront (rt_kernels.h) is 16 rounds x (extend + connect + four streaming kernels)
// whose every traversal launch ends in a drain of 0.3-0.5 ms whatever it held (DESIGN.md finding 38): 8.7 of a Tick's 9 ms on
// the instanced glass / metal scene.  Here a lane keeps its pixel: when a query ends, the lane's flush runs the body of
// Trace at the hit right there (trace_persistent's advance hook) and goes on with the next query of the same pixel -- the
// refracted ray, a shadow ray of the light loop, the most recent pending branch -- until the pixel's tree is done; only
// then does it take another pixel from the queue.  One launch, one drain.  Traversal is the same lane-granular machine
// (nearest-hit and any-hit lanes side by side, MIXED); per-pixel state lives in [field][lane of the grid] arrays, the
// pending branches of glass (reflection waits while refraction runs) and of shiny diffuse hits on a per-lane stack.
// Per pixel the segments run in the reference's depth-first order and every term is added to the radiance in the order the
// wavefront kernels add it (shade: sky / light; light: W * direct after the light loop), so the frame is theirs bit for bit.
#pragma once
#include "rt_kernels.h"
#include "rt_stream.h" // ray_decided

namespace rtd {

struct MegaState { // [field][lane of the grid]
	float4* O;   // current segment's ray, world space: origin
	float4* D;   //                                     direction
	float4* W;   // path weight xyz, w = depth (int)
	float4* E;   // energy xyz, w = pending branches (int)
	float4* L;   // radiance xyz
	float4* hI;  // diffuse hit whose light loop is running: ray.IntersectionPoint() xyz, w = material
	float4* hN;  //   normal xyz, w = light being tested (int)
	float4* hA;  //   diffuse::scatter's value for that light xyz
	float4* hS;  //   direct light so far xyz
	float4* pend; // [lane][RT_PEND_CAP][4] {O,depth} {D,-} {W,-} {E,-}
	int lanes;
	// longest first (k_mega_hist / k_mega_order below): how long a lane held each sample of the batch (s_memrealtime ticks, 10 ns),
	// written when the sample is stored, and the order the NEXT launch over the same samples deals its tiles out in
	uint* cost;        // [sample of the batch]; null: not recorded
	const uint* order; // [tile position] -> tile; null: the multiplicative permutation
	int nWork;         // work items of the launch (tile positions << permShift)
};

// ---- longest first ---------------------------------------------------------------------------------------------------
// A launch ends when its longest pixel does: on the instanced glass / metal scene the queue is dry after 0.85 ms and the last
// lane leaves 4.4 ms later (profiles/r03_mega_tail.txt) -- the tree of ONE glass pixel, walked by one lane, whenever it
// started.  An interactive loop renders the same pixels Tick after Tick, so the time a lane held a pixel in the last Tick says
// how long it will hold it in this one: tiles are dealt out by descending cost, round robin over the traversal's RT_HEADS
// sub-queues, so the long trees start at time 0 and the queue ends with sky.  Any order gives the same frame (a pixel's value
// does not depend on when it is traced); the order inside a cost class is whatever the atomics make it.
#define RT_MEGA_BUCKETS 64
#define RT_MEGA_ORDER_BLOCK 1024
__device__ __forceinline__ int mega_bucket(const uint* cost, uint tile, uint shift, uint nSamples)
{
	uint c = 0;
	for (uint k = 0; k < (1u << shift); k++) {
		const uint w = (tile << shift) + k;
		if (w < nSamples) c += cost[w];
	}
	if (c == 0) return RT_MEGA_BUCKETS - 1; // nothing known, or a tile beyond the last sample
	const int e = 31 - __clz((int)c);
	const int k = 4 * e + (e >= 2 ? (int)((c >> (e - 2)) & 3) : 0); // quarter octaves
	const int b = 95 - k;
	return b < 0 ? 0 : (b > RT_MEGA_BUCKETS - 1 ? RT_MEGA_BUCKETS - 1 : b); // class 0: the most expensive tiles
}
__global__ void __launch_bounds__(RT_MEGA_ORDER_BLOCK) k_mega_hist(const uint* cost, uint shift, uint nSamples, uint nTilesPad, uint* hist)
{
	__shared__ uint h[RT_MEGA_BUCKETS];
	if (threadIdx.x < RT_MEGA_BUCKETS) h[threadIdx.x] = 0;
	__syncthreads();
	const uint t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nTilesPad) atomicAdd(&h[mega_bucket(cost, t, shift, nSamples)], 1u);
	__syncthreads();
	if (threadIdx.x < RT_MEGA_BUCKETS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// hist[0 .. 64): tiles per class; hist[64 .. 128): tiles of the class placed so far.  nTilesPad is a multiple of RT_HEADS * 8, so
// that the traversal's sub-queues (trace_persistent: n / RT_HEADS entries each, a multiple of 64) are equally long.
__global__ void __launch_bounds__(RT_MEGA_ORDER_BLOCK) k_mega_order(const uint* cost, uint shift, uint nSamples, uint nTilesPad, uint* hist, uint* order)
{
	__shared__ uint h[RT_MEGA_BUCKETS], base[RT_MEGA_BUCKETS];
	if (threadIdx.x < RT_MEGA_BUCKETS) h[threadIdx.x] = 0;
	__syncthreads();
	const uint t = blockIdx.x * blockDim.x + threadIdx.x;
	int b = 0;
	uint mine = 0;
	if (t < nTilesPad) b = mega_bucket(cost, t, shift, nSamples), mine = atomicAdd(&h[b], 1u);
	__syncthreads();
	if (threadIdx.x < RT_MEGA_BUCKETS) {
		uint before = 0;
		for (int k = 0; k < (int)threadIdx.x; k++) before += hist[k];
		base[threadIdx.x] = before + (h[threadIdx.x] ? atomicAdd(&hist[RT_MEGA_BUCKETS + threadIdx.x], h[threadIdx.x]) : 0u);
	}
	__syncthreads();
	if (t < nTilesPad) {
		// groups of eight tiles (64 pixels: what a wave takes at a time), one tile of every eighth of the cost ranking in each,
		// every eighth in descending order along the queue: the longest trees start first, and no wave is full of them -- such a
		// wave waits at every flush for half of its lanes to finish their queries (measured: packed by cost, the Tick takes twice
		// as long as unsorted)
		const uint rank = base[b] + mine, groups = nTilesPad / 8;
		const uint r = rank % groups, slot = rank / groups;
		order[8 * ((r % RT_HEADS) * (groups / RT_HEADS) + r / RT_HEADS) + slot] = t;
	}
}

struct WhittedMegaPolicy {
	static constexpr bool kAdvance = true;
	static constexpr int kRefill = 32;
	const DScene& S;
	const DCamera& C;
	const RenderParams& R;
	const MegaState& M;
	int gl;     // this lane's column in M
	int* flag;

	__device__ __forceinline__ bool any_of(int) const { return false; } // a pixel starts with Scene::FindNearest
	// Work item -> sample.  Consecutive work items are handed to the lanes of one wave, and a pixel's cost is its tree: 1 segment
	// for the sky, up to 15 glass segments of ~50 steps each for a pixel on a glass mesh.  In pixel order the waves that draw a
	// piece of the glass region run long after the others have left (measured: a Tick of the instanced glass / metal scene 7.7 ms
	// for ~2 ms of work).  So TILES of 64 consecutive pixels are dealt out in a multiplicative permutation (tile * P mod nTiles, P
	// coprime to nTiles and about nTiles / 61): a wave's lanes still hold neighbouring pixels (coherent primary rays: permuting
	// single pixels costs the scene-BVH test scene 13 %), but consecutive chunks of the queue come from all over the frame.
	// Work items beyond the last sample (the last tile may be short) are nothing to trace.
	__device__ __forceinline__ bool sample_of(int work, uint& sid) const
	{
		uint w = (uint)work;
		if (M.order) w = (M.order[w >> R.permShift] << R.permShift) + (w & ((1u << R.permShift) - 1));
		else if (R.permMul) {
			const uint sh = R.permShift, nTiles = (R.nSamples + (1u << sh) - 1) >> sh;
			w = ((uint)(((unsigned long long)(w >> sh) * R.permMul) % nTiles) << sh) + (w & ((1u << sh) - 1));
		}
		sid = R.sampleFirst + w;
		return w < R.nSamples;
	}
	// head tests of Scene::FindNearest on a new nearest-hit ray (as emit_ray); the ray itself goes to M for the shading at its hit
	__device__ __forceinline__ void new_segment(const f3& O, const f3& D, float& tmax, HitRef& head) const
	{
		float rayT = 1e34f;
		head.kind = -1, head.inst = -1, head.prim = 0, head.t = 0;
		LaneCounters unused;
		find_nearest_head<false>(S, O, D, (float)1e-6, rayT, head, unused); // renderer.cpp:24
		M.O[gl] = mk4(O, 0.0f), M.D[gl] = mk4(D, 0.0f);
		tmax = rayT;
	}
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef& head) const
	{
		uint sid;
		if (!sample_of(work, sid)) return false;
		f3 E(1.0f);
		if (R.customO) {
			O = f3(R.customO[3 * sid], R.customO[3 * sid + 1], R.customO[3 * sid + 2]);
			D = f3(R.customD[3 * sid], R.customD[3 * sid + 1], R.customD[3 * sid + 2]);
			E = f3(R.customE[0], R.customE[1], R.customE[2]);
		} else {
			uint seed;
			sample_primary(C, R, sid, O, D, seed);
		}
		M.W[gl] = make_float4(1, 1, 1, __int_as_float(start_depth(R)));
		M.E[gl] = mk4(E, __int_as_float(0));
		M.L[gl] = make_float4(0, 0, 0, __uint_as_float((uint)__builtin_amdgcn_s_memrealtime())); // w: when the lane took the sample
		new_segment(O, D, tmax, head);
		return true;
	}
	__device__ __forceinline__ void push(int& np, const f3& O, const f3& D, const f3& W, const f3& E, int depth) const
	{
		if (np >= RT_PEND_CAP) { *flag = 2; return; }
		float4* e = M.pend + ((size_t)gl * RT_PEND_CAP + np) * 4;
		e[0] = mk4(O, __int_as_float(depth)), e[1] = mk4(D, 0.0f), e[2] = mk4(W, 0.0f), e[3] = mk4(E, 0.0f);
		np++;
	}
	// the shadow ray of light i for the diffuse hit kept in M (renderer.cpp:93-99; as ConnectPolicy::load); false: no light left
	__device__ __forceinline__ bool light_step(int i, const f3& I, const f3& normal, const f3& rayD, const DMaterial& m, f3& E, int np, const f3& direct,
	                                           f3& O, f3& D, float& tmax, bool& nextAny) const
	{
		if (i >= S.nLights) return false;
		uint unusedSeed = 0;
		const f3 pickedPos = light_position(S.lights[i], true, unusedSeed);
		f3 dir = pickedPos - I;
		const float len2 = dot(dir, dir);
		dir = normalize(dir);
		// scatter first: the energy changes even when the light turns out to be occluded (renderer.cpp:95-96)
		const f3 att = diffuse_scatter(m, rayD, dir, light_intensity(S.lights[i], I, normal, pickedPos), normal, E);
		M.E[gl] = mk4(E, __int_as_float(np));
		M.hN[gl] = mk4(normal, __int_as_float(i));
		M.hA[gl] = mk4(att, 0.0f);
		M.hS[gl] = mk4(direct, 0.0f);
		O = I + dir * 1e-4f, D = dir, tmax = sqrtf(len2), nextAny = true;
		return true;
	}
	__device__ __forceinline__ bool advance_once(int work, bool wasAny, const HitRef& res, f3& O, f3& D, float& tmax, HitRef& head, bool& nextAny) const
	{
		const float4 w4 = M.W[gl], e4 = M.E[gl], l4 = M.L[gl];
		f3 W = xyz(w4), E = xyz(e4), Lsum = xyz(l4);
		const int depth = __float_as_int(w4.w);
		int np = __float_as_int(e4.w);
		const f3 rayO = xyz(M.O[gl]), rayD = xyz(M.D[gl]);
		bool segmentEnds = true;
		nextAny = false;
		if (!wasAny) {
			// ---- Trace at the hit of this segment (renderer.cpp:25-122; k_shade's Whitted branch) ----
			int objIdx, matId;
			f3 normal;
			resolve_hit(S, res, rayO, rayD, objIdx, matId, normal);
			const float t = res.t;
			const f3 I = rayO + t * rayD;
			const int nDepth = depth - 1;
			const bool childTraces = nDepth > 0; // Trace(depth <= 0) = 0 (renderer.cpp:23)
			f3 nO(0.0f), nD(0.0f), nW(0.0f);
			if (objIdx == -1) Lsum = Lsum + W * sky_color(S, rayD);
			else if (objIdx >= 11 && objIdx < 11 + S.nLights) Lsum = Lsum + W * light_intensity(S.lights[objIdx - 11], I, normal, I);
			else {
				const DMaterial m = S.mats[matId];
				const f3 col(m.col[0], m.col[1], m.col[2]);
				if (m.type == 3) { // GLASS, renderer.cpp:45-80: refraction now, reflection waits
					const float kr = glass_fresnel(normalize(rayD), normalize(normal), m.ir);
					const bool outside = dot(rayD, normal) < 0;
					const f3 bias = 0.0001f * normal;
					const f3 norm = outside ? normal : -normal;
					const float r = !outside ? m.ir : (1 / m.ir);
					if (outside) {
						E.x *= x_expf(m.absorption[0] * -t);
						E.y *= x_expf(m.absorption[1] * -t);
						E.z *= x_expf(m.absorption[2] * -t);
					}
					const bool takeRefr = kr < 1;
					f3 refrO(0.0f), refrD(0.0f), refrW(0.0f);
					if (takeRefr) {
						refrD = normalize(glass_refract(rayD, norm, r));
						refrO = outside ? I - bias : I + bias;
						const f3 tempCol = col * E;
						refrW = W * (tempCol * (1 - kr));
					}
					const f3 reflD = normalize(reflect(rayD, norm));
					const f3 reflO = outside ? I + bias : I - bias;
					const f3 reflW = W * (col * kr);
					if (childTraces) {
						if (takeRefr) {
							push(np, reflO, reflD, reflW, E, nDepth);
							nO = refrO, nD = refrD, nW = refrW;
						} else nO = reflO, nD = reflD, nW = reflW;
						segmentEnds = false;
					}
				} else if (m.type == 2) { // METAL, renderer.cpp:81-86
					nO = I + normal * 0.001f, nD = reflect(rayD, normal);
					nW = W * (col * E);
					if (childTraces) segmentEnds = false;
				} else if (S.nLights > 0) { // DIFFUSE, renderer.cpp:87-122: the light loop, one shadow query at a time
					M.hI[gl] = mk4(I, __int_as_float(matId));
					M.L[gl] = mk4(Lsum, l4.w);
					if (light_step(0, I, normal, rayD, m, E, np, f3(0.0f), O, D, tmax, nextAny)) return true;
				}
			}
			if (!segmentEnds) {
				M.W[gl] = mk4(nW, __int_as_float(nDepth));
				M.E[gl] = mk4(E, __int_as_float(np));
				M.L[gl] = mk4(Lsum, l4.w);
				O = nO, D = nD;
				new_segment(O, D, tmax, head);
				return true;
			}
		} else {
			// ---- the occlusion answer for light i of the diffuse hit (k_light's Whitted branch) ----
			const float4 i4 = M.hI[gl], n4 = M.hN[gl], a4 = M.hA[gl], s4 = M.hS[gl];
			const f3 I = xyz(i4), normal = xyz(n4), att = xyz(a4);
			f3 direct = xyz(s4);
			const int i = __float_as_int(n4.w);
			const DMaterial m = S.mats[__float_as_int(i4.w)];
			const f3 col(m.col[0], m.col[1], m.col[2]);
			if (res.kind != 1) { // visible
				if (m.shinieness != 0 && depth - 1 > 0) // renderer.cpp:101-102: a mirror branch per visible light
					push(np, I, reflect(rayD, normal), W * ((m.shinieness * col) * E), E, depth - 1);
				direct = direct + (1 - m.shinieness) * col * att * E;
			}
			if (light_step(i + 1, I, normal, rayD, m, E, np, direct, O, D, tmax, nextAny)) return true;
			Lsum = Lsum + W * direct;
		}
		// ---- the segment is over: the most recent pending branch, or the pixel is done (k_finish) ----
		if (np > 0) {
			np--;
			const float4* pe = M.pend + ((size_t)gl * RT_PEND_CAP + np) * 4;
			const float4 o = pe[0], d = pe[1], w = pe[2], en = pe[3];
			M.W[gl] = make_float4(w.x, w.y, w.z, o.w);
			M.E[gl] = make_float4(en.x, en.y, en.z, __int_as_float(np));
			M.L[gl] = mk4(Lsum, l4.w);
			O = xyz(o), D = xyz(d);
			new_segment(O, D, tmax, head);
			return true;
		}
		uint sid;
		sample_of(work, sid);
		store_sample(R, sid, Lsum);
		if (M.cost) M.cost[sid - R.sampleFirst] = (uint)__builtin_amdgcn_s_memrealtime() - __float_as_uint(l4.w);
		return false;
	}
	// The flush: the body of Trace for the query that ended.  (Answering the pixel's further queries here when their first
	// traversal step would leave nothing to visit was measured level -- the launch is as long as its longest pixel -- and taken
	// out: profiles/patches/mega_decide_and_path_mega.diff.)
	__device__ __forceinline__ bool advance(int work, bool wasAny, const HitRef& res, f3& O, f3& D, float& tmax, HitRef& head, bool& nextAny) const
	{
		return advance_once(work, wasAny, res, O, D, tmax, head, nextAny);
	}
};

// ---- Renderer::Trace by tree levels ------------------------------------------------------------------------------------
// The single launch above is as long as its longest pixel: one lane walks the whole tree of a glass pixel, up to 15 nearest-hit
// queries and their light loops one after the other (4.4 of 5.3 ms on the bench scene, profiles/r03_mega_tail.txt).  The branches
// of a tree are independent given what Trace hands them (ray, weight, energy, depth); what ties them to one lane is the ORDER in
// which their terms enter the pixel's radiance, which is part of the result (float addition).  So: one launch per tree LEVEL.  A
// work item is a SEGMENT (level 0: the camera ray of a sample; level l + 1: the children the segments of level l appended to a
// queue); the lane that takes it runs the body of Trace at its hit as above -- the light loop of a diffuse hit stays with the
// lane, one shadow query at a time -- but hands every child to the next level's queue instead of walking it, and LOGS the one
// term a segment can add (sky / light / W * direct) under the segment's key instead of adding it.  The key is the segment's
// path in the tree, six bits per level, with the digits chosen so that numeric order is the depth-first order the reference's
// recursion adds in: glass: refraction 1, reflection 2 (renderer.cpp:61-79); metal: 1; a diffuse hit's own term sorts first (its
// digit is 0) and its mirror branches come LAST LIGHT FIRST (the single launch pops them off a stack), light i = 1 + (n - 1 - i).
// k_whitted_reduce then adds each sample's terms in key order, from 0, as the recursion would have: same bits.
// A frame is depth launches of at most 1 + lights queries per lane and a streaming pass, instead of one launch of up to 31.
#define RT_LEVEL_MAX 10   // levels (6 bits each in a 64-bit key); deeper Trace calls keep the single launch
#define RT_LEVEL_CHUNK 32 // queue slots a wave reserves at a time (one atomic per chunk, not per child)
struct LevelState {
	float4* seg[2];              // segment queues by level parity, 4 float4 per segment: {O, depth} {D, sample} {W, key lo} {E, key hi}
	int* count;                  // [RT_LEVEL_MAX + 2] slots reserved in the queue of each level (level 0: unused)
	unsigned long long* termKey; // [level][cap]
	float4* termVal;             // [level][cap] xyz, w = the sample's term logged before this one (int, -1: none)
	int* head;                   // [sample of the batch] the sample's most recent term, -1: none
	int cap;                     // segments per queue = terms per level
	int qcap;                    // queue slots in use (= cap; tests make it smaller: a queue that overflows)
	int level;
};
typedef __attribute__((address_space(3))) int lds_int;

struct WhittedLevelPolicy {
	static constexpr bool kAdvance = true;
	static constexpr int kRefill = 32;
	const DScene& S;
	const DCamera& C;
	const RenderParams& R;
	const MegaState& M;
	const LevelState& V;
	int gl;
	int* flag;
	lds_int* res; // this wave's reservation in the next level's queue: [0] next slot, [1] end

	__device__ __forceinline__ bool any_of(int) const { return false; }
	__device__ __forceinline__ bool sample_of(int work, uint& sid) const
	{
		uint w = (uint)work;
		if (M.order) w = (M.order[w >> R.permShift] << R.permShift) + (w & ((1u << R.permShift) - 1)); // longest first, as the single launch
		else if (R.permMul) {
			const uint sh = R.permShift, nTiles = (R.nSamples + (1u << sh) - 1) >> sh;
			w = ((uint)(((unsigned long long)(w >> sh) * R.permMul) % nTiles) << sh) + (w & ((1u << sh) - 1));
		}
		sid = R.sampleFirst + w;
		return w < R.nSamples;
	}
	__device__ __forceinline__ void new_segment(const f3& O, const f3& D, float& tmax, HitRef& head) const
	{
		float rayT = 1e34f;
		head.kind = -1, head.inst = -1, head.prim = 0, head.t = 0;
		LaneCounters unused;
		find_nearest_head<false>(S, O, D, (float)1e-6, rayT, head, unused); // renderer.cpp:24
		M.O[gl] = mk4(O, 0.0f), M.D[gl] = mk4(D, 0.0f);
		tmax = rayT;
	}
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef& head) const
	{
		if (V.level == 0) {
			uint sid, seed;
			if (!sample_of(work, sid)) return false;
			sample_primary(C, R, sid, O, D, seed);
			M.W[gl] = make_float4(1, 1, 1, __int_as_float(start_depth(R)));
			M.E[gl] = make_float4(1, 1, 1, 0);
			M.L[gl] = make_float4(0, 0, __uint_as_float(sid), __uint_as_float((uint)__builtin_amdgcn_s_memrealtime())); // key 0; w: when the lane took the sample
		} else {
			const float4* sg = V.seg[V.level & 1] + 4 * (size_t)work;
			const float4 s0 = sg[0], s1 = sg[1], s2 = sg[2], s3 = sg[3];
			if (__float_as_int(s0.w) < 0) return false; // a slot its wave reserved and did not fill
			O = xyz(s0), D = xyz(s1);
			M.W[gl] = make_float4(s2.x, s2.y, s2.z, s0.w);
			M.E[gl] = make_float4(s3.x, s3.y, s3.z, 0);
			M.L[gl] = make_float4(s2.w, s3.w, s1.w, 0);
		}
		new_segment(O, D, tmax, head);
		return true;
	}
	// the one term of this segment, under its key, at the front of its sample's list
	__device__ __forceinline__ void log_term(int work, unsigned long long key, uint sid, const f3& v) const
	{
		const size_t id = (size_t)V.level * (size_t)V.cap + (size_t)work;
		V.termKey[id] = key;
		const int before = atomicExch(&V.head[sid - R.sampleFirst], (int)id);
		V.termVal[id] = mk4(v, __int_as_float(before));
	}
	struct Child { bool want; f3 O, D, W, E; int depth; uint digit; };
	// the children of the lanes that are here together go to the next level's queue: one reservation for all of them
	__device__ __forceinline__ void append(const Child& a, const Child& b, unsigned long long key, uint sid) const
	{
		const uint lane = threadIdx.x & 63;
		const unsigned long long below = (1ull << lane) - 1;
		const unsigned long long ma = __ballot(a.want), mb = __ballot(b.want);
		const int na = __popcll(ma), cnt = na + __popcll(mb);
		if (cnt == 0) return;
		const int leader = __builtin_ctzll(__ballot(true));
		int next = 0, end = 0, fresh = 0;
		if ((int)lane == leader) {
			next = res[0], end = res[1];
			const int rem = end - next;
			if (cnt > rem) {
				const int need = (cnt - rem + RT_LEVEL_CHUNK - 1) / RT_LEVEL_CHUNK * RT_LEVEL_CHUNK;
				fresh = atomicAdd(&V.count[V.level + 1], need);
				res[0] = fresh + (cnt - rem), res[1] = fresh + need;
			} else res[0] = next + cnt;
		}
		next = __builtin_amdgcn_readlane(next, leader), end = __builtin_amdgcn_readlane(end, leader), fresh = __builtin_amdgcn_readlane(fresh, leader);
		const int rem = end - next;
		const int shift = 58 - 6 * V.level; // the child's digit: level l + 1 of the key
		for (int k = 0; k < 2; k++) {
			const Child& ch = k ? b : a;
			if (!ch.want) continue;
			const int i = k ? na + __popcll(mb & below) : __popcll(ma & below);
			const int slot = i < rem ? next + i : fresh + (i - rem);
			if (slot >= V.qcap) { *flag = 3; continue; } // the host repeats the frame as one launch
			const unsigned long long ck = key | ((unsigned long long)ch.digit << shift);
			float4* sg = V.seg[(V.level + 1) & 1] + 4 * (size_t)slot;
			sg[0] = mk4(ch.O, __int_as_float(ch.depth)), sg[1] = mk4(ch.D, __uint_as_float(sid));
			sg[2] = mk4(ch.W, __uint_as_float((uint)ck)), sg[3] = mk4(ch.E, __uint_as_float((uint)(ck >> 32)));
		}
	}
	// the shadow ray of light i for the diffuse hit kept in M (renderer.cpp:93-99); false: no light left
	__device__ __forceinline__ bool light_step(int i, const f3& I, const f3& normal, const f3& rayD, const DMaterial& m, f3& E, const f3& direct,
	                                           f3& O, f3& D, float& tmax, bool& nextAny) const
	{
		if (i >= S.nLights) return false;
		uint unusedSeed = 0;
		const f3 pickedPos = light_position(S.lights[i], true, unusedSeed);
		f3 dir = pickedPos - I;
		const float len2 = dot(dir, dir);
		dir = normalize(dir);
		const f3 att = diffuse_scatter(m, rayD, dir, light_intensity(S.lights[i], I, normal, pickedPos), normal, E); // before the query: E changes either way (:95-96)
		M.E[gl] = mk4(E, 0.0f);
		M.hN[gl] = mk4(normal, __int_as_float(i));
		M.hA[gl] = mk4(att, 0.0f);
		M.hS[gl] = mk4(direct, 0.0f);
		O = I + dir * 1e-4f, D = dir, tmax = sqrtf(len2), nextAny = true;
		return true;
	}
	__device__ __forceinline__ bool advance(int work, bool wasAny, const HitRef& res, f3& O, f3& D, float& tmax, HitRef& head, bool& nextAny) const
	{
		if (advance_once(work, wasAny, res, O, D, tmax, head, nextAny)) return true;
		// level 0: how long the lane held the camera ray and its light loop -- the next frame's order (k_mega_order)
		if (V.level == 0 && M.cost) {
			const float4 l4 = M.L[gl];
			M.cost[__float_as_uint(l4.z) - R.sampleFirst] = (uint)__builtin_amdgcn_s_memrealtime() - __float_as_uint(l4.w);
		}
		return false;
	}
	__device__ __forceinline__ bool advance_once(int work, bool wasAny, const HitRef& res0, f3& O, f3& D, float& tmax, HitRef& head, bool& nextAny) const
	{
		const float4 w4 = M.W[gl], e4 = M.E[gl], l4 = M.L[gl];
		const f3 W = xyz(w4);
		f3 E = xyz(e4);
		const int depth = __float_as_int(w4.w);
		const unsigned long long key = (unsigned long long)__float_as_uint(l4.x) | ((unsigned long long)__float_as_uint(l4.y) << 32);
		const uint sid = __float_as_uint(l4.z);
		const f3 rayO = xyz(M.O[gl]), rayD = xyz(M.D[gl]);
		Child a, b;
		a.want = b.want = false, a.depth = b.depth = 0, a.digit = b.digit = 0;
		bool go = false; // the lane goes on with a shadow query of this segment
		nextAny = false;
		if (!wasAny) {
			int objIdx, matId;
			f3 normal;
			resolve_hit(S, res0, rayO, rayD, objIdx, matId, normal);
			const float t = res0.t;
			const f3 I = rayO + t * rayD;
			const int nDepth = depth - 1;
			const bool childTraces = nDepth > 0; // Trace(depth <= 0) = 0 (renderer.cpp:23)
			if (objIdx == -1) log_term(work, key, sid, W * sky_color(S, rayD));
			else if (objIdx >= 11 && objIdx < 11 + S.nLights) log_term(work, key, sid, W * light_intensity(S.lights[objIdx - 11], I, normal, I));
			else {
				const DMaterial m = S.mats[matId];
				const f3 col(m.col[0], m.col[1], m.col[2]);
				if (m.type == 3) { // GLASS, renderer.cpp:45-80
					const float kr = glass_fresnel(normalize(rayD), normalize(normal), m.ir);
					const bool outside = dot(rayD, normal) < 0;
					const f3 bias = 0.0001f * normal;
					const f3 norm = outside ? normal : -normal;
					const float r = !outside ? m.ir : (1 / m.ir);
					if (outside) {
						E.x *= x_expf(m.absorption[0] * -t);
						E.y *= x_expf(m.absorption[1] * -t);
						E.z *= x_expf(m.absorption[2] * -t);
					}
					if (childTraces) {
						if (kr < 1) {
							a.want = true, a.digit = 1, a.depth = nDepth, a.E = E;
							a.D = normalize(glass_refract(rayD, norm, r));
							a.O = outside ? I - bias : I + bias;
							const f3 tempCol = col * E;
							a.W = W * (tempCol * (1 - kr));
						}
						b.want = true, b.digit = 2, b.depth = nDepth, b.E = E;
						b.D = normalize(reflect(rayD, norm));
						b.O = outside ? I + bias : I - bias;
						b.W = W * (col * kr);
					}
				} else if (m.type == 2) { // METAL, renderer.cpp:81-86
					if (childTraces) {
						a.want = true, a.digit = 1, a.depth = nDepth, a.E = E;
						a.O = I + normal * 0.001f, a.D = reflect(rayD, normal);
						a.W = W * (col * E);
					}
				} else if (S.nLights > 0) { // DIFFUSE, renderer.cpp:87-122
					M.hI[gl] = mk4(I, __int_as_float(matId));
					go = light_step(0, I, normal, rayD, m, E, f3(0.0f), O, D, tmax, nextAny);
				}
			}
		} else {
			// the occlusion answer for light i of the diffuse hit
			const float4 i4 = M.hI[gl], n4 = M.hN[gl], a4 = M.hA[gl], s4 = M.hS[gl];
			const f3 I = xyz(i4), normal = xyz(n4), att = xyz(a4);
			f3 direct = xyz(s4);
			const int i = __float_as_int(n4.w);
			const DMaterial m = S.mats[__float_as_int(i4.w)];
			const f3 col(m.col[0], m.col[1], m.col[2]);
			if (res0.kind != 1) { // visible
				if (m.shinieness != 0 && depth - 1 > 0) { // renderer.cpp:101-102: a mirror branch per visible light
					a.want = true, a.digit = (uint)(1 + (S.nLights - 1 - i)), a.depth = depth - 1, a.E = E;
					a.O = I, a.D = reflect(rayD, normal);
					a.W = W * ((m.shinieness * col) * E);
				}
				direct = direct + (1 - m.shinieness) * col * att * E;
			}
			go = light_step(i + 1, I, normal, rayD, m, E, direct, O, D, tmax, nextAny);
			if (!go) log_term(work, key, sid, W * direct);
		}
		append(a, b, key, sid);
		return go;
	}
};

#ifndef RT_MEGA_WAVES
#define RT_MEGA_WAVES 4 // measured: 3 waves (no spill) and 5 are slower, profiles/r03_tick_mega.txt
#endif
__global__ void __launch_bounds__(RT_BLOCK, RT_MEGA_WAVES) k_whitted_mega(DScene S, DCamera C, RenderParams R, MegaState M, int tuning, uint* spill, int* work)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	WhittedMegaPolicy pol{ S, C, R, M, (int)(blockIdx.x * blockDim.x + threadIdx.x), &work[1] };
	trace_persistent<false, false, false, WhittedMegaPolicy, true>(S, pol, M.nWork, work + 16, 0.0f, tuning, ldsStack, spill, &work[1], lc, rays);
}

__global__ void __launch_bounds__(RT_BLOCK, RT_MEGA_WAVES) k_whitted_level(DScene S, DCamera C, RenderParams R, MegaState M, LevelState V, int tuning, uint* spill, int* work)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	__shared__ int reservation[RT_BLOCK / 64][2];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	lds_int* res = (lds_int*)&reservation[threadIdx.x >> 6][0];
	if ((threadIdx.x & 63) == 0) res[0] = 0, res[1] = 0;
	WhittedLevelPolicy pol{ S, C, R, M, V, (int)(blockIdx.x * blockDim.x + threadIdx.x), &work[1], res };
	const int n = V.level == 0 ? M.nWork : (V.count[V.level] < V.qcap ? V.count[V.level] : V.qcap);
	trace_persistent<false, false, false, WhittedLevelPolicy, true>(S, pol, n, work + 16, 0.0f, tuning, ldsStack, spill, &work[1], lc, rays);
	// the slots this wave reserved and did not fill are nothing to trace
	const int lo = res[0] + (int)(threadIdx.x & 63), hi = res[1];
	if (lo < hi && lo < V.qcap) V.seg[(V.level + 1) & 1][4 * (size_t)lo] = make_float4(0, 0, 0, __int_as_float(-1));
}
// every sample's terms in key order, from 0 (the order Trace's recursion adds them in)
__global__ void k_whitted_reduce(RenderParams R, LevelState V)
{
	const uint s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= R.nSamples) return;
	f3 L(0.0f);
	unsigned long long last = 0;
	bool first = true;
	for (;;) {
		int best = -1;
		unsigned long long bestKey = ~0ull;
		for (int id = V.head[s]; id >= 0; id = __float_as_int(V.termVal[id].w)) {
			const unsigned long long k = V.termKey[id];
			if ((first || k > last) && k <= bestKey) best = id, bestKey = k;
		}
		if (best < 0) break;
		L = L + xyz(V.termVal[best]);
		last = bestKey, first = false;
	}
	store_sample(R, R.sampleFirst + s, L);
}

} // namespace rtd
