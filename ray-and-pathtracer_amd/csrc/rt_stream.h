// The path integrator's wavefront in DENSE form (Renderer::Sample, renderer.cpp:128-236, for batches that give every
// sample an entry of its own: the bench workloads, Renderer::Tick in path mode, rt_trace_batch).
//
// rt_kernels.h keeps one slot per sample for the whole batch and a status byte per slot; after the first bounce most
// slots are dead (the bench frame: 50 % after round 0, 87 % after round 1), so every later kernel reads and writes
// 16-byte elements scattered over mostly dead 128-byte lines -- measured round 2: k_shade moves its ~250 B per live slot
// at 0.42 of the HBM peak.  Compacting afterwards would pay for the same sparse lines once more.  Here nothing is ever
// written sparse: the entries of round r + 1 are the survivors of round r in order, and every producer writes its
// output straight to the position the survivor will have.  That position has to be known before shade runs, and it
// is: whether a path continues is decided by WHAT it hit (miss / light: no; a surface: yes until the last round; a
// diffuse surface also gets a shadow record), so the kernel that finds the hit -- extend, or the producer of a ray whose
// hit it could decide itself, see ray_decided() -- leaves a class byte per entry, and one scan (k_assign) turns the
// class bytes into positions.  Per round:
//   compact   cls -> queue of the entries that still need a traversal                        (1 B per entry)
//   extend    Scene::FindNearest for those; hit record + class byte per entry                (persistent, the dominant kernel)
//   assign    exclusive scans of the CONT and SHADOW class bits -> pos[e / 64] = (position in round r + 1, shadow position) of the group's first entry
//   shade     the body of Sample at the hit: reads entry e, writes the continuation (next ray, W, E, L) at its
//             position in the other buffer of a ping-pong pair and the shadow record at its shadow position; a path
//             that ends stores its finished sample.  Nothing else is written.
//   connect   Scene::IsOccluded per (shadow record, light)                                    } on a second stream,
//   light     the direct terms of the shadow records, in light order; updates E and L of the   } beside the next
//             continuation entry (or stores the sample in the last round)                       } round's extend
// Path depth is the same for every entry of a round (depth = start depth - round), so "last round" is a launch
// argument and W.w is free.  The order of entries is the order of samples (frame-major, then pixels) with holes
// closed: neighbours stay neighbours, which is all the coherence the later rounds have.
//
// Producer-side decisions (ray_decided): a kernel that creates a ray (generate, shade) already runs the head tests of
// Scene::FindNearest on it (lights, brute-force primitives).  It now also makes the FIRST step extend would make -- the
// root pair of the TLAS with its reach boxes, or the root pair of the scene BVH -- and when that step leaves nothing to
// visit the query is answered: the hit is the head candidate.  Such an entry gets its hit record and class from the
// producer and never enters the traversal queue; a camera ray that leaves the scene this way (sky or a light) is
// finished inside generate and never becomes an entry at all.  On the bench frame two thirds of the camera rays and
// 59 % of the first-bounce rays are of this kind (profiles/r02_step_histogram_spp8.txt: "0 instance entries"); each used
// to cost a queue slot, a refill in the 42 %-lane traversal kernel, and a ray load + hit store at random addresses.
// The step is the one trace_persistent makes, on the same operands, so results are identical by construction.
#pragma once
#include "rt_kernels.h"
#include "rt_qlearn.h"

namespace rtd {

#define CL_TRACE 1  // the entry's ray still needs Scene::FindNearest's traversal (extend)
#define CL_LIVE 2   // the entry holds a path (round 0 only: entries of samples finished inside generate are not live)
#define CL_CONT 4   // the path continues: shade writes a continuation entry at pos.x
#define CL_SHADOW 8 // diffuse hit with lights: shade writes a shadow record at pos.y

// counts[] of a StreamState
#define SC_N 0       // [3] entries of round r at [r % 3]
#define SC_FLAG 3    // overflow / debug flag (as Queues::counts[3])
#define SC_SHADOW 4  // [3] shadow records of round r at [r % 3]
#define SC_TRACE 7   // length of the traversal queue
#define SC_LEFTOVER 8 // shadow rays the 4-wide walk handed back
#define SC_DECIDED 9 // rays answered by their producer (counting launches)

struct StreamState {
	float4* O[2];     // ray origin xyz, w = ray.t after the head tests        } entry e of round parity p
	float4* D[2];     // ray direction xyz, w = head candidate (pack_head)     }
	float4* hitN[2];  // hit normal xyz, w = t
	int2* hitId[2];   // objIdx, material
	float4* W[2];     // path weight xyz
	float4* E[2];     // energy xyz, w = RNG state
	float4* L[2];     // radiance so far xyz, w = sample id
	unsigned char* cls[2]; // CL_* bits
	int2* pos;        // current round, per GROUP of 64 entries: x = entry in the next round of the group's first CL_CONT entry, y = shadow record of its first CL_SHADOW entry (k_assign; k_shade_s counts inside the group)
	float4* shI;      // shadow record: ray.IntersectionPoint() xyz, w = material
	float4* shN;      //                hit normal xyz, w = continuation entry (holds this path's E and L)
	float4* shD;      //                direction of the ray that hit xyz
	float4* shW;      //                weight of the segment xyz
	float4* shP;      // [light][cap]   sampled light position xyz
	unsigned char* vis; // [light][cap] 1: occluded
	uint* traceQ;     // entries with CL_TRACE, in order
	uint* leftover;   // connect work items the 4-wide walk handed back
	int* heads;       // work heads: extend [0, RT_HEADS), connect [RT_HEADS, 2 RT_HEADS)
	int* counts;
	int cap;          // entries the arrays hold
};

// what the hit decides about the path (path mode; 'last': the round whose hits are at depth 0, renderer.cpp:129)
// matType: the material type the primitive record carried (resolve_hit), 0 = look it up
__device__ __forceinline__ int hit_class(const DScene& S, int objIdx, int mat, int last, int matType = 0)
{
	if (objIdx == -1) return 0;                                 // renderer.cpp:134
	if (objIdx >= 11 && objIdx < 11 + S.nLights) return 0;      // :135-137
	const int type = matType ? matType : S.mats[mat].type;
	const bool shadow = type != 3 && type != 2 && S.nLights > 0; // DIFFUSE: the light loop, :158-176
	const bool cont = !last || shadow; // every material scatters one ray while depth lasts; in the last round a diffuse hit
	                                   // still needs a continuation entry: light() finds the path's E and L there
	return (cont ? CL_CONT : 0) | (shadow ? CL_SHADOW : 0);
}

// Does the first step of the traversal already end it?  (trace_persistent: link = S.rootLink, one pair step, stack empty.)
// Same operands, same tests: the reference's boxes with the clean / exact slab test, and at the TLAS level the reach
// boxes under the same guard.  All loads are wave-uniform.
__device__ __forceinline__ bool ray_decided(const DScene& S, const f3& O, const f3& D, float rayT)
{
	const uint root = S.rootLink;
	if (root == RT_EMPTY) return true;
	if (root & (RT_LEAF_BIT | RT_INST_BIT)) return false; // a single leaf / a single instance: extend's business
	const f3 rD = rcp3(D);
	const bool clean = ray_is_clean(O, D, rD);
	const float4* p = S.pairs + 4 * (size_t)root;
	const float4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
	float dist1, dist2;
	if (clean) dist1 = intersect_aabb_clean(O, rD, rayT, xyz(a0), xyz(a1)), dist2 = intersect_aabb_clean(O, rD, rayT, xyz(b0), xyz(b1));
	else dist1 = intersect_aabb_exact(O, rD, rayT, xyz(a0), xyz(a1)), dist2 = intersect_aabb_exact(O, rD, rayT, xyz(b0), xyz(b1));
	if (S.useTLAS && clean && fabsf(O.x) + fabsf(O.y) + fabsf(O.z) <= S.reachOriginMax) {
		const float4* q = S.reach + 3 * (size_t)(root - S.tlasBase);
		const float4 r0 = q[0], r1 = q[1], r2 = q[2];
		if (!box_reachable(O, rD, rayT, xyz(r0), f3(r0.w, r1.x, r1.y))) dist1 = 1e30f;
		if (!box_reachable(O, rD, rayT, f3(r1.z, r1.w, r2.x), f3(r2.y, r2.z, r2.w))) dist2 = 1e30f;
	}
	return dist1 == 1e30f && dist2 == 1e30f;
}

// A new ray for entry e of the round with parity p: head tests (as emit_ray), then the decision above.  Returns the class
// byte of the entry; a decided ray gets its hit record here.  'last': the round this ray is traced in is the last one.
// finishNoHit (generate): a decided ray that hit nothing or a light is not stored at all (the caller finishes the sample).
struct NewRay { unsigned char cls; bool decided; float t; int objIdx, mat; f3 normal; };
__device__ __forceinline__ NewRay emit_ray_s(const DScene& S, const StreamState& T, int p, int e, const f3& O, const f3& D, float t_min, int last, bool finishNoHit, bool decide)
{
	NewRay r;
	float rayT = 1e34f; // Ray constructor default (template/scene.h:42)
	HitRef head;
	head.kind = -1, head.inst = -1, head.prim = 0, head.t = 0;
	LaneCounters unused;
	find_nearest_head<false>(S, O, D, t_min, rayT, head, unused);
	r.cls = CL_LIVE | CL_TRACE, r.decided = false, r.t = rayT, r.objIdx = -1, r.mat = -1, r.normal = f3(0.0f);
	if (decide && ray_decided(S, O, D, rayT)) {
		r.decided = true;
		head.t = rayT;
		int matType;
		resolve_hit(S, head, O, D, r.objIdx, r.mat, r.normal, &matType);
		const int hc = hit_class(S, r.objIdx, r.mat, last, matType);
		// nothing hit, or a light: Sample() returns at renderer.cpp:134 / :135-137 (in the last round a glass or metal hit has class 0 too)
		if (finishNoHit && (r.objIdx == -1 || (r.objIdx >= 11 && r.objIdx < 11 + S.nLights))) { r.cls = 0; return r; }
		T.hitN[p][e] = mk4(r.normal, rayT);
		T.hitId[p][e] = make_int2(r.objIdx, r.mat);
		r.cls = (unsigned char)(CL_LIVE | hc);
	}
	T.O[p][e] = mk4(O, rayT);
	T.D[p][e] = mk4(D, __uint_as_float(pack_head(head)));
	return r;
}

// counting launches: rays answered by their producer are Scene::FindNearest calls like any other.  A thread tallies its
// own, the block adds them to counts[SC_DECIDED] with one atomic at its end (same-address atomics retire at ~88 per
// microsecond: one per wave of a 2 M-wave launch would take longer than the launch), k_fold_decided moves the batch's
// total into the counters.
__device__ __forceinline__ void flush_decided(int* counts, uint mine)
{
	__shared__ uint blockDecided;
	if (threadIdx.x == 0) blockDecided = 0;
	__syncthreads();
	uint x = mine;
	for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
	if ((threadIdx.x & 63) == 0 && x) atomicAdd(&blockDecided, x);
	__syncthreads();
	if (threadIdx.x == 0 && blockDecided) atomicAdd(&counts[SC_DECIDED], (int)blockDecided);
}
__global__ void k_fold_decided(DScene S, int* counts, DCounters* counters)
{
	const unsigned long long nd = (unsigned long long)(uint)counts[SC_DECIDED];
	counts[SC_DECIDED] = 0;
	counters->rays_nearest += nd, counters->light_tests += nd * (unsigned)S.nLights;
	if (S.useTLAS) counters->brute_tests += nd * (unsigned)(S.nBruteSph + S.nBrutePla);
}

// Round bookkeeping without launches of its own.  Round r reads its entry count at counts[SC_N + r % 3]; assign(r) adds up the
// next round's at [(r + 1) % 3] and this round's shadow records at [SC_SHADOW + r % 3].  Whatever round r + 1 needs zeroed is
// zeroed by ONE thread of shade(r) (and by generate for round 0): the traversal queue's length and extend's work heads
// (compact(r) and extend(r) are over), the entry count of round r + 2 (its slot last held round r - 1's, which nobody reads any
// more), the shadow count of round r + 1 (its slot held round r - 2's: connect / light of that round ended before shade(r - 1)
// was allowed to start), and connect's heads for connect(r) (connect(r - 1) ended before this kernel was allowed to start).
__device__ __forceinline__ void prepare_round(const StreamState& T, int next /* the round being prepared */, int n0)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	if (n0 >= 0) T.counts[SC_N + next % 3] = n0; // generate: the batch's samples are round 0's entries
	T.counts[SC_TRACE] = 0, T.counts[SC_N + (next + 1) % 3] = 0, T.counts[SC_SHADOW + next % 3] = 0;
	for (int h = 0; h < RT_HEADS; h++) T.heads[h * RT_HEAD_STRIDE] = 0;
	// connect of the round before 'next' (it starts after the kernel this runs in)
	T.counts[SC_LEFTOVER] = 0;
	for (int h = RT_HEADS; h < 2 * RT_HEADS; h++) T.heads[h * RT_HEAD_STRIDE] = 0;
}
// connect's work heads alone: the leftover launch of the 4-wide walk goes through its own list with them
__global__ void k_stream_begin(StreamState T)
{
	for (int h = RT_HEADS; h < 2 * RT_HEADS; h++) T.heads[h * RT_HEAD_STRIDE] = 0;
}

__global__ void __launch_bounds__(RT_COMPACT_BLOCK) k_compact_s(StreamState T, int round)
{
	compact_body(T.cls[round & 1], T.counts[SC_N + round % 3], CL_TRACE, T.traceQ, &T.counts[SC_TRACE]);
}

// generate: camera sample e of the batch (renderer.cpp:263-278) -> entry e of round 0.  A sample whose camera ray
// provably leaves the scene (or ends on a light) is finished here: Sample() returns at :134 / :135-137.
__global__ void __launch_bounds__(RT_BLOCK) k_generate_s(DScene S, DCamera C, RenderParams R, StreamState T, int last, int decide, int counting)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = e < (int)R.nSamples;
	bool decided = false;
	prepare_round(T, 0, (int)R.nSamples);
	if (valid) {
		const uint sid = R.sampleFirst + (uint)e;
		f3 O, D;
		uint seed = 0;
		if (R.customO) {
			O = f3(R.customO[3 * sid], R.customO[3 * sid + 1], R.customO[3 * sid + 2]);
			D = f3(R.customD[3 * sid], R.customD[3 * sid + 1], R.customD[3 * sid + 2]);
		} else sample_primary(C, R, sid, O, D, seed);
		const NewRay nr = emit_ray_s(S, T, 0, e, O, D, mode_t_min(1), last, (decide & 2) == 0, decide != 0);
		decided = nr.decided;
		if (nr.cls == 0) {
			// what the first shade would do with Lsum = 0 and W = 1 (0 + 1 * x is x, bit for bit)
			f3 Lsum(0.0f);
			const f3 W(1.0f);
			const unsigned char* texel = nullptr;
			if (nr.objIdx == -1) {
				texel = sky_texel(S, D);
				if (texel) Lsum = Lsum + W * (f3((float)texel[0], (float)texel[1], (float)texel[2]) / 255);
			} else {
				const f3 I = O + nr.t * D;
				Lsum = Lsum + W * light_intensity(S.lights[nr.objIdx - 11], I, nr.normal, I);
			}
			// the sample is a sky texel: 0 + 1 * (b / 255) is b / 255, and its gamma-corrected value one of 256 numbers the
			// device computed with the same pow (three double-precision pow calls per finished sample are most of this kernel otherwise)
			if (texel && !R.customOut && S.gammaLut) R.samples[sid] = make_float4(S.gammaLut[texel[0]], S.gammaLut[texel[1]], S.gammaLut[texel[2]], 0.0f);
			else store_sample(R, sid, Lsum);
		}
		T.cls[0][e] = nr.cls;
	}
	if (counting) flush_decided(T.counts, decided ? 1u : 0u);
}

// extend: Scene::FindNearest for the entries of the traversal queue
struct StreamExtendPolicy {
	const DScene& S;
	const StreamState& T;
	int parity, last;
	int* flag;
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef& head) const
	{
		const int e = (int)ld_stream(T.traceQ + work);
		RT_CHECK(e >= 0 && e < T.cap, 20, flag);
		const float4 o4 = ld_stream(T.O[parity] + e), d4 = ld_stream(T.D[parity] + e);
		O = xyz(o4), D = xyz(d4), tmax = o4.w;
		unpack_head(__float_as_uint(d4.w), head);
		return true;
	}
	__device__ __forceinline__ int slot_of(int work) const { return (int)T.traceQ[work]; }
	__device__ __forceinline__ void store(int work, const HitRef& hit, const f3& /*O*/, const f3& /*D*/) const
	{
		const int e = (int)ld_stream(T.traceQ + work);
		int objIdx, mat, matType;
		f3 normal;
		const StreamState& Tc = T;
		const int par = parity;
		resolve_hit_lazy(S, hit, [&](f3& o, f3& d) { o = xyz(ld_stream(Tc.O[par] + e)), d = xyz(ld_stream(Tc.D[par] + e)); }, objIdx, mat, normal, &matType);
		st_stream(T.hitN[parity] + e, mk4(normal, hit.t));
		st_stream(T.hitId[parity] + e, make_int2(objIdx, mat));
		T.cls[parity][e] = (unsigned char)(CL_LIVE | CL_TRACE | hit_class(S, objIdx, mat, last, matType)); // the record said what its material is: no second fetch
	}
};
template <bool COUNT>
__global__ void __launch_bounds__(RT_BLOCK, RT_EXTEND_WAVES) k_extend_s(DScene S, StreamState T, int parity, int last, float t_min, int tuning, uint* spill, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	StreamExtendPolicy pol{ S, T, parity, last, &T.counts[SC_FLAG] };
	trace_persistent<false, COUNT, false>(S, pol, T.counts[SC_TRACE], T.heads, t_min, tuning, ldsStack, spill, &T.counts[SC_FLAG], lc, rays);
	if (COUNT) {
		lc.light = rays * (uint)S.nLights, lc.brute = S.useTLAS ? rays * (uint)(S.nBruteSph + S.nBrutePla) : 0;
		flush_counters(counters, lc, rays, 0);
	}
}

// assign: pos[e / 64] = (number of CL_CONT entries before group e / 64, number of CL_SHADOW entries before it) + the block's bases; the
// totals become the next round's entry count and this round's shadow count.  Same shape as compact_body: a wave owns a
// contiguous range, counts it, the block reserves with one atomic per counter, then every fourth lane writes the bases of
// its group of 64 entries.  Blocks reserve in arrival order, so entry order is by block, in order within.
__global__ void __launch_bounds__(RT_COMPACT_BLOCK) k_assign(StreamState T, int round)
{
	const int parity = round & 1;
	__shared__ int waveC[RT_COMPACT_BLOCK / 64], waveS[RT_COMPACT_BLOCK / 64];
	__shared__ int baseC, baseS;
	const unsigned char* status = T.cls[parity];
	const int n = T.counts[SC_N + round % 3];
	const uint lane = threadIdx.x & 63;
	const int waves = (gridDim.x * blockDim.x) >> 6;
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	int per = (n + waves - 1) / waves;
	per = (per + 1023) & ~1023;
	const long long firstL = (long long)wave * per;
	const int first = firstL < n ? (int)firstL : n;
	const int last = firstL + per < n ? (int)(firstL + per) : n;
	int mineC = 0, mineS = 0;
	for (int s0 = first; s0 < last; s0 += 1024) {
		const uint4 v = status16(status, s0 + (int)lane * 16, last, (CL_CONT | CL_SHADOW) * 0x01010101u);
		const uint cm = CL_CONT * 0x01010101u, sm = CL_SHADOW * 0x01010101u;
		mineC += __popc(v.x & cm) + __popc(v.y & cm) + __popc(v.z & cm) + __popc(v.w & cm);
		mineS += __popc(v.x & sm) + __popc(v.y & sm) + __popc(v.z & sm) + __popc(v.w & sm);
	}
	int totC = mineC, totS = mineS;
	for (int o = 32; o > 0; o >>= 1) totC += __shfl_xor(totC, o), totS += __shfl_xor(totS, o);
	const int wib = threadIdx.x >> 6;
	if (lane == 0) waveC[wib] = totC, waveS[wib] = totS;
	__syncthreads();
	if (threadIdx.x == 0) {
		int sc = 0, ss = 0;
		for (int w = 0; w < RT_COMPACT_BLOCK / 64; w++) sc += waveC[w], ss += waveS[w];
		baseC = sc > 0 ? atomicAdd(&T.counts[SC_N + (round + 1) % 3], sc) : 0;
		baseS = ss > 0 ? atomicAdd(&T.counts[SC_SHADOW + round % 3], ss) : 0;
	}
	__syncthreads();
	int bc = baseC, bs = baseS;
	for (int w = 0; w < wib; w++) bc += waveC[w], bs += waveS[w];
	if (totC == 0) return; // no continuation in this wave's range: nobody reads its positions (CL_SHADOW implies CL_CONT)
	for (int s0 = first; s0 < last; s0 += 1024) {
		const int e0 = s0 + (int)lane * 16;
		const uint4 v = status16(status, e0, last, (CL_CONT | CL_SHADOW) * 0x01010101u);
		const uint cm = CL_CONT * 0x01010101u, sm = CL_SHADOW * 0x01010101u;
		const int cC = __popc(v.x & cm) + __popc(v.y & cm) + __popc(v.z & cm) + __popc(v.w & cm);
		const int cS = __popc(v.x & sm) + __popc(v.y & sm) + __popc(v.z & sm) + __popc(v.w & sm);
		int inC = cC, inS = cS;
		for (int o = 1; o < 64; o <<= 1) {
			const int tc = __shfl_up(inC, o), ts = __shfl_up(inS, o);
			if ((int)lane >= o) inC += tc, inS += ts;
		}
		// one pair of bases per GROUP of 64 entries (four lanes of this scan): a wave of the shading kernel holds exactly one group and
		// counts inside it with two ballots -- eight bytes per entry written here and read there otherwise (4.5 GB per bench step)
		const int pc = bc + inC - cC, ps = bs + inS - cS;
		if (e0 < last && (lane & 3) == 0) T.pos[e0 >> 6] = make_int2(pc, ps);
		bc += __shfl(inC, 63), bs += __shfl(inS, 63);
	}
}

// The shading kernels are bound by latency, not by HBM or the VALU (measured: waves waiting 66 % of the time, VALU pipes 30 %
// busy, 2.7 TB/s): a lane's chain is entry -> hit record -> material -> (next ray: brute-force primitive of the head
// candidate) -> sky texel, one memory round trip each.  The two tiny tables in that chain -- materials and the
// brute-force primitives -- are copied into LDS by every block (a ds_read instead of a trip to L2), and everything
// that does not depend on another load is issued before the first use.
#define RT_LDS_MATS 48   // materials / brute-force primitives a block keeps in LDS (scenes with more read them from memory)
#define RT_LDS_BRUTE 24
struct ShadeTables { uint mats[RT_LDS_MATS * 16]; float4 brute[RT_LDS_BRUTE * 4]; uint lights[RT_MAX_LIGHTS * (sizeof(DLight) / 4)]; };
__device__ __forceinline__ void load_tables(DScene& S, ShadeTables& L, int enable)
{
	const bool m = enable && S.nMats <= RT_LDS_MATS, b = enable && S.useTLAS && S.nBruteSph + S.nBrutePla <= RT_LDS_BRUTE && S.nBruteSph + S.nBrutePla > 0;
	const bool l = enable && S.nLights > 0 && S.nLights <= RT_MAX_LIGHTS; // the light loop of a diffuse hit reads every light's record: a round trip each otherwise
	if (m) for (int i = (int)threadIdx.x; i < S.nMats * 16; i += (int)blockDim.x) L.mats[i] = ((const uint*)S.mats)[i];
	if (b) for (int i = (int)threadIdx.x; i < (S.nBruteSph + S.nBrutePla) * 4; i += (int)blockDim.x) L.brute[i] = S.brute[i];
	if (l) for (int i = (int)threadIdx.x; i < S.nLights * (int)(sizeof(DLight) / 4); i += (int)blockDim.x) L.lights[i] = ((const uint*)S.lights)[i];
	if (m || b || l) __syncthreads();
	if (m) S.mats = (const DMaterial*)L.mats;
	if (b) S.brute = L.brute;
	if (l) S.lights = (const DLight*)L.lights;
}

// shade: everything Sample does at the hit of entry e except the occlusion-dependent direct terms (renderer.cpp:133-233).
// fresh: first segment of a sample (entry index = sample of the batch; W = 1, L = 0, E evaluated, nothing read).
#ifndef RT_SHADE_S_WAVES
#define RT_SHADE_S_WAVES 5
#endif
#ifndef RT_SHADE_Q_WAVES
#define RT_SHADE_Q_WAVES 5 // the Q-learning variant (it needs 117 VGPRs unbounded)
#endif
// QL: the indirect bounce of a diffuse hit is drawn from the Q table (rt_qlearn.h) and every hit pays its reward to the
// (cell, patch) that sent the ray; W.w of an entry carries that key (0: none).
template <bool QL>
__global__ void __launch_bounds__(RT_BLOCK, QL ? RT_SHADE_Q_WAVES : RT_SHADE_S_WAVES) k_shade_s(DScene S0, DCamera C, RenderParams R, StreamState T, int round, int fresh, int last, int lastNext, int decide, int ldsTables, int counting, QTable Qt)
{
	__shared__ ShadeTables tables;
	uint nDecided = 0;
	uint qOld = 0; // QL: the words this lane's rewards were added to, or-ed (q_reward: bit 31 = a count field past its limit)
	const int parity = round & 1;
	prepare_round(T, round + 1, -1);
	DScene S = S0;
	load_tables(S, tables, ldsTables);
	const int pout = 1 - parity;
	const int n = T.counts[SC_N + round % 3];
	const size_t cap = (size_t)T.cap;
	const int nIter = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
	for (int it = 0; it < nIter; it++) {
		const int e = it * (int)(gridDim.x * blockDim.x) + (int)(blockIdx.x * blockDim.x + threadIdx.x);
		// after round 0 every entry is live: all of an entry's loads go out together, the class byte among them (round 0
		// looks at the class first: the entries of samples finished by generate hold nothing, and they come in long runs)
		const bool spec = !fresh && e < n;
		float4 o4, d4, hn, e4, w4, l4;
		int2 id;
		const int2 grp = e < n ? T.pos[e >> 6] : make_int2(0, 0); // the bases of this wave's group of 64 entries (k_assign); the same for every lane that has an entry (the grid's last iteration reaches past n, and past pos[])
		if (spec) {
			o4 = T.O[parity][e], d4 = T.D[parity][e], hn = T.hitN[parity][e], id = T.hitId[parity][e];
			e4 = T.E[parity][e], w4 = T.W[parity][e], l4 = T.L[parity][e];
		}
		const unsigned char c = e < n ? T.cls[parity][e] : 0;
		// the entry's place in the next round and its shadow record: the group's bases + the entries before it in the wave that want one
		int2 p;
		{
			const unsigned long long mC = __ballot((c & CL_CONT) != 0), mS = __ballot((c & CL_SHADOW) != 0);
			p.x = grp.x + (int)__builtin_amdgcn_mbcnt_hi((uint)(mC >> 32), __builtin_amdgcn_mbcnt_lo((uint)mC, 0u));
			p.y = grp.y + (int)__builtin_amdgcn_mbcnt_hi((uint)(mS >> 32), __builtin_amdgcn_mbcnt_lo((uint)mS, 0u));
		}
		if (c & CL_LIVE) {
			if (!spec) {
				o4 = T.O[parity][e], d4 = T.D[parity][e], hn = T.hitN[parity][e], id = T.hitId[parity][e];
				e4 = fresh_energy(C, R, R.sampleFirst + (uint)e);
				w4 = make_float4(1, 1, 1, 0);
				l4 = make_float4(0, 0, 0, __uint_as_float(R.sampleFirst + (uint)e));
			}
			RT_CHECK(!(c & CL_CONT) || (p.x >= 0 && p.x < T.cap && p.y >= 0 && p.y < T.cap), 21, &T.counts[SC_FLAG]);
			const f3 O = xyz(o4), D = xyz(d4), normal = xyz(hn);
			const float t = hn.w;
			f3 W = xyz(w4), E = xyz(e4), Lsum = xyz(l4);
			uint seed = __float_as_uint(e4.w);
			const f3 I = O + t * D; // ray.IntersectionPoint()
			const bool childTraces = !last; // Sample(depth - 1 < 0) = 0.05 (renderer.cpp:129)
			bool segmentEnds = true, wantShadow = false;
			f3 nO(0.0f), nD(0.0f), nW(0.0f);
			uint nextKey = 0;       // QL: (cell, patch) + 1 of the ray this hit scatters
			bool deadSample = false; // QL: the guided direction points below the surface: nothing to trace
			// QL: the key word travels in W.w: bits 0-30 the (cell, patch) + 1 that sent this ray, bit 31 "this sample pays rewards"
			const uint keyWord = QL ? (fresh ? (((__float_as_uint(e4.w) & Qt.learnMask) == 0) ? RT_Q_LEARNER : 0u) : __float_as_uint(w4.w)) : 0u;
			const bool learner = (keyWord & RT_Q_LEARNER) != 0;
			const uint prevKey = learner ? (keyWord & ~RT_Q_LEARNER) : 0u;

			if (id.x == -1) {
				const f3 sky = sky_color(S, D);
				Lsum = Lsum + W * sky; // renderer.cpp:134
				if (QL && prevKey) qOld |= q_reward(Qt, prevKey, q_lum(sky));
			} else if (id.x >= 11 && id.x < 11 + S.nLights) {
				const f3 li = light_intensity(S.lights[id.x - 11], I, normal, I);
				Lsum = Lsum + W * li; // :135-137
				if (QL && prevKey) qOld |= q_reward(Qt, prevKey, q_lum(li));
			} else {
				const DMaterial m = S.mats[id.y];
				const f3 col(m.col[0], m.col[1], m.col[2]);
				if (QL && prevKey) {
					const bool dif = m.type != 3 && m.type != 2;
					const f3 albedoQ(m.albedo[0], m.albedo[1], m.albedo[2]);
					qOld |= q_reward(Qt, prevKey, q_expected(Qt, q_cell(Qt, I), normal, dif ? q_lum(col * albedoQ) : q_lum(col), dif));
				}
				if (m.type == 3) { // GLASS, renderer.cpp:198-233
					const float kr = glass_fresnel(normalize(D), normalize(normal), m.ir);
					const bool outside = dot(D, normal) < 0;
					const f3 bias = 0.0001f * normal;
					const f3 norm = outside ? normal : -normal;
					const float r = !outside ? m.ir : (1 / m.ir);
					if (outside) {
						// exp(+-0) is exactly 1 and x * 1 is x: a glass without absorption (every glass of the reference's scenes
						// but one) skips three double-precision exp calls; NaN / inf exponents still take them
						const float ax = m.absorption[0] * -t, ay = m.absorption[1] * -t, az = m.absorption[2] * -t;
						if (ax != 0) E.x *= x_expf(ax);
						if (ay != 0) E.y *= x_expf(ay);
						if (az != 0) E.z *= x_expf(az);
					}
					const bool refr = kr < RandomFloat(seed);
					if (refr) {
						nD = normalize(glass_refract(D, norm, r));
						nO = outside ? I - bias : I + bias;
						const f3 tempCol = col * E;
						nW = W * (tempCol * (1 - kr));
					} else {
						nD = normalize(reflect(D, norm));
						nO = outside ? I + bias : I - bias;
						nW = W * (col * kr);
					}
					if (childTraces) segmentEnds = false;
					else Lsum = Lsum + nW * f3(0.05f);
				} else if (m.type == 2) { // METAL, renderer.cpp:192-197, metal::scatter template/scene.h:630-635
					nO = I + normal * 0.001f, nD = reflect(D, normal);
					nW = W * col;
					if (childTraces) segmentEnds = false;
					else Lsum = Lsum + nW * f3(0.05f);
				} else { // DIFFUSE, renderer.cpp:156-191
					wantShadow = S.nLights > 0;
					for (int i = 0; i < S.nLights; i++) {
						const f3 pickedPos = light_position(S.lights[i], false, seed);
						T.shP[(size_t)i * cap + (size_t)p.y] = mk4(pickedPos, 0.0f);
					}
					const f3 albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
					if constexpr (QL) {
						// the scattered direction from the Q table of this cell instead of the uniform hemisphere
						float P;
						int patch;
						const int cell = q_cell(Qt, I);
						const f3 dirQ = q_sample(Qt, cell, seed, P, patch);
						const float c = dot(dirQ, normal);
						nO = I, nD = dirQ;
						if (c > 0) {
							const f3 cos_i(c);
							nW = W * (((1.0f / (16.0f * P)) * (col * cos_i)) * albedo); // 1 / (pi pdf) in the place of the uniform hemisphere's 2
							nextKey = learner ? 1u + (uint)cell * RT_Q_PATCHES + (uint)patch : 0u;
						} else {
							deadSample = true; // below the surface: the patch learns a reward of 0, the path carries no weight on
							if (learner) qOld |= q_reward(Qt, 1u + (uint)cell * RT_Q_PATCHES + (uint)patch, 0.0f);
						}
					} else {
						const f3 rayToHemi = RandomInHemisphere(seed, normal);
						const f3 cos_i(dot(rayToHemi, normal));
						nO = I, nD = rayToHemi;
						nW = W * ((2 * (col * cos_i)) * albedo); // child coefficient of (direct*INVPI + 2*indirect) * albedo
					}
					if (childTraces) segmentEnds = false;
					else Lsum = Lsum + nW * f3(0.05f);
				}
			}
			// the class byte promised exactly these outputs
			RT_CHECK(((c & CL_SHADOW) != 0) == wantShadow && ((c & CL_CONT) != 0) == (!segmentEnds || wantShadow), 22, &T.counts[SC_FLAG]);
			if (wantShadow) {
				T.shI[p.y] = mk4(I, __int_as_float(id.y));
				T.shN[p.y] = mk4(normal, __int_as_float(p.x));
				T.shD[p.y] = mk4(D, 0.0f);
				T.shW[p.y] = mk4(W, 0.0f);
			}
			if (!segmentEnds || wantShadow) {
				T.E[pout][p.x] = mk4(E, __uint_as_float(seed));
				T.L[pout][p.x] = mk4(Lsum, l4.w);
				if (!segmentEnds && deadSample) {
					// the entry the class byte promised, as a ray that hits nothing and weighs nothing: the next shade adds 0 and ends it
					T.O[pout][p.x] = mk4(nO, 1e34f), T.D[pout][p.x] = mk4(nD, 0.0f);
					T.hitN[pout][p.x] = make_float4(0, 0, 0, 1e34f), T.hitId[pout][p.x] = make_int2(-1, -1);
					T.W[pout][p.x] = make_float4(0, 0, 0, 0);
					T.cls[pout][p.x] = CL_LIVE;
				} else if (!segmentEnds) {
					const NewRay nr = emit_ray_s(S, T, pout, p.x, nO, nD, mode_t_min(1), lastNext, false, decide != 0);
					T.W[pout][p.x] = mk4(nW, __uint_as_float(nextKey | (keyWord & RT_Q_LEARNER)));
					T.cls[pout][p.x] = nr.cls;
					nDecided += nr.decided ? 1u : 0u;
				}
			} else store_sample(R, __float_as_uint(l4.w), Lsum);
		}
	}
	if (QL && (qOld >> 31)) *Qt.ovf = 1;
	if (counting) flush_decided(T.counts, nDecided);
}

// connect: Scene::IsOccluded from the hit point towards each sampled light position (renderer.cpp:161-165).
// Work item = light * nShadow + shadow record: consecutive items read consecutive records and aim at the same light.
struct StreamConnectPolicy {
	const StreamState& T;
	int nShadow;
	int* flag;
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef&) const
	{
		const int li = work / nShadow, s = work - li * nShadow;
		RT_CHECK(s >= 0 && s < T.cap && li >= 0 && li < RT_MAX_LIGHTS, 23, flag);
		const f3 I = xyz(ld_stream(T.shI + s));
		const f3 pickedPos = xyz(ld_stream(T.shP + ((size_t)li * (size_t)T.cap + (size_t)s)));
		f3 lightRayDirection = pickedPos - I;
		const float len2 = dot(lightRayDirection, lightRayDirection);
		lightRayDirection = normalize(lightRayDirection);
		O = I + lightRayDirection * 1e-4f, D = lightRayDirection, tmax = sqrtf(len2);
		return true;
	}
	__device__ __forceinline__ void store(int work, bool occluded) const
	{
		const int li = work / nShadow, s = work - li * nShadow;
		st_stream(T.vis + ((size_t)li * (size_t)T.cap + (size_t)s), (unsigned char)(occluded ? 1 : 0));
	}
	__device__ __forceinline__ void leftover(int work) const { T.leftover[atomicAdd(&T.counts[SC_LEFTOVER], 1)] = (uint)work; }
};
template <bool COUNT, bool WIDE = false, bool LISTED = false, bool WIDE8 = false>
__global__ void __launch_bounds__(RT_BLOCK, RT_CONNECT_WAVES) k_connect_s(DScene S, StreamState T, int round, int tuning, uint* spill, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	const int nShadow = T.counts[SC_SHADOW + round % 3];
	StreamConnectPolicy pol{ T, nShadow > 0 ? nShadow : 1, &T.counts[SC_FLAG] };
	int* heads = T.heads + RT_HEADS * RT_HEAD_STRIDE;
	if constexpr (LISTED) {
		ListedPolicy<StreamConnectPolicy> lp{ pol, T.leftover };
		trace_persistent<true, COUNT, false, ListedPolicy<StreamConnectPolicy>>(S, lp, T.counts[SC_LEFTOVER], heads, 0.0f, tuning, ldsStack, spill, &T.counts[SC_FLAG], lc, rays);
	} else
		trace_persistent<true, COUNT, false, StreamConnectPolicy, false, WIDE8 ? 8 : (WIDE ? 4 : 2)>(S, pol, nShadow * S.nLights, heads, 0.0f, tuning, ldsStack, spill, &T.counts[SC_FLAG], lc, rays);
	if (COUNT) flush_counters(counters, lc, 0, rays);
}

// traverse: extend of round r and connect of round r - 1 as ONE persistent launch over one work list (RT_FUSE=1): items
// [0, nTrace) are the traversal queue's rays (nearest hit: the long ones start first), the rest the shadow rays of the round
// before, light-major.  A lane's kind of query follows its work item (trace_persistent MIXED).  One launch has one drain where
// two have two, which is most of what a small batch's traversal costs (profiles/r03_ab_stream_fuse.txt: at the 1/8 frame an
// extend launch is busy 0.2-0.8 ms and drains for 0.5-0.7); at large batches waves that mix the two kinds cost more than the
// drains return (DESIGN.md finding 19), so this is the small batches' round loop.
struct StreamTraversePolicy {
	StreamExtendPolicy ext;
	StreamConnectPolicy con;
	int nTrace;
	__device__ __forceinline__ bool any_of(int work) const { return work >= nTrace; }
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef& head) const
	{
		if (work < nTrace) return ext.load(work, O, D, tmax, head);
		return con.load(work - nTrace, O, D, tmax, head);
	}
	__device__ __forceinline__ void store(int work, const HitRef& hit, const f3& O, const f3& D) const { ext.store(work, hit, O, D); }
	__device__ __forceinline__ void store(int work, bool occluded) const { con.store(work - nTrace, occluded); }
};
__global__ void __launch_bounds__(RT_BLOCK, RT_TRAVERSE_WAVES) k_traverse_s(DScene S, StreamState T, int round, int last, float t_min, int tuning, uint* spill)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	const int nTrace = T.counts[SC_TRACE], nShadow = T.counts[SC_SHADOW + (round + 2) % 3]; // the shadow records of round - 1
	StreamTraversePolicy pol{ { S, T, round & 1, last, &T.counts[SC_FLAG] }, { T, nShadow > 0 ? nShadow : 1, &T.counts[SC_FLAG] }, nTrace };
	trace_persistent<false, false, false, StreamTraversePolicy, true>(S, pol, nTrace + nShadow * S.nLights, T.heads, t_min, tuning, ldsStack, spill, &T.counts[SC_FLAG], lc, rays);
}

// light: the direct-light terms of a diffuse hit, in light order (renderer.cpp:158-176: occlusion test first, scatter
// only when visible), added to the radiance of the path's continuation entry; in the last round the sample is complete.
__global__ void RT_LIGHT_BOUNDS k_light_s(DScene S0, RenderParams R, StreamState T, int round, int last, int ldsTables)
{
	__shared__ ShadeTables tables;
	DScene S = S0;
	load_tables(S, tables, ldsTables);
	const int pout = 1 - (round & 1);
	const int n = T.counts[SC_SHADOW + round % 3];
	const size_t cap = (size_t)T.cap;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
		const float4 i4 = T.shI[s], n4 = T.shN[s], d4 = T.shD[s], w4 = T.shW[s];
		const int cp = __float_as_int(n4.w);
		RT_CHECK(cp >= 0 && cp < T.cap, 24, &T.counts[SC_FLAG]);
		const float4 e4 = T.E[pout][cp], l4 = T.L[pout][cp];
		const f3 I = xyz(i4), normal = xyz(n4), D = xyz(d4), W = xyz(w4);
		f3 E = xyz(e4), Lsum = xyz(l4);
		const DMaterial m = S.mats[__float_as_int(i4.w)];
		const f3 col(m.col[0], m.col[1], m.col[2]);
		f3 direct(0.0f);
		for (int i = 0; i < S.nLights; i++) {
			const f3 pickedPos = xyz(T.shP[(size_t)i * cap + (size_t)s]);
			const f3 lightRayDirection = normalize(pickedPos - I);
			if (T.vis[(size_t)i * cap + (size_t)s] != 0) continue;
			const f3 att = diffuse_scatter(m, D, lightRayDirection, light_intensity(S.lights[i], I, normal, pickedPos), normal, E);
			direct = direct + (1 - m.shinieness) * col * att * E;
		}
		const f3 albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
		Lsum = Lsum + W * ((direct * RT_INVPI) * albedo); // direct part of (direct*INVPI + 2*indirect) * albedo
		if (last) store_sample(R, __float_as_uint(l4.w), Lsum);
		else {
			if (E.x != e4.x || E.y != e4.y || E.z != e4.z) T.E[pout][cp] = mk4(E, e4.w);
			if (Lsum.x != l4.x || Lsum.y != l4.y || Lsum.z != l4.z) T.L[pout][cp] = mk4(Lsum, l4.w);
		}
	}
}

} // namespace rtd
