// HBM-resident scene layout and the traversal device functions (the hot loops).
//
// Layout (built by rt_upload_scene from the reference-shaped arrays of include/rt_amd.h):
//   pairs[]  one 64-byte record per sibling pair of BLAS nodes (the reference allocates children
//            in adjacent slots, bvh.cpp:318-319, so one inner-node visit reads exactly one record):
//              float4 {minA.xyz, linkA} {maxA.xyz, -} {minB.xyz, linkB} {maxB.xyz, -}
//            link = LEAF_BIT | first primitive slot   (leaf)
//                 = pair index of the children          (inner)
//            A lane's fetch is one aligned half cache line; a component-wise SoA would touch
//            eight lines for the same visit (gather access, not streaming).
//   prims[]  one 64-byte record per leaf slot, in leaf order (primitiveIdx already applied):
//              tri    {v0.xyz, N.x} {v1.xyz, N.y} {v2.xyz, N.z} {d, objIdx, mat, kind|last}
//              sphere {pos.xyz, r2} {invr, r, -, -} {-}          {-, objIdx, mat, kind|last}
//              plane  {N.xyz, d}    {-}            {-}          {-, objIdx, mat, kind|last}
//            'last' marks the final slot of a leaf, so a leaf is named by its first slot alone and
//            a stack entry is one dword.
//            TLAS inner nodes (tlas.h:4-11) are stored in the same form (children boxes inside the
//            parent's record, link = pair index or INST_BIT | instance), so one piece of code walks both
//            levels; tlas::Intersect tests child (leftRight & 0xFFFF) first, which is slot A here.
//   inst[]   128-byte records: invTransform rows 0-2, matTransform rows 0-2, root link.
//   reach[]  one 48-byte record per TLAS pair, beside pairs[]: {minA.xyz, maxA.x}{maxA.yz, minB.xy}{minB.z, maxB.xyz}.
//            The reference's instance bounds are the union of the BLAS's LOCAL box and its transformed
//            box (bvhInstance.h:15-30 never resets 'bounds'), so TLAS boxes overlap nearly everywhere and
//            a ray enters almost every instance only to fail the first BLAS test.  reach holds, per child,
//            the world box of what the subtree can really contain: the corners of (BLAS root's two child
//            boxes) under the exact inverse of invTransform, evaluated in double at upload, inflated by
//            a margin m = a + b * reachOriginMax that covers the float rounding of the object-space ray
//            (~500 ulps of the magnitudes involved, scaled by the transform's condition number) for every
//            ray whose world origin has |O|_1 <= reachOriginMax (64 x the extent of the instanced geometry;
//            rays from further away are not culled).  A child whose
//            inflated reach box the ray misses cannot yield a hit and is dropped; the ORDER in which the
//            remaining children are visited still comes from the reference's boxes, so the first-found
//            rule for equal t is untouched and results are identical.  Counting launches either walk like
//            the reference (no culling: the counters are the reference's tallies) or like the timed
//            kernels (RT_TUNE_CULL_COUNTED: the counters are the work actually done).
//   wide[]   4-wide nodes for occlusion queries (the reference's own 4-wide variant is bvh.cpp:335-512,
//            :658-761).  One 128-byte record = one cache line: {min.x[4]}{min.y[4]}{min.z[4]}{max.x[4]}{max.y[4]}{max.z[4]}
//            {link[4]}{source[4]}; built at upload by collapsing the reference's binary tree (a node's children are
//            expanded, largest surface first, until there are four); a child entry carries the binary node's OWN box,
//            an unused entry an inverted box.  Why this is exact for Scene::IsOccluded: the answer is "some primitive of
//            some visited leaf occludes", a leaf is visited by the reference iff every binary ancestor's box passes
//            the slab test, and for a ray that cannot produce a NaN slab product (ray_is_clean) a box that passes
//            implies every box CONTAINING it passes (rounding is monotone: (b - O) * rD keeps the order of the b's), so
//            with nested boxes -- checked at upload, parents are unions of their children after bvh::Refit -- the
//            visited leaves are exactly those whose own box passes, whichever ancestors a walk tests on the way.  Order
//            is free (a boolean).  Rays that are not clean, in world or in an instance's object space, are handed to the
//            binary walk (leftover list).  Half as many dependent fetches per ray, and each fetch is one L1 line.
//   wide8[]  8-wide nodes with QUANTISED child boxes, the re-thought form of the reference's QBVH, for occlusion
//            queries.  One 128-byte record (96 used, six dwordx4): {origin.xyz, exponents} {qlo.x[8] qlo.y[8]} {qlo.z[8] qhi.x[8]}
//            {qhi.y[8] qhi.z[8]} {link[0..3]} {link[4..7]}; a child's box is origin + q * 2^e per axis (one fma, the expression
//            the builder checked), rounded OUTWARDS at upload so that it contains the binary node's box it stands for; built
//            by collapsing the reference's binary tree (largest surface first, up to eight children).  Exactness: the argument
//            of wide[] needs the boxes a walk tests on the way to a leaf only to CONTAIN the leaf's own box (for a clean ray a
//            box that passes implies every box containing it passes), and the leaf's own, exact box to decide whether the leaf
//            is visited.  So inner entries may be quantised as long as they contain, and a leaf entry names a leafBox[] record
//            -- the reference's exact box + the first primitive slot -- that is tested (exact slab test) before any primitive:
//            visited leaves are exactly those whose own box passes, as in bvh::BIsOccluded.  A third of the dependent fetches
//            of the binary walk, half its load instructions per ray.
// All of it is read-only and a few MB at most: every XCD's 4 MiB L2 ends up holding its own copy.
#pragma once
#include "rt_dmath.h"

namespace rtd {

#define RT_LEAF_BIT 0x80000000u
#define RT_EMPTY 0xFFFFFFFFu    // root link of a bvh without primitives
#define RT_SENTINEL 0xFFFFFFFEu // stack marker: leave the current instance
#define RT_KIND_TRI 0
#define RT_KIND_SPHERE 1
#define RT_KIND_PLANE 2
#define RT_LAST_BIT 4
#define RT_TYPE_SHIFT 4 // bits 4-5 of a primitive record's kind word: rt_material::type of its material (0: unknown)
#define RT_MAX_LIGHTS 32 // per-light planes of the path state are sized by the scene's own count; this bounds memory, nothing else

#define RT_BLOCK 256
// A traversal block's LDS (trace_persistent): 20,480 bytes, eight blocks = 32 waves per CU of the 160 KB (seven blocks of 23,040
// bytes until the kernels fitted 64 registers: round 4, profiles/r04_ab_eight_waves.txt).  Three parts,
// split per scene (DScene::stackRows, set at upload):
//   [0, rows)            the top 'rows' entries of every lane's traversal stack, [entry][lane]
//   [rows, rows + 6)     the world-space ray of a lane that is inside an instance, [component][lane]
//   the rest             the block's copy of the TLAS (pairs, reach records, instance transforms), when it fits
// rows = 14 without a TLAS copy; a copy takes rows away down to RT_STACK_ROWS_MIN (the bench scene's 8 instances: 12 rows; the 16 of config 5: 10;
// measured there: 13 rows cost nothing, the copy takes 4 % off the frame set); a TLAS too large for that stays in global memory.
#ifndef RT_LDS_WORDS
#define RT_LDS_WORDS 5120 // 20 KB: eight blocks = 32 waves per CU of the 160 KB
#endif
// stack rows without a TLAS copy: what the block's LDS holds beside the six world-ray rows (14 at 5120 words), at most 16
#define RT_STACK_ROWS_MAX ((RT_LDS_WORDS / RT_BLOCK - 6) < 16 ? (RT_LDS_WORDS / RT_BLOCK - 6) : 16)
#ifndef RT_STACK_ROWS_MIN
#define RT_STACK_ROWS_MIN 8
#endif
#define RT_STACK_LDS 16 // kernels with a plain LDS stack of their own (k_sample_general)
static_assert((RT_STACK_ROWS_MAX + 6) * RT_BLOCK <= RT_LDS_WORDS && RT_STACK_ROWS_MAX >= RT_STACK_ROWS_MIN, "a traversal block's LDS holds the stack rows and the six world-ray rows");
// words of a TLAS copy: per pair 16 (boxes + links) + 12 (reach), per instance 12 (invT rows) + 1 (root link)
#define RT_TLAS_COPY_WORDS(pairs, instances) ((pairs) * 28 + (instances) * 13)
#define RT_STACK_MAX 130  // tlas::Intersect's stack[64] (tlas.cpp:67) + the instance sentinel + bvh::BIntersect's own stack[64]
                          // (bvh.cpp:608) live on ONE stack here; whatever the reference can traverse fits

// Debug build (make EXTRA=-DRT_DEBUG_CHECKS, profiles/debug_checks.sh): index checks on the structures the traversal
// and shading kernels share -- stack pointer and spill index, queue positions, slot numbers, pending-branch slots.
// A failed check stores 100 + its number in the launch's flag word (reported as RT_E_STATE by the host) instead
// of touching memory out of bounds.  Off in the product build: the checks cost registers in the hottest loop.
#ifdef RT_DEBUG_CHECKS
#define RT_CHECK(cond, code, flagptr) do { if (!(cond)) *(flagptr) = 100 + (code); } while (0)
#else
#define RT_CHECK(cond, code, flagptr) do { } while (0)
#endif

struct DLight {
	int kind, objIdx;
	float pos[3], strength, col[3], normal[3], radius, sinAngle;
};
struct DMaterial {
	int type, raytracer;
	float col[3], albedo[3];
	float specu, diffu, shinieness; int N;
	float ir, absorption[3];
};
struct DInstance {
	float invT[12];
	float T[12];
	uint rootLink;
	uint rootWide; // the same BLAS entered through its 4-wide nodes (== rootLink when the root is a leaf)
	uint rootWide8; // ... through its 8-wide quantised nodes
	uint pad[5];
};
struct DScene {
	const float4* pairs;
	const float4* prims;
	const DInstance* inst;
	const float4* brute; // TLAS mode: spheres then planes, prim-record format
	const float4* reach; // TLAS mode: per TLAS pair, the boxes its children's geometry can actually occupy (see below)
	const float4* wide;  // 4-wide nodes of every BLAS (any-hit queries of clean rays), 128-byte records; null when a tree is not nested
	uint rootWide;       // scene BVH root through the wide nodes (unused in TLAS mode: instances carry theirs)
	const uint4* wide8;  // 8-wide nodes with quantised child boxes (any-hit queries of clean rays), 128-byte records, see below; null: not built
	const float4* leafBox; // exact box + first primitive slot of every leaf the 8-wide nodes name, 32-byte records
	uint rootWide8;
	const DLight* lights;
	const DMaterial* mats;
	const unsigned char* sky;
	const float* gammaLut; // [256] pow(b / 255, GAMMA) for a sky texel byte b (the finished sample of a camera ray that leaves the scene)
	uint rootLink; // scene BVH root, or the TLAS root in TLAS mode
	uint tlasBase; // pair index of the first TLAS record
	float reachOriginMax; // reach[] boxes are inflated for world ray origins with |O|_1 up to this
	int tlasPairs; // number of TLAS pair records
	int nInst;     // instances
	int tlasLds;   // every traversal block walks its own LDS copy of the TLAS (pairs, reach records, instance transforms)
	int stackRows; // stack entries per lane kept in LDS (the split of the block's LDS, see RT_LDS_WORDS)
	int useTLAS;
	int nBruteSph, nBrutePla;
	int nLights;
	int nMats;     // entries of mats[]
	int skyW, skyH, skyN;
};

struct DCounters { // mirrors rt_counters
	unsigned long long inner_visits, prim_tests, tlas_inner, instance_visits, rays_nearest, rays_occluded, brute_tests, light_tests;
};
struct LaneCounters {
	uint inner, prim, tlasInner, inst, brute, light;
	__device__ __forceinline__ void clear() { inner = prim = tlasInner = inst = brute = light = 0; }
};

// Per-lane traversal stack: entries [0, rows) live in LDS laid out [entry][lane] (one
// bank per lane, conflict free), deeper entries spill to a per-lane column of a global buffer
// laid out [entry][global lane].  Depth is capped at the reference's 64.
// The two halves are typed by address space: with generic pointers the compiler folds "LDS or spill" into ONE
// flat_load / flat_store on a selected address, and a flat access takes the vector-memory path (the traversal's
// co-limit) even when it lands in LDS.  Typed, a pop is a ds_read_b32 and the spill a rare branch.
typedef __attribute__((address_space(3))) uint lds_uint;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
__device__ __forceinline__ float4 ld_lds(const lds_v4f* p) { const v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }
typedef __attribute__((address_space(1))) uint glb_uint;
struct Stack {
	lds_uint* lds;    // &ldsStack[0][threadIdx.x]
	glb_uint* spill;  // the grid's spill buffer, [entry][global lane] (wave-uniform: the lane's column is added where an entry is touched)
	uint spillStride; // lanes in the grid
	uint rows;        // entries held in LDS
	uint sp;
	int* overflow;
	__device__ __forceinline__ size_t spill_at(uint entry) const { return (size_t)entry * spillStride + (blockIdx.x * blockDim.x + threadIdx.x); }
	__device__ __forceinline__ void push(uint v)
	{
		RT_CHECK(sp <= RT_STACK_MAX, 1, overflow);
		if (__builtin_expect(sp < rows, 1)) lds[sp * RT_BLOCK] = v; // (the hint: the LDS rows are where nearly every entry goes)
		else if (sp < RT_STACK_MAX) spill[spill_at(sp - rows)] = v;
		else { *overflow = 1; sp = 0; return; } // reported as RT_E_OVERFLOW by the host; the ray ends at its next pop instead of walking a wrong stack
		sp++;
	}
	__device__ __forceinline__ uint pop()
	{
		RT_CHECK(sp >= 1 && sp <= RT_STACK_MAX, 2, overflow);
		sp--;
		if (__builtin_expect(sp < rows, 1)) return lds[sp * RT_BLOCK];
		return spill[spill_at(sp - rows)];
	}
};

__device__ __forceinline__ Stack make_stack(uint* ldsBase, uint* spill, int* overflow, uint rows = RT_STACK_ROWS_MAX)
{
	Stack st;
	st.rows = rows;
	st.lds = (lds_uint*)ldsBase + threadIdx.x;
	st.spillStride = gridDim.x * blockDim.x;
	st.spill = (glb_uint*)spill;
	st.sp = 0;
	st.overflow = overflow;
	return st;
}

// What a nearest-hit query tracks while it runs; resolved to normal / ids once at the end.
struct HitRef {
	float t;
	uint prim;  // slot in prims[] (or brute[]), valid when kind >= 0
	int inst;   // instance index or -1
	int kind;   // -1 none, 0 BLAS/scene prim, 1 brute prim, 2 light (prim = light index)
};

__device__ __forceinline__ f3 rcp3(const f3& D) { return f3(1 / D.x, 1 / D.y, 1 / D.z); } // template/scene.h:47

// AreaLight::Intersect (template/scene.h:105-120): overwrites t without comparing it to the
// current one, and evaluates 't - 1e-6' in double.
__device__ __forceinline__ void light_intersect(const DLight& L, int li, const f3& O, const f3& D, float t_min, float& rayT, HitRef& hit)
{
	if (L.kind != 0) return;
	const f3 normal(L.normal[0], L.normal[1], L.normal[2]), pos(L.pos[0], L.pos[1], L.pos[2]);
	float d = dot(normal, D);
	f3 dir = pos - O;
	float t = dot(dir, normal) / d;
	if (t >= t_min) {
		f3 intersection = O + D * t;
		f3 v = intersection - pos;
		float dis2 = dot(v, v);
		if (sqrtf(dis2) <= L.radius) {
			rayT = (float)((double)t - 1e-6);
			hit.t = rayT, hit.kind = 2, hit.prim = (uint)li, hit.inst = -1;
		}
	}
}

// What Scene::FindNearest tests before it enters the BVH / TLAS (template/scene.h:1257-1261): every
// light, then in TLAS mode the brute-force spheres and planes, all with the caller's t_min.  The
// render path runs this where rays are CREATED (shade / finish: dense, all lanes busy) and hands
// extend the shortened rayT plus the candidate hit; the batch queries run it when a lane picks up a ray.
template <bool COUNT>
__device__ __forceinline__ void find_nearest_head(const DScene& S, const f3& O, const f3& D, float t_min, float& rayT, HitRef& hit, LaneCounters& lc)
{
	for (int i = 0; i < S.nLights; i++) {
		light_intersect(S.lights[i], i, O, D, t_min, rayT, hit);
		if (COUNT) lc.light++;
	}
	if (S.useTLAS) {
		const int nb = S.nBruteSph + S.nBrutePla;
		for (int i = 0; i < nb; i++) {
			const float4 r0 = S.brute[4 * i];
			float t;
			const bool h = i < S.nBruteSph ? sphere_hit(O, D, rayT, t_min, xyz(r0), r0.w, t) : plane_hit(O, D, rayT, t_min, xyz(r0), r0.w, t);
			if (COUNT) lc.brute++;
			if (h) rayT = t, hit.kind = 1, hit.prim = (uint)i, hit.inst = -1;
		}
	}
}
// the candidate hit of the head tests as one dword (kept in the ray's D.w): 0 none, kind << 28 | prim + 1
__device__ __forceinline__ uint pack_head(const HitRef& h) { return h.kind < 0 ? 0u : ((uint)h.kind << 28) | (h.prim + 1); }
__device__ __forceinline__ void unpack_head(uint v, HitRef& h)
{
	h.inst = -1;
	if (v == 0) { h.kind = -1, h.prim = 0; return; }
	h.kind = (int)(v >> 28), h.prim = (v & 0x0FFFFFFFu) - 1;
}

// ---- persistent, lane-granular traversal --------------------------------------------------------
// Rays in one wave need very different numbers of node visits, so a wave that walks 64 rays to
// completion idles most of its lanes most of the time (measured: 12 % VALU lane utilisation).
// Here every lane runs a small state machine -- one step = one sibling-pair test, one primitive
// test or one TLAS node -- and a lane whose ray is finished takes the next ray of the queue while
// the others keep walking.  The visiting ORDER per ray is exactly the reference's
// (bvh.cpp:606-656, 763-806; tlas.cpp:65-122; bvhInstance.cpp:3-35): near child first, far child
// pushed, leaf primitives in primitiveIdx order, t_min 0.0001 inside the BVH.
//
// A *policy* supplies the rays and takes the results:
//   bool load(int work, f3& O, f3& D, float& tmax, HitRef& head)   world-space ray of work item 'work' (+ the
//                                                     candidate hit of head tests already made); false = nothing to trace
//   void store(int work, const HitRef&, O, D)         nearest-hit result      (ANY == false)
//   void store(int work, bool occluded)               occlusion result        (ANY == true)
// MIXED == true: one launch serves nearest-hit and any-hit work items side by side (the policy says which a work
// item is: bool any_of(int work)); ANY is then ignored.  One persistent launch per round instead of two means one
// drain per round instead of two (a launch ends when its longest ray does, several hundred microseconds after the
// queue ran dry, whatever the queue held).
#define RT_INST_BIT 0x40000000u // link names an instance (TLAS leaf)
#define RT_BOX_BIT 0x20000000u  // 8-wide walk: link names a leafBox[] record (the leaf's exact box is tested before its primitives)
#define RT_LINK_EXIT 0xFFFFFFFCu // leave the current instance (the sentinel was popped)
#define RT_LINK_DONE 0xFFFFFFFBu // this ray is finished; its result is written at the next refill
// What a lane's link asks for, as ONE unsigned compare each (the scheduler of trace_persistent takes a dozen ballots of these per
// iteration; a ballot of a compare is the compare, a ballot of an AND of conditions is the conditions, a 0 / 1 value in a vector
// register, and a compare of that).  An idle lane (work < 0) holds RT_LINK_DONE, so "live" is link != RT_LINK_DONE.
#define RT_WANTS_PAIR(lk) ((lk) < RT_INST_BIT)                            // neither RT_LEAF_BIT nor RT_INST_BIT (the special links have both)
#define RT_WANTS_ENTER(lk) ((lk) - RT_INST_BIT < RT_INST_BIT)             // RT_INST_BIT without RT_LEAF_BIT
#define RT_WANTS_LEAF(lk) ((lk) - RT_LEAF_BIT < RT_LINK_DONE - RT_LEAF_BIT) // RT_LEAF_BIT, below the special links
static_assert(RT_LEAF_BIT == 0x80000000u && RT_INST_BIT == 0x40000000u && RT_LINK_DONE + 1 == RT_LINK_EXIT, "the link tests above read the encoding this way");
#define RT_CHUNK 256 // queue entries a wave reserves per atomic on a work head (upper bound)
#ifndef RT_CHUNK_MIN
#define RT_CHUNK_MIN 64 // ... and the lower bound, near the end of a sub-queue (a power of two)
#endif
// Work distribution.  Same-address atomics retire at ~88 per microsecond on this part, so ONE work head
// shared by 7000 waves is a real cost: a wave waits in line for every reservation, and large chunks (the
// obvious cure) leave a long tail, because the wave that takes the last 256 rays works on them for
// ~0.6 ms while the rest of the machine idles.  So the queue is cut into RT_HEADS sub-queues with a head
// each (4 KB apart: separate L2 channels); a wave reserves from its home sub-queue, sizes the
// reservation by what is left there (256 entries early on, 64 near the end), and moves on to the other
// sub-queues when its own is empty.
#define RT_TUNE_CULL_COUNTED 0x10000 // the one bit of a traversal kernel's 'tuning' argument: a counting launch drops unreachable TLAS children like a timed one
// The scheduling thresholds of trace_persistent are compile-time constants (they were launch arguments, and environment variables on
// the host, through round 4's sweeps: every sweep since round 2 came out flat around these values, and as constants they leave the
// kernels five scalar registers and their selects -- k_extend_s -2.7 %, profiles/r04_ab_fixed_tuning.txt).  A sweep is a rebuild:
// make EXTRA="-DRT_STEPMIN=12" in a copy of the tree (profiles/r04_bisect.sh).  _ANY: launches of any-hit queries only.
#ifndef RT_REFILL
#define RT_REFILL 16        // free lanes a wave waits for before it flushes them and refills (a policy's kRefill overrides it)
#endif
#ifndef RT_REFILL_ANY
#define RT_REFILL_ANY 24
#endif
#ifndef RT_STEPMIN
#define RT_STEPMIN 8        // lanes that must want a leaf step before it runs (or it is the most wanted kind)
#endif
#ifndef RT_STEPMIN_ANY
#define RT_STEPMIN_ANY 8
#endif
#ifndef RT_STEPMIN_XFORM
#define RT_STEPMIN_XFORM 0  // the same for an instance entry / exit (0: as RT_STEPMIN; 4-6 neutral, less loses: round 2)
#endif
#ifndef RT_PAIRAGAIN
#define RT_PAIRAGAIN 24     // lanes that must still want a pair step for an iteration to repeat it (16 with four repeats)
#endif
#ifndef RT_PAIRAGAIN_ANY
#define RT_PAIRAGAIN_ANY 16
#endif
#ifndef RT_DRAIN_LANES
#define RT_DRAIN_LANES 64   // live lanes of a wave whose queue is dry at which they leave the scheduled machine for the per-lane loop (0: never)
#endif
#ifndef RT_DRAIN_LANES_ANY
#define RT_DRAIN_LANES_ANY 64
#endif
#define RT_HEADS 16
#ifndef RT_NO_DRAIN_LOOP
#define RT_NO_DRAIN_LOOP 0 // measurement builds: 1 keeps the drain in the scheduled machine
#endif
#ifndef RT_SHORT_QUEUE_RAYS
#define RT_SHORT_QUEUE_RAYS 32 // queue entries per wave below which further waves of the grid do not take part
#endif
#ifndef RT_PAIR_REPEAT
#define RT_PAIR_REPEAT 5 // pair steps per iteration at most (unrolled; 4 through round 4: five fit since the kernel has registers to spare, profiles/r04_sweep_repeat.txt)
#endif
#ifndef RT_CONNECT_REPEAT
#define RT_CONNECT_REPEAT 6 // the same for a launch of any-hit queries only (5-8 repeats are within the noise: profiles/r04_sweep_repeat.txt)
#endif
#define RT_HEAD_STRIDE 1024 // ints between two heads
#ifndef RT_HEADS_PROBE
#define RT_HEADS_PROBE 6 // sub-queues a wave finds empty in a row before it stops looking (every probe is one more
                         // same-address atomic per wave at the end of a launch: 16 -> 3 took a 2 M-ray frame from 5.1 to
                         // 4.7 ms, but 3 costs balance on the full frame)
#endif
#ifdef RT_TAIL_PROBE
// measurement build: when does a traversal launch run out of queue entries, and when does its last wave leave?
// [0] first wave in, [1] first wave that finds every sub-queue empty, [2] last wave out (s_memrealtime, 100 MHz)
__device__ unsigned long long g_tailProbe[4];
#endif
#ifdef RT_STEP_COUNT
// measurement build: steps per nearest-hit ray, written per work item at flush time (profiles/step_histogram.sh)
__device__ uint* g_stepOut;
#endif
#ifdef RT_SECTION_PROBE
// measurement build: where do a traversal wave's cycles go?  Shader-clock cycles per section, summed over all waves:
// [0] flush + refill  [1] pair: loads issued -> data there  [2] pair: arithmetic + stack  [3] leaf: wait  [4] leaf: rest
// [5] enter  [6] exit  [7] whole loop; step counts: [8] iterations [9] pair [10] leaf [11] enter [12] exit [13] refills
__device__ unsigned long long g_sectionProbe[16];
// (the clock is wave-uniform; the sums live in LDS, one set per wave, added to by the first enabled lane, so that probe
// points inside divergent code cost no vector registers)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
#define RT_SEC_NOW() __builtin_readcyclecounter()
#define RT_SEC_PUT(k, v) do { const unsigned long long secV = (v); if (bits_below(__ballot(true)) == 0) secAcc[k] += secV; } while (0)
#define RT_SEC_ADD(k, t0) RT_SEC_PUT(k, __builtin_readcyclecounter() - (t0))
#define RT_SEC_COUNT(k) RT_SEC_PUT(k, 1ull)
#define RT_SEC_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define RT_SEC_NOW() 0ull
#define RT_SEC_ADD(k, t0) ((void)(t0))
#define RT_SEC_COUNT(k) ((void)0)
#define RT_SEC_WAIT() ((void)0)
#endif
// WIDE == true (occlusion queries only, never counting launches): inside a BLAS / the scene BVH the walk uses the
// 4-wide nodes (wide[], see the layout notes above); a ray that is not clean is given back through
// pol.leftover(work) for the binary walk.  The TLAS level keeps its pair records and reach test.
// A policy with 'static constexpr bool kAdvance = true' keeps its lane when a query ends: instead of store(), the flush calls
//   bool advance(int work, bool wasAny, const HitRef& result, f3& O, f3& D, float& tmax, HitRef& head, bool& nextAny)
// with the finished query's result (nearest: result.t / kind / prim / inst; any-hit: result.kind == 1 means occluded); true:
// the lane goes on with the world-space ray (O, D, tmax, head candidate) as a nearest-hit or (nextAny) any-hit query of the
// SAME work item; false: the work item is complete and the lane is free.  Needs MIXED (the lane's kind of query changes).
// Such a policy may also offer 'bool starts_done()': true after load() means the query just loaded needs no walk (its result is
// what load() left in the head candidate) and goes to the next flush as it is.
template <class P, class = void> struct pol_starts_done { static constexpr bool value = false; };
template <class P> struct pol_starts_done<P, decltype((void)&P::starts_done)> { static constexpr bool value = true; };
template <class P, class = void> struct pol_advances { static constexpr bool value = false; };
template <class P> struct pol_advances<P, decltype((void)P::kAdvance)> { static constexpr bool value = P::kAdvance; };
// 'static constexpr int kRefill': the policy's own refill threshold (the Whitted launches hand out tiles, not rays)
template <class P, class = void> struct pol_refill { static constexpr int value = 0; };
template <class P> struct pol_refill<P, decltype((void)P::kRefill)> { static constexpr int value = P::kRefill; };

template <bool ANY, bool COUNT, bool HEAD, class Policy, bool MIXED = false, int WIDTH = 2>
__device__ __forceinline__ void trace_persistent(const DScene& S, Policy& pol, int n, int* heads, float t_min, int tuning,
                                                 uint* ldsStack, uint* spill, int* overflow, LaneCounters& lc, uint& rays)
{
	// WIDTH: 2 the reference's binary walk; 4 / 8: inside a BLAS / the scene BVH the walk uses the 4-wide nodes (wide[]) or the 8-wide
	// quantised ones (wide8[]) -- occlusion queries only, never counting launches
	static_assert(WIDTH == 2 || WIDTH == 4 || WIDTH == 8, "binary, 4-wide or 8-wide");
	static_assert(WIDTH == 2 || (ANY && !COUNT && !MIXED), "a wide walk is exact for any-hit queries only");
	constexpr bool WIDE = WIDTH == 4, WIDE8 = WIDTH == 8;
	constexpr bool ANYWIDE = WIDE || WIDE8; // a wide walk: rays that are not clean go back to the binary walk
	constexpr int REPEAT = ANY && !MIXED ? RT_CONNECT_REPEAT : RT_PAIR_REPEAT; // pair steps per iteration at most
	constexpr bool DRAINS = !ANYWIDE && !pol_advances<Policy>::value && !RT_NO_DRAIN_LOOP; // the last rays of a wave leave the machine for a plain per-lane loop
	const float bvh_t_min = 0.0001f; // bvh.cpp:607, :764
	constexpr bool ANYQ = ANY && !MIXED; // a launch of any-hit queries only
	constexpr int refillMin = pol_refill<Policy>::value ? pol_refill<Policy>::value : (ANYQ ? RT_REFILL_ANY : RT_REFILL);
	constexpr int stepMinBusy = ANYQ ? RT_STEPMIN_ANY : RT_STEPMIN, pairAgainBusy = ANYQ ? RT_PAIRAGAIN_ANY : RT_PAIRAGAIN;
	constexpr int stepMinXformBusy = RT_STEPMIN_XFORM ? RT_STEPMIN_XFORM : stepMinBusy; // entry / exit: arithmetic and LDS only when the TLAS is in LDS
	constexpr int drainLanes = ANYQ ? RT_DRAIN_LANES_ANY : RT_DRAIN_LANES;
	// (lane and the bits below it are recomputed where they are used: two instructions there instead of three registers kept live
	// across the hottest loop of the library, which runs at eight waves per SIMD: 64 registers)
#define lane (threadIdx.x & 63)
	auto bits_below = [](unsigned long long m) __attribute__((always_inline)) { return (int)__builtin_amdgcn_mbcnt_hi((uint)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint)m, 0u)); }; // set bits of m below this lane
	// sub-queue h = [h * subLen, (h + 1) * subLen) cut at n; all of this is wave-uniform (SGPRs)
	const int waveId = (int)(blockIdx.x * (blockDim.x >> 6)) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int subLen = ((n + RT_HEADS - 1) / RT_HEADS + 63) & ~63;
	int wavesPerHead = (int)((gridDim.x * blockDim.x) >> 6) / RT_HEADS;
	if (wavesPerHead < 1) wavesPerHead = 1;
	int home = waveId % RT_HEADS;   // sub-queue this wave is drawing from
	int tried = 0;                  // sub-queues found empty since the last successful reservation
	int chunk = subLen / (wavesPerHead * 2); // size of the next reservation
	chunk = chunk >= RT_CHUNK ? RT_CHUNK : (chunk <= RT_CHUNK_MIN ? RT_CHUNK_MIN : (chunk & ~(RT_CHUNK_MIN - 1)));
	const uint stackRows = (uint)S.stackRows;
	Stack st = make_stack(ldsStack, spill, overflow, stackRows);
	// the world-space ray while the lane walks a BLAS: rows [stackRows, stackRows + 6) of the context's LDS column, [row][lane]
#define worldRay (st.lds + stackRows * RT_BLOCK)
	// A small TLAS is walked in LDS.  The traversal is bound by the vector-memory path (lane accesses through the
	// texture addresser and L1); a TLAS visit is seven of them (pair + reach record) and an
	// instance entry four, together a quarter of all accesses on the bench scene; ds_read takes another pipe.
	// Layout: [pair][4] | [pair][3] reach | [instance][3] invT rows 0-2 | [instance] root link (rootWide for the wide walk)
	lds_v4f* const tlasL = (lds_v4f*)((lds_uint*)ldsStack + (stackRows + 6) * RT_BLOCK);
	if (S.tlasLds) {
		const int nP = S.tlasPairs * 4, nR = S.tlasPairs * 3, nI = S.nInst * 3;
		for (int i = (int)threadIdx.x; i < nP + nR + nI; i += RT_BLOCK) {
			v4f v;
			if (i < nP) v = ((const v4f*)S.pairs)[4 * (size_t)S.tlasBase + i];
			else if (i < nP + nR) v = ((const v4f*)S.reach)[i - nP];
			else { const int k = i - nP - nR; v = ((const v4f*)S.inst)[(size_t)(k / 3) * 8 + k % 3]; }
			tlasL[i] = v;
		}
		lds_uint* const roots = (lds_uint*)(tlasL + nP + nR + nI);
		for (int i = (int)threadIdx.x; i < S.nInst; i += RT_BLOCK) roots[i] = WIDE8 ? S.inst[i].rootWide8 : (WIDE ? S.inst[i].rootWide : S.inst[i].rootLink);
		__syncthreads();
	}
#ifdef RT_TAIL_PROBE
	if ((threadIdx.x & 63) == 0) atomicMin(&g_tailProbe[0], __builtin_amdgcn_s_memrealtime());
	bool probed = false;
#endif
#ifdef RT_SECTION_PROBE
	__shared__ unsigned long long secLds[(RT_BLOCK / 64) * 16];
	lds_u64* const secAcc = (lds_u64*)secLds + (threadIdx.x >> 6) * 16;
	if (lane < 16) secAcc[lane] = 0;
	const unsigned long long secStart = RT_SEC_NOW();
#endif
#ifdef RT_STEP_COUNT
	uint nsteps = 0, nenter = 0;
#endif
	int work = -1;           // queue entry this lane is tracing, -1 = idle
	int chunkNext = 0, chunkEnd = 0; // wave-uniform: reserved, not yet handed out
	// A short queue does not need the whole grid: a wave beyond one per RT_SHORT_QUEUE_RAYS entries (and beyond one per
	// head) leaves at once instead of finding every sub-queue empty one same-address atomic at a time.
	bool exhausted = n <= 0 || (waveId >= RT_HEADS && (long long)waveId * RT_SHORT_QUEUE_RAYS >= (long long)n); // wave-uniform: the queue has no more entries (for this wave)
	f3 O(0.0f), D(0.0f), rD(0.0f);
	float rayT = 0;
	uint link = RT_LINK_DONE;
	HitRef hit;
	hit.kind = -1, hit.inst = -1, hit.prim = 0, hit.t = 0;
	int inst = -1;
	bool laneAny = ANY;  // MIXED: this lane's work item is an occlusion query
	bool clean = false; // the current (world or object space) ray cannot produce a NaN slab product

	// next node for this lane: pop the stack; an empty stack ends the ray, the sentinel leaves the instance
	auto pop_next = [&]() __attribute__((always_inline)) {
		if (st.sp == 0) { link = RT_LINK_DONE; return; }
		const uint v = st.pop();
		link = v == RT_SENTINEL ? RT_LINK_EXIT : v;
	};

	// one primitive of a leaf (bvh.cpp:616-629 / :770-783) on its record, for the lanes enabled
	auto leaf_test = [&](uint lk, const float4& r0, const float4& r1, const float4& r2, const float4& r3) __attribute__((always_inline)) {
		const uint slot = lk & ~RT_LEAF_BIT;
		const int kl = __float_as_int(r3.w);
		const int kind = kl & 3;
		if (COUNT) lc.prim++;
		float t;
		bool h;
		if (kind == RT_KIND_TRI) h = tri_hit(O, D, rayT, bvh_t_min, xyz(r0), xyz(r1), xyz(r2), f3(r0.w, r1.w, r2.w), r3.x, t);
		else if (kind == RT_KIND_SPHERE) {
			if (MIXED ? laneAny : ANY) h = sphere_occludes(O, D, rayT, bvh_t_min, xyz(r0), r0.w);
			else h = sphere_hit(O, D, rayT, bvh_t_min, xyz(r0), r0.w, t);
		} else h = plane_hit(O, D, rayT, bvh_t_min, xyz(r0), r0.w, t);
		if (h && (MIXED ? laneAny : ANY)) hit.kind = 1, link = RT_LINK_DONE; // first occluder ends the query
		else {
			if (h) rayT = t, hit.prim = slot, hit.inst = inst, hit.kind = 0;
			if (kl & RT_LAST_BIT) pop_next();
			else link = lk + 1;
		}
	};
	// ---- the four kinds of step live in rt_step_*.inc: the code of one step for the lanes enabled where the file is included, on the
	// lane's state as it is named here (lk: the link the step is for).  Text inclusion, not lambdas: the scheduled machine below and
	// the drain loop behind it make the same steps, and with closures the compiler's register allocation of the hot loop moved
	// (k_extend_s: 0 -> 20 bytes of scratch at its 72 registers, which costs 10 % of the kernel; profiles/r04_ab_drain_loop.txt).
	while (true) {
		// ---- flush finished lanes and refill, once enough lanes have nothing to do ----
		bool doneLane = work >= 0 && link == RT_LINK_DONE;
		unsigned long long freeMask = __ballot(link == RT_LINK_DONE); // lanes that can take a new work item: idle ones (they hold RT_LINK_DONE too) and finished ones
		RT_SEC_COUNT(8);
		if (freeMask != 0) {
			const int cnt = __popcll(freeMask);
			const bool nothingLeft = freeMask == ~0ull;
			// (once the queue is dry there is nothing to hand out: the block is for finished lanes only -- a wave's last rays then
			// skip it on all the iterations in which none of them finished)
			if ((cnt >= refillMin || nothingLeft) && (!exhausted || (__ballot(work >= 0) & freeMask) != 0)) {
				const unsigned long long secT = RT_SEC_NOW();
				RT_SEC_COUNT(13);
				if (doneLane) {
					RT_CHECK(work >= 0 && work < n && st.sp <= RT_STACK_MAX, 4, overflow);
#ifdef RT_STEP_COUNT
					if constexpr (!ANY && !MIXED) { if (g_stepOut) g_stepOut[pol.slot_of(work)] = (nsteps & 0xFFFFu) | (nenter << 16); }
					nsteps = 0, nenter = 0;
#endif
					// results are written here, many lanes at a time, not one lane per iteration
					if constexpr (pol_advances<Policy>::value) {
						static_assert(MIXED, "an advancing policy changes its lane's kind of query");
						HitRef res = hit;
						res.t = rayT;
						float tmax = 0;
						bool nextAny = false;
						hit.kind = -1, hit.prim = 0, hit.inst = -1;
						if (pol.advance(work, laneAny, res, O, D, tmax, hit, nextAny)) {
							laneAny = nextAny, rayT = tmax;
							st.sp = 0, inst = -1;
							rD = rcp3(D);
							clean = ray_is_clean(O, D, rD);
							link = S.rootLink;
							if (link == RT_EMPTY) link = RT_LINK_DONE;
							rays++;
						} else work = -1;
					} else {
						if constexpr (MIXED) {
							if (laneAny) pol.store(work, hit.kind == 1);
							else { hit.t = rayT; pol.store(work, hit, O, D); }
						} else if constexpr (ANY) pol.store(work, hit.kind == 1);
						else { hit.t = rayT; pol.store(work, hit, O, D); }
						work = -1;
					}
				}
				int cntFree = cnt; // lanes to hand new work to
				if constexpr (pol_advances<Policy>::value) {
					freeMask = __ballot(work < 0); // a lane that went on with its work item is not free
					cntFree = __popcll(freeMask);
				}
				if (!exhausted) {
					while (chunkNext >= chunkEnd && !exhausted) {
						const int lo = home * subLen, hi = lo + subLen < n ? lo + subLen : n;
						int base = subLen; // "empty"
						if (lo < hi) {
							if (lane == 0) base = atomicAdd(heads + home * RT_HEAD_STRIDE, chunk);
							base = __builtin_amdgcn_readfirstlane(base); // lane 0's value, in an SGPR: everything derived from it stays scalar
						}
						if (lo + base < hi) {
							chunkNext = lo + base, chunkEnd = chunkNext + chunk < hi ? chunkNext + chunk : hi;
							// the next reservation: a share of what is left in this sub-queue
							int next = (hi - chunkEnd) / (wavesPerHead * 2);
							chunk = next >= RT_CHUNK ? RT_CHUNK : (next <= RT_CHUNK_MIN ? RT_CHUNK_MIN : (next & ~(RT_CHUNK_MIN - 1)));
							tried = 0;
						} else {
							// this sub-queue is empty: help with the next one, in small pieces
							home = home + 1 == RT_HEADS ? 0 : home + 1;
							chunk = RT_CHUNK_MIN;
							if (++tried == RT_HEADS_PROBE) exhausted = true;
						}
					}
					if (!exhausted) {
						const int mine = chunkNext + bits_below(freeMask);
						const int avail = chunkEnd - chunkNext;
						float tmax = 0;
						const bool take = mine < chunkEnd && ((freeMask >> lane) & 1);
						RT_CHECK(!take || (mine >= 0 && mine < n), 3, overflow);
						if (take) hit.kind = -1, hit.prim = 0, hit.inst = -1;
						// a work item may turn out to be nothing to trace: the lane stays idle
						if (take && pol.load(mine, O, D, tmax, hit)) {
							work = mine;
							if constexpr (MIXED) laneAny = pol.any_of(mine);
							rayT = tmax;
							st.sp = 0, inst = -1;
							if (!ANY && HEAD) find_nearest_head<COUNT>(S, O, D, t_min, rayT, hit, lc);
							rD = rcp3(D);
							clean = ray_is_clean(O, D, rD);
							link = ANYWIDE && !S.useTLAS ? (WIDE8 ? S.rootWide8 : S.rootWide) : S.rootLink;
							if (link == RT_EMPTY) link = RT_LINK_DONE;
							if constexpr (pol_starts_done<Policy>::value) { if (pol.starts_done()) link = RT_LINK_DONE; } // the policy answered the query itself
							rays++;
							if constexpr (ANYWIDE) { if (!clean) { pol.leftover(mine); work = -1, link = RT_LINK_DONE; } } // the binary walk answers this one
						}
						chunkNext += cntFree < avail ? cntFree : avail;
					}
				}
				RT_SEC_WAIT();
				RT_SEC_ADD(0, secT);
			}
		}
#ifdef RT_TAIL_PROBE
		if (exhausted && !probed) { probed = true; if (lane == 0) atomicMin(&g_tailProbe[1], __builtin_amdgcn_s_memrealtime()); }
#endif
		const unsigned long long stepMask = __ballot(link != RT_LINK_DONE);
		if (stepMask == 0) {
			if (exhausted && __ballot(work >= 0) == 0) break;
			continue; // only finished lanes left (they flush above), or nothing was handed out this time
		}

		// ---- the drain: the last rays of a wave leave the machine (see after the loop) ----
		if constexpr (DRAINS) { if (exhausted && __popcll(stepMask) <= drainLanes) break; }

		// ---- one iteration of the state machine ----
		// Four kinds of step, four pieces of code.  A kind that only a few lanes want this iteration
		// is postponed until at least stepMin lanes want it or it is the most wanted kind, so its
		// instructions run with more lanes enabled; postponed lanes just wait.  Pair steps are what most
		// lanes want most of the time (34 of 64 on the bench scene, against 9 / 5 / 6 waiting for a leaf,
		// an entry or an exit), so an iteration repeats the pair step while at least pairAgain lanes still
		// want one: the bookkeeping around the steps is paid once for up to RT_PAIR_REPEAT pair steps,
		// and the rarer kinds find more lanes waiting when their turn comes.
		// Once the queue is dry no new lane will ever join a postponed kind: waiting for company only stretches the
		// dependent chains of the last rays (the drain of the launch), so every wanted kind runs in every iteration.
		const int stepMin = exhausted ? 1 : stepMinBusy, pairAgain = exhausted ? 1 : pairAgainBusy, stepMinXform = exhausted ? 1 : stepMinXformBusy;
#pragma unroll
		for (int rep = 0; rep < REPEAT; rep++) {
			const uint lk = link;
			const bool wantPair = RT_WANTS_PAIR(lk);
			const int nP = __popcll(__ballot(wantPair));
			if (nP == 0) break;
			if (rep == 0) {
				if (nP < stepMin) {
					// fewer than stepMin: only if nothing else is wanted more
					const int nL = __popcll(__ballot(RT_WANTS_LEAF(lk)));
					const int nN = __popcll(__ballot(RT_WANTS_ENTER(lk)));
					const int nE = __popcll(__ballot(lk == RT_LINK_EXIT));
					if (nP < nL || nP < nN || nP < nE) break;
				}
			} else if (nP < pairAgain) break;
			if (wantPair) {
#ifdef RT_STEP_COUNT
				nsteps++;
#endif
#include "rt_step_pair.inc"
			}
		}
		// the other kinds, on the links as they are now
		uint lk = link;
		bool live = lk != RT_LINK_DONE;
		bool wantLeaf = RT_WANTS_LEAF(lk);
		bool wantExit = lk == RT_LINK_EXIT;
		bool wantEnter = RT_WANTS_ENTER(lk);
		const int nL = __popcll(__ballot(wantLeaf)), nP = __popcll(__ballot(RT_WANTS_PAIR(lk)));
		const int nN = __popcll(__ballot(wantEnter)), nE = __popcll(__ballot(wantExit));
		int most = nL > nP ? nL : nP;
		most = nN > most ? nN : most;
		most = nE > most ? nE : most;
		const bool runLeaf = nL > 0 && (nL >= stepMin || nL == most);
		const bool runEnter = nN > 0 && (nN >= stepMinXform || nN == most);
		const bool runExit = nE > 0 && (nE >= stepMinXform || nE == most);
		if (most == 0 && __ballot(live) != 0) {
			// live lanes whose link no step kind understands (a corrupt tree): nothing would ever change again.  Every
			// wave must reach its exit, so the rays are dropped and the launch reports RT_E_STATE.
			*overflow = 199;
			if (live) link = RT_LINK_DONE;
			continue;
		}

#ifdef RT_STEP_COUNT
		if ((runLeaf && wantLeaf) || (runEnter && wantEnter) || (runExit && wantExit)) nsteps++;
		if (runEnter && wantEnter) nenter++;
#endif
		if (runLeaf && wantLeaf) {
#include "rt_step_leaf.inc"
		}
		if (runEnter && wantEnter) {
#include "rt_step_enter.inc"
		}
		if (runExit && wantExit) {
#include "rt_step_exit.inc"
		}
	}
	// ---- the drain: the last rays of a wave walk alone ----
	// Once the queue is dry and only a few lanes still hold a ray, the scheduled machine above is mostly overhead: a wave-64
	// instruction takes its issue slots whatever the number of enabled lanes, and an iteration of the machine is ~300 of them
	// (ballots, thresholds, the unrolled repeats) for the one or two steps it makes -- measured: a lone lane's step takes
	// 0.61-0.65 us and does not wait for memory (DESIGN.md finding 52).  So these lanes leave the loop for good (nothing will be
	// handed out any more) and each walks its ray to the end in the classic per-lane loop: one plain step after the other, ~100
	// instructions per step.  Every ray makes the steps it would make in the machine, in the same order: same bits.  The code sits
	// BEHIND the loop on purpose: inside it (tried first) the hot loop of k_extend_s went from 0 to 92 bytes of scratch at its 72
	// registers.  The wide walks and the policies that keep a lane for the next query (advance) stay in the machine.
	if constexpr (DRAINS) {
		while (work >= 0 && link != RT_LINK_DONE) {
			const uint lk = link;
#ifdef RT_STEP_COUNT
			nsteps++;
#endif
			if (!(lk & (RT_LEAF_BIT | RT_INST_BIT))) {
#include "rt_step_pair.inc"
			} else if (lk == RT_LINK_EXIT) {
#include "rt_step_exit.inc"
			} else if (lk < RT_LINK_EXIT && (lk & RT_LEAF_BIT)) {
#include "rt_step_leaf.inc"
			}
			else if (!(lk & RT_LEAF_BIT)) {
#ifdef RT_STEP_COUNT
				nenter++;
#endif
#include "rt_step_enter.inc"
			} else { *overflow = 199; link = RT_LINK_DONE; } // a link no step understands (a corrupt tree): reported as RT_E_STATE
		}
		if (work >= 0) {
#ifdef RT_STEP_COUNT
			if constexpr (!ANY && !MIXED) { if (g_stepOut) g_stepOut[pol.slot_of(work)] = (nsteps & 0xFFFFu) | (nenter << 16); }
#endif
			if constexpr (MIXED) {
				if (laneAny) pol.store(work, hit.kind == 1);
				else { hit.t = rayT; pol.store(work, hit, O, D); }
			} else if constexpr (ANY) pol.store(work, hit.kind == 1);
			else { hit.t = rayT; pol.store(work, hit, O, D); }
		}
	}
#ifdef RT_TAIL_PROBE
	if (lane == 0) atomicMax(&g_tailProbe[2], __builtin_amdgcn_s_memrealtime());
#endif
#ifdef RT_SECTION_PROBE
	if (lane == 0) secAcc[7] = RT_SEC_NOW() - secStart;
	if (lane < 14) atomicAdd(&g_sectionProbe[lane], (unsigned long long)secAcc[lane]);
#endif
}

#undef worldRay
#undef lane

// ---- one ray at a time (the general path-mode kernel) ----------------------------------------------
// The same walk as trace_persistent for a single ray held by one lane: used by k_sample_general, where
// a lane runs a whole sample and queries are interleaved with random draws.
template <bool ANY>
__device__ __forceinline__ bool trace_one(const DScene& S, const f3& Ow, const f3& Dw, float& rayT, HitRef& hit, Stack& st)
{
	const float bvh_t_min = 0.0001f;
	f3 O = Ow, D = Dw, rD = rcp3(Dw);
	int inst = -1;
	uint link = S.rootLink;
	st.sp = 0;
	if (link == RT_EMPTY) return false;
	while (true) {
		bool needPop = false;
		if (link == RT_LINK_EXIT) {
			O = Ow, D = Dw, rD = rcp3(Dw), inst = -1;
			needPop = true;
		} else if (link & RT_LEAF_BIT) {
			const uint slot = link & ~RT_LEAF_BIT;
			const float4* rec = S.prims + 4 * (size_t)slot;
			const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
			const int kl = __float_as_int(r3.w);
			const int kind = kl & 3;
			float t;
			bool h;
			if (kind == RT_KIND_TRI) h = tri_hit(O, D, rayT, bvh_t_min, xyz(r0), xyz(r1), xyz(r2), f3(r0.w, r1.w, r2.w), r3.x, t);
			else if (kind == RT_KIND_SPHERE) {
				if (ANY) h = sphere_occludes(O, D, rayT, bvh_t_min, xyz(r0), r0.w);
				else h = sphere_hit(O, D, rayT, bvh_t_min, xyz(r0), r0.w, t);
			} else h = plane_hit(O, D, rayT, bvh_t_min, xyz(r0), r0.w, t);
			if (h) {
				if (ANY) return true;
				rayT = t, hit.prim = slot, hit.inst = inst, hit.kind = 0;
			}
			if (kl & RT_LAST_BIT) needPop = true;
			else link++;
		} else if (link & RT_INST_BIT) {
			inst = (int)(link & ~RT_INST_BIT);
			const DInstance* I = S.inst + inst;
			const f3 Oo = xform_pos(I->invT, O), Do = xform_vec(I->invT, D);
			O = Oo, D = Do, rD = rcp3(Do);
			link = I->rootLink;
			if (link == RT_EMPTY) link = RT_LINK_EXIT;
			else st.push(RT_SENTINEL);
		} else {
			const float4* p = S.pairs + 4 * (size_t)link;
			const float4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
			float dist1 = intersect_aabb_exact(O, rD, rayT, xyz(a0), xyz(a1));
			float dist2 = intersect_aabb_exact(O, rD, rayT, xyz(b0), xyz(b1));
			uint c1 = __float_as_uint(a0.w), c2 = __float_as_uint(b0.w);
			if (dist1 > dist2) { float td = dist1; dist1 = dist2; dist2 = td; uint tc = c1; c1 = c2; c2 = tc; }
			if (dist1 == 1e30f) needPop = true;
			else {
				link = c1;
				if (dist2 != 1e30f) st.push(c2);
			}
		}
		if (needPop) {
			if (st.sp == 0) return false;
			const uint v = st.pop();
			link = v == RT_SENTINEL ? RT_LINK_EXIT : v;
		}
	}
}
// Scene::FindNearest / Scene::IsOccluded for one ray
__device__ __forceinline__ void find_nearest_one(const DScene& S, const f3& O, const f3& D, float rayT, float t_min, HitRef& hit, Stack& st)
{
	LaneCounters unused;
	hit.kind = -1, hit.inst = -1, hit.prim = 0;
	find_nearest_head<false>(S, O, D, t_min, rayT, hit, unused);
	trace_one<false>(S, O, D, rayT, hit, st);
	hit.t = rayT;
}
__device__ __forceinline__ bool is_occluded_one(const DScene& S, const f3& O, const f3& D, float rayT, Stack& st)
{
	HitRef unused;
	return trace_one<true>(S, O, D, rayT, unused, st);
}

// Fill the fields FindNearest leaves in the Ray: objIdx, material, hitNormal (for the sphere the
// normal is (P - pos) * invr at the accepted t, template/scene.h:361; for an instanced triangle
// normalize(TransformVector(N, matTransform)), bvhInstance.cpp:19).
// 'ray' is called only for a sphere (it is the one primitive whose normal needs the ray), so a caller that
// would have to fetch the ray from memory does not do so for the triangles, planes and lights.
// matType: rt_material::type of the hit primitive's material as the record carries it (0 for a miss, a light, or unknown)
template <class RayFn>
__device__ __forceinline__ void resolve_hit_lazy(const DScene& S, const HitRef& hit, RayFn ray, int& objIdx, int& mat, f3& normal, int* matType = nullptr)
{
	objIdx = -1, mat = -1, normal = f3(0.0f);
	if (matType) *matType = 0;
	if (hit.kind < 0) return;
	if (hit.kind == 2) {
		const DLight& L = S.lights[hit.prim];
		objIdx = L.objIdx, normal = f3(L.normal[0], L.normal[1], L.normal[2]);
		return;
	}
	const float4* rec = (hit.kind == 1 ? S.brute : S.prims) + 4 * (size_t)hit.prim;
	const float4 r0 = rec[0], r3 = rec[3];
	objIdx = __float_as_int(r3.y), mat = __float_as_int(r3.z);
	const int kind = __float_as_int(r3.w) & 3;
	if (matType) *matType = (__float_as_int(r3.w) >> RT_TYPE_SHIFT) & 3;
	if (kind == RT_KIND_TRI) {
		// (N.y and N.z alone, not the vectors they ride in: this runs in the traversal kernels' flush, where every register counts)
		normal = f3(r0.w, ((const float*)rec)[7], ((const float*)rec)[11]);
		if (hit.inst >= 0) normal = normalize(xform_vec(S.inst[hit.inst].T, normal));
	} else if (kind == RT_KIND_SPHERE) {
		const float4 r1 = rec[1];
		f3 O, D;
		ray(O, D);
		normal = ((O + hit.t * D) - xyz(r0)) * r1.x;
	} else {
		normal = xyz(r0);
	}
}
__device__ __forceinline__ void resolve_hit(const DScene& S, const HitRef& hit, const f3& O, const f3& D, int& objIdx, int& mat, f3& normal, int* matType = nullptr)
{
	resolve_hit_lazy(S, hit, [&](f3& o, f3& d) { o = O, d = D; }, objIdx, mat, normal, matType);
}

} // namespace rtd
