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
//   tlas[]   the reference's 32-byte TLASNode records (tlas.h:4-11), 2 float4 each.
//   inst[]   128-byte records: invTransform rows 0-2, matTransform rows 0-2, root link.
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
#define RT_MAX_LIGHTS 8

#define RT_BLOCK 256
#define RT_STACK_LDS 16   // stack entries per lane held in LDS
#define RT_STACK_MAX 64   // the reference's stack[64] (bvh.cpp:608, tlas.cpp:67)

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
	uint pad[7];
};
struct DScene {
	const float4* pairs;
	const float4* prims;
	const float4* tlas;
	const DInstance* inst;
	const float4* brute; // TLAS mode: spheres then planes, prim-record format
	const DLight* lights;
	const DMaterial* mats;
	const unsigned char* sky;
	uint rootLink; // non-TLAS scene BVH
	int useTLAS;
	int nBruteSph, nBrutePla;
	int nLights;
	int skyW, skyH, skyN;
};

struct DCounters { // mirrors rt_counters
	unsigned long long inner_visits, prim_tests, tlas_inner, instance_visits, rays_nearest, rays_occluded, brute_tests, light_tests;
};
struct LaneCounters {
	uint inner, prim, tlasInner, inst, brute, light;
	__device__ __forceinline__ void clear() { inner = prim = tlasInner = inst = brute = light = 0; }
};

// Per-lane traversal stack: entries [0, RT_STACK_LDS) live in LDS laid out [entry][lane] (one
// bank per lane, conflict free), deeper entries spill to a per-lane column of a global buffer
// laid out [entry][global lane].  Depth is capped at the reference's 64.
struct Stack {
	uint* lds;        // &ldsStack[0][threadIdx.x]
	uint* spill;      // &spill[0][global lane]
	uint spillStride; // lanes in the grid
	uint sp;
	int* overflow;
	__device__ __forceinline__ void push(uint v)
	{
		if (sp < RT_STACK_LDS) lds[sp * RT_BLOCK] = v;
		else if (sp < RT_STACK_MAX) spill[(size_t)(sp - RT_STACK_LDS) * spillStride] = v;
		else { *overflow = 1; return; }
		sp++;
	}
	__device__ __forceinline__ uint pop()
	{
		sp--;
		return sp < RT_STACK_LDS ? lds[sp * RT_BLOCK] : spill[(size_t)(sp - RT_STACK_LDS) * spillStride];
	}
};

// What a nearest-hit query tracks while it runs; resolved to normal / ids once at the end.
struct HitRef {
	float t;
	uint prim;  // slot in prims[] (or brute[]), valid when kind >= 0
	int inst;   // instance index or -1
	int kind;   // -1 none, 0 BLAS/scene prim, 1 brute prim, 2 light (prim = light index)
};

__device__ __forceinline__ f3 rcp3(const f3& D) { return f3(1 / D.x, 1 / D.y, 1 / D.z); } // template/scene.h:47

// Leaf loop shared by both traversals (bvh.cpp:616-629 / :770-783).  ANY: return true at the first
// occluder.  t_min is the BVH's hard-coded 0.0001f (bvh.cpp:607, :764).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool leaf_prims(const float4* __restrict__ prims, uint first, const f3& O, const f3& D, float& rayT,
                                           HitRef& hit, int inst, LaneCounters& lc)
{
	const float t_min = 0.0001f;
	uint slot = first;
	while (true) {
		const float4 r0 = prims[4 * slot + 0];
		const float4 r3 = prims[4 * slot + 3];
		const int kl = __float_as_int(r3.w);
		const int kind = kl & 3;
		if (COUNT) lc.prim++;
		float t;
		bool h;
		if (kind == RT_KIND_TRI) {
			const float4 r1 = prims[4 * slot + 1];
			const float4 r2 = prims[4 * slot + 2];
			h = tri_hit(O, D, rayT, t_min, xyz(r0), xyz(r1), xyz(r2), f3(r0.w, r1.w, r2.w), r3.x, t);
		} else if (kind == RT_KIND_SPHERE) {
			if (ANY) h = sphere_occludes(O, D, rayT, t_min, xyz(r0), r0.w);
			else h = sphere_hit(O, D, rayT, t_min, xyz(r0), r0.w, t);
		} else {
			h = plane_hit(O, D, rayT, t_min, xyz(r0), r0.w, t);
		}
		if (h) {
			if (ANY) return true;
			rayT = t, hit.t = t, hit.prim = slot, hit.inst = inst, hit.kind = 0;
		}
		if (kl & RT_LAST_BIT) break;
		slot++;
	}
	return false;
}

// bvh::BIntersect / BIsOccluded (bvh.cpp:606-656, 763-806) on one BLAS, entered with an empty
// sub-stack (stack.sp == base).  Ordered traversal: near child first, far child pushed when hit.
// Returns true (ANY only) when an occluder was found.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse_blas(const DScene& S, uint link, const f3& O, const f3& D, const f3& rD, float& rayT,
                                              HitRef& hit, int inst, Stack& st, uint base, LaneCounters& lc)
{
	if (link == RT_EMPTY) return false;
	while (true) {
		if (link & RT_LEAF_BIT) {
			if (leaf_prims<ANY, COUNT>(S.prims, link & ~RT_LEAF_BIT, O, D, rayT, hit, inst, lc)) return true;
			if (st.sp == base) return false;
			link = st.pop();
			continue;
		}
		if (COUNT) lc.inner++;
		const float4* p = S.pairs + 4 * (size_t)link;
		const float4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
		float dist1 = intersect_aabb(O, rD, rayT, xyz(a0), xyz(a1));
		float dist2 = intersect_aabb(O, rD, rayT, xyz(b0), xyz(b1));
		uint c1 = __float_as_uint(a0.w), c2 = __float_as_uint(b0.w);
		if (dist1 > dist2) { float td = dist1; dist1 = dist2; dist2 = td; uint tc = c1; c1 = c2; c2 = tc; }
		if (dist1 == 1e30f) {
			if (st.sp == base) return false;
			link = st.pop();
		} else {
			link = c1;
			if (dist2 != 1e30f) st.push(c2);
		}
	}
}

// tlas::Intersect / IsOccluded (tlas.cpp:65-122) with bvhInstance::BIntersect / IsOccluded
// (bvhInstance.cpp:3-35) inlined: the ray is taken to object space with invTransform (direction not
// renormalised, so t is shared by both spaces), the BLAS is walked on the same stack above the
// TLAS entries, and the world-space ray is re-derived from (Ow, Dw) on the way out.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse_tlas(const DScene& S, const f3& Ow, const f3& Dw, const f3& rDw, float& rayT,
                                              HitRef& hit, Stack& st, LaneCounters& lc)
{
	uint node = 0;
	const uint base = st.sp;
	while (true) {
		const float4 n0 = S.tlas[2 * node], n1 = S.tlas[2 * node + 1];
		const uint leftRight = __float_as_uint(n0.w);
		if (leftRight == 0) {
			const int inst = (int)__float_as_uint(n1.w);
			if (COUNT) lc.inst++;
			const DInstance* I = S.inst + inst;
			const f3 O = xform_pos(I->invT, Ow);
			const f3 D = xform_vec(I->invT, Dw);
			const f3 rD = rcp3(D);
			if (traverse_blas<ANY, COUNT>(S, I->rootLink, O, D, rD, rayT, hit, inst, st, st.sp, lc)) return true;
			if (st.sp == base) return false;
			node = st.pop();
			continue;
		}
		if (COUNT) lc.tlasInner++;
		uint c1 = leftRight & 0xFFFFu, c2 = leftRight >> 16;
		const float4 a0 = S.tlas[2 * c1], a1 = S.tlas[2 * c1 + 1];
		const float4 b0 = S.tlas[2 * c2], b1 = S.tlas[2 * c2 + 1];
		float dist1 = intersect_aabb(Ow, rDw, rayT, xyz(a0), xyz(a1));
		float dist2 = intersect_aabb(Ow, rDw, rayT, xyz(b0), xyz(b1));
		if (dist1 > dist2) { float td = dist1; dist1 = dist2; dist2 = td; uint tc = c1; c1 = c2; c2 = tc; }
		if (dist1 == 1e30f) {
			if (st.sp == base) return false;
			node = st.pop();
		} else {
			node = c1;
			if (dist2 != 1e30f) st.push(c2);
		}
	}
}

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

// Scene::FindNearest (template/scene.h:1248-1267)
template <bool COUNT>
__device__ __forceinline__ void find_nearest(const DScene& S, const f3& O, const f3& D, float rayT, float t_min, HitRef& hit, Stack& st, LaneCounters& lc)
{
	hit.kind = -1, hit.inst = -1, hit.prim = 0, hit.t = rayT;
	for (int i = 0; i < S.nLights; i++) {
		light_intersect(S.lights[i], i, O, D, t_min, rayT, hit);
		if (COUNT) lc.light++;
	}
	const f3 rD = rcp3(D);
	if (S.useTLAS) {
		const int nb = S.nBruteSph + S.nBrutePla;
		for (int i = 0; i < nb; i++) {
			const float4 r0 = S.brute[4 * i];
			float t;
			bool h = i < S.nBruteSph ? sphere_hit(O, D, rayT, t_min, xyz(r0), r0.w, t) : plane_hit(O, D, rayT, t_min, xyz(r0), r0.w, t);
			if (COUNT) lc.brute++;
			if (h) rayT = t, hit.t = t, hit.kind = 1, hit.prim = (uint)i, hit.inst = -1;
		}
		traverse_tlas<false, COUNT>(S, O, D, rD, rayT, hit, st, lc);
	} else {
		traverse_blas<false, COUNT>(S, S.rootLink, O, D, rD, rayT, hit, -1, st, st.sp, lc);
	}
	hit.t = rayT;
}

// Scene::IsOccluded(Ray&) (template/scene.h:1286-1291)
template <bool COUNT>
__device__ __forceinline__ bool is_occluded(const DScene& S, const f3& O, const f3& D, float rayT, Stack& st, LaneCounters& lc)
{
	HitRef dummy;
	const f3 rD = rcp3(D);
	if (S.useTLAS) return traverse_tlas<true, COUNT>(S, O, D, rD, rayT, dummy, st, lc);
	return traverse_blas<true, COUNT>(S, S.rootLink, O, D, rD, rayT, dummy, -1, st, st.sp, lc);
}

// Fill the fields FindNearest leaves in the Ray: objIdx, material, hitNormal (for the sphere the
// normal is (P - pos) * invr at the accepted t, template/scene.h:361; for an instanced triangle
// normalize(TransformVector(N, matTransform)), bvhInstance.cpp:19).
__device__ __forceinline__ void resolve_hit(const DScene& S, const HitRef& hit, const f3& O, const f3& D, int& objIdx, int& mat, f3& normal)
{
	objIdx = -1, mat = -1, normal = f3(0.0f);
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
	if (kind == RT_KIND_TRI) {
		const float4 r1 = rec[1], r2 = rec[2];
		normal = f3(r0.w, r1.w, r2.w);
		if (hit.inst >= 0) normal = normalize(xform_vec(S.inst[hit.inst].T, normal));
	} else if (kind == RT_KIND_SPHERE) {
		const float4 r1 = rec[1];
		normal = ((O + hit.t * D) - xyz(r0)) * r1.x;
	} else {
		normal = xyz(r0);
	}
}

} // namespace rtd
