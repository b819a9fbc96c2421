// ORACLE (test infrastructure, NOT product code) -- parity unpinned, see oracle/README.md.
//
// CPU restatement of the reference's acceleration structures: bvh (bvh.cpp), bvhInstance
// (bvhInstance.cpp) and tlas (tlas.cpp), builders and traversals, including the builder quirks
// Q1-Q5 of SURVEY.md section 7 because the stored boxes decide which primitive a ray reports.
#pragma once
#include "orc_scene.h"
#include <algorithm>
#include <tuple>

namespace orc {

struct BVHNode { // bvh.h:10-24, 32 bytes
	float3 aabbMin; uint leftFirst;
	float3 aabbMax; uint primCount;
};

enum SplitMethod { BINNEDSAH = 0, SAMESIZE = 1, LONGESTAXIS = 2, SAH = 3 }; // bvh.h:38-43

// bvh::IntersectAABB (bvh.cpp:819-828)
static inline float IntersectAABB(const Ray& ray, const float3& bmin, const float3& bmax)
{
	float tx1 = (bmin.x - ray.O.x) * ray.rD.x, tx2 = (bmax.x - ray.O.x) * ray.rD.x;
	float tmin = std_min(tx1, tx2), tmax = std_max(tx1, tx2);
	float ty1 = (bmin.y - ray.O.y) * ray.rD.y, ty2 = (bmax.y - ray.O.y) * ray.rD.y;
	tmin = std_max(tmin, std_min(ty1, ty2)), tmax = std_min(tmax, std_max(ty1, ty2));
	float tz1 = (bmin.z - ray.O.z) * ray.rD.z, tz2 = (bmax.z - ray.O.z) * ray.rD.z;
	tmin = std_max(tmin, std_min(tz1, tz2)), tmax = std_min(tmax, std_max(tz1, tz2));
	if (tmax >= tmin && tmin < ray.t && tmax > 0) return tmin; else return 1e30f;
}

struct bvh {
	const Scene* scene = nullptr; // bvh(Scene*): all triangles, spheres and planes of the scene
	const Mesh* mesh = nullptr;   // bvh(Mesh*): one mesh's triangles
	uint rootNodeIdx = 0, nodesUsed = 2, NTri = 0, NSph = 0, NPla = 0, N = 0; // bvh.h:75 (Q4: node 1 unused)
	std::vector<uint> primitiveIdx;
	std::vector<BVHNode> bvhNode;
	aabb bounds;
	int splitMethod = BINNEDSAH; // bvh.cpp:7, 13
	int maxDepth = 0;            // deepest leaf, counted in nodes from the root (diagnostic)

	explicit bvh(const Scene* s) : scene(s) {}
	explicit bvh(const Mesh* m) : mesh(m) {}

	const Triangle& getTriangle(uint idx) const { return scene ? scene->getTriangle(idx) : mesh->tri[idx]; } // :58-65

	void Build() // :18-56 (the QBVH branch is disabled in the reference, template/scene.h:704-707)
	{
		if (scene) { NTri = scene->getTriangleNb(); NSph = (uint)scene->spheres.size(); NPla = (uint)scene->planes.size(); }
		else { NTri = (uint)mesh->tri.size(); NSph = 0; NPla = 0; }
		N = NTri + NSph + NPla;
		primitiveIdx.resize(N);
		bvhNode.assign(2 * (N + 1) - 1 + 1, BVHNode{ float3(0), 0, float3(0), 0 });
		for (uint i = 0; i < N; ++i) primitiveIdx[i] = i;
		nodesUsed = 2;
		BVHNode& root = bvhNode[rootNodeIdx];
		root.primCount = N;
		root.leftFirst = 0;
		UpdateNodeBounds(rootNodeIdx);
		separatePlanes(rootNodeIdx);
		bounds.grow(bvhNode[rootNodeIdx].aabbMin);
		bounds.grow(bvhNode[rootNodeIdx].aabbMax);
		Refit();
	}

	void UpdateNodeBounds(uint nodeIdx) // :67-114
	{
		BVHNode& node = bvhNode[nodeIdx];
		node.aabbMin = float3(1e30f);
		node.aabbMax = float3(-1e30f);
		for (uint first = node.leftFirst, i = 0; i < node.primCount; i++) {
			uint leafIdx = primitiveIdx[first + i];
			if (leafIdx < NTri) {
				const Triangle& leafTri = getTriangle(leafIdx);
				node.aabbMin = t_fminf(node.aabbMin, leafTri.v0);
				node.aabbMin = t_fminf(node.aabbMin, leafTri.v1);
				node.aabbMin = t_fminf(node.aabbMin, leafTri.v2);
				node.aabbMax = t_fmaxf(node.aabbMax, leafTri.v0);
				node.aabbMax = t_fmaxf(node.aabbMax, leafTri.v1);
				node.aabbMax = t_fmaxf(node.aabbMax, leafTri.v2);
			} else if (leafIdx >= NTri && leafIdx < NTri + NSph) {
				leafIdx -= NTri;
				const Sphere& leafSph = scene->spheres[leafIdx];
				node.aabbMin = t_fminf(node.aabbMin, leafSph.pos - float3(leafSph.r));
				node.aabbMax = t_fmaxf(node.aabbMax, leafSph.pos + float3(leafSph.r));
			} else {
				leafIdx -= NTri + NSph;
				const Plane& leafPla = scene->planes[leafIdx];
				float3 normal = normalize(leafPla.N);
				// Q1: only +X/+Y/+Z planes get a slab, placed at coordinate 0 whatever 'd' is
				if (normal.x + normal.y + normal.z == 1 && (normal.x == 1 || normal.y == 1 || normal.z == 1)) {
					if (normal.x == 1) {
						node.aabbMin = t_fminf(node.aabbMin, float3(0, -1e30f, -1e30f));
						node.aabbMax = t_fmaxf(node.aabbMax, float3(0, 1e30f, 1e30f));
					} else if (normal.y == 1) {
						node.aabbMin = t_fminf(node.aabbMin, float3(-1e30f, 0, -1e30f));
						node.aabbMax = t_fmaxf(node.aabbMax, float3(1e30f, 0, 1e30f));
					} else if (normal.z == 1) {
						node.aabbMin = t_fminf(node.aabbMin, float3(-1e30f, -1e30f, 0));
						node.aabbMax = t_fmaxf(node.aabbMax, float3(1e30f, 1e30f, 0));
					}
				} else {
					node.aabbMin = float3(-1e30f);
					node.aabbMax = float3(1e30f);
					return;
				}
			}
		}
	}

	struct Bin { aabb bounds; int primCount = 0; }; // bvh.h:88

	float FindBestSplitPlane(const BVHNode& node, int& axis, float& splitPos) const // :116-193
	{
		const int BINS = 8;
		float bestCost = 1e30f;
		for (int a = 0; a < 3; a++) {
			float boundsMin = 1e30f, boundsMax = -1e30f;
			for (uint i = 0; i < node.primCount; i++) {
				uint primIdx = primitiveIdx[node.leftFirst + i];
				if (primIdx < NTri) {
					const Triangle& triangle = getTriangle(primIdx);
					boundsMin = std_min(boundsMin, triangle.centroid[a]);
					boundsMax = std_max(boundsMax, triangle.centroid[a]);
				} else if (primIdx >= NTri && primIdx < NTri + NSph) {
					primIdx -= NTri;
					const Sphere& sphere = scene->spheres[primIdx];
					boundsMin = std_min(boundsMin, sphere.pos[a]);
					boundsMax = std_max(boundsMax, sphere.pos[a]);
				}
			}
			if (boundsMin == boundsMax) continue;
			Bin bin[BINS];
			float scale = BINS / (boundsMax - boundsMin);
			for (uint i = 0; i < node.primCount; i++) {
				uint primIdx = primitiveIdx[node.leftFirst + i];
				int binIdx;
				if (primIdx < NTri) {
					const Triangle& triangle = getTriangle(primIdx);
					binIdx = std::min(BINS - 1, f2i((triangle.centroid[a] - boundsMin) * scale));
					bin[binIdx].primCount++;
					bin[binIdx].bounds.grow(triangle.v0);
					bin[binIdx].bounds.grow(triangle.v1);
					bin[binIdx].bounds.grow(triangle.v2);
				} else if (primIdx >= NTri && primIdx < NTri + NSph) {
					primIdx -= NTri;
					const Sphere& sphere = scene->spheres[primIdx];
					binIdx = std::min(BINS - 1, f2i((sphere.pos[a] - boundsMin) * scale));
					bin[binIdx].primCount++;
					bin[binIdx].bounds.grow(sphere.pos - float3(2 * sphere.r)); // Q3: bins grow by 2r
					bin[binIdx].bounds.grow(sphere.pos + float3(2 * sphere.r));
				}
			}
			float leftArea[BINS - 1], rightArea[BINS - 1];
			int leftCount[BINS - 1], rightCount[BINS - 1];
			aabb leftBox, rightBox;
			int leftSum = 0, rightSum = 0;
			for (int i = 0; i < BINS - 1; i++) {
				leftSum += bin[i].primCount;
				leftCount[i] = leftSum;
				leftBox.grow(bin[i].bounds);
				leftArea[i] = leftBox.area();
				rightSum += bin[BINS - 1 - i].primCount;
				rightCount[BINS - 2 - i] = rightSum;
				rightBox.grow(bin[BINS - 1 - i].bounds);
				rightArea[BINS - 2 - i] = rightBox.area();
			}
			scale = (boundsMax - boundsMin) / BINS;
			for (int i = 0; i < BINS - 1; i++) {
				float planeCost = leftCount[i] * leftArea[i] + rightCount[i] * rightArea[i];
				if (planeCost < bestCost)
					axis = a, splitPos = boundsMin + scale * (i + 1), bestCost = planeCost;
			}
		}
		return bestCost;
	}

	float CalculateNodeCost(const BVHNode& node) const // :196-200
	{
		float3 e = node.aabbMax - node.aabbMin;
		float surfaceArea = e.x * e.y + e.y * e.z + e.z * e.x;
		return node.primCount * surfaceArea;
	}

	float EvaluateSAH(const BVHNode& node, int axis, float pos) const // :514-554
	{
		aabb leftBox, rightBox;
		int leftCount = 0, rightCount = 0;
		for (uint i = 0; i < node.primCount; i++) {
			uint primIdx = primitiveIdx[node.leftFirst + i];
			if (primIdx < NTri) {
				const Triangle& triangle = getTriangle(primIdx);
				if (triangle.centroid[axis] < pos) {
					leftCount++;
					leftBox.grow(triangle.v0); leftBox.grow(triangle.v1); leftBox.grow(triangle.v2);
				} else {
					rightCount++;
					rightBox.grow(triangle.v0); rightBox.grow(triangle.v1); rightBox.grow(triangle.v2);
				}
			} else if (primIdx >= NTri && primIdx < NTri + NSph) { // the reference tests '< N' and would index past the spheres for a plane
				primIdx -= NTri;
				const Sphere& sphere = scene->spheres[primIdx];
				// the reference grows by the scalar pos[axis] -/+ r broadcast to all three axes
				if (sphere.pos[axis] < pos) {
					leftCount++;
					leftBox.grow(sphere.pos[axis] - float3(sphere.r));
					leftBox.grow(sphere.pos[axis] + float3(sphere.r));
				} else {
					rightCount++;
					rightBox.grow(sphere.pos[axis] - float3(sphere.r));
					rightBox.grow(sphere.pos[axis] + float3(sphere.r));
				}
			}
		}
		float cost = leftCount * leftBox.area() + rightCount * rightBox.area();
		return cost > 0 ? cost : 1e30f;
	}

	void separatePlanes(uint nodeIdx) // :202-221 (Q2)
	{
		if (NPla > 0 && (NSph + NTri > 0)) {
			int leftChildIdx = nodesUsed++;
			int rightChildIdx = nodesUsed++;
			bvhNode[leftChildIdx].leftFirst = 0;
			bvhNode[leftChildIdx].primCount = NTri + NSph;
			bvhNode[rightChildIdx].leftFirst = NTri + NSph;
			bvhNode[rightChildIdx].primCount = NPla;
			bvhNode[nodeIdx].leftFirst = leftChildIdx;
			bvhNode[nodeIdx].primCount = 0;
			UpdateNodeBounds(leftChildIdx);
			UpdateNodeBounds(rightChildIdx);
			Subdivide(leftChildIdx, 2);
		} else {
			Subdivide(nodeIdx, 1);
		}
	}

	void Subdivide(uint nodeIdx, int depth) // :223-333
	{
		if (depth > maxDepth) maxDepth = depth;
		int axis = 0; float splitPos = 0;
		{
			const BVHNode& node = bvhNode[nodeIdx];
			switch (splitMethod) {
			case BINNEDSAH: {
				float splitCost = FindBestSplitPlane(node, axis, splitPos);
				float nosplitCost = CalculateNodeCost(node);
				if (splitCost >= nosplitCost) return;
				break;
			}
			case LONGESTAXIS: {
				float3 extent = node.aabbMax - node.aabbMin;
				axis = 0;
				if (extent.y > extent.x) axis = 1;
				if (extent.z > extent[axis]) axis = 2;
				splitPos = node.aabbMin[axis] + extent[axis] * 0.5f;
				break;
			}
			case SAMESIZE: {
				float3 extent = node.aabbMax - node.aabbMin;
				axis = 0;
				if (extent.y > extent.x) axis = 1;
				if (extent.z > extent[axis]) axis = 2;
				int m = node.primCount / 2;
				std::vector<std::tuple<float, int>> sorted;
				for (uint i = 0; i < node.primCount; i++) {
					uint primIdx = primitiveIdx[node.leftFirst + i];
					if (primIdx < NTri) sorted.push_back(std::make_tuple(getTriangle(primIdx).centroid[axis], (int)primIdx));
					else if (primIdx >= NTri && primIdx < NTri + NSph) { primIdx -= NTri; sorted.push_back(std::make_tuple(scene->spheres[primIdx].pos[axis], (int)primIdx)); }
				}
				std::sort(sorted.begin(), sorted.end());
				splitPos = std::get<0>(sorted[m]);
				break;
			}
			case SAH: {
				int bestAxis = -1;
				float bestPos = 0, bestCost = 1e30f;
				float candidatePos = 0;
				for (int a = 0; a < 3; a++) for (uint i = 0; i < node.primCount; i++) {
					uint primIdx = primitiveIdx[node.leftFirst + i];
					if (primIdx < NTri) candidatePos = getTriangle(primIdx).centroid[a];
					else if (primIdx >= NTri && primIdx < NTri + NSph) { primIdx -= NTri; candidatePos = scene->spheres[primIdx].pos[a]; }
					float splitCost = EvaluateSAH(node, a, candidatePos);
					if (splitCost < bestCost) bestPos = candidatePos, bestAxis = a, bestCost = splitCost;
				}
				// the reference indexes centroid[-1] when no candidate has a finite cost; defined as "no split"
				if (bestAxis < 0) return;
				axis = bestAxis;
				splitPos = bestPos;
				break;
			}
			}
		}
		// in-place partition (:296-313)
		const uint first = bvhNode[nodeIdx].leftFirst, count = bvhNode[nodeIdx].primCount;
		int i = first;
		int j = i + count - 1;
		while (i <= j) {
			uint primIdx = primitiveIdx[i];
			if (primIdx < NTri) {
				if (getTriangle(primIdx).centroid[axis] < splitPos) i++;
				else std::swap(primitiveIdx[i], primitiveIdx[j--]);
			} else if (primIdx >= NTri && primIdx < NTri + NSph) {
				primIdx -= NTri;
				if (scene->spheres[primIdx].pos[axis] < splitPos) i++;
				else std::swap(primitiveIdx[i], primitiveIdx[j--]);
			} else {
				// a plane: the reference ('< N') indexes past its sphere array here; defined as "goes right"
				std::swap(primitiveIdx[i], primitiveIdx[j--]);
			}
		}
		int leftCount = i - first;
		if (leftCount == 0 || leftCount == (int)count) return;
		int leftChildIdx = nodesUsed++;
		int rightChildIdx = nodesUsed++;
		bvhNode[leftChildIdx].leftFirst = first;
		bvhNode[leftChildIdx].primCount = leftCount;
		bvhNode[rightChildIdx].leftFirst = i;
		bvhNode[rightChildIdx].primCount = count - leftCount;
		bvhNode[nodeIdx].leftFirst = leftChildIdx;
		bvhNode[nodeIdx].primCount = 0;
		UpdateNodeBounds(leftChildIdx);
		UpdateNodeBounds(rightChildIdx);
		Subdivide(leftChildIdx, depth + 1);
		Subdivide(rightChildIdx, depth + 1);
	}

	void Refit() // :556-594
	{
		for (int i = nodesUsed - 1; i >= 0; i--) if (i != 1) {
			BVHNode& node = bvhNode[i];
			if (node.primCount > 0) { UpdateNodeBounds(i); continue; }
			const BVHNode& leftChild = bvhNode[node.leftFirst];
			const BVHNode& rightChild = bvhNode[node.leftFirst + 1];
			node.aabbMin = t_fminf(leftChild.aabbMin, rightChild.aabbMin);
			node.aabbMax = t_fmaxf(leftChild.aabbMax, rightChild.aabbMax);
		}
	}

	// leaf dispatch shared by both traversals (:616-629, :770-783)
	inline void leafIntersect(uint primIdx, Ray& ray, float t_min) const
	{
		if (primIdx < NTri) getTriangle(primIdx).Intersect(ray, t_min);
		else if (primIdx >= NTri && primIdx < NTri + NSph) scene->spheres[primIdx - NTri].Intersect(ray, t_min);
		else scene->planes[primIdx - (NTri + NSph)].Intersect(ray, t_min);
	}
	inline bool leafOccludes(uint primIdx, const Ray& ray, float t_min) const
	{
		if (primIdx < NTri) return getTriangle(primIdx).IsOccluding(ray, t_min);
		else if (primIdx >= NTri && primIdx < NTri + NSph) return scene->spheres[primIdx - NTri].IsOccluding(ray, t_min);
		else return scene->planes[primIdx - (NTri + NSph)].IsOccluding(ray, t_min);
	}

	// bvh::BIntersect (:606-656): ordered closest-hit traversal, t_min fixed at 0.0001
	void Intersect(Ray& ray, Counters& cnt) const
	{
		float t_min = 0.0001f;
		uint node = rootNodeIdx, stack[64];
		uint stackPtr = 0;
		while (1) {
			const BVHNode& n = bvhNode[node];
			if (n.primCount > 0) {
				for (uint i = 0; i < n.primCount; i++) {
					leafIntersect(primitiveIdx[n.leftFirst + i], ray, t_min);
					cnt.prim_tests++;
					if (primitiveIdx[n.leftFirst + i] < NTri) cnt.tri_intersect_calls++;
				}
				if (stackPtr == 0) break; else node = stack[--stackPtr];
				continue;
			}
			cnt.inner_visits++;
			uint c1 = n.leftFirst, c2 = n.leftFirst + 1;
			float dist1 = IntersectAABB(ray, bvhNode[c1].aabbMin, bvhNode[c1].aabbMax);
			float dist2 = IntersectAABB(ray, bvhNode[c2].aabbMin, bvhNode[c2].aabbMax);
			if (dist1 > dist2) { std::swap(dist1, dist2); std::swap(c1, c2); }
			if (dist1 == 1e30f) {
				if (stackPtr == 0) break; else node = stack[--stackPtr];
			} else {
				node = c1;
				if (dist2 != 1e30f) stack[stackPtr++] = c2;
			}
		}
	}

	// bvh::BIsOccluded (:763-806): same order, first occluder wins
	bool IsOccluded(const Ray& ray, Counters& cnt) const
	{
		float t_min = 0.0001f;
		uint node = rootNodeIdx, stack[64];
		uint stackPtr = 0;
		while (1) {
			const BVHNode& n = bvhNode[node];
			if (n.primCount > 0) {
				for (uint i = 0; i < n.primCount; i++) {
					cnt.prim_tests++;
					if (leafOccludes(primitiveIdx[n.leftFirst + i], ray, t_min)) return true;
				}
				if (stackPtr == 0) return false; else node = stack[--stackPtr];
				continue;
			}
			cnt.inner_visits++;
			uint c1 = n.leftFirst, c2 = n.leftFirst + 1;
			float dist1 = IntersectAABB(ray, bvhNode[c1].aabbMin, bvhNode[c1].aabbMax);
			float dist2 = IntersectAABB(ray, bvhNode[c2].aabbMin, bvhNode[c2].aabbMax);
			if (dist1 > dist2) { std::swap(dist1, dist2); std::swap(c1, c2); }
			if (dist1 == 1e30f) {
				if (stackPtr == 0) return false; else node = stack[--stackPtr];
			} else {
				node = c1;
				if (dist2 != 1e30f) stack[stackPtr++] = c2;
			}
		}
	}
};

// bvhInstance (bvhInstance.h, bvhInstance.cpp)
struct bvhInstance {
	const bvh* blas = nullptr;
	int blasIdx = -1;
	mat4 invTransform, matTransform;
	aabb bounds; // world space; union(local, transformed) because it is never reset (Q5)

	bvhInstance() = default;
	explicit bvhInstance(const bvh* b) : blas(b) // bvhInstance.h:9
	{
		mat4 ident;
		SetTransform(ident);
		bounds = blas->bounds;
	}
	void SetTransform(const mat4& transform) // bvhInstance.cpp:37-44
	{
		invTransform = transform.Inverted();
		matTransform = transform;
		float3 bmin = blas->bounds.bmin, bmax = blas->bounds.bmax;
		for (int i = 0; i < 8; i++)
			bounds.grow(TransformPosition(float3(i & 1 ? bmax.x : bmin.x, i & 2 ? bmax.y : bmin.y, i & 4 ? bmax.z : bmin.z), transform));
	}
	void BIntersect(Ray& ray, Counters& cnt) const // bvhInstance.cpp:3-21
	{
		Ray backupRay = ray;
		ray.O = TransformPosition(ray.O, invTransform);
		ray.D = TransformVector(ray.D, invTransform);
		ray.rD = float3(1 / ray.D.x, 1 / ray.D.y, 1 / ray.D.z);
		blas->Intersect(ray, cnt);
		if (backupRay.t > ray.t) {
			backupRay.mat = ray.mat;
			backupRay.t = ray.t;
			backupRay.objIdx = ray.objIdx;
			backupRay.hitNormal = normalize(TransformVector(ray.hitNormal, matTransform));
		}
		ray = backupRay;
	}
	bool IsOccluded(const Ray& rayIn, Counters& cnt) const // bvhInstance.cpp:23-35
	{
		Ray ray = rayIn;
		ray.O = TransformPosition(ray.O, invTransform);
		ray.D = TransformVector(ray.D, invTransform);
		ray.rD = float3(1 / ray.D.x, 1 / ray.D.y, 1 / ray.D.z);
		return blas->IsOccluded(ray, cnt);
	}
};

struct TLASNode { // tlas.h:4-11, 32 bytes
	float3 aabbMin; uint leftRight;
	float3 aabbMax; uint BLAS;
	bool isLeaf() const { return leftRight == 0; }
};

struct tlas {
	std::vector<TLASNode> tlasNode;
	uint nodesUsed = 0;
	std::vector<bvhInstance*> blas;
	uint blasCount = 0;

	tlas(const std::vector<bvhInstance*>& list) : blas(list), blasCount((uint)list.size()) // tlas.cpp:3-11
	{
		tlasNode.assign(2 * blasCount + 1, TLASNode{ float3(0), 0, float3(0), 0 });
		nodesUsed = 2;
	}
	int FindBestMatch(const int* list, int Ncount, int A) const // tlas.cpp:50-63
	{
		float smallest = 1e30f;
		int bestB = -1;
		for (int B = 0; B < Ncount; B++) if (B != A) {
			float3 bmax = t_fmaxf(tlasNode[list[A]].aabbMax, tlasNode[list[B]].aabbMax);
			float3 bmin = t_fminf(tlasNode[list[A]].aabbMin, tlasNode[list[B]].aabbMin);
			float3 e = bmax - bmin;
			float surfaceArea = e.x * e.y + e.y * e.z + e.z * e.x;
			if (surfaceArea < smallest) smallest = surfaceArea, bestB = B;
		}
		return bestB;
	}
	bool build() // tlas.cpp:13-48; the reference's scratch array caps the instance count at 256
	{
		if (blasCount == 0 || blasCount > 256) return false;
		int nodeIdx[256], nodeIndices = blasCount;
		nodesUsed = 1;
		for (uint i = 0; i < blasCount; i++) {
			nodeIdx[i] = nodesUsed;
			tlasNode[nodesUsed].aabbMin = blas[i]->bounds.bmin;
			tlasNode[nodesUsed].aabbMax = blas[i]->bounds.bmax;
			tlasNode[nodesUsed].BLAS = i;
			tlasNode[nodesUsed++].leftRight = 0;
		}
		int A = 0, B = FindBestMatch(nodeIdx, nodeIndices, A);
		while (nodeIndices > 1) {
			int C = FindBestMatch(nodeIdx, nodeIndices, B);
			if (A == C) {
				int nodeIdxA = nodeIdx[A], nodeIdxB = nodeIdx[B];
				const TLASNode& nodeA = tlasNode[nodeIdxA];
				const TLASNode& nodeB = tlasNode[nodeIdxB];
				TLASNode& newNode = tlasNode[nodesUsed];
				newNode.leftRight = nodeIdxA + (nodeIdxB << 16);
				newNode.aabbMin = t_fminf(nodeA.aabbMin, nodeB.aabbMin);
				newNode.aabbMax = t_fmaxf(nodeA.aabbMax, nodeB.aabbMax);
				nodeIdx[A] = nodesUsed++;
				nodeIdx[B] = nodeIdx[nodeIndices - 1];
				B = FindBestMatch(nodeIdx, --nodeIndices, A);
			} else A = B, B = C;
		}
		tlasNode[0] = tlasNode[nodeIdx[A]];
		return true;
	}
	void Intersect(Ray& ray, Counters& cnt) const // tlas.cpp:65-92
	{
		uint node = 0, stack[64];
		uint stackPtr = 0;
		while (1) {
			const TLASNode& n = tlasNode[node];
			if (n.isLeaf()) {
				cnt.instance_visits++;
				blas[n.BLAS]->BIntersect(ray, cnt);
				if (stackPtr == 0) break; else node = stack[--stackPtr];
				continue;
			}
			cnt.tlas_inner++;
			uint child1 = n.leftRight & 0x0000FFFF;
			uint child2 = n.leftRight >> 16;
			float dist1 = IntersectAABB(ray, tlasNode[child1].aabbMin, tlasNode[child1].aabbMax);
			float dist2 = IntersectAABB(ray, tlasNode[child2].aabbMin, tlasNode[child2].aabbMax);
			if (dist1 > dist2) { std::swap(dist1, dist2); std::swap(child1, child2); }
			if (dist1 == 1e30f) {
				if (stackPtr == 0) break; else node = stack[--stackPtr];
			} else {
				node = child1;
				if (dist2 != 1e30f) stack[stackPtr++] = child2;
			}
		}
	}
	bool IsOccluded(const Ray& ray, Counters& cnt) const // tlas.cpp:94-122
	{
		uint node = 0, stack[64];
		uint stackPtr = 0;
		while (1) {
			const TLASNode& n = tlasNode[node];
			if (n.isLeaf()) {
				cnt.instance_visits++;
				if (blas[n.BLAS]->IsOccluded(ray, cnt)) return true;
				if (stackPtr == 0) break; else node = stack[--stackPtr];
				continue;
			}
			cnt.tlas_inner++;
			uint child1 = n.leftRight & 0x0000FFFF;
			uint child2 = n.leftRight >> 16;
			float dist1 = IntersectAABB(ray, tlasNode[child1].aabbMin, tlasNode[child1].aabbMax);
			float dist2 = IntersectAABB(ray, tlasNode[child2].aabbMin, tlasNode[child2].aabbMax);
			if (dist1 > dist2) { std::swap(dist1, dist2); std::swap(child1, child2); }
			if (dist1 == 1e30f) {
				if (stackPtr == 0) break; else node = stack[--stackPtr];
			} else {
				node = child1;
				if (dist2 != 1e30f) stack[stackPtr++] = child2;
			}
		}
		return false;
	}
};

// Scene::SetTime (template/scene.h:1228-1244): every mesh vertex is rotated about z by an angle
// proportional to its own height, triangles are re-derived (Mesh::update, Triangle::update) and the
// BVH is refitted.  The deformation is a pure function of the ORIGINAL vertex, so applying it per
// triangle corner gives the values the reference computes per shared vertex.  sinf/cosf as f64-
// rounded-once (orc_math.h).
inline void Scene::SetTime(float t)
{
	float r = fmodf(t, 2 * PI);
	float a = x_sinf(r) * 0.5f;
	for (auto& m : meshes) {
		if (m.orig.size() != m.tri.size()) m.orig = m.tri;
		for (size_t i = 0; i < m.tri.size(); i++) {
			const Triangle& o = m.orig[i];
			float3 v[3] = { o.v0, o.v1, o.v2 };
			for (int k = 0; k < 3; k++) {
				float sft = a * v[k].y * 0.2f;
				float x = v[k].x * x_cosf(sft) - v[k].y * x_sinf(sft);
				float y = v[k].x * x_sinf(sft) + v[k].y * x_cosf(sft);
				v[k] = float3(x, y, v[k].z);
			}
			m.tri[i] = Triangle(o.objIdx, o.mat, v[0], v[1], v[2]);
		}
	}
	if (b) b->Refit();
}

// Scene::FindNearest (template/scene.h:1248-1267)
inline void Scene::FindNearest(Ray& ray, float t_min, Counters& cnt) const
{
	cnt.rays_nearest++;
	ray.objIdx = -1;
	for (size_t i = 0; i < lights.size(); ++i) { lights[i].Intersect(ray, t_min); cnt.light_tests++; }
	if (useTLAS) {
		for (size_t i = 0; i < spheres.size(); ++i) { spheres[i].Intersect(ray, t_min); cnt.brute_tests++; }
		for (size_t i = 0; i < planes.size(); ++i) { planes[i].Intersect(ray, t_min); cnt.brute_tests++; }
		tl->Intersect(ray, cnt);
	} else {
		b->Intersect(ray, cnt);
	}
}
// Scene::IsOccluded(Ray&) (template/scene.h:1286-1291); in TLAS mode the brute-force spheres
// and planes cast no shadows
inline bool Scene::IsOccluded(Ray& ray, Counters& cnt) const
{
	cnt.rays_occluded++;
	if (useTLAS) return tl->IsOccluded(ray, cnt);
	else return b->IsOccluded(ray, cnt);
}

} // namespace orc
