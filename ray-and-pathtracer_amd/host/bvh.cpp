// Host-side acceleration-structure builders: bvh (bvh.cpp:18-333, 514-594), bvhInstance
// (bvhInstance.cpp:37-44) and tlas (tlas.cpp:3-63) of the reference, re-implemented over flat
// primitive views with an explicit work stack instead of recursion.  The output contract is the
// reference's: identical node numbering, boxes and primitiveIdx order (the traversal kernels
// reproduce the reference's visiting order on top of it, which is what makes hit ids agree).
#include "rapt.h"
#include <algorithm>
#include <stdexcept>
#include <tuple>

namespace rapt {

// ---- math that needs a translation unit ----------------------------------------------------
mat4 operator*(const mat4& a, const mat4& b)
{
	mat4 r;
	for (int row = 0; row < 4; row++)
		for (int colm = 0; colm < 4; colm++) {
			const float* ar = a.cell + 4 * row;
			r.cell[4 * row + colm] = (ar[0] * b.cell[colm]) + (ar[1] * b.cell[colm + 4]) + (ar[2] * b.cell[colm + 8]) + (ar[3] * b.cell[colm + 12]);
		}
	return r;
}
float3 TransformPosition(const float3& a, const mat4& M)
{
	const float* c = M.cell;
	return float3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 1.0f,
	              c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 1.0f,
	              c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 1.0f);
}
float3 TransformVector(const float3& a, const mat4& M)
{
	const float* c = M.cell;
	return float3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 0.0f,
	              c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 0.0f,
	              c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 0.0f);
}
// Cofactor inverse.  Each adjugate entry is a sum of six signed triple products taken in the
// published (MESA gluInvertMatrix) term order; the table below lists, per entry, the six
// (sign, i, j, k) terms so the arithmetic is a*b*c accumulated left to right.
mat4 mat4::Inverted() const
{
	struct Term { int s, i, j, k; };
	static const Term T[16][6] = {
		{ { 1, 5, 10, 15 }, { -1, 5, 11, 14 }, { -1, 9, 6, 15 }, { 1, 9, 7, 14 }, { 1, 13, 6, 11 }, { -1, 13, 7, 10 } },
		{ { -1, 1, 10, 15 }, { 1, 1, 11, 14 }, { 1, 9, 2, 15 }, { -1, 9, 3, 14 }, { -1, 13, 2, 11 }, { 1, 13, 3, 10 } },
		{ { 1, 1, 6, 15 }, { -1, 1, 7, 14 }, { -1, 5, 2, 15 }, { 1, 5, 3, 14 }, { 1, 13, 2, 7 }, { -1, 13, 3, 6 } },
		{ { -1, 1, 6, 11 }, { 1, 1, 7, 10 }, { 1, 5, 2, 11 }, { -1, 5, 3, 10 }, { -1, 9, 2, 7 }, { 1, 9, 3, 6 } },
		{ { -1, 4, 10, 15 }, { 1, 4, 11, 14 }, { 1, 8, 6, 15 }, { -1, 8, 7, 14 }, { -1, 12, 6, 11 }, { 1, 12, 7, 10 } },
		{ { 1, 0, 10, 15 }, { -1, 0, 11, 14 }, { -1, 8, 2, 15 }, { 1, 8, 3, 14 }, { 1, 12, 2, 11 }, { -1, 12, 3, 10 } },
		{ { -1, 0, 6, 15 }, { 1, 0, 7, 14 }, { 1, 4, 2, 15 }, { -1, 4, 3, 14 }, { -1, 12, 2, 7 }, { 1, 12, 3, 6 } },
		{ { 1, 0, 6, 11 }, { -1, 0, 7, 10 }, { -1, 4, 2, 11 }, { 1, 4, 3, 10 }, { 1, 8, 2, 7 }, { -1, 8, 3, 6 } },
		{ { 1, 4, 9, 15 }, { -1, 4, 11, 13 }, { -1, 8, 5, 15 }, { 1, 8, 7, 13 }, { 1, 12, 5, 11 }, { -1, 12, 7, 9 } },
		{ { -1, 0, 9, 15 }, { 1, 0, 11, 13 }, { 1, 8, 1, 15 }, { -1, 8, 3, 13 }, { -1, 12, 1, 11 }, { 1, 12, 3, 9 } },
		{ { 1, 0, 5, 15 }, { -1, 0, 7, 13 }, { -1, 4, 1, 15 }, { 1, 4, 3, 13 }, { 1, 12, 1, 7 }, { -1, 12, 3, 5 } },
		{ { -1, 0, 5, 11 }, { 1, 0, 7, 9 }, { 1, 4, 1, 11 }, { -1, 4, 3, 9 }, { -1, 8, 1, 7 }, { 1, 8, 3, 5 } },
		{ { -1, 4, 9, 14 }, { 1, 4, 10, 13 }, { 1, 8, 5, 14 }, { -1, 8, 6, 13 }, { -1, 12, 5, 10 }, { 1, 12, 6, 9 } },
		{ { 1, 0, 9, 14 }, { -1, 0, 10, 13 }, { -1, 8, 1, 14 }, { 1, 8, 2, 13 }, { 1, 12, 1, 10 }, { -1, 12, 2, 9 } },
		{ { -1, 0, 5, 14 }, { 1, 0, 6, 13 }, { 1, 4, 1, 14 }, { -1, 4, 2, 13 }, { -1, 12, 1, 6 }, { 1, 12, 2, 5 } },
		{ { 1, 0, 5, 10 }, { -1, 0, 6, 9 }, { -1, 4, 1, 10 }, { 1, 4, 2, 9 }, { 1, 8, 1, 6 }, { -1, 8, 2, 5 } },
	};
	float inv[16];
	for (int e = 0; e < 16; e++) {
		// first term carries its sign on the first factor (-c[i]*c[j]*c[k]), later terms are added or subtracted
		const Term& t0 = T[e][0];
		float acc = (t0.s > 0 ? cell[t0.i] : -cell[t0.i]) * cell[t0.j] * cell[t0.k];
		for (int q = 1; q < 6; q++) {
			const Term& t = T[e][q];
			float p = cell[t.i] * cell[t.j] * cell[t.k];
			acc = t.s > 0 ? acc + p : acc - p;
		}
		inv[e] = acc;
	}
	mat4 r;
	const float det = cell[0] * inv[0] + cell[1] * inv[4] + cell[2] * inv[8] + cell[3] * inv[12];
	if (det != 0) {
		const float invdet = 1.0f / det;
		for (int i = 0; i < 16; i++) r.cell[i] = inv[i] * invdet;
	}
	return r;
}

// ---- bvh -----------------------------------------------------------------------------------
bvh::bvh(Scene* s) : scene(s) {}
bvh::bvh(Mesh* m) : mesh(m) {}
bvh::~bvh() { delete[] primitiveIdx; delete[] bvhNode; }

// Flat view of the primitives a bvh indexes: triangles first, then spheres, then planes
// (bvh.cpp:618-627).
struct bvh::Builder {
	bvh& B;
	std::vector<const Triangle*> tri;
	const Sphere* sph = nullptr;
	const Plane* pla = nullptr;
	uint nTri, nSph, nPla, nAll;

	explicit Builder(bvh& b) : B(b)
	{
		if (B.scene) {
			for (auto& m : B.scene->meshes) for (auto& t : m.tri) tri.push_back(&t);
			sph = B.scene->spheres.data(), pla = B.scene->planes.data();
			nSph = (uint)B.scene->spheres.size(), nPla = (uint)B.scene->planes.size();
		} else {
			for (auto& t : B.mesh->tri) tri.push_back(&t);
			nSph = nPla = 0;
		}
		nTri = (uint)tri.size();
		nAll = nTri + nSph + nPla;
	}
	bool isTri(uint p) const { return p < nTri; }
	bool isSph(uint p) const { return p >= nTri && p < nTri + nSph; }

	// split coordinate of a primitive on an axis (triangle centroid / sphere centre); planes have none
	bool key(uint p, int axis, float& out) const
	{
		if (isTri(p)) { out = tri[p]->centroid[axis]; return true; }
		if (isSph(p)) { out = sph[p - nTri].pos[axis]; return true; }
		return false;
	}

	void nodeBounds(uint nodeIdx) // bvh.cpp:67-114
	{
		BVHNode& node = B.bvhNode[nodeIdx];
		float3 lo(1e30f), hi(-1e30f);
		for (uint i = 0; i < node.primCount; i++) {
			uint p = B.primitiveIdx[node.leftFirst + i];
			if (isTri(p)) {
				const Triangle& t = *tri[p];
				lo = fminf(lo, t.v0), lo = fminf(lo, t.v1), lo = fminf(lo, t.v2);
				hi = fmaxf(hi, t.v0), hi = fmaxf(hi, t.v1), hi = fmaxf(hi, t.v2);
			} else if (isSph(p)) {
				const Sphere& s = sph[p - nTri];
				lo = fminf(lo, s.pos - float3(s.r));
				hi = fmaxf(hi, s.pos + float3(s.r));
			} else {
				// planes: a slab through coordinate 0 for +X / +Y / +Z normals, otherwise everything
				// (and the loop stops there) -- bvh.cpp:90-111
				const float3 n = normalize(pla[p - (nTri + nSph)].N);
				int ax = -1;
				if (n.x + n.y + n.z == 1 && (n.x == 1 || n.y == 1 || n.z == 1)) ax = n.x == 1 ? 0 : (n.y == 1 ? 1 : 2);
				if (ax < 0) { lo = float3(-1e30f), hi = float3(1e30f); break; }
				float3 slabLo(-1e30f), slabHi(1e30f);
				slabLo[ax] = 0, slabHi[ax] = 0;
				lo = fminf(lo, slabLo), hi = fmaxf(hi, slabHi);
			}
		}
		node.aabbMin = lo, node.aabbMax = hi;
	}

	// 8-bin SAH over the three axes (bvh.cpp:116-193); returns the best cost, 1e30f if none
	float binnedSplit(const BVHNode& node, int& axis, float& splitPos) const
	{
		enum { BINS = 8 };
		float best = 1e30f;
		for (int a = 0; a < 3; a++) {
			float kmin = 1e30f, kmax = -1e30f, k;
			for (uint i = 0; i < node.primCount; i++)
				if (key(B.primitiveIdx[node.leftFirst + i], a, k)) { kmin = (k < kmin) ? k : kmin; kmax = (kmax < k) ? k : kmax; }
			if (kmin == kmax) continue;
			aabb box[BINS];
			int cnt[BINS] = { 0 };
			float scale = BINS / (kmax - kmin);
			for (uint i = 0; i < node.primCount; i++) {
				uint p = B.primitiveIdx[node.leftFirst + i];
				if (!key(p, a, k)) continue;
				int bi = std::min(BINS - 1, (int)((k - kmin) * scale));
				cnt[bi]++;
				if (isTri(p)) { box[bi].grow(tri[p]->v0); box[bi].grow(tri[p]->v1); box[bi].grow(tri[p]->v2); }
				else { const Sphere& s = sph[p - nTri]; box[bi].grow(s.pos - float3(2 * s.r)); box[bi].grow(s.pos + float3(2 * s.r)); }
			}
			float areaL[BINS - 1], areaR[BINS - 1];
			int cntL[BINS - 1], cntR[BINS - 1];
			aabb accL, accR;
			int sumL = 0, sumR = 0;
			for (int i = 0; i < BINS - 1; i++) {
				sumL += cnt[i], cntL[i] = sumL, accL.grow(box[i]), areaL[i] = accL.area();
				sumR += cnt[BINS - 1 - i], cntR[BINS - 2 - i] = sumR, accR.grow(box[BINS - 1 - i]), areaR[BINS - 2 - i] = accR.area();
			}
			scale = (kmax - kmin) / BINS;
			for (int i = 0; i < BINS - 1; i++) {
				float c = cntL[i] * areaL[i] + cntR[i] * areaR[i];
				if (c < best) axis = a, splitPos = kmin + scale * (i + 1), best = c;
			}
		}
		return best;
	}

	float sweepCost(const BVHNode& node, int axis, float pos) const // bvh.cpp:514-554
	{
		aabb L, R;
		int nL = 0, nR = 0;
		for (uint i = 0; i < node.primCount; i++) {
			uint p = B.primitiveIdx[node.leftFirst + i];
			if (isTri(p)) {
				const Triangle& t = *tri[p];
				aabb& dst = (t.centroid[axis] < pos) ? (nL++, L) : (nR++, R);
				dst.grow(t.v0), dst.grow(t.v1), dst.grow(t.v2);
			} else if (isSph(p)) {
				const Sphere& s = sph[p - nTri];
				aabb& dst = (s.pos[axis] < pos) ? (nL++, L) : (nR++, R);
				dst.grow(s.pos[axis] - float3(s.r)); // sic: scalar coordinate broadcast, as in the reference
				dst.grow(s.pos[axis] + float3(s.r));
			}
		}
		float cost = nL * L.area() + nR * R.area();
		return cost > 0 ? cost : 1e30f;
	}

	static int longestAxis(const float3& e)
	{
		int axis = 0;
		if (e.y > e.x) axis = 1;
		if (e.z > e[axis]) axis = 2;
		return axis;
	}

	// decide the split plane of one node; false = keep as leaf (bvh.cpp:226-293)
	bool choose(const BVHNode& node, int& axis, float& splitPos) const
	{
		switch (B.splitMethod) {
		case BINNEDSAH: {
			float c = binnedSplit(node, axis, splitPos);
			float3 e = node.aabbMax - node.aabbMin;
			float leafCost = node.primCount * (e.x * e.y + e.y * e.z + e.z * e.x);
			return !(c >= leafCost);
		}
		case LONGESTAXIS: {
			float3 e = node.aabbMax - node.aabbMin;
			axis = longestAxis(e);
			splitPos = node.aabbMin[axis] + e[axis] * 0.5f;
			return true;
		}
		case SAMESIZE: {
			float3 e = node.aabbMax - node.aabbMin;
			axis = longestAxis(e);
			std::vector<std::tuple<float, int>> keys;
			float k;
			for (uint i = 0; i < node.primCount; i++) {
				uint p = B.primitiveIdx[node.leftFirst + i];
				if (key(p, axis, k)) keys.push_back(std::make_tuple(k, (int)(isTri(p) ? p : p - nTri)));
			}
			std::sort(keys.begin(), keys.end());
			splitPos = std::get<0>(keys[node.primCount / 2]);
			return true;
		}
		case SAH: {
			int bestAxis = -1;
			float bestPos = 0, bestCost = 1e30f, cand = 0, k;
			for (int a = 0; a < 3; a++) for (uint i = 0; i < node.primCount; i++) {
				if (key(B.primitiveIdx[node.leftFirst + i], a, k)) cand = k;
				float c = sweepCost(node, a, cand);
				if (c < bestCost) bestPos = cand, bestAxis = a, bestCost = c;
			}
			if (bestAxis < 0) return false;
			axis = bestAxis, splitPos = bestPos;
			return true;
		}
		}
		return false;
	}

	// in-place partition (bvh.cpp:296-313); returns the first index of the right part.  A primitive
	// without a split coordinate (a plane; the reference indexes past its sphere array there) goes right.
	int partition(uint first, uint count, int axis, float splitPos)
	{
		int i = (int)first, j = i + (int)count - 1;
		float k;
		while (i <= j) {
			if (key(B.primitiveIdx[i], axis, k) && k < splitPos) i++;
			else std::swap(B.primitiveIdx[i], B.primitiveIdx[j--]);
		}
		return i;
	}

	void subdivideFrom(uint start, int startDepth) // bvh.cpp:223-333, depth-first, left before right
	{
		struct Item { uint node; int depth; };
		std::vector<Item> todo;
		todo.push_back({ start, startDepth });
		while (!todo.empty()) {
			Item it = todo.back();
			todo.pop_back();
			if (it.depth > B.treeDepth) B.treeDepth = it.depth;
			BVHNode& node = B.bvhNode[it.node];
			int axis = 0;
			float splitPos = 0;
			if (!choose(node, axis, splitPos)) continue;
			const uint first = node.leftFirst, count = node.primCount;
			int mid = partition(first, count, axis, splitPos);
			int leftCount = mid - (int)first;
			if (leftCount == 0 || leftCount == (int)count) continue;
			uint l = B.nodesUsed++, r = B.nodesUsed++;
			B.bvhNode[l].leftFirst = first, B.bvhNode[l].primCount = leftCount;
			B.bvhNode[r].leftFirst = mid, B.bvhNode[r].primCount = count - leftCount;
			node.leftFirst = l, node.primCount = 0;
			nodeBounds(l), nodeBounds(r);
			todo.push_back({ r, it.depth + 1 });
			todo.push_back({ l, it.depth + 1 });
		}
	}
};

void bvh::Build(bool isQ)
{
	if (isQ) throw std::runtime_error("QBVH is not part of this path (disabled in the reference)");
	Builder bl(*this);
	NTri = bl.nTri, NSph = bl.nSph, NPla = bl.nPla, N = bl.nAll;
	delete[] primitiveIdx;
	delete[] bvhNode;
	primitiveIdx = new uint[N ? N : 1];
	bvhNode = new BVHNode[2 * (N + 1)]();
	for (uint i = 0; i < N; ++i) primitiveIdx[i] = i;
	nodesUsed = 2, treeDepth = 0;
	bvhNode[rootNodeIdx].primCount = N;
	bvhNode[rootNodeIdx].leftFirst = 0;
	bl.nodeBounds(rootNodeIdx);
	// planes are split off first: root -> {triangles + spheres, planes} (bvh.cpp:202-221)
	if (NPla > 0 && (NSph + NTri > 0)) {
		uint l = nodesUsed++, r = nodesUsed++;
		bvhNode[l].leftFirst = 0, bvhNode[l].primCount = NTri + NSph;
		bvhNode[r].leftFirst = NTri + NSph, bvhNode[r].primCount = NPla;
		bvhNode[rootNodeIdx].leftFirst = l, bvhNode[rootNodeIdx].primCount = 0;
		bl.nodeBounds(l), bl.nodeBounds(r);
		bl.subdivideFrom(l, 2);
	} else {
		bl.subdivideFrom(rootNodeIdx, 1);
	}
	bounds.grow(bvhNode[rootNodeIdx].aabbMin);
	bounds.grow(bvhNode[rootNodeIdx].aabbMax);
	Refit();
}

void bvh::BuildOnDevice(rt_ctx* ctx)
{
	Builder bl(*this);
	NTri = bl.nTri, NSph = bl.nSph, NPla = bl.nPla, N = bl.nAll;
	std::vector<rt_triangle> T(NTri);
	std::vector<rt_sphere> S(NSph);
	std::vector<rt_plane> P(NPla);
	for (uint i = 0; i < NTri; i++) {
		const Triangle& t = *bl.tri[i];
		memset(&T[i], 0, sizeof(rt_triangle));
		memcpy(T[i].v0, &t.v0, 12), memcpy(T[i].v1, &t.v1, 12), memcpy(T[i].v2, &t.v2, 12);
	}
	for (uint i = 0; i < NSph; i++) { memset(&S[i], 0, sizeof(rt_sphere)); memcpy(S[i].pos, &bl.sph[i].pos, 12); S[i].r = bl.sph[i].r; }
	for (uint i = 0; i < NPla; i++) { memset(&P[i], 0, sizeof(rt_plane)); memcpy(P[i].N, &bl.pla[i].N, 12); P[i].d = bl.pla[i].d; }
	delete[] primitiveIdx;
	delete[] bvhNode;
	primitiveIdx = new uint[N ? N : 1];
	bvhNode = new BVHNode[2 * (N + 1)]();
	static_assert(sizeof(BVHNode) == sizeof(rt_bvh_node), "BVHNode must match rt_bvh_node");
	uint32_t used = 0;
	if (rt_build_bvh_split(ctx, splitMethod, T.data(), NTri, S.data(), NSph, P.data(), NPla, (rt_bvh_node*)bvhNode, primitiveIdx, &used) != RT_OK)
		throw std::runtime_error(std::string("bvh::BuildOnDevice: ") + rt_last_error(ctx));
	nodesUsed = used;
	// DataCollector::maxTreeDepth analogue: deepest leaf, counted in nodes from the root
	treeDepth = 0;
	std::vector<std::pair<uint, int>> walk(1, { rootNodeIdx, 1 });
	while (!walk.empty()) {
		const uint n = walk.back().first;
		const int d = walk.back().second;
		walk.pop_back();
		if (d > treeDepth) treeDepth = d;
		if (!bvhNode[n].isLeaf()) { walk.push_back({ bvhNode[n].leftFirst, d + 1 }); walk.push_back({ bvhNode[n].leftFirst + 1, d + 1 }); }
	}
	bounds.grow(bvhNode[rootNodeIdx].aabbMin);
	bounds.grow(bvhNode[rootNodeIdx].aabbMax);
}

void bvh::Refit() // bvh.cpp:556-594
{
	Builder bl(*this);
	for (int i = (int)nodesUsed - 1; i >= 0; i--) {
		if (i == 1) continue; // node 1 is never allocated (nodesUsed starts at 2)
		BVHNode& node = bvhNode[i];
		if (node.isLeaf()) { bl.nodeBounds(i); continue; }
		const BVHNode& l = bvhNode[node.leftFirst];
		const BVHNode& r = bvhNode[node.leftFirst + 1];
		node.aabbMin = fminf(l.aabbMin, r.aabbMin);
		node.aabbMax = fmaxf(l.aabbMax, r.aabbMax);
	}
}

// ---- bvhInstance ---------------------------------------------------------------------------
bvhInstance::bvhInstance(bvh* b) : blas(b)
{
	mat4 ident;
	SetTransform(ident); // grows the still-empty box with the identity-transformed corners ...
	bounds = blas->bounds; // ... and is then overwritten with the local box (bvhInstance.h:9)
}
void bvhInstance::SetTransform(mat4& transform)
{
	invTransform = transform.Inverted();
	matTransform = transform;
	// 'bounds' is grown, never reset: world box = union(previous, transformed corners)
	const float3 lo = blas->bounds.bmin, hi = blas->bounds.bmax;
	for (int i = 0; i < 8; i++)
		bounds.grow(TransformPosition(float3(i & 1 ? hi.x : lo.x, i & 2 ? hi.y : lo.y, i & 4 ? hi.z : lo.z), transform));
}

// ---- tlas ------------------------------------------------------------------------------------
tlas::tlas(bvhInstance* bvhList, int Ncount) : blas(bvhList), blasCount((uint)Ncount)
{
	tlasNode = new TLASNode[2 * Ncount + 1]();
	nodesUsed = 2;
}
tlas::~tlas() { delete[] tlasNode; }

int tlas::FindBestMatch(int* list, int Ncount, int A) // tlas.cpp:50-63
{
	float smallest = 1e30f;
	int bestB = -1;
	const TLASNode& a = tlasNode[list[A]];
	for (int B = 0; B < Ncount; B++) {
		if (B == A) continue;
		const TLASNode& b = tlasNode[list[B]];
		float3 e = fmaxf(a.aabbMax, b.aabbMax) - fminf(a.aabbMin, b.aabbMin);
		float surfaceArea = e.x * e.y + e.y * e.z + e.z * e.x;
		if (surfaceArea < smallest) smallest = surfaceArea, bestB = B;
	}
	return bestB;
}

void tlas::BuildOnDevice(rt_ctx* ctx) // the same nodes from rt_build_tlas
{
	if (blasCount == 0 || blasCount > 256) throw std::runtime_error("tlas::build supports 1..256 instances (nodeIdx[256] in the reference)");
	std::vector<float> b6((size_t)6 * blasCount);
	for (uint i = 0; i < blasCount; i++) {
		const aabb& bb = blas[i].bounds;
		b6[6 * i] = bb.bmin.x, b6[6 * i + 1] = bb.bmin.y, b6[6 * i + 2] = bb.bmin.z, b6[6 * i + 3] = bb.bmax.x, b6[6 * i + 4] = bb.bmax.y, b6[6 * i + 5] = bb.bmax.z;
	}
	static_assert(sizeof(TLASNode) == sizeof(rt_tlas_node), "TLASNode must match rt_tlas_node");
	uint32_t used = 0;
	if (rt_build_tlas(ctx, b6.data(), blasCount, (rt_tlas_node*)tlasNode, &used) != RT_OK)
		throw std::runtime_error(std::string("tlas::BuildOnDevice: ") + rt_last_error(ctx));
	nodesUsed = used;
}

void tlas::build() // tlas.cpp:13-48: agglomerative clustering by smallest union area
{
	if (blasCount == 0 || blasCount > 256) throw std::runtime_error("tlas::build supports 1..256 instances (nodeIdx[256] in the reference)");
	int nodeIdx[256], live = (int)blasCount;
	nodesUsed = 1;
	for (uint i = 0; i < blasCount; i++) {
		TLASNode& leaf = tlasNode[nodesUsed];
		leaf.aabbMin = blas[i].bounds.bmin, leaf.aabbMax = blas[i].bounds.bmax;
		leaf.BLAS = i, leaf.leftRight = 0;
		nodeIdx[i] = nodesUsed++;
	}
	int A = 0, B = FindBestMatch(nodeIdx, live, A);
	while (live > 1) {
		int C = FindBestMatch(nodeIdx, live, B);
		if (A != C) { A = B, B = C; continue; }
		const int ia = nodeIdx[A], ib = nodeIdx[B];
		TLASNode& parent = tlasNode[nodesUsed];
		parent.leftRight = ia + (ib << 16);
		parent.aabbMin = fminf(tlasNode[ia].aabbMin, tlasNode[ib].aabbMin);
		parent.aabbMax = fmaxf(tlasNode[ia].aabbMax, tlasNode[ib].aabbMax);
		nodeIdx[A] = nodesUsed++;
		nodeIdx[B] = nodeIdx[live - 1];
		B = FindBestMatch(nodeIdx, --live, A);
	}
	tlasNode[0] = tlasNode[nodeIdx[A]];
}

} // namespace rapt
