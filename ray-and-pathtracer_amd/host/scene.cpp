// Host-side scene container: Mesh loaders (template/scene.h:258-313), Scene bookkeeping
// (template/scene.h:685-1397) and the flattening of host objects into the rt_scene_desc that
// rt_upload_scene() consumes.  The three queries forward to the device library.
#include "rapt.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <fstream>
#include <map>
#include <stdexcept>

namespace rapt {

// ---- Mesh ------------------------------------------------------------------------------------
static bool read_file(const char* path, std::string& out)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	out.resize(n > 0 ? (size_t)n : 0);
	size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
	fclose(f);
	out.resize(got);
	return true;
}

// .tri: nine floats per record.  The reference's loop (template/scene.h:266-282) pushes a triangle
// for every fscanf call, including the final call that only reports end-of-file, so the last
// record appears twice; a short final record overwrites a prefix of the previous values.
Mesh::Mesh(int idGroup, const char* path, material* m) : mat(m), groupIdx(idGroup)
{
	std::string text;
	if (!read_file(path, text)) throw std::runtime_error(std::string("cannot open ") + path);
	float rec[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	const char* p = text.c_str();
	int count = 0;
	bool more = true;
	while (more) {
		int got = 0;
		for (; got < 9; got++) {
			char* end = nullptr;
			float v = strtof(p, &end);
			if (end == p) break; // end of input or a token that is not a number
			rec[got] = v, p = end;
		}
		more = got > 0;
		float3 a(rec[0], rec[1], rec[2]), b(rec[3], rec[4], rec[5]), c(rec[6], rec[7], rec[8]);
		originalVerts.push_back(a), originalVerts.push_back(b), originalVerts.push_back(c);
		vertices.push_back(a), vertices.push_back(b), vertices.push_back(c);
		int n = (int)vertices.size();
		int3 f; f.x = n - 2, f.y = n - 1, f.z = n;
		faces.push_back(f);
		tri.push_back(Triangle(1000 * idGroup + count, m, a, b, c));
		count++;
	}
}

// .obj: only 'v x y z' and 'f a//n b//n c//n' records are read (template/scene.h:294-308);
// vertices are scaled, then offset; triangle ids are 1000*group + face number.
Mesh::Mesh(int idGroup, std::string path, material* m, float3 pos, float scale) : mat(m), groupIdx(idGroup)
{
	std::ifstream file(path, std::ios::in);
	if (!file) throw std::runtime_error("Cannot open " + path);
	std::string line;
	while (std::getline(file, line)) {
		if (line.compare(0, 2, "v ") == 0) {
			const char* p = line.c_str() + 2;
			char* end = nullptr;
			float xyz[3] = { 0, 0, 0 };
			for (int k = 0; k < 3; k++) { xyz[k] = strtof(p, &end); p = end; }
			float3 v(xyz[0] * scale + pos.x, xyz[1] * scale + pos.y, xyz[2] * scale + pos.z);
			vertices.push_back(v), originalVerts.push_back(v);
		} else if (line.compare(0, 2, "f ") == 0) {
			int3 f;
			int skip;
			sscanf(line.c_str(), "f %i//%i %i//%i %i//%i", &f.x, &skip, &f.y, &skip, &f.z, &skip);
			faces.push_back(f);
		}
	}
	for (size_t i = 0; i < faces.size(); i++) {
		int3 f = faces[i];
		f.x--, f.y--, f.z--;
		const int nv = (int)vertices.size();
		if (f.x < 0 || f.y < 0 || f.z < 0 || f.x >= nv || f.y >= nv || f.z >= nv) throw std::runtime_error("face index out of range in " + path);
		tri.push_back(Triangle(1000 * idGroup + (int)i, m, f, vertices));
	}
}

Mesh::Mesh(int idGroup, material* m, const float* v9, int n) : mat(m), groupIdx(idGroup)
{
	for (int i = 0; i < n; i++) {
		const float* p = v9 + 9 * i;
		float3 a(p[0], p[1], p[2]), b(p[3], p[4], p[5]), c(p[6], p[7], p[8]);
		vertices.push_back(a), vertices.push_back(b), vertices.push_back(c);
		originalVerts.push_back(a), originalVerts.push_back(b), originalVerts.push_back(c);
		int3 f; f.x = 3 * i + 1, f.y = 3 * i + 2, f.z = 3 * i + 3;
		faces.push_back(f);
		tri.push_back(Triangle(1000 * idGroup + i, m, a, b, c));
	}
}

// ---- sky texture -------------------------------------------------------------------------------
// The reference loads its skydome with stbi_load("...hdr", &x, &y, &n, 3) (template/scene.h:792 ...):
// a Radiance RGBE picture decoded to float and squeezed to 8 bits per channel.  Own reader for the
// published format (header lines, "-Y h +X w", flat or new-style run-length scanlines) followed by the
// same float pipeline: c * 2^(e-136), then (float)pow(v, 1/2.2f) * 255 + 0.5f, clamped, truncated.
// Pinned against the reference's vendored stb_image by tests/test_sky_hdr.py.
static bool hdr_token(const std::string& d, size_t& pos, std::string& out)
{
	out.clear();
	while (pos < d.size()) {
		char c = d[pos++];
		if (c == '\n') return true;
		out.push_back(c);
	}
	return !out.empty();
}
bool Scene::LoadSkyHDR(const char* path, std::string* why)
{
	auto fail = [&](const char* msg) { if (why) *why = msg; return false; };
	std::string d;
	if (!read_file(path, d)) return fail("cannot open file");
	size_t pos = 0;
	std::string tok;
	hdr_token(d, pos, tok);
	if (tok != "#?RADIANCE" && tok != "#?RGBE") return fail("not a Radiance HDR file");
	bool valid = false;
	for (;;) {
		if (!hdr_token(d, pos, tok)) return fail("truncated header");
		if (tok.empty()) break;
		if (tok == "FORMAT=32-bit_rle_rgbe") valid = true;
	}
	if (!valid) return fail("unsupported HDR format");
	hdr_token(d, pos, tok);
	int hgt = 0, wid = 0;
	if (sscanf(tok.c_str(), "-Y %d +X %d", &hgt, &wid) != 2 || hgt <= 0 || wid <= 0) return fail("unsupported data layout");
	std::vector<unsigned char> rgbe((size_t)wid * hgt * 4);
	auto need = [&](size_t n) { return pos + n <= d.size(); };
	const unsigned char* u = reinterpret_cast<const unsigned char*>(d.data());
	bool flat = wid < 8 || wid >= 32768;
	if (!flat && need(4) && !(u[pos] == 2 && u[pos + 1] == 2 && !(u[pos + 2] & 0x80))) flat = true; // old-style file: plain pixels
	if (flat) {
		if (!need(rgbe.size())) return fail("truncated pixel data");
		memcpy(rgbe.data(), u + pos, rgbe.size());
	} else {
		for (int j = 0; j < hgt; j++) {
			if (!need(4)) return fail("truncated scanline");
			if (u[pos] != 2 || u[pos + 1] != 2 || (u[pos + 2] & 0x80)) return fail("mixed scanline encodings");
			if (((u[pos + 2] << 8) | u[pos + 3]) != wid) return fail("invalid decoded scanline length");
			pos += 4;
			for (int k = 0; k < 4; k++) {
				int i = 0;
				while (i < wid) {
					if (!need(1)) return fail("truncated run");
					int count = u[pos++];
					if (count > 128) {
						count -= 128;
						if (count > wid - i || !need(1)) return fail("bad RLE data in HDR");
						unsigned char v = u[pos++];
						for (int z = 0; z < count; z++) rgbe[((size_t)j * wid + i++) * 4 + k] = v;
					} else {
						if (count > wid - i || !need((size_t)count)) return fail("bad RLE data in HDR");
						for (int z = 0; z < count; z++) rgbe[((size_t)j * wid + i++) * 4 + k] = u[pos++];
					}
				}
			}
		}
	}
	skydome.resize((size_t)wid * hgt * 3);
	const float gamma_i = 1.0f / 2.2f, scale_i = 1.0f;
	for (size_t px = 0; px < (size_t)wid * hgt; px++) {
		const unsigned char* q = &rgbe[px * 4];
		float f1 = q[3] ? (float)ldexp(1.0f, (int)q[3] - (128 + 8)) : 0.0f;
		for (int k = 0; k < 3; k++) {
			float lin = q[3] ? q[k] * f1 : 0.0f;
			float z = (float)pow(lin * scale_i, gamma_i) * 255 + 0.5f;
			if (z < 0) z = 0;
			if (z > 255) z = 255;
			skydome[px * 3 + k] = (unsigned char)(int)z;
		}
	}
	skydomeX = wid, skydomeY = hgt, skydomeN = 3;
	return true;
}

// ---- Scene -------------------------------------------------------------------------------------
struct Scene::Flat {
	rt_scene_desc desc;
	std::vector<rt_blas> blas;
	std::vector<std::vector<rt_triangle>> tris; // per blas
	std::vector<rt_sphere> spheres;
	std::vector<rt_plane> planes;
	std::vector<rt_instance> instances;
	std::vector<rt_light> lights;
	std::vector<rt_material> materials;
	std::map<const material*, int> matIndex;
};

Scene::Scene() : flat(new Flat()) {}
Scene::~Scene()
{
	delete b;
	delete tl;
	delete[] bvhList;
	delete[] Transforms;
	for (auto* p : blasOwned) delete p;
	for (auto* l : lights) delete l;
	for (auto* m : materials) delete m;
	delete flat;
}

uint Scene::getTriangleNb() const { uint acc = 0; for (auto& m : meshes) acc += m.getSize(); return acc; }
const Triangle& Scene::getTriangle(uint idx) const
{
	size_t i = 0;
	while (idx >= meshes[i].getSize()) { idx -= meshes[i].getSize(); i++; }
	return meshes[i].tri[idx];
}
void Scene::toogleRaytracer()
{
	raytracer = !raytracer;
	SetIterationNumber(1);
	for (auto* l : lights) l->updateTracing(raytracer);
}

void Scene::BuildBVH(int splitMethod)
{
	delete b;
	useTLAS = false;
	b = new bvh(this);
	b->splitMethod = splitMethod;
	if (deviceBuild) b->BuildOnDevice(deviceBuild);
	else b->Build(false);
}

void Scene::BuildTLAS(const std::vector<int>& meshOfInstance, const std::vector<mat4>& transforms, int splitMethod)
{
	if (meshOfInstance.size() != transforms.size() || meshOfInstance.empty()) throw std::runtime_error("BuildTLAS: bad instance list");
	useTLAS = true;
	bvhCount = (uint)meshOfInstance.size();
	delete[] bvhList;
	delete[] Transforms;
	bvhList = new bvhInstance[bvhCount];
	Transforms = new mat4[bvhCount];
	std::map<int, bvh*> blasOfMesh;
	for (uint i = 0; i < bvhCount; i++) {
		int mi = meshOfInstance[i];
		if (mi < 0 || mi >= (int)meshes.size()) throw std::runtime_error("BuildTLAS: bad mesh index");
		bvh*& bl = blasOfMesh[mi];
		if (!bl) {
			bl = new bvh(&meshes[mi]);
			bl->splitMethod = splitMethod;
			if (deviceBuild) bl->BuildOnDevice(deviceBuild);
			else bl->Build();
			blasOwned.push_back(bl);
		}
		Transforms[i] = transforms[i];
		bvhList[i] = bvhInstance(bl);
		bvhList[i].SetTransform(Transforms[i]);
	}
	delete tl;
	tl = new tlas(bvhList, (int)bvhCount);
	if (deviceBuild) tl->BuildOnDevice(deviceBuild);
	else tl->build();
}

int Scene::materialIndex(const material* m) const
{
	auto it = flat->matIndex.find(m);
	if (it != flat->matIndex.end()) return it->second;
	int idx = (int)flat->materials.size();
	rt_material r;
	memset(&r, 0, sizeof(r));
	r.type = m->type, r.raytracer = m->raytracer ? 1 : 0;
	r.col[0] = m->col.x, r.col[1] = m->col.y, r.col[2] = m->col.z;
	r.albedo[0] = m->albedo.x, r.albedo[1] = m->albedo.y, r.albedo[2] = m->albedo.z;
	if (m->type == DIFFUSE) {
		const diffuse* d = static_cast<const diffuse*>(m);
		r.specu = d->specu, r.diffu = d->diffu, r.shinieness = d->shinieness, r.N = d->N;
	} else if (m->type == GLASS) {
		const glass* g = static_cast<const glass*>(m);
		r.ir = g->ir, r.absorption[0] = g->absorption.x, r.absorption[1] = g->absorption.y, r.absorption[2] = g->absorption.z;
	}
	flat->materials.push_back(r);
	flat->matIndex[m] = idx;
	return idx;
}

static void put3(float* d, const float3& v) { d[0] = v.x, d[1] = v.y, d[2] = v.z; }

const rt_scene_desc& Scene::Describe()
{
	Flat& F = *flat;
	F.blas.clear(), F.tris.clear(), F.spheres.clear(), F.planes.clear(), F.instances.clear(), F.lights.clear();
	F.materials.clear(), F.matIndex.clear();
	for (auto* m : materials) materialIndex(m); // registered order first, so indices follow creation order

	auto flattenTri = [&](const Triangle& t) {
		rt_triangle r;
		put3(r.v0, t.v0), put3(r.v1, t.v1), put3(r.v2, t.v2), put3(r.N, t.N);
		r.obj_idx = t.objIdx, r.material = materialIndex(t.mat);
		return r;
	};
	for (auto& s : spheres) {
		rt_sphere r;
		put3(r.pos, s.pos), r.r2 = s.r2, r.invr = s.invr, r.r = s.r, r.obj_idx = s.objIdx, r.material = materialIndex(s.mat);
		F.spheres.push_back(r);
	}
	for (auto& p : planes) {
		rt_plane r;
		put3(r.N, p.N), r.d = p.d, r.obj_idx = p.objIdx, r.material = materialIndex(p.mat);
		F.planes.push_back(r);
	}
	auto addBlas = [&](const bvh* bv, bool withAnalytic) {
		F.tris.emplace_back();
		std::vector<rt_triangle>& T = F.tris.back();
		if (bv->scene) { for (auto& m : bv->scene->meshes) for (auto& t : m.tri) T.push_back(flattenTri(t)); }
		else for (auto& t : bv->mesh->tri) T.push_back(flattenTri(t));
		rt_blas r;
		memset(&r, 0, sizeof(r));
		r.nodes = reinterpret_cast<const rt_bvh_node*>(bv->bvhNode), r.nodes_used = bv->nodesUsed;
		r.prim_idx = bv->primitiveIdx, r.n_prims = bv->N;
		r.n_tri = bv->NTri;
		if (withAnalytic) r.n_sph = bv->NSph, r.n_pla = bv->NPla;
		F.blas.push_back(r);
	};
	memset(&F.desc, 0, sizeof(F.desc));
	if (!useTLAS) {
		if (!b) throw std::runtime_error("Scene: BuildBVH() has not been called");
		addBlas(b, true);
	} else {
		if (!tl) throw std::runtime_error("Scene: BuildTLAS() has not been called");
		std::map<const bvh*, int> idx;
		for (uint i = 0; i < bvhCount; i++) {
			const bvh* bv = bvhList[i].blas;
			if (!idx.count(bv)) { idx[bv] = (int)F.blas.size(); addBlas(bv, false); }
			rt_instance in;
			in.blas = idx[bv];
			memcpy(in.transform, bvhList[i].matTransform.cell, 64);
			memcpy(in.inv_transform, bvhList[i].invTransform.cell, 64);
			F.instances.push_back(in);
		}
		F.desc.instances = F.instances.data(), F.desc.n_instances = (uint32_t)F.instances.size();
		F.desc.tlas_nodes = reinterpret_cast<const rt_tlas_node*>(tl->tlasNode), F.desc.tlas_nodes_used = tl->nodesUsed;
		F.desc.brute_spheres = F.spheres.data(), F.desc.n_brute_spheres = (uint32_t)F.spheres.size();
		F.desc.brute_planes = F.planes.data(), F.desc.n_brute_planes = (uint32_t)F.planes.size();
	}
	// vectors are complete now: take the element pointers
	for (size_t i = 0; i < F.blas.size(); i++) {
		F.blas[i].tris = F.tris[i].data();
		if (!useTLAS) F.blas[i].spheres = F.spheres.data(), F.blas[i].planes = F.planes.data();
	}
	for (auto* l : lights) {
		rt_light r;
		memset(&r, 0, sizeof(r));
		r.kind = l->kind(), r.obj_idx = l->objIdx, r.strength = l->strength;
		put3(r.pos, l->pos), put3(r.col, l->col), put3(r.normal, l->normal);
		if (r.kind == RT_LIGHT_AREA) r.radius = static_cast<AreaLight*>(l)->radius;
		if (r.kind == RT_LIGHT_DIRECTIONAL) r.sin_angle = static_cast<DirectionalLight*>(l)->sinAngle;
		F.lights.push_back(r);
	}
	F.desc.use_tlas = useTLAS ? 1 : 0;
	F.desc.blas = F.blas.data(), F.desc.n_blas = (uint32_t)F.blas.size();
	F.desc.lights = F.lights.data(), F.desc.n_lights = (uint32_t)F.lights.size();
	F.desc.materials = F.materials.data(), F.desc.n_materials = (uint32_t)F.materials.size();
	F.desc.sky_pixels = skydome.empty() ? nullptr : skydome.data();
	F.desc.sky_w = skydomeX, F.desc.sky_h = skydomeY, F.desc.sky_n = skydomeN;
	return F.desc;
}

static void check(rt_ctx* ctx, int rc)
{
	if (rc != RT_OK) throw std::runtime_error(std::string("rt_amd: ") + rt_last_error(ctx));
}

void Scene::Commit(rt_ctx* c)
{
	ctx = c;
	alsoCtx.clear(); // a re-commit starts over: Renderer::Commit names the other contexts again
	check(ctx, rt_upload_scene(ctx, &Describe()));
	// the accelerator objects learn where they live, for their own Intersect / IsOccluded members
	if (!useTLAS) { if (b) b->owner = this, b->blasIndex = 0; }
	else {
		std::map<const bvh*, int> idx;
		for (uint i = 0; i < bvhCount; i++) {
			bvh* bv = bvhList[i].blas;
			if (!idx.count(bv)) { const int k = (int)idx.size(); idx[bv] = k; }
			bv->owner = this, bv->blasIndex = idx[bv]; // the order Describe() flattened them in
			bvhList[i].owner = this, bvhList[i].index = (int)i;
		}
		if (tl) tl->owner = this;
	}
}

void Scene::CommitAlso(rt_ctx* other)
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	check(other, rt_upload_scene(other, &Describe()));
	if (other != ctx && std::find(alsoCtx.begin(), alsoCtx.end(), other) == alsoCtx.end()) alsoCtx.push_back(other);
}

// every context holds its own copy of the geometry and its own refitted tree: all of them move to time t
// (a multi-GPU frame interleaves rows rendered from each copy)
void Scene::SetTime(float t)
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	check(ctx, rt_set_time(ctx, t));
	for (rt_ctx* o : alsoCtx) check(o, rt_set_time(o, t));
}

void Scene::FindNearestBatch(int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out) const
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	check(ctx, rt_intersect_batch(ctx, n, O, D, tmax, t_min, out));
}
void Scene::IsOccludedBatch(int n, const float* O, const float* D, const float* tmax, uint8_t* out) const
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	check(ctx, rt_occluded_batch(ctx, n, O, D, tmax, out));
}

// Same contract as the reference: results are returned by mutating the Ray (t, objIdx, m,
// hitNormal); a miss leaves objIdx == -1.  One device round trip per call: use the batch forms
// on any path that matters.
void Scene::FindNearest(Ray& ray, float t_min) const
{
	rt_hit h;
	FindNearestBatch(1, &ray.O.x, &ray.D.x, &ray.t, t_min, &h);
	ray.objIdx = h.obj_idx;
	ray.t = h.t;
	if (h.obj_idx != -1) {
		ray.hitNormal = float3(h.normal[0], h.normal[1], h.normal[2]);
		ray.m = nullptr;
		for (auto& kv : flat->matIndex) if (kv.second == h.material) ray.m = const_cast<material*>(kv.first);
	}
}
bool Scene::IsOccluded(Ray& ray) const
{
	uint8_t o = 0;
	IsOccludedBatch(1, &ray.O.x, &ray.D.x, &ray.t, &o);
	return o != 0;
}
float3 Scene::GetSkyColor(Ray& ray) const
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	float rgb[3];
	check(ctx, rt_sky_color_batch(ctx, 1, &ray.D.x, rgb));
	return float3(rgb[0], rgb[1], rgb[2]);
}

// bvh::Intersect, tlas::Intersect, bvhInstance::BIntersect and their IsOccluded forms: the same contract as
// FindNearest (the Ray is mutated when something closer than ray.t is hit), one device round trip per call
void Scene::ScopeNearest(int scope, int index, Ray& ray) const
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	rt_hit h;
	check(ctx, rt_intersect_scope(ctx, scope, index, 1, &ray.O.x, &ray.D.x, &ray.t, 0.0f, &h));
	if (h.obj_idx == -1) return; // nothing closer than ray.t: the ray keeps what it had
	ray.objIdx = h.obj_idx, ray.t = h.t;
	ray.hitNormal = float3(h.normal[0], h.normal[1], h.normal[2]);
	ray.m = nullptr;
	for (auto& kv : flat->matIndex) if (kv.second == h.material) ray.m = const_cast<material*>(kv.first);
}
bool Scene::ScopeOccluded(int scope, int index, Ray& ray) const
{
	if (!ctx) throw std::runtime_error("Scene: Commit() has not been called");
	uint8_t o = 0;
	check(ctx, rt_occluded_scope(ctx, scope, index, 1, &ray.O.x, &ray.D.x, &ray.t, &o));
	return o != 0;
}
static Scene* committed(Scene* owner, const char* who)
{
	if (!owner) throw std::runtime_error(std::string(who) + ": the scene has not been committed (Scene::Commit)");
	return owner;
}
void bvh::Intersect(Ray& ray) { committed(owner, "bvh::Intersect")->ScopeNearest(RT_SCOPE_BLAS, blasIndex, ray); }
bool bvh::IsOccluded(Ray& ray) { return committed(owner, "bvh::IsOccluded")->ScopeOccluded(RT_SCOPE_BLAS, blasIndex, ray); }
void tlas::Intersect(Ray& ray) { committed(owner, "tlas::Intersect")->ScopeNearest(RT_SCOPE_ACCEL, 0, ray); }
bool tlas::IsOccluded(Ray& ray) { return committed(owner, "tlas::IsOccluded")->ScopeOccluded(RT_SCOPE_ACCEL, 0, ray); }
void bvhInstance::BIntersect(Ray& ray) { committed(owner, "bvhInstance::BIntersect")->ScopeNearest(RT_SCOPE_INSTANCE, index, ray); }
bool bvhInstance::IsOccluded(Ray& ray) { return committed(owner, "bvhInstance::IsOccluded")->ScopeOccluded(RT_SCOPE_INSTANCE, index, ray); }

} // namespace rapt
