// ORACLE (test infrastructure, NOT product code) -- parity unpinned, see oracle/README.md.
//
// C entry points over the CPU restatement so that tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive it through ctypes.  Nothing in the product path links or loads this.
#include "orc_render.h"
#include <omp.h>

using namespace orc;

struct OrcScene {
	Scene sc;
	std::string err;
};
struct OrcRenderer {
	Renderer r;
};

static float3 f3(const float* p) { return float3(p[0], p[1], p[2]); }

extern "C" {

void* orc_scene_new() { return new OrcScene(); }
void orc_scene_free(void* h)
{
	OrcScene* s = (OrcScene*)h;
	if (!s) return;
	delete s->sc.b;
	delete s->sc.tl;
	for (auto* p : s->sc.instances) delete p;
	for (auto* p : s->sc.blasList) delete p;
	delete s;
}
const char* orc_last_error(void* h) { return ((OrcScene*)h)->err.c_str(); }

// diffuse(a, c, ks, kd, n, rt, e, s) -- template/scene.h:595-601
int orc_add_diffuse(void* h, const float* albedo, const float* col, float ks, float kd, int n, float emission, float shininess, int rt)
{
	Scene& sc = ((OrcScene*)h)->sc;
	Material m;
	m.type = DIFFUSE, m.albedo = f3(albedo), m.col = f3(col), m.specu = ks, m.diffu = kd, m.N = n;
	m.emission = float3(emission), m.shinieness = shininess, m.raytracer = rt != 0;
	sc.materials.push_back(m);
	return (int)sc.materials.size() - 1;
}
// metal(f, c, rt) -- template/scene.h:629
int orc_add_metal(void* h, float fuzzy, const float* col, int rt)
{
	Scene& sc = ((OrcScene*)h)->sc;
	Material m;
	m.type = METAL, m.col = f3(col), m.fuzzy = fuzzy < 1 ? fuzzy : 1, m.raytracer = rt != 0;
	sc.materials.push_back(m);
	return (int)sc.materials.size() - 1;
}
// glass(refIndex, c, a, r, n, rt) -- template/scene.h:643-646
int orc_add_glass(void* h, float ir, const float* col, const float* absorption, int rt)
{
	Scene& sc = ((OrcScene*)h)->sc;
	Material m;
	m.type = GLASS, m.col = f3(col), m.ir = ir, m.invIr = 1 / ir, m.absorption = f3(absorption), m.raytracer = rt != 0;
	sc.materials.push_back(m);
	return (int)sc.materials.size() - 1;
}
// AreaLight(idx, p, str, c, r, n, s, rt) -- template/scene.h:97-103
int orc_add_area_light(void* h, int idx, const float* pos, float strength, const float* col, float radius, const float* normal)
{
	Scene& sc = ((OrcScene*)h)->sc;
	Light l;
	l.kind = 0, l.objIdx = idx, l.pos = f3(pos), l.strength = strength, l.col = f3(col), l.normal = f3(normal);
	l.radius = radius, l.radius2 = radius * radius, l.area = 2 * l.radius2 * PI, l.raytracer = sc.raytracer;
	sc.lights.push_back(l);
	return (int)sc.lights.size() - 1;
}
// DirectionalLight(idx, p, str, c, n, r, rt) -- template/scene.h:146-148
int orc_add_dir_light(void* h, int idx, const float* pos, float strength, const float* col, const float* normal, float r)
{
	Scene& sc = ((OrcScene*)h)->sc;
	Light l;
	l.kind = 1, l.objIdx = idx, l.pos = f3(pos), l.strength = strength, l.col = f3(col), l.normal = f3(normal);
	l.sinAngle = x_sinf(r * PI / 2), l.raytracer = sc.raytracer;
	sc.lights.push_back(l);
	return (int)sc.lights.size() - 1;
}
int orc_add_sphere(void* h, int idx, int mat, const float* pos, float r)
{
	Scene& sc = ((OrcScene*)h)->sc;
	sc.spheres.push_back(Sphere(idx, mat, f3(pos), r));
	return (int)sc.spheres.size() - 1;
}
int orc_add_plane(void* h, int idx, int mat, const float* N, float d)
{
	Scene& sc = ((OrcScene*)h)->sc;
	sc.planes.push_back(Plane(idx, mat, f3(N), d));
	return (int)sc.planes.size() - 1;
}
int orc_add_mesh_raw(void* h, int group, int mat, const float* v9, int n)
{
	Scene& sc = ((OrcScene*)h)->sc;
	sc.meshes.push_back(Mesh(group, mat, v9, n));
	return (int)sc.meshes.size() - 1;
}
int orc_add_mesh_obj(void* h, int group, const char* path, int mat, const float* pos, float scale)
{
	OrcScene* s = (OrcScene*)h;
	Mesh m;
	if (!Mesh::LoadObj(m, group, path, mat, f3(pos), scale)) { s->err = std::string("cannot load obj ") + path; return -1; }
	s->sc.meshes.push_back(m);
	return (int)s->sc.meshes.size() - 1;
}
int orc_add_mesh_tri(void* h, int group, const char* path, int mat)
{
	OrcScene* s = (OrcScene*)h;
	Mesh m;
	if (!Mesh::LoadTri(m, group, path, mat)) { s->err = std::string("cannot load tri ") + path; return -1; }
	s->sc.meshes.push_back(m);
	return (int)s->sc.meshes.size() - 1;
}
int orc_mesh_count(void* h, int mesh) { return (int)((OrcScene*)h)->sc.meshes[mesh].tri.size(); }
// per triangle: v0 v1 v2 N centroid (15 floats) and objIdx
void orc_mesh_get(void* h, int mesh, float* out15, int* outIdx)
{
	const Mesh& m = ((OrcScene*)h)->sc.meshes[mesh];
	for (size_t i = 0; i < m.tri.size(); i++) {
		const Triangle& t = m.tri[i];
		const float3* src[5] = { &t.v0, &t.v1, &t.v2, &t.N, &t.centroid };
		for (int k = 0; k < 5; k++) { out15[15 * i + 3 * k] = src[k]->x; out15[15 * i + 3 * k + 1] = src[k]->y; out15[15 * i + 3 * k + 2] = src[k]->z; }
		outIdx[i] = t.objIdx;
	}
}
int orc_set_sky(void* h, int w, int hgt, int n, const unsigned char* px)
{
	Scene& sc = ((OrcScene*)h)->sc;
	sc.skydomeX = w, sc.skydomeY = hgt, sc.skydomeN = n;
	sc.skydome.assign(px, px + (size_t)w * hgt * n);
	return 0;
}
// Scene::toogleRaytracer semantics: lights follow the scene flag, materials keep theirs
void orc_set_raytracer(void* h, int rt)
{
	Scene& sc = ((OrcScene*)h)->sc;
	if (sc.raytracer != (rt != 0)) sc.toogleRaytracer();
}

void orc_set_time(void* h, float t) { ((OrcScene*)h)->sc.SetTime(t); }

// non-TLAS scene: new bvh(this); Build(false) -- template/scene.h:700-702
int orc_build(void* h, int splitMethod)
{
	Scene& sc = ((OrcScene*)h)->sc;
	delete sc.b;
	sc.useTLAS = false;
	sc.b = new bvh(&sc);
	sc.b->splitMethod = splitMethod;
	sc.b->Build();
	return 0;
}
// TLAS scene in the style of TLASSceneTest2 (template/scene.h:941-972): one bvh per referenced
// mesh, one bvhInstance per (mesh, transform), then tlas::build
int orc_build_tlas(void* h, int splitMethod, int nInst, const int* meshIdx, const float* transforms)
{
	OrcScene* s = (OrcScene*)h;
	Scene& sc = s->sc;
	sc.useTLAS = true;
	std::vector<int> blasOfMesh(sc.meshes.size(), -1);
	for (int i = 0; i < nInst; i++) {
		int mi = meshIdx[i];
		if (mi < 0 || mi >= (int)sc.meshes.size()) { s->err = "bad mesh index"; return -1; }
		if (blasOfMesh[mi] < 0) {
			bvh* b = new bvh(&sc.meshes[mi]);
			b->splitMethod = splitMethod;
			b->Build();
			blasOfMesh[mi] = (int)sc.blasList.size();
			sc.blasList.push_back(b);
		}
		bvhInstance* inst = new bvhInstance(sc.blasList[blasOfMesh[mi]]);
		inst->blasIdx = blasOfMesh[mi];
		mat4 T;
		memcpy(T.cell, transforms + 16 * i, 64);
		inst->SetTransform(T);
		sc.instances.push_back(inst);
	}
	sc.tl = new tlas(sc.instances);
	if (!sc.tl->build()) { s->err = "tlas build failed (1..256 instances)"; return -1; }
	return 0;
}

static const bvh* pick_bvh(const Scene& sc, int blas) { return blas < 0 ? sc.b : sc.blasList[blas]; }
int orc_blas_count(void* h) { return (int)((OrcScene*)h)->sc.blasList.size(); }
// info[0..6] = nodesUsed, N, NTri, NSph, NPla, maxDepth, allocated nodes
void orc_bvh_info(void* h, int blas, int* info)
{
	const bvh* b = pick_bvh(((OrcScene*)h)->sc, blas);
	info[0] = b->nodesUsed, info[1] = b->N, info[2] = b->NTri, info[3] = b->NSph, info[4] = b->NPla, info[5] = b->maxDepth, info[6] = (int)b->bvhNode.size();
}
// nodes: nodesUsed records of 32 bytes (reference BVHNode layout); primIdx: N uints
void orc_bvh_get(void* h, int blas, void* nodes, unsigned* primIdx)
{
	const bvh* b = pick_bvh(((OrcScene*)h)->sc, blas);
	memcpy(nodes, b->bvhNode.data(), (size_t)b->nodesUsed * sizeof(BVHNode));
	memcpy(primIdx, b->primitiveIdx.data(), (size_t)b->N * 4);
}
int orc_tlas_nodes_used(void* h) { return (int)((OrcScene*)h)->sc.tl->nodesUsed; }
void orc_tlas_get(void* h, void* nodes) { const tlas* t = ((OrcScene*)h)->sc.tl; memcpy(nodes, t->tlasNode.data(), (size_t)t->nodesUsed * sizeof(TLASNode)); }
// per instance: blas index, transform[16], invTransform[16], bounds min/max [6]
void orc_instance_get(void* h, int i, int* blas, float* T, float* invT, float* bounds)
{
	const bvhInstance* in = ((OrcScene*)h)->sc.instances[i];
	*blas = in->blasIdx;
	memcpy(T, in->matTransform.cell, 64);
	memcpy(invT, in->invTransform.cell, 64);
	bounds[0] = in->bounds.bmin.x, bounds[1] = in->bounds.bmin.y, bounds[2] = in->bounds.bmin.z;
	bounds[3] = in->bounds.bmax.x, bounds[4] = in->bounds.bmax.y, bounds[5] = in->bounds.bmax.z;
}

static void counters_out(const Counters& c, unsigned long long* out)
{
	if (!out) return;
	out[0] = c.inner_visits, out[1] = c.prim_tests, out[2] = c.tlas_inner, out[3] = c.instance_visits;
	out[4] = c.rays_nearest, out[5] = c.rays_occluded, out[6] = c.brute_tests, out[7] = c.light_tests;
	out[8] = c.tri_intersect_calls;
}

// Scene::FindNearest on n rays (O, D: n*3 floats; tmax: n floats or NULL for 1e34f).
// Outputs: t, objIdx, material index, hit normal.
int orc_find_nearest_batch(void* h, int n, const float* O, const float* D, const float* tmax, float t_min,
                           float* outT, int* outObj, int* outMat, float* outN, unsigned long long* counters)
{
	const Scene& sc = ((OrcScene*)h)->sc;
	Counters cnt;
	for (int i = 0; i < n; i++) {
		Ray r(f3(O + 3 * i), f3(D + 3 * i), tmax ? tmax[i] : 1e34f);
		sc.FindNearest(r, t_min, cnt);
		outT[i] = r.t, outObj[i] = r.objIdx;
		if (outMat) outMat[i] = r.objIdx == -1 ? -1 : r.mat;
		if (outN) { outN[3 * i] = r.hitNormal.x, outN[3 * i + 1] = r.hitNormal.y, outN[3 * i + 2] = r.hitNormal.z; }
	}
	counters_out(cnt, counters);
	return 0;
}
// The queries below Scene level: scope 1 = the scene's accelerator alone (bvh::Intersect of the scene bvh / tlas::Intersect:
// no lights, no brute-force primitives), 2 = bvh::Intersect of BLAS 'index' in its own object space, 3 =
// bvhInstance::BIntersect of instance 'index' (bvh.cpp:596-604, tlas.cpp:65-122, bvhInstance.cpp:3-35).
int orc_scope_nearest(void* h, int scope, int index, int n, const float* O, const float* D, const float* tmax,
                      float* outT, int* outObj, int* outMat, float* outN)
{
	const Scene& sc = ((OrcScene*)h)->sc;
	Counters cnt;
	for (int i = 0; i < n; i++) {
		Ray r(f3(O + 3 * i), f3(D + 3 * i), tmax ? tmax[i] : 1e34f);
		r.objIdx = -1;
		if (scope == 1) { if (sc.useTLAS) sc.tl->Intersect(r, cnt); else sc.b->Intersect(r, cnt); }
		else if (scope == 2) (sc.useTLAS ? sc.blasList[index] : sc.b)->Intersect(r, cnt);
		else sc.instances[index]->BIntersect(r, cnt);
		outT[i] = r.t, outObj[i] = r.objIdx, outMat[i] = r.objIdx == -1 ? -1 : r.mat;
		outN[3 * i] = r.hitNormal.x, outN[3 * i + 1] = r.hitNormal.y, outN[3 * i + 2] = r.hitNormal.z;
	}
	return 0;
}
int orc_scope_occluded(void* h, int scope, int index, int n, const float* O, const float* D, const float* tmax, unsigned char* out)
{
	const Scene& sc = ((OrcScene*)h)->sc;
	Counters cnt;
	for (int i = 0; i < n; i++) {
		Ray r(f3(O + 3 * i), f3(D + 3 * i), tmax ? tmax[i] : 1e34f);
		bool o;
		if (scope == 1) o = sc.useTLAS ? sc.tl->IsOccluded(r, cnt) : sc.b->IsOccluded(r, cnt);
		else if (scope == 2) o = (sc.useTLAS ? sc.blasList[index] : sc.b)->IsOccluded(r, cnt);
		else o = sc.instances[index]->IsOccluded(r, cnt);
		out[i] = o ? 1 : 0;
	}
	return 0;
}
// Scene::GetSkyColor (template/scene.h:1312-1327) for n directions
void orc_sky_color(void* h, int n, const float* D, float* rgb)
{
	const Scene& sc = ((OrcScene*)h)->sc;
	for (int i = 0; i < n; i++) {
		Ray r(float3(0), f3(D + 3 * i));
		const float3 c = sc.GetSkyColor(r);
		rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
	}
}
int orc_is_occluded_batch(void* h, int n, const float* O, const float* D, const float* tmax, unsigned char* out, unsigned long long* counters)
{
	const Scene& sc = ((OrcScene*)h)->sc;
	Counters cnt;
	for (int i = 0; i < n; i++) {
		Ray r(f3(O + 3 * i), f3(D + 3 * i), tmax ? tmax[i] : 1e34f);
		out[i] = sc.IsOccluded(r, cnt) ? 1 : 0;
	}
	counters_out(cnt, counters);
	return 0;
}

// ---- renderer ---------------------------------------------------------------------------------
void* orc_renderer_new(void* scene, int w, int hgt)
{
	OrcRenderer* r = new OrcRenderer();
	r->r.scene = &((OrcScene*)scene)->sc;
	r->r.Init(w, hgt);
	return r;
}
void orc_renderer_free(void* h) { delete (OrcRenderer*)h; }
void orc_renderer_set_camera(void* h, const float* camPos, const float* TL, const float* TR, const float* BL, int fisheye, float viewAngle, float yAngle)
{
	Camera& c = ((OrcRenderer*)h)->r.camera;
	c.camPos = f3(camPos), c.topLeft = f3(TL), c.topRight = f3(TR), c.bottomLeft = f3(BL);
	c.fishEye = fisheye != 0, c.viewAngle = viewAngle, c.yAngle = yAngle;
}
void orc_renderer_get_camera(void* h, float* out12)
{
	const Camera& c = ((OrcRenderer*)h)->r.camera;
	const float3* v[4] = { &c.camPos, &c.topLeft, &c.topRight, &c.bottomLeft };
	for (int k = 0; k < 4; k++) out12[3 * k] = v[k]->x, out12[3 * k + 1] = v[k]->y, out12[3 * k + 2] = v[k]->z;
}
void orc_renderer_clear(void* h)
{
	Renderer& r = ((OrcRenderer*)h)->r;
	std::fill(r.accumulator.begin(), r.accumulator.end(), float4{ 0, 0, 0, 0 });
	r.iterationNumber = 1;
}
// The pixel loop of Renderer::Tick for frames [frame0, frame0+nframes), rows [y0, y1).
// Mode follows the scene's raytracer flag.  nthreads <= 0: OpenMP default.
int orc_render(void* h, unsigned frame0, int nframes, unsigned seedBase, int y0, int y1, int nthreads, int maxDepthTrace, unsigned long long* counters)
{
	Renderer& r = ((OrcRenderer*)h)->r;
	r.max_depth_trace = maxDepthTrace;
	const int W = r.camera.width;
	Counters total;
	if (nthreads > 0) omp_set_num_threads(nthreads);
	for (int f = 0; f < nframes; f++) {
#pragma omp parallel
		{
			Counters local;
#pragma omp for schedule(dynamic)
			for (int y = y0; y < y1; ++y)
				for (int x = 0; x < W; ++x) r.Pixel(x, y, frame0 + f, seedBase, local);
#pragma omp critical
			total.add(local);
		}
		if (!r.scene->raytracer) r.iterationNumber++; // renderer.cpp:293-294
	}
	counters_out(total, counters);
	return 0;
}
// Q-learning guided sampler (orc_qlearn.h): grid == 0 switches it off
void orc_qlearn_enable(void* h, int grid, const float* lo, const float* hi, float alpha, float eps, float qInit, unsigned learnMask)
{
	QLearn& q = ((OrcRenderer*)h)->r.ql;
	if (grid <= 0) { q = QLearn(); return; }
	q.enable(grid, lo, hi, alpha, eps, qInit, learnMask);
}
void orc_qlearn_apply(void* h) { ((OrcRenderer*)h)->r.ql.apply(); }
void orc_qlearn_get(void* h, long long* sums, unsigned* counts, float* table)
{
	const QLearn& q = ((OrcRenderer*)h)->r.ql;
	const size_t cells = (size_t)q.grid * q.grid * q.grid;
	if (sums) memcpy(sums, q.sum.data(), cells * 64 * 8);
	if (counts) memcpy(counts, q.cnt.data(), cells * 64 * 4);
	if (table) for (size_t c = 0; c < cells; c++) memcpy(table + c * 64, &q.q[c * 72 + 8], 64 * 4);
}
void orc_qlearn_set_table(void* h, const float* table) { ((OrcRenderer*)h)->r.ql.set_table(table); }
int orc_qlearn_patch_of(const float* n) { return QLearn::patch_of(float3(n[0], n[1], n[2])); }
void orc_qlearn_get_v(void* h, float* v, float* centres)
{
	const QLearn& q = ((OrcRenderer*)h)->r.ql;
	memcpy(v, q.v.data(), q.v.size() * 4);
	for (int p = 0; p < 64; p++) centres[3 * p] = q.centre[p].x, centres[3 * p + 1] = q.centre[p].y, centres[3 * p + 2] = q.centre[p].z;
}
void orc_qlearn_set(void* h, const long long* sums, const unsigned* counts)
{
	QLearn& q = ((OrcRenderer*)h)->r.ql;
	const size_t cells = (size_t)q.grid * q.grid * q.grid;
	memcpy(q.sum.data(), sums, cells * 64 * 8);
	memcpy(q.cnt.data(), counts, cells * 64 * 4);
}
// Renderer::Tick: one frame with the reference's iteration bookkeeping; *camChanged in/out
int orc_tick(void* h, int* camChanged, unsigned frame, unsigned seedBase, int nthreads, unsigned* pixels)
{
	Renderer& r = ((OrcRenderer*)h)->r;
	if (nthreads > 0) omp_set_num_threads(nthreads);
	bool ch = *camChanged != 0;
	r.max_depth_trace = 4;
	r.Tick(ch, frame, seedBase, pixels);
	*camChanged = ch ? 1 : 0;
	return r.iterationNumber;
}
// Renderer::Trace (mode 0) / Renderer::Sample (mode 1) on caller rays with a caller energy; ray i draws from the
// stream StreamSeed(seedBase + i), as rt_trace_batch does
void orc_trace_rays(void* h, int mode, int n, const float* O, const float* D, int depth, const float* energy, unsigned seedBase, float* rgb)
{
	Renderer& r = ((OrcRenderer*)h)->r;
	Counters cnt;
	for (int i = 0; i < n; i++) {
		uint seed = StreamSeed(seedBase + (uint)i);
		Ray ray(f3(O + 3 * i), f3(D + 3 * i));
		const float3 c = mode == 0 ? r.Trace(ray, depth, f3(energy), seed, cnt) : r.Sample(ray, depth, f3(energy), seed, cnt);
		rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
	}
}
void orc_get_accumulator(void* h, float* out) { const Renderer& r = ((OrcRenderer*)h)->r; memcpy(out, r.accumulator.data(), r.accumulator.size() * 16); }
// screen->pixels for iteration count 'it' (renderer.cpp:287-290)
void orc_resolve(void* h, int it, unsigned* out)
{
	const Renderer& r = ((OrcRenderer*)h)->r;
	for (size_t i = 0; i < r.accumulator.size(); i++) out[i] = r.ResolvePixel(i, it);
}
// primary rays only: GetPrimaryRay + FindNearest(t_min) per pixel
void orc_primary_hits(void* h, float t_min, int* outObj, float* outT, unsigned long long* counters)
{
	const Renderer& r = ((OrcRenderer*)h)->r;
	Counters cnt;
	const int W = r.camera.width, H = r.camera.height;
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
		Ray pr = r.camera.GetPrimaryRay(x, y);
		r.scene->FindNearest(pr, t_min, cnt);
		outObj[x + y * W] = pr.objIdx, outT[x + y * W] = pr.t;
	}
	counters_out(cnt, counters);
}
void orc_primary_rays(void* h, float* O, float* D)
{
	const Renderer& r = ((OrcRenderer*)h)->r;
	const int W = r.camera.width, H = r.camera.height;
	for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
		Ray pr = r.camera.GetPrimaryRay(x, y);
		size_t i = (size_t)x + (size_t)y * W;
		O[3 * i] = pr.O.x, O[3 * i + 1] = pr.O.y, O[3 * i + 2] = pr.O.z;
		D[3 * i] = pr.D.x, D[3 * i + 1] = pr.D.y, D[3 * i + 2] = pr.D.z;
	}
}

// ---- unit-level entry points (per-function parity vectors) -----------------------------------
void orc_rng_stream(unsigned seedBase, int n, unsigned* outU, float* outF)
{
	uint s = StreamSeed(seedBase);
	for (int i = 0; i < n; i++) { uint before = s; (void)before; float f = RandomFloat(s); outU[i] = s; outF[i] = f; }
}
void orc_hemisphere(unsigned seedBase, int n, const float* normals, float* out)
{
	uint s = StreamSeed(seedBase);
	for (int i = 0; i < n; i++) { float3 v = RandomInHemisphere(s, f3(normals + 3 * i)); out[3 * i] = v.x, out[3 * i + 1] = v.y, out[3 * i + 2] = v.z; }
}
float orc_intersect_aabb(const float* O, const float* D, float t, const float* bmin, const float* bmax)
{
	Ray r(f3(O), f3(D), t);
	return IntersectAABB(r, f3(bmin), f3(bmax));
}
void orc_fresnel(int n, const float* I, const float* N, float ior, float* kr)
{
	for (int i = 0; i < n; i++) glass_fresnel(f3(I + 3 * i), f3(N + 3 * i), ior, kr[i]);
}
void orc_refract(int n, const float* I, const float* N, float ratio, float* out)
{
	for (int i = 0; i < n; i++) { float3 v = glass_refract(f3(I + 3 * i), f3(N + 3 * i), ratio); out[3 * i] = v.x, out[3 * i + 1] = v.y, out[3 * i + 2] = v.z; }
}
void orc_mat4_inverse(const float* m, float* out) { mat4 a; memcpy(a.cell, m, 64); mat4 r = a.Inverted(); memcpy(out, r.cell, 64); }
// Translate * Scale * RotateX * RotateY * RotateZ, the product every TLAS scene uses (template/scene.h:927)
void orc_mat4_trs(const float* t, float s, float rx, float ry, float rz, float* out)
{
	mat4 M = mat4::Translate(f3(t)) * mat4::Scale(s) * mat4::RotateX(rx) * mat4::RotateY(ry) * mat4::RotateZ(rz);
	memcpy(out, M.cell, 64);
}

} // extern "C"
