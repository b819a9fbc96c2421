// C entry points over the C++ host mirror (rapt::Scene / rapt::Renderer) so that Python tests and
// bench.py can build scenes with the product's own loaders and builders.  A standalone scene
// (rth_scene_new) needs no GPU: it is what the CPU test tier uses to compare builder output with
// the oracle.  A renderer (rth_renderer_new) owns a device context.
#include "rapt.h"
#include <stdexcept>

using namespace rapt;

struct RthScene {
	Scene* sc = nullptr;
	bool owned = false;
	std::vector<int> instMesh;
	std::string err;
};
struct RthRenderer {
	Renderer* r = nullptr;
	RthScene scene;
	std::string err;
};

static float3 f3(const float* p) { return float3(p[0], p[1], p[2]); }
#define GUARD(s, expr) try { expr; } catch (const std::exception& e) { (s)->err = e.what(); return -1; }

extern "C" {

void* rth_scene_new() { RthScene* s = new RthScene(); s->sc = new Scene(); s->owned = true; return s; }
void rth_scene_free(void* h) { RthScene* s = (RthScene*)h; if (s && s->owned) { delete s->sc; delete s; } }
const char* rth_last_error(void* h) { return ((RthScene*)h)->err.c_str(); }

int rth_add_diffuse(void* h, const float* albedo, const float* col, float ks, float kd, int n, float emission, float shininess, int rt)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.materials.push_back(new diffuse(f3(albedo), f3(col), ks, kd, n, rt != 0, emission, shininess));
	return (int)sc.materials.size() - 1;
}
int rth_add_metal(void* h, float fuzzy, const float* col, int rt)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.materials.push_back(new metal(fuzzy, f3(col), rt != 0));
	return (int)sc.materials.size() - 1;
}
int rth_add_glass(void* h, float ir, const float* col, const float* absorption, int rt)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.materials.push_back(new glass(ir, f3(col), f3(absorption), 0.0f, 0, rt != 0));
	return (int)sc.materials.size() - 1;
}
int rth_add_area_light(void* h, int idx, const float* pos, float strength, const float* col, float radius, const float* normal)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.lights.push_back(new AreaLight(idx, f3(pos), strength, f3(col), radius, f3(normal), 4, sc.raytracer));
	return (int)sc.lights.size() - 1;
}
int rth_add_dir_light(void* h, int idx, const float* pos, float strength, const float* col, const float* normal, float r)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.lights.push_back(new DirectionalLight(idx, f3(pos), strength, f3(col), f3(normal), r, sc.raytracer));
	return (int)sc.lights.size() - 1;
}
int rth_add_sphere(void* h, int idx, int mat, const float* pos, float r)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.spheres.push_back(Sphere(idx, sc.materials[mat], f3(pos), r));
	return (int)sc.spheres.size() - 1;
}
int rth_add_plane(void* h, int idx, int mat, const float* N, float d)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.planes.push_back(Plane(idx, sc.materials[mat], f3(N), d));
	return (int)sc.planes.size() - 1;
}
int rth_add_mesh_raw(void* h, int group, int mat, const float* v9, int n)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.meshes.push_back(Mesh(group, sc.materials[mat], v9, n));
	return (int)sc.meshes.size() - 1;
}
int rth_add_mesh_obj(void* h, int group, const char* path, int mat, const float* pos, float scale)
{
	RthScene* s = (RthScene*)h;
	GUARD(s, s->sc->meshes.push_back(Mesh(group, std::string(path), s->sc->materials[mat], f3(pos), scale)));
	return (int)s->sc->meshes.size() - 1;
}
int rth_add_mesh_tri(void* h, int group, const char* path, int mat)
{
	RthScene* s = (RthScene*)h;
	GUARD(s, s->sc->meshes.push_back(Mesh(group, path, s->sc->materials[mat])));
	return (int)s->sc->meshes.size() - 1;
}
int rth_meshes(void* h) { return (int)((RthScene*)h)->sc->meshes.size(); }
int rth_mesh_count(void* h, int mesh) { const auto& ms = ((RthScene*)h)->sc->meshes; return mesh < 0 || mesh >= (int)ms.size() ? -1 : (int)ms[mesh].tri.size(); }
void rth_mesh_get(void* h, int mesh, float* out15, int* outIdx)
{
	if (rth_mesh_count(h, mesh) < 0) return;
	const Mesh& m = ((RthScene*)h)->sc->meshes[mesh];
	for (size_t i = 0; i < m.tri.size(); i++) {
		const Triangle& t = m.tri[i];
		const float3* src[5] = { &t.v0, &t.v1, &t.v2, &t.N, &t.centroid };
		for (int k = 0; k < 5; k++) { out15[15 * i + 3 * k] = src[k]->x; out15[15 * i + 3 * k + 1] = src[k]->y; out15[15 * i + 3 * k + 2] = src[k]->z; }
		outIdx[i] = t.objIdx;
	}
}
int rth_set_sky(void* h, int w, int hgt, int n, const unsigned char* px)
{
	Scene& sc = *((RthScene*)h)->sc;
	sc.skydomeX = w, sc.skydomeY = hgt, sc.skydomeN = n;
	sc.skydome.assign(px, px + (size_t)w * hgt * n);
	return 0;
}
int rth_scene_load_file(void* h, const char* path)
{
	RthScene* s = (RthScene*)h;
	GUARD(s, s->sc->LoadFile(path));
	return 0;
}
int rth_load_sky_hdr(void* h, const char* path)
{
	RthScene* s = (RthScene*)h;
	std::string why;
	if (!s->sc->LoadSkyHDR(path, &why)) { s->err = std::string(path) + ": " + why; return -1; }
	return 0;
}
// dims[0..2] = width, height, channels; returns the 8-bit texels
const unsigned char* rth_get_sky(void* h, int* dims)
{
	const Scene& sc = *((RthScene*)h)->sc;
	dims[0] = sc.skydomeX, dims[1] = sc.skydomeY, dims[2] = sc.skydomeN;
	return sc.skydome.data();
}
void rth_set_raytracer(void* h, int rt)
{
	Scene& sc = *((RthScene*)h)->sc;
	if (sc.raytracer != (rt != 0)) sc.toogleRaytracer();
}
int rth_get_raytracer(void* h) { return ((RthScene*)h)->sc->raytracer ? 1 : 0; }
void rth_scene_device_build(void* h, void* ctx) { ((RthScene*)h)->sc->deviceBuild = (rt_ctx*)ctx; }
int rth_build(void* h, int splitMethod)
{
	RthScene* s = (RthScene*)h;
	GUARD(s, s->sc->BuildBVH(splitMethod));
	return 0;
}
int rth_build_tlas(void* h, int splitMethod, int nInst, const int* meshIdx, const float* transforms)
{
	RthScene* s = (RthScene*)h;
	std::vector<int> mi(meshIdx, meshIdx + nInst);
	std::vector<mat4> T(nInst);
	for (int i = 0; i < nInst; i++) memcpy(T[i].cell, transforms + 16 * i, 64);
	GUARD(s, s->sc->BuildTLAS(mi, T, splitMethod));
	return 0;
}
void rth_mat4_trs(const float* t, float sc, float rx, float ry, float rz, float* out)
{
	mat4 M = mat4::Translate(f3(t)) * mat4::Scale(sc) * mat4::RotateX(rx) * mat4::RotateY(ry) * mat4::RotateZ(rz);
	memcpy(out, M.cell, 64);
}
void rth_mat4_inverse(const float* m, float* out) { mat4 a; memcpy(a.cell, m, 64); mat4 r = a.Inverted(); memcpy(out, r.cell, 64); }

// ---- dumps (what the builders produced) ----
static const bvh* pick(const Scene& sc, int blas)
{
	if (blas < 0) return sc.b;
	// distinct BLASes in order of first use by an instance (matches Scene::Describe)
	std::vector<const bvh*> seen;
	for (uint i = 0; i < sc.bvhCount; i++) {
		const bvh* b = sc.bvhList[i].blas;
		bool known = false;
		for (auto* q : seen) known |= (q == b);
		if (!known) seen.push_back(b);
	}
	return seen[blas];
}
int rth_blas_count(void* h)
{
	const Scene& sc = *((RthScene*)h)->sc;
	std::vector<const bvh*> seen;
	for (uint i = 0; i < sc.bvhCount; i++) { bool known = false; for (auto* q : seen) known |= (q == sc.bvhList[i].blas); if (!known) seen.push_back(sc.bvhList[i].blas); }
	return (int)seen.size();
}
void rth_bvh_info(void* h, int blas, int* info)
{
	const bvh* b = pick(*((RthScene*)h)->sc, blas);
	info[0] = b->nodesUsed, info[1] = b->N, info[2] = b->NTri, info[3] = b->NSph, info[4] = b->NPla, info[5] = b->treeDepth, info[6] = 0;
}
void rth_bvh_get(void* h, int blas, void* nodes, unsigned* primIdx)
{
	const bvh* b = pick(*((RthScene*)h)->sc, blas);
	memcpy(nodes, b->bvhNode, (size_t)b->nodesUsed * sizeof(BVHNode));
	memcpy(primIdx, b->primitiveIdx, (size_t)b->N * 4);
}
int rth_tlas_nodes_used(void* h) { return (int)((RthScene*)h)->sc->tl->nodesUsed; }
void rth_tlas_get(void* h, void* nodes) { const tlas* t = ((RthScene*)h)->sc->tl; memcpy(nodes, t->tlasNode, (size_t)t->nodesUsed * sizeof(TLASNode)); }
void rth_instance_get(void* h, int i, int* blas, float* T, float* invT, float* bounds)
{
	const Scene& sc = *((RthScene*)h)->sc;
	const bvhInstance& in = sc.bvhList[i];
	int k = 0, n = rth_blas_count(h);
	for (k = 0; k < n; k++) if (pick(sc, k) == in.blas) break;
	*blas = k;
	memcpy(T, in.matTransform.cell, 64);
	memcpy(invT, in.invTransform.cell, 64);
	bounds[0] = in.bounds.bmin.x, bounds[1] = in.bounds.bmin.y, bounds[2] = in.bounds.bmin.z;
	bounds[3] = in.bounds.bmax.x, bounds[4] = in.bounds.bmax.y, bounds[5] = in.bounds.bmax.z;
}
// flattened scene as handed to rt_upload_scene (pointer stays valid until the scene changes)
const void* rth_describe(void* h)
{
	RthScene* s = (RthScene*)h;
	try { return &s->sc->Describe(); } catch (const std::exception& e) { s->err = e.what(); return nullptr; }
}

// ---- renderer (needs a GPU) ----
void* rth_renderer_new(int w, int hgt, int device)
{
	RthRenderer* r = new RthRenderer();
	r->r = new Renderer(w, hgt, device);
	r->scene.sc = &r->r->scene;
	r->scene.owned = false;
	return r;
}
// one context per entry of devices[] (an entry may repeat: two contexts on one GPU)
void* rth_renderer_new_multi(int w, int hgt, int n, const int* devices)
{
	RthRenderer* r = (RthRenderer*)rth_renderer_new(w, hgt, n > 0 ? devices[0] : 0);
	if (n > 0) r->r->UseDevices(std::vector<int>(devices, devices + n));
	return r;
}
int rth_renderer_contexts(void* h) { return (int)((RthRenderer*)h)->r->ctxs.size(); }
void rth_renderer_free(void* h) { RthRenderer* r = (RthRenderer*)h; if (r) { delete r->r; delete r; } }
const char* rth_renderer_error(void* h) { return ((RthRenderer*)h)->err.c_str(); }
void* rth_renderer_scene(void* h) { return &((RthRenderer*)h)->scene; }
int rth_renderer_init(void* h) { RthRenderer* r = (RthRenderer*)h; GUARD(r, r->r->Init()); return 0; }
void* rth_renderer_ctx(void* h) { return ((RthRenderer*)h)->r->ctx; }
int rth_renderer_commit(void* h) { RthRenderer* r = (RthRenderer*)h; GUARD(r, r->r->Commit()); return 0; }
void rth_renderer_set_camera(void* h, const float* camPos, const float* TL, const float* TR, const float* BL, int fisheye, float viewAngle, float yAngle)
{
	Camera& c = ((RthRenderer*)h)->r->camera;
	c.camPos = f3(camPos), c.topLeft = f3(TL), c.topRight = f3(TR), c.bottomLeft = f3(BL);
	c.fishEye = fisheye != 0, c.viewAngle = viewAngle, c.yAngle = yAngle, c.changed = true;
}
void rth_renderer_get_camera(void* h, float* out12)
{
	const Camera& c = ((RthRenderer*)h)->r->camera;
	const float3* v[4] = { &c.camPos, &c.topLeft, &c.topRight, &c.bottomLeft };
	for (int k = 0; k < 4; k++) out12[3 * k] = v[k]->x, out12[3 * k + 1] = v[k]->y, out12[3 * k + 2] = v[k]->z;
}
int rth_renderer_sync_camera(void* h) { RthRenderer* r = (RthRenderer*)h; GUARD(r, r->r->SyncCamera()); return 0; }
void rth_renderer_set_download(void* h, int on) { ((RthRenderer*)h)->r->downloadEachTick = on != 0; }
int rth_renderer_iteration(void* h) { return ((RthRenderer*)h)->r->scene.GetIterationNumber(); }
int rth_renderer_tick(void* h) { RthRenderer* r = (RthRenderer*)h; GUARD(r, r->r->Tick(0.0f)); return 0; }
int rth_renderer_qlearning(void* h, int grid, const float* lo, const float* hi, float alpha, float eps, float qInit)
{
	RthRenderer* r = (RthRenderer*)h;
	GUARD(r, if (grid > 0) r->r->EnableQLearning(grid, float3(lo[0], lo[1], lo[2]), float3(hi[0], hi[1], hi[2]), alpha, eps, qInit); else r->r->DisableQLearning());
	return 0;
}
const float* rth_renderer_accumulator(void* h) { return &((RthRenderer*)h)->r->accumulator[0].x; }
const unsigned* rth_renderer_pixels(void* h) { return ((RthRenderer*)h)->r->screenPixels; }
int rth_renderer_trace(void* h, int path, const float* O, const float* D, int depth, const float* energy, float* rgb)
{
	RthRenderer* r = (RthRenderer*)h;
	Ray ray(f3(O), f3(D), float3(0));
	float3 c;
	GUARD(r, c = path ? r->r->Sample(ray, depth, f3(energy)) : r->r->Trace(ray, depth, f3(energy)));
	rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
	return 0;
}
int rth_scene_set_time(void* h, float t)
{
	RthScene* s = (RthScene*)h;
	GUARD(s, s->sc->SetTime(t));
	return 0;
}
int rth_scene_find_nearest(void* h, const float* O, const float* D, float tmax, float t_min, float* t, int* obj, float* normal)
{
	RthScene* s = (RthScene*)h;
	Ray ray(f3(O), f3(D), float3(0), tmax);
	GUARD(s, s->sc->FindNearest(ray, t_min));
	*t = ray.t, *obj = ray.objIdx;
	normal[0] = ray.hitNormal.x, normal[1] = ray.hitNormal.y, normal[2] = ray.hitNormal.z;
	return 0;
}
// the members below Scene level: which = 0 bvh (scene bvh, or BLAS 'index' in TLAS mode), 1 tlas, 2 bvhInstance 'index'
int rth_member_intersect(void* h, int which, int index, const float* O, const float* D, float tmax, float* t, int* obj, float* normal)
{
	RthScene* s = (RthScene*)h;
	Ray ray(f3(O), f3(D), float3(0), tmax);
	ray.objIdx = -1;
	GUARD(s, {
		Scene& sc = *s->sc;
		if (which == 0) { bvh* bv = sc.useTLAS ? nullptr : sc.b; if (sc.useTLAS) for (uint i = 0; i < sc.bvhCount; i++) if (sc.bvhList[i].blas->blasIndex == index) bv = sc.bvhList[i].blas; if (!bv) throw std::runtime_error("no such bvh"); bv->Intersect(ray); }
		else if (which == 1) { if (!sc.tl) throw std::runtime_error("no tlas"); sc.tl->Intersect(ray); }
		else { if (index < 0 || index >= (int)sc.bvhCount) throw std::runtime_error("no such instance"); sc.bvhList[index].BIntersect(ray); }
	});
	*t = ray.t, *obj = ray.objIdx;
	normal[0] = ray.hitNormal.x, normal[1] = ray.hitNormal.y, normal[2] = ray.hitNormal.z;
	return 0;
}
int rth_member_occluded(void* h, int which, int index, const float* O, const float* D, float tmax)
{
	RthScene* s = (RthScene*)h;
	Ray ray(f3(O), f3(D), float3(0), tmax);
	bool o = false;
	GUARD(s, {
		Scene& sc = *s->sc;
		if (which == 0) { bvh* bv = sc.useTLAS ? nullptr : sc.b; if (sc.useTLAS) for (uint i = 0; i < sc.bvhCount; i++) if (sc.bvhList[i].blas->blasIndex == index) bv = sc.bvhList[i].blas; if (!bv) throw std::runtime_error("no such bvh"); o = bv->IsOccluded(ray); }
		else if (which == 1) { if (!sc.tl) throw std::runtime_error("no tlas"); o = sc.tl->IsOccluded(ray); }
		else { if (index < 0 || index >= (int)sc.bvhCount) throw std::runtime_error("no such instance"); o = sc.bvhList[index].IsOccluded(ray); }
	});
	return o ? 1 : 0;
}
int rth_scene_sky_color(void* h, const float* D, float* rgb)
{
	RthScene* s = (RthScene*)h;
	Ray ray(float3(0), f3(D), float3(0));
	float3 c;
	GUARD(s, c = s->sc->GetSkyColor(ray));
	rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
	return 0;
}
int rth_scene_is_occluded(void* h, const float* O, const float* D, float tmax)
{
	RthScene* s = (RthScene*)h;
	Ray ray(f3(O), f3(D), float3(0), tmax);
	bool o = false;
	GUARD(s, o = s->sc->IsOccluded(ray));
	return o ? 1 : 0;
}

} // extern "C"
