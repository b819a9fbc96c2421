// rapt.h -- C++ host mirror of the reference's interface for the trace-loop path.
//
// Same class names, member names, constructor argument orders and call shapes as the reference
// (template/scene.h, bvh.h, bvhInstance.h, tlas.h, camera.h, renderer.h) so that host code written
// against it -- scene factories, the OBJ/.tri loaders' callers, the frame loop -- ports by changing
// the namespace.  What differs: the queries and the pixel loop execute on the GPU through the C ABI
// of include/rt_amd.h; host objects hold parameters and the builders' output only.
//
//   reference                                   here
//   Scene::FindNearest(Ray&, float) const       Scene::FindNearest (1 ray) / FindNearestBatch
//   Scene::IsOccluded(Ray&) const               Scene::IsOccluded  (1 ray) / IsOccludedBatch
//   Renderer::Init / Tick / Trace / Sample      same names, same arguments
//   bvh::Build, Refit, bvhInstance::SetTransform, tlas::build   host-side, same outputs
//
// After the last scene edit call Scene::Commit(): it builds nothing, it only flattens the objects
// into an rt_scene_desc and uploads it (the reference needs no such step because its queries chase
// host pointers).
#pragma once
#include "rapt_math.h"
#include "../../include/rt_amd.h"
#include <string>
#include <vector>

namespace rapt {

enum MAT_TYPE { DIFFUSE = 1, METAL = 2, GLASS = 3 }; // template/scene.h:33-37

// template/scene.h:582-676 (parameters only; scatter / fresnel / RefractRay run on the device)
class material {
public:
	material(float3 c, bool rt) : col(c), raytracer(rt) {}
	virtual ~material() {}
	void SetColor(float3 c) { col = c; }
	float3 col, albedo = float3(0), emission = float3(0);
	int type = DIFFUSE;
	bool raytracer;
};
class diffuse : public material {
public:
	diffuse(float3 a = float3(0), float3 c = float3(0), float ks = 0.2f, float kd = 0.8f, int n = 2, bool rt = true, float e = 0, float s = 0)
		: material(c, rt), specu(ks), diffu(kd), shinieness(s), N(n) { type = DIFFUSE; albedo = a; emission = float3(e); }
	float specu, diffu, shinieness;
	int N;
};
class metal : public material {
public:
	metal(float f, float3 c, bool rt) : material(c, rt), fuzzy(f < 1 ? f : 1) { type = METAL; }
	float fuzzy;
};
class glass : public material {
public:
	glass(float refIndex, float3 c, float3 a, float r, float n, bool rt)
		: material(c, rt), ir(refIndex), specu(r), N(n), absorption(a) { type = GLASS; invIr = 1 / ir; }
	float ir, specu, N, invIr;
	float3 absorption;
};

// template/scene.h:38-73
class Ray {
public:
	Ray() = default;
	Ray(float3 origin, float3 direction, float3 color, float distance = 1e34f)
	{
		O = origin, D = direction, t = distance;
		rD = float3(1 / D.x, 1 / D.y, 1 / D.z);
		this->color = color;
		exists = true;
	}
	float3 IntersectionPoint() const { return O + t * D; }
	void SetMaterial(material* mat) { m = mat; }
	material* GetMaterial() { return m; }
	void SetNormal(float3 normal) { hitNormal = normal; }
	float3 O, D, rD;
	float t = 1e34f;
	int objIdx = -1;
	bool inside = false, exists = false;
	float3 color = float3(0);
	float3 hitNormal;
	material* m = nullptr;
};

// template/scene.h:75-168 (parameters; Intersect / GetLightIntensityAt / GetLightPosition on device)
class Light {
public:
	Light() = default;
	Light(int idx, float3 p, float str, float3 c, float3 n, bool rt) : pos(p), raytracer(rt), col(c), strength(str), objIdx(idx), normal(n) {}
	virtual ~Light() {}
	virtual int kind() const { return RT_LIGHT_BASE; }
	virtual void updateTracing(bool rt) { raytracer = rt; }
	float3 pos;
	bool raytracer = true;
	float3 col;
	float strength = 1;
	int objIdx = 0;
	float3 normal;
};
class AreaLight : public Light {
public:
	AreaLight(int idx, float3 p, float str, float3 c, float r, float3 n, int s, bool rt) : Light(idx, p, str, c, n, rt)
	{
		radius = r, radius2 = r * r, samples = s, area = 2 * radius2 * 3.14159265358979323846264f;
	}
	int kind() const override { return RT_LIGHT_AREA; }
	int samples;
	float radius, radius2, area;
};
class DirectionalLight : public Light {
public:
	DirectionalLight(int idx, float3 p, float str, float3 c, float3 n, float r, bool rt) : Light(idx, p, str, c, n, rt)
	{
		sinAngle = sinf_r(r * 3.14159265358979323846264f / 2);
	}
	int kind() const override { return RT_LIGHT_DIRECTIONAL; }
	float sinAngle;
};

// template/scene.h:175-251
class Triangle {
public:
	Triangle() = default;
	Triangle(int idx, material* m, float3 ver0, float3 ver1, float3 ver2) : v0(ver0), v1(ver1), v2(ver2), objIdx(idx), mat(m) { derive(); }
	Triangle(int idx, material* m, int3 facesIdx, const std::vector<float3>& vertices)
		: v0(vertices[facesIdx.x]), v1(vertices[facesIdx.y]), v2(vertices[facesIdx.z]), objIdx(idx), mat(m) { derive(); }
	void update(int3 faces, const std::vector<float3>& vertices) { v0 = vertices[faces.x], v1 = vertices[faces.y], v2 = vertices[faces.z]; derive(); }
	float3 GetNormal(const float3) const { return N; }
	float3 v0, v1, v2, e1, e2, centroid, N;
	int objIdx = -1;
	material* mat = nullptr;
private:
	void derive()
	{
		e1 = v1 - v0;
		e2 = v2 - v0;
		N = normalize(cross(e1, e2));
		centroid = (v0 + v1 + v2) * 0.333f;
	}
};

// template/scene.h:258-340.  The file constructors keep the reference's argument orders; failures
// throw std::runtime_error (the reference exits or crashes).
class Mesh {
public:
	Mesh() = default;
	Mesh(int idGroup, const char* path, material* m);                             // .tri
	Mesh(int idGroup, std::string path, material* m, float3 pos, float scale);    // .obj
	Mesh(int idGroup, material* m, const float* v9, int n);                       // in-memory triangles
	uint getSize() const { return (uint)tri.size(); }
	void update() { for (size_t i = 0; i < faces.size(); i++) { int3 f = faces[i]; f.x--, f.y--, f.z--; tri[i].update(f, vertices); } }
	std::vector<float3> vertices;
	std::vector<int3> faces;
	std::vector<Triangle> tri;
	std::vector<float3> originalVerts;
	material* mat = nullptr;
	int groupIdx = -1;
};

// template/scene.h:347-394
class Sphere {
public:
	Sphere() = default;
	Sphere(int idx, material* m, float3 p, float r) : pos(p), r2(r * r), invr(1 / r), r(r), objIdx(idx), mat(m) {}
	float3 pos = float3(0);
	float r2 = 0, invr = 0, r = 0;
	int objIdx = -1;
	material* mat = nullptr;
};
// template/scene.h:401-448
class Plane {
public:
	Plane() = default;
	Plane(int idx, material* m, float3 normal, float dist) : N(normal), d(dist), objIdx(idx), mat(m) {}
	float3 N;
	float d = 0;
	int objIdx = -1;
	material* mat = nullptr;
};

class Scene;

struct BVHNode { // bvh.h:10-24
	float3 aabbMin; uint leftFirst;
	float3 aabbMax; uint primCount;
	bool isLeaf() const { return primCount > 0; }
};
static_assert(sizeof(BVHNode) == 32, "BVHNode must match rt_bvh_node");

enum SplitMethod { BINNEDSAH = 0, SAMESIZE = 1, LONGESTAXIS = 2, SAH = 3 }; // bvh.h:38-43

// bvh.h:45-86.  Build() and Refit() run on the host and reproduce the reference's node array
// and primitiveIdx order (builder quirks included: they decide hit ids).  QBVH is not built
// (disabled in the reference, template/scene.h:704-707).
class bvh {
public:
	bvh(Scene* s);
	bvh(Mesh* m);
	~bvh();
	void Build(bool isQ = false);
	// Build() through rt_build_bvh (include/rt_amd.h): the same tree, made on the device.  BINNEDSAH only;
	// throws for what the device builder refuses (no triangle or sphere, non-finite geometry)
	void BuildOnDevice(rt_ctx* ctx);
	void Refit();
	// bvh.h:57, :65 -- on the device, in this bvh's own space (rt_intersect_scope RT_SCOPE_BLAS); valid after Scene::Commit
	void Intersect(Ray& ray);
	bool IsOccluded(Ray& ray);
	Scene* owner = nullptr; int blasIndex = -1; // set by Scene::Commit
	uint rootNodeIdx = 0, nodesUsed = 2, NTri = 0, NSph = 0, NPla = 0, N = 0;
	uint* primitiveIdx = nullptr;
	Scene* scene = nullptr;
	BVHNode* bvhNode = nullptr;
	Mesh* mesh = nullptr;
	aabb bounds;
	int splitMethod = BINNEDSAH;
	int treeDepth = 0; // DataCollector::maxTreeDepth analogue
private:
	struct Builder;
};

// bvhInstance.h
class bvhInstance {
public:
	bvhInstance() = default;
	bvhInstance(bvh* blas);
	void SetTransform(mat4& transform);
	// bvhInstance.h:11-12 -- on the device (RT_SCOPE_INSTANCE); valid after Scene::Commit
	void BIntersect(Ray& ray);
	bool IsOccluded(Ray& ray);
	Scene* owner = nullptr; int index = -1; // set by Scene::Commit
	mat4 invTransform, matTransform;
	bvh* blas = nullptr; // the reference names this member 'bvh'
	aabb bounds;
};

struct TLASNode { // tlas.h:4-11
	float3 aabbMin; uint leftRight;
	float3 aabbMax; uint BLAS;
	bool isLeaf() const { return leftRight == 0; }
};
static_assert(sizeof(TLASNode) == 32, "TLASNode must match rt_tlas_node");

// tlas.h:13-29
class tlas {
public:
	tlas(bvhInstance* bvhList, int N);
	~tlas();
	void build();
	void BuildOnDevice(rt_ctx* ctx); // tlas::build through rt_build_tlas
	// tlas.h:20-21 -- on the device (RT_SCOPE_ACCEL); valid after Scene::Commit
	void Intersect(Ray& ray);
	bool IsOccluded(Ray& ray);
	Scene* owner = nullptr; // set by Scene::Commit
	int FindBestMatch(int* list, int N, int A);
	TLASNode* tlasNode = nullptr;
	uint nodesUsed = 0;
	bvhInstance* blas = nullptr;
	uint blasCount = 0;
};

// camera.h (state read by GetPrimaryRay; the interactive move/rotate helpers are UI, out of scope)
class Camera {
public:
	Camera() { Reset(600, 400); }
	void Reset(int w, int h)
	{
		aspect = (float)w / (float)h;
		camPos = float3(0, 1, -2);
		topLeft = float3(-aspect, 2, 0);
		topRight = float3(aspect, 2, 0);
		bottomLeft = float3(-aspect, 0, 0);
		changed = false; // camera.h:52: the constructor does not mark a change; the mutators do
	}
	void ToogleFisheye() { changed = true; fishEye = !fishEye; }
	void SetChange(bool s) { changed = s; }
	bool GetChange() const { return changed; }
	float aspect = 1.5f;
	float viewAngle = 0.25f;
	float3 camPos, topLeft, topRight, bottomLeft;
	float yAngle = 0;
	bool paused = false, fishEye = false, changed = false;
};

// template/scene.h:685-1397: container + the three queries.  Owns what is added to it.
class Scene {
public:
	Scene();
	~Scene();
	// the default construction path of the reference (template/scene.h:688-716): after filling the
	// containers call BuildBVH() (new bvh(this); Build) or BuildTLAS() (tlas(bvhList, bvhCount); build)
	// deviceBuild != nullptr: BuildBVH / BuildTLAS make their trees (any split method) and the TLAS on that context's GPU
	// (bvh::BuildOnDevice -> rt_build_bvh_split, tlas::BuildOnDevice -> rt_build_tlas)
	rt_ctx* deviceBuild = nullptr;
	void BuildBVH(int splitMethod = BINNEDSAH);
	// instances reference meshes by index; one bvh per distinct mesh (TLASSceneTest2 shares one)
	void BuildTLAS(const std::vector<int>& meshOfInstance, const std::vector<mat4>& transforms, int splitMethod = BINNEDSAH);
	// Scene description file (host/scene_file.cpp): the data a reference scene factory hard-codes
	// (template/scene.h:791-1209) as a text file; builds the BVH / TLAS it names.  Throws on errors.
	void LoadFile(const std::string& path);
	// skydome = stbi_load(path, &x, &y, &n, 3) for a Radiance .hdr file (template/scene.h:792)
	bool LoadSkyHDR(const char* path, std::string* why = nullptr);
	// flatten + upload to the device context (must be called before queries / rendering)
	void Commit(rt_ctx* ctx);
	void CommitAlso(rt_ctx* other); // the same flattened scene to one more context (multi-GPU: every GPU holds a copy); SetTime reaches it too

	// template/scene.h:1210: mesh wobble + bvh::Refit, executed on the device (rt_set_time); the
	// reference runs it only when animOn (raytracer && defaultAnim && !useTLAS)
	void SetTime(float t);
	void FindNearest(Ray& ray, float t_min) const;   // template/scene.h:1248
	bool IsOccluded(Ray& ray) const;                 // template/scene.h:1286
	float3 GetSkyColor(Ray& ray) const;              // template/scene.h:1312, evaluated on the device (rt_sky_color_batch)
	// the queries below Scene level (bvh / tlas / bvhInstance members call these)
	void ScopeNearest(int scope, int index, Ray& ray) const;
	bool ScopeOccluded(int scope, int index, Ray& ray) const;
	void FindNearestBatch(int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out) const;
	void IsOccludedBatch(int n, const float* O, const float* D, const float* tmax, uint8_t* out) const;

	uint getTriangleNb() const;                      // :1329
	const Triangle& getTriangle(uint idx) const;     // :1337
	void toogleRaytracer();                          // :1352
	void SetIterationNumber(int i) { iterationNumber = i; }
	int GetIterationNumber() const { return iterationNumber; }

	// flattening (also used by tests to inspect what is uploaded)
	const rt_scene_desc& Describe();
	int materialIndex(const material* m) const;

	int skydomeX = 0, skydomeY = 0, skydomeN = 3;
	std::vector<unsigned char> skydome;
	bvh* b = nullptr; tlas* tl = nullptr; bvhInstance* bvhList = nullptr;
	uint bvhCount = 0;
	mat4* Transforms = nullptr;
	std::vector<Light*> lights;
	std::vector<Sphere> spheres;
	std::vector<Mesh> meshes;
	std::vector<Plane> planes;
	std::vector<material*> materials; // owned; the reference leaks them
	int aaSamples = 1;
	int iterationNumber = 1;
	int totIterationNumber = 0;
	bool raytracer = true;
	bool useTLAS = false;
	rt_ctx* ctx = nullptr;
	std::vector<rt_ctx*> alsoCtx; // the further contexts the scene was committed to (CommitAlso): SetTime animates every copy
private:
	struct Flat;
	Flat* flat = nullptr;
	std::vector<bvh*> blasOwned;
};

// renderer.h / renderer.cpp
// One GPU: exactly the reference's shape.  Several GPUs: UseDevices() before Init() makes Init create
// one rt_ctx per device; Commit() uploads the scene to each; Tick() renders the interleaved rows k, k + n, ... on
// context k from one host thread per context, gathers the rows into context 0's accumulator device to device
// (rt_gather_rows) and resolves there.  Pixels are independent (per-pixel RNG streams), so the frame is the
// one-GPU frame bit for bit.
class Renderer {
public:
	Renderer(int width, int height, int device = 0);
	~Renderer();
	void UseDevices(const std::vector<int>& devs);         // before Init(); a device may be named twice (two contexts on it)
	void UseAllDevices();                                  // every device rt_device_count() reports
	void Init();                                           // renderer.cpp:5-11
	void Commit();                                         // scene.Commit() on every context
	float3 Trace(Ray& ray, int depth, float3 energy);      // renderer.cpp:21   } with scene.raytracer as the Scene holds it: every combination of function
	float3 Sample(Ray& ray, int depth, float3 energy);     // renderer.cpp:128  } and flag the reference's bodies contain (:33-43, :107-121, :143-153 included)
	void Tick(float deltaTime);                            // renderer.cpp:240
	void Shutdown();
	// Q-learning guided sampling of the indirect bounce in path mode (README.md:36-42 of the reference names Dahm & Keller 2017;
	// the snapshot holds no code for it, so this is the library's own statement: rt_qlearn_*, PARITY UNPINNED).  Tick folds the
	// frame's rewards into the table after every path frame; with several contexts their integer reward sums are added first, so
	// every GPU learns the same table and the frame does not depend on how its rows were sharded.
	void EnableQLearning(int grid, float3 lo, float3 hi, float alpha = 0.3f, float epsilon = 0.2f, float qInit = 1.0f);
	void DisableQLearning();
	bool qlearning = false;
	int qgrid = 0;
	uint32_t qlearnMask = 0;          // rt_qlearn_params::learn_mask for the next EnableQLearning (0: every sample pays rewards)
	float4* accumulator = nullptr; // host copy, refreshed by Tick
	uint32_t* screenPixels = nullptr; // Surface::pixels analogue (template/precomp.h:134)
	Scene scene;
	Camera camera;
	int width, height, device;
	rt_ctx* ctx = nullptr;            // context 0: holds the gathered accumulator, resolves the pixels
	std::vector<rt_ctx*> ctxs;        // all contexts, ctxs[0] == ctx
	std::vector<int> devices;
	uint32_t seedBase = 0x12345678; // template/template.cpp:671
	uint32_t frame = 0;
	bool downloadEachTick = true;
	void SyncCamera();
private:
	struct Workers;                   // one parked host thread per context (several contexts only)
	Workers* workers = nullptr;
};

} // namespace rapt
