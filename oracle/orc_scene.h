// ORACLE (test infrastructure, NOT product code) -- parity unpinned, see oracle/README.md.
//
// CPU restatement of the reference's scene data and leaf tests: Ray, lights, Triangle, Mesh
// loaders, Sphere, Plane, materials and the Scene container (template/scene.h).  Materials are
// tagged records addressed by index instead of a class hierarchy reached through pointers; the
// arithmetic of every function follows the cited reference lines.
#pragma once
#include "orc_math.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace orc {

enum MatType { DIFFUSE = 1, METAL = 2, GLASS = 3 }; // template/scene.h:33-37

// Work counters: the reference's DataCollector tallies (bvh.cpp:610-631) plus ray counts; these
// define the "algorithmic work" of SURVEY.md section 8(d).
struct Counters {
	uint64_t inner_visits = 0;   // BLAS inner-node visits (two children fetched, bvh.cpp:638-645)
	uint64_t prim_tests = 0;     // leaf primitive tests (bvh.cpp:616-629)
	uint64_t tlas_inner = 0;     // TLAS inner-node visits (tlas.cpp:77-80)
	uint64_t instance_visits = 0;// bvhInstance entries (tlas.cpp:73)
	uint64_t rays_nearest = 0;   // Scene::FindNearest calls
	uint64_t rays_occluded = 0;  // Scene::IsOccluded calls
	uint64_t brute_tests = 0;    // TLAS-mode brute force sphere / plane tests (template/scene.h:1260-1261)
	uint64_t light_tests = 0;    // light->Intersect calls (template/scene.h:1257)
	uint64_t tri_intersect_calls = 0; // Triangle::Intersect calls from BIntersect leaves (bvh.cpp:619): the figure gprof reports for the reference
	void add(const Counters& o)
	{
		inner_visits += o.inner_visits; prim_tests += o.prim_tests; tlas_inner += o.tlas_inner;
		instance_visits += o.instance_visits; rays_nearest += o.rays_nearest; rays_occluded += o.rays_occluded;
		brute_tests += o.brute_tests; light_tests += o.light_tests; tri_intersect_calls += o.tri_intersect_calls;
	}
};

// template/scene.h:38-73.  'color' is carried by the reference but never reaches a pixel; it
// is omitted.  'mat' is an index into Scene::materials (-1: none).
struct Ray {
	float3 O = float3(0), D = float3(0), rD = float3(0);
	float t = 1e34f;
	int objIdx = -1;
	float3 hitNormal = float3(0);
	int mat = -1;
	Ray() = default;
	Ray(const float3& origin, const float3& direction, float distance = 1e34f)
	{
		O = origin, D = direction, t = distance;
		rD = float3(1 / D.x, 1 / D.y, 1 / D.z); // :47
	}
	float3 IntersectionPoint() const { return O + t * D; } // :52
};

// template/scene.h:582-676, flattened.  'raytracer' is the flag captured at construction time
// (material::raytracer); Scene::toogleRaytracer does NOT update it (template/scene.h:1352-1357).
struct Material {
	int type = DIFFUSE;
	float3 col = float3(0), albedo = float3(0), emission = float3(0);
	bool raytracer = true;
	// diffuse
	float specu = 0.2f, diffu = 0.8f, shinieness = 0;
	int N = 2;
	// metal
	float fuzzy = 0;
	// glass
	float ir = 1, gspecu = 0, gN = 0, invIr = 1;
	float3 absorption = float3(0);
};

struct Scene;

// diffuse::scatter (template/scene.h:605-620).  Returns 'att', updates 'energy', and in a
// material built with raytracer == false draws one hemisphere sample (whose value only matters
// to the dead indirect block of Renderer::Trace; quirk Q13).
static inline void diffuse_scatter(const Material& m, const Ray& ray, float3& att, float3& scatteredDir,
                                   const float3& lightDir, const float3& lightIntensity, const float3& normal,
                                   float3& energy, uint& seed)
{
	float3 reflectionDirection = reflect(-lightDir, normal);
	float3 specularColor = x_powf(libm_fmaxf(0.0f, -dot(reflectionDirection, ray.D)), (float)m.N) * lightIntensity;
	float3 lightAttenuation = lightIntensity;
	att = m.albedo * lightAttenuation * m.diffu + specularColor * m.specu;
	float3 dir(0); // uninitialised in the reference when raytracer is set; never read in that case
	if (!m.raytracer) dir = RandomInHemisphere(seed, normal);
	scatteredDir = dir;
	float3 retention = float3(1) - m.albedo;
	float3 newEnergy(energy - retention);
	energy = newEnergy.x > 0 ? newEnergy : float3(0);
}

// metal::scatter (template/scene.h:630-635)
static inline Ray metal_scatter(const Ray& ray, const float3& normal)
{
	float3 dir = reflect(ray.D, normal);
	return Ray(ray.IntersectionPoint() + normal * 0.001f, dir);
}

// glass::fresnel (template/scene.h:647-666)
static inline void glass_fresnel(const float3& I, const float3& N, float ior, float& kr)
{
	float cosi = t_clamp(dot(I, N), -1.0f, 1.0f);
	float etai = 1, etat = ior;
	if (cosi > 0) { float tmp = etai; etai = etat; etat = tmp; }
	float sint = etai / etat * sqrtf(t_fmaxf(0.f, 1 - cosi * cosi));
	if (sint >= 1) {
		kr = 1;
	} else {
		float cost = sqrtf(t_fmaxf(0.f, 1 - sint * sint));
		cosi = fabsf(cosi);
		float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
		float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
		kr = (Rs * Rs + Rp * Rp) / 2;
	}
}

// glass::RefractRay (template/scene.h:667-672).  fmin(.,1.0), 1.0 - pow(len,2), fabs and sqrt
// run in double in the reference; pow(x,2) of a float x is exactly x*x in double.
static inline float3 glass_refract(const float3& oRayDir, const float3& normal, float refRatio)
{
	double thd = (double)dot(-oRayDir, normal);
	float theta = (float)(thd < 1.0 ? thd : 1.0);
	float3 perpendicular = refRatio * (oRayDir + theta * normal);
	double len = (double)length(perpendicular);
	float par = (float)(-sqrt(fabs(1.0 - len * len)));
	float3 parallel = par * normal;
	return perpendicular + parallel;
}

// template/scene.h:75-168.  kind 0 = AreaLight, 1 = DirectionalLight, 2 = base Light.
struct Light {
	int kind = 0;
	int objIdx = 0;
	float3 pos = float3(0), col = float3(1), normal = float3(0, -1, 0);
	float strength = 1;
	bool raytracer = true;
	float radius = 0, radius2 = 0, area = 0; // AreaLight
	float sinAngle = 0;                      // DirectionalLight

	// AreaLight::Intersect (:105-120); the other kinds do nothing (:84, :149-151).  Note the
	// missing 't < ray.t' test (quirk Q6) and 't - 1e-6' evaluated in double.
	void Intersect(Ray& ray, float t_min) const
	{
		if (kind != 0) return;
		float d = dot(normal, ray.D);
		float3 dir = pos - ray.O;
		float t = dot(dir, normal) / d;
		if (t >= t_min) {
			float3 intersection = ray.O + ray.D * t;
			float3 v = intersection - pos;
			float dis2 = dot(v, v);
			if (sqrtf(dis2) <= radius) {
				ray.t = (float)((double)t - 1e-6), ray.hitNormal = normal;
				ray.objIdx = objIdx;
			}
		}
	}
	// :121-129 (area), :155-165 (directional), :83 (base)
	float3 GetLightIntensityAt(const float3& p, const float3& n, const float3& from) const
	{
		if (kind == 0) {
			float dis = length(from - p);
			float3 dir = from - p;
			float cos_ang = dot(normalize(n), normalize(dir));
			float relStr = 1 / (dis * PI) * strength;
			if (dis <= radius && isZero(float3(cos_ang))) return float3(strength * col);
			return relStr * col;
		}
		if (kind == 1) {
			float3 dir = p - pos;
			float sTheta = length(cross(dir, normal)) / length(dir) * length(normal);
			if (dot(dir, normal) < 0) return float3(0);
			float dis = length(dir);
			float str = sinAngle - sTheta > 0 ? x_asinf(sinAngle) - x_asinf(sTheta) : 0;
			return float3(1 / dis * str * strength);
		}
		return float3(1);
	}
	// :132-137 (area: two draws, radius then angle), :152-154, :81
	float3 GetLightPosition(uint& seed) const
	{
		if (kind != 0) return pos;
		if (raytracer) return pos;
		float newRad = radius * sqrtf(RandomFloat(seed));
		float theta = RandomFloat(seed) * 2 * PI;
		return float3(pos.x + newRad * x_cosf(theta), pos.y + newRad * x_sinf(theta), pos.z);
	}
};

// template/scene.h:175-251
struct Triangle {
	float3 v0, v1, v2, e1, e2, centroid, N;
	int objIdx = -1;
	int mat = -1;
	Triangle() = default;
	Triangle(int idx, int m, const float3& a, const float3& b, const float3& c) : v0(a), v1(b), v2(c), objIdx(idx), mat(m)
	{
		e1 = v1 - v0;
		e2 = v2 - v0;
		N = normalize(cross(e1, e2));
		centroid = (v0 + v1 + v2) * 0.333f;
	}
	// :190-215; returns true when the edge tests pass and t lies in (t_min, ray.t)
	bool HitT(const Ray& ray, float t_min, float& tOut) const
	{
		float NdotRayDir = dot(N, ray.D);
		if (fabsf(NdotRayDir) < t_min) return false; // quirk Q15: t_min doubles as the parallel epsilon
		float d = -dot(N, v0);
		float t = -(dot(N, ray.O) + d) / NdotRayDir;
		if (t < 0) return false;
		float3 p = ray.O + t * ray.D;
		float3 c;
		float3 vp0 = p - v0;
		c = cross(e1, vp0);
		if (dot(N, c) < 0) return false;
		float3 vp1 = p - v1;
		float3 e3 = v2 - v1;
		c = cross(e3, vp1);
		if (dot(N, c) < 0) return false;
		float3 e4 = v0 - v2;
		float3 vp2 = p - v2;
		c = cross(e4, vp2);
		if (dot(N, c) < 0) return false;
		if (t < ray.t && t > t_min) { tOut = t; return true; }
		return false; // IsOccluding falls off the end here in the reference (quirk Q9); defined as false
	}
	void Intersect(Ray& ray, float t_min) const
	{
		float t;
		if (HitT(ray, t_min, t)) ray.t = t, ray.objIdx = objIdx, ray.mat = mat, ray.hitNormal = N;
	}
	bool IsOccluding(const Ray& ray, float t_min) const { float t; return HitT(ray, t_min, t); } // :216-237
};

// template/scene.h:347-394
struct Sphere {
	float3 pos = float3(0);
	float r2 = 0, invr = 0, r = 0;
	int objIdx = -1;
	int mat = -1;
	Sphere() = default;
	Sphere(int idx, int m, const float3& p, float rad) : pos(p), r2(rad * rad), invr(1 / rad), r(rad), objIdx(idx), mat(m) {}
	void Intersect(Ray& ray, float t_min) const // :351-371
	{
		float3 oc = ray.O - pos;
		float b = dot(oc, ray.D);
		float c = dot(oc, oc) - r2;
		float t, d = b * b - c;
		if (d <= 0) return;
		d = sqrtf(d), t = -b - d;
		if (t < ray.t && t > t_min) {
			ray.t = t, ray.objIdx = objIdx, ray.mat = mat;
			ray.hitNormal = (ray.IntersectionPoint() - pos) * invr;
			return;
		}
		t = d - b;
		if (t < ray.t && t > t_min) {
			ray.t = t, ray.objIdx = objIdx, ray.mat = mat;
			ray.hitNormal = (ray.IntersectionPoint() - pos) * invr;
			return;
		}
	}
	bool IsOccluding(const Ray& ray, float t_min) const // :372-381
	{
		float3 oc = ray.O - pos;
		float b = dot(oc, ray.D);
		float c = dot(oc, oc) - r2;
		float t, d = b * b - c;
		if (d <= 0) return false;
		d = sqrtf(d), t = -b - d;
		float t2 = d - b;
		return ((t < ray.t && t > t_min) || (t2 < ray.t && t2 > t_min));
	}
};

// template/scene.h:401-448
struct Plane {
	float3 N = float3(0, 1, 0);
	float d = 0;
	int objIdx = -1;
	int mat = -1;
	Plane() = default;
	Plane(int idx, int m, const float3& normal, float dist) : N(normal), d(dist), objIdx(idx), mat(m) {}
	void Intersect(Ray& ray, float t_min) const // :405-410
	{
		float t = -(dot(ray.O, N) + d) / (dot(ray.D, N));
		if (t < ray.t && t > t_min) ray.t = t, ray.objIdx = objIdx, ray.mat = mat, ray.hitNormal = N;
	}
	bool IsOccluding(const Ray& ray, float t_min) const // :411-415
	{
		float t = -(dot(ray.O, N) + d) / (dot(ray.D, N));
		return (t < ray.t && t > t_min);
	}
};

// template/scene.h:258-340
struct Mesh {
	std::vector<Triangle> tri;
	std::vector<Triangle> orig; // originalVerts of the reference (template/scene.h:337), kept per triangle
	int groupIdx = -1;
	int mat = -1;
	Mesh() = default;
	// in-memory form of either loader: n triangles, 9 floats each, ids 1000*group + i
	Mesh(int idGroup, int m, const float* v9, int n) : groupIdx(idGroup), mat(m)
	{
		for (int i = 0; i < n; i++) {
			const float* p = v9 + 9 * i;
			tri.push_back(Triangle(1000 * idGroup + i, m, float3(p[0], p[1], p[2]), float3(p[3], p[4], p[5]), float3(p[6], p[7], p[8])));
		}
	}
	// .tri loader (:261-284): the loop tests fscanf's result only after using the values, so the
	// last record is pushed twice (quirk Q10).
	static bool LoadTri(Mesh& out, int idGroup, const char* path, int m)
	{
		FILE* file = fopen(path, "r");
		if (!file) return false; // the reference dereferences NULL here
		out = Mesh();
		out.groupIdx = idGroup, out.mat = m;
		float a = 0, c = 0, d = 0, e = 0, f = 0, g = 0, h = 0, i = 0, j = 0;
		int res = 1;
		int count = 0;
		while (res > 0) {
			res = fscanf(file, "%f %f %f %f %f %f %f %f %f\n", &a, &c, &d, &e, &f, &g, &h, &i, &j);
			out.tri.push_back(Triangle(1000 * idGroup + count, m, float3(a, c, d), float3(e, f, g), float3(h, i, j)));
			count++;
		}
		fclose(file);
		return true;
	}
	// .obj loader (:285-313): 'v x y z' and 'f a//n b//n c//n' only; vertices scaled then offset
	static bool LoadObj(Mesh& out, int idGroup, const char* path, int m, const float3& pos, float scale)
	{
		std::ifstream file(path, std::ios::in);
		if (!file) return false; // the reference calls exit(1)
		out = Mesh();
		out.groupIdx = idGroup, out.mat = m;
		std::vector<float3> vertices;
		std::vector<int> faces;
		std::string line;
		float x, y, z;
		while (std::getline(file, line)) {
			if (line.substr(0, 2) == "v ") {
				std::istringstream v(line.substr(2));
				v >> x; v >> y; v >> z;
				vertices.push_back(float3(x * scale + pos.x, y * scale + pos.y, z * scale + pos.z));
			} else if (line.substr(0, 2) == "f ") {
				int v0 = 0, v1 = 0, v2 = 0, temp;
				sscanf(line.c_str(), "f %i//%i %i//%i %i//%i", &v0, &temp, &v1, &temp, &v2, &temp);
				faces.push_back(v0), faces.push_back(v1), faces.push_back(v2);
			}
		}
		for (size_t i = 0; i < faces.size() / 3; i++) {
			int a = faces[3 * i] - 1, b = faces[3 * i + 1] - 1, c = faces[3 * i + 2] - 1;
			if (a < 0 || b < 0 || c < 0 || a >= (int)vertices.size() || b >= (int)vertices.size() || c >= (int)vertices.size()) return false;
			out.tri.push_back(Triangle(1000 * idGroup + (int)i, m, vertices[a], vertices[b], vertices[c]));
		}
		return true;
	}
};

struct bvh;
struct tlas;
struct bvhInstance;

// template/scene.h:685-1397: the container and its three queries.  Scene factories are data
// supplied by the caller (tests build them through oracle_capi.cpp).
struct Scene {
	std::vector<Material> materials;
	std::vector<Light> lights;
	std::vector<Sphere> spheres;
	std::vector<Plane> planes;
	std::vector<Mesh> meshes;
	std::vector<unsigned char> skydome;
	int skydomeX = 0, skydomeY = 0, skydomeN = 3;
	bool raytracer = true; // :1384
	bool useTLAS = false;  // :1388
	bvh* b = nullptr;
	tlas* tl = nullptr;
	std::vector<bvh*> blasList;         // TLAS mode: one BLAS per referenced mesh
	std::vector<bvhInstance*> instances;

	uint getTriangleNb() const { uint acc = 0; for (auto& m : meshes) acc += (uint)m.tri.size(); return acc; } // :1329-1335
	const Triangle& getTriangle(uint idx) const // :1337-1344
	{
		size_t i = 0;
		while (idx >= meshes[i].tri.size()) { idx -= (uint)meshes[i].tri.size(); i++; }
		return meshes[i].tri[idx];
	}
	// :1352-1357: flips the scene flag and the lights' flags; materials keep theirs
	void toogleRaytracer() { raytracer = !raytracer; for (auto& l : lights) l.raytracer = raytracer; }

	void SetTime(float t); // :1210-1246 (mesh wobble + bvh::Refit; the reference gates it with animOn)
	void FindNearest(Ray& ray, float t_min, Counters& cnt) const; // :1248-1267
	bool IsOccluded(Ray& ray, Counters& cnt) const;               // :1286-1291

	// :1312-1327.  With no sky texture loaded the reference would read through a null pointer;
	// defined here as black.
	float3 GetSkyColor(const Ray& r) const
	{
		if (skydome.empty()) return float3(0);
		float3 horizontalProj = float3(r.D.x, 0, r.D.z);
		float cHeight = dot(r.D, float3(0, -1, 0));
		float cOrient = dot(float3(0, 0, 1), normalize(horizontalProj));
		float sOrient = dot(float3(1, 0, 0), normalize(horizontalProj));
		sOrient = sOrient > 0 ? 1 : -1;
		int y = f2i(((cHeight + 1) / 2) * (skydomeY - 1));
		int x = f2i((((sOrient * x_acosf(cOrient)) + PI) / TWOPI) * (skydomeX - 1));
		// the reference clamps to skydomeX / skydomeY (one past the end, quirk Q19); indices that
		// large cannot be produced by the expressions above, so the clamp is kept in-bounds here
		if (x >= skydomeX) x = skydomeX - 1;
		if (y >= skydomeY) y = skydomeY - 1;
		if (y < 0) y = 0;
		if (x < 0) x = 0;
		const unsigned char* p = skydome.data() + (size_t)(x + skydomeX * y) * skydomeN;
		return float3((float)p[0], (float)p[1], (float)p[2]) / 255;
	}
};

} // namespace orc
