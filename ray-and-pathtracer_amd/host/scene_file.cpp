// Scene description files ("rapt-scene 1"): what the reference expresses as edited C++ scene
// factories (template/scene.h:791-1209: materials, lights, spheres, planes, meshes, instance
// transforms, which acceleration structure to build) as data, one record per line:
//
//   rapt-scene 1
//   sky hdr <file.hdr> | sky raw <file.bin> <w> <h> <n>
//   material diffuse <albedo.xyz> <col.xyz> <ks> <kd> <n> <emission> <shinieness> <rt>
//   material metal <fuzzy> <col.xyz> <rt>
//   material glass <ir> <col.xyz> <absorption.xyz> <rt>
//   light area <idx> <pos.xyz> <strength> <col.xyz> <radius> <normal.xyz>
//   light dir <idx> <pos.xyz> <strength> <col.xyz> <normal.xyz> <r>
//   sphere <idx> <material#> <pos.xyz> <r>
//   plane <idx> <material#> <N.xyz> <d>
//   mesh obj <group> <material#> <pos.xyz> <scale> <file.obj>
//   mesh tri <group> <material#> <file.tri>
//   mesh raw <group> <material#> <file.bin>          (float32, 9 per triangle)
//   instance <mesh#> <16 floats, row-major transform>
//   build bvh <split> | build tlas <split>            (split: 0 binned SAH, 1 median, 2 longest axis, 3 SAH)
//
// Arguments keep the constructor orders of the reference classes.  Relative file names resolve against
// the scene file's directory.  Numbers are read with strtof, so "%.9g" text round-trips float32.
#include "rapt.h"
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace rapt {

namespace {
struct Line {
	std::vector<std::string> tok;
	size_t at = 0;
	int number = 0;
	std::string file;
	[[noreturn]] void fail(const std::string& msg) const { throw std::runtime_error(file + ":" + std::to_string(number) + ": " + msg); }
	const std::string& word()
	{
		if (at >= tok.size()) fail("missing argument");
		return tok[at++];
	}
	float f()
	{
		const std::string& w = word();
		char* end = nullptr;
		float v = strtof(w.c_str(), &end);
		if (end == w.c_str() || *end) fail("not a number: " + w);
		return v;
	}
	int i()
	{
		const std::string& w = word();
		char* end = nullptr;
		long v = strtol(w.c_str(), &end, 10);
		if (end == w.c_str() || *end) fail("not an integer: " + w);
		return (int)v;
	}
	float3 v3() { float a = f(), b = f(), c = f(); return float3(a, b, c); }
	void done() const { if (at != tok.size()) fail("unexpected extra argument: " + tok[at]); }
};
std::vector<unsigned char> slurp(const std::string& path)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("cannot open " + path);
	return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
} // namespace

void Scene::LoadFile(const std::string& path)
{
	std::ifstream in(path);
	if (!in) throw std::runtime_error("cannot open " + path);
	const size_t slash = path.find_last_of('/');
	const std::string dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
	auto resolve = [&](const std::string& p) { return (!p.empty() && p[0] == '/') ? p : dir + p; };
	std::vector<int> instMesh;
	std::vector<mat4> instT;
	bool header = false, built = false;
	std::string text;
	int number = 0;
	while (std::getline(in, text)) {
		number++;
		Line L;
		L.file = path, L.number = number;
		std::istringstream ss(text);
		for (std::string w; ss >> w;) L.tok.push_back(w);
		if (L.tok.empty() || L.tok[0][0] == '#') continue;
		const std::string kind = L.word();
		if (!header) {
			if (kind != "rapt-scene" || L.i() != 1) L.fail("expected 'rapt-scene 1'");
			header = true;
			continue;
		}
		if (built) L.fail("records after 'build'");
		auto mat = [&]() -> material* { int m = L.i(); if (m < 0 || m >= (int)materials.size()) L.fail("material index out of range"); return materials[m]; };
		if (kind == "sky") {
			const std::string how = L.word();
			if (how == "hdr") {
				std::string why;
				if (!LoadSkyHDR(resolve(L.word()).c_str(), &why)) L.fail("sky: " + why);
			} else if (how == "raw") {
				const std::string file = resolve(L.word());
				int w = L.i(), h = L.i(), n = L.i();
				std::vector<unsigned char> px = slurp(file);
				if (w <= 0 || h <= 0 || n < 3 || px.size() != (size_t)w * h * n) L.fail("sky raw: size mismatch");
				skydome = px, skydomeX = w, skydomeY = h, skydomeN = n;
			} else L.fail("sky hdr|raw expected");
		} else if (kind == "material") {
			const std::string type = L.word();
			if (type == "diffuse") { float3 a = L.v3(), c = L.v3(); float ks = L.f(), kd = L.f(); int n = L.i(); float e = L.f(), s = L.f(); int rt = L.i(); materials.push_back(new diffuse(a, c, ks, kd, n, rt != 0, e, s)); }
			else if (type == "metal") { float fz = L.f(); float3 c = L.v3(); int rt = L.i(); materials.push_back(new metal(fz, c, rt != 0)); }
			else if (type == "glass") { float ir = L.f(); float3 c = L.v3(), a = L.v3(); int rt = L.i(); materials.push_back(new glass(ir, c, a, 0.0f, 0, rt != 0)); }
			else L.fail("material diffuse|metal|glass expected");
		} else if (kind == "light") {
			const std::string type = L.word();
			if (type == "area") { int idx = L.i(); float3 p = L.v3(); float str = L.f(); float3 c = L.v3(); float r = L.f(); float3 n = L.v3(); lights.push_back(new AreaLight(idx, p, str, c, r, n, 4, raytracer)); }
			else if (type == "dir") { int idx = L.i(); float3 p = L.v3(); float str = L.f(); float3 c = L.v3(), n = L.v3(); float r = L.f(); lights.push_back(new DirectionalLight(idx, p, str, c, n, r, raytracer)); }
			else L.fail("light area|dir expected");
		} else if (kind == "sphere") {
			int idx = L.i(); material* m = mat(); float3 p = L.v3(); float r = L.f();
			spheres.push_back(Sphere(idx, m, p, r));
		} else if (kind == "plane") {
			int idx = L.i(); material* m = mat(); float3 n = L.v3(); float d = L.f();
			planes.push_back(Plane(idx, m, n, d));
		} else if (kind == "mesh") {
			const std::string type = L.word();
			int group = L.i();
			material* m = mat();
			if (type == "obj") { float3 p = L.v3(); float sc = L.f(); meshes.push_back(Mesh(group, resolve(L.word()), m, p, sc)); }
			else if (type == "tri") meshes.push_back(Mesh(group, resolve(L.word()).c_str(), m));
			else if (type == "raw") {
				std::vector<unsigned char> raw = slurp(resolve(L.word()));
				if (raw.size() % 36) L.fail("mesh raw: size is not a multiple of 9 floats");
				meshes.push_back(Mesh(group, m, reinterpret_cast<const float*>(raw.data()), (int)(raw.size() / 36)));
			} else L.fail("mesh obj|tri|raw expected");
		} else if (kind == "instance") {
			int mi = L.i();
			if (mi < 0 || mi >= (int)meshes.size()) L.fail("mesh index out of range");
			mat4 T;
			for (int k = 0; k < 16; k++) T.cell[k] = L.f();
			instMesh.push_back(mi), instT.push_back(T);
		} else if (kind == "build") {
			const std::string what = L.word();
			int split = L.i();
			if (split < 0 || split > 3) L.fail("split method 0..3 expected");
			if (what == "bvh") BuildBVH(split);
			else if (what == "tlas") { if (instMesh.empty()) L.fail("build tlas without instances"); BuildTLAS(instMesh, instT, split); }
			else L.fail("build bvh|tlas expected");
			built = true;
		} else L.fail("unknown record '" + kind + "'");
		L.done();
	}
	if (!header) throw std::runtime_error(path + ": empty scene file");
	if (!built) throw std::runtime_error(path + ": no 'build' record");
}

} // namespace rapt
