// ORACLE (test infrastructure, NOT product code) -- parity unpinned, see oracle/README.md.
//
// CPU restatement of Camera::GetPrimaryRay (camera.h:24-41) and of the integrators and pixel
// loop of renderer.cpp: Renderer::Trace (:21-126), Renderer::Sample (:128-236) and the body of
// Renderer::Tick (:256-291), recursion kept as recursion.  The only intended difference from
// the reference is the random stream: one xorshift32 state per (pixel, frame), threaded through
// every call that draws, instead of one process-global state (see orc_math.h).
#pragma once
#include "orc_bvh.h"
#include "orc_qlearn.h"

namespace orc {

// camera.h:10-22, 42-52; SCRWIDTH/SCRHEIGHT (camera.h:4-5) are run-time values here
struct Camera {
	int width = 600, height = 400;
	float aspect = 1.5f;
	float viewAngle = 0.25f;
	float3 camPos, topLeft, topRight, bottomLeft;
	bool fishEye = false;
	float yAngle = 0;
	Camera() { Reset(600, 400); }
	void Reset(int w, int h)
	{
		width = w, height = h;
		aspect = (float)w / (float)h;
		camPos = float3(0, 1, -2);
		topLeft = float3(-aspect, 2, 0);
		topRight = float3(aspect, 2, 0);
		bottomLeft = float3(-aspect, 0, 0);
	}
	// camera.h:141-154 (RotateY) and :157-174 (RotateX), used by the fisheye branch only
	float3 RotateY(const float3& p, const float3& center, float theta) const
	{
		double c = cos((double)theta), s = sin((double)theta);
		float3 res(0.f);
		float3 vect = p - center;
		float3 xTransform((float)c, 0, (float)-s);
		float3 zTransform((float)s, 0, (float)c);
		res.x = dot(vect, xTransform);
		res.y = vect.y;
		res.z = dot(vect, zTransform);
		return res + center;
	}
	float3 RotateX(const float3& p, const float3& center, float theta) const
	{
		double c = cos((double)theta), s = sin((double)theta);
		float3 res(0.f);
		float3 vect = p - center;
		vect = RotateY(vect, float3(0.f), -yAngle);
		float3 zTransform(0, (float)-s, (float)c);
		float3 yTransform(0, (float)c, (float)s);
		res.x = vect.x;
		res.y = dot(vect, yTransform);
		res.z = dot(vect, zTransform);
		res = RotateY(res, float3(0.f), yAngle);
		return res + center;
	}
	Ray GetPrimaryRay(const int x, const int y) const // camera.h:24-41
	{
		if (fishEye) {
			float3 screenCenter = topLeft + .5f * (topRight - topLeft) + .5f * (bottomLeft - topLeft);
			const float u = (float)(x - width / 2) * (aspect * viewAngle / width);
			const float v = (float)(y - width / 2) * (viewAngle / height);
			float3 newRay = RotateX(RotateY(normalize(screenCenter - camPos), camPos, -u), camPos, -v);
			return Ray(camPos, normalize(newRay));
		}
		const float u = (float)x * (1.0f / width);
		const float v = (float)y * (1.0f / height);
		const float3 P = topLeft + u * (topRight - topLeft) + v * (bottomLeft - topLeft);
		return Ray(camPos, normalize(P - camPos));
	}
};

struct Renderer {
	Scene* scene = nullptr;
	Camera camera;
	std::vector<float4> accumulator; // renderer.cpp:8
	int iterationNumber = 1;          // Scene::iterationNumber (template/scene.h:1381)
	int max_depth_trace = 4;          // renderer.cpp:269; tests may lower it ("primary rays only")
	mutable QLearn ql;                // the Q-learning guided sampler (orc_qlearn.h; no reference code: parity unpinned), off by default

	void Init(int w, int h) // :5-11
	{
		camera.Reset(w, h);
		accumulator.assign((size_t)w * h, float4{ 0, 0, 0, 0 });
		iterationNumber = 1;
	}

	bool isLight(int objIdx) const { return objIdx >= 11 && objIdx < 11 + (int)scene->lights.size(); } // :27, :135

	// Renderer::Trace (:21-126)
	float3 Trace(Ray& ray, int depth, float3 energy, uint& seed, Counters& cnt) const
	{
		const Scene& sc = *scene;
		if (depth <= 0) return float3(0, 0, 0);
		float t_min = 1e-6;
		sc.FindNearest(ray, t_min, cnt);
		if (ray.objIdx == -1) return sc.GetSkyColor(ray);
		if (isLight(ray.objIdx))
			return sc.lights[ray.objIdx - 11].GetLightIntensityAt(ray.IntersectionPoint(), ray.hitNormal, ray.IntersectionPoint());
		float3 totCol = float3(0);
		const Material& m = sc.materials[ray.mat];
		float3 f = m.col;
		if (!sc.raytracer) { // :33-43, unreachable from Tick (which calls Trace only when raytracer is set)
			double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z;
			if (depth < 5 || !p) {
				if ((double)RandomFloat(seed) < p) f = f * (float)(1 / p);
				else return totCol;
			}
		}
		switch (m.type) {
		case GLASS: { // :45-80
			float3 refractionColor = float3(0);
			float kr;
			glass_fresnel(normalize(ray.D), normalize(ray.hitNormal), m.ir, kr);
			bool outside = dot(ray.D, ray.hitNormal) < 0;
			float3 bias = 0.0001f * ray.hitNormal;
			float3 norm = outside ? ray.hitNormal : -ray.hitNormal;
			float r = !outside ? m.ir : (1 / m.ir);
			if (outside) {
				energy.x *= x_expf(m.absorption.x * -ray.t);
				energy.y *= x_expf(m.absorption.y * -ray.t);
				energy.z *= x_expf(m.absorption.z * -ray.t);
			}
			if (kr < 1) {
				float3 refractionDirection = normalize(glass_refract(ray.D, norm, r));
				float3 refractionRayOrig = outside ? ray.IntersectionPoint() - bias : ray.IntersectionPoint() + bias;
				Ray refrRay(refractionRayOrig, refractionDirection);
				float3 tempCol = m.col * energy;
				refractionColor = tempCol * Trace(refrRay, depth - 1, energy, seed, cnt);
			}
			float3 reflectionDirection = normalize(reflect(ray.D, norm));
			float3 reflectionRayOrig = outside ? ray.IntersectionPoint() + bias : ray.IntersectionPoint() - bias;
			Ray reflRay(reflectionRayOrig, reflectionDirection);
			float3 reflectionColor = m.col * Trace(reflRay, depth - 1, energy, seed, cnt);
			totCol += reflectionColor * kr + refractionColor * (1 - kr);
			break;
		}
		case METAL: { // :81-86
			Ray reflected = metal_scatter(ray, ray.hitNormal);
			totCol += m.col * Trace(reflected, depth - 1, energy, seed, cnt) * energy;
			break;
		}
		case DIFFUSE: { // :87-122
			float3 scatteredDir(0);
			for (size_t i = 0; i < sc.lights.size(); i++) {
				float3 attenuation;
				float3 pickedPos = sc.lights[i].GetLightPosition(seed);
				float3 lightRayDirection = pickedPos - ray.IntersectionPoint();
				float len2 = dot(lightRayDirection, lightRayDirection);
				lightRayDirection = normalize(lightRayDirection);
				Ray r(ray.IntersectionPoint() + lightRayDirection * 1e-4f, lightRayDirection, sqrtf(len2));
				diffuse_scatter(m, ray, attenuation, scatteredDir, lightRayDirection,
				                sc.lights[i].GetLightIntensityAt(ray.IntersectionPoint(), ray.hitNormal, pickedPos), ray.hitNormal, energy, seed);
				if (sc.IsOccluded(r, cnt)) continue;
				if (m.shinieness != 0) {
					Ray refl(ray.IntersectionPoint(), reflect(ray.D, ray.hitNormal));
					totCol += m.shinieness * m.col * Trace(refl, depth - 1, energy, seed, cnt) * energy;
				}
				totCol += (1 - m.shinieness) * m.col * attenuation * energy;
			}
			if (!sc.raytracer) { // :107-121, unreachable from Tick
				float3 indirectLightning = float3(0);
				int N = 1;
				for (int i = 0; i < N; i++) {
					float3 cos_i = float3(dot(scatteredDir, float3((float)N)));
					Ray scattered(ray.IntersectionPoint(), scatteredDir);
					indirectLightning += cos_i * Trace(scattered, depth - 1, energy, seed, cnt) * 2 * PI;
				}
				indirectLightning /= (float)N;
				totCol = totCol * INVPI;
				totCol += indirectLightning;
			}
			break;
		}
		}
		return totCol;
	}

	// Renderer::Sample (:128-236)
	// Q-learning only: prevKey = 1 + cell * 64 + patch of the scattering that sent this ray, 0 for none; learner = this sample pays rewards
	float3 Sample(Ray& ray, int depth, float3 energy, uint& seed, Counters& cnt, uint prevKey = 0, bool learner = false) const
	{
		const Scene& sc = *scene;
		if (depth < 0) return float3(0.05f);
		float3 totCol = float3(0);
		float t_min = 0.001f;
		sc.FindNearest(ray, t_min, cnt);
		if (ray.objIdx == -1) {
			const float3 sky = sc.GetSkyColor(ray);
			if (ql.on && prevKey) ql.reward(prevKey, QLearn::lum(sky));
			return sky;
		}
		if (isLight(ray.objIdx)) {
			const float3 li = sc.lights[ray.objIdx - 11].GetLightIntensityAt(ray.IntersectionPoint(), ray.hitNormal, ray.IntersectionPoint());
			if (ql.on && prevKey) ql.reward(prevKey, QLearn::lum(li));
			return li;
		}
		float3 intersectionPoint = ray.IntersectionPoint();
		float3 normal = ray.hitNormal;
		const Material& m = sc.materials[ray.mat];
		float3 f = m.col;
		if (ql.on && prevKey) {
			const bool dif = m.type == DIFFUSE;
			ql.reward(prevKey, ql.expected(ql.cell(intersectionPoint), normal, dif ? QLearn::lum(m.col * m.albedo) : QLearn::lum(m.col), dif));
		}
		if (sc.raytracer) { // :143-153, unreachable from Tick (which calls Sample only when raytracer is clear)
			double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z;
			if (depth < 5 || !p) {
				if ((double)RandomFloat(seed) < p) f = f * (float)(1 / p);
				else return totCol;
			}
		}
		switch (m.type) {
		case DIFFUSE: { // :156-191
			float3 directLightning = float3(0);
			for (size_t i = 0; i < sc.lights.size(); i++) {
				float3 pickedPos = sc.lights[i].GetLightPosition(seed);
				float3 lightRayDirection = pickedPos - ray.IntersectionPoint();
				float len2 = dot(lightRayDirection, lightRayDirection);
				lightRayDirection = normalize(lightRayDirection);
				Ray r(ray.IntersectionPoint() + lightRayDirection * 1e-4f, lightRayDirection, sqrtf(len2));
				if (sc.IsOccluded(r, cnt)) continue;
				float3 scatteredDir, attenuation;
				diffuse_scatter(m, ray, attenuation, scatteredDir, lightRayDirection,
				                sc.lights[i].GetLightIntensityAt(ray.IntersectionPoint(), normal, pickedPos), normal, energy, seed);
				if (m.shinieness != 0) {
					Ray refl(ray.IntersectionPoint(), reflect(ray.D, ray.hitNormal));
					directLightning += m.shinieness * m.col * Sample(refl, depth - 1, energy, seed, cnt);
				}
				directLightning += (1 - m.shinieness) * m.col * attenuation * energy;
			}
			float3 indirectLightning = float3(0);
			if (ql.on) {
				// guided: the direction from the Q table of this cell; 1 / (16 P) = 1 / (pi pdf) stands where the uniform hemisphere has 2
				float P;
				int patch;
				const int cell = ql.cell(intersectionPoint);
				const float3 d = ql.sample(cell, seed, P, patch);
				const float c = dot(d, normal);
				float fq = 0;
				if (c > 0) {
					float3 cos_i = float3(c);
					Ray next(intersectionPoint, d);
					indirectLightning += m.col * cos_i * Sample(next, depth - 1, energy, seed, cnt, learner ? 1u + (uint)cell * 64 + (uint)patch : 0u, learner);
					fq = 1.0f / (16.0f * P);
				} else if (learner) ql.reward(1u + (uint)cell * 64 + (uint)patch, 0.0f); // below the surface: nothing to trace, the patch learns 0
				totCol = (directLightning * INVPI + fq * indirectLightning) * m.albedo;
				break;
			}
			int N = 1;
			for (int i = 0; i < N; ++i) {
				float3 rayToHemi = RandomInHemisphere(seed, normal);
				float3 cos_i = float3(dot(rayToHemi, normal));
				Ray next(intersectionPoint, rayToHemi);
				indirectLightning += m.col * cos_i * Sample(next, depth - 1, energy, seed, cnt);
			}
			indirectLightning /= (float)N;
			totCol = (directLightning * INVPI + 2 * indirectLightning) * m.albedo;
			break;
		}
		case METAL: { // :192-197
			Ray reflected = metal_scatter(ray, normal);
			totCol += m.col * Sample(reflected, depth - 1, energy, seed, cnt, 0, learner);
			break;
		}
		case GLASS: { // :198-233
			float3 refractionColor = float3(0);
			float kr;
			glass_fresnel(normalize(ray.D), normalize(ray.hitNormal), m.ir, kr);
			bool outside = dot(ray.D, ray.hitNormal) < 0;
			float3 bias = 0.0001f * ray.hitNormal;
			float3 norm = outside ? ray.hitNormal : -ray.hitNormal;
			float r = !outside ? m.ir : (1 / m.ir);
			if (outside) {
				energy.x *= x_expf(m.absorption.x * -ray.t);
				energy.y *= x_expf(m.absorption.y * -ray.t);
				energy.z *= x_expf(m.absorption.z * -ray.t);
			}
			float odds = kr;
			if (odds < RandomFloat(seed)) {
				float3 refractionDirection = normalize(glass_refract(ray.D, norm, r));
				float3 refractionRayOrig = outside ? ray.IntersectionPoint() - bias : ray.IntersectionPoint() + bias;
				Ray refrRay(refractionRayOrig, refractionDirection);
				float3 tempCol = m.col * energy;
				refractionColor = tempCol * Sample(refrRay, depth - 1, energy, seed, cnt, 0, learner);
				totCol += refractionColor * (1 - kr);
			} else {
				float3 reflectionDirection = normalize(reflect(ray.D, norm));
				float3 reflectionRayOrig = outside ? ray.IntersectionPoint() + bias : ray.IntersectionPoint() - bias;
				Ray reflRay(reflectionRayOrig, reflectionDirection);
				float3 reflectionColor = m.col * Sample(reflRay, depth - 1, energy, seed, cnt, 0, learner);
				totCol += reflectionColor * kr;
			}
			break;
		}
		}
		return totCol;
	}

	// One pixel of the Tick loop (:263-285) for frame 'frame'.  aaSamples is 1 and invAaSamples
	// is the int 1 (template/scene.h:1379-1380).
	void Pixel(int x, int y, uint frame, uint seedBase, Counters& cnt)
	{
		const int W = camera.width, H = camera.height;
		const size_t idx = (size_t)x + (size_t)y * W;
		uint seed = StreamSeed(seedBase + (uint)idx + frame * (uint)(W * H));
		float4& acc = accumulator[idx];
		if (scene->raytracer) {
			Ray pr = camera.GetPrimaryRay(x, y);
			float3 totCol = float3(0);
			totCol += Trace(pr, max_depth_trace, float3(1), seed, cnt);
			float3 v = totCol / (float)1;
			acc = float4{ v.x, v.y, v.z, 0 }; // :270, float4(float3) sets w = 0 (template.cpp:785-789)
		} else {
			float newX = x + (RandomFloat(seed) * 2 - 1);
			float newY = y + (RandomFloat(seed) * 2 - 1);
			Ray pr = camera.GetPrimaryRay((int)newX, (int)newY); // Q11: jitter truncated by the int parameters
			float3 totCol = float3(0);
			totCol += Sample(pr, 4, float3(1), seed, cnt, 0, ql.on && (seed & ql.learnMask) == 0);
			float r = x_powf(totCol.x * 1, GAMMA); // Q12: gamma per sample, before accumulation
			float g = x_powf(totCol.y * 1, GAMMA);
			float b = x_powf(totCol.z * 1, GAMMA);
			acc.x += r, acc.y += g, acc.z += b, acc.w += 0;
		}
	}

	// Renderer::Tick (:240-305) without animation, input and the performance printf.  'it' is read BEFORE a
	// camera change resets the iteration number (:247 vs :253-255), so the frame on which the camera changed is
	// resolved with the stale count, and the count is not advanced on that frame (:293-294).
	void Tick(bool& cameraChanged, uint frame, uint seedBase, uint* pixels)
	{
		const int W = camera.width, H = camera.height;
		int it = iterationNumber;
		if (cameraChanged && !scene->raytracer) iterationNumber = 1;
		const bool changed = cameraChanged;
#pragma omp parallel for schedule(dynamic)
		for (int y = 0; y < H; ++y) {
			Counters local;
			for (int x = 0; x < W; ++x) {
				if (!scene->raytracer && changed) accumulator[(size_t)x + (size_t)y * W] = float4{ 0, 0, 0, 0 }; // :273-275
				Pixel(x, y, frame, seedBase, local);
			}
			for (int x = 0; x < W; ++x) pixels[(size_t)y * W + x] = ResolvePixel((size_t)x + (size_t)y * W, it);
		}
		if (!scene->raytracer && !changed) iterationNumber = it + 1;
		cameraChanged = false;
	}

	// RGBF32_to_RGB8 of accumulator / it (:287-290, template/precomp.h:445-448, non-MSVC branch)
	uint ResolvePixel(size_t idx, int it) const
	{
		const float4& a = accumulator[idx];
		float fx = a.x / it, fy = a.y / it, fz = a.z / it;
		auto conv = [](float v) -> uint {
			float m = std_min(1.0f, v);
			float s = 255.0f * m;
			// (uint) of a negative or NaN float is undefined in C++; x86-64 compilers emit cvttss2si
			// to a 64-bit register and keep the low 32 bits
			long long q = (s > -9.2e18f && s < 9.2e18f) ? (long long)s : (long long)0x8000000000000000ull;
			return (uint)q;
		};
		uint r = conv(fx), g = conv(fy), b = conv(fz);
		return (r << 16) + (g << 8) + b;
	}
};

} // namespace orc
