// Device-side arithmetic of the trace loop for gfx950.  Compiled with -ffp-contract=off and
// without fast-math: every +, *, / and sqrtf rounds exactly once (v_div_* / v_sqrt sequences are the
// IEEE forms, hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), and NaN / inf propagate as
// IEEE prescribes, so a lane reproduces the reference's scalar arithmetic bit for bit.  Operand
// orders follow the cited reference lines (file:line in Fannollost/Ray-and-pathtracer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

typedef unsigned int uint;

struct f3 {
	float x, y, z;
	__device__ __forceinline__ f3() {}
	__device__ __forceinline__ f3(float a, float b, float c) : x(a), y(b), z(c) {}
	__device__ __forceinline__ explicit f3(float s) : x(s), y(s), z(s) {}
};
__device__ __forceinline__ f3 xyz(const float4& v) { return f3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 mk4(const f3& v, float w) { return make_float4(v.x, v.y, v.z, w); }

// template/precomp.h:541-758
__device__ __forceinline__ f3 operator-(const f3& a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator+(const f3& a, const f3& b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(const f3& a, const f3& b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(const f3& a, const f3& b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(const f3& a, float b) { return f3(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ f3 operator*(float b, const f3& a) { return f3(b * a.x, b * a.y, b * a.z); }
__device__ __forceinline__ f3 operator/(const f3& a, float b) { return f3(a.x / b, a.y / b, a.z / b); }
// template/precomp.h:813, 827, 835, 861, 863
__device__ __forceinline__ float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length(const f3& v) { return sqrtf(dot(v, v)); }
__device__ __forceinline__ f3 normalize(const f3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
__device__ __forceinline__ f3 reflect(const f3& i, const f3& n) { return i - 2.0f * n * dot(n, i); }
__device__ __forceinline__ f3 cross(const f3& a, const f3& b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// std::min / std::max (bvh.cpp:822-826 via 'using namespace std'): NaN-asymmetric ternaries, which
// v_min_f32 / v_max_f32 are not
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
// Tmpl8::fminf / fmaxf (template/precomp.h:479-480)
__device__ __forceinline__ float t_fminf(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float t_fmaxf(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float t_clamp(float f, float a, float b) { return t_fmaxf(a, t_fminf(f, b)); }
// libm fmax reached through std::fmax in diffuse::scatter (template/scene.h:608)
__device__ __forceinline__ float libm_fmaxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }

// float -> int with the x86 cvttss2si result for NaN / out-of-range (0x80000000); the reference
// reaches those inputs in Scene::GetSkyColor (template/scene.h:1319-1320)
__device__ __forceinline__ int f2i(float f)
{
	if (!(f > -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000;
	return (int)f;
}

// Transcendentals: evaluated in f64, rounded once to f32 (DESIGN.md "Transcendentals"); the same
// definition is used by every CPU implementation this path is compared with.
// They are real functions, not inlined: the double-precision library bodies pushed k_shade to 128 VGPRs (its bound; four
// waves per SIMD) and k_light to 181 (two); called, the kernels need 89 and 114 (five and four waves) and the shading
// side of a bench step takes 15.6 instead of 16.5 ms.
#define RT_MATH_FN __device__ __noinline__
RT_MATH_FN float x_cosf(float x) { return (float)cos((double)x); }
RT_MATH_FN float x_sinf(float x) { return (float)sin((double)x); }
RT_MATH_FN float x_acosf(float x) { return (float)acos((double)x); }
RT_MATH_FN float x_asinf(float x) { return (float)asin((double)x); }
RT_MATH_FN float x_expf(float x) { return (float)exp((double)x); }
RT_MATH_FN float x_powf(float a, float b) { return (float)pow((double)a, (double)b); }

#define RT_PI 3.14159265358979323846264f
#define RT_INVPI 0.31830988618379067153777f
#define RT_GAMMA 0.57142857142857142857143f
#define RT_TWOPI 6.28318530717958647692528f

// RNG (template/template.cpp:672-724), one stream per (pixel, frame)
__device__ __forceinline__ uint WangHash(uint s) { s = (s ^ 61) ^ (s >> 16); s *= 9; s = s ^ (s >> 4); s *= 0x27d4eb2d; s = s ^ (s >> 15); return s; }
__device__ __forceinline__ uint InitSeed(uint seedBase) { return WangHash((seedBase + 1) * 17); }
// start of the stream of one (pixel, frame): the one index whose hash is 0 would park xorshift32 at 0
// (every draw 0, RandomVectorInUnitSphere never returns); it gets a fixed non-zero state instead
__device__ __forceinline__ uint StreamSeed(uint index) { uint s = InitSeed(index); return s ? s : 0x9E3779B9u; }
__device__ __forceinline__ uint RandomUInt(uint& seed) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; return seed; }
__device__ __forceinline__ float RandomFloat(uint& seed) { return RandomUInt(seed) * 2.3283064365387e-10f; }
__device__ __forceinline__ f3 RandomVectorInUnitSphere(uint& seed)
{
	while (true) {
		float ax = RandomFloat(seed) * 2 - 1; // draw order x, y, z (unspecified in the reference, template.cpp:711)
		float ay = RandomFloat(seed) * 2 - 1;
		float az = RandomFloat(seed) * 2 - 1;
		f3 a(ax, ay, az);
		if (dot(a, a) > 1) continue;
		return a;
	}
}
__device__ __forceinline__ f3 RandomInHemisphere(uint& seed, const f3& normal)
{
	f3 a = RandomVectorInUnitSphere(seed);
	if (dot(a, normal) > 0.0f) return normalize(a);
	return -normalize(a);
}

// template/template.cpp:846-860: float4(a, w) * M kept term by term (w = 1 / w = 0)
__device__ __forceinline__ f3 xform_pos(const float* c, const f3& a)
{
	return f3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 1.0f,
	          c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 1.0f,
	          c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 1.0f);
}
__device__ __forceinline__ f3 xform_vec(const float* c, const f3& a)
{
	return f3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 0.0f,
	          c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 0.0f,
	          c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 0.0f);
}

// bvh::IntersectAABB (bvh.cpp:819-828), literal form: std::min / std::max ternaries
__device__ __forceinline__ float intersect_aabb_exact(const f3& O, const f3& rD, float rayT, const f3& bmin, const f3& bmax)
{
	float tx1 = (bmin.x - O.x) * rD.x, tx2 = (bmax.x - O.x) * rD.x;
	float tmin = std_min(tx1, tx2), tmax = std_max(tx1, tx2);
	float ty1 = (bmin.y - O.y) * rD.y, ty2 = (bmax.y - O.y) * rD.y;
	tmin = std_max(tmin, std_min(ty1, ty2)), tmax = std_min(tmax, std_max(ty1, ty2));
	float tz1 = (bmin.z - O.z) * rD.z, tz2 = (bmax.z - O.z) * rD.z;
	tmin = std_max(tmin, std_min(tz1, tz2)), tmax = std_min(tmax, std_max(tz1, tz2));
	if (tmax >= tmin && tmin < rayT && tmax > 0) return tmin;
	return 1e30f;
}
// Same function on the hardware min/max.  v_min_f32 / v_max_f32 select one of their operands and
// differ from the ternaries only when an operand is NaN (and in the sign of a zero result, which no
// comparison below can see).  A slab product (b - O) * rD is NaN only through 0 * inf, inf * 0 or a
// NaN input.  ray_is_clean() rules all of those out once per ray: |O| < 1e30 and D finite make
// b - O finite (boxes are bounded by +-1e30, bvh.cpp:96-109) and rD non-zero, and rD finite leaves
// only finite * finite.  Clean rays take this path, every other ray the literal one.
__device__ __forceinline__ bool ray_is_clean(const f3& O, const f3& D, const f3& rD)
{
	const float inf = __builtin_inff();
	return fabsf(O.x) < 1e30f && fabsf(O.y) < 1e30f && fabsf(O.z) < 1e30f && fabsf(D.x) < inf && fabsf(D.y) < inf && fabsf(D.z) < inf &&
	       fabsf(rD.x) < inf && fabsf(rD.y) < inf && fabsf(rD.z) < inf;
}
// v_min_f32 / v_max_f32 / v_min3 / v_max3 issued directly: through fminf/fmaxf hipcc puts a
// canonicalising v_max x,x in front of every operand (12 extra VALU per box).  Operands here are
// never NaN (clean rays), so the selection is exact.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float intersect_aabb_clean(const f3& O, const f3& rD, float rayT, const f3& bmin, const f3& bmax)
{
	const float tx1 = (bmin.x - O.x) * rD.x, tx2 = (bmax.x - O.x) * rD.x;
	const float ty1 = (bmin.y - O.y) * rD.y, ty2 = (bmax.y - O.y) * rD.y;
	const float tz1 = (bmin.z - O.z) * rD.z, tz2 = (bmax.z - O.z) * rD.z;
	const float tmin = hw_max3(hw_min(tx1, tx2), hw_min(ty1, ty2), hw_min(tz1, tz2));
	const float tmax = hw_min3(hw_max(tx1, tx2), hw_max(ty1, ty2), hw_max(tz1, tz2));
	// the three comparisons have no side effects: '&' gives the value of '&&' without the branches
	const bool hit = (tmax >= tmin) & (tmin < rayT) & (tmax > 0);
	return hit ? tmin : 1e30f;
}
// Conservative reachability test used for culling only (never for ordering or for accepting a hit):
// false means that a ray which is clean in the sense above passes outside the box (which the caller has
// inflated by the rounding margin).
__device__ __forceinline__ bool box_reachable(const f3& O, const f3& rD, float rayT, const f3& bmin, const f3& bmax)
{
	const float tx1 = (bmin.x - O.x) * rD.x, tx2 = (bmax.x - O.x) * rD.x;
	const float ty1 = (bmin.y - O.y) * rD.y, ty2 = (bmax.y - O.y) * rD.y;
	const float tz1 = (bmin.z - O.z) * rD.z, tz2 = (bmax.z - O.z) * rD.z;
	const float tmin = hw_max3(hw_min(tx1, tx2), hw_min(ty1, ty2), hw_min(tz1, tz2));
	const float tmax = hw_min3(hw_max(tx1, tx2), hw_max(ty1, ty2), hw_max(tz1, tz2));
	return (tmax >= tmin) & (tmin < rayT) & (tmax > 0);
}
__device__ __forceinline__ float intersect_aabb(bool clean, const f3& O, const f3& rD, float rayT, const f3& bmin, const f3& bmax)
{
	return clean ? intersect_aabb_clean(O, rD, rayT, bmin, bmax) : intersect_aabb_exact(O, rD, rayT, bmin, bmax);
}

// Triangle::Intersect / IsOccluding (template/scene.h:190-237): true when the hit lies in
// (t_min, rayT); tOut receives t.  d = -dot(N, v0) is precomputed at upload with the same expression.
__device__ __forceinline__ bool tri_hit(const f3& O, const f3& D, float rayT, float t_min,
                                        const f3& v0, const f3& v1, const f3& v2, const f3& N, float d, float& tOut)
{
	// The reference returns false at five places on the way (:195, :201, :210, :216, :222); here everything is computed and the
	// verdict is ONE conjunction of the same comparisons, negated where the reference leaves -- a NaN falls the same way through
	// either form, and nothing computed has a side effect.  Lanes of a wave do not leave together, so the early returns saved no
	// instruction and cost a mask save, a branch and a restore each.
	const float NdotRayDir = dot(N, D);
	const float t = -(dot(N, O) + d) / NdotRayDir;
	const f3 p = O + t * D;
	const float e0 = dot(N, cross(v1 - v0, p - v0));
	const float e1 = dot(N, cross(v2 - v1, p - v1));
	const float e2 = dot(N, cross(v0 - v2, p - v2));
	tOut = t;
	return !(fabsf(NdotRayDir) < t_min) & !(t < 0) & !(e0 < 0) & !(e1 < 0) & !(e2 < 0) & (t < rayT) & (t > t_min);
}
// Sphere::Intersect (template/scene.h:351-371): nearest accepted root
__device__ __forceinline__ bool sphere_hit(const f3& O, const f3& D, float rayT, float t_min, const f3& pos, float r2, float& tOut)
{
	f3 oc = O - pos;
	float b = dot(oc, D);
	float c = dot(oc, oc) - r2;
	float d = b * b - c;
	if (d <= 0) return false;
	d = sqrtf(d);
	float t = -b - d;
	if (t < rayT && t > t_min) { tOut = t; return true; }
	t = d - b;
	if (t < rayT && t > t_min) { tOut = t; return true; }
	return false;
}
// Sphere::IsOccluding (template/scene.h:372-381)
__device__ __forceinline__ bool sphere_occludes(const f3& O, const f3& D, float rayT, float t_min, const f3& pos, float r2)
{
	f3 oc = O - pos;
	float b = dot(oc, D);
	float c = dot(oc, oc) - r2;
	float d = b * b - c;
	if (d <= 0) return false;
	d = sqrtf(d);
	float t = -b - d, t2 = d - b;
	return ((t < rayT && t > t_min) || (t2 < rayT && t2 > t_min));
}
// Plane::Intersect / IsOccluding (template/scene.h:405-415)
__device__ __forceinline__ bool plane_hit(const f3& O, const f3& D, float rayT, float t_min, const f3& N, float d, float& tOut)
{
	float t = -(dot(O, N) + d) / (dot(D, N));
	if (t < rayT && t > t_min) { tOut = t; return true; }
	return false;
}

// glass::fresnel (template/scene.h:647-666)
__device__ __forceinline__ float glass_fresnel(const f3& I, const f3& N, float ior)
{
	float cosi = t_clamp(dot(I, N), -1.0f, 1.0f);
	float etai = 1, etat = ior;
	if (cosi > 0) { float tmp = etai; etai = etat; etat = tmp; }
	float sint = etai / etat * sqrtf(t_fmaxf(0.f, 1 - cosi * cosi));
	if (sint >= 1) return 1;
	float cost = sqrtf(t_fmaxf(0.f, 1 - sint * sint));
	cosi = fabsf(cosi);
	float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
	float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
	return (Rs * Rs + Rp * Rp) / 2;
}
// glass::RefractRay (template/scene.h:667-672): the fmin / 1.0 - len^2 / fabs / sqrt chain is double
__device__ __forceinline__ f3 glass_refract(const f3& oRayDir, const f3& normal, float refRatio)
{
	double thd = (double)dot(-oRayDir, normal);
	float theta = (float)(thd < 1.0 ? thd : 1.0);
	f3 perpendicular = refRatio * (oRayDir + theta * normal);
	double len = (double)length(perpendicular);
	float par = (float)(-sqrt(fabs(1.0 - len * len)));
	f3 parallel = par * normal;
	return perpendicular + parallel;
}

} // namespace rtd
