// ORACLE (test infrastructure, NOT product code) -- parity unpinned, see oracle/README.md.
//
// CPU restatement of the vector / matrix arithmetic the reference's trace loop is built on.
// Every helper states the reference line whose arithmetic (operand order, rounding points,
// NaN behaviour of min/max) it follows.  All float code in oracle/ is compiled with
// -ffp-contract=off -fno-fast-math so that each '+', '*', '/' and sqrtf rounds exactly once.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
// oracle/.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

typedef unsigned int uint;

// ---- scalar helpers ---------------------------------------------------------------------
// std::min / std::max as pulled in by 'using namespace std' (template/precomp.h:34); used by
// bvh::IntersectAABB (bvh.cpp:819-828).  min(a,b) = (b<a)?b:a ; max(a,b) = (a<b)?b:a.
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
// Tmpl8::fminf / fmaxf overrides (template/precomp.h:479-480): plain ternaries, NOT libm.
static inline float t_fminf(float a, float b) { return a < b ? a : b; }
static inline float t_fmaxf(float a, float b) { return a > b ? a : b; }
// libm fmax as reached through std::fmax(float,float) in diffuse::scatter (template/scene.h:608):
// returns the non-NaN operand.
static inline float libm_fmaxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
// template/precomp.h:481
static inline float t_rsqrtf(float x) { return 1.0f / sqrtf(x); }
// template/precomp.h:790
static inline float t_clamp(float f, float a, float b) { return t_fmaxf(a, t_fminf(f, b)); }

// float -> int conversion with the x86 cvttss2si result for NaN / out-of-range inputs
// (0x80000000).  C++ leaves those cases undefined; the reference reaches them in
// Scene::GetSkyColor (template/scene.h:1319-1320, quirk Q19).  Both the oracle and the HIP
// path use this explicit definition.
static inline int f2i(float f)
{
	if (!(f > -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000;
	return (int)f;
}

// Transcendentals.  The reference calls the float overloads of cos/sin/acos/exp/pow from the
// MSVC runtime, whose last-ulp behaviour is not reproducible here (quirk Q18).  Oracle and
// HIP path both DEFINE them as: evaluate in f64, round once to f32.  Two sub-ulp-accurate f64
// libraries then agree after rounding except when the f64 value lies within ~2^-29 (relative)
// of an f32 rounding boundary.
static inline float x_cosf(float x) { return (float)cos((double)x); }
static inline float x_sinf(float x) { return (float)sin((double)x); }
static inline float x_acosf(float x) { return (float)acos((double)x); }
static inline float x_asinf(float x) { return (float)asin((double)x); }
static inline float x_expf(float x) { return (float)exp((double)x); }
static inline float x_powf(float a, float b) { return (float)pow((double)a, (double)b); }

// ---- float3 -------------------------------------------------------------------------------
struct float3 {
	float x, y, z;
	float3() = default;
	float3(float a, float b, float c) : x(a), y(b), z(c) {}
	float3(float s) : x(s), y(s), z(s) {} // template/precomp.h float3(float) broadcast ctor
	float operator[](int i) const { return (&x)[i]; }
	float& operator[](int i) { return (&x)[i]; }
};
struct float4 { float x, y, z, w; };

// template/precomp.h:541, 577, 585, 653, 661, 664, 718-721, 754-758
static inline float3 operator-(const float3& a) { return float3(-a.x, -a.y, -a.z); }
static inline float3 operator+(const float3& a, const float3& b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float3 operator-(const float3& a, const float3& b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float3 operator*(const float3& a, const float3& b) { return float3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float3 operator*(const float3& a, float b) { return float3(a.x * b, a.y * b, a.z * b); }
static inline float3 operator*(float b, const float3& a) { return float3(b * a.x, b * a.y, b * a.z); }
static inline float3 operator/(const float3& a, float b) { return float3(a.x / b, a.y / b, a.z / b); }
static inline float3 operator/(const float3& a, const float3& b) { return float3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline void operator+=(float3& a, const float3& b) { a.x += b.x; a.y += b.y; a.z += b.z; }
static inline void operator*=(float3& a, const float3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
static inline void operator/=(float3& a, float b) { a.x /= b; a.y /= b; a.z /= b; }

// template/precomp.h:766, 776
static inline float3 t_fminf(const float3& a, const float3& b) { return float3(t_fminf(a.x, b.x), t_fminf(a.y, b.y), t_fminf(a.z, b.z)); }
static inline float3 t_fmaxf(const float3& a, const float3& b) { return float3(t_fmaxf(a.x, b.x), t_fmaxf(a.y, b.y), t_fmaxf(a.z, b.z)); }
// template/precomp.h:813 -- (x*x + y*y) + z*z, left to right
static inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float sqrLength(const float3& v) { return dot(v, v); }          // :823
static inline float length(const float3& v) { return sqrtf(dot(v, v)); }      // :827
static inline float3 normalize(const float3& v) { float invLen = t_rsqrtf(dot(v, v)); return v * invLen; } // :835
static inline float3 reflect(const float3& i, const float3& n) { return i - 2.0f * n * dot(n, i); }        // :861
static inline float3 cross(const float3& a, const float3& b)                                               // :863
{
	return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// template/precomp.h:885 -- 'fabs(r.x < s)' takes the absolute value of a bool (quirk Q8), so
// the function is true iff every component compares below 1e-4 (compared in double).
static inline bool isZero(const float3& r) { double s = 1e-4; return ((double)r.x < s) && ((double)r.y < s) && ((double)r.z < s); }

// ---- mat4 (row major, template/precomp.h:965-1204) --------------------------------------------
struct mat4 {
	float cell[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	static mat4 Identity() { return mat4(); }
	// :994-996 use cosf/sinf; see the transcendental note above
	static mat4 RotateX(float a) { mat4 r; r.cell[5] = x_cosf(a); r.cell[6] = -x_sinf(a); r.cell[9] = x_sinf(a); r.cell[10] = x_cosf(a); return r; }
	static mat4 RotateY(float a) { mat4 r; r.cell[0] = x_cosf(a); r.cell[2] = x_sinf(a); r.cell[8] = -x_sinf(a); r.cell[10] = x_cosf(a); return r; }
	static mat4 RotateZ(float a) { mat4 r; r.cell[0] = x_cosf(a); r.cell[1] = -x_sinf(a); r.cell[4] = x_sinf(a); r.cell[5] = x_cosf(a); return r; }
	static mat4 Scale(float s) { mat4 r; r.cell[0] = r.cell[5] = r.cell[10] = s; return r; }      // :997
	static mat4 Translate(const float3& P) { mat4 r; r.cell[3] = P.x; r.cell[7] = P.y; r.cell[11] = P.z; return r; } // :1056
	// :1126-1166, the non-MSVC branch: cofactor expansion published with MESA's gluInvertMatrix
	// (quirk Q17: the Windows build runs an SSE variant whose bits differ in the last place).
	mat4 Inverted() const
	{
		const float* c = cell;
		mat4 r;
		float inv[16];
		inv[0] = c[5] * c[10] * c[15] - c[5] * c[11] * c[14] - c[9] * c[6] * c[15] + c[9] * c[7] * c[14] + c[13] * c[6] * c[11] - c[13] * c[7] * c[10];
		inv[1] = -c[1] * c[10] * c[15] + c[1] * c[11] * c[14] + c[9] * c[2] * c[15] - c[9] * c[3] * c[14] - c[13] * c[2] * c[11] + c[13] * c[3] * c[10];
		inv[2] = c[1] * c[6] * c[15] - c[1] * c[7] * c[14] - c[5] * c[2] * c[15] + c[5] * c[3] * c[14] + c[13] * c[2] * c[7] - c[13] * c[3] * c[6];
		inv[3] = -c[1] * c[6] * c[11] + c[1] * c[7] * c[10] + c[5] * c[2] * c[11] - c[5] * c[3] * c[10] - c[9] * c[2] * c[7] + c[9] * c[3] * c[6];
		inv[4] = -c[4] * c[10] * c[15] + c[4] * c[11] * c[14] + c[8] * c[6] * c[15] - c[8] * c[7] * c[14] - c[12] * c[6] * c[11] + c[12] * c[7] * c[10];
		inv[5] = c[0] * c[10] * c[15] - c[0] * c[11] * c[14] - c[8] * c[2] * c[15] + c[8] * c[3] * c[14] + c[12] * c[2] * c[11] - c[12] * c[3] * c[10];
		inv[6] = -c[0] * c[6] * c[15] + c[0] * c[7] * c[14] + c[4] * c[2] * c[15] - c[4] * c[3] * c[14] - c[12] * c[2] * c[7] + c[12] * c[3] * c[6];
		inv[7] = c[0] * c[6] * c[11] - c[0] * c[7] * c[10] - c[4] * c[2] * c[11] + c[4] * c[3] * c[10] + c[8] * c[2] * c[7] - c[8] * c[3] * c[6];
		inv[8] = c[4] * c[9] * c[15] - c[4] * c[11] * c[13] - c[8] * c[5] * c[15] + c[8] * c[7] * c[13] + c[12] * c[5] * c[11] - c[12] * c[7] * c[9];
		inv[9] = -c[0] * c[9] * c[15] + c[0] * c[11] * c[13] + c[8] * c[1] * c[15] - c[8] * c[3] * c[13] - c[12] * c[1] * c[11] + c[12] * c[3] * c[9];
		inv[10] = c[0] * c[5] * c[15] - c[0] * c[7] * c[13] - c[4] * c[1] * c[15] + c[4] * c[3] * c[13] + c[12] * c[1] * c[7] - c[12] * c[3] * c[5];
		inv[11] = -c[0] * c[5] * c[11] + c[0] * c[7] * c[9] + c[4] * c[1] * c[11] - c[4] * c[3] * c[9] - c[8] * c[1] * c[7] + c[8] * c[3] * c[5];
		inv[12] = -c[4] * c[9] * c[14] + c[4] * c[10] * c[13] + c[8] * c[5] * c[14] - c[8] * c[6] * c[13] - c[12] * c[5] * c[10] + c[12] * c[6] * c[9];
		inv[13] = c[0] * c[9] * c[14] - c[0] * c[10] * c[13] - c[8] * c[1] * c[14] + c[8] * c[2] * c[13] + c[12] * c[1] * c[10] - c[12] * c[2] * c[9];
		inv[14] = -c[0] * c[5] * c[14] + c[0] * c[6] * c[13] + c[4] * c[1] * c[14] - c[4] * c[2] * c[13] - c[12] * c[1] * c[6] + c[12] * c[2] * c[5];
		inv[15] = c[0] * c[5] * c[10] - c[0] * c[6] * c[9] - c[4] * c[1] * c[10] + c[4] * c[2] * c[9] + c[8] * c[1] * c[6] - c[8] * c[2] * c[5];
		const float det = c[0] * inv[0] + c[1] * inv[4] + c[2] * inv[8] + c[3] * inv[12];
		if (det != 0) {
			const float invdet = 1.0f / det;
			for (int i = 0; i < 16; i++) r.cell[i] = inv[i] * invdet;
		}
		return r;
	}
};
// template/template.cpp:800-813
static inline mat4 operator*(const mat4& a, const mat4& b)
{
	mat4 r;
	for (uint i = 0; i < 16; i += 4)
		for (uint j = 0; j < 4; ++j)
			r.cell[i + j] = (a.cell[i + 0] * b.cell[j + 0]) + (a.cell[i + 1] * b.cell[j + 4]) +
			                (a.cell[i + 2] * b.cell[j + 8]) + (a.cell[i + 3] * b.cell[j + 12]);
	return r;
}
// template/template.cpp:846-860: float4(a, w) * M, then xyz.  The w term is kept (x + c*0 turns
// -0 into +0 and propagates non-finite cells exactly as the reference does).
static inline float3 TransformPosition(const float3& a, const mat4& M)
{
	const float* c = M.cell;
	return float3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 1.0f,
	              c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 1.0f,
	              c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 1.0f);
}
static inline float3 TransformVector(const float3& a, const mat4& M)
{
	const float* c = M.cell;
	return float3(c[0] * a.x + c[1] * a.y + c[2] * a.z + c[3] * 0.0f,
	              c[4] * a.x + c[5] * a.y + c[6] * a.z + c[7] * 0.0f,
	              c[8] * a.x + c[9] * a.y + c[10] * a.z + c[11] * 0.0f);
}

// ---- aabb (bvh.h:26-36) ----------------------------------------------------------------------
struct aabb {
	float3 bmin = float3(1e30f), bmax = float3(-1e30f);
	void grow(const float3& p) { bmin = t_fminf(bmin, p); bmax = t_fmaxf(bmax, p); }
	void grow(const aabb& b) { if (b.bmin.x != 1e30f) { grow(b.bmin); grow(b.bmax); } }
	float area() const { float3 e = bmax - bmin; return e.x * e.y + e.y * e.z + e.z * e.x; }
};

// ---- constants (template/common.h:8-12) -------------------------------------------------------
static const float PI = 3.14159265358979323846264f;
static const float INVPI = 0.31830988618379067153777f;
static const float INV2PI = 0.15915494309189533576888f;
static const float GAMMA = 0.57142857142857142857143f;
static const float TWOPI = 6.28318530717958647692528f;

// ---- RNG (template/template.cpp:670-724) ------------------------------------------------------
// The reference draws from ONE process-global xorshift32 state in scanline order (and races on
// it under OpenMP).  A GPU cannot replay that order, so oracle and HIP path both use the
// template's own per-stream forms (template.cpp:680-683, 695-702) with one stream per
// (pixel, frame): seed = InitSeed(seed_base + pixel + frame*W*H).  Deliberate, documented
// deviation (SURVEY.md section 7 "RNG semantics").
static inline uint WangHash(uint s) { s = (s ^ 61) ^ (s >> 16); s *= 9; s = s ^ (s >> 4); s *= 0x27d4eb2d; s = s ^ (s >> 15); return s; }
static inline uint InitSeed(uint seedBase) { return WangHash((seedBase + 1) * 17); }
// Start of the stream of one (pixel, frame).  WangHash is a bijection, so exactly one index maps to
// state 0, where xorshift32 is stuck (every draw 0) and RandomVectorInUnitSphere never terminates.  A
// single global stream can never reach 0; per-sample streams can (at 4K, frame 176 of seed base
// 0x12345678 contains that index), so the zero state is replaced by a fixed non-zero one.
static inline uint StreamSeed(uint index) { uint s = InitSeed(index); return s ? s : 0x9E3779B9u; }
static inline uint RandomUInt(uint& seed) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; return seed; }
static inline float RandomFloat(uint& seed) { return RandomUInt(seed) * 2.3283064365387e-10f; }
// template.cpp:709-715.  The three draws are constructor arguments in the reference, so their
// order is unspecified by C++ (quirk Q14); defined here as x, then y, then z.
static inline float3 RandomVectorInUnitSphere(uint& seed)
{
	while (true) {
		float ax = RandomFloat(seed) * 2 - 1;
		float ay = RandomFloat(seed) * 2 - 1;
		float az = RandomFloat(seed) * 2 - 1;
		float3 a(ax, ay, az);
		if (sqrLength(a) > 1) continue;
		return a;
	}
}
// template.cpp:717-724
static inline float3 RandomInHemisphere(uint& seed, const float3& normal)
{
	float3 a = RandomVectorInUnitSphere(seed);
	if (dot(a, normal) > 0.0) return normalize(a);
	else return -normalize(a);
}

} // namespace orc
