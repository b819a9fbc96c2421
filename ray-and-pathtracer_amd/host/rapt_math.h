// Host-side math for the C++ mirror of the reference's interface.  Types and free functions keep
// the reference's names (float3, mat4, aabb, dot, normalize, TransformPosition ... --
// template/precomp.h:191-283, 478-885, 965-1216; template/template.cpp:800-860) so host code
// written against the reference reads the same.  Only what scene setup and the acceleration-
// structure builders need lives here: shading arithmetic runs on the device (csrc/).
// Built with -ffp-contract=off: builder output must be bit-stable, it decides hit ids.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rapt {

typedef unsigned int uint;

struct float3 {
	float x = 0, y = 0, z = 0;
	float3() = default;
	float3(float a, float b, float c) : x(a), y(b), z(c) {}
	float3(float s) : x(s), y(s), z(s) {}
	float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
	float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct float4 {
	float x = 0, y = 0, z = 0, w = 0;
	float4() = default;
	float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
	float4(const float3& a, float d = 0) : x(a.x), y(a.y), z(a.z), w(d) {}
};
struct int3 { int x = 0, y = 0, z = 0; };

// component-wise min / max with the template's ternary semantics (template/precomp.h:479-480)
inline float fminf_t(float a, float b) { return a < b ? a : b; }
inline float fmaxf_t(float a, float b) { return a > b ? a : b; }
inline float3 fminf(const float3& a, const float3& b) { return float3(fminf_t(a.x, b.x), fminf_t(a.y, b.y), fminf_t(a.z, b.z)); }
inline float3 fmaxf(const float3& a, const float3& b) { return float3(fmaxf_t(a.x, b.x), fmaxf_t(a.y, b.y), fmaxf_t(a.z, b.z)); }

inline float3 operator-(const float3& a) { return float3(-a.x, -a.y, -a.z); }
inline float3 operator+(const float3& a, const float3& b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(const float3& a, const float3& b) { return float3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float3 operator*(const float3& a, float b) { return float3(a.x * b, a.y * b, a.z * b); }
inline float3 operator*(float b, const float3& a) { return float3(b * a.x, b * a.y, b * a.z); }
inline float3 operator/(const float3& a, float b) { return float3(a.x / b, a.y / b, a.z / b); }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(const float3& v) { return sqrtf(dot(v, v)); }
inline float3 normalize(const float3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
inline float3 cross(const float3& a, const float3& b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// bvh.h:26-36
struct aabb {
	float3 bmin = float3(1e30f), bmax = float3(-1e30f);
	void grow(const float3& p) { bmin = fminf(bmin, p); bmax = fmaxf(bmax, p); }
	void grow(const aabb& b) { if (b.bmin.x != 1e30f) { grow(b.bmin); grow(b.bmax); } }
	float area() const { float3 e = bmax - bmin; return e.x * e.y + e.y * e.z + e.z * e.x; }
};

// Rotation matrices evaluate cos/sin in f64 and round once to f32 -- the definition shared with
// the device code for every transcendental on this path ("Transcendentals").
inline float cosf_r(float a) { return (float)cos((double)a); }
inline float sinf_r(float a) { return (float)sin((double)a); }

// row-major 4x4 (template/precomp.h:965-1204)
struct mat4 {
	float cell[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	float& operator[](int idx) { return cell[idx]; }
	float operator()(int i, int j) const { return cell[i * 4 + j]; }
	static mat4 Identity() { return mat4(); }
	static mat4 RotateX(float a) { mat4 r; r.cell[5] = cosf_r(a); r.cell[6] = -sinf_r(a); r.cell[9] = sinf_r(a); r.cell[10] = cosf_r(a); return r; }
	static mat4 RotateY(float a) { mat4 r; r.cell[0] = cosf_r(a); r.cell[2] = sinf_r(a); r.cell[8] = -sinf_r(a); r.cell[10] = cosf_r(a); return r; }
	static mat4 RotateZ(float a) { mat4 r; r.cell[0] = cosf_r(a); r.cell[1] = -sinf_r(a); r.cell[4] = sinf_r(a); r.cell[5] = cosf_r(a); return r; }
	static mat4 Scale(float s) { mat4 r; r.cell[0] = r.cell[5] = r.cell[10] = s; return r; }
	static mat4 Translate(const float3& P) { mat4 r; r.cell[3] = P.x; r.cell[7] = P.y; r.cell[11] = P.z; return r; }
	mat4 Inverted() const; // general inverse by cofactors (the portable branch, template/precomp.h:1126-1166)
};
mat4 operator*(const mat4& a, const mat4& b);
float3 TransformPosition(const float3& a, const mat4& M);
float3 TransformVector(const float3& a, const mat4& M);

} // namespace rapt
