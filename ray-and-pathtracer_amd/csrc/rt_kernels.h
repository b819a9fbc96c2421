// Wavefront kernels of the pixel loop: generate -> [extend -> shade -> connect -> light -> finish]* -> accumulate.
//
// Work unit: one *sample* = (pixel, frame).  The samples of a render call form a pool ordered
// frame-major; a fixed set of *slots* (SoA path state in HBM, 16-byte elements so every lane moves
// whole dwordx4s) pulls samples from the pool, so every round has work for (nearly) every slot until
// the pool is empty, however path lengths vary over the image.  A status byte per slot says what the
// slot needs in the current round; there are no compacted queues (appending to one costs a same-
// address atomic per wave, which measured as THE cost of the shading kernels), so the shading kernels
// are plain one-thread-per-slot passes with coalesced loads and stores:
//   extend   Scene::FindNearest for every ACTIVE slot   (persistent waves, the dominant kernel)
//   shade    the body of Renderer::Trace / Renderer::Sample at the hit: leaf terms, material switch,
//            light sampling, the next ray
//   connect  Scene::IsOccluded towards every sampled light (traversal only, one byte per light)
//   light    the direct-light terms of a diffuse hit, in light order (their energy bookkeeping
//            feeds the next bounce)
//   finish   slots whose segment ended: resume a pending Whitted branch, or store the finished
//            sample and pull the next one from the pool (new primary ray)
// Finished samples go to a [frame][pixel] buffer; accumulate adds them to the accumulator in frame
// order, so the sum is the one a sequential Tick loop produces (renderer.cpp:279-282).  Radiance is
// carried forward as path weights (W) instead of being combined on return from recursion; per sample
// the segment order is the reference's depth-first order.
#pragma once
#include "rt_scene_dev.h"

namespace rtd {

#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 8 // waves per SIMD the extend kernel is compiled for (launch bound: 64 VGPRs; it needs exactly that since round 4, DESIGN.md finding 55)
#endif
#define RT_PEND_CAP 12 // pending Whitted branches per pixel (glass: <= 3 at depth 4; shiny diffuse: more)

struct DCamera {
	float camPos[3], topLeft[3], topRight[3], bottomLeft[3];
	int fisheye; float viewAngle, yAngle;
	int width, height;
};

struct PathState {
	float4* O[2];    // ray origin xyz, w = ray.t on entry (tmax); double buffered by round parity
	float4* D[2];    // ray direction xyz
	float4* hitN[2]; // hit normal xyz, w = t; double buffered by round parity like the rays
	int2* hitId[2];  // objIdx, material
	float4* W;       // path weight xyz, w = depth (int bits)
	float4* E;       // energy xyz, w = RNG state (uint bits)
	float4* L;       // radiance of the current sample xyz, w = sample id in the pool (uint bits)
	float4* sh;      // [light][slot] sampled light position xyz (plane 0: w = flags); plane nLights: weight of the segment
	float4* hitP;    // ray.IntersectionPoint() of a diffuse hit (written by shade: connect needs nothing else of the ray)
	unsigned char* vis; // [light][slot] 1: the light is occluded
	unsigned char* status; // [slot] ST_* bits: what the slot needs this round
	float4* pend;    // [slot][RT_PEND_CAP][4]: pending Whitted branches {O,depth} {D,-} {W,-} {E,-}
	int* pendCount;  // [slot]
	int nSlots;
};

struct RenderParams {
	int mode;          // RT_MODE_WHITTED / RT_MODE_PATH
	uint frame0;       // first frame of this batch
	uint nSamples;     // samples of this pool
	uint sampleFirst;  // id of the pool's first sample within the batch (batch = frames * tile pixels, frame-major)
	uint tilePixels;   // pixels in the tile (rows of this shard * width)
	float4* samples;   // [batch frame][tile pixel] finished samples (gamma applied in path mode)
	uint seedBase;
	int rowFirst, rowStride; // slot s is pixel x = s % width, y = rowFirst + (s / width) * rowStride
	int maxDepth;      // depth argument of Trace
	float4* accum;     // accumulator, whole image
	// rt_trace_batch: caller rays instead of camera rays, raw radiance out instead of accumulation
	const float* customO; const float* customD; float4* customOut; int customDepth;
	float customE[3];  // rt_trace_batch_energy: the 'energy' argument of Trace / Sample
	uint permShift;    // rt_mega.h: log2 of the tile the permutation deals out
	uint permMul;      // rt_mega.h: work item w is sample (w * permMul) % nSamples (0: w itself), see run_whitted_mega
	int finishInline;  // path mode with a slot per sample: nothing to resume and no sample to pull when a segment ends,
	                   // so shade / light store the finished sample themselves and the round has no finish pass
	int deferGamma;    // path mode: a finished sample is stored raw with w = 1 and k_accumulate applies the gamma (see store_sample)
	int sceneRt;       // rt_trace_batch with rt_set_scene_raytracer: scene.raytracer as the caller's Scene holds it, when it is NOT what the
	                   // function called implies (Trace with the flag clear, Sample with it set: renderer.cpp:33-43, 107-121, 143-153) -- the
	                   // general kernels below; -1: the flag follows the function, as Tick calls it
};

#define ST_ACTIVE 1      // has a ray for extend + shade
#define ST_SHADOW 2      // diffuse hit: connect + light
#define ST_ENDS_AFTER 4  // ... and the segment ends once light has run
#define ST_ENDED 8       // segment ended: finish resumes a pending branch or starts a new sample

struct Queues {
	uint* active; // compacted ACTIVE slots (built per round by k_compact)
	uint* shadow; // compacted SHADOW slots
	uint* ended;  // compacted ENDED slots (built after light)
	int* heads;   // work heads of extend ([0, RT_HEADS)) and connect ([RT_HEADS, 2 RT_HEADS)), RT_HEAD_STRIDE ints apart
	int* counts;  // [0] active count, [1] ended count, [2] shadow count, [3] overflow flag, [4] extend head, [6] connect head, [7] next sample in the pool,
	              // [8] shadow rays the wide walk handed back
	uint* leftover; // connect work items of rays that are not clean (the 4-wide walk is exact for clean rays only): redone by the binary walk
};

// ---- camera (camera.h:24-41) ---------------------------------------------------------------
__device__ __forceinline__ f3 cam_rotate_y(const f3& p, const f3& center, float theta)
{
	double c = cos((double)theta), s = sin((double)theta);
	f3 vect = p - center;
	f3 xT((float)c, 0, (float)-s), zT((float)s, 0, (float)c);
	f3 res(dot(vect, xT), vect.y, dot(vect, zT));
	return res + center;
}
__device__ __forceinline__ f3 cam_rotate_x(const f3& p, const f3& center, float theta, float yAngle)
{
	double c = cos((double)theta), s = sin((double)theta);
	f3 vect = p - center;
	vect = cam_rotate_y(vect, f3(0.f), -yAngle);
	f3 zT(0, (float)-s, (float)c), yT(0, (float)c, (float)s);
	f3 res(vect.x, dot(vect, yT), dot(vect, zT));
	res = cam_rotate_y(res, f3(0.f), yAngle);
	return res + center;
}
__device__ __forceinline__ void primary_ray(const DCamera& C, int x, int y, f3& O, f3& D)
{
	const f3 camPos(C.camPos[0], C.camPos[1], C.camPos[2]), TL(C.topLeft[0], C.topLeft[1], C.topLeft[2]);
	const f3 TR(C.topRight[0], C.topRight[1], C.topRight[2]), BL(C.bottomLeft[0], C.bottomLeft[1], C.bottomLeft[2]);
	O = camPos;
	if (C.fisheye) {
		const float aspect = (float)C.width / (float)C.height;
		f3 screenCenter = TL + .5f * (TR - TL) + .5f * (BL - TL);
		const float u = (float)(x - C.width / 2) * (aspect * C.viewAngle / C.width);
		const float v = (float)(y - C.width / 2) * (C.viewAngle / C.height);
		f3 newRay = cam_rotate_x(cam_rotate_y(normalize(screenCenter - camPos), camPos, -u), camPos, -v, C.yAngle);
		D = normalize(newRay);
		return;
	}
	const float u = (float)x * (1.0f / C.width);
	const float v = (float)y * (1.0f / C.height);
	const f3 P = TL + u * (TR - TL) + v * (BL - TL);
	D = normalize(P - camPos);
}

// ---- lights (template/scene.h:121-137, 152-165) -----------------------------------------------
__device__ __forceinline__ f3 light_intensity(const DLight& L, const f3& p, const f3& n, const f3& from)
{
	const f3 col(L.col[0], L.col[1], L.col[2]);
	if (L.kind == 0) {
		float dis = length(from - p);
		f3 dir = from - p;
		float cos_ang = dot(normalize(n), normalize(dir));
		float relStr = 1 / (dis * RT_PI) * L.strength;
		// isZero(float3) compares each component '< 1e-4' in double (template/precomp.h:885)
		if (dis <= L.radius && ((double)cos_ang < 1e-4)) return L.strength * col;
		return relStr * col;
	}
	if (L.kind == 1) {
		const f3 pos(L.pos[0], L.pos[1], L.pos[2]), normal(L.normal[0], L.normal[1], L.normal[2]);
		f3 dir = p - pos;
		float sTheta = length(cross(dir, normal)) / length(dir) * length(normal);
		if (dot(dir, normal) < 0) return f3(0.0f);
		float dis = length(dir);
		float str = L.sinAngle - sTheta > 0 ? x_asinf(L.sinAngle) - x_asinf(sTheta) : 0;
		return f3(1 / dis * str * L.strength);
	}
	return f3(1.0f);
}
__device__ __forceinline__ f3 light_position(const DLight& L, bool raytracer, uint& seed)
{
	const f3 pos(L.pos[0], L.pos[1], L.pos[2]);
	if (L.kind != 0 || raytracer) return pos;
	float newRad = L.radius * sqrtf(RandomFloat(seed));
	float theta = RandomFloat(seed) * 2 * RT_PI;
	return f3(pos.x + newRad * x_cosf(theta), pos.y + newRad * x_sinf(theta), pos.z);
}

// Scene::GetSkyColor (template/scene.h:1312-1327): the texel the direction D looks at (null without a sky texture)
__device__ __forceinline__ const unsigned char* sky_texel(const DScene& S, const f3& D)
{
	if (!S.sky) return nullptr;
	f3 horizontalProj(D.x, 0, D.z);
	float cHeight = dot(D, f3(0, -1, 0));
	f3 nh = normalize(horizontalProj);
	float cOrient = dot(f3(0, 0, 1), nh);
	float sOrient = dot(f3(1, 0, 0), nh);
	sOrient = sOrient > 0 ? 1 : -1;
	int y = f2i(((cHeight + 1) / 2) * (S.skyH - 1));
	int x = f2i((((sOrient * x_acosf(cOrient)) + RT_PI) / RT_TWOPI) * (S.skyW - 1));
	if (x >= S.skyW) x = S.skyW - 1;
	if (y >= S.skyH) y = S.skyH - 1;
	if (y < 0) y = 0;
	if (x < 0) x = 0;
	return S.sky + (size_t)(x + S.skyW * y) * S.skyN;
}
__device__ __forceinline__ f3 sky_color(const DScene& S, const f3& D)
{
	const unsigned char* p = sky_texel(S, D);
	if (!p) return f3(0.0f);
	return f3((float)p[0], (float)p[1], (float)p[2]) / 255;
}
// The gamma itself is evaluated in single precision (round 4; the reference's expression is a double-precision pow rounded to
// float, which is what this was until round 3: 200 M double-precision pow per bench step, 1.45 ms of k_accumulate; the device
// library's powf: 1.0 ms).  It is an OUTPUT transform: its value goes into the accumulator and nowhere else -- no ray, no texel
// index, no random draw depends on it, so no hit id can move.  x^g = 2^(g log2 x) on the hardware's log2 / exp2 (1 ulp each)
// with the two roundings that would be amplified by the exponent's size repaired: the rounding of log2 x from x / 2^l0 = 1 + d
// (log2(1 + d) = d log2 e to first order, d ~ 1e-6), the rounding of g * l0 by an fma; the result is 2^t_hi * (1 + ln 2 * t_lo).
// Error against the rounded double-precision value: <= 4 ulp with every hardware result a full ulp off the worst way (4.8e-7 relative:
// 200 x inside the 1e-4 radiance bar), 1.5 ulp with correctly rounded ones (profiles/r04_gamma_error.txt); measured
// <= 2 on the 256 sky values: tests/test_gpu_parity.py::test_gamma_of_finished_samples).  Values outside [1e-30, 1e30] -- zero,
// negatives, NaN, inf (a directly viewed light, Q7), denormals -- take the library's powf.  ONE function everywhere a finished
// sample is made (here, k_accumulate, k_gamma_lut), so the knobs that move the gamma between kernels still give identical bits.
// RT_EXACT_GAMMA=1 (read at rt_create, one value per device): the reference's expression itself, pow in double rounded to float
// (x_powf) -- what rounds 1-3 computed.  With it the accumulator's bits are the oracle's wherever the radiance's are; the parity
// tier runs once that way (tests/test_gpu_parity.py::test_exact_gamma_switch), the product keeps the fast form.
// (Only where samples are READ -- k_accumulate, k_gamma_lut -- : the switch forces the deferred gamma, so the shading kernels,
// whose register budget a double-precision pow in their call graph would eat, never see it.)
__device__ int g_exactGamma;
__device__ __forceinline__ float gamma_powf(float x)
{
	if (!(x >= 1e-30f && x <= 1e30f)) return powf(x, RT_GAMMA);
	const float l0 = __builtin_amdgcn_logf(x);                       // log2 x, rounded
	const float d = __builtin_fmaf(x, __builtin_amdgcn_exp2f(-l0), -1.0f); // x / 2^l0 - 1
	const float tHi = RT_GAMMA * l0;
	const float tLo = __builtin_fmaf(RT_GAMMA, l0, -tHi) + RT_GAMMA * (d * 1.4426950408889634f);
	const float p = __builtin_amdgcn_exp2f(tHi);
	return __builtin_fmaf(p, tLo * 0.6931471805599453f, p);
}
// gammaLut[b] = the finished path-mode sample of radiance b / 255 (store_sample: pow(c, GAMMA) per channel, renderer.cpp:279-282)
__global__ void k_gamma_lut(float* lut)
{
	const int b = (int)threadIdx.x;
	const f3 c = f3((float)b, (float)b, (float)b) / 255;
	lut[b] = g_exactGamma ? x_powf(c.x * 1, RT_GAMMA) : gamma_powf(c.x * 1);
}

// diffuse::scatter (template/scene.h:605-620): att out, energy in/out
__device__ __forceinline__ f3 diffuse_scatter(const DMaterial& m, const f3& rayD, const f3& lightDir, const f3& lightIntensity, const f3& normal, f3& energy)
{
	const f3 albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
	f3 reflectionDirection = reflect(-lightDir, normal);
	f3 specularColor = x_powf(libm_fmaxf(0.0f, -dot(reflectionDirection, rayD)), (float)m.N) * lightIntensity;
	f3 att = albedo * lightIntensity * m.diffu + specularColor * m.specu;
	f3 retention = f3(1.0f) - albedo;
	f3 newEnergy = energy - retention;
	energy = newEnergy.x > 0 ? newEnergy : f3(0.0f);
	return att;
}

// Per-wave ranges: wave w of the grid owns the contiguous slots [w*R, (w+1)*R), R a multiple of 64.
// A pass that must hand out consecutive ids (queue positions, pool samples) to the slots that ask
// runs over its range twice: count the askers, reserve them with ONE atomic per wave per launch,
// then assign base + running prefix.  (One atomic per 64 slots on a single counter measured as the
// whole cost of such passes: ~88 same-address atomics per microsecond.)
__device__ __forceinline__ void wave_range(int nSlots, int& first, int& last)
{
	const int waves = (gridDim.x * blockDim.x) >> 6;
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	int per = (nSlots + waves - 1) / waves;
	per = (per + 63) & ~63;
	first = wave * per;
	last = first + per < nSlots ? first + per : nSlots;
	if (first > nSlots) first = nSlots;
}

// compact: queue <- slots whose status has the (single) bit 'bit', in slot order.  A lane reads 16
// status bytes at a time (one dwordx4), a wave 1024 slots per iteration; wave w owns a contiguous
// range (a multiple of 1024 slots), counts it, reserves its queue positions with one atomic, then writes.
__device__ __forceinline__ uint4 status16(const unsigned char* status, int slot0, int last, uint bits)
{
	if (slot0 >= last) return make_uint4(0, 0, 0, 0);
	uint4 v = *(const uint4*)(status + slot0);
	v.x &= bits, v.y &= bits, v.z &= bits, v.w &= bits;
	const int valid = last - slot0; // bytes of this vector that are slots
	if (valid < 16) {
		uint w[4] = { v.x, v.y, v.z, v.w };
		for (int k = 0; k < 4; k++) {
			const int vb = valid - 4 * k;
			if (vb <= 0) w[k] = 0;
			else if (vb < 4) w[k] &= (1u << (8 * vb)) - 1;
		}
		v = make_uint4(w[0], w[1], w[2], w[3]);
	}
	return v;
}
#define RT_COMPACT_BLOCK 512
// status[0 .. nSlots) (readable up to the next multiple of 16) -> queue of the entries that have 'bit', *count += their number
__device__ __forceinline__ void compact_body(const unsigned char* status, int nSlots, int bit, uint* queue, int* count)
{
	__shared__ int waveTotal[RT_COMPACT_BLOCK / 64];
	__shared__ int blockBase;
	__shared__ uint stage[RT_COMPACT_BLOCK / 64][1024]; // per wave: the slots selected in one iteration, in order
	const uint lane = threadIdx.x & 63;
	const uint bits = (uint)bit * 0x01010101u;
	const int waves = (gridDim.x * blockDim.x) >> 6;
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	int per = (nSlots + waves - 1) / waves;
	per = (per + 1023) & ~1023;
	const long long firstL = (long long)wave * per;
	const int first = firstL < nSlots ? (int)firstL : nSlots;
	const int last = firstL + per < nSlots ? (int)(firstL + per) : nSlots;
	int mine = 0;
	for (int s0 = first; s0 < last; s0 += 1024) {
		const uint4 v = status16(status, s0 + (int)lane * 16, last, bits);
		mine += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
	}
	int total = mine;
	for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
	// one atomic per BLOCK (same-address atomics retire at ~88 per microsecond: one per wave of a
	// 7000-wave grid was most of this kernel's time)
	const int wib = threadIdx.x >> 6;
	if (lane == 0) waveTotal[wib] = total;
	__syncthreads();
	if (threadIdx.x == 0) {
		int sum = 0;
		for (int w = 0; w < RT_COMPACT_BLOCK / 64; w++) sum += waveTotal[w];
		blockBase = sum > 0 ? atomicAdd(count, sum) : 0;
	}
	__syncthreads();
	int base = blockBase;
	for (int w = 0; w < wib; w++) base += waveTotal[w];
	if (total == 0) return;
	uint* mystage = stage[wib];
	for (int s0 = first; s0 < last; s0 += 1024) {
		const int slot0 = s0 + (int)lane * 16;
		const uint4 v = status16(status, slot0, last, bits);
		const int c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
		int incl = c; // inclusive prefix sum over the wave
		for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
		// a lane's slots go to LDS at its prefix; the wave then writes them out 64 consecutive dwords at
		// a time (a lane storing its own up-to-16 entries straight to memory touches 16 cache lines)
		int at = incl - c;
		const uint w[4] = { v.x, v.y, v.z, v.w };
		for (int k = 0; k < 4; k++) {
			uint x = w[k];
			while (x) {
				const int b = __ffs(x) - 1;
				mystage[at++] = (uint)(slot0 + 4 * k + (b >> 3));
				x &= x - 1;
			}
		}
		const int n = __shfl(incl, 63);
		__builtin_amdgcn_wave_barrier(); // the stage is private to this wave and its LDS accesses execute in order: no s_barrier needed
		for (int i = (int)lane; i < n; i += 64) queue[base + i] = mystage[i];
		__builtin_amdgcn_wave_barrier();
		base += n;
	}
}
__global__ void __launch_bounds__(RT_COMPACT_BLOCK) k_compact(PathState P, int bit, uint* queue, int* count)
{
	compact_body(P.status, P.nSlots, bit, queue, count);
}

__device__ __forceinline__ void flush_counters(DCounters* g, const LaneCounters& lc, uint nearest, uint occluded)
{
	// wave reduction, then one atomic per field per wave
	uint v[8] = { lc.inner, lc.prim, lc.tlasInner, lc.inst, nearest, occluded, lc.brute, lc.light };
	unsigned long long* out = (unsigned long long*)g;
	for (int k = 0; k < 8; k++) {
		uint x = v[k];
		for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
		if ((threadIdx.x & 63) == 0 && x) atomicAdd(out + k, (unsigned long long)x);
	}
}

// Store a freshly created ray: run the head tests of Scene::FindNearest on it now (dense kernel), keep
// the shortened ray.t in O.w and the candidate hit in D.w for extend.
__device__ __forceinline__ void emit_ray(const DScene& S, PathState& P, int buf, int slot, const f3& O, const f3& D, float t_min)
{
	float rayT = 1e34f; // Ray constructor default (template/scene.h:42)
	HitRef head;
	head.kind = -1, head.inst = -1, head.prim = 0, head.t = 0;
	LaneCounters unused;
	find_nearest_head<false>(S, O, D, t_min, rayT, head, unused);
	P.O[buf][slot] = mk4(O, rayT);
	P.D[buf][slot] = mk4(D, __uint_as_float(pack_head(head)));
}
__device__ __forceinline__ float mode_t_min(int mode) { return mode == 0 ? (float)1e-6 : 0.001f; } // renderer.cpp:24, :131

// Put sample 'sid' of the pool into a slot: seed, jitter, primary ray (renderer.cpp:263-278)
// depth a fresh sample starts with (renderer.cpp:269 / :278; rt_trace_batch: the caller's)
__device__ __forceinline__ int start_depth(const RenderParams& R) { return R.customO ? R.customDepth : (R.mode == 0 ? R.maxDepth : 4); }
// camera sample 'sid' of the pool (renderer.cpp:263-278): its seed, jitter and primary ray; deterministic in (R, C, sid)
__device__ __forceinline__ void sample_primary(const DCamera& C, const RenderParams& R, uint sid, f3& O, f3& D, uint& seed)
{
	const uint lp = sid % R.tilePixels, frame = R.frame0 + sid / R.tilePixels;
	const int x = (int)(lp % (uint)C.width), y = R.rowFirst + (int)(lp / (uint)C.width) * R.rowStride;
	const int pixel = y * C.width + x;
	seed = StreamSeed(R.seedBase + (uint)pixel + frame * (uint)(C.width * C.height));
	if (R.mode == 0) primary_ray(C, x, y, O, D);
	else {
		float newX = x + (RandomFloat(seed) * 2 - 1);
		float newY = y + (RandomFloat(seed) * 2 - 1);
		primary_ray(C, (int)newX, (int)newY, O, D); // jitter truncated by the int parameters (renderer.cpp:276-278)
	}
}
// energy and RNG state a sample starts its first segment with
__device__ __forceinline__ float4 fresh_energy(const DCamera& C, const RenderParams& R, uint sid)
{
	if (R.customO) return make_float4(R.customE[0], R.customE[1], R.customE[2], __uint_as_float(StreamSeed(R.seedBase + sid)));
	f3 O, D;
	uint seed;
	sample_primary(C, R, sid, O, D, seed); // the ray itself is not needed here (and not computed: dead code)
	return make_float4(1, 1, 1, __uint_as_float(seed));
}
// writeWL = false: the caller guarantees that the first shade of this slot runs with 'fresh' set and rebuilds
// W = (1,1,1,depth), L = (0,0,0,sid) and E = fresh_energy() itself instead of reading them
__device__ __forceinline__ void start_sample(const DScene& S, const DCamera& C, const RenderParams& R, PathState& P, int slot, uint sid, int parityOut, bool writeWL = true)
{
	f3 O, D;
	uint seed = 0;
	int depth;
	if (R.customO) {
		O = f3(R.customO[3 * sid], R.customO[3 * sid + 1], R.customO[3 * sid + 2]);
		D = f3(R.customD[3 * sid], R.customD[3 * sid + 1], R.customD[3 * sid + 2]);
		seed = StreamSeed(R.seedBase + sid);
		depth = R.customDepth;
	} else {
		sample_primary(C, R, sid, O, D, seed);
		depth = R.mode == 0 ? R.maxDepth : 4;
	}
	emit_ray(S, P, parityOut, slot, O, D, mode_t_min(R.mode));
	if (writeWL) {
		P.E[slot] = fresh_energy(C, R, sid);
		P.W[slot] = make_float4(1, 1, 1, __int_as_float(depth));
		P.L[slot] = make_float4(0, 0, 0, __uint_as_float(sid));
	}
	if (P.pendCount) P.pendCount[slot] = 0;
}

// A finished sample goes to the [frame][pixel] buffer (renderer.cpp:270 / :279-282: gamma per sample in path
// mode) or to the caller's array (rt_trace_batch).
// Deferred gamma (R.deferGamma): the three double-precision pow() of a path-mode sample are the same function of the same three
// floats wherever they run.  In the shading kernels they run for the lanes of a wave whose path just ended (a third of them,
// the others wait) at five waves per SIMD, and their registers are part of those kernels' floor; k_accumulate reads every
// sample anyway, dense and with the whole machine: the sample is stored raw with w = 1 and gets its gamma there (w = 0: the
// value is final -- the sky texel table of generate).  Same bits, same order of the accumulator's additions.
__device__ __forceinline__ float4 gamma_sample(const f3& L) { return make_float4(gamma_powf(L.x * 1), gamma_powf(L.y * 1), gamma_powf(L.z * 1), 0.0f); }
// the same as a called function: the kernels that store finished samples (RT_DEFER_GAMMA=0, the slot pipeline) do not carry three inlined copies
__device__ __noinline__ void store_gamma_sample(float4* dst, float x, float y, float z) { *dst = gamma_sample(f3(x, y, z)); }
__device__ __forceinline__ void store_sample(const RenderParams& R, uint sid, const f3& L)
{
	if (R.customOut) R.customOut[sid] = mk4(L, 0.0f);
	else if (R.mode == 0) R.samples[sid] = make_float4(L.x / (float)1, L.y / (float)1, L.z / (float)1, 0.0f);
	else if (R.deferGamma) R.samples[sid] = mk4(L, 1.0f);
	else store_gamma_sample(R.samples + sid, L.x, L.y, L.z);
}

__device__ __forceinline__ void push_pending(PathState& P, int slot, const f3& O, const f3& D, const f3& W, const f3& E, int depth, int* overflow)
{
	int n = P.pendCount[slot];
	if (n >= RT_PEND_CAP) { *overflow = 2; return; }
	float4* e = P.pend + ((size_t)slot * RT_PEND_CAP + n) * 4;
	e[0] = mk4(O, __int_as_float(depth)), e[1] = mk4(D, 0.0f), e[2] = mk4(W, 0.0f), e[3] = mk4(E, 0.0f);
	P.pendCount[slot] = n + 1;
}

// ---- kernels -----------------------------------------------------------------------------------

__global__ void __launch_bounds__(RT_BLOCK) k_generate(DScene S, DCamera C, RenderParams R, PathState P, Queues Q)
{
	const int slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= P.nSlots) return;
	// slots take the first nSlots samples of the pool; with a slot per sample (finishInline) round 0 shades every
	// slot 'fresh', so the constant W and L are not written here and not read there (64 B per sample)
	start_sample(S, C, R, P, slot, R.sampleFirst + (uint)slot, 0, !R.finishInline);
	P.status[slot] = ST_ACTIVE;
	if (slot == 0) Q.counts[7] = P.nSlots, Q.counts[1] = 0;
}

// round bookkeeping between kernels: reset the work heads and the queue counts
// which = 1: the extend side (active / ended counts, extend's work heads); 2: the connect side (shadow count, connect's
// work heads); 3: both.  They are reset apart when connect of round r shares a launch with extend of round r + 1.
__global__ void k_round_begin(Queues Q, int poolFollowsEnded, int allActive, int which)
{
	if (which & 1) {
		if (poolFollowsEnded) Q.counts[7] += Q.counts[1]; // k_finish handed out one pool sample per ended slot
		Q.counts[0] = allActive, Q.counts[1] = 0; // allActive: the number of slots when all are ACTIVE and no queue is built (0 otherwise)
		for (int h = 0; h < RT_HEADS; h++) Q.heads[h * RT_HEAD_STRIDE] = 0;
	}
	if (which & 2) {
		Q.counts[2] = 0, Q.counts[8] = 0;
		for (int h = RT_HEADS; h < 2 * RT_HEADS; h++) Q.heads[h * RT_HEAD_STRIDE] = 0;
	}
	if (which & 4) // connect's work heads only: the leftover launch walks its own list with them
		for (int h = RT_HEADS; h < 2 * RT_HEADS; h++) Q.heads[h * RT_HEAD_STRIDE] = 0;
}

// Path-state accesses of the traversal kernels go through these two (non-temporal variants were measured and lost: a refill
// takes ~20 consecutive queue entries, so a ray's line is used by two or three refills and must stay cached;
// profiles/r02_sweep_nontemporal.txt, profiles/patches/measurement_builds.diff).
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void st_stream(T* p, const T& v) { *p = v; }

// extend: Scene::FindNearest for every active slot.  t_min is Trace's 1e-6 or Sample's 0.001
// (renderer.cpp:24, :131); it applies to lights and brute-force primitives, the BVH uses 0.0001.
template <bool DENSE> // DENSE: every slot is active and the queue is the identity (round 0 with a slot per sample)
struct ExtendPolicy {
	const DScene& S;
	PathState& P;
	const uint* queue;
	int parity;
	int* flag;
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef& head) const
	{
		const int slot = DENSE ? work : (int)ld_stream(queue + work);
		RT_CHECK(slot >= 0 && slot < P.nSlots, 10, flag);
		const float4 o4 = ld_stream(P.O[parity] + slot), d4 = ld_stream(P.D[parity] + slot);
		O = xyz(o4), D = xyz(d4), tmax = o4.w; // o4.w: ray.t after the head tests made when the ray was created
		unpack_head(__float_as_uint(d4.w), head);
		return true;
	}
	__device__ __forceinline__ int slot_of(int work) const { return DENSE ? work : (int)queue[work]; }
	__device__ __forceinline__ void store(int work, const HitRef& hit, const f3& /*O*/, const f3& /*D*/) const
	{
		// the ray in registers may be the object-space one; the sphere normal needs the world ray
		const int slot = DENSE ? work : (int)ld_stream(queue + work);
		int objIdx, mat;
		f3 normal;
		const PathState& Pc = P;
		const int par = parity;
		resolve_hit_lazy(S, hit, [&](f3& o, f3& d) { o = xyz(ld_stream(Pc.O[par] + slot)), d = xyz(ld_stream(Pc.D[par] + slot)); }, objIdx, mat, normal);
		st_stream(P.hitN[parity] + slot, mk4(normal, hit.t));
		st_stream(P.hitId[parity] + slot, make_int2(objIdx, mat));
	}
};
template <bool COUNT, bool DENSE>
__global__ void __launch_bounds__(RT_BLOCK, RT_EXTEND_WAVES) k_extend(DScene S, PathState P, Queues Q, int parity, float t_min, int tuning, uint* spill, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	ExtendPolicy<DENSE> pol{ S, P, Q.active, parity, &Q.counts[3] };
	trace_persistent<false, COUNT, false>(S, pol, Q.counts[0], Q.heads, t_min, tuning, ldsStack, spill, &Q.counts[3], lc, rays);
	if (COUNT) {
		// the head tests ran where the rays were created: per ray, every light and every brute-force primitive
		lc.light = rays * (uint)S.nLights, lc.brute = S.useTLAS ? rays * (uint)(S.nBruteSph + S.nBrutePla) : 0;
		flush_counters(counters, lc, rays, 0);
	}
}

// shade: everything Trace / Sample do at a hit except the occlusion-dependent direct terms.
// Outcomes per slot: continue with a new ray (next active queue), hand the diffuse direct terms to
// connect + light (shadow queue), or end the segment (done queue).
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
__global__ void __launch_bounds__(RT_BLOCK, RT_SHADE_WAVES) k_shade(DScene S, DCamera C, RenderParams R, PathState P, Queues Q, int parity, int fresh)
{
	const int pout = 1 - parity;
	const int nActive = Q.counts[0];
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nActive; e += gridDim.x * blockDim.x) {
		const int slot = fresh ? e : (int)Q.active[e]; // the slots extend just traced, in slot order (fresh: all of them, no queue)
		RT_CHECK(slot >= 0 && slot < P.nSlots, 12, &Q.counts[3]);
		bool keep = false, wantShadow = false, ended = false;
		{
			const float4 o4 = P.O[parity][slot], d4 = P.D[parity][slot], hn = P.hitN[parity][slot];
			const int2 id = P.hitId[parity][slot];
			// fresh: first segment of a sample whose slot index is its sample (k_generate did not write W, L and E)
			const float4 e4 = fresh ? fresh_energy(C, R, R.sampleFirst + (uint)slot) : P.E[slot];
			const float4 w4 = fresh ? make_float4(1, 1, 1, __int_as_float(start_depth(R))) : P.W[slot];
			const float4 l4 = fresh ? make_float4(0, 0, 0, __uint_as_float(R.sampleFirst + (uint)slot)) : P.L[slot];
			const f3 O = xyz(o4), D = xyz(d4), normal = xyz(hn);
			const float t = hn.w;
			f3 W = xyz(w4), E = xyz(e4), Lsum = xyz(l4);
			const int depth = __float_as_int(w4.w);
			uint seed = __float_as_uint(e4.w);
			const bool path = R.mode != 0;
			const f3 I = O + t * D; // ray.IntersectionPoint()

			bool segmentEnds = true; // no continuation ray unless a material creates one
			f3 nO(0.0f), nD(0.0f), nW(0.0f);
			const int nDepth = depth - 1;
			// a child at depth-1 that cannot trace contributes its terminal value right here:
			// Trace(depth <= 0) = 0 (renderer.cpp:23), Sample(depth < 0) = 0.05 (:129)
			const bool childTraces = path ? (nDepth >= 0) : (nDepth > 0);

			if (id.x == -1) {
				Lsum = Lsum + W * sky_color(S, D); // renderer.cpp:26 / :134
			} else if (id.x >= 11 && id.x < 11 + S.nLights) {
				Lsum = Lsum + W * light_intensity(S.lights[id.x - 11], I, normal, I); // :27-28 / :135-137
			} else {
				const DMaterial m = S.mats[id.y];
				const f3 col(m.col[0], m.col[1], m.col[2]);
				if (m.type == 3) { // GLASS, renderer.cpp:45-80 / :198-233
					const float kr = glass_fresnel(normalize(D), normalize(normal), m.ir);
					const bool outside = dot(D, normal) < 0;
					const f3 bias = 0.0001f * normal;
					const f3 norm = outside ? normal : -normal;
					const float r = !outside ? m.ir : (1 / m.ir);
					if (outside) {
						E.x *= x_expf(m.absorption[0] * -t);
						E.y *= x_expf(m.absorption[1] * -t);
						E.z *= x_expf(m.absorption[2] * -t);
					}
					bool takeRefr, takeRefl;
					if (path) {
						const bool refr = kr < RandomFloat(seed);
						takeRefr = refr, takeRefl = !refr;
					} else {
						takeRefr = kr < 1, takeRefl = true;
					}
					f3 refrO(0.0f), refrD(0.0f), refrW(0.0f), reflO(0.0f), reflD(0.0f), reflW(0.0f);
					if (takeRefr) {
						refrD = normalize(glass_refract(D, norm, r));
						refrO = outside ? I - bias : I + bias;
						const f3 tempCol = col * E;
						refrW = W * (tempCol * (1 - kr));
					}
					if (takeRefl) {
						reflD = normalize(reflect(D, norm));
						reflO = outside ? I + bias : I - bias;
						reflW = W * (col * kr);
					}
					if (!childTraces) {
						if (path) Lsum = Lsum + (takeRefr ? refrW : reflW) * f3(0.05f);
					} else if (takeRefr) {
						if (takeRefl) push_pending(P, slot, reflO, reflD, reflW, E, nDepth, &Q.counts[3]); // refraction first, reflection resumes later
						nO = refrO, nD = refrD, nW = refrW, segmentEnds = false;
					} else {
						nO = reflO, nD = reflD, nW = reflW, segmentEnds = false;
					}
				} else if (m.type == 2) { // METAL, renderer.cpp:81-86 / :192-197, metal::scatter template/scene.h:630-635
					nO = I + normal * 0.001f, nD = reflect(D, normal);
					nW = path ? W * col : W * (col * E);
					if (childTraces) segmentEnds = false;
					else if (path) Lsum = Lsum + nW * f3(0.05f);
				} else { // DIFFUSE, renderer.cpp:87-122 / :156-191
					for (int i = 0; i < S.nLights; i++) {
						const f3 pickedPos = light_position(S.lights[i], !path, seed);
						P.sh[(size_t)i * P.nSlots + slot] = mk4(pickedPos, 0.0f);
					}
					wantShadow = S.nLights > 0;
					if (path) {
						const f3 albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
						const f3 rayToHemi = RandomInHemisphere(seed, normal);
						const f3 cos_i(dot(rayToHemi, normal));
						nO = I, nD = rayToHemi;
						nW = W * ((2 * (col * cos_i)) * albedo); // child coefficient of (direct*INVPI + 2*indirect) * albedo
						if (childTraces) segmentEnds = false;
						else Lsum = Lsum + nW * f3(0.05f);
					}
				}
			}

			// a slot that is finished right here (stored below) is never read again; otherwise E and L go back
			// only if they changed (a fresh slot's L has never been written)
			const bool doneHere = R.finishInline && segmentEnds && !wantShadow;
			if (!doneHere) {
				if (fresh || E.x != e4.x || E.y != e4.y || E.z != e4.z || seed != __float_as_uint(e4.w)) P.E[slot] = mk4(E, __uint_as_float(seed));
				if (fresh || Lsum.x != l4.x || Lsum.y != l4.y || Lsum.z != l4.z) P.L[slot] = mk4(Lsum, l4.w);
			}
			if (wantShadow) {
				// light keeps working on THIS segment: it needs the segment's own weight and whether the
				// segment ends there (the continuation's weight goes to P.W below)
				P.sh[(size_t)S.nLights * P.nSlots + slot] = mk4(W, segmentEnds ? 1.0f : 0.0f);
				P.hitP[slot] = mk4(I, 0.0f);
			}
			if (!segmentEnds) {
				emit_ray(S, P, pout, slot, nO, nD, mode_t_min(R.mode));
				P.W[slot] = mk4(nW, __int_as_float(nDepth));
				keep = true;
			} else if (!wantShadow) {
				ended = true;
				if (R.finishInline) store_sample(R, __float_as_uint(l4.w), Lsum), ended = false; // the slot is simply done
			}
		}
		P.status[slot] = (keep ? ST_ACTIVE : 0) | (wantShadow ? ST_SHADOW : 0) | (wantShadow && !keep ? ST_ENDS_AFTER : 0) | (ended ? ST_ENDED : 0);
	}
}

// connect: Scene::IsOccluded from the hit point towards each sampled light position
// (renderer.cpp:93-99 / :161-165).  Traversal only.  Work item = (shadow-queue entry, light);
// vis[light][slot] = 1 when that light is occluded.
struct ConnectPolicy {
	PathState& P;
	const uint* queue;
	int parity, nLights;
	int* flag;
	__device__ __forceinline__ bool load(int work, f3& O, f3& D, float& tmax, HitRef&) const
	{
		const int slot = (int)ld_stream(queue + work / nLights), li = work % nLights;
		RT_CHECK(slot >= 0 && slot < P.nSlots && li >= 0 && li < nLights, 11, flag);
		const f3 I = xyz(ld_stream(P.hitP + slot)); // O + t * D, as shade computed it
		const f3 pickedPos = xyz(ld_stream(P.sh + ((size_t)li * P.nSlots + slot)));
		f3 lightRayDirection = pickedPos - I;
		const float len2 = dot(lightRayDirection, lightRayDirection);
		lightRayDirection = normalize(lightRayDirection);
		O = I + lightRayDirection * 1e-4f, D = lightRayDirection, tmax = sqrtf(len2);
		return true;
	}
	__device__ __forceinline__ void store(int work, bool occluded) const
	{
		const int slot = (int)ld_stream(queue + work / nLights), li = work % nLights;
		st_stream(P.vis + ((size_t)li * P.nSlots + slot), (unsigned char)(occluded ? 1 : 0));
	}
	uint* leftoverList; int* leftoverCount;
	__device__ __forceinline__ void leftover(int work) const { leftoverList[atomicAdd(leftoverCount, 1)] = (uint)work; }
};
// a list of work items of another policy (the rays a wide walk handed back)
template <class Base>
struct ListedPolicy {
	Base base;
	const uint* list;
	__device__ __forceinline__ bool load(int i, f3& O, f3& D, float& tmax, HitRef& head) const { return base.load((int)list[i], O, D, tmax, head); }
	__device__ __forceinline__ void store(int i, bool occluded) const { base.store((int)list[i], occluded); }
};
#ifndef RT_CONNECT_WAVES
#define RT_CONNECT_WAVES 8
#endif
// WIDE: the 4-wide walk (needs S.wide); LISTED: the work items are Q.leftover[0 .. Q.counts[8])
template <bool COUNT, bool WIDE = false, bool LISTED = false>
__global__ void __launch_bounds__(RT_BLOCK, RT_CONNECT_WAVES) k_connect(DScene S, PathState P, Queues Q, int parity, int tuning, uint* spill, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	ConnectPolicy pol{ P, Q.shadow, parity, S.nLights, &Q.counts[3], Q.leftover, &Q.counts[8] };
	if constexpr (LISTED) {
		ListedPolicy<ConnectPolicy> lp{ pol, Q.leftover };
		trace_persistent<true, COUNT, false, ListedPolicy<ConnectPolicy>>(S, lp, Q.counts[8], Q.heads + RT_HEADS * RT_HEAD_STRIDE, 0.0f, tuning, ldsStack, spill, &Q.counts[3], lc, rays);
	} else
		trace_persistent<true, COUNT, false, ConnectPolicy, false, WIDE ? 4 : 2>(S, pol, Q.counts[2] * S.nLights, Q.heads + RT_HEADS * RT_HEAD_STRIDE, 0.0f, tuning, ldsStack, spill, &Q.counts[3], lc, rays);
	if (COUNT) flush_counters(counters, lc, 0, rays);
}

#ifndef RT_TRAVERSE_WAVES
#define RT_TRAVERSE_WAVES 8 // k_traverse_s (rt_stream.h)
#endif
// light: the direct-light terms of a diffuse hit, in light order.
// Whitted (renderer.cpp:89-105): scatter first (energy changes even when occluded), then the
// occlusion test.  Path (renderer.cpp:158-176): occlusion test first, scatter only when visible.
#ifdef RT_LIGHT_WAVES
#define RT_LIGHT_BOUNDS __launch_bounds__(RT_BLOCK, RT_LIGHT_WAVES)
#else
#define RT_LIGHT_BOUNDS __launch_bounds__(RT_BLOCK)
#endif
__global__ void RT_LIGHT_BOUNDS k_light(DScene S, RenderParams R, PathState P, Queues Q, int parity)
{
	const int nShadow = Q.counts[2];
	for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nShadow; e += gridDim.x * blockDim.x) {
		const int slot = (int)Q.shadow[e]; // the slots connect just tested, in slot order
		RT_CHECK(slot >= 0 && slot < P.nSlots, 13, &Q.counts[3]);
		const unsigned char stBits = P.status[slot];
		{
			const float4 o4 = P.O[parity][slot], d4 = P.D[parity][slot], hn = P.hitN[parity][slot];
			const int2 id = P.hitId[parity][slot];
			const float4 e4 = P.E[slot], l4 = P.L[slot];
			const float4 w4 = P.sh[(size_t)S.nLights * P.nSlots + slot];
			const f3 D = xyz(d4), normal = xyz(hn);
			const f3 I = xyz(o4) + hn.w * D;
			const f3 W = xyz(w4);
			f3 E = xyz(e4), Lsum = xyz(l4);
			const bool path = R.mode != 0;
			const DMaterial m = S.mats[id.y];
			const f3 col(m.col[0], m.col[1], m.col[2]);
			const int depth = path ? 0 : __float_as_int(P.W[slot].w); // Whitted: depth of this segment (a diffuse hit stores no continuation)
			f3 direct(0.0f);
			for (int i = 0; i < S.nLights; i++) {
				const f3 pickedPos = xyz(P.sh[(size_t)i * P.nSlots + slot]);
				const f3 lightRayDirection = normalize(pickedPos - I);
				const bool occluded = P.vis[(size_t)i * P.nSlots + slot] != 0;
				f3 att(0.0f);
				if (!path) att = diffuse_scatter(m, D, lightRayDirection, light_intensity(S.lights[i], I, normal, pickedPos), normal, E);
				if (occluded) continue;
				if (path) att = diffuse_scatter(m, D, lightRayDirection, light_intensity(S.lights[i], I, normal, pickedPos), normal, E);
				if (!path && m.shinieness != 0 && depth - 1 > 0) // renderer.cpp:101-102: a mirror branch per visible light
					push_pending(P, slot, I, reflect(D, normal), W * ((m.shinieness * col) * E), E, depth - 1, &Q.counts[3]);
				direct = direct + (1 - m.shinieness) * col * att * E;
			}
			if (path) {
				const f3 albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
				Lsum = Lsum + W * ((direct * RT_INVPI) * albedo); // direct part of (direct*INVPI + 2*indirect) * albedo
			} else {
				Lsum = Lsum + W * direct;
			}
			const bool endsHere = (stBits & ST_ENDS_AFTER) != 0;
			if (endsHere && R.finishInline) store_sample(R, __float_as_uint(l4.w), Lsum); // finished: E and L are not read again
			else {
				if (E.x != e4.x || E.y != e4.y || E.z != e4.z) P.E[slot] = mk4(E, e4.w);
				if (Lsum.x != l4.x || Lsum.y != l4.y || Lsum.z != l4.z) P.L[slot] = mk4(Lsum, l4.w);
			}
			P.status[slot] = (stBits & ST_ACTIVE) | (endsHere && !R.finishInline ? ST_ENDED : 0);
		}
	}
}

// finish: a slot's segment ended without a continuation ray.  Resume the most recent pending
// Whitted branch if there is one; otherwise the sample is complete: store it (renderer.cpp:270 /
// :279-282, gamma per sample) and pull the next sample from the pool.
__global__ void __launch_bounds__(RT_BLOCK) k_finish(DScene S, DCamera C, RenderParams R, PathState P, Queues Q, int parity)
{
	const uint lane = threadIdx.x & 63;
	const int pout = 1 - parity;
	const uint* queue = Q.ended; // slots with ST_ENDED, in slot order
	int first, last;
	wave_range(Q.counts[1], first, last);
	if (first >= last) return;
	int base = 0;
	if (!P.pendCount) {
		// every entry completes a sample: entry e takes pool sample counts[7] + e; k_round_begin advances
		// counts[7] by the queue length afterwards (no atomics)
		base = Q.counts[7] + first;
	} else {
		// pass 1: how many entries of this wave's range complete a sample (no pending branch to resume)
		int total = 0;
		for (int e0 = first; e0 < last; e0 += 64) {
			const int e = e0 + (int)lane;
			const bool completes = e < last && !(P.pendCount[queue[e]] > 0);
			total += __popcll(__ballot(completes));
		}
		// one atomic per wave: the next 'total' samples of the pool
		if (total > 0) {
			if (lane == 0) base = atomicAdd(&Q.counts[7], total);
			base = __shfl(base, 0);
		}
	}
	// pass 2
	for (int e0 = first; e0 < last; e0 += 64) {
		const int e = e0 + (int)lane;
		const int slot = e < last ? (int)queue[e] : -1;
		RT_CHECK(slot < P.nSlots, 14, &Q.counts[3]);
		unsigned char stBits = slot >= 0 ? P.status[slot] : 0;
		bool completes = false;
		if (slot >= 0) {
			stBits &= ~ST_ENDED;
			int np = P.pendCount ? P.pendCount[slot] : 0;
			if (np > 0) {
				np--;
				const float4* pe = P.pend + ((size_t)slot * RT_PEND_CAP + np) * 4;
				const float4 o = pe[0], d = pe[1], w = pe[2], en = pe[3];
				emit_ray(S, P, pout, slot, xyz(o), xyz(d), mode_t_min(R.mode));
				P.W[slot] = make_float4(w.x, w.y, w.z, o.w);
				P.E[slot] = make_float4(en.x, en.y, en.z, P.E[slot].w);
				P.pendCount[slot] = np;
				stBits |= ST_ACTIVE;
			} else {
				const float4 l4 = P.L[slot];
				const uint sid = __float_as_uint(l4.w);
				store_sample(R, sid, xyz(l4));
				completes = true;
			}
		}
		const unsigned long long m = __ballot(completes);
		if (completes) {
			const uint sidNext = (uint)(base + __popcll(m & ((1ull << lane) - 1)));
			if (sidNext < R.nSamples) {
				start_sample(S, C, R, P, slot, R.sampleFirst + sidNext, pout);
				stBits |= ST_ACTIVE;
			}
		}
		base += __popcll(m);
		if (slot >= 0) P.status[slot] = stBits;
	}
}

// ---- general path-mode kernel -------------------------------------------------------------------
// Renderer::Sample (renderer.cpp:128-236) for scenes the wavefront cannot replay: a diffuse material with
// shinieness != 0 (a recursive Sample INSIDE the light loop, :172-173) or built with raytracer == false
// (diffuse::scatter then draws a hemisphere sample after every visible light, template/scene.h:612-614).
// In both cases random draws interleave with occlusion queries in depth-first order, so one lane runs
// one whole sample: recursion becomes a stack of suspended light loops, radiance is carried forward as
// weights exactly as in the wavefront kernels.  Slow path by design (divergent, scratch-heavy).
struct Suspended { // a DIFFUSE hit whose light loop is waiting for a shiny branch to return
	f3 P, N, rayD, W, E;
	int mat, nextLight, depth;
};
__global__ void __launch_bounds__(RT_BLOCK) k_sample_general(DScene S, DCamera C, RenderParams R, uint* spill, int* overflow)
{
	__shared__ uint ldsStack[RT_STACK_LDS * RT_BLOCK];
	Stack st = make_stack(ldsStack, spill, overflow, RT_STACK_LDS);
	for (uint sid = blockIdx.x * blockDim.x + threadIdx.x; sid < R.nSamples; sid += gridDim.x * blockDim.x) {
		f3 O, D;
		uint seed;
		int depth;
		if (R.customO) {
			O = f3(R.customO[3 * sid], R.customO[3 * sid + 1], R.customO[3 * sid + 2]);
			D = f3(R.customD[3 * sid], R.customD[3 * sid + 1], R.customD[3 * sid + 2]);
			seed = StreamSeed(R.seedBase + sid), depth = R.customDepth;
		} else {
			const uint lp = sid % R.tilePixels, frame = R.frame0 + sid / R.tilePixels;
			const int x = (int)(lp % (uint)C.width), y = R.rowFirst + (int)(lp / (uint)C.width) * R.rowStride;
			seed = StreamSeed(R.seedBase + (uint)(y * C.width + x) + frame * (uint)(C.width * C.height));
			float newX = x + (RandomFloat(seed) * 2 - 1);
			float newY = y + (RandomFloat(seed) * 2 - 1);
			primary_ray(C, (int)newX, (int)newY, O, D);
			depth = 4;
		}
		f3 W(1.0f), E(1.0f), Lsum(0.0f);
		if (R.customO) E = f3(R.customE[0], R.customE[1], R.customE[2]);
		Suspended stack[6];
		int sp = 0;
		bool resume = false;
		Suspended cur;
		while (true) {
			if (!resume) {
				// ---- Sample(ray, depth, energy) from its first line ----
				if (depth < 0) { Lsum = Lsum + W * f3(0.05f); }
				else {
					HitRef hit;
					find_nearest_one(S, O, D, 1e34f, 0.001f, hit, st);
					int objIdx, matId;
					f3 normal;
					resolve_hit(S, hit, O, D, objIdx, matId, normal);
					const f3 I = O + hit.t * D;
					if (objIdx == -1) Lsum = Lsum + W * sky_color(S, D);
					else if (objIdx >= 11 && objIdx < 11 + S.nLights) Lsum = Lsum + W * light_intensity(S.lights[objIdx - 11], I, normal, I);
					else {
						const DMaterial m = S.mats[matId];
						const f3 col(m.col[0], m.col[1], m.col[2]);
						bool survives = true;
						if (R.sceneRt == 1) { // Sample with scene.raytracer set (renderer.cpp:143-153): Russian roulette on the largest colour component
							const double p = col.x > col.y && col.x > col.z ? col.x : col.y > col.z ? col.y : col.z;
							if (depth < 5 || !p) survives = (double)RandomFloat(seed) < p; // (the survivor's f = f * (1 / p) is never read)
						}
						if (!survives) {
							// return totCol (= 0): nothing is added
						} else
						if (m.type == 2) {
							O = I + normal * 0.001f, D = reflect(D, normal), W = W * col, depth--;
							continue;
						} else
						if (m.type == 3) {
							const float kr = glass_fresnel(normalize(D), normalize(normal), m.ir);
							const bool outside = dot(D, normal) < 0;
							const f3 bias = 0.0001f * normal, norm = outside ? normal : -normal;
							const float r = !outside ? m.ir : (1 / m.ir);
							if (outside) {
								E.x *= x_expf(m.absorption[0] * -hit.t);
								E.y *= x_expf(m.absorption[1] * -hit.t);
								E.z *= x_expf(m.absorption[2] * -hit.t);
							}
							if (kr < RandomFloat(seed)) {
								const f3 nd = normalize(glass_refract(D, norm, r));
								O = outside ? I - bias : I + bias, D = nd, W = W * ((col * E) * (1 - kr));
							} else {
								const f3 nd = normalize(reflect(D, norm));
								O = outside ? I + bias : I - bias, D = nd, W = W * (col * kr);
							}
							depth--;
							continue;
						} else {
							cur.P = I, cur.N = normal, cur.rayD = D, cur.W = W, cur.E = E, cur.mat = matId, cur.nextLight = 0, cur.depth = depth;
							resume = true;
							continue;
						}
					}
				}
				// this Sample returned: resume the innermost suspended light loop, or finish
				if (sp == 0) break;
				cur = stack[--sp];
				resume = true;
				continue;
			}
			// ---- the light loop of a DIFFUSE hit (renderer.cpp:158-176), possibly resumed ----
			resume = false;
			const DMaterial m = S.mats[cur.mat];
			const f3 col(m.col[0], m.col[1], m.col[2]), albedo(m.albedo[0], m.albedo[1], m.albedo[2]);
			bool branched = false;
			for (int i = cur.nextLight; i < S.nLights; i++) {
				const f3 pickedPos = light_position(S.lights[i], R.sceneRt == 1, seed); // (a light follows scene.raytracer: its position itself when the flag is set, template/scene.h:133)
				f3 L = pickedPos - cur.P;
				const float len2 = dot(L, L);
				L = normalize(L);
				if (is_occluded_one(S, cur.P + L * 1e-4f, L, sqrtf(len2), st)) continue;
				const f3 att = diffuse_scatter(m, cur.rayD, L, light_intensity(S.lights[i], cur.P, cur.N, pickedPos), cur.N, cur.E);
				if (!m.raytracer) (void)RandomInHemisphere(seed, cur.N); // diffuse::scatter's own draw (template/scene.h:612-614)
				Lsum = Lsum + cur.W * ((((1 - m.shinieness) * col * att * cur.E) * RT_INVPI) * albedo);
				if (m.shinieness != 0) {
					// directLightning += shinieness * col * Sample(mirror ray, depth - 1, energy): run it now
					// (its draws come before the next light's), keep the rest of this loop for later
					cur.nextLight = i + 1;
					if (sp < 6) stack[sp++] = cur; else *overflow = 2;
					O = cur.P, D = reflect(cur.rayD, cur.N), E = cur.E, depth = cur.depth - 1;
					W = cur.W * (((m.shinieness * col) * RT_INVPI) * albedo);
					branched = true;
					break;
				}
			}
			if (branched) continue;
			// indirect term: one hemisphere sample (renderer.cpp:181-186)
			const f3 rayToHemi = RandomInHemisphere(seed, cur.N);
			const f3 cos_i(dot(rayToHemi, cur.N));
			O = cur.P, D = rayToHemi, E = cur.E, depth = cur.depth - 1;
			W = cur.W * ((2 * (col * cos_i)) * albedo);
		}
		if (R.customOut) R.customOut[sid] = mk4(Lsum, 0.0f);
		else store_sample(R, sid, Lsum);
	}
}

// ---- general Whitted kernel: Renderer::Trace with scene.raytracer == false (renderer.cpp:21-126 with :33-43 and :107-121 taken) ----
// Unreachable from Tick (which calls Trace only while the flag is set), reachable through the preserved Renderer::Trace.  With the
// flag clear Trace draws random numbers -- Russian roulette at every surface hit, the lights' sampled positions, diffuse::scatter's
// hemisphere sample, whose LAST value becomes the direction of an extra indirect child -- interleaved with occlusion queries and
// child calls in depth-first order, so one lane runs one whole call tree: recursion becomes a stack of frames (a child still to be
// called, or a light loop to resume), radiance is carried forward as weights like everywhere else.
struct TraceFrame {
	f3 P, N, rayD, W, E, sdir; // kind 0: a child Trace(P, rayD) still to be called with weight W and energy E; kind 1: a suspended light loop
	int mat, nextLight, depth, kind;
};
#define RT_TRACE_FRAMES 12
__global__ void __launch_bounds__(RT_BLOCK) k_trace_general(DScene S, DCamera C, RenderParams R, uint* spill, int* overflow)
{
	__shared__ uint ldsStack[RT_STACK_LDS * RT_BLOCK];
	Stack st = make_stack(ldsStack, spill, overflow, RT_STACK_LDS);
	for (uint sid = blockIdx.x * blockDim.x + threadIdx.x; sid < R.nSamples; sid += gridDim.x * blockDim.x) {
		f3 O(R.customO[3 * sid], R.customO[3 * sid + 1], R.customO[3 * sid + 2]), D(R.customD[3 * sid], R.customD[3 * sid + 1], R.customD[3 * sid + 2]);
		uint seed = StreamSeed(R.seedBase + sid);
		int depth = R.customDepth;
		f3 W(1.0f), E(R.customE[0], R.customE[1], R.customE[2]), Lsum(0.0f);
		TraceFrame stack[RT_TRACE_FRAMES];
		int sp = 0;
		bool resume = false;
		TraceFrame cur;
		while (true) {
			if (!resume) {
				// ---- Trace(ray, depth, energy) from its first line ----
				bool returned = true;
				if (depth > 0) {
					HitRef hit;
					find_nearest_one(S, O, D, 1e34f, (float)1e-6, hit, st);
					int objIdx, matId;
					f3 normal;
					resolve_hit(S, hit, O, D, objIdx, matId, normal);
					const f3 I = O + hit.t * D;
					if (objIdx == -1) Lsum = Lsum + W * sky_color(S, D);
					else if (objIdx >= 11 && objIdx < 11 + S.nLights) Lsum = Lsum + W * light_intensity(S.lights[objIdx - 11], I, normal, I);
					else {
						const DMaterial m = S.mats[matId];
						const f3 col(m.col[0], m.col[1], m.col[2]);
						const double p = col.x > col.y && col.x > col.z ? col.x : col.y > col.z ? col.y : col.z; // :34
						bool survives = true;
						if (depth < 5 || !p) survives = (double)RandomFloat(seed) < p; // :35-40
						if (survives && m.type == 3) { // GLASS :45-80: the refraction child first, the reflection child after it
							const float kr = glass_fresnel(normalize(D), normalize(normal), m.ir);
							const bool outside = dot(D, normal) < 0;
							const f3 bias = 0.0001f * normal, norm = outside ? normal : -normal;
							const float r = !outside ? m.ir : (1 / m.ir);
							if (outside) {
								E.x *= x_expf(m.absorption[0] * -hit.t);
								E.y *= x_expf(m.absorption[1] * -hit.t);
								E.z *= x_expf(m.absorption[2] * -hit.t);
							}
							const f3 reflD = normalize(reflect(D, norm)), reflO = outside ? I + bias : I - bias, reflW = W * (col * kr);
							if (kr < 1) {
								if (sp < RT_TRACE_FRAMES) { TraceFrame& f = stack[sp++]; f.kind = 0, f.P = reflO, f.rayD = reflD, f.W = reflW, f.E = E, f.depth = depth - 1; } else *overflow = 2;
								const f3 refrD = normalize(glass_refract(D, norm, r));
								O = outside ? I - bias : I + bias, D = refrD, W = W * ((col * E) * (1 - kr));
							} else O = reflO, D = reflD, W = reflW;
							depth--, returned = false;
						} else if (survives && m.type == 2) { // METAL :81-86
							O = I + normal * 0.001f, D = reflect(D, normal), W = W * (col * E), depth--, returned = false;
						} else if (survives) { // DIFFUSE :87-122
							cur.kind = 1, cur.P = I, cur.N = normal, cur.rayD = D, cur.W = W, cur.E = E, cur.sdir = f3(0.0f), cur.mat = matId, cur.nextLight = 0, cur.depth = depth;
							resume = true, returned = false;
						}
					}
				}
				if (!returned) continue;
				// this Trace returned: the innermost frame goes on, or the tree is finished
				if (sp == 0) break;
				cur = stack[--sp];
				if (cur.kind == 0) { O = cur.P, D = cur.rayD, W = cur.W, E = cur.E, depth = cur.depth; continue; }
				resume = true;
				continue;
			}
			// ---- the light loop of a DIFFUSE hit (:89-106), possibly resumed, then the indirect child (:107-121) ----
			resume = false;
			const DMaterial m = S.mats[cur.mat];
			const f3 col(m.col[0], m.col[1], m.col[2]);
			bool branched = false;
			for (int i = cur.nextLight; i < S.nLights; i++) {
				const f3 pickedPos = light_position(S.lights[i], false, seed); // the flag is clear: a sampled position
				f3 L = pickedPos - cur.P;
				const float len2 = dot(L, L);
				L = normalize(L);
				const f3 att = diffuse_scatter(m, cur.rayD, L, light_intensity(S.lights[i], cur.P, cur.N, pickedPos), cur.N, cur.E); // before the occlusion test (:97-98)
				if (!m.raytracer) cur.sdir = RandomInHemisphere(seed, cur.N); // diffuse::scatter's own draw: scatteredDir (template/scene.h:612-614)
				if (is_occluded_one(S, cur.P + L * 1e-4f, L, sqrtf(len2), st)) continue;
				Lsum = Lsum + cur.W * ((((1 - m.shinieness) * col * att) * cur.E) * RT_INVPI); // (totCol = totCol * INVPI at :119 scales the whole loop's sum)
				if (m.shinieness != 0) {
					cur.nextLight = i + 1;
					if (sp < RT_TRACE_FRAMES) stack[sp++] = cur; else *overflow = 2;
					O = cur.P, D = reflect(cur.rayD, cur.N), E = cur.E, depth = cur.depth - 1;
					W = cur.W * ((((m.shinieness * col)) * cur.E) * RT_INVPI);
					branched = true;
					break;
				}
			}
			if (branched) continue;
			const f3 cos_i(dot(cur.sdir, f3(1.0f))); // float3(dot(scatteredDir, float3((float)N))), N = 1 (:111)
			O = cur.P, D = cur.sdir, E = cur.E, depth = cur.depth - 1;
			W = cur.W * ((cos_i * 2) * RT_PI);
		}
		R.customOut[sid] = mk4(Lsum, 0.0f);
	}
}

// accumulate: add the finished samples of a batch to the accumulator in frame order
// (renderer.cpp:270: overwrite in Whitted mode; :282: += in path mode)
__global__ void k_accumulate(DCamera C, RenderParams R, int batchFrames)
{
	const uint lp = blockIdx.x * blockDim.x + threadIdx.x;
	if (lp >= R.tilePixels) return;
	const int x = (int)(lp % (uint)C.width), y = R.rowFirst + (int)(lp / (uint)C.width) * R.rowStride;
	const int pixel = y * C.width + x;
	if (R.mode == 0) { R.accum[pixel] = R.samples[lp]; return; }
	float4 a = R.accum[pixel];
	for (int f = 0; f < batchFrames; f++) {
		float4 s = R.samples[(size_t)f * R.tilePixels + lp];
		if (s.w != 0) s = g_exactGamma ? make_float4(x_powf(s.x * 1, RT_GAMMA), x_powf(s.y * 1, RT_GAMMA), x_powf(s.z * 1, RT_GAMMA), 0.0f) : gamma_sample(xyz(s)); // stored raw (R.deferGamma)
		a.x += s.x, a.y += s.y, a.z += s.z, a.w += 0;
	}
	R.accum[pixel] = a;
}

// ---- batch queries -------------------------------------------------------------------------------
struct QueryHit { float t; int objIdx; int mat; float nx, ny, nz; };

struct ArrayRays {
	const float* O3; const float* D3; const float* tmax;
	__device__ __forceinline__ bool load(int i, f3& O, f3& D, float& tm, HitRef&) const
	{
		O = f3(O3[3 * i], O3[3 * i + 1], O3[3 * i + 2]), D = f3(D3[3 * i], D3[3 * i + 1], D3[3 * i + 2]);
		tm = tmax ? tmax[i] : 1e34f;
		return true;
	}
};
struct NearestQueryPolicy : ArrayRays {
	const DScene& S; QueryHit* out;
	__device__ __forceinline__ int slot_of(int i) const { return i; }
	__device__ __forceinline__ NearestQueryPolicy(const DScene& s, const float* o, const float* d, const float* t, QueryHit* q) : ArrayRays{ o, d, t }, S(s), out(q) {}
	__device__ __forceinline__ void store(int i, const HitRef& hit, const f3&, const f3&) const
	{
		f3 O, D;
		float tm;
		HitRef unused;
		load(i, O, D, tm, unused);
		int objIdx, mat;
		f3 normal;
		resolve_hit(S, hit, O, D, objIdx, mat, normal);
		QueryHit q;
		q.t = hit.t, q.objIdx = objIdx, q.mat = mat, q.nx = normal.x, q.ny = normal.y, q.nz = normal.z;
		out[i] = q;
	}
};
struct OccludedQueryPolicy : ArrayRays {
	unsigned char* out;
	uint* leftoverList; int* leftoverCount;
	__device__ __forceinline__ OccludedQueryPolicy(const float* o, const float* d, const float* t, unsigned char* q, uint* ll, int* lc) : ArrayRays{ o, d, t }, out(q), leftoverList(ll), leftoverCount(lc) {}
	__device__ __forceinline__ void store(int i, bool occluded) const { out[i] = occluded ? 1 : 0; }
	__device__ __forceinline__ void leftover(int i) const { leftoverList[atomicAdd(leftoverCount, 1)] = (uint)i; }
};
// Camera::GetPrimaryRay + Scene::FindNearest for every pixel
struct PrimaryPolicy {
	const DScene& S; const DCamera& C; int* objOut; float* tOut;
	__device__ __forceinline__ int slot_of(int i) const { return i; }
	__device__ __forceinline__ bool load(int i, f3& O, f3& D, float& tm, HitRef&) const { primary_ray(C, i % C.width, i / C.width, O, D); tm = 1e34f; return true; }
	__device__ __forceinline__ void store(int i, const HitRef& hit, const f3&, const f3&) const
	{
		f3 O, D;
		float tm;
		HitRef unused;
		load(i, O, D, tm, unused);
		int objIdx, mat;
		f3 normal;
		resolve_hit(S, hit, O, D, objIdx, mat, normal);
		objOut[i] = objIdx, tOut[i] = hit.t;
	}
};

// work[0] is the queue head (zeroed by the host before the launch), work[1] the overflow flag
// HEAD: Scene::FindNearest (lights and brute-force primitives first); !HEAD: the accelerator alone (bvh::Intersect,
// tlas::Intersect, bvhInstance::BIntersect -- the host passes a DScene rooted at what is asked for)
template <bool COUNT, bool HEAD = true>
__global__ void __launch_bounds__(RT_BLOCK) k_query_nearest(DScene S, int n, const float* O3, const float* D3, const float* tmax, float t_min, int tuning,
                                                            QueryHit* out, uint* spill, int* work, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	NearestQueryPolicy pol(S, O3, D3, tmax, out);
	trace_persistent<false, COUNT, HEAD>(S, pol, n, work + 16, t_min, tuning, ldsStack, spill, &work[1], lc, rays);
	if (COUNT) flush_counters(counters, lc, rays, 0);
}

// work[0] unused, work[1] overflow flag, work[2] rays handed back by the wide walk; LISTED: the items are leftover[0 .. work[2])
template <bool COUNT, bool WIDE = false, bool LISTED = false, bool WIDE8 = false>
__global__ void __launch_bounds__(RT_BLOCK) k_query_occluded(DScene S, int n, const float* O3, const float* D3, const float* tmax, int tuning,
                                                             unsigned char* out, uint* spill, int* work, DCounters* counters, uint* leftover)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	OccludedQueryPolicy pol(O3, D3, tmax, out, leftover, &work[2]);
	if constexpr (LISTED) {
		ListedPolicy<OccludedQueryPolicy> lp{ pol, leftover };
		trace_persistent<true, COUNT, false>(S, lp, work[2], work + 16, 0.0f, tuning, ldsStack, spill, &work[1], lc, rays);
	} else
		trace_persistent<true, COUNT, false, OccludedQueryPolicy, false, WIDE8 ? 8 : (WIDE ? 4 : 2)>(S, pol, n, work + 16, 0.0f, tuning, ldsStack, spill, &work[1], lc, rays);
	if (COUNT) flush_counters(counters, lc, 0, rays);
}

template <bool COUNT>
__global__ void __launch_bounds__(RT_BLOCK) k_primary_hits(DScene S, DCamera C, float t_min, int tuning, int* objOut, float* tOut, uint* spill, int* work, DCounters* counters)
{
	__shared__ __attribute__((aligned(16))) uint ldsStack[RT_LDS_WORDS];
	LaneCounters lc;
	lc.clear();
	uint rays = 0;
	PrimaryPolicy pol{ S, C, objOut, tOut };
	trace_persistent<false, COUNT, true>(S, pol, C.width * C.height, work + 16, t_min, tuning, ldsStack, spill, &work[1], lc, rays);
	if (COUNT) flush_counters(counters, lc, rays, 0);
}

// Scene::GetSkyColor (template/scene.h:1312-1327) for n directions
__global__ void k_sky_color(DScene S, int n, const float* D3, float* rgb)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const f3 c = sky_color(S, f3(D3[3 * i], D3[3 * i + 1], D3[3 * i + 2]));
	rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
}

// ---- animation: Scene::SetTime + bvh::Refit on the device ------------------------------------------
// k_animate: one thread per leaf slot of the scene BVH; triangles are rebuilt from their ORIGINAL
// vertices (template/scene.h:1233-1241) and re-derived like Triangle::update (:238-246).
__global__ void k_animate(const float4* __restrict__ orig, float4* prims, int nSlots, float a)
{
	const int slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= nSlots) return;
	const float4 r0 = orig[4 * slot], r1 = orig[4 * slot + 1], r2 = orig[4 * slot + 2], r3 = orig[4 * slot + 3];
	if ((__float_as_int(r3.w) & 3) != RT_KIND_TRI) return;
	f3 v[3] = { xyz(r0), xyz(r1), xyz(r2) };
	for (int k = 0; k < 3; k++) {
		const float sft = a * v[k].y * 0.2f;
		const float x = v[k].x * x_cosf(sft) - v[k].y * x_sinf(sft);
		const float y = v[k].x * x_sinf(sft) + v[k].y * x_cosf(sft);
		v[k] = f3(x, y, v[k].z);
	}
	const f3 N = normalize(cross(v[1] - v[0], v[2] - v[0]));
	const float d = -dot(N, v[0]);
	prims[4 * slot] = mk4(v[0], N.x), prims[4 * slot + 1] = mk4(v[1], N.y), prims[4 * slot + 2] = mk4(v[2], N.z);
	prims[4 * slot + 3] = make_float4(d, r3.y, r3.z, r3.w);
}

// bounds of one leaf: bvh::UpdateNodeBounds (bvh.cpp:67-114) over its slots, in slot order
__device__ __forceinline__ void leaf_bounds(const float4* prims, uint first, f3& lo, f3& hi)
{
	lo = f3(1e30f), hi = f3(-1e30f);
	auto fmin3 = [](const f3& a, const f3& b) { return f3(t_fminf(a.x, b.x), t_fminf(a.y, b.y), t_fminf(a.z, b.z)); };
	auto fmax3 = [](const f3& a, const f3& b) { return f3(t_fmaxf(a.x, b.x), t_fmaxf(a.y, b.y), t_fmaxf(a.z, b.z)); };
	for (uint slot = first;; slot++) {
		const float4 r0 = prims[4 * slot], r1 = prims[4 * slot + 1], r2 = prims[4 * slot + 2], r3 = prims[4 * slot + 3];
		const int kl = __float_as_int(r3.w);
		const int kind = kl & 3;
		if (kind == RT_KIND_TRI) {
			lo = fmin3(lo, xyz(r0)), lo = fmin3(lo, xyz(r1)), lo = fmin3(lo, xyz(r2));
			hi = fmax3(hi, xyz(r0)), hi = fmax3(hi, xyz(r1)), hi = fmax3(hi, xyz(r2));
		} else if (kind == RT_KIND_SPHERE) {
			lo = fmin3(lo, xyz(r0) - f3(r1.y)), hi = fmax3(hi, xyz(r0) + f3(r1.y));
		} else {
			const f3 n = normalize(xyz(r0));
			if (n.x + n.y + n.z == 1 && (n.x == 1 || n.y == 1 || n.z == 1)) {
				f3 slabLo(-1e30f), slabHi(1e30f);
				if (n.x == 1) slabLo.x = 0, slabHi.x = 0; else if (n.y == 1) slabLo.y = 0, slabHi.y = 0; else slabLo.z = 0, slabHi.z = 0;
				lo = fmin3(lo, slabLo), hi = fmax3(hi, slabHi);
			} else { lo = f3(-1e30f), hi = f3(1e30f); return; }
		}
		if (kl & RT_LAST_BIT) return;
	}
}

// k_refit: bvh::Refit (bvh.cpp:556-594).  A pair record holds the boxes of its two children, so the
// records are processed deepest level first; 'order' lists them by level, levelStart[l] .. [l+1].
// One workgroup walks all levels (a tree of 10^4..10^5 nodes has a few dozen), barrier between levels.
__device__ __forceinline__ void refit_record(float4* pairs, const float4* prims, uint record)
{
	float4* rec = pairs + 4 * (size_t)record;
	for (int s = 0; s < 2; s++) {
		const uint lk = __float_as_uint(rec[2 * s].w);
		f3 lo, hi;
		if (lk & RT_LEAF_BIT) leaf_bounds(prims, lk & ~RT_LEAF_BIT, lo, hi);
		else {
			const float4* ch = pairs + 4 * (size_t)lk; // child pair: already refitted (deeper level)
			lo = f3(t_fminf(ch[0].x, ch[2].x), t_fminf(ch[0].y, ch[2].y), t_fminf(ch[0].z, ch[2].z));
			hi = f3(t_fmaxf(ch[1].x, ch[3].x), t_fmaxf(ch[1].y, ch[3].y), t_fmaxf(ch[1].z, ch[3].z));
		}
		rec[2 * s] = mk4(lo, rec[2 * s].w);
		rec[2 * s + 1] = mk4(hi, 0.0f);
	}
}
// the levels [0, nLevels) in one workgroup, deepest first, a barrier between levels (the narrow top of the tree)
__global__ void __launch_bounds__(1024) k_refit(float4* pairs, const float4* prims, const uint* order, const int* levelStart, int nLevels)
{
	for (int l = nLevels - 1; l >= 0; l--) {
		for (int i = levelStart[l] + (int)threadIdx.x; i < levelStart[l + 1]; i += (int)blockDim.x) refit_record(pairs, prims, order[i]);
		__threadfence_block();
		__syncthreads();
	}
}
// one wide level across the whole chip: records order[first .. first + count), one thread each
__global__ void __launch_bounds__(256) k_refit_level(float4* pairs, const float4* prims, const uint* order, int first, int count)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) refit_record(pairs, prims, order[first + i]);
}

// after k_refit: the 4-wide nodes' child boxes are copies of binary node boxes (record src / 2, side src & 1)
__global__ void k_wide_sync(float4* wide, const float4* pairs, int nWide)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nWide * 4) return;
	const int node = i >> 2, j = i & 3;
	float* w = (float*)(wide + 8 * (size_t)node);
	const uint src = __float_as_uint(w[28 + j]);
	if (src == 0xFFFFFFFFu) return;
	const float4* rec = pairs + 4 * (size_t)(src >> 1);
	const float4 lo = rec[2 * (src & 1)], hi = rec[2 * (src & 1) + 1];
	w[0 + j] = lo.x, w[4 + j] = lo.y, w[8 + j] = lo.z, w[12 + j] = hi.x, w[16 + j] = hi.y, w[20 + j] = hi.z;
}

// RGBF32_to_RGB8(accumulator / it) (renderer.cpp:287-290, template/precomp.h:445-448)
__global__ void k_resolve(const float4* accum, int first, int n, int it, uint* out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 a = accum[first + i];
	const float v[3] = { a.x / it, a.y / it, a.z / it };
	uint c[3];
	for (int k = 0; k < 3; k++) {
		float m = std_min(1.0f, v[k]);
		float s = 255.0f * m;
		// (uint) of a negative / NaN float: x86-64 converts through a 64-bit cvttss2si and keeps the low half
		long long q = (s > -9.2e18f && s < 9.2e18f) ? (long long)s : (long long)0x8000000000000000ull;
		c[k] = (uint)q;
	}
	out[i] = (c[0] << 16) + (c[1] << 8) + c[2];
}

} // namespace rtd
