// Microbenchmark: how fast can a CU gather 64-byte records from an L2-resident table when the walk is
// data dependent (next record = link read from the current one), as in BVH traversal?
//   A: one lane per walker, 4 x dwordx4 per record (what k_extend does per pair visit)
//   Q: four lanes per walker, 1 x dwordx4 per lane per record (a quad reads the 64 contiguous bytes together)
//   P: two lanes per walker, 2 x dwordx4 per lane
// Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 7) walkA(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		const float4* p = tab + 4 * (size_t)idx;
		const float4 a = p[0], b = p[1], c = p[2], d = p[3];
		acc += a.x + b.y + c.z + d.x;
		idx = __float_as_uint(a.w) & mask;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256, 7) walkQ(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned j = threadIdx.x & 3;
	unsigned idx = (tid >> 2) * 2654435761u & mask;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		const float4 a = tab[4 * (size_t)idx + j];
		acc += a.x + a.y;
		// link lives in lane 0's w: quad broadcast
		idx = (unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a.w), 0x00 /* quad_perm [0,0,0,0] */, 0xF, 0xF, true) & mask;
	}
	out[tid] = acc;
}
__global__ void __launch_bounds__(256, 7) walkP(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned j = threadIdx.x & 1;
	unsigned idx = (tid >> 1) * 2654435761u & mask;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		const float4 a = tab[4 * (size_t)idx + 2 * j], b = tab[4 * (size_t)idx + 2 * j + 1];
		acc += a.x + b.y;
		idx = (unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a.w), 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true) & mask;
	}
	out[tid] = acc;
}

// Q with N independent walkers per quad (memory-level parallelism per wave)
template <int N>
__global__ void __launch_bounds__(256, 7) walkQN(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned j = threadIdx.x & 3;
	unsigned idx[N];
	for (int k = 0; k < N; k++) idx[k] = ((tid >> 2) * N + k) * 2654435761u & mask;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		float4 a[N];
#pragma unroll
		for (int k = 0; k < N; k++) a[k] = tab[4 * (size_t)idx[k] + j];
#pragma unroll
		for (int k = 0; k < N; k++) {
			acc += a[k].x + a[k].y;
			idx[k] = (unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a[k].w), 0x00, 0xF, 0xF, true) & mask;
		}
	}
	out[tid] = acc;
}
// A with only every KEEP-th lane walking (the others idle, exec-masked): is the cost per instruction or per active lane?
template <int KEEP>
__global__ void __launch_bounds__(256, 7) walkAmask(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	const bool on = (threadIdx.x % KEEP) == 0;
	for (int s = 0; s < steps; s++) {
		if (on) {
			const float4* p = tab + 4 * (size_t)idx;
			const float4 a = p[0], b = p[1], c = p[2], d = p[3];
			acc += a.x + b.y + c.z + d.x;
			idx = __float_as_uint(a.w) & mask;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// A with only the FIRST N lanes of every wave walking (contiguous active lanes): does the texture addresser skip quads (or
// larger groups) that have no enabled lane?  Compare with every 4th lane (the same number of lanes, one in every quad).
template <int N>
__global__ void __launch_bounds__(256, 7) walkAfirst(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	const bool on = (threadIdx.x & 63) < N;
	for (int s = 0; s < steps; s++) {
		if (on) {
			const float4* p = tab + 4 * (size_t)idx;
			const float4 a = p[0], b = p[1], c = p[2], d = p[3];
			acc += a.x + b.y + c.z + d.x;
			idx = __float_as_uint(a.w) & mask;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// ... and with N lanes scattered irregularly over the wave (a traversal's pair step)
template <int N>
__global__ void __launch_bounds__(256, 7) walkAscatter(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	const unsigned l = threadIdx.x & 63;
	const bool on = ((l * 37u + 11u) & 63u) < N; // a permutation of the lanes: N of 64 on, irregularly placed
	for (int s = 0; s < steps; s++) {
		if (on) {
			const float4* p = tab + 4 * (size_t)idx;
			const float4 a = p[0], b = p[1], c = p[2], d = p[3];
			acc += a.x + b.y + c.z + d.x;
			idx = __float_as_uint(a.w) & mask;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// A with a wave-uniform base in SGPRs and a 32-bit byte offset per lane (global_load ... v_off, s[base]): does the
// address form matter to the vector-memory path?
__global__ void __launch_bounds__(256, 7) walkAs(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	const char* base = (const char*)tab;
	for (int s = 0; s < steps; s++) {
		const unsigned off = idx << 6;
		float4 a, b, c, d;
		asm volatile("global_load_dwordx4 %0, %4, %5\n global_load_dwordx4 %1, %4, %5 offset:16\n global_load_dwordx4 %2, %4, %5 offset:32\n global_load_dwordx4 %3, %4, %5 offset:48\n s_waitcnt vmcnt(0)"
		             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(off), "s"(base) : "memory");
		acc += a.x + b.y + c.z + d.x;
		idx = __float_as_uint(a.w) & mask;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// A with cache-policy bits on the loads (sc0: wave scope, sc1: system scope, nt: streaming): does a policy make an L1 miss cheaper?
#define WALK_POLICY(NAME, MOD) \
__global__ void __launch_bounds__(256, 7) NAME(const float4* __restrict__ tab, int steps, unsigned mask, float* out) \
{ \
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask; \
	float acc = 0; \
	const char* base = (const char*)tab; \
	for (int s = 0; s < steps; s++) { \
		const unsigned off = idx << 6; \
		float4 a, b, c, d; \
		asm volatile("global_load_dwordx4 %0, %4, %5 " MOD "\n global_load_dwordx4 %1, %4, %5 offset:16 " MOD "\n global_load_dwordx4 %2, %4, %5 offset:32 " MOD "\n global_load_dwordx4 %3, %4, %5 offset:48 " MOD "\n s_waitcnt vmcnt(0)" \
		             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(off), "s"(base) : "memory"); \
		acc += a.x + b.y + c.z + d.x; \
		idx = __float_as_uint(a.w) & mask; \
	} \
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc; \
}
WALK_POLICY(walkSc0, "sc0")
WALK_POLICY(walkSc1, "sc1")
WALK_POLICY(walkNt, "nt")
WALK_POLICY(walkSc01, "sc0 sc1")
// one lane per walker, ONE dwordx4 per step (16-byte records at a 64-byte stride)
__global__ void __launch_bounds__(256, 7) walkA1(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		const float4 a = tab[4 * (size_t)idx];
		acc += a.x;
		idx = __float_as_uint(a.w) & mask;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// one lane per walker, 8 x dwordx4 per step (128-byte records: a 4-wide node)
__global__ void __launch_bounds__(256, 7) walkA8(const float4* __restrict__ tab, int steps, unsigned mask, float* out)
{
	unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask & ~1u;
	float acc = 0;
	for (int s = 0; s < steps; s++) {
		const float4* p = tab + 4 * (size_t)idx;
		const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
		acc += a.x + b.y + c.z + d.x + e.x + f.y + g.z + h.x;
		idx = __float_as_uint(a.w) & mask & ~1u;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
	const bool only_new = argc > 1; // any argument: only the lane-placement variants
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int blocks = prop.multiProcessorCount * 7, threads = 256, steps = 2000;
	float* out;
	CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
	for (unsigned logR : { 13u, 15u, 17u }) {
		const unsigned R = 1u << logR;
		std::vector<float> h((size_t)R * 16);
		std::vector<unsigned> perm(R);
		for (unsigned i = 0; i < R; i++) perm[i] = i;
		std::mt19937 g(7);
		std::shuffle(perm.begin(), perm.end(), g);
		for (unsigned i = 0; i < R; i++) {
			for (int k = 0; k < 16; k++) h[(size_t)i * 16 + k] = 1.0f;
			unsigned l = perm[i];
			memcpy(&h[(size_t)i * 16 + 3], &l, 4);
			l = perm[(i * 7 + 3) % R];
			memcpy(&h[(size_t)i * 16 + 11], &l, 4);
		}
		float4* tab;
		CHK(hipMalloc(&tab, (size_t)R * 64));
		CHK(hipMemcpy(tab, h.data(), (size_t)R * 64, hipMemcpyHostToDevice));
		hipEvent_t a, b;
		CHK(hipEventCreate(&a));
		CHK(hipEventCreate(&b));
		for (int variant = (only_new ? 14 : 0); variant < 20; variant++) {
			for (int rep = 0; rep < 2; rep++) {
				CHK(hipEventRecord(a));
				if (variant == 0) hipLaunchKernelGGL(walkA, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 1) hipLaunchKernelGGL(walkQ, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 2) hipLaunchKernelGGL(walkP, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 3) hipLaunchKernelGGL(walkQN<2>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 4) hipLaunchKernelGGL(walkQN<4>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 5) hipLaunchKernelGGL(walkA1, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 7) hipLaunchKernelGGL(walkAmask<2>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 8) hipLaunchKernelGGL(walkAmask<4>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 9) hipLaunchKernelGGL(walkAs, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 10) hipLaunchKernelGGL(walkSc0, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 11) hipLaunchKernelGGL(walkSc1, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 12) hipLaunchKernelGGL(walkNt, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 13) hipLaunchKernelGGL(walkSc01, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 14) hipLaunchKernelGGL(walkAfirst<16>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 15) hipLaunchKernelGGL(walkAfirst<32>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 16) hipLaunchKernelGGL(walkAscatter<16>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 17) hipLaunchKernelGGL(walkAscatter<26>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 18) hipLaunchKernelGGL(walkAfirst<26>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 19) hipLaunchKernelGGL(walkAfirst<8>, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				if (variant == 6) hipLaunchKernelGGL(walkA8, dim3(blocks), dim3(threads), 0, 0, tab, steps, R - 1, out);
				CHK(hipEventRecord(b));
				CHK(hipEventSynchronize(b));
				float ms;
				CHK(hipEventElapsedTime(&ms, a, b));
				const double per[20] = { 1, 0.25, 0.5, 0.5, 1, 1, 1, 0.5, 0.25, 1, 1, 1, 1, 1, 16 / 64.0, 32 / 64.0, 16 / 64.0, 26 / 64.0, 26 / 64.0, 8 / 64.0 };
				const char* names[20] = { "A  lane, 4 loads/rec", "Q  quad, 1 load/lane", "P  pair, 2 loads/lane", "Q2 quad, 2 chains", "Q4 quad, 4 chains", "A1 lane, 1 load (16 B)", "A8 lane, 8 loads (128 B)", "A  every 2nd lane only", "A  every 4th lane only", "As lane, 4 loads, saddr+voffset", "As + sc0", "As + sc1", "As + nt", "As + sc0 sc1", "A  first 16 lanes only", "A  first 32 lanes only", "A  16 lanes, scattered", "A  26 lanes, scattered", "A  first 26 lanes only", "A  first 8 lanes only" };
				const double walkers = (double)blocks * threads * per[variant];
				if (rep == 1) printf("table %5u KB  %-26s %8.3f ms  %8.2f G records/s  (%.0f walkers)\n", R * 64 / 1024, names[variant], ms, walkers * steps / ms / 1e6, walkers);
			}
		}
		CHK(hipFree(tab));
	}
	return 0;
}
