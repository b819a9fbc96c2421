// How many cycles does a SIMD of gfx950 need per wave64 VALU instruction (plain f32, no packing) when several
// waves are resident?  The answer decides what "VALU busy" from the SQ counters means for the traversal kernels
// (SQ_ACTIVE_INST_VALU counts 4 cycles per instruction per wave).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
			if (KIND == 0) { // v_fma_f32
				asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
				             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
			} else if (KIND == 1) { // v_mul_f32 / v_sub_f32 / v_min_f32 / v_cndmask mix (what a slab test issues)
				asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
				             "v_sub_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	float* out;
	CHK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
	hipEvent_t a, b;
	CHK(hipEventCreate(&a));
	CHK(hipEventCreate(&b));
	const int iters = 20000;
	for (int kind = 0; kind < 2; kind++)
		for (int wavesPerSimd : { 1, 2, 4, 7, 8 }) {
			const int blocks = prop.multiProcessorCount * wavesPerSimd; // 256 threads = 4 waves = one per SIMD
			float ms = 0;
			for (int rep = 0; rep < 2; rep++) {
				CHK(hipEventRecord(a));
				if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
				else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
				CHK(hipEventRecord(b));
				CHK(hipEventSynchronize(b));
				CHK(hipEventElapsedTime(&ms, a, b));
			}
			const double instPerSimd = (double)iters * 64 * wavesPerSimd; // wave-instructions issued on one SIMD
			const double cycles = ms * 1e-3 * 2.4e9;
			printf("%s  %d wave(s)/SIMD  %8.3f ms  %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", kind ? "slab mix " : "v_fma_f32", wavesPerSimd, ms, cycles / instPerSimd);
		}
	return 0;
}
