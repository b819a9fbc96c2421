// librt_amd.so: implementation of the C ABI in include/rt_amd.h for gfx950.
// This file: context, camera, accumulator access, counters and timers, build / tuning info; the other units of the library
// (scene upload, device builders, the render loops, gather, batch queries, the Q-learning sampler) are the rt_api_*.inc files
// included below -- one translation unit, see rt_ctx.h.
#include "rt_ctx.h"

extern "C" {

int rt_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// "domain:bus:device.function" of a HIP device: what tells two ranks of a multi-process run that they sit on the same GPU
int rt_device_pci_bus_id(int device, char* out, int cap)
{
	if (!out || cap < 16) return fail(nullptr, RT_E_ARG, "rt_device_pci_bus_id: a buffer of at least 16 bytes");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(nullptr, RT_E_NODEVICE, "rt_device_pci_bus_id: device %d of %d", device, n);
	if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess) return fail(nullptr, RT_E_HIP, "hipDeviceGetPCIBusId(%d) failed", device);
	return RT_OK;
}

rt_ctx* rt_create(int device, int width, int height)
{
	if (width <= 0 || height <= 0) { fail(nullptr, RT_E_ARG, "rt_create: bad size %dx%d", width, height); return nullptr; }
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fail(nullptr, RT_E_NODEVICE, "rt_create: no HIP device (this library has no CPU path)"); return nullptr; }
	if (device < 0 || device >= n) { fail(nullptr, RT_E_ARG, "rt_create: device %d of %d", device, n); return nullptr; }
	if (hipSetDevice(device) != hipSuccess) { fail(nullptr, RT_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(nullptr, RT_E_HIP, "hipGetDeviceProperties failed"); return nullptr; }
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fail(nullptr, RT_E_NODEVICE, "rt_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName); return nullptr; }
	rt_ctx* c = new rt_ctx();
	c->device = device, c->width = width, c->height = height;
	memset(&c->S, 0, sizeof(c->S));
	memset(&c->slot.P, 0, sizeof(PathState)), memset(&c->slot.Q, 0, sizeof(Queues));
	read_knobs(c->knobs);
	c->fuseTraversal = c->knobs.fuse, c->useStream = c->knobs.stream, c->decideRays = c->knobs.decide;
	c->useMega = c->knobs.mega, c->megaLpt = c->knobs.megaLpt, c->megaLevels = c->knobs.megaLevels, c->deferGamma = c->knobs.deferGamma;
	memset(&c->M, 0, sizeof(c->M));
	memset(&c->Qt, 0, sizeof(c->Qt));
	c->shadeLds = c->knobs.shadeLds;
	memset(&c->T, 0, sizeof(c->T));
	memset(&c->prof, 0, sizeof(c->prof));
	memset(&c->C, 0, sizeof(c->C));
	bool ok = hipStreamCreate(&c->stream) == hipSuccess;
	ok = ok && hipMalloc((void**)&c->accum, (size_t)width * height * sizeof(float4)) == hipSuccess;
	ok = ok && hipMemset(c->accum, 0, (size_t)width * height * sizeof(float4)) == hipSuccess;
	// launch geometry: enough 256-lane blocks to fill 256 CUs at 8 blocks per CU; queues are drained
	// through shared work heads, so the same grid serves every queue length
	c->gridBlocks = prop.multiProcessorCount * 8;
	{
		// the persistent traversal kernels are launched with exactly the blocks a CU can hold (occupancy calculator, per
		// kernel): a block that had to wait for a slot would find its share of the queue already taken
		const int capBlocks = RT_GRID_CAP; // (measurement builds: -DRT_GRID_CAP=4 halves the resident blocks of every persistent kernel)
		auto resident = [&](const void* fn) { int b = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, fn, RT_BLOCK, 0) != hipSuccess || b < 1) b = 1; if (b > 8) b = 8; if (capBlocks >= 1 && b > capBlocks) b = capBlocks; return b * prop.multiProcessorCount; };
		const int e0 = std::min(resident((const void*)k_extend<false, false>), resident((const void*)k_extend<false, true>)), e1 = std::min(resident((const void*)k_extend<true, false>), resident((const void*)k_extend<true, true>));
		const int c0 = resident((const void*)k_connect<false>), c1 = resident((const void*)k_connect<true>);
		c->gridExtend = e0 < e1 ? e0 : e1, c->gridConnect = c0 < c1 ? c0 : c1;
		c->gridConnectWide = resident((const void*)k_connect<false, true>);
		c->gridLeftover = std::min(resident((const void*)k_connect<false, false, true>), prop.multiProcessorCount); // a short list: one block per CU is plenty
		c->gridExtendS = std::min(resident((const void*)k_extend_s<false>), resident((const void*)k_extend_s<true>));
		c->gridConnectS = std::min(resident((const void*)k_connect_s<false>), resident((const void*)k_connect_s<true>));
		c->gridConnectWideS = resident((const void*)k_connect_s<false, true>);
		c->gridLeftoverS = std::min(resident((const void*)k_connect_s<false, false, true>), prop.multiProcessorCount);
		c->gridConnectWide8S = resident((const void*)k_connect_s<false, false, false, true>);
		c->gridTraverseS = resident((const void*)k_traverse_s);
		c->gridMega = resident((const void*)k_whitted_mega);
		c->gridLevel = resident((const void*)k_whitted_level);
		c->gridShadeS = std::min(resident((const void*)k_shade_s<false>), resident((const void*)k_shade_s<true>));
		c->gridLightS = resident((const void*)k_light_s);
		int q = c->gridConnect < c->gridExtend ? c->gridConnect : c->gridExtend;
		const void* qk[9] = { (const void*)k_query_nearest<false>, (const void*)k_query_nearest<true>, (const void*)k_query_occluded<false>, (const void*)k_query_occluded<true>, (const void*)k_primary_hits<false>, (const void*)k_primary_hits<true>,
		                      (const void*)k_query_occluded<false, true>, (const void*)k_query_occluded<false, false, true>, (const void*)k_query_occluded<false, false, false, true> };
		for (int i = 0; i < 9; i++) { const int r = resident(qk[i]); if (r < q) q = r; }
		c->gridQuery = q;
	}
	ok = ok && hipMalloc((void**)&c->spill, (size_t)(RT_STACK_MAX - RT_STACK_ROWS_MIN) * c->gridBlocks * RT_BLOCK * sizeof(uint)) == hipSuccess;
	ok = ok && hipMalloc((void**)&c->flags, (16 + RT_HEADS * RT_HEAD_STRIDE) * sizeof(int)) == hipSuccess;
	ok = ok && hipMemset(c->flags, 0, (16 + RT_HEADS * RT_HEAD_STRIDE) * sizeof(int)) == hipSuccess;
	ok = ok && hipMalloc((void**)&c->counters, 2 * sizeof(DCounters)) == hipSuccess;
	ok = ok && hipMemset(c->counters, 0, 2 * sizeof(DCounters)) == hipSuccess;
	ok = ok && hipHostMalloc((void**)&c->hostCounts, 16 * sizeof(int)) == hipSuccess;
	ok = ok && hipMalloc((void**)&c->gammaLut, 256 * sizeof(float)) == hipSuccess;
	if (ok) {
		const int exact = c->knobs.exactGamma; // rt_kernels.h gamma_powf
		ok = hipMemcpyToSymbol(HIP_SYMBOL(g_exactGamma), &exact, sizeof(int), 0, hipMemcpyHostToDevice) == hipSuccess;
		c->exactGamma = exact;
		if (exact) c->deferGamma = 1; // the exact form lives where samples are read (k_accumulate), not in the shading kernels
	}
	if (ok) hipLaunchKernelGGL(k_gamma_lut, dim3(1), dim3(256), 0, c->stream, c->gammaLut);
	if (!ok) { fail(nullptr, RT_E_HIP, "rt_create: device allocation failed: %s", hipGetErrorString(hipGetLastError())); rt_destroy(c); return nullptr; }
	// default camera = Camera::Camera (camera.h:10-22) for this aspect
	const float aspect = (float)width / (float)height;
	rt_camera cam;
	memset(&cam, 0, sizeof(cam));
	cam.cam_pos[0] = 0, cam.cam_pos[1] = 1, cam.cam_pos[2] = -2;
	cam.top_left[0] = -aspect, cam.top_left[1] = 2, cam.top_right[0] = aspect, cam.top_right[1] = 2;
	cam.bottom_left[0] = -aspect, cam.view_angle = 0.25f;
	rt_set_camera(c, &cam);
	return c;
}

void rt_destroy(rt_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	prof_collect(c);
	free_pool(c->sceneAllocs);
	free_pool(c->slot.allocs);
	free_pool(c->streamAllocs);
	free_pool(c->megaAllocs);
	free_pool(c->megaOrderAllocs);
	free_pool(c->levelAllocs);
	free_pool(c->qAllocs);
	if (c->streamSide) { (void)hipStreamSynchronize(c->streamSide); (void)hipStreamDestroy(c->streamSide); }
	if (c->streamSideSpill) (void)hipFree(c->streamSideSpill);
	if (c->streamFork) (void)hipEventDestroy(c->streamFork);
	if (c->streamJoin) (void)hipEventDestroy(c->streamJoin);
	if (c->gatherDone) (void)hipEventDestroy(c->gatherDone);
	if (c->gatherReady) (void)hipEventDestroy(c->gatherReady);
	if (c->rowsFree) (void)hipEventDestroy(c->rowsFree);
	for (int k = 0; k < 2; k++) if (c->megaEv[k]) (void)hipEventDestroy(c->megaEv[k]);
	if (c->accum && c->accumOwned) (void)hipFree(c->accum);
	if (c->spill) (void)hipFree(c->spill);
	if (c->samples) (void)hipFree(c->samples);
	if (c->resolveBuf) (void)hipFree(c->resolveBuf);
	if (c->gammaLut) (void)hipFree(c->gammaLut);
	if (c->flags) (void)hipFree(c->flags);
	if (c->counters) (void)hipFree(c->counters);
	if (c->hostCounts) (void)hipHostFree(c->hostCounts);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

int rt_set_scene_raytracer(rt_ctx* c, int flag)
{
	if (!c || flag < -1 || flag > 1) return fail(c, RT_E_ARG, "rt_set_scene_raytracer: flag %d (-1, 0, 1)", flag);
	c->sceneRaytracer = flag;
	return RT_OK;
}

int rt_set_camera(rt_ctx* c, const rt_camera* cam)
{
	if (!c || !cam) return fail(c, RT_E_ARG, "rt_set_camera: null argument");
	memcpy(c->C.camPos, cam->cam_pos, 12), memcpy(c->C.topLeft, cam->top_left, 12);
	memcpy(c->C.topRight, cam->top_right, 12), memcpy(c->C.bottomLeft, cam->bottom_left, 12);
	c->C.fisheye = cam->fisheye, c->C.viewAngle = cam->view_angle, c->C.yAngle = cam->y_angle;
	c->C.width = c->width, c->C.height = c->height;
	c->cameraSet = true;
	return RT_OK;
}

#include "rt_api_upload.inc"
#include "rt_api_build.inc"
#include "rt_api_render.inc"

int rt_clear(rt_ctx* c)
{
	if (!c) return RT_E_ARG;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemsetAsync(c->accum, 0, (size_t)c->width * c->height * sizeof(float4), c->stream));
	return RT_OK;
}

int rt_download_accumulator(rt_ctx* c, int y0, int y1, float* out)
{
	if (!c || !out || y0 < 0 || y1 > c->height || y0 >= y1) return fail(c, RT_E_ARG, "rt_download_accumulator: bad argument");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(out, c->accum + (size_t)y0 * c->width, (size_t)(y1 - y0) * c->width * sizeof(float4), hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_resolve(rt_ctx* c, int iteration, int y0, int y1, uint32_t* rgb8_out)
{
	if (!c || !rgb8_out || y0 < 0 || y1 > c->height || y0 >= y1 || iteration == 0) return fail(c, RT_E_ARG, "rt_resolve: bad argument");
	HIPCHK(c, hipSetDevice(c->device));
	const int n = (y1 - y0) * c->width;
	// one frame-sized pixel buffer per context, kept: Tick resolves every frame
	if (!c->resolveBuf) HIPCHK(c, hipMalloc((void**)&c->resolveBuf, (size_t)c->width * c->height * 4));
	hipLaunchKernelGGL(k_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->accum, y0 * c->width, n, iteration, c->resolveBuf);
	HIPCHK(c, hipMemcpyAsync(rgb8_out, c->resolveBuf, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return RT_OK;
}

void* rt_accumulator_device_ptr(rt_ctx* c) { return c ? c->accum : nullptr; }
int rt_bind_accumulator(rt_ctx* c, void* p)
{
	if (!c || !p) return fail(c, RT_E_ARG, "rt_bind_accumulator: null argument");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (c->accum && c->accumOwned) (void)hipFree(c->accum);
	c->accum = (float4*)p, c->accumOwned = false;
	return RT_OK;
}

#include "rt_api_gather.inc"
#include "rt_api_query.inc"

// ---- measurement ------------------------------------------------------------------------------
int rt_set_counting(rt_ctx* c, int counting)
{
	if (!c) return RT_E_ARG;
	if (counting < 0 || counting > RT_COUNT_EXECUTED) return fail(c, RT_E_ARG, "rt_set_counting: mode %d", counting);
	c->counting = counting;
	return RT_OK;
}
int rt_get_counters_split(rt_ctx* c, rt_counters* nearest, rt_counters* occluded, int reset)
{
	if (!c || !nearest || !occluded) return fail(c, RT_E_ARG, "rt_get_counters: null argument");
	static_assert(sizeof(rt_counters) == sizeof(DCounters), "rt_counters layout");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	DCounters both[2];
	HIPCHK(c, hipMemcpy(both, c->counters, 2 * sizeof(DCounters), hipMemcpyDeviceToHost));
	memcpy(nearest, &both[0], sizeof(DCounters)), memcpy(occluded, &both[1], sizeof(DCounters));
	if (reset) HIPCHK(c, hipMemset(c->counters, 0, 2 * sizeof(DCounters)));
	return RT_OK;
}
int rt_get_counters(rt_ctx* c, rt_counters* out, int reset)
{
	if (!out) return fail(c, RT_E_ARG, "rt_get_counters: null argument");
	rt_counters a, b;
	int rc = rt_get_counters_split(c, &a, &b, reset);
	if (rc != RT_OK) return rc;
	const uint64_t *pa = (const uint64_t*)&a, *pb = (const uint64_t*)&b;
	uint64_t* po = (uint64_t*)out;
	for (int i = 0; i < 8; i++) po[i] = pa[i] + pb[i];
	return RT_OK;
}
int rt_set_profiling(rt_ctx* c, int profiling) { if (!c) return RT_E_ARG; prof_collect(c); c->profiling = profiling != 0; return RT_OK; }
int rt_get_profile(rt_ctx* c, rt_profile* out, int reset)
{
	if (!c || !out) return fail(c, RT_E_ARG, "rt_get_profile: null argument");
	prof_collect(c);
	*out = c->prof;
	if (reset) memset(&c->prof, 0, sizeof(c->prof));
	return RT_OK;
}
#include "rt_api_qlearn.inc"

#ifndef RT_EXTRA_FLAGS
#define RT_EXTRA_FLAGS ""
#endif
#define RT_STR2(x) #x
#define RT_STR(x) RT_STR2(x)
const char* rt_build_info(void)
{
	return "flags=[" RT_EXTRA_FLAGS "] RT_PAIR_REPEAT=" RT_STR(RT_PAIR_REPEAT) " RT_CONNECT_REPEAT=" RT_STR(RT_CONNECT_REPEAT) " RT_HEADS=" RT_STR(RT_HEADS)
	       " RT_HEADS_PROBE=" RT_STR(RT_HEADS_PROBE) " RT_EXTEND_WAVES=" RT_STR(RT_EXTEND_WAVES) " RT_CONNECT_WAVES=" RT_STR(RT_CONNECT_WAVES)
	       " RT_SHADE_S_WAVES=" RT_STR(RT_SHADE_S_WAVES) " RT_STACK_ROWS_MIN=" RT_STR(RT_STACK_ROWS_MIN) " RT_SHORT_QUEUE_RAYS=" RT_STR(RT_SHORT_QUEUE_RAYS)
	       " RT_CHUNK_MIN=" RT_STR(RT_CHUNK_MIN) " RT_LDS_WORDS=" RT_STR(RT_LDS_WORDS);
}
const char* rt_tuning_info(rt_ctx* c)
{
	if (!c) return "";
	char buf[640];
	snprintf(buf, sizeof(buf), "stream=%d decide=%d fuse=%d refill=%d refill_any=%d stepmin=%d stepmin_any=%d stepmin_xform=%d pairagain=%d pairagain_any=%d drain=%d drain_any=%d shade_lds=%d gamma_lut=%d exact_gamma=%d defer_gamma=%d wide=%d wide8=%d mega=%d mega_levels=%d mega_lpt=%d qlearn=%d tlas_lds=%d stack_rows=%d slots=%d",
	         c->useStream, c->decideRays, c->fuseTraversal, RT_REFILL, RT_REFILL_ANY, RT_STEPMIN, RT_STEPMIN_ANY, RT_STEPMIN_XFORM,
	         RT_PAIRAGAIN, RT_PAIRAGAIN_ANY, RT_DRAIN_LANES, RT_DRAIN_LANES_ANY, c->shadeLds, c->S.gammaLut ? 1 : 0, c->exactGamma, c->deferGamma, c->S.wide ? 1 : 0, c->S.wide8 ? 1 : 0, c->useMega, c->megaLevels, c->megaLpt, c->Qt.on,
	         c->S.tlasLds, c->S.stackRows, slot_budget(c));
	c->tuningInfo = buf;
	return c->tuningInfo.c_str();
}

int rt_synchronize(rt_ctx* c)
{
	if (!c) return RT_E_ARG;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return RT_OK;
}

} // extern "C"

