// librt_amd.so: implementation of the C ABI in include/rt_amd.h for gfx950.
// Host part: context, repacking of the reference-shaped scene arrays into the HBM traversal layout
// (rt_scene_dev.h), the round loop of the wavefront pixel loop, batch queries, counters, profiling.
#include "rt_kernels.h"
#include "rt_stream.h"
#include "rt_mega.h"
#include "rt_build.h"
#include "../../include/rt_amd.h"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

using namespace rtd;

#ifndef RT_GRID_CAP
#define RT_GRID_CAP 8
#endif
static std::string g_err;

struct Timer {
	hipEvent_t a = nullptr, b = nullptr;
};

struct rt_ctx {
	int device = 0, width = 0, height = 0;
	hipStream_t stream = nullptr;
	std::string err, tuningInfo;
	// scene
	DScene S;
	bool sceneLoaded = false;
	std::vector<void*> sceneAllocs;
	bool pathUnsupported = false; // shiny or rt==0 diffuse present: path mode runs k_sample_general instead of the wavefront
	std::string pathUnsupportedWhy;
	// animation (rt_set_time): original leaf records of the scene BVH and its pair records by level
	float4* primsOrig = nullptr;
	float4* pairsMut = nullptr; float4* primsMut = nullptr;
	std::vector<uint> blasRoot, blasRootWide, blasRootWide8; int nInstances = 0; // roots for the scoped queries (rt_intersect_scope)
	float4* wideMut = nullptr; int wideNodes = 0; // 4-wide nodes: their boxes follow the pair records after a refit
	uint* refitOrder = nullptr; int* refitLevelStart = nullptr;
	int refitLevels = 0, animSlots = 0;
	std::vector<int> refitLevelHost; // levelStart[] on the host: which levels are wide enough for a launch of their own
	// camera
	DCamera C;
	bool cameraSet = false;
	// accumulator
	float4* accum = nullptr;
	bool accumOwned = true;
	// the slot wavefront of rt_kernels.h (Whitted rounds with RT_MEGA=0, path batches above the slot budget, RT_COUNT_REFERENCE
	// launches, RT_STREAM=0): slots, status bytes, queues; it runs on the context's stream
	struct SlotState {
		PathState P;
		Queues Q;
		int stateSlots = 0, stateLights = -1;
		bool statePend = false, stateWide = false;
		std::vector<void*> allocs;
	};
	SlotState slot;
	int fuseTraversal = -1;  // RT_FUSE: how connect(r) + light(r) share the machine with round r + 1 in the dense pipeline (run_rounds_stream)

	// the dense path-mode pipeline (rt_stream.h): its state, the second stream for connect + light, and whether it is on
	StreamState T;
	std::vector<void*> streamAllocs;
	int streamCap = 0, streamLights = -1;
	bool streamWide = false;
	hipStream_t streamSide = nullptr;
	uint* streamSideSpill = nullptr;
	hipEvent_t streamFork = nullptr, streamJoin = nullptr;
	hipEvent_t gatherDone = nullptr; // rt_gather_rows with this context as the source: its rows have arrived at the destination
	hipEvent_t gatherReady = nullptr; // ... and, recorded on the destination's stream before the push: what the destination had queued is done
	hipEvent_t rowsFree = nullptr;    // this context as a DESTINATION: recorded by rt_gather_begin on its stream, once per frame, before its own share is queued
	int useStream = 1;       // RT_STREAM: 1 the dense pipeline for path batches with an entry per sample (default), 0 the slot pipeline of rt_kernels.h
	// Renderer::Trace as one persistent launch per frame (rt_mega.h)
	MegaState M;
	std::vector<void*> megaAllocs;
	int megaLanes = 0, gridMega = 0, gridLevel = 0;
	// Whitted frames by tree levels (rt_mega.h LevelState): queues, term log
	std::vector<void*> levelAllocs;
	LevelState V;
	size_t levelCap = 0; int levelLevels = 0; size_t levelSamples = 0;
	int megaLevels = 2;      // RT_MEGA_LEVELS: Whitted frames up to RT_LEVEL_SAMPLES_MAX samples run one launch per tree level (1), as one launch (0), or
	                         // as whichever of the two was faster when this context last tried both on a batch of this shape (2, default: the frames
	                         // are identical either way; deep glass trees gain 40 %, shallow scenes lose 10 % to the extra launches)
	struct { unsigned nSamples = 0; int depth = 0; int tried[2] = { 0, 0 }; float ms[2] = { 0, 0 }; int choice = -1; } megaAuto; // [0] single launch, [1] levels
	hipEvent_t megaEv[2] = { nullptr, nullptr };
	// longest first (rt_mega.h): per-sample cost of the last Whitted launch and the tile order made from it
	std::vector<void*> megaOrderAllocs;
	uint* megaCost = nullptr; uint* megaOrder = nullptr; uint* megaHist = nullptr;
	size_t megaCostCap = 0;
	unsigned megaCostSamples = 0, megaCostFirst = 0; // the batch megaCost describes (0 samples: nothing yet)
	int exactGamma = 0;      // RT_EXACT_GAMMA: the gamma of a finished sample as the reference's double-precision pow (rt_kernels.h gamma_powf)
	int deferGamma = 1;      // RT_DEFER_GAMMA: path-mode samples get their gamma in k_accumulate (rt_kernels.h store_sample)
	int megaLpt = 1;         // RT_MEGA_LPT: 0 keeps the multiplicative permutation
	int useMega = 1;         // RT_MEGA: 1 Whitted frames as one launch (default), 0 the wavefront rounds of rt_kernels.h
	// Q-learning guided sampling (rt_qlearn.h)
	QTable Qt;
	std::vector<void*> qAllocs;
	int shadeLds = 1;        // RT_SHADE_LDS: the shading kernels keep the material / brute-force primitive tables in LDS
	int gridShadeS = 0, gridLightS = 0; // resident blocks of the grid-stride shading kernels (a second, partial round of blocks would run at low occupancy)
	int decideRays = 1;      // RT_DECIDE: producers answer rays whose first traversal step leaves nothing to visit (rt_stream.h ray_decided)
	int gridTraverseS = 0;
	int gridExtendS = 0, gridConnectS = 0, gridConnectWideS = 0, gridLeftoverS = 0, gridConnectWide8S = 0;
	// traversal stack spill of the context's stream (rounds and batch queries) + flags
	uint* spill = nullptr;
	int gridBlocks = 0;
	std::vector<int> matTypes; // material types of the uploaded scene (measurement builds)
	int gridExtend = 0, gridConnect = 0, gridQuery = 0, gridConnectWide = 0, gridLeftover = 0; // resident blocks of the persistent traversal kernels (every wave owns a first chunk: none may wait for a slot)
	float* gammaLut = nullptr; // DScene::gammaLut
	int* flags = nullptr; // [0] overflow for batch queries
	DCounters* counters = nullptr;
	int counting = 0; // 0 off, 1 the reference's walk (RT_COUNT_REFERENCE), 2 the walk the timed kernels make (RT_COUNT_EXECUTED)
	bool profiling = false;
	rt_profile prof;
	std::vector<Timer> timers; // pending event pairs, resolved lazily
	std::vector<int> timerKind;
	int* hostCounts = nullptr; // pinned
	uint* resolveBuf = nullptr; // rt_resolve's device pixels (width * height), allocated on first use
	float4* samples = nullptr; // finished samples of the current batch, [frame][tile pixel]
	size_t sampleCap = 0;
};

static int fail(rt_ctx* c, int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (c) c->err = buf; else g_err = buf;
	return code;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, RT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

template <typename T>
static hipError_t dalloc(std::vector<void*>& pool, T** p, size_t count)
{
	void* q = nullptr;
	hipError_t e = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
	if (e == hipSuccess) { pool.push_back(q); *p = (T*)q; }
	return e;
}
static void free_pool(std::vector<void*>& pool)
{
	for (void* p : pool) (void)hipFree(p);
	pool.clear();
}

static int tuning(const rt_ctx* c) { return c->counting == RT_COUNT_EXECUTED ? RT_TUNE_CULL_COUNTED : 0; } // the traversal kernels' one launch-time flag (the thresholds are constants: rt_scene_dev.h)

// ---- profiling helpers ---------------------------------------------------------------------
enum { K_GENERATE = 0, K_EXTEND, K_SHADE, K_CONNECT, K_QUERY };
static void prof_begin(rt_ctx* c, int kind, hipStream_t stream = nullptr)
{
	if (!c->profiling) return;
	Timer t;
	(void)hipEventCreate(&t.a);
	(void)hipEventCreate(&t.b);
	(void)hipEventRecord(t.a, stream ? stream : c->stream);
	c->timers.push_back(t);
	c->timerKind.push_back(kind);
}
static void prof_end(rt_ctx* c, hipStream_t stream = nullptr)
{
	if (!c->profiling) return;
	(void)hipEventRecord(c->timers.back().b, stream ? stream : c->stream);
}
static void prof_collect(rt_ctx* c)
{
	if (c->timers.empty()) return;
	(void)hipStreamSynchronize(c->stream);
	rt_kernel_time* slot[5] = { &c->prof.generate, &c->prof.extend, &c->prof.shade, &c->prof.connect, &c->prof.query };
	for (size_t i = 0; i < c->timers.size(); i++) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, c->timers[i].a, c->timers[i].b);
		slot[c->timerKind[i]]->launches++;
		slot[c->timerKind[i]]->ms += ms;
		(void)hipEventDestroy(c->timers[i].a);
		(void)hipEventDestroy(c->timers[i].b);
	}
	c->timers.clear();
	c->timerKind.clear();
}

#ifdef RT_TAIL_PROBE
// measurement build only: print, per traversal launch of the plain round loop (RT_FUSE=0), how long it ran with work left in
// its queue and how long its drain was
static void tail_probe_reset(hipStream_t st)
{
	const unsigned long long init[4] = { ~0ull, ~0ull, 0ull, 0ull };
	(void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tailProbe), init, sizeof(init), 0, hipMemcpyHostToDevice, st);
}
static void tail_probe_print(hipStream_t st, const char* what, int round)
{
	unsigned long long v[4];
	(void)hipStreamSynchronize(st);
	(void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tailProbe), sizeof(v), 0, hipMemcpyDeviceToHost);
	fprintf(stderr, "tail probe %s round %d: busy %.1f us, drain %.1f us\n", what, round, (double)(v[1] - v[0]) / 100.0, (double)(v[2] - v[1]) / 100.0);
}
#endif

#ifdef RT_STEP_COUNT
// measurement build only: steps per nearest-hit ray of every extend launch of the plain round loop, by the material
// type of the hit the ray left from (previous round's hit record) -- which rays are the long ones?
static uint* g_stepBuf = nullptr;
static size_t g_stepCap = 0;
static void step_count_begin(rt_ctx* c, hipStream_t st, int nSlots)
{
	if (g_stepCap < (size_t)nSlots) { if (g_stepBuf) (void)hipFree(g_stepBuf); (void)hipMalloc((void**)&g_stepBuf, (size_t)nSlots * 4); g_stepCap = (size_t)nSlots; }
	(void)hipMemsetAsync(g_stepBuf, 0xFF, (size_t)nSlots * 4, st);
	(void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_stepOut), &g_stepBuf, sizeof(g_stepBuf), 0, hipMemcpyHostToDevice, st);
}
static void step_count_print(rt_ctx* c, hipStream_t st, const PathState& P, int parity, int round, const std::vector<int>& matType)
{
	(void)hipStreamSynchronize(st);
	const int n = P.nSlots;
	std::vector<uint> steps((size_t)n);
	std::vector<int2> prevHit((size_t)n);
	(void)hipMemcpy(steps.data(), g_stepBuf, (size_t)n * 4, hipMemcpyDeviceToHost);
	(void)hipMemcpy(prevHit.data(), P.hitId[1 - parity], (size_t)n * 8, hipMemcpyDeviceToHost);
	std::vector<uint> v;
	double sum = 0;
	unsigned long long byType[8] = { 0 }, longByType[8] = { 0 }, stepsByType[8] = { 0 }, byEnt[12] = { 0 }, longByEnt[12] = { 0 }, stepsByEnt[12] = { 0 };
	for (int i = 0; i < n; i++) {
		if (steps[(size_t)i] == 0xFFFFFFFFu) continue;
		{ const uint e = std::min(11u, steps[(size_t)i] >> 16); steps[(size_t)i] &= 0xFFFFu; byEnt[e]++, stepsByEnt[e] += steps[(size_t)i]; if (steps[(size_t)i] > 150) longByEnt[e]++; }
		v.push_back(steps[(size_t)i]);
		sum += steps[(size_t)i];
		int t = 7; // 7: no previous hit record (round 0)
		if (round > 0) { const int m = prevHit[(size_t)i].y; t = m >= 0 && m < (int)matType.size() ? matType[(size_t)m] & 3 : 6; }
		byType[t]++, stepsByType[t] += steps[(size_t)i];
		if (steps[(size_t)i] > 150) longByType[t]++;
	}
	if (v.empty()) return;
	std::sort(v.begin(), v.end());
	auto q = [&](double f) { return v[(size_t)std::min<double>((double)v.size() - 1, f * (double)v.size())]; };
	fprintf(stderr, "step count extend round %d: %zu rays, mean %.1f, p50 %u p90 %u p99 %u p99.9 %u p99.99 %u max %u\n", round, v.size(), sum / (double)v.size(), q(0.5), q(0.9), q(0.99), q(0.999), q(0.9999), v.back());
	for (int e = 0; e < 12; e++)
		if (byEnt[e]) fprintf(stderr, "   %d instance entries: %llu rays (%.2f %%), mean %.1f steps, %llu with > 150 steps (%.2f %% of them)\n", e, byEnt[e], 100.0 * (double)byEnt[e] / (double)v.size(), (double)stepsByEnt[e] / (double)byEnt[e], longByEnt[e], 100.0 * (double)longByEnt[e] / (double)byEnt[e]);
	for (int t = 0; t < 8; t++)
		if (byType[t]) fprintf(stderr, "   left a surface of type %d: %llu rays (%.1f %%), mean %.1f steps, %llu with > 150 steps (%.2f %% of them)\n", t, byType[t], 100.0 * (double)byType[t] / (double)v.size(), (double)stepsByType[t] / (double)byType[t], longByType[t], 100.0 * (double)longByType[t] / (double)byType[t]);
}
#endif
#ifdef RT_SECTION_PROBE
// measurement build only: per traversal launch of the plain round loop, the waves' cycles by section (rt_scene_dev.h)
static void section_probe_reset(hipStream_t st)
{
	const unsigned long long zero[16] = { 0 };
	(void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sectionProbe), zero, sizeof(zero), 0, hipMemcpyHostToDevice, st);
}
static void section_probe_print(hipStream_t st, const char* what, int round)
{
	unsigned long long v[16];
	(void)hipStreamSynchronize(st);
	(void)hipMemcpyFromSymbol(v, HIP_SYMBOL(g_sectionProbe), sizeof(v), 0, hipMemcpyDeviceToHost);
	const double tot = (double)v[7] > 0 ? (double)v[7] : 1;
	fprintf(stderr, "section probe %s round %d: refill %.1f%% (%llu, %.0f cyc)  pair wait %.1f%% rest %.1f%% (%llu steps, %.0f + %.0f cyc)  leaf wait %.1f%% rest %.1f%% (%llu, %.0f + %.0f)  enter %.1f%% (%llu, %.0f)  exit %.1f%% (%llu, %.0f)  other %.1f%%  iterations %llu (%.0f cyc)\n",
	        what, round, 100 * v[0] / tot, v[13], v[13] ? (double)v[0] / v[13] : 0, 100 * v[1] / tot, 100 * v[2] / tot, v[9], v[9] ? (double)v[1] / v[9] : 0, v[9] ? (double)v[2] / v[9] : 0,
	        100 * v[3] / tot, 100 * v[4] / tot, v[10], v[10] ? (double)v[3] / v[10] : 0, v[10] ? (double)v[4] / v[10] : 0, 100 * v[5] / tot, v[11], v[11] ? (double)v[5] / v[11] : 0,
	        100 * v[6] / tot, v[12], v[12] ? (double)v[6] / v[12] : 0, 100 * (tot - v[0] - v[1] - v[2] - v[3] - v[4] - v[5] - v[6]) / tot, v[8], v[8] ? tot / v[8] : 0);
}
#endif

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
	if (getenv("RT_FUSE")) { const int f = atoi(getenv("RT_FUSE")); c->fuseTraversal = f < 0 ? -1 : (f > 2 ? 2 : f); } // 0 one kernel at a time, 1 one traversal launch per round, 2 connect + light on a second stream; negative: the default by batch size
	if (getenv("RT_STREAM")) c->useStream = atoi(getenv("RT_STREAM")) != 0;
	{
		// the scheduling thresholds are compile-time constants since round 4 (rt_scene_dev.h): a sweep script that still sets them in
		// the environment would read as a flat sweep -- say so once
		static bool warned = false;
		const char* retired[] = { "RT_REFILL", "RT_REFILL_ANY", "RT_STEPMIN", "RT_STEPMIN_ANY", "RT_STEPMIN_XFORM", "RT_PAIRAGAIN", "RT_PAIRAGAIN_ANY", "RT_DRAIN_LANES", "RT_DRAIN_LANES_ANY" };
		for (const char* name : retired)
			if (!warned && getenv(name)) { fprintf(stderr, "rt_amd: %s is a compile-time constant (rebuild with make EXTRA=-D%s=N); the environment variable is ignored\n", name, name); warned = true; }
	}
	if (getenv("RT_DECIDE")) c->decideRays = atoi(getenv("RT_DECIDE")) & 3; // 0 off, 1 on, 2 / 3 on, but generate leaves the finished camera samples to the first shade
	if (getenv("RT_MEGA")) c->useMega = atoi(getenv("RT_MEGA")) != 0;
	if (getenv("RT_MEGA_LPT")) c->megaLpt = atoi(getenv("RT_MEGA_LPT")) != 0;
	if (getenv("RT_MEGA_LEVELS")) c->megaLevels = atoi(getenv("RT_MEGA_LEVELS"));
	if (getenv("RT_DEFER_GAMMA")) c->deferGamma = atoi(getenv("RT_DEFER_GAMMA")) != 0;
	memset(&c->M, 0, sizeof(c->M));
	memset(&c->Qt, 0, sizeof(c->Qt));
	if (getenv("RT_SHADE_LDS")) c->shadeLds = atoi(getenv("RT_SHADE_LDS")) != 0;
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
		const int exact = getenv("RT_EXACT_GAMMA") && atoi(getenv("RT_EXACT_GAMMA")) != 0 ? 1 : 0; // rt_kernels.h gamma_powf
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

// ---- scene upload: reference-shaped arrays -> HBM traversal layout ---------------------------------
// matTypes: rt_material::type per material (null: none known); the type of the primitive's material rides in bits 4-5 of the
// record's kind word, so that the kernel that resolves a hit knows what the hit means for the path without a second fetch
static void pack_prim(float* rec, const rt_blas& b, uint p, bool last, const rt_material* mats = nullptr, uint nMats = 0)
{
	memset(rec, 0, 64);
	int kind, obj, mat;
	if (p < b.n_tri) {
		const rt_triangle& t = b.tris[p];
		rec[0] = t.v0[0], rec[1] = t.v0[1], rec[2] = t.v0[2], rec[3] = t.N[0];
		rec[4] = t.v1[0], rec[5] = t.v1[1], rec[6] = t.v1[2], rec[7] = t.N[1];
		rec[8] = t.v2[0], rec[9] = t.v2[1], rec[10] = t.v2[2], rec[11] = t.N[2];
		// float d = -dot(N, v0) (template/scene.h:193), same expression tree as the per-ray evaluation
		volatile float p0 = t.N[0] * t.v0[0], p1 = t.N[1] * t.v0[1], p2 = t.N[2] * t.v0[2];
		volatile float s01 = p0 + p1;
		volatile float s = s01 + p2;
		rec[12] = -s;
		kind = RT_KIND_TRI, obj = t.obj_idx, mat = t.material;
	} else if (p < b.n_tri + b.n_sph) {
		const rt_sphere& s = b.spheres[p - b.n_tri];
		rec[0] = s.pos[0], rec[1] = s.pos[1], rec[2] = s.pos[2], rec[3] = s.r2;
		rec[4] = s.invr, rec[5] = s.r;
		kind = RT_KIND_SPHERE, obj = s.obj_idx, mat = s.material;
	} else {
		const rt_plane& q = b.planes[p - b.n_tri - b.n_sph];
		rec[0] = q.N[0], rec[1] = q.N[1], rec[2] = q.N[2], rec[3] = q.d;
		kind = RT_KIND_PLANE, obj = q.obj_idx, mat = q.material;
	}
	int kl = kind | (last ? RT_LAST_BIT : 0);
	if (mats && mat >= 0 && (uint)mat < nMats) kl |= (mats[mat].type & 3) << RT_TYPE_SHIFT;
	memcpy(rec + 13, &obj, 4), memcpy(rec + 14, &mat, 4), memcpy(rec + 15, &kl, 4);
}

int rt_upload_scene(rt_ctx* c, const rt_scene_desc* d)
{
	if (!c || !d) return fail(c, RT_E_ARG, "rt_upload_scene: null argument");
	HIPCHK(c, hipSetDevice(c->device));
	if (d->n_lights > RT_MAX_LIGHTS) return fail(c, RT_E_UNSUPPORTED, "rt_upload_scene: %u lights (limit %d)", d->n_lights, RT_MAX_LIGHTS);
	if (d->n_blas < 1 || !d->blas) return fail(c, RT_E_ARG, "rt_upload_scene: no bvh");
	if (d->use_tlas && (d->n_instances < 1 || d->n_instances > 256 || !d->tlas_nodes)) return fail(c, RT_E_UNSUPPORTED, "rt_upload_scene: TLAS needs 1..256 instances (nodeIdx[256], tlas.cpp:16)");
	for (uint i = 0; i < d->n_materials; i++) {
		const rt_material& m = d->materials[i];
		if (m.type < 1 || m.type > 3) return fail(c, RT_E_ARG, "rt_upload_scene: material %u has type %d", i, m.type);
	}
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->megaAuto = {}; // another scene: the Whitted forms are timed again
	free_pool(c->sceneAllocs);
	c->sceneLoaded = false;
	c->pathUnsupported = false;
	for (uint i = 0; i < d->n_materials; i++) {
		const rt_material& m = d->materials[i];
		if (m.type == RT_MAT_DIFFUSE && (m.shinieness != 0 || m.raytracer == 0)) {
			c->pathUnsupported = true;
			c->pathUnsupportedWhy = m.shinieness != 0 ? "a diffuse material has shinieness != 0" : "a diffuse material was built with raytracer == 0";
		}
	}

	// pairs + prims of every BLAS, concatenated
	std::vector<float> pairs, prims;
	std::vector<uint> rootLink(d->n_blas), pairOffOf(d->n_blas);
	for (uint k = 0; k < d->n_blas; k++) {
		const rt_blas& b = d->blas[k];
		if (b.nodes_used < 1 || (b.nodes_used & 1) || !b.nodes) return fail(c, RT_E_ARG, "rt_upload_scene: blas %u has %u nodes (expected an even count >= 2)", k, b.nodes_used);
		if (b.n_prims != b.n_tri + b.n_sph + b.n_pla) return fail(c, RT_E_ARG, "rt_upload_scene: blas %u primitive counts disagree", k);
		// an instanced BLAS is a mesh (bvh(Mesh*), bvh.cpp:5-16): the instance path resolves triangle normals only
		// (bvhInstance.cpp:19), so spheres / planes below an instance are refused rather than shaded wrongly
		if (d->use_tlas && (b.n_sph > 0 || b.n_pla > 0)) return fail(c, RT_E_UNSUPPORTED, "rt_upload_scene: blas %u holds spheres / planes; an instanced BLAS must be triangles only", k);
		// node boxes must be numbers within the reference's +-1e30 sentinels (bvh.cpp:96-109): the hardware min / max
		// slab test of clean rays relies on it
		for (uint i = 0; i < b.nodes_used && b.n_prims > 0; i++) {
			if (i == 1) continue;
			for (int a = 0; a < 3; a++)
				if (!(std::fabs(b.nodes[i].aabb_min[a]) <= 1e30f) || !(std::fabs(b.nodes[i].aabb_max[a]) <= 1e30f))
					return fail(c, RT_E_UNSUPPORTED, "rt_upload_scene: blas %u node %u has a non-finite bound or one beyond 1e30", k, i);
		}
		const uint pairOff = (uint)(pairs.size() / 16), primOff = (uint)(prims.size() / 16);
		pairOffOf[k] = pairOff;
		std::vector<char> last(b.n_prims, 0);
		auto link_of = [&](uint i) -> uint {
			const rt_bvh_node& nd = b.nodes[i];
			if (nd.prim_count > 0) return RT_LEAF_BIT | (primOff + nd.left_first);
			return pairOff + nd.left_first / 2;
		};
		for (uint i = 0; i < b.nodes_used; i++) {
			if (i == 1) continue;
			const rt_bvh_node& nd = b.nodes[i];
			if (nd.prim_count > 0) {
				if (nd.left_first + nd.prim_count > b.n_prims) return fail(c, RT_E_ARG, "rt_upload_scene: blas %u node %u leaf range out of bounds", k, i);
				last[nd.left_first + nd.prim_count - 1] = 1;
			} else if (b.n_prims > 0 && ((nd.left_first & 1) || nd.left_first < 2 || nd.left_first + 1 >= b.nodes_used)) {
				return fail(c, RT_E_ARG, "rt_upload_scene: blas %u node %u child index %u invalid", k, i, nd.left_first);
			}
		}
		rootLink[k] = b.n_prims == 0 ? RT_EMPTY : link_of(0);
		pairs.resize(pairs.size() + (size_t)(b.nodes_used / 2) * 16, 0.0f);
		for (uint i = 2; i + 1 < b.nodes_used + 0u && b.n_prims > 0; i += 2) {
			float* rec = &pairs[(size_t)(pairOff + i / 2) * 16];
			for (int s = 0; s < 2; s++) {
				const rt_bvh_node& nd = b.nodes[i + s];
				uint lk = link_of(i + s);
				memcpy(rec + 8 * s, nd.aabb_min, 12), memcpy(rec + 8 * s + 3, &lk, 4);
				memcpy(rec + 8 * s + 4, nd.aabb_max, 12);
			}
		}
		prims.resize(prims.size() + (size_t)b.n_prims * 16);
		for (uint j = 0; j < b.n_prims; j++) {
			const uint p = b.prim_idx[j];
			if (p >= b.n_prims) return fail(c, RT_E_ARG, "rt_upload_scene: blas %u prim_idx[%u] = %u out of range", k, j, p);
			pack_prim(&prims[(size_t)(primOff + j) * 16], b, p, last[j] != 0, d->materials, d->n_materials);
		}
	}
	// 4-wide nodes (rt_scene_dev.h, wide[]): collapse every BLAS; all or nothing (one flag for the kernels)
	std::vector<float> wide;
	std::vector<uint> rootWide(d->n_blas);
	// Default: on for a scene BVH, off with a TLAS (RT_WIDE=0 / 1 forces).  Exact either way (tests/test_gpu_parity.py::
	// test_wide_walk_*).  Measured (profiles/r02_wide_by_config.txt): one big tree gains (config 4: connect 19.3 -> 16.3 ms
	// per step, config 2: 2.63 -> 2.47), instanced scenes lose or stay level (config 5: 625 -> 660 ms, config 3: 10.1 -> 10.2):
	// there the walk spends its steps on the TLAS level and the first levels of many BLASes, where a 4-wide step
	// replaces fewer binary ones than it costs (four slab tests, a sorting network and up to three pushes).
	bool wideOK = getenv("RT_WIDE") ? atoi(getenv("RT_WIDE")) != 0 : !d->use_tlas;
	{
		size_t primBase = 0;
		for (uint k = 0; k < d->n_blas && wideOK; k++) {
			const rt_blas& b = d->blas[k];
			const uint primOff = (uint)primBase;
			primBase += b.n_prims;
			rootWide[k] = rootLink[k];
			if (b.n_prims == 0 || b.nodes[0].prim_count > 0) continue; // empty, or the root is a leaf
			// nested?  (parents are unions of their children after bvh::Refit, bvh.cpp:556-594)
			for (uint i = 0; i < b.nodes_used && wideOK; i++) {
				if (i == 1 || b.nodes[i].prim_count > 0) continue;
				for (uint ci = b.nodes[i].left_first; ci < b.nodes[i].left_first + 2; ci++)
					for (int a = 0; a < 3; a++)
						if (!(b.nodes[ci].aabb_min[a] >= b.nodes[i].aabb_min[a]) || !(b.nodes[ci].aabb_max[a] <= b.nodes[i].aabb_max[a])) wideOK = false;
			}
			if (!wideOK) break;
			auto area = [&](uint i) { const rt_bvh_node& n = b.nodes[i]; const double ex = (double)n.aabb_max[0] - n.aabb_min[0], ey = (double)n.aabb_max[1] - n.aabb_min[1], ez = (double)n.aabb_max[2] - n.aabb_min[2]; return ex * ey + ey * ez + ez * ex; };
			// depth-first emission; a record is reserved when its binary node is first reached
			std::vector<std::pair<uint, uint>> todo; // (binary inner node, wide record)
			auto reserve = [&]() { const uint w = (uint)(wide.size() / 32); wide.resize(wide.size() + 32, 0.0f); return w; };
			const uint rootRec = reserve();
			rootWide[k] = rootRec;
			todo.push_back({ 0u, rootRec });
			while (!todo.empty()) {
				const uint node = todo.back().first, rec = todo.back().second;
				todo.pop_back();
				uint ch[4];
				int n = 2;
				ch[0] = b.nodes[node].left_first, ch[1] = ch[0] + 1;
				while (n < 4) {
					int best = -1;
					double bestA = -1;
					for (int j = 0; j < n; j++)
						if (b.nodes[ch[j]].prim_count == 0 && area(ch[j]) > bestA) best = j, bestA = area(ch[j]);
					if (best < 0) break;
					const uint lf = b.nodes[ch[best]].left_first;
					for (int j = n; j > best + 1; j--) ch[j] = ch[j - 1];
					ch[best] = lf, ch[best + 1] = lf + 1;
					n++;
				}
				float* r = &wide[(size_t)rec * 32];
				for (int j = 0; j < 4; j++) {
					uint link = RT_EMPTY, src = 0xFFFFFFFFu;
					float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f }; // inverted: never hit
					if (j < n) {
						const rt_bvh_node& c2 = b.nodes[ch[j]];
						memcpy(lo, c2.aabb_min, 12), memcpy(hi, c2.aabb_max, 12);
						src = pairOffOf[k] * 2 + ch[j]; // global binary node number: pair record (src / 2), side (src & 1)
						if (c2.prim_count > 0) link = RT_LEAF_BIT | (primOff + c2.left_first);
						else { link = reserve(); r = &wide[(size_t)rec * 32]; todo.push_back({ ch[j], link }); }
					}
					for (int a = 0; a < 3; a++) r[4 * a + j] = lo[a], r[12 + 4 * a + j] = hi[a];
					memcpy(&r[24 + j], &link, 4), memcpy(&r[28 + j], &src, 4);
				}
			}
		}
		if (!wideOK) wide.clear();
	}
	// 8-wide nodes with quantised child boxes (rt_scene_dev.h, wide8[] / leafBox[]): the same collapse, up to eight
	// children, each child's box rounded outwards onto the node's 8-bit grid.  All or nothing.  RT_WIDE8=1 / 0 forces; default: see below.
	std::vector<uint> wide8;     // 32 uints per node
	std::vector<float> leafBox;  // 8 floats per leaf: {min.xyz, first slot}{max.xyz, -}
	std::vector<uint> rootWide8(d->n_blas);
	bool wide8OK = getenv("RT_WIDE8") ? atoi(getenv("RT_WIDE8")) != 0 : false;
	{
		size_t primBase = 0;
		for (uint k = 0; k < d->n_blas && wide8OK; k++) {
			const rt_blas& b = d->blas[k];
			const uint primOff = (uint)primBase;
			primBase += b.n_prims;
			rootWide8[k] = RT_EMPTY;
			if (b.n_prims == 0) continue;
			// nested and bounded?  (outward rounding needs finite boxes; a plane's +-1e30 slab has no useful grid)
			for (uint i = 0; i < b.nodes_used && wide8OK; i++) {
				if (i == 1) continue;
				for (int a = 0; a < 3; a++)
					if (!(fabsf(b.nodes[i].aabb_min[a]) < 1e29f) || !(fabsf(b.nodes[i].aabb_max[a]) < 1e29f) || !(b.nodes[i].aabb_min[a] <= b.nodes[i].aabb_max[a])) wide8OK = false;
				if (b.nodes[i].prim_count > 0) continue;
				for (uint ci = b.nodes[i].left_first; ci < b.nodes[i].left_first + 2; ci++)
					for (int a = 0; a < 3; a++)
						if (!(b.nodes[ci].aabb_min[a] >= b.nodes[i].aabb_min[a]) || !(b.nodes[ci].aabb_max[a] <= b.nodes[i].aabb_max[a])) wide8OK = false;
			}
			if (!wide8OK) break;
			auto area = [&](uint i) { const rt_bvh_node& n = b.nodes[i]; const double ex = (double)n.aabb_max[0] - n.aabb_min[0], ey = (double)n.aabb_max[1] - n.aabb_min[1], ez = (double)n.aabb_max[2] - n.aabb_min[2]; return ex * ey + ey * ez + ez * ex; };
			auto leaf_link = [&](uint node) {
				const rt_bvh_node& n = b.nodes[node];
				const uint li = (uint)(leafBox.size() / 8);
				const uint slot = primOff + n.left_first;
				float rec[8] = { n.aabb_min[0], n.aabb_min[1], n.aabb_min[2], 0, n.aabb_max[0], n.aabb_max[1], n.aabb_max[2], 0 };
				memcpy(&rec[3], &slot, 4);
				leafBox.insert(leafBox.end(), rec, rec + 8);
				return RT_BOX_BIT | li;
			};
			if (b.nodes[0].prim_count > 0) { rootWide8[k] = leaf_link(0); continue; } // the root is a leaf
			std::vector<std::pair<uint, uint>> todo; // (binary inner node, wide8 record)
			auto reserve = [&]() { const uint w = (uint)(wide8.size() / 32); wide8.resize(wide8.size() + 32, 0u); return w; };
			const uint rootRec = reserve();
			rootWide8[k] = rootRec;
			todo.push_back({ 0u, rootRec });
			while (!todo.empty() && wide8OK) {
				const uint node = todo.back().first, rec = todo.back().second;
				todo.pop_back();
				uint ch[8];
				int n = 2;
				ch[0] = b.nodes[node].left_first, ch[1] = ch[0] + 1;
				while (n < 8) {
					int best = -1;
					double bestA = -1;
					for (int j = 0; j < n; j++)
						if (b.nodes[ch[j]].prim_count == 0 && area(ch[j]) > bestA) best = j, bestA = area(ch[j]);
					if (best < 0) break;
					const uint lf = b.nodes[ch[best]].left_first;
					for (int j = n; j > best + 1; j--) ch[j] = ch[j - 1];
					ch[best] = lf, ch[best + 1] = lf + 1;
					n++;
				}
				uint links[8];
				for (int j = 0; j < 8; j++) {
					links[j] = RT_EMPTY;
					if (j >= n) continue;
					if (b.nodes[ch[j]].prim_count > 0) links[j] = leaf_link(ch[j]);
					else { links[j] = reserve(); todo.push_back({ ch[j], links[j] }); }
				}
				uint* r = &wide8[(size_t)rec * 32];
				unsigned char q[6][8]; // qlo.x qlo.y qlo.z qhi.x qhi.y qhi.z
				uint exps = 0;
				for (int a = 0; a < 3 && wide8OK; a++) {
					float org = b.nodes[ch[0]].aabb_min[a], top = b.nodes[ch[0]].aabb_max[a];
					for (int j = 1; j < n; j++) org = std::min(org, b.nodes[ch[j]].aabb_min[a]), top = std::max(top, b.nodes[ch[j]].aabb_max[a]);
					const double extent = (double)top - (double)org;
					int e = extent > 0 ? (int)std::ceil(std::log2(extent / 255.0)) : -100;
					if (e < -100) e = -100;
					for (bool again = true; again && wide8OK;) {
						again = false;
						if (e > 100) { wide8OK = false; break; }
						const float sc = std::ldexp(1.0f, e);
						for (int j = 0; j < 8 && !again; j++) {
							if (j >= n) { q[a][j] = 255, q[3 + a][j] = 0; continue; } // inverted: never hit (and the link says empty)
							const float lo = b.nodes[ch[j]].aabb_min[a], hi = b.nodes[ch[j]].aabb_max[a];
							int ql = (int)std::floor(((double)lo - (double)org) / (double)sc), qh = (int)std::ceil(((double)hi - (double)org) / (double)sc);
							ql = std::max(0, std::min(255, ql)), qh = std::max(0, std::min(255, qh));
							// the device's own expression must contain the box: fma(q, 2^e, origin)
							while (ql > 0 && std::fmaf((float)ql, sc, org) > lo) ql--;
							while (qh < 255 && std::fmaf((float)qh, sc, org) < hi) qh++;
							if (std::fmaf((float)ql, sc, org) > lo || std::fmaf((float)qh, sc, org) < hi) { e++; again = true; break; } // a coarser grid
							q[a][j] = (unsigned char)ql, q[3 + a][j] = (unsigned char)qh;
						}
					}
					memcpy(&r[a], &org, 4);
					exps |= (uint)(e + 128) << (8 * a);
				}
				r[3] = exps;
				for (int a = 0; a < 6; a++) memcpy(&r[4 + 2 * a], q[a], 8);
				memcpy(&r[16], links, 32);
			}
		}
		if (!wide8OK) wide8.clear(), leafBox.clear();
	}
	// TLAS inner nodes -> pair records appended to the BLAS pairs (children boxes inside the parent's
	// record; child A = leftRight & 0xFFFF, the one tlas::Intersect tests first)
	uint tlasRoot = RT_EMPTY;
	std::vector<uint> tlasSlots; // TLAS node index of every TLAS pair record, in record order
	uint tlasSlotBase = 0;
	if (d->use_tlas) {
		for (uint i = 0; i < d->tlas_nodes_used; i++) {
			const rt_tlas_node& nd = d->tlas_nodes[i];
			if (nd.left_right == 0) { if (nd.blas >= d->n_instances && (i != 0 || d->tlas_nodes_used == 1)) return fail(c, RT_E_ARG, "rt_upload_scene: tlas node %u instance %u out of range", i, nd.blas); }
			else if ((nd.left_right & 0xFFFF) >= d->tlas_nodes_used || (nd.left_right >> 16) >= d->tlas_nodes_used) return fail(c, RT_E_ARG, "rt_upload_scene: tlas node %u child out of range", i);
		}
		for (uint i = 0; i < d->tlas_nodes_used; i++)
			for (int a = 0; a < 3; a++)
				if (!(std::fabs(d->tlas_nodes[i].aabb_min[a]) <= 1e30f) || !(std::fabs(d->tlas_nodes[i].aabb_max[a]) <= 1e30f))
					return fail(c, RT_E_UNSUPPORTED, "rt_upload_scene: tlas node %u has a non-finite bound or one beyond 1e30", i);
		const uint nT = d->tlas_nodes_used;
		std::vector<uint> slotOf(nT, 0);
		uint next = (uint)(pairs.size() / 16);
		tlasSlotBase = next;
		for (uint i = 0; i < nT; i++) if (d->tlas_nodes[i].left_right != 0) slotOf[i] = next++, tlasSlots.push_back(i);
		auto tlink = [&](uint i) -> uint { const rt_tlas_node& nd = d->tlas_nodes[i]; return nd.left_right == 0 ? (RT_INST_BIT | nd.blas) : slotOf[i]; };
		pairs.resize((size_t)next * 16, 0.0f);
		for (uint i = 0; i < nT; i++) {
			const rt_tlas_node& nd = d->tlas_nodes[i];
			if (nd.left_right == 0) continue;
			const uint ch[2] = { nd.left_right & 0xFFFFu, nd.left_right >> 16 };
			float* rec = &pairs[(size_t)slotOf[i] * 16];
			for (int s = 0; s < 2; s++) {
				const rt_tlas_node& cn = d->tlas_nodes[ch[s]];
				const uint lk = tlink(ch[s]);
				memcpy(rec + 8 * s, cn.aabb_min, 12), memcpy(rec + 8 * s + 3, &lk, 4);
				memcpy(rec + 8 * s + 4, cn.aabb_max, 12);
			}
		}
		tlasRoot = tlink(0);
	}
	DScene S;
	memset(&S, 0, sizeof(S));
	float* dp = nullptr;
	HIPCHK(c, dalloc(c->sceneAllocs, &dp, pairs.size() + 16));
	HIPCHK(c, hipMemcpy(dp, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
	S.pairs = (const float4*)dp;

	HIPCHK(c, dalloc(c->sceneAllocs, &dp, prims.size() + 16));
	HIPCHK(c, hipMemcpy(dp, prims.data(), prims.size() * 4, hipMemcpyHostToDevice));
	S.prims = (const float4*)dp;
	S.rootLink = d->use_tlas ? tlasRoot : rootLink[0];
	if (!wide.empty() || wideOK) {
		HIPCHK(c, dalloc(c->sceneAllocs, &dp, wide.size() + 32));
		if (!wide.empty()) HIPCHK(c, hipMemcpy(dp, wide.data(), wide.size() * 4, hipMemcpyHostToDevice));
		S.wide = wideOK ? (const float4*)dp : nullptr;
		S.rootWide = rootWide[0];
		c->wideMut = (float4*)dp, c->wideNodes = (int)(wide.size() / 32);
	} else c->wideMut = nullptr, c->wideNodes = 0;
	S.wide8 = nullptr, S.leafBox = nullptr, S.rootWide8 = RT_EMPTY;
	if (wide8OK && !wide8.empty()) {
		uint* dw = nullptr;
		float* dl = nullptr;
		HIPCHK(c, dalloc(c->sceneAllocs, &dw, wide8.size() + 32));
		HIPCHK(c, hipMemcpy(dw, wide8.data(), wide8.size() * 4, hipMemcpyHostToDevice));
		HIPCHK(c, dalloc(c->sceneAllocs, &dl, leafBox.size() + 8));
		if (!leafBox.empty()) HIPCHK(c, hipMemcpy(dl, leafBox.data(), leafBox.size() * 4, hipMemcpyHostToDevice));
		S.wide8 = (const uint4*)dw, S.leafBox = (const float4*)dl, S.rootWide8 = rootWide8[0];
	}
	c->pairsMut = (float4*)S.pairs, c->primsMut = (float4*)S.prims;
	c->primsOrig = nullptr, c->refitLevels = 0, c->animSlots = 0;
	if (!d->use_tlas && d->blas[0].n_prims > 0) {
		// rt_set_time support: a copy of the uploaded leaf records, and the pair records of the scene
		// BVH grouped by depth (children are deeper than their parents) for the bottom-up refit
		const rt_blas& b = d->blas[0];
		std::vector<uint> order;
		std::vector<int> levelStart(1, 0);
		std::vector<uint> level;
		if (b.nodes[0].prim_count == 0) level.push_back(b.nodes[0].left_first / 2);
		while (!level.empty()) {
			std::vector<uint> next;
			for (uint pr : level) {
				order.push_back(pr);
				for (int sI = 0; sI < 2; sI++) { const rt_bvh_node& nd = b.nodes[2 * pr + sI]; if (nd.prim_count == 0) next.push_back(nd.left_first / 2); }
			}
			levelStart.push_back((int)order.size());
			level.swap(next);
		}
		c->refitLevels = (int)levelStart.size() - 1;
		c->refitLevelHost = levelStart;
		c->animSlots = (int)b.n_prims;
		HIPCHK(c, dalloc(c->sceneAllocs, &c->primsOrig, (size_t)b.n_prims * 4));
		HIPCHK(c, hipMemcpy(c->primsOrig, prims.data(), (size_t)b.n_prims * 64, hipMemcpyHostToDevice));
		HIPCHK(c, dalloc(c->sceneAllocs, &c->refitOrder, order.size() + 1));
		HIPCHK(c, hipMemcpy(c->refitOrder, order.data(), order.size() * 4, hipMemcpyHostToDevice));
		HIPCHK(c, dalloc(c->sceneAllocs, &c->refitLevelStart, levelStart.size()));
		HIPCHK(c, hipMemcpy(c->refitLevelStart, levelStart.data(), levelStart.size() * 4, hipMemcpyHostToDevice));
	}
	S.useTLAS = d->use_tlas ? 1 : 0;
	S.stackRows = RT_STACK_ROWS_MAX;

	if (d->use_tlas) {
		std::vector<DInstance> inst(d->n_instances);
		for (uint i = 0; i < d->n_instances; i++) {
			const rt_instance& in = d->instances[i];
			if (in.blas < 0 || (uint)in.blas >= d->n_blas) return fail(c, RT_E_ARG, "rt_upload_scene: instance %u blas %d out of range", i, in.blas);
			memset(&inst[i], 0, sizeof(DInstance));
			memcpy(inst[i].invT, in.inv_transform, 48), memcpy(inst[i].T, in.transform, 48);
			inst[i].rootLink = rootLink[in.blas];
			inst[i].rootWide = rootWide[in.blas];
			inst[i].rootWide8 = wide8OK ? rootWide8[in.blas] : RT_EMPTY;
		}
		DInstance* di = nullptr;
		HIPCHK(c, dalloc(c->sceneAllocs, &di, inst.size()));
		HIPCHK(c, hipMemcpy(di, inst.data(), inst.size() * sizeof(DInstance), hipMemcpyHostToDevice));
		S.inst = di;
		// brute-force primitives in prim-record form
		rt_blas fake;
		memset(&fake, 0, sizeof(fake));
		fake.spheres = d->brute_spheres, fake.n_sph = d->n_brute_spheres, fake.planes = d->brute_planes, fake.n_pla = d->n_brute_planes;
		const uint nb = fake.n_sph + fake.n_pla;
		std::vector<float> brute((size_t)nb * 16 + 16);
		for (uint j = 0; j < nb; j++) pack_prim(&brute[(size_t)j * 16], fake, j, true, d->materials, d->n_materials);
		HIPCHK(c, dalloc(c->sceneAllocs, &dp, brute.size()));
		HIPCHK(c, hipMemcpy(dp, brute.data(), brute.size() * 4, hipMemcpyHostToDevice));
		S.brute = (const float4*)dp;
		S.nBruteSph = (int)fake.n_sph, S.nBrutePla = (int)fake.n_pla;

		// reach[]: per TLAS pair, the world boxes its children's geometry can occupy (rt_scene_dev.h)
		struct Reach { double lo[3], hi[3], a, b; };
		const double big = 1e30, rel = 1.0 / 32768.0;
		auto unbounded = [&]() { Reach r; for (int k = 0; k < 3; k++) r.lo[k] = -big, r.hi[k] = big; r.a = 0, r.b = 0; return r; };
		std::vector<Reach> instReach(d->n_instances);
		for (uint i = 0; i < d->n_instances; i++) {
			const rt_instance& in = d->instances[i];
			const rt_blas& b = d->blas[in.blas];
			Reach r = unbounded();
			instReach[i] = r;
			// object-space box of everything a ray can hit in this BLAS: triangles and spheres; a plane is
			// unbounded.  Triangle::Intersect rejects a triangle with N == 0 for every ray with finite D
			// (|dot(N, D)| < t_min), and one with a NaN in N for every ray (t is NaN and fails the final
			// range test) -- the two sentinel triangles a .tri mesh ends with (v0 = v1 = v2 = 999) are such.
			if (b.n_prims == 0 || b.n_pla > 0) continue;
			double lo[3] = { big, big, big }, hi[3] = { -big, -big, -big };
			bool finite = true;
			for (uint j = 0; j < b.n_tri; j++) {
				const rt_triangle& t = b.tris[j];
				if ((t.N[0] == 0 && t.N[1] == 0 && t.N[2] == 0) || t.N[0] != t.N[0] || t.N[1] != t.N[1] || t.N[2] != t.N[2]) continue;
				for (int k = 0; k < 3; k++) {
					lo[k] = std::min(lo[k], (double)std::min(t.v0[k], std::min(t.v1[k], t.v2[k])));
					hi[k] = std::max(hi[k], (double)std::max(t.v0[k], std::max(t.v1[k], t.v2[k])));
					if (!(std::fabs(t.v0[k]) < 1e29f && std::fabs(t.v1[k]) < 1e29f && std::fabs(t.v2[k]) < 1e29f)) finite = false;
				}
			}
			for (uint j = 0; j < b.n_sph; j++) {
				const rt_sphere& q = b.spheres[j];
				const double rr = std::max(std::fabs((double)q.r), std::sqrt(std::fabs((double)q.r2))); // r and r2 are separate inputs
				for (int k = 0; k < 3; k++) {
					lo[k] = std::min(lo[k], (double)q.pos[k] - rr), hi[k] = std::max(hi[k], (double)q.pos[k] + rr);
					if (!(std::fabs(q.pos[k]) < 1e29f && rr < 1e29)) finite = false;
				}
			}
			if (lo[0] > hi[0]) { for (int k = 0; k < 3; k++) lo[k] = hi[k] = 0; } // nothing hittable: an empty box at the origin
			if (!finite) continue;
			// exact inverse of the float matrix the device applies (rows 0-2 of invTransform), in double
			const float* m = in.inv_transform;
			const double A[3][3] = { { m[0], m[1], m[2] }, { m[4], m[5], m[6] }, { m[8], m[9], m[10] } }, tv[3] = { m[3], m[7], m[11] };
			const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
			                   A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
			if (!(std::fabs(det) > 1e-30) || !std::isfinite(det)) continue;
			double Ai[3][3];
			for (int rI = 0; rI < 3; rI++)
				for (int cI = 0; cI < 3; cI++) {
					const int r1 = (cI + 1) % 3, r2 = (cI + 2) % 3, c1 = (rI + 1) % 3, c2 = (rI + 2) % 3;
					Ai[rI][cI] = (A[r1][c1] * A[r2][c2] - A[r1][c2] * A[r2][c1]) / det;
				}
			double nA = 0, nAi = 0, lmax = 0, wmax = 0;
			for (int rI = 0; rI < 3; rI++) {
				nA = std::max(nA, std::fabs(A[rI][0]) + std::fabs(A[rI][1]) + std::fabs(A[rI][2]));
				nAi = std::max(nAi, std::fabs(Ai[rI][0]) + std::fabs(Ai[rI][1]) + std::fabs(Ai[rI][2]));
			}
			for (int k = 0; k < 3; k++) r.lo[k] = big, r.hi[k] = -big, lmax = std::max(lmax, std::max(std::fabs(lo[k]), std::fabs(hi[k])) + std::fabs(tv[k]));
			for (int corner = 0; corner < 8; corner++) {
				const double l[3] = { (corner & 1 ? hi[0] : lo[0]) - tv[0], (corner & 2 ? hi[1] : lo[1]) - tv[1], (corner & 4 ? hi[2] : lo[2]) - tv[2] };
				for (int k = 0; k < 3; k++) {
					const double w = Ai[k][0] * l[0] + Ai[k][1] * l[1] + Ai[k][2] * l[2];
					r.lo[k] = std::min(r.lo[k], w), r.hi[k] = std::max(r.hi[k], w);
				}
			}
			for (int k = 0; k < 3; k++) wmax = std::max(wmax, std::max(std::fabs(r.lo[k]), std::fabs(r.hi[k])));
			const double cond = std::max(1.0, nA * nAi);
			r.a = rel * (cond * wmax + nAi * lmax) + 1e-20, r.b = rel * cond;
			if (!(wmax < 1e29) || !std::isfinite(r.a) || !std::isfinite(r.b) || r.a > 1e29 || r.b > 1e10) continue;
			instReach[i] = r;
		}
		const uint nT = d->tlas_nodes_used;
		std::vector<Reach> nodeReach(nT);
		std::vector<char> state(nT, 0); // 0 unseen, 1 open, 2 done
		std::vector<uint> walk;
		if (nT > 0 && d->tlas_nodes[0].left_right != 0) walk.push_back(0);
		while (!walk.empty()) {
			const uint i = walk.back();
			const rt_tlas_node& nd = d->tlas_nodes[i];
			if (nd.left_right == 0) { nodeReach[i] = nd.blas < d->n_instances ? instReach[nd.blas] : unbounded(); state[i] = 2; walk.pop_back(); continue; }
			const uint ch[2] = { nd.left_right & 0xFFFFu, nd.left_right >> 16 };
			if (state[i] == 0) {
				state[i] = 1;
				for (int sI = 0; sI < 2; sI++) {
					if (state[ch[sI]] == 1) return fail(c, RT_E_ARG, "rt_upload_scene: tlas node %u is its own ancestor", ch[sI]);
					if (state[ch[sI]] == 0) walk.push_back(ch[sI]);
				}
				continue;
			}
			Reach r = nodeReach[ch[0]];
			const Reach& o = nodeReach[ch[1]];
			for (int k = 0; k < 3; k++) r.lo[k] = std::min(r.lo[k], o.lo[k]), r.hi[k] = std::max(r.hi[k], o.hi[k]);
			r.a = std::max(r.a, o.a), r.b = std::max(r.b, o.b);
			nodeReach[i] = r, state[i] = 2;
			walk.pop_back();
		}
		const uint tlasBase = (uint)tlasSlots.size() ? tlasSlotBase : 0;
		// the boxes are stored inflated for every world origin with |O|_1 <= originMax: 64 x the extent of
		// the instanced geometry (camera and bounce rays start inside that; a ray from further away is
		// simply not culled)
		double extent = 1.0;
		for (uint i = 0; i < d->n_instances; i++)
			if (instReach[i].b > 0) for (int k = 0; k < 3; k++) extent = std::max(extent, std::max(std::fabs(instReach[i].lo[k]), std::fabs(instReach[i].hi[k])));
		const double originMax = 64.0 * 3.0 * extent;
		std::vector<float> reach((size_t)tlasSlots.size() * 12 + 16, 0.0f);
		for (size_t sI = 0; sI < tlasSlots.size(); sI++) {
			const rt_tlas_node& nd = d->tlas_nodes[tlasSlots[sI]];
			const uint ch[2] = { nd.left_right & 0xFFFFu, nd.left_right >> 16 };
			for (int h = 0; h < 2; h++) {
				const Reach& r = state[ch[h]] == 2 ? nodeReach[ch[h]] : unbounded();
				float* rec = &reach[sI * 12 + 6 * h]; // {min.xyz, max.xyz} of child h
				const double m = r.a + r.b * originMax;
				// round outwards: the float box must contain the inflated double one
				for (int k = 0; k < 3; k++) {
					rec[k] = std::nextafterf((float)std::max(r.lo[k] - m, -big), -INFINITY), rec[3 + k] = std::nextafterf((float)std::min(r.hi[k] + m, big), INFINITY);
				}
			}
		}
		HIPCHK(c, dalloc(c->sceneAllocs, &dp, reach.size()));
		HIPCHK(c, hipMemcpy(dp, reach.data(), reach.size() * 4, hipMemcpyHostToDevice));
		S.reach = (const float4*)dp;
		S.tlasBase = tlasBase;
		S.reachOriginMax = (float)originMax;
		S.tlasPairs = (int)tlasSlots.size();
		S.nInst = (int)d->n_instances;
		// the split of a traversal block's LDS (rt_scene_dev.h RT_LDS_WORDS): as many stack rows as the TLAS copy leaves
		{
			const int words = RT_TLAS_COPY_WORDS(S.tlasPairs, S.nInst);
			const int rows = (RT_LDS_WORDS - ((words + 3) & ~3)) / RT_BLOCK - 6;
			S.tlasLds = rows >= RT_STACK_ROWS_MIN ? 1 : 0;
			if (getenv("RT_TLAS_LDS")) S.tlasLds = S.tlasLds && atoi(getenv("RT_TLAS_LDS")) != 0;
			if (S.tlasLds) S.stackRows = rows < RT_STACK_ROWS_MAX ? rows : RT_STACK_ROWS_MAX;
		}
	}
	std::vector<DLight> lights(d->n_lights ? d->n_lights : 1);
	for (uint i = 0; i < d->n_lights; i++) {
		const rt_light& l = d->lights[i];
		DLight& o = lights[i];
		o.kind = l.kind, o.objIdx = l.obj_idx, o.strength = l.strength, o.radius = l.radius, o.sinAngle = l.sin_angle;
		memcpy(o.pos, l.pos, 12), memcpy(o.col, l.col, 12), memcpy(o.normal, l.normal, 12);
	}
	DLight* dl = nullptr;
	HIPCHK(c, dalloc(c->sceneAllocs, &dl, lights.size()));
	HIPCHK(c, hipMemcpy(dl, lights.data(), lights.size() * sizeof(DLight), hipMemcpyHostToDevice));
	S.lights = dl, S.nLights = (int)d->n_lights;
	std::vector<DMaterial> mats(d->n_materials ? d->n_materials : 1);
	for (uint i = 0; i < d->n_materials; i++) {
		const rt_material& m = d->materials[i];
		DMaterial& o = mats[i];
		o.type = m.type, o.raytracer = m.raytracer, o.specu = m.specu, o.diffu = m.diffu, o.shinieness = m.shinieness, o.N = m.N, o.ir = m.ir;
		memcpy(o.col, m.col, 12), memcpy(o.albedo, m.albedo, 12), memcpy(o.absorption, m.absorption, 12);
	}
	c->matTypes.clear();
	for (size_t i = 0; i < mats.size(); i++) c->matTypes.push_back(mats[i].type);
	DMaterial* dm = nullptr;
	HIPCHK(c, dalloc(c->sceneAllocs, &dm, mats.size()));
	HIPCHK(c, hipMemcpy(dm, mats.data(), mats.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
	S.mats = dm, S.nMats = (int)mats.size();
	S.gammaLut = getenv("RT_GAMMA_LUT") && atoi(getenv("RT_GAMMA_LUT")) == 0 ? nullptr : c->gammaLut;
	if (d->sky_pixels && d->sky_w > 0 && d->sky_h > 0 && d->sky_n >= 3) {
		unsigned char* ds = nullptr;
		const size_t nbytes = (size_t)d->sky_w * d->sky_h * d->sky_n;
		HIPCHK(c, dalloc(c->sceneAllocs, &ds, nbytes));
		HIPCHK(c, hipMemcpy(ds, d->sky_pixels, nbytes, hipMemcpyHostToDevice));
		S.sky = ds, S.skyW = d->sky_w, S.skyH = d->sky_h, S.skyN = d->sky_n;
	}
	c->S = S;
	c->blasRootWide8 = rootWide8;
	c->blasRoot = rootLink, c->blasRootWide = wideOK ? rootWide : rootLink, c->nInstances = d->use_tlas ? (int)d->n_instances : 0;
	c->sceneLoaded = true;
	return RT_OK;
}

// bvh::Build (splitMethod BINNEDSAH) on the device; see rt_build.h.  Planes never take part in the subdivision
// (separatePlanes, bvh.cpp:202-221: they become the right child of the root), so their node is made here.
int rt_build_bvh(rt_ctx* c, const rt_triangle* tris, uint32_t n_tri, const rt_sphere* spheres, uint32_t n_sph, const rt_plane* planes, uint32_t n_pla,
                 rt_bvh_node* nodes_out, uint32_t* prim_idx_out, uint32_t* nodes_used_out)
{
	return rt_build_bvh_split(c, RT_SPLIT_BINNEDSAH, tris, n_tri, spheres, n_sph, planes, n_pla, nodes_out, prim_idx_out, nodes_used_out);
}

int rt_build_bvh_split(rt_ctx* c, int split_method, const rt_triangle* tris, uint32_t n_tri, const rt_sphere* spheres, uint32_t n_sph, const rt_plane* planes, uint32_t n_pla,
                       rt_bvh_node* nodes_out, uint32_t* prim_idx_out, uint32_t* nodes_used_out)
{
	if (split_method < 0 || split_method > 3) return fail(c, RT_E_ARG, "rt_build_bvh: split method %d (bvh.h:38-43 knows 0..3)", split_method);
	if (!c || !nodes_out || !prim_idx_out || !nodes_used_out) return fail(c, RT_E_ARG, "rt_build_bvh: null argument");
	if ((n_tri && !tris) || (n_sph && !spheres) || (n_pla && !planes)) return fail(c, RT_E_ARG, "rt_build_bvh: null primitive array");
	const uint M = n_tri + n_sph, N = M + n_pla;
	if (M == 0) return fail(c, RT_E_UNSUPPORTED, "rt_build_bvh: no triangle or sphere to subdivide (the reference's paths for an empty or plane-only bvh stay on the host)");
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<void*> tmp;
	struct Guard { std::vector<void*>& v; ~Guard() { free_pool(v); } } guard{ tmp };
	BuildArrays B;
	memset(&B, 0, sizeof(B));
	float *dTri = nullptr, *dSph = nullptr;
	int *openA = nullptr, *openB = nullptr;
	HIPCHK(c, dalloc(tmp, &dTri, (size_t)n_tri * 14 + 4));
	HIPCHK(c, dalloc(tmp, &dSph, (size_t)n_sph * 8 + 4));
	HIPCHK(c, dalloc(tmp, &B.cen, M));
	HIPCHK(c, dalloc(tmp, &B.nlo, M));
	HIPCHK(c, dalloc(tmp, &B.nhi, M));
	HIPCHK(c, dalloc(tmp, &B.blo, M));
	HIPCHK(c, dalloc(tmp, &B.bhi, M));
	HIPCHK(c, dalloc(tmp, &B.idx, M));
	HIPCHK(c, dalloc(tmp, &B.tmp, M));
	HIPCHK(c, dalloc(tmp, &B.hpos, M));
	HIPCHK(c, dalloc(tmp, &B.fpos, M));
	HIPCHK(c, dalloc(tmp, &B.nodes, (size_t)2 * M + 2));
	HIPCHK(c, dalloc(tmp, &B.counters, 8));
	HIPCHK(c, dalloc(tmp, &openA, (size_t)2 * M + 2));
	HIPCHK(c, dalloc(tmp, &openB, (size_t)2 * M + 2));
	static_assert(sizeof(rt_triangle) == 56 && sizeof(rt_sphere) == 32, "primitive layouts of rt_amd.h");
	if (n_tri) HIPCHK(c, hipMemcpyAsync(dTri, tris, (size_t)n_tri * sizeof(rt_triangle), hipMemcpyHostToDevice, c->stream));
	if (n_sph) HIPCHK(c, hipMemcpyAsync(dSph, spheres, (size_t)n_sph * sizeof(rt_sphere), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipMemsetAsync(B.counters, 0, 8 * sizeof(int), c->stream));
	const int rootList[1] = { 0 }, one = 1;
	HIPCHK(c, hipMemcpyAsync(openA, rootList, sizeof(int), hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(k_build_prep, dim3((M + 255) / 256), dim3(256), 0, c->stream, dTri, 14, (int)n_tri, dSph, 8, (int)n_sph, B);
	hipLaunchKernelGGL(k_build_root, dim3(1), dim3(RT_BUILD_THREADS), 0, c->stream, B, M);
	HIPCHK(c, hipMemcpyAsync(B.counters + 4, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
	// Levels are launched in groups without looking at the device: a level's launch covers the most nodes the
	// level can hold (2^level, at most M) and its surplus blocks return at once; the open count is read back
	// after each group.
	int host[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	int *open = openA, *next = openB;
	int level = 0;
	for (bool done = false; !done;) {
		for (int g = 0; g < 16 && (uint)level <= M + 2; g++, level++) { // a tree over M primitives has at most M levels
			const unsigned cap = level < 31 && (1u << level) < M ? (1u << level) : M;
			hipLaunchKernelGGL(k_build_level, dim3(cap), dim3(RT_BUILD_THREADS), 0, c->stream, B, open, next, level, split_method);
			hipLaunchKernelGGL(k_build_advance, dim3(1), dim3(1), 0, c->stream, B, level);
			std::swap(open, next);
		}
		HIPCHK(c, hipMemcpyAsync(host, B.counters, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if (host[2]) return fail(c, RT_E_UNSUPPORTED, "rt_build_bvh: non-finite vertex, centre or radius (the reference's NaN-order-dependent min / max stay on the host)");
		done = host[4 + (level & 1)] == 0;
		if (!done && (uint)level > M + 2) return fail(c, RT_E_STATE, "rt_build_bvh: more levels than primitives");
	}
	HIPCHK(c, hipGetLastError());
	const int nT = host[0];
	std::vector<TNode> t((size_t)nT);
	HIPCHK(c, hipMemcpy(t.data(), B.nodes, (size_t)nT * sizeof(TNode), hipMemcpyDeviceToHost));
	HIPCHK(c, hipMemcpy(prim_idx_out, B.idx, (size_t)M * sizeof(uint), hipMemcpyDeviceToHost));
	for (uint i = 0; i < n_pla; i++) prim_idx_out[M + i] = M + i;

	// the reference's numbering: Subdivide allocates a node's children (an adjacent pair) when it reaches the
	// node, depth first, left subtree before right (bvh.cpp:317-332)
	memset(nodes_out, 0, (size_t)2 * (N + 1) * sizeof(rt_bvh_node));
	uint nodesUsed = 2;
	uint rootAt = 0;
	if (n_pla > 0) {
		// separatePlanes: root -> { node 2 = what was subdivided above, node 3 = the planes }
		nodesUsed = 4, rootAt = 2;
		rt_bvh_node& pl = nodes_out[3];
		float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
		for (uint i = 0; i < n_pla; i++) { // UpdateNodeBounds for planes, Q1 (bvh.cpp:90-111)
			const float* Np = planes[i].N;
			const float invLen = 1.0f / sqrtf(Np[0] * Np[0] + Np[1] * Np[1] + Np[2] * Np[2]);
			const float n[3] = { Np[0] * invLen, Np[1] * invLen, Np[2] * invLen };
			int ax = -1;
			if (n[0] + n[1] + n[2] == 1 && (n[0] == 1 || n[1] == 1 || n[2] == 1)) ax = n[0] == 1 ? 0 : (n[1] == 1 ? 1 : 2);
			if (ax < 0) { for (int k = 0; k < 3; k++) lo[k] = -1e30f, hi[k] = 1e30f; break; }
			for (int k = 0; k < 3; k++) {
				const float sl = k == ax ? 0.0f : -1e30f, sh = k == ax ? 0.0f : 1e30f;
				lo[k] = lo[k] < sl ? lo[k] : sl, hi[k] = hi[k] > sh ? hi[k] : sh;
			}
		}
		memcpy(pl.aabb_min, lo, 12), memcpy(pl.aabb_max, hi, 12);
		pl.left_first = M, pl.prim_count = n_pla;
	}
	std::vector<std::pair<int, uint>> stack; // (TNode, final index)
	stack.push_back({ 0, rootAt });
	while (!stack.empty()) {
		const int id = stack.back().first;
		const uint at = stack.back().second;
		stack.pop_back();
		const TNode& tn = t[(size_t)id];
		rt_bvh_node& o = nodes_out[at];
		memcpy(o.aabb_min, tn.lo, 12), memcpy(o.aabb_max, tn.hi, 12);
		if (tn.left < 0) { o.left_first = tn.first, o.prim_count = tn.count; continue; }
		const uint pair = nodesUsed;
		nodesUsed += 2;
		o.left_first = pair, o.prim_count = 0;
		stack.push_back({ tn.right, pair + 1 }); // popped after the whole left subtree
		stack.push_back({ tn.left, pair });
	}
	if (n_pla > 0) { // Refit of the root (bvh.cpp:556-594): union of its two children
		rt_bvh_node& r = nodes_out[0];
		const rt_bvh_node &a = nodes_out[2], &b = nodes_out[3];
		for (int k = 0; k < 3; k++) r.aabb_min[k] = a.aabb_min[k] < b.aabb_min[k] ? a.aabb_min[k] : b.aabb_min[k], r.aabb_max[k] = a.aabb_max[k] > b.aabb_max[k] ? a.aabb_max[k] : b.aabb_max[k];
		r.left_first = 2, r.prim_count = 0;
	}
	*nodes_used_out = nodesUsed;
	return RT_OK;
}

// tlas::build on the device (rt_build.h k_build_tlas): bounds6 = per instance the world box bvhInstance::SetTransform
// left in 'bounds' (min.xyz, max.xyz); nodes_out has room for 2 n + 1 nodes.
int rt_build_tlas(rt_ctx* c, const float* bounds6, uint32_t n, rt_tlas_node* nodes_out, uint32_t* nodes_used_out)
{
	if (!c || !bounds6 || !nodes_out || !nodes_used_out) return fail(c, RT_E_ARG, "rt_build_tlas: null argument");
	if (n < 1 || n > RT_TLAS_MAX) return fail(c, RT_E_UNSUPPORTED, "rt_build_tlas: %u instances (the reference's nodeIdx[256] holds 1..256, tlas.cpp:16)", n);
	for (uint32_t i = 0; i < 6 * n; i++)
		if (!(std::fabs(bounds6[i]) <= 1e30f)) return fail(c, RT_E_UNSUPPORTED, "rt_build_tlas: instance %u has a non-finite bound", i / 6);
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<void*> tmp;
	struct Guard { std::vector<void*>& v; ~Guard() { free_pool(v); } } guard{ tmp };
	float* dB = nullptr;
	TlasNodeDev* dN = nullptr;
	int* dU = nullptr;
	static_assert(sizeof(TlasNodeDev) == sizeof(rt_tlas_node), "TLAS node layout");
	HIPCHK(c, dalloc(tmp, &dB, (size_t)6 * n));
	HIPCHK(c, dalloc(tmp, &dN, (size_t)2 * n + 1));
	HIPCHK(c, dalloc(tmp, &dU, 1));
	HIPCHK(c, hipMemsetAsync(dN, 0, ((size_t)2 * n + 1) * sizeof(TlasNodeDev), c->stream));
	HIPCHK(c, hipMemcpyAsync(dB, bounds6, (size_t)24 * n, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(k_build_tlas, dim3(1), dim3(64), 0, c->stream, dB, (int)n, dN, dU);
	int used = 0;
	HIPCHK(c, hipMemcpyAsync(&used, dU, sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(nodes_out, dN, ((size_t)2 * n + 1) * sizeof(rt_tlas_node), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipGetLastError());
	if (used < 0) return fail(c, RT_E_UNSUPPORTED, "rt_build_tlas: no finite union area left (degenerate instance bounds)");
	*nodes_used_out = (uint32_t)used;
	return RT_OK;
}

int rt_set_time(rt_ctx* c, float t)
{
	if (!c) return RT_E_ARG;
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_set_time: no scene uploaded");
	if (c->S.useTLAS) return fail(c, RT_E_UNSUPPORTED, "rt_set_time: the reference animates only without the TLAS (animOn, template/scene.h:1389)");
	if (!c->primsOrig) return RT_OK; // nothing to animate
	c->S.wide8 = nullptr; // the quantised boxes were rounded around the uploaded geometry: a refitted tree is walked through the exact nodes
	HIPCHK(c, hipSetDevice(c->device));
	// float r = fmodf(t, 2 * PI); float a = sinf(r) * 0.5f;  (template/scene.h:1229-1230; sinf in f64, rounded once)
	const float r = fmodf(t, 2 * RT_PI);
	const float a = (float)sin((double)r) * 0.5f;
	hipLaunchKernelGGL(k_animate, dim3((c->animSlots + 255) / 256), dim3(256), 0, c->stream, c->primsOrig, c->primsMut, c->animSlots, a);
	if (c->refitLevels > 0) {
		// bottom up: a level with more records than one workgroup covers in two passes gets a launch of its own across the
		// chip; the narrow levels above the last such level share one workgroup with barriers between them
		int top = c->refitLevels; // levels [0, top) are left for the single-workgroup kernel
		for (int l = c->refitLevels - 1; l >= 0; l--) {
			const int count = c->refitLevelHost[l + 1] - c->refitLevelHost[l];
			if (count < 2048) continue;
			// everything deeper than l must be done first: the narrow levels between two wide ones go with the next wide launch's predecessor
			for (int m = top - 1; m > l; m--) {
				const int cm = c->refitLevelHost[m + 1] - c->refitLevelHost[m];
				if (cm > 0) hipLaunchKernelGGL(k_refit_level, dim3((cm + 255) / 256), dim3(256), 0, c->stream, c->pairsMut, c->primsMut, c->refitOrder, c->refitLevelHost[m], cm);
			}
			hipLaunchKernelGGL(k_refit_level, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->pairsMut, c->primsMut, c->refitOrder, c->refitLevelHost[l], count);
			top = l;
		}
		if (top > 0) hipLaunchKernelGGL(k_refit, dim3(1), dim3(1024), 0, c->stream, c->pairsMut, c->primsMut, c->refitOrder, c->refitLevelStart, top);
	}
	if (c->S.wide && c->wideNodes > 0) hipLaunchKernelGGL(k_wide_sync, dim3((c->wideNodes * 4 + 255) / 256), dim3(256), 0, c->stream, c->wideMut, c->pairsMut, c->wideNodes);
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}

static int check_overflow(rt_ctx* c);

// ---- path state -------------------------------------------------------------------------------
static int ensure_state(rt_ctx* c, int nSlots, bool pend)
{
	rt_ctx::SlotState& pl = c->slot;
	const bool wide = c->S.wide != nullptr; // the 4-wide occlusion walk hands rays back through Q.leftover: allocated only for scenes that have it
	if (pl.stateSlots >= nSlots && pl.stateLights == c->S.nLights && (pl.statePend || !pend) && (pl.stateWide || !wide)) { pl.P.nSlots = nSlots; return RT_OK; }
	HIPCHK(c, hipStreamSynchronize(c->stream));
	free_pool(pl.allocs);
	pl.stateSlots = 0;
	PathState P;
	memset(&P, 0, sizeof(P));
	const size_t n = (size_t)nSlots;
	for (int b = 0; b < 2; b++) { HIPCHK(c, dalloc(pl.allocs, &P.O[b], n)); HIPCHK(c, dalloc(pl.allocs, &P.D[b], n)); }
	for (int b = 0; b < 2; b++) { HIPCHK(c, dalloc(pl.allocs, &P.hitN[b], n)); HIPCHK(c, dalloc(pl.allocs, &P.hitId[b], n)); }
	HIPCHK(c, dalloc(pl.allocs, &P.hitP, n));
	HIPCHK(c, dalloc(pl.allocs, &P.W, n));
	HIPCHK(c, dalloc(pl.allocs, &P.E, n));
	HIPCHK(c, dalloc(pl.allocs, &P.L, n));
	HIPCHK(c, dalloc(pl.allocs, &P.sh, n * (size_t)(c->S.nLights + 1)));
	HIPCHK(c, dalloc(pl.allocs, &P.vis, n * (size_t)(c->S.nLights + 1)));
	if (pend) {
		HIPCHK(c, dalloc(pl.allocs, &P.pend, n * RT_PEND_CAP * 4));
		HIPCHK(c, dalloc(pl.allocs, &P.pendCount, n));
	}
	Queues Q;
	memset(&Q, 0, sizeof(Q));
	HIPCHK(c, dalloc(pl.allocs, &P.status, n + 16));
	HIPCHK(c, dalloc(pl.allocs, &Q.active, n));
	HIPCHK(c, dalloc(pl.allocs, &Q.shadow, n));
	HIPCHK(c, dalloc(pl.allocs, &Q.ended, n));
	HIPCHK(c, dalloc(pl.allocs, &Q.leftover, wide ? n * (size_t)(c->S.nLights > 0 ? c->S.nLights : 1) : (size_t)4));
	HIPCHK(c, dalloc(pl.allocs, &Q.counts, 16));
	HIPCHK(c, dalloc(pl.allocs, &Q.heads, (size_t)2 * RT_HEADS * RT_HEAD_STRIDE));
	HIPCHK(c, hipMemset(Q.counts, 0, 16 * sizeof(int)));
	P.nSlots = nSlots;
	pl.P = P, pl.Q = Q;
	pl.stateSlots = nSlots, pl.stateLights = c->S.nLights, pl.statePend = pend, pl.stateWide = wide;
	return RT_OK;
}
static int ensure_samples(rt_ctx* c, size_t count)
{
	if (c->sampleCap >= count) return RT_OK;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (c->samples) (void)hipFree(c->samples);
	c->samples = nullptr, c->sampleCap = 0;
	HIPCHK(c, hipMalloc((void**)&c->samples, count * sizeof(float4)));
	c->sampleCap = count;
	return RT_OK;
}

static int slot_budget(const rt_ctx* c);
// Scene::IsOccluded for the shadow queue.  Counting launches walk like the reference; timed launches take the 4-wide
// walk when the scene has wide nodes, followed by the binary walk over the (normally empty) list of rays the wide
// walk handed back because they are not clean.
static void launch_connect(rt_ctx* c, hipStream_t st, const PathState& P, const Queues& Q, int parity, uint* spill)
{
	// the any-hit walk has its own thresholds (RT_REFILL_ANY, RT_STEPMIN_ANY, RT_PAIRAGAIN_ANY)
	const int tun = tuning(c);
	if (c->counting) hipLaunchKernelGGL((k_connect<true>), dim3(c->gridConnect), dim3(RT_BLOCK), 0, st, c->S, P, Q, parity, tun, spill, c->counters + 1);
	else if (!c->S.wide) hipLaunchKernelGGL((k_connect<false>), dim3(c->gridConnect), dim3(RT_BLOCK), 0, st, c->S, P, Q, parity, tun, spill, c->counters + 1);
	else {
		hipLaunchKernelGGL((k_connect<false, true>), dim3(c->gridConnectWide), dim3(RT_BLOCK), 0, st, c->S, P, Q, parity, tun, spill, c->counters + 1);
		hipLaunchKernelGGL(k_round_begin, dim3(1), dim3(1), 0, st, Q, 0, 0, 4);
		hipLaunchKernelGGL((k_connect<false, false, true>), dim3(c->gridLeftover), dim3(RT_BLOCK), 0, st, c->S, P, Q, parity, tun, spill, c->counters + 1);
	}
}
// The round loop of the slot wavefront (rt_kernels.h), shared by rt_render_rows and rt_trace_batch: Whitted rounds (RT_MEGA=0 and
// counting launches), path batches with fewer slots than samples, RT_COUNT_REFERENCE launches, RT_STREAM=0.  One kernel at a time on
// the context's stream.  (Several sample pools on separate streams, and extend(r + 1) beside connect(r) on two streams or as one
// launch, were measured on this pipeline in rounds 1-2 and superseded by the dense pipeline of rt_stream.h:
// profiles/patches/slot_pipeline_pools_and_fusing.diff.)
static int run_rounds(rt_ctx* c, const RenderParams& R, int maxRounds, int knownRounds)
{
	// knownRounds > 0: the caller knows how many rounds empty the slots (path mode with a slot per sample:
	// one round per path segment, depth + 1 of them), so no queue length is read back before the end
	if (knownRounds > 0) maxRounds = knownRounds;
	const int mode = R.mode;
	const float t_min = mode == RT_MODE_WHITTED ? (float)1e-6 : 0.001f; // renderer.cpp:24, :131
	const int grid = c->gridBlocks;
	PathState P = c->slot.P;
	if (mode != RT_MODE_WHITTED) P.pend = nullptr, P.pendCount = nullptr;
	const Queues Q = c->slot.Q;
	hipStream_t st = c->stream;
	prof_begin(c, K_GENERATE, st);
	hipLaunchKernelGGL(k_generate, dim3((P.nSlots + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, c->S, c->C, R, P, Q);
	prof_end(c, st);
	int parity = 0, rc = RT_OK;
	for (int round = 0; round < maxRounds && rc == RT_OK; round++) {
		// round 0 of a batch with a slot per sample: every slot is ACTIVE, so no queue is built and extend /
		// shade address slots directly
		const int allActive = round == 0 && R.finishInline ? P.nSlots : 0;
		hipLaunchKernelGGL(k_round_begin, dim3(1), dim3(1), 0, st, Q, P.pendCount ? 0 : 1, allActive, 3);
		if (!allActive) hipLaunchKernelGGL(k_compact, dim3(grid / 2), dim3(RT_COMPACT_BLOCK), 0, st, P, (int)ST_ACTIVE, Q.active, &Q.counts[0]);
#ifdef RT_TAIL_PROBE
		tail_probe_reset(st);
#endif
#ifdef RT_SECTION_PROBE
		section_probe_reset(st);
#endif
#ifdef RT_STEP_COUNT
		step_count_begin(c, st, P.nSlots);
#endif
		prof_begin(c, K_EXTEND, st);
		{
			auto extendKernel = c->counting ? (allActive ? k_extend<true, true> : k_extend<true, false>) : (allActive ? k_extend<false, true> : k_extend<false, false>);
			hipLaunchKernelGGL(extendKernel, dim3(c->gridExtend), dim3(RT_BLOCK), 0, st, c->S, P, Q, parity, t_min, tuning(c), c->spill, c->counters);
		}
		prof_end(c, st);
#ifdef RT_TAIL_PROBE
		tail_probe_print(st, "extend", round);
#endif
#ifdef RT_SECTION_PROBE
		section_probe_print(st, "extend", round);
#endif
#ifdef RT_STEP_COUNT
		step_count_print(c, st, P, parity, round, c->matTypes);
#endif
		prof_begin(c, K_SHADE, st);
		hipLaunchKernelGGL(k_shade, dim3(grid), dim3(RT_BLOCK), 0, st, c->S, c->C, R, P, Q, parity, round == 0 && R.finishInline ? 1 : 0);
		prof_end(c, st);
		hipLaunchKernelGGL(k_compact, dim3(grid / 2), dim3(RT_COMPACT_BLOCK), 0, st, P, (int)ST_SHADOW, Q.shadow, &Q.counts[2]);
#ifdef RT_TAIL_PROBE
		tail_probe_reset(st);
#endif
#ifdef RT_SECTION_PROBE
		section_probe_reset(st);
#endif
		prof_begin(c, K_CONNECT, st);
		launch_connect(c, st, P, Q, parity, c->spill);
		prof_end(c, st);
#ifdef RT_TAIL_PROBE
		tail_probe_print(st, "connect", round);
#endif
#ifdef RT_SECTION_PROBE
		section_probe_print(st, "connect", round);
#endif
		prof_begin(c, K_SHADE, st);
		hipLaunchKernelGGL(k_light, dim3(grid), dim3(RT_BLOCK), 0, st, c->S, R, P, Q, parity);
		if (!R.finishInline) {
			hipLaunchKernelGGL(k_compact, dim3(grid / 2), dim3(RT_COMPACT_BLOCK), 0, st, P, (int)ST_ENDED, Q.ended, &Q.counts[1]);
			hipLaunchKernelGGL(k_finish, dim3(grid), dim3(RT_BLOCK), 0, st, c->S, c->C, R, P, Q, parity);
		}
		prof_end(c, st);
		parity = 1 - parity;
		// look at the queue lengths every few rounds (one small D2H copy + sync); the loop stops when the active queue is empty
		if ((knownRounds <= 0 && (round & 3) == 3) || round + 1 == maxRounds) {
			HIPCHK(c, hipMemcpyAsync(c->hostCounts, Q.counts, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
			HIPCHK(c, hipStreamSynchronize(st));
			const int* hc = c->hostCounts;
			if (hc[3] == 1) rc = fail(c, RT_E_OVERFLOW, "traversal stack deeper than %d entries", RT_STACK_MAX);
			else if (hc[3] == 2) rc = fail(c, RT_E_OVERFLOW, "more than %d pending Whitted branches in one pixel", RT_PEND_CAP);
			else if (hc[3] == 199) rc = fail(c, RT_E_STATE, "a traversal launch met a link no step understands (corrupt tree?) and dropped rays");
			else if (hc[3] >= 100) rc = fail(c, RT_E_STATE, "debug check %d failed in a wavefront kernel (RT_DEBUG_CHECKS build)", hc[3] - 100);
			if (hc[3] != 0) (void)hipMemsetAsync(Q.counts + 3, 0, sizeof(int), st);
			if (rc != RT_OK || hc[0] == 0 || knownRounds > 0) break;
			if (round + 1 == maxRounds) rc = fail(c, RT_E_STATE, "paths still active after %d rounds", maxRounds);
		}
	}
	if (rc != RT_OK) return rc;
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}

// ---- Whitted frames as one persistent launch (rt_mega.h) -----------------------------------------------
#define RT_LEVEL_SAMPLES_MAX (16u << 20) // larger batches keep the single launch: their drain is a small part of them, and the queues would take GBs
static int run_levels(rt_ctx* c, const RenderParams& R, const MegaState& M0, int grid0, bool* redo);
static int run_mega(rt_ctx* c, const RenderParams& R0)
{
	RenderParams R = R0;
	// deal the frame out in tiles of 64 pixels from all over it (rt_mega.h sample_of): the multiplier nearest nTiles / 61 that is coprime to nTiles
	if (R.nSamples >= 16384) {
		auto gcd = [](unsigned a, unsigned b) { while (b) { const unsigned t = a % b; a = b; b = t; } return a; };
		R.permShift = 3; // tiles of 8 pixels (profiles/r03_tick_mega.txt: tile sizes 2^0 .. 2^6)
		const unsigned nTiles = (R.nSamples + (1u << R.permShift) - 1) >> R.permShift;
		unsigned p = nTiles / 61 | 1;
		while (gcd(p, nTiles) != 1) p += 2;
		R.permMul = p;
	}
	const int gridMax = c->gridMega;
	const int lanes = std::max(c->gridMega, c->gridLevel) * RT_BLOCK; // run_levels indexes the same arrays by lane of ITS grid
	if (c->megaLanes < lanes) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		free_pool(c->megaAllocs);
		MegaState M;
		memset(&M, 0, sizeof(M));
		std::vector<void*>& A = c->megaAllocs;
		const size_t n = (size_t)lanes;
		HIPCHK(c, dalloc(A, &M.O, n)); HIPCHK(c, dalloc(A, &M.D, n)); HIPCHK(c, dalloc(A, &M.W, n)); HIPCHK(c, dalloc(A, &M.E, n)); HIPCHK(c, dalloc(A, &M.L, n));
		HIPCHK(c, dalloc(A, &M.hI, n)); HIPCHK(c, dalloc(A, &M.hN, n)); HIPCHK(c, dalloc(A, &M.hA, n)); HIPCHK(c, dalloc(A, &M.hS, n));
		HIPCHK(c, dalloc(A, &M.pend, n * RT_PEND_CAP * 4));
		M.lanes = lanes;
		c->M = M, c->megaLanes = lanes;
	}
	int grid = ((int)R.nSamples + RT_SHORT_QUEUE_RAYS * 64 - 1) / (RT_SHORT_QUEUE_RAYS * 64) + 1; // a short queue does not need the whole grid
	if (grid > gridMax) grid = gridMax;
	(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream); // work heads
	// longest first (rt_mega.h): a Whitted launch over the samples the last one rendered deals its tiles out by what they cost then
	MegaState M = c->M;
	M.cost = nullptr, M.order = nullptr;
	M.nWork = (int)(((R.nSamples + (1u << R.permShift) - 1) >> R.permShift) << R.permShift);
	bool useLevels = c->megaLevels && R.mode == RT_MODE_WHITTED && !R.customO && R.maxDepth >= 1 && R.maxDepth <= RT_LEVEL_MAX && R.nSamples <= RT_LEVEL_SAMPLES_MAX && c->S.nLights <= 32;
	// RT_MEGA_LEVELS=2: both forms give the same frame, so the context may simply time them -- each form twice on the first four batches
	// of a shape (the first run of a form pays its allocations and has no cost history), then whichever was faster
	int probe = -1; // the form this batch is timed as
	if (useLevels && c->megaLevels == 2) {
		auto& A = c->megaAuto;
		if (A.nSamples != R.nSamples || A.depth != R.maxDepth) A = {}, A.nSamples = R.nSamples, A.depth = R.maxDepth;
		if (A.choice < 0) {
			probe = A.tried[0] < 2 ? 0 : 1;
			if (!c->megaEv[0]) { (void)hipEventCreate(&c->megaEv[0]); (void)hipEventCreate(&c->megaEv[1]); }
			(void)hipEventRecord(c->megaEv[0], c->stream);
		}
		useLevels = A.choice >= 0 ? A.choice == 1 : probe == 1;
	}
	auto probe_end = [&]() {
		if (probe < 0) return;
		auto& A = c->megaAuto;
		float ms = 0;
		(void)hipEventRecord(c->megaEv[1], c->stream);
		(void)hipEventSynchronize(c->megaEv[1]);
		(void)hipEventElapsedTime(&ms, c->megaEv[0], c->megaEv[1]);
		A.ms[probe] = ms, A.tried[probe]++;
		if (A.tried[0] >= 2 && A.tried[1] >= 2) A.choice = A.ms[1] < A.ms[0] ? 1 : 0;
	};
	if (c->megaLpt && R.mode == RT_MODE_WHITTED && R.permMul && !R.customO) {
		const unsigned tilesPerHead = RT_HEADS * 8u; // sub-queues of n / RT_HEADS entries, a multiple of 64 entries = 8 tiles of 8
		const unsigned nTiles = (R.nSamples + (1u << R.permShift) - 1) >> R.permShift;
		// k_mega_order's slot mapping is a bijection only when groups = nTilesPad / 8 is a multiple of RT_HEADS: pad to 8 * RT_HEADS tiles whatever the tile size
		const unsigned unit = std::max(tilesPerHead * (64u >> R.permShift ? 64u >> R.permShift : 1u) / 8u, 8u * RT_HEADS);
		const unsigned nTilesPad = (nTiles + unit - 1) / unit * unit;
		if (c->megaCostCap < (size_t)R.nSamples) {
			HIPCHK(c, hipStreamSynchronize(c->stream));
			free_pool(c->megaOrderAllocs);
			c->megaCostCap = 0, c->megaCostSamples = 0;
			HIPCHK(c, dalloc(c->megaOrderAllocs, &c->megaCost, (size_t)R.nSamples));
			HIPCHK(c, dalloc(c->megaOrderAllocs, &c->megaOrder, (size_t)nTilesPad + 1024));
			HIPCHK(c, dalloc(c->megaOrderAllocs, &c->megaHist, (size_t)2 * RT_MEGA_BUCKETS));
			c->megaCostCap = (size_t)R.nSamples;
		}
		if (c->megaCostSamples == R.nSamples && c->megaCostFirst == R.sampleFirst) {
			(void)hipMemsetAsync(c->megaHist, 0, 2 * RT_MEGA_BUCKETS * sizeof(uint), c->stream);
			const unsigned blocks = (nTilesPad + RT_MEGA_ORDER_BLOCK - 1) / RT_MEGA_ORDER_BLOCK;
			hipLaunchKernelGGL(k_mega_hist, dim3(blocks), dim3(RT_MEGA_ORDER_BLOCK), 0, c->stream, c->megaCost, R.permShift, R.nSamples, nTilesPad, c->megaHist);
			hipLaunchKernelGGL(k_mega_order, dim3(blocks), dim3(RT_MEGA_ORDER_BLOCK), 0, c->stream, c->megaCost, R.permShift, R.nSamples, nTilesPad, c->megaHist, c->megaOrder);
			M.order = c->megaOrder;
			M.nWork = (int)(nTilesPad << R.permShift);
		}
		M.cost = c->megaCost;
		c->megaCostSamples = R.nSamples, c->megaCostFirst = R.sampleFirst;
	}
	// a flush runs the body of Trace for the lanes that finished a query: it waits for more of them than a plain store does (32 instead of 16: 5.0 -> 4.85 ms)
	if (useLevels) {
		bool redo = false;
		const int rc = run_levels(c, R, M, grid, &redo); // level 0 deals its tiles out like the single launch (M.order), and records what they cost
		if (rc != RT_OK || !redo) { probe_end(); return rc; }
		probe = -1, c->megaAuto.choice = 0; // this scene overflows the level queues: the single launch from now on
		// a queue overflowed (more than two live branches per sample on average): the frame again, as one launch
		(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream);
	}
#ifdef RT_TAIL_PROBE
	tail_probe_reset(c->stream);
#endif
	prof_begin(c, K_EXTEND);
	hipLaunchKernelGGL(k_whitted_mega, dim3(grid), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, R, M, tuning(c), c->spill, c->flags);
	prof_end(c);
#ifdef RT_TAIL_PROBE
	tail_probe_print(c->stream, "mega", 0);
#endif
	int f = 0;
	HIPCHK(c, hipMemcpyAsync(c->hostCounts, c->flags + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	f = c->hostCounts[0];
	if (f != 0) (void)hipMemsetAsync(c->flags, 0, 2 * sizeof(int), c->stream);
	if (f == 2) return fail(c, RT_E_OVERFLOW, "more than %d pending Whitted branches in one pixel", RT_PEND_CAP);
	if (f == 199) return fail(c, RT_E_STATE, "a traversal launch met a link no step understands (corrupt tree?) and dropped rays");
	if (f >= 100) return fail(c, RT_E_STATE, "debug check %d failed in the Whitted kernel (RT_DEBUG_CHECKS build)", f - 100);
	if (f) return fail(c, RT_E_OVERFLOW, "traversal stack deeper than %d entries", RT_STACK_MAX);
	HIPCHK(c, hipGetLastError());
	probe_end();
	return RT_OK;
}

// Whitted frames by tree levels (rt_mega.h): depth launches of k_whitted_level + k_whitted_reduce.  *redo: a queue overflowed.
static int run_levels(rt_ctx* c, const RenderParams& R, const MegaState& M0, int grid0, bool* redo)
{
	const int levels = R.maxDepth;
	const size_t cap = ((size_t)2 * R.nSamples + 65536 + 63) & ~(size_t)63;
	if (c->levelCap < cap || c->levelLevels < levels || c->levelSamples < (size_t)R.nSamples) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		free_pool(c->levelAllocs);
		c->levelCap = 0, c->levelLevels = 0, c->levelSamples = 0;
		LevelState V;
		memset(&V, 0, sizeof(V));
		std::vector<void*>& A = c->levelAllocs;
		HIPCHK(c, dalloc(A, &V.seg[0], cap * 4)); HIPCHK(c, dalloc(A, &V.seg[1], cap * 4));
		HIPCHK(c, dalloc(A, &V.count, (size_t)RT_LEVEL_MAX + 2));
		HIPCHK(c, dalloc(A, &V.termKey, cap * (size_t)levels)); HIPCHK(c, dalloc(A, &V.termVal, cap * (size_t)levels));
		HIPCHK(c, dalloc(A, &V.head, (size_t)R.nSamples));
		V.cap = (int)cap;
		c->V = V, c->levelCap = cap, c->levelLevels = levels, c->levelSamples = (size_t)R.nSamples;
	}
	LevelState V = c->V;
	V.qcap = V.cap;
	if (getenv("RT_LEVEL_CAP") && atoi(getenv("RT_LEVEL_CAP")) >= 64 && atoi(getenv("RT_LEVEL_CAP")) < V.cap) V.qcap = atoi(getenv("RT_LEVEL_CAP")) & ~63; // tests: queues that overflow
	(void)hipMemsetAsync(V.count, 0, (RT_LEVEL_MAX + 2) * sizeof(int), c->stream);
	(void)hipMemsetAsync(V.head, 0xFF, (size_t)R.nSamples * sizeof(int), c->stream);
	prof_begin(c, K_EXTEND);
	for (int level = 0; level < levels; level++) {
		if (level > 0) (void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream); // work heads
		V.level = level;
		const int grid = level == 0 ? std::min(grid0, c->gridLevel) : c->gridLevel;
		hipLaunchKernelGGL(k_whitted_level, dim3(grid), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, R, M0, V, tuning(c), c->spill, c->flags);
	}
	hipLaunchKernelGGL(k_whitted_reduce, dim3((R.nSamples + 255) / 256), dim3(256), 0, c->stream, R, V);
	prof_end(c);
	HIPCHK(c, hipMemcpyAsync(c->hostCounts, c->flags + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const int f = c->hostCounts[0];
	if (f != 0) (void)hipMemsetAsync(c->flags, 0, 2 * sizeof(int), c->stream);
	*redo = f == 3;
	if (f == 3) return RT_OK;
	if (f == 199) return fail(c, RT_E_STATE, "a traversal launch met a link no step understands (corrupt tree?) and dropped rays");
	if (f >= 100) return fail(c, RT_E_STATE, "debug check %d failed in the Whitted kernel (RT_DEBUG_CHECKS build)", f - 100);
	if (f) return fail(c, RT_E_OVERFLOW, "traversal stack deeper than %d entries", RT_STACK_MAX);
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}

// ---- the dense path-mode pipeline (rt_stream.h) ------------------------------------------------------
static int ensure_stream_state(rt_ctx* c, int n)
{
	const bool wide = c->S.wide != nullptr || c->S.wide8 != nullptr;
	if (!c->streamSide) {
		HIPCHK(c, hipStreamCreate(&c->streamSide));
		HIPCHK(c, hipMalloc((void**)&c->streamSideSpill, (size_t)(RT_STACK_MAX - RT_STACK_ROWS_MIN) * c->gridBlocks * RT_BLOCK * sizeof(uint)));
		HIPCHK(c, hipEventCreateWithFlags(&c->streamFork, hipEventDisableTiming));
		HIPCHK(c, hipEventCreateWithFlags(&c->streamJoin, hipEventDisableTiming));
	}
	if (c->streamCap >= n && c->streamLights == c->S.nLights && (c->streamWide || !wide)) return RT_OK;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipStreamSynchronize(c->streamSide));
	free_pool(c->streamAllocs);
	c->streamCap = 0;
	StreamState T;
	memset(&T, 0, sizeof(T));
	// k_assign writes the positions of whole 16-entry vectors and the compactions read whole 16-byte vectors of class bytes
	const size_t cap = ((size_t)n + 1023) & ~(size_t)1023;
	const size_t nl = (size_t)(c->S.nLights > 0 ? c->S.nLights : 1);
	std::vector<void*>& A = c->streamAllocs;
	for (int b = 0; b < 2; b++) {
		HIPCHK(c, dalloc(A, &T.O[b], cap)); HIPCHK(c, dalloc(A, &T.D[b], cap));
		HIPCHK(c, dalloc(A, &T.hitN[b], cap)); HIPCHK(c, dalloc(A, &T.hitId[b], cap));
		HIPCHK(c, dalloc(A, &T.W[b], cap)); HIPCHK(c, dalloc(A, &T.E[b], cap)); HIPCHK(c, dalloc(A, &T.L[b], cap));
		HIPCHK(c, dalloc(A, &T.cls[b], cap));
	}
	HIPCHK(c, dalloc(A, &T.pos, cap / 64 + 64)); // a pair of bases per group of 64 entries
	HIPCHK(c, dalloc(A, &T.shI, cap)); HIPCHK(c, dalloc(A, &T.shN, cap)); HIPCHK(c, dalloc(A, &T.shD, cap)); HIPCHK(c, dalloc(A, &T.shW, cap));
	HIPCHK(c, dalloc(A, &T.shP, cap * nl));
	HIPCHK(c, dalloc(A, &T.vis, cap * nl));
	HIPCHK(c, dalloc(A, &T.traceQ, cap));
	HIPCHK(c, dalloc(A, &T.leftover, wide ? cap * nl : (size_t)4));
	HIPCHK(c, dalloc(A, &T.counts, 16));
	HIPCHK(c, dalloc(A, &T.heads, (size_t)2 * RT_HEADS * RT_HEAD_STRIDE));
	HIPCHK(c, hipMemset(T.counts, 0, 16 * sizeof(int)));
	T.cap = (int)cap;
	c->T = T;
	c->streamCap = (int)cap, c->streamLights = c->S.nLights, c->streamWide = wide;
	return RT_OK;
}
static void launch_connect_s(rt_ctx* c, hipStream_t st, const StreamState& T, int round, uint* spill)
{
	const int tun = tuning(c);
	if (c->counting) hipLaunchKernelGGL((k_connect_s<true>), dim3(c->gridConnectS), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
	else if (c->S.wide8) {
		// the 8-wide quantised walk, then the binary walk over the rays it handed back (not clean: normally none)
		hipLaunchKernelGGL((k_connect_s<false, false, false, true>), dim3(c->gridConnectWide8S), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
		hipLaunchKernelGGL(k_stream_begin, dim3(1), dim3(1), 0, st, T);
		hipLaunchKernelGGL((k_connect_s<false, false, true>), dim3(c->gridLeftoverS), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
	} else if (!c->S.wide) hipLaunchKernelGGL((k_connect_s<false>), dim3(c->gridConnectS), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
	else {
		hipLaunchKernelGGL((k_connect_s<false, true>), dim3(c->gridConnectWideS), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
		hipLaunchKernelGGL(k_stream_begin, dim3(1), dim3(1), 0, st, T);
		hipLaunchKernelGGL((k_connect_s<false, false, true>), dim3(c->gridLeftoverS), dim3(RT_BLOCK), 0, st, c->S, T, round, tun, spill, c->counters + 1);
	}
}
// Path mode, an entry per sample: 'rounds' = start depth + 1 rounds, no queue length is read back.  connect(r) and light(r)
// run on the second stream beside compact / extend / assign of round r + 1 (twoStreams; light(r) is the last writer of
// the E and L that shade(r + 1) reads, so the main stream joins before shade); RT_FUSE=0 keeps one kernel at a time.
//   generate | begin compact extend(0) assign shade(0) | begin compact extend(1) assign {|| connect(0) light(0)} shade(1) | ...
static int run_rounds_stream(rt_ctx* c, const RenderParams& R, int rounds)
{
	const float t_min = 0.001f; // renderer.cpp:131
	const int grid = c->gridBlocks;
	// How connect(r) + light(r) share the machine with round r + 1 (profiles/r03_ab_stream_fuse.txt, r04_ab_gate.txt): one kernel at a
	// time (RT_FUSE=0) pays every traversal launch's drain in full; both streams started together (RT_FUSE=2) make the two persistent
	// kernels share the machine for their whole length: good below ~100 M samples per batch (1/8 frame 11.0 -> 10.1 ms, 33 M samples
	// 17.1 -> 16.4), level or slightly worse above, where the default therefore keeps one kernel at a time -- every kernel's time and
	// counters are then its own, which is what the roofline block of the bench line is made of.  (Holding the second stream at a gate
	// until extend(r + 1) has found its queue dry, so that connect(r) fills that drain and nothing else, was round 3's default below
	// 100 M samples -- with a gate that never waited, ADVICE.md r3.  Made to wait, it is no better than no gate at any size
	// (10.37 against 10.10 ms, 16.36 / 16.39, 85.2 / 83.6): taken out, profiles/patches/gate.diff.)
	// Small batches (RT_FUSE=1; default below RT_MIXED_MAX samples): extend(r) and connect(r - 1) as ONE launch (k_traverse_s), one drain
	// per round instead of two.
	// Measured (profiles/r03_ab_one_launch_per_round.txt, 1080p x spp): 1: 3.73 -> 3.24 ms, 2: 4.50 -> 4.18, 4: 6.10 -> 6.02, 8: 9.00 -> 9.08, 16: 14.8 -> 15.8.
	const unsigned mixedMax = getenv("RT_MIXED_MAX") ? (unsigned)atol(getenv("RT_MIXED_MAX")) : 10000000u;
	const bool mixed = !c->counting && (c->fuseTraversal == 1 || (c->fuseTraversal < 0 && R.nSamples < mixedMax)) &&
	                   (unsigned long long)R.nSamples * (unsigned)(c->S.nLights + 1) < 0x7FFFFFFFull;
	const bool twoStreams = !mixed && (c->fuseTraversal < 0 ? R.nSamples < 100000000u : c->fuseTraversal != 0);
	const StreamState& T = c->T;
	hipStream_t st = c->stream, sb = twoStreams ? c->streamSide : c->stream;
	const int cnt = c->counting ? 1 : 0;
	const int n = (int)R.nSamples;
	prof_begin(c, K_GENERATE, st);
	hipLaunchKernelGGL(k_generate_s, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, c->S, c->C, R, T, rounds == 1 ? 1 : 0, c->decideRays, cnt);
	prof_end(c, st);
	bool pendingJoin = false;
	for (int round = 0; round < rounds; round++) {
		const int last = round + 1 == rounds ? 1 : 0, lastNext = round + 2 == rounds ? 1 : 0;
		hipLaunchKernelGGL(k_compact_s, dim3(grid / 2), dim3(RT_COMPACT_BLOCK), 0, st, T, round);
#ifdef RT_TAIL_PROBE
		tail_probe_reset(st);
#endif
		prof_begin(c, K_EXTEND, st);
		if (mixed && round > 0) hipLaunchKernelGGL(k_traverse_s, dim3(c->gridTraverseS), dim3(RT_BLOCK), 0, st, c->S, T, round, last, t_min, tuning(c), c->spill);
		else if (c->counting) hipLaunchKernelGGL((k_extend_s<true>), dim3(c->gridExtendS), dim3(RT_BLOCK), 0, st, c->S, T, round & 1, last, t_min, tuning(c), c->spill, c->counters);
		else hipLaunchKernelGGL((k_extend_s<false>), dim3(c->gridExtendS), dim3(RT_BLOCK), 0, st, c->S, T, round & 1, last, t_min, tuning(c), c->spill, c->counters);
		prof_end(c, st);
#ifdef RT_TAIL_PROBE
		tail_probe_print(st, "extend_s", round);
#endif
		if (mixed && round > 0) { // the shadow answers of the round before came with this round's hits
			prof_begin(c, K_SHADE, st);
			hipLaunchKernelGGL(k_light_s, dim3(c->gridLightS), dim3(RT_BLOCK), 0, st, c->S, R, T, round - 1, 0, c->shadeLds);
			prof_end(c, st);
		}
		hipLaunchKernelGGL(k_assign, dim3(grid / 2), dim3(RT_COMPACT_BLOCK), 0, st, T, round);
		if (pendingJoin) { HIPCHK(c, hipStreamWaitEvent(st, c->streamJoin, 0)); pendingJoin = false; }
		prof_begin(c, K_SHADE, st);
		if (c->Qt.on) hipLaunchKernelGGL(k_shade_s<true>, dim3(c->gridShadeS), dim3(RT_BLOCK), 0, st, c->S, c->C, R, T, round, round == 0 ? 1 : 0, last, lastNext, c->decideRays, c->shadeLds, cnt, c->Qt);
		else hipLaunchKernelGGL(k_shade_s<false>, dim3(c->gridShadeS), dim3(RT_BLOCK), 0, st, c->S, c->C, R, T, round, round == 0 ? 1 : 0, last, lastNext, c->decideRays, c->shadeLds, cnt, c->Qt);
		prof_end(c, st);
		if (mixed && !last) continue; // this round's shadow rays ride in the next round's traversal launch
		if (twoStreams) {
			HIPCHK(c, hipEventRecord(c->streamFork, st));
			HIPCHK(c, hipStreamWaitEvent(sb, c->streamFork, 0));
		}
		prof_begin(c, K_CONNECT, sb);
		launch_connect_s(c, sb, T, round, twoStreams ? c->streamSideSpill : c->spill);
		prof_end(c, sb);
		prof_begin(c, K_SHADE, sb);
		hipLaunchKernelGGL(k_light_s, dim3(c->gridLightS), dim3(RT_BLOCK), 0, sb, c->S, R, T, round, last, c->shadeLds);
		prof_end(c, sb);
		if (twoStreams) { HIPCHK(c, hipEventRecord(c->streamJoin, sb)); pendingJoin = true; }
	}
	if (pendingJoin) HIPCHK(c, hipStreamWaitEvent(st, c->streamJoin, 0));
	if (cnt) hipLaunchKernelGGL(k_fold_decided, dim3(1), dim3(1), 0, st, c->S, T.counts, c->counters);
	HIPCHK(c, hipMemcpyAsync(c->hostCounts, T.counts, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
	c->hostCounts[4] = 0;
	if (c->Qt.on) {
		// the batch's packed reward words -> the wide sums, in stream order; the overflow word rides home with the round flags
		const int nq = c->Qt.grid * c->Qt.grid * c->Qt.grid * RT_Q_PATCHES;
		hipLaunchKernelGGL(k_q_fold, dim3((nq + 255) / 256), dim3(256), 0, st, c->Qt);
		HIPCHK(c, hipMemcpyAsync(c->hostCounts + 4, c->Qt.ovf, sizeof(int), hipMemcpyDeviceToHost, st));
	}
	HIPCHK(c, hipStreamSynchronize(st));
	const int* hc = c->hostCounts;
	int rc = RT_OK;
	if (hc[4] != 0) {
		(void)hipMemsetAsync(c->Qt.ovf, 0, sizeof(int), st);
		return fail(c, RT_E_OVERFLOW, "Q-learning sampler: more than %u rewards for one (cell, direction) within one batch of frames: render fewer frames per call", RT_Q_ACC_LIMIT);
	}
	if (hc[3] == 1) rc = fail(c, RT_E_OVERFLOW, "traversal stack deeper than %d entries", RT_STACK_MAX);
	else if (hc[3] == 199) rc = fail(c, RT_E_STATE, "a traversal launch met a link no step understands (corrupt tree?) and dropped rays");
	else if (hc[3] >= 100) rc = fail(c, RT_E_STATE, "debug check %d failed in a wavefront kernel (RT_DEBUG_CHECKS build)", hc[3] - 100);
	if (hc[3] != 0) (void)hipMemsetAsync(T.counts + 3, 0, sizeof(int), st);
	if (rc != RT_OK) return rc;
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}
// the dense pipeline serves path batches with an entry per sample; RT_COUNT_REFERENCE tallies are the slot pipeline's
// (the reference's walk visits the root pair for every ray, which a producer-side decision skips)
static bool stream_eligible(const rt_ctx* c, int mode, size_t samples)
{
	return (c->useStream || c->Qt.on) && mode == RT_MODE_PATH && !c->pathUnsupported && c->counting != RT_COUNT_REFERENCE && samples <= (size_t)slot_budget(c);
}

// Size the slots of the slot wavefront for 'total' samples: one each while the budget lasts.
static int setup_slots(rt_ctx* c, size_t total, bool pend, int& slots, bool& slotPerSample)
{
	const size_t budget = (size_t)slot_budget(c);
	slots = (int)(total < budget ? total : budget);
	if (slots < 1) slots = 1;
	slotPerSample = (size_t)slots >= total;
	return ensure_state(c, slots, pend);
}

static int slot_budget(const rt_ctx* c)
{
	// slots in flight.  Every round pays a fixed tail: once the work head runs dry the waves of a traversal
	// launch drain unevenly, and the longest rays finish alone at memory latency per step (0.3-0.8 ms per traversal
	// launch whatever it held, DESIGN.md finding 38).  Fewer, larger rounds win until every sample of the batch has its
	// own slot: 16M -> 64M -> 128M slots took the 1080p x 64 spp frame from 97.8 to 79.0 to 76.4 ms (round 1), and
	// 128M -> 256M takes 1080p x 256 spp from 148 to 138.5 ms and 4K x 1024 spp from 2.96 to 2.88 s (32-frame instead
	// of 16-frame batches).  ~250 B of state per slot: 256M slots = 67 GB of the 288 GB, allocated for the slots a batch
	// really has.  The shadow rays of a round are counted in an int: slots x lights stays below 2^31.  RT_SLOTS overrides.
	const char* e = getenv("RT_SLOTS");
	long v = e ? atol(e) : 0;
	long b = v > 0 ? v : (1l << 28);
	const long lights = c && c->S.nLights > 1 ? c->S.nLights : 1;
	if (b > 0x7FFFFFFFl / lights) b = 0x7FFFFFFFl / lights;
	return (int)b;
}
static int segments_per_sample(int mode, int depth, int nLights)
{
	// path: at most 5 segments (depth 4..0); Whitted: at most 2^depth - 1 glass segments, times the
	// mirror branches of shiny diffuse hits
	if (mode == RT_MODE_PATH) return depth + 1;
	return (1 << (depth < 12 ? depth : 12)) * (1 + nLights);
}

int rt_render_rows(rt_ctx* c, int mode, uint32_t frame0, int nframes, uint32_t seed_base, int row_first, int row_stride, int row_count, int max_depth)
{
	if (!c) return RT_E_ARG;
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_render: no scene uploaded");
	if (mode != RT_MODE_WHITTED && mode != RT_MODE_PATH) return fail(c, RT_E_ARG, "rt_render: mode %d", mode);
	if (row_first < 0 || row_stride < 1 || row_count < 1 || row_first + (row_count - 1) * row_stride >= c->height)
		return fail(c, RT_E_ARG, "rt_render: rows %d + k*%d (k < %d) outside 0..%d", row_first, row_stride, row_count, c->height);
	if (nframes < 1 || (mode == RT_MODE_WHITTED && nframes != 1)) return fail(c, RT_E_ARG, "rt_render: nframes %d (Whitted frames overwrite the accumulator: 1 only)", nframes);
	HIPCHK(c, hipSetDevice(c->device));
	const int nSlots = c->width * row_count;
	if (mode == RT_MODE_WHITTED && max_depth <= 0) { // Trace(depth <= 0) returns black without tracing
		for (int k = 0; k < row_count; k++)
			HIPCHK(c, hipMemsetAsync(c->accum + (size_t)(row_first + k * row_stride) * c->width, 0, (size_t)c->width * sizeof(float4), c->stream));
		return RT_OK;
	}
	// batches of frames: the finished samples of a batch live in a [frame][pixel] buffer (<= 4 GiB)
	const size_t tilePixels = (size_t)nSlots;
	const size_t sampleGiB = getenv("RT_SAMPLE_GIB") && atol(getenv("RT_SAMPLE_GIB")) > 0 ? (size_t)atol(getenv("RT_SAMPLE_GIB")) : 4;
	int batchFrames = (int)((sampleGiB << 30) / (tilePixels * sizeof(float4)));
	if (batchFrames < 1) batchFrames = 1;
	if (batchFrames > nframes) batchFrames = nframes;
	// path mode: keep a batch within the slot budget when a frame fits, so that every sample has its own slot
	// (exactly depth + 1 rounds, finished samples stored by shade / light, no finish pass; 4K: 16-frame batches)
	if (mode == RT_MODE_PATH && tilePixels <= (size_t)slot_budget(c) && (size_t)batchFrames * tilePixels > (size_t)slot_budget(c))
		batchFrames = (int)((size_t)slot_budget(c) / tilePixels);
	int rc = ensure_samples(c, tilePixels * batchFrames);
	if (rc != RT_OK) return rc;
	for (int f = 0; f < nframes; f += batchFrames) {
		const int bf = nframes - f < batchFrames ? nframes - f : batchFrames;
		const size_t total = tilePixels * bf;
		RenderParams R;
		memset(&R, 0, sizeof(R));
		R.mode = mode, R.frame0 = frame0 + (uint)f, R.nSamples = (uint)total, R.tilePixels = (uint)tilePixels, R.samples = c->samples;
		R.seedBase = seed_base, R.rowFirst = row_first, R.rowStride = row_stride, R.maxDepth = max_depth, R.accum = c->accum;
		R.deferGamma = mode == RT_MODE_PATH ? c->deferGamma : 0;
		if (mode == RT_MODE_PATH && c->pathUnsupported) {
			// random draws interleave with occlusion queries (shiny / raytracer == 0 diffuse): one lane per sample
			hipLaunchKernelGGL(k_sample_general, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, R, c->spill, c->flags + 1);
			hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((tilePixels + 255) / 256)), dim3(256), 0, c->stream, c->C, R, bf);
			rc = check_overflow(c);
			if (rc != RT_OK) return rc;
			continue;
		}
		if (mode == RT_MODE_PATH && c->Qt.on && !stream_eligible(c, mode, total))
			return fail(c, RT_E_UNSUPPORTED, "rt_render: the Q-learning sampler needs a path batch with an entry per sample (within the slot budget, no RT_COUNT_REFERENCE)");
		if (c->useMega && !c->counting && mode == RT_MODE_WHITTED) {
			rc = run_mega(c, R);
			if (rc != RT_OK) return rc;
			hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((tilePixels + 255) / 256)), dim3(256), 0, c->stream, c->C, R, bf);
			continue;
		}
		if (stream_eligible(c, mode, total)) {
			rc = ensure_stream_state(c, (int)total);
			if (rc != RT_OK) return rc;
			R.finishInline = 1;
			rc = run_rounds_stream(c, R, 4 + 1); // Sample starts at depth 4 (renderer.cpp:278): five hit levels
			if (rc != RT_OK) return rc;
			hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((tilePixels + 255) / 256)), dim3(256), 0, c->stream, c->C, R, bf);
			continue;
		}
		int slots = 1;
		bool slotPerSample = false;
		rc = setup_slots(c, total, mode == RT_MODE_WHITTED, slots, slotPerSample);
		if (rc != RT_OK) return rc;
		const int seg = segments_per_sample(mode, mode == RT_MODE_PATH ? 4 : max_depth, c->S.nLights);
		const int maxRounds = (int)((total + slots) / slots) * seg + seg + 4;
		const bool direct = mode == RT_MODE_PATH && slotPerSample;
		R.finishInline = direct ? 1 : 0;
		rc = run_rounds(c, R, maxRounds, direct ? seg : 0);
		if (rc != RT_OK) return rc;
		hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((tilePixels + 255) / 256)), dim3(256), 0, c->stream, c->C, R, bf);
	}
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}

int rt_render(rt_ctx* c, int mode, uint32_t frame0, int nframes, uint32_t seed_base, int y0, int y1, int max_depth)
{
	if (!c) return RT_E_ARG;
	if (y0 < 0 || y1 > c->height || y0 >= y1) return fail(c, RT_E_ARG, "rt_render: rows [%d,%d) outside 0..%d", y0, y1, c->height);
	return rt_render_rows(c, mode, frame0, nframes, seed_base, y0, 1, y1 - y0, max_depth);
}

int rt_trace_batch(rt_ctx* c, int mode, int n, const float* O, const float* D, int depth, uint32_t seed_base, float* rgb_out)
{
	const float one[3] = { 1, 1, 1 };
	return rt_trace_batch_energy(c, mode, n, O, D, depth, seed_base, one, rgb_out);
}

int rt_trace_batch_energy(rt_ctx* c, int mode, int n, const float* O, const float* D, int depth, uint32_t seed_base, const float* energy, float* rgb_out)
{
	if (!energy) return fail(c, RT_E_ARG, "rt_trace_batch: null energy");
	if (!c || !O || !D || !rgb_out || n < 0) return fail(c, RT_E_ARG, "rt_trace_batch: bad argument");
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_trace_batch: no scene uploaded");
	if (n == 0) return RT_OK;
	HIPCHK(c, hipSetDevice(c->device));
	if ((mode == RT_MODE_WHITTED && depth <= 0) || (mode == RT_MODE_PATH && depth < 0)) {
		const float v = mode == RT_MODE_WHITTED ? 0.0f : 0.05f; // renderer.cpp:23, :129
		for (int i = 0; i < 3 * n; i++) rgb_out[i] = v;
		return RT_OK;
	}
	int rc = RT_OK;
	float *dO = nullptr, *dD = nullptr;
	float4* dOut = nullptr;
	std::vector<void*> tmp;
	struct Guard { rt_ctx* c; std::vector<void*>& v; ~Guard() { (void)hipStreamSynchronize(c->stream); free_pool(v); } } guard{ c, tmp }; // every return path frees
	HIPCHK(c, dalloc(tmp, &dO, (size_t)3 * n));
	HIPCHK(c, dalloc(tmp, &dD, (size_t)3 * n));
	HIPCHK(c, dalloc(tmp, &dOut, (size_t)n));
	HIPCHK(c, hipMemcpyAsync(dO, O, (size_t)12 * n, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipMemcpyAsync(dD, D, (size_t)12 * n, hipMemcpyHostToDevice, c->stream));
	RenderParams R;
	memset(&R, 0, sizeof(R));
	R.mode = mode, R.nSamples = (uint)n, R.tilePixels = (uint)n, R.seedBase = seed_base, R.maxDepth = depth, R.accum = c->accum;
	R.customO = dO, R.customD = dD, R.customOut = dOut, R.customDepth = depth;
	R.customE[0] = energy[0], R.customE[1] = energy[1], R.customE[2] = energy[2];
	if (mode == RT_MODE_PATH && c->pathUnsupported) {
		hipLaunchKernelGGL(k_sample_general, dim3(c->gridBlocks), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, R, c->spill, c->flags + 1);
		rc = check_overflow(c);
	} else if (c->useMega && !c->counting && mode == RT_MODE_WHITTED) {
		rc = run_mega(c, R);
	} else if (stream_eligible(c, mode, (size_t)n)) {
		rc = ensure_stream_state(c, n);
		R.finishInline = 1;
		if (rc == RT_OK) rc = run_rounds_stream(c, R, depth + 1);
	} else {
		int slots = 1;
		bool slotPerSample = false;
		rc = setup_slots(c, (size_t)n, mode == RT_MODE_WHITTED, slots, slotPerSample);
		const int seg = segments_per_sample(mode, depth, c->S.nLights);
		const bool direct = mode == RT_MODE_PATH && slotPerSample;
		R.finishInline = direct ? 1 : 0;
		if (rc == RT_OK) rc = run_rounds(c, R, ((n + slots) / slots) * seg + seg + 4, direct ? seg : 0);
	}
	if (rc == RT_OK) {
		std::vector<float> out4((size_t)4 * n);
		hipError_t e = hipMemcpy(out4.data(), dOut, (size_t)16 * n, hipMemcpyDeviceToHost);
		if (e != hipSuccess) rc = fail(c, RT_E_HIP, "rt_trace_batch: copy back failed: %s", hipGetErrorString(e));
		else for (int i = 0; i < n; i++) { rgb_out[3 * i] = out4[4 * i], rgb_out[3 * i + 1] = out4[4 * i + 1], rgb_out[3 * i + 2] = out4[4 * i + 2]; }
	}
	return rc;
}

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

int rt_device_of(const rt_ctx* c) { return c ? c->device : -1; }

int rt_gather_begin(rt_ctx* dst)
{
	if (!dst) return RT_E_ARG;
	HIPCHK(dst, hipSetDevice(dst->device));
	if (!dst->rowsFree) HIPCHK(dst, hipEventCreateWithFlags(&dst->rowsFree, hipEventDisableTiming));
	HIPCHK(dst, hipEventRecord(dst->rowsFree, dst->stream));
	return RT_OK;
}

int rt_gather_rows(rt_ctx* dst, rt_ctx* src, int row_first, int row_stride, int row_count)
{
	// (every error of this call is reported on SRC -- rt_last_error(src) -- : the call may come from the source context's host
	// thread while another thread drives dst, whose error string must not be written from here)
	if (!dst || !src) return fail(src, RT_E_ARG, "rt_gather_rows: null context");
	if (dst->width != src->width || dst->height != src->height) return fail(src, RT_E_ARG, "rt_gather_rows: contexts differ in size (%dx%d vs %dx%d)", dst->width, dst->height, src->width, src->height);
	if (row_first < 0 || row_stride < 1 || row_count < 1 || row_first + (row_count - 1) * row_stride >= dst->height)
		return fail(src, RT_E_ARG, "rt_gather_rows: rows %d + k*%d (k < %d) outside 0..%d", row_first, row_stride, row_count, dst->height);
	if (dst == src) return RT_OK;
	// A PUSH on the source's stream: the copy follows the source's rendering in stream order (no host wait), every source
	// context pushes over its own link to the destination at the same time (xGMI is point to point), and the destination's
	// stream waits for the source's event -- what it does next (resolve, the next frame) sees the rows.
	// The push must also come AFTER whatever is already queued on the destination's stream and touches these rows (rt_clear's
	// whole-frame memset on a camera change, the destination's own resolve of the frame before): the source's stream waits for
	// an event recorded on the destination's stream first.  Work queued on dst->stream after this call is ordered by the second event.
	// Which event: the destination's own "rows free" mark of this frame when its owner set one (rt_gather_begin: before dst's share
	// of the frame was queued, so the push waits neither for dst's rendering nor for the sources that pushed earlier -- the pushes
	// overlap, one xGMI link each); otherwise one recorded now, behind everything dst's stream holds at this moment.
	if (dst->rowsFree) {
		HIPCHK(src, hipSetDevice(src->device));
		HIPCHK(src, hipStreamWaitEvent(src->stream, dst->rowsFree, 0));
	} else {
		HIPCHK(src, hipSetDevice(dst->device));
		if (!src->gatherReady) HIPCHK(src, hipEventCreateWithFlags(&src->gatherReady, hipEventDisableTiming));
		HIPCHK(src, hipEventRecord(src->gatherReady, dst->stream));
		HIPCHK(src, hipSetDevice(src->device));
		HIPCHK(src, hipStreamWaitEvent(src->stream, src->gatherReady, 0));
	}
	const size_t rowBytes = (size_t)dst->width * sizeof(float4), pitch = rowBytes * (size_t)row_stride;
	const float4* from = src->accum + (size_t)row_first * src->width;
	float4* to = dst->accum + (size_t)row_first * dst->width;
	bool direct = dst->device == src->device;
	if (!direct) {
		// peer access src -> dst: one strided copy engine transfer over the link between the two GPUs
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, src->device, dst->device) == hipSuccess && can) {
			const hipError_t e = hipDeviceEnablePeerAccess(dst->device, 0);
			direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
			(void)hipGetLastError();
		}
	}
	if (direct) HIPCHK(src, hipMemcpy2DAsync(to, pitch, from, pitch, rowBytes, (size_t)row_count, hipMemcpyDeviceToDevice, src->stream));
	else
		for (int k = 0; k < row_count; k++)
			HIPCHK(src, hipMemcpyPeerAsync((char*)to + (size_t)k * pitch, dst->device, (const char*)from + (size_t)k * pitch, src->device, rowBytes, src->stream));
	if (!src->gatherDone) HIPCHK(src, hipEventCreateWithFlags(&src->gatherDone, hipEventDisableTiming));
	HIPCHK(src, hipEventRecord(src->gatherDone, src->stream));
	HIPCHK(src, hipSetDevice(dst->device));
	HIPCHK(src, hipStreamWaitEvent(dst->stream, src->gatherDone, 0));
	return RT_OK;
}

// ---- batch queries -------------------------------------------------------------------------------
static int check_overflow(rt_ctx* c)
{
	int f = 0;
	HIPCHK(c, hipMemcpy(&f, c->flags + 1, sizeof(int), hipMemcpyDeviceToHost));
	if (f == 199) { (void)hipMemset(c->flags, 0, 2 * sizeof(int)); return fail(c, RT_E_STATE, "a traversal launch met a link no step understands (corrupt tree?) and dropped rays"); }
	if (f >= 100) { (void)hipMemset(c->flags, 0, 2 * sizeof(int)); return fail(c, RT_E_STATE, "debug check %d failed in a query kernel (RT_DEBUG_CHECKS build)", f - 100); }
	if (f) { (void)hipMemset(c->flags, 0, 2 * sizeof(int)); return fail(c, RT_E_OVERFLOW, "traversal stack deeper than %d entries", RT_STACK_MAX); }
	return RT_OK;
}
static int query_grid(rt_ctx* c, int n) { int g = (n + RT_CHUNK - 1) / RT_CHUNK / 4 + 1; return g > c->gridQuery ? c->gridQuery : g; }

// the scene as a query of the given scope sees it: rooted at the accelerator, one BLAS or one instance
static int scoped_scene(rt_ctx* c, int scope, int index, DScene& S, const char* who)
{
	S = c->S;
	if (scope == RT_SCOPE_SCENE || scope == RT_SCOPE_ACCEL) return RT_OK;
	if (scope == RT_SCOPE_BLAS) {
		if (index < 0 || index >= (int)c->blasRoot.size()) return fail(c, RT_E_ARG, "%s: blas %d of %d", who, index, (int)c->blasRoot.size());
		S.useTLAS = 0, S.tlasLds = 0, S.rootLink = c->blasRoot[(size_t)index], S.rootWide = c->blasRootWide[(size_t)index], S.rootWide8 = S.wide8 ? c->blasRootWide8[(size_t)index] : RT_EMPTY, S.nBruteSph = S.nBrutePla = 0;
		return RT_OK;
	}
	if (scope == RT_SCOPE_INSTANCE) {
		if (!c->S.useTLAS || index < 0 || index >= c->nInstances) return fail(c, RT_E_ARG, "%s: instance %d of %d", who, index, c->S.useTLAS ? c->nInstances : 0);
		S.rootLink = RT_INST_BIT | (uint)index;
		return RT_OK;
	}
	return fail(c, RT_E_ARG, "%s: scope %d", who, scope);
}

int rt_intersect_batch(rt_ctx* c, int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out)
{
	return rt_intersect_scope(c, RT_SCOPE_SCENE, 0, n, O, D, tmax, t_min, out);
}

int rt_intersect_scope(rt_ctx* c, int scope, int index, int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out)
{
	if (!c || !O || !D || !out || n < 0) return fail(c, RT_E_ARG, "rt_intersect_batch: bad argument");
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_intersect_batch: no scene uploaded");
	DScene S;
	{ const int src = scoped_scene(c, scope, index, S, "rt_intersect_scope"); if (src != RT_OK) return src; }
	if (n == 0) return RT_OK;
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<void*> tmp;
	float *dO = nullptr, *dD = nullptr, *dT = nullptr;
	QueryHit* dH = nullptr;
	int rc = RT_OK;
	hipError_t e = dalloc(tmp, &dO, (size_t)3 * n);
	if (e == hipSuccess) e = dalloc(tmp, &dD, (size_t)3 * n);
	if (e == hipSuccess && tmax) e = dalloc(tmp, &dT, (size_t)n);
	if (e == hipSuccess) e = dalloc(tmp, &dH, (size_t)n);
	if (e == hipSuccess) e = hipMemcpyAsync(dO, O, (size_t)12 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dD, D, (size_t)12 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && tmax) e = hipMemcpyAsync(dT, tmax, (size_t)4 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream); // work heads
#ifdef RT_SECTION_PROBE
		section_probe_reset(c->stream);
#endif
		prof_begin(c, K_QUERY);
		const bool head = scope == RT_SCOPE_SCENE;
		if (c->counting) {
			if (head) hipLaunchKernelGGL((k_query_nearest<true, true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, t_min, tuning(c), dH, c->spill, c->flags, c->counters);
			else hipLaunchKernelGGL((k_query_nearest<true, false>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, t_min, tuning(c), dH, c->spill, c->flags, c->counters);
		} else {
			if (head) hipLaunchKernelGGL((k_query_nearest<false, true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, t_min, tuning(c), dH, c->spill, c->flags, c->counters);
			else hipLaunchKernelGGL((k_query_nearest<false, false>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, t_min, tuning(c), dH, c->spill, c->flags, c->counters);
		}
		prof_end(c);
#ifdef RT_SECTION_PROBE
		if (!c->counting) section_probe_print(c->stream, "query", n);
#endif
		e = hipStreamSynchronize(c->stream);
	}
	static_assert(sizeof(QueryHit) == sizeof(rt_hit), "rt_hit layout");
	if (e == hipSuccess) e = hipMemcpy(out, dH, (size_t)n * sizeof(rt_hit), hipMemcpyDeviceToHost);
	if (e != hipSuccess) rc = fail(c, RT_E_HIP, "rt_intersect_batch: %s", hipGetErrorString(e));
	free_pool(tmp);
	if (rc == RT_OK) rc = check_overflow(c);
	return rc;
}

int rt_occluded_batch(rt_ctx* c, int n, const float* O, const float* D, const float* tmax, uint8_t* out)
{
	return rt_occluded_scope(c, RT_SCOPE_SCENE, 0, n, O, D, tmax, out);
}

int rt_sky_color_batch(rt_ctx* c, int n, const float* D, float* rgb_out)
{
	if (!c || !D || !rgb_out || n < 0) return fail(c, RT_E_ARG, "rt_sky_color_batch: bad argument");
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_sky_color_batch: no scene uploaded");
	if (n == 0) return RT_OK;
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<void*> tmp;
	struct Guard { std::vector<void*>& v; ~Guard() { free_pool(v); } } guard{ tmp };
	float *dD = nullptr, *dC = nullptr;
	HIPCHK(c, dalloc(tmp, &dD, (size_t)3 * n));
	HIPCHK(c, dalloc(tmp, &dC, (size_t)3 * n));
	HIPCHK(c, hipMemcpyAsync(dD, D, (size_t)12 * n, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(k_sky_color, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, n, dD, dC);
	HIPCHK(c, hipMemcpyAsync(rgb_out, dC, (size_t)12 * n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}

int rt_occluded_scope(rt_ctx* c, int scope, int index, int n, const float* O, const float* D, const float* tmax, uint8_t* out)
{
	if (!c || !O || !D || !out || n < 0) return fail(c, RT_E_ARG, "rt_occluded_batch: bad argument");
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_occluded_batch: no scene uploaded");
	DScene S;
	{ const int src = scoped_scene(c, scope, index, S, "rt_occluded_scope"); if (src != RT_OK) return src; }
	if (n == 0) return RT_OK;
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<void*> tmp;
	float *dO = nullptr, *dD = nullptr, *dT = nullptr;
	unsigned char* dR = nullptr;
	uint* dL = nullptr;
	int rc = RT_OK;
	const bool wide8Walk = !c->counting && S.wide8;
	const bool wideWalk = !c->counting && (c->S.wide || wide8Walk);
	hipError_t e = dalloc(tmp, &dO, (size_t)3 * n);
	if (e == hipSuccess) e = dalloc(tmp, &dD, (size_t)3 * n);
	if (e == hipSuccess && tmax) e = dalloc(tmp, &dT, (size_t)n);
	if (e == hipSuccess) e = dalloc(tmp, &dR, (size_t)n);
	if (e == hipSuccess && wideWalk) e = dalloc(tmp, &dL, (size_t)n);
	if (e == hipSuccess) e = hipMemcpyAsync(dO, O, (size_t)12 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dD, D, (size_t)12 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && tmax) e = hipMemcpyAsync(dT, tmax, (size_t)4 * n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream); // work heads
		prof_begin(c, K_QUERY);
		if (c->counting) hipLaunchKernelGGL((k_query_occluded<true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, tuning(c), dR, c->spill, c->flags, c->counters + 1, dL);
		else if (!wideWalk) hipLaunchKernelGGL((k_query_occluded<false>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, tuning(c), dR, c->spill, c->flags, c->counters + 1, dL);
		else {
			// the 4-wide walk, then the binary walk over the rays it handed back (not clean: normally none)
			(void)hipMemsetAsync(c->flags + 2, 0, sizeof(int), c->stream);
			if (wide8Walk) hipLaunchKernelGGL((k_query_occluded<false, false, false, true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, tuning(c), dR, c->spill, c->flags, c->counters + 1, dL);
			else hipLaunchKernelGGL((k_query_occluded<false, true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, tuning(c), dR, c->spill, c->flags, c->counters + 1, dL);
			(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream);
			hipLaunchKernelGGL((k_query_occluded<false, false, true>), dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, S, n, dO, dD, dT, tuning(c), dR, c->spill, c->flags, c->counters + 1, dL);
		}
		prof_end(c);
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess) e = hipMemcpy(out, dR, (size_t)n, hipMemcpyDeviceToHost);
	if (e != hipSuccess) rc = fail(c, RT_E_HIP, "rt_occluded_batch: %s", hipGetErrorString(e));
	free_pool(tmp);
	if (rc == RT_OK) rc = check_overflow(c);
	return rc;
}

int rt_primary_hits(rt_ctx* c, float t_min, int32_t* obj_out, float* t_out)
{
	if (!c || !obj_out || !t_out) return fail(c, RT_E_ARG, "rt_primary_hits: null argument");
	if (!c->sceneLoaded) return fail(c, RT_E_STATE, "rt_primary_hits: no scene uploaded");
	HIPCHK(c, hipSetDevice(c->device));
	const int n = c->width * c->height;
	std::vector<void*> tmp;
	int* dO = nullptr;
	float* dT = nullptr;
	int rc = RT_OK;
	hipError_t e = dalloc(tmp, &dO, (size_t)n);
	if (e == hipSuccess) e = dalloc(tmp, &dT, (size_t)n);
	if (e == hipSuccess) {
		(void)hipMemsetAsync(c->flags + 16, 0, RT_HEADS * RT_HEAD_STRIDE * sizeof(int), c->stream); // work heads
		prof_begin(c, K_QUERY);
		if (c->counting) hipLaunchKernelGGL(k_primary_hits<true>, dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, t_min, tuning(c), dO, dT, c->spill, c->flags, c->counters);
		else hipLaunchKernelGGL(k_primary_hits<false>, dim3(query_grid(c, n)), dim3(RT_BLOCK), 0, c->stream, c->S, c->C, t_min, tuning(c), dO, dT, c->spill, c->flags, c->counters);
		prof_end(c);
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess) e = hipMemcpy(obj_out, dO, (size_t)n * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(t_out, dT, (size_t)n * 4, hipMemcpyDeviceToHost);
	if (e != hipSuccess) rc = fail(c, RT_E_HIP, "rt_primary_hits: %s", hipGetErrorString(e));
	free_pool(tmp);
	if (rc == RT_OK) rc = check_overflow(c);
	return rc;
}

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
int rt_qlearn_enable(rt_ctx* c, const rt_qlearn_params* p)
{
	if (!c) return RT_E_ARG;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	free_pool(c->qAllocs);
	memset(&c->Qt, 0, sizeof(c->Qt));
	if (!p) return RT_OK;
	if (p->grid < 1 || p->grid > 64) return fail(c, RT_E_ARG, "rt_qlearn_enable: grid %d (1..64)", p->grid);
	for (int a = 0; a < 3; a++)
		if (!(p->hi[a] > p->lo[a])) return fail(c, RT_E_ARG, "rt_qlearn_enable: empty box on axis %d", a);
	if (!(p->alpha > 0 && p->alpha <= 1) || !(p->epsilon >= 0 && p->epsilon <= 1) || !(p->q_init > 0)) return fail(c, RT_E_ARG, "rt_qlearn_enable: alpha in (0, 1], epsilon in [0, 1], q_init > 0");
	if (c->pathUnsupported) return fail(c, RT_E_UNSUPPORTED, "rt_qlearn_enable: this scene's path mode runs the general kernel (%s)", c->pathUnsupportedWhy.c_str());
	QTable Q;
	memset(&Q, 0, sizeof(Q));
	const size_t cells = (size_t)p->grid * p->grid * p->grid;
	float4* centre = nullptr;
	float* wgt = nullptr;
	HIPCHK(c, dalloc(c->qAllocs, &Q.q, cells * RT_Q_ROW));
	HIPCHK(c, dalloc(c->qAllocs, &Q.v, cells * RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &Q.sum, cells * RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &Q.cnt, cells * RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &Q.acc, cells * RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &centre, (size_t)RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &wgt, (size_t)RT_Q_PATCHES * RT_Q_PATCHES));
	HIPCHK(c, dalloc(c->qAllocs, &Q.ovf, (size_t)4)); // the sampler's own overflow word (flags[2] belongs to the wide occlusion walk's leftover count)
	HIPCHK(c, hipMemsetAsync(Q.ovf, 0, 4 * sizeof(int), c->stream));
	Q.centre = centre, Q.wgt = wgt, Q.grid = p->grid, Q.on = 1;
	for (int a = 0; a < 3; a++) Q.lo[a] = p->lo[a], Q.inv[a] = (float)p->grid / (p->hi[a] - p->lo[a]);
	Q.eps = p->epsilon, Q.alpha = p->alpha, Q.qMin = 1e-4f, Q.learnMask = p->learn_mask;
	hipLaunchKernelGGL(k_q_weights, dim3(RT_Q_PATCHES), dim3(RT_Q_PATCHES), 0, c->stream, centre, wgt);
	hipLaunchKernelGGL(k_q_init, dim3(((int)cells + 63) / 64), dim3(64), 0, c->stream, Q, p->q_init);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->Qt = Q;
	return RT_OK;
}
int rt_qlearn_apply(rt_ctx* c)
{
	if (!c || !c->Qt.on) return fail(c, RT_E_STATE, "rt_qlearn_apply: the sampler is off");
	HIPCHK(c, hipSetDevice(c->device));
	// stream-ordered, no host wait: the rewards of every batch were folded into the wide sums at the batch's end (run_rounds_stream),
	// where a count field past its limit was reported by the render call itself
	const int cells = c->Qt.grid * c->Qt.grid * c->Qt.grid;
	hipLaunchKernelGGL(k_q_apply, dim3((cells + 63) / 64), dim3(64), 0, c->stream, c->Qt); // a 64 x 64 product per cell: one wave per block spreads the cells over the CUs
	HIPCHK(c, hipGetLastError());
	return RT_OK;
}
int rt_qlearn_get_sums(rt_ctx* c, int64_t* sums, uint32_t* counts)
{
	if (!c || !c->Qt.on || !sums || !counts) return fail(c, RT_E_STATE, "rt_qlearn_get_sums: the sampler is off, or a null argument");
	HIPCHK(c, hipSetDevice(c->device));
	const size_t n = (size_t)c->Qt.grid * c->Qt.grid * c->Qt.grid * RT_Q_PATCHES;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(sums, c->Qt.sum, n * 8, hipMemcpyDeviceToHost));
	HIPCHK(c, hipMemcpy(counts, c->Qt.cnt, n * 4, hipMemcpyDeviceToHost));
	return RT_OK;
}
int rt_qlearn_set_sums(rt_ctx* c, const int64_t* sums, const uint32_t* counts)
{
	if (!c || !c->Qt.on || !sums || !counts) return fail(c, RT_E_STATE, "rt_qlearn_set_sums: the sampler is off, or a null argument");
	HIPCHK(c, hipSetDevice(c->device));
	const size_t n = (size_t)c->Qt.grid * c->Qt.grid * c->Qt.grid * RT_Q_PATCHES;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemset(c->Qt.acc, 0, n * 8)); // the caller's sums replace everything gathered so far
	HIPCHK(c, hipMemcpy(c->Qt.sum, sums, n * 8, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->Qt.cnt, counts, n * 4, hipMemcpyHostToDevice));
	return RT_OK;
}
// The pending reward sums in the CALLER's device memory (several processes: the sums are all-reduced in place between the ranks,
// RCCL on device pointers, instead of four host copies per exchange -- VERDICT r5 item 5).  What was pending moves over.
int rt_qlearn_bind_sums(rt_ctx* c, int64_t* dev_sums, uint32_t* dev_counts)
{
	if (!c || !c->Qt.on || !dev_sums || !dev_counts) return fail(c, RT_E_STATE, "rt_qlearn_bind_sums: the sampler is off, or a null argument");
	HIPCHK(c, hipSetDevice(c->device));
	const size_t n = (size_t)c->Qt.grid * c->Qt.grid * c->Qt.grid * RT_Q_PATCHES;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(dev_sums, c->Qt.sum, n * 8, hipMemcpyDeviceToDevice));
	HIPCHK(c, hipMemcpy(dev_counts, c->Qt.cnt, n * 4, hipMemcpyDeviceToDevice));
	c->Qt.sum = (long long*)dev_sums, c->Qt.cnt = dev_counts; // (the library's own arrays stay in qAllocs until the sampler is switched off)
	return RT_OK;
}
int rt_qlearn_get_table(rt_ctx* c, float* q_out)
{
	if (!c || !c->Qt.on || !q_out) return fail(c, RT_E_STATE, "rt_qlearn_get_table: the sampler is off, or a null argument");
	HIPCHK(c, hipSetDevice(c->device));
	const size_t cells = (size_t)c->Qt.grid * c->Qt.grid * c->Qt.grid;
	std::vector<float> rows(cells * RT_Q_ROW);
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(rows.data(), c->Qt.q, rows.size() * 4, hipMemcpyDeviceToHost));
	for (size_t v = 0; v < cells; v++) memcpy(q_out + v * RT_Q_PATCHES, &rows[v * RT_Q_ROW + 8], RT_Q_PATCHES * 4);
	return RT_OK;
}

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
