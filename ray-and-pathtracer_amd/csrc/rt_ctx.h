// The context behind the C ABI (struct rt_ctx) and what every unit of rt_api.hip shares: error reporting, device allocation pools, the
// HIP-event kernel timers, the measurement builds' probes.  rt_api.hip is ONE translation unit on purpose -- the kernels live in headers and the
// hot ones' code generation is pinned to this compilation (bench.py kernel_hash) -- cut into units by what they do:
//   rt_api.hip          context, camera, accumulator access, counters / timers, build + tuning info
//   rt_api_upload.inc   rt_upload_scene
//   rt_api_build.inc    device builders, refit
//   rt_api_render.inc   rt_render*, rt_trace_batch*: round loops and batches
//   rt_api_gather.inc   rt_gather_*
//   rt_api_query.inc    batch queries
//   rt_api_qlearn.inc   rt_qlearn_*
#pragma once
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

// The environment, read ONCE per context (rt_create -> read_knobs: the library's only getenv site).
struct Knobs {
	// what a user may want to set
	long slots = 0;        // RT_SLOTS: path samples in flight (default 2^28: ~67 GB of path state at the largest batches)
	long sampleGiB = 4;    // RT_SAMPLE_GIB: size of the finished-sample buffer of a batch of frames
	int exactGamma = 0;    // RT_EXACT_GAMMA=1: the gamma of a finished sample as the reference's double-precision pow (bit-equal accumulators)
	int wide = -1;         // RT_WIDE=0/1: the 4-wide any-hit walk off / on (default: on for a scene BVH, off with a TLAS)
	int wide8 = 0;         // RT_WIDE8=1: the 8-wide quantised any-hit walk
	// switches between EQUIVALENT code paths (same results: tests/test_gpu_parity.py renders every one of them against the oracle, and the
	// profiles/ A/B scripts time them); the defaults are the measured optimum, none of them is a tuning knob a user needs
	int fuse = -1;         // RT_FUSE: 0 one kernel at a time, 1 one traversal launch per round, 2 connect + light on a second stream; < 0: by batch size
	int stream = 1;        // RT_STREAM=0: the slot wavefront of rt_kernels.h instead of the dense pipeline
	int decide = 1;        // RT_DECIDE: producers answer rays whose first traversal step leaves nothing to visit
	int mega = 1, megaLpt = 1, megaLevels = 2; // RT_MEGA / RT_MEGA_LPT / RT_MEGA_LEVELS: the forms of a Whitted frame (rt_mega.h)
	int deferGamma = 1;    // RT_DEFER_GAMMA=0: the gamma where a sample is stored instead of in k_accumulate
	int shadeLds = 1;      // RT_SHADE_LDS=0: material / light tables from memory
	int tlasLds = 1;       // RT_TLAS_LDS=0: the TLAS from global memory even when it fits a block's LDS
	int gammaLut = 1;      // RT_GAMMA_LUT=0: no 256-entry table for finished sky samples
	int levelCap = 0;      // RT_LEVEL_CAP: (tests) level queues that overflow
	unsigned mixedMax = 10000000u; // RT_MIXED_MAX: batches below this many samples run one traversal launch per round
};
static void read_knobs(Knobs& k)
{
	auto num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
	k.slots = num("RT_SLOTS", 0), k.sampleGiB = num("RT_SAMPLE_GIB", 4) > 0 ? num("RT_SAMPLE_GIB", 4) : 4;
	k.exactGamma = num("RT_EXACT_GAMMA", 0) != 0, k.wide = (int)num("RT_WIDE", -1), k.wide8 = num("RT_WIDE8", 0) != 0;
	{ const long f = num("RT_FUSE", -1); k.fuse = f < 0 ? -1 : (f > 2 ? 2 : (int)f); }
	k.stream = num("RT_STREAM", 1) != 0, k.decide = (int)num("RT_DECIDE", 1) & 3, k.mega = num("RT_MEGA", 1) != 0, k.megaLpt = num("RT_MEGA_LPT", 1) != 0;
	k.megaLevels = (int)num("RT_MEGA_LEVELS", 2), k.deferGamma = num("RT_DEFER_GAMMA", 1) != 0, k.shadeLds = num("RT_SHADE_LDS", 1) != 0;
	k.tlasLds = num("RT_TLAS_LDS", 1) != 0, k.gammaLut = num("RT_GAMMA_LUT", 1) != 0, k.levelCap = (int)num("RT_LEVEL_CAP", 0);
	k.mixedMax = (unsigned)num("RT_MIXED_MAX", 10000000);
	// the scheduling thresholds are compile-time constants since round 4 (rt_scene_dev.h): a sweep script that still sets them in
	// the environment would read as a flat sweep -- say so once
	static bool warned = false;
	const char* retired[] = { "RT_REFILL", "RT_REFILL_ANY", "RT_STEPMIN", "RT_STEPMIN_ANY", "RT_STEPMIN_XFORM", "RT_PAIRAGAIN", "RT_PAIRAGAIN_ANY", "RT_DRAIN_LANES", "RT_DRAIN_LANES_ANY" };
	for (const char* name : retired)
		if (!warned && getenv(name)) { fprintf(stderr, "rt_amd: %s is a compile-time constant (rebuild with make EXTRA=-D%s=N); the environment variable is ignored\n", name, name); warned = true; }
}

struct rt_ctx {
	int device = 0, width = 0, height = 0;
	Knobs knobs;
	hipStream_t stream = nullptr;
	std::string err, tuningInfo;
	// scene
	DScene S;
	bool sceneLoaded = false;
	std::vector<void*> sceneAllocs;
	int sceneRaytracer = -1; // rt_set_scene_raytracer: scene.raytracer as the caller's Scene holds it (-1: it follows the function called)
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

