// Renderer (renderer.h, renderer.cpp of the reference): Init / Tick / Trace / Sample with the
// reference's signatures.  The pixel loop of Tick (renderer.cpp:259-291) is one rt_render() +
// rt_resolve() pair on the device; Trace and Sample evaluate a caller-supplied ray there too.
#include "rapt.h"
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace rapt {

// One host thread per context, started once and parked on a condition variable between Ticks (a context is not thread safe,
// and each thread binds its context's device once): the scanline loop of renderer.cpp:259 with one interleaved share of the rows
// per GPU.  Round 3 spawned and joined N std::threads in every Tick.
struct Renderer::Workers {
	struct Slot { std::thread th; std::function<void()> job; bool busy = false; std::exception_ptr err; };
	std::vector<Slot> slots;
	std::mutex m;
	std::condition_variable wake, done;
	bool quit = false;
	explicit Workers(int n) : slots((size_t)n)
	{
		for (int k = 0; k < n; k++)
			slots[(size_t)k].th = std::thread([this, k]() {
				Slot& s = slots[(size_t)k];
				std::unique_lock<std::mutex> lk(m);
				for (;;) {
					wake.wait(lk, [&]() { return quit || s.busy; });
					if (quit) return;
					std::function<void()> job = s.job;
					lk.unlock();
					std::exception_ptr e;
					try { job(); } catch (...) { e = std::current_exception(); }
					lk.lock();
					s.err = e, s.busy = false;
					done.notify_all();
				}
			});
	}
	~Workers()
	{
		{ std::lock_guard<std::mutex> lk(m); quit = true; }
		wake.notify_all();
		for (Slot& s : slots) s.th.join();
	}
	// run job(k) on worker k for every k, wait for all, rethrow the first failure
	void run(const std::function<void(int)>& job)
	{
		{
			std::lock_guard<std::mutex> lk(m);
			for (size_t k = 0; k < slots.size(); k++) slots[k].job = [&job, k]() { job((int)k); }, slots[k].busy = true, slots[k].err = nullptr;
		}
		wake.notify_all();
		std::unique_lock<std::mutex> lk(m);
		done.wait(lk, [&]() { for (const Slot& s : slots) if (s.busy) return false; return true; });
		for (Slot& s : slots) if (s.err) std::rethrow_exception(s.err);
	}
};

static void check(rt_ctx* ctx, int rc)
{
	if (rc != RT_OK) throw std::runtime_error(std::string("rt_amd: ") + rt_last_error(ctx));
}

Renderer::Renderer(int w, int h, int dev) : width(w), height(h), device(dev) { camera.Reset(w, h); devices.assign(1, dev); }

void Renderer::UseDevices(const std::vector<int>& devs)
{
	if (ctx) throw std::runtime_error("Renderer::UseDevices: call before Init()");
	if (devs.empty()) throw std::runtime_error("Renderer::UseDevices: no device");
	devices = devs;
	device = devs[0];
}
void Renderer::UseAllDevices()
{
	std::vector<int> d;
	for (int i = 0; i < rt_device_count(); i++) d.push_back(i);
	if (d.empty()) throw std::runtime_error("rt_amd: no HIP device");
	UseDevices(d);
}
Renderer::~Renderer() { Shutdown(); }

void Renderer::Init() // renderer.cpp:5-11: allocate and zero the float4 accumulator
{
	if (ctx) return;
	for (int d : devices) {
		rt_ctx* c = rt_create(d, width, height);
		if (!c) { const std::string why = rt_last_error(nullptr); for (rt_ctx* o : ctxs) rt_destroy(o); ctxs.clear(); throw std::runtime_error("rt_amd: " + why); }
		ctxs.push_back(c);
	}
	ctx = ctxs[0];
	if (ctxs.size() > 1) workers = new Workers((int)ctxs.size());
	accumulator = new float4[(size_t)width * height]();
	screenPixels = new uint32_t[(size_t)width * height];
	memset(screenPixels, 0, 4 * (size_t)width * height);
	frame = 0;
}

void Renderer::Shutdown()
{
	delete workers;
	workers = nullptr;
	for (rt_ctx* c : ctxs) rt_destroy(c);
	ctxs.clear();
	ctx = nullptr;
	scene.ctx = nullptr, scene.alsoCtx.clear(); // the scene's contexts are gone with ours
	delete[] accumulator;
	delete[] screenPixels;
	accumulator = nullptr, screenPixels = nullptr;
}

void Renderer::Commit()
{
	if (!ctx) Init();
	scene.Commit(ctx);
	for (size_t k = 1; k < ctxs.size(); k++) scene.CommitAlso(ctxs[k]);
}

void Renderer::SyncCamera()
{
	rt_camera c;
	memset(&c, 0, sizeof(c));
	c.cam_pos[0] = camera.camPos.x, c.cam_pos[1] = camera.camPos.y, c.cam_pos[2] = camera.camPos.z;
	c.top_left[0] = camera.topLeft.x, c.top_left[1] = camera.topLeft.y, c.top_left[2] = camera.topLeft.z;
	c.top_right[0] = camera.topRight.x, c.top_right[1] = camera.topRight.y, c.top_right[2] = camera.topRight.z;
	c.bottom_left[0] = camera.bottomLeft.x, c.bottom_left[1] = camera.bottomLeft.y, c.bottom_left[2] = camera.bottomLeft.z;
	c.fisheye = camera.fishEye ? 1 : 0, c.view_angle = camera.viewAngle, c.y_angle = camera.yAngle;
	for (rt_ctx* k : ctxs) check(k, rt_set_camera(k, &c));
}

// renderer.cpp:240-305 without the animation, input and printf parts.  As in the reference, 'it' is read before a
// camera change resets the iteration number (:247 vs :253-255): the frame on which the camera changed is resolved
// with the stale count and does not advance it (:293-294).
void Renderer::Tick(float /*deltaTime*/)
{
	if (!ctx) Init();
	scene.totIterationNumber++;
	const int it = scene.GetIterationNumber();
	const bool camChanged = camera.GetChange();
	if (camChanged && !scene.raytracer) {
		scene.SetIterationNumber(1);
		for (rt_ctx* k : ctxs) check(k, rt_clear(k)); // accumulator[...] = float3(0) for every pixel (:273-275)
	}
	SyncCamera();
	const int mode = scene.raytracer ? RT_MODE_WHITTED : RT_MODE_PATH;
	const int n = (int)ctxs.size();
	if (n == 1) check(ctx, rt_render(ctx, mode, frame, 1, seedBase, 0, height, 4));
	else {
		// the scanline loop of renderer.cpp:259, one interleaved share of the rows per GPU, each on its context's own (parked)
		// host thread; a share's rows are pushed to context 0 from that thread as soon as they are queued -- every gather is
		// issued before anything waits (rt_gather_rows is asynchronous: context 0's stream waits for the rows, not the host)
		check(ctx, rt_gather_begin(ctx)); // context 0's rows are free from here on (after the clear above and the resolve of the frame before): the pushes wait for this mark alone
		workers->run([&](int k) {
			const int count = (height - k + n - 1) / n;
			if (count <= 0) return;
			check(ctxs[k], rt_render_rows(ctxs[k], mode, frame, 1, seedBase, k, n, count, 4));
			if (k > 0) check(ctxs[k], rt_gather_rows(ctx, ctxs[k], k, n, count)); // (its errors are reported on the source context)
		});
	}
	if (!scene.raytracer && qlearning) {
		// learning happens between frames: add the contexts' pending (integer) reward sums, give every context the total, apply
		if (n > 1) {
			std::vector<int64_t> sum, s1;
			std::vector<uint32_t> cnt, c1;
			const size_t cells = (size_t)qgrid * qgrid * qgrid * 64;
			sum.assign(cells, 0), cnt.assign(cells, 0), s1.resize(cells), c1.resize(cells);
			for (rt_ctx* k : ctxs) {
				check(k, rt_qlearn_get_sums(k, s1.data(), c1.data()));
				for (size_t i = 0; i < cells; i++) sum[i] += s1[i], cnt[i] += c1[i];
			}
			for (rt_ctx* k : ctxs) check(k, rt_qlearn_set_sums(k, sum.data(), cnt.data()));
		}
		for (rt_ctx* k : ctxs) check(k, rt_qlearn_apply(k));
	}
	if (!scene.raytracer) frame++;
	check(ctx, rt_resolve(ctx, it, 0, height, screenPixels));
	if (downloadEachTick) check(ctx, rt_download_accumulator(ctx, 0, height, &accumulator[0].x));
	if (!scene.raytracer && !camChanged) scene.SetIterationNumber(it + 1);
	camera.SetChange(false);
}

void Renderer::EnableQLearning(int grid, float3 lo, float3 hi, float alpha, float epsilon, float qInit)
{
	if (!ctx) Init();
	rt_qlearn_params p;
	p.grid = grid, p.alpha = alpha, p.epsilon = epsilon, p.q_init = qInit, p.learn_mask = qlearnMask;
	p.lo[0] = lo.x, p.lo[1] = lo.y, p.lo[2] = lo.z, p.hi[0] = hi.x, p.hi[1] = hi.y, p.hi[2] = hi.z;
	for (rt_ctx* k : ctxs) check(k, rt_qlearn_enable(k, &p));
	qlearning = true, qgrid = grid;
}
void Renderer::DisableQLearning()
{
	for (rt_ctx* k : ctxs) check(k, rt_qlearn_enable(k, nullptr));
	qlearning = false;
}

static float3 eval(rt_ctx* ctx, int mode, const Ray& ray, int depth, const float3& energy, uint32_t seed)
{
	float rgb[3];
	check(ctx, rt_trace_batch_energy(ctx, mode, 1, &ray.O.x, &ray.D.x, depth, seed, &energy.x, rgb));
	return float3(rgb[0], rgb[1], rgb[2]);
}
// Trace / Sample on a caller's ray, with scene.raytracer as the Scene holds it: the combinations Tick makes (Trace with the flag set, Sample
// with it clear) run the wavefront kernels, the other two (renderer.cpp:33-43, 107-121, 143-153) the general kernels (rt_set_scene_raytracer).
// (the flag is the context's only for the duration of the call: direct users of the C ABI keep their own setting's default)
struct ScopedSceneFlag {
	rt_ctx* ctx;
	ScopedSceneFlag(rt_ctx* c, bool raytracer) : ctx(c) { check(c, rt_set_scene_raytracer(c, raytracer ? 1 : 0)); }
	~ScopedSceneFlag() { (void)rt_set_scene_raytracer(ctx, -1); }
};
float3 Renderer::Trace(Ray& ray, int depth, float3 energy)
{
	ScopedSceneFlag flag(ctx, scene.raytracer);
	return eval(ctx, RT_MODE_WHITTED, ray, depth, energy, seedBase);
}
float3 Renderer::Sample(Ray& ray, int depth, float3 energy)
{
	ScopedSceneFlag flag(ctx, scene.raytracer);
	return eval(ctx, RT_MODE_PATH, ray, depth, energy, seedBase);
}

} // namespace rapt
