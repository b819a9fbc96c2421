/* rt_amd.h -- C ABI of the MI355X trace-loop library (librt_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of Fannollost/Ray-and-pathtracer: the per-pixel
 * trace loop.  The reference has no device boundary (everything is one address space); this
 * library inserts one at the `#pragma omp parallel for` line of Renderer::Tick
 * (renderer.cpp:259): one rt_render() call replaces the whole parallel region, and the batch
 * queries replace Scene::FindNearest / Scene::IsOccluded.  Citations are file:line in the
 * reference repository.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every call returns 0 on success or a
 * negative rt_status; rt_last_error() gives the text.  The caller owns every host buffer.  A
 * context is bound to one GPU and is NOT thread-safe (one host thread per context, one context per
 * GPU).  The library never falls back to a CPU path: without a usable gfx950 device rt_create()
 * fails.
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum {
	RT_OK = 0,
	RT_E_NODEVICE = -1,    /* no HIP device / not gfx950 */
	RT_E_ARG = -2,         /* bad argument */
	RT_E_HIP = -3,         /* HIP runtime error */
	RT_E_UNSUPPORTED = -4, /* scene feature outside the device path (see rt_upload_scene) */
	RT_E_STATE = -5,       /* call order (e.g. render before upload) */
	RT_E_OVERFLOW = -6     /* traversal stack deeper than 130 = tlas.cpp:67 stack[64] + instance sentinel + bvh.cpp:608 stack[64], too many pending Whitted branches, or (Q-learning sampler) more than 2^19 rewards for one (cell, direction) within one batch of frames of a render call (reported by that call) */
} rt_status;

/* ---- scene records -------------------------------------------------------------------------
 * Shapes follow the reference's own structures so a binding can fill them with a few loops.
 * 'material' fields are indices into rt_scene_desc.materials (the reference stores material*). */

/* BVHNode (bvh.h:10-24), 32 bytes, identical layout. leaf: prim_count > 0, left_first = first
 * entry in prim_idx; inner: prim_count == 0, children at left_first and left_first + 1. */
typedef struct { float aabb_min[3]; uint32_t left_first; float aabb_max[3]; uint32_t prim_count; } rt_bvh_node;

/* TLASNode (tlas.h:4-11), 32 bytes, identical layout. leaf: left_right == 0, blas = instance
 * index; inner: children = left_right & 0xFFFF and left_right >> 16. Node 0 is the root. */
typedef struct { float aabb_min[3]; uint32_t left_right; float aabb_max[3]; uint32_t blas; } rt_tlas_node;

/* Triangle (template/scene.h:175-251): v0, v1, v2, N, objIdx, mat (e1, e2, centroid are derived) */
typedef struct { float v0[3], v1[3], v2[3], N[3]; int32_t obj_idx; int32_t material; } rt_triangle;
/* Sphere (template/scene.h:347-394) */
typedef struct { float pos[3]; float r2, invr, r; int32_t obj_idx; int32_t material; } rt_sphere;
/* Plane (template/scene.h:401-448) */
typedef struct { float N[3]; float d; int32_t obj_idx; int32_t material; } rt_plane;

/* Light / AreaLight / DirectionalLight (template/scene.h:75-168) */
enum { RT_LIGHT_AREA = 0, RT_LIGHT_DIRECTIONAL = 1, RT_LIGHT_BASE = 2 };
typedef struct {
	int32_t kind; int32_t obj_idx;
	float pos[3]; float strength; float col[3]; float normal[3];
	float radius;    /* AreaLight */
	float sin_angle; /* DirectionalLight */
} rt_light;

/* material / diffuse / metal / glass (template/scene.h:582-676) */
enum { RT_MAT_DIFFUSE = 1, RT_MAT_METAL = 2, RT_MAT_GLASS = 3 };
typedef struct {
	int32_t type;
	int32_t raytracer; /* material::raytracer as captured at construction */
	float col[3], albedo[3];
	float specu, diffu, shinieness; int32_t N; /* diffuse */
	float ir; float absorption[3];           /* glass */
} rt_material;

/* One bvh object (bvh.h:45-86): node array, primitiveIdx, and the primitive arrays it indexes.
 * prim_idx values < n_tri address triangles, then spheres, then planes (bvh.cpp:618-627).
 * A mesh BVH (bvh(Mesh*)) has n_sph = n_pla = 0. */
typedef struct {
	const rt_bvh_node* nodes; uint32_t nodes_used;
	const uint32_t* prim_idx; uint32_t n_prims;
	const rt_triangle* tris; uint32_t n_tri;
	const rt_sphere* spheres; uint32_t n_sph;
	const rt_plane* planes; uint32_t n_pla;
} rt_blas;

/* bvhInstance (bvhInstance.h): BLAS index + matTransform + invTransform (row-major mat4) */
typedef struct { int32_t blas; float transform[16]; float inv_transform[16]; } rt_instance;

typedef struct {
	/* use_tlas == 0 (template/scene.h:1388): blas[0] is the scene BVH, nothing else is used.
	 * use_tlas != 0: tlas_nodes/instances/blas[] are traversed, brute_spheres/brute_planes are
	 * tested by brute force in FindNearest and ignored by IsOccluded (template/scene.h:1259-1263,
	 * 1288). */
	int32_t use_tlas;
	const rt_blas* blas; uint32_t n_blas;
	const rt_instance* instances; uint32_t n_instances;
	const rt_tlas_node* tlas_nodes; uint32_t tlas_nodes_used;
	const rt_sphere* brute_spheres; uint32_t n_brute_spheres;
	const rt_plane* brute_planes; uint32_t n_brute_planes;
	const rt_light* lights; uint32_t n_lights;
	const rt_material* materials; uint32_t n_materials;
	/* skydome as loaded by stbi_load(..., 3): 8-bit, sky_n channels (template/scene.h:1312-1327).
	 * sky_pixels == NULL: misses return black. */
	const uint8_t* sky_pixels; int32_t sky_w, sky_h, sky_n;
} rt_scene_desc;

/* Camera state read by Camera::GetPrimaryRay (camera.h:24-52) */
typedef struct {
	float cam_pos[3], top_left[3], top_right[3], bottom_left[3];
	int32_t fisheye; float view_angle; float y_angle;
} rt_camera;

/* result of one nearest-hit query: the fields Scene::FindNearest leaves in the Ray
 * (template/scene.h:66-72): t, objIdx (-1 = miss), material index, hitNormal */
typedef struct { float t; int32_t obj_idx; int32_t material; float normal[3]; } rt_hit;

/* work counters = the reference's DataCollector tallies (bvh.cpp:610-631) plus ray counts */
typedef struct {
	uint64_t inner_visits, prim_tests, tlas_inner, instance_visits;
	uint64_t rays_nearest, rays_occluded, brute_tests, light_tests;
} rt_counters;

/* per-kernel device time, measured with HIP events on the context's stream */
typedef struct { uint64_t launches; double ms; } rt_kernel_time;
typedef struct { rt_kernel_time generate, extend, shade, connect, query; } rt_profile;

enum { RT_MODE_WHITTED = 0, RT_MODE_PATH = 1 };

/* ---- lifetime ------------------------------------------------------------------------------ */
int rt_device_count(void);
/* Replaces Renderer::Init (renderer.cpp:5-11): allocates the float4 accumulator [width*height] in
 * HBM (zeroed) plus the path-state arrays.  Returns NULL on failure (rt_last_error(NULL)). */
rt_ctx* rt_create(int device, int width, int height);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);

/* ---- scene / camera -------------------------------------------------------------------------- */
/* Copies the scene into HBM in the traversal layout.  Replaces the pointers Scene keeps to
 * bvh / tlas / bvhInstance / primitive vectors (template/scene.h:1371-1378).
 * RT_E_UNSUPPORTED: more than 32 lights, or a TLAS with more than 256 instances. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_desc* desc);
int rt_set_camera(rt_ctx* ctx, const rt_camera* cam);
/* Scene::SetTime(t) with animation on (template/scene.h:1228-1244): every triangle of the scene BVH
 * is deformed from its ORIGINAL (uploaded) vertices -- rotation about z by a*y*0.2, a = sin(fmod(t,
 * 2*pi))/2 -- its normal re-derived (Triangle::update, template/scene.h:238-246) and the BVH refitted
 * bottom-up (bvh::Refit, bvh.cpp:556-594), all on the GPU.  Non-TLAS scenes only (the reference's
 * animOn is false under useTLAS, template/scene.h:1389).  t = 0 restores the uploaded geometry. */
int rt_set_time(rt_ctx* ctx, float t);

/* ---- the pixel loop ---------------------------------------------------------------------------- */
/* Replaces the pixel loop of Renderer::Tick (renderer.cpp:259-285) for frames
 * [frame0, frame0 + nframes) and rows [y0, y1): per pixel, Camera::GetPrimaryRay then
 * Renderer::Trace (mode RT_MODE_WHITTED, renderer.cpp:21-126; nframes must be 1) or
 * Renderer::Sample (RT_MODE_PATH, renderer.cpp:128-236), accumulated into the accumulator exactly
 * as :270 / :279-282 do.  The random stream of pixel p in frame f starts at
 * InitSeed(seed_base + p + f*width*height) (template/template.cpp:680-683; the single index whose
 * hash is 0 starts at 0x9E3779B9 instead, because xorshift32 cannot leave state 0).
 * max_depth is the depth argument of Trace (4 at renderer.cpp:269); Sample always starts at 4.
 * In RT_MODE_PATH a scene with a diffuse material that has shinieness != 0 or raytracer == 0 (random
 * draws then interleave with shadow queries in depth-first order) is rendered by a slower
 * one-lane-per-sample kernel instead of the wavefront kernels; results follow the same definition. */
int rt_render(rt_ctx* ctx, int mode, uint32_t frame0, int nframes, uint32_t seed_base, int y0, int y1, int max_depth);
/* Same for the rows row_first + k*row_stride, k < row_count: the row-interleaved pixel shard used
 * when the frame is split over several GPUs (rank r of n renders row_first = r, row_stride = n). */
int rt_render_rows(rt_ctx* ctx, int mode, uint32_t frame0, int nframes, uint32_t seed_base, int row_first, int row_stride, int row_count, int max_depth);
/* memset of the accumulator (renderer.cpp:9, :274) */
int rt_clear(rt_ctx* ctx);
/* rows [y0, y1) of the float4 accumulator -> host (Renderer::accumulator, renderer.h:98) */
int rt_download_accumulator(rt_ctx* ctx, int y0, int y1, float* out);
/* screen->pixels for rows [y0, y1): RGBF32_to_RGB8(accumulator / iteration) (renderer.cpp:287-290,
 * template/precomp.h:445-448) */
int rt_resolve(rt_ctx* ctx, int iteration, int y0, int y1, uint32_t* rgb8_out);
/* Device address of the accumulator (width*height float4), and rebinding it to caller-owned device
 * memory (e.g. a torch tensor handed to an RCCL gather).  The caller keeps that memory alive. */
void* rt_accumulator_device_ptr(rt_ctx* ctx);
int rt_bind_accumulator(rt_ctx* ctx, void* device_ptr);
/* Multi-GPU from one process (one host thread + one rt_ctx per GPU): copy the accumulator rows
 * row_first + k*row_stride, k < row_count, that context src rendered into the same rows of context dst's
 * accumulator -- device to device over xGMI (peer access is enabled on first use; without it the rows are
 * copied one by one through hipMemcpyPeerAsync).  Asynchronous: the copy is a push queued on src's stream behind
 * the rendering already queued there (several sources push over their own links at the same time) AND behind
 * whatever was queued on dst's stream when the call was made (rt_clear's memset, dst's own resolve of the frame
 * before: src's stream waits for an event recorded on dst's stream first), and dst's stream is made to wait for
 * it, so whatever is queued on dst afterwards (rt_resolve, rt_download_accumulator, the next frame) sees the
 * rows; the host does not wait.  src's accumulator must not be re-bound or freed before dst has synchronised.
 * May be called from src's host thread while another thread drives dst (only stream operations touch dst; EVERY
 * error of the call, the argument checks included, is reported on src: rt_last_error(src)).  Both contexts
 * must have the same width and height.  rt_device_of: the HIP device a context lives on.
 * rt_gather_begin(dst): called by dst's owner ONCE PER FRAME, after whatever must precede the frame's rows on dst (rt_clear, the
 * resolve of the frame before) and BEFORE dst's own share of the frame is queued: it marks "dst's rows are free" on dst's stream,
 * and every rt_gather_rows into dst until the next rt_gather_begin waits for that mark only -- not for dst's own rendering of
 * the frame, and not for the pushes of the sources that called earlier: the pushes then run side by side, each over its own
 * xGMI link.  Without it (never called on dst) every gather orders itself behind all that dst's stream held when it was
 * called, which is correct and serialises the pushes behind dst's rendering. */
int rt_gather_begin(rt_ctx* dst);
int rt_gather_rows(rt_ctx* dst, rt_ctx* src, int row_first, int row_stride, int row_count);
int rt_device_of(const rt_ctx* ctx);
/* PCI address ("0000:c1:00.0") of HIP device 'device' into out[cap >= 16]: the ranks of a multi-process run exchange these to
 * prove that no two of them render on the same GPU (bench.py ranks_devices). */
int rt_device_pci_bus_id(int device, char* out, int cap);

/* ---- batch queries ---------------------------------------------------------------------------- */
/* Scene::FindNearest(ray, t_min) (template/scene.h:1248-1267) for n rays. O, D: n*3 floats;
 * tmax: n floats or NULL (1e34f, the Ray constructor default, template/scene.h:42). */
int rt_intersect_batch(rt_ctx* ctx, int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out);
/* Scene::IsOccluded(ray) (template/scene.h:1286-1291) for n rays; out[i] = 0 / 1 */
int rt_occluded_batch(rt_ctx* ctx, int n, const float* O, const float* D, const float* tmax, uint8_t* out);
/* The queries BELOW Scene level, with the reference's own names: scope
 *   RT_SCOPE_SCENE     Scene::FindNearest / Scene::IsOccluded (what the two calls above do)
 *   RT_SCOPE_ACCEL     the scene's accelerator alone: bvh::Intersect / IsOccluded of the scene bvh (bvh.cpp:596-604) or
 *                      tlas::Intersect / IsOccluded (tlas.cpp:65-122): no lights, no brute-force primitives
 *   RT_SCOPE_BLAS      bvh::Intersect / IsOccluded of BLAS 'index' in its own object space
 *   RT_SCOPE_INSTANCE  bvhInstance::BIntersect / IsOccluded of instance 'index' (bvhInstance.cpp:3-35): ray to object
 *                      space, normal back to world
 * t_min matters for RT_SCOPE_SCENE only (the BVH uses 0.0001, bvh.cpp:607). */
enum { RT_SCOPE_SCENE = 0, RT_SCOPE_ACCEL = 1, RT_SCOPE_BLAS = 2, RT_SCOPE_INSTANCE = 3 };
int rt_intersect_scope(rt_ctx* ctx, int scope, int index, int n, const float* O, const float* D, const float* tmax, float t_min, rt_hit* out);
int rt_occluded_scope(rt_ctx* ctx, int scope, int index, int n, const float* O, const float* D, const float* tmax, uint8_t* out);
/* Scene::GetSkyColor (template/scene.h:1312-1327) for n directions (D: n*3 floats, rgb_out: n*3 floats) */
int rt_sky_color_batch(rt_ctx* ctx, int n, const float* D, float* rgb_out);
/* Camera::GetPrimaryRay(x, y) + Scene::FindNearest(t_min) for every pixel ("primary rays only") */
int rt_primary_hits(rt_ctx* ctx, float t_min, int32_t* obj_idx_out, float* t_out);
/* Renderer::Trace / Renderer::Sample on caller-supplied rays: rgb_out[n*3].  Stream i starts at
 * InitSeed(seed_base + i). */
int rt_trace_batch(rt_ctx* ctx, int mode, int n, const float* O, const float* D, int depth, uint32_t seed_base, float* rgb_out);
/* scene.raytracer as the caller's Scene holds it, for rt_trace_batch*: -1 (default) the flag follows the function called -- Trace with the
 * flag set, Sample with it clear, as Renderer::Tick calls them (renderer.cpp:268-283); 0 / 1: the flag's value.  Trace with the flag clear
 * (Russian roulette, sampled light positions, an indirect child: renderer.cpp:33-43, 107-121) and Sample with it set (:143-153) are what the
 * reference's Renderer::Trace / Sample compute when called that way; the device evaluates them one lane per call tree (slow path,
 * correctness only; not with the Q-learning sampler on: RT_E_UNSUPPORTED).  rt_render ignores the flag (it is Tick's loop). */
int rt_set_scene_raytracer(rt_ctx* ctx, int flag);
/* The same with Trace / Sample's third argument: energy[3] instead of float3(1) */
int rt_trace_batch_energy(rt_ctx* ctx, int mode, int n, const float* O, const float* D, int depth, uint32_t seed_base, const float* energy, float* rgb_out);

/* ---- Q-learning guided sampling ("next" row N4) ------------------------------------------------------
 * The reference snapshot has no code for it: README.md:36-42 names Dahm & Keller 2017, "Learning Light Transport
 * the Reinforced Way", and lists "initialize sampling positions; pick sampling direction according to the QValue of neighboring
 * points; store and update directions with a corresponding probability per sampling point".  These calls are this library's
 * statement of that scheme (csrc/rt_qlearn.h; PARITY UNPINNED: there is nothing in the reference to compare with).
 * rt_qlearn_enable: path-mode batches draw the indirect bounce of every diffuse hit from a table of grid^3 cells over the box
 *   [lo, hi] x 64 direction patches, P(patch) = (1 - epsilon) Q / sum Q + epsilon / 64, and collect rewards (integer sums);
 *   the table starts at q_init everywhere.  params == NULL switches the sampler off and frees the table.  Batches larger than
 *   the slot budget, and scenes that need the general path kernel, return RT_E_UNSUPPORTED while it is on.
 * rt_qlearn_apply: Q <- (1 - alpha) Q + alpha * mean reward, for every (cell, patch) that received one; call it BETWEEN
 *   batches (within a batch the table is read-only, so a frame does not depend on scheduling or sharding).
 * rt_qlearn_get_sums / rt_qlearn_set_sums: the pending rewards (int64 sums in 48.16 fixed point, uint32 counts, grid^3 * 64
 *   each) -- with several ranks the sums are all-reduced between the ranks before every rank applies them.
 * rt_qlearn_get_table: Q as grid^3 * 64 floats (cell-major, patch = 8 * band + sector). */
typedef struct {
	int32_t grid; float lo[3], hi[3]; float alpha, epsilon, q_init;
	/* learn_mask: a sample pays rewards iff (the state of its random stream after the pixel jitter) & learn_mask == 0 -- 0: every
	 * sample learns; 3: every fourth one (a surface hit's reward reads 64 table values: the picks stay guided for all samples,
	 * the table is taught by a fixed, scheduling-independent quarter of them) */
	uint32_t learn_mask;
} rt_qlearn_params;
int rt_qlearn_enable(rt_ctx* ctx, const rt_qlearn_params* params);
int rt_qlearn_apply(rt_ctx* ctx);
int rt_qlearn_get_sums(rt_ctx* ctx, int64_t* sums_out, uint32_t* counts_out);
int rt_qlearn_set_sums(rt_ctx* ctx, const int64_t* sums, const uint32_t* counts);
int rt_qlearn_get_table(rt_ctx* ctx, float* q_out);
/* The pending sums live in the caller's DEVICE arrays from here on (grid^3 * 64 int64 / uint32; what was pending is copied over;
 * the caller keeps them alive until the sampler is switched off): one process per GPU all-reduces them in place between the
 * ranks (RCCL on device memory) before every rank's rt_qlearn_apply -- no host copy in the exchange. */
int rt_qlearn_bind_sums(rt_ctx* ctx, int64_t* dev_sums, uint32_t* dev_counts);

/* ---- acceleration structure build ("next" row N1) ------------------------------------------------ */
/* bvh::Build() with splitMethod BINNEDSAH (bvh.cpp:18-56; FindBestSplitPlane :116-193, Subdivide :223-333,
 * separatePlanes :202-221, Refit :556-594) on the device.  The result is the reference's tree bit for bit: node
 * numbering, boxes and primitiveIdx order.  nodes_out: 2 * (n_tri + n_sph + n_pla + 1) entries (bvh.cpp:33),
 * prim_idx_out: n_tri + n_sph + n_pla entries.  RT_E_UNSUPPORTED (never a silent fallback) for an input with
 * no triangle or sphere, or with a non-finite vertex / centre / radius. */
int rt_build_bvh(rt_ctx* ctx, const rt_triangle* tris, uint32_t n_tri, const rt_sphere* spheres, uint32_t n_sph, const rt_plane* planes, uint32_t n_pla,
                 rt_bvh_node* nodes_out, uint32_t* prim_idx_out, uint32_t* nodes_used_out);
/* The same with bvh.h:38-43's splitMethod: 0 BINNEDSAH (what rt_build_bvh builds), 1 SAMESIZE (median of the longest
 * axis, bvh.cpp:233-252), 2 LONGESTAXIS (spatial middle, :226-232), 3 SAH (every centroid tried as the plane, :275-293
 * with EvaluateSAH :514-554 -- quadratic in the node size like the reference). */
int rt_build_bvh_split(rt_ctx* ctx, int split_method, const rt_triangle* tris, uint32_t n_tri, const rt_sphere* spheres, uint32_t n_sph,
                       const rt_plane* planes, uint32_t n_pla, rt_bvh_node* nodes_out, uint32_t* prim_idx_out, uint32_t* nodes_used_out);
/* tlas::build (tlas.cpp:13-48: agglomerative clustering by smallest union surface area, FindBestMatch :50-63) on the
 * device.  bounds6: per instance the world box its bvhInstance holds (min.xyz, max.xyz, template bvhInstance.cpp:37-44);
 * 1 <= n <= 256; nodes_out has room for 2 n + 1 nodes.  Same node order and boxes as the host builder. */
int rt_build_tlas(rt_ctx* ctx, const float* bounds6, uint32_t n, rt_tlas_node* nodes_out, uint32_t* nodes_used_out);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Kernels tally rt_counters (slower; keep off when timing).
 *   RT_COUNT_REFERENCE  the walk bvh::BIntersect / tlas::Intersect make: the DataCollector tallies
 *                       (bvh.cpp:610-631), identical to the oracle's
 *   RT_COUNT_EXECUTED   the walk the timed kernels make: they skip TLAS children whose geometry the ray
 *                       cannot reach, so fewer instance / node visits; same results */
#define RT_COUNT_OFF 0
#define RT_COUNT_REFERENCE 1
#define RT_COUNT_EXECUTED 2
int rt_set_counting(rt_ctx* ctx, int counting);
int rt_get_counters(rt_ctx* ctx, rt_counters* out, int reset);
/* the same tallies kept apart: nearest-hit queries (extend kernel) / occlusion queries (connect) */
int rt_get_counters_split(rt_ctx* ctx, rt_counters* nearest, rt_counters* occluded, int reset);
/* profiling != 0: HIP events bracket every kernel launch on the context's stream */
int rt_set_profiling(rt_ctx* ctx, int profiling);
int rt_get_profile(rt_ctx* ctx, rt_profile* out, int reset);
int rt_synchronize(rt_ctx* ctx);
/* What the library was built with (the -D flags given to the build and the compile-time tuning macros), and the tuning a
 * context resolved from its environment at rt_create (RT_* variables): measurement files are stamped with both, so that
 * counters taken on one build / tuning are not priced against timings of another.  Static / context-owned strings. */
const char* rt_build_info(void);
const char* rt_tuning_info(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
