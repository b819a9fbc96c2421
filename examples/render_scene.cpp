// Headless C++ host over the rapt:: mirror of the reference's interface: load a scene description,
// run Renderer::Tick like the reference's main loop does (template/template.cpp:266-270), write the
// resolved frame as a binary PPM.  Everything below Renderer / Scene goes through the C ABI of
// include/rt_amd.h.
//
//   g++ -std=c++17 -O2 examples/render_scene.cpp -Iray-and-pathtracer_amd/host \
//       -Lray-and-pathtracer_amd/host -lrapt_host -Lray-and-pathtracer_amd/csrc -lrt_amd \
//       -Wl,-rpath,$PWD/ray-and-pathtracer_amd/host -Wl,-rpath,$PWD/ray-and-pathtracer_amd/csrc -o render_scene
//   ./render_scene scene.rapt out.ppm 640 360 path 16 [all | d0,d1,...]
// The optional last argument spreads every Tick over several GPUs (one rt_ctx and one host thread per device,
// rows interleaved, rt_gather_rows into device d0): "all" = every visible device.
#include "rapt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

using namespace rapt;

int main(int argc, char** argv)
{
	if (argc < 7) { fprintf(stderr, "usage: %s scene.rapt out.ppm width height whitted|path frames [all | d0,d1,...]\n", argv[0]); return 2; }
	const int w = atoi(argv[3]), h = atoi(argv[4]), frames = atoi(argv[6]);
	const bool path = strcmp(argv[5], "path") == 0;
	try {
		Renderer app(w, h, 0);
		if (argc > 7) {
			if (strcmp(argv[7], "all") == 0) app.UseAllDevices();
			else {
				std::vector<int> devs;
				for (const char* p = argv[7]; *p;) { devs.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p) p++; }
				app.UseDevices(devs);
			}
		}
		app.Init();                          // allocates the accumulator on the GPU(s)
		app.scene.LoadFile(argv[1]);         // meshes, materials, lights, BVH / TLAS (host builders)
		if (path) app.scene.toogleRaytracer(); // key 'P' in the reference: path tracing, lights sampled
		app.Commit();                        // flatten + rt_upload_scene on every context
		for (int f = 0; f < (path ? frames : 1); f++) app.Tick(0.0f);
		FILE* out = fopen(argv[2], "wb");
		if (!out) { perror(argv[2]); return 1; }
		fprintf(out, "P6\n%d %d\n255\n", w, h);
		for (int i = 0; i < w * h; i++) {
			const uint32_t p = app.screenPixels[i]; // 0x00RRGGBB, as Surface::pixels
			const unsigned char rgb[3] = { (unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p };
			fwrite(rgb, 1, 3, out);
		}
		fclose(out);
		app.Shutdown();
	} catch (const std::exception& e) {
		fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
