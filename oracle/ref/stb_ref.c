/* ORACLE/_ref (test infrastructure): the reference's own vendored image loader, compiled from
 * the source WHERE IT LIES (/root/reference/lib/stb_image.h, stb_image v2.27) -- nothing is copied
 * into this repository.  It is the only piece of the reference's hot-path inputs that builds in
 * this image without stand-ins: Scene loads its skydome with stbi_load(path, &w, &h, &n, 3)
 * (template/scene.h:792 and the other scene factories).  Used by tests/test_sky_hdr.py to pin the
 * host's .hdr loader (8-bit LDR conversion of Radiance RGBE files) against the real thing. */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_ONLY_HDR
#define STBI_NO_LINEAR
#include "/root/reference/lib/stb_image.h"

unsigned char* ref_stbi_load(const char* path, int* w, int* h, int* n, int req) { return stbi_load(path, w, h, n, req); }
void ref_stbi_free(void* p) { stbi_image_free(p); }
