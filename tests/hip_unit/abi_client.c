/* Plain-C client of the drop-in boundary: no torch, no Python, no C++ -- only include/g4s_rasterizer.h,
 * the HIP runtime's C API and libg4s_hip.so, exactly what a non-Python host (or the reference's
 * rasterize_points.cpp of INTEGRATION.md) would link.  tests/test_gpu_abi_client.py writes the inputs,
 * runs this program on the GPU box and checks the outputs it writes against the CPU oracle.
 *
 *   gcc -std=c99 tests/hip_unit/abi_client.c -I/opt/rocm/include -Iinclude -Lg4splat_amd -lg4s_hip \
 *       -L/opt/rocm/lib -lamdhip64 -o abi_client
 *   abi_client <in.bin> <out.bin>
 *
 * in.bin : int32 P, D, M, W, H; float32 tanfovx, tanfovy, scale_modifier; then float32 arrays
 *          bg[3] means3D[3P] sh[3MP] opacity[P] scales[2P] rotations[4P] view[16] proj[16] campos[3]
 *          dL_dcolor[3HW] dL_dothers[7HW]
 * out.bin: int32 R; float32 color[3HW] others[7HW]; int32 radii[P]; float32 dmean2D[3P] dopacity[P]
 *          dmean3D[3P] dsh[3MP] dscale[2P] drot[4P]; uint8 present[P]; float32 knn[P]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "g4s_rasterizer.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

/* std::function<char*(size_t)> of the reference -> (fn, ctx): ctx is a slot holding the current allocation */
typedef struct { char* ptr; size_t cap; int calls; } Chunk;
static char* resize_chunk(void* ctx, size_t n) {
    Chunk* c = (Chunk*)ctx;
    c->calls++;
    if (n > c->cap) {
        if (c->ptr) hipFree(c->ptr);
        if (hipMalloc((void**)&c->ptr, n) != hipSuccess) return NULL;
        c->cap = n;
    }
    return c->ptr ? c->ptr : (char*)1;
}

static float* upload(FILE* f, size_t n) {
    float* h = (float*)malloc(n * 4 + 4);
    float* d = NULL;
    if (fread(h, 4, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    if (hipMalloc((void**)&d, n * 4 + 4) != hipSuccess) exit(4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    free(h);
    return d;
}
static void download(FILE* f, const void* d, size_t bytes) {
    void* h = malloc(bytes + 4);
    hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
    fwrite(h, 1, bytes, f);
    free(h);
}
static void* dalloc(size_t bytes) { void* p = NULL; if (hipMalloc(&p, bytes + 4) != hipSuccess) exit(4); return p; }

int main(int argc, char** argv) {
    if (argc != 3) return 1;
    FILE* in = fopen(argv[1], "rb");
    if (!in) return 1;
    int hdr[5]; float sc[3];
    if (fread(hdr, 4, 5, in) != 5 || fread(sc, 4, 3, in) != 3) return 3;
    const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4];
    const size_t N = (size_t)W * H;
    float* bg = upload(in, 3); float* means = upload(in, 3 * (size_t)P); float* sh = upload(in, 3 * (size_t)M * P);
    float* opa = upload(in, P); float* scales = upload(in, 2 * (size_t)P); float* rots = upload(in, 4 * (size_t)P);
    float* view = upload(in, 16); float* proj = upload(in, 16); float* campos = upload(in, 3);
    float* gcol = upload(in, 3 * N); float* goth = upload(in, 7 * N);
    fclose(in);

    hipStream_t stream;
    CK(hipStreamCreate(&stream));  /* a non-default stream: nothing may depend on the legacy stream */
    float* color = (float*)dalloc(3 * N * 4); float* others = (float*)dalloc(7 * N * 4);
    int* radii = (int*)dalloc((size_t)P * 4);
    Chunk geom = {0}, binning = {0}, img = {0};
    const int R = g4s_rasterizer_forward(resize_chunk, &geom, resize_chunk, &binning, resize_chunk, &img, P, D, M, bg, W, H,
                                         means, sh, NULL, opa, scales, sc[2], rots, NULL, view, proj, campos, sc[0], sc[1], 0,
                                         color, others, radii, 0, stream);
    if (R < 0) { fprintf(stderr, "forward: %s\n", g4s_last_error()); return 5; }
    if (geom.calls != 1 || binning.calls != 1 || img.calls != 1) { fprintf(stderr, "callback protocol\n"); return 6; }

    const size_t wsb = g4s_rasterizer_backward_workspace(P, R);
    char* ws = (char*)dalloc(wsb);
    float* dm2 = (float*)dalloc((size_t)P * 12); float* dop = (float*)dalloc((size_t)P * 4);
    float* dcol = (float*)dalloc((size_t)P * 12); float* dm3 = (float*)dalloc((size_t)P * 12);
    float* dT = (float*)dalloc((size_t)P * 36); float* dsh = (float*)dalloc((size_t)P * M * 12);
    float* dsc = (float*)dalloc((size_t)P * 8); float* drot = (float*)dalloc((size_t)P * 16);
    int rc = g4s_rasterizer_backward(P, D, M, R, bg, W, H, means, sh, NULL, scales, sc[2], rots, NULL, view, proj, campos,
                                     sc[0], sc[1], radii, geom.ptr, binning.ptr, img.ptr, gcol, goth, dm2, NULL, dop, dcol,
                                     dm3, dT, dsh, dsc, drot, ws, wsb, 0, stream);
    if (rc != G4S_OK) { fprintf(stderr, "backward: %s\n", g4s_last_error()); return 7; }

    unsigned char* present = (unsigned char*)dalloc(P);
    if (g4s_rasterizer_mark_visible(P, means, view, proj, present, stream) != G4S_OK) return 8;
    float* knn = (float*)dalloc((size_t)P * 4);
    const size_t kwb = g4s_knn_workspace(P);
    char* kws = (char*)dalloc(kwb);
    if (g4s_knn_mean_dist(P, means, knn, kws, kwb, stream) != G4S_OK) return 9;
    CK(hipStreamSynchronize(stream));

    /* error paths return codes + messages, never abort */
    if (g4s_rasterizer_forward(resize_chunk, &geom, resize_chunk, &binning, resize_chunk, &img, P, D, M, bg, -1, H, means, sh,
                               NULL, opa, scales, 1.0f, rots, NULL, view, proj, campos, sc[0], sc[1], 0, color, others, radii,
                               0, stream) != G4S_ERR_INVALID_ARGUMENT || strlen(g4s_last_error()) == 0) return 10;
    if (g4s_rasterizer_backward(P, D, M, R, bg, W, H, means, sh, NULL, scales, 1.0f, rots, NULL, view, proj, campos, sc[0],
                                sc[1], radii, geom.ptr, binning.ptr, img.ptr, gcol, goth, dm2, NULL, dop, dcol, dm3, dT, dsh,
                                dsc, drot, ws, 16, 0, stream) != G4S_ERR_INVALID_ARGUMENT) return 11;

    FILE* out = fopen(argv[2], "wb");
    if (!out) return 1;
    fwrite(&R, 4, 1, out);
    download(out, color, 3 * N * 4); download(out, others, 7 * N * 4); download(out, radii, (size_t)P * 4);
    download(out, dm2, (size_t)P * 12); download(out, dop, (size_t)P * 4); download(out, dm3, (size_t)P * 12);
    download(out, dsh, (size_t)P * M * 12); download(out, dsc, (size_t)P * 8); download(out, drot, (size_t)P * 16);
    download(out, present, P); download(out, knn, (size_t)P * 4);
    fclose(out);
    printf("abi_client ok: %s, R = %d\n", g4s_version(), R);
    return 0;
}
