// extern "C" entry points of libg4s_hip.so (see include/g4s_rasterizer.h) and the host-side
// sequencing of the kernels.  Mirrors Rasterizer::forward / ::backward / ::markVisible
// (dsr/cuda_rasterizer/rasterizer_impl.cu:141-153,198-448) and SimpleKNN::knn
// (knn/simple_knn.cu:185-221).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "g4s_internal.h"

using namespace g4s;

extern "C" int g4s_knn_launch_internal(int P, const float* points, float* meanDists, char* workspace, hipStream_t s);
extern "C" void g4s_maps_launch_internal(int fwd, int W, int H, float depth_ratio, const float* allmap, const float* wvt,
                                         const float* fpt, float* cam, float* const* outs, const float* surf_depth_in,
                                         const float* const* grads, float* g_allmap, hipStream_t s);

namespace {

thread_local char t_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

// One pinned, device-mapped word block per host thread for the read-back of the instance counts: the totals kernel
// stores them straight into host memory (no copy launch between it and the event the host waits on).
// hipHostMallocPortable: the same host thread may drive several devices (one process, eight GPUs), and only a portable
// allocation is pinned / mapped for all of them.
uint32_t* pinned_word() {
    thread_local uint32_t* p = nullptr;
    if (!p) {
        if (hipHostMalloc((void**)&p, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) p = nullptr;
    }
    return p;
}
// its address as the current device sees it
uint32_t* pinned_word_device(uint32_t* host) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) return nullptr;
    return (uint32_t*)d;
}

// One event per host thread and device, recorded behind the read-back copy: the forward waits on it instead of on
// the whole stream, so work queued after the copy keeps the GPU busy while the host reads the totals.
hipEvent_t readback_event() {
    thread_local hipEvent_t ev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ev[dev]) {
        if (hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
    }
    return ev[dev];
}

// Diagnostic switches (include/g4s_rasterizer.h: g4s_set_option).  Plain process-wide integers: read on every call,
// written only by g4s_set_option -- the call paths never touch the environment.
struct Option { const char* name; std::atomic<int> value; };
Option g_options[] = {{"box_only", {0}}, {"no_fastpath", {0}}, {"bwd_fwd_order", {0}},
                      {"bwd_hot_threshold", {G4S_OPTION_UNSET}}, {"no_side_zero", {0}}};
enum OptionId { OPT_BOX_ONLY, OPT_NO_FASTPATH, OPT_BWD_FWD_ORDER, OPT_BWD_HOT_THRESHOLD, OPT_NO_SIDE_ZERO };
inline int opt(OptionId id) { return g_options[id].value.load(std::memory_order_relaxed); }

inline bool trace_on() {
    static const bool on = getenv("G4S_TRACE") != nullptr;
    return on;
}
inline bool misaligned(const void* p, size_t a) { return p != nullptr && ((size_t)p % a) != 0; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(G4S_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// CHECK_CUDA of the reference (auxiliary.h:295-302): launch errors always, sync + check in debug.
// G4S_TRACE=1 in the environment: synchronise after every stage and log it to stderr.
#define CHECK_LAUNCH(what)                                                                          \
    do {                                                                                            \
        hipError_t _e = hipGetLastError();                                                          \
        if (_e == hipSuccess && (debug || trace_on())) _e = hipStreamSynchronize(stream);           \
        if (trace_on()) { fprintf(stderr, "[g4s] %s: %s\n", what, hipGetErrorString(_e)); fflush(stderr); } \
        if (_e != hipSuccess) return fail(G4S_ERR_HIP, "%s: %s", what, hipGetErrorString(_e));      \
    } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline) ----
enum ProfId { PF_PREPROCESS_FWD, PF_DEPTH_SORT, PF_COUNT_SCAN, PF_EMIT, PF_TILE_SORT, PF_TILE_RANGES, PF_BLEND_FWD,
              PF_BLEND_BWD, PF_PREPROCESS_BWD, PF_MAPS_FWD, PF_MAPS_BWD, PF_PHOTO_LOSS, PF_ADAM, PF_GEO_REG, PF_COUNT };
const char* const kProfNames[PF_COUNT] = {"preprocess_fwd", "depth_sort", "count_scan", "emit", "tile_sort",
                                          "tile_ranges",    "blend_fwd",  "blend_bwd",  "preprocess_bwd",
                                          "maps_fwd",       "maps_bwd",   "photometric_loss",
                                          "adam", "geometry_regularizers"};
struct ProfRec { int id; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;          // recorded (kernel group, start, stop)
std::vector<hipEvent_t> g_prof_pool;  // events are created up front, never inside a timed region
size_t g_prof_next = 0;
constexpr size_t PROF_POOL = 16384;

struct ProfScope {
    int id; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on = false;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) {
        if (!g_prof_on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_next + 2 > g_prof_pool.size()) return;  // pool exhausted: stop recording
        a = g_prof_pool[g_prof_next++];
        b = g_prof_pool[g_prof_next++];
        on = hipEventRecord(a, s) == hipSuccess;
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(ProfRec{id, a, b});
    }
};

// Bits of the tile field the partition sorts on: the reference's getHigherMsb(tiles) (rasterizer_impl.cu:301), capped
// at the 32 bits the field has.
uint32_t higher_msb(uint32_t n) {  // rasterizer_impl.cu:35-50
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
int tile_sort_bits(int tiles) {
    const int b = (int)higher_msb((uint32_t)tiles);
    return b < 32 ? b : 32;
}

}  // namespace

extern "C" const char* g4s_last_error(void) { return t_err; }

extern "C" void g4s_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on && g_prof_pool.empty()) {
        g_prof_pool.reserve(PROF_POOL);
        for (size_t i = 0; i < PROF_POOL; i++) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            g_prof_pool.push_back(e);
        }
    }
    g_prof_on = on != 0;
}
extern "C" int g4s_profile_kernels(void) { return PF_COUNT; }
extern "C" const char* g4s_profile_name(int id) { return (id >= 0 && id < PF_COUNT) ? kProfNames[id] : ""; }
// Sum of the recorded durations of kernel group `id` (ms) and the number of recordings.
// Synchronises on the recorded events.  g4s_profile_reset() drops all recordings.
extern "C" int g4s_profile_read(int id, double* total_ms, int* count) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0; int n = 0;
    for (const ProfRec& r : g_prof) {
        if (r.id != id) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return G4S_ERR_HIP;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return G4S_ERR_HIP;
        tot += ms; n++;
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return G4S_OK;
}
extern "C" void g4s_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.clear();
    g_prof_next = 0;  // the pooled events are reused
}
extern "C" int g4s_set_option(const char* name, int value) {
    for (Option& o : g_options)
        if (name && !strcmp(name, o.name)) { o.value.store(value, std::memory_order_relaxed); return G4S_OK; }
    return fail(G4S_ERR_INVALID_ARGUMENT, "unknown option '%s'", name ? name : "(null)");
}
extern "C" int g4s_get_option(const char* name, int* value) {
    for (Option& o : g_options)
        if (name && !strcmp(name, o.name)) { if (value) *value = o.value.load(std::memory_order_relaxed); return G4S_OK; }
    return fail(G4S_ERR_INVALID_ARGUMENT, "unknown option '%s'", name ? name : "(null)");
}
#ifndef G4S_BUILD_ID
#define G4S_BUILD_ID "unknown"
#endif
// "... build <id>": the id is the digest of the library's sources (csrc/Makefile)
extern "C" const char* g4s_version(void) { return "g4s-hip 0.1.0 gfx950 build " G4S_BUILD_ID; }

extern "C" int g4s_rasterizer_layout(int P, int R, int width, int height, g4s_layout* out) {
    if (!out || P < 0 || R < 0 || width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "bad layout query");
    const int tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    const GeomLayout g = geom_layout((size_t)P);
    const BinLayout b = bin_layout((size_t)R);
    const ImgLayout im = img_layout((size_t)width * height, (size_t)tiles);
    // which ping-pong half holds the results is fixed by the (even / data-independent) pass counts
    const int tile_bits = tile_sort_bits(tiles);
    const int passes = (tile_bits + 7) / 8;
    out->rec = g.rec;
    out->clamped = g.clamped;
    out->depth_sorted = g.vals_b;  // wherever the sort starts and however many passes it takes (g4s_rasterizer_forward)
    out->tiles_touched = g.tiles_touched;
    out->geom_bytes = g.bytes;
    out->entries = (passes & 1) ? b.ent_b : b.ent_a;
    out->qhit = b.qhit;
    out->binning_bytes = b.bytes;
    out->ranges = im.ranges;
    out->final_T = im.final_T;
    out->n_contrib = im.n_contrib;
    out->tile_order = im.tile_order;
    out->image_bytes = im.bytes;
    out->hot_count = im.hot_count;
    return G4S_OK;
}

// Fixed buffers as "resize callbacks" (g4s_rasterizer_forward_presized): the callback hands the caller's chunk out
// if it is large enough.
struct FixedChunk { char* ptr; size_t bytes; };
char* fixed_chunk_cb(void* ctx, size_t n) {
    FixedChunk* c = (FixedChunk*)ctx;
    return n <= c->bytes ? c->ptr : nullptr;
}

// shs_rest == NULL: shs is the packed [P,M,3] tensor; otherwise shs = [P,1,3] and shs_rest = [P,M-1,3].
// capacity >= 0: the presized, host-synchronisation-free form -- the binning chunk holds `capacity` instances, the
// instance counts stay on the device (status_dev), nothing is read back.
static int rasterizer_forward_impl(
    int capacity, uint32_t* status_dev,
    g4s_resize_fn geometry_buffer, void* geometry_ctx, g4s_resize_fn binning_buffer, void* binning_ctx,
    g4s_resize_fn image_buffer, void* image_ctx, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* scales,
    float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
    float* out_others, int* radii, int debug, void* stream_) {
    (void)tan_fovx; (void)tan_fovy; (void)prefiltered;
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0 || width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P, width, height must be positive");
    if (!geometry_buffer || !binning_buffer || !image_buffer)
        return fail(G4S_ERR_INVALID_ARGUMENT, "resize callbacks must not be NULL");
    if (!background || !viewmatrix || !projmatrix || !cam_pos || !out_color || !out_others)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    const int tiles_x = (width + TILE - 1) / TILE, tiles_y = (height + TILE - 1) / TILE;
    // (tile coordinates travel in 16 bits each -- the binned rect of a Gaussian -- and the tile id in 32)
    if (tiles_x > 65535 || tiles_y > 32767 || (long long)tiles_x * tiles_y > 0x7FFFFFFFll)
        return fail(G4S_ERR_INVALID_ARGUMENT, "image too large: %d x %d tiles (at most 65535 across, 32767 down)", tiles_x, tiles_y);
    const int tiles = tiles_x * tiles_y;
    const size_t N = (size_t)width * height;

    // image chunk first: with P == 0 the frame is still background (rasterize_points.cu:85-99
    // returns zero-filled outputs in that case; the binding handles P == 0 itself)
    const ImgLayout IL = img_layout(N, (size_t)tiles);
    char* img = image_buffer(image_ctx, IL.bytes);
    if (!img) return fail(G4S_ERR_ALLOC, "image buffer callback returned NULL");
    img = align_ptr(img);
    uint32_t* ranges = (uint32_t*)(img + IL.ranges);
    float* final_T = (float*)(img + IL.final_T);
    uint32_t* n_contrib = (uint32_t*)(img + IL.n_contrib);
    if (P <= 0) HIP_TRY(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, stream));  // rasterizer_impl.cu:311 (P > 0: cleared by the totals scan)
    const bool presized = capacity >= 0;
    if (presized && status_dev && P <= 0) HIP_TRY(hipMemsetAsync(status_dev, 0, 16, stream));

    int R = 0;
    const float* rec_ptr = nullptr;
    const uint64_t* entries_ptr = nullptr;
    uint8_t* qhit_ptr = nullptr;
    if (P > 0) {
        if (!means3D || !opacities) return fail(G4S_ERR_INVALID_ARGUMENT, "means3D / opacities must not be NULL");
        if (!shs && !colors_precomp)  // NUM_CHANNELS == 3 here; mirrors rasterizer_impl.cu:243-246
            return fail(G4S_ERR_UNSUPPORTED, "provide SHs or precomputed colours");
        if (!transMat_precomp && (!scales || !rotations))
            return fail(G4S_ERR_INVALID_ARGUMENT, "provide scales+rotations or transMat_precomp");
        if (misaligned(rotations, 16) || misaligned(scales, 8))
            return fail(G4S_ERR_INVALID_ARGUMENT, "rotations must be 16-byte and scales 8-byte aligned");
        if (shs && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
            return fail(G4S_ERR_INVALID_ARGUMENT, "SH degree %d does not fit M = %d coefficients", D, M);

        const GeomLayout GL = geom_layout((size_t)P);
        char* geom = geometry_buffer(geometry_ctx, GL.bytes);
        if (!geom) return fail(G4S_ERR_ALLOC, "geometry buffer callback returned NULL");
        geom = align_ptr(geom);
        float* rec = (float*)(geom + GL.rec);
        uint32_t* tiles_touched = (uint32_t*)(geom + GL.tiles_touched);
        uint32_t* keys_a = (uint32_t*)(geom + GL.keys_a);
        uint32_t* keys_b = (uint32_t*)(geom + GL.keys_b);
        uint32_t* vals_a = (uint32_t*)(geom + GL.vals_a);
        uint32_t* vals_b = (uint32_t*)(geom + GL.vals_b);
        uint32_t* block_sums = (uint32_t*)(geom + GL.block_sums);
        uint32_t* block_offs = (uint32_t*)(geom + GL.block_offs);
        uint32_t* d_total = (uint32_t*)(geom + GL.total);
        if (radii == nullptr) radii = (int*)(geom + GL.internal_radii);  // rasterizer_impl.cu:230-233

        PreprocessArgs pa{};
        pa.P = P; pa.D = D; pa.M = M; pa.W = width; pa.H = height; pa.tiles_x = tiles_x; pa.tiles_y = tiles_y;
        pa.means3D = means3D; pa.scales = scales; pa.rotations = rotations; pa.opacities = opacities; pa.shs = shs;
        pa.transMat_precomp = transMat_precomp; pa.colors_precomp = colors_precomp;
        pa.viewmatrix = viewmatrix; pa.projmatrix = projmatrix; pa.cam_pos = cam_pos;
        pa.scale_modifier = scale_modifier;
        pa.shs_rest = shs_rest;
        pa.sh_vec16 = (shs != nullptr && shs_rest == nullptr && M == 16 && !misaligned(shs, 16));
        pa.no_fastpath = opt(OPT_NO_FASTPATH) != 0;
        pa.rec = rec; pa.clamped = (uint8_t*)(geom + GL.clamped); pa.tiles_touched = tiles_touched; pa.radii = radii;
        pa.tight_rect = (uint2*)(geom + GL.tight_rect);
        // The depth sort ping-pongs between the a and the b arrays and its result is to land in vals_b whatever the
        // number of passes (g4s_rasterizer_layout().depth_sorted): three passes (the regular forward) start in the a
        // arrays, four (presized) in the b arrays; the preprocess writes the unpacked keys into the other key array.
        uint32_t* const k_first = presized ? keys_b : keys_a;
        uint32_t* const k_other = presized ? keys_a : keys_b;
        uint32_t* const v_first = presized ? vals_b : vals_a;
        uint32_t* const v_other = presized ? vals_a : vals_b;
        pa.depth_keys = k_other;
        pa.ref_block_sums = (uint32_t*)(geom + GL.ref_block_sums);
        pa.idx_block_sums = (uint32_t*)(geom + GL.idx_block_sums);
        pa.vis_block_sums = (uint32_t*)(geom + GL.vis_block_sums);
        pa.key_min_blocks = (uint32_t*)(geom + GL.key_min_blocks);
        pa.key_max_blocks = (uint32_t*)(geom + GL.key_max_blocks);
        { ProfScope ps(PF_PREPROCESS_FWD, stream); launch_preprocess_fwd(pa, stream); }
        CHECK_LAUNCH("preprocess_fwd");

        // Everything the host has to know comes out of the preprocess' per-block partial sums: the one host
        // synchronisation of the forward (rasterizer_impl.cu:281-282) sits right behind it, and every later launch
        // is sized for the Gaussians that actually emit instances.
        uint32_t* idx_block_offs = (uint32_t*)(geom + GL.idx_block_offs);
        uint32_t* vis_block_offs = (uint32_t*)(geom + GL.vis_block_offs);
        uint32_t* h_total = nullptr;
        uint32_t* h_total_dev = nullptr;
        hipEvent_t totals_ready = nullptr;
        if (!presized) {
            h_total = pinned_word();
            h_total_dev = h_total ? pinned_word_device(h_total) : nullptr;
            if (!h_total || !h_total_dev) return fail(G4S_ERR_HIP, "hipHostMalloc / hipHostGetDevicePointer failed");
            totals_ready = readback_event();
            if (!totals_ready) return fail(G4S_ERR_HIP, "hipEventCreate failed");
        }
        { ProfScope ps(PF_COUNT_SCAN, stream);
          launch_scan_totals(pa.idx_block_sums, idx_block_offs, pa.ref_block_sums, pa.vis_block_sums, vis_block_offs,
                             d_total, GL.nblocks, ranges, tiles * 2, stream, presized ? (uint32_t)capacity : 0xFFFFFFFFu,
                             h_total_dev, presized ? status_dev : nullptr, pa.key_min_blocks, pa.key_max_blocks); }
        CHECK_LAUNCH("scan totals");
        if (!presized) {
            HIP_TRY(hipEventRecord(totals_ready, stream));  // (the kernel above stored the three counts in host memory)
        }
        // (presized: the kernel wrote status_dev[0..3] = num_rendered, instances binned, emitting Gaussians, overflow flag)

        // Queued BEFORE the host waits for the totals: nothing below needs them on the host -- the gradient slots do
        // not depend on them, and the pack / depth sort / count of the emitting Gaussians read V (d_total[2]) on the
        // device, their launches sized for P.  The GPU therefore has ~0.1 ms of work queued while the host reads the
        // totals back, sizes the binning chunk and issues the rest: the read-back no longer drains the queue.
        const uint32_t* d_V = d_total + 2;
        const uint32_t* d_key_min = d_total + 5;  // smallest depth key of the frame (the totals scan)
        uint32_t* sort_hist = (uint32_t*)(geom + GL.hist);
        uint32_t* sort_bins = (uint32_t*)(geom + GL.bin_total);
        int cur;
        {   // depth order of the emitting Gaussians (stable => ties by ascending index): pack, then sort
            ProfScope ps(PF_DEPTH_SORT, stream);
            launch_slots_and_compact(P, tiles_touched, idx_block_offs, rec, k_other, vis_block_offs, k_first, v_first,
                                     GL.nblocks, stream);
            // three passes over the low 27 bits of (key - smallest key): the whole sort unless the frame's depths span a
            // ratio of 2^16 or more, which the host learns with the totals below.  Presized (no read-back): four passes.
            cur = presized ? radix_sort_u32_pairs(k_first, k_other, v_first, v_other, P, sort_hist, sort_bins, stream, d_V)
                           : radix_sort_depth_low(k_first, k_other, v_first, v_other, P, sort_hist, sort_bins, stream, d_V,
                                                  d_key_min);
        }
        CHECK_LAUNCH("depth sort");

        int R_binned, V_emit, nblocks_v;
        const uint32_t* d_counts = nullptr;   // presized: (V, min(binned, capacity)) on the device
        const uint32_t* d_nbinned = nullptr;
        if (!presized) {
            // the one host wait of the forward (rasterizer_impl.cu:281-282), on the read-back only
            HIP_TRY(hipEventSynchronize(totals_ready));
            // h_total[0]: instances actually binned (3-sigma rect intersected with the alpha-cutoff box),
            // h_total[1]: the reference's count (3-sigma rect only) = the num_rendered this call returns,
            // h_total[2]: Gaussians that emit at least one instance.
            // All buffers are laid out for the reference count, which bounds the binned one.
            if (h_total[1] > 0x7FFFFFFFu) return fail(G4S_ERR_INVALID_ARGUMENT, "num_rendered overflows int");
            R = (int)h_total[1];
            R_binned = (int)h_total[0];
            V_emit = (int)h_total[2];
            nblocks_v = (V_emit + 255) / 256;
            if (V_emit > 0 && ((uint64_t)h_total[4] - h_total[3]) >> DEPTH_SORT_LOW_BITS) {  // a frame that deep: the bits above
                ProfScope ps(PF_DEPTH_SORT, stream);
                cur = radix_sort_depth_top(k_first, k_other, v_first, v_other, P, cur, sort_hist, sort_bins, stream, d_V,
                                           d_key_min);
                CHECK_LAUNCH("depth sort, upper bits");
                // (a fourth pass: the order is in vals_a now -- back to where everything else expects it)
                HIP_TRY(hipMemcpyAsync(vals_b, vals_a, (size_t)V_emit * 4, hipMemcpyDeviceToDevice, stream));
            }
        } else {
            // no read-back: every launch below is sized for the capacity and reads the counts on the device
            R = capacity;  // what the layouts (here and in the backward) are computed from
            R_binned = capacity;
            V_emit = P;
            nblocks_v = GL.nblocks;
            d_counts = d_total + 2;
            d_nbinned = d_total + 3;
        }

        (void)cur;
        const uint32_t* gidx_sorted = vals_b;
        uint32_t* rank_local = keys_a;  // (neither key array is needed after the sort)
        {
            ProfScope ps(PF_COUNT_SCAN, stream);
            launch_count_scan(P, gidx_sorted, tiles_touched, block_sums, block_offs, rank_local, d_total + 8, GL.nblocks,
                              stream, d_V);
        }
        CHECK_LAUNCH("count scan");

        const BinLayout BL = bin_layout((size_t)R);
        char* bin = binning_buffer(binning_ctx, BL.bytes);
        if (!bin) return fail(G4S_ERR_ALLOC, presized ? "binning buffer smaller than g4s_rasterizer_layout(P, capacity).binning_bytes"
                                                      : "binning buffer callback returned NULL");
        bin = align_ptr(bin);
        uint64_t* ent_a = (uint64_t*)(bin + BL.ent_a);
        uint64_t* ent_b = (uint64_t*)(bin + BL.ent_b);
        entries_ptr = ent_a;
        qhit_ptr = (uint8_t*)(bin + BL.qhit);
        if (R_binned > 0) {
            { ProfScope ps(PF_EMIT, stream);  // (also clears the contribution masks qhit[0, R_binned))
              launch_emit(V_emit, (uint32_t)R_binned, tiles_x, gidx_sorted, block_offs, nblocks_v, rank_local,
                          (const uint2*)(geom + GL.tight_rect), ent_a, qhit_ptr, (uint8_t*)(bin + BL.rec_flag), stream,
                          d_counts); }
            CHECK_LAUNCH("emit");
            const int tile_bits = tile_sort_bits(tiles);  // rasterizer_impl.cu:301
            int c2;
            { ProfScope ps(PF_TILE_SORT, stream);
              c2 = radix_sort_u64_keys(ent_a, ent_b, R_binned, ENTRY_TILE_SHIFT, ENTRY_TILE_SHIFT + tile_bits,
                                       (uint32_t*)(bin + BL.hist), (uint32_t*)(bin + BL.bin_total), stream, d_nbinned); }
            CHECK_LAUNCH("tile partition");
            entries_ptr = c2 ? ent_b : ent_a;
            { ProfScope ps(PF_TILE_RANGES, stream); launch_tile_ranges(R_binned, entries_ptr, ranges, stream, d_nbinned); }
            CHECK_LAUNCH("tile ranges");
        }
        rec_ptr = rec;
    } else {
        // keep the callback protocol: zero-sized chunks are still requested
        (void)geometry_buffer(geometry_ctx, 0);
        (void)binning_buffer(binning_ctx, 0);
    }

    uint32_t* tile_order = (uint32_t*)(img + IL.tile_order);
    launch_tile_order(tiles, ranges, tile_order, stream);
    CHECK_LAUNCH("tile order");
    BlendFwdArgs ba{};
    ba.W = width; ba.H = height; ba.tiles_x = tiles_x; ba.tiles_y = tiles_y;
    ba.tile_depth = (uint32_t*)(img + IL.tile_depth);
    ba.ranges = ranges; ba.tile_order = tile_order; ba.entries = entries_ptr; ba.rec = rec_ptr; ba.bg = background;
    ba.final_T = final_T; ba.n_contrib = n_contrib; ba.out_color = out_color; ba.out_others = out_others;
    ba.qhit = qhit_ptr;
    ba.box_only = opt(OPT_BOX_ONLY) != 0;
    { ProfScope ps(PF_BLEND_FWD, stream); launch_blend_fwd(ba, stream); }
    CHECK_LAUNCH("blend_fwd");
    return R;
}

extern "C" int g4s_rasterizer_forward(
    g4s_resize_fn geometry_buffer, void* geometry_ctx, g4s_resize_fn binning_buffer, void* binning_ctx,
    g4s_resize_fn image_buffer, void* image_ctx, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
    float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
    float* out_others, int* radii, int debug, void* stream) {
    return rasterizer_forward_impl(-1, nullptr, geometry_buffer, geometry_ctx, binning_buffer, binning_ctx, image_buffer, image_ctx, P, D,
                                   M, background, width, height, means3D, shs, nullptr, colors_precomp, opacities, scales,
                                   scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                   tan_fovy, prefiltered, out_color, out_others, radii, debug, stream);
}

extern "C" int g4s_rasterizer_forward_split_sh(
    g4s_resize_fn geometry_buffer, void* geometry_ctx, g4s_resize_fn binning_buffer, void* binning_ctx,
    g4s_resize_fn image_buffer, void* image_ctx, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* sh_dc, const float* sh_rest, const float* opacities, const float* scales,
    float scale_modifier, const float* rotations, const float* transMat_precomp, const float* viewmatrix,
    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
    float* out_others, int* radii, int debug, void* stream) {
    t_err[0] = 0;
    if (P > 0 && (!sh_dc || M < 1 || (M > 1 && !sh_rest)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "split SH needs sh_dc [P,1,3] and, for M > 1, sh_rest [P,M-1,3]");
    // M == 1: there is no rest tensor; the packed layout [P,1,3] is the same memory
    return rasterizer_forward_impl(-1, nullptr, geometry_buffer, geometry_ctx, binning_buffer, binning_ctx, image_buffer, image_ctx, P, D,
                                   M, background, width, height, means3D, sh_dc, M > 1 ? sh_rest : nullptr, nullptr, opacities,
                                   scales, scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, cam_pos,
                                   tan_fovx, tan_fovy, prefiltered, out_color, out_others, radii, debug, stream);
}

// The forward without its host synchronisation (include/g4s_rasterizer.h).  sh_rest == NULL: packed SH.
extern "C" int g4s_rasterizer_forward_presized(
    char* geom_buffer, size_t geom_bytes, char* binning_buffer, size_t binning_bytes, char* image_buffer, size_t image_bytes,
    int instance_capacity, uint32_t* status_dev, int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* sh_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
    float* out_color, float* out_others, int* radii, int debug, void* stream) {
    t_err[0] = 0;
    if (instance_capacity < 0 || !status_dev) return fail(G4S_ERR_INVALID_ARGUMENT, "capacity must be >= 0 and status_dev non-NULL");
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(G4S_ERR_INVALID_ARGUMENT, "state buffers must not be NULL");
    FixedChunk g{geom_buffer, geom_bytes}, b{binning_buffer, binning_bytes}, im{image_buffer, image_bytes};
    const int rc = rasterizer_forward_impl(instance_capacity, status_dev, fixed_chunk_cb, &g, fixed_chunk_cb, &b, fixed_chunk_cb,
                                           &im, P, D, M, background, width, height, means3D, shs, (M > 1 ? sh_rest : nullptr),
                                           colors_precomp, opacities, scales, scale_modifier, rotations, transMat_precomp,
                                           viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_others, radii,
                                           debug, stream);
    return rc < 0 ? rc : G4S_OK;  // (the impl returns the capacity as "R"; the real count is status_dev[0])
}

extern "C" size_t g4s_rasterizer_backward_workspace(int P, int R) {
    // gradient records   (their validity bytes live in the forward's binning chunk, the deep-tile list in its image chunk)
    (void)P;
    return align_up((size_t)(R > 0 ? R : 1) * GRAD_STRIDE * 4) + 256;
}

static int rasterizer_backward_impl(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
    const float* shs, const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
    const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
    float* dL_drot, char* workspace, size_t workspace_bytes, int debug, void* stream_, bool accumulate = false,
    void* after_event = nullptr, float* view_stats = nullptr, const g4s_packed_rows* packed = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "bad sizes");
    if (P == 0) return G4S_OK;  // rasterize_points.cu:197: nothing to do, outputs are [0,*]
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer))
        return fail(G4S_ERR_INVALID_ARGUMENT, "state buffers must not be NULL");
    if (!dL_dpix || !dL_depths || !dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D ||
        !dL_dscale || !dL_drot || (M > 0 && !dL_dsh) || (shs_rest && M > 1 && !dL_dsh_rest))
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL gradient pointer");
    if (workspace_bytes < g4s_rasterizer_backward_workspace(P, R) || !workspace)
        return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    if (misaligned(rotations, 16) || misaligned(scales, 8) || misaligned(dL_drot, 16) || misaligned(dL_dscale, 8))
        return fail(G4S_ERR_INVALID_ARGUMENT, "rotations/dL_drot must be 16-byte, scales/dL_dscale 8-byte aligned");

    const int tiles_x = (width + TILE - 1) / TILE, tiles_y = (height + TILE - 1) / TILE;
    const int tiles = tiles_x * tiles_y;
    const GeomLayout GL = geom_layout((size_t)P);
    const BinLayout BL = bin_layout((size_t)R);
    const ImgLayout IL = img_layout((size_t)width * height, (size_t)tiles);
    char* geom = align_ptr(geom_buffer);
    char* img = align_ptr(image_buffer);
    const float* rec = (const float*)(geom + GL.rec);
    if (radii == nullptr) radii = (const int*)(geom + GL.internal_radii);
    float* grad_inst = (float*)align_ptr(workspace);

    // gradient records: only instances that receive a contribution are written by the blend backward; instead of
    // clearing 80 B per instance, one validity byte per instance is cleared and the fold selects on it
    // the deep-tile counter + list sit in the image chunk; the counter is cleared by the tile-order kernel below
    uint32_t* hot_count = (uint32_t*)(img + IL.hot_count);
    uint32_t* hot_list = (uint32_t*)(img + IL.hot_list);
    // One validity byte per record slot says which records the blend backward wrote; the bytes live in the binning chunk
    // and were cleared by the forward (emit).  A second backward over the same forward state finds them set -- to the
    // values it is going to write again (which records are written depends on the forward's state only).
    uint8_t* rec_flag = nullptr;
    bool sh_prezeroed = false;
    if (R > 0) {
        char* bin = align_ptr(binning_buffer);
        rec_flag = (uint8_t*)(bin + BL.rec_flag);
        const int tile_bits = tile_sort_bits(tiles);
        const int passes = (tile_bits + 7) / 8;
        BlendBwdArgs bb{};
        bb.W = width; bb.H = height; bb.tiles_x = tiles_x; bb.tiles_y = tiles_y;
        bb.ranges = (const uint32_t*)(img + IL.ranges);
        // backward order: most blended (entry, quadrant) pairs first -- the forward counted them per tile
        uint32_t* tile_order_bwd = (uint32_t*)(img + IL.tile_order_bwd);
        launch_tile_order(tiles, (const uint32_t*)(img + IL.tile_depth), tile_order_bwd, stream, hot_count);
        bb.tile_order = opt(OPT_BWD_FWD_ORDER) ? (const uint32_t*)(img + IL.tile_order) : tile_order_bwd;
        bb.entries = (const uint64_t*)(bin + ((passes & 1) ? BL.ent_b : BL.ent_a));
        bb.rec = rec; bb.bg = background;
        bb.final_T = (const float*)(img + IL.final_T);
        bb.n_contrib = (const uint32_t*)(img + IL.n_contrib);
        bb.qhit = (const uint8_t*)(bin + BL.qhit);
        bb.dL_dpix = dL_dpix; bb.dL_depths = dL_depths; bb.grad_inst = grad_inst; bb.rec_flag = rec_flag;
        bb.n_slots = (uint32_t)R;
        // One wave per tile is the efficient form when there are enough tiles to fill the GPU (1 024 SIMDs x 3
        // waves); a small frame (<= 768 tiles, e.g. 256 x 256) runs about twice as fast with four waves per tile,
        // and so does any single tile that is much deeper than the rest (measured: tools/deep_tile_bench.py).
        // In a full-size frame only OUTLIERS are handed over -- tiles at least four times deeper than the average
        // list (and deeper than 2 048): when every tile is deep (3 M surfels at 1200x680) one wave each stays the
        // faster form, the four-wave kernel does ~1.9x the work per tile.
        const long long avg_list = (long long)R / (tiles > 0 ? tiles : 1);
        const long long outlier = 4 * avg_list > BWD_HOT_THRESHOLD ? 4 * avg_list : BWD_HOT_THRESHOLD;
        const int hot_override = opt(OPT_BWD_HOT_THRESHOLD);
        bb.hot_threshold = hot_override != G4S_OPTION_UNSET ? hot_override
                           : (tiles <= BWD_FOUR_WAVE_MAX_TILES ? -1 : (int)(outlier < 0x7fffffff ? outlier : 0x7fffffff));
        bb.hot_count = hot_count; bb.hot_list = hot_list;
        // dL_dsh is mostly zero rows (invisible Gaussians).  When the one-wave kernel runs, its workgroups clear the
        // tensor on the side (blend.hip) and K8 writes the visible rows only; otherwise K8 clears the rows it skips.
        // (accumulating: dL_dsh holds the sum over the previous views -- nothing is cleared anywhere)
        if (bb.hot_threshold >= 0 && M > 0 && !opt(OPT_NO_SIDE_ZERO) && !accumulate) {
            float* zb[2] = {dL_dsh, dL_dsh_rest};
            const size_t zn[2] = {(size_t)P * (dL_dsh_rest ? 1 : M) * 3, dL_dsh_rest ? (size_t)P * (M - 1) * 3 : 0};
            bool ok = true;
            for (int z = 0; z < 2; z++) ok = ok && (zn[z] == 0 || (!misaligned(zb[z], 16) && (zn[z] >> 2) < 0x80000000ull));  // (the kernel's u32 loop index must not wrap)
            if (ok) {
                for (int z = 0; z < 2; z++) {
                    if (zn[z] == 0) continue;
                    bb.zero_base[z] = zb[z]; bb.zero_quads[z] = (uint32_t)(zn[z] >> 2); bb.zero_tail[z] = (uint32_t)(zn[z] & 3);
                }
                sh_prezeroed = true;
            }
        }
        { ProfScope ps(PF_BLEND_BWD, stream); launch_blend_bwd(bb, stream); }
        CHECK_LAUNCH("blend_bwd");
    }

    // backward.cu:618-619: W,H re-derived through float truncation (may be W-1 / H-1)
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    PreprocessBwdArgs pb{};
    pb.P = P; pb.D = D; pb.M = M;
    pb.W = (int)(focal_x * tan_fovx * 2);
    pb.H = (int)(focal_y * tan_fovy * 2);
    pb.means3D = means3D; pb.scales = scales; pb.rotations = rotations; pb.shs = shs;
    pb.transMat_precomp = transMat_precomp; pb.colors_precomp = colors_precomp;
    pb.viewmatrix = viewmatrix; pb.projmatrix = projmatrix; pb.campos = campos;
    pb.radii = radii; pb.rec = rec; pb.clamped = (const uint8_t*)(geom + GL.clamped); pb.grad_inst = grad_inst;
    pb.rec_flag = rec_flag; pb.n_slots = (uint32_t)R;
    // the forward's own T (scale_modifier applied, exact W / H) is what the blend kernels' moments refer to
    pb.frame_W = width; pb.frame_H = height; pb.scale_modifier = scale_modifier;
    pb.shs_rest = shs_rest; pb.dL_dsh_rest = dL_dsh_rest; pb.sh_prezeroed = sh_prezeroed;
    pb.accumulate = accumulate;
    pb.view_stats = view_stats;
    if (packed != nullptr && packed->rows != nullptr) {
        if (accumulate) return fail(G4S_ERR_INVALID_ARGUMENT, "packed rows: only the first view of a batch (first_view != 0) can write them");
        if (!packed->block_offs || packed->capacity < 0 || shs == nullptr)
            return fail(G4S_ERR_INVALID_ARGUMENT, "packed rows: block_offs must not be NULL, capacity >= 0, colours from SH");
        pb.packed_rows = packed->rows; pb.packed_block_offs = packed->block_offs;
        pb.packed_capacity = (uint32_t)(packed->capacity < 0xFFFFFFFFll ? packed->capacity : 0xFFFFFFFFll);
    }
    // The blend backward above touches only this call's own state; the per-Gaussian kernel below adds into tensors that
    // the previous view's backward -- on another stream -- may still be adding into: it waits for the caller's event.
    if (after_event) HIP_TRY(hipStreamWaitEvent(stream, (hipEvent_t)after_event, 0));
    pb.sh_vec16 = (shs != nullptr && shs_rest == nullptr && M == 16 && !misaligned(shs, 16) && !misaligned(dL_dsh, 16));
    pb.dL_dmean2D = dL_dmean2D; pb.dL_dnormal = dL_dnormal; pb.dL_dopacity = dL_dopacity; pb.dL_dcolor = dL_dcolor;
    pb.dL_dmean3D = dL_dmean3D; pb.dL_dtransMat = dL_dtransMat; pb.dL_dsh = dL_dsh; pb.dL_dscale = dL_dscale;
    pb.dL_drot = dL_drot;
    { ProfScope ps(PF_PREPROCESS_BWD, stream); launch_preprocess_bwd(pb, stream); }
    CHECK_LAUNCH("preprocess_bwd");
    return G4S_OK;
}

extern "C" int g4s_rasterizer_backward(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
    const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot,
    char* workspace, size_t workspace_bytes, int debug, void* stream) {
    return rasterizer_backward_impl(P, D, M, R, background, width, height, means3D, shs, nullptr, colors_precomp, scales,
                                    scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx,
                                    tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_depths,
                                    dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh, nullptr,
                                    dL_dscale, dL_drot, workspace, workspace_bytes, debug, stream);
}

extern "C" int g4s_rasterizer_backward_split_sh(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
    const float* sh_dc, const float* sh_rest, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale,
    float* dL_drot, char* workspace, size_t workspace_bytes, int debug, void* stream) {
    t_err[0] = 0;
    if (P > 0 && (!sh_dc || M < 1 || (M > 1 && (!sh_rest || !dL_dsh_rest))))
        return fail(G4S_ERR_INVALID_ARGUMENT, "split SH needs sh_dc / dL_dsh_dc and, for M > 1, sh_rest / dL_dsh_rest");
    return rasterizer_backward_impl(P, D, M, R, background, width, height, means3D, sh_dc, M > 1 ? sh_rest : nullptr, nullptr,
                                    scales, scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos,
                                    tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_depths,
                                    dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh_dc,
                                    M > 1 ? dL_dsh_rest : nullptr, dL_dscale, dL_drot, workspace, workspace_bytes, debug,
                                    stream);
}

// Gradient accumulation over views (include/g4s_rasterizer.h).  sh_rest == NULL: sh_dc is the packed [P,M,3] tensor.
extern "C" int g4s_rasterizer_backward_accumulate_packed(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
    const float* sh_dc, const float* sh_rest, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale,
    float* dL_drot, float* view_stats, int first_view, const g4s_packed_rows* packed, char* workspace, size_t workspace_bytes,
    void* after_event, int debug, void* stream) {
    t_err[0] = 0;
    if (P > 0 && (!sh_dc || M < 1 || (sh_rest && M > 1 && !dL_dsh_rest)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "accumulating backward needs SH coefficients (packed, or sh_dc + sh_rest / dL_dsh_rest)");
    return rasterizer_backward_impl(P, D, M, R, background, width, height, means3D, sh_dc, (sh_rest && M > 1) ? sh_rest : nullptr,
                                    nullptr, scales, scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos,
                                    tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_depths,
                                    dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh_dc,
                                    (sh_rest && M > 1) ? dL_dsh_rest : nullptr, dL_dscale, dL_drot, workspace, workspace_bytes,
                                    debug, stream, first_view == 0, after_event, view_stats, packed);
}

extern "C" int g4s_rasterizer_backward_accumulate(
    int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
    const float* sh_dc, const float* sh_rest, const float* scales, float scale_modifier, const float* rotations,
    const float* transMat_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity,
    float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat, float* dL_dsh_dc, float* dL_dsh_rest, float* dL_dscale,
    float* dL_drot, float* view_stats, int first_view, char* workspace, size_t workspace_bytes, void* after_event, int debug,
    void* stream) {
    t_err[0] = 0;
    if (P > 0 && (!sh_dc || M < 1 || (sh_rest && M > 1 && !dL_dsh_rest)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "accumulating backward needs SH coefficients (packed, or sh_dc + sh_rest / dL_dsh_rest)");
    return rasterizer_backward_impl(P, D, M, R, background, width, height, means3D, sh_dc, (sh_rest && M > 1) ? sh_rest : nullptr,
                                    nullptr, scales, scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos,
                                    tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_depths,
                                    dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dtransMat, dL_dsh_dc,
                                    (sh_rest && M > 1) ? dL_dsh_rest : nullptr, dL_dscale, dL_drot, workspace, workspace_bytes,
                                    debug, stream, first_view == 0, after_event, view_stats);
}

extern "C" int g4s_rasterizer_mark_visible(int P, const float* means3D, const float* viewmatrix,
                                           const float* projmatrix, uint8_t* present, void* stream_) {
    (void)projmatrix;
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return G4S_OK;
    if (!means3D || !viewmatrix || !present) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL pointer");
    launch_mark_visible(P, means3D, viewmatrix, present, stream);
    CHECK_LAUNCH("mark_visible");
    return G4S_OK;
}

extern "C" int g4s_knn_mean_dist(int P, const float* points, float* meanDists, char* workspace,
                                 size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return G4S_OK;
    if (!points || !meanDists || !workspace) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL pointer");
    if (workspace_bytes < g4s_knn_workspace(P)) return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    g4s_knn_launch_internal(P, points, meanDists, workspace, stream);
    CHECK_LAUNCH("knn");
    return G4S_OK;
}

// ---- fused photometric loss (include/g4s_losses.h) ------------------------------------------------
#include "../../include/g4s_losses.h"
extern "C" void g4s_photometric_launch_internal(int W, int H, const float* image, const float* gt, float lambda, float* out3,
                                                float* dL_dimage, char* workspace, hipStream_t s);

extern "C" int g4s_photometric_loss(int width, int height, const float* image, const float* gt, float lambda_dssim,
                                    float* out3, float* dL_dimage, char* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "width, height must be positive");
    if (!image || !gt || !out3) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (!workspace || workspace_bytes < g4s_photometric_workspace(width, height))
        return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    { ProfScope ps(PF_PHOTO_LOSS, stream);
      g4s_photometric_launch_internal(width, height, image, gt, lambda_dssim, out3, dL_dimage, workspace, stream); }
    CHECK_LAUNCH("photometric_loss");
    return G4S_OK;
}

extern "C" void g4s_georeg_launch_internal(int fwd, int W, int H, const float* rn, const float* sn, const float* dist, float* out2,
                                           const float* g2, float* d_rn, float* d_sn, float* d_dist, char* workspace,
                                           hipStream_t s);

extern "C" int g4s_geometry_regularizers_forward(int width, int height, const float* rend_normal, const float* surf_normal,
                                                 const float* rend_dist, float* out2, char* workspace, size_t workspace_bytes,
                                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "width, height must be positive");
    if (!rend_normal || !surf_normal || !rend_dist || !out2) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (!workspace || workspace_bytes < g4s_geometry_regularizers_workspace(width, height))
        return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    { ProfScope ps(PF_GEO_REG, stream);
      g4s_georeg_launch_internal(1, width, height, rend_normal, surf_normal, rend_dist, out2, nullptr, nullptr, nullptr, nullptr,
                                 workspace, stream); }
    CHECK_LAUNCH("geometry_regularizers_forward");
    return G4S_OK;
}

extern "C" int g4s_geometry_regularizers_backward(int width, int height, const float* rend_normal, const float* surf_normal,
                                                  const float* grad_out2, float* dL_drend_normal, float* dL_dsurf_normal,
                                                  float* dL_drend_dist, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "width, height must be positive");
    if (!rend_normal || !surf_normal || !grad_out2 || !dL_drend_normal || !dL_dsurf_normal || !dL_drend_dist)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    { ProfScope ps(PF_GEO_REG, stream);
      g4s_georeg_launch_internal(0, width, height, rend_normal, surf_normal, nullptr, nullptr, grad_out2, dL_drend_normal,
                                 dL_dsurf_normal, dL_drend_dist, nullptr, stream); }
    CHECK_LAUNCH("geometry_regularizers_backward");
    return G4S_OK;
}

// ---- fused Adam (include/g4s_optim.h) ---------------------------------------------------------------
#include "../../include/g4s_optim.h"
extern "C" void g4s_adam_launch_internal(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, const long long* numel, const double* lr, const int* step,
                                         double beta1, double beta2, double eps, hipStream_t s);

extern "C" int g4s_adam_step(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const long long* numel, const double* lr, const int* step, double beta1,
                             double beta2, double eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (nseg < 1 || nseg > 8) return fail(G4S_ERR_INVALID_ARGUMENT, "1..8 segments");
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL array");
    for (int i = 0; i < nseg; i++) {
        if (numel[i] < 0 || step[i] < 1) return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: numel < 0 or step < 1", i);
        if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))
            return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: NULL pointer", i);
    }
    { ProfScope ps(PF_ADAM, stream);
      g4s_adam_launch_internal(nseg, params, grads, exp_avg, exp_avg_sq, numel, lr, step, beta1, beta2, eps, stream); }
    CHECK_LAUNCH("adam_step");
    return G4S_OK;
}

extern "C" void g4s_adam_device_launch_internal(int nseg, float* const* params, const float* const* grads,
                                                float* const* exp_avg, float* const* exp_avg_sq, const long long* numel,
                                                const double* lr_dev, float* const* step_dev, float* coef_dev, double beta1,
                                                double beta2, double eps, hipStream_t s);

extern "C" int g4s_adam_step_device(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                                    float* const* exp_avg_sq, const long long* numel, const double* lr_dev,
                                    float* const* step_dev, float* coef_dev, double beta1, double beta2, double eps,
                                    void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (nseg < 1 || nseg > 8) return fail(G4S_ERR_INVALID_ARGUMENT, "1..8 segments");
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr_dev || !step_dev || !coef_dev)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL array");
    for (int i = 0; i < nseg; i++) {
        if (numel[i] < 0 || !step_dev[i]) return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: numel < 0 or NULL step", i);
        if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))
            return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: NULL pointer", i);
    }
    { ProfScope ps(PF_ADAM, stream);
      g4s_adam_device_launch_internal(nseg, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, step_dev, coef_dev, beta1, beta2,
                                      eps, stream); }
    CHECK_LAUNCH("adam_step_device");
    return G4S_OK;
}

// ---- stream compaction of Gaussian rows (include/g4s_optim.h) ------------------------------------
extern "C" int g4s_compact_scan_launch_internal(int P, const uint8_t* keep, int* out_count, char* workspace, hipStream_t s);
extern "C" int g4s_compact_gather_launch_internal(int P, const uint8_t* keep, const char* workspace, int nseg,
                                                  const float* const* src, float* const* dst, const int* widths,
                                                  long long dst_row0, hipStream_t s);

extern "C" int g4s_compact_scan(int P, const unsigned char* keep, int* out_count, char* workspace, size_t workspace_bytes,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P must not be negative");
    if (!out_count || !workspace || (P > 0 && !keep)) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (workspace_bytes < g4s_compact_workspace(P)) return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    if (P == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int), stream) != hipSuccess) return fail(G4S_ERR_HIP, "memset failed");
        return G4S_OK;
    }
    g4s_compact_scan_launch_internal(P, keep, out_count, workspace, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G4S_ERR_HIP, "compact_scan launch: %s", hipGetErrorString(e));
    return G4S_OK;
}

extern "C" int g4s_compact_gather(int P, const unsigned char* keep, const char* workspace, int nseg, const float* const* src,
                                  float* const* dst, const int* widths, long long dst_row0, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0 || nseg < 0 || dst_row0 < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P, nseg, dst_row0 must not be negative");
    if (P == 0 || nseg == 0) return G4S_OK;
    if (!keep || !workspace || !src || !dst || !widths) return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    for (int i = 0; i < nseg; i++)
        if (!src[i] || !dst[i] || widths[i] <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "tensor %d: NULL pointer or width <= 0", i);
    g4s_compact_gather_launch_internal(P, keep, workspace, nseg, src, dst, widths, dst_row0, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G4S_ERR_HIP, "compact_gather launch: %s", hipGetErrorString(e));
    return G4S_OK;
}

// ---- packed rows for the visible-rows gradient exchange ------------------------------------------
extern "C" void g4s_densify_stats_launch_internal(int P, const float* grad, const unsigned char* filter, const int* radii,
                                                  float* accum, float* denom, float* max_radii, hipStream_t s);

extern "C" int g4s_densify_stats(int P, const float* grad_mean2D, const unsigned char* update_filter, const int* radii,
                                 float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P must not be negative");
    if (P == 0) return G4S_OK;
    if (!grad_mean2D || !update_filter || !xyz_gradient_accum || !denom || (max_radii2D && !radii))
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    g4s_densify_stats_launch_internal(P, grad_mean2D, update_filter, radii, xyz_gradient_accum, denom, max_radii2D, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G4S_ERR_HIP, "densify_stats launch: %s", hipGetErrorString(e));
    return G4S_OK;
}

extern "C" void g4s_activations_launch_internal(int fwd, int P, const float* scaling_or_scales, const float* rotation,
                                                const float* opacity_or_opac, const float* g_scales, const float* g_rots,
                                                const float* g_opac, float* out_s, float* out_r, float* out_o, hipStream_t s);

extern "C" int g4s_activations_forward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                       float* scales, float* rotations, float* opacities, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P must not be negative");
    if (P == 0) return G4S_OK;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !scales || !rotations || !opacities)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (misaligned(scaling_raw, 8) || misaligned(scales, 8) || misaligned(rotation_raw, 16) || misaligned(rotations, 16))
        return fail(G4S_ERR_INVALID_ARGUMENT, "scaling / scales must be 8-byte, rotations 16-byte aligned");
    g4s_activations_launch_internal(1, P, scaling_raw, rotation_raw, opacity_raw, nullptr, nullptr, nullptr, scales, rotations,
                                    opacities, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G4S_ERR_HIP, "activations launch: %s", hipGetErrorString(e));
    return G4S_OK;
}

extern "C" int g4s_activations_backward(int P, const float* scales, const float* rotation_raw, const float* opacities,
                                        const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                                        float* dL_dscaling_raw, float* dL_drotation_raw, float* dL_dopacity_raw, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    t_err[0] = 0;
    if (P < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "P must not be negative");
    if (P == 0) return G4S_OK;
    if (!scales || !rotation_raw || !opacities || !dL_dscales || !dL_drotations || !dL_dopacities || !dL_dscaling_raw ||
        !dL_drotation_raw || !dL_dopacity_raw)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (misaligned(scales, 8) || misaligned(dL_dscales, 8) || misaligned(dL_dscaling_raw, 8) || misaligned(rotation_raw, 16) ||
        misaligned(dL_drotations, 16) || misaligned(dL_drotation_raw, 16))
        return fail(G4S_ERR_INVALID_ARGUMENT, "scale tensors must be 8-byte, rotation tensors 16-byte aligned");
    g4s_activations_launch_internal(0, P, scales, rotation_raw, opacities, dL_dscales, dL_drotations, dL_dopacities,
                                    dL_dscaling_raw, dL_drotation_raw, dL_dopacity_raw, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G4S_ERR_HIP, "activations backward launch: %s", hipGetErrorString(e));
    return G4S_OK;
}

extern "C" void g4s_pack_rows_launch_internal(int nseg, float* const* ptrs, const int* widths, const long long* idx, int n,
                                              float* packed, int unpack, hipStream_t s);

extern "C" int g4s_pack_rows(int nseg, float* const* segments, const int* widths, const long long* row_index, int n,
                             float* packed, int unpack, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (nseg < 1 || nseg > 8 || n < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "1..8 segments, n >= 0");
    if (unpack < 0 || unpack > 15 || ((unpack & 4) && !(unpack & 1)) || ((unpack & 8) && !(unpack & 2)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "mode: bit 0 unpack, bit 1 row-major buffer, bit 2 add (unpack only), "
                                              "bit 3 index column (row-major only)");
    const bool idx_from_buffer = (unpack & 9) == 9;
    if (!segments || !widths || (n > 0 && ((!row_index && !idx_from_buffer) || !packed)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL pointer");
    for (int i = 0; i < nseg; i++)
        if (!segments[i] || widths[i] <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: NULL pointer or width <= 0", i);
    g4s_pack_rows_launch_internal(nseg, segments, widths, row_index, n, packed, unpack, stream);
    CHECK_LAUNCH("pack_rows");
    return G4S_OK;
}

extern "C" int g4s_accumulate_rows_launch_internal(int nseg, float* const* ptrs, const int* widths, int nsrc, const int* src_off,
                                                   const int* src_cnt, const float* packed, int row_lo, int row_hi,
                                                   hipStream_t s, int own_pos);

static int accumulate_rows_impl(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                                const int* src_counts, const float* packed, int row_lo, int row_hi, int own_position, void* stream_);

extern "C" int g4s_accumulate_rows(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                                   const int* src_counts, const float* packed, int row_lo, int row_hi, void* stream_) {
    return accumulate_rows_impl(nseg, segments, widths, nsrc, src_offsets, src_counts, packed, row_lo, row_hi, 0, stream_);
}

extern "C" int g4s_accumulate_rows_ordered(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                                           const int* src_counts, const float* packed, int row_lo, int row_hi, int own_position,
                                           void* stream_) {
    t_err[0] = 0;
    if (own_position < 0 || own_position > nsrc) return fail(G4S_ERR_INVALID_ARGUMENT, "0 <= own_position <= nsrc");
    if (own_position != 0 && nsrc > 8) return fail(G4S_ERR_UNSUPPORTED, "the ordered accumulation takes at most 8 sources");
    return accumulate_rows_impl(nseg, segments, widths, nsrc, src_offsets, src_counts, packed, row_lo, row_hi, own_position, stream_);
}

static int accumulate_rows_impl(int nseg, float* const* segments, const int* widths, int nsrc, const int* src_offsets,
                                const int* src_counts, const float* packed, int row_lo, int row_hi, int own_position, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (nseg < 1 || nseg > 8 || nsrc < 0 || row_lo < 0 || row_hi < row_lo)
        return fail(G4S_ERR_INVALID_ARGUMENT, "1..8 segments, nsrc >= 0, 0 <= row_lo <= row_hi");
    if (!segments || !widths || (nsrc > 0 && (!src_offsets || !src_counts || !packed)))
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL pointer");
    for (int i = 0; i < nseg; i++)
        if (!segments[i] || widths[i] <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "segment %d: NULL pointer or width <= 0", i);
    for (int i = 0; i < nsrc; i++)
        if (src_offsets[i] < 0 || src_counts[i] < 0) return fail(G4S_ERR_INVALID_ARGUMENT, "source %d: negative offset / count", i);
    if (nsrc == 0 || row_hi == row_lo) return G4S_OK;
    if (g4s_accumulate_rows_launch_internal(nseg, segments, widths, nsrc, src_offsets, src_counts, packed, row_lo, row_hi, stream,
                                            own_position) != 0)
        return fail(G4S_ERR_UNSUPPORTED, "rows wider than 240 floats");
    CHECK_LAUNCH("accumulate_rows");
    return G4S_OK;
}

// ---- fused render() map post-processing (include/g4s_render_maps.h) -------------------------------
#include "../../include/g4s_render_maps.h"

extern "C" size_t g4s_render_maps_workspace(void) { return 512; }

extern "C" int g4s_render_maps_forward(int width, int height, const float* allmap, const float* world_view_transform,
                                       const float* full_proj_transform, float depth_ratio, float* rend_alpha,
                                       float* rend_normal, float* rend_normal_cam, float* rend_depth, float* rend_dist,
                                       float* surf_depth, float* surf_normal, float* surf_normal_cam, char* workspace,
                                       size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "width, height must be positive");
    if (!allmap || !world_view_transform || !full_proj_transform || !rend_alpha || !rend_normal || !rend_normal_cam ||
        !rend_depth || !rend_dist || !surf_depth || !surf_normal || !surf_normal_cam)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (!workspace || workspace_bytes < g4s_render_maps_workspace()) return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    float* outs[8] = {rend_alpha, rend_normal, rend_normal_cam, rend_depth, rend_dist, surf_depth, surf_normal, surf_normal_cam};
    { ProfScope ps(PF_MAPS_FWD, stream);
      g4s_maps_launch_internal(1, width, height, depth_ratio, allmap, world_view_transform, full_proj_transform,
                               (float*)align_ptr(workspace), outs, nullptr, nullptr, nullptr, stream); }
    CHECK_LAUNCH("render_maps_forward");
    return G4S_OK;
}

extern "C" int g4s_render_maps_backward(int width, int height, const float* allmap, const float* surf_depth,
                                        const float* world_view_transform, const float* full_proj_transform,
                                        float depth_ratio, const float* dL_rend_alpha, const float* dL_rend_normal,
                                        const float* dL_rend_normal_cam, const float* dL_rend_depth,
                                        const float* dL_rend_dist, const float* dL_surf_depth, const float* dL_surf_normal,
                                        const float* dL_surf_normal_cam, float* dL_dallmap, char* workspace,
                                        size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    t_err[0] = 0;
    if (width <= 0 || height <= 0) return fail(G4S_ERR_INVALID_ARGUMENT, "width, height must be positive");
    if (!allmap || !surf_depth || !world_view_transform || !full_proj_transform || !dL_dallmap)
        return fail(G4S_ERR_INVALID_ARGUMENT, "NULL required pointer");
    if (!workspace || workspace_bytes < g4s_render_maps_workspace()) return fail(G4S_ERR_INVALID_ARGUMENT, "workspace too small");
    const float* grads[8] = {dL_rend_alpha, dL_rend_normal, dL_rend_normal_cam, dL_rend_depth, dL_rend_dist,
                             dL_surf_depth, dL_surf_normal, dL_surf_normal_cam};
    { ProfScope ps(PF_MAPS_BWD, stream);
      g4s_maps_launch_internal(0, width, height, depth_ratio, allmap, world_view_transform, full_proj_transform,
                               (float*)align_ptr(workspace), nullptr, surf_depth, grads, dL_dallmap, stream); }
    CHECK_LAUNCH("render_maps_backward");
    return G4S_OK;
}
