// C ABI of libvo_mi355x.so (include/vo_mi355x.h): contexts, device buffers, host<->device
// staging and the entry points that mirror the reference's three front-end functions
// (camera_utility.hpp:742 opt_DetectFeatures, :633 optRelativePose, :288 triangPoints).
#include <algorithm>
#include <cmath>

#include "vo_common.hpp"

// ---------------------------------------------------------------- errors
static thread_local std::string g_hip_err;

void vo_set_hip_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_hip_err = buf;
}

void vo_set_error_text(const char* text) { g_hip_err = text; }

extern "C" int vo_abi_version(void) { return VO_ABI_VERSION; }

extern "C" const char* vo_last_hip_error(void) { return g_hip_err.c_str(); }

extern "C" const char* vo_strerror(int s) {
    switch (s) {
        case VO_OK: return "ok";
        case VO_ERR_INVALID_ARG: return "invalid argument";
        case VO_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
        case VO_ERR_HIP: return "HIP runtime error";
        case VO_ERR_CAPACITY: return "output capacity exceeded";
        case VO_ERR_UNSUPPORTED: return "unsupported parameter combination";
        case VO_ERR_TOO_FEW_POINTS: return "fewer than 5 correspondences";
        case VO_ERR_NO_MODEL: return "RANSAC found no model";
        default: return "unknown status";
    }
}

extern "C" void vo_params_default(vo_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->hessian_threshold = 50.f;  // camera_utility.hpp:33
    p->n_octaves = 4;             // :35
    p->n_octave_layers = 3;       // :36
    p->extended = 0;              // :37
    p->upright = 0;               // :38
    p->ratio_thresh = 0.7f;       // :41
    p->roi_x_lo = 28;             // :44
    p->roi_y_lo = 20;             // :45
    p->roi_x_hi = 614;            // :46
    p->roi_y_hi = 482;            // :47
    p->ransac_prob = 0.99;        // :53
    p->ransac_threshold = 2.0;    // :54
    p->ransac_max_iters = 1000;
    p->ransac_seed = (uint64_t)-1;
    p->inlier_threshold = 0.3f;   // :56
    p->vpf_threshold = 0.85f;     // :58
    p->distance_threshold = 50.0; // :60
    p->reprojection_tolerance = 0.5f;  // :63
    p->kp_capacity = 16384;
    p->match_capacity = 16384;
}

// ---------------------------------------------------------------- profiling
StageScope::StageScope(vo_ctx* c, int s, hipStream_t on) : ctx(c), stage(s), stream(on ? on : c->stream) {
    if (!ctx->profiling) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        a = b = nullptr;
        return;
    }
    hipEventRecord(a, stream);
}
StageScope::~StageScope() {
    if (!a) return;
    hipEventRecord(b, stream);
    ctx->timers[stage].pending.emplace_back(a, b);
}

static void resolve_timers(vo_ctx* ctx) {
    for (int s = 0; s < ST_COUNT; s++) {
        StageTimer& t = ctx->timers[s];
        for (auto& pr : t.pending) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                t.ms += ms;
                t.launches++;
            }
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        t.pending.clear();
    }
}

static const char* kStageNames[ST_COUNT] = {"integral",      "hessian_oct1",  "hessian_oct0_23",    "sort",          "orientation",
                                            "describe_prep", "describe_w768", "describe_w320",      "describe_w128", "describe_w64",
                                            "describe_seq",  "match_prep",    "match_mfma",         "match_select",  "ransac_sample",
                                            "ransac_solve_score", "ransac_select", "pose",          "triangulate",
                                            "prep_resize",   "prep_remap_gray", "prep_clahe"};

extern "C" int vo_stage_count(void) { return ST_COUNT; }
extern "C" const char* vo_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }
extern "C" int vo_profile_enable(vo_ctx* ctx, int on) {
    if (!ctx) return VO_ERR_INVALID_ARG;
    if (ctx->profiling != (on != 0)) {
        // The two modes order their launches differently (with the timers on everything is serialised on the main stream;
        // otherwise the auxiliary stream's order carries dependencies between consecutive runs): drain before switching
        VO_HIP(hipSetDevice(ctx->device));
        VO_HIP(hipStreamSynchronize(ctx->stream));
        VO_HIP(hipStreamSynchronize(ctx->stream_aux));
        VO_HIP(hipStreamSynchronize(ctx->stream_geo));
    }
    ctx->profiling = on != 0;
    return VO_OK;
}
extern "C" int vo_profile_get(vo_ctx* ctx, int stage, double* ms, int64_t* launches) {
    if (!ctx || stage < 0 || stage >= ST_COUNT) return VO_ERR_INVALID_ARG;
    resolve_timers(ctx);
    if (ms) *ms = ctx->timers[stage].ms;
    if (launches) *launches = ctx->timers[stage].launches;
    return VO_OK;
}
extern "C" int vo_profile_reset(vo_ctx* ctx) {
    if (!ctx) return VO_ERR_INVALID_ARG;
    resolve_timers(ctx);
    for (int s = 0; s < ST_COUNT; s++) {
        ctx->timers[s].ms = 0;
        ctx->timers[s].launches = 0;
    }
    return VO_OK;
}

// ---------------------------------------------------------------- buffers
template <class T>
static int dalloc(T*& p, size_t n) {
    VO_HIP(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)));
    return VO_OK;
}
template <class T>
static void dfree(T*& p) {
    if (p) hipFree(p);
    p = nullptr;
}

int vo_frameset_alloc(FrameSet& fs, int F, int H, int W, int cap, int dlen) {
    if (fs.F == F && fs.H == H && fs.W == W && fs.cap == cap && fs.dlen == dlen && fs.img) return VO_OK;
    vo_frameset_free(fs);
    fs.dlen = dlen;
    fs.F = F;
    fs.H = H;
    fs.W = W;
    fs.cap = cap;
    const int nbands = (H + 15) / 16;
    VO_TRY(dalloc(fs.img, (size_t)F * H * W + 256));  // tile staging may read up to 111 bytes past a row end
    VO_TRY(dalloc(fs.sum, (size_t)F * (H + 1) * (W + 1) + 64));
    VO_TRY(dalloc(fs.sum4, (size_t)F * fs.sum4_stride() + 64));
    VO_TRY(dalloc(fs.colsum, (size_t)F * nbands * W));
    VO_TRY(dalloc(fs.quad, (size_t)F * ((W + 7) / 8) * 8 * H + 64));  // 8-pixel column strips (vo_detect.hip, quad_index)
    VO_TRY(dalloc(fs.kp_raw, (size_t)F * cap));
    VO_TRY(dalloc(fs.kp, (size_t)F * cap));
    VO_TRY(dalloc(fs.rank, (size_t)F * cap));
    VO_TRY(dalloc(fs.desc, (size_t)F * cap * dlen));
    VO_TRY(dalloc(fs.descT, (size_t)F * cap * 72));  // 9 slabs of 32 B per row
    VO_TRY(dalloc(fs.nrm, (size_t)F * cap));
    VO_TRY(dalloc(fs.nmax, (size_t)F));
    VO_TRY(dalloc(fs.dir_sc, (size_t)F * cap));
    VO_TRY(dalloc(fs.cnt, (size_t)F));
    VO_TRY(dalloc(fs.cls_list, (size_t)5 * F * cap));
    VO_TRY(dalloc(fs.cls_cnt, (size_t)kClsCntInts));
    VO_TRY(dalloc(fs.cls_hist, (size_t)4 * F * (((W + 255) / 256) * ((H + 255) / 256)) + 4));
    VO_HIP(hipMemset(fs.rank, 0, (size_t)F * cap * sizeof(int)));
    VO_HIP(hipMemset(fs.cnt, 0, (size_t)F * sizeof(FrameCounters)));
    return VO_OK;
}

void vo_frameset_free(FrameSet& fs) {
    dfree(fs.img);
    dfree(fs.sum);
    dfree(fs.sum4);
    dfree(fs.colsum);
    dfree(fs.quad);
    dfree(fs.kp_raw);
    dfree(fs.kp);
    dfree(fs.rank);
    dfree(fs.desc);
    dfree(fs.descT);
    dfree(fs.nrm);
    dfree(fs.nmax);
    dfree(fs.dir_sc);
    dfree(fs.cnt);
    dfree(fs.cls_list);
    dfree(fs.cls_cnt);
    dfree(fs.cls_hist);
    fs.F = fs.H = fs.W = fs.cap = 0;
}

int vo_ctx_tmp(vo_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->d_tmp_bytes) {
        if (ctx->d_tmp) hipFree(ctx->d_tmp);
        ctx->d_tmp = nullptr;
        ctx->d_tmp_bytes = 0;
        VO_HIP(hipMalloc(&ctx->d_tmp, bytes));
        ctx->d_tmp_bytes = bytes;
    }
    *out = ctx->d_tmp;
    return VO_OK;
}

static int ctx_bounce(vo_ctx* ctx, size_t bytes, bool down, uint8_t** out);

// ---------------------------------------------------------------- context
extern "C" int vo_ctx_create(const vo_params* params, int hip_device, vo_ctx** out) {
    if (!out) return VO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hip_device < 0 || hip_device >= ndev) return VO_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) return VO_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VO_ERR_NO_DEVICE;  // code objects are gfx950 only
    vo_params p;
    if (params)
        p = *params;
    else
        vo_params_default(&p);
    // the reference's values (4, 3, false, false) take the tuned kernels; 1..5 octaves, 1..6 layers per octave, upright and
    // extended (128-float descriptors) take plain forms of the same arithmetic
    // (a fifth octave doubles the largest window once more: accepted with at most the reference's three layers)
    if (p.n_octaves < 1 || p.n_octaves > kMaxOctaves || p.n_octave_layers < 1 || p.n_octave_layers > kMaxOctaveLayers ||
        (p.n_octaves > kNumOctaves && p.n_octave_layers > kLayersPerOctave - 2))
        return VO_ERR_UNSUPPORTED;
    p.upright = p.upright ? 1 : 0;
    p.extended = p.extended ? 1 : 0;
    if (p.kp_capacity < 64 || p.match_capacity < 8 || p.ransac_max_iters < 1 || p.ransac_max_iters > kMaxHyp)
        return VO_ERR_INVALID_ARG;
    p.kp_capacity = (p.kp_capacity + 63) & ~63;
    VO_HIP(hipSetDevice(hip_device));
    vo_ctx* ctx = new vo_ctx();
    ctx->prm = p;
    ctx->device = hip_device;
    // (stream priorities were measured: a low-priority geometry stream changes nothing, a low-priority auxiliary stream costs
    // 1.4 % - the two descriptor launches that run side by side do best when the hardware interleaves them freely)
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_geo, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_quads, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_integral, hipEventDisableTiming) != hipSuccess) {
        if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
        if (ctx->ev_quads) hipEventDestroy(ctx->ev_quads);
        if (ctx->ev_integral) hipEventDestroy(ctx->ev_integral);
        if (ctx->stream_aux) hipStreamDestroy(ctx->stream_aux);
        if (ctx->stream_geo) hipStreamDestroy(ctx->stream_geo);
        if (ctx->stream) hipStreamDestroy(ctx->stream);
        delete ctx;
        return VO_ERR_HIP;
    }
    DetectTables t;
    vo_build_detect_tables(t, 0);
    int st = vo_upload_detect_tables(t);
    if (st == VO_OK) st = vo_build_window_tables(ctx);
    if (st == VO_OK) st = vo_geometry_init(ctx);
    if (st == VO_OK) st = vo_detect_init(ctx);
    // the pinned bounce buffers exist from the start, large enough for vo_triang_points_noalloc's maximum (it may not grow them)
    uint8_t* unused = nullptr;
    if (st == VO_OK) st = ctx_bounce(ctx, (size_t)1 << 20, false, &unused);
    if (st == VO_OK) st = ctx_bounce(ctx, (size_t)1 << 20, true, &unused);
    if (st != VO_OK) {
        vo_geometry_release(ctx);
        if (ctx->h_up) hipHostFree(ctx->h_up);
        if (ctx->h_down) hipHostFree(ctx->h_down);
            if (ctx->d_win_cell) hipFree(ctx->d_win_cell);
        if (ctx->d_win_pix) hipFree(ctx->d_win_pix);
        hipEventDestroy(ctx->ev_fork);
        hipEventDestroy(ctx->ev_join);
        hipEventDestroy(ctx->ev_quads);
        hipEventDestroy(ctx->ev_integral);
        hipStreamDestroy(ctx->stream_aux);
        hipStreamDestroy(ctx->stream_geo);
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return st;
    }
    *out = ctx;
    return VO_OK;
}

extern "C" void vo_ctx_destroy(vo_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->pb) vo_batch_destroy(ctx->pb);
    ctx->pb = nullptr;
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->stream_geo);
    hipStreamSynchronize(ctx->stream_aux);
    resolve_timers(ctx);
    vo_frameset_free(ctx->fs);
    vo_pairset_free(ctx->ps);
    vo_geometry_release(ctx);
    if (ctx->d_tmp) hipFree(ctx->d_tmp);
    if (ctx->d_gen) hipFree(ctx->d_gen);
    if (ctx->h_up) hipHostFree(ctx->h_up);
    if (ctx->h_down) hipHostFree(ctx->h_down);
    if (ctx->d_win_cell) hipFree(ctx->d_win_cell);
    if (ctx->d_win_pix) hipFree(ctx->d_win_pix);
    hipEventDestroy(ctx->ev_fork);
    hipEventDestroy(ctx->ev_join);
    hipEventDestroy(ctx->ev_quads);
    hipEventDestroy(ctx->ev_integral);
    hipStreamDestroy(ctx->stream_aux);
    hipStreamDestroy(ctx->stream_geo);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int vo_ctx_get_params(const vo_ctx* ctx, vo_params* out) {
    if (!ctx || !out) return VO_ERR_INVALID_ARG;
    *out = ctx->prm;
    return VO_OK;
}

// ---------------------------------------------------------------- staging helpers
static int check_image(const uint8_t* img, int rows, int cols, size_t step) {
    if (!img || rows <= 0 || cols <= 0 || step < (size_t)cols) return VO_ERR_INVALID_ARG;
    if (rows < 16 || cols < 16 || cols > 8192 || rows > 8192) return VO_ERR_UNSUPPORTED;
    return VO_OK;
}

// pinned bounce buffer of the context (up = host -> device, down = device -> host), grown on demand.  The synchronous
// entry points that use it have nothing in flight when they are entered, so growing it is safe.
static int ctx_bounce(vo_ctx* ctx, size_t bytes, bool down, uint8_t** out) {
    uint8_t*& p = down ? ctx->h_down : ctx->h_up;
    size_t& cap = down ? ctx->h_down_bytes : ctx->h_up_bytes;
    if (bytes > cap) {
        VO_HIP(hipStreamSynchronize(ctx->stream));
        if (p) hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t nb = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        VO_HIP(hipHostMalloc((void**)&p, nb, hipHostMallocDefault));
        cap = nb;
    }
    *out = p;
    return VO_OK;
}

static int upload_frame(vo_ctx* ctx, FrameSet& fs, int frame, const uint8_t* img, size_t step) {
    uint8_t* up = nullptr;
    VO_TRY(ctx_bounce(ctx, (size_t)fs.F * fs.img_stride(), false, &up));
    uint8_t* slot = up + (size_t)frame * fs.img_stride();
    if (step == (size_t)fs.W)
        std::memcpy(slot, img, fs.img_stride());
    else
        for (int y = 0; y < fs.H; y++) std::memcpy(slot + (size_t)y * fs.W, img + (size_t)y * step, (size_t)fs.W);
    VO_HIP(hipMemcpyAsync(fs.img + (size_t)frame * fs.img_stride(), slot, fs.img_stride(), hipMemcpyHostToDevice, ctx->stream));
    return VO_OK;
}

// Several device -> host results of a synchronous call through the pinned bounce buffer: add() enqueues the copies,
// finish() waits once and hands the bytes to the caller's (pageable) buffers.
struct BounceDown {
    struct Item {
        void* dst;
        size_t off, bytes;
    };
    vo_ctx* ctx;
    uint8_t* base = nullptr;
    size_t off = 0;
    Item items[8];
    int n = 0;
    int begin(vo_ctx* c, size_t total) {
        ctx = c;
        return ctx_bounce(c, total + 8 * 64, true, &base);
    }
    int add(void* dst, const void* dev, size_t bytes) {
        if (!bytes) return VO_OK;
        VO_HIP(hipMemcpyAsync(base + off, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        items[n++] = Item{dst, off, bytes};
        off += (bytes + 63) & ~(size_t)63;
        return VO_OK;
    }
    int finish() {
        VO_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n; i++) std::memcpy(items[i].dst, base + items[i].off, items[i].bytes);
        return VO_OK;
    }
};

// host -> device through the pinned buffer, at byte offset `off` of it (the caller sized it with ctx_bounce)
static int bounce_up(vo_ctx* ctx, uint8_t* up, size_t off, void* dev, const void* src, size_t bytes) {
    if (!bytes) return VO_OK;
    std::memcpy(up + off, src, bytes);
    VO_HIP(hipMemcpyAsync(dev, up + off, bytes, hipMemcpyHostToDevice, ctx->stream));
    return VO_OK;
}

static int reset_counters(vo_ctx* ctx, FrameSet& fs) {
    VO_HIP(hipMemsetAsync(fs.cnt, 0, (size_t)fs.F * sizeof(FrameCounters), ctx->stream));
    VO_HIP(hipMemsetAsync(fs.cls_cnt, 0, kClsCntInts * sizeof(int), ctx->stream));
    VO_HIP(hipMemsetAsync(fs.cls_hist, 0, ((size_t)4 * fs.F * fs.nbins() + 4) * sizeof(int), ctx->stream));
    return VO_OK;
}

// ---------------------------------------------------------------- a1 / a2 inspection entry points
extern "C" int vo_integral(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, int32_t* sum) {
    if (!ctx || !sum) return VO_ERR_INVALID_ARG;
    VO_TRY(check_image(img, rows, cols, step));
    VO_HIP(hipSetDevice(ctx->device));
    VO_TRY(vo_frameset_alloc(ctx->fs, 1, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx)));
    VO_TRY(upload_frame(ctx, ctx->fs, 0, img, step));
    VO_TRY(vo_enqueue_integral(ctx, ctx->fs, ctx->stream));
    VO_HIP(hipMemcpyAsync(sum, ctx->fs.sum, ctx->fs.sum_stride() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_hessian_layer(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, int octave, int layer,
                                float* det, float* trace) {
    if (!ctx || !det || !trace || octave < 0 || octave >= kNumOctaves || layer < 0 || layer >= kLayersPerOctave)
        return VO_ERR_INVALID_ARG;
    VO_TRY(check_image(img, rows, cols, step));
    VO_HIP(hipSetDevice(ctx->device));
    VO_TRY(vo_frameset_alloc(ctx->fs, 1, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx)));
    VO_TRY(upload_frame(ctx, ctx->fs, 0, img, step));
    VO_TRY(vo_enqueue_integral(ctx, ctx->fs, ctx->stream));
    const size_t n = (size_t)(rows >> octave) * (cols >> octave);
    void* tmp = nullptr;
    VO_TRY(vo_ctx_tmp(ctx, 2 * n * sizeof(float) + 16, &tmp));
    float* d_det = (float*)tmp;
    float* d_tr = d_det + n;
    VO_TRY(vo_enqueue_hessian_layer(ctx, ctx->fs, 0, octave, layer, d_det, d_tr));
    VO_HIP(hipMemcpyAsync(det, d_det, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipMemcpyAsync(trace, d_tr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// det planes of all 20 layers as the PRODUCT kernels (k_hessian_nms_fused / k_hessian_nms_lds) hold them in LDS when their
// NMS runs - not the inspection kernel of vo_hessian_layer.  Packed layer after layer (octave-major), [rows >> o][cols >> o].
extern "C" int vo_hessian_product_det(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, float* det_all,
                                      size_t n_floats) {
    if (!ctx || !det_all) return VO_ERR_INVALID_ARG;
    // the inspection reads the LDS planes of the tuned kernels: five layers per octave, all four octaves
    if (ctx->prm.n_octave_layers != kLayersPerOctave - 2 || ctx->prm.n_octaves != kNumOctaves) return VO_ERR_UNSUPPORTED;
    VO_TRY(check_image(img, rows, cols, step));
    size_t S = 0;
    for (int o = 0; o < kNumOctaves; o++) S += (size_t)kLayersPerOctave * (rows >> o) * (cols >> o);
    if (n_floats != S) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(ctx->device));
    FrameSet& fs = ctx->fs;
    VO_TRY(vo_frameset_alloc(fs, 1, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx)));
    VO_TRY(reset_counters(ctx, fs));
    VO_TRY(upload_frame(ctx, fs, 0, img, step));
    VO_TRY(vo_enqueue_integral(ctx, fs, ctx->stream));
    void* tmp = nullptr;
    VO_TRY(vo_ctx_tmp(ctx, S * sizeof(float), &tmp));
    VO_HIP(hipMemsetAsync(tmp, 0xff, S * sizeof(float), ctx->stream));  // NaN pattern: a sample no tile wrote would show
    VO_TRY(vo_enqueue_hessian_nms(ctx, fs, true, (float*)tmp));
    VO_HIP(hipMemcpyAsync(det_all, tmp, S * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// ---------------------------------------------------------------- a1-a6
extern "C" int vo_detect_describe(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, vo_keypoint* kps,
                                  float* desc64, int cap, int* n) {
    if (!ctx || !kps || !n || cap < 0) return VO_ERR_INVALID_ARG;
    *n = 0;
    VO_TRY(check_image(img, rows, cols, step));
    VO_HIP(hipSetDevice(ctx->device));
    FrameSet& fs = ctx->fs;
    VO_TRY(vo_frameset_alloc(fs, 1, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx)));
    VO_TRY(reset_counters(ctx, fs));
    VO_TRY(upload_frame(ctx, fs, 0, img, step));
    VO_TRY(vo_enqueue_detect(ctx, fs, desc64 != nullptr));
    FrameCounters c;
    {
        BounceDown dn;
        VO_TRY(dn.begin(ctx, sizeof(c)));
        VO_TRY(dn.add(&c, fs.cnt, sizeof(c)));
        VO_TRY(dn.finish());
    }
    *n = c.n_raw;
    if (c.n_raw > fs.cap || c.n_raw > cap) return VO_ERR_CAPACITY;
    if (c.n_raw > 0) {
        BounceDown dn;
        VO_TRY(dn.begin(ctx, (size_t)c.n_raw * (sizeof(vo_keypoint) + fs.dlen * sizeof(float))));
        VO_TRY(dn.add(kps, fs.kp, (size_t)c.n_raw * sizeof(vo_keypoint)));
        if (desc64) VO_TRY(dn.add(desc64, fs.desc, (size_t)c.n_raw * fs.dlen * sizeof(float)));
        VO_TRY(dn.finish());
    }
    return VO_OK;
}

// ---------------------------------------------------------------- pair buffers
int vo_pairset_alloc(PairSet& ps, int P, int kp_cap, int match_cap, int max_hyp) {
    if (ps.P == P && ps.kp_cap == kp_cap && ps.match_cap == match_cap && ps.max_hyp == max_hyp && ps.st) return VO_OK;
    vo_pairset_free(ps);
    ps.P = P;
    ps.kp_cap = kp_cap;
    ps.match_cap = match_cap;
    ps.max_hyp = max_hyp;
    ps.n_split = 8;
    const size_t pk = (size_t)P * kp_cap, pm = (size_t)P * match_cap, ph = (size_t)P * max_hyp;
    VO_TRY(dalloc(ps.cand_val, pk * 32));
    VO_TRY(dalloc(ps.cand_idx, pk * 32));
    VO_TRY(dalloc(ps.redo_list, pk));
    VO_TRY(dalloc(ps.redo_cnt, 1));
    VO_HIP(hipMemset(ps.redo_cnt, 0, sizeof(int)));
    VO_TRY(dalloc(ps.knn_idx, pk * 2));
    VO_TRY(dalloc(ps.knn_dist, pk * 2));
    VO_TRY(dalloc(ps.p1, pm * 2));
    VO_TRY(dalloc(ps.p2, pm * 2));
    VO_TRY(dalloc(ps.matches, pm));
    VO_TRY(dalloc(ps.m1, pm * 2));
    VO_TRY(dalloc(ps.m2, pm * 2));
    VO_TRY(dalloc(ps.tuples, ph * 5));
    VO_TRY(dalloc(ps.models, ph * kMaxModels * 9));
    VO_TRY(dalloc(ps.nmodels, ph));
    VO_TRY(dalloc(ps.counts, ph * kMaxModels));
    VO_TRY(dalloc(ps.mask, pm));
    VO_TRY(dalloc(ps.in1, pm * 2));
    VO_TRY(dalloc(ps.in2, pm * 2));
    VO_TRY(dalloc(ps.X4, pm * 4));
    VO_TRY(dalloc(ps.Xpre, pm * 3));
    VO_TRY(dalloc(ps.Xk, pm * 3));
    VO_TRY(dalloc(ps.flag, pm));
    VO_TRY(dalloc(ps.tmp, pm * 4));
    VO_TRY(dalloc(ps.st, (size_t)P));
    VO_HIP(hipMemset(ps.st, 0, (size_t)P * sizeof(PairState)));
    VO_TRY(dalloc(ps.frames, (size_t)P));
    {
        std::vector<int2> f((size_t)P);
        for (int p = 0; p < P; p++) f[p] = make_int2(2 * p, 2 * p + 1);
        VO_HIP(hipMemcpy(ps.frames, f.data(), f.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    return VO_OK;
}

void vo_pairset_free(PairSet& ps) {
    dfree(ps.frames);
    dfree(ps.cand_val);
    dfree(ps.cand_idx);
    dfree(ps.redo_list);
    dfree(ps.redo_cnt);
    dfree(ps.knn_idx);
    dfree(ps.knn_dist);
    dfree(ps.p1);
    dfree(ps.p2);
    dfree(ps.matches);
    dfree(ps.m1);
    dfree(ps.m2);
    dfree(ps.tuples);
    dfree(ps.models);
    dfree(ps.nmodels);
    dfree(ps.counts);
    dfree(ps.mask);
    dfree(ps.in1);
    dfree(ps.in2);
    dfree(ps.X4);
    dfree(ps.Xpre);
    dfree(ps.Xk);
    dfree(ps.flag);
    dfree(ps.tmp);
    dfree(ps.st);
    ps.P = ps.kp_cap = ps.match_cap = ps.max_hyp = 0;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// The context's own pair buffers (P = 1) of the single-call entry points only ever GROW: the node calls opt_DetectFeatures,
// optRelativePose and triangPoints once per frame, with different sizes, and re-allocating ~24 device buffers between them
// (hipFree synchronises the device) cost more than the kernels.  With one pair the strides do not matter to any kernel.
static int ctx_pairset(vo_ctx* ctx, int kp_cap, int match_cap) {
    PairSet& ps = ctx->ps;
    if (ps.st && ps.P == 1) {
        if (ps.kp_cap >= kp_cap && ps.match_cap >= match_cap) return VO_OK;
        kp_cap = std::max(kp_cap, ps.kp_cap);
        match_cap = std::max(match_cap, ps.match_cap);
    }
    if (ctx->d_redo_cnt == ps.redo_cnt) ctx->d_redo_cnt = nullptr;  // about to be freed
    return vo_pairset_alloc(ps, 1, kp_cap, match_cap, kMaxHyp);
}

static void init_state(PairState& S, int n, const double* K) {
    std::memset(&S, 0, sizeof(S));
    S.n_match = S.n_used = n;
    S.best_hyp = -1;
    if (K) std::memcpy(S.K, K, 9 * sizeof(double));
    S.R[0] = S.R[4] = S.R[8] = 1.0;
}

// single pair of point lists on the device (P = 1)
static int stage_points(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, PairState& S, bool as_inliers) {
    PairSet& ps = ctx->ps;
    VO_TRY(ctx_pairset(ctx, 64, round_up(std::max(n, 64), 64)));
    float* d1 = as_inliers ? ps.in1 : ps.p1;
    float* d2 = as_inliers ? ps.in2 : ps.p2;
    const size_t pb = (size_t)n * 2 * sizeof(float), sb = (sizeof(S) + 63) & ~(size_t)63;
    uint8_t* up = nullptr;
    VO_TRY(ctx_bounce(ctx, sb + 2 * pb, false, &up));
    VO_TRY(bounce_up(ctx, up, 0, ps.st, &S, sizeof(S)));
    VO_TRY(bounce_up(ctx, up, sb, d1, p1xy, pb));
    VO_TRY(bounce_up(ctx, up, sb + pb, d2, p2xy, pb));
    return VO_OK;
}

static int fetch_state(vo_ctx* ctx, PairSet& ps, int pair, PairState& S) {
    uint8_t* dn = nullptr;
    VO_TRY(ctx_bounce(ctx, sizeof(S), true, &dn));
    VO_HIP(hipMemcpyAsync(dn, ps.st + pair, sizeof(S), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(&S, dn, sizeof(S));
    return VO_OK;
}

// ---------------------------------------------------------------- a7
extern "C" int vo_match_knn2(vo_ctx* ctx, const float* descA, int nA, const float* descB, int nB, int32_t* idx2,
                             float* dist2) {
    if (!ctx || !descA || !descB || !idx2 || !dist2 || nA < 0 || nB < 0) return VO_ERR_INVALID_ARG;
    if (nA == 0) return VO_OK;
    VO_HIP(hipSetDevice(ctx->device));
    const int cap = round_up(std::max(std::max(nA, nB), 64), 64);
    FrameSet& fs = ctx->fs;
    VO_TRY(vo_frameset_alloc(fs, 2, 16, 16, cap, vo_desc_len(ctx)));
    VO_TRY(ctx_pairset(ctx, cap, 64));
    FrameCounters c[2];
    std::memset(c, 0, sizeof(c));
    c[0].n_raw = c[0].n_kp = nA;
    c[1].n_raw = c[1].n_kp = nB;
    const size_t bA = (size_t)nA * fs.dlen * sizeof(float), bB = (size_t)nB * fs.dlen * sizeof(float);
    uint8_t* up = nullptr;
    VO_TRY(ctx_bounce(ctx, 64 + bA + bB, false, &up));
    VO_TRY(bounce_up(ctx, up, 0, fs.cnt, c, sizeof(c)));
    VO_TRY(bounce_up(ctx, up, 64, fs.desc, descA, bA));
    VO_TRY(bounce_up(ctx, up, 64 + bA, fs.desc + (size_t)cap * fs.dlen, descB, bB));
    VO_TRY(vo_enqueue_match(ctx, fs, ctx->ps, false, ctx->stream));
    BounceDown dn;
    VO_TRY(dn.begin(ctx, (size_t)nA * 2 * (sizeof(int) + sizeof(float))));
    VO_TRY(dn.add(idx2, ctx->ps.knn_idx, (size_t)nA * 2 * sizeof(int)));
    VO_TRY(dn.add(dist2, ctx->ps.knn_dist, (size_t)nA * 2 * sizeof(float)));
    return dn.finish();
}

// ---------------------------------------------------------------- opt_DetectFeatures
extern "C" int vo_detect_match(vo_ctx* ctx, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2, int rows,
                               int cols, float* p1xy, float* p2xy, vo_dmatch* m, int cap, int* n) {
    if (!ctx || !n || cap < 0) return VO_ERR_INVALID_ARG;
    *n = 0;
    VO_TRY(check_image(img1, rows, cols, step1));
    VO_TRY(check_image(img2, rows, cols, step2));
    VO_HIP(hipSetDevice(ctx->device));
    FrameSet& fs = ctx->fs;
    PairSet& ps = ctx->ps;
    VO_TRY(vo_frameset_alloc(fs, 2, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx)));
    VO_TRY(ctx_pairset(ctx, ctx->prm.kp_capacity, round_up(ctx->prm.match_capacity, 64)));
    VO_TRY(reset_counters(ctx, fs));
    VO_TRY(upload_frame(ctx, fs, 0, img1, step1));
    VO_TRY(upload_frame(ctx, fs, 1, img2, step2));
    VO_TRY(vo_enqueue_detect(ctx, fs, true));
    VO_TRY(vo_enqueue_match(ctx, fs, ps, true, ctx->stream));
    // counters, pair state and the (capacity-sized) result arrays come down through the pinned bounce buffer in one go:
    // one synchronisation, then plain memcpys of the n_match entries the caller gets
    const size_t mc = (size_t)ps.match_cap;
    const size_t o_cnt = 0, o_st = 64, o_p1 = o_st + ((sizeof(PairState) + 63) & ~(size_t)63), o_p2 = o_p1 + mc * 2 * sizeof(float);
    const size_t o_m = o_p2 + mc * 2 * sizeof(float), total = o_m + mc * sizeof(vo_dmatch);
    static_assert(2 * sizeof(FrameCounters) <= 64, "bounce layout");
    uint8_t* dn = nullptr;
    VO_TRY(ctx_bounce(ctx, total, true, &dn));
    VO_HIP(hipMemcpyAsync(dn + o_cnt, fs.cnt, 2 * sizeof(FrameCounters), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipMemcpyAsync(dn + o_st, ps.st, sizeof(PairState), hipMemcpyDeviceToHost, ctx->stream));
    if (p1xy) VO_HIP(hipMemcpyAsync(dn + o_p1, ps.p1, mc * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (p2xy) VO_HIP(hipMemcpyAsync(dn + o_p2, ps.p2, mc * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (m) VO_HIP(hipMemcpyAsync(dn + o_m, ps.matches, mc * sizeof(vo_dmatch), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    PairState S;
    FrameCounters c[2];
    std::memcpy(c, dn + o_cnt, sizeof(c));
    std::memcpy(&S, dn + o_st, sizeof(S));
    if (c[0].n_raw > fs.cap || c[1].n_raw > fs.cap) return VO_ERR_CAPACITY;
    *n = S.n_match;
    if (S.n_match > ps.match_cap || S.n_match > cap) return VO_ERR_CAPACITY;
    const size_t k = (size_t)S.n_match;
    if (k) {
        if (p1xy) std::memcpy(p1xy, dn + o_p1, k * 2 * sizeof(float));
        if (p2xy) std::memcpy(p2xy, dn + o_p2, k * 2 * sizeof(float));
        if (m) std::memcpy(m, dn + o_m, k * sizeof(vo_dmatch));
    }
    return VO_OK;
}

// ---------------------------------------------------------------- a10
extern "C" int vo_essential_ransac(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9], double prob,
                                   double thr_px, int max_iters, uint64_t seed, double E[9], uint8_t* mask, int* n_inl,
                                   int* iters_run) {
    if (!ctx || !K || !E || !mask || n < 0 || (n > 0 && (!p1xy || !p2xy))) return VO_ERR_INVALID_ARG;
    if (max_iters < 1 || max_iters > kMaxHyp) return VO_ERR_INVALID_ARG;
    if (n_inl) *n_inl = 0;
    if (iters_run) *iters_run = 0;
    if (n < 5) return VO_ERR_TOO_FEW_POINTS;
    VO_HIP(hipSetDevice(ctx->device));
    PairState S;
    init_state(S, n, K);
    VO_TRY(stage_points(ctx, p1xy, p2xy, n, S, false));
    GeoOptions o;
    o.run_pose = o.run_triang = false;
    o.prob = prob;
    o.thr_px = thr_px;
    o.max_iters = max_iters;
    o.seed = seed;
    VO_TRY(vo_enqueue_geometry(ctx, ctx->ps, o, ctx->stream));
    VO_HIP(hipMemcpyAsync(mask, ctx->ps.mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    VO_TRY(fetch_state(ctx, ctx->ps, 0, S));
    if (iters_run) *iters_run = S.iters_run;
    if (S.status != VO_OK) return S.status;
    std::memcpy(E, S.E, 9 * sizeof(double));
    if (n_inl) *n_inl = S.n_inl;
    return VO_OK;
}

extern "C" int vo_five_point(vo_ctx* ctx, const double* x1, const double* x2, int n_hyp, double* E_out, int32_t* n_models) {
    if (!ctx || !x1 || !x2 || !E_out || !n_models || n_hyp < 0) return VO_ERR_INVALID_ARG;
    if (n_hyp == 0) return VO_OK;
    VO_HIP(hipSetDevice(ctx->device));
    const size_t pts = (size_t)n_hyp * 10 * sizeof(double), eb = (size_t)n_hyp * 90 * sizeof(double);
    const size_t tb = (size_t)n_hyp * 5 * sizeof(int), nb = (size_t)n_hyp * sizeof(int);
    void* tmp = nullptr;
    VO_TRY(vo_ctx_tmp(ctx, 2 * pts + eb + tb + nb + 64, &tmp));
    double* d_x1 = (double*)tmp;
    double* d_x2 = d_x1 + (size_t)n_hyp * 10;
    double* d_E = d_x2 + (size_t)n_hyp * 10;
    int* d_t = (int*)(d_E + (size_t)n_hyp * 90);
    int* d_nm = d_t + (size_t)n_hyp * 5;
    std::vector<int> tup((size_t)n_hyp * 5);
    for (size_t i = 0; i < tup.size(); i++) tup[i] = (int)i;
    VO_HIP(hipMemcpyAsync(d_x1, x1, pts, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(d_x2, x2, pts, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(d_t, tup.data(), tb, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemsetAsync(d_E, 0, eb, ctx->stream));
    VO_TRY(vo_enqueue_five_point(ctx, d_x1, d_x2, n_hyp, d_t, d_E, d_nm));
    VO_HIP(hipMemcpyAsync(E_out, d_E, eb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipMemcpyAsync(n_models, d_nm, nb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// normalisation kernel lives in vo_geometry.hip (k_geo_prep); here the points go through a PairSet
extern "C" int vo_sampson_masks(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9], double thr_px,
                                const double* E, int n_models, uint8_t* masks, int32_t* counts) {
    if (!ctx || !K || !E || !masks || !counts || n < 0 || n_models < 0 || (n > 0 && (!p1xy || !p2xy)))
        return VO_ERR_INVALID_ARG;
    for (int k = 0; k < n_models; k++) counts[k] = 0;
    if (n == 0 || n_models == 0) return VO_OK;
    VO_HIP(hipSetDevice(ctx->device));
    PairState S;
    init_state(S, n, K);
    VO_TRY(stage_points(ctx, p1xy, p2xy, n, S, false));
    GeoOptions o;
    o.run_ransac = o.run_pose = o.run_triang = false;
    VO_TRY(vo_enqueue_normalise(ctx, ctx->ps, thr_px));
    void* tmp = nullptr;
    const size_t eb = (size_t)n_models * 9 * sizeof(double), mb = (size_t)n_models * n, cb = (size_t)n_models * sizeof(int);
    VO_TRY(vo_ctx_tmp(ctx, eb + cb + mb + 64, &tmp));
    double* d_E = (double*)tmp;
    int* d_c = (int*)(d_E + (size_t)n_models * 9);
    uint8_t* d_m = (uint8_t*)(d_c + n_models);
    VO_HIP(hipMemcpyAsync(d_E, E, eb, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemsetAsync(d_c, 0, cb, ctx->stream));
    VO_TRY(vo_enqueue_sampson_masks(ctx, ctx->ps.m1, ctx->ps.m2, n, vo_thr_sq(thr_px, K), d_E, n_models, d_m, d_c));
    VO_HIP(hipMemcpyAsync(masks, d_m, mb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipMemcpyAsync(counts, d_c, cb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// ---------------------------------------------------------------- a12
extern "C" int vo_recover_pose(vo_ctx* ctx, const double E[9], const float* p1xy, const float* p2xy, int n, const double K[9],
                               double dist_thresh, double R[9], double t[3], int* n_good) {
    if (!ctx || !E || !K || !R || !t || n <= 0 || !p1xy || !p2xy) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(ctx->device));
    PairState S;
    init_state(S, n, K);
    S.n_inl = n;
    S.status = VO_OK;
    std::memcpy(S.E, E, 9 * sizeof(double));
    VO_TRY(stage_points(ctx, p1xy, p2xy, n, S, true));
    GeoOptions o;
    o.run_ransac = o.run_triang = false;
    o.gate = false;
    o.dist_thresh = dist_thresh;
    VO_TRY(vo_enqueue_geometry(ctx, ctx->ps, o, ctx->stream));
    VO_TRY(fetch_state(ctx, ctx->ps, 0, S));
    std::memcpy(R, S.R, 9 * sizeof(double));
    std::memcpy(t, S.t, 3 * sizeof(double));
    if (n_good) *n_good = S.n_good;
    return VO_OK;
}

// ---------------------------------------------------------------- optRelativePose
extern "C" int vo_relative_pose(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9], double R_io[9],
                                double t_io[3], float* in1xy, float* in2xy, int* n_inl, int* success) {
    if (!ctx || !K || !R_io || !t_io || !n_inl || !success || n < 0 || (n > 0 && (!p1xy || !p2xy))) return VO_ERR_INVALID_ARG;
    *n_inl = 0;
    *success = 0;
    if (n < 5) return VO_ERR_TOO_FEW_POINTS;
    VO_HIP(hipSetDevice(ctx->device));
    PairState S;
    init_state(S, n, K);
    std::memcpy(S.R, R_io, 9 * sizeof(double));
    std::memcpy(S.t, t_io, 3 * sizeof(double));
    VO_TRY(stage_points(ctx, p1xy, p2xy, n, S, false));
    GeoOptions o;
    o.run_triang = false;
    o.prob = ctx->prm.ransac_prob;
    o.thr_px = ctx->prm.ransac_threshold;
    o.max_iters = ctx->prm.ransac_max_iters;
    o.seed = ctx->prm.ransac_seed;
    o.dist_thresh = ctx->prm.distance_threshold;
    VO_TRY(vo_enqueue_geometry(ctx, ctx->ps, o, ctx->stream));
    VO_TRY(fetch_state(ctx, ctx->ps, 0, S));
    if (S.status != VO_OK) return S.status;
    *n_inl = S.n_inl;
    if (S.n_inl > 0) {
        const size_t ib = (size_t)S.n_inl * 2 * sizeof(float);
        BounceDown dn;
        VO_TRY(dn.begin(ctx, 2 * ib));
        if (in1xy) VO_TRY(dn.add(in1xy, ctx->ps.in1, ib));
        if (in2xy) VO_TRY(dn.add(in2xy, ctx->ps.in2, ib));
        VO_TRY(dn.finish());
    }
    *success = S.success;
    if (S.success) {
        std::memcpy(R_io, S.R, 9 * sizeof(double));
        std::memcpy(t_io, S.t, 3 * sizeof(double));
    }
    return VO_OK;
}

// ---------------------------------------------------------------- a13 / triangPoints
extern "C" int vo_triangulate(vo_ctx* ctx, const double P0[12], const double P1[12], const float* p1xy, const float* p2xy, int n,
                              float* X4) {
    if (!ctx || !P0 || !P1 || !X4 || n < 0 || (n > 0 && (!p1xy || !p2xy))) return VO_ERR_INVALID_ARG;
    if (n == 0) return VO_OK;
    VO_HIP(hipSetDevice(ctx->device));
    void* tmp = nullptr;
    const size_t pb = (size_t)n * 2 * sizeof(float), xb = (size_t)n * 4 * sizeof(float);
    VO_TRY(vo_ctx_tmp(ctx, 24 * sizeof(double) + 2 * pb + xb + 64, &tmp));
    double* d_P = (double*)tmp;
    float* d_p1 = (float*)(d_P + 24);
    float* d_p2 = d_p1 + (size_t)n * 2;
    float* d_X = d_p2 + (size_t)n * 2;
    double P01[24];
    std::memcpy(P01, P0, 12 * sizeof(double));
    std::memcpy(P01 + 12, P1, 12 * sizeof(double));
    VO_HIP(hipMemcpyAsync(d_P, P01, sizeof(P01), hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(d_p1, p1xy, pb, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(d_p2, p2xy, pb, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));  // P01 is a stack buffer
    VO_TRY(vo_enqueue_triangulate(ctx, d_P, d_p1, d_p2, n, d_X));
    VO_HIP(hipMemcpyAsync(X4, d_X, xb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_triang_points(vo_ctx* ctx, const float* in1xy, const float* in2xy, int n, const double R[9], const double t[3],
                                const double K[9], double* Xk, int cap, int* m) {
    if (!ctx || !R || !t || !K || !m || n < 0 || cap < 0 || (n > 0 && (!in1xy || !in2xy))) return VO_ERR_INVALID_ARG;
    *m = 0;
    if (n == 0) return VO_OK;
    VO_HIP(hipSetDevice(ctx->device));
    PairState S;
    init_state(S, n, K);
    S.n_inl = n;
    S.success = 1;
    std::memcpy(S.R, R, 9 * sizeof(double));
    std::memcpy(S.t, t, 3 * sizeof(double));
    VO_TRY(stage_points(ctx, in1xy, in2xy, n, S, true));
    GeoOptions o;
    o.run_ransac = o.run_pose = false;
    VO_TRY(vo_enqueue_geometry(ctx, ctx->ps, o, ctx->stream));
    VO_TRY(fetch_state(ctx, ctx->ps, 0, S));
    *m = S.n_world;
    if (S.n_world > cap) return VO_ERR_CAPACITY;
    if (S.n_world > 0 && Xk) {
        const size_t wb = (size_t)S.n_world * sizeof(double);
        BounceDown dn;
        VO_TRY(dn.begin(ctx, 3 * wb));
        for (int r = 0; r < 3; r++) VO_TRY(dn.add(Xk + (size_t)r * cap, ctx->ps.Xk + (size_t)r * ctx->ps.match_cap, wb));
        VO_TRY(dn.finish());
    }
    return VO_OK;
}

// ---------------------------------------------------------------- batch
static int batch_create_impl(vo_ctx* ctx, int n_frames, int n_pairs, bool is_seq, int rows, int cols, vo_batch** out) {
    if (!ctx || !out || n_pairs <= 0 || n_frames < 2) return VO_ERR_INVALID_ARG;
    *out = nullptr;
    if (rows < 16 || cols < 16 || rows > 8192 || cols > 8192) return VO_ERR_UNSUPPORTED;
    VO_HIP(hipSetDevice(ctx->device));
    vo_batch* b = new vo_batch();
    b->ctx = ctx;
    b->n_pairs = n_pairs;
    b->n_frames = n_frames;
    b->is_seq = is_seq;
    b->rows = rows;
    b->cols = cols;
    int st = vo_frameset_alloc(b->fs, n_frames, rows, cols, ctx->prm.kp_capacity, vo_desc_len(ctx));
    if (st == VO_OK) st = vo_pairset_alloc(b->ps, n_pairs, ctx->prm.kp_capacity, round_up(ctx->prm.match_capacity, 64), kMaxHyp);
    if (st != VO_OK) {
        vo_batch_destroy(b);
        return st;
    }
    const size_t fbytes = (size_t)n_frames * rows * cols;
    b->d_img[0] = b->fs.img;  // owned by the frame set
    bool ok = hipMalloc((void**)&b->d_img[1], fbytes + 256) == hipSuccess &&
              hipStreamCreateWithFlags(&b->stream_copy, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&b->ev_detect, hipEventDisableTiming) == hipSuccess;
    {  // second set of the feature buffers the matcher reads (FeatBuf): ~9.3 KB per keypoint slot
        const FrameSet& fs = b->fs;
        const size_t fc = (size_t)n_frames * fs.cap;
        b->feat[0] = FeatBuf{fs.kp, fs.desc, fs.descT, fs.nrm, fs.nmax, fs.cnt};
        FeatBuf& x = b->feat[1];
        ok = ok && dalloc(x.kp, fc) == VO_OK && dalloc(x.desc, fc * fs.dlen) == VO_OK && dalloc(x.descT, fc * 72) == VO_OK &&
             dalloc(x.nrm, fc) == VO_OK && dalloc(x.nmax, (size_t)n_frames) == VO_OK && dalloc(x.cnt, (size_t)n_frames) == VO_OK &&
             hipMemset(x.cnt, 0, (size_t)n_frames * sizeof(FrameCounters)) == hipSuccess;
    }
    for (int s = 0; s < 2 && ok; s++) {
        RunHost& rh = b->rh[s];
        ok = hipHostMalloc((void**)&b->h_stage[s], fbytes, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&rh.init, (size_t)n_pairs * sizeof(PairState), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&rh.res, (size_t)n_pairs * sizeof(PairState), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&rh.frames, (size_t)n_pairs * sizeof(int2), hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void**)&rh.cnt, (size_t)n_frames * sizeof(FrameCounters), hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&b->ev_h2d[s], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b->ev_img_free[s], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b->ev_prep[s], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&rh.ev_state, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&rh.ev_done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        vo_set_hip_error("batch allocation (hipMalloc / hipHostMalloc / hipEventCreate)", hipGetLastError(), __FILE__, __LINE__);
        vo_batch_destroy(b);
        return VO_ERR_HIP;
    }
    b->h_init.resize(n_pairs);
    b->rh[0].unset.assign(n_pairs, 0);
    b->rh[1].unset.assign(n_pairs, 0);
    b->h_frames.resize(n_pairs);
    b->staged.assign(n_frames, 0);
    for (int p = 0; p < n_pairs; p++) b->h_frames[p] = is_seq ? make_int2(p, p + 1) : make_int2(2 * p, 2 * p + 1);
    for (auto& S : b->h_init) init_state(S, 0, nullptr);
    *out = b;
    return VO_OK;
}

extern "C" int vo_batch_create(vo_ctx* ctx, int n_pairs, int rows, int cols, vo_batch** out) {
    return batch_create_impl(ctx, 2 * n_pairs, n_pairs, false, rows, cols, out);
}

// Sequence batch: n_frames frames, each detected and described ONCE, and n_frames - 1 pairs, pair p = (frame p, frame p + 1)
// until vo_seq_set_pair re-pairs it.  The reference re-detects prev_img in every iteration (visual_Odometry.cpp:296);
// for an offline replay the features of a frame serve both pairs it belongs to, which halves the detection work.
extern "C" int vo_seq_create(vo_ctx* ctx, int n_frames, int rows, int cols, vo_batch** out) {
    return batch_create_impl(ctx, n_frames, n_frames - 1, true, rows, cols, out);
}

// First frame staged into the fill slot since the latest run: its staging memory must have left the host (the upload issued
// two runs ago) and, on the device, the run that read the slot must be past description before the copy stream overwrites it.
static int begin_fill(vo_batch* b) {
    if (b->n_staged) return VO_OK;
    const int s = b->fill_slot;
    if (b->h2d_pending[s]) {
        VO_HIP(hipEventSynchronize(b->ev_h2d[s]));
        b->h2d_pending[s] = false;
    }
    if (b->img_free_pending[s]) {
        VO_HIP(hipStreamWaitEvent(b->stream_copy, b->ev_img_free[s], 0));
        b->img_free_pending[s] = false;
    }
    return VO_OK;
}

int vo_batch_frame_slot(vo_batch* b, int frame, uint8_t** d_dst) {
    if (!b || !d_dst || frame < 0 || frame >= b->n_frames) return VO_ERR_INVALID_ARG;
    VO_TRY(begin_fill(b));
    if (!b->staged[frame]) {
        b->staged[frame] = 1;
        b->n_staged++;
    }
    *d_dst = b->d_img[b->fill_slot] + (size_t)frame * b->fs.img_stride();
    return VO_OK;
}

static int stage_frame(vo_batch* b, int frame, const uint8_t* img, size_t step) {
    // the caller's buffer may be pageable and may be reused as soon as this returns: the frame goes into the batch's pinned
    // staging copy of the frame slot the latest run does not read.  The H2D transfers are issued by vo_batch_run, one per
    // contiguous range of staged frames, on the batch's COPY stream (128 separate 2 MB transfers beside a run's kernels cost
    // the kernels 2.4 ms per step; one 265 MB transfer costs them nothing - tools/pcie_probe*.py)
    uint8_t* d_dst = nullptr;
    VO_TRY(vo_batch_frame_slot(b, frame, &d_dst));
    b->staged[frame] = 2;  // 2 = the frame is in the pinned staging memory and still has to be uploaded
    const size_t fbytes = (size_t)b->rows * b->cols;
    uint8_t* dst = b->h_stage[b->fill_slot] + (size_t)frame * fbytes;
    if (step == (size_t)b->cols)
        std::memcpy(dst, img, fbytes);
    else
        for (int r = 0; r < b->rows; r++) std::memcpy(dst + (size_t)r * b->cols, img + (size_t)r * step, b->cols);
    return VO_OK;
}

extern "C" int vo_seq_set_frame(vo_batch* b, int frame, const uint8_t* img, size_t step) {
    if (!b || !b->is_seq || frame < 0 || frame >= b->n_frames) return VO_ERR_INVALID_ARG;
    VO_TRY(check_image(img, b->rows, b->cols, step));
    VO_HIP(hipSetDevice(b->ctx->device));
    return stage_frame(b, frame, img, step);
}

extern "C" int vo_seq_set_pair(vo_batch* b, int pair, int frame_a, int frame_b, const double K[9]) {
    if (!b || !b->is_seq || pair < 0 || pair >= b->n_pairs || !K) return VO_ERR_INVALID_ARG;
    if (frame_a < 0 || frame_a >= b->n_frames || frame_b < 0 || frame_b >= b->n_frames) return VO_ERR_INVALID_ARG;
    b->h_frames[pair] = make_int2(frame_a, frame_b);
    init_state(b->h_init[pair], 0, K);
    return VO_OK;
}

extern "C" void vo_batch_destroy(vo_batch* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    hipStreamSynchronize(b->ctx->stream_geo);
    if (b->stream_copy) hipStreamSynchronize(b->stream_copy);
    if (b->ctx->d_redo_cnt == b->ps.redo_cnt) b->ctx->d_redo_cnt = nullptr;
    if (b->d_img[0]) b->fs.img = b->d_img[0];
    if (b->feat[0].kp) {  // the frame set frees its own allocation, whichever set the latest run used
        const FeatBuf& x = b->feat[0];
        b->fs.kp = x.kp, b->fs.desc = x.desc, b->fs.descT = x.descT, b->fs.nrm = x.nrm, b->fs.nmax = x.nmax, b->fs.cnt = x.cnt;
    }
    {
        FeatBuf& x = b->feat[1];
        dfree(x.kp), dfree(x.desc), dfree(x.descT), dfree(x.nrm), dfree(x.nmax), dfree(x.cnt);
    }
    vo_frameset_free(b->fs);
    vo_pairset_free(b->ps);
    if (b->d_img[1]) hipFree(b->d_img[1]);
    for (int s = 0; s < 2; s++) {
        RunHost& rh = b->rh[s];
        if (b->h_stage[s]) hipHostFree(b->h_stage[s]);
        if (rh.init) hipHostFree(rh.init);
        if (rh.res) hipHostFree(rh.res);
        if (rh.frames) hipHostFree(rh.frames);
        if (rh.cnt) hipHostFree(rh.cnt);
        if (b->ev_h2d[s]) hipEventDestroy(b->ev_h2d[s]);
        if (b->ev_img_free[s]) hipEventDestroy(b->ev_img_free[s]);
        if (b->ev_prep[s]) hipEventDestroy(b->ev_prep[s]);
        if (rh.ev_state) hipEventDestroy(rh.ev_state);
        if (rh.ev_done) hipEventDestroy(rh.ev_done);
    }
    if (b->ev_detect) hipEventDestroy(b->ev_detect);
    if (b->h_bulk) hipHostFree(b->h_bulk);
    if (b->stream_copy) hipStreamDestroy(b->stream_copy);
    delete b;
}

extern "C" int vo_batch_set_pair(vo_batch* b, int pair, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2,
                                 const double K[9]) {
    if (!b || b->is_seq || pair < 0 || pair >= b->n_pairs || !K) return VO_ERR_INVALID_ARG;
    VO_TRY(check_image(img1, b->rows, b->cols, step1));
    VO_TRY(check_image(img2, b->rows, b->cols, step2));
    VO_HIP(hipSetDevice(b->ctx->device));
    VO_TRY(stage_frame(b, 2 * pair, img1, step1));
    VO_TRY(stage_frame(b, 2 * pair + 1, img2, step2));
    init_state(b->h_init[pair], 0, K);
    return VO_OK;
}

extern "C" int vo_batch_set_k(vo_batch* b, int pair, const double K[9]) {
    if (!b || b->is_seq || pair < 0 || pair >= b->n_pairs || !K) return VO_ERR_INVALID_ARG;
    init_state(b->h_init[pair], 0, K);
    return VO_OK;
}

extern "C" int vo_batch_run(vo_batch* b) {
    if (!b) return VO_ERR_INVALID_ARG;
    vo_ctx* ctx = b->ctx;
    VO_HIP(hipSetDevice(ctx->device));
    RunHost& rh = b->rh[b->runs & 1];
    // snapshot of K / initial pose (identity: every run starts every pair from R = I, t = 0, whatever the previous run left)
    // and of the pair -> frame map; the upload issued from this memory two runs ago has to be over
    if (rh.state_pending) {
        VO_HIP(hipEventSynchronize(rh.ev_state));
        rh.state_pending = false;
    }
    std::memcpy(rh.init, b->h_init.data(), b->h_init.size() * sizeof(PairState));
    // A pair whose K was never given (vo_batch_set_pair / vo_batch_set_k / vo_seq_set_pair not called for it) would divide by
    // zero on the device.  It does not fail the run: it runs with K = I so that nothing faults, and its OWN record comes back
    // with status VO_ERR_INVALID_ARG (vo_batch_get_run); the other pairs are unaffected.
    for (int p = 0; p < b->n_pairs; p++) {
        PairState& S = rh.init[p];
        rh.unset[p] = S.K[0] == 0.0 || S.K[4] == 0.0;
        if (rh.unset[p]) S.K[0] = S.K[4] = S.K[8] = 1.0;
    }
    std::memcpy(rh.frames, b->h_frames.data(), b->h_frames.size() * sizeof(int2));
    hipEvent_t img_ready = nullptr;
    if (b->n_staged > 0) {  // frames were staged since the latest run: this run reads the other slot
        const int s = b->fill_slot;
        const size_t fst = b->fs.img_stride();
        // frames this run gets from vo_prep_to_batch were written by kernels on the main stream: mark the slot, so that a
        // LATER carry-over out of it (copy stream) waits for them
        const bool has_prep = std::find(b->staged.begin(), b->staged.end(), (uint8_t)1) != b->staged.end();
        if (has_prep) {
            VO_HIP(hipEventRecord(b->ev_prep[s], ctx->stream));
            b->prep_pending[s] = true;
        } else {
            b->prep_pending[s] = false;  // every frame of the slot now comes through the copy stream itself
        }
        // carry-over out of a slot that holds prep-written frames: the copy stream is not ordered against those kernels.
        // (Frames carried INTO such a slot by an earlier copy are ordered by the copy stream; waiting covers both.)
        if (b->cur_slot >= 0 && b->prep_pending[b->cur_slot] &&
            std::find(b->staged.begin(), b->staged.end(), (uint8_t)0) != b->staged.end())
            VO_HIP(hipStreamWaitEvent(b->stream_copy, b->ev_prep[b->cur_slot], 0));
        for (int f = 0; f < b->n_frames;) {  // one transfer per run of frames with the same origin
            const uint8_t kind = b->staged[f];
            int e = f + 1;
            while (e < b->n_frames && b->staged[e] == kind) e++;
            const size_t off = (size_t)f * fst, bytes = (size_t)(e - f) * fst;
            if (kind == 2)  // staged by vo_batch_set_pair / vo_seq_set_frame: host -> device
                VO_HIP(hipMemcpyAsync(b->d_img[s] + off, b->h_stage[s] + off, bytes, hipMemcpyHostToDevice, b->stream_copy));
            else if (kind == 0 && b->cur_slot >= 0)  // not replaced: carries over from the slot the latest run read
                VO_HIP(hipMemcpyAsync(b->d_img[s] + off, b->d_img[b->cur_slot] + off, bytes, hipMemcpyDeviceToDevice, b->stream_copy));
            f = e;  // kind 1: written on the device by vo_prep_to_batch (main stream), nothing to copy
        }
        VO_HIP(hipEventRecord(b->ev_h2d[s], b->stream_copy));
        b->h2d_pending[s] = true;
        img_ready = b->ev_h2d[s];  // the detection waits for it on the stream its integral pass runs on
        // frames written by vo_prep_to_batch (kernels on the main stream): the auxiliary stream has to see them too
        if (has_prep) {
            VO_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            VO_HIP(hipStreamWaitEvent(ctx->stream_aux, ctx->ev_fork, 0));
        }
        b->cur_slot = s;
        b->fill_slot = s ^ 1;
        b->n_staged = 0;
        std::fill(b->staged.begin(), b->staged.end(), 0);
    }
    if (b->cur_slot < 0) {
        b->cur_slot = 0;
        b->fill_slot = 1;
    }
    b->fs.img = b->d_img[b->cur_slot];
    {
        // feature buffers of this run; the matcher of the run before the previous one read the same set (geometry stream)
        const FeatBuf& x = b->feat[b->runs & 1];
        b->fs.kp = x.kp, b->fs.desc = x.desc, b->fs.descT = x.descT, b->fs.nrm = x.nrm, b->fs.nmax = x.nmax, b->fs.cnt = x.cnt;
        if (b->runs >= 2) VO_HIP(hipStreamWaitEvent(ctx->stream, rh.ev_done, 0));
    }
    VO_TRY(reset_counters(ctx, b->fs));
    VO_TRY(vo_enqueue_detect(ctx, b->fs, true, true, img_ready));
    VO_HIP(hipEventRecord(b->ev_img_free[b->cur_slot], ctx->stream));
    b->img_free_pending[b->cur_slot] = true;
    // Everything from the matcher on goes to the geometry stream, in order behind the previous run's geometry (which may still
    // be working on this batch's pair buffers).  The main stream is free as soon as description is enqueued: the next run's
    // detection (VALU-bound) then runs beside this run's matcher (MFMA- and gather-bound) and its RANSAC / pose / triangulation
    // (a few single-wavefront kernels, ~1.6 ms of mostly idle GPU per step).
    hipStream_t sg = ctx->stream_geo;
    VO_HIP(hipEventRecord(b->ev_detect, ctx->stream));
    VO_HIP(hipStreamWaitEvent(sg, b->ev_detect, 0));
    VO_HIP(hipMemcpyAsync(b->ps.frames, rh.frames, (size_t)b->n_pairs * sizeof(int2), hipMemcpyHostToDevice, sg));
    VO_HIP(hipMemcpyAsync(b->ps.st, rh.init, (size_t)b->n_pairs * sizeof(PairState), hipMemcpyHostToDevice, sg));
    VO_HIP(hipEventRecord(rh.ev_state, sg));
    rh.state_pending = true;
    VO_TRY(vo_enqueue_match(ctx, b->fs, b->ps, true, sg));
    VO_HIP(hipMemcpyAsync(rh.cnt, b->fs.cnt, (size_t)b->n_frames * sizeof(FrameCounters), hipMemcpyDeviceToHost, sg));
    GeoOptions o;
    o.prob = ctx->prm.ransac_prob;
    o.thr_px = ctx->prm.ransac_threshold;
    o.max_iters = ctx->prm.ransac_max_iters;
    o.seed = ctx->prm.ransac_seed;
    o.dist_thresh = ctx->prm.distance_threshold;
    VO_TRY(vo_enqueue_geometry(ctx, b->ps, o, sg));
    // the result records travel with the run: they are on the host when ev_done fires
    VO_HIP(hipMemcpyAsync(rh.res, b->ps.st, (size_t)b->n_pairs * sizeof(PairState), hipMemcpyDeviceToHost, sg));
    VO_HIP(hipEventRecord(rh.ev_done, sg));
    // with the stage timers on every stage is measured alone: nothing of the next run starts before this one is over
    if (ctx->profiling) VO_HIP(hipStreamWaitEvent(ctx->stream, rh.ev_done, 0));
    b->runs++;
    return VO_OK;
}

extern "C" int vo_batch_sync(vo_batch* b) {
    if (!b) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(b->ctx->device));
    VO_HIP(hipStreamSynchronize(b->stream_copy));
    VO_HIP(hipStreamSynchronize(b->ctx->stream));
    if (b->runs > 0) VO_HIP(hipEventSynchronize(b->rh[(b->runs - 1) & 1].ev_done));
    return VO_OK;
}

// records of the latest run (age 0) or the one before it (age 1): waits for that run only
static int run_records(vo_batch* b, int age, RunHost** out) {
    if (age < 0 || age > 1 || b->runs - 1 - age < 0) return VO_ERR_INVALID_ARG;
    RunHost& rh = b->rh[(b->runs - 1 - age) & 1];
    VO_HIP(hipEventSynchronize(rh.ev_done));
    *out = &rh;
    return VO_OK;
}

extern "C" int vo_batch_get_run(vo_batch* b, int age, int pair, vo_pair_out* out) {
    if (!b || !out || pair < 0 || pair >= b->n_pairs) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(b->ctx->device));
    RunHost* rh = nullptr;
    VO_TRY(run_records(b, age, &rh));
    const PairState& S = rh->res[pair];
    std::memset(out, 0, sizeof(*out));
    out->n_kp1 = rh->cnt[rh->frames[pair].x].n_raw;
    out->n_kp2 = rh->cnt[rh->frames[pair].y].n_raw;
    out->n_match = S.n_match;
    out->status = S.status;
    if (out->n_kp1 > b->fs.cap || out->n_kp2 > b->fs.cap || S.n_match > b->ps.match_cap) out->status = VO_ERR_CAPACITY;
    out->n_inl = S.n_inl;
    out->n_good = S.n_good;
    out->n_world = S.n_world;
    out->success = S.success;
    out->iters = S.iters_run;
    std::memcpy(out->E, S.E, sizeof(out->E));
    std::memcpy(out->R, S.R, sizeof(out->R));
    std::memcpy(out->t, S.t, sizeof(out->t));
    if (rh->unset[pair]) {  // no camera matrix was given for this pair: nothing of its geometry means anything
        out->status = VO_ERR_INVALID_ARG;
        out->success = 0;
    }
    return VO_OK;
}

extern "C" int vo_batch_get(vo_batch* b, int pair, vo_pair_out* out) { return vo_batch_get_run(b, 0, pair, out); }

extern "C" int vo_batch_get_run_all(vo_batch* b, int age, vo_pair_out* outs) {
    if (!b || !outs) return VO_ERR_INVALID_ARG;
    for (int p = 0; p < b->n_pairs; p++) VO_TRY(vo_batch_get_run(b, age, p, outs + p));
    return VO_OK;
}

extern "C" int vo_batch_get_points(vo_batch* b, int pair, float* p1xy, float* p2xy, vo_dmatch* m, uint8_t* mask, int cap,
                                   double* Xk, int cap_world) {
    if (!b || pair < 0 || pair >= b->n_pairs || cap < 0 || cap_world < 0) return VO_ERR_INVALID_ARG;
    vo_ctx* ctx = b->ctx;
    VO_HIP(hipSetDevice(ctx->device));
    RunHost* rh = nullptr;
    VO_TRY(run_records(b, 0, &rh));  // the bulk arrays on the device are those of the latest run, and it has finished
    const PairState& S = rh->res[pair];
    const PairSet& ps = b->ps;
    const size_t mc = ps.match_cap;
    const int nm = std::min(S.n_used, (int)mc);
    if (nm > cap || S.n_world > cap_world) return VO_ERR_CAPACITY;
    BounceDown dn;  // through the context's pinned buffer: one synchronisation, no pinning of the caller's pages
    const size_t wb = (size_t)std::max(S.n_world, 0) * sizeof(double);
    VO_TRY(dn.begin(ctx, (size_t)nm * (4 * sizeof(float) + sizeof(vo_dmatch) + 1) + 3 * wb));
    if (nm > 0) {
        if (p1xy) VO_TRY(dn.add(p1xy, ps.p1 + pair * mc * 2, (size_t)nm * 2 * sizeof(float)));
        if (p2xy) VO_TRY(dn.add(p2xy, ps.p2 + pair * mc * 2, (size_t)nm * 2 * sizeof(float)));
        if (m) VO_TRY(dn.add(m, ps.matches + pair * mc, (size_t)nm * sizeof(vo_dmatch)));
        if (mask) VO_TRY(dn.add(mask, ps.mask + pair * mc, (size_t)nm));
    }
    if (Xk && S.n_world > 0)
        for (int r = 0; r < 3; r++) VO_TRY(dn.add(Xk + (size_t)r * cap_world, ps.Xk + (pair * 3 + r) * mc, wb));
    return dn.finish();
}

// Points of EVERY pair of the latest run in one download: strided device -> pinned copies on the batch's COPY stream (the main
// stream may already be busy with another batch of the context), one wait, one pass of host memcpys.
extern "C" int vo_batch_get_points_all(vo_batch* b, float* p1xy, float* p2xy, vo_dmatch* m, uint8_t* mask, int cap, double* Xk,
                                       int cap_world) {
    if (!b || cap < 0 || cap_world < 0) return VO_ERR_INVALID_ARG;
    vo_ctx* ctx = b->ctx;
    VO_HIP(hipSetDevice(ctx->device));
    RunHost* rh = nullptr;
    VO_TRY(run_records(b, 0, &rh));
    const PairSet& ps = b->ps;
    const size_t mc = ps.match_cap, P = (size_t)b->n_pairs;
    for (size_t p = 0; p < P; p++) {
        const PairState& S = rh->res[p];
        if (std::min(S.n_used, (int)mc) > cap || (Xk && S.n_world > cap_world)) return VO_ERR_CAPACITY;
    }
    const size_t w = std::min((size_t)cap, mc), ww = std::min((size_t)cap_world, mc);
    struct Part {
        void* dst;
        const void* src;
        size_t elem, width, rows, dst_width;
    };
    const Part parts[5] = {{p1xy, ps.p1, 2 * sizeof(float), w, P, (size_t)cap},   {p2xy, ps.p2, 2 * sizeof(float), w, P, (size_t)cap},
                           {m, ps.matches, sizeof(vo_dmatch), w, P, (size_t)cap}, {mask, ps.mask, 1, w, P, (size_t)cap},
                           {Xk, ps.Xk, sizeof(double), ww, 3 * P, (size_t)cap_world}};
    size_t total = 0;
    for (const Part& q : parts)
        if (q.dst && q.width) total += (q.elem * q.width * q.rows + 255) & ~(size_t)255;
    if (!total) return VO_OK;
    if (total > b->h_bulk_bytes) {  // the batch's own pinned buffer: nothing of this batch is in flight after run_records
        if (b->h_bulk) hipHostFree(b->h_bulk);
        b->h_bulk = nullptr;
        b->h_bulk_bytes = 0;
        const size_t nb = (total + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        VO_HIP(hipHostMalloc((void**)&b->h_bulk, nb, hipHostMallocDefault));
        b->h_bulk_bytes = nb;
    }
    size_t off = 0;
    for (const Part& q : parts) {
        if (!q.dst || !q.width) continue;
        VO_HIP(hipMemcpy2DAsync(b->h_bulk + off, q.elem * q.width, q.src, q.elem * mc, q.elem * q.width, q.rows, hipMemcpyDeviceToHost,
                                b->stream_copy));
        off += (q.elem * q.width * q.rows + 255) & ~(size_t)255;
    }
    VO_HIP(hipStreamSynchronize(b->stream_copy));
    off = 0;
    for (const Part& q : parts) {
        if (!q.dst || !q.width) continue;
        const size_t rb = q.elem * q.width;
        if (q.width == q.dst_width)
            std::memcpy(q.dst, b->h_bulk + off, rb * q.rows);
        else
            for (size_t r = 0; r < q.rows; r++) std::memcpy((uint8_t*)q.dst + r * q.elem * q.dst_width, b->h_bulk + off + r * rb, rb);
        off += (rb * q.rows + 255) & ~(size_t)255;
    }
    return VO_OK;
}

// ---------------------------------------------------------------- inspection of the last single-call RANSAC
// what: 0 = tuples [max_hyp][5] int32, 1 = nmodels [max_hyp] int32, 2 = counts [max_hyp][10] int32,
//       3 = models [max_hyp][10][9] double.  Copies min(bytes, size) bytes of pair 0 of the context's scratch set.
extern "C" int vo_debug_fetch(vo_ctx* ctx, int what, void* dst, size_t bytes) {
    if (!ctx || !dst) return VO_ERR_INVALID_ARG;
    if (what == 4) {  // queries of the last matcher launch (any batch of this context) that took the exact scan
        if (bytes < sizeof(int)) return VO_ERR_INVALID_ARG;
        VO_HIP(hipSetDevice(ctx->device));
        VO_HIP(hipStreamSynchronize(ctx->stream));
        VO_HIP(hipStreamSynchronize(ctx->stream_geo));
        if (!ctx->d_redo_cnt) return VO_ERR_INVALID_ARG;  // no matcher launch yet, or its batch is gone
        VO_HIP(hipMemcpy(dst, ctx->d_redo_cnt, sizeof(int), hipMemcpyDeviceToHost));
        return VO_OK;
    }
    if (!ctx->ps.st) return VO_ERR_INVALID_ARG;
    const PairSet& ps = ctx->ps;
    const void* src = nullptr;
    size_t size = 0;
    switch (what) {
        case 0: src = ps.tuples; size = (size_t)ps.max_hyp * 5 * sizeof(int); break;
        case 1: src = ps.nmodels; size = (size_t)ps.max_hyp * sizeof(int); break;
        case 2: src = ps.counts; size = (size_t)ps.max_hyp * kMaxModels * sizeof(int); break;
        case 3: src = ps.models; size = (size_t)ps.max_hyp * kMaxModels * 9 * sizeof(double); break;
        case 5: src = ps.cand_idx; size = (size_t)ps.kp_cap * 32 * sizeof(int); break;    // matcher shortlist of pair 0: block ids
        case 6: src = ps.cand_val; size = (size_t)ps.kp_cap * 32 * sizeof(float); break;  // block maxima / bounds
        default: return VO_ERR_INVALID_ARG;
    }
    VO_HIP(hipSetDevice(ctx->device));
    VO_HIP(hipMemcpy(dst, src, std::min(bytes, size), hipMemcpyDeviceToHost));
    return VO_OK;
}

extern "C" int vo_debug_set(vo_ctx* ctx, int key, int value) {
    if (!ctx) return VO_ERR_INVALID_ARG;
    switch (key) {
        case 0: ctx->force_seq_describe = value ? 1 : 0; return VO_OK;
        case 1: ctx->force_exact_match = value ? 1 : 0; return VO_OK;
        case 2: ctx->match_fp32 = value ? 1 : 0; return VO_OK;
        default: return VO_ERR_INVALID_ARG;
    }
}
