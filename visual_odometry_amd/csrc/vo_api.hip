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
StageScope::StageScope(vo_ctx* c, int s) : ctx(c), stage(s) {
    if (!ctx->profiling) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        a = b = nullptr;
        return;
    }
    hipEventRecord(a, ctx->stream);
}
StageScope::~StageScope() {
    if (!a) return;
    hipEventRecord(b, ctx->stream);
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

static const char* kStageNames[ST_COUNT] = {"integral",      "hessian_nms",        "sort",          "describe",
                                            "match_prep",    "match_mfma",         "match_select",  "ransac_sample",
                                            "ransac_solve_score", "ransac_select", "pose",          "triangulate"};

extern "C" int vo_stage_count(void) { return ST_COUNT; }
extern "C" const char* vo_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }
extern "C" int vo_profile_enable(vo_ctx* ctx, int on) {
    if (!ctx) return VO_ERR_INVALID_ARG;
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

int vo_frameset_alloc(FrameSet& fs, int F, int H, int W, int cap) {
    if (fs.F == F && fs.H == H && fs.W == W && fs.cap == cap && fs.img) return VO_OK;
    vo_frameset_free(fs);
    fs.F = F;
    fs.H = H;
    fs.W = W;
    fs.cap = cap;
    const int nbands = (H + 15) / 16;
    VO_TRY(dalloc(fs.img, (size_t)F * H * W + 64));
    VO_TRY(dalloc(fs.sum, (size_t)F * (H + 1) * (W + 1) + 64));
    VO_TRY(dalloc(fs.colsum, (size_t)F * nbands * W));
    VO_TRY(dalloc(fs.kp_raw, (size_t)F * cap));
    VO_TRY(dalloc(fs.kp, (size_t)F * cap));
    VO_TRY(dalloc(fs.rank, (size_t)F * cap));
    VO_TRY(dalloc(fs.desc, (size_t)F * cap * 64));
    VO_TRY(dalloc(fs.descT, (size_t)F * cap * 64));
    VO_TRY(dalloc(fs.nrm, (size_t)F * cap));
    VO_TRY(dalloc(fs.cnt, (size_t)F));
    VO_HIP(hipMemset(fs.rank, 0, (size_t)F * cap * sizeof(int)));
    VO_HIP(hipMemset(fs.cnt, 0, (size_t)F * sizeof(FrameCounters)));
    return VO_OK;
}

void vo_frameset_free(FrameSet& fs) {
    dfree(fs.img);
    dfree(fs.sum);
    dfree(fs.colsum);
    dfree(fs.kp_raw);
    dfree(fs.kp);
    dfree(fs.rank);
    dfree(fs.desc);
    dfree(fs.descT);
    dfree(fs.nrm);
    dfree(fs.cnt);
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
    if (p.n_octaves != 4 || p.n_octave_layers != 3 || p.extended || p.upright) return VO_ERR_UNSUPPORTED;
    if (p.kp_capacity < 64 || p.match_capacity < 8 || p.ransac_max_iters < 1 || p.ransac_max_iters > kMaxHyp)
        return VO_ERR_INVALID_ARG;
    p.kp_capacity = (p.kp_capacity + 63) & ~63;
    VO_HIP(hipSetDevice(hip_device));
    vo_ctx* ctx = new vo_ctx();
    ctx->prm = p;
    ctx->device = hip_device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return VO_ERR_HIP;
    }
    DetectTables t;
    vo_build_detect_tables(t, 0);
    int st = vo_upload_detect_tables(t);
    if (st == VO_OK) st = vo_geometry_init(ctx);
    if (st != VO_OK) {
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
    hipStreamSynchronize(ctx->stream);
    resolve_timers(ctx);
    vo_frameset_free(ctx->fs);
    vo_pairset_free(ctx->ps);
    if (ctx->d_tmp) hipFree(ctx->d_tmp);
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

static int upload_frame(vo_ctx* ctx, FrameSet& fs, int frame, const uint8_t* img, size_t step) {
    VO_HIP(hipMemcpy2DAsync(fs.img + (size_t)frame * fs.img_stride(), fs.W, img, step, fs.W, fs.H, hipMemcpyHostToDevice,
                            ctx->stream));
    return VO_OK;
}

static int reset_counters(vo_ctx* ctx, FrameSet& fs) {
    VO_HIP(hipMemsetAsync(fs.cnt, 0, (size_t)fs.F * sizeof(FrameCounters), ctx->stream));
    return VO_OK;
}

// ---------------------------------------------------------------- a1 / a2 inspection entry points
extern "C" int vo_integral(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, int32_t* sum) {
    if (!ctx || !sum) return VO_ERR_INVALID_ARG;
    VO_TRY(check_image(img, rows, cols, step));
    VO_HIP(hipSetDevice(ctx->device));
    VO_TRY(vo_frameset_alloc(ctx->fs, 1, rows, cols, ctx->prm.kp_capacity));
    VO_TRY(upload_frame(ctx, ctx->fs, 0, img, step));
    VO_TRY(vo_enqueue_integral(ctx, ctx->fs));
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
    VO_TRY(vo_frameset_alloc(ctx->fs, 1, rows, cols, ctx->prm.kp_capacity));
    VO_TRY(upload_frame(ctx, ctx->fs, 0, img, step));
    VO_TRY(vo_enqueue_integral(ctx, ctx->fs));
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

// ---------------------------------------------------------------- a1-a6
extern "C" int vo_detect_describe(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, vo_keypoint* kps,
                                  float* desc64, int cap, int* n) {
    if (!ctx || !kps || !n || cap < 0) return VO_ERR_INVALID_ARG;
    *n = 0;
    VO_TRY(check_image(img, rows, cols, step));
    VO_HIP(hipSetDevice(ctx->device));
    FrameSet& fs = ctx->fs;
    VO_TRY(vo_frameset_alloc(fs, 1, rows, cols, ctx->prm.kp_capacity));
    VO_TRY(reset_counters(ctx, fs));
    VO_TRY(upload_frame(ctx, fs, 0, img, step));
    VO_TRY(vo_enqueue_detect(ctx, fs, desc64 != nullptr));
    FrameCounters c;
    VO_HIP(hipMemcpyAsync(&c, fs.cnt, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    *n = c.n_raw;
    if (c.n_raw > fs.cap || c.n_raw > cap) return VO_ERR_CAPACITY;
    if (c.n_raw > 0) {
        VO_HIP(hipMemcpyAsync(kps, fs.kp, (size_t)c.n_raw * sizeof(vo_keypoint), hipMemcpyDeviceToHost, ctx->stream));
        if (desc64)
            VO_HIP(hipMemcpyAsync(desc64, fs.desc, (size_t)c.n_raw * 64 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return VO_OK;
}
