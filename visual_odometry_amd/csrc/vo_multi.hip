// vo_pair_batch: the multi-GPU entry point of the C ABI.  Host code only: one host thread per context,
// every thread drives the pipelined batch loop (vo_batch_*) of ITS context on its share of the pairs - pair p belongs to
// context p mod n_ctx - and writes its records straight into the caller's array ("host gather": the records are 264 bytes
// each, nothing crosses between GPUs).  Replaces the reference's single sequential loop over frame pairs,
// src/visual_Odometry.cpp:268-470, for offline batches.
#include <algorithm>
#include <thread>

#include "vo_common.hpp"

namespace {

// pairs per step of one context: what the benchmark uses at 1080p (64 pairs = ~6 GB of device buffers), scaled with the
// frame area so that small frames still fill the chip and 4K frames still fit
int pairs_per_step(int rows, int cols) {
    const double scale = (1920.0 * 1080.0) / ((double)rows * cols);
    return std::max(4, std::min(256, (int)(64.0 * scale)));
}

struct Share {
    vo_ctx* ctx;
    int first, stride;  // pairs first, first + stride, ...
    const vo_pair_in* pairs;
    int n_pairs;
    vo_pair_out* out;
    int status = VO_OK;
    std::string err;
};

int ensure_batch(vo_ctx* ctx, int n, int rows, int cols) {
    if (ctx->pb && ctx->pb->n_pairs == n && ctx->pb->rows == rows && ctx->pb->cols == cols) return VO_OK;
    if (ctx->pb) vo_batch_destroy(ctx->pb);
    ctx->pb = nullptr;
    return vo_batch_create(ctx, n, rows, cols, &ctx->pb);
}

int run_share(Share& sh) {
    const int mine = sh.first < sh.n_pairs ? (sh.n_pairs - sh.first + sh.stride - 1) / sh.stride : 0;
    if (mine == 0) return VO_OK;
    const int rows = sh.pairs[0].rows, cols = sh.pairs[0].cols;
    const int cap = pairs_per_step(rows, cols);
    const int steps = (mine + cap - 1) / cap;
    const int per = (mine + steps - 1) / steps;  // equal steps: the last one is not a sliver
    VO_TRY(ensure_batch(sh.ctx, per, rows, cols));
    vo_batch* b = sh.ctx->pb;
    std::vector<vo_pair_out> recs((size_t)per);
    auto stage = [&](int step) {
        for (int i = 0; i < per && step * per + i < mine; i++) {
            const vo_pair_in& in = sh.pairs[sh.first + (size_t)(step * per + i) * sh.stride];
            VO_TRY(vo_batch_set_pair(b, i, in.img1, in.step1, in.img2, in.step2, in.K));
        }
        return (int)VO_OK;  // slots past the end of the last step keep the previous step's frames; their records are dropped
    };
    auto collect = [&](int step, int age) {
        VO_TRY(vo_batch_get_run_all(b, age, recs.data()));
        for (int i = 0; i < per && step * per + i < mine; i++) sh.out[sh.first + (size_t)(step * per + i) * sh.stride] = recs[i];
        return (int)VO_OK;
    };
    VO_TRY(stage(0));
    for (int k = 0; k < steps; k++) {
        VO_TRY(vo_batch_run(b));                    // asynchronous
        if (k + 1 < steps) VO_TRY(stage(k + 1));    // the next step's frames go up while this one runs
        if (k > 0) VO_TRY(collect(k - 1, 1));       // waits for step k - 1 only
    }
    return collect(steps - 1, 0);
}

}  // namespace

extern "C" int vo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int vo_pair_batch(vo_ctx* const* ctx_per_gpu, int n_ctx, const vo_pair_in* pairs, int n_pairs, vo_pair_out* out) {
    if (!ctx_per_gpu || n_ctx <= 0 || n_pairs < 0 || (n_pairs > 0 && (!pairs || !out))) return VO_ERR_INVALID_ARG;
    for (int g = 0; g < n_ctx; g++) {
        if (!ctx_per_gpu[g]) return VO_ERR_INVALID_ARG;
        for (int h = 0; h < g; h++)
            if (ctx_per_gpu[h] == ctx_per_gpu[g]) return VO_ERR_INVALID_ARG;  // a context serves one host thread
    }
    if (n_pairs == 0) return VO_OK;
    for (int p = 0; p < n_pairs; p++) {
        const vo_pair_in& in = pairs[p];
        if (!in.img1 || !in.img2 || in.rows != pairs[0].rows || in.cols != pairs[0].cols) {
            vo_set_error_text("vo_pair_batch: every pair needs two frames, all of one size");
            return VO_ERR_INVALID_ARG;
        }
    }
    std::vector<Share> shares((size_t)n_ctx);
    for (int g = 0; g < n_ctx; g++) {
        shares[g].ctx = ctx_per_gpu[g];
        shares[g].first = g;
        shares[g].stride = n_ctx;
        shares[g].pairs = pairs;
        shares[g].n_pairs = n_pairs;
        shares[g].out = out;
    }
    auto body = [](Share* sh) {
        sh->status = run_share(*sh);
        if (sh->status != VO_OK) sh->err = vo_last_hip_error();  // the error text is per thread: hand it to the caller's
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < n_ctx; g++) threads.emplace_back(body, &shares[g]);
    body(&shares[0]);  // the calling thread drives the first context
    for (auto& t : threads) t.join();
    for (const Share& sh : shares)
        if (sh.status != VO_OK) {
            vo_set_error_text(sh.err.c_str());
            return sh.status;
        }
    return VO_OK;
}
