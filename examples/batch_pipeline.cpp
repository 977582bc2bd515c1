// The pipelined batch loop of INTEGRATION.md section 3 as a complete C++ host program on the C ABI (no OpenCV, no torch):
// frames of step k + 1 are staged while step k runs, step k - 1's records are read with age 1, and every step is checked
// against a synchronous run of the same inputs on a second batch.
// Build: g++ -std=c++17 -I include examples/batch_pipeline.cpp -L visual_odometry_amd -lvo_mi355x -o batch_pipeline
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vo_mi355x.h"

namespace {
constexpr int H = 480, W = 640, P = 4, STEPS = 5;
const double K[9] = {407.0646129842357, 0, 384.5, 0, 407.0646129842357, 246.5, 0, 0, 1};  // config/camera_intrinsic.yaml

struct Pair {
    std::vector<uint8_t> a, b;
};

// textured spots at different depths, second view shifted along x by a per-spot disparity
Pair make_pair(unsigned seed) {
    Pair pr{std::vector<uint8_t>((size_t)H * W, 128), std::vector<uint8_t>((size_t)H * W, 128)};
    unsigned s = seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffff; };
    for (int k = 0; k < 260; k++) {
        const int cx = 40 + rnd() % (W - 100), cy = 40 + rnd() % (H - 80), disp = 10 + rnd() % 31;
        for (int part = 0; part < 4; part++) {
            const int ox = part ? (int)(rnd() % 15) - 7 : 0, oy = part ? (int)(rnd() % 15) - 7 : 0;
            const int amp = (part ? 20 : 40) + rnd() % 60;
            const double sg = (part ? 1.0 : 1.5) + (rnd() % 250) / 100.0;
            for (int y = -12; y <= 12; y++)
                for (int x = -12; x <= 12; x++) {
                    const int v = (int)(amp * std::exp(-(x * x + y * y) / (2 * sg * sg)));
                    auto put = [&](std::vector<uint8_t>& im, int px, int py) {
                        if (px >= 0 && px < W && py >= 0 && py < H)
                            im[(size_t)py * W + px] = (uint8_t)std::min(255, im[(size_t)py * W + px] + v);
                    };
                    put(pr.a, cx + ox + x, cy + oy + y);
                    put(pr.b, cx + ox + x - disp, cy + oy + y);
                }
        }
    }
    return pr;
}

bool same(const vo_pair_out& x, const vo_pair_out& y) { return std::memcmp(&x, &y, sizeof(x)) == 0; }

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int st_ = (call);                                                                      \
        if (st_ != VO_OK) {                                                                          \
            std::fprintf(stderr, "%s -> %s (%s)\n", #call, vo_strerror(st_), vo_last_hip_error());   \
            return 2;                                                                                \
        }                                                                                            \
    } while (0)
}  // namespace

int main() {
    std::vector<Pair> pairs;
    for (int i = 0; i < STEPS + P; i++) pairs.push_back(make_pair(1000 + 17 * i));  // step k uses pairs k .. k + P - 1
    vo_ctx* ctx = nullptr;
    CHECK(vo_ctx_create(nullptr, 0, &ctx));
    vo_batch *b = nullptr, *ref = nullptr;
    CHECK(vo_batch_create(ctx, P, H, W, &b));
    CHECK(vo_batch_create(ctx, P, H, W, &ref));
    auto stage = [&](vo_batch* bb, int step) {
        for (int p = 0; p < P; p++) {
            const Pair& pr = pairs[step + p];
            const int st = vo_batch_set_pair(bb, p, pr.a.data(), W, pr.b.data(), W, K);
            if (st != VO_OK) return st;
        }
        return (int)VO_OK;
    };
    // what every step must give: the same inputs through a batch that is synchronised after every run
    std::vector<std::vector<vo_pair_out>> want(STEPS, std::vector<vo_pair_out>(P));
    for (int k = 0; k < STEPS; k++) {
        CHECK(stage(ref, k));
        CHECK(vo_batch_run(ref));
        CHECK(vo_batch_sync(ref));
        CHECK(vo_batch_get_run_all(ref, 0, want[k].data()));
    }
    // the pipelined loop
    std::vector<vo_pair_out> got(P);
    int bad = 0;
    CHECK(stage(b, 0));
    for (int k = 0; k < STEPS; k++) {
        CHECK(vo_batch_run(b));                    // asynchronous: one upload on the copy stream, then the kernels
        if (k + 1 < STEPS) CHECK(stage(b, k + 1));  // frames of step k + 1 while step k executes
        if (k > 0) {
            CHECK(vo_batch_get_run_all(b, 1, got.data()));  // records of step k - 1: waits for that step only
            for (int p = 0; p < P; p++) bad += !same(got[p], want[k - 1][p]);
        }
    }
    CHECK(vo_batch_get_run_all(b, 0, got.data()));
    for (int p = 0; p < P; p++) bad += !same(got[p], want[STEPS - 1][p]);
    int posed = 0;
    for (const auto& o : got) posed += o.success;
    std::printf("pipelined steps: %d, pairs per step: %d, records differing from the synchronous runs: %d, last step: %d matches, %d inliers, %d of %d pairs with a pose\n",
                STEPS, P, bad, got[0].n_match, got[0].n_inl, posed, P);
    vo_batch_destroy(b);
    vo_batch_destroy(ref);
    vo_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
