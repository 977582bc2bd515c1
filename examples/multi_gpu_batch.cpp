// vo_pair_batch on every GPU of the node from one C++ host program (no OpenCV, no torch, no Python): one context and one
// host thread per GPU, pair p on GPU p mod n, host gather - checked against the same pairs through ONE context.
// With a single GPU it runs two contexts on device 0 from two threads (per-thread error state, shared device symbols).
// Usage: multi_gpu_batch [n_pairs] [n_ctx]
// Build: g++ -std=c++17 -I include examples/multi_gpu_batch.cpp -L visual_odometry_amd -lvo_mi355x -pthread -o multi_gpu_batch
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vo_mi355x.h"

namespace {
constexpr int H = 480, W = 640;
const double K[9] = {407.0646129842357, 0, 384.5, 0, 407.0646129842357, 246.5, 0, 0, 1};  // config/camera_intrinsic.yaml

struct Frames {
    std::vector<uint8_t> a, b;
};

// Gaussian spots at different depths; the second view is shifted along x by a per-spot disparity
Frames make_frames(unsigned seed) {
    Frames f{std::vector<uint8_t>((size_t)H * W, 128), std::vector<uint8_t>((size_t)H * W, 128)};
    unsigned s = seed;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffff; };
    for (int k = 0; k < 240; k++) {
        const int cx = 40 + rnd() % (W - 100), cy = 40 + rnd() % (H - 80), disp = 10 + rnd() % 31;
        for (int part = 0; part < 3; part++) {
            const int ox = part ? (int)(rnd() % 15) - 7 : 0, oy = part ? (int)(rnd() % 15) - 7 : 0;
            const int amp = (part ? 20 : 40) + rnd() % 60;
            const double sg = (part ? 1.0 : 1.5) + (rnd() % 250) / 100.0;
            for (int y = -12; y <= 12; y++)
                for (int x = -12; x <= 12; x++) {
                    const int v = (int)(amp * std::exp(-(x * x + y * y) / (2 * sg * sg)));
                    auto put = [&](std::vector<uint8_t>& im, int px, int py) {
                        if (px >= 0 && px < W && py >= 0 && py < H) im[(size_t)py * W + px] = (uint8_t)std::min(255, im[(size_t)py * W + px] + v);
                    };
                    put(f.a, cx + ox + x, cy + oy + y);
                    put(f.b, cx + ox + x - disp, cy + oy + y);
                }
        }
    }
    return f;
}

#define CHECK(call)                                                                                \
    do {                                                                                           \
        const int st_ = (call);                                                                    \
        if (st_ != VO_OK) {                                                                        \
            std::fprintf(stderr, "%s -> %s (%s)\n", #call, vo_strerror(st_), vo_last_hip_error()); \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)
}  // namespace

int main(int argc, char** argv) {
    const int n_pairs = argc > 1 ? std::atoi(argv[1]) : 150;
    const int n_dev = vo_device_count();
    if (n_dev <= 0) {
        std::fprintf(stderr, "no HIP device\n");
        return 2;
    }
    const int n_ctx = argc > 2 ? std::atoi(argv[2]) : (n_dev > 1 ? n_dev : 2);
    std::vector<Frames> scenes;
    for (int i = 0; i < 12; i++) scenes.push_back(make_frames(2000 + 31 * i));
    std::vector<vo_pair_in> pairs((size_t)n_pairs);
    for (int p = 0; p < n_pairs; p++) {  // every pair distinct in content or direction
        const Frames& f = scenes[p % scenes.size()];
        const bool swap = (p / scenes.size()) & 1;
        pairs[p] = vo_pair_in{swap ? f.b.data() : f.a.data(), W, swap ? f.a.data() : f.b.data(), W, H, W, {}};
        std::memcpy(pairs[p].K, K, sizeof(K));
    }
    std::vector<vo_ctx*> ctx((size_t)n_ctx, nullptr);
    for (int g = 0; g < n_ctx; g++) CHECK(vo_ctx_create(nullptr, g % n_dev, &ctx[g]));
    std::vector<vo_pair_out> one((size_t)n_pairs), many((size_t)n_pairs);
    std::memset(one.data(), 0xee, one.size() * sizeof(vo_pair_out));
    std::memset(many.data(), 0xdd, many.size() * sizeof(vo_pair_out));
    using clk = std::chrono::steady_clock;
    CHECK(vo_pair_batch(ctx.data(), 1, pairs.data(), n_pairs, one.data()));  // also warms context 0 up
    const auto t0 = clk::now();
    CHECK(vo_pair_batch(ctx.data(), 1, pairs.data(), n_pairs, one.data()));
    const auto t1 = clk::now();
    CHECK(vo_pair_batch(ctx.data(), n_ctx, pairs.data(), n_pairs, many.data()));  // warm-up of the other contexts
    const auto t2 = clk::now();
    CHECK(vo_pair_batch(ctx.data(), n_ctx, pairs.data(), n_pairs, many.data()));
    const auto t3 = clk::now();
    int bad = 0, posed = 0;
    for (int p = 0; p < n_pairs; p++) {
        bad += std::memcmp(&one[p], &many[p], sizeof(vo_pair_out)) != 0;
        posed += many[p].success == 1;
    }
    // an argument error must come back as a status, from whichever thread hit it
    vo_pair_in broken = pairs[0];
    broken.cols = W - 1;
    std::vector<vo_pair_in> mixed = {pairs[0], broken};
    const int st_mixed = vo_pair_batch(ctx.data(), n_ctx, mixed.data(), 2, many.data());
    const double s1 = std::chrono::duration<double>(t1 - t0).count(), sn = std::chrono::duration<double>(t3 - t2).count();
    std::printf("pairs %d, contexts %d on %d device(s): 1 context %.1f pairs/s, %d contexts %.1f pairs/s, records differing: %d, "
                "pairs with a pose: %d, mixed sizes -> %s\n",
                n_pairs, n_ctx, n_dev, n_pairs / s1, n_ctx, n_pairs / sn, bad, posed, vo_strerror(st_mixed));
    for (vo_ctx* c : ctx) vo_ctx_destroy(c);
    return (bad || posed < n_pairs / 2 || st_mixed != VO_ERR_INVALID_ARG) ? 1 : 0;
}
