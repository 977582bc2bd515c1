// Minimal C++ host program on vo_frontend.hpp (no OpenCV): two synthetic frames -> matches -> pose -> points.
// Build: g++ -std=c++17 -I include examples/frontend_smoke.cpp -L visual_odometry_amd -lvo_mi355x -o frontend_smoke
#include <cmath>
#include <cstdio>
#include <vector>

#include "vo_frontend.hpp"

int main() {
    const int H = 480, W = 640;
    std::vector<uint8_t> a((size_t)H * W, 128), b((size_t)H * W, 128);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffff; };
    for (int k = 0; k < 220; k++) {  // textured features at different depths: camera moves along x, disparity 10..40 px
        const int cx = 40 + rnd() % (W - 100), cy = 40 + rnd() % (H - 80);
        const int disp = 10 + rnd() % 31;
        for (int part = 0; part < 4; part++) {  // a spot and three random satellites give every feature an orientation
            const int ox = part ? (int)(rnd() % 15) - 7 : 0, oy = part ? (int)(rnd() % 15) - 7 : 0;
            const int amp = (part ? 20 : 40) + rnd() % 60;
            const double sg = (part ? 1.0 : 1.5) + (rnd() % 250) / 100.0;
            for (int y = -12; y <= 12; y++)
                for (int x = -12; x <= 12; x++) {
                    const int v = (int)(amp * std::exp(-(x * x + y * y) / (2 * sg * sg)));
                    auto put = [&](std::vector<uint8_t>& im, int px, int py) {
                        if (px >= 0 && px < W && py >= 0 && py < H) im[(size_t)py * W + px] = (uint8_t)std::min(255, im[(size_t)py * W + px] + v);
                    };
                    put(a, cx + ox + x, cy + oy + y);
                    put(b, cx + ox + x - disp, cy + oy + y);
                }
        }
    }
    try {
        vo::Frontend fe;
        vo::KpAsPoint2f_Match kP, inl;
        fe.opt_DetectFeatures(vo::Image{a.data(), H, W, (size_t)W}, vo::Image{b.data(), H, W, (size_t)W}, kP);
        std::printf("matches: %zu\n", kP.match.size());
        vo::Mat33 K = {407.0646129842357, 0, 384.5, 0, 407.0646129842357, 246.5, 0, 0, 1}, R = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        vo::Vec3 t = {0, 0, 0};
        bool ok = false;
        if (!vo::Frontend::checkMinFeat(kP) && vo::Frontend::checkIfMoving(kP)) fe.optRelativePose(kP, K, R, t, inl, ok);
        std::printf("inliers: %zu success: %d t = (%.3f %.3f %.3f)\n", inl.Kpoints1.size(), (int)ok, t[0], t[1], t[2]);
        if (ok) {
            int m = 0;
            const std::vector<double> wp = fe.triangPoints(inl.Kpoints1, inl.Kpoints2, R, t, K, &m);
            std::printf("world points: %d\n", m);
            if (m > 0) {
                const double SF = vo::Frontend::scaleFactor(7.5f, wp);
                const vo::Mat33 I3 = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                const auto ap = vo::Frontend::absolutePose(I3, vo::Vec3{0, 0, 0}, R, t, SF, &wp);
                std::printf("scale factor %.3f, camera at (%.3f %.3f %.3f)\n", SF, ap.scaled_locW[0], ap.scaled_locW[1], ap.scaled_locW[2]);
            }
        }
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
