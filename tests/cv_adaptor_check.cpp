// Compiles include/vo_cv_adaptor.hpp and integration/ros/vo_mi355x_switch_*.hpp the way the node would use them, against
// tests/mock_cv (syntax + link check; run only on a GPU box).  The stand-in for camera_utility.hpp declares the node's own
// match structure and four functions with the reference's signatures, which the switch headers must push aside.
#define VO_USE_MI355X 1
#include <opencv2/core.hpp>

#include <cstdio>
#include <vector>
using namespace cv;
using namespace std;

#include "vo_mi355x_switch_begin.hpp"
// ---- what camera_utility.hpp declares (signatures only: :73-78, :147, :742, :633, :288)
struct KpAsPoint2f_Match {
    vector<Point2f> Kpoints1;
    vector<Point2f> Kpoints2;
    vector<DMatch> match;
};
Mat get_image(Mat, int, int, Mat, Mat) { return Mat(); }
void opt_DetectFeatures(Mat, Mat, KpAsPoint2f_Match&) {}
void optRelativePose(KpAsPoint2f_Match, Mat, Mat&, Mat&, KpAsPoint2f_Match&, bool&) {}
Mat triangPoints(vector<Point2f>, vector<Point2f>, Mat, Mat, Mat) { return Mat(); }
// ----
#include "vo_mi355x_switch_end.hpp"

int main(int argc, char**) {
    if (argc < 2) return 0;  // built and linked: enough without a GPU
    // the call sequence of src/visual_Odometry.cpp:228-384 on a synthetic pair
    Mat K(3, 3, CV_64F), dist(1, 4, CV_64F), R(3, 3, CV_64F), t(3, 1, CV_64F);
    K.at<double>(0, 0) = K.at<double>(1, 1) = 400, K.at<double>(0, 2) = 320, K.at<double>(1, 2) = 240, K.at<double>(2, 2) = 1;
    for (int i = 0; i < 3; i++) R.at<double>(i, i) = 1;
    Mat a(480, 640, CV_8UC3), b(480, 640, CV_8UC3);
    unsigned s = 1;
    for (int i = 0; i < 900; i++) {  // bright squares, shifted 30 px between the frames
        s = s * 1664525u + 1013904223u;
        const int x = 40 + (s >> 8) % 520, y = 20 + (s >> 20) % 430, w = 3 + (s >> 4) % 6, v = 120 + (s >> 12) % 130;
        for (int dy = 0; dy < w; dy++)
            for (int dx = 0; dx < w; dx++)
                for (int c = 0; c < 3; c++) a.data[(y + dy) * a.step + (x + dx) * 3 + c] = b.data[(y + dy) * b.step + (x + dx + 30) * 3 + c] = (uint8_t)v;
    }
    Mat prev_img = get_image(a, 640, 480, K, dist), curr_img = get_image(b, 640, 480, K, dist);
    KpAsPoint2f_Match kP_converted, inlier_converted;
    opt_DetectFeatures(prev_img, curr_img, kP_converted);
    bool success = false;
    if (kP_converted.Kpoints1.size() >= 20) optRelativePose(kP_converted, K, R, t, inlier_converted, success);
    Mat wp;
    if (success) wp = triangPoints(inlier_converted.Kpoints1, inlier_converted.Kpoints2, R, t, K);
    printf("matches %zu inliers %zu success %d world points %d\n", kP_converted.Kpoints1.size(), inlier_converted.Kpoints1.size(), (int)success, wp.cols);
    return kP_converted.Kpoints1.size() > 50 ? 0 : 1;
}
