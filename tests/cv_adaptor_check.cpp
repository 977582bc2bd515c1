// Compiles include/vo_cv_adaptor.hpp and integration/ros/vo_mi355x_switch_*.hpp the way the node would use them, against
// tests/mock_cv (syntax + link check; run only on a GPU box).  The stand-in for camera_utility.hpp declares the node's own
// match structure and four functions with the reference's signatures, which the switch headers must push aside.
#define VO_USE_MI355X 1
#include <opencv2/core.hpp>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
using namespace cv;
using namespace std;

#include "vo_mi355x_switch_begin.hpp"
// ---- what camera_utility.hpp declares: its parameter globals (names and types of :21, :33-63; the switch-end header reads
// them) and the signatures :73-78, :147, :742, :633, :288
enum rel_pose_method { ESSENTIAL, HOMOGRAPHY };
int minHessian = 50, nOctaves = 4, nOctaveLayers = 3;
bool extended = false, upright = false;
const float ratio_thresh = 0.7f;
int width_low = 28, height_low = 20, width_high = 614, height_high = 482;
double ransac_prob[] = {0.99, 0.99}, ransac_threshold[] = {2.0, 5.0};
const float inlier_threshold[] = {0.3, 0.3};
const float VPF_threshold = 0.85;
const double distance_threshold = 50.0;
const float reprojection_tolerance = 0.5;
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

// `files a.bgr b.bgr rows cols dst_cols dst_rows fx fy cx cy k1 k2 p1 p2 out.txt`: the same call sequence on two raw BGR8 frames,
// every output written with %a (exact) so that the test can compare it with the CPU oracle
static int run_files(char** argv) {
    const int rows = atoi(argv[4]), cols = atoi(argv[5]), dcols = atoi(argv[6]), drows = atoi(argv[7]);
    Mat K(3, 3, CV_64F), dist(1, 4, CV_64F), R(3, 3, CV_64F), t(3, 1, CV_64F);
    K.at<double>(0, 0) = atof(argv[8]), K.at<double>(1, 1) = atof(argv[9]), K.at<double>(0, 2) = atof(argv[10]), K.at<double>(1, 2) = atof(argv[11]);
    K.at<double>(2, 2) = 1;
    for (int i = 0; i < 4; i++) dist.at<double>(0, i) = atof(argv[12 + i]);
    for (int i = 0; i < 3; i++) R.at<double>(i, i) = 1;
    Mat img[2] = {Mat(rows, cols, CV_8UC3), Mat(rows, cols, CV_8UC3)};
    for (int f = 0; f < 2; f++) {
        FILE* fp = fopen(argv[2 + f], "rb");
        if (!fp || fread(img[f].data, 1, (size_t)rows * cols * 3, fp) != (size_t)rows * cols * 3) return 2;
        fclose(fp);
    }
    Mat prev_img = get_image(img[0], dcols, drows, K, dist), curr_img = get_image(img[1], dcols, drows, K, dist);
    KpAsPoint2f_Match kP, inl;
    opt_DetectFeatures(prev_img, curr_img, kP);
    bool success = false;
    if (kP.Kpoints1.size() >= 20) optRelativePose(kP, K, R, t, inl, success);
    Mat wp;
    if (success) wp = triangPoints(inl.Kpoints1, inl.Kpoints2, R, t, K);
    FILE* fo = fopen(argv[16], "w");
    if (!fo) return 3;
    fprintf(fo, "gray %d %d\n", prev_img.rows, prev_img.cols);
    unsigned long long h0 = 1469598103934665603ull, h1 = h0;  // FNV-1a of the two preprocessed frames
    for (int r = 0; r < prev_img.rows; r++)
        for (int c = 0; c < prev_img.cols; c++) {
            h0 = (h0 ^ prev_img.data[r * prev_img.step + c]) * 1099511628211ull;
            h1 = (h1 ^ curr_img.data[r * curr_img.step + c]) * 1099511628211ull;
        }
    fprintf(fo, "hash %llu %llu\n", h0, h1);
    fprintf(fo, "matches %zu\n", kP.Kpoints1.size());
    for (size_t i = 0; i < kP.Kpoints1.size(); i++)
        fprintf(fo, "%a %a %a %a %d %d %a\n", kP.Kpoints1[i].x, kP.Kpoints1[i].y, kP.Kpoints2[i].x, kP.Kpoints2[i].y, kP.match[i].queryIdx,
                kP.match[i].trainIdx, kP.match[i].distance);
    fprintf(fo, "success %d inliers %zu\n", (int)success, inl.Kpoints1.size());
    for (size_t i = 0; i < inl.Kpoints1.size(); i++)
        fprintf(fo, "%a %a %a %a\n", inl.Kpoints1[i].x, inl.Kpoints1[i].y, inl.Kpoints2[i].x, inl.Kpoints2[i].y);
    fprintf(fo, "R");
    for (int i = 0; i < 9; i++) fprintf(fo, " %a", R.at<double>(i / 3, i % 3));
    fprintf(fo, "\nt %a %a %a\n", t.at<double>(0, 0), t.at<double>(1, 0), t.at<double>(2, 0));
    fprintf(fo, "world %d\n", wp.cols);
    for (int j = 0; j < wp.cols; j++) fprintf(fo, "%a %a %a\n", wp.at<double>(0, j), wp.at<double>(1, j), wp.at<double>(2, j));
    fclose(fo);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 0;  // built and linked: enough without a GPU
    if (argc >= 17 && std::string(argv[1]) == "files") return run_files(argv);
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
    // the node's globals are the parameters: a narrower ROI (camera_utility.hpp:44-47 are plain ints) must narrow the matches,
    // a higher minHessian must thin the keypoints, and restoring them must restore the first result
    const size_t n_all = kP_converted.Kpoints1.size();
    size_t left_of_300 = 0;
    for (const Point2f& p : kP_converted.Kpoints1) left_of_300 += p.x < 300.f;
    width_low = 300;
    KpAsPoint2f_Match narrow;
    opt_DetectFeatures(prev_img, curr_img, narrow);
    bool roi_ok = narrow.Kpoints1.size() + left_of_300 == n_all && left_of_300 > 0;
    for (const Point2f& p : narrow.Kpoints1) roi_ok = roi_ok && p.x >= 300.f;
    width_low = 28;
    minHessian = 5000;
    KpAsPoint2f_Match sparse;
    opt_DetectFeatures(prev_img, curr_img, sparse);
    const bool hess_ok = sparse.Kpoints1.size() < n_all;
    minHessian = 50;
    // the SURF constructor arguments (:35-38) are globals too: another layer count and 128-float descriptors must reach the library
    // (different matches, not an error), and restoring them must restore the first result
    nOctaveLayers = 4;
    KpAsPoint2f_Match deeper;
    opt_DetectFeatures(prev_img, curr_img, deeper);
    nOctaveLayers = 3;
    extended = true;
    KpAsPoint2f_Match wide;
    opt_DetectFeatures(prev_img, curr_img, wide);
    extended = false;
    const bool surf_ok = deeper.Kpoints1.size() > 50 && deeper.Kpoints1.size() != n_all && wide.Kpoints1.size() > 20 && wide.Kpoints1.size() != n_all;
    KpAsPoint2f_Match again;
    opt_DetectFeatures(prev_img, curr_img, again);
    bool same = again.Kpoints1.size() == n_all;
    for (size_t i = 0; same && i < n_all; i++) same = again.Kpoints1[i].x == kP_converted.Kpoints1[i].x && again.Kpoints2[i].y == kP_converted.Kpoints2[i].y;
    printf("globals: roi %s (%zu of %zu matches right of x = 300) minHessian %s (%zu matches at 5000) nOctaveLayers / extended %s (%zu / %zu matches) restored %s\n",
           roi_ok ? "ok" : "BAD", narrow.Kpoints1.size(), n_all, hess_ok ? "ok" : "BAD", sparse.Kpoints1.size(), surf_ok ? "ok" : "BAD",
           deeper.Kpoints1.size(), wide.Kpoints1.size(), same ? "ok" : "BAD");
    return kP_converted.Kpoints1.size() > 50 && roi_ok && hess_ok && surf_ok && same ? 0 : 1;
}
