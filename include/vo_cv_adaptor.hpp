// vo_cv_adaptor.hpp -- the reference's hot-path signatures on top of libvo_mi355x.so, for a tree that has OpenCV.
//
// Replaces, with identical signatures (reference include/visual_odometry/camera_utility.hpp):
//     Mat  get_image(Mat current_img, int desired_width, int desired_height, Mat cameraMatrix, Mat distortionCoeff)   :147
//     void opt_DetectFeatures(Mat img1, Mat img2, KpAsPoint2f_Match& kP_converted)                                     :742
//     void optRelativePose(KpAsPoint2f_Match kP, Mat K, Mat& R, Mat& t, KpAsPoint2f_Match& inlier, bool& success)       :633
//     Mat  triangPoints(vector<Point2f> inlier1, vector<Point2f> inlier2, Mat R, Mat t, Mat cameraMatrix)              :288
// so that src/visual_Odometry.cpp:228, 290, 296, 336, 384 compile unchanged.  The match structure is a template
// parameter: the node's own `KpAsPoint2f_Match` (:73-78, members Kpoints1, Kpoints2, match) is used as it is.
// integration/ros/ shows the three-line change that switches the node over; tests/mock_cv holds the minimal cv:: types
// this header is syntax-checked against in this repository (OpenCV is not installed here).
#pragma once
#include <opencv2/core.hpp>

#include <cstdlib>
#include <memory>

#include "vo_frontend.hpp"

namespace vo_cv {

// Where the algorithm parameters come from.  The reference keeps them in mutable header globals (camera_utility.hpp:33-63:
// minHessian, ratio_thresh, width_low ... height_high, ransac_prob[], ransac_threshold[], inlier_threshold[], VPF_threshold,
// distance_threshold, reprojection_tolerance); integration/ros/vo_mi355x_switch_end.hpp, which is included after that header
// and therefore sees them, registers a provider that reads them.  Without a provider the library's defaults (the same
// values) apply.
using ParamsProvider = vo_params (*)();
inline ParamsProvider& params_provider() {
    static ParamsProvider p = nullptr;
    return p;
}
inline void set_params_provider(ParamsProvider p) { params_provider() = p; }

inline int hip_device_from_env() {  // VO_HIP_DEVICE picks the GPU of the node's context (default 0)
    const char* e = std::getenv("VO_HIP_DEVICE");
    return e && *e ? std::atoi(e) : 0;
}

// One context per process: the node is single-threaded (visual_Odometry.cpp:179-480).  The provider is asked on every call
// (a few loads); when a global has changed since the context was made - the reference's are plain non-const ints - the
// context is rebuilt with the new values, so an edited ROI or threshold takes effect exactly as it would in the reference.
inline vo::Frontend& frontend() {
    static std::unique_ptr<vo::Frontend> f;
    static vo_params live;
    vo_params want;
    if (params_provider())
        want = params_provider()();
    else
        vo_params_default(&want);
    if (!f || std::memcmp(&want, &live, sizeof(want)) != 0) {
        f.reset();  // the old context goes first: one context per GPU at a time
        f.reset(new vo::Frontend(&want, hip_device_from_env()));
        live = want;
    }
    return *f;
}

inline vo::Image view(const cv::Mat& m) {
    CV_Assert(m.type() == CV_8UC1);
    return vo::Image{m.data, m.rows, m.cols, (size_t)m.step};
}

inline vo::Mat33 mat33(const cv::Mat& m) {
    cv::Mat d;
    m.convertTo(d, CV_64F);
    CV_Assert(d.rows == 3 && d.cols == 3);
    vo::Mat33 o;
    for (int i = 0; i < 9; i++) o[i] = d.at<double>(i / 3, i % 3);
    return o;
}

template <class P>
inline std::vector<vo::Point2f> points(const std::vector<P>& v) {
    std::vector<vo::Point2f> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = vo::Point2f{v[i].x, v[i].y};
    return o;
}

template <class KP>
inline void store(const vo::KpAsPoint2f_Match& k, KP& out) {
    out.Kpoints1.clear();
    out.Kpoints2.clear();
    out.match.clear();
    for (const vo::Point2f& p : k.Kpoints1) out.Kpoints1.emplace_back(p.x, p.y);
    for (const vo::Point2f& p : k.Kpoints2) out.Kpoints2.emplace_back(p.x, p.y);
    for (const vo_dmatch& m : k.match) out.match.emplace_back(m.queryIdx, m.trainIdx, m.imgIdx, m.distance);
}

inline cv::Mat get_image(cv::Mat current_img, int desired_width, int desired_height, cv::Mat cameraMatrix, cv::Mat distortionCoeff) {
    CV_Assert(current_img.type() == CV_8UC3);
    cv::Mat d;
    distortionCoeff.convertTo(d, CV_64F);
    std::array<double, 4> dist{0, 0, 0, 0};
    for (int i = 0; i < 4 && i < (int)d.total(); i++) dist[i] = d.at<double>(i);
    const std::vector<uint8_t> g = frontend().get_image(current_img.data, current_img.rows, current_img.cols, (size_t)current_img.step,
                                                        desired_width, desired_height, mat33(cameraMatrix), dist);
    return cv::Mat(desired_height, desired_width, CV_8UC1, const_cast<uint8_t*>(g.data())).clone();
}

template <class KP>
inline void opt_DetectFeatures(cv::Mat img1, cv::Mat img2, KP& kP_converted) {
    vo::KpAsPoint2f_Match k;
    frontend().opt_DetectFeatures(view(img1), view(img2), k);
    store(k, kP_converted);
}

template <class KP>
inline void optRelativePose(KP kP_converted, cv::Mat cameraMatrix, cv::Mat& R, cv::Mat& t, KP& inlier_converted, bool& success) {
    vo::KpAsPoint2f_Match in, out;
    in.Kpoints1 = points(kP_converted.Kpoints1);
    in.Kpoints2 = points(kP_converted.Kpoints2);
    vo::Mat33 Rm = mat33(R);
    vo::Vec3 tv;
    cv::Mat td;
    t.convertTo(td, CV_64F);
    for (int i = 0; i < 3; i++) tv[i] = td.at<double>(i);
    frontend().optRelativePose(in, mat33(cameraMatrix), Rm, tv, out, success);
    store(out, inlier_converted);
    if (success) {  // R, t are in/out: untouched unless success (camera_utility.hpp:673-679)
        R = cv::Mat(3, 3, CV_64F, Rm.data()).clone();
        t = cv::Mat(3, 1, CV_64F, tv.data()).clone();
    }
}

template <class P>
inline cv::Mat triangPoints(std::vector<P> inlier1, std::vector<P> inlier2, cv::Mat R, cv::Mat t, cv::Mat cameraMatrix) {
    vo::Vec3 tv;
    cv::Mat td;
    t.convertTo(td, CV_64F);
    for (int i = 0; i < 3; i++) tv[i] = td.at<double>(i);
    int m = 0;
    std::vector<double> X = frontend().triangPoints(points(inlier1), points(inlier2), mat33(R), tv, mat33(cameraMatrix), &m);
    if (m == 0) return cv::Mat();  // the caller checks .empty() (visual_Odometry.cpp:386)
    return cv::Mat(3, m, CV_64F, X.data()).clone();
}

}  // namespace vo_cv
