// vo_cv_adaptor.hpp -- the three reference signatures on top of libvo_mi355x.so, for a tree that has OpenCV.
// Include this INSTEAD of the function bodies of include/visual_odometry/camera_utility.hpp:742-809, 633-685,
// 288-351; src/visual_Odometry.cpp:296, 336, 384 then compile unchanged.  Not compiled in this repository
// (OpenCV is not installed here); see INTEGRATION.md.
#pragma once
#include <opencv2/core.hpp>

#include "vo_frontend.hpp"

namespace vo_cv {

inline vo::Frontend& frontend() {
    static vo::Frontend f;  // one context per process: the node is single-threaded (visual_Odometry.cpp:179-480)
    return f;
}

struct KpAsPoint2f_Match {  // camera_utility.hpp:73-78
    std::vector<cv::Point2f> Kpoints1, Kpoints2;
    std::vector<cv::DMatch> match;
};

inline vo::Image view(const cv::Mat& m) {
    CV_Assert(m.type() == CV_8UC1);
    return vo::Image{m.data, m.rows, m.cols, (size_t)m.step};
}

inline void opt_DetectFeatures(cv::Mat img1, cv::Mat img2, KpAsPoint2f_Match& out) {
    vo::KpAsPoint2f_Match k;
    frontend().opt_DetectFeatures(view(img1), view(img2), k);
    out.Kpoints1.assign(reinterpret_cast<const cv::Point2f*>(k.Kpoints1.data()),
                        reinterpret_cast<const cv::Point2f*>(k.Kpoints1.data()) + k.Kpoints1.size());
    out.Kpoints2.assign(reinterpret_cast<const cv::Point2f*>(k.Kpoints2.data()),
                        reinterpret_cast<const cv::Point2f*>(k.Kpoints2.data()) + k.Kpoints2.size());
    out.match.clear();
    for (const vo_dmatch& m : k.match) out.match.emplace_back(m.queryIdx, m.trainIdx, m.imgIdx, m.distance);
}

inline void optRelativePose(KpAsPoint2f_Match kP, cv::Mat K, cv::Mat& R, cv::Mat& t, KpAsPoint2f_Match& inlier, bool& success) {
    vo::KpAsPoint2f_Match in, out;
    in.Kpoints1.assign(reinterpret_cast<const vo::Point2f*>(kP.Kpoints1.data()),
                       reinterpret_cast<const vo::Point2f*>(kP.Kpoints1.data()) + kP.Kpoints1.size());
    in.Kpoints2.assign(reinterpret_cast<const vo::Point2f*>(kP.Kpoints2.data()),
                       reinterpret_cast<const vo::Point2f*>(kP.Kpoints2.data()) + kP.Kpoints2.size());
    vo::Mat33 Km, Rm;
    vo::Vec3 tv;
    cv::Mat Kd, Rd, td;
    K.convertTo(Kd, CV_64F);
    R.convertTo(Rd, CV_64F);
    t.convertTo(td, CV_64F);
    for (int i = 0; i < 9; i++) {
        Km[i] = Kd.at<double>(i / 3, i % 3);
        Rm[i] = Rd.at<double>(i / 3, i % 3);
    }
    for (int i = 0; i < 3; i++) tv[i] = td.at<double>(i);
    frontend().optRelativePose(in, Km, Rm, tv, out, success);
    inlier.Kpoints1.assign(reinterpret_cast<const cv::Point2f*>(out.Kpoints1.data()),
                           reinterpret_cast<const cv::Point2f*>(out.Kpoints1.data()) + out.Kpoints1.size());
    inlier.Kpoints2.assign(reinterpret_cast<const cv::Point2f*>(out.Kpoints2.data()),
                           reinterpret_cast<const cv::Point2f*>(out.Kpoints2.data()) + out.Kpoints2.size());
    inlier.match.clear();
    for (const vo_dmatch& m : out.match) inlier.match.emplace_back(m.queryIdx, m.trainIdx, m.imgIdx, m.distance);
    if (success) {  // R, t are in/out: untouched unless success (camera_utility.hpp:673-679)
        R = cv::Mat(3, 3, CV_64F, Rm.data()).clone();
        t = cv::Mat(3, 1, CV_64F, tv.data()).clone();
    }
}

inline cv::Mat triangPoints(std::vector<cv::Point2f> inl1, std::vector<cv::Point2f> inl2, cv::Mat R, cv::Mat t, cv::Mat K) {
    std::vector<vo::Point2f> a(reinterpret_cast<const vo::Point2f*>(inl1.data()), reinterpret_cast<const vo::Point2f*>(inl1.data()) + inl1.size());
    std::vector<vo::Point2f> b(reinterpret_cast<const vo::Point2f*>(inl2.data()), reinterpret_cast<const vo::Point2f*>(inl2.data()) + inl2.size());
    vo::Mat33 Km, Rm;
    vo::Vec3 tv;
    for (int i = 0; i < 9; i++) {
        Km[i] = K.at<double>(i / 3, i % 3);
        Rm[i] = R.at<double>(i / 3, i % 3);
    }
    for (int i = 0; i < 3; i++) tv[i] = t.at<double>(i);
    int m = 0;
    std::vector<double> X = frontend().triangPoints(a, b, Rm, tv, Km, &m);
    if (m == 0) return cv::Mat();  // caller checks .empty() (visual_Odometry.cpp:386)
    return cv::Mat(3, m, CV_64F, X.data()).clone();
}

}  // namespace vo_cv
