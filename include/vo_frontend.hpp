// vo_frontend.hpp -- C++ host mirror of the reference front end on top of the C ABI (vo_mi355x.h).
//
// The reference's boundary is three header-defined free functions with OpenCV value types
// (include/visual_odometry/camera_utility.hpp:742 opt_DetectFeatures, :633 optRelativePose, :288 triangPoints).
// This header re-creates them with plain std types (no OpenCV needed); vo_cv_adaptor.hpp maps cv::Mat /
// cv::Point2f / cv::DMatch onto these where OpenCV is installed.  Same names, same argument meaning, same
// in/out semantics (R, t are overwritten only on success); errors that would be cv::Exception in the reference
// are std::runtime_error here.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "vo_mi355x.h"

namespace vo {

struct Point2f {
    float x, y;
};
static_assert(sizeof(Point2f) == 8, "interleaved xy");

struct Image {  // CV_8UC1 view: cv::Mat::{data, rows, cols, step}
    const uint8_t* data;
    int rows, cols;
    size_t step;
};

struct KpAsPoint2f_Match {  // camera_utility.hpp:73-78
    std::vector<Point2f> Kpoints1, Kpoints2;
    std::vector<vo_dmatch> match;
};

using Mat33 = std::array<double, 9>;  // row-major, cv::Mat1d(3,3)
using Vec3 = std::array<double, 3>;

class Frontend {
   public:
    explicit Frontend(const vo_params* params = nullptr, int hip_device = 0) {
        const int st = vo_ctx_create(params, hip_device, &ctx_);
        if (st != VO_OK) throw std::runtime_error(std::string("vo_ctx_create: ") + vo_strerror(st) + " " + vo_last_hip_error());
        vo_ctx_get_params(ctx_, &prm_);
    }
    ~Frontend() {
        if (prep_) vo_prep_destroy(prep_);
        vo_ctx_destroy(ctx_);
    }
    Frontend(const Frontend&) = delete;
    Frontend& operator=(const Frontend&) = delete;

    // void opt_DetectFeatures(Mat img1, Mat img2, KpAsPoint2f_Match& kP_converted)   camera_utility.hpp:742
    void opt_DetectFeatures(const Image& img1, const Image& img2, KpAsPoint2f_Match& kP) {
        if (img1.rows != img2.rows || img1.cols != img2.cols) throw std::runtime_error("frame sizes differ");
        const int cap = prm_.match_capacity;
        kP.Kpoints1.resize(cap);
        kP.Kpoints2.resize(cap);
        kP.match.resize(cap);
        int n = 0;
        check(vo_detect_match(ctx_, img1.data, img1.step, img2.data, img2.step, img1.rows, img1.cols,
                              reinterpret_cast<float*>(kP.Kpoints1.data()), reinterpret_cast<float*>(kP.Kpoints2.data()),
                              kP.match.data(), cap, &n),
              "vo_detect_match");
        kP.Kpoints1.resize(n);
        kP.Kpoints2.resize(n);
        kP.match.resize(n);
    }

    // void optRelativePose(KpAsPoint2f_Match kP, Mat K, Mat& R, Mat& t, KpAsPoint2f_Match& inlier, bool& success)  :633
    void optRelativePose(const KpAsPoint2f_Match& kP, const Mat33& K, Mat33& R, Vec3& t, KpAsPoint2f_Match& inlier,
                         bool& success) {
        const int n = (int)kP.Kpoints1.size();
        inlier.Kpoints1.assign(n, Point2f{0, 0});
        inlier.Kpoints2.assign(n, Point2f{0, 0});
        int n_inl = 0, ok = 0;
        const int st = vo_relative_pose(ctx_, reinterpret_cast<const float*>(kP.Kpoints1.data()),
                                        reinterpret_cast<const float*>(kP.Kpoints2.data()), n, K.data(), R.data(), t.data(),
                                        reinterpret_cast<float*>(inlier.Kpoints1.data()),
                                        reinterpret_cast<float*>(inlier.Kpoints2.data()), &n_inl, &ok);
        // findEssentialMat with < 5 points / without a model hands an empty E to recoverPose, which throws
        if (st != VO_OK) check(st, "vo_relative_pose");
        inlier.Kpoints1.resize(n_inl);
        inlier.Kpoints2.resize(n_inl);
        inlier.match.resize(n_inl);
        for (int i = 0; i < n_inl; i++) inlier.match[i] = vo_dmatch{i, i, 0, 0.f};  // extract_Inlier :238-239
        success = ok != 0;
    }

    // Mat triangPoints(vector<Point2f> inl1, vector<Point2f> inl2, Mat R, Mat t, Mat K)   :288
    // returns 3 x M row-major (column j = point j); M may be 0 (the caller checks .empty(), visual_Odometry.cpp:386)
    std::vector<double> triangPoints(const std::vector<Point2f>& inl1, const std::vector<Point2f>& inl2, const Mat33& R,
                                     const Vec3& t, const Mat33& K, int* m_out) {
        const int n = (int)inl1.size();
        std::vector<double> X((size_t)3 * (n > 0 ? n : 1));
        int m = 0;
        check(vo_triang_points(ctx_, reinterpret_cast<const float*>(inl1.data()), reinterpret_cast<const float*>(inl2.data()), n,
                               R.data(), t.data(), K.data(), X.data(), n > 0 ? n : 1, &m),
              "vo_triang_points");
        std::vector<double> out((size_t)3 * m);
        for (int r = 0; r < 3; r++)
            for (int j = 0; j < m; j++) out[(size_t)r * m + j] = X[(size_t)r * (n > 0 ? n : 1) + j];
        if (m_out) *m_out = m;
        return out;
    }

    // checkMinFeat (:528): fewer than MIN_NUM_FEATURES matches -> FAIL_DETECTION
    static bool checkMinFeat(const KpAsPoint2f_Match& kP) { return vo_check_min_feat((int)kP.Kpoints1.size(), 20.0) != 0; }

    // checkIfMoving (:503): median match displacement >= DISTANCE (20 px); false -> NOT_MOVING
    static bool checkIfMoving(const KpAsPoint2f_Match& kP) {
        int moving = 0;
        check(vo_check_if_moving(reinterpret_cast<const float*>(kP.Kpoints1.data()), reinterpret_cast<const float*>(kP.Kpoints2.data()),
                                 (int)kP.Kpoints1.size(), 20.0, &moving, nullptr),
              "vo_check_if_moving");
        return moving != 0;
    }

    // scaleFactor (:384): world_points is the 3 x M row-major matrix triangPoints returned
    static double scaleFactor(float distance, const std::vector<double>& world_points) {
        double sf = 1.0;
        check(vo_scale_factor(distance, world_points.data(), (int)(world_points.size() / 3), &sf), "vo_scale_factor");
        return sf;
    }

    // absolutePose (:414, :467): returns scaled_locW, orient_wk and (when world_points is given) world_pointsW
    struct AbsPose {
        Vec3 scaled_locW;
        Mat33 orient_wk;
        std::vector<double> world_pointsW;
    };
    static AbsPose absolutePose(const Mat33& rotm, const Vec3& tran, const Mat33& orient, const Vec3& loc, double SF,
                                const std::vector<double>* world_points = nullptr) {
        AbsPose a;
        const int n = world_points ? (int)(world_points->size() / 3) : 0;
        a.world_pointsW.resize((size_t)3 * n);
        check(vo_absolute_pose(rotm.data(), tran.data(), orient.data(), loc.data(), SF, world_points ? world_points->data() : nullptr,
                               n, a.scaled_locW.data(), a.orient_wk.data(), n ? a.world_pointsW.data() : nullptr),
              "vo_absolute_pose");
        return a;
    }

    // Mat get_image(Mat current_img, int desired_width, int desired_height, Mat cameraMatrix, Mat distortionCoeff)   :147
    // bgr: CV_8UC3 view (data, rows, cols, step); K: camera matrix of the RESIZED frames (the node calls updateCameraMatrix
    // before, visual_Odometry.cpp:225); dist: k1 k2 p1 p2.  Returns the desired_height x desired_width gray frame, row-major.
    // The maps and tables of the camera are built on first use and kept (one camera per node).
    std::vector<uint8_t> get_image(const uint8_t* bgr, int rows, int cols, size_t step, int desired_width, int desired_height,
                                   const Mat33& K, const std::array<double, 4>& dist) {
        vo_prep_params p{};
        p.src_rows = rows, p.src_cols = cols, p.dst_rows = desired_height, p.dst_cols = desired_width;
        for (int i = 0; i < 9; i++) p.K[i] = K[i];
        for (int i = 0; i < 4; i++) p.dist[i] = dist[i];
        p.clip_limit = 16.0;  // CLIP_LIMIT, camera_utility.hpp:10
        p.tiles_x = p.tiles_y = 8;
        if (!prep_ || std::memcmp(&p, &prep_params_, sizeof(p)) != 0) {
            if (prep_) vo_prep_destroy(prep_);
            prep_ = nullptr;
            check(vo_prep_create(ctx_, &p, &prep_), "vo_prep_create");
            prep_params_ = p;
        }
        std::vector<uint8_t> gray((size_t)desired_height * desired_width);
        check(vo_prep_run(prep_, bgr, step, gray.data(), (size_t)desired_width), "vo_prep_run");
        return gray;
    }

    vo_ctx* handle() { return ctx_; }

   private:
    static void check(int st, const char* what) {
        if (st != VO_OK) throw std::runtime_error(std::string(what) + ": " + vo_strerror(st) + " " + vo_last_hip_error());
    }
    vo_ctx* ctx_ = nullptr;
    vo_params prm_{};
    vo_prep* prep_ = nullptr;
    vo_prep_params prep_params_{};
};

}  // namespace vo
