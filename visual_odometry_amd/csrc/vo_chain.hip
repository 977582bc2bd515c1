// N1: the gates and the pose chain the node runs around the three hot-path calls.  Host arithmetic only -
// microseconds per frame pair, no GPU work - exported through the same C ABI so that an offline replay tool (or the
// node itself) can produce the reference's vo_results / fail_check fields without ROS, tf or OpenCV.
// Reference: include/visual_odometry/camera_utility.hpp:384-531, math_utility.hpp:109-357, src/visual_Odometry.cpp:296-430.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/vo_mi355x.h"

// camera_utility.hpp:528-531   return kP.Kpoints1.size() < MIN_NUM_FEATURES   (MIN_NUM_FEATURES = 20.0, :12)
extern "C" int vo_check_min_feat(int n_matches, double min_features) { return (double)n_matches < min_features ? 1 : 0; }

// math_utility.hpp:214-233: sort, mean of the two middle elements for an even count
static double median_of(std::vector<double> v) {
    const size_t n = v.size();
    std::sort(v.begin(), v.end());
    if (n % 2 == 0) return (v[n / 2 - 1] + v[n / 2]) / 2.0;
    return v[n / 2];
}

// camera_utility.hpp:503-526: displacement of every match computed in FLOAT (powf, float sqrt), median in double,
// moving iff median >= DISTANCE (= 20.0 px, :11).  The reference is undefined for n = 0 (median of an empty vector);
// here that is VO_ERR_INVALID_ARG - the node never gets there (checkMinFeat runs first).
extern "C" int vo_check_if_moving(const float* p1xy, const float* p2xy, int n, double distance, int* moving, double* median_px) {
    if (!p1xy || !p2xy || n <= 0 || !moving) return VO_ERR_INVALID_ARG;
    std::vector<double> d((size_t)n);
    for (int i = 0; i < n; i++) {
        const float dx = p1xy[2 * i] - p2xy[2 * i], dy = p1xy[2 * i + 1] - p2xy[2 * i + 1];
        const float s = dx * dx + dy * dy;  // powf(x, 2.0) is exactly x * x
        d[i] = (double)std::sqrt(s);        // std::sqrt(float) -> float
    }
    const double med = median_of(d);
    if (median_px) *median_px = med;
    *moving = med < distance ? 0 : 1;
    return VO_OK;
}

// camera_utility.hpp:384-412: distance / mean(Z), Z summed in column order; 1.0 when the mean is 0
extern "C" int vo_scale_factor(float distance, const double* Xk, int n, double* sf) {
    if (!Xk || n <= 0 || !sf) return VO_ERR_INVALID_ARG;
    double zsum = 0.0;
    for (int i = 0; i < n; i++) zsum += Xk[(size_t)2 * n + i];  // row 2 of the 3 x n matrix
    const double zmean = zsum / n;
    *sf = zmean == 0 ? 1.0 : (double)distance / zmean;
    return VO_OK;
}

static void mat3_mul_bt(const double A[9], const double B[9], double C[9]) {  // C = A * B^T
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// camera_utility.hpp:414-501 (both overloads): orient_wk = rotm * orient^T; scaled_locW = rotm * (-SF * orient^T * loc) + tran;
// world point j = orient_wk * (SF * X_j) + scaled_locW.  Xk / Xw are 3 x n row-major (column j = point j), Xw may be NULL.
extern "C" int vo_absolute_pose(const double rotm[9], const double tran[3], const double orient[9], const double loc[3], double sf,
                                const double* Xk, int n, double t_wk[3], double R_wk[9], double* Xw) {
    if (!rotm || !tran || !orient || !loc || !t_wk || !R_wk || n < 0 || (n > 0 && Xw && !Xk)) return VO_ERR_INVALID_ARG;
    mat3_mul_bt(rotm, orient, R_wk);
    double otl[3];
    for (int i = 0; i < 3; i++) otl[i] = -sf * (orient[i] * loc[0] + orient[3 + i] * loc[1] + orient[6 + i] * loc[2]);
    double tl[3];
    mat3_vec(rotm, otl, tl);
    for (int i = 0; i < 3; i++) t_wk[i] = tl[i] + tran[i];
    if (Xw)
        for (int j = 0; j < n; j++) {
            const double x[3] = {sf * Xk[j], sf * Xk[(size_t)n + j], sf * Xk[(size_t)2 * n + j]};
            double y[3];
            mat3_vec(R_wk, x, y);
            for (int i = 0; i < 3; i++) Xw[(size_t)i * n + j] = y[i] + t_wk[i];
        }
    return VO_OK;
}

// math_utility.hpp:109-119 mat2Euler = tf::Matrix3x3::getRPY (getEulerYPR, solution 1)
extern "C" int vo_mat2euler(const double R[9], double rpy[3]) {
    if (!R || !rpy) return VO_ERR_INVALID_ARG;
    const double m20 = R[6];
    if (std::fabs(m20) >= 1) {
        const double delta = m20 < 0 ? std::atan2(R[1], R[2]) : std::atan2(-R[1], -R[2]);
        rpy[2] = 0;
        rpy[1] = m20 < 0 ? M_PI / 2.0 : -M_PI / 2.0;
        rpy[0] = delta;
    } else {
        const double pitch = -std::asin(m20), c = std::cos(pitch);
        rpy[1] = pitch;
        rpy[0] = std::atan2(R[7] / c, R[8] / c);
        rpy[2] = std::atan2(R[3] / c, R[0] / c);
    }
    return VO_OK;
}

// math_utility.hpp:260-263
extern "C" double vo_wrap2pi(double angle) { return std::atan2(std::sin(angle), std::cos(angle)); }

// math_utility.hpp:265-310 (all three overloads)
extern "C" int vo_abs_diff3(const double a[3], const double b[3], int is_angle, double out[3]) {
    if (!a || !b || !out) return VO_ERR_INVALID_ARG;
    for (int i = 0; i < 3; i++) out[i] = std::fabs(is_angle ? vo_wrap2pi(a[i] - b[i]) : a[i] - b[i]);
    return VO_OK;
}

// math_utility.hpp:312-357 estimateTwist + computeAngularVel.  lin = Rbc * (-SF * R^T * t / dt);
// 2-D motion: ang = (0, 0, atan2(R10, R00) / dt); 3-D: w = J(rpy) * (mat2Euler(R) / dt) as written in the reference
// (Rbc does not enter the angular part there either).
extern "C" int vo_estimate_twist(double dt, const double R[9], const double t[3], double sf, const double rpy[3],
                                 const double Rbc[9], int motion2d, double lin[3], double ang[3]) {
    if (!R || !t || !rpy || !Rbc || !lin || !ang || dt == 0) return VO_ERR_INVALID_ARG;
    double v[3];
    for (int i = 0; i < 3; i++) v[i] = -sf * (R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]) / dt;
    mat3_vec(Rbc, v, lin);
    if (motion2d) {
        ang[0] = 0;
        ang[1] = 0;
        ang[2] = std::atan2(R[3], R[0]) / dt;
    } else {
        double d[3];
        vo_mat2euler(R, d);
        for (int i = 0; i < 3; i++) d[i] /= dt;
        ang[0] = (std::sin(rpy[0]) / std::cos(rpy[1])) * d[1] + (std::cos(rpy[0]) / std::cos(rpy[1])) * d[2];
        ang[1] = -d[0] - std::sin(rpy[0]) * std::tan(rpy[1]) * d[1] - std::cos(rpy[0]) * std::tan(rpy[1]) * d[2];
        ang[2] = -std::cos(rpy[0]) * d[1] + std::sin(rpy[0]) * d[2];
    }
    return VO_OK;
}

// One pose update of the node's loop as the offline replay performs it, in one call: absolutePose on the running pose
// (src/visual_Odometry.cpp:402-410 -> camera_utility.hpp:414-501), then what publish_VOResults derives from it (:553-570):
// rpy = mat2Euler(orientation * Rbc^T) and the twist (zero when dt == 0).  The same arithmetic as the three separate calls.
extern "C" int vo_loop_pose_update(double orientation_io[9], double location_io[3], const double R[9], const double t[3], double sf,
                                   const double* Xk, int n, double* Xw, double dt, const double Rbc[9], int motion2d, double rpy[3],
                                   double lin[3], double ang[3]) {
    if (!orientation_io || !location_io || !R || !t || !Rbc || !rpy || !lin || !ang) return VO_ERR_INVALID_ARG;
    double t_wk[3], R_wk[9];
    const int st = vo_absolute_pose(orientation_io, location_io, R, t, sf, Xk, n, t_wk, R_wk, Xw);
    if (st != VO_OK) return st;
    for (int i = 0; i < 3; i++) location_io[i] = t_wk[i];
    for (int i = 0; i < 9; i++) orientation_io[i] = R_wk[i];
    double M[9];
    mat3_mul_bt(orientation_io, Rbc, M);
    vo_mat2euler(M, rpy);
    if (dt != 0)
        return vo_estimate_twist(dt, R, t, sf, rpy, Rbc, motion2d, lin, ang);
    for (int i = 0; i < 3; i++) lin[i] = ang[i] = 0;
    return VO_OK;
}
