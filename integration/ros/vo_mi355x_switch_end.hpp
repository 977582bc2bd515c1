// Include immediately AFTER visual_odometry/camera_utility.hpp (see README.md).
#ifdef VO_USE_MI355X
#undef get_image
#undef opt_DetectFeatures
#undef optRelativePose
#undef triangPoints
#include "vo_cv_adaptor.hpp"
// The reference's parameter globals (camera_utility.hpp:33-63) are in scope here: the library's context is configured from
// them, at first use and again whenever one of them has been edited (vo_cv::frontend()).
namespace vo_cv_switch {
inline vo_params params_from_reference_globals() {
    vo_params p;
    std::memset(&p, 0, sizeof(p));  // padding too: the adaptor compares parameter sets bytewise
    vo_params_default(&p);
    p.hessian_threshold = (float)minHessian;          // :33
    p.n_octaves = nOctaves;                           // :35
    p.n_octave_layers = nOctaveLayers;                // :36
    p.extended = extended ? 1 : 0;                    // :37
    p.upright = upright ? 1 : 0;                      // :38
    p.ratio_thresh = ratio_thresh;                    // :41
    p.roi_x_lo = (float)width_low;                    // :44
    p.roi_y_lo = (float)height_low;                   // :45
    p.roi_x_hi = (float)width_high;                   // :46
    p.roi_y_hi = (float)height_high;                  // :47
    p.ransac_prob = ransac_prob[ESSENTIAL];           // :53 (rel_method = ESSENTIAL, :51)
    p.ransac_threshold = ransac_threshold[ESSENTIAL]; // :54
    p.inlier_threshold = inlier_threshold[ESSENTIAL]; // :56
    p.vpf_threshold = VPF_threshold;                  // :58
    p.distance_threshold = distance_threshold;        // :60
    p.reprojection_tolerance = reprojection_tolerance;  // :63
    return p;
}
inline const bool registered = (vo_cv::set_params_provider(&params_from_reference_globals), true);
}  // namespace vo_cv_switch
using vo_cv::get_image;           // camera_utility.hpp:147
using vo_cv::opt_DetectFeatures;  // :742
using vo_cv::optRelativePose;     // :633
using vo_cv::triangPoints;        // :288
#endif
