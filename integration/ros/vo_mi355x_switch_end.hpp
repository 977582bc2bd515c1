// Include immediately AFTER visual_odometry/camera_utility.hpp (see README.md).
#ifdef VO_USE_MI355X
#undef get_image
#undef opt_DetectFeatures
#undef optRelativePose
#undef triangPoints
#include "vo_cv_adaptor.hpp"
using vo_cv::get_image;           // camera_utility.hpp:147
using vo_cv::opt_DetectFeatures;  // :742
using vo_cv::optRelativePose;     // :633
using vo_cv::triangPoints;        // :288
#endif
