// Include immediately BEFORE visual_odometry/camera_utility.hpp (see README.md).  With VO_USE_MI355X the reference's four
// OpenCV implementations are parsed under other names, so that the names themselves are free for the library's functions.
#ifdef VO_USE_MI355X
#define get_image get_image_opencv
#define opt_DetectFeatures opt_DetectFeatures_opencv
#define optRelativePose optRelativePose_opencv
#define triangPoints triangPoints_opencv
#endif
