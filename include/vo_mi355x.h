/*
 * vo_mi355x.h -- C ABI of libvo_mi355x.so, the MI355X (gfx950) implementation of the
 * per-frame-pair visual-odometry front end of Elektron97/Visual_Odometry.
 *
 * The reference has no FFI/plugin interface: its boundary is three header-defined free
 * functions taking OpenCV value types.  Every entry point below names the
 * reference call it replaces (path:line under the reference checkout).  All buffers are
 * caller owned, sizes are explicit, no exception crosses the boundary: every function
 * returns VO_OK (0) or a negative vo_status; vo_strerror() gives the text.
 *
 * Images are passed exactly as cv::Mat exposes them: (data, rows, cols, step_bytes) of a
 * CV_8UC1 matrix.  3x3 / 3x4 matrices are row-major doubles (cv::Mat1d layout).  Point
 * arrays are interleaved float xy (std::vector<cv::Point2f> layout).
 *
 * A vo_ctx owns one HIP device, its streams and all device scratch.  It is NOT thread-safe:
 * use one context per host thread / GPU.  There is no CPU fallback: if the HIP runtime or a
 * gfx950 device is missing, vo_ctx_create fails with VO_ERR_NO_DEVICE.
 */
#ifndef VO_MI355X_H
#define VO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_ABI_VERSION 1

typedef enum vo_status {
    VO_OK = 0,
    VO_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad step            */
    VO_ERR_NO_DEVICE = -2,     /* no HIP device / not gfx950 / device index out of range */
    VO_ERR_HIP = -3,           /* a HIP runtime call failed (see vo_last_hip_error)   */
    VO_ERR_CAPACITY = -4,      /* an output did not fit the caller's capacity          */
    VO_ERR_UNSUPPORTED = -5,   /* parameter value not implemented (e.g. n_octaves > 5) */
    VO_ERR_TOO_FEW_POINTS = -6,/* fewer than 5 correspondences (findEssentialMat fails)*/
    VO_ERR_NO_MODEL = -7       /* RANSAC found no model with more than 4 inliers       */
} vo_status;

/* cv::KeyPoint mirror (pt.x, pt.y, size, angle, response, octave, class_id): 28 bytes.  */
typedef struct vo_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} vo_keypoint;

/* cv::DMatch mirror (queryIdx, trainIdx, imgIdx, distance).                             */
typedef struct vo_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} vo_dmatch;

/* enum fail_V0 of the node: src/visual_Odometry.cpp:58                                  */
typedef enum vo_fail { VO_FAIL_DETECTION = 0, VO_NOT_MOVING = 1, VO_FAIL_RECOVER = 2, VO_SUCCESS = 3 } vo_fail;

/*
 * Algorithm parameters.  In the reference these are header globals
 * (include/visual_odometry/camera_utility.hpp:11-13, 33-63); vo_params_default() fills the
 * same values.  The ROI of the reference (28/20/614/482) is tied to its 640x505 frames, so
 * it is a parameter here.
 */
typedef struct vo_params {
    float hessian_threshold;      /* minHessian = 50                       :33 */
    int32_t n_octaves;            /* 4 (1..5 accepted; a fifth octave runs :35
                                     plain kernels, with <= 3 layers only)    */
    int32_t n_octave_layers;      /* 3 (1..6 accepted; other than 3: plain :36
                                     det-plane / NMS kernels)                 */
    int32_t extended;             /* false = 64 floats per descriptor;     :37
                                     true accepted: 128 floats, sequential
                                     describer, exact-scan matcher            */
    int32_t upright;              /* false (true accepted: angle 270,      :38
                                     axis-aligned windows, sequential kernel) */
    float ratio_thresh;           /* 0.7f                                  :41 */
    float roi_x_lo, roi_y_lo;     /* width_low = 28, height_low = 20       :44-45 */
    float roi_x_hi, roi_y_hi;     /* width_high = 614, height_high = 482   :46-47 */
    double ransac_prob;           /* 0.99                                  :53 */
    double ransac_threshold;      /* 2.0 px                                :54 */
    int32_t ransac_max_iters;     /* 1000 (7-argument findEssentialMat)        */
    uint64_t ransac_seed;         /* cv::RNG((uint64)-1) inside the registrator */
    float inlier_threshold;       /* 0.3                                   :56 */
    float vpf_threshold;          /* 0.85                                  :58 */
    double distance_threshold;    /* 50.0 (recoverPose 6-arg default)      :60 */
    float reprojection_tolerance; /* 0.5 px                                :63 */
    int32_t kp_capacity;          /* max keypoints per frame kept on device (default 16384) */
    int32_t match_capacity;       /* max matches per pair (default = kp_capacity)           */
} vo_params;

typedef struct vo_ctx vo_ctx;
typedef struct vo_batch vo_batch;

int vo_abi_version(void);
const char* vo_strerror(int status);
/* text of the last failing HIP call on this thread ("" if none) */
const char* vo_last_hip_error(void);
void vo_params_default(vo_params* p);

/* Context: one per GPU / host thread.  params may be NULL (defaults).                  */
int vo_ctx_create(const vo_params* params, int hip_device, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
int vo_ctx_get_params(const vo_ctx* ctx, vo_params* out);

/* ---- a1-a2 (test / inspection entry points; the product path fuses a2 into a3) ------ */
/* cv::integral(img, sum, CV_32S) inside SURF detectAndCompute (camera_utility.hpp:749).
 * sum is (rows+1)*(cols+1) int32, row-major.                                            */
int vo_integral(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, int32_t* sum);
/* Fast-Hessian det/trace of one of the 20 layers (octave 0..3, layer 0..4).  Output
 * arrays are (rows/step_o)*(cols/step_o) floats; untouched border samples are 0.         */
int vo_hessian_layer(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step,
                     int octave, int layer, float* det, float* trace);

/* The same det values as the PRODUCT kernels hold them (the fused Hessian + NMS launches of vo_detect_describe keep the five
 * planes of an octave in LDS and never write them out; this inspection call makes them store what their NMS reads).
 * det_all: all 20 layers packed octave-major, layer l of octave o is [rows >> o][cols >> o]; n_floats must be the total.
 * VO_ERR_UNSUPPORTED for a context with other than the reference's 4 octaves x 3 layers (those run other kernels).          */
int vo_hessian_product_det(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step,
                           float* det_all, size_t n_floats);

/* ---- a1-a6: SURF::create(50,4,3,false,false)->detectAndCompute(img, noArray(), kps, desc)
 * camera_utility.hpp:746-750.  kps/desc64 hold up to cap entries; *n = count written.
 * Keypoints come out in OpenCV's KeypointGreater order.  desc64 may be NULL (detect only:
 * angle is then -1 as after fastHessianDetector).  A descriptor is 64 floats - 128 when
 * the context was created with vo_params.extended, here and in vo_match_knn2.           */
int vo_detect_describe(vo_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step,
                       vo_keypoint* kps, float* desc64, int cap, int* n);

/* ---- a7: knnMatch(desc1, desc2, knn, 2), exact brute-force L2 (camera_utility.hpp:759-762;
 * the reference uses FLANN, the "OLD METHOD" :752-755 and the north star are brute force).
 * idx2/dist2 are [nA][2]; dist = sqrt(sum (a-b)^2), ties to the lower train index.       */
int vo_match_knn2(vo_ctx* ctx, const float* descA, int nA, const float* descB, int nB,
                  int32_t* idx2, float* dist2);

/* ---- a1-a9: opt_DetectFeatures(img1, img2, kP) camera_utility.hpp:742-809.
 * p1xy/p2xy: matched keypoint positions (kP.Kpoints1/2), m: kP.match, all up to cap.     */
int vo_detect_match(vo_ctx* ctx, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2,
                    int rows, int cols, float* p1xy, float* p2xy, vo_dmatch* m, int cap, int* n);

/* ---- a10: findEssentialMat(p1, p2, K, RANSAC, prob, thr, mask) camera_utility.hpp:642.
 * mask is n bytes of 0/1.  *n_inl = inlier count, *iters_run = hypotheses consumed by the
 * sequential-equivalent loop (may be NULL).                                              */
int vo_essential_ransac(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9],
                        double prob, double thr_px, int max_iters, uint64_t seed,
                        double E[9], uint8_t* mask, int* n_inl, int* iters_run);

/* The deterministic 5-tuples the registrator draws (cv::RNG stream).
 * idx5 is [n_tuples][5].  Pure host logic, no GPU needed.                               */
int vo_ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5);

/* Nister five-point solver for explicit 5-tuples (test entry point for "one wavefront per
 * hypothesis").  pts are normalised coordinates, [n_hyp][5][2] doubles for each image.
 * E_out [n_hyp][10][9], n_models [n_hyp]; candidates ordered by ascending real root z.   */
int vo_five_point(vo_ctx* ctx, const double* x1, const double* x2, int n_hyp,
                  double* E_out, int32_t* n_models);

/* Sampson scoring of given models (EMEstimatorCallback::computeError + findInliers).
 * pts are PIXEL coordinates; masks [n_models][n] bytes, counts [n_models].               */
int vo_sampson_masks(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9],
                     double thr_px, const double* E, int n_models, uint8_t* masks, int32_t* counts);

/* ---- a12: recoverPose(E, inl1, inl2, K, R, t) camera_utility.hpp:667                  */
int vo_recover_pose(vo_ctx* ctx, const double E[9], const float* p1xy, const float* p2xy, int n,
                    const double K[9], double dist_thresh, double R[9], double t[3], int* n_good);

/* ---- a10-a12 + gates: optRelativePose(kP, K, R, t, inlier, success) camera_utility.hpp:633-685.
 * R_io/t_io are overwritten only on success.  in1xy/in2xy need room for n points.        */
int vo_relative_pose(vo_ctx* ctx, const float* p1xy, const float* p2xy, int n, const double K[9],
                     double R_io[9], double t_io[3], float* in1xy, float* in2xy, int* n_inl, int* success);

/* ---- a13: triangulatePoints(P0, P1, pts1, pts2, X4) camera_utility.hpp:320.
 * X4 is [4][n] row-major CV_32F, unit 4-vectors (sign arbitrary).                        */
int vo_triangulate(vo_ctx* ctx, const double P0[12], const double P1[12],
                   const float* p1xy, const float* p2xy, int n, float* X4);

/* ---- a13-a14: triangPoints(inl1, inl2, R, t, K) camera_utility.hpp:288-351.
 * Xk is [3][cap] row-major doubles (column j = point j, frame {k}); *m = points kept.    */
int vo_triang_points(vo_ctx* ctx, const float* in1xy, const float* in2xy, int n,
                     const double R[9], const double t[3], const double K[9],
                     double* Xk, int cap, int* m);

/* ---- triangPoints without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot allocate
 * device memory - tight serving loops, harnesses that forbid hipMalloc (the call still copies through the context's pinned
 * buffers and waits on the context's stream, so it cannot itself be CAPTURED into a HIP graph).  Same arguments,
 * results and status codes as vo_triang_points (bit-identical output), but every intermediate lives in a static device
 * array of the library (1.77 MiB per GPU, held until the process ends) and the host staging goes through the pinned
 * buffers vo_ctx_create made: the call issues no hipMalloc / hipFree / hipHostMalloc.  Limits: n <= VO_NOALLOC_MAX_POINTS,
 * otherwise VO_ERR_CAPACITY (vo_last_hip_error() says so); VO_ERR_UNSUPPORTED while vo_profile_enable is on (the stage
 * timers create events).  All contexts of one GPU share the array: calls that overlap on one GPU race.                */
#define VO_NOALLOC_MAX_POINTS 16384
int vo_triang_points_noalloc(vo_ctx* ctx, const float* in1xy, const float* in2xy, int n,
                             const double R[9], const double t[3], const double K[9],
                             double* Xk, int cap, int* m);

/* ---- batched, device-resident pipeline: all pairs of a batch run through
 * detect x2 -> match -> E-RANSAC -> recoverPose -> triangulate with no host round trip.  */
typedef struct vo_pair_out {
    int32_t status;         /* vo_status of this pair                                    */
    int32_t n_kp1, n_kp2;   /* keypoints per frame                                       */
    int32_t n_match;        /* matches after ratio + ROI                                 */
    int32_t n_inl;          /* RANSAC inliers                                            */
    int32_t n_good;         /* recoverPose cheirality count                              */
    int32_t n_world;        /* triangulated points kept                                  */
    int32_t success;        /* optRelativePose success flag                              */
    int32_t iters;          /* RANSAC hypotheses consumed                                */
    int32_t pad_;
    double E[9];
    double R[9];
    double t[3];
} vo_pair_out;

int vo_batch_create(vo_ctx* ctx, int n_pairs, int rows, int cols, vo_batch** out);
void vo_batch_destroy(vo_batch* b);
/* Copy one pair's frames host -> device.  The frames exist twice on the device: this writes the slot the latest
 * vo_batch_run does NOT read, through pinned staging memory (the caller's buffers are free when the call returns) and a
 * copy stream of the batch's own, so the transfer of step k + 1 runs beside the kernels of step k.  Frames that are not
 * replaced between two runs carry over; every run starts every pair from R = I, t = 0.                                 */
int vo_batch_set_pair(vo_batch* b, int pair, const uint8_t* img1, size_t step1,
                      const uint8_t* img2, size_t step2, const double K[9]);
/* enqueue the whole pipeline for all pairs on the frames staged so far; returns without waiting.  A pair that has no
 * camera matrix yet (K[0] or K[4] zero: no vo_batch_set_pair / vo_batch_set_k / vo_seq_set_pair for it) does not fail the
 * run: its own record comes back with status VO_ERR_INVALID_ARG and success 0, the other pairs run as usual.            */
int vo_batch_run(vo_batch* b);
/* wait for the uploads issued so far and for the latest run                                                            */
int vo_batch_sync(vo_batch* b);
/* result record of a pair: waits for the latest run only (the records travel to pinned host memory with the run)       */
int vo_batch_get(vo_batch* b, int pair, vo_pair_out* out);
/* the same for the latest run (age 0) or the run before it (age 1): a pipelined caller enqueues run k + 1 and then reads
 * run k's records while k + 1 executes.  The records of a run stay valid until the second run after it is enqueued.    */
int vo_batch_get_run(vo_batch* b, int age, int pair, vo_pair_out* out);
/* the records of all n_pairs pairs of that run in one call (outs[n_pairs]; same waiting rule)                            */
int vo_batch_get_run_all(vo_batch* b, int age, vo_pair_out* outs);
/* optional bulk outputs of a pair of the LATEST run (any pointer may be NULL); waits for that run                      */
int vo_batch_get_points(vo_batch* b, int pair, float* p1xy, float* p2xy, vo_dmatch* m, uint8_t* mask,
                        int cap, double* Xk, int cap_world);
/* The same for EVERY pair of the latest run in one download (one synchronisation instead of one per pair; the offline replay
 * reads all pairs of a chunk).  Row p of an output belongs to pair p: p1xy / p2xy [n_pairs][cap][2], m and mask
 * [n_pairs][cap], Xk [n_pairs][3][cap_world]; only the first n_match (n_world) entries of a row are written with results.
 * Null outputs are skipped.  VO_ERR_CAPACITY when some pair holds more than cap matches (cap_world world points).
 * Replaces the per-pair reads of KpAsPoint2f_Match / world_points after camera_utility.hpp:795-808, 343-350. */
int vo_batch_get_points_all(vo_batch* b, float* p1xy, float* p2xy, vo_dmatch* m, uint8_t* mask, int cap, double* Xk,
                            int cap_world);

/* ---- multi-GPU: vo_pair_batch(ctx_per_gpu[], n_gpu, pairs, n_pairs, out).  Replaces, for offline batches,
 * the node's single sequential loop over frame pairs (src/visual_Odometry.cpp:268-470: opt_DetectFeatures :296,
 * optRelativePose :336, triangPoints :384 once per iteration).  One host thread per context (the calling thread drives the
 * first one), pair p runs on context p mod n_ctx through that context's pipelined batch loop (uploads beside kernels, steps
 * of up to 64 pairs at 1080p), and every thread writes its records into out[p] - a host gather, nothing crosses between
 * GPUs.  Contexts are normally one per GPU (vo_ctx_create(params, g, &ctx[g]) for g < vo_device_count()); several contexts
 * on one GPU are allowed.  Frames are cv::Mat-style CV_8UC1 buffers, all of one size; records are bit-identical to those of
 * a single context running the same pairs.  Returns the first failing context's status (text: vo_last_hip_error).      */
typedef struct vo_pair_in {
    const uint8_t* img1;    /* prev_img: data, step of a CV_8UC1 cv::Mat          */
    size_t step1;
    const uint8_t* img2;    /* curr_img                                            */
    size_t step2;
    int32_t rows, cols;     /* the same for every pair of a call                   */
    double K[9];            /* camera matrix, row-major                            */
} vo_pair_in;
int vo_device_count(void);  /* HIP devices visible to the process (0 when there is no runtime / device) */
int vo_pair_batch(vo_ctx* const* ctx_per_gpu, int n_ctx, const vo_pair_in* pairs, int n_pairs, vo_pair_out* out);

/* Sequence batch (offline replay of consecutive frames; no counterpart in the reference, which re-detects prev_img
 * in every iteration, src/visual_Odometry.cpp:296): n_frames frames are detected and described once each;
 * n_frames - 1 pairs, pair p = (frame p, frame p + 1) unless vo_seq_set_pair re-pairs it (the NOT_MOVING case keeps
 * prev_img, :316-326).  vo_seq_set_pair gives a pair its K; a pair without one reports VO_ERR_INVALID_ARG in its own
 * record (see vo_batch_run).  Run / sync / get / get_points / destroy are the vo_batch_* calls.                  */
int vo_seq_create(vo_ctx* ctx, int n_frames, int rows, int cols, vo_batch** out);
int vo_seq_set_frame(vo_batch* b, int frame, const uint8_t* img, size_t step);
int vo_seq_set_pair(vo_batch* b, int pair, int frame_a, int frame_b, const double K[9]);

/* Inspection of the context's last single-call RANSAC (tests): what = 0 tuples [4096][5] int32,
 * 1 nmodels [4096] int32, 2 counts [4096][10] int32, 3 models [4096][10][9] double;
 * what = 4: one int32, the queries of the context's last matcher launch whose MFMA shortlist could not be
 * certified and that took the exact scan instead; what = 5 / 6: the shortlist that launch stored for
 * pair 0, [kp_cap][32] int32 block ids (4 * tile + j, -1 = empty) / float block maxima, per query
 * 8 groups (train split * 2 + lane half) of {3 kept blocks, bound on everything else}.           */
int vo_debug_fetch(vo_ctx* ctx, int what, void* dst, size_t bytes);
/* Test hook, no counterpart in the reference.  key 0: value != 0 sends EVERY keypoint's descriptor window
 * through the sequential coordinate walk (the path otherwise taken only when the closed-form window
 * coordinates are not exactly representable); results are identical either way.
 * key 1: value != 0 certifies no matcher shortlist, so that EVERY query takes the exact scan of all train rows
 * (otherwise a handful per frame pair); results are identical either way.
 * key 2: value != 0 runs the matcher's dense contraction on the FP32 matrix instruction (v_mfma_f32_32x32x2_f32, the
 * formulation BASELINE's north star names) instead of three split-bf16 products; same shortlist, certificate and
 * results, about 2.5 times the kernel time.                                                               */
int vo_debug_set(vo_ctx* ctx, int key, int value);

/* ---- N1: gates and pose chain of the node, host arithmetic only (no GPU, no context) ----------------
 * These are the functions src/visual_Odometry.cpp calls around the three hot-path entry points, so that an offline
 * replay can emit the reference's vo_results / fail_check fields.  Matrices row-major, 3-vectors plain.            */
/* checkMinFeat, camera_utility.hpp:528 (MIN_NUM_FEATURES = 20.0, :12): 1 when the pair must be skipped           */
int vo_check_min_feat(int n_matches, double min_features);
/* checkIfMoving, camera_utility.hpp:503 (DISTANCE = 20.0 px, :11): float displacements, double median            */
int vo_check_if_moving(const float* p1xy, const float* p2xy, int n, double distance, int* moving, double* median_px);
/* scaleFactor, camera_utility.hpp:384: distance / mean(Z) of the 3 x n world points                              */
int vo_scale_factor(float distance, const double* Xk, int n, double* sf);
/* absolutePose (both overloads), camera_utility.hpp:414-501.  Xw may be NULL (the success = false overload)      */
int vo_absolute_pose(const double rotm[9], const double tran[3], const double orient[9], const double loc[3], double sf,
                     const double* Xk, int n, double t_wk[3], double R_wk[9], double* Xw);
/* mat2Euler = tf::Matrix3x3::getRPY, math_utility.hpp:109                                                        */
int vo_mat2euler(const double R[9], double rpy[3]);
/* wrap2pi, math_utility.hpp:260; absDiff_Vec3 (three overloads), :265-310                                        */
double vo_wrap2pi(double angle);
int vo_abs_diff3(const double a[3], const double b[3], int is_angle, double out[3]);
/* estimateTwist / computeAngularVel, math_utility.hpp:312-357                                                    */
int vo_estimate_twist(double dt, const double R[9], const double t[3], double sf, const double rpy[3], const double Rbc[9],
                      int motion2d, double lin[3], double ang[3]);
/* One pose update of the replayed loop in one call: absolutePose on the running pose (src/visual_Odometry.cpp:402-410), then
 * publish_VOResults' rpy = mat2Euler(orientation * Rbc^T) and twist (:553-570; zero when dt == 0).  Xk / Xw as above.   */
int vo_loop_pose_update(double orientation_io[9], double location_io[3], const double R[9], const double t[3], double sf,
                        const double* Xk, int n, double* Xw, double dt, const double Rbc[9], int motion2d, double rpy[3],
                        double lin[3], double ang[3]);

/* ---- N2: get_image, camera_utility.hpp:147-179 -- the per-frame preprocessing of the node -----------
 * resize(INTER_AREA) :119-136 -> undistort(K, dist) :155 -> cvtColor(COLOR_RGB2GRAY) on the node's BGR frames :160 ->
 * CLAHE(clip 16, 8 x 8 tiles) :165-169, all on the device.  One vo_prep holds the maps and tables of one camera.   */
typedef struct vo_prep vo_prep;
typedef struct vo_prep_params {
    int32_t src_rows, src_cols;   /* size of the incoming BGR8 frames                                             */
    int32_t dst_rows, dst_cols;   /* desired_size (:151); equal to the source size = no resize (:125)             */
    double K[9];                  /* camera matrix of the RESIZED frames, i.e. after vo_update_camera_matrix       */
    double dist[4];               /* k1 k2 p1 p2 (the node's yaml holds four coefficients)                        */
    double clip_limit;            /* CLIP_LIMIT = 16.0, camera_utility.hpp:10                                     */
    int32_t tiles_x, tiles_y;     /* createCLAHE() default 8 x 8                                                  */
} vo_prep_params;
/* updateCameraMatrix, camera_utility.hpp:138-145: diag(RW, RH, 1) * K                                            */
int vo_update_camera_matrix(int src_rows, int src_cols, int dst_rows, int dst_cols, const double K[9], double K_out[9]);
int vo_prep_create(vo_ctx* ctx, const vo_prep_params* p, vo_prep** out);
void vo_prep_destroy(vo_prep* pp);
/* get_image on one frame: host BGR8 in, host gray8 out (synchronous)                                              */
int vo_prep_run(vo_prep* pp, const uint8_t* bgr, size_t step, uint8_t* gray, size_t gray_step);
/* get_image with the result left on the device as frame `frame` of a batch (frame = 2 * pair + {0, 1} in a pair
 * batch, the frame index in a sequence batch); asynchronous like vo_batch_set_pair.  The pair's K still comes from
 * vo_batch_set_k / vo_seq_set_pair.                                                                               */
int vo_prep_to_batch(vo_prep* pp, const uint8_t* bgr, size_t step, vo_batch* b, int frame);
/* K of a pair whose frames arrive through vo_prep_to_batch (pair batches)                                         */
int vo_batch_set_k(vo_batch* b, int pair, const double K[9]);
/* Measurement hook, no counterpart in the reference: enqueue the four passes `reps` more times on the frame already
 * resident on the device (no upload), asynchronously on the context's stream -- the "inputs resident in HBM" timing. */
int vo_prep_rerun(vo_prep* pp, int reps);
/* inspection of the last frame's intermediates (tests): what = 0 resized BGR [dst_rows][dst_cols][3], 1 undistorted
 * BGR, 2 gray, 3 CLAHE look-up tables [tiles_y][tiles_x][256], 4 map1 int16 [dst_rows][dst_cols][2], 5 map2 uint16 */
int vo_prep_fetch(vo_prep* pp, int what, void* dst, size_t bytes);

/* Per-stage device timing (HIP events on the context's compute stream).  Enable, run,
 * sync, then read.  Stage names: vo_stage_name(i), i in [0, vo_stage_count()).           */
int vo_profile_enable(vo_ctx* ctx, int on);
int vo_stage_count(void);
const char* vo_stage_name(int stage);
/* accumulated milliseconds and launch count of a stage since the last reset             */
int vo_profile_get(vo_ctx* ctx, int stage, double* ms, int64_t* launches);
int vo_profile_reset(vo_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VO_MI355X_H */
