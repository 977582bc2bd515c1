// Internal declarations of libvo_mi355x.so (gfx950 only; no CPU fallback, no dual paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vo_mi355x.h"

// ---------------------------------------------------------------- errors
void vo_set_hip_error(const char* what, hipError_t e, const char* file, int line);
void vo_set_error_text(const char* text);  // what vo_last_hip_error() returns next (errors that are not HIP's)
#define VO_HIP(call)                                             \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) {                                  \
            vo_set_hip_error(#call, e_, __FILE__, __LINE__);     \
            return VO_ERR_HIP;                                   \
        }                                                        \
    } while (0)
#define VO_TRY(call)               \
    do {                           \
        int s_ = (call);           \
        if (s_ != VO_OK) return s_; \
    } while (0)

// ---------------------------------------------------------------- constants of the algorithm
constexpr int kNumOctaves = 4;
constexpr int kLayersPerOctave = 5;  // nOctaveLayers + 2
constexpr int kNumLayers = kNumOctaves * kLayersPerOctave;
constexpr int kMaxOctaves = 5;       // vo_params.n_octaves 1..5 (octaves 0..3: the tuned kernels; a fifth: plain kernels, <= 3 layers)
constexpr int kMaxOctaveLayers = 6;  // vo_params.n_octave_layers 1..6 (3 = the reference's, the LDS-tile kernels; others: plain kernels)
constexpr int kOriSamples = 113;  // lattice points with i^2 + j^2 <= 36 (SURVEY A.5 says 109; the loop it quotes yields 113)
constexpr int kPatch = 20;        // PATCH_SZ
constexpr int kMaxHyp = 4096;     // largest max_iters accepted (BASELINE config 5)
constexpr int kMaxModels = 10;

// Box pattern of one fast-Hessian layer, edges shared between adjacent boxes.
struct LayerTab {
    int size, step, margin, pad0;
    int xdx[4], ydx[2];  // Dx: 3 boxes side by side in x
    int xdy[2], ydy[4];  // Dy: 3 boxes stacked in y
    int cxy[4];          // Dxy: edges {1,4,5,8} scaled, same for x and y
    float w[10];         // weights: Dx[3], Dy[3], Dxy[4]
    float pad1[2];
};

struct DetectTables {
    LayerTab layers[kNumLayers];
    int aptx[kOriSamples + 15], apty[kOriSamples + 15];
    float aptw[kOriSamples + 15];
    float DW[kPatch * kPatch];
};

// per-frame device counters
struct FrameCounters {
    int n_raw;    // keypoints appended by NMS (may exceed capacity)
    int n_kp;     // min(n_raw, cap): keypoints in the sorted list
    int pad[2];
};

// per-pair device state of matching + geometry
struct PairState {
    int n_match;       // matches after ratio + ROI (may exceed capacity -> clipped in n_used)
    int n_used;        // min(n_match, match_cap)
    int niters;        // current RANSAC iteration bound (sequential semantics)
    int best_count;    // maxGoodCount
    int best_hyp;      // hypothesis index of the best model (-1 none)
    int best_model;    // model index within the hypothesis
    int scanned;       // hypotheses already consumed by the in-order scan
    int n_inl;
    int good[4];       // cheirality counts of the 4 pose candidates
    int n_good;
    int success;
    int status;
    int n_pre;         // triangulated points passing reprojection + Z>0
    int n_world;
    int iters_run;
    float thr_sq;      // (threshold / f)^2 as float (findInliers)
    int pad[1];
    double K[9];
    double E[9];
    double R[9];       // in/out pose (identity / zero before the first success)
    double t[3];
    double Rc[2][9];   // decomposeEssentialMat candidates
    double tc[3];
    double zstat[2];   // mean / variance of Z for the 3-sigma filter
};

// ---------------------------------------------------------------- profiling stages
enum Stage {
    ST_INTEGRAL = 0,
    ST_HESSIAN_OCT1,    // k_hessian_nms_lds<24, 1>
    ST_HESSIAN_OCT0,    // k_hessian_nms_fused (octave 0 from LDS + octaves 2-3)
    ST_SORT,
    ST_ORIENTATION,
    ST_DESC_PREP,       // k_quads + k_classify
    ST_DESC_W768,       // k_describe_tiled, one stage per window class
    ST_DESC_W320,
    ST_DESC_W128,
    ST_DESC_W64,
    ST_DESC_SEQ,        // k_describe: sequential-walk fallback list
    ST_MATCH_PREP,
    ST_MATCH_MFMA,
    ST_MATCH_SELECT,
    ST_RANSAC_SAMPLE,
    ST_RANSAC_SOLVE_SCORE,
    ST_RANSAC_SELECT,
    ST_POSE,
    ST_TRIANGULATE,
    ST_PREP_RESIZE,     // vo_prep.hip (N2): INTER_AREA resize
    ST_PREP_REMAP,      // undistort remap + gray
    ST_PREP_CLAHE,      // tile look-up tables + blending
    ST_COUNT
};

struct StageTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // events of unresolved launches
    double ms = 0;
    int64_t launches = 0;
};

// ---------------------------------------------------------------- device buffers
// list lengths [0..4]; then one work-queue ticket counter per (class, list segment), each on a cache line of its own:
// atomics on one line complete one after the other (~11 ns each), and at the end of a launch every workgroup draws once
// from every segment
constexpr int kTicketStride = 32;  // ints = 128 bytes
constexpr int kTicketBase = 64;
__host__ __device__ constexpr int ticket_index(int cls, int seg) { return kTicketBase + (cls * 8 + seg) * kTicketStride; }
constexpr int kClsCntInts = kTicketBase + 5 * 8 * kTicketStride;

struct FrameSet {
    int F = 0, H = 0, W = 0, cap = 0;
    int dlen = 64;               // floats per descriptor: 64, or 128 with vo_params.extended
    uint8_t* img = nullptr;      // [F][H][W]
    int32_t* sum = nullptr;      // [F][H+1][W+1]
    int32_t* sum4 = nullptr;     // [F][4][H+1][sum4_pitch] the same values by column phase: sum[y][x] at plane x & 3, word x >> 2 (coarse octaves)
    int32_t* colsum = nullptr;   // [F][nbands][W]
    uint32_t* quad = nullptr;    // [F][ceil(W/8)][H][8] bilinear footprints p(y,x) | p(y,x+1)<<8 | p(y+1,x)<<16 | p(y+1,x+1)<<24 in 8-pixel column strips
    vo_keypoint* kp_raw = nullptr;  // [F][cap]
    vo_keypoint* kp = nullptr;      // [F][cap] sorted
    int* rank = nullptr;            // [F][cap]
    float* desc = nullptr;          // [F][cap][dlen]
    float* descT = nullptr;         // [F][cap/32][9][64] x 16 B: MFMA operand layout (bf16 high parts, low parts, norm term)
    float* nrm = nullptr;           // [F][cap]  -0.5*||b||^2
    int* nmax = nullptr;            // [F] bits of the largest ||b||^2 of the frame
    float2* dir_sc = nullptr;       // [F][cap] (sin, cos) of every keypoint's descriptor direction
    FrameCounters* cnt = nullptr;   // [F]
    int* cls_list = nullptr;        // [5][F*cap] keypoints (frame*cap + k) of each descriptor window class; list 4 = sequential-walk fallback
    int* cls_cnt = nullptr;         // [kClsCntInts] list lengths [0..4]; work-queue tickets at ticket_index(cls, segment)
    int* cls_hist = nullptr;        // [4][F*nbins] keypoints per (class, frame, 256-px tile): counting sort of the class lists
    int nbins() const { return ((W + 255) / 256) * ((H + 255) / 256); }
    size_t img_stride() const { return (size_t)H * W; }
    size_t sum_stride() const { return (size_t)(H + 1) * (W + 1); }
    int sum4_pitch() const { return ((W + 1 + 3) / 4 + 31) & ~31; }  // words per plane row: whole 128-byte lines
    size_t sum4_plane() const { return (size_t)(H + 1) * sum4_pitch(); }
    size_t sum4_stride() const { return 4 * sum4_plane(); }
};

struct PairSet {
    int P = 0, kp_cap = 0, match_cap = 0, max_hyp = 0;
    int n_split = 0;            // train-range splits of the MFMA pass
    int2* frames = nullptr;     // [P] (query frame, train frame) of every pair; default (2p, 2p+1)
    float* cand_val = nullptr;  // [P][kp_cap][32] shortlist scores (4 per group = lane half x train split)
    int* cand_idx = nullptr;
    int* knn_idx = nullptr;     // [P][kp_cap][2]
    int* redo_list = nullptr;   // [P*kp_cap] queries whose shortlist could not be certified (pair*cap + q)
    int* redo_cnt = nullptr;    // [1] length of redo_list (per pair set: matchers of two sets may run on two streams at once)
    float* knn_dist = nullptr;  // [P][kp_cap][2]
    float* p1 = nullptr;        // [P][match_cap][2]
    float* p2 = nullptr;
    vo_dmatch* matches = nullptr;  // [P][match_cap]
    double* m1 = nullptr;       // [P][match_cap][2] normalised
    double* m2 = nullptr;
    int* tuples = nullptr;      // [P][max_hyp][5]
    double* models = nullptr;   // [P][max_hyp][10][9]
    int* nmodels = nullptr;     // [P][max_hyp]
    int* counts = nullptr;      // [P][max_hyp][10]
    uint8_t* mask = nullptr;    // [P][match_cap]
    float* in1 = nullptr;       // [P][match_cap][2] inliers
    float* in2 = nullptr;
    float* X4 = nullptr;        // [P][4][match_cap]
    double* Xpre = nullptr;     // [P][match_cap][3] points passing the reprojection filter (frame k)
    double* Xk = nullptr;       // [P][3][match_cap]
    uint8_t* flag = nullptr;    // [P][match_cap] reprojection filter flags
    double* tmp = nullptr;      // [P][match_cap*4] scratch of the world-point filter
    PairState* st = nullptr;    // [P]
};

struct vo_ctx {
    vo_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    // Auxiliary stream, in order per run: [batch runs: the integral pass, which thereby runs under the PREVIOUS run's
    // descriptor kernels] -> Hessian octave 1 (beside the octave-0 grid) -> the footprint image (beside sort and orientation)
    // -> the largest descriptor class (beside the other three: its long keypoints leave a tail in which most CUs idle).
    // Forked / joined with the main stream by events; unused while the stage timers are on.
    hipStream_t stream_aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_quads = nullptr;     // the footprint image of the current detection is complete (auxiliary stream)
    hipEvent_t ev_integral = nullptr;  // the integral image of the current detection is complete (auxiliary stream, batch runs)
    hipStream_t stream_geo = nullptr;  // RANSAC / pose / triangulation of a batch run here, under the next run's detection
    DetectTables* d_tables = nullptr;
    void* d_win_cell = nullptr;     // INTER_AREA decimation tables of every window size (vo_detect.hip: CellTab [769][21 + 1])
    void* d_win_pix = nullptr;      // PixTab [769][768]
    // cv::RNG streams of the registrator, one device table per seed seen by THIS context (never shared between contexts:
    // a table is written once, before anything reads it, so streams of one context can use different seeds side by side)
    struct RngTable {
        uint64_t seed;
        uint32_t* d_raw;
    };
    std::vector<RngTable> rng_tables;
    int* d_signs = nullptr;         // 36 sign bits of JacobiSVD's completion vectors (vo_geometry.hip)
    unsigned char* d_noalloc = nullptr;  // this device's address of the static scratch array of vo_triang_points_noalloc
    bool profiling = false;
    int force_seq_describe = 0;     // vo_debug_set key 0
    int match_fp32 = 0;             // vo_debug_set key 2: shortlist on the FP32 MFMA (v_mfma_f32_32x32x2_f32) instead of split bf16
    int force_exact_match = 0;      // vo_debug_set key 1: no shortlist is certified, every query takes the exact scan
    int* d_redo_cnt = nullptr;      // redo_cnt of the pair set of the latest matcher launch (not owned; vo_debug_fetch 4 reads it)
    StageTimer timers[ST_COUNT];
    // lazily sized scratch for the single-call entry points
    FrameSet fs;
    PairSet ps;
    void* d_tmp = nullptr;
    size_t d_tmp_bytes = 0;
    void* d_gen = nullptr;          // det planes of the n_octave_layers != 3 detector (vo_detect.hip), grown on demand
    size_t d_gen_bytes = 0;
    // pinned bounce buffers of the synchronous entry points: frames go up and results come down through them (a copy
    // between the device and PAGEABLE memory makes the runtime pin the caller's pages for every call: ~4 ms per frame pair)
    uint8_t* h_up = nullptr;
    size_t h_up_bytes = 0;
    uint8_t* h_down = nullptr;
    size_t h_down_bytes = 0;
    struct vo_batch* pb = nullptr;  // pair batch of vo_pair_batch (vo_multi.hip), kept from call to call, owned by the context
};

// Feature buffers of a frame set that the matcher reads.  A batch owns two sets and alternates between them from run to run:
// the matcher of run k works on the geometry stream while the main stream is already detecting run k + 1 into the other set
// (the matcher is MFMA- and gather-bound, detection is VALU-bound: side by side they hide most of the matcher's time).
struct FeatBuf {
    vo_keypoint* kp = nullptr;
    float* desc = nullptr;
    float* descT = nullptr;
    float* nrm = nullptr;
    int* nmax = nullptr;
    FrameCounters* cnt = nullptr;
};

// Host-side memory of one run of a batch, pinned: what the run uploads (a snapshot of the pairs' K / initial pose and of the
// pair -> frame map, taken when vo_batch_run is called) and what it brings back.  A batch keeps two, used alternately, so
// that the host can prepare and enqueue run k + 1 while run k is executing, and read run k's records while k + 1 runs.
struct RunHost {
    PairState* init = nullptr;     // [n_pairs]
    int2* frames = nullptr;        // [n_pairs]
    PairState* res = nullptr;      // [n_pairs] device state after the run
    FrameCounters* cnt = nullptr;  // [n_frames]
    hipEvent_t ev_state = nullptr; // the upload of init / frames has finished (geometry stream)
    hipEvent_t ev_done = nullptr;  // matcher and geometry finished, res / cnt are on the host (geometry stream)
    bool state_pending = false;
    std::vector<uint8_t> unset;    // [n_pairs] 1 = the pair had no camera matrix when the run was enqueued (pageable)
};

struct vo_batch {
    vo_ctx* ctx = nullptr;
    int n_pairs = 0, rows = 0, cols = 0;
    int n_frames = 0;                // frames held (2 * n_pairs for a pair batch, n_pairs + 1 for a sequence)
    bool is_seq = false;
    FrameSet fs;  // pair batch: frame 2p = img1 of pair p, 2p+1 = img2.  fs.img points at the frame slot the current run reads
    PairSet ps;
    std::vector<PairState> h_init;   // K and initial pose of every pair as the setters left them (pageable; snapshot per run)
    std::vector<int2> h_frames;      // pair -> (query frame, train frame)
    // Frame slots: the frames exist twice on the device (and in pinned staging memory).  The setters write the slot the
    // latest run does NOT read and upload it on the copy stream, so the transfer of the next step's frames runs beside the
    // kernels of the current one; vo_batch_run flips the slots when anything was staged.
    uint8_t* d_img[2] = {nullptr, nullptr};
    uint8_t* h_stage[2] = {nullptr, nullptr};
    hipStream_t stream_copy = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};       // every upload into the slot issued so far has finished (copy stream)
    hipEvent_t ev_img_free[2] = {nullptr, nullptr};  // the run that read the slot is past detection + description (main stream)
    bool h2d_pending[2] = {false, false}, img_free_pending[2] = {false, false};
    // A slot that received frames from vo_prep_to_batch was written by kernels on the context's MAIN stream: the copy stream
    // has to wait for them before it carries frames of that slot over into the other one (ev_prep, recorded by the run
    // that first reads the slot).
    hipEvent_t ev_prep[2] = {nullptr, nullptr};
    bool prep_pending[2] = {false, false};
    int cur_slot = -1;               // slot the latest run read (-1: no run yet)
    int fill_slot = 0;               // slot the setters write
    int n_staged = 0;                // distinct frames staged into fill_slot since the latest run
    std::vector<uint8_t> staged;     // [n_frames]
    RunHost rh[2];
    int64_t runs = 0;                // runs enqueued so far; run k uses rh[k & 1]
    hipEvent_t ev_detect = nullptr;  // detection + description of the latest run finished (main stream)
    FeatBuf feat[2];                 // run k detects into feat[k & 1]; feat[0] is the frame set's own allocation
    uint8_t* h_bulk = nullptr;       // pinned landing buffer of vo_batch_get_points_all, grown on demand
    size_t h_bulk_bytes = 0;
};

// ---------------------------------------------------------------- helpers (vo_api.cpp)
int vo_frameset_alloc(FrameSet& fs, int F, int H, int W, int cap, int dlen = 64);
inline int vo_desc_len(const vo_ctx* ctx) { return ctx->prm.extended ? 128 : 64; }
void vo_frameset_free(FrameSet& fs);
int vo_pairset_alloc(PairSet& ps, int P, int kp_cap, int match_cap, int max_hyp);
void vo_pairset_free(PairSet& ps);
int vo_ctx_tmp(vo_ctx* ctx, size_t bytes, void** out);
// device address of frame `frame` in the slot the NEXT vo_batch_run will read; marks the frame as staged (vo_prep_to_batch
// writes there with kernels on the context's main stream)
int vo_batch_frame_slot(vo_batch* b, int frame, uint8_t** d_dst);

struct StageScope {  // records HIP events around a stage when profiling is on
    vo_ctx* ctx;
    int stage;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(vo_ctx* c, int s, hipStream_t on = nullptr);  // on = nullptr: the context's main stream
    ~StageScope();
};

// ---------------------------------------------------------------- kernels' host launchers
// vo_detect.hip
void vo_build_detect_tables(DetectTables& t, int sum_cols_unused);
int vo_enqueue_integral(vo_ctx* ctx, FrameSet& fs, hipStream_t stream);
int vo_enqueue_hessian_layer(vo_ctx* ctx, FrameSet& fs, int frame, int octave, int layer, float* d_det, float* d_trace);
int vo_enqueue_detect(vo_ctx* ctx, FrameSet& fs, bool describe, bool integral_beside = false, hipEvent_t img_ready = nullptr);
int vo_enqueue_hessian_nms(vo_ctx* ctx, FrameSet& fs, bool describe, float* dbg_det);
// vo_match.hip
// every launch goes to `stream` (the main stream for the single-call entry points, the geometry stream for a batch)
int vo_enqueue_match(vo_ctx* ctx, FrameSet& fs, PairSet& ps, bool filter, hipStream_t stream);
// vo_geometry.hip
struct GeoOptions {
    bool run_ransac = true, run_pose = true, run_triang = true;
    bool gate = true;  // optRelativePose gates (inlier ratio, VPF); false = plain recoverPose
    double prob = 0.99, thr_px = 2.0, dist_thresh = 50.0;
    int max_iters = 1000;
    uint64_t seed = (uint64_t)-1;
};
// every launch goes to `stream` (the context's main stream for the single-call entry points, its geometry stream for a batch)
int vo_enqueue_geometry(vo_ctx* ctx, PairSet& ps, const GeoOptions& o, hipStream_t stream);
int vo_enqueue_five_point(vo_ctx* ctx, const double* d_x1, const double* d_x2, int n_hyp, const int* d_tuples,
                          double* d_E, int* d_nm);
int vo_enqueue_sampson_masks(vo_ctx* ctx, const double* d_m1, const double* d_m2, int n, float thr_sq, const double* d_E,
                             int n_models, uint8_t* d_masks, int* d_counts);
int vo_enqueue_triangulate(vo_ctx* ctx, const double* d_P01, const float* d_p1, const float* d_p2, int n, float* d_X4);
float vo_thr_sq(double thr_px, const double K[9]);
int vo_enqueue_normalise(vo_ctx* ctx, PairSet& ps, double thr_px);
int vo_upload_detect_tables(const DetectTables& t);
int vo_build_window_tables(vo_ctx* ctx);
int vo_geometry_init(vo_ctx* ctx);
void vo_geometry_release(vo_ctx* ctx);
int vo_detect_init(vo_ctx* ctx);  // per-device kernel attributes (called with the context's device current)
