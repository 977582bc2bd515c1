// Brute-force L2 2-NN matching + Lowe ratio / ROI filter for gfx950.
//
// Replaces matcher->knnMatch(desc1, desc2, knn, 2) and the filter/gather loops of
// opt_DetectFeatures (reference: include/visual_odometry/camera_utility.hpp:759-808; the
// reference's FLANN matcher is replaced by the exact brute-force search of its "OLD METHOD"
// comment, :752-755, as the north star asks).
//
//   k_match_prep   : train descriptors -> MFMA operand layout, plus -0.5*||b||^2
//   k_match_mfma   : dense contraction a.b - 0.5||b||^2 on v_mfma_f32_32x32x2_f32 with queries on
//                    the lanes (columns) and train rows in the accumulator registers, so the
//                    running top-2 per query is pure in-lane VALU work; the train range is split
//                    8 ways over wavefronts, each keeps a top-2 per lane half -> 32 candidates
//   k_match_select : exact re-scoring of the candidates with the oracle's sequential
//                    sum (a-b)^2, 32-lane (dist, index) min-reductions -> knn[2]
//   k_match_filter : d0 < 0.7 d1 and ROI on keypoint 1, ordered ballot compaction, point gather
// The MFMA value is only used to shortlist; the distances that reach the ratio test are exact.
#include <cfloat>

#include "vo_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSplit = 8;            // train-range splits
constexpr int kCand = 4 * kSplit;    // candidates per query (top-2 per lane half per split)

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_match_prep(const float* __restrict__ desc_base, const FrameCounters* __restrict__ cnt,
                                                   int cap, float4* __restrict__ descT_base, float* __restrict__ nrm_base) {
    const int frame = blockIdx.y, tile = blockIdx.x;
    const int n = min(cnt[frame].n_raw, cap);
    if (tile * 32 >= n) return;
    const int lane = threadIdx.x, h = lane >> 5;
    const int r = tile * 32 + (lane & 31);
    const float4* row = reinterpret_cast<const float4*>(desc_base + ((size_t)frame * cap + r) * 64);
    float4* out = descT_base + ((size_t)frame * (cap / 32) + tile) * 8 * 64;
    float ss = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = row[8 * h + s4];
        out[s4 * 64 + lane] = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss += __shfl_xor(ss, 32, 64);
    if (h == 0) nrm_base[(size_t)frame * cap + r] = (r < n) ? -0.5f * ss : -1e30f;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void top2_update(float v, int idx, float& v1, int& i1, float& v2, int& i2) {
    const bool g1 = v > v1, g2 = v > v2;
    v2 = g1 ? v1 : (g2 ? v : v2);
    i2 = g1 ? i1 : (g2 ? idx : i2);
    v1 = g1 ? v : v1;
    i1 = g1 ? idx : i1;
}

template <int NC>  // 32-query blocks per wavefront
__global__ __launch_bounds__(256) void k_match_mfma(const float* __restrict__ desc_base,
                                                    const float4* __restrict__ descT_base,
                                                    const float* __restrict__ nrm_base,
                                                    const FrameCounters* __restrict__ cnt, int cap,
                                                    const int2* __restrict__ pair_frames,
                                                    float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    const int pair = blockIdx.y;
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap), nB = min(cnt[fb].n_raw, cap);
    const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int qb = w / kSplit, sp = w - qb * kSplit;
    const int q0 = qb * 32 * NC;
    if (q0 >= nA) return;
    const int ntiles = (nB + 31) >> 5;

    // query operands: lane holds query[q0 + 32c + l31][32h + s], s = 0..31 (k is permuted
    // identically on both operands: lane half h covers k in [32h, 32h+32))
    float bq[NC][32];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int row = q0 + 32 * c + l31;
        const float4* p = reinterpret_cast<const float4*>(desc_base + ((size_t)fa * cap + row) * 64 + 32 * h);
#pragma unroll
        for (int s4 = 0; s4 < 8; s4++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nA) v = p[s4];
            bq[c][4 * s4 + 0] = v.x;
            bq[c][4 * s4 + 1] = v.y;
            bq[c][4 * s4 + 2] = v.z;
            bq[c][4 * s4 + 3] = v.w;
        }
    }
    float v1[NC], v2[NC];
    int i1[NC], i2[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        v1[c] = v2[c] = -FLT_MAX;
        i1[c] = i2[c] = -1;
    }
    const float4* __restrict__ descT = descT_base + (size_t)fb * (cap / 32) * 8 * 64;
    const float* __restrict__ nrm = nrm_base + (size_t)fb * cap;

    // Operands of a tile come in two halves of four float4 per lane; each half is re-loaded for the NEXT tile right after
    // its MFMAs were issued, so the loads fly under the other half's MFMAs and the top-2 update (register budget:
    // 2 x 16 operand registers instead of 32 + 32 for a whole-tile double buffer).
    float4 ha[4], hb[4];
    float an_next = 0.f;
    if (sp < ntiles) {
        const float4* tp = descT + (size_t)sp * 8 * 64 + lane;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            ha[s4] = tp[s4 * 64];
            hb[s4] = tp[(s4 + 4) * 64];
        }
        an_next = (h == 0) ? nrm[sp * 32 + l31] : 0.f;
    }
    for (int t = sp; t < ntiles; t += kSplit) {
        const float an = an_next;
        const bool more = t + kSplit < ntiles;
        const float4* tn = descT + (size_t)(more ? t + kSplit : t) * 8 * 64 + lane;
        f32x16 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
#define VO_MFMA4(A, S4)                                                                                       \
    _Pragma("unroll") for (int c = 0; c < NC; c++)                                                            \
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, bq[c][4 * (S4) + 0], acc[c], 0, 0, 0);             \
    _Pragma("unroll") for (int c = 0; c < NC; c++)                                                            \
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, bq[c][4 * (S4) + 1], acc[c], 0, 0, 0);             \
    _Pragma("unroll") for (int c = 0; c < NC; c++)                                                            \
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, bq[c][4 * (S4) + 2], acc[c], 0, 0, 0);             \
    _Pragma("unroll") for (int c = 0; c < NC; c++)                                                            \
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, bq[c][4 * (S4) + 3], acc[c], 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { VO_MFMA4(ha[s4], s4) }
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) ha[s4] = tn[s4 * 64];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { VO_MFMA4(hb[s4], s4 + 4) }
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) hb[s4] = tn[(s4 + 4) * 64];
#undef VO_MFMA4
        an_next = (h == 0) ? nrm[(more ? t + kSplit : t) * 32 + l31] : 0.f;
        // one more k-step adds -0.5*||b_i||^2 to every column of row i
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(an, 1.0f, acc[c], 0, 0, 0);
        // accumulator register r of lane (h, l31) is train row (r&3) + 8*(r>>2) + 4h of the tile
        const int ibase = t * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int idx = ibase + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int c = 0; c < NC; c++) top2_update(acc[c][r], idx, v1[c], i1[c], v2[c], i2[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int row = q0 + 32 * c + l31;
        if (row < nA) {
            const size_t o = ((size_t)pair * cap + row) * kCand + sp * 4 + 2 * h;
            // entries whose value never left the sentinel (padding rows) are not candidates
            cand_idx[o] = (v1[c] > -1e29f) ? i1[c] : -1;
            cand_idx[o + 1] = (v2[c] > -1e29f) ? i2[c] : -1;
            cand_val[o] = v1[c];
            cand_val[o + 1] = v2[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void min_pair32(float& d, int& i) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const float od = __shfl_xor(d, m, 32);
        const int oi = __shfl_xor(i, m, 32);
        if (od < d || (od == d && oi < i)) {
            d = od;
            i = oi;
        }
    }
}

__global__ __launch_bounds__(256) void k_match_select(const float* __restrict__ desc_base,
                                                      const FrameCounters* __restrict__ cnt, int cap,
                                                      const int2* __restrict__ pair_frames,
                                                      const int* __restrict__ cand_idx, int* __restrict__ knn_idx,
                                                      float* __restrict__ knn_dist) {
    const int pair = blockIdx.y;
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap), nB = min(cnt[fb].n_raw, cap);
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), sub = threadIdx.x & 31;
    if (q >= nA) return;
    int idx = (nB > 0) ? cand_idx[((size_t)pair * cap + q) * kCand + sub] : -1;
    float d = FLT_MAX;
    if (idx >= 0) {
        const float* a = desc_base + ((size_t)fa * cap + q) * 64;
        const float* b = desc_base + ((size_t)fb * cap + idx) * 64;
        float s = 0.f;
#pragma unroll 16
        for (int k = 0; k < 64; k++) {  // sequential float accumulation: the oracle's normL2Sqr (A.7)
            const float t = a[k] - b[k];
            s += t * t;
        }
        d = s;
    } else {
        idx = 0x7fffffff;
    }
    float d0 = d;
    int i0 = idx;
    min_pair32(d0, i0);
    float d1 = (idx == i0) ? FLT_MAX : d;
    int i1 = (idx == i0) ? 0x7fffffff : idx;
    min_pair32(d1, i1);
    if (sub == 0) {
        const size_t o = ((size_t)pair * cap + q) * 2;
        const bool h0 = i0 != 0x7fffffff, h1 = i1 != 0x7fffffff;
        knn_idx[o] = h0 ? i0 : -1;
        knn_idx[o + 1] = h1 ? i1 : -1;
        knn_dist[o] = h0 ? sqrtf(d0) : FLT_MAX;
        knn_dist[o + 1] = h1 ? sqrtf(d1) : FLT_MAX;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_match_filter(const vo_keypoint* __restrict__ kp_base,
                                                       const FrameCounters* __restrict__ cnt, int cap,
                                                       const int2* __restrict__ pair_frames,
                                                       const int* __restrict__ knn_idx, const float* __restrict__ knn_dist,
                                                       float ratio, float x_lo, float y_lo, float x_hi, float y_hi,
                                                       int match_cap, float* __restrict__ p1, float* __restrict__ p2,
                                                       vo_dmatch* __restrict__ matches, PairState* __restrict__ st) {
    __shared__ int s_wtot[16];
    __shared__ int s_base;
    const int pair = blockIdx.x;
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const vo_keypoint* kA = kp_base + (size_t)fa * cap;
    const vo_keypoint* kB = kp_base + (size_t)fb * cap;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < nA; base += 1024) {
        const int q = base + tid;
        bool pass = false;
        int t0 = -1;
        float d0 = 0.f;
        float ax = 0.f, ay = 0.f;
        if (q < nA) {
            const size_t o = ((size_t)pair * cap + q) * 2;
            t0 = knn_idx[o];
            const int t1 = knn_idx[o + 1];
            d0 = knn_dist[o];
            const float d1 = knn_dist[o + 1];
            ax = kA[q].x;
            ay = kA[q].y;
            const bool lowe = (t0 >= 0) && (t1 >= 0) && (d0 < ratio * d1);            // :776
            const bool roi = (ax >= x_lo) && (ax <= x_hi) && (ay >= y_lo) && (ay <= y_hi);  // :777
            pass = lowe && roi;
        }
        const unsigned long long m = __ballot(pass);
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        if (pass) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < match_cap) {
                const size_t o = (size_t)pair * match_cap + pos;
                vo_dmatch dm;
                dm.queryIdx = q;
                dm.trainIdx = t0;
                dm.imgIdx = 0;
                dm.distance = d0;
                matches[o] = dm;
                p1[2 * o] = ax;
                p1[2 * o + 1] = ay;
                p2[2 * o] = kB[t0].x;
                p2[2 * o + 1] = kB[t0].y;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wtot[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        st[pair].n_match = s_base;
        st[pair].n_used = min(s_base, match_cap);
    }
}

// ------------------------------------------------------------------------------------------------
int vo_enqueue_match(vo_ctx* ctx, FrameSet& fs, PairSet& ps, bool filter) {
    const int cap = fs.cap, P = ps.P;
    {
        StageScope sc(ctx, ST_MATCH_PREP);
        hipLaunchKernelGGL(k_match_prep, dim3(cap / 32, fs.F), dim3(64), 0, ctx->stream, fs.desc, fs.cnt, cap,
                           reinterpret_cast<float4*>(fs.descT), fs.nrm);
        VO_HIP(hipGetLastError());
    }
    {
        StageScope sc(ctx, ST_MATCH_MFMA);
        constexpr int kQueryBlocks = 2;  // 64 queries per wavefront (1 measured slower: 5.65 vs 5.25 ms)
        const int waves = (cap / (32 * kQueryBlocks)) * kSplit;
        hipLaunchKernelGGL(k_match_mfma<kQueryBlocks>, dim3(waves / 4, P), dim3(256), 0, ctx->stream, fs.desc,
                           reinterpret_cast<const float4*>(fs.descT), fs.nrm, fs.cnt, cap, ps.frames, ps.cand_val, ps.cand_idx);
        VO_HIP(hipGetLastError());
    }
    {
        StageScope sc(ctx, ST_MATCH_SELECT);
        hipLaunchKernelGGL(k_match_select, dim3(cap / 8, P), dim3(256), 0, ctx->stream, fs.desc, fs.cnt, cap, ps.frames,
                           ps.cand_idx, ps.knn_idx, ps.knn_dist);
        if (filter) {
            const vo_params& p = ctx->prm;
            hipLaunchKernelGGL(k_match_filter, dim3(P), dim3(1024), 0, ctx->stream, fs.kp, fs.cnt, cap, ps.frames,
                               ps.knn_idx, ps.knn_dist, p.ratio_thresh, p.roi_x_lo, p.roi_y_lo, p.roi_x_hi, p.roi_y_hi, ps.match_cap,
                               ps.p1, ps.p2, ps.matches, ps.st);
        }
        VO_HIP(hipGetLastError());
    }
    return VO_OK;
}
