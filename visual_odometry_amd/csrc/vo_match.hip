// Brute-force L2 2-NN matching + Lowe ratio / ROI filter for gfx950.
//
// Replaces matcher->knnMatch(desc1, desc2, knn, 2) and the filter/gather loops of
// opt_DetectFeatures (reference: include/visual_odometry/camera_utility.hpp:759-808; the
// reference's FLANN matcher is replaced by the exact brute-force search of its "OLD METHOD"
// comment, :752-755, as the north star asks).
//
//   k_match_prep   : train descriptors -> MFMA operand layout (bf16 high / low parts), -0.5*||b||^2, max ||b||^2
//   k_match_bf3    : dense contraction a.b - 0.5||b||^2 on v_mfma_f32_32x32x16_bf16 with each float split into two bf16
//                    (a = ah + al): ah.bh + ah.bl + al.bh, 12 MFMAs of 32 cycles per 32 x 32 tile instead of 32 FP32
//                    MFMAs of 64 cycles, error <= 3 * 2^-16 |a||b|.  Queries on the lanes (columns), train rows in the
//                    accumulator registers, so the shortlist (the three best 4-row blocks per group and a bound on the
//                    rest) is in-lane VALU work: max, med3, compare + branch per block, insertion only on a hit
//   k_match_select : exact re-scoring of the candidates with the oracle's sequential sum (a-b)^2, (dist, index)
//                    min-reductions -> knn[2]; CERTIFIES the shortlist: every train row left out of a group scored no
//                    better than the group's bound t4, so if ||a||^2 - 2*t4 exceeds the second-best exact
//                    distance by more than the error bound for every group, no left-out row can be a neighbour.
//                    Queries that fail the test go to a list
//   k_match_exact  : exact scan of all train rows for the listed queries (a handful per frame pair)
//   k_match_filter : d0 < 0.7 d1 and ROI on keypoint 1, ordered ballot compaction, point gather
// The MFMA value is only used to shortlist; the distances that reach the ratio test are exact.
#include <cfloat>

#include "vo_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));


// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bf16_rne(float f) {  // round to nearest even, finite inputs
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// a = hi + lo + r with hi, lo bf16 (8 significand bits, round to nearest even): |a - hi| <= 2^-8 |a|, |r| <= 2^-16 |a|
__device__ __forceinline__ void bf16_split(float a, uint32_t& hi, uint32_t& lo) {
    hi = bf16_rne(a);
    lo = bf16_rne(a - __uint_as_float(hi << 16));  // the subtraction is exact
}
// eight consecutive floats -> their high parts and low parts, packed for one MFMA operand (element j in half-word j)
__device__ __forceinline__ void bf16_split8(const float4& v0, const float4& v1, uint4& hi, uint4& lo) {
    uint32_t h[8], l[8];
    bf16_split(v0.x, h[0], l[0]);
    bf16_split(v0.y, h[1], l[1]);
    bf16_split(v0.z, h[2], l[2]);
    bf16_split(v0.w, h[3], l[3]);
    bf16_split(v1.x, h[4], l[4]);
    bf16_split(v1.y, h[5], l[5]);
    bf16_split(v1.z, h[6], l[6]);
    bf16_split(v1.w, h[7], l[7]);
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

constexpr int kSlabs = 9;
// Operand layout of a 32-row tile: 9 slabs of 64 lanes x 16 bytes.  Lane (h, r) of slab m (0..3) holds the high parts of
// row r, elements 32h + 8m .. + 7; slab 4 + m the low parts; slab 8 the norm term (below).
// F32: the layout of the FP32-MFMA variant instead (slab s4 = floats 32h + 4 s4 .. + 3 of row r, slab 8 = the norm as one float).
template <bool F32>
__global__ __launch_bounds__(64) void k_match_prep(const float* __restrict__ desc_base, const FrameCounters* __restrict__ cnt,
                                                   int cap, float4* __restrict__ descT_base, float* __restrict__ nrm_base,
                                                   int* __restrict__ nmax_bits) {
    const int frame = blockIdx.y, tile = blockIdx.x;
    const int n = min(cnt[frame].n_raw, cap);
    if (tile * 32 >= n) return;
    const int lane = threadIdx.x, h = lane >> 5;
    const int r = tile * 32 + (lane & 31);
    float4* out = descT_base + ((size_t)frame * (cap / 32) + tile) * kSlabs * 64;
    // The tile's 8 KB are read as eight fully coalesced 1 KB loads and turned row-per-lane through LDS (a lane reading its own
    // row directly touches 64 cache lines per load instruction: 0.38 ms per 128 frames, bound by the texture addresser).
    __shared__ float4 s_t[32][17];  // 272-byte pitch: the rows of a 16-lane phase fall into different banks
    const float4* tile_base = reinterpret_cast<const float4*>(desc_base + ((size_t)frame * cap + (size_t)tile * 32) * 64);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = i * 64 + lane, rr = idx >> 4, c4 = idx & 15;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tile * 32 + rr < n) t = tile_base[idx];
        s_t[rr][c4] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float ss = 0.f;
    float4 v[8];
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        v[s4] = s_t[lane & 31][8 * h + s4];
        ss += v[s4].x * v[s4].x + v[s4].y * v[s4].y + v[s4].z * v[s4].z + v[s4].w * v[s4].w;
    }
    if (F32) {
#pragma unroll
        for (int s4 = 0; s4 < 8; s4++) out[s4 * 64 + lane] = v[s4];
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            uint4 hi, lo;
            bf16_split8(v[2 * m], v[2 * m + 1], hi, lo);
            reinterpret_cast<uint4*>(out)[m * 64 + lane] = hi;
            reinterpret_cast<uint4*>(out)[(4 + m) * 64 + lane] = lo;
        }
    }
    ss += __shfl_xor(ss, 32, 64);
    const float nb = (r < n) ? -0.5f * ss : -1e30f;
    if (h == 0) nrm_base[(size_t)frame * cap + r] = nb;
    if (F32) {
        out[8 * 64 + lane] = make_float4(h == 0 ? nb : 0.f, 0.f, 0.f, 0.f);
    } else {  // slab 8: -0.5||b||^2 as three bf16 (24 bits) in k slots 0..2 of lane half 0; the query side holds ones there
        uint32_t p0, p1, p2, rest;
        bf16_split(nb, p0, p1);
        bf16_split(nb - __uint_as_float(p0 << 16) - __uint_as_float(p1 << 16), p2, rest);
        reinterpret_cast<uint4*>(out)[8 * 64 + lane] = (h == 0) ? make_uint4(p0 | (p1 << 16), p2, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
    // largest squared norm of the frame (non-negative floats order like their bit patterns): scales the certificate's margin
    float mx = (r < n) ? ss : 0.f;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if (lane == 0) atomicMax(&nmax_bits[frame], __float_as_int(mx));
}

// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Bf16Frag {
    uint4 u;
    bf16x8 v;
};

// Shortlist bookkeeping.  A lane sees 16 scores per tile: 4 BLOCKS of 4 consecutive train rows (accumulator registers
// 4j .. 4j+3 = tile rows 8j + 4h ..+3).  Per group (lane half x train split) it keeps the three blocks with the largest
// maximum, and t4 = the largest maximum among all other blocks.  The two best rows of the group always lie in its two best
// blocks (the runner-up row is either in the winner's block or is the maximum of its own block), so the rows of the kept
// blocks are a complete shortlist as far as the scores can be trusted, and t4 bounds every row that was left out.
// Cost per block: v_max3 + v_max, one v_med3 (t4 = med3(m, b3, t4) is "max of everything not kept" without a branch) and
// one compare + branch; the 13-instruction insertion runs only when some lane's block beats its third best.
__device__ __forceinline__ void top3_insert(float m, int id, float (&bv)[3], int (&bi)[3]) {
    const bool g0 = m > bv[0], g1 = m > bv[1], g2 = m > bv[2];
    bv[2] = g1 ? bv[1] : (g2 ? m : bv[2]);
    bi[2] = g1 ? bi[1] : (g2 ? id : bi[2]);
    bv[1] = g0 ? bv[0] : (g1 ? m : bv[1]);
    bi[1] = g0 ? bi[0] : (g1 ? id : bi[1]);
    bv[0] = g0 ? m : bv[0];
    bi[0] = g0 ? id : bi[0];
}

__device__ __forceinline__ float max4(float a, float b, float c, float d) {  // finite inputs: no NaN canonicalisation wanted
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(d));
    return m;
}

// max4 reads accumulator registers from inline assembly, which the compiler's hazard recogniser does not look into: an
// MFMA result must not be read by a VALU instruction earlier than 19 wait states after a 16-pass MFMA was issued (the
// hardware does not interlock).  Inside the pipeline every check is separated from the MFMAs that produced its tile by a
// barrier and two later MFMAs on the in-order matrix pipe; the checks that follow the LAST MFMAs directly wait here first.
template <int NC>
__device__ __forceinline__ void mfma_settle(f32x16 (&acc)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; c++) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[c]));
}

constexpr int kKeep = 3;         // blocks kept per group
constexpr int kCandStride = 32;  // slots per query in cand_val / cand_idx: 8 groups x (3 block ids | t4)

// One shortlist check: block j (registers 4j .. 4j+3) of query block c of a finished accumulator tile
template <int NC>
__device__ __forceinline__ void bf3_check(const f32x16 (&acc)[NC], int c, int j, int tile, float (&bv)[NC][kKeep], int (&bi)[NC][kKeep],
                                          float (&t4)[NC]) {
    const float m = max4(acc[c][4 * j], acc[c][4 * j + 1], acc[c][4 * j + 2], acc[c][4 * j + 3]);
    t4[c] = __builtin_amdgcn_fmed3f(m, bv[c][kKeep - 1], t4[c]);
    if (__any(m > bv[c][kKeep - 1])) top3_insert(m, 4 * tile + j, bv[c], bi[c]);  // rarer and rarer as the scan goes on
}

// One step of the software pipeline: the 1 + 12 NC MFMAs of the tile staged in LDS (st) go into `an`, and BETWEEN them
// (an MFMA occupies the matrix pipe for 32 cycles, the wavefront is free to issue ~7 VALU instructions meanwhile) the
// shortlist checks of the PREVIOUS tile's accumulators `ao`.  Operand slabs are read from LDS one MFMA group ahead.
template <int NC, bool CHECK>
__device__ __forceinline__ void bf3_step(f32x16 (&an)[NC], const f32x16 (&ao)[NC], const uint4* st, const Bf16Frag (&qh)[NC][4],
                                         const Bf16Frag (&ql)[NC][4], int tile_prev, float (&bv)[NC][kKeep], int (&bi)[NC][kKeep],
                                         float (&t4)[NC]) {
    static_assert(NC <= 3, "the checks of a block row are spread over the three partial products");
    Bf16Frag ones, tnrm, th[2], tl[2];
    ones.u = make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u);  // bf16 1, 1, 1, 0 ...: picks the three norm parts of slab 8
    tnrm.u = st[8 * 64];
    th[0].u = st[0];
    tl[0].u = st[4 * 64];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the norm term first (accumulator input = literal zero: no register initialisation); both query blocks start from it
    const f32x16 nterm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tnrm.v, ones.v, zero, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if (m < 3) {
            th[(m + 1) & 1].u = st[(m + 1) * 64];
            tl[(m + 1) & 1].u = st[(5 + m) * 64];
        }
#pragma unroll
        for (int part = 0; part < 3; part++) {  // ah.bh, ah.bl (query low), al.bh (train low)
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const bf16x8 a = part == 2 ? tl[m & 1].v : th[m & 1].v, b = part == 1 ? ql[c][m].v : qh[c][m].v;
                an[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, (m == 0 && part == 0) ? nterm : an[c], 0, 0, 0);
            }
            if (CHECK && part < NC) bf3_check<NC>(ao, part, m, tile_prev, bv, bi, t4);
        }
    }
}

// The same step on v_mfma_f32_32x32x2_f32 with FP32 operands (the north star's formulation; vo_debug_set key 2): 32 MFMAs of
// 64 cycles per 32 x 32 tile plus the norm step.  Lane (h, r) supplies element 32h + 4 s4 + e of its row to MFMA (s4, e).
template <int NC, bool CHECK>
__device__ __forceinline__ void f32_step(f32x16 (&an)[NC], const f32x16 (&ao)[NC], const uint4* st, const float (&q)[NC][32], int tile_prev,
                                         float (&bv)[NC][kKeep], int (&bi)[NC][kKeep], float (&t4)[NC]) {
    const float4* sf = reinterpret_cast<const float4*>(st);
    float4 t[2];
    t[0] = sf[0];
    const float an_norm = sf[8 * 64].x;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 nterm = __builtin_amdgcn_mfma_f32_32x32x2f32(an_norm, 1.0f, zero, 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        if (s4 < 7) t[(s4 + 1) & 1] = sf[(s4 + 1) * 64];
        const float4 a = t[s4 & 1];
#pragma unroll
        for (int c = 0; c < NC; c++) an[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[c][4 * s4 + 0], s4 == 0 ? nterm : an[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; c++) an[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[c][4 * s4 + 1], an[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; c++) an[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[c][4 * s4 + 2], an[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NC; c++) an[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[c][4 * s4 + 3], an[c], 0, 0, 0);
        if (CHECK) {  // the 4 NC checks of the previous tile, spread over the eight MFMA groups
#pragma unroll
            for (int idx = s4 * 4 * NC / 8; idx < (s4 + 1) * 4 * NC / 8; idx++) bf3_check<NC>(ao, idx % NC, idx / NC, tile_prev, bv, bi, t4);
        }
    }
}

// query operands of one wavefront in the form its MFMA variant takes them
template <int NC, bool F32>
struct QueryOps;
template <int NC>
struct QueryOps<NC, false> {
    Bf16Frag qh[NC][4], ql[NC][4];  // elements 32h + 8m .. + 7 of the lane's query row for MFMA m, high and low parts
    __device__ __forceinline__ void load(const float* __restrict__ desc_base, int fa, int cap, int q0, int l31, int h, int nA) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int row = q0 + 32 * c + l31;
            const float4* p = reinterpret_cast<const float4*>(desc_base + ((size_t)fa * cap + row) * 64 + 32 * h);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (row < nA) {
                    v0 = p[2 * m];
                    v1 = p[2 * m + 1];
                }
                bf16_split8(v0, v1, qh[c][m].u, ql[c][m].u);
            }
        }
    }
    template <bool CHECK>
    __device__ __forceinline__ void step(f32x16 (&an)[NC], const f32x16 (&ao)[NC], const uint4* st, int tile_prev, float (&bv)[NC][kKeep],
                                         int (&bi)[NC][kKeep], float (&t4)[NC]) const {
        bf3_step<NC, CHECK>(an, ao, st, qh, ql, tile_prev, bv, bi, t4);
    }
};
template <int NC>
struct QueryOps<NC, true> {
    float q[NC][32];  // elements 32h .. 32h + 31 of the lane's query row
    __device__ __forceinline__ void load(const float* __restrict__ desc_base, int fa, int cap, int q0, int l31, int h, int nA) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int row = q0 + 32 * c + l31;
            const float4* p = reinterpret_cast<const float4*>(desc_base + ((size_t)fa * cap + row) * 64 + 32 * h);
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nA) v = p[s4];
                q[c][4 * s4 + 0] = v.x, q[c][4 * s4 + 1] = v.y, q[c][4 * s4 + 2] = v.z, q[c][4 * s4 + 3] = v.w;
            }
        }
    }
    template <bool CHECK>
    __device__ __forceinline__ void step(f32x16 (&an)[NC], const f32x16 (&ao)[NC], const uint4* st, int tile_prev, float (&bv)[NC][kKeep],
                                         int (&bi)[NC][kKeep], float (&t4)[NC]) const {
        f32_step<NC, CHECK>(an, ao, st, q, tile_prev, bv, bi, t4);
    }
};

template <int NC, bool F32>  // 32-query blocks per wavefront; F32: FP32 MFMA instead of split bf16
__global__ __launch_bounds__(256) void k_match_bf3(const float* __restrict__ desc_base, const uint4* __restrict__ descT_base,
                                                   const float* __restrict__ nrm_base, const FrameCounters* __restrict__ cnt, int cap,
                                                   const int2* __restrict__ pair_frames, int ksplit, int bpp, int xcd_map,
                                                   float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    // Workgroups are dealt to the 8 XCDs round robin (linear id % 8).  All workgroups of a pair scan the same 2 MB of train
    // operands, so a pair is kept on ONE XCD (pair = xcd + 8 * ...): its L2 then streams two pairs at a time instead of
    // every pair in flight, and HBM delivers each operand tile once.
    int pair, bx;
    if (xcd_map) {
        const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
        pair = xcd + 8 * (k / bpp);
        bx = k - (k / bpp) * bpp;
    } else {
        pair = blockIdx.x / bpp;
        bx = blockIdx.x - pair * bpp;
    }
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap), nB = min(cnt[fb].n_raw, cap);
    const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31, wave = threadIdx.x >> 6;
    // the four wavefronts of a workgroup take four consecutive query blocks and the SAME train split: they share every
    // operand tile through LDS
    const int qgroup = bx / ksplit, sp = bx - qgroup * ksplit;
    const int q0 = (qgroup * 4 + wave) * 32 * NC;
    if (qgroup * 4 * 32 * NC >= nA) return;  // no query in the whole workgroup (wavefronts past nA still help with the staging)
    const int ntiles = (nB + 31) >> 5;

    QueryOps<NC, F32> Q;
    Q.load(desc_base, fa, cap, q0, l31, h, nA);
    float bv[NC][kKeep], t4[NC];
    int bi[NC][kKeep];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        t4[c] = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < kKeep; k++) {
            bv[c][k] = -FLT_MAX;
            bi[c][k] = -1;
        }
    }
    const uint4* __restrict__ descT = descT_base + (size_t)fb * (cap / 32) * kSlabs * 64;

    // Operand tiles (9 KB) go global -> registers -> LDS, each wavefront staging a quarter (slabs w and w + 4; wavefront 0
    // also the norm slab), double-buffered: the loads of tile i + 1 are issued before the MFMAs of tile i and written to the
    // other buffer after them, one barrier per step.  (Straight from L1 each wavefront pulled the whole 9 KB per step:
    // 19 GB per launch through the 64 B/clk vector L1s, which held the MFMA pipe at half rate.)
    // The workgroups of a pair start their scan at different tiles (and wrap around), so that first touches of operand
    // tiles (HBM misses) are spread over the pair's workgroups.
    __shared__ uint4 s_tile[2][kSlabs * 64];
    const int nsteps = (ntiles - sp + ksplit - 1) / ksplit;  // tiles sp, sp + ksplit, ... of this workgroup's split
    const int ngroups = max(bpp / ksplit, 1);
    const int s0 = nsteps > 0 ? (int)(((long long)qgroup * nsteps) / ngroups) % nsteps : 0;
    auto tile_of = [&](int i) {  // tile of step i (i <= nsteps: step nsteps wraps to a harmless re-load of the first tile)
        int sc = s0 + i;
        sc = sc >= nsteps ? sc - nsteps : sc;
        return sp + sc * ksplit;
    };
    uint4 g0, g1, g2 = make_uint4(0u, 0u, 0u, 0u);
    auto stage_load = [&](int tile) {
        const uint4* tp = descT + (size_t)tile * kSlabs * 64 + lane;
        g0 = tp[wave * 64];
        g1 = tp[(wave + 4) * 64];
        if (wave == 0) g2 = tp[8 * 64];
    };
    auto stage_store = [&](int buf) {
        s_tile[buf][wave * 64 + lane] = g0;
        s_tile[buf][(wave + 4) * 64 + lane] = g1;
        if (wave == 0) s_tile[buf][8 * 64 + lane] = g2;
    };
    if (nsteps > 0) {
        stage_load(tile_of(0));
        stage_store(0);
        __syncthreads();
        // two accumulator sets: while the MFMAs of step i fill one, the checks of step i - 1 read the other
        f32x16 accA[NC], accB[NC];
        stage_load(tile_of(1));
        Q.template step<false>(accA, accB, &s_tile[0][lane], 0, bv, bi, t4);
        stage_store(1);
        __syncthreads();
        int i = 1;
        for (; i + 1 < nsteps; i += 2) {  // i odd: tile i sits in buffer 1
            stage_load(tile_of(i + 1));
            Q.template step<true>(accB, accA, &s_tile[1][lane], tile_of(i - 1), bv, bi, t4);
            stage_store(0);
            __syncthreads();
            stage_load(tile_of(i + 2));
            Q.template step<true>(accA, accB, &s_tile[0][lane], tile_of(i), bv, bi, t4);
            stage_store(1);
            __syncthreads();
        }
        if (i < nsteps) {  // one more tile: into B, then B's checks
            Q.template step<true>(accB, accA, &s_tile[1][lane], tile_of(i - 1), bv, bi, t4);
            mfma_settle<NC>(accB);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int c = 0; c < NC; c++) bf3_check<NC>(accB, c, j, tile_of(i), bv, bi, t4);
        } else {
            mfma_settle<NC>(accA);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int c = 0; c < NC; c++) bf3_check<NC>(accA, c, j, tile_of(i - 1), bv, bi, t4);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int row = q0 + 32 * c + l31;
        if (row < nA) {
            const size_t o = ((size_t)pair * cap + row) * kCandStride + (sp * 2 + h) * 4;
#pragma unroll
            for (int k = 0; k < kKeep; k++) {
                cand_idx[o + k] = (bv[c][k] > -1e29f) ? bi[c][k] : -1;  // sentinel / all-padding blocks
                cand_val[o + k] = bv[c][k];
            }
            cand_val[o + 3] = t4[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void min_pair(float& d, int& i) {  // (distance, index) minimum over W consecutive lanes
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) {
        const float od = __shfl_xor(d, m, W);
        const int oi = __shfl_xor(i, m, W);
        if (od < d || (od == d && oi < i)) {
            d = od;
            i = oi;
        }
    }
}

__device__ __forceinline__ void knn_store(int* __restrict__ knn_idx, float* __restrict__ knn_dist, size_t o, float d0, int i0,
                                          float d1, int i1) {
    const bool h0 = i0 != 0x7fffffff, h1 = i1 != 0x7fffffff;
    knn_idx[o] = h0 ? i0 : -1;
    knn_idx[o + 1] = h1 ? i1 : -1;
    knn_dist[o] = h0 ? sqrtf(d0) : FLT_MAX;
    knn_dist[o + 1] = h1 ? sqrtf(d1) : FLT_MAX;
}

// The certificate's margin (ONE derivation; DESIGN.md 5.0 and tests/test_gpu_match.py quote the same numbers).  Score =
// a.b - 0.5 ||b||^2, distance D = ||a||^2 - 2 score; everything below is for ||a|| = ||b|| = 1 and scales with
// N = max(1, (||a||^2 + max ||b||^2) / 2) >= ||a|| ||b||, which is what the margin is multiplied by.
//   eps_split = 3 * 2^-16 = 4.6e-5   bf16 round-to-nearest keeps 8 significand bits: |a - hi| <= 2^-8 |a|, the residue left by
//                                    hi + lo is <= 2^-16 |a|; dropped: a_lo.b_lo (2^-16) + r_a.b + a.r_b (2 * 2^-16), Cauchy-Schwarz
//   eps_acc   = 195 * 2^-24 * 1.5 = 1.7e-5   FP32 accumulation of 3 x 64 products + 3 norm pieces, sum of |terms| <= 1.5
//   eps_norm  = 2e-6                 -0.5 ||b||^2 itself (64 fp32 terms) and its 24-bit three-piece bf16 form
//   eps = eps_split + eps_acc + eps_norm = 6.5e-5  >= |MFMA score - exact score|      (observed worst: 5e-6)
//   delta = 64 * 2^-24 * 4 = 1.5e-5  the re-scored float distance (sequential sum, D <= 4), delta_n = 4e-6 the same for ||a||^2
// Pruning needs margin >= 2 eps = 1.3e-4 (a block below s2 - margin cannot beat the two rows at s1, s2), the certificate
// needs margin >= 2 eps + delta + delta_n = 1.5e-4 (a left-out row has float distance > d1, strictly: ties cannot hide).
// 2e-4 covers the worst-case bound 1.33 times and the observed error 20 times.
constexpr float kCertMargin = 2e-4f;

// 32 lanes per query; candidate c of a query = row e of kept block k of group g, c = 12 g + 4 k + e
__global__ __launch_bounds__(256) void k_match_select(const float* __restrict__ desc_base, const FrameCounters* __restrict__ cnt, int cap,
                                                      const int2* __restrict__ pair_frames, const int* __restrict__ cand_idx,
                                                      const float* __restrict__ cand_val, const int* __restrict__ nmax_bits, int groups,
                                                      float cert_scale, int* __restrict__ knn_idx, float* __restrict__ knn_dist,
                                                      int* __restrict__ redo_list, int* __restrict__ redo_cnt) {
    const int pair = blockIdx.y;
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap), nB = min(cnt[fb].n_raw, cap);
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), sub = threadIdx.x & 31;
    if (q >= nA) return;
    const size_t co = ((size_t)pair * cap + q) * kCandStride;
    const float* a = desc_base + ((size_t)fa * cap + q) * 64;
    // ||a||^2 (any summation order: it only enters margins)
    float na2 = a[2 * sub] * a[2 * sub] + a[2 * sub + 1] * a[2 * sub + 1];
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) na2 += __shfl_xor(na2, m, 32);
    const float margin = cert_scale * kCertMargin * fmaxf(1.f, 0.5f * (na2 + __int_as_float(nmax_bits[fb])));
    // Second largest of the kept blocks' maxima: two different rows score at least s2, so a block whose maximum lies more
    // than the error margin below s2 cannot hold one of the two nearest rows and is not re-scored.
    float bm = -FLT_MAX;  // lane sub < 4 * groups: slot sub of the query (k = 3 is t4, not a block)
    if (sub < 4 * groups && (sub & 3) < kKeep && cand_idx[co + sub] >= 0) bm = cand_val[co + sub];
    float s1 = bm, s2 = -FLT_MAX;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {  // (largest, second largest) of 32 lanes
        const float o1 = __shfl_xor(s1, m, 32), o2 = __shfl_xor(s2, m, 32);
        s2 = fmaxf(fminf(s1, o1), fmaxf(s2, o2));
        s1 = fmaxf(s1, o1);
    }
    const float keep_from = s2 - margin;  // needs margin >= 2 eps (see kCertMargin)
    float b0 = FLT_MAX, b1 = FLT_MAX;  // this lane's two smallest (distance, row), lexicographic
    int j0 = 0x7fffffff, j1 = 0x7fffffff;
    for (int c = sub; c < groups * 4 * kKeep; c += 32) {
        const int g = c / (4 * kKeep), k = (c - g * 4 * kKeep) >> 2, e = c & 3;
        const int blk = cand_idx[co + g * 4 + k];
        const int row = (blk >> 2) * 32 + 8 * (blk & 3) + 4 * (g & 1) + e;
        if (blk < 0 || row >= nB || cand_val[co + g * 4 + k] < keep_from) continue;
        const float* b = desc_base + ((size_t)fb * cap + row) * 64;
        float s = 0.f;
#pragma unroll 16
        for (int kk = 0; kk < 64; kk++) {  // sequential float accumulation: the oracle's normL2Sqr (A.7)
            const float t = a[kk] - b[kk];
            s += t * t;
        }
        if (s < b0 || (s == b0 && row < j0)) {
            b1 = b0, j1 = j0;
            b0 = s, j0 = row;
        } else if (s < b1 || (s == b1 && row < j1)) {
            b1 = s, j1 = row;
        }
    }
    float d0 = b0;
    int i0 = j0;
    min_pair<32>(d0, i0);
    float d1 = (j0 == i0) ? b1 : b0;  // the winner lane offers its runner-up
    int i1 = (j0 == i0) ? j1 : j0;
    min_pair<32>(d1, i1);
    // certificate: t4 of a group bounds the score of every train row the group left out
    bool doubt = false;
    if (sub < groups && i1 != 0x7fffffff) {
        const float t4 = cand_val[co + sub * 4 + 3];
        doubt = t4 > -1e29f && !((na2 - 2.f * t4) > d1 + margin);
    }
    const unsigned long long m = __ballot(doubt) >> (threadIdx.x & 32) & 0xffffffffull;
    if (sub == 0) {
        knn_store(knn_idx, knn_dist, ((size_t)pair * cap + q) * 2, d0, i0, d1, i1);
        if (m != 0ull) redo_list[atomicAdd(redo_cnt, 1)] = pair * cap + q;
    }
}

// Exact 2-NN of the listed queries over ALL train rows: one 256-thread workgroup per query, thread T takes rows T, T + 256, ...
__global__ __launch_bounds__(256) void k_match_exact(const float* __restrict__ desc_base, const FrameCounters* __restrict__ cnt, int cap,
                                                     const int2* __restrict__ pair_frames, const int* __restrict__ redo_list,
                                                     const int* __restrict__ redo_cnt, int* __restrict__ knn_idx,
                                                     float* __restrict__ knn_dist) {
    __shared__ float s_d[4][2];
    __shared__ int s_i[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, total = *redo_cnt;
    for (int e = blockIdx.x; e < total; e += gridDim.x) {
        const int pq = redo_list[e], pair = pq / cap, q = pq - pair * cap;
        const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
        const int nB = min(cnt[fb].n_raw, cap);
        float a[64];
        const float4* ap = reinterpret_cast<const float4*>(desc_base + ((size_t)fa * cap + q) * 64);
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
            const float4 v = ap[k4];
            a[4 * k4] = v.x, a[4 * k4 + 1] = v.y, a[4 * k4 + 2] = v.z, a[4 * k4 + 3] = v.w;
        }
        float b0 = FLT_MAX, b1 = FLT_MAX;  // the thread's two smallest distances; rows come in ascending order, ties keep the earlier
        int j0 = 0x7fffffff, j1 = 0x7fffffff;
        for (int r = tid; r < nB; r += 256) {
            const float4* bp = reinterpret_cast<const float4*>(desc_base + ((size_t)fb * cap + r) * 64);
            float s = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 16; k4++) {
                const float4 v = bp[k4];
                float t = a[4 * k4] - v.x;
                s += t * t;
                t = a[4 * k4 + 1] - v.y;
                s += t * t;
                t = a[4 * k4 + 2] - v.z;
                s += t * t;
                t = a[4 * k4 + 3] - v.w;
                s += t * t;
            }
            if (s < b0) {
                b1 = b0, j1 = j0;
                b0 = s, j0 = r;
            } else if (s < b1) {
                b1 = s, j1 = r;
            }
        }
        float d0 = b0;
        int i0 = j0;
        min_pair<64>(d0, i0);
        float d1 = (j0 == i0) ? b1 : b0;  // the winner lane offers its runner-up, the others their best
        int i1 = (j0 == i0) ? j1 : j0;
        min_pair<64>(d1, i1);
        if (lane == 0) {
            s_d[wave][0] = d0, s_i[wave][0] = i0;
            s_d[wave][1] = d1, s_i[wave][1] = i1;
        }
        __syncthreads();
        if (tid == 0) {  // merge four sorted pairs, (distance, row) lexicographic
            float m0 = FLT_MAX, m1 = FLT_MAX;
            int n0 = 0x7fffffff, n1 = 0x7fffffff;
            for (int w = 0; w < 4; w++)
                for (int k = 0; k < 2; k++) {
                    const float d = s_d[w][k];
                    const int i = s_i[w][k];
                    if (d < m0 || (d == m0 && i < n0)) {
                        m1 = m0, n1 = n0;
                        m0 = d, n0 = i;
                    } else if (d < m1 || (d == m1 && i < n1)) {
                        m1 = d, n1 = i;
                    }
                }
            knn_store(knn_idx, knn_dist, ((size_t)pair * cap + q) * 2, m0, n0, m1, n1);
        }
        __syncthreads();
    }
}

// Exact 2-NN of EVERY query, any descriptor length: the matcher of the 128-float (`extended`) descriptors, which the reference
// switches off (camera_utility.hpp:37) and the MFMA shortlist above - built for 64 floats - does not serve.  One query per
// thread with its descriptor in registers, train rows staged in LDS 32 at a time and read as broadcasts, distances summed
// in the oracle's order (A.7), rows visited in ascending order with strict '<': ties keep the lower train index.
template <int D>
__global__ __launch_bounds__(256) void k_match_scan(const float* __restrict__ desc_base, const FrameCounters* __restrict__ cnt, int cap,
                                                    const int2* __restrict__ pair_frames, int* __restrict__ knn_idx,
                                                    float* __restrict__ knn_dist) {
    constexpr int TR = 32;
    __shared__ float4 s_b[TR * D / 4];
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int fa = pair_frames[pair].x, fb = pair_frames[pair].y;
    const int nA = min(cnt[fa].n_raw, cap), nB = min(cnt[fb].n_raw, cap);
    if ((int)blockIdx.x * 256 >= nA) return;
    const int q = blockIdx.x * 256 + tid;
    float a[D];
    {
        const float4* ap = reinterpret_cast<const float4*>(desc_base + ((size_t)fa * cap + min(q, nA - 1)) * D);
#pragma unroll
        for (int k4 = 0; k4 < D / 4; k4++) {
            const float4 v = ap[k4];
            a[4 * k4] = v.x, a[4 * k4 + 1] = v.y, a[4 * k4 + 2] = v.z, a[4 * k4 + 3] = v.w;
        }
    }
    float d0 = FLT_MAX, d1 = FLT_MAX;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    const float4* bsrc = reinterpret_cast<const float4*>(desc_base + (size_t)fb * cap * D);
    for (int r0 = 0; r0 < nB; r0 += TR) {
        const int nr = min(TR, nB - r0);
        __syncthreads();
        for (int e = tid; e < nr * (D / 4); e += 256) s_b[e] = bsrc[(size_t)r0 * (D / 4) + e];
        __syncthreads();
        for (int r = 0; r < nr; r++) {
            float sacc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < D / 4; k4++) {
                const float4 v = s_b[r * (D / 4) + k4];
                float t = a[4 * k4] - v.x;
                sacc += t * t;
                t = a[4 * k4 + 1] - v.y;
                sacc += t * t;
                t = a[4 * k4 + 2] - v.z;
                sacc += t * t;
                t = a[4 * k4 + 3] - v.w;
                sacc += t * t;
            }
            if (sacc < d0) {
                d1 = d0, i1 = i0;
                d0 = sacc, i0 = r0 + r;
            } else if (sacc < d1) {
                d1 = sacc, i1 = r0 + r;
            }
        }
    }
    if (q < nA) knn_store(knn_idx, knn_dist, ((size_t)pair * cap + q) * 2, d0, i0, d1, i1);
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
static int enqueue_filter(vo_ctx* ctx, FrameSet& fs, PairSet& ps, hipStream_t stream) {
    const vo_params& p = ctx->prm;
    hipLaunchKernelGGL(k_match_filter, dim3(ps.P), dim3(1024), 0, stream, fs.kp, fs.cnt, fs.cap, ps.frames, ps.knn_idx, ps.knn_dist,
                       p.ratio_thresh, p.roi_x_lo, p.roi_y_lo, p.roi_x_hi, p.roi_y_hi, ps.match_cap, ps.p1, ps.p2, ps.matches, ps.st);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

int vo_enqueue_match(vo_ctx* ctx, FrameSet& fs, PairSet& ps, bool filter, hipStream_t stream) {
    ctx->d_redo_cnt = ps.redo_cnt;
    const int cap = fs.cap, P = ps.P;
    if (fs.dlen != 64) {
        // 128-float descriptors: the plain exact scan (see k_match_scan), timed as the shortlist stage
        {
            StageScope sc(ctx, ST_MATCH_MFMA, stream);
            VO_HIP(hipMemsetAsync(ps.redo_cnt, 0, sizeof(int), stream));
            hipLaunchKernelGGL(k_match_scan<128>, dim3((cap + 255) / 256, P), dim3(256), 0, stream, fs.desc, fs.cnt, cap, ps.frames,
                               ps.knn_idx, ps.knn_dist);
            VO_HIP(hipGetLastError());
        }
        if (filter) {
            StageScope sc(ctx, ST_MATCH_SELECT, stream);
            VO_TRY(enqueue_filter(ctx, fs, ps, stream));
        }
        return VO_OK;
    }
    {
        StageScope sc(ctx, ST_MATCH_PREP, stream);
        VO_HIP(hipMemsetAsync(fs.nmax, 0, (size_t)fs.F * sizeof(int), stream));
        VO_HIP(hipMemsetAsync(ps.redo_cnt, 0, sizeof(int), stream));
        auto prep = ctx->match_fp32 ? k_match_prep<true> : k_match_prep<false>;
        hipLaunchKernelGGL(prep, dim3(cap / 32, fs.F), dim3(64), 0, stream, fs.desc, fs.cnt, cap, reinterpret_cast<float4*>(fs.descT),
                           fs.nrm, fs.nmax);
        VO_HIP(hipGetLastError());
    }
    // Train-range splits: one per query block when the batch alone fills the chip (long scans make top-4 insertions rare),
    // four for a few pairs so that a single pair still spreads over 512 wavefronts.
    constexpr int kQueryBlocks = 2;
    const int ksplit = (P >= 8) ? 1 : 4;
    {
        StageScope sc(ctx, ST_MATCH_MFMA, stream);
        const int bpp = ((cap + 128 * kQueryBlocks - 1) / (128 * kQueryBlocks)) * ksplit;  // workgroups per pair: 4 query blocks of 64, one train split each
        auto shortlist = ctx->match_fp32 ? k_match_bf3<kQueryBlocks, true> : k_match_bf3<kQueryBlocks, false>;
        hipLaunchKernelGGL(shortlist, dim3(bpp * P), dim3(256), 0, stream, fs.desc,
                           reinterpret_cast<const uint4*>(fs.descT), fs.nrm, fs.cnt, cap, ps.frames, ksplit, bpp, (P % 8 == 0) ? 1 : 0,
                           ps.cand_val, ps.cand_idx);
        VO_HIP(hipGetLastError());
    }
    {
        StageScope sc(ctx, ST_MATCH_SELECT, stream);
        const float cert_scale = ctx->force_exact_match ? 1e30f : 1.f;
        hipLaunchKernelGGL(k_match_select, dim3(cap / 8, P), dim3(256), 0, stream, fs.desc, fs.cnt, cap, ps.frames, ps.cand_idx,
                           ps.cand_val, fs.nmax, 2 * ksplit, cert_scale, ps.knn_idx, ps.knn_dist, ps.redo_list, ps.redo_cnt);
        hipLaunchKernelGGL(k_match_exact, dim3(1024), dim3(256), 0, stream, fs.desc, fs.cnt, cap, ps.frames, ps.redo_list,
                           ps.redo_cnt, ps.knn_idx, ps.knn_dist);
        VO_HIP(hipGetLastError());
        if (filter) VO_TRY(enqueue_filter(ctx, fs, ps, stream));
    }
    return VO_OK;
}
