// TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.  PARITY UNPINNED (no reference fixtures exist).
//
// CPU restatement of the calib3d calls of the reference:
//   findEssentialMat(p1,p2,K,RANSAC,0.99,2.0,mask)      camera_utility.hpp:642   
//   recoverPose(E,inl1,inl2,K,R,t)                      camera_utility.hpp:667   (A.10)
//   triangulatePoints(P0,P1,inl1,inl2,X4)               camera_utility.hpp:320   (A.11)
//   optRelativePose gates                               camera_utility.hpp:644-682
//   triangPoints post-processing                        camera_utility.hpp:326-348, 353-382, 533-587;
//                                                       math_utility.hpp:150-154, 235-258 (A.12)
// OpenCV sources (calib3d/five-point.cpp, ptsetreg.cpp, triangulate.cpp, core/lapack.cpp,
// core/mathfuncs.cpp) are not vendored by the reference; restated after SURVEY.md Appendix A.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle_internal.h"

namespace vo_oracle {

// ---------------------------------------------------------------- core/lapack.cpp JacobiSVDImpl_<double>
// At: n1 rows x m (row stride astep); first n rows hold the vectors to orthogonalise.
// Vt: n x n (may be null).  On return rows of At are the left vectors (normalised), W the
// singular values (descending), rows n..n1-1 are completed as OpenCV does (RNG 0x12345678).
void jacobi_svd(double* At, int astep, double* Wout, double* Vt, int vstep, int m, int n, int n1) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    std::vector<double> W(n);
    int i, j, k, iter, max_iter = std::max(m, 30);
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = std::sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            if (Vt) {
                for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
                for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
            }
        }
    }
    for (i = 0; i < n; i++) Wout[i] = W[i];
    if (!Vt) return;
    RNG rng(0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (rng.next() & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += std::fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = std::sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

// cv::SVD::compute(A (m x m), W, U, Vt) for square A, m <= 4: U columns / Vt rows.
static void svd_square(const double* A, int m, double* W, double* U, double* Vt) {
    double At[16], V[16];
    for (int i = 0; i < m; i++)
        for (int k = 0; k < m; k++) At[i * m + k] = A[k * m + i];  // rows of At = columns of A
    jacobi_svd(At, m, W, V, m, m, m, m);
    for (int i = 0; i < m; i++)
        for (int k = 0; k < m; k++) {
            if (U) U[k * m + i] = At[i * m + k];
            Vt[i * m + k] = V[i * m + k];
        }
}

// ---------------------------------------------------------------- A.8 sample stream
void ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5) {
    RNG rng(seed);
    for (int t = 0; t < n_tuples; t++) {
        int* idx = idx5 + 5 * t;
        for (int i = 0; i < 5; i++) {
            int idx_i;
            for (;;) {
                idx_i = rng.uniform(0, count);
                bool dup = false;
                for (int q = 0; q < i; q++) dup |= (idx[q] == idx_i);
                if (!dup) break;
            }
            idx[i] = idx_i;
        }
    }
}

// ---------------------------------------------------------------- A.9 five-point solver
// Polynomials in (x,y,z) of total degree <= 3 stored densely by exponent.
struct Poly {
    double c[4][4][4];
    Poly() { std::memset(c, 0, sizeof(c)); }
};
static Poly padd(const Poly& a, const Poly& b, double sb = 1.0) {
    Poly r;
    for (int i = 0; i < 64; i++) (&r.c[0][0][0])[i] = (&a.c[0][0][0])[i] + sb * (&b.c[0][0][0])[i];
    return r;
}
static Poly pmul(const Poly& a, const Poly& b) {
    Poly r;
    for (int a0 = 0; a0 < 4; a0++)
        for (int a1 = 0; a0 + a1 < 4; a1++)
            for (int a2 = 0; a0 + a1 + a2 < 4; a2++) {
                double va = a.c[a0][a1][a2];
                if (va == 0) continue;
                for (int b0 = 0; a0 + b0 < 4; b0++)
                    for (int b1 = 0; a1 + b1 < 4; b1++)
                        for (int b2 = 0; a2 + b2 < 4; b2++) {
                            if (a0 + a1 + a2 + b0 + b1 + b2 > 3) continue;
                            r.c[a0 + b0][a1 + b1][a2 + b2] += va * b.c[b0][b1][b2];
                        }
            }
    return r;
}
static Poly pscale(const Poly& a, double s) {
    Poly r;
    for (int i = 0; i < 64; i++) (&r.c[0][0][0])[i] = s * (&a.c[0][0][0])[i];
    return r;
}

// monomial (column) order of A.9 step 3
static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                 {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                 {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

// cv::solvePoly (Durand-Kerner, maxIters = 300).  coeffs ascending, n0 = degree.
static void solve_poly(const double* coeffs0, int n0, double* re, double* im, int* nroots) {
    struct C {
        double re, im;
    };
    auto mul = [](C a, C b) { return C{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; };
    auto sub = [](C a, C b) { return C{a.re - b.re, a.im - b.im}; };
    auto addc = [](C a, C b) { return C{a.re + b.re, a.im + b.im}; };
    auto divc = [](C a, C b) {
        double t = 1. / (b.re * b.re + b.im * b.im);
        return C{(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
    };
    int n = n0;
    std::vector<C> coeffs(n0 + 1), roots(n0 + 1);
    for (int i = 0; i <= n0; i++) coeffs[i] = C{coeffs0[i], 0};
    for (; n > 1; n--)
        if (std::fabs(coeffs[n].re) + std::fabs(coeffs[n].im) > DBL_EPSILON) break;
    C p{1, 0}, r{1, 1};
    for (int i = 0; i < n; i++) {
        roots[i] = p;
        p = mul(p, r);
    }
    const int maxIters = 300;
    for (int iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            C num = coeffs[n], denom = coeffs[n];
            int num_same_root = 1;
            for (int j = 0; j < n; j++) {
                num = addc(mul(num, p), coeffs[n - j - 1]);
                if (j != i) {
                    C d = sub(p, roots[j]);
                    if (d.re != 0 || d.im != 0)
                        denom = mul(denom, d);
                    else
                        num_same_root++;
                }
            }
            num = divc(num, denom);
            if (num_same_root > 1) {
                double old_re = num.re, old_im = num.im;
                int square_root_times = num_same_root % 2 == 0 ? num_same_root / 2 : num_same_root / 2 - 1;
                for (int q = 0; q < square_root_times; q++) {
                    num.re = old_re * old_re + old_im * old_im;
                    num.re = std::sqrt(num.re);
                    num.re += old_re;
                    num.im = num.re - old_re;
                    num.re /= 2;
                    num.re = std::sqrt(num.re);
                    num.im /= 2;
                    num.im = std::sqrt(num.im);
                    if (old_re < 0) num.im = -num.im;
                }
                if (num_same_root % 2 != 0) {
                    double ang = std::atan2(num.im, num.re) / 3, mag = std::cbrt(std::hypot(num.re, num.im));
                    num = C{mag * std::cos(ang), mag * std::sin(ang)};
                }
            }
            roots[i] = sub(p, num);
            maxDiff = std::max(maxDiff, std::hypot(num.re, num.im));
        }
        if (maxDiff <= 0) break;
    }
    for (int i = 0; i < n; i++) {
        if (std::fabs(roots[i].im) < 1e-100) roots[i].im = 0;
        re[i] = roots[i].re;
        im[i] = roots[i].im;
    }
    *nroots = n;
}

// EMEstimatorCallback::runKernel.  x1,x2: 5 normalised points each ([5][2]).  Returns the
// number of models (<= 10); E_out rows are 3x3 row-major, unit Frobenius norm, ordered by
// ascending real root z (the oracle's definition of the candidate order, A.9 last paragraph).
int five_point(const double x1[10], const double x2[10], double E_out[90]) {
    double At[9 * 9];
    std::memset(At, 0, sizeof(At));
    for (int i = 0; i < 5; i++) {
        double a = x1[2 * i], b = x1[2 * i + 1], c = x2[2 * i], d = x2[2 * i + 1];
        double* q = At + 9 * i;
        // row of Q for vec(E) row-major and the constraint x2^T E x1 = 0 (the one computeError tests)
        q[0] = c * a;
        q[1] = c * b;
        q[2] = c;
        q[3] = d * a;
        q[4] = d * b;
        q[5] = d;
        q[6] = a;
        q[7] = b;
        q[8] = 1.0;
    }
    // SVD::compute(Q, W, U, Vt, FULL_UV): m=5 < n=9 -> rows of Q are orthogonalised, rows
    // 5..8 of Vt come out of JacobiSVD's completion step.
    double W[5], V5[25];
    jacobi_svd(At, 9, W, V5, 5, 9, 5, 9);
    const double* EE[4] = {At + 9 * 5, At + 9 * 6, At + 9 * 7, At + 9 * 8};

    // E(x,y,z) = x E1 + y E2 + z E3 + E4, entry (i,j) = vec index 3*i + j
    Poly Ep[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Ep[i][j].c[1][0][0] = EE[0][3 * i + j];
            Ep[i][j].c[0][1][0] = EE[1][3 * i + j];
            Ep[i][j].c[0][0][1] = EE[2][3 * i + j];
            Ep[i][j].c[0][0][0] = EE[3][3 * i + j];
        }
    Poly EEt[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Poly s;
            for (int k = 0; k < 3; k++) s = padd(s, pmul(Ep[i][k], Ep[j][k]));
            EEt[i][j] = s;
        }
    Poly tr = padd(padd(EEt[0][0], EEt[1][1]), EEt[2][2]);
    Poly rows[10];
    rows[0] = padd(padd(pmul(Ep[0][0], padd(pmul(Ep[1][1], Ep[2][2]), pmul(Ep[1][2], Ep[2][1]), -1.0)),
                        pmul(Ep[0][1], padd(pmul(Ep[1][0], Ep[2][2]), pmul(Ep[1][2], Ep[2][0]), -1.0)), -1.0),
                   pmul(Ep[0][2], padd(pmul(Ep[1][0], Ep[2][1]), pmul(Ep[1][1], Ep[2][0]), -1.0)));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Poly s;
            for (int k = 0; k < 3; k++) {
                Poly f = pscale(EEt[i][k], 2.0);
                if (i == k) f = padd(f, tr, -1.0);
                s = padd(s, pmul(f, Ep[k][j]));
            }
            rows[1 + 3 * i + j] = s;
        }
    double A[10][20];
    for (int r = 0; r < 10; r++)
        for (int cidx = 0; cidx < 20; cidx++) A[r][cidx] = rows[r].c[kMono[cidx][0]][kMono[cidx][1]][kMono[cidx][2]];
    // A = A[:, :10]^-1 * A[:, 10:]  (Gauss-Jordan, partial pivoting)
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
        if (A[piv][col] == 0) return 0;
        if (piv != col)
            for (int cidx = 0; cidx < 20; cidx++) std::swap(A[piv][cidx], A[col][cidx]);
        double inv = 1.0 / A[col][col];
        for (int cidx = 0; cidx < 20; cidx++) A[col][cidx] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            double f = A[r][col];
            if (f == 0) continue;
            for (int cidx = 0; cidx < 20; cidx++) A[r][cidx] -= f * A[col][cidx];
        }
    }
    // B rows (step 4): [x-block z^3..1 | y-block z^3..1 | 1-block z^4..1]
    double b[3][13];
    for (int i = 0; i < 3; i++) {
        const double* ar1 = &A[4 + 2 * i][10];
        const double* ar2 = &A[5 + 2 * i][10];
        double row1[13] = {0}, row2[13] = {0};
        for (int q = 0; q < 3; q++) row1[1 + q] = ar1[q];
        for (int q = 0; q < 3; q++) row1[5 + q] = ar1[3 + q];
        for (int q = 0; q < 4; q++) row1[9 + q] = ar1[6 + q];
        for (int q = 0; q < 3; q++) row2[0 + q] = ar2[q];
        for (int q = 0; q < 3; q++) row2[4 + q] = ar2[3 + q];
        for (int q = 0; q < 4; q++) row2[8 + q] = ar2[6 + q];
        for (int q = 0; q < 13; q++) b[i][q] = row1[q] - row2[q];
    }
    // det B(z): degree 10.  Univariate polynomials, ascending powers.
    auto upoly = [&](int row, int blk, double* out) -> int {  // returns degree
        if (blk < 2) {
            for (int q = 0; q < 4; q++) out[q] = b[row][4 * blk + 3 - q];
            return 3;
        }
        for (int q = 0; q < 5; q++) out[q] = b[row][8 + 4 - q];
        return 4;
    };
    auto umul = [](const double* a, int da, const double* bb, int db, double* out) {
        for (int q = 0; q <= da + db; q++) out[q] = 0;
        for (int i = 0; i <= da; i++)
            for (int j = 0; j <= db; j++) out[i + j] += a[i] * bb[j];
    };
    double c[11] = {0};
    {
        double p[3][4], q[3][4], r[3][5];
        for (int i = 0; i < 3; i++) {
            upoly(i, 0, p[i]);
            upoly(i, 1, q[i]);
            upoly(i, 2, r[i]);
        }
        auto minor_acc = [&](const double* f, int df, const double* g1, const double* h1, const double* g2,
                             const double* h2, int dg, int dh, double sign) {
            // sign * f * (g1*h1 - g2*h2)
            double t1[8], t2[8], t[8], out[12];
            umul(g1, dg, h1, dh, t1);
            umul(g2, dg, h2, dh, t2);
            for (int k = 0; k <= dg + dh; k++) t[k] = t1[k] - t2[k];
            umul(f, df, t, dg + dh, out);
            for (int k = 0; k <= df + dg + dh; k++) c[k] += sign * out[k];
        };
        minor_acc(p[0], 3, q[1], r[2], q[2], r[1], 3, 4, 1.0);
        minor_acc(q[0], 3, p[1], r[2], p[2], r[1], 3, 4, -1.0);
        minor_acc(r[0], 4, p[1], q[2], p[2], q[1], 3, 3, 1.0);
    }
    double re[11], im[11];
    int nroots = 0;
    solve_poly(c, 10, re, im, &nroots);
    std::vector<double> zs;
    for (int i = 0; i < nroots; i++)
        if (!(std::fabs(im[i]) > 1e-10)) zs.push_back(re[i]);
    std::sort(zs.begin(), zs.end());
    int count = 0;
    for (double z1 : zs) {
        double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b[j];
            bz[3 * j + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[3 * j + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[3 * j + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double Wz[3], Vt[9];
        svd_square(bz, 3, Wz, nullptr, Vt);  // SVD::solveZ -> last row of Vt
        const double* xy1 = Vt + 6;
        if (std::fabs(xy1[2]) < 1e-10) continue;
        double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; k++) {
            Ev[k] = EE[0][k] * x + EE[1][k] * y + EE[2][k] * z1 + EE[3][k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; k++) E_out[9 * count + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

// EMEstimatorCallback::computeError (Sampson), m1/m2 interleaved normalised xy doubles
void sampson_errors(const double* m1, const double* m2, int n, const double E[9], float* err) {
    for (int i = 0; i < n; i++) {
        double x1[3] = {m1[2 * i], m1[2 * i + 1], 1.}, x2[3] = {m2[2 * i], m2[2 * i + 1], 1.};
        double Ex1[3], Etx2[3];
        for (int r = 0; r < 3; r++) {
            Ex1[r] = E[3 * r] * x1[0] + E[3 * r + 1] * x1[1] + E[3 * r + 2] * x1[2];
            Etx2[r] = E[r] * x2[0] + E[3 + r] * x2[1] + E[6 + r] * x2[2];
        }
        double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
        double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
}

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cvRound(num / denom);
}

static void normalise_points(const float* p, int n, const double K[9], std::vector<double>& out) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];  // skew K[1] ignored (A.8)
    out.resize(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        out[2 * i] = ((double)p[2 * i] - cx) / fx;
        out[2 * i + 1] = ((double)p[2 * i + 1] - cy) / fy;
    }
}

// findEssentialMat(RANSAC) == RANSACPointSetRegistrator::run (A.8)
int essential_ransac(const float* p1, const float* p2, int n, const double K[9], double prob, double thr_px,
                     int max_iters, uint64_t seed, double E[9], uint8_t* mask, int* n_inl, int* iters_run,
                     const double* replay_models, const int* replay_nmodels) {
    if (n_inl) *n_inl = 0;
    if (iters_run) *iters_run = 0;
    if (n < 5) return VO_ERR_TOO_FEW_POINTS;
    std::vector<double> m1, m2;
    normalise_points(p1, n, K, m1);
    normalise_points(p2, n, K, m2);
    double threshold = thr_px / ((K[0] + K[4]) / 2);
    const int modelPoints = 5;
    int niters = std::max(max_iters, 1);
    int maxGoodCount = 0;
    std::vector<float> err(n);
    std::vector<uint8_t> cur(n), best(n, 0);
    double bestE[9] = {0}, models[90];
    if (n == modelPoints) {
        int nm = five_point(m1.data(), m2.data(), models);
        if (nm <= 0) return VO_ERR_NO_MODEL;
        std::memcpy(E, models, 9 * sizeof(double));
        for (int i = 0; i < n; i++) mask[i] = 1;
        if (n_inl) *n_inl = n;
        if (iters_run) *iters_run = 1;
        return VO_OK;
    }
    RNG rng(seed);
    const float t = (float)(threshold * threshold);
    int iter = 0;
    for (; iter < niters; iter++) {
        if (replay_models) {
            // replay mode (tests only): the candidate models of hypothesis `iter` are supplied by the
            // caller instead of being solved here; the registrator loop is otherwise unchanged
            int nmodels = replay_nmodels[iter];
            const double* mdl = replay_models + (size_t)iter * 90;
            for (int mi = 0; mi < nmodels; mi++) {
                sampson_errors(m1.data(), m2.data(), n, mdl + 9 * mi, err.data());
                int good = 0;
                for (int i = 0; i < n; i++) good += (cur[i] = (uint8_t)(err[i] <= t));
                if (good > std::max(maxGoodCount, modelPoints - 1)) {
                    std::swap(cur, best);
                    std::memcpy(bestE, mdl + 9 * mi, 9 * sizeof(double));
                    maxGoodCount = good;
                    niters = ransac_update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
                }
            }
            continue;
        }
        int idx[5];
        for (int i = 0; i < 5; i++) {
            int idx_i;
            for (;;) {
                idx_i = rng.uniform(0, n);
                bool dup = false;
                for (int q = 0; q < i; q++) dup |= (idx[q] == idx_i);
                if (!dup) break;
            }
            idx[i] = idx_i;
        }
        double s1[10], s2[10];
        for (int i = 0; i < 5; i++) {
            s1[2 * i] = m1[2 * idx[i]];
            s1[2 * i + 1] = m1[2 * idx[i] + 1];
            s2[2 * i] = m2[2 * idx[i]];
            s2[2 * i + 1] = m2[2 * idx[i] + 1];
        }
        int nmodels = five_point(s1, s2, models);
        if (nmodels <= 0) continue;
        for (int mi = 0; mi < nmodels; mi++) {
            sampson_errors(m1.data(), m2.data(), n, models + 9 * mi, err.data());
            int good = 0;
            for (int i = 0; i < n; i++) good += (cur[i] = (uint8_t)(err[i] <= t));
            if (good > std::max(maxGoodCount, modelPoints - 1)) {
                std::swap(cur, best);
                std::memcpy(bestE, models + 9 * mi, 9 * sizeof(double));
                maxGoodCount = good;
                niters = ransac_update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
            }
        }
    }
    if (iters_run) *iters_run = iter;
    if (maxGoodCount <= 0) {
        std::memset(mask, 0, n);
        return VO_ERR_NO_MODEL;
    }
    std::memcpy(E, bestE, sizeof(bestE));
    std::memcpy(mask, best.data(), n);
    if (n_inl) *n_inl = maxGoodCount;
    return VO_OK;
}

// ---------------------------------------------------------------- A.11 triangulatePoints
void triangulate(const double P0[12], const double P1[12], const double* x1, const double* y1, const double* x2,
                 const double* y2, int n, double* X4) {
    const double* P[2] = {P0, P1};
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        double A[16], W[4], Vt[16];
        double xs[2] = {x1[i], x2[i]}, ys[2] = {y1[i], y2[i]};
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 4; k++) {
                A[(2 * j + 0) * 4 + k] = xs[j] * P[j][8 + k] - P[j][k];
                A[(2 * j + 1) * 4 + k] = ys[j] * P[j][8 + k] - P[j][4 + k];
            }
        svd_square(A, 4, W, nullptr, Vt);
        for (int k = 0; k < 4; k++) X4[(size_t)k * n + i] = Vt[12 + k];
    }
}

// ---------------------------------------------------------------- A.10 recoverPose
static double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static void mat3mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

int recover_pose(const double E[9], const float* p1, const float* p2, int n, const double K[9], double dist_thresh,
                 double R[9], double t[3], int* n_good) {
    std::vector<double> m1, m2;
    normalise_points(p1, n, K, m1);
    normalise_points(p2, n, K, m2);
    std::vector<double> x1(n), y1(n), x2(n), y2(n);
    for (int i = 0; i < n; i++) {
        x1[i] = m1[2 * i];
        y1[i] = m1[2 * i + 1];
        x2[i] = m2[2 * i];
        y2[i] = m2[2 * i + 1];
    }
    // decomposeEssentialMat
    double D[3], U[9], Vt[9];
    svd_square(E, 3, D, U, Vt);
    if (det3(U) < 0)
        for (double& v : U) v = -v;
    if (det3(Vt) < 0)
        for (double& v : Vt) v = -v;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double UW[9], R1[9], R2[9], tt[3] = {U[2], U[5], U[8]};
    mat3mul(U, Wm, UW);
    mat3mul(UW, Vt, R1);
    mat3mul(U, Wt, UW);
    mat3mul(UW, Vt, R2);
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int good[4];
    std::vector<double> Q(4 * (size_t)n);
    for (int c = 0; c < 4; c++) {
        const double* Rc = (c == 0 || c == 2) ? R1 : R2;
        double sgn = c < 2 ? 1.0 : -1.0;
        double Pk[12];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) Pk[4 * i + j] = Rc[3 * i + j];
            Pk[4 * i + 3] = sgn * tt[i];
        }
        triangulate(P0, Pk, x1.data(), y1.data(), x2.data(), y2.data(), n, Q.data());
        int g = 0;
        for (int i = 0; i < n; i++) {
            double q0 = Q[i], q1 = Q[(size_t)n + i], q2 = Q[2 * (size_t)n + i], q3 = Q[3 * (size_t)n + i];
            bool m = (q2 * q3 > 0);
            q0 /= q3;
            q1 /= q3;
            q2 /= q3;
            q3 /= q3;
            m = m && (q2 < dist_thresh);
            double z = Pk[8] * q0 + Pk[9] * q1 + Pk[10] * q2 + Pk[11] * q3;
            m = m && (z > 0) && (z < dist_thresh);
            g += m;
        }
        good[c] = g;
    }
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3])
        sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3])
        sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3])
        sel = 2;
    else
        sel = 3;
    const double* Rs = (sel == 0 || sel == 2) ? R1 : R2;
    double sg = sel < 2 ? 1.0 : -1.0;
    for (int i = 0; i < 9; i++) R[i] = Rs[i];
    for (int i = 0; i < 3; i++) t[i] = sg * tt[i];
    if (n_good) *n_good = good[sel];
    return VO_OK;
}

// ---------------------------------------------------------------- triangPoints (camera_utility.hpp:288-351)
int triang_points(const vo_params& P, const float* in1, const float* in2, int n, const double R[9], const double t[3],
                  const double K[9], std::vector<double>& Xk) {
    Xk.clear();
    if (n <= 0) return 0;
    // projectionMatrix(): K * [R|t]  (camera_utility.hpp:280-286)
    double P0[12], P1[12];
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
    auto proj = [&](const double* Rm, const double* tv, double* Pm) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += K[3 * i + k] * (j < 3 ? Rm[3 * k + j] : tv[k]);
                Pm[4 * i + j] = s;
            }
    };
    proj(I3, z3, P0);
    proj(R, t, P1);
    std::vector<double> x1(n), y1(n), x2(n), y2(n), X4(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        x1[i] = in1[2 * i];
        y1[i] = in1[2 * i + 1];
        x2[i] = in2[2 * i];
        y2[i] = in2[2 * i + 1];
    }
    triangulate(P0, P1, x1.data(), y1.data(), x2.data(), y2.data(), n, X4.data());
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> good;  // rows of goodWP (frame k)
    for (int i = 0; i < n; i++) {
        // X4 is CV_32F; my_convertFromHom divides in float (:544), then convertTo CV_64F (:327)
        float h[4];
        for (int k = 0; k < 4; k++) h[k] = (float)X4[(size_t)k * n + i];
        double X[3] = {(double)(h[0] / h[3]), (double)(h[1] / h[3]), (double)(h[2] / h[3])};
        // reproject_error() x2 (:353-382): projectPoints without distortion
        auto reproj = [&](const double* Rm, const double* tv, const float* pt) {
            double x = Rm[0] * X[0] + Rm[1] * X[1] + Rm[2] * X[2] + tv[0];
            double y = Rm[3] * X[0] + Rm[4] * X[1] + Rm[5] * X[2] + tv[1];
            double z = Rm[6] * X[0] + Rm[7] * X[1] + Rm[8] * X[2] + tv[2];
            z = z ? 1. / z : 1;
            x *= z;
            y *= z;
            double u = x * fx + cx, v = y * fy + cy;
            double du = (double)pt[0] - u, dv = (double)pt[1] - v;
            return std::sqrt(du * du + dv * dv);
        };
        double e0 = reproj(I3, z3, in1 + 2 * i), e1 = reproj(R, t, in2 + 2 * i);
        double mean = (e0 + e1) / 2.0;
        if (mean < (double)P.reprojection_tolerance && X[2] > 0) {  // filter_convertWP (:559)
            for (int r = 0; r < 3; r++) good.push_back(R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r]);
        }
    }
    int M = (int)(good.size() / 3);
    if (M == 0) return 0;
    // var_avgMat (math_utility.hpp:235-258): sequential mean += v/N; population variance
    double mean = 0, var = 0;
    for (int i = 0; i < M; i++) mean += good[3 * i + 2] / M;
    for (int i = 0; i < M; i++) {
        double d = good[3 * i + 2] - mean;
        var += d * d / M;
    }
    double lo = mean - 3.0 * std::sqrt(var), hi = mean + 3.0 * std::sqrt(var);
    for (int i = 0; i < M; i++)
        if (good[3 * i + 2] <= hi && good[3 * i + 2] >= lo) {
            Xk.push_back(good[3 * i]);
            Xk.push_back(good[3 * i + 1]);
            Xk.push_back(good[3 * i + 2]);
        }
    return (int)(Xk.size() / 3);
}

}  // namespace vo_oracle
