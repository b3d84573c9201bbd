// sddp_kernels.hpp -- the DDP engine as CDNA4 HIP kernels (gfx950, fp64, wave64).
//
// Replaces the arithmetic the reference delegates to the external `pyddp.DdpSolver.solve()` (reference
// python/ddp.py:101): derivative evaluation at every knot, backward Riccati sweep, forward line-search rollout,
// acceptance / termination -- see DESIGN.md "Algorithm" for the exact iteration (it is the one restated in
// oracle/ddp.py).
//
// Mapping (DESIGN.md "Kernel design"):
//   * one 64-lane wavefront (= one workgroup) per MPC instance, persistent over all DDP iterations: no host
//     round trip and no inter-workgroup traffic inside a solve;
//   * derivative phase: one LANE per knot (scalar register code, all knots of the horizon at once);
//   * Riccati sweep: serial over knots; the Vxx / [fx fu] / Q tiles of the current knot live in LDS, one lane per
//     tile element, Quu solved by a register-resident Gauss-Jordan (one lane per right-hand-side column),
//     small reductions by wavefront shuffles;
//   * line search: one LANE per step length alpha_j = alpha_0 * factor^j -- the whole backtracking ladder is
//     rolled out in one pass, the largest acceptable alpha is picked by a ballot;
//   * the knot sequence (trajectory, parameters, gains, derivative records) is read and written coalesced,
//     one knot per wave-wide access, and stays L2-resident (no MFMA: 13x13 / 6x6 tiles are too small to win).
#pragma once
#include <hip/hip_runtime.h>

#include "sddp_models.hpp"

namespace sddp {

constexpr int kWave = 64;
constexpr int kScal = 16;  // doubles per instance in the scratch `scal` record (test kernels / diagnostic stamps)

// Diagnostic build only (-DSDDP_STAMPS): per-phase shader-cycle sums, written to `scal`; never in the shipped library.
#ifdef SDDP_STAMPS
#define SDDP_T_DECL unsigned long long T_[kScal] = {0}; unsigned long long t_last_ = clock64();
#define SDDP_T_ARG , unsigned long long* T_, unsigned long long& t_last_
#define SDDP_T_PASS , T_, t_last_
#define SDDP_TICK(i) { const unsigned long long t_ = clock64(); T_[i] += t_ - t_last_; t_last_ = t_; }
#else
#define SDDP_T_DECL
#define SDDP_T_ARG
#define SDDP_T_PASS
#define SDDP_TICK(i)
#endif

struct SolveArgs {
    DevConsts c;
    sddp_options o;
    int N, B;
    const double* x0;   // [B][NX]
    const double* P;    // [B][N+1][NP]
    double* xs;         // [B][N+1][NX]  current iterate (in: warm start, out: solution)
    double* us;         // [B][N][NU]
    double* xn;         // [B][N+1][NX]  candidate
    double* un;         // [B][N][NU]
    double* dft;        // [B][N][NX]    defects d_{k+1} stored at k
    double* gains;      // [B][N][NU*(NX+1)]  kff (NU) then K (NU x NX, row-major)
    double* rec;        // [B][N+1][NREC]
    sddp_stats* stats;  // [B]
    double* scal;       // [B][kScal] (test kernels / diagnostic stamps)
    double alpha;       // forward test kernel only
    double mu;          // backward test kernel only
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
    return v;
}

// LDS tile layout of one instance (offsets in doubles)
template <class M>
struct Lds {
    static constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ;
    static constexpr int VXX = 0;
    static constexpr int VX = VXX + NX * NX;
    static constexpr int VP = VX + NX;
    static constexpr int DK = VP + NX;
    static constexpr int F = DK + NX;
    static constexpr int W = F + NX * NZ;
    static constexpr int Q = W + NX * NZ;
    static constexpr int QV = Q + NZ * NZ;
    static constexpr int REC = QV + NZ;
    static constexpr int PK = REC + M::NREC;
    static constexpr int KT = PK + M::NP;          // [NU][NX+1]: kff | K
    static constexpr int PIV = KT + NU * (NX + 1);
    static constexpr int TOTAL = PIV + NU + 2;
    static constexpr size_t BYTES = size_t(TOTAL) * sizeof(double);
};

// -----------------------------------------------------------------------------------------------------------------
// derivative phase: one lane per knot -> compact records in HBM/L2
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void phase_derivs(const DevConsts& c, int N, const double* __restrict__ xs,
                                             const double* __restrict__ us, const double* __restrict__ P,
                                             double* __restrict__ rec, int lane) {
    for (int k = lane; k <= N; k += kWave) {
        double x[M::NX], u[M::NU];
#pragma unroll
        for (int i = 0; i < M::NX; ++i) x[i] = xs[k * M::NX + i];
        const int ku = k < N ? k : N - 1;
#pragma unroll
        for (int i = 0; i < M::NU; ++i) u[i] = us[ku * M::NU + i];
        M::derivs(c, x, u, P + k * M::NP, k, N, rec + size_t(k) * M::NREC);
    }
}

// initial defects d_{k+1} = f(x_k,u_k) - x_{k+1}, total cost and defect 1-norm of the current iterate
template <class M>
__device__ __forceinline__ void phase_defects(const DevConsts& c, int N, const double* __restrict__ xs,
                                              const double* __restrict__ us, const double* __restrict__ P,
                                              double* __restrict__ dft, int lane, double& J, double& gap) {
    double Jl = 0.0, gl = 0.0;
    for (int k = lane; k <= N; k += kWave) {
        double x[M::NX];
#pragma unroll
        for (int i = 0; i < M::NX; ++i) x[i] = xs[k * M::NX + i];
        if (k < N) {
            double u[M::NU], xn[M::NX];
#pragma unroll
            for (int i = 0; i < M::NU; ++i) u[i] = us[k * M::NU + i];
            Jl += M::step(c, x, u, P + k * M::NP, k, xn);
#pragma unroll
            for (int i = 0; i < M::NX; ++i) {
                const double d = xn[i] - xs[(k + 1) * M::NX + i];
                dft[k * M::NX + i] = d;
                gl += fabs(d);
            }
        } else {
            Jl += M::term_cost(c, x, P + k * M::NP);
        }
    }
    J = wave_sum(Jl);
    gap = wave_sum(gl);
}

// -----------------------------------------------------------------------------------------------------------------
// backward Riccati sweep.  Returns false when a Quu is not positive definite (caller bumps mu, ddp.py:34-35).
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__device__ bool backward_sweep(const DevConsts& c, int N, const double* __restrict__ P, const double* __restrict__ dft,
                               const double* __restrict__ rec, double* __restrict__ gains, double mu, double* s, int lane,
                               double& dV1, double& G1, double& G2, double& qu_inf SDDP_T_ARG) {
    using L = Lds<M>;
    constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NREC = M::NREC, NP = M::NP;
    constexpr int NCOL = NU + 1 + NX;
    static_assert(NCOL <= kWave, "one lane per augmented column");
    static_assert(NX <= kWave, "one lane per state row");
    dV1 = G1 = G2 = qu_inf = 0.0;
    bool ok = true;
    // ---- terminal node: Vx = lx_N, Vxx = lxx_N (ddp.py:216-226)
    {
        const double* rN = rec + size_t(N) * NREC;
        for (int e = lane; e < NREC; e += kWave) s[L::REC + e] = rN[e];
        for (int e = lane; e < NP; e += kWave) s[L::PK + e] = P[N * NP + e];
        __syncthreads();
        for (int e = lane; e < NX * NX; e += kWave) s[L::VXX + e] = M::H_entry(c, s + L::REC, s + L::PK, N, N, e / NX, e % NX);
        if (lane < NX) s[L::VX + lane] = s[L::REC + M::REC_G + lane];
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {
        // ---- stage this knot (coalesced: one knot per wave-wide access)
        const double* rk = rec + size_t(k) * NREC;
        for (int e = lane; e < NREC; e += kWave) s[L::REC + e] = rk[e];
        for (int e = lane; e < NP; e += kWave) s[L::PK + e] = P[k * NP + e];
        if (lane < NX) s[L::DK + lane] = dft[k * NX + lane];
        __syncthreads();
        SDDP_TICK(1)
        // ---- expand [fx fu], GN Hessian, gradient; v' = Vx + Vxx d ; gap terms
        for (int e = lane; e < NX * NZ; e += kWave) s[L::F + e] = M::F_entry(c, s + L::REC, e / NZ, e % NZ);
        for (int e = lane; e < NZ * NZ; e += kWave) s[L::Q + e] = M::H_entry(c, s + L::REC, s + L::PK, k, N, e / NZ, e % NZ);
        for (int e = lane; e < NZ; e += kWave) s[L::QV + e] = s[L::REC + M::REC_G + e];
        double g1 = 0.0, g2 = 0.0;
        if (lane < NX) {
            double acc = 0.0;
            for (int j = 0; j < NX; ++j) acc += s[L::VXX + lane * NX + j] * s[L::DK + j];
            const double d = s[L::DK + lane], vx = s[L::VX + lane];
            s[L::VP + lane] = vx + acc;
            g1 = d * vx;
            g2 = 0.5 * d * acc;
        }
        G1 += wave_sum(g1);
        G2 += wave_sum(g2);
        __syncthreads();
        SDDP_TICK(2)
        // ---- W = Vxx [fx fu]
        for (int e = lane; e < NX * NZ; e += kWave) {
            const int i = e / NZ, j = e % NZ;
            double acc = 0.0;
            for (int l = 0; l < NX; ++l) acc += s[L::VXX + i * NX + l] * s[L::F + l * NZ + j];
            s[L::W + e] = acc;
        }
        __syncthreads();
        SDDP_TICK(3)
        // ---- Q = H + F^T W ; q = g + F^T v'
        for (int e = lane; e < NZ * NZ; e += kWave) {
            const int i = e / NZ, j = e % NZ;
            double acc = s[L::Q + e];
            for (int l = 0; l < NX; ++l) acc += s[L::F + l * NZ + i] * s[L::W + l * NZ + j];
            s[L::Q + e] = acc;
        }
        for (int e = lane; e < NZ; e += kWave) {
            double acc = s[L::QV + e];
            for (int l = 0; l < NX; ++l) acc += s[L::F + l * NZ + e] * s[L::VP + l];
            s[L::QV + e] = acc;
        }
        __syncthreads();
        SDDP_TICK(4)
        // ---- [k K] = -Quu^-1 [Qu Qux]: Gauss-Jordan, lane j owns column j of [Quu+mu I | Qu | Qux]
        double a[NU];
        if (lane < NCOL) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double v;
                if (lane < NU) v = s[L::Q + (NX + i) * NZ + NX + lane] + (i == lane ? mu : 0.0);
                else if (lane == NU) v = s[L::QV + NX + i];
                else v = s[L::Q + (NX + i) * NZ + (lane - NU - 1)];
                a[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) a[i] = 0.0;
        }
        double qu_abs = 0.0, qu_save[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { qu_save[i] = a[i]; qu_abs = fmax(qu_abs, fabs(a[i])); }
        qu_inf = fmax(qu_inf, __shfl(qu_abs, NU, kWave));
#pragma unroll
        for (int p = 0; p < NU; ++p) {
            if (lane == p) {
#pragma unroll
                for (int i = 0; i < NU; ++i) s[L::PIV + i] = a[i];
            }
            __syncthreads();
            const double piv = s[L::PIV + p];
            if (!(piv > 0.0) || !(piv < 1e300)) ok = false;
            const double t = a[p] / piv;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (i == p) a[i] = t;
                else a[i] -= s[L::PIV + i] * t;
            }
            __syncthreads();
        }
        // a = Quu^-1 * column ; publish kff | K (negated)
        if (lane >= NU && lane < NCOL) {
#pragma unroll
            for (int i = 0; i < NU; ++i) s[L::KT + i * (NX + 1) + (lane - NU)] = -a[i];
        }
        // dV1 += kff . Qu   (dV2 = 1/2 kff^T Quu kff = -1/2 dV1 exactly, not accumulated separately)
        double dv = 0.0;
        if (lane == NU) {
#pragma unroll
            for (int i = 0; i < NU; ++i) dv += -a[i] * qu_save[i];
        }
        dV1 += __shfl(dv, NU, kWave);
        __syncthreads();
        SDDP_TICK(5)
        if (!ok) return false;
        // ---- Vx = Qx + Qux^T kff ; Vxx = sym(Qxx + Qux^T K)
        if (lane < NX) {
            double acc = s[L::QV + lane];
#pragma unroll
            for (int i = 0; i < NU; ++i) acc += s[L::Q + (NX + i) * NZ + lane] * s[L::KT + i * (NX + 1)];
            s[L::VX + lane] = acc;
        }
        for (int e = lane; e < NX * NX; e += kWave) {
            const int i = e / NX, j = e % NX;
            double acc = s[L::Q + i * NZ + j] + s[L::Q + j * NZ + i];
#pragma unroll
            for (int l = 0; l < NU; ++l)
                acc += s[L::Q + (NX + l) * NZ + i] * s[L::KT + l * (NX + 1) + 1 + j] +
                       s[L::Q + (NX + l) * NZ + j] * s[L::KT + l * (NX + 1) + 1 + i];
            s[L::VXX + e] = 0.5 * acc;
        }
        // ---- gains to HBM/L2: kff (NU) then K (NU x NX) row-major
        double* gk = gains + size_t(k) * (NU * (NX + 1));
        for (int e = lane; e < NU * (NX + 1); e += kWave) {
            double v;
            if (e < NU) v = s[L::KT + e * (NX + 1)];
            else { const int i = (e - NU) / NX, j = (e - NU) % NX; v = s[L::KT + i * (NX + 1) + 1 + j]; }
            gk[e] = v;
        }
        __syncthreads();
        SDDP_TICK(6)
    }
    return ok;
}

// -----------------------------------------------------------------------------------------------------------------
// forward pass: one lane per step length.  Every lane rolls the whole horizon with its own alpha; the lane
// `store_lane` also writes its trajectory to xn/un.  OPEN_LOOP: plain rollout of us (single-shooting start).
// -----------------------------------------------------------------------------------------------------------------
template <class M, bool OPEN_LOOP>
__device__ double rollout(const DevConsts& c, int N, const double* __restrict__ x0, const double* __restrict__ P,
                          const double* __restrict__ xs, const double* __restrict__ us, const double* __restrict__ dft,
                          const double* __restrict__ gains, double* __restrict__ xn, double* __restrict__ un,
                          double alpha, int store_lane, int lane) {
    constexpr int NX = M::NX, NU = M::NU;
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    double J = 0.0;
    const double oma = 1.0 - alpha;
    for (int k = 0; k < N; ++k) {
        double u[NU], xnext[NX];
        if (OPEN_LOOP) {
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = us[k * NU + i];
        } else {
            double dx[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) dx[j] = x[j] - xs[k * NX + j];
            const double* gk = gains + size_t(k) * (NU * (NX + 1));
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double acc = us[k * NU + i] + alpha * gk[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += gk[NU + i * NX + j] * dx[j];
                u[i] = acc;
            }
        }
        J += M::step(c, x, u, P + k * M::NP, k, xnext);
        if (lane == store_lane) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[k * NX + i] = x[i];
#pragma unroll
            for (int i = 0; i < NU; ++i) un[k * NU + i] = u[i];
        }
        if (OPEN_LOOP) {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xnext[i];
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xnext[i] - oma * dft[k * NX + i];
        }
    }
    J += M::term_cost(c, x, P + N * M::NP);
    if (lane == store_lane) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[N * NX + i] = x[i];
    }
    return J;
}

// -----------------------------------------------------------------------------------------------------------------
// fused persistent solve: one wavefront per MPC instance, all iterations in one launch (replaces ddp.py:101)
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(kWave) void solve_kernel(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N;
    const sddp_options& o = A.o;
    const double* x0 = A.x0 + size_t(b) * NX;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    double* xn = A.xn + size_t(b) * (N + 1) * NX;
    double* un = A.un + size_t(b) * N * NU;
    double* dft = A.dft + size_t(b) * N * NX;
    double* gains = A.gains + size_t(b) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(b) * (N + 1) * NREC;

    double J = 0.0, gap = 0.0;
    SDDP_T_DECL
    // ---- starting point
    if (o.initial_rollout) {
        J = rollout<M, true>(A.c, N, x0, P, xs, us, dft, gains, xn, un, 0.0, 0, lane);
        __syncthreads();
        for (int e = lane; e < (N + 1) * NX; e += kWave) xs[e] = xn[e];
        for (int e = lane; e < N * NX; e += kWave) dft[e] = 0.0;
        __syncthreads();
    } else {
        if (lane < NX) xs[lane] = x0[lane];
        __syncthreads();
        phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
        __syncthreads();
    }
    double mu = o.mu0, rho = 0.0, alpha = 0.0, expected = 0.0;
    int iters = 0, converged = 0, status = 1, rollouts = 0;
    if (!(fabs(J) < 1e300)) { status = 3; }
    else
        while (iters < o.max_iters) {
            SDDP_TICK(9)
            phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
            __syncthreads();
            SDDP_TICK(0)
            double dV1, G1, G2, qu_inf;
            bool ok;
            while (true) {
                ok = backward_sweep<M>(A.c, N, P, dft, rec, gains, mu, s, lane, dV1, G1, G2, qu_inf SDDP_T_PASS);
                if (ok) break;
                mu = fmax(mu, 0.0) * 10.0 + o.mu_min;
                if (mu > o.mu_max) break;
            }
            if (!ok) { status = 2; break; }
            const double dV2 = -0.5 * dV1;
            expected = -(dV1 + dV2);
            if (expected < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; break; }
            const double A1 = dV1 + G1, B2 = dV2 + G2;
            if (gap > 0.0) rho = fmax(rho, 2.0 * fmax(fmax(A1, A1 + B2), 0.0) / gap);
            const double slack = 1e-13 * (fabs(J) + rho * gap);
            // ---- line search: lane j tries alpha_0 * factor^j (the ladder of ddp.py:20-28 in one pass)
            bool accepted = false;
            double a_base = o.alpha_0, a_win = 0.0, J_win = 0.0;
            while (a_base >= o.alpha_converge_threshold) {
                double a = a_base;
                for (int j = 0; j < lane; ++j) a *= o.line_search_decrease_factor;
                const bool valid = a >= o.alpha_converge_threshold;
                SDDP_TICK(9)
                double Jl = rollout<M, false>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a, 0, lane);
                SDDP_TICK(8)
                ++rollouts;
                const double pred = a * A1 + a * a * B2 - a * rho * gap;
                const double dphi = (Jl + rho * (1.0 - a) * gap) - (J + rho * gap);
                const bool good = valid && (fabs(Jl) < 1e300) && (dphi <= o.beta * pred + slack);
                const unsigned long long mask = __ballot(good);
                if (mask) {
                    const int win = __ffsll((long long)mask) - 1;
                    a_win = __shfl(a, win, kWave);
                    J_win = __shfl(Jl, win, kWave);
                    if (win != 0) {
                        __syncthreads();
                        rollout<M, false>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a, win, lane);
                        ++rollouts;
                    }
                    accepted = true;
                    break;
                }
                a_base = __shfl(a, kWave - 1, kWave) * o.line_search_decrease_factor;
            }
            if (!accepted) { alpha = 0.0; converged = 1; status = 0; break; }  // alpha fell below alpha_converge_threshold
            alpha = a_win;
            const double dJ = J - J_win;
            J = J_win;
            __syncthreads();
            { double* t = xs; xs = xn; xn = t; }
            { double* t = us; us = un; un = t; }
            const double oma = 1.0 - alpha;
            for (int e = lane; e < N * NX; e += kWave) dft[e] *= oma;
            gap *= oma;
            ++iters;
            if (mu > o.mu0) mu = fmax(o.mu0, mu * 0.1);
            __syncthreads();
            if (fabs(dJ) < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; break; }
        }
    // ---- results live in A.xs/A.us: copy back if the iterate ended in the candidate buffers
    __syncthreads();
    double* xs0 = A.xs + size_t(b) * (N + 1) * NX;
    if (xs != xs0) {
        double* us0 = A.us + size_t(b) * N * NU;
        for (int e = lane; e < (N + 1) * NX; e += kWave) xs0[e] = xs[e];
        for (int e = lane; e < N * NU; e += kWave) us0[e] = us[e];
    }
#ifdef SDDP_STAMPS
    SDDP_TICK(9)
    if (lane == 0) for (int i = 0; i < kScal; ++i) A.scal[size_t(b) * kScal + i] = (double)T_[i];
#endif
    if (lane == 0) {
        sddp_stats st;
        st.cost = J; st.alpha = alpha; st.gap = gap; st.mu = mu; st.expected = expected;
        st.iters = iters; st.converged = converged; st.status = status; st.rollouts = rollouts;
        A.stats[b] = st;
    }
}

// -----------------------------------------------------------------------------------------------------------------
// single-phase kernels for the parity tests (same device code as the fused kernel)
// -----------------------------------------------------------------------------------------------------------------
// one wavefront per knot: lane 0 runs the scalar model code, all lanes expand the dense tiles
template <class M>
__global__ __launch_bounds__(kWave) void eval_knots_kernel(DevConsts c, int N, int nk, const int* __restrict__ kk,
                                                           const double* __restrict__ x, const double* __restrict__ u,
                                                           const double* __restrict__ p, double* __restrict__ rec,
                                                           double* f_out, double* F_out, double* H_out, double* g_out,
                                                           double* L_out) {
    constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NP = M::NP, NREC = M::NREC;
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= nk) return;
    const int k = kk[t];
    double* r = rec + size_t(t) * NREC;
    if (lane == 0) {
        double xl[NX], ul[NU], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xl[i] = x[t * NX + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) ul[i] = u[t * NU + i];
        M::derivs(c, xl, ul, p + t * NP, k, N, r);
        double L;
        if (k < N) {
            L = M::step(c, xl, ul, p + t * NP, k, xn);
        } else {
            L = M::term_cost(c, xl, p + t * NP);
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = xl[i];
        }
        L_out[t] = L;
#pragma unroll
        for (int i = 0; i < NX; ++i) f_out[t * NX + i] = xn[i];
    }
    __syncthreads();
    for (int e = lane; e < NX * NZ; e += kWave) F_out[size_t(t) * NX * NZ + e] = k < N ? M::F_entry(c, r, e / NZ, e % NZ) : 0.0;
    for (int e = lane; e < NZ * NZ; e += kWave) {
        const int i = e / NZ, j = e % NZ;
        H_out[size_t(t) * NZ * NZ + e] = (k < N || (i < NX && j < NX)) ? M::H_entry(c, r, p + t * NP, k, N, i, j) : 0.0;
    }
    for (int e = lane; e < NZ; e += kWave) g_out[t * NZ + e] = r[M::REC_G + e];
}

template <class M>
__global__ __launch_bounds__(kWave) void backward_kernel(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    double* dft = A.dft + size_t(b) * N * NX;
    double* gains = A.gains + size_t(b) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(b) * (N + 1) * NREC;
    double J, gap;
    phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
    phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
    __syncthreads();
    double dV1, G1, G2, qu_inf;
    SDDP_T_DECL
    const bool ok = backward_sweep<M>(A.c, N, P, dft, rec, gains, A.mu, s, lane, dV1, G1, G2, qu_inf SDDP_T_PASS);
    if (lane == 0) {
        double* sc = A.scal + size_t(b) * kScal;
        sc[0] = dV1; sc[1] = -0.5 * dV1; sc[2] = G1; sc[3] = G2; sc[4] = ok ? 1.0 : 0.0; sc[5] = A.mu; sc[6] = qu_inf; sc[7] = J;
    }
}

template <class M>
__global__ __launch_bounds__(kWave) void forward_kernel(SolveArgs A) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N;
    const double J = rollout<M, false>(A.c, N, A.x0 + size_t(b) * NX, A.P + size_t(b) * (N + 1) * NP,
                                       A.xs + size_t(b) * (N + 1) * NX, A.us + size_t(b) * N * NU,
                                       A.dft + size_t(b) * N * NX, A.gains + size_t(b) * N * (NU * (NX + 1)),
                                       A.xn + size_t(b) * (N + 1) * NX, A.un + size_t(b) * N * NU, A.alpha, 0, lane);
    if (lane == 0) A.scal[size_t(b) * kScal] = J;
}

}  // namespace sddp
