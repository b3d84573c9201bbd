// sddp_kernels_mw.hpp -- the same DDP iteration as sddp_kernels.hpp for the LARGE models (srbd37: 37x37 / 61x61 tiles,
// lip30), mapped on one 256-thread workgroup (4 wavefronts, one per SIMD of a CU) per MPC instance:
//   * the tile phases of the Riccati sweep run one THREAD per tile element (the 1891 lower-triangle elements of Q at srbd37
//     are 30 rounds of one wavefront but 8 rounds of four), separated by workgroup barriers;
//   * wave 0 owns the scalar solver state and runs the lane-parallel phases (derivatives: lane per knot; line search: lane
//     per step length, per-lane vectors in LDS columns; Quu Gauss-Jordan: lane per right-hand side); its decisions reach
//     the other waves through the LDS control words CTL[1..2];
//   * these models need > 48 KB of LDS per instance, i.e. at most two workgroups per CU, so the 4 waves do not fight the
//     register budget that rules this mapping out for srbd13 at four instances per CU (experiment log).
// Same device model code, same arithmetic, same parity tests as the single-wavefront kernel.
#pragma once
#include "sddp_kernels.hpp"

namespace sddp {

constexpr int kThreadsMW = 256;
constexpr int kLastWaveMW = kThreadsMW / kWave - 1;

template <class M>
struct LdsMW {
    static constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NI = NX + NE;
    static constexpr int NXP = (NX + 1) & ~1;
    static constexpr int NIP = (NI + 1) & ~1;
    static constexpr int NZP = (NZ + 3) & ~3;
    static constexpr int NUP = (NU + 1) & ~1;
    static constexpr int NSTG = M::NREC + M::NP + NX;          // one knot's staged operands: record | params | defect
    static constexpr int NSTGP = (NSTG + 1) & ~1;
    static constexpr int VXX = 0;
    static constexpr int FT = VXX + NXP * NXP;
    static constexpr int WT = FT + NZP * NIP;                  // WT and Q are adjacent: the wide rollout's columns alias them
    static constexpr int Q = WT + NZP * NIP;
    static constexpr int VX = Q + NZP * NZP;
    static constexpr int VP = VX + NXP;
    static constexpr int QV = VP + NXP;
    static constexpr int STG = QV + NZP;
    static constexpr int REC = STG, PK = STG + M::NREC, DK = PK + M::NP;
    static constexpr int KT = STG + NSTGP;             // [NXP][NUP]: KT[c][i] = K[i][c]
    static constexpr int KF = KT + NXP * NUP;          // kff [NUP]
    static constexpr int DS = KF + NUP;                // constant diagonal, state part   [NZP]
    static constexpr int DG = DS + NZP;                // constant diagonal, stage part   [NZP]
    static constexpr int LS = DG + NZP;                // extra-row weights, state / stage [NE] each
    static constexpr int LG = LS + ((NE + 1) & ~1);
    static constexpr int CTL = LG + ((NE + 1) & ~1);   // control words shared by the 4 waves [8]
    static constexpr int KI = CTL + 8;                 // ints: dkind[NZP], dci[NZP], lower-triangle element LUTs of Q and Vxx
    static constexpr int NTRIQ = NZ * (NZ + 1) / 2, NTRIV = NX * (NX + 1) / 2;
    static constexpr int KI_INTS = 2 * NZP + NTRIQ + NTRIV;
    static constexpr int TOTAL = KI + (KI_INTS + 1) / 2;
    static constexpr size_t BYTES = size_t(TOTAL) * sizeof(double);
    static constexpr bool WIDE = (NX > 16) && ((2 * NX + NU) * kWave <= NZP * NIP + NZP * NZP);
};

template <class M>
__device__ void sweep_tables_mw(const DevConsts& c, double* s, int tid) {
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NZ = M::NZ, NE = M::NE;
    for (int e = tid; e < L::TOTAL; e += kThreadsMW) s[e] = 0.0;    // also: zero record -> constant part of F below
    __syncthreads();
    for (int e = tid; e < NZ * NX; e += kThreadsMW) {
        const int j = e / NX, i = e % NX;
        s[L::FT + j * L::NIP + i] = M::F_entry(c, s + L::REC, i, j);
    }
    for (int e = tid; e < NZ * NE; e += kThreadsMW) {
        const int j = e / NE, m = e % NE;
        s[L::FT + j * L::NIP + NX + m] = M::E_const(c, m, j);
    }
    int* ki = reinterpret_cast<int*>(s + L::KI);
    for (int i = tid; i < NZ; i += kThreadsMW) {
        s[L::DS + i] = M::dg_state(c, i);
        s[L::DG + i] = M::dg_stage(c, i);
        ki[i] = M::dkind(i);
        ki[L::NZP + i] = M::dci(i);
    }
    for (int m = tid; m < NE; m += kThreadsMW) {
        s[L::LS + m] = M::lam_state(c, m);
        s[L::LG + m] = M::lam_stage(c, m);
    }
    for (int t = tid; t < L::NTRIQ + L::NTRIV; t += kThreadsMW) {   // t -> (a << 8) | b with a >= b
        const int tt = t < L::NTRIQ ? t : t - L::NTRIQ;
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= tt) ++a;
        ki[2 * L::NZP + t] = (a << 8) | (tt - a * (a + 1) / 2);
    }
    __syncthreads();
}

// backward Riccati sweep on 4 waves.  Returns false (to every thread) when a Quu is not positive definite.
// dV1 / G1 / G2 / qu_inf are valid in wave 0 only.
template <class M>
__device__ bool backward_sweep_mw(const DevConsts& c, int N, const double* __restrict__ P, const double* __restrict__ dft,
                                  const double* __restrict__ rec, double* __restrict__ gains, double mu, double theta,
                                  double* s, int tid, double& dV1, double& G1, double& G2, double& qu_inf SDDP_T_ARG) {
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NREC = M::NREC, NP = M::NP;
    constexpr int NXP = L::NXP, NIP = L::NIP, NZP = L::NZP, NUP = L::NUP, NSTG = L::NSTG;
    constexpr int NCOL = NU + 1 + NX;
    constexpr int RS = (NSTG + kThreadsMW - 1) / kThreadsMW;
    static_assert(NCOL <= kWave, "one lane per augmented column");
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const int* ki = reinterpret_cast<const int*>(s + L::KI);
    dV1 = G1 = G2 = qu_inf = 0.0;
    auto stage_word = [&](int k, int w) -> double {    // [0,NREC) record | [NREC,NREC+NP) parameters | then the defect
        if (w < NREC) return rec[size_t(k) * NREC + w];
        if (w < NREC + NP) return P[k * NP + (w - NREC)];
        if (w < NSTG) return dft[k * NX + (w - NREC - NP)];
        return 0.0;
    };
    // ---- terminal node: Vx = lx_N, Vxx = lxx_N = diag(D_state) + Je^T Lambda_state Je  (ddp.py:216-226)
    if (tid < NXP) s[L::VX + tid] = tid < NX ? rec[size_t(N) * NREC + M::REC_G + tid] : 0.0;
    if (tid < NP) s[L::PK + tid] = P[N * NP + tid];
    __syncthreads();
    for (int e = tid; e < NXP * NXP; e += kThreadsMW) {
        const int a = e / NXP, b = e % NXP;
        double v = 0.0;
        if (a < NX && b < NX) {
            for (int m = 0; m < NE; ++m) v += s[L::LS + m] * s[L::FT + a * NIP + NX + m] * s[L::FT + b * NIP + NX + m];
            if (a == b) v += s[L::DS + a] + M::dparam(c, s + L::PK, ki[a], ki[NZP + a], 1.0, 0.0);
        }
        s[L::VXX + e] = v;
    }
    double r_stage[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) r_stage[t] = stage_word(N - 1, tid + t * kThreadsMW);
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        // ---- stage this knot from the prefetch registers; start the next knot's loads
#pragma unroll
        for (int t = 0; t < RS; ++t)
            if (tid + t * kThreadsMW < NSTG) s[L::STG + tid + t * kThreadsMW] = r_stage[t];
        if (k > 0) {
#pragma unroll
            for (int t = 0; t < RS; ++t) r_stage[t] = stage_word(k - 1, tid + t * kThreadsMW);
        }
        __syncthreads();
        SDDP_TICK(1)
        const double state = k >= 1 ? 1.0 : 0.0;
        // ---- expand the variable entries of F~^T (all threads) ; v' = Vx + Vxx d and the gap terms (wave 0)
        M::expand_var(c, s + L::REC, s + L::FT, NIP, tid, kThreadsMW);
        if (wave == 0) {
            double g1 = 0.0, g2 = 0.0;
            if (lane < NX) {
                double acc = 0.0;
                for (int m = 0; m < NX; ++m) acc += s[L::VXX + lane * NXP + m] * s[L::DK + m];
                const double d = s[L::DK + lane], vx = s[L::VX + lane];
                s[L::VP + lane] = vx + acc;
                g1 = d * vx;
                g2 = 0.5 * d * acc;
            }
            G1 += wave_sum(g1);
            G2 += wave_sum(g2);
        }
        __syncthreads();
        SDDP_TICK(2)
        // ---- WT = (V~ F~)^T : one thread per element; dynamics rows by an NX-deep product, extra rows by a scaling
        for (int e = tid; e < NZ * NX; e += kThreadsMW) {
            const int j = e / NX, l = e % NX;
            double acc = 0.0;
#pragma unroll 4
            for (int m = 0; m < NXP; m += 2) {
                const double2_t v = lds2(s + L::VXX + l * NXP + m), f = lds2(s + L::FT + j * NIP + m);
                acc += v.x * f.x + v.y * f.y;                    // Vxx pad column is zero: the FT word beyond NX is harmless
            }
            s[L::WT + j * NIP + l] = acc;
        }
        for (int e = tid; e < NZ * NE; e += kThreadsMW) {
            const int j = e / NE, m = e % NE;
            const double lam = state * s[L::LS + m] + s[L::LG + m];
            s[L::WT + j * NIP + NX + m] = lam * s[L::FT + j * NIP + NX + m];
        }
        __syncthreads();
        SDDP_TICK(3)
        // ---- Q = diag(D) + F~^T (V~ F~): one thread per lower-triangle element, mirrored ; q = g + F^T v'
        for (int t = tid; t < L::NTRIQ; t += kThreadsMW) {
            const int code = ki[2 * NZP + t];
            const int a = code >> 8, b = code & 255;
            double acc = 0.0;
#pragma unroll 4
            for (int l = 0; l < NIP; l += 2) {
                const double2_t f = lds2(s + L::FT + a * NIP + l), w = lds2(s + L::WT + b * NIP + l);
                acc += f.x * w.x + f.y * w.y;
            }
            if (a == b) acc += state * s[L::DS + a] + s[L::DG + a] + M::dparam(c, s + L::PK, ki[a], ki[NZP + a], state, 1.0);
            s[L::Q + a * NZP + b] = acc;
            s[L::Q + b * NZP + a] = acc;
        }
        if (wave == kLastWaveMW) {
            for (int j = lane; j < NZ; j += kWave) {
                double acc = s[L::REC + M::REC_G + j];
                for (int m = 0; m < NX; ++m) acc += s[L::FT + j * NIP + m] * s[L::VP + m];
                s[L::QV + j] = acc;
            }
        }
        __syncthreads();
        if (theta != 0.0) {   // exact second-order torque term (uniform switch)
            M::add_second_order(c, s + L::REC, s + L::VP, s + L::Q, NZP, theta, tid, kThreadsMW);
            __syncthreads();
        }
        SDDP_TICK(4)
        // ---- [k K] = -Quu^-1 [Qu Qux]: Gauss-Jordan in wave 0, lane j owns column j of [Quu+mu I | Qu | Qux]
        if (wave == 0) {
            double a[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double v = 0.0;
                if (lane < NU) v = s[L::Q + (NX + i) * NZP + NX + lane] + (i == lane ? mu : 0.0);
                else if (lane == NU) v = s[L::QV + NX + i];
                else if (lane < NCOL) v = s[L::Q + (NX + i) * NZP + (lane - NU - 1)];
                a[i] = v;
            }
            double qu_abs = 0.0, dv = 0.0, qu_save[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) { qu_save[i] = a[i]; qu_abs = fmax(qu_abs, fabs(a[i])); }
            qu_inf = fmax(qu_inf, readlane_d(qu_abs, NU));
            bool ok = true;
#pragma unroll
            for (int p = 0; p < NU; ++p) {
                double pv[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) pv[i] = readlane_d(a[i], p);
                if (!(pv[p] > 0.0) || !(pv[p] < 1e300)) ok = false;
                const double t = a[p] * fast_rcp(pv[p]);
#pragma unroll
                for (int i = 0; i < NU; ++i) a[i] = (i == p) ? t : fma(-pv[i], t, a[i]);
            }
            if (lane == NU) {
#pragma unroll
                for (int i = 0; i < NU; ++i) s[L::KF + i] = -a[i];
            }
            if (lane > NU && lane < NCOL) {
#pragma unroll
                for (int i = 0; i < NU; ++i) s[L::KT + (lane - NU - 1) * NUP + i] = -a[i];
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) dv += -a[i] * qu_save[i];
            dV1 += readlane_d(dv, NU);
            if (lane == 0) s[L::CTL + 0] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        SDDP_TICK(5)
        if (s[L::CTL + 0] == 0.0) return false;
        // ---- Vx = Qx + Qux^T kff ; Vxx = Qxx + 1/2 (Qux^T K + K^T Qux): one thread per lower-triangle element, mirrored
        if (wave == kLastWaveMW && lane < NX) {
            double acc = s[L::QV + lane];
            for (int i = 0; i < NU; ++i) acc += s[L::Q + lane * NZP + NX + i] * s[L::KF + i];
            s[L::VX + lane] = acc;
        }
        for (int t = tid; t < L::NTRIV; t += kThreadsMW) {
            const int code = ki[2 * NZP + L::NTRIQ + t];
            const int a = code >> 8, b = code & 255;
            double acc = 0.0;
#pragma unroll 4
            for (int i = 0; i < NU; ++i)
                acc += s[L::Q + a * NZP + NX + i] * s[L::KT + b * NUP + i] + s[L::Q + b * NZP + NX + i] * s[L::KT + a * NUP + i];
            acc = s[L::Q + a * NZP + b] + 0.5 * acc;
            s[L::VXX + a * NXP + b] = acc;
            s[L::VXX + b * NXP + a] = acc;
        }
        // ---- gains to HBM/L2: kff (NU) then K (NU x NX) row-major, coalesced
        {
            double* gk = gains + size_t(k) * (NU * (NX + 1));
            for (int e = tid; e < NU * (NX + 1); e += kThreadsMW) {
                double v;
                if (e < NU) v = s[L::KF + e];
                else { const int i = (e - NU) / NX, j = (e - NU) % NX; v = s[L::KT + j * NUP + i]; }
                gk[e] = v;
            }
        }
        __syncthreads();
        SDDP_TICK(6)
    }
    return true;
}

// forward pass on ONE wave, no workgroup barrier inside (the other waves wait at the next barrier meanwhile): per-lane
// vectors in LDS columns for the wide models, in registers with direct (wave-uniform) operand loads otherwise
template <class M, bool OPEN_LOOP>
__device__ double rollout_w(const DevConsts& c, int N, const double* __restrict__ x0, const double* __restrict__ P,
                            const double* __restrict__ xs, const double* __restrict__ us, const double* __restrict__ dft,
                            const double* __restrict__ gains, double* __restrict__ xn, double* __restrict__ un,
                            double alpha, int store_lane, int lane, double* s SDDP_T_ARG) {
    if constexpr (LdsMW<M>::WIDE) {
        return rollout_lds_core<M, OPEN_LOOP>(c, N, x0, P, xs, us, dft, gains, xn, un, alpha, store_lane, lane, s + LdsMW<M>::WT SDDP_T_PASS);
    } else {
        constexpr int NX = M::NX, NU = M::NU;
        double x[NX];
        for (int i = 0; i < NX; ++i) x[i] = x0[i];
        double J = 0.0;
        const double oma = 1.0 - alpha;
        for (int k = 0; k < N; ++k) {
            double u[NU], xnext[NX];
            if (OPEN_LOOP) {
                for (int i = 0; i < NU; ++i) u[i] = us[k * NU + i];
            } else {
                double dx[NX];
                for (int j = 0; j < NX; ++j) dx[j] = x[j] - xs[k * NX + j];
                const double* gk = gains + size_t(k) * (NU * (NX + 1));
                for (int i = 0; i < NU; ++i) {
                    double acc = us[k * NU + i] + alpha * gk[i];
                    for (int j = 0; j < NX; ++j) acc += gk[NU + i * NX + j] * dx[j];
                    u[i] = acc;
                }
            }
            J += M::step(c, x, u, P + k * M::NP, k, xnext);
            if (lane == store_lane) {
                for (int i = 0; i < NX; ++i) xn[k * NX + i] = x[i];
                for (int i = 0; i < NU; ++i) un[k * NU + i] = u[i];
            }
            for (int i = 0; i < NX; ++i) x[i] = OPEN_LOOP ? xnext[i] : xnext[i] - oma * dft[k * NX + i];
        }
        J += M::term_cost(c, x, P + N * M::NP);
        if (lane == store_lane) {
            for (int i = 0; i < NX; ++i) xn[N * NX + i] = x[i];
        }
        return J;
    }
}

// fused persistent solve, 4 waves per instance.  Control words: CTL[1] = code (0 step accepted, 1 accepted and stop,
// 2 stop without a step, 3 redo the iteration with theta = 0, 4 non-finite start), CTL[2] = accepted alpha.
template <class M>
__global__ __launch_bounds__(kThreadsMW) void solve_kernel_mw(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
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
    sweep_tables_mw<M>(A.c, s, tid);
    if (o.initial_rollout) {
        if (wave == 0) J = rollout_w<M, true>(A.c, N, x0, P, xs, us, dft, gains, xn, un, 0.0, 0, lane, s SDDP_T_PASS);
        __syncthreads();
        for (int e = tid; e < (N + 1) * NX; e += kThreadsMW) xs[e] = xn[e];
        for (int e = tid; e < N * NX; e += kThreadsMW) dft[e] = 0.0;
    } else {
        if (tid < NX) xs[tid] = x0[tid];
        __syncthreads();
        if (wave == 0) phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
    }
    if (tid == 0) s[L::CTL + 1] = (fabs(J) < 1e300) ? 0.0 : 4.0;
    __syncthreads();
    double mu = o.mu0, rho = 0.0, alpha = 0.0, expected = 0.0, theta = 0.0;
    int iters = 0, converged = 0, status = 1, rollouts = 0, guess = 0;
    if (s[L::CTL + 1] == 4.0) status = 3;
    else
        while (iters < o.max_iters) {
            SDDP_TICK(9)
            if (wave == 0) phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
            SDDP_TICK(0)
            if (L::WIDE)   // the wide rollout's columns lived in the WT / Q tiles: restore the zero pads of WT
                for (int e = tid; e < L::NZP * L::NIP; e += kThreadsMW) s[L::WT + e] = 0.0;
            __syncthreads();
            double dV1, G1, G2, qu_inf;
            bool ok;
            while (true) {
                ok = backward_sweep_mw<M>(A.c, N, P, dft, rec, gains, mu, theta, s, tid, dV1, G1, G2, qu_inf SDDP_T_PASS);
                if (ok) break;
                if (theta != 0.0) { theta = 0.0; continue; }      // identical in every wave
                mu = fmax(mu, 0.0) * 10.0 + o.mu_min;
                if (mu > o.mu_max) break;
            }
            if (!ok) { status = 2; break; }
            if (wave == 0) {   // convergence test, merit weight, line search
                double code = 0.0;
                const double dV2 = -0.5 * dV1;
                expected = -(dV1 + dV2);
                if (expected < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; code = 2.0; }
                else {
                    const double A1 = dV1 + G1, B2 = dV2 + G2;
                    if (gap > 0.0) rho = fmax(rho, 2.0 * fmax(fmax(A1, A1 + B2), 0.0) / gap);
                    const double slack = 1e-13 * (fabs(J) + rho * gap);
                    bool accepted = false;
                    double a_base = o.alpha_0, a_win = 0.0, J_win = 0.0;
                    while (a_base >= o.alpha_converge_threshold) {
                        double a = a_base;
                        for (int j = 0; j < lane; ++j) a *= o.line_search_decrease_factor;
                        const bool valid = a >= o.alpha_converge_threshold;
                        SDDP_TICK(9)
                        const double Jl = rollout_w<M, false>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a, guess, lane, s SDDP_T_PASS);
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
                            if (win != guess) {
                                rollout_w<M, false>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a, win, lane, s SDDP_T_PASS);
                                ++rollouts;
                            }
                            guess = win;
                            accepted = true;
                            break;
                        }
                        a_base = __shfl(a, kWave - 1, kWave) * o.line_search_decrease_factor;
                        guess = 0;
                    }
                    if (!accepted) {
                        if (theta != 0.0) code = 3.0;                                          // redo with plain Gauss-Newton
                        else { alpha = 0.0; converged = 1; status = 0; code = 2.0; }           // alpha below the threshold
                    } else {
                        alpha = a_win;
                        const double dJ = J - J_win;
                        J = J_win;
                        gap *= (1.0 - alpha);
                        ++iters;
                        if (fabs(dJ) < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; code = 1.0; }
                    }
                }
                if (lane == 0) { s[L::CTL + 1] = code; s[L::CTL + 2] = alpha; }
            }
            __syncthreads();
            const double code = s[L::CTL + 1];
            if (code == 2.0) break;
            if (code == 3.0) { theta = 0.0; __syncthreads(); continue; }
            const double a_acc = s[L::CTL + 2];
            { double* t = xs; xs = xn; xn = t; }
            { double* t = us; us = un; un = t; }
            const double oma = 1.0 - a_acc;
            for (int e = tid; e < N * NX; e += kThreadsMW) dft[e] *= oma;
            if (wave != 0) ++iters;
            theta = (o.second_order && a_acc == o.alpha_0) ? 1.0 : 0.0;
            if (mu > o.mu0) mu = fmax(o.mu0, mu * 0.1);
            __syncthreads();
            if (code == 1.0) break;
        }
    __syncthreads();
    double* xs0 = A.xs + size_t(b) * (N + 1) * NX;
    if (xs != xs0) {
        double* us0 = A.us + size_t(b) * N * NU;
        for (int e = tid; e < (N + 1) * NX; e += kThreadsMW) xs0[e] = xs[e];
        for (int e = tid; e < N * NU; e += kThreadsMW) us0[e] = us[e];
    }
#ifdef SDDP_STAMPS
    SDDP_TICK(9)
    if (tid == 0) for (int i = 0; i < kScal; ++i) A.scal[size_t(b) * kScal + i] = (double)T_[i];
#endif
    if (tid == 0) {
        sddp_stats st;
        st.cost = J; st.alpha = alpha; st.gap = gap; st.mu = mu; st.expected = expected;
        st.iters = iters; st.converged = converged; st.status = status; st.rollouts = rollouts;
        A.stats[b] = st;
    }
}

template <class M>
__global__ __launch_bounds__(kThreadsMW) void backward_kernel_mw(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (b >= A.B) return;
    const int N = A.N;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    double* dft = A.dft + size_t(b) * N * NX;
    double* gains = A.gains + size_t(b) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(b) * (N + 1) * NREC;
    double J = 0.0, gap = 0.0;
    sweep_tables_mw<M>(A.c, s, tid);
    if (wave == 0) {
        phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
        phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
    }
    __syncthreads();
    double dV1, G1, G2, qu_inf;
    SDDP_T_DECL
    const bool ok = backward_sweep_mw<M>(A.c, N, P, dft, rec, gains, A.mu, A.alpha, s, tid, dV1, G1, G2, qu_inf SDDP_T_PASS);
    if (tid == 0) {
        double* sc = A.scal + size_t(b) * kScal;
        sc[0] = dV1; sc[1] = -0.5 * dV1; sc[2] = G1; sc[3] = G2; sc[4] = ok ? 1.0 : 0.0; sc[5] = A.mu; sc[6] = qu_inf; sc[7] = J;
    }
}

}  // namespace sddp
