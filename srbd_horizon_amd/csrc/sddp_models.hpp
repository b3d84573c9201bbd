// sddp_models.hpp -- hand-written analytic device models for the SRBD / LIP DDP engine (gfx950, fp64).
//
// Replaces the CasADi graphs the reference builds in python/prb.py and hands to pyddp through
// python/ddp.py:179-230 (f_k = explicit Euler of xdot, L_k = sum ||res||^2 + 1e6 sum ||g||^2, L_N).
//
// A model provides
//   * per-lane scalar code (one lane = one knot or one line-search candidate; everything in registers):
//       step()      x+ = f(x,u,p) and the stage cost L_k(x,u,p) in one pass (they share wdot)
//       term_cost() L_N(x,p)
//       derivs()    compact derivative record of one knot: the only non-constant Jacobian blocks
//                   (A = d wdot / d z, quaternion-rate blocks) and the cost gradient
//   * wave-cooperative expansion used inside the Riccati sweep (one lane = one tile element):
//       F_entry(i,j)  element of [fx fu]
//       H_entry(i,j)  element of the Gauss-Newton Hessian  D(p_k) + 2 g_q A^T A
#pragma once
#include <hip/hip_runtime.h>

#include "sddp.h"

namespace sddp {

constexpr double kGravity = 9.81;          // prb.py:243, prb.py:317
constexpr double kConstraintWeight = 1e6;  // ddp.py:181

// constants derived on the host from sddp_model_consts, passed to kernels by value (kernarg -> SGPRs)
struct DevConsts {
    double dt, inv_ms, Is[9], com_z;
    double w_rz, w_rd, w_w, w_rel, w_f, w_sw, gq, w_pen, w_zmp, w_rxy, eta2;
    double w_rv;                // weight of the relative-velocity constraints inside a foot (prb.py:166-170): w_pen, or 0 with contact_model = 1
    double d1x, d1y, d2x, d2y;  // rel_pos offsets d1 = p2 - p0, d2 = p3 - p1 (prb.py:153-154)
    double lever;
    double mu_lin, bar_w, bar_s;   // friction-cone exponential barrier (BAR models only): linearised coefficient, weight, sharpness
    int inertia_mode;
    double box_w, box_s;           // bound barrier (BAR models only, sddp.h): weight (0: off), sharpness
    unsigned long long box_lm, box_um;   // ... bit j: z_j has a finite lower / upper bound
    const double* box;             // ... device array lower[64] | upper[64] (a pointer, not 128 kernel-argument words: those are
                                   //     loop-invariant scalar loads the compiler hoists and keeps live -- 3.9 KB of scratch)
    // user-declared linear residual rows ("_x" builds, sddp.h extra_*): rows in use, and the device table
    //   a [kXrRows][kXrStride] | weight as a state row [8] | weight as a stage row [8] | constant part of the reference [8]
    int xr_n;
    const double* xr;
};
constexpr int kXrRows = 8, kXrStride = 128, kXrWS = kXrRows * kXrStride, kXrWG = kXrWS + 8, kXrC = kXrWG + 8, kXrWords = kXrC + 8;

// The user rows of one knot: cost sum_j w_j e_j^2 with e_j = a_j . z - (p_ref[j] + const_j), w_j = the row's weight where its kind is
// active (state rows: nodes >= 1, terminal included; stage rows: nodes < N), and -- g != nullptr -- the gradient 2 w_j e_j a_j added
// into g[NX + NU].  Compile-time indices into x / u (register arrays); the coefficients are wave-uniform loads.
template <int NX, int NU, class XV, class UV>
__device__ __forceinline__ double xr_eval(const DevConsts& c, XV x, UV u, bool has_u, const double* pref, bool state_on, bool stage_on,
                                          double* g = nullptr) {
    double L = 0.0;
    for (int j = 0; j < c.xr_n; ++j) {
        const double* a = c.xr + j * kXrStride;
        const double w = (state_on ? c.xr[kXrWS + j] : 0.0) + (stage_on ? c.xr[kXrWG + j] : 0.0);
        double e = -pref[j] - c.xr[kXrC + j];
#pragma unroll
        for (int i = 0; i < NX; ++i) e = fma(a[i], x[i], e);
        if (has_u) {
#pragma unroll
            for (int i = 0; i < NU; ++i) e = fma(a[NX + i], u[i], e);
        }
        L = fma(w * e, e, L);
        if (g) {
            const double t = 2.0 * w * e;
#pragma unroll
            for (int i = 0; i < NX; ++i) g[i] = fma(t, a[i], g[i]);
            if (has_u) {
#pragma unroll
                for (int i = 0; i < NU; ++i) g[NX + i] = fma(t, a[NX + i], g[NX + i]);
            }
        }
    }
    return L;
}

inline DevConsts make_dev_consts(const sddp_model_consts& c) {
    DevConsts d;
    d.dt = c.dt;
    d.inv_ms = c.force_scaling / c.m;                      // 1 / (m / force_scaling)   prb.py:99
    for (int i = 0; i < 9; ++i) d.Is[i] = c.I[i] / c.force_scaling;
    d.com_z = c.com[2];
    d.w_rz = c.r_tracking_gain;                            // residual sqrt(g)*e -> cost g*e^2 (prb.py:184)
    d.w_rxy = c.r_tracking_gain;                           // prb.py:391
    d.w_rd = c.rdot_tracking_gain;                         // prb.py:190
    d.w_w = c.w_tracking_gain;                             // prb.py:191
    d.w_rel = c.rel_pos_gain;                              // prb.py:192-199
    d.w_f = c.force_scaling * c.force_scaling * c.min_f_gain;            // prb.py:202
    d.w_sw = c.force_scaling * c.force_scaling * c.force_switch_weight;  // prb.py:203-204
    d.gq = c.min_qddot_gain;                               // prb.py:200
    d.w_pen = kConstraintWeight;                           // ddp.py:181,:196
    d.w_rv = c.relative_velocity_constraints ? kConstraintWeight : 0.0;   // prb.py:166: only `if contact_model > 1`
    d.w_zmp = c.zmp_tracking_gain;                         // prb.py:393
    d.eta2 = kGravity / c.lip_height;                      // prb.py:317
    d.d1x = c.feet[6] - c.feet[0];
    d.d1y = c.feet[7] - c.feet[1];
    d.d2x = c.feet[9] - c.feet[3];
    d.d2y = c.feet[10] - c.feet[4];
    d.lever = c.lever_sign;
    d.inertia_mode = c.inertia_mode;
    d.mu_lin = c.friction_cone_coefficient / sqrt(2.0);
    d.bar_w = c.friction_barrier_weight;
    d.bar_s = c.friction_barrier_sharpness;
    d.box_w = c.bound_barrier_weight;
    d.box_s = c.bound_barrier_sharpness;
    d.xr_n = 0;
    d.xr = nullptr;
    d.box_lm = d.box_um = 0;
    for (int i = 0; i < 64; ++i) {
        if (c.lower[i] > -1e300) d.box_lm |= 1ull << i;
        if (c.upper[i] < 1e300) d.box_um |= 1ull << i;
    }
    d.box = nullptr;               // set by the caller that owns the device copy of the bounds (sddp_api.hip)
    return d;
}

// Per-lane vector kept as an LDS column (element i of lane l at p[i*64 + l]): lets the wide models (srbd37) roll out without
// 135 doubles of per-lane registers.  Reads/writes touch only the lane's own column: no bank conflict, no synchronisation.
struct LdsCol {
    double* p;
    __device__ __forceinline__ double& operator[](int i) const { return p[i * 64]; }
    __device__ __forceinline__ LdsCol operator+(int off) const { return LdsCol{p + off * 64}; }
};

// Store-only view of such a column that closes a knot of the multiple-shooting rollout while the model step writes its result:
// element i receives v - oma * d[i]  (x_{k+1} = f(x_k, u_k) - (1 - alpha) d_k; d: the knot's staged defect, broadcast reads)
struct LdsColClose {
    double* p; const double* d; double oma;
    struct Ref {
        double* q; double di, oma;
        __device__ __forceinline__ void operator=(double v) const { *q = fma(-oma, di, v); }
    };
    __device__ __forceinline__ Ref operator[](int i) const { return Ref{p + i * 64, d[i], oma}; }
};

// ---------------------------------------------------------------------------------------------------------
// small fixed-size helpers (all indices compile-time after unrolling -> registers)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void matvec3(const double* M, const double* v, double* o) {
    o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
// out = M * skew(v)
__device__ __forceinline__ void mat_skew3(const double* M, const double* v, double* o) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[3 * a + 0] = M[3 * a + 1] * v[2] - M[3 * a + 2] * v[1];
        o[3 * a + 1] = -M[3 * a + 0] * v[2] + M[3 * a + 2] * v[0];
        o[3 * a + 2] = M[3 * a + 0] * v[1] - M[3 * a + 1] * v[0];
    }
}
// out = skew(v) * M
__device__ __forceinline__ void skew_mat3(const double* v, const double* M, double* o) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        o[0 + b] = -v[2] * M[3 + b] + v[1] * M[6 + b];
        o[3 + b] = v[2] * M[0 + b] - v[0] * M[6 + b];
        o[6 + b] = -v[1] * M[0 + b] + v[0] * M[3 + b];
    }
}
__device__ __forceinline__ void matmul3(const double* A, const double* B, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// 1/x to fp64 accuracy: hardware estimate + 2 Newton steps, 5 instructions instead of the ~30 of an IEEE division.
// x = 0 or non-finite gives a non-finite result, which the callers treat like the division's (pivot checks, cost checks).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void inv3(const double* M, double* o) {
    const double c00 = M[4] * M[8] - M[5] * M[7];
    const double c01 = M[5] * M[6] - M[3] * M[8];
    const double c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = fast_rcp(det);
    o[0] = c00 * id;
    o[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    o[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    o[3] = c01 * id;
    o[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    o[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    o[6] = c02 * id;
    o[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    o[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}
// Horizon utils.toRot (prb.py:97): xyzw quaternion -> matrix, no normalisation
__device__ __forceinline__ void quat_to_rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
template <int A>
__device__ __forceinline__ void quat_to_rot_d(const double* q, double* D) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    if (A == 0) { D[0] = 0; D[1] = 2 * y; D[2] = 2 * z; D[3] = 2 * y; D[4] = -4 * x; D[5] = -2 * w; D[6] = 2 * z; D[7] = 2 * w; D[8] = -4 * x; }
    if (A == 1) { D[0] = -4 * y; D[1] = 2 * x; D[2] = 2 * w; D[3] = 2 * x; D[4] = 0; D[5] = 2 * z; D[6] = -2 * w; D[7] = 2 * z; D[8] = -4 * y; }
    if (A == 2) { D[0] = -4 * z; D[1] = -2 * w; D[2] = 2 * x; D[3] = 2 * w; D[4] = -4 * z; D[5] = 2 * y; D[6] = 2 * x; D[7] = 2 * y; D[8] = 0; }
    if (A == 3) { D[0] = 0; D[1] = -2 * z; D[2] = 2 * y; D[3] = 2 * z; D[4] = 0; D[5] = -2 * x; D[6] = -2 * y; D[7] = 2 * x; D[8] = 0; }
}
// world inertia: prb.py:99 `w_R_b * (I/force_scaling) * w_R_b.T` is ELEMENT-WISE in CasADi (mode 0); mode 1 = R I R^T
__device__ __forceinline__ void world_inertia(const DevConsts& c, const double* R, double* M) {
    if (c.inertia_mode == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[3 * i + j] = R[3 * i + j] * c.Is[3 * i + j] * R[3 * j + i];
    } else {
        double T[9];  // Is R^T
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[3 * i + j] = c.Is[3 * i] * R[3 * j] + c.Is[3 * i + 1] * R[3 * j + 1] + c.Is[3 * i + 2] * R[3 * j + 2];
        matmul3(R, T, M);
    }
}
__device__ __forceinline__ void world_inertia_d(const DevConsts& c, const double* R, const double* dR, double* dM) {
    if (c.inertia_mode == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dM[3 * i + j] = c.Is[3 * i + j] * (dR[3 * i + j] * R[3 * j + i] + R[3 * i + j] * dR[3 * j + i]);
    } else {
        double T[9], U[9];  // T = Is R^T ; U = dR * T ; dM = U + U^T
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[3 * i + j] = c.Is[3 * i] * R[3 * j] + c.Is[3 * i + 1] * R[3 * j + 1] + c.Is[3 * i + 2] * R[3 * j + 2];
        matmul3(dR, T, U);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) dM[3 * i + j] = U[3 * i + j] + U[3 * j + i];
    }
}

// d2 I_w / dq_p dq_q from R, dR/dq_p, dR/dq_q and the (constant) d2R/dq_p dq_q
__device__ __forceinline__ void world_inertia_d2(const DevConsts& c, const double* R, const double* dRp, const double* dRq, const double* d2R,
                                                 double* d2M) {
    if (c.inertia_mode == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                d2M[3 * i + j] = c.Is[3 * i + j] * (d2R[3 * i + j] * R[3 * j + i] + dRp[3 * i + j] * dRq[3 * j + i] +
                                                    dRq[3 * i + j] * dRp[3 * j + i] + R[3 * i + j] * d2R[3 * j + i]);
    } else {   // d2R Is R^T + dRp Is dRq^T + dRq Is dRp^T + R Is d2R^T = X + X^T,  X = d2R Is R^T + dRp Is dRq^T
        double T[9], U[9], X[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[3 * i + j] = c.Is[3 * i] * R[3 * j] + c.Is[3 * i + 1] * R[3 * j + 1] + c.Is[3 * i + 2] * R[3 * j + 2];
        matmul3(d2R, T, X);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[3 * i + j] = c.Is[3 * i] * dRq[3 * j] + c.Is[3 * i + 1] * dRq[3 * j + 1] + c.Is[3 * i + 2] * dRq[3 * j + 2];
        matmul3(dRp, T, U);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) d2M[3 * i + j] = (X[3 * i + j] + U[3 * i + j]) + (X[3 * j + i] + U[3 * j + i]);
    }
}

// Q as the model code sees it: a symmetric (NZ x NZ) matrix addressed by (row, column).  The single-wavefront kernel stores it
// whole (QFull).  The 4-wavefront kernel (QSplit) keeps the state block inside its Vxx tile, the input rows [Qux | Quu] as a tile
// of their own and does not store the (state row, input column) block at all: accesses to it go to a dump word.
struct QFull {
    double* q; int ld;
    __device__ __forceinline__ double& at(int r, int c) const { return q[r * ld + c]; }
};
template <int NXS>
struct QSplit {
    double* xx; int ldx; double* u; int ldu; double* dump;
    __device__ __forceinline__ double& at(int r, int c) const {
        double* pu = u + (r - NXS) * ldu + c;
        double* px = c < NXS ? xx + r * ldx + c : dump;
        return *(r >= NXS ? pu : px);
    }
};

// ---------------------------------------------------------------------------------------------------------
// SRBD family.  CS=false, NC=2: srbd13 (metric model, contacts are parameters);  CS=true: contacts are states
// (reference problem, prb.py:32-68) -- NC=4: srbd37 (contact_model = 2, the launch file's), NC=8: srbd61 (contact_model = 4,
// the default in the code, prb.py:39-41).  Two legs (number_of_legs = 2), CM = NC / 2 contact points per foot.
// ---------------------------------------------------------------------------------------------------------
template <int NC_, bool CS_, bool BAR_ = false, bool SO2_ = false, int XR_ = 0>
struct SrbdModel {
    static constexpr int NC = NC_;
    static constexpr int NXR = XR_;        // user-declared linear residual rows ("_x" builds, sddp.h extra_*): 0 or kXrRows; their per-knot
                                           // references are NXR further parameter columns behind the model's own
    static_assert(XR_ == 0 || (XR_ == kXrRows && !BAR_ && !SO2_), "the _x builds are plain builds with kXrRows user rows");
    static constexpr bool CS = CS_;
    static constexpr bool BAR = BAR_;      // friction-cone exponential barrier on the contact forces (sddp.h), separate builds
    static constexpr bool SO2 = SO2_;      // full second-order builds (sddp_options.second_order = 2): the record also carries the
                                           // second derivatives of wdot, add_second_order adds the whole term
    static constexpr int NX = CS ? 13 + 6 * NC : 13;
    static constexpr int NU = CS ? 6 * NC : 3 * NC;
    static constexpr int NZ = NX + NU;
    static constexpr int CM = NC / 2;      // contact points per foot (rosparam contact_model, prb.py:39)
    // parameters in creation order (ddp.py:173-177): rdot_ref(3) w_ref(3) otg(1) (c_ref_i, cdot_switch_i) x NC, oref(4)
    static constexpr int NPB = CS ? 11 + 2 * NC : 19, NP = NPB + NXR;
    static_assert((CS && (NC == 4 || NC == 8)) || (!CS && NC == 2), "srbd37 / srbd61 / srbd13");
    // the bound barrier's masks and bound arrays (sddp_model_consts.lower / upper) cover 64 entries of z: models with more
    // (srbd61: 109) take the friction-cone barrier only -- sddp_create refuses bound_barrier_weight > 0 for them, as the oracle does
    static constexpr bool BOX = BAR_ && (CS_ ? 13 + 12 * NC_ : 13 + 3 * NC_) <= 64;
    // relative-velocity penalty pairs (prb.py:166-170): q-th pair = (first contact of the leg, its i-th other contact)
    static constexpr int NRV = CS ? 2 * (CM - 1) : 0, CM1 = CM > 1 ? CM - 1 : 1;
    __device__ __forceinline__ static constexpr int rv_a(int q) { return (q / CM1) * CM; }
    __device__ __forceinline__ static constexpr int rv_b(int q) { return (q / CM1) * CM + 1 + q % CM1; }
    // number of pairs contact i is part of; whether (i, j), i != j, is a pair
    __device__ __forceinline__ static constexpr int rv_count(int i) { return i % CM == 0 ? CM - 1 : 1; }
    __device__ __forceinline__ static constexpr bool rv_paired(int i, int j) { return i / CM == j / CM && (i % CM == 0 || j % CM == 0); }
    // state offsets (prb.py:32-59 creation order)
    static constexpr int XR = 0, XO = 3, XC = 7, XRD = CS ? 7 + 3 * NC : 7, XW = XRD + 3, XCD = XW + 3;
    // compact columns of A = d wdot / d z : r(0..2) o(3..6) w(7..9) [c(3NC)] f(3NC)
    static constexpr int AC = 10, AF = 10 + (CS ? 3 * NC : 0), NA = AF + 3 * NC;
    // derivative record of one knot
    static constexpr int REC_A = 0, REC_JO = 3 * NA, REC_JW = REC_JO + 16, REC_MI = REC_JW + 12, REC_G = REC_MI + 9,
                         REC_B = REC_G + NZ,                       // BAR: barrier Hessian per contact: hxx hyy hzz hxz hyz
                         REC_BB = REC_B + (BAR ? 5 * NC : 0),      // BAR: Gauss-Newton Hessian (diagonal) of the bound barrier, NZ words
                         REC_WD = REC_BB + (BAR ? NZ : 0),         // SO2 (full second-order builds): compact factors of the wdot
                         REC_M = REC_WD + 3,                       //   Hessian, contracted in the sweep (add_second_order):
                         REC_DM = REC_M + 9,                       //   wdot (3) | I_w (9) | dI_w/do_a (4 x 9) |
                         REC_COO = REC_DM + 36,                    //   w x (d2I_w/do_a do_b w) + d2I_w/do_a do_b wdot for the 10
                         REC_W = REC_COO + 30,                     //   pairs a >= b (10 x 3) | w (3)
                         NTRI = NA * (NA + 1) / 2,
                         NREC = SO2 ? REC_W + 3 : REC_WD,
                         // SO2: per-knot factors of the contraction, computed once per knot in LDS (so2_prepare): y (3) | dI_a y
                         // (4 x 3) | h_b = (dI_b w) x y + dI_b (y x w) (4 x 3) | P[a][b] = (I_w e_b x y)_a (9) | y . c_ab (10)
                         //   then two zero words (the slot of "no term")
                         T_Y = 0, T_G = 3, T_H = 15, T_P = 27, T_D = 36, T_ZERO = 46, NSO2T = SO2 ? 48 : 0,
                         NSO2L = SO2 ? 2 * NTRI : 0;       // ints: pair codes of the contraction (so2_pair_code), built once per instance

    __device__ __forceinline__ static int uf(int i) { return CS ? 6 * i + 3 : 3 * i; }  // prb.py:66-68 interleaved
    // parameter layouts: srbd37 = creation order; srbd13 = App. A.7
    __device__ __forceinline__ static double p_rdref(const double* p, int a) { return p[a]; }
    __device__ __forceinline__ static double p_wref(const double* p, int a) { return p[3 + a]; }
    __device__ __forceinline__ static double p_otg(const double* p) { return p[6]; }
    __device__ __forceinline__ static double p_oref(const double* p, int a) { return CS ? p[7 + 2 * NC + a] : p[7 + a]; }
    __device__ __forceinline__ static double p_sw(const double* p, int i) { return CS ? p[8 + 2 * i] : p[17 + i]; }
    __device__ __forceinline__ static double p_cref(const double* p, int i) { return p[7 + 2 * i]; }

    struct Core {
        double Mi[9], M[9], R[9], Mw[3], wdot[3], rddot[3];
    };

    // rddot, wdot (Horizon kin_dyn.fSRBD, prb.py:99; App. A.3)
    __device__ __forceinline__ static void core(const DevConsts& c, const double* r, const double* o, const double* w,
                                                const double (*cp)[3], const double (*f)[3], Core& k) {
        quat_to_rot(o, k.R);
        world_inertia(c, k.R, k.M);
        inv3(k.M, k.Mi);
        matvec3(k.M, w, k.Mw);
        double tau[3] = {0, 0, 0}, fs[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            double l[3] = {cp[i][0] - r[0], cp[i][1] - r[1], cp[i][2] - r[2]}, t[3];
            cross3(l, f[i], t);
#pragma unroll
            for (int a = 0; a < 3; ++a) { tau[a] += c.lever * t[a]; fs[a] += f[i][a]; }
        }
        double g[3];
        cross3(w, k.Mw, g);
#pragma unroll
        for (int a = 0; a < 3; ++a) tau[a] -= g[a];
        matvec3(k.Mi, tau, k.wdot);
        k.rddot[0] = fs[0] * c.inv_ms;
        k.rddot[1] = fs[1] * c.inv_ms;
        k.rddot[2] = fs[2] * c.inv_ms - kGravity;
    }

    template <class XV, class UV>
    __device__ __forceinline__ static void load_contacts(XV x, UV u, const double* p, double (*cp)[3], double (*f)[3]) {
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cp[i][a] = CS ? x[XC + 3 * i + a] : p[11 + 3 * i + a];
                f[i][a] = u[uf(i) + a];
            }
    }

    // cost of the state residuals (nodes 1..ns, prb.py:184-199)
    template <class XV>
    __device__ __forceinline__ static double state_cost(const DevConsts& c, XV x, const double* p) {
        double L = 0;
        const double ez = x[XR + 2] - c.com_z;
        L += c.w_rz * ez * ez;
        const double o[4] = {x[XO], x[XO + 1], x[XO + 2], x[XO + 3]};
        const double qx = p_oref(p, 0), qy = p_oref(p, 1), qz = p_oref(p, 2), qw = p_oref(p, 3);
        const double e0 = o[3] * qx + qw * o[0] + (o[1] * qz - o[2] * qy);
        const double e1 = o[3] * qy + qw * o[1] + (o[2] * qx - o[0] * qz);
        const double e2 = o[3] * qz + qw * o[2] + (o[0] * qy - o[1] * qx);
        const double e3 = o[3] * qw - (o[0] * qx + o[1] * qy + o[2] * qz) - 1.0;
        const double otg = p_otg(p);
        L += otg * otg * (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double ed = x[XRD + a] - p_rdref(p, a), ew = x[XW + a] - p_wref(p, a);
            L += c.w_rd * ed * ed + c.w_w * ew * ew;
        }
        if (CS) {
            const double r1y = -x[XC + 1] + x[XC + 7] - c.d1y, r1x = -x[XC + 0] + x[XC + 6] - c.d1x;
            const double r2y = -x[XC + 4] + x[XC + 10] - c.d2y, r2x = -x[XC + 3] + x[XC + 9] - c.d2x;
            L += c.w_rel * (r1y * r1y + r1x * r1x + r2y * r2y + r2x * r2x);
        }
        return L;
    }

    // friction-cone exponential barrier of one contact force (BAR builds): value w sum_j exp(s a_j.f) over the rows
    // a_j = (+-1, 0, -mu), (0, +-1, -mu), (0, 0, -1) of the linearised cone; gradient; Gauss-Newton Hessian of the residual form
    // r_j = sqrt(w) exp(s a_j.f / 2), i.e. (w s^2 / 2) sum_j e_j a_j a_j^T -> hxx hyy hzz hxz hyz (hxy = 0)
    __device__ __forceinline__ static double barrier(const DevConsts& c, const double* f, double* grad, double* h) {
        const double m = c.mu_lin, sz = -m * f[2];
        const double e1 = exp(c.bar_s * (f[0] + sz)), e2 = exp(c.bar_s * (-f[0] + sz));
        const double e3 = exp(c.bar_s * (f[1] + sz)), e4 = exp(c.bar_s * (-f[1] + sz));
        const double e5 = exp(-c.bar_s * f[2]);
        if (grad) {
            const double ws = c.bar_w * c.bar_s;
            grad[0] = ws * (e1 - e2);
            grad[1] = ws * (e3 - e4);
            grad[2] = -ws * (m * (e1 + e2 + e3 + e4) + e5);
        }
        if (h) {
            const double k2 = 0.5 * c.bar_w * c.bar_s * c.bar_s;
            h[0] = k2 * (e1 + e2);
            h[1] = k2 * (e3 + e4);
            h[2] = k2 * (m * m * (e1 + e2 + e3 + e4) + e5);
            h[3] = -k2 * m * (e1 - e2);
            h[4] = -k2 * m * (e3 - e4);
        }
        return c.bar_w * (e1 + e2 + e3 + e4 + e5);
    }
    // entry (a, b) of that 3x3 Hessian from its 5 stored words
    __device__ __forceinline__ static double barrier_h(const double* h, int a, int b) {
        if (a == b) return h[a];
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        return hi == 2 ? h[3 + lo] : 0.0;
    }

    // opt-in barrier on the bounds of z = [x u] (BAR builds, sddp.h; ddp.py:203-208): w sum_j [exp(s (z_j - ub_j)) + exp(s (lb_j - z_j))]
    // over the finite bounds.  Compile-time j: whether z_j is bounded is a bit test on a scalar, only bounded entries load and exp.
    // grad / hdiag (derivative phase): gradient added into grad[j], Gauss-Newton Hessian (w s^2 / 2) e of the residual form
    // r = sqrt(w) exp(s (z - ub) / 2) written to hdiag[j] (0 where unbounded)
    template <class XV, class UV>
    __device__ __forceinline__ static double bound_cost(const DevConsts& c, XV x, UV u, double* grad = nullptr, double* hdiag = nullptr) {
        double L = 0.0;
        bool on = false;
        if constexpr (BOX) {
            if (c.box_w > 0.0) {
                on = true;
                const double ws = c.box_w * c.box_s, k2 = 0.5 * ws * c.box_s;
#pragma unroll
                for (int j = 0; j < NZ; ++j) {
                    const double z = j < NX ? x[j < NX ? j : 0] : u[j < NX ? 0 : j - NX];
                    double eu = 0.0, el = 0.0;
                    if ((c.box_um >> j) & 1) eu = exp(c.box_s * (z - c.box[64 + j]));
                    if ((c.box_lm >> j) & 1) el = exp(c.box_s * (c.box[j] - z));
                    L += eu + el;
                    if (grad) grad[j] += ws * (eu - el);
                    if (hdiag) hdiag[j] = k2 * (eu + el);
                }
                L *= c.box_w;
            }
        }
        if (!on && hdiag) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) hdiag[j] = 0.0;
        }
        return L;
    }

    // cost of the input residuals + penalties given rddot/wdot (nodes 0..ns-1, prb.py:200-204, :166-181)
    template <class XV, class UV>
    __device__ __forceinline__ static double input_cost(const DevConsts& c, XV x, UV u, const double* p,
                                                        const double (*f)[3], const Core& k) {
        double L = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) L += c.gq * (k.rddot[a] * k.rddot[a] + k.wdot[a] * k.wdot[a]);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const double s1 = 1.0 - p_sw(p, i);
            const double wf = c.w_f + c.w_sw * s1 * s1;
            L += wf * (f[i][0] * f[i][0] + f[i][1] * f[i][1] + f[i][2] * f[i][2]);
            if (BAR) L += barrier(c, f[i], nullptr, nullptr);
        }
        if (CS) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const double cdd0 = u[6 * i], cdd1 = u[6 * i + 1], cdd2 = u[6 * i + 2];
                L += c.gq * (cdd0 * cdd0 + cdd1 * cdd1 + cdd2 * cdd2);
                const double ez = x[XC + 3 * i + 2] - p_cref(p, i);
                const double sw = p_sw(p, i);
                const double vx = sw * x[XCD + 3 * i], vy = sw * x[XCD + 3 * i + 1];
                L += c.w_pen * (ez * ez + vx * vx + vy * vy);
            }
#pragma unroll
            for (int q = 0; q < NRV; ++q) {
                const int a = rv_a(q), b = rv_b(q);
                const double ex = x[XCD + 3 * a] - x[XCD + 3 * b], ey = x[XCD + 3 * a + 1] - x[XCD + 3 * b + 1];
                L += c.w_rv * (ex * ex + ey * ey);
            }
        }
        return L;
    }

    // x+ = x + dt*xdot (explicit Euler, ddp.py:228-230) and L_k(x,u,p) (ddp.py:179-214) in one pass
    template <class XV, class UV, class XN>
    __device__ __forceinline__ static double step(const DevConsts& c, XV x, UV u, const double* p, int k, XN xn) {
        double cp[NC][3], f[NC][3];
        load_contacts(x, u, p, cp, f);
        Core q;
        const double r[3] = {x[XR], x[XR + 1], x[XR + 2]}, o[4] = {x[XO], x[XO + 1], x[XO + 2], x[XO + 3]};
        const double w[3] = {x[XW], x[XW + 1], x[XW + 2]};
        core(c, r, o, w, cp, f, q);
        double L = input_cost(c, x, u, p, f, q);
        if (BAR) L += bound_cost(c, x, u);
        if (k >= 1) L += state_cost(c, x, p);
        if (NXR) L += xr_eval<NX, NU>(c, x, u, true, p + NPB, k >= 1, true);
        // every component of x+ is formed from values read BEFORE the first store: x and xn may be the same array (in-place
        // rollout) or LDS columns the compiler cannot tell apart (then interleaved loads and stores would serialise)
        const double dt = c.dt;
        double xp[NX];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            xp[XR + a] = x[XR + a] + dt * x[XRD + a];
            xp[XRD + a] = x[XRD + a] + dt * q.rddot[a];
            xp[XW + a] = x[XW + a] + dt * q.wdot[a];
        }
        // odot = 1/2 [w;0] (x) o  (LOCAL_WORLD_ALIGNED, prb.py:107-108)
        double wxo[3];
        cross3(w, o, wxo);
#pragma unroll
        for (int a = 0; a < 3; ++a) xp[XO + a] = o[a] + dt * 0.5 * (o[3] * w[a] + wxo[a]);
        xp[XO + 3] = o[3] - dt * 0.5 * (w[0] * o[0] + w[1] * o[1] + w[2] * o[2]);
        if (CS) {
#pragma unroll
            for (int i = 0; i < 3 * NC; ++i) {
                xp[XC + i] = x[XC + i] + dt * x[XCD + i];
                xp[XCD + i] = x[XCD + i] + dt * u[6 * (i / 3) + (i % 3)];
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = xp[i];
        return L;
    }

    template <class XV>
    __device__ __forceinline__ static double term_cost(const DevConsts& c, XV x, const double* p) {
        double L = state_cost(c, x, p);  // ddp.py:216-226: residuals only, no constraints
        if (NXR) L += xr_eval<NX, NU>(c, x, x, false, p + NPB, true, false);
        return L;
    }

    // compact derivative record of knot k (k == N: terminal -> gradient only)
    __device__ __forceinline__ static void derivs(const DevConsts& c, const double* x, const double* u, const double* p,
                                                  int k, int N, double* rec) {
        double g[NZ];
#pragma unroll
        for (int i = 0; i < NZ; ++i) g[i] = 0.0;
        const double* r = x + XR; const double* o = x + XO; const double* w = x + XW;
        if (k >= 1) {  // state residual gradients
            g[XR + 2] += 2 * c.w_rz * (x[XR + 2] - c.com_z);
            const double qv[3] = {p_oref(p, 0), p_oref(p, 1), p_oref(p, 2)};
            const double qw = p_oref(p, 3);
            double oxq[3];
            cross3(o, qv, oxq);
            double ev[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) ev[a] = o[3] * qv[a] + qw * o[a] + oxq[a];
            const double ew = o[3] * qw - (o[0] * qv[0] + o[1] * qv[1] + o[2] * qv[2]) - 1.0;
            const double otg = p_otg(p);
            const double s2 = 2 * otg * otg;
            double qxe[3];
            cross3(qv, ev, qxe);
#pragma unroll
            for (int a = 0; a < 3; ++a) g[XO + a] += s2 * (qw * ev[a] + qxe[a] - qv[a] * ew);
            g[XO + 3] += s2 * (qv[0] * ev[0] + qv[1] * ev[1] + qv[2] * ev[2] + qw * ew);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                g[XRD + a] += 2 * c.w_rd * (x[XRD + a] - p_rdref(p, a));
                g[XW + a] += 2 * c.w_w * (x[XW + a] - p_wref(p, a));
            }
            if (CS) {
                const double r1y = -x[XC + 1] + x[XC + 7] - c.d1y, r1x = -x[XC + 0] + x[XC + 6] - c.d1x;
                const double r2y = -x[XC + 4] + x[XC + 10] - c.d2y, r2x = -x[XC + 3] + x[XC + 9] - c.d2x;
                const double s = 2 * c.w_rel;
                g[XC + 1] -= s * r1y; g[XC + 7] += s * r1y; g[XC + 0] -= s * r1x; g[XC + 6] += s * r1x;
                g[XC + 4] -= s * r2y; g[XC + 10] += s * r2y; g[XC + 3] -= s * r2x; g[XC + 9] += s * r2x;
            }
        }
        if (k < N) {
            double cp[NC][3], f[NC][3];
            load_contacts(x, u, p, cp, f);
            Core q;
            core(c, r, o, w, cp, f, q);
            double A[3][NA];
            // d wdot / d r = Mi * (s * skew(sum f))
            {
                double sf[3] = {0, 0, 0}, T[9];
#pragma unroll
                for (int i = 0; i < NC; ++i)
#pragma unroll
                    for (int a = 0; a < 3; ++a) sf[a] += c.lever * f[i][a];
                mat_skew3(q.Mi, sf, T);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) A[a][b] = T[3 * a + b];
            }
            // d wdot / d w = Mi * (skew(M w) - skew(w) M)
            {
                double T[9], U[9], V[9];
                skew_mat3(w, q.M, T);
                const double mw[3] = {q.Mw[0], q.Mw[1], q.Mw[2]};
                U[0] = 0 - T[0]; U[1] = -mw[2] - T[1]; U[2] = mw[1] - T[2];
                U[3] = mw[2] - T[3]; U[4] = 0 - T[4]; U[5] = -mw[0] - T[5];
                U[6] = -mw[1] - T[6]; U[7] = mw[0] - T[7]; U[8] = 0 - T[8];
                matmul3(q.Mi, U, V);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) A[a][7 + b] = V[3 * a + b];
            }
            // d wdot / d o_a = -Mi (dM_a wdot + w x (dM_a w))
            {
                double col[4][3];
                dwdot_do<0>(c, o, w, q, col[0]);
                dwdot_do<1>(c, o, w, q, col[1]);
                dwdot_do<2>(c, o, w, q, col[2]);
                dwdot_do<3>(c, o, w, q, col[3]);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) A[a][3 + b] = col[b][a];
            }
            // d wdot / d c_i = Mi * (-s skew(f_i)) ; d wdot / d f_i = Mi * (s skew(c_i - r))
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                double T[9];
                const double l[3] = {c.lever * (cp[i][0] - r[0]), c.lever * (cp[i][1] - r[1]), c.lever * (cp[i][2] - r[2])};
                mat_skew3(q.Mi, l, T);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) A[a][AF + 3 * i + b] = T[3 * a + b];
                if (CS) {
                    const double mf[3] = {-c.lever * f[i][0], -c.lever * f[i][1], -c.lever * f[i][2]};
                    mat_skew3(q.Mi, mf, T);
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) A[a][AC + 3 * i + b] = T[3 * a + b];
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int j = 0; j < NA; ++j) rec[REC_A + a * NA + j] = A[a][j];
#pragma unroll
            for (int i = 0; i < 9; ++i) rec[REC_MI + i] = q.Mi[i];      // I_w^-1, for the second-order torque term
            // quaternion-rate blocks: d odot / d o (4x4), d odot / d w (4x3)
            {
                double Jo[16], Jw[12];
                Jo[0] = 0; Jo[1] = -0.5 * w[2]; Jo[2] = 0.5 * w[1]; Jo[3] = 0.5 * w[0];
                Jo[4] = 0.5 * w[2]; Jo[5] = 0; Jo[6] = -0.5 * w[0]; Jo[7] = 0.5 * w[1];
                Jo[8] = -0.5 * w[1]; Jo[9] = 0.5 * w[0]; Jo[10] = 0; Jo[11] = 0.5 * w[2];
                Jo[12] = -0.5 * w[0]; Jo[13] = -0.5 * w[1]; Jo[14] = -0.5 * w[2]; Jo[15] = 0;
                // 1/2 (o_w I - skew(o_v)) ; last row -1/2 o_v
                Jw[0] = 0.5 * o[3]; Jw[1] = 0.5 * o[2]; Jw[2] = -0.5 * o[1];
                Jw[3] = -0.5 * o[2]; Jw[4] = 0.5 * o[3]; Jw[5] = 0.5 * o[0];
                Jw[6] = 0.5 * o[1]; Jw[7] = -0.5 * o[0]; Jw[8] = 0.5 * o[3];
                Jw[9] = -0.5 * o[0]; Jw[10] = -0.5 * o[1]; Jw[11] = -0.5 * o[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) rec[REC_JO + i] = Jo[i];
#pragma unroll
                for (int i = 0; i < 12; ++i) rec[REC_JW + i] = Jw[i];
            }
            // gradient of the input residuals: min_qddot rows [rddot; wdot; cddot], min_f, f_active, penalties
            const double s = 2 * c.gq;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const double v = s * (A[0][j] * q.wdot[0] + A[1][j] * q.wdot[1] + A[2][j] * q.wdot[2]);
                g[zcol(j)] += v;
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const double s1 = 1.0 - p_sw(p, i);
                const double wf = 2 * (c.w_f + c.w_sw * s1 * s1);
#pragma unroll
                for (int a = 0; a < 3; ++a) g[NX + uf(i) + a] += s * q.rddot[a] * c.inv_ms + wf * f[i][a];
                if (BAR) {
                    double bg[3], bh[5];
                    barrier(c, f[i], bg, bh);
#pragma unroll
                    for (int a = 0; a < 3; ++a) g[NX + uf(i) + a] += bg[a];
#pragma unroll
                    for (int t = 0; t < 5; ++t) rec[REC_B + 5 * i + t] = bh[t];
                }
            }
            if (CS) {
                const double sp = 2 * c.w_pen;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) g[NX + 6 * i + a] += s * u[6 * i + a];
                    g[XC + 3 * i + 2] += sp * (x[XC + 3 * i + 2] - p_cref(p, i));
                    const double sw = p_sw(p, i);
                    g[XCD + 3 * i] += sp * sw * sw * x[XCD + 3 * i];
                    g[XCD + 3 * i + 1] += sp * sw * sw * x[XCD + 3 * i + 1];
                }
#pragma unroll
                for (int q = 0; q < NRV; ++q) {
                    const int a = rv_a(q), b = rv_b(q);
                    const double ex = x[XCD + 3 * a] - x[XCD + 3 * b], ey = x[XCD + 3 * a + 1] - x[XCD + 3 * b + 1];
                    const double sr = 2 * c.w_rv;
                    g[XCD + 3 * a] += sr * ex; g[XCD + 3 * b] -= sr * ex;
                    g[XCD + 3 * a + 1] += sr * ey; g[XCD + 3 * b + 1] -= sr * ey;
                }
            }
            if (BAR) (void)bound_cost(c, x, u, g, rec + REC_BB);   // bound barrier (off: zeros): gradient into g, GN Hessian diagonal into the record
            if (NXR) (void)xr_eval<NX, NU>(c, x, u, true, p + NPB, k >= 1, true, g);   // user rows: their gradient
#pragma unroll
            for (int i = 0; i < NZ; ++i) rec[REC_G + i] = g[i];
            return;                                     // (SO2 builds: the second-order factors are a pass of their own, so2_knot)
        }
        if (NXR) (void)xr_eval<NX, NU>(c, x, x, false, p + NPB, true, false, g);       // terminal node: the state rows
#pragma unroll
        for (int i = 0; i < NZ; ++i) rec[REC_G + i] = g[i];
    }

    // Full second-order builds: what the sweep needs to contract the second derivatives of wdot = I_w(o)^-1 n(z),
    // n = sum s (c_i - r) x f_i - w x I_w(o) w, with a multiplier it only knows then (add_second_order).  Differentiating
    // I_w wdot = n twice (oracle/models.py srbd_wdot_hess):
    //   d_a d_b wdot = I_w^-1 (d_a d_b n - d_a d_b I_w wdot - d_a I_w d_b wdot - d_b I_w d_a wdot),
    // so  lam . d_a d_b wdot = y . V_ab  with  y = I_w^-1 lam  and V_ab built from I_w, dI_w/do_a, w, the first derivatives A
    // (already in the record) and, for the quaternion pairs, c_ab = w x (d2I_w/do_a do_b w) + d2I_w/do_a do_b wdot: 81 words
    // instead of the 3 NA (NA + 1) / 2 tensor entries of the first version (408 for srbd13, 1785 for srbd37).  Straight-line
    // code on compile-time indices: everything stays in registers (the tensor version ran loops over private arrays: 4 KB of
    // scratch per lane).
    template <int PA, int PB>
    __device__ __forceinline__ static void so2_coo_pair(const DevConsts& c, const double* o, const double* w, const Core& q, double* out) {
        const double eb[4] = {PB == 0 ? 1.0 : 0.0, PB == 1 ? 1.0 : 0.0, PB == 2 ? 1.0 : 0.0, PB == 3 ? 1.0 : 0.0};
        double dRp[9], dRq[9], d2R[9], d2M[9], t[3], cr[3], t2[3];
        quat_to_rot_d<PA>(o, dRp);                     // recomputed per pair (9 trivial operations) rather than kept live for all ten
        quat_to_rot_d<PB>(o, dRq);
        quat_to_rot_d<PA>(eb, d2R);                    // dR/dq_a is linear in q: its q_b derivative is dR/dq_a at e_b
        world_inertia_d2(c, q.R, dRp, dRq, d2R, d2M);
        matvec3(d2M, w, t);
        cross3(w, t, cr);
        matvec3(d2M, q.wdot, t2);
#pragma unroll
        for (int m = 0; m < 3; ++m) out[m] = cr[m] + t2[m];
    }
    // the second-order factors of one stage knot, as a pass of its own after derivs(): the accelerations are recomputed (a few
    // hundred flops) so that this code does not share its registers with the first-derivative code (two-per-SIMD build: scratch)
    __device__ __forceinline__ static void so2_knot(const DevConsts& c, const double* x, const double* u, const double* p, double* rec) {
        if (!SO2) return;
        double cp[NC][3], f[NC][3];
        load_contacts(x, u, p, cp, f);
        Core q;
        core(c, x + XR, x + XO, x + XW, cp, f, q);
        so2_record(c, x + XO, x + XW, q, rec);
    }
    __device__ __forceinline__ static void so2_record(const DevConsts& c, const double* o, const double* w, const Core& q, double* rec) {
        double dR[9], dM[9];
#pragma unroll
        for (int m = 0; m < 3; ++m) { rec[REC_WD + m] = q.wdot[m]; rec[REC_W + m] = w[m]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) rec[REC_M + i] = q.M[i];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a == 0) quat_to_rot_d<0>(o, dR);
            if (a == 1) quat_to_rot_d<1>(o, dR);
            if (a == 2) quat_to_rot_d<2>(o, dR);
            if (a == 3) quat_to_rot_d<3>(o, dR);
            world_inertia_d(c, q.R, dR, dM);
#pragma unroll
            for (int i = 0; i < 9; ++i) rec[REC_DM + 9 * a + i] = dM[i];
        }
        double v[3];
#define SDDP_COO(PA, PB) so2_coo_pair<PA, PB>(c, o, w, q, v); rec[REC_COO + 3 * (PA * (PA + 1) / 2 + PB)] = v[0]; \
                         rec[REC_COO + 3 * (PA * (PA + 1) / 2 + PB) + 1] = v[1]; rec[REC_COO + 3 * (PA * (PA + 1) / 2 + PB) + 2] = v[2];
        SDDP_COO(0, 0) SDDP_COO(1, 0) SDDP_COO(1, 1) SDDP_COO(2, 0) SDDP_COO(2, 1) SDDP_COO(2, 2)
        SDDP_COO(3, 0) SDDP_COO(3, 1) SDDP_COO(3, 2) SDDP_COO(3, 3)
#undef SDDP_COO
    }

    template <int QA>
    __device__ __forceinline__ static void dwdot_do(const DevConsts& c, const double* o, const double* w, const Core& q,
                                                    double* out) {
        double dR[9], dM[9], a[3], b[3], cr[3];
        quat_to_rot_d<QA>(o, dR);
        world_inertia_d(c, q.R, dR, dM);
        matvec3(dM, q.wdot, a);
        matvec3(dM, w, b);
        cross3(w, b, cr);
        const double t[3] = {-(a[0] + cr[0]), -(a[1] + cr[1]), -(a[2] + cr[2])};
        matvec3(q.Mi, t, out);
    }

    // z index of compact A column j (compile-time after unrolling)
    __device__ __forceinline__ static constexpr int zcol(int j) {
        return j < 7 ? j : (j < 10 ? XW + (j - 7) : (CS && j < AF ? XC + (j - AC) : NX + (CS ? 6 * ((j - AF) / 3) + 3 + (j - AF) % 3 : (j - AF))));
    }
    // compact A column of z index j, or -1
    __device__ __forceinline__ static int acol(int j) {
        if (j < 7) return j;
        if (CS && j < XRD) return AC + (j - XC);
        if (j < XW) return -1;
        if (j < XW + 3) return 7 + (j - XW);
        if (j < NX) return -1;
        const int uj = j - NX;
        if (CS) {
            const int i = uj / 6, a = uj % 6;
            return a >= 3 ? AF + 3 * i + (a - 3) : -1;
        }
        return AF + uj;
    }

    // variable class / contact / axis of z index j
    enum { V_R = 0, V_O, V_C, V_RD, V_W, V_CD, V_CDD, V_F };
    __device__ __forceinline__ static void decode(int j, int& cls, int& ci, int& ax) {
        ci = 0;
        if (j < 3) { cls = V_R; ax = j; return; }
        if (j < 7) { cls = V_O; ax = j - 3; return; }
        if (CS && j < XRD) { cls = V_C; ci = (j - XC) / 3; ax = (j - XC) % 3; return; }
        if (j < XW) { cls = V_RD; ax = j - XRD; return; }
        if (j < XW + 3) { cls = V_W; ax = j - XW; return; }
        if (j < NX) { cls = V_CD; ci = (j - XCD) / 3; ax = (j - XCD) % 3; return; }
        const int uj = j - NX;
        if (CS) {
            ci = uj / 6;
            const int a = uj % 6;
            if (a < 3) { cls = V_CDD; ax = a; } else { cls = V_F; ax = a - 3; }
            return;
        }
        cls = V_F; ci = uj / 3; ax = uj % 3;
    }

    // element (i,j) of [fx fu] = [I 0] + dt * d xdot / d z   (rec: this knot's record)
    __device__ __forceinline__ static double F_entry(const DevConsts& c, const double* rec, int i, int j) {
        int ci, cli, ai, cj, clj, aj;
        decode(i, cli, ci, ai);
        decode(j, clj, cj, aj);
        double s = 0.0;
        switch (cli) {
            case V_R: s = (clj == V_RD && aj == ai) ? 1.0 : 0.0; break;
            case V_O:
                if (clj == V_O) s = rec[REC_JO + 4 * ai + aj];
                else if (clj == V_W) s = rec[REC_JW + 3 * ai + aj];
                break;
            case V_C: s = (clj == V_CD && cj == ci && aj == ai) ? 1.0 : 0.0; break;
            case V_RD: s = (clj == V_F && aj == ai) ? c.inv_ms : 0.0; break;
            case V_W: { const int a = acol(j); s = a >= 0 ? rec[REC_A + ai * NA + a] : 0.0; } break;
            case V_CD: s = (clj == V_CDD && cj == ci && aj == ai) ? 1.0 : 0.0; break;
            default: break;
        }
        return (i == j ? 1.0 : 0.0) + c.dt * s;
    }

    // Sparsity of [fx fu] by COLUMN z (the 4-wavefront kernel's products run over it): besides the identity, column z has
    // entries only in the ND = 7 "dense" next-state rows -- o (4) and w (3), where every per-knot variable entry lives -- and in
    // at most ONE other row (F_entry): rdot_a -> r_a (dt), cdot_i,a -> c_i,a (dt), cddot_i,a -> cdot_i,a (dt), f_i,a -> rdot_a
    // (dt / m).  So for any X indexed by the next state:  (F^T X)[z] = idc(z) X[z] + beta(z) X[n(z)] + sum_d F[D_d][z] X[D_d].
    static constexpr int ND = 7;
    __device__ __forceinline__ static constexpr int dense_row(int d) { return d < 4 ? XO + d : XW + (d - 4); }
    // compact column of next-state row `row` (dense rows: 0..6) or of extra row `row - NX` (ND + m); -1 for any other row
    __device__ __forceinline__ static constexpr int dense_col(int row) {
        return row >= NX ? ND + (row - NX) : (row >= XO && row < XO + 4 ? row - XO : (row >= XW && row < XW + 3 ? 4 + (row - XW) : -1));
    }
    // column z: idc = 1 when the identity entry is not part of a dense row, (n, beta) = the one other row and its entry (0, 0: none)
    __device__ static void nbr(const DevConsts& c, int z, int& n, double& beta, double& idc) {
        int cls, ci, ax;
        decode(z, cls, ci, ax);
        idc = (z < NX && cls != V_O && cls != V_W) ? 1.0 : 0.0;
        n = 0; beta = 0.0;
        switch (cls) {
            case V_RD: n = XR + ax; beta = c.dt; break;
            case V_CD: n = XC + 3 * ci + ax; beta = c.dt; break;
            case V_CDD: n = XCD + 3 * ci + ax; beta = c.dt; break;
            case V_F: n = XRD + ax; beta = c.dt * c.inv_ms; break;
            default: break;
        }
    }

    // element (i,j) of the Gauss-Newton Hessian of L_k (k<N) or L_N (k==N)
    __device__ __forceinline__ static double H_entry(const DevConsts& c, const double* rec, const double* p, int k, int N,
                                                     int i, int j) {
        int ci, cli, ai, cj, clj, aj;
        decode(i, cli, ci, ai);
        decode(j, clj, cj, aj);
        const bool state = k >= 1, stage = k < N;
        double v = 0.0;
        if (cli == clj && ai == aj) {
            switch (cli) {
                case V_R: if (state && ai == 2) v = 2 * c.w_rz; break;
                case V_O: if (state) {
                    const double otg = p_otg(p);
                    const double n2 = p_oref(p, 0) * p_oref(p, 0) + p_oref(p, 1) * p_oref(p, 1) + p_oref(p, 2) * p_oref(p, 2) + p_oref(p, 3) * p_oref(p, 3);
                    v = 2 * otg * otg * n2;
                } break;
                case V_RD: if (state) v = 2 * c.w_rd; break;
                case V_W: if (state) v = 2 * c.w_w; break;
                case V_C:
                    if (state && ai < 2 && ci < 4 && cj < 4) {   // rel_pos names contact points 0, 2 and 1, 3 (prb.py:192-199), whatever nc is
                        if (ci == cj) v += 2 * c.w_rel; else if (ci + 2 == cj || cj + 2 == ci) v -= 2 * c.w_rel;
                    }
                    if (stage && ci == cj && ai == 2) v += 2 * c.w_pen;
                    break;
                case V_CD:
                    if (stage && ai < 2) {
                        if (ci == cj) { const double sw = p_sw(p, ci); v = 2 * c.w_rv * double(rv_count(ci)) + 2 * c.w_pen * sw * sw; }
                        else if (rv_paired(ci, cj)) v = -2 * c.w_rv;
                    }
                    break;
                case V_CDD: if (stage && ci == cj) v = 2 * c.gq; break;
                case V_F:
                    if (stage) {
                        v = 2 * c.gq * c.inv_ms * c.inv_ms;
                        if (ci == cj) { const double s1 = 1.0 - p_sw(p, ci); v += 2 * (c.w_f + c.w_sw * s1 * s1); }
                    }
                    break;
                default: break;
            }
        }
        if (BAR && stage && cli == V_F && clj == V_F && ci == cj) v += barrier_h(rec + REC_B + 5 * ci, ai, aj);
        if (BAR && stage && i == j) v += rec[REC_BB + i];
        if (stage) {
            const int a = acol(i), b = acol(j);
            if (a >= 0 && b >= 0)
                v += 2 * c.gq * (rec[REC_A + a] * rec[REC_A + b] + rec[REC_A + NA + a] * rec[REC_A + NA + b] +
                                 rec[REC_A + 2 * NA + a] * rec[REC_A + 2 * NA + b]);
        }
        if (NXR) {   // user rows: 2 w_j a_j a_j^T with the weight of the node
            for (int r = 0; r < c.xr_n; ++r) {
                const double w = (state ? c.xr[kXrWS + r] : 0.0) + (stage ? c.xr[kXrWG + r] : 0.0);
                v += 2 * w * c.xr[r * kXrStride + i] * c.xr[r * kXrStride + j];
            }
        }
        return v;
    }
    // -------------------------------------------------------------------------------------------------------------
    // Branch-free tile expansion used by the Riccati sweep (DESIGN.md "Kernel design").
    // The Gauss-Newton Hessian is written  H = diag(D) + Je^T Lambda Je : single-variable residuals go to the diagonal,
    // every multi-variable residual row (wdot, rddot, rel_pos, relative-velocity penalty) is an EXTRA ROW of the
    // augmented Jacobian F~ = [F ; Je] with its weight on the diagonal of V~ = blockdiag(Vxx+, Lambda), so that
    // Q = diag(D) + F~^T V~ F~ needs no special cases.  Only the wdot rows (A) and the quaternion blocks vary per knot.
    // -------------------------------------------------------------------------------------------------------------
    static constexpr int NEB = 6 + (CS ? 4 + 2 * NRV : 0);   // wdot(3) rddot(3) [rel_pos(4) rel_vel(2 per pair)]
    static constexpr int NE = NEB + NXR;                     // ... then the user rows (definition rows NEB .. NEB + NXR - 1)
    // Extra rows m >= NEV are constant (E_const) and their weights do not depend on the node: their contribution
    // sum_m lambda_m e_m e_m^T to Q is a constant matrix that the one-wave kernel adds instead of carrying the rows through the
    // tile products (srbd13: the three rddot rows -> product depth 20 -> 16).  NEV = NE: every row goes through the product.
    // Row order: first the NEV rows that must go through the tile products -- the variable wdot rows and, for the
    // reference model, the rel_pos rows whose weight depends on the node (state nodes only) -- then the constant rows with
    // node-independent weights (rddot, rel_vel).  erow() maps this order to the order of the definitions below.
    // The user rows go through the tile products too (their weight depends on the node: state rows / stage rows), right behind the
    // model's own product rows: sweep rows NEVB .. NEVB + NXR - 1.
    static constexpr int NEVB = CS ? 7 : 3, NEV = NEVB + NXR;
    static constexpr bool CONST_ROWS_STATE_WEIGHTED = false;
    __device__ __forceinline__ static int erow(int m) {
        if (NXR) {
            if (m >= NEVB && m < NEV) return NEB + (m - NEVB);
            if (m >= NEV) m -= NXR;
        }
        if (!CS) return m;                       // wdot(0-2) rddot(3-5)
        if (m < 3) return m;                     // wdot
        if (m < 7) return m + 3;                 // rel_pos   (definition rows 6..9)
        if (m < 10) return m - 4;                // rddot     (definition rows 3..5)
        return m;                                // rel_vel   (definition rows 10..13)
    }
    static constexpr int NVAR = 28 + 3 * NA;           // per-knot variable entries: quaternion blocks + A

    // constant entry of extra row m w.r.t. z_j (one-time table fill; variable wdot rows are 0 here)
    __device__ static double E_const(const DevConsts& c, int mrow, int j) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? c.xr[(m - NEB) * kXrStride + j] : 0.0;   // user rows
        int cls, ci, ax;
        decode(j, cls, ci, ax);
        if (m < 3) return 0.0;
        if (m < 6) return (cls == V_F && ax == m - 3) ? c.inv_ms : 0.0;                    // rddot rows   prb.py:200
        if (!CS) return 0.0;
        if (m < 10) {                                                                       // rel_pos rows prb.py:192-199
            const int pair = (m - 6) / 2, comp = ((m - 6) % 2 == 0) ? 1 : 0;                // y first, then x
            if (cls != V_C || ax != comp) return 0.0;
            return ci == pair ? -1.0 : (ci == pair + 2 ? 1.0 : 0.0);
        }
        const int pair = (m - 10) / 2, comp = (m - 10) % 2;                                 // relative_vel rows prb.py:166-170
        if (cls != V_CD || ax != comp) return 0.0;
        return ci == rv_a(pair) ? 1.0 : (ci == rv_b(pair) ? -1.0 : 0.0);
    }
    __device__ static double lam_state(const DevConsts& c, int mrow) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? 2 * c.xr[kXrWS + m - NEB] : 0.0;
        return (CS && m >= 6 && m < 10) ? 2 * c.w_rel : 0.0;
    }
    __device__ static double lam_stage(const DevConsts& c, int mrow) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? 2 * c.xr[kXrWG + m - NEB] : 0.0;
        if (m < 6) return 2 * c.gq;
        return (CS && m >= 10) ? 2 * c.w_rv : 0.0;
    }
    // diagonal of the Hessian: constant parts and the kind of parameter-dependent part (0 none, 1 orientation gain,
    // 2 force switch, 3 contact-velocity switch)
    __device__ static double dg_state(const DevConsts& c, int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        if (cls == V_R) return ax == 2 ? 2 * c.w_rz : 0.0;
        if (cls == V_RD) return 2 * c.w_rd;
        if (cls == V_W) return 2 * c.w_w;
        return 0.0;
    }
    __device__ static double dg_stage(const DevConsts& c, int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        if (cls == V_C) return ax == 2 ? 2 * c.w_pen : 0.0;
        if (cls == V_CDD) return 2 * c.gq;
        if (cls == V_F) return 2 * c.w_f;
        return 0.0;
    }
    __device__ static int dkind(int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        if (cls == V_O) return 1;
        if (cls == V_F) return 2;
        if (cls == V_CD && ax < 2) return 3;
        return 0;
    }
    __device__ static int dci(int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        return ci;
    }
    // parameter-dependent diagonal term of one knot (branch-free: all candidates evaluated, one selected)
    __device__ __forceinline__ static double dparam(const DevConsts& c, const double* p, int kind, int ci, double state, double stage) {
        // every parameter read requested unconditionally and together (pinned): left alone the compiler sinks each read into the
        // select that consumes it and the selects become branches with an exposed LDS round trip each
        double otg = p_otg(p), o0 = p_oref(p, 0), o1 = p_oref(p, 1), o2 = p_oref(p, 2), o3 = p_oref(p, 3);
        double sw = p_sw(p, ci);
        asm volatile("" : "+v"(otg), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "+v"(sw));
        // (state / stage through the same pin: `stage * 2 * c.w_sw` is otherwise hoisted out of the callers' loops as a VALU result, spilled
        // where registers are short, and every call then waits for a scratch reload -- two multiplications cost less)
        asm volatile("" : "+v"(state), "+v"(stage));
        const double n2 = o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3;
        const double v1 = state * 2 * otg * otg * n2;
        const double v2 = stage * 2 * c.w_sw * (1.0 - sw) * (1.0 - sw);
        const double v3 = stage * 2 * c.w_pen * sw * sw;
        return kind == 1 ? v1 : (kind == 2 ? v2 : (kind == 3 ? v3 : 0.0));
    }
    // per-knot variable entries of F~^T (FT[j*NIP + l] = F~[l][j]); branch-free index arithmetic, one entry per lane-step
    __device__ __forceinline__ static void expand_var(const DevConsts& c, const double* rec, double* FT, int NIP, int lane) {
        expand_var(c, rec, FT, NIP, lane, 64);
    }
    // WT != nullptr: also writes lam[m] * entry into the extra-row part of (V~ F~)^T (one-wave kernel).
    // COMPACT (4-wave kernel): FT is the compact tile [z][dense rows | extra rows] (dense_col), NIP its row stride
    template <bool COMPACT = false>
    __device__ __forceinline__ static void expand_var(const DevConsts& c, const double* rec, double* FT, int NIP, int tid, int nthreads,
                                                      double* WT = nullptr, const double* lam = nullptr) {
        // two entries per thread and trip, both record reads (and weights) requested before the first store: the single-wave
        // kernel (64 threads, NVAR > 64) needs two entries per lane and would otherwise sit out two LDS round trips in a row
        auto decode_entry = [&](int e, int& row, int& col, int& row2, int& src) {
            row2 = -1;
            if (e < 28) {
                const int a = e / 7, t = e % 7;
                row = XO + a;
                const bool isq = t < 4;
                col = isq ? XO + t : XW + (t - 4);
                src = isq ? REC_JO + 4 * a + t : REC_JW + 3 * a + (t - 4);
            } else {
                const int m = (e - 28) / NA, j = (e - 28) % NA;
                row = XW + m;
                row2 = NX + m;                                   // the same A entry is also extra row m (wdot residual)
                col = zcol(j);
                src = REC_A + m * NA + j;
            }
        };
        for (int e0 = tid; e0 < NVAR; e0 += 2 * nthreads) {
            const int e1 = e0 + nthreads;
            const bool two = e1 < NVAR;
            int row[2], col[2], row2[2], src[2];
            decode_entry(e0, row[0], col[0], row2[0], src[0]);
            decode_entry(two ? e1 : e0, row[1], col[1], row2[1], src[1]);
            double raw[2], lm[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                raw[q] = rec[src[q]];
                lm[q] = (WT && row2[q] >= 0) ? lam[row2[q] - NX] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) { asm volatile("" : "+v"(raw[q])); asm volatile("" : "+v"(lm[q])); }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && !two) break;
                const int r1 = COMPACT ? dense_col(row[q]) : row[q], r2 = COMPACT ? dense_col(row2[q] >= 0 ? row2[q] : NX) : row2[q];
                FT[col[q] * NIP + r1] = (row[q] == col[q] ? 1.0 : 0.0) + c.dt * raw[q];
                if (row2[q] >= 0) {
                    FT[col[q] * NIP + r2] = raw[q];
                    if (WT) WT[col[q] * NIP + r2] = lm[q] * raw[q];
                }
            }
        }
    }

    // exact second-order dynamics term of the DDP sweep: v'.f_ux restricted to the bilinear torque (c - r) x f
    //   Qux[f_i,a][r_b] += -theta * s * skew(y)[a][b] ,  Qux[f_i,a][c_i,b] += +theta * s * skew(y)[a][b] ,  y = I_w^-1 (dt v'_w)
    // (constant tensor; Q is kept symmetric: both triangles are updated)
    static constexpr int NSO = 9 * NC * (CS ? 2 : 1);
    // SO2 builds: the 46 per-knot factors of the second-order contraction (layout T_*), one per lane, into `tmp` (LDS); the
    // caller synchronises before add_second_order reads them.  lam = dt v'_w + 2 gq wdot, y = I_w^-1 lam.
    __device__ __forceinline__ static void so2_prepare(const DevConsts& c, const double* rec, const double* vp, double* tmp, int lane,
                                                       int nlanes) {
        if (!SO2) return;
        for (int t = lane; t < NSO2T; t += nlanes) {
            if (t >= T_ZERO) { tmp[t] = 0.0; continue; }
            const double l0 = c.dt * vp[XW] + 2 * c.gq * rec[REC_WD], l1 = c.dt * vp[XW + 1] + 2 * c.gq * rec[REC_WD + 1],
                         l2 = c.dt * vp[XW + 2] + 2 * c.gq * rec[REC_WD + 2];
            const double* Mi = rec + REC_MI;
            const double y0 = Mi[0] * l0 + Mi[1] * l1 + Mi[2] * l2, y1 = Mi[3] * l0 + Mi[4] * l1 + Mi[5] * l2,
                         y2 = Mi[6] * l0 + Mi[7] * l1 + Mi[8] * l2;
            auto sel = [](int i, double a0, double a1, double a2) { return i == 0 ? a0 : (i == 1 ? a1 : a2); };
            double v;
            if (t < T_G) {
                v = sel(t, y0, y1, y2);
            } else if (t < T_H) {                      // (dI_a y)_m
                const double* D = rec + REC_DM + 9 * ((t - T_G) / 3) + 3 * ((t - T_G) % 3);
                v = D[0] * y0 + D[1] * y1 + D[2] * y2;
            } else if (t < T_P) {                      // h_b[m] = ((dI_b w) x y + dI_b (y x w))_m
                const double* D = rec + REC_DM + 9 * ((t - T_H) / 3);
                const int m = (t - T_H) % 3;
                const double w0 = rec[REC_W], w1 = rec[REC_W + 1], w2 = rec[REC_W + 2];
                const double t0 = D[0] * w0 + D[1] * w1 + D[2] * w2, t1 = D[3] * w0 + D[4] * w1 + D[5] * w2,
                             t2 = D[6] * w0 + D[7] * w1 + D[8] * w2;
                const double z0 = y1 * w2 - y2 * w1, z1 = y2 * w0 - y0 * w2, z2 = y0 * w1 - y1 * w0;
                const double cr = sel(m, t1 * y2 - t2 * y1, t2 * y0 - t0 * y2, t0 * y1 - t1 * y0);
                v = cr + D[3 * m] * z0 + D[3 * m + 1] * z1 + D[3 * m + 2] * z2;
            } else if (t < T_D) {                      // P[a][b] = (I_w e_b x y)_a = m_{a+1} y_{a+2} - m_{a+2} y_{a+1}, m = column b
                const int a = (t - T_P) / 3, b = (t - T_P) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                v = rec[REC_M + 3 * a1 + b] * sel(a2, y0, y1, y2) - rec[REC_M + 3 * a2 + b] * sel(a1, y0, y1, y2);
            } else {                                   // y . c_pair
                const double* cc = rec + REC_COO + 3 * (t - T_D);
                v = y0 * cc[0] + y1 * cc[1] + y2 * cc[2];
            }
            tmp[t] = v;
        }
    }

    // Pair e = a (a + 1) / 2 + b (a >= b) of compact columns -> two ints that make the per-knot contraction branch-free:
    //   w0: row | col << 8 | i1 << 16 | i2 << 24     z indices of the Q entry; tmp slots of the two direct terms (T_ZERO: none)
    //   w1: qa | colb << 2 | m1 << 9 | qb << 10 | cola << 12 | m2 << 19 | qo << 20 | so << 22 | c1 << 24 | bar << 26
    //       -(dI_qa y) . A[:, colb] if m1, -(dI_qb y) . A[:, cola] if m2; quaternion-rate term so (1: +, 2: -) * v'_o[qo];
    //       c1: coefficient of the first direct term (0: -1, 1: +lever, 2: -lever); the second one is -1;
    //       bar: i2 holds contact << 4 | fa << 2 | fb of a barrier block instead of a tmp slot.
    // Built once per instance into LDS (sweep tables); the class logic is the one of oracle/models.py srbd_wdot_hess.
    __device__ __forceinline__ static void so2_pair_code(int e, int& w0, int& w1) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= e) ++a;
        const int b = e - a * (a + 1) / 2;
        const bool ao = a >= 3 && a < 7, bo = b >= 3 && b < 7, aw = a >= 7 && a < 10, bw = b >= 7 && b < 10;
        const bool af = a >= AF, br = b < 3, bc = CS && b >= AC && b < AF;
        int i1 = T_ZERO, i2 = T_ZERO, c1 = 0, bar = 0, so = 0, qo = 0;
        if (af && (br || bc)) {          // d2 n / (dc df) = s e_c x e_f, d2 n / (dr df) = -s e_r x e_f
            const int i = (a - AF) / 3, fa = (a - AF) % 3;
            const bool hit = br || (b - AC) / 3 == i;
            const int xa = br ? b : (b - AC) % 3;
            if (hit && xa != fa) {
                const bool pos = ((fa - xa + 3) % 3 == 1);                 // (e_x x e_f)[third] = +1
                i1 = T_Y + (3 - xa - fa);
                c1 = (pos != br) ? 1 : 2;                                  // (br ? -lever : lever) * (pos ? 1 : -1)
            }
        } else if (aw && bw) {
            i1 = T_P + 3 * (a - 7) + (b - 7);
            i2 = T_P + 3 * (b - 7) + (a - 7);
        } else if (aw && bo) {
            i1 = T_H + 3 * (b - 3) + (a - 7);
            // quaternion rate: sum_q v'_o[q] d2 odot_q / do_b dw_c = 1/2 (+-) v'_o[q] for exactly one q
            const int ob = b - 3, cc = a - 7;
            for (int q = 0; q < 4; ++q) {
                if (q == ob) continue;
                int comp; bool plus;
                if (q == 3) { comp = ob; plus = false; }
                else if (ob == 3) { comp = q; plus = true; }
                else { comp = 3 - q - ob; plus = !((ob - q + 3) % 3 == 1); }
                if (comp == cc) { qo = q; so = plus ? 1 : 2; }
            }
        } else if (ao && bo) {
            i1 = T_D + (a - 3) * (a - 2) / 2 + (b - 3);
        }
        if (BAR && a >= AF && b >= AF && (a - AF) / 3 == (b - AF) / 3) {
            bar = 1;
            i2 = (((a - AF) / 3) << 4) | (((a - AF) % 3) << 2) | ((b - AF) % 3);
        }
        w0 = zcol(a) | (zcol(b) << 8) | (i1 << 16) | (i2 << 24);
        w1 = (ao ? a - 3 : 0) | (b << 2) | ((ao ? 1 : 0) << 9) | ((bo ? b - 3 : 0) << 10) | (a << 12) | ((bo ? 1 : 0) << 19) | (qo << 20) |
             (so << 22) | (c1 << 24) | (bar << 26);
    }

    template <class QM>
    __device__ __forceinline__ static void add_second_order(const DevConsts& c, const double* rec, const double* vp, QM Q,
                                                            double theta, int lane, int nlanes, const double* tmp = nullptr,
                                                            const int* lut = nullptr) {
        if (SO2) {
            // full term (second_order = 2): Q += theta * (sum_m lam_m d2 wdot_m + dt v'_o . d2 odot), lam = dt v'_w + 2 gq wdot: the
            // dynamics tensor contracted with v' plus the exact-minus-Gauss-Newton Hessian of the wdot rows of min_qddot.
            // lam . d_a d_b wdot = y . V_ab, y = I_w^-1 lam (so2_record): one lane per pair (a >= b) of compact columns, from the
            // per-knot factors of so2_prepare (tmp) and the pair codes (lut), every operand requested before the first use:
            //   f_i x (r | c_i):  +- s y . (e x e)                                       (the bilinear torque)
            //   w x w:            -(P[a][b] + P[b][a]),  P[a][b] = (I_w e_b x y)_a        (gyroscopic term)
            //   w_a x o_b:        -h_b[a],  h_b = (dI_b w) x y + dI_b (y x w)
            //   o_a x o_b:        -y . c_ab
            //   o_a x any b:      -(dI_a y) . A[:, b]      (and symmetrically for b in o)
            for (int e = lane; e < NTRI; e += nlanes) {
                const int w0 = lut[2 * e], w1 = lut[2 * e + 1];
                const int row = w0 & 255, col = (w0 >> 8) & 255, i1 = (w0 >> 16) & 255, i2f = (w0 >> 24) & 255;
                const int qa = w1 & 3, colb = (w1 >> 2) & 127, qb = (w1 >> 10) & 3, cola = (w1 >> 12) & 127, qo = (w1 >> 20) & 3;
                const int so = (w1 >> 22) & 3, c1 = (w1 >> 24) & 3;
                const bool m1 = (w1 >> 9) & 1, m2 = (w1 >> 19) & 1, bar = BAR && ((w1 >> 26) & 1);
                const int i2 = bar ? int(T_ZERO) : i2f;
                const double t1 = tmp[i1], t2 = tmp[i2], vo = vp[XO + qo];
                const double* g1 = tmp + T_G + 3 * qa;
                const double* g2 = tmp + T_G + 3 * qb;
                const double g10 = g1[0], g11 = g1[1], g12 = g1[2], g20 = g2[0], g21 = g2[1], g22 = g2[2];
                const double a10 = rec[REC_A + colb], a11 = rec[REC_A + NA + colb], a12 = rec[REC_A + 2 * NA + colb];
                const double a20 = rec[REC_A + cola], a21 = rec[REC_A + NA + cola], a22 = rec[REC_A + 2 * NA + cola];
                double q0 = Q.at(row, col);
                const double cf = c1 == 0 ? -1.0 : (c1 == 1 ? c.lever : -c.lever);
                double sv = cf * t1 - t2;
                sv -= m1 ? (g10 * a10 + g11 * a11 + g12 * a12) : 0.0;
                sv -= m2 ? (g20 * a20 + g21 * a21 + g22 * a22) : 0.0;
                double val = theta * sv + (so == 0 ? 0.0 : (so == 1 ? 0.5 : -0.5) * theta * c.dt * vo);
                if (bar) val += theta * barrier_h(rec + REC_B + 5 * (i2f >> 4), (i2f >> 2) & 3, i2f & 3);
                q0 += val;
                Q.at(row, col) = q0;
                if (row != col) Q.at(col, row) = q0;     // Q is symmetric here: both triangles hold the same value
            }
            return;
        }
        for (int e = lane; e < NSO; e += nlanes) {
            const int blk = e / 9, a = (e % 9) / 3, b = e % 3;
            const int i = blk % NC;
            const bool isc = blk >= NC;
            const int yi = (a == b) ? 0 : 3 - a - b;                      // index of the y component in skew(y)[a][b], a != b
            const double* mi = rec + REC_MI + 3 * yi;
            const int row = NX + uf(i) + a, col = isc ? XC + 3 * i + b : XR + b;
            // every operand (and the two Q entries to update) requested before the first use: one LDS round trip, not five
            double v0 = vp[XW], v1 = vp[XW + 1], v2 = vp[XW + 2], m0 = mi[0], m1 = mi[1], m2 = mi[2];
            double q0 = Q.at(row, col), q1 = Q.at(col, row);
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(q0), "+v"(q1));
            const double l0 = c.dt * v0, l1 = c.dt * v1, l2 = c.dt * v2;
            const double y = m0 * l0 + m1 * l1 + m2 * l2;
            const int d = (b - a + 3) % 3;                                // 2: +y , 1: -y , 0: diagonal (zero)
            const double sk = d == 2 ? y : (d == 1 ? -y : 0.0);
            const double val = theta * c.lever * (isc ? sk : -sk);
            Q.at(row, col) = q0 + val;
            Q.at(col, row) = q1 + val;
        }
    }

    // The constant extra rows' share of Q (rows m >= NEV: rddot and, with contact states, the relative-velocity penalties; weights
    // lam_stage) as a SPARSE pass, one thread per non-zero element of the lower triangle, for kernels that cannot afford to keep it
    // as a per-thread constant block (the 4-wavefront kernel with several Q blocks per thread: srbd61).  rddot_a = inv_ms sum_i f_i,a:
    // 2 gq inv_ms^2 on every (f_i,a ; f_j,a); relative velocity, component e: 2 w_pen (pairs of contact i) on the diagonal of cdot_i,e,
    // -2 w_pen on (cdot_lead,e ; cdot_follower,e) (prb.py:166-170).  Disjoint from the entries add_second_order touches (f x r, f x c).
    static constexpr int NFF = NC * (NC + 1) / 2, NCONST = 3 * NFF + (CS ? 2 * (NC + NRV) : 0);
    template <class QM>
    __device__ __forceinline__ static void add_const_rows(const DevConsts& c, QM Q, int tid, int nthreads) {
        for (int e = tid; e < NCONST; e += nthreads) {
            int row, col;
            double val;
            if (e < 3 * NFF) {
                const int ax = e / NFF, t = e % NFF;
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                const int j = t - i * (i + 1) / 2;
                row = NX + uf(i) + ax; col = NX + uf(j) + ax;
                val = 2 * c.gq * c.inv_ms * c.inv_ms;
            } else {
                const int e2 = e - 3 * NFF, comp = e2 / (NC + NRV), t = e2 % (NC + NRV);
                if (t < NC) { row = col = XCD + 3 * t + comp; val = 2 * c.w_rv * double(rv_count(t)); }
                else { const int q = t - NC; row = XCD + 3 * rv_b(q) + comp; col = XCD + 3 * rv_a(q) + comp; val = -2 * c.w_rv; }
            }
            const double q0 = Q.at(row, col) + val;
            Q.at(row, col) = q0;
            if (row != col) Q.at(col, row) = q0;
        }
    }

    // BAR builds: the barrier's Gauss-Newton Hessian blocks (3x3 per contact force, from the record) added to Quu
    // ... and the diagonal of the bound barrier; so2_theta (SO2 builds: theta of this sweep, else 0): its exact Hessian is twice the
    // Gauss-Newton one, like the friction barrier's (whose share is added in add_second_order)
    template <class QM>
    __device__ __forceinline__ static void add_barrier(const double* rec, QM Q, int lane, int nlanes, double so2_theta = 0.0) {
        for (int e = lane; e < 9 * NC + NZ; e += nlanes) {      // every entry of Q is updated by exactly one lane
            if (e < 9 * NC) {
                const int i = e / 9, a = (e % 9) / 3, b = e % 3, j = NX + uf(i) + a;
                double v = barrier_h(rec + REC_B + 5 * i, a, b);
                if (a == b) v += (1.0 + so2_theta) * rec[REC_BB + j];          // the force diagonals take their bound term here
                Q.at(j, NX + uf(i) + b) += v;
            } else {
                const int j = e - 9 * NC;
                const bool force = j >= NX && (!CS || (j - NX) % 6 >= 3);
                if (!force) Q.at(j, j) += (1.0 + so2_theta) * rec[REC_BB + j];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// LIP (prb.py:248-441): x = r | c0..3 | rdot | cdot0..3 (30), u = z | cddot0..3 (15), p = rdot_ref | (c_ref_i, sw_i)x4.
// Linear dynamics + quadratic cost: F and H are constant in (x,u); the record is the gradient only.
// ---------------------------------------------------------------------------------------------------------
template <int XR_ = 0>
struct LipModel {
    static constexpr int NC = 4;
    static constexpr int NXR = XR_;        // user-declared linear residual rows (see SrbdModel)
    static_assert(XR_ == 0 || XR_ == kXrRows, "0 or kXrRows user rows");
    static constexpr bool BAR = false;
    template <class QM> __device__ __forceinline__ static void add_barrier(const double*, QM, int, int, double = 0.0) {}
    template <class QM> __device__ __forceinline__ static void add_const_rows(const DevConsts&, QM, int, int) {}   // (one Q block per thread: unused)
    static constexpr int NX = 30, NU = 15, NZ = 45, NPB = 11, NP = NPB + NXR;
    static constexpr int XR = 0, XC = 3, XRD = 15, XCD = 18;
    static constexpr int REC_G = 0, NREC = NZ, NSO2T = 0, NSO2L = 0;
    static constexpr bool SO2 = false;
    __device__ __forceinline__ static void so2_prepare(const DevConsts&, const double*, const double*, double*, int, int) {}
    __device__ __forceinline__ static void so2_pair_code(int, int&, int&) {}
    __device__ __forceinline__ static void so2_knot(const DevConsts&, const double*, const double*, const double*, double*) {}
    __device__ __forceinline__ static double p_cref(const double* p, int i) { return p[3 + 2 * i]; }
    __device__ __forceinline__ static double p_sw(const double* p, int i) { return p[4 + 2 * i]; }

    template <class XV>
    __device__ __forceinline__ static double state_cost(const DevConsts& c, XV x, const double* p) {
        const double ez = x[2] - c.com_z;
        double L = c.w_rz * ez * ez;                                              // rz_tracking  prb.py:390
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double m = 0.25 * (x[XC + a] + x[XC + 3 + a] + x[XC + 6 + a] + x[XC + 9 + a]);
            const double e = x[a] - m;
            L += c.w_rxy * e * e;                                                 // rxy_tracking prb.py:391
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { const double e = x[XRD + a] - p[a]; L += c.w_rd * e * e; }   // prb.py:392
        const double r1y = -x[XC + 1] + x[XC + 7] - c.d1y, r1x = -x[XC + 0] + x[XC + 6] - c.d1x;
        const double r2y = -x[XC + 4] + x[XC + 10] - c.d2y, r2x = -x[XC + 3] + x[XC + 9] - c.d2x;
        L += c.w_rel * (r1y * r1y + r1x * r1x + r2y * r2y + r2x * r2x);           // prb.py:394-401
        return L;
    }

    template <class XV, class UV, class XN>
    __device__ __forceinline__ static double step(const DevConsts& c, XV x, UV u, const double* p, int k, XN xn) {
        double L = 0, rddot[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rddot[a] = c.eta2 * (x[a] - u[a]) - (a == 2 ? kGravity : 0.0);        // prb.py:317-319
            const double m = 0.25 * (x[XC + a] + x[XC + 3 + a] + x[XC + 6 + a] + x[XC + 9 + a]);
            const double e = u[a] - m;
            L += c.w_zmp * e * e + c.gq * rddot[a] * rddot[a];                    // prb.py:393, :402
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) L += c.gq * u[3 + i] * u[3 + i];
#pragma unroll
        for (int i = 0; i < NC; ++i) {                                            // penalties prb.py:379-387
            const double ez = x[XC + 3 * i + 2] - p_cref(p, i), sw = p_sw(p, i);
            const double vx = sw * x[XCD + 3 * i], vy = sw * x[XCD + 3 * i + 1];
            L += c.w_pen * (ez * ez + vx * vx + vy * vy);
        }
#pragma unroll
        for (int b = 0; b < NC; b += 2) {
            const double ex = x[XCD + 3 * b] - x[XCD + 3 * b + 3], ey = x[XCD + 3 * b + 1] - x[XCD + 3 * b + 4];
            L += c.w_rv * (ex * ex + ey * ey);
        }
        if (k >= 1) L += state_cost(c, x, p);
        if (NXR) L += xr_eval<NX, NU>(c, x, u, true, p + NPB, k >= 1, true);
        const double dt = c.dt;
        double xp[NX];                                                            // read everything before the first store
#pragma unroll
        for (int i = 0; i < 15; ++i) xp[i] = x[i] + dt * x[15 + i];               // q += dt qdot  prb.py:323-328
#pragma unroll
        for (int a = 0; a < 3; ++a) xp[XRD + a] = x[XRD + a] + dt * rddot[a];
#pragma unroll
        for (int i = 0; i < 12; ++i) xp[XCD + i] = x[XCD + i] + dt * u[3 + i];
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = xp[i];
        return L;
    }

    template <class XV>
    __device__ __forceinline__ static double term_cost(const DevConsts& c, XV x, const double* p) {
        double L = state_cost(c, x, p);
        if (NXR) L += xr_eval<NX, NU>(c, x, x, false, p + NPB, true, false);
        return L;
    }

    __device__ __forceinline__ static void derivs(const DevConsts& c, const double* x, const double* u, const double* p,
                                                  int k, int N, double* rec) {
        double g[NZ];
#pragma unroll
        for (int i = 0; i < NZ; ++i) g[i] = 0.0;
        if (k >= 1) {
            g[2] += 2 * c.w_rz * (x[2] - c.com_z);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const double m = 0.25 * (x[XC + a] + x[XC + 3 + a] + x[XC + 6 + a] + x[XC + 9 + a]);
                const double e = 2 * c.w_rxy * (x[a] - m);
                g[a] += e;
#pragma unroll
                for (int i = 0; i < NC; ++i) g[XC + 3 * i + a] -= 0.25 * e;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) g[XRD + a] += 2 * c.w_rd * (x[XRD + a] - p[a]);
            const double r1y = -x[XC + 1] + x[XC + 7] - c.d1y, r1x = -x[XC + 0] + x[XC + 6] - c.d1x;
            const double r2y = -x[XC + 4] + x[XC + 10] - c.d2y, r2x = -x[XC + 3] + x[XC + 9] - c.d2x;
            const double s = 2 * c.w_rel;
            g[XC + 1] -= s * r1y; g[XC + 7] += s * r1y; g[XC + 0] -= s * r1x; g[XC + 6] += s * r1x;
            g[XC + 4] -= s * r2y; g[XC + 10] += s * r2y; g[XC + 3] -= s * r2x; g[XC + 9] += s * r2x;
        }
        if (k < N) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double m = 0.25 * (x[XC + a] + x[XC + 3 + a] + x[XC + 6 + a] + x[XC + 9 + a]);
                const double e = 2 * c.w_zmp * (u[a] - m);
                g[NX + a] += e;
#pragma unroll
                for (int i = 0; i < NC; ++i) g[XC + 3 * i + a] -= 0.25 * e;
                const double rdd = c.eta2 * (x[a] - u[a]) - (a == 2 ? kGravity : 0.0);
                const double t = 2 * c.gq * c.eta2 * rdd;
                g[a] += t;
                g[NX + a] -= t;
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) g[NX + 3 + i] += 2 * c.gq * u[3 + i];
            const double sp = 2 * c.w_pen;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                g[XC + 3 * i + 2] += sp * (x[XC + 3 * i + 2] - p_cref(p, i));
                const double sw = p_sw(p, i);
                g[XCD + 3 * i] += sp * sw * sw * x[XCD + 3 * i];
                g[XCD + 3 * i + 1] += sp * sw * sw * x[XCD + 3 * i + 1];
            }
#pragma unroll
            for (int b = 0; b < NC; b += 2) {
                const double ex = x[XCD + 3 * b] - x[XCD + 3 * b + 3], ey = x[XCD + 3 * b + 1] - x[XCD + 3 * b + 4];
                const double sr = 2 * c.w_rv;
                g[XCD + 3 * b] += sr * ex; g[XCD + 3 * b + 3] -= sr * ex;
                g[XCD + 3 * b + 1] += sr * ey; g[XCD + 3 * b + 4] -= sr * ey;
            }
        }
        if (NXR) (void)xr_eval<NX, NU>(c, x, u, k < N, p + NPB, k >= 1, k < N, g);     // user rows: their gradient
#pragma unroll
        for (int i = 0; i < NZ; ++i) rec[REC_G + i] = g[i];
    }

    enum { V_R = 0, V_C, V_RD, V_CD, V_Z, V_CDD };
    __device__ __forceinline__ static void decode(int j, int& cls, int& ci, int& ax) {
        ci = 0;
        if (j < 3) { cls = V_R; ax = j; return; }
        if (j < 15) { cls = V_C; ci = (j - 3) / 3; ax = (j - 3) % 3; return; }
        if (j < 18) { cls = V_RD; ax = j - 15; return; }
        if (j < 30) { cls = V_CD; ci = (j - 18) / 3; ax = (j - 18) % 3; return; }
        if (j < 33) { cls = V_Z; ax = j - 30; return; }
        cls = V_CDD; ci = (j - 33) / 3; ax = (j - 33) % 3;
    }

    __device__ __forceinline__ static double F_entry(const DevConsts& c, const double*, int i, int j) {
        int ci, cli, ai, cj, clj, aj;
        decode(i, cli, ci, ai);
        decode(j, clj, cj, aj);
        double s = 0.0;
        if (ai == aj) {
            switch (cli) {
                case V_R: s = clj == V_RD ? 1.0 : 0.0; break;
                case V_C: s = (clj == V_CD && ci == cj) ? 1.0 : 0.0; break;
                case V_RD: s = clj == V_R ? c.eta2 : (clj == V_Z ? -c.eta2 : 0.0); break;
                case V_CD: s = (clj == V_CDD && ci == cj) ? 1.0 : 0.0; break;
                default: break;
            }
        }
        return (i == j ? 1.0 : 0.0) + c.dt * s;
    }

    __device__ __forceinline__ static double H_entry(const DevConsts& c, const double*, const double* p, int k, int N, int i, int j) {
        int ci, cli, ai, cj, clj, aj;
        decode(i, cli, ci, ai);
        decode(j, clj, cj, aj);
        const bool state = k >= 1, stage = k < N;
        double vx = 0.0;
        if (NXR) {   // user rows: 2 w_j a_j a_j^T with the weight of the node
            for (int r = 0; r < c.xr_n; ++r) {
                const double w = (state ? c.xr[kXrWS + r] : 0.0) + (stage ? c.xr[kXrWG + r] : 0.0);
                vx += 2 * w * c.xr[r * kXrStride + i] * c.xr[r * kXrStride + j];
            }
        }
        if (ai != aj) return vx;
        if (cli > clj) { int t = cli; cli = clj; clj = t; t = ci; ci = cj; cj = t; }
        const double e4 = 2 * c.gq * c.eta2 * c.eta2;
        double v = 0.0;
        if (cli == V_R && clj == V_R) {
            if (state) v += ai == 2 ? 2 * c.w_rz : 2 * c.w_rxy;
            if (stage) v += e4;
        } else if (cli == V_R && clj == V_C) {
            if (state && ai < 2) v = -0.5 * c.w_rxy;
        } else if (cli == V_R && clj == V_Z) {
            if (stage) v = -e4;
        } else if (cli == V_C && clj == V_C) {
            if (state && ai < 2) {
                v += 0.125 * c.w_rxy;
                if (ci == cj) v += 2 * c.w_rel; else if (ci + 2 == cj || cj + 2 == ci) v -= 2 * c.w_rel;
            }
            if (stage) {
                v += 0.125 * c.w_zmp;
                if (ci == cj && ai == 2) v += 2 * c.w_pen;
            }
        } else if (cli == V_C && clj == V_Z) {
            if (stage) v = -0.5 * c.w_zmp;
        } else if (cli == V_RD && clj == V_RD) {
            if (state) v = 2 * c.w_rd;
        } else if (cli == V_CD && clj == V_CD) {
            if (stage && ai < 2) {
                if (ci == cj) { const double sw = p_sw(p, ci); v = 2 * c.w_rv + 2 * c.w_pen * sw * sw; }
                else if (ci / 2 == cj / 2) v = -2 * c.w_rv;
            }
        } else if (cli == V_Z && clj == V_Z) {
            if (stage) v = 2 * c.w_zmp + e4;
        } else if (cli == V_CDD && clj == V_CDD) {
            if (stage && ci == cj) v = 2 * c.gq;
        }
        return v + vx;
    }
    // ---- branch-free expansion hooks (see SrbdModel): everything is constant, all couplings are extra rows
    static constexpr int NEB = 16;  // rxy(2) zmp(3) rddot(3) rel_pos(4) rel_vel(4)
    static constexpr int NE = NEB + NXR;   // ... then the user rows
    // product rows first: rxy(2) and rel_pos(4) carry state-node weights; then zmp(3) rddot(3) rel_vel(4): constant rows with
    // node-independent weights, which enter Q as a constant matrix
    static constexpr int NEVB = 6, NEV = NEVB + NXR;     // the user rows go through the products too (node-dependent weights)
    static constexpr bool CONST_ROWS_STATE_WEIGHTED = false;
    __device__ __forceinline__ static int erow(int m) {
        if (NXR) {
            if (m >= NEVB && m < NEV) return NEB + (m - NEVB);
            if (m >= NEV) m -= NXR;
        }
        if (m < 2) return m;                     // rxy
        if (m < 6) return m + 6;                 // rel_pos   (definition rows 8..11)
        if (m < 12) return m - 4;                // zmp, rddot (definition rows 2..7)
        return m;                                // rel_vel   (definition rows 12..15)
    }
    __device__ static double E_const(const DevConsts& c, int mrow, int j) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? c.xr[(m - NEB) * kXrStride + j] : 0.0;   // user rows
        int cls, ci, ax;
        decode(j, cls, ci, ax);
        if (m < 2) { if (ax != m) return 0.0; return cls == V_R ? 1.0 : (cls == V_C ? -0.25 : 0.0); }           // prb.py:391
        if (m < 5) { if (ax != m - 2) return 0.0; return cls == V_Z ? 1.0 : (cls == V_C ? -0.25 : 0.0); }       // prb.py:393
        if (m < 8) { if (ax != m - 5) return 0.0; return cls == V_R ? c.eta2 : (cls == V_Z ? -c.eta2 : 0.0); }  // prb.py:317,:402
        if (m < 12) {
            const int pair = (m - 8) / 2, comp = ((m - 8) % 2 == 0) ? 1 : 0;
            if (cls != V_C || ax != comp) return 0.0;
            return ci == pair ? -1.0 : (ci == pair + 2 ? 1.0 : 0.0);
        }
        const int pair = (m - 12) / 2, comp = (m - 12) % 2;
        if (cls != V_CD || ax != comp) return 0.0;
        return ci == 2 * pair ? 1.0 : (ci == 2 * pair + 1 ? -1.0 : 0.0);
    }
    __device__ static double lam_state(const DevConsts& c, int mrow) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? 2 * c.xr[kXrWS + m - NEB] : 0.0;
        return m < 2 ? 2 * c.w_rxy : ((m >= 8 && m < 12) ? 2 * c.w_rel : 0.0);
    }
    __device__ static double lam_stage(const DevConsts& c, int mrow) {
        const int m = erow(mrow);
        if (NXR && m >= NEB) return (c.xr && m - NEB < c.xr_n) ? 2 * c.xr[kXrWG + m - NEB] : 0.0;
        if (m < 2) return 0.0;
        if (m < 5) return 2 * c.w_zmp;
        if (m < 8) return 2 * c.gq;
        return m >= 12 ? 2 * c.w_rv : 0.0;
    }
    __device__ static double dg_state(const DevConsts& c, int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        if (cls == V_R) return ax == 2 ? 2 * c.w_rz : 0.0;
        if (cls == V_RD) return 2 * c.w_rd;
        return 0.0;
    }
    __device__ static double dg_stage(const DevConsts& c, int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        if (cls == V_C) return ax == 2 ? 2 * c.w_pen : 0.0;
        if (cls == V_CDD) return 2 * c.gq;
        return 0.0;
    }
    __device__ static int dkind(int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        return (cls == V_CD && ax < 2) ? 3 : 0;
    }
    __device__ static int dci(int i) {
        int cls, ci, ax;
        decode(i, cls, ci, ax);
        return ci;
    }
    __device__ __forceinline__ static double dparam(const DevConsts& c, const double* p, int kind, int ci, double state, double stage) {
        double sw = p[4 + 2 * ci];
        asm volatile("" : "+v"(sw));                       // unconditional read (see SrbdModel::dparam)
        return kind == 3 ? stage * 2 * c.w_pen * sw * sw : 0.0;
    }
    __device__ __forceinline__ static void expand_var(const DevConsts&, const double*, double*, int, int) {}
    template <bool COMPACT = false>
    __device__ __forceinline__ static void expand_var(const DevConsts&, const double*, double*, int, int, int, double* = nullptr,
                                                      const double* = nullptr) {}
    // sparsity of [fx fu] by column (see SrbdModel): no dense rows, every column has the identity and at most one other entry
    static constexpr int ND = 0;
    __device__ __forceinline__ static constexpr int dense_row(int) { return 0; }
    __device__ __forceinline__ static constexpr int dense_col(int row) { return row >= NX ? row - NX : -1; }
    __device__ static void nbr(const DevConsts& c, int z, int& n, double& beta, double& idc) {
        int cls, ci, ax;
        decode(z, cls, ci, ax);
        idc = z < NX ? 1.0 : 0.0;
        n = 0; beta = 0.0;
        switch (cls) {
            case V_R: n = XRD + ax; beta = c.dt * c.eta2; break;              // F_entry: rdot row, r column
            case V_RD: n = XR + ax; beta = c.dt; break;
            case V_CD: n = XC + 3 * ci + ax; beta = c.dt; break;
            case V_Z: n = XRD + ax; beta = -c.dt * c.eta2; break;
            case V_CDD: n = XCD + 3 * ci + ax; beta = c.dt; break;
            default: break;
        }
    }
    template <class QM>
    __device__ __forceinline__ static void add_second_order(const DevConsts&, const double*, const double*, QM, double, int, int,
                                                            const double* = nullptr, const int* = nullptr) {}

};

using Srbd13 = SrbdModel<2, false>;
using Srbd37 = SrbdModel<4, true>;
using Srbd13B = SrbdModel<2, false, true>;   // with the friction-cone barrier (sddp_model_consts.friction_barrier_weight > 0)
using Srbd37B = SrbdModel<4, true, true>;
using Srbd13S = SrbdModel<2, false, false, true>;   // full second-order builds (sddp_options.second_order = 2)
using Srbd37S = SrbdModel<4, true, false, true>;
using Srbd13BS = SrbdModel<2, false, true, true>;    // barrier + full second order
using Srbd37BS = SrbdModel<4, true, true, true>;
using Srbd61 = SrbdModel<8, true>;                  // contact_model = 4 (prb.py:39-41); Srbd61X / Srbd61B below, no second_order = 2 build
using Lip30 = LipModel<>;
using Srbd13X = SrbdModel<2, false, false, false, kXrRows>;   // with user-declared linear residual rows (sddp_model_consts.n_extra > 0)
using Srbd37X = SrbdModel<4, true, false, false, kXrRows>;
using Lip30X = LipModel<kXrRows>;
using Srbd61X = SrbdModel<8, true, false, false, kXrRows>;
using Srbd61B = SrbdModel<8, true, true>;            // friction-cone barrier only (BOX = false)

}  // namespace sddp
