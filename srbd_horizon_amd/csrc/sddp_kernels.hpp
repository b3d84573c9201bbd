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
#ifndef SDDP_KSLOTS
#define SDDP_KSLOTS 6
#endif
constexpr int kSlots = SDDP_KSLOTS;   // line-search candidates whose trajectories are kept per pass (one-wave kernel)
constexpr int kScal = 24;  // doubles per instance in the scratch `scal` record (test kernels / diagnostic stamps)

// Diagnostic build only (-DSDDP_STAMPS): per-phase shader-cycle sums, written to `scal`; never in the shipped library.
#ifdef SDDP_STAMPS
#define SDDP_T_DECL unsigned long long T_[kScal] = {0}; unsigned long long t_last_ = clock64();
#define SDDP_T_ARG , unsigned long long* T_, unsigned long long& t_last_
#define SDDP_T_PASS , T_, t_last_
#define SDDP_TICK(i) { const unsigned long long t_ = clock64(); T_[i] += t_ - t_last_; t_last_ = t_; }
#elif defined(SDDP_MARKS)
// Listing build only (-DSDDP_MARKS, `hipcc -S`): the phase boundaries as assembler comments, for tools/isa_phase_mix.py.  No
// instruction is emitted; the volatile asm only keeps the compiler from moving code across a boundary.
#define SDDP_T_DECL
#define SDDP_T_ARG
#define SDDP_T_PASS
#define SDDP_TICK(i) asm volatile("; SDDP_MARK " #i ::: "memory");
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
    double* xn;         // [B][N+1][NX]  candidate (4-wave kernels, single-phase test kernels)
    double* un;         // [B][N][NU]
    double* xc;         // [B][2][kSlots][N+1][NX]  line-search candidates of the one-wave kernel: the first kSlots step lengths of
    double* uc;         // [B][2][kSlots][N][NU]    a pass each keep their trajectory; two sets, written alternately
    double* dft;        // [B][N][NX]    defects d_{k+1} stored at k
    double* gains;      // [B][N][NU*(NX+1)]  kff (NU) then K (NU x NX, row-major)
    double* rec;        // [B][N+1][NREC]
    sddp_stats* stats;  // [B]
    double* scal;       // [B][kScal] (test kernels / diagnostic stamps)
    double alpha;       // forward test kernel only
    double mu;          // backward test kernel only
    // work queue (solve kernels).  The launch covers `count` instances; the grid is the resident workgroups ("slots"), at most
    // `count` of them.  qhead == nullptr: workgroup w solves instance first + w (grid == count).  Otherwise every workgroup
    // takes queue positions i = atomicAdd(qhead, 1) until i >= count and solves instance order[i] (order == nullptr:
    // first + i).  The work buffers xn/un/xc/uc/dft/gains/rec are per SLOT (indexed by blockIdx.x), x0/P/xs/us/stats/scal/hist
    // per instance.
    int first, count;
    int* qhead;
    const int* order;   // [count] absolute instance indices, or nullptr
    int* hist;          // [B] iterations of the last solve of each instance (-1: never solved): the queue-order key; behind it (no
                        // kernel argument of its own: the solve kernels live on their scalar registers) the slot clocks [slots][2]:
                        // constant-rate clock (wall_clock64, 100 MHz) when a slot started its first instance and when it found the
                        // queue empty -- how long a launch drains (bench.py drain_frac)
    __device__ __forceinline__ unsigned long long* slot_clock(int slot) const {
        return reinterpret_cast<unsigned long long*>(hist + ((B + 1) & ~1)) + 2 * slot;
    }
};

// Hand-off between phases of a ONE-WAVEFRONT workgroup.  LDS (and global) accesses of one wave are performed in issue order, so a
// later ds_read of any lane sees an earlier ds_write of any lane without waiting; what must not happen is the COMPILER moving
// accesses across the hand-off.  A wavefront-scope fence does exactly that and costs no instruction, where __syncthreads()
// costs an s_waitcnt lgkmcnt(0) that also drains loads still in flight.  (The 4-wave kernels use real barriers.)
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt 0, expcnt / lgkmcnt at their maxima): every load and store of this wave has completed.
// In front of a knot loop: whatever the code before the loop left in flight is otherwise "pending" on the loop's entry edge in the
// compiler's wait-count bookkeeping, and the waits it inserts for that stay inside the loop body, behind the knot's prefetch.
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
// value of lane `l` (compile-time constant after unrolling) broadcast to the wave through SGPRs: 2 x v_readlane_b32
__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
    return v;
}

// LDS tile layout of one instance (offsets in doubles).  Tiles are stored so that every inner product of the sweep
// runs over CONTIGUOUS, 16-byte aligned rows (ds_read_b128, two fp64 per load):
//   VXX [NXP][NXP]  Vxx+ (symmetric; pad row/col zero)
//   FT  [NZP][NIP]  F~^T : FT[j][l] = F~[l][j],  F~ = [fx fu ; Je]  (NX dynamics rows + NE extra residual rows)
//   WT  [NZP][NIP]  (V~ F~)^T : WT[j][l<NX] = sum_m Vxx[l][m] F[m][j] ;  WT[j][NX+m] = lambda_m(k) * Je[m][j]
//   Q   [NZP][NZP]  diag(D) + F~^T (V~ F~)   (symmetric, both triangles written)
template <class M>
struct Lds {
    static constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NEV = M::NEV, NI = NX + NEV;   // product rows only
    static constexpr int NXP = (NX + 1) & ~1;
    static constexpr int NIP = (((NI + 1) & ~1) % 4 == 0) ? ((NI + 1) & ~1) + 2 : ((NI + 1) & ~1);   // even, == 2 (mod 4): rows on distinct 16-B slots
    static constexpr int NZP = (NZ + 3) & ~3;
    static constexpr int SQ = (NZP % 4 == 0) ? NZP + 2 : NZP;          // row stride of the Q tile: == 2 (mod 4) as well
    static constexpr int NUP = (NU + 1) & ~1;
    static constexpr int NRECP = (((M::NREC + 1) & ~1) + M::NSO2T + 1) & ~1;   // record, then the second-order factors (SO2 builds)
    static constexpr int SO2T = (M::NREC + 1) & ~1;                             // ... at this offset from REC
    static constexpr int NPP = (M::NP + 1) & ~1;
    static constexpr int VXX = 0;
    static constexpr int FT = VXX + NXP * NXP;
    static constexpr int WT = FT + NZP * NIP;
    static constexpr int Q = WT + NZP * NIP;
    static constexpr int VX = Q + NZP * SQ;
    static constexpr int VP = VX + NXP;
    static constexpr int DK = VP + NXP;
    static constexpr int QV = DK + NXP;
    static constexpr int REC = QV + NZP;
    static constexpr int PK = REC + NRECP;
    static constexpr int KT = PK + NPP;            // [NXP][NUP]: KT[c][i] = K[i][c]
    static constexpr int KF = KT + NXP * NUP;      // kff [NUP]
    static constexpr int PIV = KF + NUP;           // [NUP]
    static constexpr int DS = PIV + NUP;           // constant diagonal, state part   [NZP]
    static constexpr int DG = DS + NZP;            // constant diagonal, stage part   [NZP]
    static constexpr int LS = DG + NZP;            // extra-row weights, state / stage [NE] each
    static constexpr int LG = LS + ((NE + 1) & ~1);
    static constexpr int LAM = LG + ((NE + 1) & ~1);   // LS + LG: extra-row weights of a stage node k >= 1
    static constexpr int KI = LAM + ((NE + 1) & ~1);   // ints: dkind[NZP], dci[NZP], tri LUT
    static constexpr int NBQ = NZP / 2, NTRIQ = NBQ * (NBQ + 1) / 2;
    static constexpr int ntriv(int nx) { int n = 0; for (int a = 0; a < nx; ++a) n += a / 2 + 1; return n; }
    static constexpr int NTRIV = ntriv(NX);        // Vxx update: 1 x 2 blocks (row a, column pair 2bp, 2bp+1 with 2bp <= a)
    static constexpr int SO2L = 2 * NZP + NTRIQ + NTRIV;       // SO2 builds: pair codes of the second-order contraction
    static constexpr int KI_INTS = SO2L + M::NSO2L;
    static constexpr int RO = KI + (KI_INTS + 1) / 2;  // rollout staging: xs | us | p | gains | dft of one knot
    static constexpr int RO_X = 0, RO_U = NXP, RO_P = RO_U + NUP, RO_G = RO_P + NPP, RO_D = RO_G + ((NU * (NX + 1) + 1) & ~1);
    static constexpr int RO_N = RO_D + NXP;
    static constexpr int TOTAL = RO + RO_N;
    static constexpr size_t BYTES = size_t(TOTAL) * sizeof(double);
};

typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2_t lds2(const double* p) { return *reinterpret_cast<const double2_t*>(p); }
// Pins a loaded pair in its registers: the read that produces it has completed here, and (placed after the reads of later
// stages in program order) those stay in flight behind it.  See DESIGN.md section 5, "exposed LDS round trips".
__device__ __forceinline__ void pin2(double2_t& v) { asm volatile("" : "+v"(v)); }

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
    if (M::SO2) {   // full second-order builds: the factors of the wdot Hessian, stage knots only (a pass of its own: registers)
        for (int k = lane; k < N; k += kWave) {
            double x[M::NX], u[M::NU];
#pragma unroll
            for (int i = 0; i < M::NX; ++i) x[i] = xs[k * M::NX + i];
#pragma unroll
            for (int i = 0; i < M::NU; ++i) u[i] = us[k * M::NU + i];
            M::so2_knot(c, x, u, P + k * M::NP, rec + size_t(k) * M::NREC);
        }
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
// one-time (per kernel) constant tables in LDS: constant part of F~^T, diagonal tables, extra-row weights, block LUTs
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__device__ void sweep_tables(const DevConsts& c, double* s, int lane) {
    using L = Lds<M>;
    constexpr int NX = M::NX, NZ = M::NZ, NE = M::NE, NEV = M::NEV;
    for (int e = lane; e < L::TOTAL; e += kWave) s[e] = 0.0;
    wave_sync();
    for (int e = lane; e < M::NREC; e += kWave) s[L::REC + e] = 0.0;         // zero record -> constant part of F
    wave_sync();
    for (int e = lane; e < NZ * NX; e += kWave) {
        const int j = e / NX, i = e % NX;
        s[L::FT + j * L::NIP + i] = M::F_entry(c, s + L::REC, i, j);
    }
    for (int e = lane; e < NZ * NEV; e += kWave) {
        const int j = e / NEV, m = e % NEV;
        s[L::FT + j * L::NIP + NX + m] = M::E_const(c, m, j);
    }
    int* ki = reinterpret_cast<int*>(s + L::KI);
    for (int i = lane; i < NZ; i += kWave) {
        s[L::DS + i] = M::dg_state(c, i);
        s[L::DG + i] = M::dg_stage(c, i);
        ki[i] = M::dkind(i);
        ki[L::NZP + i] = M::dci(i);
    }
    for (int m = lane; m < NE; m += kWave) {
        s[L::LS + m] = M::lam_state(c, m);
        s[L::LG + m] = M::lam_stage(c, m);
        s[L::LAM + m] = M::lam_state(c, m) + M::lam_stage(c, m);
    }
    // lower-triangle block LUTs: t -> (ba << 8) | bc with ba >= bc
    for (int t = lane; t < L::NTRIQ; t += kWave) {
        int ba = 0;
        while ((ba + 1) * (ba + 2) / 2 <= t) ++ba;
        ki[2 * L::NZP + t] = (ba << 8) | (t - ba * (ba + 1) / 2);
    }
    for (int t = lane; t < L::NTRIV; t += kWave) {          // t -> (row a << 8) | column pair bp
        int a = 0, first = 0;
        while (first + a / 2 + 1 <= t) { first += a / 2 + 1; ++a; }
        ki[2 * L::NZP + L::NTRIQ + t] = (a << 8) | (t - first);
    }
    for (int e = lane; e < M::NSO2L / 2; e += kWave) M::so2_pair_code(e, ki[L::SO2L + 2 * e], ki[L::SO2L + 2 * e + 1]);
    wave_sync();
}

// -----------------------------------------------------------------------------------------------------------------
// backward Riccati sweep.  Returns false when a Quu is not positive definite (caller bumps mu, ddp.py:34-35).
// Requires sweep_tables() to have run in this kernel.  One knot per loop trip; the next knot's record / parameters /
// defect are prefetched into registers while the current knot is processed.
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__device__ bool backward_sweep(const DevConsts& c, int N, const double* __restrict__ P, const double* __restrict__ dft,
                               const double* __restrict__ rec, double* __restrict__ gains, double mu, double theta, double* s,
                               int lane, double& dV1, double& G1, double& G2, double& qu_inf, bool has_gap SDDP_T_ARG) {
    // has_gap = false: all defects are zero (every iteration after the first full step): v' = Vx, no Vxx d product
    using L = Lds<M>;
    constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NEV = M::NEV, NREC = M::NREC, NP = M::NP;
    constexpr int NXP = L::NXP, NIP = L::NIP, NZP = L::NZP, NUP = L::NUP, SQ = L::SQ;
    constexpr int NCOL = NU + 1 + NX;
    static_assert(NEV == NE || !M::CONST_ROWS_STATE_WEIGHTED, "constant rows with node-dependent weights must stay in the product");
    constexpr int RREC = (NREC + kWave - 1) / kWave;
    constexpr bool PADROW = (NXP > NX);      // odd NX: the pad row of the Vxx tile carries v' through the W product
    static_assert(NCOL <= kWave, "one lane per augmented column");
    static_assert(NX <= kWave && NP <= kWave, "one lane per state row / parameter");
    const int* ki = reinterpret_cast<const int*>(s + L::KI);
    dV1 = G1 = G2 = qu_inf = 0.0;
    bool ok = true;
    // the blocks this lane owns in the Q and Vxx phases never change: read the block LUT once per sweep, not once per knot
    // (one dependent LDS round trip less in each of the two phases)
    constexpr int PQ = (L::NTRIQ + kWave - 1) / kWave, PV = (L::NTRIV + kWave - 1) / kWave;
    int codeq[PQ], codev[PV];
#pragma unroll
    for (int q = 0; q < PQ; ++q) codeq[q] = (lane + q * kWave < L::NTRIQ) ? ki[2 * NZP + lane + q * kWave] : 0;
#pragma unroll
    for (int q = 0; q < PV; ++q) codev[q] = (lane + q * kWave < L::NTRIV) ? ki[2 * NZP + L::NTRIQ + lane + q * kWave] : 0;
    // constant extra rows (m >= NEV): their share sum_m lambda_m E[m][row] E[m][col] of the lane's four Q entries, once per sweep
    double qconst[PQ][4];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int a0 = 2 * (codeq[q] >> 8), c0 = 2 * (codeq[q] & 255);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double v = 0.0;
            if (NEV < NE) {
                const int row = a0 + (e >> 1), col = c0 + (e & 1);
                if (row < NZ && col < NZ)
                    for (int m = NEV; m < NE; ++m) v += s[L::LG + m] * M::E_const(c, m, row) * M::E_const(c, m, col);
            }
            qconst[q][e] = v;
        }
    }
    // diagonal blocks: what the diagonal term D of the lane's two diagonal entries needs is knot-invariant except for the knot's
    // parameters, so it is read once per sweep (zeros on the lanes that own an off-diagonal block: the per-knot code below is
    // then branch-free and its parameter reads are issued ahead of the tile product)
    int dkind[PQ][2], dcidx[PQ][2];
    double dsv[PQ][2], dgv[PQ][2];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int a0 = 2 * (codeq[q] >> 8), c0 = 2 * (codeq[q] & 255);
        const bool on = lane + q * kWave < L::NTRIQ && a0 == c0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool one = on && a0 + e < NZ;
            const int idx = one ? a0 + e : 0;
            dkind[q][e] = one ? ki[idx] : 0;
            dcidx[q][e] = one ? ki[NZP + idx] : 0;
            dsv[q][e] = one ? s[L::DS + idx] : 0.0;
            dgv[q][e] = one ? s[L::DG + idx] : 0.0;
        }
    }
    // ---- terminal node: Vx = lx_N, Vxx = lxx_N = diag(D_state) + Je^T Lambda_state Je  (ddp.py:216-226)
    {
        const double* rN = rec + size_t(N) * NREC;
        if (lane < NXP) s[L::VX + lane] = lane < NX ? rN[M::REC_G + lane] : 0.0;
        if (lane < NP) s[L::PK + lane] = P[N * NP + lane];
        wave_sync();
        for (int e = lane; e < NXP * NXP; e += kWave) {
            const int a = e / NXP, b = e % NXP;
            double v = 0.0;
            if (a < NX && b < NX) {
                for (int m = 0; m < NEV; ++m) v += s[L::LS + m] * s[L::FT + a * NIP + NX + m] * s[L::FT + b * NIP + NX + m];
                if (a == b) v += s[L::DS + a] + M::dparam(c, s + L::PK, ki[a], ki[NZP + a], 1.0, 0.0);
            }
            s[L::VXX + e] = v;
        }
        // extra-row part of WT = (V~ F~)^T for the nodes k >= 1: lambda_m Je[m][j].  Constant entries are written here once per
        // sweep, the per-knot variable ones by expand_var together with F~^T; node 0 (other weights) rescales them below
        for (int e = lane; e < NZ * NEV; e += kWave) {
            const int j = e / NEV, m = e % NEV;
            s[L::WT + j * NIP + NX + m] = s[L::LAM + m] * s[L::FT + j * NIP + NX + m];
        }
        wave_sync();
    }
    // ---- knot operands record | parameters | defect.  The last record word group and the parameters share one load: lanes past
    //      the record's tail take the parameters (fixed per-lane source / knot stride / LDS slot).  The defect is staged only
    //      while there is one (has_gap); otherwise its slot stays zero.
    constexpr int RTAIL = NREC - (RREC - 1) * kWave;                  // record words in the last group
    constexpr bool MERGE_P = RTAIL + NP <= kWave;
    const bool tail_rec = lane < RTAIL, tail_par = MERGE_P && lane >= RTAIL && lane < RTAIL + NP;
    const double* t_src = tail_rec ? rec + (RREC - 1) * kWave + lane : P + (lane - RTAIL);
    const int t_stride = tail_rec ? NREC : NP;
    const int t_dst = tail_rec ? L::REC + (RREC - 1) * kWave + lane : L::PK + (lane - RTAIL);
    const bool t_on = tail_rec || tail_par;
    double r_rec[RREC], r_p = 0.0, r_d = 0.0;
    auto fetch = [&](int k) {
        const double* rk = rec + size_t(k) * NREC;
#pragma unroll
        for (int t = 0; t + 1 < RREC; ++t) r_rec[t] = rk[lane + t * kWave];
        if (t_on) r_rec[RREC - 1] = t_src[size_t(k) * t_stride];
        if (!MERGE_P && lane < NP) r_p = P[k * NP + lane];
        if (has_gap && lane < NX) r_d = dft[k * NX + lane];
    };
    if (lane < NXP) s[L::DK + lane] = 0.0;
    // gains of knot kk, LDS (KF | K^T, as the solve phase left them) -> HBM/L2: word e of the knot's record = kff[e] for e < NU, else
    // K[i][c] = KT[c][i] with (i, c) = divmod(e - NU, NX); whole-wave contiguous stores
    constexpr int NGW = NU * (NX + 1), RGW = (NGW + kWave - 1) / kWave;
    int g_src[RGW];
#pragma unroll
    for (int t = 0; t < RGW; ++t) {
        const int e = lane + t * kWave, ec = e < NGW ? e : 0;
        g_src[t] = ec < NU ? L::KF + ec : L::KT + ((ec - NU) % NX) * NUP + (ec - NU) / NX;
    }
    auto store_gains = [&](int kk) {
        double* gk = gains + size_t(kk) * NGW;
        double v[RGW];
#pragma unroll
        for (int t = 0; t < RGW; ++t) v[t] = s[g_src[t]];
#pragma unroll
        for (int t = 0; t < RGW; ++t)
            if (lane + t * kWave < NGW) gk[lane + t * kWave] = v[t];
    };
    // Experiment (-DSDDP_LDS_DMA, profiles/r05/experiments): the knot's record | parameters | defect go HBM/L2 -> LDS by LDS-DMA
    // (global_load_lds: lane l's bytes land at M0 + size * l, no VGPR round trip, no ds_write), issued for knot k - 1 where knot
    // k's copies are dead (behind the Q phase) and waited for at the top of knot k - 1.  Record: NREC * 8 bytes = a whole number of
    // 16-byte pieces, 16-byte aligned; parameters and defect rows are only 8-byte aligned: 4-byte pieces.
#ifdef SDDP_LDS_DMA
    constexpr bool kDma = (NREC * 8) % 16 == 0 && NREC * 8 / 16 <= kWave && NP * 2 <= kWave && NX * 2 <= kWave;
#else
    constexpr bool kDma = false;
#endif
    auto dma_fetch = [&](int k) {
        typedef const __attribute__((address_space(1))) void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        const char* rk = reinterpret_cast<const char*>(rec + size_t(k) * NREC);
        const char* pk = reinterpret_cast<const char*>(P + size_t(k) * NP);
        const char* dk = reinterpret_cast<const char*>(dft + size_t(k) * NX);
        if (lane < NREC * 8 / 16) __builtin_amdgcn_global_load_lds((gptr)(rk + 16 * lane), (lptr)(s + L::REC), 16, 0, 0);
        if (lane < NP * 2) __builtin_amdgcn_global_load_lds((gptr)(pk + 4 * lane), (lptr)(s + L::PK), 4, 0, 0);
        if (has_gap && lane < NX * 2) __builtin_amdgcn_global_load_lds((gptr)(dk + 4 * lane), (lptr)(s + L::DK), 4, 0, 0);
    };
    drain_vmem();
    if constexpr (kDma) dma_fetch(N - 1); else fetch(N - 1);
    for (int k = N - 1; k >= 0; --k) {
        if constexpr (kDma) {
            drain_vmem();                                    // this knot's operands have landed (issued a knot ago)
            if (k < N - 1) store_gains(k + 1);
        } else {
            // ---- stage this knot from the prefetch registers; start the next knot's loads; then the previous knot's gains go out
#pragma unroll
            for (int t = 0; t + 1 < RREC; ++t) s[L::REC + lane + t * kWave] = r_rec[t];
            if (t_on) s[t_dst] = r_rec[RREC - 1];
            if (!MERGE_P && lane < NP) s[L::PK + lane] = r_p;
            if (has_gap && lane < NX) s[L::DK + lane] = r_d;
            if (k < N - 1) store_gains(k + 1);
            if (k > 0) fetch(k - 1);
        }
        wave_sync();
        SDDP_TICK(1)
        const double state = k >= 1 ? 1.0 : 0.0;
        // ---- expand: variable entries of F~^T; v' = Vx + Vxx d ; gap terms
        M::expand_var(c, s + L::REC, s + L::FT, NIP, lane, kWave, s + L::WT, s + L::LAM);
        if (lane < NX) {
            double acc = 0.0;
            if (has_gap) {
                double2_t vv[NXP / 2], dd[NXP / 2];
#pragma unroll
                for (int m = 0; m < NXP; m += 2) { vv[m / 2] = lds2(s + L::VXX + lane * NXP + m); dd[m / 2] = lds2(s + L::DK + m); }
#pragma unroll
                for (int m = 0; m < NXP / 2; ++m) { pin2(vv[m]); pin2(dd[m]); }
#pragma unroll
                for (int m = 0; m < NXP / 2; ++m) acc = fma(vv[m].y, dd[m].y, fma(vv[m].x, dd[m].x, acc));
            }
            const double d = s[L::DK + lane], vx = s[L::VX + lane];
            s[L::VP + lane] = vx + acc;
            if (PADROW) s[L::VXX + NX * NXP + lane] = vx + acc;   // pad row of Vxx := v': the W product then yields F^T v'
            G1 = fma(d, vx, G1);                 // per-lane partial sums, reduced once after the sweep
            G2 = fma(0.5 * d, acc, G2);
        }
        wave_sync();
        SDDP_TICK(2)
        // ---- WT = (V~ F~)^T : 2 (l) x JB (j) register blocks, inner product over the NX dynamics rows.  JB = 3 when that still
        //      fits one pass of the wave (more lanes, a shorter chain per lane), else 4.  A block may reach one row past the
        //      F~^T tile (the row after it is the first row of WT: finite, never stored).
        constexpr int JB = ((NXP / 2) * ((NZP + 2) / 3) <= kWave) ? 3 : 4;
        constexpr int NJB = (NZP + JB - 1) / JB;
        for (int blk = lane; blk < (NXP / 2) * NJB; blk += kWave) {
            const int l0 = 2 * (blk % (NXP / 2)), j0 = JB * (blk / (NXP / 2));
            double a0[JB], a1[JB];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) a0[jj] = a1[jj] = 0.0;
            constexpr int NSW = NXP / 2, PFW = 2;       // operands two steps ahead, as in the Q phase below
            double2_t v0[NSW], v1[NSW], ff[NSW][JB];
            auto ldw = [&](int st) {
                v0[st] = lds2(s + L::VXX + l0 * NXP + 2 * st); v1[st] = lds2(s + L::VXX + (l0 + 1) * NXP + 2 * st);
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) ff[st][jj] = lds2(s + L::FT + (j0 + jj) * NIP + 2 * st);
            };
#pragma unroll
            for (int st = 0; st < PFW && st < NSW; ++st) ldw(st);
#pragma unroll
            for (int st = 0; st < NSW; ++st) {
                if (st + PFW < NSW) ldw(st + PFW);
                pin2(v0[st]); pin2(v1[st]);
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) pin2(ff[st][jj]);
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
                    a0[jj] = fma(v0[st].y, ff[st][jj].y, fma(v0[st].x, ff[st][jj].x, a0[jj]));
                    a1[jj] = fma(v1[st].y, ff[st][jj].y, fma(v1[st].x, ff[st][jj].x, a1[jj]));
                }
            }
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                if (j0 + jj >= NZP) continue;
                if (l0 + 1 < NX) {
                    double2_t w;
                    w.x = a0[jj];
                    w.y = a1[jj];
                    *reinterpret_cast<double2_t*>(s + L::WT + (j0 + jj) * NIP + l0) = w;
                } else {
                    s[L::WT + (j0 + jj) * NIP + l0] = a0[jj];        // odd NX: slot NX belongs to the first extra row;
                    if (j0 + jj < NZ) s[L::QV + j0 + jj] = s[L::REC + M::REC_G + j0 + jj] + a1[jj];   // row NX of Vxx holds v'
                }
            }
        }
        if (k == 0) {   // node 0 carries no state residuals: rescale the extra rows with the stage-only weights
            wave_sync();
            for (int e = lane; e < NZ * NEV; e += kWave) {
                const int j = e / NEV, m = e % NEV;
                s[L::WT + j * NIP + NX + m] = s[L::LG + m] * s[L::FT + j * NIP + NX + m];
            }
        }
        wave_sync();
        SDDP_TICK(3)
        // ---- Q = diag(D) + F~^T (V~ F~): lower-triangle 2x2 blocks, mirrored ; q = g + F^T v'
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (lane + q * kWave >= L::NTRIQ) break;
            const int code = codeq[q];
            const int a0 = 2 * (code >> 8), c0 = 2 * (code & 255);
            double q00 = 0, q01 = 0, q10 = 0, q11 = 0;
            // operands two steps ahead of the step being multiplied (three register stages, pinned stage by stage): the reads of
            // the whole product stay in flight behind the FMAs instead of a round trip every other step
            constexpr int NS = NIP / 2, PF = 2;
            double2_t fa[NS], fb[NS], wa[NS], wb[NS];
            auto ldq = [&](int st) {
                fa[st] = lds2(s + L::FT + a0 * NIP + 2 * st); fb[st] = lds2(s + L::FT + (a0 + 1) * NIP + 2 * st);
                wa[st] = lds2(s + L::WT + c0 * NIP + 2 * st); wb[st] = lds2(s + L::WT + (c0 + 1) * NIP + 2 * st);
            };
#pragma unroll
            for (int st = 0; st < PF && st < NS; ++st) ldq(st);
            // diagonal term of this knot (zero on off-diagonal blocks): its parameter reads queue behind the first product operands
            const double dd0 = state * dsv[q][0] + dgv[q][0] + M::dparam(c, s + L::PK, dkind[q][0], dcidx[q][0], state, 1.0);
            const double dd1 = state * dsv[q][1] + dgv[q][1] + M::dparam(c, s + L::PK, dkind[q][1], dcidx[q][1], state, 1.0);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                if (st + PF < NS) ldq(st + PF);
                pin2(fa[st]); pin2(fb[st]); pin2(wa[st]); pin2(wb[st]);
                q00 = fma(fa[st].y, wa[st].y, fma(fa[st].x, wa[st].x, q00));
                q01 = fma(fa[st].y, wb[st].y, fma(fa[st].x, wb[st].x, q01));
                q10 = fma(fb[st].y, wa[st].y, fma(fb[st].x, wa[st].x, q10));
                q11 = fma(fb[st].y, wb[st].y, fma(fb[st].x, wb[st].x, q11));
            }
            if (NEV < NE) { q00 += qconst[q][0]; q01 += qconst[q][1]; q10 += qconst[q][2]; q11 += qconst[q][3]; }
            {   // diagonal block: add D, keep it exactly symmetric (selects, no branch)
                q00 += dd0;
                q11 += dd1;
                const double off = 0.5 * (q01 + q10);
                q01 = a0 == c0 ? off : q01;
                q10 = a0 == c0 ? off : q10;
            }
            double2_t r0, r1;
            r0.x = q00; r0.y = q01; r1.x = q10; r1.y = q11;
            *reinterpret_cast<double2_t*>(s + L::Q + a0 * SQ + c0) = r0;
            *reinterpret_cast<double2_t*>(s + L::Q + (a0 + 1) * SQ + c0) = r1;
            if (a0 != c0) {
                double2_t m0, m1;
                m0.x = q00; m0.y = q10; m1.x = q01; m1.y = q11;
                *reinterpret_cast<double2_t*>(s + L::Q + c0 * SQ + a0) = m0;
                *reinterpret_cast<double2_t*>(s + L::Q + (c0 + 1) * SQ + a0) = m1;
            }
        }
        if (!PADROW) {
            for (int j = lane; j < NZ; j += kWave) {
                double acc = s[L::REC + M::REC_G + j];
#pragma unroll
                for (int m = 0; m < NXP; m += 2) {
                    const double2_t f = lds2(s + L::FT + j * NIP + m), v = lds2(s + L::VP + m);
                    acc = fma(f.y, (m + 1 < NX) ? v.y : 0.0, fma(f.x, (m < NX) ? v.x : 0.0, acc));
                }
                s[L::QV + j] = acc;
            }
        }
        wave_sync();
        if (theta != 0.0) {   // exact second-order torque term (wave-uniform switch)
            if (M::NSO2T) {   // full second-order builds: the per-knot factors of the contraction first
                M::so2_prepare(c, s + L::REC, s + L::VP, s + L::REC + L::SO2T, lane, kWave);
                wave_sync();
            }
            M::add_second_order(c, s + L::REC, s + L::VP, QFull{s + L::Q, SQ}, theta, lane, kWave, s + L::REC + L::SO2T, ki + L::SO2L);
            wave_sync();
        }
        if (M::BAR) {         // friction-cone barrier builds: its Hessian blocks on the force-force diagonal of Q
            M::add_barrier(s + L::REC, QFull{s + L::Q, SQ}, lane, kWave, M::SO2 ? theta : 0.0);
            wave_sync();
        }
        SDDP_TICK(4)
        if constexpr (kDma) { if (k > 0) dma_fetch(k - 1); }   // REC / PK / DK of this knot are dead from here on
        // ---- [k K] = -Quu^-1 [Qu Qux]: Gauss-Jordan, lane j owns column j of [Quu+mu I | Qu | Qux]
        double a[NU];
        const double qx = s[L::QV + (lane > NU && lane < NCOL ? lane - NU - 1 : 0)];     // Qx of this lane's state column, for the Vx update
        {
            // column of Q this lane reads (any valid one for lane NU, which takes q instead; lanes >= NCOL are zeroed)
            const int qcol = lane < NU ? NX + lane : (lane > NU && lane < NCOL ? lane - NU - 1 : 0);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const double qv = s[L::QV + NX + i];
                double v = s[L::Q + (NX + i) * SQ + qcol];
                v = lane == NU ? qv : v;
                v += (i == lane) ? mu : 0.0;
                a[i] = lane < NCOL ? v : 0.0;
            }
        }
        double qu_abs = 0.0, qu_save[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { qu_save[i] = a[i]; qu_abs = fmax(qu_abs, fabs(a[i])); }
        qu_inf = fmax(qu_inf, readlane_d(qu_abs, NU));
        // pivot column broadcast with v_readlane (through SGPRs): no LDS hand-off, no barrier inside the elimination
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
        // a = Quu^-1 * column ; publish kff and K^T (negated) in LDS.  The copy to HBM/L2 (kff (NU) then K (NU x NX) row-major) is
        // made from there one knot later, behind the next staging wait (store_gains): a store in flight makes every wait for a
        // prefetched load a full vmcnt(0) (gfx9 counts loads and stores in one counter and returns them out of order with respect
        // to each other), so stores issued at the end of a knot were waited for at the start of the next
        if (lane >= NU && lane < NCOL) {
            double* dst = lane == NU ? s + L::KF : s + L::KT + (lane - NU - 1) * NUP;            // KF follows KT: one region
#pragma unroll
            for (int i = 0; i < NU; ++i) dst[i] = -a[i];
        }
        // kff broadcast from lane NU; then ONE dot product per lane with the column it loaded gives both
        //   lane NU:        kff . Qu           -> dV1 (dV2 = 1/2 kff^T Quu kff = -1/2 dV1 exactly, not accumulated separately)
        //   lane NU+1+c:    (Qux^T kff)[c]     -> Vx[c] = Qx[c] + Qux^T kff
        {
            double dot = 0.0;
#pragma unroll
            for (int i = 0; i < NU; ++i) dot = fma(readlane_d(-a[i], NU), qu_save[i], dot);
            dV1 += readlane_d(dot, NU);
            if (lane > NU && lane < NCOL) s[L::VX + lane - NU - 1] = qx + dot;
        }
        wave_sync();
        SDDP_TICK(5)
        if (!ok) return false;
        // ---- Vxx = Qxx + Qux^T K.  Qux^T K = -Qux^T (Quu + mu I)^-1 Qux is symmetric, so only the lower triangle is formed (one
        //      product per element, 1 x 2 blocks: 49 lanes with a 12-FMA chain each) and mirrored
#pragma unroll
        for (int q = 0; q < PV; ++q) {
            if (lane + q * kWave >= L::NTRIV) break;
            const int code = codev[q];
            const int a0 = code >> 8, c0 = 2 * (code & 255);        // row a0, columns c0, c0 + 1 (c0 <= a0)
            const bool two = c0 + 1 <= a0;                          // the second column is in the lower triangle too
            const int c1 = two ? c0 + 1 : c0;
            double v0 = s[L::Q + a0 * SQ + c0], v1 = s[L::Q + a0 * SQ + c1];
            double qa[NU], k0[NU], k1[NU];           // all operands in flight before the first FMA
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                qa[i] = s[L::Q + a0 * SQ + NX + i];
                k0[i] = s[L::KT + c0 * NUP + i];
                k1[i] = s[L::KT + c1 * NUP + i];
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) { asm volatile("" : "+v"(qa[i])); asm volatile("" : "+v"(k0[i])); asm volatile("" : "+v"(k1[i])); }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                v0 = fma(qa[i], k0[i], v0);
                v1 = fma(qa[i], k1[i], v1);
            }
            s[L::VXX + a0 * NXP + c0] = v0;
            if (c0 != a0) s[L::VXX + c0 * NXP + a0] = v0;
            if (two) {
                s[L::VXX + a0 * NXP + c1] = v1;
                if (c1 != a0) s[L::VXX + c1 * NXP + a0] = v1;
            }
        }
        wave_sync();
        SDDP_TICK(6)
    }
    store_gains(0);
    G1 = wave_sum(G1);
    G2 = wave_sum(G2);
    return ok;
}

// -----------------------------------------------------------------------------------------------------------------
// forward pass: one lane per step length.  Every lane rolls the whole horizon with its own alpha; the lane
// `store_lane` also writes its trajectory to xn/un.  OPEN_LOOP: plain rollout of us (single-shooting start).
// The knot's wave-uniform operands (x_k, u_k, p_k, gains_k, d_k) are fetched by ONE coalesced wave-wide load each,
// staged in LDS and read back as broadcasts; the next knot is prefetched into registers meanwhile.
// -----------------------------------------------------------------------------------------------------------------
template <class M, bool OPEN_LOOP>
__device__ double rollout(const DevConsts& c, int N, const double* __restrict__ x0, const double* __restrict__ P,
                          const double* __restrict__ xs, const double* __restrict__ us, const double* __restrict__ dft,
                          const double* __restrict__ gains, double* __restrict__ xn, double* __restrict__ un,
                          double alpha, int store_lane, int lane, double* s, bool has_gap = true) {
    // has_gap = false: all defects are zero, the (1 - alpha) d correction is skipped.
    // store_lane >= 0: that lane writes its trajectory to xn / un.  store_lane < 0: lanes 0 .. kSlots-1 each write theirs to
    // slot `lane` of xn / un (slot strides (N+1) NX and N NU): same instruction count, no second pass to fetch the winner.
    using L = Lds<M>;
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NG = NU * (NX + 1), RG = (NG + kWave - 1) / kWave;
    double* ro = s + L::RO;
    const bool storing = store_lane >= 0 ? lane == store_lane : lane < kSlots;
    if (store_lane < 0 && storing) { xn += size_t(lane) * (N + 1) * NX; un += size_t(lane) * N * NU; }
    double x[NX];
    if (lane < NX) ro[L::RO_X + lane] = x0[lane];
    wave_sync();
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = ro[L::RO_X + i];
    wave_sync();
    // knot operands x_k | u_k | p_k | d_k: when they fit one wave (2 NX + NU + NP <= 64) every lane owns ONE word of the knot:
    // its source pointer / knot stride / LDS slot are fixed for the whole pass, so a knot costs one load and one LDS store
    // (plus the gain rows) instead of four masked pairs
    constexpr bool MERGED = !OPEN_LOOP && (2 * NX + NU + NP <= kWave);
    const double* m_src = nullptr;
    int m_stride = 0, m_dst = 0;
    bool m_on = false;
    if (MERGED) {
        int l = lane;
        if (l < NX) { m_src = xs + l; m_stride = NX; m_dst = L::RO_X + l; m_on = true; }
        else if ((l -= NX) < NU) { m_src = us + l; m_stride = NU; m_dst = L::RO_U + l; m_on = true; }
        else if ((l -= NU) < NP) { m_src = P + l; m_stride = NP; m_dst = L::RO_P + l; m_on = true; }
        else if ((l -= NP) < NX) { m_src = dft + l; m_stride = NX; m_dst = L::RO_D + l; m_on = true; }
    }
    double r_x = 0, r_u = 0, r_p = 0, r_d = 0, r_m = 0, r_g[RG];
    auto fetch = [&](int k) {
        // the gain rows first: their registers are zero-filled for the lanes past the row's end, and that write waits for whatever
        // the compiler still counts as pending on those registers -- in front of the knot's first load it waits for nothing
        if (!OPEN_LOOP) {
            const double* gk = gains + size_t(k) * NG;
#pragma unroll
            for (int t = 0; t < RG; ++t) r_g[t] = (lane + t * kWave < NG) ? gk[lane + t * kWave] : 0.0;
        }
        if (MERGED) {
            if (m_on) r_m = m_src[size_t(k) * m_stride];
        } else {
            if (lane < NU) r_u = us[k * NU + lane];
            if (lane < NP) r_p = P[k * NP + lane];
            if (!OPEN_LOOP && lane < NX) { r_x = xs[k * NX + lane]; r_d = dft[k * NX + lane]; }
        }
    };
    drain_vmem();
    fetch(0);
    double J = 0.0;
    const double oma = 1.0 - alpha;
    for (int k = 0; k < N; ++k) {
        if (MERGED) {
            if (m_on) ro[m_dst] = r_m;
        } else {
            if (lane < NU) ro[L::RO_U + lane] = r_u;
            if (lane < NP) ro[L::RO_P + lane] = r_p;
            if (!OPEN_LOOP && lane < NX) { ro[L::RO_X + lane] = r_x; ro[L::RO_D + lane] = r_d; }
        }
        if (!OPEN_LOOP) {
#pragma unroll
            for (int t = 0; t < RG; ++t)
                if (lane + t * kWave < NG) ro[L::RO_G + lane + t * kWave] = r_g[t];
        }
        if (k + 1 < N) fetch(k + 1);
        wave_sync();
        double u[NU];
        if (OPEN_LOOP) {
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = ro[L::RO_U + i];
        } else {
            // u = u_k + alpha kff + K (x - x_k).  The gain rows are fetched one row ahead of the row being used (two register rows,
            // pinned row by row): left to the compiler every LDS read of the NU NX-long chain is issued just in time, one exposed
            // round trip per two FMAs
            double dx[NX], ub[NU], kf[NU], g[2][NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) dx[j] = x[j] - ro[L::RO_X + j];
#pragma unroll
            for (int i = 0; i < NU; ++i) { ub[i] = ro[L::RO_U + i]; kf[i] = ro[L::RO_G + i]; }
#pragma unroll
            for (int j = 0; j < NX; ++j) g[0][j] = ro[L::RO_G + NU + j];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (i + 1 < NU) {
#pragma unroll
                    for (int j = 0; j < NX; ++j) g[(i + 1) & 1][j] = ro[L::RO_G + NU + (i + 1) * NX + j];
                }
#pragma unroll
                for (int j = 0; j < NX; ++j) asm volatile("" : "+v"(g[i & 1][j]));
                double acc = ub[i] + alpha * kf[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += g[i & 1][j] * dx[j];
                u[i] = acc;
            }
        }
        if (storing) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[k * NX + i] = x[i];
#pragma unroll
            for (int i = 0; i < NU; ++i) un[k * NU + i] = u[i];
        }
        J += M::step(c, x, u, ro + L::RO_P, k, x);               // in place: every component is read before it is written
        if (!OPEN_LOOP && has_gap) {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] -= oma * ro[L::RO_D + i];
        }
        wave_sync();
    }
    if (lane < NP) ro[L::RO_P + lane] = P[N * NP + lane];
    wave_sync();
    J += M::term_cost(c, x, ro + L::RO_P);
    if (storing) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[N * NX + i] = x[i];
    }
    wave_sync();
    return J;
}

// -----------------------------------------------------------------------------------------------------------------
// fused persistent solve: one wavefront per MPC instance, all iterations in one launch (replaces ddp.py:101)
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void solve_instance(const SolveArgs& A, double* s, const int b, const int slot) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int lane = threadIdx.x;
    const int N = A.N;
    const sddp_options& o = A.o;
    const double* x0 = A.x0 + size_t(b) * NX;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    // candidate sets: set w of this instance = kSlots trajectories; the pass of an iteration writes set `wr`, the current
    // iterate lives in the other set (or in A.xs / A.us before the first accepted step)
    const size_t XS = size_t(N + 1) * NX, US = size_t(N) * NU;
    double* xc = A.xc + size_t(slot) * 2 * kSlots * XS;
    double* uc = A.uc + size_t(slot) * 2 * kSlots * US;
    int wr = 0;
    double* dft = A.dft + size_t(slot) * N * NX;
    double* gains = A.gains + size_t(slot) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(slot) * (N + 1) * NREC;

    double J = 0.0, gap = 0.0;
    SDDP_T_DECL
    sweep_tables<M>(A.c, s, lane);
    // ---- starting point
    if (o.initial_rollout) {
        J = rollout<M, true>(A.c, N, x0, P, xs, us, dft, gains, xc, uc, 0.0, 0, lane, s);
        wave_sync();
        for (int e = lane; e < (N + 1) * NX; e += kWave) xs[e] = xc[e];
        for (int e = lane; e < N * NX; e += kWave) dft[e] = 0.0;
        wave_sync();
    } else {
        if (lane < NX) xs[lane] = x0[lane];
        wave_sync();
        phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
        wave_sync();
    }
    double mu = o.mu0, rho = 0.0, alpha = 0.0, expected = 0.0, theta = 0.0;
    int iters = 0, converged = 0, status = 1, rollouts = 0, win = 0;
    if (!(fabs(J) < 1e300)) { status = 3; }
    else
        while (iters < o.max_iters) {
            SDDP_TICK(9)
            phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
            wave_sync();
            SDDP_TICK(0)
            double dV1, G1, G2, qu_inf, a_win = 0.0, J_win = 0.0;
            bool ok = true, stop = false, accepted = false;
            do {   // at most twice: a failed sweep / line search with the second-order term is redone without it
                while (true) {
                    ok = backward_sweep<M>(A.c, N, P, dft, rec, gains, mu, theta, s, lane, dV1, G1, G2, qu_inf, gap > 0.0 SDDP_T_PASS);
                    if (ok) break;
                    if (theta != 0.0) { theta = 0.0; continue; }
                    mu = fmax(mu, 0.0) * 10.0 + o.mu_min;
                    if (mu > o.mu_max) break;
                }
                if (!ok) { status = 2; stop = true; break; }
                const double dV2 = -0.5 * dV1;
                expected = -(dV1 + dV2);
                if (expected < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; stop = true; break; }
                const double A1 = dV1 + G1, B2 = dV2 + G2;
                if (gap > 0.0) rho = fmax(rho, 2.0 * fmax(fmax(A1, A1 + B2), 0.0) / gap);
                const double slack = 1e-13 * (fabs(J) + rho * gap);
                // ---- line search: lane j tries alpha_0 * factor^j (the ladder of ddp.py:20-28 in one pass)
                accepted = false;
                double a_base = o.alpha_0;
                while (a_base >= o.alpha_converge_threshold) {
                    double a = a_base;
                    for (int j = 0; j < lane; ++j) a *= o.line_search_decrease_factor;
                    const bool valid = a >= o.alpha_converge_threshold;
                    SDDP_TICK(9)
                    double* xw = xc + size_t(wr) * kSlots * XS;
                    double* uw = uc + size_t(wr) * kSlots * US;
                    double Jl = rollout<M, false>(A.c, N, x0, P, xs, us, dft, gains, xw, uw, a, -1, lane, s, gap > 0.0);
                    SDDP_TICK(8)
                    ++rollouts;
                    const double pred = a * A1 + a * a * B2 - a * rho * gap;
                    const double dphi = (Jl + rho * (1.0 - a) * gap) - (J + rho * gap);
                    const bool good = valid && (fabs(Jl) < 1e300) && (dphi <= o.beta * pred + slack);
                    const unsigned long long mask = __ballot(good);
                    if (mask) {
                        win = __ffsll((long long)mask) - 1;
                        a_win = __shfl(a, win, kWave);
                        J_win = __shfl(Jl, win, kWave);
                        if (win >= kSlots) {   // the winner is not one of the kept candidates: roll it again into slot 0
                            wave_sync();
                            rollout<M, false>(A.c, N, x0, P, xs, us, dft, gains, xw, uw, a, win, lane, s, gap > 0.0);
                            ++rollouts;
                            win = 0;
                        }
                        accepted = true;
                        break;
                    }
                    a_base = __shfl(a, kWave - 1, kWave) * o.line_search_decrease_factor;
                }
                if (accepted) break;
                if (theta != 0.0) { theta = 0.0; continue; }                       // redo with the plain Gauss-Newton step
                // alpha fell below alpha_converge_threshold: stop.  An optimum only with closed gaps and (next to) no predicted
                // decrease either; otherwise the line search has stalled (status 4, not converged)
                alpha = 0.0; stop = true;
                converged = (gap <= o.gap_tol && expected <= o.cost_reduction_ths * fmax(1.0, fabs(J))) ? 1 : 0;
                status = converged ? 0 : 4;
                break;
            } while (true);
            if (stop) break;
            alpha = a_win;
            theta = (o.second_order && alpha == o.alpha_0) ? 1.0 : 0.0;
            const double dJ = J - J_win;
            J = J_win;
            wave_sync();
            xs = xc + (size_t(wr) * kSlots + win) * XS;          // the accepted candidate becomes the iterate; next pass writes
            us = uc + (size_t(wr) * kSlots + win) * US;          // the other set
            wr ^= 1;
            const double oma = 1.0 - alpha;
            if (gap > 0.0) {
                for (int e = lane; e < N * NX; e += kWave) dft[e] *= oma;     // a full step writes exact zeros
            }
            gap *= oma;
            ++iters;
            if (mu > o.mu0) mu = fmax(o.mu0, mu * 0.1);
            wave_sync();
            if (fabs(dJ) < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; break; }
        }
    // ---- results live in A.xs/A.us: copy back if the iterate ended in the candidate buffers
    wave_sync();
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
        st.cost = J; st.alpha = alpha; st.gap = gap; st.mu = mu; st.expected = expected; st.rho = rho;
        st.iters = iters; st.converged = converged; st.status = status; st.rollouts = rollouts;
        A.stats[b] = st;
        A.hist[b] = iters;
    }
    wave_sync();
}

// The launch as a work queue: one resident wavefront (slot) per workgroup, each solving instances until the queue is empty.
// No workgroup ever waits for another one, so any grid size terminates; a slot's work buffers are private to its wavefront
// (its own loads and stores are seen in program order), the per-instance inputs were written before the launch.
// Every instance starts from freshly built LDS tables, so a result does not depend on which slot solved it or on what that
// slot solved before: bit-identical to one launch per instance.
template <class M>
__device__ __forceinline__ void solve_queue(const SolveArgs& A, double* s) {
    const int slot = blockIdx.x;
    const bool queued = A.qhead != nullptr;
    int i = slot;                                      // no queue: workgroup w solves instance first + w
#ifndef SDDP_NO_SLOT_CLOCK
    if (threadIdx.x == 0) A.slot_clock(slot)[0] = wall_clock64();
#endif
    if (queued) {
        if (threadIdx.x == 0) i = atomicAdd(A.qhead, 1);
        i = __builtin_amdgcn_readfirstlane(i);
    }
    while (i < A.count) {                              // every wavefront reaches the exit: the head only grows
        const int b = (queued && A.order) ? A.order[i] : A.first + i;
        solve_instance<M>(A, s, b, slot);              // one call site: the body is compiled once
        if (!queued) break;
        if (threadIdx.x == 0) i = atomicAdd(A.qhead, 1);
        i = __builtin_amdgcn_readfirstlane(i);
    }
#ifndef SDDP_NO_SLOT_CLOCK
    if (threadIdx.x == 0) A.slot_clock(slot)[1] = wall_clock64();
#endif
}

// two builds of the same body: the register allocation is the only difference (sddp_options.waves_per_simd)
template <class M>
__global__ __launch_bounds__(kWave) void solve_kernel(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    solve_queue<M>(A, s);
}
template <class M>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2))) void solve_kernel_w2(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    solve_queue<M>(A, s);
}

// Queue order for a COLD queue (sddp_options.queue_order = 2): the key of an instance is the total cost of its warm start
// (x_0 := x0 as the solve does; multiple-shooting cost of the given xs / us), evaluated here by one wavefront per instance, one
// lane per knot -- the same model code and the same sum the solve itself starts from.  The keys are then sorted in descending
// order (sddp_sort.hip): the instances farthest from their optimum start first.  Nothing here depends on an earlier solve.
template <class M>
__global__ __launch_bounds__(kWave) void queue_cost_key_kernel(DevConsts c, int N, int first, int count, const double* __restrict__ x0,
                                                               const double* __restrict__ P, const double* __restrict__ xs,
                                                               const double* __restrict__ us, double* __restrict__ key,
                                                               int* __restrict__ idx) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= count) return;
    const int b = first + i;
    const double* xb = xs + size_t(b) * (N + 1) * NX;
    const double* ub = us + size_t(b) * N * NU;
    const double* Pb = P + size_t(b) * (N + 1) * NP;
    double Jl = 0.0;
    for (int k = lane; k <= N; k += kWave) {
        double x[NX];
        const double* xk = k == 0 ? x0 + size_t(b) * NX : xb + k * NX;
#pragma unroll
        for (int j = 0; j < NX; ++j) x[j] = xk[j];
        if (k < N) {
            double u[NU], xn[NX];
#pragma unroll
            for (int j = 0; j < NU; ++j) u[j] = ub[k * NU + j];
            Jl += M::step(c, x, u, Pb + k * NP, k, xn);
        } else {
            Jl += M::term_cost(c, x, Pb + k * NP);
        }
    }
    const double J = wave_sum(Jl);
    if (lane == 0) {
        key[i] = (J == J) ? J : __builtin_huge_val();      // a non-finite start sorts first (it ends at once with status 3)
        idx[i] = b;
    }
}

// -----------------------------------------------------------------------------------------------------------------
// single-phase kernels for the parity tests (same device code as the fused kernel)
// -----------------------------------------------------------------------------------------------------------------
// one model step per instance, x+ = f(x, u, p_k) (ddp.py:228-230): the closed-loop simulator step of the examples
// (dsrbd_example.py:158-159) through the same device model code as the solver.  One thread per instance.
// -----------------------------------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(kWave) void model_step_kernel(DevConsts c, int B, int k, const double* __restrict__ x,
                                                           const double* __restrict__ u, const double* __restrict__ p,
                                                           double* __restrict__ xn) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int b = blockIdx.x * kWave + threadIdx.x;
    if (b >= B) return;
    double xv[NX], uv[NU], pv[NP], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = x[size_t(b) * NX + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) uv[i] = u[size_t(b) * NU + i];
#pragma unroll
    for (int i = 0; i < NP; ++i) pv[i] = p[size_t(b) * NP + i];
    (void)M::step(c, xv, uv, pv, k, xo);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[size_t(b) * NX + i] = xo[i];
}

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
    wave_sync();
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
    sweep_tables<M>(A.c, s, lane);
    phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
    phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
    wave_sync();
    double dV1, G1, G2, qu_inf;
    SDDP_T_DECL
    const bool ok = backward_sweep<M>(A.c, N, P, dft, rec, gains, A.mu, A.alpha, s, lane, dV1, G1, G2, qu_inf, true SDDP_T_PASS);
    if (lane == 0) {
        double* sc = A.scal + size_t(b) * kScal;
        sc[0] = dV1; sc[1] = -0.5 * dV1; sc[2] = G1; sc[3] = G2; sc[4] = ok ? 1.0 : 0.0; sc[5] = A.mu; sc[6] = qu_inf; sc[7] = J;
    }
}

template <class M>
__global__ __launch_bounds__(kWave) void forward_kernel(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N;
    const double J = rollout<M, false>(A.c, N, A.x0 + size_t(b) * NX, A.P + size_t(b) * (N + 1) * NP,
                                       A.xs + size_t(b) * (N + 1) * NX, A.us + size_t(b) * N * NU,
                                       A.dft + size_t(b) * N * NX, A.gains + size_t(b) * N * (NU * (NX + 1)),
                                       A.xn + size_t(b) * (N + 1) * NX, A.un + size_t(b) * N * NU, A.alpha, 0, lane, s);
    if (lane == 0) A.scal[size_t(b) * kScal] = J;
}

}  // namespace sddp
