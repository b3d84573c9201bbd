// sddp_kernels_mw.hpp -- the same DDP iteration as sddp_kernels.hpp for the LARGE models (srbd37: 37x37 / 61x61 tiles, srbd61:
// 61x61 / 109x109, lip30), mapped on one 256-thread workgroup (4 wavefronts, one per SIMD of a CU) per MPC instance.
//
//   * Tile phases of the Riccati sweep: one THREAD per 3x3 block of Q / 2x2 block of Vxx (srbd61: three / two blocks per thread),
//     operands read with ds_read_b128 from rows whose stride is == 2 (mod 4) doubles, so that the rows of a block column fall on
//     16 distinct 16-byte LDS slots (MI355X_MICROARCH.md, LDS: b128 reads conflict per 16-lane group on (addr/4) mod 64).
//   * The products run over the column sparsity of [fx fu]; the SRBD models do so without the dense (Vxx F)^T tile ("W-free":
//     LdsMW, mw_wfree).
//   * Quu solve: block Gauss-Jordan across the 4 waves.  A lane owns column lane (+ 64 for the second one at srbd61) of
//     [Quu + mu I | Qu | Qux]; wave w owns rows [w RPW, (w+1) RPW).  Step b: wave b reduces its own rows (pivot column by
//     v_readlane), publishes them in LDS, the other waves eliminate their rows against them: 4 hand-offs per knot instead of NU.
//   * Forward pass (line search, one lane per step length): the feedback product u = u_k + alpha k + K (x - x_k) is split
//     by rows over the 4 waves with the knot's gains staged in LDS (broadcast reads) and the per-lane vectors kept as LDS
//     columns; the scalar model step runs in wave 0 while the other waves stage the next knot's gains.
//   * Every wave keeps an identical copy of the scalar solver state (cost, merit weight, regularisation, counters); values
//     produced by one wave only reach the others through the control words CTL[..] behind a workgroup barrier.
//
//   * LDS per instance: 52.0 KB (srbd37; two workgroups share a CU: solve_kernel_mw_w2), 32.0 KB (lip30), 149.9 KB (srbd61, one
//     per CU); Q has no tile of its own -- see LdsMW.  tests/test_lds_budget.py pins these.
//
// These models need > 40 KB of LDS per instance (<= 2 workgroups per CU at 256 registers), so the 4 waves do not fight the
// register budget that rules this mapping out for srbd13 at four instances per CU (profiles/experiment_log_r01_r03.md).
// Same device model code, same arithmetic up to summation order, same parity tests as the single-wavefront kernel.
#pragma once
#include "sddp_kernels.hpp"

namespace sddp {

constexpr int kThreadsMW = 256;
constexpr int kWavesMW = kThreadsMW / kWave;

__host__ __device__ constexpr int pad2mod4(int n) { return ((n + 1) & ~3) + 2; }   // smallest m >= n with m == 2 (mod 4)
__host__ __device__ constexpr int round_up(int n, int m) { return (n + m - 1) / m * m; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

// Whether the sweep of model M runs WITHOUT the dense (Vxx F)^T tile ("W-free"): the SRBD models, whose
// [fx fu] has dense rows (ND > 0).  srbd61's tiles would not fit a CU's 160 KB with the tile; srbd37 drops from 63.8 to 50.9 KB (52.0 with the Gauss-Jordan's multiplier block) and
// gains 14 % at two workgroups per CU (the W phase becomes a compact 111 x 7 product).  lip30 (no dense rows: the tile IS the
// product) keeps it: measured neutral to - 1 % without.  -DSDDP_WFREE_ALL / -DSDDP_WFREE_NONE: diagnostic builds.
template <class M>
constexpr bool mw_wfree() {
#if defined(SDDP_WFREE_ALL)
    return true;
#elif defined(SDDP_WFREE_NONE)
    return M::NX > 40;
#else
    return M::ND > 0;
#endif
}

// Whether a build of the kernel recomputes per knot what derives from the thread index (row / column decodes, LDS and HBM addresses)
// instead of carrying it across the knot loops: the thread index passes through an opaque register copy at the top of every knot, so
// LLVM's loop-invariant code motion has nothing to hoist.  Where registers are short the hoisted values are spilled, and a spill
// reload is a `s_waitcnt vmcnt(0)` behind the knot's prefetch (loads return in order).  The half-register-file builds and the wide
// model: scratch of solve_kernel_mw_w2<srbd37> 1 500 -> 1 324 B, + 3 % (srbd37, lip30), + 1.5 % (srbd61); the full-register-file
// builds of srbd37 / lip30 lose 4.5 % with it (profiles/r04/experiments/README.md).
template <class M>
constexpr bool mw_sink(bool half_register_file) { return half_register_file || M::NX > 40; }

#ifndef SDDP_MW_W2_WAVES
#define SDDP_MW_W2_WAVES 2      // diagnostic: 3 = the half-register-file build capped at a third of the register file instead
#endif

template <class M>
struct LdsMW {
    static constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NEV = M::NEV;
    static constexpr bool WFREE = mw_wfree<M>();
    // register-block shapes: W = (V~F~)^T in JW x LW blocks, Q in 3x3 lower-triangle blocks, Vxx in 2x2 lower-triangle blocks
    static constexpr int JW = 3;
    static constexpr int LW = (round_up(NZ, 3) / 3) * (round_up(NX, 3) / 3) <= kThreadsMW ? 3 : 4;
    static constexpr int NJB = round_up(NZ, JW) / JW, NLB = round_up(NX, LW) / LW;
    static constexpr int NBQ = round_up(NZ, 3) / 3, NTRIQ = NBQ * (NBQ + 1) / 2;
    static constexpr int NBV = round_up(NX, 2) / 2, NTRIV = NBV * (NBV + 1) / 2;
    static constexpr int TQ = (NTRIQ + kThreadsMW - 1) / kThreadsMW;            // Q blocks per thread (1: srbd37, lip30; 3: srbd61)
    static constexpr int TV = (NTRIV + kThreadsMW - 1) / kThreadsMW;            // Vxx blocks per thread
    // row strides (doubles), all == 2 (mod 4); row counts padded to the block shapes (pad rows stay zero)
    static constexpr int SV = pad2mod4(NX), RV = round_up(NX, imax(WFREE ? 2 : LW, 2));     // VXX [RV][SV]
    static constexpr int RZ = round_up(NZ, 3);                                  // WT [RZ][SV], FC / WC [RZ][SC]
    static constexpr int ND = M::ND, SC = pad2mod4(ND + NEV);                   // compact columns: dense rows of F | variable extra rows
    static constexpr int DG8 = (ND + 1) & ~1, SGC = pad2mod4(imax(ND, 1));      // W-free: GC [RZ][SGC], product depth DG8 (pad: zero)
    static constexpr int TBL = round_up(RZ, 2);                                 // per-column tables (pad rows: zero)
    static constexpr int SQ = pad2mod4(NZ);                                     // QU [NU][SQ]; the diagonal tables [SQ]
    static constexpr int SK = pad2mod4(NU);                                     // KT [NX][SK]: KT[c][i] = K[i][c]
    static constexpr int RPW = (NU + kWavesMW - 1) / kWavesMW;                  // Gauss-Jordan rows per wave
    static constexpr int NCOL = NU + 1 + NX, CPL = (NCOL + kWave - 1) / kWave;  // augmented columns [Quu | Qu | Qux], columns per lane
    static constexpr int GTS = CPL * kWave;                                     // hand-off row stride
    static constexpr int NSTG = M::NREC + M::NP + NX;                           // staged knot: record | params | defect
    static constexpr int NRECP = (((M::NREC + 1) & ~1) + M::NSO2T + 1) & ~1, NPP = (M::NP + 1) & ~1;   // record + second-order factors (SO2)
    static constexpr int SO2T = (M::NREC + 1) & ~1;
    static constexpr int SG = (NX + 1) & ~1;                                    // staged gain rows in the forward pass
    // ---- tables of the kernel: they survive the forward pass
    static constexpr int DS = 0;                       // constant diagonal, state part [SQ]
    static constexpr int DG = DS + SQ;                 // constant diagonal, stage part [SQ]
    static constexpr int LS = DG + SQ;                 // extra-row weights, state / stage [NE] each
    static constexpr int LG = LS + ((NE + 1) & ~1);
    static constexpr int CTL = LG + ((NE + 1) & ~1);   // control words shared by the 4 waves [16]
    static constexpr int DUMP = CTL + 16;              // where the model code's writes to the unstored (x row, u column) part of Q go
    static constexpr int BET = DUMP + 2;               // sparsity of [fx fu] by column z (M::nbr): entry beta[z] in row nbi[z] ...
    static constexpr int IDC = BET + TBL;              // ... and 1.0 where the identity entry is not part of a dense row [TBL] each
    static constexpr int KI = IDC + TBL;               // ints: dkind[SQ], dci[SQ], nbi[TBL]
    static constexpr int NBI = 2 * SQ;
    static constexpr int SO2L = NBI + TBL;                       // SO2 builds: pair codes of the second-order contraction
    static constexpr int KI_INTS = SO2L + M::NSO2L;
    static constexpr int WORK = KI + ((KI_INTS + 1) / 2 + 1) / 2 * 2;
    // ---- work tiles of the sweep; the forward pass aliases ALL of them (zero_work_mw / ft_constants_mw put back what the sweep relies on).
    // Q is not a tile of its own (two workgroups must fit a CU's 160 KB): its state block Qxx is written INTO the Vxx tile --
    // Vxx_{k+1} is dead once W = (V~ F~)^T is formed, and Vxx_k = Qxx + Qux^T K then updates the tile in place -- its input rows
    // [Qux | Quu] are the QU tile, and the (state row, input column) block, the transpose of Qux, is not stored at all.
    // F~^T is kept COMPACT: FC[z] = (F[D_d][z], d < ND | E[m][z], m < NEV) -- everything of column z that is not the identity or
    // its one neighbour entry (M::nbr); WT = (Vxx F)^T dense over the NX next-state columns, WC the same compact columns of
    // (V~ F~)^T (dense-row columns of WT duplicated | lambda_m E[m][z]).  The products run over the compact index (depth
    // ND + NEV instead of NX + NEV) plus two single terms per row.
    // W-free layout (WFREE): no WT at all.  With column z of F = s_z + f_z (s_z: the identity and neighbour entries, f_z: the dense-row
    // part), Q[z][z'] = s_z^T V s_z' + sum_d FC[z'][d] GC[z][d] + sum_c FC[z][c] WC[z'][c], where GC[z][d] = (V s_z)[D_d] is a compact
    // RZ x ND tile, WC's dense-row columns are GC + (V f_z)[D_d] -- a depth-ND product with the ND x ND block of V -- and the first
    // term is four gathers of V per element.  The gain tile and the hand-off rows then share the region of GC | WC, which are
    // dead between the Q products and the next knot.
    static constexpr int VXX = WORK;
    static constexpr int FC = VXX + RV * SV;
    static constexpr int VX = FC + RZ * SC;
    static constexpr int VP = VX + SV;
    static constexpr int QV = VP + SV;
    static constexpr int REC = QV + SQ;
    static constexpr int PK = REC + NRECP;
    static constexpr int DK = PK + NPP;                 // [SV], pad zero
    static constexpr int KF = DK + SV;                  // kff [SK]
    static constexpr int WT = KF + SK;                  // (W-free: the start of the shared region)
    // the gain tile and the Gauss-Jordan hand-off rows live where WT (W-free: GC | WC) is dead: after the Q phase, until the next knot
    static constexpr int KT = WT;                       // KT [NX][SK]: KT[c][i] = K[i][c]
    static constexpr int GT = KT + ((NX * SK + 1) & ~1);   // hand-off rows, double buffered [2][RPW][GTS]
    static constexpr int RPWE = (RPW + 1) & ~1;
    static constexpr int MU = GT + 2 * RPW * GTS;          // each wave's multipliers for the block being eliminated [4][RPW][RPWE] (row r: its RPW rows)
    static constexpr int MU_END = MU + kWavesMW * RPW * RPWE;
    static constexpr int GC = WT;                                                              // W-free only
    static constexpr int WC = WFREE ? GC + RZ * SGC : imax(WT + RZ * SV, MU_END);              // (small models: the tile is sized by what it hosts)
    static constexpr int QU = WFREE ? imax(WC + RZ * SC, MU_END) : WC + RZ * SC;               // [NU][SQ]: row i = row NX + i of Q (columns: state | input)
    static constexpr int SWEEP_END = QU + NU * SQ;
    // forward pass: per-lane vector columns X | U and the staged gains of one knot
    static constexpr int RO_X = WORK, RO_U = RO_X + NX * kWave, RO_G = RO_U + NU * kWave;
    static constexpr int RO_K = RO_G + ((NU + 1) & ~1);                        // kff [NU] | K [NU][SG]
    // knot operands staged beside the gains, double buffered: x_k [SG] | u_k [NUE] | d_k [SG] | p_k [NPE]
    static constexpr int NUE = (NU + 1) & ~1, NPE = (M::NP + 1) & ~1;
    static constexpr int SB_X = 0, SB_U = SG, SB_D = SB_U + NUE, SB_P = SB_D + SG, SB_N = SB_P + NPE;
    static constexpr int RO_S = RO_K + NU * SG;
    static constexpr int RO_END = RO_S + 2 * SB_N;
    static constexpr int TOTAL = imax(SWEEP_END, RO_END);
    static constexpr size_t BYTES = size_t(TOTAL) * sizeof(double);
    static_assert((FC | VX | VP | QV | REC | PK | DK | KT | KF | GT | MU | GC | DS | DG | LS | LG | CTL | BET | IDC | KI | WORK | WT | WC | QU | RO_G) % 2 == 0, "16-byte aligned sections");
    static_assert(BYTES <= size_t(160) * 1024, "the tiles of one instance must fit a CU's LDS");
    static_assert(NU < kWave, "the pivot columns and the Qu column sit in the first column of every lane");
};

// C[i][j] += sum_m A[i][m] B[j][m], m < DEPTH (even): RA x RB register block, rows read two fp64 at a time (ds_read_b128).
// Software pipelined: the operands of pair q + PF are requested before the FMAs of pair q, so that with one wave per SIMD the
// LDS round trip overlaps the arithmetic.  Ring of PF + 1 register buffers with static indices: the steady-state loop runs
// over whole groups of PF + 1 pairs, the remainder is peeled.  The last groups request up to PF pairs past the end of the
// rows (in bounds: every tile is followed by another LDS section); those values are never used.
template <int RA, int RB, int DEPTH, int PF = 2>
__device__ __forceinline__ void dot_block(const double* A, int lda, const double* B, int ldb, double (&acc)[RA][RB]) {
    constexpr int NPAIR = DEPTH / 2, NB = PF + 1, NGRP = NPAIR / NB, REM = NPAIR % NB;
    static_assert(DEPTH % 2 == 0 && NPAIR >= PF, "even depth, at least PF pairs");
    double2_t a[NB][RA], b[NB][RB];
    auto load = [&](int buf, int pair) {
#pragma unroll
        for (int i = 0; i < RA; ++i) a[buf][i] = lds2(A + i * lda + 2 * pair);
#pragma unroll
        for (int j = 0; j < RB; ++j) b[buf][j] = lds2(B + j * ldb + 2 * pair);
    };
    auto mac = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int j = 0; j < RB; ++j) acc[i][j] = fma(a[buf][i].y, b[buf][j].y, fma(a[buf][i].x, b[buf][j].x, acc[i][j]));
    };
#pragma unroll
    for (int q = 0; q < PF; ++q) load(q, q);
#pragma unroll 1
    for (int g = 0; g < NGRP; ++g) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            load((u + PF) % NB, g * NB + u + PF);
            mac(u);
        }
    }
#pragma unroll
    for (int u = 0; u < REM; ++u) mac(u);
}

// NN consecutive doubles of a row; even NN: the address is 16-byte aligned (callers: even row strides, even offsets)
template <int NN>
__device__ __forceinline__ void load_run(const double* p, double (&v)[NN]) {
    if constexpr (NN % 2 == 0) {
#pragma unroll
        for (int i = 0; i < NN; i += 2) { const double2_t t = lds2(p + i); v[i] = t.x; v[i + 1] = t.y; }
    } else {
#pragma unroll
        for (int i = 0; i < NN; ++i) v[i] = p[i];
    }
}

// Pins an array of loaded values: every element must be in its register here, so all the reads that produce them are issued
// (in flight together) before the first use.  Left alone, the scheduler of these long straight-line phases issues each LDS read
// just before its use: one exposed round trip per element (tools/isa_wait_batches.py shows them).
template <int NN>
__device__ __forceinline__ void pin_regs(double (&v)[NN]) {
#pragma unroll
    for (int i = 0; i < NN; ++i) asm volatile("" : "+v"(v[i]));
}

// t-th block of a lower triangle, row-major: (block row << 8) | block column, row >= column.  A thread owns the same block of Q
// (and of Vxx) for the whole kernel and decodes it once per sweep.
__device__ __forceinline__ int tri_code(int t) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    return (a << 8) | (t - a * (a + 1) / 2);
}

// The work tiles in the state the sweep relies on: everything zero (pad rows and columns of the tiles are summed over), then the
// constant part of F~^T (F_entry on a zero record).  At the start of a solve and after every forward pass, which uses the whole
// work area for its per-lane columns.  zero_work_mw by a group of threads, a barrier, ft_constants_mw by a group, a barrier.
template <class M>
__device__ __forceinline__ void zero_work_mw(double* s, int t, int nthreads) {
    using L = LdsMW<M>;
    for (int e = L::WORK + t; e < L::TOTAL; e += nthreads) s[e] = 0.0;
}
template <class M>
__device__ void ft_constants_mw(const DevConsts& c, double* s, int t, int nthreads) {
    using L = LdsMW<M>;
    constexpr int NZ = M::NZ, NEV = M::NEV, ND = M::ND;
    for (int e = t; e < NZ * (ND + NEV); e += nthreads) {
        const int j = e / (ND + NEV), q = e % (ND + NEV);
        s[L::FC + j * L::SC + q] = q < ND ? M::F_entry(c, s + L::REC, M::dense_row(q), j) : M::E_const(c, q - ND, j);
    }
}

template <class M>
__device__ void sweep_tables_mw(const DevConsts& c, double* s, int tid) {
    using L = LdsMW<M>;
    constexpr int NZ = M::NZ, NE = M::NE;
    for (int e = tid; e < L::WORK; e += kThreadsMW) s[e] = 0.0;
    zero_work_mw<M>(s, tid, kThreadsMW);
    __syncthreads();
    ft_constants_mw<M>(c, s, tid, kThreadsMW);
    __syncthreads();
    int* ki = reinterpret_cast<int*>(s + L::KI);
    for (int i = tid; i < NZ; i += kThreadsMW) {
        s[L::DS + i] = M::dg_state(c, i);
        s[L::DG + i] = M::dg_stage(c, i);
        ki[i] = M::dkind(i);
        ki[L::SQ + i] = M::dci(i);
        M::nbr(c, i, ki[L::NBI + i], s[L::BET + i], s[L::IDC + i]);
    }
    for (int m = tid; m < NE; m += kThreadsMW) {
        s[L::LS + m] = M::lam_state(c, m);
        s[L::LG + m] = M::lam_stage(c, m);
    }
    for (int e = tid; e < M::NSO2L / 2; e += kThreadsMW) M::so2_pair_code(e, ki[L::SO2L + 2 * e], ki[L::SO2L + 2 * e + 1]);
    __syncthreads();
}

// The constant extra rows (m >= NEV: node-independent weights) contribute sum_m lambda_m E[m][row] E[m][col] to Q: a constant
// matrix.  Each thread keeps the 3x3 blocks it owns in the Q phase (LdsMW::TQ of them) in registers for the whole kernel.
template <class M>
__device__ void mw_const_block(const DevConsts& c, int tid, double (&qconst)[LdsMW<M>::TQ][3][3]) {
    using L = LdsMW<M>;
    if constexpr (L::TQ > 1) return;        // several blocks per thread: the constant rows are a sparse pass in the sweep (add_const_rows)
#pragma unroll
    for (int tq = 0; tq < L::TQ; ++tq) {
        const int t = tid + tq * kThreadsMW;
        const int code = t < L::NTRIQ ? tri_code(t) : 0;
        const int a0 = 3 * (code >> 8), b0 = 3 * (code & 255);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double v = 0.0;
                if (M::NEV < M::NE && t < L::NTRIQ && a0 + i < M::NZ && b0 + j < M::NZ)
                    for (int m = M::NEV; m < M::NE; ++m) v += M::lam_stage(c, m) * M::E_const(c, m, a0 + i) * M::E_const(c, m, b0 + j);
                qconst[tq][i][j] = v;
            }
    }
}

// backward Riccati sweep on 4 waves; every thread gets the same return value and the same dV1 / G1 / G2 / qu_inf.
template <class M, bool SINK = false>
__device__ __forceinline__ bool backward_sweep_mw(const DevConsts& c, int N, const double* __restrict__ P, const double* __restrict__ dft,
                                  const double* __restrict__ rec, double* __restrict__ gains, double mu, double theta,
                                  double* s, int tid, double& dV1, double& G1, double& G2, double& qu_inf,
                                  const double (&qconst)[LdsMW<M>::TQ][3][3] SDDP_T_ARG) {
    // qconst: the constant extra rows' share of this thread's 3x3 Q blocks (mw_const_block, once per kernel)
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NE = M::NE, NEV = M::NEV, NREC = M::NREC, NP = M::NP;
    static_assert(!M::CONST_ROWS_STATE_WEIGHTED, "constant rows must have node-independent weights");
    constexpr int SV = L::SV, SC = L::SC, ND = L::ND, SQ = L::SQ, SK = L::SK, NSTG = L::NSTG, RPW = L::RPW;
    constexpr int NCOL = L::NCOL, CPL = L::CPL, GTS = L::GTS, TQ = L::TQ, TV = L::TV;
    constexpr bool WFREE = L::WFREE;
    constexpr int RS = (NSTG + kThreadsMW - 1) / kThreadsMW;
    constexpr int kLast = kWavesMW - 1;
    static_assert(NX <= kWave, "one lane per state in the v' and Vx phases");
    int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);   // uniform: indices and branches on it are scalar
    const int* ki = reinterpret_cast<const int*>(s + L::KI);
    const QSplit<NX> qm{s + L::VXX, SV, s + L::QU, SQ, s + L::DUMP};   // Q as the model code addresses it (LdsMW)
    double g1_acc = 0.0, g2_acc = 0.0, dv_acc = 0.0, qu_acc = 0.0;   // per-wave partial sums, combined after the sweep
    dV1 = G1 = G2 = qu_inf = 0.0;
    auto stage_word = [&](int k, int w) -> double {    // [0,NREC) record | [NREC,NREC+NP) parameters | then the defect
        if (w < NREC) return rec[size_t(k) * NREC + w];
        if (w < NREC + NP) return P[k * NP + (w - NREC)];
        if (w < NSTG) return dft[k * NX + (w - NREC - NP)];
        return 0.0;
    };
    // ---- terminal node: Vx = lx_N, Vxx = lxx_N = diag(D_state) + Je^T Lambda_state Je  (ddp.py:216-226)
    if (tid < SV) s[L::VX + tid] = tid < NX ? rec[size_t(N) * NREC + M::REC_G + tid] : 0.0;
    if (tid < NP) s[L::PK + tid] = P[N * NP + tid];
    __syncthreads();
    for (int e = tid; e < NX * NX; e += kThreadsMW) {
        const int a = e / NX, b = e % NX;
        double v = 0.0;
        for (int m = 0; m < NEV; ++m) v += s[L::LS + m] * s[L::FC + a * SC + ND + m] * s[L::FC + b * SC + ND + m];
        if (a == b) v += s[L::DS + a] + M::dparam(c, s + L::PK, ki[a], ki[SQ + a], 1.0, 0.0);
        s[L::VXX + a * SV + b] = v;
    }
    // the blocks this thread owns in the Q / Vxx phases never change: decoded once per sweep
    int code_q[TQ], code_v[TV];
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq) code_q[tq] = tid + tq * kThreadsMW < L::NTRIQ ? tri_code(tid + tq * kThreadsMW) : 0;
#pragma unroll
    for (int tv = 0; tv < TV; ++tv) code_v[tv] = tid + tv * kThreadsMW < L::NTRIV ? tri_code(tid + tv * kThreadsMW) : 0;
    // where a 3x3 block of Q goes (LdsMW: state rows inside the Vxx tile, input rows in QU, (state row, input column) nowhere):
    // row offsets and one validity bit per element, for the block itself (d) and for its mirror image (m)
    auto q_place = [&](int code, int (&off_d)[3], int (&off_m)[3], unsigned& ok_d, unsigned& ok_m) {
        const int a0 = 3 * (code >> 8), b0 = 3 * (code & 255);
        ok_d = ok_m = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rd = a0 + i, rm = b0 + i;
            off_d[i] = (rd >= NX ? L::QU + (rd - NX) * SQ : L::VXX + rd * SV) + b0;
            off_m[i] = (rm >= NX ? L::QU + (rm - NX) * SQ : L::VXX + rm * SV) + a0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int cd = b0 + j, cm = a0 + j;
                if (rd < NZ && cd < NZ && (rd >= NX || cd < NX)) ok_d |= 1u << (3 * i + j);               // element (a0+i, b0+j)
                if (a0 != b0 && rm < NZ && cm < NZ && (rm >= NX || cm < NX)) ok_m |= 1u << (3 * i + j);   // element (b0+i, a0+j)
            }
        }
    };
    int q_off_d[3], q_off_m[3];      // one block per thread (TQ == 1): placed once per sweep; else per trip
    unsigned q_ok_d = 0, q_ok_m = 0;
    if (TQ == 1) q_place(code_q[0], q_off_d, q_off_m, q_ok_d, q_ok_m);
    double r_stage[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) r_stage[t] = stage_word(N - 1, tid + t * kThreadsMW);
    // gains of knot kk, LDS (KF | K^T as the solve left them; intact until the next knot's W / GC | WC phase) -> HBM/L2, whole-wave
    // contiguous stores one knot later: kff (NU) then K (NU x NX) row-major.  A store in flight turns every wait for a prefetched
    // load into a full vmcnt(0) (one counter for loads and stores on gfx9), so they are issued right after such a wait, not before it
    constexpr int NGW = NU * (NX + 1);
    auto store_gains = [&](int kk) {
        double* gk = gains + size_t(kk) * NGW;
        for (int e = tid; e < NGW; e += kThreadsMW) gk[e] = e < NU ? s[L::KF + e] : s[L::KT + ((e - NU) % NX) * SK + (e - NU) / NX];
    };
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        if constexpr (SINK) {   // (mw_sink: nothing derived from the thread index is carried across knots -- it would be spilled)
            asm volatile("" : "+v"(tid));
            lane = tid & (kWave - 1);
        }
        // ---- stage this knot from the prefetch registers; start the next knot's loads; the previous knot's gains go out
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            const int w = tid + t * kThreadsMW;
            if (w < NREC) s[L::REC + w] = r_stage[t];
            else if (w < NREC + NP) s[L::PK + w - NREC] = r_stage[t];
            else if (w < NSTG) s[L::DK + w - NREC - NP] = r_stage[t];
        }
        if (k > 0) {
#pragma unroll
            for (int t = 0; t < RS; ++t) r_stage[t] = stage_word(k - 1, tid + t * kThreadsMW);
        }
        if (k < N - 1) store_gains(k + 1);
        __syncthreads();
        SDDP_TICK(1)
        const double state = k >= 1 ? 1.0 : 0.0;
        // ---- waves 0..2: variable entries of F~^T ; last wave: v' = Vx + Vxx d and the gap terms
        if (wave != kLast) {
            M::template expand_var<true>(c, s + L::REC, s + L::FC, SC, tid, kThreadsMW - kWave);
        } else if (lane < NX) {
            // one row of Vxx against d per lane: four independent partial sums (the FMA chain is the latency of this phase),
            // operands in groups of four pairs, the next group in flight behind the FMAs of this one
            constexpr int NP2 = SV / 2, G = 4, NG = (NP2 + G - 1) / G;
            double a4[4] = {0.0, 0.0, 0.0, 0.0};
            double2_t vv[2][G], dd[2][G];
            auto load_g = [&](int g, int buf) {
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (g * G + q < NP2) { vv[buf][q] = lds2(s + L::VXX + lane * SV + 2 * (g * G + q)); dd[buf][q] = lds2(s + L::DK + 2 * (g * G + q)); }
            };
            load_g(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_g(g + 1, (g + 1) & 1);
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (g * G + q < NP2) {
                        asm volatile("" : "+v"(vv[g & 1][q]), "+v"(dd[g & 1][q]));
                        a4[q] = fma(vv[g & 1][q].y, dd[g & 1][q].y, fma(vv[g & 1][q].x, dd[g & 1][q].x, a4[q]));   // pads of DK and Vxx are zero
                    }
            }
            const double acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            const double d = s[L::DK + lane], vx = s[L::VX + lane];
            s[L::VP + lane] = vx + acc;
            g1_acc += d * vx;
            g2_acc += 0.5 * d * acc;
        }
        __syncthreads();
        SDDP_TICK(2)
        double qacc[TQ][3][3];       // the Q blocks of this thread (W-free: their s_z^T V s_z' part is formed BEFORE the barrier below,
                                     // while V is still intact; the products and the stores -- Qxx goes into the V tile -- come after it)
        if constexpr (!WFREE) {
        // ---- WT = (Vxx F)^T, JW x LW register blocks: row z = idc(z) Vxx[z] + beta(z) Vxx[n(z)] + sum_d F[D_d][z] Vxx[D_d] (the
        // column sparsity of F, M::nbr); the columns that are dense rows of F also go into the compact tile WC
        for (int blk = tid; blk < L::NJB * L::NLB; blk += kThreadsMW) {
            constexpr int JW = L::JW, LW = L::LW;
            const int j0 = JW * (blk % L::NJB), l0 = LW * (blk / L::NJB);
            double acc[JW][LW];
            {   // the two single terms of every row first: their operands are dead before the dense rows are requested
                double idc[JW], bet[JW], va[JW][LW], vb[JW][LW];
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    const int z = j0 + jj, zc = z < NX ? z : 0, nb = ki[L::NBI + z];      // pad rows: zero tables, zero FC row
                    idc[jj] = s[L::IDC + z];
                    bet[jj] = s[L::BET + z];
                    load_run<LW>(s + L::VXX + zc * SV + l0, va[jj]);
                    load_run<LW>(s + L::VXX + nb * SV + l0, vb[jj]);
                }
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) { pin_regs(va[jj]); pin_regs(vb[jj]); }
#pragma unroll
                for (int jj = 0; jj < JW; ++jj)
#pragma unroll
                    for (int ll = 0; ll < LW; ++ll) acc[jj][ll] = fma(bet[jj], vb[jj][ll], idc[jj] * va[jj][ll]);
            }
            if (ND > 0) {   // dense rows of F: row D_d of Vxx and column d of the compact tile, one step ahead
                double vd[2][LW], fd[2][JW];
                auto load_d = [&](int d, int buf) {
                    load_run<LW>(s + L::VXX + M::dense_row(d) * SV + l0, vd[buf]);
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) fd[buf][jj] = s[L::FC + (j0 + jj) * SC + d];
                };
                load_d(0, 0);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (d + 1 < ND) load_d(d + 1, (d + 1) & 1);
                    pin_regs(vd[d & 1]);
                    pin_regs(fd[d & 1]);
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
                        for (int ll = 0; ll < LW; ++ll) acc[jj][ll] = fma(fd[d & 1][jj], vd[d & 1][ll], acc[jj][ll]);
                }
            }
#pragma unroll
            for (int ll = 0; ll < LW; ++ll) {
                const int l = l0 + ll, wc = M::dense_col(l < NX ? l : 0);
                if (l < NX) {
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) {
                        s[L::WT + (j0 + jj) * SV + l] = acc[jj][ll];
                        if (wc >= 0) s[L::WC + (j0 + jj) * SC + wc] = acc[jj][ll];
                    }
                }
            }
        }
        } else {
        // ---- W-free: the compact tiles GC[z][d] = (V s_z)[D_d] = idc(z) V[z][D_d] + beta(z) V[n(z)][D_d] and
        // WC[z][d] = GC[z][d] + sum_d' FC[z][d'] V[D_d'][D_d], one thread per column z (the ND x ND block of V: broadcast reads)
        if constexpr (ND > 0) {
            for (int z = tid; z < NZ; z += kThreadsMW) {
                const int zc = z < NX ? z : 0, nb = ki[L::NBI + z];
                const double idc = s[L::IDC + z], bet = s[L::BET + z];
                double fc[L::DG8], g[ND], w[ND];
                load_run<L::DG8>(s + L::FC + z * SC, fc);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const double va = s[L::VXX + zc * SV + M::dense_row(d)], vb = s[L::VXX + nb * SV + M::dense_row(d)];
                    g[d] = fma(bet, vb, idc * va);
                    w[d] = g[d];
                }
#pragma unroll
                for (int dp = 0; dp < ND; ++dp) {
                    double vrow[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d) vrow[d] = s[L::VXX + M::dense_row(dp) * SV + M::dense_row(d)];
                    pin_regs(vrow);
#pragma unroll
                    for (int d = 0; d < ND; ++d) w[d] = fma(fc[dp], vrow[d], w[d]);
                }
#pragma unroll
                for (int d = 0; d < L::SGC; ++d) s[L::GC + z * L::SGC + d] = d < ND ? g[d < ND ? d : 0] : 0.0;   // pad columns zero: the product runs over DG8
#pragma unroll
                for (int d = 0; d < ND; ++d) s[L::WC + z * SC + d] = w[d];
            }
        }
        // ... and the s_z^T V s_z' part of this thread's Q blocks: four gathers of V per element
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
            const int a0 = 3 * (code_q[tq] >> 8), b0 = 3 * (code_q[tq] & 255);
            double ia[3], ba[3], ib[3], bb[3];
            int za[3], na[3], zb[3], nb[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int z = a0 + i, y = b0 + i;
                za[i] = z < NX ? z : 0; na[i] = ki[L::NBI + z]; ia[i] = s[L::IDC + z]; ba[i] = s[L::BET + z];
                zb[i] = y < NX ? y : 0; nb[i] = ki[L::NBI + y]; ib[i] = s[L::IDC + y]; bb[i] = s[L::BET + y];
            }
            double v00[3][3], v01[3][3], v10[3][3], v11[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    v00[i][j] = s[L::VXX + za[i] * SV + zb[j]];
                    v01[i][j] = s[L::VXX + za[i] * SV + nb[j]];
                    v10[i][j] = s[L::VXX + na[i] * SV + zb[j]];
                    v11[i][j] = s[L::VXX + na[i] * SV + nb[j]];
                }
#pragma unroll
            for (int i = 0; i < 3; ++i) { pin_regs(v00[i]); pin_regs(v01[i]); pin_regs(v10[i]); pin_regs(v11[i]); }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    qacc[tq][i][j] = fma(ia[i], fma(ib[j], v00[i][j], bb[j] * v01[i][j]), ba[i] * fma(ib[j], v10[i][j], bb[j] * v11[i][j]));
        }
        }
        for (int e = tid; e < NZ * (SC - ND); e += kThreadsMW) {    // extra rows: a scaling of F~^T (pad columns of WC: zero)
            const int j = e / (SC - ND), m = e % (SC - ND);
            const double lam = m < NEV ? state * s[L::LS + m] + s[L::LG + m] : 0.0;
            s[L::WC + j * SC + ND + m] = m < NEV ? lam * s[L::FC + j * SC + ND + m] : 0.0;
        }
        __syncthreads();
        SDDP_TICK(3)
        // ---- Q = diag(D) + F~^T (V~ F~): 3x3 lower-triangle blocks, mirrored ; q = g + F^T v'
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
        if (tid + tq * kThreadsMW < L::NTRIQ) {
            const int code = code_q[tq];
            const int a0 = 3 * (code >> 8), b0 = 3 * (code & 255);
            double acc[3][3] = {};
            if constexpr (!WFREE) {   // the two single terms of every row a0 + i against rows b0 + j of WT first (dead before the product's operand ring)
                double idc[3], bet[3], wa[3][3], wb[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int z = a0 + i, zc = z < NX ? z : 0, nb = ki[L::NBI + z];
                    idc[i] = s[L::IDC + z];
                    bet[i] = s[L::BET + z];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { wa[i][j] = s[L::WT + (b0 + j) * SV + zc]; wb[i][j] = s[L::WT + (b0 + j) * SV + nb]; }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) { pin_regs(wa[i]); pin_regs(wb[i]); }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = fma(bet[i], wb[i][j], idc[i] * wa[i][j]);
            } else {                  // W-free: s_z^T V s_z' from before the barrier, then s_z^T V f_z' = sum_d GC[z][d] FC[z'][d]
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = qacc[tq][i][j];
                if constexpr (ND > 0) dot_block<3, 3, L::DG8>(s + L::GC + a0 * L::SGC, L::SGC, s + L::FC + b0 * SC, SC, acc);
            }
            dot_block<3, 3, SC>(s + L::FC + a0 * SC, SC, s + L::WC + b0 * SC, SC, acc);
            if (NEV < NE && TQ == 1) {      // (several blocks per thread: the constant rows are a sparse pass after the barrier)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] += qconst[tq][i][j];
            }
            if (a0 == b0) {   // diagonal block: add D, keep it exactly symmetric
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (a0 + i < NZ)
                        acc[i][i] += state * s[L::DS + a0 + i] + s[L::DG + a0 + i] +
                                     M::dparam(c, s + L::PK, ki[a0 + i], ki[SQ + a0 + i], state, 1.0);
#pragma unroll
                    for (int j = 0; j < i; ++j) { const double off = 0.5 * (acc[i][j] + acc[j][i]); acc[i][j] = acc[j][i] = off; }
                }
            }
            // state rows go into the Vxx tile (dead since the W phase; both triangles: the Vxx update reads whole 2x2 blocks),
            // input rows into QU (both triangles of Quu: the solve reads whole rows), (state row, input column) is not stored
            if (TQ > 1) q_place(code, q_off_d, q_off_m, q_ok_d, q_ok_m);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if ((q_ok_d >> (3 * i + j)) & 1u) s[q_off_d[i] + j] = acc[i][j];
                    if ((q_ok_m >> (3 * j + i)) & 1u) s[q_off_m[j] + i] = acc[i][j];      // (b0+j, a0+i)
                }
        }
        }
        SDDP_TICK(13)
        if (wave == kLast) {
            for (int j = lane; j < NZ; j += kWave) {
                const int zc = j < NX ? j : 0, nb = ki[L::NBI + j];
                double acc = s[L::REC + M::REC_G + j] + fma(s[L::BET + j], s[L::VP + nb], s[L::IDC + j] * s[L::VP + zc]);
#pragma unroll
                for (int d = 0; d < ND; ++d) acc = fma(s[L::FC + j * SC + d], s[L::VP + M::dense_row(d)], acc);
                s[L::QV + j] = acc;
            }
        }
        __syncthreads();
        SDDP_TICK(14)
        constexpr bool kConstPass = TQ > 1 && NEV < NE;     // the constant extra rows as a sparse pass (entries disjoint from the torque term's)
        if (kConstPass) M::add_const_rows(c, qm, tid, kThreadsMW);
        if (theta != 0.0) {   // exact second-order torque term (uniform switch)
            if (M::NSO2T) {   // full second-order builds: the per-knot factors of the contraction first
                M::so2_prepare(c, s + L::REC, s + L::VP, s + L::REC + L::SO2T, tid, kThreadsMW);
                __syncthreads();
            }
            M::add_second_order(c, s + L::REC, s + L::VP, qm, theta, tid, kThreadsMW, s + L::REC + L::SO2T,
                                reinterpret_cast<const int*>(s + L::KI) + L::SO2L);
            __syncthreads();
        } else if (kConstPass) __syncthreads();
        if (M::BAR) {         // friction-cone barrier builds: its Hessian blocks on the force-force diagonal of Q
            M::add_barrier(s + L::REC, qm, tid, kThreadsMW, M::SO2 ? theta : 0.0);
            __syncthreads();
        }
        SDDP_TICK(4)
#ifdef SDDP_GJ_REDUNDANT
        constexpr bool kGjRedundant = NU <= 24 && NU + (NX + 1 + kWavesMW - 1) / kWavesMW <= kWave;
#else
        constexpr bool kGjRedundant = false;
#endif
        if constexpr (kGjRedundant) {
            // Experiment (profiles/r05/experiments): NO hand-off inside the solve.  Every wave eliminates the
            // whole [Quu + mu I] block REDUNDANTLY (lanes 0..NU-1: one column each, all NU rows in registers) together with ITS
            // quarter of the right-hand sides [Qu | Qux] (lanes NU..NU+CW-1); pivot columns by v_readlane inside the wave.  The four
            // waves compute the same pivots bit for bit, so the positive-definiteness decision needs no exchange.
            constexpr int CW = (NX + 1 + kWavesMW - 1) / kWavesMW;
            const int rc = wave * CW + (lane - NU);                       // right-hand side this lane holds: 0 = Qu, c + 1 = Qux column c
            const bool is_rhs = lane >= NU && lane < NU + CW && rc <= NX;
            const int qc = lane < NU ? NX + lane : (is_rhs && rc > 0 ? rc - 1 : 0);
            double a[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double v = s[L::QU + i * SQ + qc];
                v = (is_rhs && rc == 0) ? s[L::QV + NX + i] : v;
                v += (i == lane) ? mu : 0.0;
                a[i] = (lane < NU || is_rhs) ? v : 0.0;
            }
            if (is_rhs && rc == 0) {
#pragma unroll
                for (int i = 0; i < NU; ++i) qu_acc = fmax(qu_acc, fabs(a[i]));
            }
            SDDP_TICK(16)
            bool okp = true;
#pragma unroll
            for (int p = 0; p < NU; ++p) {
                double pv[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) pv[i] = readlane_d(a[i], p);
                if (!(pv[p] > 0.0) || !(pv[p] < 1e300)) okp = false;
                const double t = a[p] * fast_rcp(pv[p]);
#pragma unroll
                for (int i = 0; i < NU; ++i) a[i] = (i == p) ? t : fma(-pv[i], t, a[i]);
            }
            if (!okp) return false;                                       // the same in every wave
            SDDP_TICK(19)
            if (is_rhs) {
                double* dst = rc == 0 ? s + L::KF : s + L::KT + (rc - 1) * SK;
                double dv = 0.0;
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    dst[i] = -a[i];
                    if (rc == 0) dv += -a[i] * s[L::QV + NX + i];
                }
                // dv_acc / qu_acc are read from lane NU of every wave at the end of the sweep: the Qu column is lane NU of wave 0
                if (rc == 0) dv_acc += dv;
            }
        } else {
            double a[RPW][CPL], qu_save[RPW];
            // column of Q a slot reads (any valid one for the slot of column NU, which takes q instead; columns >= NCOL are zeroed): branch-free
            int qcol[CPL];
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const int col = lane + kWave * cc;
                qcol[cc] = col < NU ? NX + col : (col > NU && col < NCOL ? col - NU - 1 : 0);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = wave * RPW + r, ic = i < NU ? i : NU - 1;
                const double qv = s[L::QV + NX + ic];
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const int col = lane + kWave * cc;
                    double v = s[L::QU + ic * SQ + qcol[cc]];
                    v = col == NU ? qv : v;
                    v += (i == col) ? mu : 0.0;
                    v = (i < NU && col < NCOL) ? v : 0.0;
                    a[r][cc] = v;
                }
                qu_save[r] = a[r][0];
                qu_acc = fmax(qu_acc, fabs(a[r][0]));          // only lane NU's value is used
            }
            SDDP_TICK(16)
#pragma unroll
            for (int blk = 0; blk < kWavesMW; ++blk) {
                double* gt = s + L::GT + (blk & 1) * RPW * GTS;
                if (wave == blk) {
                    bool ok = true;
#ifdef SDDP_GJ_PAIR
                    // Experiment (profiles/r05/experiments #11): the owner's pivots two at a time -- the 2 x 2 diagonal block is
                    // inverted with ONE reciprocal (of its determinant) and both pivot columns are fetched before either is used, so
                    // the serial chain readlane -> reciprocal -> scale -> update runs RPW / 2 times per block instead of RPW times.
                    // Positive definite <=> a00 > 0 and det > 0 (the second pivot of the one-at-a-time order is det / a00).
#pragma unroll
                    for (int r = 0; r + 1 < RPW; r += 2) {
                        const int p = blk * RPW + r;
                        if (p + 1 < NU) {
                            double p0[RPW], p1[RPW];
#pragma unroll
                            for (int rr = 0; rr < RPW; ++rr) { p0[rr] = readlane_d(a[rr][0], p); p1[rr] = readlane_d(a[rr][0], p + 1); }
                            const double det = fma(p0[r], p1[r + 1], -p1[r] * p0[r + 1]);
                            if (!(p0[r] > 0.0) || !(det > 0.0) || !(det < 1e300)) ok = false;
                            const double id = fast_rcp(det);
                            const double i00 = p1[r + 1] * id, i01 = -p1[r] * id, i10 = -p0[r + 1] * id, i11 = p0[r] * id;
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) {
                                const double t0 = fma(i00, a[r][cc], i01 * a[r + 1][cc]);
                                const double t1 = fma(i10, a[r][cc], i11 * a[r + 1][cc]);
#pragma unroll
                                for (int rr = 0; rr < RPW; ++rr)
                                    a[rr][cc] = (rr == r) ? t0 : (rr == r + 1) ? t1 : fma(-p1[rr], t1, fma(-p0[rr], t0, a[rr][cc]));
                            }
                        }
                    }
                    // an odd last row of the block, or a last pivot without a partner, goes alone (always behind the pairs)
                    auto paired = [&](int r) { return (r & ~1) + 1 < RPW && blk * RPW + (r & ~1) + 1 < NU; };
#else
                    auto paired = [](int) { return false; };
#endif
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int p = blk * RPW + r;
                        if (p < NU && !paired(r)) {
                            double pv[RPW];
#pragma unroll
                            for (int rr = 0; rr < RPW; ++rr) pv[rr] = readlane_d(a[rr][0], p);
                            if (!(pv[r] > 0.0) || !(pv[r] < 1e300)) ok = false;
                            const double ip = fast_rcp(pv[r]);
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) {
                                const double t = a[r][cc] * ip;
#pragma unroll
                                for (int rr = 0; rr < RPW; ++rr) a[rr][cc] = (rr == r) ? t : fma(-pv[rr], t, a[rr][cc]);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) gt[r * GTS + lane + kWave * cc] = a[r][cc];
                    if (lane == 0) s[L::CTL + 14 + (blk & 1)] = ok ? 1.0 : 0.0;
                }
                // The multipliers of this block -- a[rr] in the lanes of the block's pivot columns -- go to LDS while the owner reduces
                // its rows; the apply step below reads them back as broadcasts (one ds_read_b128 per two multipliers instead of two
                // v_readlane per multiplier; the same values, so the same results).
                double* mul = s + L::MU + wave * RPW * L::RPWE;
                if (wave != blk) {
                    const int r = lane - blk * RPW;
                    if (r >= 0 && r < RPW) {
#pragma unroll
                        for (int rr = 0; rr < RPW; ++rr) mul[r * L::RPWE + rr] = a[rr][0];
                    }
                }
                SDDP_TICK(17)
                __syncthreads();
                SDDP_TICK(18)
                if (s[L::CTL + 14 + (blk & 1)] == 0.0) return false;
                if (wave != blk) {
                    if constexpr (RPW * RPW <= 36) {
                        double pv[RPW][L::RPWE], tv[RPW][CPL];
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) tv[r][cc] = gt[r * GTS + lane + kWave * cc];
#pragma unroll
                        for (int r = 0; r < RPW; ++r) load_run<L::RPWE>(mul + r * L::RPWE, pv[r]);             // pv[r][rr]: multiplier of row rr against published row r
#pragma unroll
                        for (int r = 0; r < RPW; ++r) { pin_regs(tv[r]); pin_regs(pv[r]); }
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            if (blk * RPW + r < NU) {
#pragma unroll
                                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                                    for (int cc = 0; cc < CPL; ++cc) a[rr][cc] = fma(-pv[r][rr], tv[r][cc], a[rr][cc]);
                            }
                        }
                    } else {
                        // many rows per wave: the multipliers of one published row at a time (RPW x RPW of them do not fit the
                        // registers).  Row r of the reduced block is the unit vector in the block's own columns but for rounding,
                        // so the multiplier a[rr] at lane p_r is still the original entry when row r is reached.
                        double tv[2][CPL], pv[2][L::RPWE];
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) tv[0][cc] = gt[lane + kWave * cc];
                        load_run<L::RPWE>(mul, pv[0]);
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            if (r + 1 < RPW) {
#pragma unroll
                                for (int cc = 0; cc < CPL; ++cc) tv[(r + 1) & 1][cc] = gt[(r + 1) * GTS + lane + kWave * cc];
                                load_run<L::RPWE>(mul + (r + 1) * L::RPWE, pv[(r + 1) & 1]);
                            }
                            pin_regs(tv[r & 1]);
                            pin_regs(pv[r & 1]);
                            if (blk * RPW + r < NU) {
#pragma unroll
                                for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                                    for (int cc = 0; cc < CPL; ++cc) a[rr][cc] = fma(-pv[r & 1][rr], tv[r & 1][cc], a[rr][cc]);
                            }
                        }
                    }
                }
            }
            SDDP_TICK(19)
            // a = Quu^-1 * column ; publish kff and K^T (negated) ; dV1 += kff . Qu
            double dv = 0.0;
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const int col = lane + kWave * cc;
                double* dst = col == NU ? s + L::KF : s + L::KT + (col - NU - 1) * SK;
                const bool pub = col >= NU && col < NCOL;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int i = wave * RPW + r;
                    if (i < NU) {
                        if (pub) dst[i] = -a[r][cc];          // (the copy to HBM/L2: store_gains, one knot later)
                        if (cc == 0) dv += -a[r][0] * qu_save[r];
                    }
                }
            }
            dv_acc += dv;                                 // only lane NU's value is used
        }
        SDDP_TICK(20)
        __syncthreads();
        SDDP_TICK(5)
        // ---- Vx = Qx + Qux^T kff ; Vxx = Qxx + Qux^T K (symmetric: lower triangle in 2x2 blocks, mirrored)
        if (wave == kLast && lane < NX) {
            double acc = s[L::QV + lane], qr[NU], kv[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) { qr[i] = s[L::QU + i * SQ + lane]; kv[i] = s[L::KF + i]; }     // Qxu[lane][i] = Qux[i][lane]
            pin_regs(qr);
            pin_regs(kv);
#pragma unroll
            for (int i = 0; i < NU; ++i) acc += qr[i] * kv[i];
            s[L::VX + lane] = acc;
        }
        if constexpr (TV > 1) {
            // several blocks per thread (srbd61, one wave per SIMD): their products run INTERLEAVED -- TV independent chains and TV times
            // the LDS reads in flight per input row; one block after the other leaves every round trip exposed.  Same sums, same order.
            int a0[TV], c0[TV], a1[TV], c1[TV];
            bool on[TV];
            double v[TV][4];
#pragma unroll
            for (int tv = 0; tv < TV; ++tv) {
                on[tv] = tid + tv * kThreadsMW < L::NTRIV;
                const int code = on[tv] ? code_v[tv] : 0;
                a0[tv] = 2 * (code >> 8); c0[tv] = 2 * (code & 255);
                a1[tv] = a0[tv] + 1 < NX ? a0[tv] + 1 : a0[tv]; c1[tv] = c0[tv] + 1 < NX ? c0[tv] + 1 : c0[tv];
                v[tv][0] = v[tv][1] = v[tv][2] = v[tv][3] = 0.0;
            }
#pragma unroll 4
            for (int i = 0; i < NU; ++i) {
                double2_t q2[TV];
                double kc[TV], kd[TV];
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    q2[tv] = lds2(s + L::QU + i * SQ + a0[tv]);
                    kc[tv] = s[L::KT + c0[tv] * SK + i];
                    kd[tv] = s[L::KT + c1[tv] * SK + i];
                }
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    v[tv][0] = fma(q2[tv].x, kc[tv], v[tv][0]);
                    v[tv][1] = fma(q2[tv].x, kd[tv], v[tv][1]);
                    v[tv][2] = fma(q2[tv].y, kc[tv], v[tv][2]);
                    v[tv][3] = fma(q2[tv].y, kd[tv], v[tv][3]);
                }
            }
#pragma unroll
            for (int tv = 0; tv < TV; ++tv) {
                if (!on[tv]) continue;
                double v00 = v[tv][0] + s[L::VXX + a0[tv] * SV + c0[tv]], v01 = v[tv][1] + s[L::VXX + a0[tv] * SV + c1[tv]];
                double v10 = v[tv][2] + s[L::VXX + a1[tv] * SV + c0[tv]], v11 = v[tv][3] + s[L::VXX + a1[tv] * SV + c1[tv]];
                if (a0[tv] == c0[tv]) { const double off = 0.5 * (v01 + v10); v01 = v10 = off; }
                const bool ha = a0[tv] + 1 < NX, hc = c0[tv] + 1 < NX;
                s[L::VXX + a0[tv] * SV + c0[tv]] = v00;
                if (hc) s[L::VXX + a0[tv] * SV + c0[tv] + 1] = v01;
                if (ha) s[L::VXX + (a0[tv] + 1) * SV + c0[tv]] = v10;
                if (ha && hc) s[L::VXX + (a0[tv] + 1) * SV + c0[tv] + 1] = v11;
                if (a0[tv] != c0[tv]) {
                    s[L::VXX + c0[tv] * SV + a0[tv]] = v00;
                    if (hc) s[L::VXX + (c0[tv] + 1) * SV + a0[tv]] = v01;
                    if (ha) s[L::VXX + c0[tv] * SV + a0[tv] + 1] = v10;
                    if (ha && hc) s[L::VXX + (c0[tv] + 1) * SV + a0[tv] + 1] = v11;
                }
            }
        } else {
#pragma unroll
        for (int tv = 0; tv < TV; ++tv) {
        if (tid + tv * kThreadsMW < L::NTRIV) {
            const int code = code_v[tv];
            const int a0 = 2 * (code >> 8), c0 = 2 * (code & 255);
            const int a1 = a0 + 1 < NX ? a0 + 1 : a0, c1 = c0 + 1 < NX ? c0 + 1 : c0;    // odd NX: clamp, not stored
            double v00 = 0, v01 = 0, v10 = 0, v11 = 0;
#pragma unroll 4
            for (int i = 0; i < NU; ++i) {
                const double2_t q2 = lds2(s + L::QU + i * SQ + a0);       // Qux[i][a0], Qux[i][a0 + 1]: a0 is even, so are SQ and QU
                const double qa = q2.x, qb = q2.y;                        // (odd NX, last block: qb is a finite neighbour, not stored)
                const double kc = s[L::KT + c0 * SK + i], kd = s[L::KT + c1 * SK + i];
                v00 = fma(qa, kc, v00);
                v01 = fma(qa, kd, v01);
                v10 = fma(qb, kc, v10);
                v11 = fma(qb, kd, v11);
            }
            v00 += s[L::VXX + a0 * SV + c0];         // Qxx sits in the tile it is about to become (in place: a thread reads only
            v01 += s[L::VXX + a0 * SV + c1];         // the blocks it owns; the mirrored stores below go to blocks above the
            v10 += s[L::VXX + a1 * SV + c0];         // diagonal, which no thread reads here)
            v11 += s[L::VXX + a1 * SV + c1];
            if (a0 == c0) { const double off = 0.5 * (v01 + v10); v01 = v10 = off; }
            const bool ha = a0 + 1 < NX, hc = c0 + 1 < NX;
            s[L::VXX + a0 * SV + c0] = v00;
            if (hc) s[L::VXX + a0 * SV + c0 + 1] = v01;
            if (ha) s[L::VXX + (a0 + 1) * SV + c0] = v10;
            if (ha && hc) s[L::VXX + (a0 + 1) * SV + c0 + 1] = v11;
            if (a0 != c0) {
                s[L::VXX + c0 * SV + a0] = v00;
                if (hc) s[L::VXX + (c0 + 1) * SV + a0] = v01;
                if (ha) s[L::VXX + c0 * SV + a0 + 1] = v10;
                if (ha && hc) s[L::VXX + (c0 + 1) * SV + a0 + 1] = v11;
            }
        }
        }
        }
        __syncthreads();
        SDDP_TICK(6)
    }
    store_gains(0);
    // ---- combine the per-wave partial sums (same values, same order in every thread)
    if (wave == kLast) {
        g1_acc = wave_sum(g1_acc);
        g2_acc = wave_sum(g2_acc);
        if (lane == 0) { s[L::CTL + 1] = g1_acc; s[L::CTL + 2] = g2_acc; }
    }
    if (lane == NU) { s[L::CTL + 4 + wave] = dv_acc; s[L::CTL + 8 + wave] = qu_acc; }
    __syncthreads();
    G1 = s[L::CTL + 1];
    G2 = s[L::CTL + 2];
#pragma unroll
    for (int w = 0; w < kWavesMW; ++w) { dV1 += s[L::CTL + 4 + w]; qu_inf = fmax(qu_inf, s[L::CTL + 8 + w]); }
    __syncthreads();
    return true;
}

// forward pass on 4 waves (called by every thread): lane l of every wave works on step length alpha_l.
// Returns the cost of lane l's trajectory in wave 0 (other waves: unspecified).  Clobbers the whole work area of LdsMW (zero_work_mw / ft_constants_mw put it back).
// Per knot: (A) wave 0 closes the previous knot, (B) every wave computes its rows of the feedback law, (C) wave 0 steps the
// model while the other waves fetch the next knot's operands (gains, x_k, u_k, d_k, p_k: coalesced loads into LDS, read back
// as broadcasts) and write the stored lane's x_k / u_k to HBM.
template <class M, bool OPEN_LOOP, bool SINK = false>
__device__ __forceinline__ double rollout_mw(const DevConsts& c, int N, const double* __restrict__ x0, const double* __restrict__ P,
                             const double* __restrict__ xs, const double* __restrict__ us, const double* __restrict__ dft,
                             const double* __restrict__ gains, double* __restrict__ xn, double* __restrict__ un,
                             double alpha, int store_lane, int tid, double* s SDDP_T_ARG) {
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NG = NU * (NX + 1), SG = L::SG;
    constexpr int UPW = (NU + kWavesMW - 1) / kWavesMW;            // feedback rows per wave
    constexpr int kStagers = kThreadsMW - kWave;
    constexpr int NSB = 2 * NX + NU + NP;
    int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);   // uniform: indices and branches on it are scalar
    const LdsCol X{s + L::RO_X + lane}, U{s + L::RO_U + lane};
    double* kf = s + L::RO_G;                                      // staged gains of one knot: kff [NU]
    double* kb = s + L::RO_K;                                      //                           K [NU][SG]
    // Staging of a knot's operands by waves 1..3 (stager thread se = tid - 64): every load of the knot first, into registers, then
    // the LDS writes -- one memory round trip per knot.  (A loop of load / wait / LDS store pairs pays one per trip: 5 for the
    // gains of srbd37 plus the operand word, each behind reloads of spilled addresses in the two-per-SIMD build.)  Where the
    // registers allow (kEarlyFetch) the loads are issued before the feedback phase and land during it.
    constexpr int TS = (NSB + kStagers - 1) / kStagers, TG = (NG + kStagers - 1) / kStagers;
    constexpr bool kEarlyFetch = false;
    int se = tid - kWave;
    double r_s[TS], r_g[TG];
    auto fetch_knot = [&](int k) {
        if (!OPEN_LOOP) {
            const double* gk = gains + size_t(k) * NG;
#pragma unroll
            for (int t = 0; t < TG; ++t) { const int e = se + t * kStagers; r_g[t] = e < NG ? gk[e] : 0.0; }
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            const int e = se + t * kStagers;
            const double* src = e < NX ? xs + k * NX + e : e < NX + NU ? us + k * NU + (e - NX)
                              : e < 2 * NX + NU ? dft + k * NX + (e - NX - NU) : P + k * NP + (e - 2 * NX - NU);
            r_s[t] = e < NSB ? *src : 0.0;
        }
    };
    auto put_knot = [&](int k) {
        double* sb = s + L::RO_S + (k & 1) * L::SB_N;
        if (!OPEN_LOOP) {
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int e = se + t * kStagers;
                if (e < NG) {
                    if (e < NU) kf[e] = r_g[t];
                    else { const int i = (e - NU) / NX, j = (e - NU) % NX; kb[i * SG + j] = r_g[t]; }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            const int e = se + t * kStagers;
            if (e < NSB) sb[e < NX ? L::SB_X + e : e < NX + NU ? L::SB_U + e - NX : e < 2 * NX + NU ? L::SB_D + e - NX - NU : L::SB_P + e - 2 * NX - NU] = r_s[t];
        }
    };
    __syncthreads();                                               // the tiles this pass aliases are no longer read
    if (wave == 0) {
        for (int i = 0; i < NX; ++i) X[i] = x0[i];
    } else {
        // the pad column of the staged gain rows (odd nx) is read by the wide feedback product against a zero of x - x_k: it has to
        // be finite, and a kernel that starts with this pass (forward_kernel_mw) finds whatever the LDS held before
        if (SG > NX && se < NU) kb[se * SG + NX] = 0.0;
        fetch_knot(0);
        put_knot(0);
    }
    double J = 0.0;
    const double oma = 1.0 - alpha;
    for (int k = 0; k < N; ++k) {
        if constexpr (SINK) {
            asm volatile("" : "+v"(tid));
            lane = tid & (kWave - 1);
            se = tid - kWave;
        }
        SDDP_TICK(8)
        const double* sb = s + L::RO_S + (k & 1) * L::SB_N;
        // (x_k is in the X columns: x0, or written by the step of knot k - 1, which also closed that knot -- LdsColClose)
        __syncthreads();
        SDDP_TICK(11)
        if (kEarlyFetch && wave != 0 && k + 1 < N) fetch_knot(k + 1);
        // ---- B: every wave computes UPW rows of u = u_k + alpha kff + K (x - x_k)
        if (wave == kWavesMW - 1 && lane < NX) xn[k * NX + lane] = s[L::RO_X + lane * kWave + store_lane];
        if constexpr (NX <= 40) {
            double dx[NX];
            if (!OPEN_LOOP) {
                // x - x_k in chunks of CH entries, the next chunk's reads in flight behind the subtractions of this one
                constexpr int CH = 10, NCH = (NX + CH - 1) / CH;
                double xk[2][CH];
                auto load_chunk = [&](int cI) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        if (cI * CH + j < NX) { dx[cI * CH + j] = X[cI * CH + j]; xk[cI & 1][j] = sb[L::SB_X + cI * CH + j]; }
                };
                load_chunk(0);
#pragma unroll
                for (int cI = 0; cI < NCH; ++cI) {
                    if (cI + 1 < NCH) load_chunk(cI + 1);
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        if (cI * CH + j < NX) { asm volatile("" : "+v"(dx[cI * CH + j])); asm volatile("" : "+v"(xk[cI & 1][j])); }
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        if (cI * CH + j < NX) dx[cI * CH + j] -= xk[cI & 1][j];
                }
            }
            // gain rows in units of HU entries, one unit ahead of the unit being used (two register units, pinned unit by unit):
            // the reads stay in flight behind the FMAs without holding whole rows in registers
            constexpr int NXE = (NX + 1) & ~1, HU = 10, NUNIT = (NXE + HU - 1) / HU, NT = UPW * NUNIT;
            double gb[2][HU], ub[UPW], kv[UPW];
            auto load_unit = [&](int t, double (&dst)[HU]) {
                const int r = t / NUNIT, u = t % NUNIT;
                const int i = wave * UPW + r < NU ? wave * UPW + r : NU - 1;
                const double* row = kb + i * SG + u * HU;
#pragma unroll
                for (int j = 0; j < HU; j += 2) {
                    if (u * HU + j < NXE) {                      // SG >= NXE: the pad column is finite, multiplied by nothing
                        const double2_t v = lds2(row + j);
                        dst[j] = v.x;
                        dst[j + 1] = v.y;
                    }
                }
            };
#pragma unroll
            for (int r = 0; r < UPW; ++r) {
                const int i = wave * UPW + r < NU ? wave * UPW + r : NU - 1;
                ub[r] = sb[L::SB_U + i];
                kv[r] = OPEN_LOOP ? 0.0 : kf[i];
            }
            if (OPEN_LOOP) {
#pragma unroll
                for (int r = 0; r < UPW; ++r)
                    if (wave * UPW + r < NU) U[wave * UPW + r] = ub[r];
            } else {
                load_unit(0, gb[0]);
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int r = t / NUNIT, u = t % NUNIT;
                    if (t + 1 < NT) load_unit(t + 1, gb[(t + 1) & 1]);
#pragma unroll
                    for (int j = 0; j < HU; ++j)
                        if (u * HU + j < NXE) asm volatile("" : "+v"(gb[t & 1][j]));
                    if (u == 0) acc = fma(alpha, kv[r], ub[r]);
#pragma unroll
                    for (int j = 0; j < HU; ++j)
                        if (u * HU + j < NX) acc = fma(gb[t & 1][j], dx[u * HU + j], acc);
                    if (u == NUNIT - 1 && wave * UPW + r < NU) U[wave * UPW + r] = acc;
                }
            }
        }
        else {
            // Wide models (nx = 61): x - x_k does not fit the registers beside the gain units (all of it live: 61 doubles spill, and
            // the spills of a whole workgroup fall out of the L2).  Chunks of CH columns in a RUNTIME loop, the wave's UPW rows
            // inside: x - x_k of one chunk lives in CH registers, every row keeps one accumulator, the gain rows are read CH entries at
            // a time one row ahead.  Per row the sum still runs over the columns in ascending order, starting from u_k + alpha kff.
            constexpr int CH = 10, NCH = (NX + CH - 1) / CH;
            static_assert(NCH * CH <= SG + CH, "the last chunk reads at most one unit past a gain row (finite staged data)");
            double acc[UPW];
#pragma unroll
            for (int r = 0; r < UPW; ++r) {
                const int i = wave * UPW + r < NU ? wave * UPW + r : NU - 1;
                const double ub = sb[L::SB_U + i];
                if (OPEN_LOOP) { if (wave * UPW + r < NU) U[wave * UPW + r] = ub; }
                else acc[r] = fma(alpha, kf[i], ub);
            }
            if (!OPEN_LOOP) {
#pragma unroll 1
                for (int cI = 0; cI < NCH; ++cI) {
                    const int c0 = cI * CH;
                    double dx[CH], xv[CH], xkv[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) { xv[j] = X[min(c0 + j, NX - 1)]; xkv[j] = sb[L::SB_X + min(c0 + j, NX - 1)]; }
                    pin_regs(xv);
                    pin_regs(xkv);
#pragma unroll
                    for (int j = 0; j < CH; ++j) dx[j] = c0 + j < NX ? xv[j] - xkv[j] : 0.0;     // columns past nx: multiplied by 0
                    // gain rows RING - 1 rows ahead of the row being used: one wave per SIMD has nothing else to cover an LDS round trip
                    // (one row ahead = 10 FMAs = 40 cycles against > 100)
                    constexpr int RING = 4;
                    double gb[RING][CH];
                    auto load_unit = [&](int r, double (&dst)[CH]) {
                        const int i = wave * UPW + r < NU ? wave * UPW + r : NU - 1;
                        load_run<CH>(kb + i * SG + c0, dst);                                      // c0, SG even: ds_read_b128
                    };
#pragma unroll
                    for (int r = 0; r < RING - 1 && r < UPW; ++r) load_unit(r, gb[r]);
#pragma unroll
                    for (int r = 0; r < UPW; ++r) {
                        if (r + RING - 1 < UPW) load_unit(r + RING - 1, gb[(r + RING - 1) % RING]);
                        pin_regs(gb[r % RING]);
#pragma unroll
                        for (int j = 0; j < CH; ++j) acc[r] = fma(gb[r % RING][j], dx[j], acc[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < UPW; ++r)
                    if (wave * UPW + r < NU) U[wave * UPW + r] = acc[r];
            }
        }
        __syncthreads();
        SDDP_TICK(10)
        // ---- C: wave 0 steps the model; the other waves write the stored lane's u_k and stage the next knot
        if (wave == 0) {
            // x_{k+1} = f(x_k, u_k) - (1 - alpha) d_k straight into the X columns (the step forms x+ before its first store)
            if (OPEN_LOOP) J += M::step(c, X, U, sb + L::SB_P, k, X);
            else J += M::step(c, X, U, sb + L::SB_P, k, LdsColClose{X.p, sb + L::SB_D, oma});
        } else {
            if (k + 1 < N) {                         // (loads before the store below: a store in flight would be waited for with them -- vmcnt counts both)
                if (!kEarlyFetch) fetch_knot(k + 1);
                put_knot(k + 1);
            }
            if (tid - kWave < NU) un[k * NU + tid - kWave] = s[L::RO_U + (tid - kWave) * kWave + store_lane];
        }
        SDDP_TICK(12)
    }
    if (wave == 0) {
        J += M::term_cost(c, X, P + N * NP);
        if (lane == store_lane) {
            for (int i = 0; i < NX; ++i) xn[N * NX + i] = X[i];
        }
    }
    return J;
}

// fused persistent solve, 4 waves per instance
template <class M, bool SINK>
__device__ __forceinline__ void solve_instance_mw(const SolveArgs& A, double* s, const int b, const int slot) {
    using L = LdsMW<M>;
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int N = A.N;
    const sddp_options& o = A.o;
    const double* x0 = A.x0 + size_t(b) * NX;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    double* xn = A.xn + size_t(slot) * (N + 1) * NX;      // work buffers: per slot (SolveArgs)
    double* un = A.un + size_t(slot) * N * NU;
    double* dft = A.dft + size_t(slot) * N * NX;
    double* gains = A.gains + size_t(slot) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(slot) * (N + 1) * NREC;

    double J = 0.0, gap = 0.0;
    SDDP_T_DECL
    sweep_tables_mw<M>(A.c, s, tid);
    double qconst[LdsMW<M>::TQ][3][3];
    mw_const_block<M>(A.c, tid, qconst);
    // ---- starting point (cost and defect norm computed by wave 0, shared through CTL)
    if (o.initial_rollout) {
        J = rollout_mw<M, true, SINK>(A.c, N, x0, P, xs, us, dft, gains, xn, un, 0.0, 0, tid, s SDDP_T_PASS);
        __syncthreads();
        for (int e = tid; e < (N + 1) * NX; e += kThreadsMW) xs[e] = xn[e];
        for (int e = tid; e < N * NX; e += kThreadsMW) dft[e] = 0.0;
    } else {
        if (tid < NX) xs[tid] = x0[tid];
        __syncthreads();
        if (wave == 0) phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
    }
    if (tid == 0) { s[L::CTL + 12] = J; s[L::CTL + 13] = gap; }
    __syncthreads();
    J = s[L::CTL + 12];
    gap = s[L::CTL + 13];
    __syncthreads();
    double mu = o.mu0, rho = 0.0, alpha = 0.0, expected = 0.0, theta = 0.0;
    int iters = 0, converged = 0, status = 1, rollouts = 0, guess = 0;
    if (!(fabs(J) < 1e300)) { status = 3; }
    else
        while (iters < o.max_iters) {
            SDDP_TICK(9)
            const bool work_dirty = iters > 0 || o.initial_rollout;        // a forward pass used the whole work area:
            if (work_dirty) {                                              // zero it, then the constants of F~^T by waves 1..3
                zero_work_mw<M>(s, tid, kThreadsMW);                       // while wave 0 (one lane per knot) differentiates
                __syncthreads();
            }
            if (wave == 0) phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
            else if (work_dirty) ft_constants_mw<M>(A.c, s, tid - kWave, kThreadsMW - kWave);
            __syncthreads();
            SDDP_TICK(0)
            double dV1, G1, G2, qu_inf, a_win = 0.0, J_win = 0.0;
            bool ok = true, stop = false, accepted = false;
            do {   // at most twice: a failed sweep / line search with the second-order term is redone without it
                while (true) {
                    ok = backward_sweep_mw<M, SINK>(A.c, N, P, dft, rec, gains, mu, theta, s, tid, dV1, G1, G2, qu_inf, qconst SDDP_T_PASS);
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
                bool tiles_dirty = false;
                double a_base = o.alpha_0;
                // Wide models: ONE call site of the pass (a pass that only has to store another lane's trajectory is a second trip of
                // this loop).  With two, the compiler keeps the pass of srbd61 out of line: a real call, generic pointers (flat loads),
                // the constants by reference through scratch -- 28.9 -> 31.8 k solves/s.  The narrow models keep their two inlined
                // copies (one call site costs them 2 %).
                constexpr bool kOneCallSite = M::NX > 40;
                int reroll = -1;
                while (a_base >= o.alpha_converge_threshold) {
                    double a = a_base;
                    for (int j = 0; j < lane; ++j) a *= o.line_search_decrease_factor;
                    const bool valid = a >= o.alpha_converge_threshold;
                    SDDP_TICK(9)
                    const double Jl = rollout_mw<M, false, SINK>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a,
                                                                 kOneCallSite && reroll >= 0 ? reroll : guess, tid, s SDDP_T_PASS);
                    SDDP_TICK(8)
                    tiles_dirty = true;
                    ++rollouts;
                    if (kOneCallSite && reroll >= 0) { guess = reroll; accepted = true; break; }
                    if (wave == 0) {
                        const double pred = a * A1 + a * a * B2 - a * rho * gap;
                        const double dphi = (Jl + rho * (1.0 - a) * gap) - (J + rho * gap);
                        const bool good = valid && (fabs(Jl) < 1e300) && (dphi <= o.beta * pred + slack);
                        const unsigned long long mask = __ballot(good);
                        const int win = mask ? __ffsll((long long)mask) - 1 : -1;
                        const double jw = __shfl(Jl, win < 0 ? 0 : win, kWave);
                        if (lane == 0) { s[L::CTL + 12] = double(win); s[L::CTL + 13] = jw; }
                    }
                    __syncthreads();
                    const int win = int(s[L::CTL + 12]);
                    if (win >= 0) {
                        a_win = __shfl(a, win, kWave);
                        J_win = s[L::CTL + 13];
                        if (win != guess) {   // the accepted lane's trajectory was not the one stored: roll it again
                            if constexpr (kOneCallSite) { reroll = win; continue; }
                            else {
                                rollout_mw<M, false, SINK>(A.c, N, x0, P, xs, us, dft, gains, xn, un, a, win, tid, s SDDP_T_PASS);
                                ++rollouts;
                            }
                        }
                        guess = win;
                        accepted = true;
                        break;
                    }
                    a_base = __shfl(a, kWave - 1, kWave) * o.line_search_decrease_factor;
                    guess = 0;
                }
                if (!accepted && theta != 0.0) {   // fall back to the plain Gauss-Newton sweep once
                    theta = 0.0;
                    if (tiles_dirty) {
                        __syncthreads();
                        zero_work_mw<M>(s, tid, kThreadsMW);
                        __syncthreads();
                        ft_constants_mw<M>(A.c, s, tid, kThreadsMW);
                        __syncthreads();
                    }
                    continue;
                }
                break;
            } while (true);
            if (stop) break;
            if (!accepted) {   // alpha below the threshold: stop; converged only with closed gaps and no predicted decrease (status 4)
                alpha = 0.0;
                converged = (gap <= o.gap_tol && expected <= o.cost_reduction_ths * fmax(1.0, fabs(J))) ? 1 : 0;
                status = converged ? 0 : 4;
                break;
            }
            alpha = a_win;
            const double dJ = J - J_win;
            J = J_win;
            { double* t = xs; xs = xn; xn = t; }
            { double* t = us; us = un; un = t; }
            const double oma = 1.0 - alpha;
            __syncthreads();
            for (int e = tid; e < N * NX; e += kThreadsMW) dft[e] *= oma;
            gap *= oma;
            ++iters;
            theta = (o.second_order && alpha == o.alpha_0) ? 1.0 : 0.0;
            if (mu > o.mu0) mu = fmax(o.mu0, mu * 0.1);
            __syncthreads();
            if (fabs(dJ) < o.cost_reduction_ths && gap <= o.gap_tol) { converged = 1; status = 0; break; }
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
        st.cost = J; st.alpha = alpha; st.gap = gap; st.mu = mu; st.expected = expected; st.rho = rho;
        st.iters = iters; st.converged = converged; st.status = status; st.rollouts = rollouts;
        A.stats[b] = st;
        A.hist[b] = iters;
    }
    __syncthreads();
}

// work queue over the resident workgroups (see solve_queue in sddp_kernels.hpp); the queue position travels through LDS
template <class M, bool SINK>
__device__ __forceinline__ void solve_queue_mw(const SolveArgs& A, double* s) {
    // the queue position lives in a control word of the dynamic LDS block (CTL + 15), so that the occupancy query and the
    // dynamic-LDS attribute cover every byte of LDS the kernel uses
    int* q_pos = reinterpret_cast<int*>(s + LdsMW<M>::CTL + 15);
    const int slot = blockIdx.x;
    const bool queued = A.qhead != nullptr;
    if (threadIdx.x == 0) A.slot_clock(slot)[0] = wall_clock64();
    if (queued && threadIdx.x == 0) *q_pos = atomicAdd(A.qhead, 1);
    __syncthreads();
    int i = queued ? *q_pos : slot;
    __syncthreads();                                   // every thread has read it before the solve re-zeroes the LDS block
    while (i < A.count) {
        const int b = (queued && A.order) ? A.order[i] : A.first + i;
        solve_instance_mw<M, SINK>(A, s, b, slot);           // ends with a barrier: q_pos may be rewritten
        if (!queued) break;
        if (threadIdx.x == 0) *q_pos = atomicAdd(A.qhead, 1);
        __syncthreads();
        i = *q_pos;
        __syncthreads();
    }
    if (threadIdx.x == 0) A.slot_clock(slot)[1] = wall_clock64();
}
template <class M>
__global__ __launch_bounds__(kThreadsMW) void solve_kernel_mw(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    solve_queue_mw<M, mw_sink<M>(false)>(A, s);
}

// the same body capped at half the register file: two workgroups per CU where the tiles of two instances fit its LDS;
// sddp_options.waves_per_simd = 2 picks it, results are identical
template <class M>
__global__ __launch_bounds__(kThreadsMW) __attribute__((amdgpu_waves_per_eu(SDDP_MW_W2_WAVES))) void solve_kernel_mw_w2(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    solve_queue_mw<M, mw_sink<M>(true)>(A, s);
}

template <class M>
__global__ __launch_bounds__(kThreadsMW) void backward_kernel_mw(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NREC = M::NREC;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (b >= A.B) return;
    const int N = A.N;
    const double* P = A.P + size_t(b) * (N + 1) * NP;
    double* xs = A.xs + size_t(b) * (N + 1) * NX;
    double* us = A.us + size_t(b) * N * NU;
    double* dft = A.dft + size_t(b) * N * NX;
    double* gains = A.gains + size_t(b) * N * (NU * (NX + 1));
    double* rec = A.rec + size_t(b) * (N + 1) * NREC;
    double J = 0.0, gap = 0.0;
    SDDP_T_DECL
    sweep_tables_mw<M>(A.c, s, tid);
    if (wave == 0) {
        phase_defects<M>(A.c, N, xs, us, P, dft, lane, J, gap);
        phase_derivs<M>(A.c, N, xs, us, P, rec, lane);
    }
    __syncthreads();
    double dV1, G1, G2, qu_inf;
    double qconst[LdsMW<M>::TQ][3][3];
    mw_const_block<M>(A.c, tid, qconst);
    const bool ok = backward_sweep_mw<M, mw_sink<M>(false)>(A.c, N, P, dft, rec, gains, A.mu, A.alpha, s, tid, dV1, G1, G2, qu_inf, qconst SDDP_T_PASS);
    if (tid == 0) {
        double* sc = A.scal + size_t(b) * kScal;
        sc[0] = dV1; sc[1] = -0.5 * dV1; sc[2] = G1; sc[3] = G2; sc[4] = ok ? 1.0 : 0.0; sc[5] = A.mu; sc[6] = qu_inf; sc[7] = J;
    }
}

template <class M>
__global__ __launch_bounds__(kThreadsMW) void forward_kernel_mw(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double s[];
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N;
    SDDP_T_DECL
    const double J = rollout_mw<M, false, mw_sink<M>(false)>(A.c, N, A.x0 + size_t(b) * NX, A.P + size_t(b) * (N + 1) * NP,
                                          A.xs + size_t(b) * (N + 1) * NX, A.us + size_t(b) * N * NU,
                                          A.dft + size_t(b) * N * NX, A.gains + size_t(b) * N * (NU * (NX + 1)),
                                          A.xn + size_t(b) * (N + 1) * NX, A.un + size_t(b) * N * NU, A.alpha, 0, tid, s SDDP_T_PASS);
    if (tid == 0) A.scal[size_t(b) * kScal] = J;
}

}  // namespace sddp
