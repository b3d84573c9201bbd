/* TEST INFRASTRUCTURE ONLY -- plain C restatement of the four problems (srbd13, srbd37, srbd61, lip30) and, through ddp_engine.inc, of
 * the MS-DDP iteration (same steps, same order as oracle/models.py + oracle/ddp.py, which carry the reference file:line
 * citations).
 * PARITY UNPINNED upstream (the reference engine `pyddp` is absent): this file is pinned against the numpy oracle in
 * tests/test_oracle_c.py.  Used for (1) the `cpu_baseline` leg of bench.py (kind "port"), (2) large-batch / long-horizon parity
 * checks.  Never linked into or called from the product (srbd_horizon_amd/).
 *
 * Reference lines restated: dynamics prb.py:92-110 (fSRBD, element-wise inertia prb.py:99, Euler ddp.py:228-230);
 * costs and penalties prb.py:166-204 through ddp.py:179-226; LIP prb.py:315-328, :379-402; solver options ddp.py:14-35.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GRAV 9.81
#define CW 1e6 /* equality-constraint weight, ddp.py:181 */

typedef struct {
    double dt, inv_ms, Is[9], com_z, w_rz, w_rd, w_w, w_f, w_sw, gq, lever;
    double mu_lin, bar_w, bar_s;   /* friction-cone exponential barrier (oracle/models.py _force_rows): off when bar_w == 0 */
    double w_rel, w_zmp, lip_h, feet[12];
    int inertia_mode;
    double box_w, box_s, lower[64], upper[64];   /* bound barrier (oracle/models.py _bound_rows, ddp.py:203-208): off when box_w == 0 */
    int rv;                        /* relative-velocity constraints inside a foot present (prb.py:166: `if contact_model > 1`) */
    /* user-declared linear residual rows (oracle/models.py WithLinearRows): xr_on: the parameter vector is NXR columns wider and
     * row j = sqrt(xw[j]) (xa[j] . z - p[np + j] - xc[j]), kind 0 "state" (nodes 1..N) / 1 "stage" (nodes 0..N-1); nz entries of xa used */
    int xr_on, xr_n, xkind[8];
    double xw[8], xc[8], xa[8][128];
} consts_t;
#define NXR 8

/* packed constants from Python: m, I[9], com_z, dt, force_scaling, r_gain, rdot_gain, w_gain, fsw, qddot, minf, inertia_mode, lever,
 * friction_cone_coefficient, friction_barrier_weight, friction_barrier_sharpness, rel_pos_gain, zmp_gain, lip_height, feet[12],
 * bound_barrier_weight, bound_barrier_sharpness, lower[64], upper[64], relative_velocity_constraints,
 * xr_on, xr_n, then per extra row (8 of them): kind, w, const, a[128] */
static void unpack_consts(const double* c, consts_t* k) {
    const double m = c[0], fs = c[12];
    k->inv_ms = fs / m;
    for (int i = 0; i < 9; ++i) k->Is[i] = c[1 + i] / fs;
    k->com_z = c[10]; k->dt = c[11];
    k->w_rz = c[13]; k->w_rd = c[14]; k->w_w = c[15];
    k->w_sw = fs * fs * c[16]; k->gq = c[17]; k->w_f = fs * fs * c[18];
    k->inertia_mode = (int)c[19]; k->lever = c[20];
    k->mu_lin = c[21] / sqrt(2.0); k->bar_w = c[22]; k->bar_s = c[23];
    k->w_rel = c[24]; k->w_zmp = c[25]; k->lip_h = c[26];
    for (int i = 0; i < 12; ++i) k->feet[i] = c[27 + i];
    k->box_w = c[39]; k->box_s = c[40];
    for (int i = 0; i < 64; ++i) { k->lower[i] = c[41 + i]; k->upper[i] = c[105 + i]; }
    k->rv = c[169] != 0.0;
    k->xr_on = c[170] != 0.0; k->xr_n = (int)c[171];
    for (int j = 0; j < 8; ++j) {
        const double* q = c + 172 + j * 131;
        k->xkind[j] = (int)q[0]; k->xw[j] = q[1]; k->xc[j] = q[2];
        for (int i = 0; i < 128; ++i) k->xa[j][i] = q[3 + i];
    }
}

/* the user rows of node k (u == NULL: terminal) appended to r / J (row stride nz) from row n on; pref = the node's NXR reference columns */
static int xr_rows(const consts_t* c, int nx, int nu, const double* x, const double* u, const double* pref, int k, double* r, double* J, int nz, int n) {
    if (!c->xr_on) return n;
    for (int j = 0; j < c->xr_n; ++j) {
        const int active = c->xkind[j] == 0 ? (k >= 1) : (u != NULL);
        if (!active) continue;
        const double g = sqrt(c->xw[j]);
        double e = -pref[j] - c->xc[j];
        for (int i = 0; i < nx; ++i) e += c->xa[j][i] * x[i];
        if (u) for (int i = 0; i < nu; ++i) e += c->xa[j][nx + i] * u[i];
        r[n] = g * e;
        if (J) { for (int i = 0; i < nz; ++i) J[n * nz + i] = (i < nx || u) ? g * c->xa[j][i] : 0.0; }
        ++n;
    }
    return n;
}
static void cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void skew(const double* v, double* S) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void mm3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0; for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}
static void mv3(const double* A, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
/* LU solve of a 3x3 system with partial pivoting (numpy.linalg.solve analogue); returns inverse */
static void inv3(const double* M, double* o) {
    double a[3][6];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = M[3 * i + j]; a[i][3 + j] = i == j; }
    for (int p = 0; p < 3; ++p) {
        int r = p; for (int i = p + 1; i < 3; ++i) if (fabs(a[i][p]) > fabs(a[r][p])) r = i;
        if (r != p) for (int j = 0; j < 6; ++j) { double t = a[p][j]; a[p][j] = a[r][j]; a[r][j] = t; }
        const double d = 1.0 / a[p][p];
        for (int j = 0; j < 6; ++j) a[p][j] *= d;
        for (int i = 0; i < 3; ++i) if (i != p) { const double f = a[i][p]; for (int j = 0; j < 6; ++j) a[i][j] -= f * a[p][j]; }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[3 * i + j] = a[i][3 + j];
}
static void quat_to_rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void quat_to_rot_d(const double* q, int a, double* D) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double t[4][9] = {{0, 2 * y, 2 * z, 2 * y, -4 * x, -2 * w, 2 * z, 2 * w, -4 * x},
                            {-4 * y, 2 * x, 2 * w, 2 * x, 0, 2 * z, -2 * w, 2 * z, -4 * y},
                            {-4 * z, -2 * w, 2 * x, 2 * w, -4 * z, 2 * y, 2 * x, 2 * y, 0},
                            {0, -2 * z, 2 * y, 2 * z, 0, -2 * x, -2 * y, 2 * x, 0}};
    memcpy(D, t[a], sizeof(double) * 9);
}
static void world_inertia(const consts_t* c, const double* R, double* M) {
    if (c->inertia_mode == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[3 * i + j] = R[3 * i + j] * c->Is[3 * i + j] * R[3 * j + i]; }
    else { double T[9], Rt[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i]; mm3(c->Is, Rt, T); mm3(R, T, M); }
}
static void world_inertia_d(const consts_t* c, const double* R, const double* dR, double* dM) {
    if (c->inertia_mode == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dM[3 * i + j] = c->Is[3 * i + j] * (dR[3 * i + j] * R[3 * j + i] + R[3 * i + j] * dR[3 * j + i]); }
    else { double T[9], U[9], Rt[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
        mm3(c->Is, Rt, T); mm3(dR, T, U); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dM[3 * i + j] = U[3 * i + j] + U[3 * j + i]; }
}

/* Cholesky of an n x n SPD matrix (lower, in place); returns 0 on failure */
static int chol(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j]; for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) { double s = A[i * n + j]; for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k]; A[i * n + j] = s / d; }
    }
    return 1;
}

/* ---- SRBD accelerations for nc contacts (Horizon kin_dyn.fSRBD, prb.py:99; oracle/models.py srbd_acc / srbd_acc_jac) ---- */
typedef struct { double R[9], M[9], Mi[9], wdot[3], rddot[3]; } core_t;
typedef struct { double Wr[9], Wo[12], Ww[9], Wc[8][9], Wf[8][9]; } corejac_t;   /* d wdot / d r, o, w, c_i, f_i */

static void core_n(const consts_t* c, const double* r, const double* o, const double* w, int nc, const double* const* cs,
                   const double* const* fs, core_t* k) {
    quat_to_rot(o, k->R); world_inertia(c, k->R, k->M); inv3(k->M, k->Mi);
    double tau[3] = {0, 0, 0}, fsum[3] = {0, 0, 0}, Mw[3], g[3];
    for (int i = 0; i < nc; ++i) {
        const double* cp = cs[i]; const double* f = fs[i];
        double l[3] = {cp[0] - r[0], cp[1] - r[1], cp[2] - r[2]}, t[3];
        cross(l, f, t);
        for (int a = 0; a < 3; ++a) { tau[a] += c->lever * t[a]; fsum[a] += f[a]; }
    }
    mv3(k->M, w, Mw); cross(w, Mw, g);
    for (int a = 0; a < 3; ++a) tau[a] -= g[a];
    mv3(k->Mi, tau, k->wdot);
    k->rddot[0] = fsum[0] * c->inv_ms; k->rddot[1] = fsum[1] * c->inv_ms; k->rddot[2] = fsum[2] * c->inv_ms - GRAV;
}
static void corejac_n(const consts_t* c, const double* r, const double* o, const double* w, int nc, const double* const* cs,
                      const double* const* fs, const core_t* k, corejac_t* J) {
    double S[9], sf[3] = {0, 0, 0};
    for (int i = 0; i < nc; ++i) for (int a = 0; a < 3; ++a) sf[a] += c->lever * fs[i][a];
    skew(sf, S); mm3(k->Mi, S, J->Wr);
    double Mw[3], SMw[9], Sw[9], SwM[9], U[9];
    mv3(k->M, w, Mw); skew(Mw, SMw); skew(w, Sw); mm3(Sw, k->M, SwM);
    for (int i = 0; i < 9; ++i) U[i] = SMw[i] - SwM[i];
    mm3(k->Mi, U, J->Ww);
    for (int q = 0; q < 4; ++q) {
        double dR[9], dM[9], a1[3], b1[3], cr[3], t[3], col[3];
        quat_to_rot_d(o, q, dR); world_inertia_d(c, k->R, dR, dM);
        mv3(dM, k->wdot, a1); mv3(dM, w, b1); cross(w, b1, cr);
        for (int a = 0; a < 3; ++a) t[a] = -(a1[a] + cr[a]);
        mv3(k->Mi, t, col);
        for (int a = 0; a < 3; ++a) J->Wo[a * 4 + q] = col[a];
    }
    for (int i = 0; i < nc; ++i) {
        double l[3] = {c->lever * (cs[i][0] - r[0]), c->lever * (cs[i][1] - r[1]), c->lever * (cs[i][2] - r[2])};
        skew(l, S); mm3(k->Mi, S, J->Wf[i]);
        double nf[3] = {-c->lever * fs[i][0], -c->lever * fs[i][1], -c->lever * fs[i][2]};
        skew(nf, S); mm3(k->Mi, S, J->Wc[i]);
    }
}
/* Second derivatives of wdot = I_w(o)^-1 n(z), n = sum s (c_i - r) x f_i - w x I_w(o) w (oracle/models.py srbd_wdot_hess), from
 * differentiating I_w wdot = n twice:  d_a d_b wdot = I_w^-1 (d_a d_b n - d_a d_b I_w wdot - d_a I_w d_b wdot - d_b I_w d_a wdot).
 * Local variable order z = r(3) | o(4) | w(3) | c_0..(3 each) | f_0..(3 each), n = 10 + 6 nc.  S (n x n) += sum_m lam_m T[m]. */
static void world_inertia_d2(const consts_t* c, const double* R, const double* dRp, const double* dRq, const double* d2R, double* d2M) {
    if (c->inertia_mode == 0) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            d2M[3 * i + j] = c->Is[3 * i + j] * (d2R[3 * i + j] * R[3 * j + i] + dRp[3 * i + j] * dRq[3 * j + i] + dRq[3 * i + j] * dRp[3 * j + i]
                                                  + R[3 * i + j] * d2R[3 * j + i]);
    } else {   /* d2R Is R^T + dRp Is dRq^T + dRq Is dRp^T + R Is d2R^T */
        const double* L[4] = {d2R, dRp, dRq, R}; const double* Rr[4] = {R, dRq, dRp, d2R};
        memset(d2M, 0, sizeof(double) * 9);
        for (int t = 0; t < 4; ++t) {
            double Rt[9], T[9], U[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = Rr[t][3 * j + i];
            mm3(c->Is, Rt, T); mm3(L[t], T, U);
            for (int i = 0; i < 9; ++i) d2M[i] += U[i];
        }
    }
}
static void wdot_hess_contract(const consts_t* c, const double* r, const double* o, const double* w, int nc, const double* const* cs,
                               const double* const* fs, const double* lam, double* S /* n x n, accumulated */) {
    (void)cs; (void)r;
    const int n = 10 + 6 * nc;
    core_t k; corejac_t J;
    core_n(c, r, o, w, nc, cs, fs, &k); corejac_n(c, r, o, w, nc, cs, fs, &k, &J);
    double Jw[3][58];                                   /* d wdot / d z in the local order */
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) { Jw[a][b] = J.Wr[3 * a + b]; Jw[a][7 + b] = J.Ww[3 * a + b]; }
        for (int b = 0; b < 4; ++b) Jw[a][3 + b] = J.Wo[4 * a + b];
        for (int i = 0; i < nc; ++i) for (int b = 0; b < 3; ++b) { Jw[a][10 + 3 * i + b] = J.Wc[i][3 * a + b]; Jw[a][10 + 3 * nc + 3 * i + b] = J.Wf[i][3 * a + b]; }
    }
    double dR[4][9], dM[4][9], d2M[4][4][9];
    for (int q = 0; q < 4; ++q) { quat_to_rot_d(o, q, dR[q]); world_inertia_d(c, k.R, dR[q], dM[q]); }
    for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) {
        double eq[4] = {0, 0, 0, 0}, d2R[9]; eq[q] = 1.0;
        quat_to_rot_d(eq, p, d2R);                       /* dR/dq_p is linear in the quaternion: its q-derivative = dR/dq_p at e_q */
        world_inertia_d2(c, k.R, dR[p], dR[q], d2R, d2M[p][q]);
    }
    double y[3]; mv3(k.Mi, lam, y);                      /* I_w symmetric: lam^T I_w^-1 v = y . v */
    const double s = c->lever;
    for (int a = 0; a < n; ++a) for (int b = 0; b <= a; ++b) {
        double v[3] = {0, 0, 0};
        const int ao = a >= 3 && a < 7, bo = b >= 3 && b < 7, aw = a >= 7 && a < 10, bw = b >= 7 && b < 10;
        const int af = a >= 10 + 3 * nc, bf = b >= 10 + 3 * nc, ac = a >= 10 && !af, bc = b >= 10 && !bf, br = b < 3;
        if (af && (br || bc)) {                          /* bilinear torque: d2/(dc df) = s e_c x e_f, d2/(dr df) = -s e_r x e_f */
            const int i = (a - 10 - 3 * nc) / 3, fa = (a - 10 - 3 * nc) % 3;
            const int ok = br || (b - 10) / 3 == i, xa = br ? b : (b - 10) % 3;
            if (ok) { double ex[3] = {0, 0, 0}, ef[3] = {0, 0, 0}, t[3]; ex[xa] = 1; ef[fa] = 1; cross(ex, ef, t);
                      for (int m = 0; m < 3; ++m) v[m] = (br ? -s : s) * t[m]; }
        } else if (aw && bw) {
            double ea[3] = {0, 0, 0}, eb[3] = {0, 0, 0}, Ma[3], Mb[3], t1[3], t2[3]; ea[a - 7] = 1; eb[b - 7] = 1;
            mv3(k.M, ea, Ma); mv3(k.M, eb, Mb); cross(ea, Mb, t1); cross(eb, Ma, t2);
            for (int m = 0; m < 3; ++m) v[m] = -(t1[m] + t2[m]);
        } else if (aw && bo) {
            double ea[3] = {0, 0, 0}, t0[3], t1[3], t2[3], t3[3]; ea[a - 7] = 1;
            mv3(dM[b - 3], w, t0); cross(ea, t0, t1); mv3(dM[b - 3], ea, t2); cross(w, t2, t3);
            for (int m = 0; m < 3; ++m) v[m] = -(t1[m] + t3[m]);
        } else if (ao && bo) {
            double t0[3], t1[3]; mv3(d2M[a - 3][b - 3], w, t0); cross(w, t0, t1);
            double t2[3]; mv3(d2M[a - 3][b - 3], k.wdot, t2);
            for (int m = 0; m < 3; ++m) v[m] = -t1[m] - t2[m];
        }
        (void)ac;
        if (ao) { double jb[3] = {Jw[0][b], Jw[1][b], Jw[2][b]}, t[3]; mv3(dM[a - 3], jb, t); for (int m = 0; m < 3; ++m) v[m] -= t[m]; }
        if (bo) { double ja[3] = {Jw[0][a], Jw[1][a], Jw[2][a]}, t[3]; mv3(dM[b - 3], ja, t); for (int m = 0; m < 3; ++m) v[m] -= t[m]; }
        const double val = y[0] * v[0] + y[1] * v[1] + y[2] * v[2];
        S[a * n + b] += val; if (a != b) S[b * n + a] += val;
    }
}
/* full second-order correction of a stage node (oracle/models.py _srbd_second_order_full): Q += theta * (sum_m lam_m d2 wdot_m +
 * dt v'_o . d2 odot), lam = dt v'_w + 2 min_qddot_gain wdot; gl maps the local order to z columns (-1: not a variable) */
static void srbd_second_order_full(const consts_t* c, const double* r, const double* o, const double* w, int nc, const double* const* cs,
                                   const double* const* fs, const double* vp_o, const double* vp_w, const int* gl, int o0, int w0,
                                   double theta, double* Q, int nz) {
    const int n = 10 + 6 * nc;
    core_t k; core_n(c, r, o, w, nc, cs, fs, &k);
    double lam[3], S[58 * 58];
    for (int m = 0; m < 3; ++m) lam[m] = c->dt * vp_w[m] + 2.0 * c->gq * k.wdot[m];
    memset(S, 0, sizeof(double) * n * n);
    wdot_hess_contract(c, r, o, w, nc, cs, fs, lam, S);
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (gl[a] >= 0 && gl[b] >= 0) Q[gl[a] * nz + gl[b]] += theta * S[a * n + b];
    if (c->bar_w > 0.0) {   /* friction-cone barrier: exact - Gauss-Newton Hessian = (w s^2 / 2) sum_j e_j a_j a_j^T once more */
        const double ml = c->mu_lin;
        const double A[5][3] = {{1, 0, -ml}, {-1, 0, -ml}, {0, 1, -ml}, {0, -1, -ml}, {0, 0, -1}};
        for (int i = 0; i < nc; ++i) {
            const int f0 = gl[10 + 3 * nc + 3 * i];
            for (int j = 0; j < 5; ++j) {
                const double e = c->bar_w * exp(c->bar_s * (A[j][0] * fs[i][0] + A[j][1] * fs[i][1] + A[j][2] * fs[i][2]));
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Q[(f0 + a) * nz + f0 + b] += theta * 0.5 * c->bar_s * c->bar_s * e * A[j][a] * A[j][b];
            }
        }
    }
    for (int cc = 0; cc < 3; ++cc) {                     /* odot bilinear in (o, w): d(d odot/d o)/dw_c = Jo at w = e_c */
        double e[3] = {0, 0, 0}; e[cc] = 1.0;
        const double Jo[16] = {0, -0.5 * e[2], 0.5 * e[1], 0.5 * e[0], 0.5 * e[2], 0, -0.5 * e[0], 0.5 * e[1],
                               -0.5 * e[1], 0.5 * e[0], 0, 0.5 * e[2], -0.5 * e[0], -0.5 * e[1], -0.5 * e[2], 0};
        for (int b = 0; b < 4; ++b) {
            double v = 0; for (int a = 0; a < 4; ++a) v += vp_o[a] * Jo[4 * a + b];
            v *= theta * c->dt;
            Q[(o0 + b) * nz + w0 + cc] += v; Q[(w0 + cc) * nz + o0 + b] += v;
        }
    }
}
/* quaternion kinematics odot = 1/2 [w;0] (x) o and its Jacobians (prb.py:107-108) */
static void quat_step(const double* o, const double* w, double dt, double* on) {
    double wxo[3]; cross(w, o, wxo);
    for (int a = 0; a < 3; ++a) on[a] = o[a] + dt * 0.5 * (o[3] * w[a] + wxo[a]);
    on[3] = o[3] - dt * 0.5 * (w[0] * o[0] + w[1] * o[1] + w[2] * o[2]);
}
static void quat_jac(const double* o, const double* w, double dt, double* F, int nz, int O0, int W0) {
    const double Jo[16] = {0, -0.5 * w[2], 0.5 * w[1], 0.5 * w[0], 0.5 * w[2], 0, -0.5 * w[0], 0.5 * w[1],
                           -0.5 * w[1], 0.5 * w[0], 0, 0.5 * w[2], -0.5 * w[0], -0.5 * w[1], -0.5 * w[2], 0};
    const double Jw[12] = {0.5 * o[3], 0.5 * o[2], -0.5 * o[1], -0.5 * o[2], 0.5 * o[3], 0.5 * o[0],
                           0.5 * o[1], -0.5 * o[0], 0.5 * o[3], -0.5 * o[0], -0.5 * o[1], -0.5 * o[2]};
    for (int a = 0; a < 4; ++a) {
        for (int b = 0; b < 4; ++b) F[(O0 + a) * nz + O0 + b] += dt * Jo[4 * a + b];
        for (int b = 0; b < 3; ++b) F[(O0 + a) * nz + W0 + b] += dt * Jw[3 * a + b];
    }
}
/* state residual rows of nodes 1..ns (prb.py:184-191; oracle/models.py _srbd_state_rows): 11 rows */
static int srbd_state_rows(const consts_t* c, const double* x, int R0, int O0, int RD0, int W0, const double* rdot_ref, const double* w_ref,
                           double otg, const double* q, double* r, double* J, int nz, int n) {
    const double g = sqrt(c->w_rz);
    r[n] = g * (x[R0 + 2] - c->com_z); if (J) J[n * nz + R0 + 2] = g; ++n;
    const double* o = x + O0;
    double oxq[3]; cross(o, q, oxq);
    for (int a = 0; a < 3; ++a) r[n + a] = otg * (o[3] * q[a] + q[3] * o[a] + oxq[a]);
    r[n + 3] = otg * (o[3] * q[3] - (o[0] * q[0] + o[1] * q[1] + o[2] * q[2]) - 1.0);
    if (J) {
        double Sq[9]; skew(q, Sq);
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) J[(n + a) * nz + O0 + b] = otg * ((a == b ? q[3] : 0.0) - Sq[3 * a + b]);
            J[(n + a) * nz + O0 + 3] = otg * q[a];
            J[(n + 3) * nz + O0 + a] = -otg * q[a];
        }
        J[(n + 3) * nz + O0 + 3] = otg * q[3];
    }
    n += 4;
    const double gd = sqrt(c->w_rd), gw = sqrt(c->w_w);
    for (int a = 0; a < 3; ++a) { r[n + a] = gd * (x[RD0 + a] - rdot_ref[a]); if (J) J[(n + a) * nz + RD0 + a] = gd; }
    n += 3;
    for (int a = 0; a < 3; ++a) { r[n + a] = gw * (x[W0 + a] - w_ref[a]); if (J) J[(n + a) * nz + W0 + a] = gw; }
    return n + 3;
}
/* min_f_i, f_i_active (prb.py:202-204) and the optional friction-cone barrier rows; ucol = column of f in z = [x u] */
static int force_rows(const consts_t* c, const double* f, double sw, int ucol, double* r, double* J, int nz, int n) {
    const double g1 = sqrt(c->w_f), g2 = sqrt(c->w_sw) * (1.0 - sw);
    for (int a = 0; a < 3; ++a) { r[n + a] = g1 * f[a]; if (J) J[(n + a) * nz + ucol + a] = g1; }
    n += 3;
    for (int a = 0; a < 3; ++a) { r[n + a] = g2 * f[a]; if (J) J[(n + a) * nz + ucol + a] = g2; }
    n += 3;
    if (c->bar_w > 0.0) {   /* r_j = sqrt(w) exp(s a_j.f / 2), rows of the linearised cone A f <= 0 */
        const double ml = c->mu_lin;
        const double A[5][3] = {{1, 0, -ml}, {-1, 0, -ml}, {0, 1, -ml}, {0, -1, -ml}, {0, 0, -1}};
        for (int j = 0; j < 5; ++j) {
            const double gj = A[j][0] * f[0] + A[j][1] * f[1] + A[j][2] * f[2];
            const double rj = sqrt(c->bar_w) * exp(0.5 * c->bar_s * gj);
            r[n + j] = rj;
            if (J) for (int a = 0; a < 3; ++a) J[(n + j) * nz + ucol + a] = 0.5 * c->bar_s * rj * A[j][a];
        }
        n += 5;
    }
    return n;
}
/* opt-in exponential barrier on the bounds of z = [x u] (ddp.py:203-208, commented out upstream): one row per finite bound,
 * r = sqrt(w) exp(s (z_j - ub_j) / 2) resp. sqrt(w) exp(s (lb_j - z_j) / 2), upper before lower (oracle/models.py _bound_rows) */
static int bound_rows(const consts_t* c, const double* x, const double* u, int nx, int nu, double* r, double* J, int n) {
    if (!(c->box_w > 0.0)) return n;
    const int nz = nx + nu;
    const double sw = sqrt(c->box_w), hs = 0.5 * c->box_s;
    for (int j = 0; j < nz; ++j) {
        const double z = j < nx ? x[j] : u[j - nx];
        if (isfinite(c->upper[j])) { r[n] = sw * exp(hs * (z - c->upper[j])); if (J) J[n * nz + j] = hs * r[n]; ++n; }
        if (isfinite(c->lower[j])) { r[n] = sw * exp(hs * (c->lower[j] - z)); if (J) J[n * nz + j] = -hs * r[n]; ++n; }
    }
    return n;
}
/* exact minus Gauss-Newton Hessian of the bound barrier = its (diagonal) Gauss-Newton Hessian once more */
static void bound_second_order(const consts_t* c, const double* x, const double* u, int nx, int nu, double theta, double* Q) {
    if (!(c->box_w > 0.0)) return;
    const int nz = nx + nu;
    for (int j = 0; j < nz; ++j) {
        const double z = j < nx ? x[j] : u[j - nx];
        double e = 0.0;
        if (isfinite(c->upper[j])) e += exp(c->box_s * (z - c->upper[j]));
        if (isfinite(c->lower[j])) e += exp(c->box_s * (c->lower[j] - z));
        Q[j * nz + j] += theta * 0.5 * c->box_w * c->box_s * c->box_s * e;
    }
}
/* rel_pos_{y,x}_1_4 and _3_6 (prb.py:192-199), d1 = p2 - p0, d2 = p3 - p1 (prb.py:153-154): 4 rows */
static int rel_pos_rows(const consts_t* c, const double* x, const int* C0, double* r, double* J, int nz, int n) {
    const double g = sqrt(c->w_rel);
    const int pa[2] = {0, 1}, pb[2] = {2, 3};
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 2; ++e) {
        const int comp = e == 0 ? 1 : 0, a = pa[t], b = pb[t];          /* y first, then x */
        const double d = c->feet[3 * b + comp] - c->feet[3 * a + comp];
        r[n] = g * (-x[C0[a] + comp] + x[C0[b] + comp] - d);
        if (J) { J[n * nz + C0[a] + comp] = -g; J[n * nz + C0[b] + comp] = g; }
        ++n;
    }
    return n;
}
/* equality constraints as sqrt(1e6)-weighted residuals (ddp.py:195-196; prb.py:166-170, :179-181) for nc contacts, contact_model
 * cm = nc / 2 per foot: relative_vel_left_i (cdot_0 - cdot_i, i = 1..cm-1), relative_vel_right_i (cdot_cm - cdot_i, i = cm+1..2cm-1),
 * then per contact cz_tracking_i and cdotxy_tracking_i: 4 (cm - 1) + 3 nc rows */
static int contact_penalty_rows(const double* x, int nc, const int* C0, const int* CD0, const double* cref, const double* sw, double* r,
                                double* J, int nz, int n, int rv) {
    const double g = sqrt(CW), grv = rv ? g : 0.0;     /* rv == 0 (contact_model = 1): the rows stay, as zeros */
    const int cm = nc / 2;
    for (int leg = 0; leg < 2; ++leg)
        for (int i = 1; i < cm; ++i) {
            const int lead = leg * cm, foll = leg * cm + i;
            for (int e = 0; e < 2; ++e) {
                r[n + e] = grv * (x[CD0[lead] + e] - x[CD0[foll] + e]);
                if (J) { J[(n + e) * nz + CD0[lead] + e] = grv; J[(n + e) * nz + CD0[foll] + e] = -grv; }
            }
            n += 2;
        }
    for (int i = 0; i < nc; ++i) {
        r[n] = g * (x[C0[i] + 2] - cref[i]); if (J) J[n * nz + C0[i] + 2] = g; ++n;                 /* cz_tracking_i */
        for (int e = 0; e < 2; ++e) { r[n + e] = g * sw[i] * x[CD0[i] + e]; if (J) J[(n + e) * nz + CD0[i] + e] = g * sw[i]; }
        n += 2;                                                                                      /* cdotxy_tracking_i */
    }
    return n;
}

/* =================================================== srbd13 ===========================================================
 * x = r|o|rdot|w, u = f_L|f_R, p = rdot_ref(3) | w_ref(3) | otg | oref(4) | cL(3) | cR(3) | swL | swR   */
#define NX 13
#define NU 6
#define NP 19
#define NR 77 /* residual rows of a stage node: 11 state + 18 input (+ 10 friction-cone barrier rows, + 2 x 19 bound rows when enabled) */
#define MDL(n) s13_##n
static void s13_core(const consts_t* c, const double* x, const double* u, const double* p, core_t* k, corejac_t* J) {
    const double* cs[2] = {p + 11, p + 14}; const double* fs[2] = {u, u + 3};
    core_n(c, x, x + 3, x + 10, 2, cs, fs, k);
    if (J) corejac_n(c, x, x + 3, x + 10, 2, cs, fs, k, J);
}
static void s13_dyn_k(const consts_t* c, const double* x, const core_t* kp, double* xn) {
    const core_t k = *kp;
    const double dt = c->dt;
    double on[4]; quat_step(x + 3, x + 10, dt, on);
    for (int a = 0; a < 3; ++a) {
        const double rd = x[7 + a], w = x[10 + a];
        xn[a] = x[a] + dt * rd;
        xn[7 + a] = rd + dt * k.rddot[a];
        xn[10 + a] = w + dt * k.wdot[a];
    }
    for (int a = 0; a < 4; ++a) xn[3 + a] = on[a];
}
static void s13_dyn(const consts_t* c, const double* x, const double* u, const double* p, double* xn) {
    core_t k; s13_core(c, x, u, p, &k, NULL);
    s13_dyn_k(c, x, &k, xn);
}
static void s13_wrows(const corejac_t* Jc, double g, double* M, int nz, int row0) {   /* rows of d wdot/dz scaled by g */
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) { M[(row0 + a) * nz + b] += g * Jc->Wr[3 * a + b]; M[(row0 + a) * nz + 10 + b] += g * Jc->Ww[3 * a + b]; }
        for (int b = 0; b < 4; ++b) M[(row0 + a) * nz + 3 + b] += g * Jc->Wo[4 * a + b];
        for (int i = 0; i < 2; ++i) for (int b = 0; b < 3; ++b) M[(row0 + a) * nz + 13 + 3 * i + b] += g * Jc->Wf[i][3 * a + b];
    }
}
static void s13_dyn_jac(const consts_t* c, const double* x, const double* u, const double* p, double* F /*13x19*/) {
    core_t k; corejac_t Jc; s13_core(c, x, u, p, &k, &Jc);
    const double dt = c->dt; const int nz = 19;
    memset(F, 0, sizeof(double) * 13 * nz);
    for (int i = 0; i < 13; ++i) F[i * nz + i] = 1.0;
    for (int a = 0; a < 3; ++a) {
        F[a * nz + 7 + a] += dt;
        for (int i = 0; i < 2; ++i) F[(7 + a) * nz + 13 + 3 * i + a] = dt * c->inv_ms;
    }
    s13_wrows(&Jc, dt, F, nz, 10);
    quat_jac(x + 3, x + 10, dt, F, nz, 3, 10);
}
static int s13_residual_k(const consts_t* c, const double* x, const double* u, const double* p, int k, double* r, double* J, const core_t* pre) {
    int n = 0; const int nz = 19;
    if (J) memset(J, 0, sizeof(double) * NR * nz);
    if (!u || k >= 1) n = srbd_state_rows(c, x, 0, 3, 7, 10, p, p + 3, p[6], p + 7, r, J, nz, n);
    if (u) {
        core_t kk; corejac_t Jc;
        if (pre && !J) kk = *pre; else s13_core(c, x, u, p, &kk, J ? &Jc : NULL);
        const double g = sqrt(c->gq);
        for (int a = 0; a < 3; ++a) { r[n + a] = g * kk.rddot[a]; r[n + 3 + a] = g * kk.wdot[a]; }
        if (J) {
            for (int a = 0; a < 3; ++a) for (int i = 0; i < 2; ++i) J[(n + a) * nz + 13 + 3 * i + a] = g * c->inv_ms;
            s13_wrows(&Jc, g, J, nz, n + 3);
        }
        n += 6;
        for (int i = 0; i < 2; ++i) n = force_rows(c, u + 3 * i, p[17 + i], 13 + 3 * i, r, J, nz, n);
        n = bound_rows(c, x, u, 13, 6, r, J, n);
    }
    return n;
}
static int s13_residual(const consts_t* c, const double* x, const double* u, const double* p, int k, double* r, double* J) {
    return s13_residual_k(c, x, u, p, k, r, J, NULL);
}
/* one rollout knot: cost of node k at (x, u) and x+ = f(x, u), the accelerations evaluated once for both */
static double s13_stepcost(const consts_t* c, const double* x, const double* u, const double* p, int k, double* xn) {
    core_t kk; s13_core(c, x, u, p, &kk, NULL);
    double r[NR]; const int n = s13_residual_k(c, x, u, p, k, r, NULL, &kk);
    double s = 0; for (int i = 0; i < n; ++i) s += r[i] * r[i];
    s13_dyn_k(c, x, &kk, xn);
    return s;
}
/* exact bilinear-torque term: Qux[f_a][r_b] -= theta * s * skew(y)[a][b], y = I_w^-1 (dt v'_w)   */
static void s13_second_order(const consts_t* c, const double* x, const double* u, const double* p, const double* vp, double theta, int mode,
                             double* Q) {
    if (mode == 2) {
        const double* cs[2] = {p + 11, p + 14}; const double* fs[2] = {u, u + 3};
        const int gl[22] = {0, 1, 2, 3, 4, 5, 6, 10, 11, 12, -1, -1, -1, -1, -1, -1, 13, 14, 15, 16, 17, 18};
        srbd_second_order_full(c, x, x + 3, x + 10, 2, cs, fs, vp + 3, vp + 10, gl, 3, 10, theta, Q, 19);
        bound_second_order(c, x, u, 13, 6, theta, Q);
        return;
    }
    double R[9], M[9], Mi[9], lam[3], y[3], S[9];
    quat_to_rot(x + 3, R); world_inertia(c, R, M); inv3(M, Mi);
    for (int a = 0; a < 3; ++a) lam[a] = c->dt * vp[10 + a];
    mv3(Mi, lam, y); skew(y, S);
    for (int i = 0; i < 2; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
        const double v = -theta * c->lever * S[3 * a + b];
        Q[(13 + 3 * i + a) * 19 + b] += v; Q[b * 19 + 13 + 3 * i + a] += v;
    }
}
#include "ddp_engine.inc"
#undef NX
#undef NU
#undef NP
#undef NR
#undef MDL

/* ============================================== srbd37 and srbd61 ======================================================
 * the reference problem with the contacts as states (srbd_cs.inc): nc = 4 (launch file, contact_model = 2) and nc = 8 (the
 * code default contact_model = 4, prb.py:39-41) */
#define NCC 4
#define NR 218 /* 15 state + 18 min_qddot + 24 force (+ 20 barrier) + 16 penalty rows (+ 2 x 61 bound rows) */
#define MDL(n) s37_##n
#include "srbd_cs.inc"
#undef NCC
#undef NR
#undef MDL
#define NCC 8
#define NR 169 /* 15 state + 30 min_qddot + 48 force (+ 40 barrier) + 36 penalty rows (12 relative-velocity + 24); no bound rows */
#define MDL(n) s61_##n
#include "srbd_cs.inc"
#undef NCC
#undef NR
#undef MDL

/* ==================================================== lip30 ===========================================================
 * x = r | c0..3 | rdot | cdot0..3 ; u = z | cddot0..3 ; p = rdot_ref | (c_ref_i, sw_i) x 4  (prb.py:248-441, App. A.5) */
#define NX 30
#define NU 15
#define NP 11
#define NR 48 /* 6 + 3 zmp + 4 rel_pos + 15 min_qddot + 16 penalty rows */
#define MDL(n) l30_##n
static const int L30_C[4] = {3, 6, 9, 12}, L30_CD[4] = {18, 21, 24, 27};
static void l30_dyn(const consts_t* c, const double* x, const double* u, const double* p, double* xn) {
    (void)p;
    const double dt = c->dt, eta2 = GRAV / c->lip_h;
    double tmp[30];
    for (int i = 0; i < 15; ++i) tmp[i] = x[i] + dt * x[15 + i];
    for (int a = 0; a < 3; ++a) tmp[15 + a] = x[15 + a] + dt * (eta2 * (x[a] - u[a]) - (a == 2 ? GRAV : 0.0));
    for (int i = 0; i < 12; ++i) tmp[18 + i] = x[18 + i] + dt * u[3 + i];
    memcpy(xn, tmp, sizeof(tmp));
}
static void l30_dyn_jac(const consts_t* c, const double* x, const double* u, const double* p, double* F /*30x45*/) {
    (void)x; (void)u; (void)p;
    const double dt = c->dt, eta2 = GRAV / c->lip_h; const int nz = 45;
    memset(F, 0, sizeof(double) * 30 * nz);
    for (int i = 0; i < 30; ++i) F[i * nz + i] = 1.0;
    for (int i = 0; i < 15; ++i) F[i * nz + 15 + i] += dt;
    for (int a = 0; a < 3; ++a) { F[(15 + a) * nz + a] += dt * eta2; F[(15 + a) * nz + 30 + a] = -dt * eta2; }
    for (int i = 0; i < 12; ++i) F[(18 + i) * nz + 30 + 3 + i] = dt;
}
static int l30_residual(const consts_t* c, const double* x, const double* u, const double* p, int k, double* r, double* J) {
    int n = 0; const int nz = 45;
    if (J) memset(J, 0, sizeof(double) * NR * nz);
    double mean[3];
    for (int a = 0; a < 3; ++a) mean[a] = 0.25 * (x[3 + a] + x[6 + a] + x[9 + a] + x[12 + a]);
    const int state = !u || k >= 1;
    if (state) {
        const double g = sqrt(c->w_rz);
        r[n] = g * (x[2] - c->com_z); if (J) J[n * nz + 2] = g; ++n;                         /* rz_tracking  prb.py:390 */
        for (int e = 0; e < 2; ++e) {                                                          /* rxy_tracking prb.py:391 */
            r[n + e] = g * (x[e] - mean[e]);
            if (J) { J[(n + e) * nz + e] = g; for (int i = 0; i < 4; ++i) J[(n + e) * nz + L30_C[i] + e] = -0.25 * g; }
        }
        n += 2;
        const double gd = sqrt(c->w_rd);
        for (int a = 0; a < 3; ++a) { r[n + a] = gd * (x[15 + a] - p[a]); if (J) J[(n + a) * nz + 15 + a] = gd; }   /* prb.py:392 */
        n += 3;
    }
    if (u) {
        const double g = sqrt(c->w_zmp);                                                      /* zmp_tracking prb.py:393 */
        for (int a = 0; a < 3; ++a) {
            r[n + a] = g * (u[a] - mean[a]);
            if (J) { J[(n + a) * nz + 30 + a] = g; for (int i = 0; i < 4; ++i) J[(n + a) * nz + L30_C[i] + a] = -0.25 * g; }
        }
        n += 3;
    }
    if (state) n = rel_pos_rows(c, x, L30_C, r, J, nz, n);                                    /* prb.py:394-401 */
    if (u) {
        const double eta2 = GRAV / c->lip_h, g = sqrt(c->gq);                                 /* min_qddot prb.py:402 */
        for (int a = 0; a < 3; ++a) {
            r[n + a] = g * (eta2 * (x[a] - u[a]) - (a == 2 ? GRAV : 0.0));
            if (J) { J[(n + a) * nz + a] = g * eta2; J[(n + a) * nz + 30 + a] = -g * eta2; }
        }
        for (int i = 0; i < 12; ++i) { r[n + 3 + i] = g * u[3 + i]; if (J) J[(n + 3 + i) * nz + 33 + i] = g; }
        n += 15;
        double cref[4], sw[4];
        for (int i = 0; i < 4; ++i) { cref[i] = p[3 + 2 * i]; sw[i] = p[4 + 2 * i]; }
        n = contact_penalty_rows(x, 4, L30_C, L30_CD, cref, sw, r, J, nz, n, c->rv);
    }
    return n;
}
static double l30_stepcost(const consts_t* c, const double* x, const double* u, const double* p, int k, double* xn) {
    double r[NR]; const int n = l30_residual(c, x, u, p, k, r, NULL);
    double s = 0; for (int i = 0; i < n; ++i) s += r[i] * r[i];
    l30_dyn(c, x, u, p, xn);
    return s;
}
static void l30_second_order(const consts_t* c, const double* x, const double* u, const double* p, const double* vp, double theta, int mode,
                             double* Q) {
    (void)c; (void)x; (void)u; (void)p; (void)vp; (void)theta; (void)Q; (void)mode;                      /* linear dynamics */
}
#include "ddp_engine.inc"
#undef NX
#undef NU
#undef NP
#undef NR
#undef MDL

/* ---- exported entry points: model 0 srbd13, 1 srbd37, 2 lip30, 3 srbd61 (ids of include/sddp.h) ------------------------------------ */
int oracle_solve_batch(int model, const double* cpack, int N, int B, const double* x0, const double* P, double* xs, double* us,
                       const double* o, double* stats, int threads) {
    switch (model) {
        case 0: return s13_solve_batch(cpack, N, B, x0, P, xs, us, o, stats, threads);
        case 1: return s37_solve_batch(cpack, N, B, x0, P, xs, us, o, stats, threads);
        case 2: return l30_solve_batch(cpack, N, B, x0, P, xs, us, o, stats, threads);
        case 3: return s61_solve_batch(cpack, N, B, x0, P, xs, us, o, stats, threads);
    }
    return -1;
}
/* -> number of trace records written (ddp_engine.inc: TRACE_W doubles each), or -1 */
int oracle_trace_width(void) { return TRACE_W; }
int oracle_solve_trace(int model, const double* cpack, int N, const double* x0, const double* P, double* xs, double* us,
                       const double* o, double* stats, double* trace, int trace_cap) {
    switch (model) {
        case 0: return s13_solve_trace(cpack, N, x0, P, xs, us, o, stats, trace, trace_cap);
        case 1: return s37_solve_trace(cpack, N, x0, P, xs, us, o, stats, trace, trace_cap);
        case 2: return l30_solve_trace(cpack, N, x0, P, xs, us, o, stats, trace, trace_cap);
        case 3: return s61_solve_trace(cpack, N, x0, P, xs, us, o, stats, trace, trace_cap);
    }
    return -1;
}
int oracle_eval(int model, const double* cpack, const double* x, const double* u, const double* p, int k, int terminal,
                double* f, double* F, double* H, double* g, double* L) {
    switch (model) {
        case 0: return s13_eval(cpack, x, u, p, k, terminal, f, F, H, g, L);
        case 1: return s37_eval(cpack, x, u, p, k, terminal, f, F, H, g, L);
        case 2: return l30_eval(cpack, x, u, p, k, terminal, f, F, H, g, L);
        case 3: return s61_eval(cpack, x, u, p, k, terminal, f, F, H, g, L);
    }
    return -1;
}
/* kept names of the first version (srbd13 only) */
int oracle_srbd13_solve_batch(const double* cpack, int N, int B, const double* x0, const double* P, double* xs, double* us,
                              const double* o, double* stats, int threads) {
    return s13_solve_batch(cpack, N, B, x0, P, xs, us, o, stats, threads);
}
int oracle_srbd13_eval(const double* cpack, const double* x, const double* u, const double* p, int k, int terminal,
                       double* f, double* F, double* H, double* g, double* L) {
    return s13_eval(cpack, x, u, p, k, terminal, f, F, H, g, L);
}
