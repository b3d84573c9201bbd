/* TEST INFRASTRUCTURE ONLY -- plain C restatement of the srbd13 problem and of the MS-DDP iteration (same steps, same
 * order as oracle/models.py + oracle/ddp.py, which carry the reference file:line citations).
 * PARITY UNPINNED upstream (the reference engine `pyddp` is absent): this file is pinned against the numpy oracle in
 * tests/test_oracle_c.py.  Used for (1) the `cpu_baseline` leg of bench.py (kind "port"), (2) large-batch parity checks.
 * Never linked into or called from the product (srbd_horizon_amd/).
 *
 * Reference lines restated: dynamics prb.py:92-110 (fSRBD, element-wise inertia prb.py:99, Euler ddp.py:228-230);
 * costs prb.py:184-204 through ddp.py:179-226; solver options ddp.py:14-35.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NX 13
#define NU 6
#define NZ 19
#define NP 19
#define NR 39 /* residual rows of a stage node: 11 state + 18 input (+ 10 friction-cone barrier rows when enabled) */
#define GRAV 9.81

typedef struct {
    double dt, inv_ms, Is[9], com_z, w_rz, w_rd, w_w, w_f, w_sw, gq, lever;
    double mu_lin, bar_w, bar_s;   /* friction-cone exponential barrier (oracle/models.py _force_rows): off when bar_w == 0 */
    int inertia_mode;
} consts_t;

/* packed constants from Python: m, I[9], com_z, dt, force_scaling, r_gain, rdot_gain, w_gain, fsw, qddot, minf, inertia_mode, lever,
 * friction_cone_coefficient, friction_barrier_weight, friction_barrier_sharpness */
static void unpack_consts(const double* c, consts_t* k) {
    const double m = c[0], fs = c[12];
    k->inv_ms = fs / m;
    for (int i = 0; i < 9; ++i) k->Is[i] = c[1 + i] / fs;
    k->com_z = c[10]; k->dt = c[11];
    k->w_rz = c[13]; k->w_rd = c[14]; k->w_w = c[15];
    k->w_sw = fs * fs * c[16]; k->gq = c[17]; k->w_f = fs * fs * c[18];
    k->inertia_mode = (int)c[19]; k->lever = c[20];
    k->mu_lin = c[21] / sqrt(2.0); k->bar_w = c[22]; k->bar_s = c[23];
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

typedef struct { double R[9], M[9], Mi[9], wdot[3], rddot[3]; } core_t;

/* p = rdot_ref(3) | w_ref(3) | otg | oref(4) | cL(3) | cR(3) | swL | swR */
static void core(const consts_t* c, const double* x, const double* u, const double* p, core_t* k) {
    const double *r = x, *o = x + 3, *w = x + 10;
    quat_to_rot(o, k->R); world_inertia(c, k->R, k->M); inv3(k->M, k->Mi);
    double tau[3] = {0, 0, 0}, fs[3] = {0, 0, 0}, Mw[3], g[3];
    for (int i = 0; i < 2; ++i) {
        const double* cp = p + 11 + 3 * i; const double* f = u + 3 * i;
        double l[3] = {cp[0] - r[0], cp[1] - r[1], cp[2] - r[2]}, t[3];
        cross(l, f, t);
        for (int a = 0; a < 3; ++a) { tau[a] += c->lever * t[a]; fs[a] += f[a]; }
    }
    mv3(k->M, w, Mw); cross(w, Mw, g);
    for (int a = 0; a < 3; ++a) tau[a] -= g[a];
    mv3(k->Mi, tau, k->wdot);
    k->rddot[0] = fs[0] * c->inv_ms; k->rddot[1] = fs[1] * c->inv_ms; k->rddot[2] = fs[2] * c->inv_ms - GRAV;
}

static void dyn(const consts_t* c, const double* x, const double* u, const double* p, double* xn) {
    core_t k; core(c, x, u, p, &k);
    const double *o = x + 3, *w = x + 10, dt = c->dt;
    double wxo[3]; cross(w, o, wxo);
    for (int a = 0; a < 3; ++a) {
        xn[a] = x[a] + dt * x[7 + a];
        xn[3 + a] = o[a] + dt * 0.5 * (o[3] * w[a] + wxo[a]);
        xn[7 + a] = x[7 + a] + dt * k.rddot[a];
        xn[10 + a] = x[10 + a] + dt * k.wdot[a];
    }
    xn[6] = o[3] - dt * 0.5 * (w[0] * o[0] + w[1] * o[1] + w[2] * o[2]);
}

/* A = d wdot / d [r(3) o(4) w(3) fL(3) fR(3)] (3 x 16) */
static void wdot_jac(const consts_t* c, const double* x, const double* u, const double* p, const core_t* k, double* A) {
    const double *r = x, *o = x + 3, *w = x + 10;
    double S[9], T[9], sf[3] = {0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int a = 0; a < 3; ++a) sf[a] += c->lever * u[3 * i + a];
    skew(sf, S); mm3(k->Mi, S, T);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a * 16 + b] = T[3 * a + b];
    double Mw[3], SMw[9], Sw[9], SwM[9], U[9];
    mv3(k->M, w, Mw); skew(Mw, SMw); skew(w, Sw); mm3(Sw, k->M, SwM);
    for (int i = 0; i < 9; ++i) U[i] = SMw[i] - SwM[i];
    mm3(k->Mi, U, T);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a * 16 + 7 + b] = T[3 * a + b];
    for (int q = 0; q < 4; ++q) {
        double dR[9], dM[9], a1[3], b1[3], cr[3], t[3], col[3];
        quat_to_rot_d(o, q, dR); world_inertia_d(c, k->R, dR, dM);
        mv3(dM, k->wdot, a1); mv3(dM, w, b1); cross(w, b1, cr);
        for (int a = 0; a < 3; ++a) t[a] = -(a1[a] + cr[a]);
        mv3(k->Mi, t, col);
        for (int a = 0; a < 3; ++a) A[a * 16 + 3 + q] = col[a];
    }
    for (int i = 0; i < 2; ++i) {
        const double* cp = p + 11 + 3 * i;
        double l[3] = {c->lever * (cp[0] - r[0]), c->lever * (cp[1] - r[1]), c->lever * (cp[2] - r[2])};
        skew(l, S); mm3(k->Mi, S, T);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a * 16 + 10 + 3 * i + b] = T[3 * a + b];
    }
}
static const int ZCOL[16] = {0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18};

static void dyn_jac(const consts_t* c, const double* x, const double* u, const double* p, double* F /*13x19*/) {
    core_t k; core(c, x, u, p, &k);
    double A[48]; wdot_jac(c, x, u, p, &k, A);
    const double *o = x + 3, *w = x + 10, dt = c->dt;
    memset(F, 0, sizeof(double) * NX * NZ);
    for (int i = 0; i < NX; ++i) F[i * NZ + i] = 1.0;
    for (int a = 0; a < 3; ++a) {
        F[a * NZ + 7 + a] += dt;
        for (int i = 0; i < 2; ++i) F[(7 + a) * NZ + NX + 3 * i + a] = dt * c->inv_ms;
        for (int j = 0; j < 16; ++j) F[(10 + a) * NZ + ZCOL[j]] += dt * A[a * 16 + j];
    }
    const double Jo[16] = {0, -0.5 * w[2], 0.5 * w[1], 0.5 * w[0], 0.5 * w[2], 0, -0.5 * w[0], 0.5 * w[1],
                           -0.5 * w[1], 0.5 * w[0], 0, 0.5 * w[2], -0.5 * w[0], -0.5 * w[1], -0.5 * w[2], 0};
    const double Jw[12] = {0.5 * o[3], 0.5 * o[2], -0.5 * o[1], -0.5 * o[2], 0.5 * o[3], 0.5 * o[0],
                           0.5 * o[1], -0.5 * o[0], 0.5 * o[3], -0.5 * o[0], -0.5 * o[1], -0.5 * o[2]};
    for (int a = 0; a < 4; ++a) {
        for (int b = 0; b < 4; ++b) F[(3 + a) * NZ + 3 + b] += dt * Jo[4 * a + b];
        for (int b = 0; b < 3; ++b) F[(3 + a) * NZ + 10 + b] += dt * Jw[3 * a + b];
    }
}

/* stacked residual and Jacobian of node k; terminal: u == NULL.  Returns the number of rows. */
static int residual(const consts_t* c, const double* x, const double* u, const double* p, int k, double* r, double* J /*NR x NZ or NULL*/) {
    int n = 0;
    if (J) memset(J, 0, sizeof(double) * NR * NZ);
    if (!u || k >= 1) {
        const double g = sqrt(c->w_rz);
        r[n] = g * (x[2] - c->com_z); if (J) J[n * NZ + 2] = g; ++n;
        const double *o = x + 3, *q = p + 7, otg = p[6];
        double oxq[3]; cross(o, q, oxq);
        for (int a = 0; a < 3; ++a) r[n + a] = otg * (o[3] * q[a] + q[3] * o[a] + oxq[a]);
        r[n + 3] = otg * (o[3] * q[3] - (o[0] * q[0] + o[1] * q[1] + o[2] * q[2]) - 1.0);
        if (J) {
            double Sq[9]; skew(q, Sq);
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) J[(n + a) * NZ + 3 + b] = otg * ((a == b ? q[3] : 0.0) - Sq[3 * a + b]);
                J[(n + a) * NZ + 6] = otg * q[a];
                J[(n + 3) * NZ + 3 + a] = -otg * q[a];
            }
            J[(n + 3) * NZ + 6] = otg * q[3];
        }
        n += 4;
        const double gd = sqrt(c->w_rd), gw = sqrt(c->w_w);
        for (int a = 0; a < 3; ++a) { r[n + a] = gd * (x[7 + a] - p[a]); if (J) J[(n + a) * NZ + 7 + a] = gd; }
        n += 3;
        for (int a = 0; a < 3; ++a) { r[n + a] = gw * (x[10 + a] - p[3 + a]); if (J) J[(n + a) * NZ + 10 + a] = gw; }
        n += 3;
    }
    if (u) {
        core_t kk; core(c, x, u, p, &kk);
        const double g = sqrt(c->gq);
        for (int a = 0; a < 3; ++a) { r[n + a] = g * kk.rddot[a]; r[n + 3 + a] = g * kk.wdot[a]; }
        if (J) {
            double A[48]; wdot_jac(c, x, u, p, &kk, A);
            for (int a = 0; a < 3; ++a) {
                for (int i = 0; i < 2; ++i) J[(n + a) * NZ + NX + 3 * i + a] = g * c->inv_ms;
                for (int j = 0; j < 16; ++j) J[(n + 3 + a) * NZ + ZCOL[j]] = g * A[a * 16 + j];
            }
        }
        n += 6;
        for (int i = 0; i < 2; ++i) {
            const double s1 = 1.0 - p[17 + i];
            const double g1 = sqrt(c->w_f), g2 = sqrt(c->w_sw) * s1;
            for (int a = 0; a < 3; ++a) { r[n + a] = g1 * u[3 * i + a]; if (J) J[(n + a) * NZ + NX + 3 * i + a] = g1; }
            n += 3;
            for (int a = 0; a < 3; ++a) { r[n + a] = g2 * u[3 * i + a]; if (J) J[(n + a) * NZ + NX + 3 * i + a] = g2; }
            n += 3;
            if (c->bar_w > 0.0) {   /* r_j = sqrt(w) exp(s a_j.f / 2), rows of the linearised cone A f <= 0 */
                const double ml = c->mu_lin;
                const double A[5][3] = {{1, 0, -ml}, {-1, 0, -ml}, {0, 1, -ml}, {0, -1, -ml}, {0, 0, -1}};
                for (int j = 0; j < 5; ++j) {
                    const double gj = A[j][0] * u[3 * i] + A[j][1] * u[3 * i + 1] + A[j][2] * u[3 * i + 2];
                    const double rj = sqrt(c->bar_w) * exp(0.5 * c->bar_s * gj);
                    r[n + j] = rj;
                    if (J) for (int a = 0; a < 3; ++a) J[(n + j) * NZ + NX + 3 * i + a] = 0.5 * c->bar_s * rj * A[j][a];
                }
                n += 5;
            }
        }
    }
    return n;
}

static double cost(const consts_t* c, const double* x, const double* u, const double* p, int k) {
    double r[NR]; const int n = residual(c, x, u, p, k, r, NULL);
    double s = 0; for (int i = 0; i < n; ++i) s += r[i] * r[i]; return s;
}
/* g = 2 J^T r (NZ), H = 2 J^T J (NZ x NZ) */
static double cost_derivs(const consts_t* c, const double* x, const double* u, const double* p, int k, double* g, double* H) {
    double r[NR], J[NR * NZ]; const int n = residual(c, x, u, p, k, r, J);
    double s = 0;
    memset(g, 0, sizeof(double) * NZ); memset(H, 0, sizeof(double) * NZ * NZ);
    for (int i = 0; i < n; ++i) {
        s += r[i] * r[i];
        for (int a = 0; a < NZ; ++a) {
            const double ja = J[i * NZ + a]; if (ja == 0.0) continue;
            g[a] += 2 * ja * r[i];
            for (int b = 0; b < NZ; ++b) H[a * NZ + b] += 2 * ja * J[i * NZ + b];
        }
    }
    return s;
}

static double total_cost(const consts_t* c, int N, const double* xs, const double* us, const double* P) {
    double J = 0; for (int k = 0; k < N; ++k) J += cost(c, xs + k * NX, us + k * NU, P + k * NP, k);
    return J + cost(c, xs + N * NX, NULL, P + N * NP, N);
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

static int backward(const consts_t* c, int N, const double* xs, const double* us, const double* P, const double* d, double mu,
                    double theta, double* K /*N x NU x NX*/, double* kff /*N x NU*/, double* sc /*dV1,dV2,G1,G2*/) {
    double Vx[NX], Vxx[NX * NX], g[NZ], H[NZ * NZ];
    cost_derivs(c, xs + N * NX, NULL, P + N * NP, N, g, H);
    for (int i = 0; i < NX; ++i) { Vx[i] = g[i]; for (int j = 0; j < NX; ++j) Vxx[i * NX + j] = H[i * NZ + j]; }
    double dV1 = 0, dV2 = 0, G1 = 0, G2 = 0;
    for (int k = N - 1; k >= 0; --k) {
        double F[NX * NZ], vp[NX], W[NX * NZ], Q[NZ * NZ], q[NZ];
        dyn_jac(c, xs + k * NX, us + k * NU, P + k * NP, F);
        cost_derivs(c, xs + k * NX, us + k * NU, P + k * NP, k, g, H);
        const double* dk = d + k * NX;
        for (int i = 0; i < NX; ++i) {
            double s = 0; for (int j = 0; j < NX; ++j) s += Vxx[i * NX + j] * dk[j];
            vp[i] = Vx[i] + s; G1 += dk[i] * Vx[i]; G2 += 0.5 * dk[i] * s;
        }
        for (int i = 0; i < NX; ++i) for (int j = 0; j < NZ; ++j) { double s = 0; for (int l = 0; l < NX; ++l) s += Vxx[i * NX + l] * F[l * NZ + j]; W[i * NZ + j] = s; }
        for (int a = 0; a < NZ; ++a) {
            double s = g[a]; for (int l = 0; l < NX; ++l) s += F[l * NZ + a] * vp[l]; q[a] = s;
            for (int b = 0; b < NZ; ++b) { double t = H[a * NZ + b]; for (int l = 0; l < NX; ++l) t += F[l * NZ + a] * W[l * NZ + b]; Q[a * NZ + b] = t; }
        }
        if (theta != 0.0) { /* exact bilinear-torque term: Qux[f_a][r_b] -= theta * s * skew(y)[a][b], y = I_w^-1 (dt v'_w) */
            double R[9], M[9], Mi[9], lam[3], y[3], S[9];
            quat_to_rot(xs + k * NX + 3, R); world_inertia(c, R, M); inv3(M, Mi);
            for (int a = 0; a < 3; ++a) lam[a] = c->dt * vp[10 + a];
            mv3(Mi, lam, y); skew(y, S);
            for (int i = 0; i < 2; ++i) for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) {
                const double v = -theta * c->lever * S[3 * a + b2];
                Q[(NX + 3 * i + a) * NZ + b2] += v; Q[b2 * NZ + NX + 3 * i + a] += v;
            }
        }
        double L[NU * NU], Quu[NU * NU];
        for (int i = 0; i < NU; ++i) for (int j = 0; j < NU; ++j) Quu[i * NU + j] = L[i * NU + j] = Q[(NX + i) * NZ + NX + j] + (i == j ? mu : 0.0);
        if (!chol(L, NU)) return 0;
        /* solve for the NX+1 right-hand sides [Qu | Qux] */
        double sol[NU * (NX + 1)];
        for (int col = 0; col <= NX; ++col) {
            double y[NU];
            for (int i = 0; i < NU; ++i) { double s = col == 0 ? q[NX + i] : Q[(NX + i) * NZ + (col - 1)]; for (int kx = 0; kx < i; ++kx) s -= L[i * NU + kx] * y[kx]; y[i] = s / L[i * NU + i]; }
            for (int i = NU - 1; i >= 0; --i) { double s = y[i]; for (int kx = i + 1; kx < NU; ++kx) s -= L[kx * NU + i] * y[kx]; y[i] = s / L[i * NU + i]; }
            for (int i = 0; i < NU; ++i) sol[i * (NX + 1) + col] = -y[i];
        }
        double* Kk = K + (size_t)k * NU * NX; double* kk = kff + k * NU;
        for (int i = 0; i < NU; ++i) { kk[i] = sol[i * (NX + 1)]; for (int j = 0; j < NX; ++j) Kk[i * NX + j] = sol[i * (NX + 1) + 1 + j]; }
        for (int i = 0; i < NU; ++i) { dV1 += kk[i] * q[NX + i]; double s = 0; for (int j = 0; j < NU; ++j) s += Quu[i * NU + j] * kk[j]; dV2 += 0.5 * kk[i] * s; }
        double Vn[NX * NX];
        for (int a = 0; a < NX; ++a) {
            double s = q[a]; for (int i = 0; i < NU; ++i) s += Q[(NX + i) * NZ + a] * kk[i]; Vx[a] = s;
            for (int b = 0; b < NX; ++b) { double t = Q[a * NZ + b]; for (int i = 0; i < NU; ++i) t += Q[(NX + i) * NZ + a] * Kk[i * NX + b]; Vn[a * NX + b] = t; }
        }
        for (int a = 0; a < NX; ++a) for (int b = 0; b < NX; ++b) Vxx[a * NX + b] = 0.5 * (Vn[a * NX + b] + Vn[b * NX + a]);
    }
    sc[0] = dV1; sc[1] = dV2; sc[2] = G1; sc[3] = G2;
    return 1;
}

static double forward(const consts_t* c, int N, const double* x0, const double* xs, const double* us, const double* P, const double* d,
                      const double* K, const double* kff, double alpha, double* xn, double* un) {
    memcpy(xn, x0, sizeof(double) * NX);
    double J = 0;
    for (int k = 0; k < N; ++k) {
        double* x = xn + k * NX; double* u = un + k * NU;
        for (int i = 0; i < NU; ++i) {
            double s = us[k * NU + i] + alpha * kff[k * NU + i];
            for (int j = 0; j < NX; ++j) s += K[((size_t)k * NU + i) * NX + j] * (x[j] - xs[k * NX + j]);
            u[i] = s;
        }
        J += cost(c, x, u, P + k * NP, k);
        dyn(c, x, u, P + k * NP, x + NX);
        for (int i = 0; i < NX; ++i) x[NX + i] -= (1.0 - alpha) * d[k * NX + i];
    }
    return J + cost(c, xn + N * NX, NULL, P + N * NP, N);
}

/* opts: max_iters, alpha_0, alpha_converge_threshold, factor, beta, cost_reduction_ths, mu0, initial_rollout, gap_tol, mu_min, mu_max, second_order
 * stats out: cost, iters, converged, alpha, gap, mu, status */
int oracle_srbd13_solve(const double* cpack, int N, const double* x0, const double* P, double* xs, double* us, const double* o, double* stats) {
    consts_t c; unpack_consts(cpack, &c);
    const int max_iters = (int)o[0];
    const double a0 = o[1], athr = o[2], fac = o[3], beta = o[4], ths = o[5], mu0 = o[6], gap_tol = o[8], mu_min = o[9], mu_max = o[10];
    double* d = (double*)calloc((size_t)N * NX, sizeof(double));
    double* K = (double*)malloc(sizeof(double) * (size_t)N * NU * NX);
    double* kff = (double*)malloc(sizeof(double) * (size_t)N * NU);
    double* xn = (double*)malloc(sizeof(double) * (size_t)(N + 1) * NX);
    double* un = (double*)malloc(sizeof(double) * (size_t)N * NU);
    if ((int)o[7]) { memcpy(xs, x0, sizeof(double) * NX); for (int k = 0; k < N; ++k) dyn(&c, xs + k * NX, us + k * NU, P + k * NP, xs + (k + 1) * NX); }
    else {
        memcpy(xs, x0, sizeof(double) * NX);
        for (int k = 0; k < N; ++k) { double f[NX]; dyn(&c, xs + k * NX, us + k * NU, P + k * NP, f); for (int i = 0; i < NX; ++i) d[k * NX + i] = f[i] - xs[(k + 1) * NX + i]; }
    }
    double J = total_cost(&c, N, xs, us, P), gap = 0;
    for (int i = 0; i < N * NX; ++i) gap += fabs(d[i]);
    double mu = mu0, rho = 0, alpha = 0, theta = 0;
    const int second_order = (int)o[11];
    int iters = 0, converged = 0, status = 1;
    if (!isfinite(J)) status = 3;
    else while (iters < max_iters) {
        double sc[4]; int ok;
        for (;;) {
            ok = backward(&c, N, xs, us, P, d, mu, theta, K, kff, sc);
            if (ok) break;
            if (theta != 0.0) { theta = 0.0; continue; }
            mu = fmax(mu, 0.0) * 10.0 + mu_min; if (mu > mu_max) break;
        }
        if (!ok) { status = 2; break; }
        const double expected = -(sc[0] + sc[1]);
        if (expected < ths && gap <= gap_tol) { converged = 1; status = 0; break; }
        const double A1 = sc[0] + sc[2], B2 = sc[1] + sc[3];
        if (gap > 0.0) rho = fmax(rho, 2.0 * fmax(fmax(A1, A1 + B2), 0.0) / gap);
        const double slack = 1e-13 * (fabs(J) + rho * gap);
        double a = a0, Jn = 0; int accepted = 0;
        while (a >= athr) {
            Jn = forward(&c, N, x0, xs, us, P, d, K, kff, a, xn, un);
            const double pred = a * A1 + a * a * B2 - a * rho * gap;
            const double dphi = (Jn + rho * (1.0 - a) * gap) - (J + rho * gap);
            if (isfinite(Jn) && dphi <= beta * pred + slack) { accepted = 1; break; }
            a *= fac;
        }
        if (!accepted) {
            if (theta != 0.0) { theta = 0.0; continue; }
            status = 4; converged = (gap <= gap_tol && expected <= ths * fmax(1.0, fabs(J))) ? 1 : 0; alpha = 0.0; break;
        }
        alpha = a;
        theta = (second_order && a == a0) ? 1.0 : 0.0;
        const double dJ = J - Jn;
        memcpy(xs, xn, sizeof(double) * (size_t)(N + 1) * NX); memcpy(us, un, sizeof(double) * (size_t)N * NU);
        J = Jn;
        for (int i = 0; i < N * NX; ++i) d[i] *= (1.0 - a);
        gap *= (1.0 - a);
        ++iters;
        if (mu > mu0) mu = fmax(mu0, mu * 0.1);
        if (fabs(dJ) < ths && gap <= gap_tol) { converged = 1; status = 0; break; }
    }
    stats[0] = J; stats[1] = iters; stats[2] = converged; stats[3] = alpha; stats[4] = gap; stats[5] = mu; stats[6] = status;
    free(d); free(K); free(kff); free(xn); free(un);
    return 0;
}

/* batch of independent instances; OpenMP over instances when compiled with -fopenmp */
int oracle_srbd13_solve_batch(const double* cpack, int N, int B, const double* x0, const double* P, double* xs, double* us,
                              const double* o, double* stats, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int b = 0; b < B; ++b)
        oracle_srbd13_solve(cpack, N, x0 + (size_t)b * NX, P + (size_t)b * (N + 1) * NP, xs + (size_t)b * (N + 1) * NX,
                            us + (size_t)b * N * NU, o, stats + (size_t)b * 7);
    return 0;
}

/* per-knot evaluation for the cross-check against the numpy oracle: f[NX], F[NX*NZ], H[NZ*NZ], g[NZ], L */
int oracle_srbd13_eval(const double* cpack, const double* x, const double* u, const double* p, int k, int terminal,
                       double* f, double* F, double* H, double* g, double* L) {
    consts_t c; unpack_consts(cpack, &c);
    if (!terminal) { dyn(&c, x, u, p, f); dyn_jac(&c, x, u, p, F); }
    *L = cost_derivs(&c, x, terminal ? NULL : u, p, k, g, H);
    return 0;
}
