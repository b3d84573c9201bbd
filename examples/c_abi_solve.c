/* The C ABI of include/sddp.h used from plain C, no Python in the process: one standing srbd13 robot (BASELINE's metric model,
 * N = 30) solved from the warm start the reference's example computes (x = x0 at every node, u = static input;
 * dsrbd_example.py:61-68) with the example's solver options (dsrbd_example.py:55-58).
 *
 *   gcc -O2 -Iinclude examples/c_abi_solve.c -o build/c_abi_solve -Lsrbd_horizon_amd -lsddp_hip -Wl,-rpath,$PWD/srbd_horizon_amd -lm
 *   build/c_abi_solve [ticks]        -> one line: iterations, converged flag, cost, u_0 of the last tick, mean ms per warm-started tick
 *
 * tests/test_gpu_shim.py::test_c_host_program_gets_the_same_solve builds and runs it and compares with the Python path.
 * This is what a C / C++ maintainer of a controller would write: sddp_create replaces pyddp.DdpSolver(...) (ddp.py:93-94),
 * sddp_solve replaces .solve(param_values_list) (ddp.py:101). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sddp.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != SDDP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sddp_last_error(h)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = 30, ticks = argc > 1 ? atoi(argv[1]) : 20;
    int nx, nu, np;
    sddp_handle* h = NULL;
    if (sddp_model_dims(SDDP_MODEL_SRBD13, &nx, &nu, &np) != SDDP_OK) return 2;
    sddp_options o;
    sddp_model_consts c;
    sddp_default_options(&o);
    sddp_default_consts(&c);
    o.max_iters = 100; o.alpha_converge_threshold = 1e-12; o.beta = 1e-3;          /* dsrbd_example.py:55-58 */
    CHECK(sddp_create(&h, SDDP_MODEL_SRBD13, N, 1, &o, &c));

    double* x0 = calloc(nx, sizeof(double));
    double* P = calloc((size_t)(N + 1) * np, sizeof(double));
    double* xs = calloc((size_t)(N + 1) * nx, sizeof(double));
    double* us = calloc((size_t)N * nu, sizeof(double));
    /* x = r | o (x,y,z,w) | rdot | w, a little off its reference (prb.py:224-240 pattern) */
    x0[0] = 0.01; x0[1] = -0.005; x0[2] = c.com[2] + 0.01; x0[6] = 1.0; x0[7] = 0.02;
    /* p = rdot_ref(3) | w_ref(3) | otg | oref(4) | c_L(3) | c_R(3) | sw_L | sw_R: both feet on the ground */
    for (int k = 0; k <= N; ++k) {
        double* p = P + (size_t)k * np;
        p[6] = 1e2; p[10] = 1.0;                                                    /* otg, oref = (-0,-0,-0,1) */
        for (int a = 0; a < 3; ++a) { p[11 + a] = 0.5 * (c.feet[a] + c.feet[3 + a]); p[14 + a] = 0.5 * (c.feet[6 + a] + c.feet[9 + a]); }
        p[17] = p[18] = 1.0;
    }
    for (int k = 0; k <= N; ++k) memcpy(xs + (size_t)k * nx, x0, nx * sizeof(double));
    for (int k = 0; k < N; ++k) { us[k * nu + 2] = us[k * nu + 5] = c.m * 9.81 / c.force_scaling / 2.0; }   /* prb.py:242-246 */
    double* x = malloc((size_t)(N + 1) * nx * sizeof(double));
    double* u = malloc((size_t)N * nu * sizeof(double));
    sddp_stats st;
    CHECK(sddp_set_initial_state(h, x0));
    CHECK(sddp_set_x_warmstart(h, xs));
    CHECK(sddp_set_u_warmstart(h, us));
    CHECK(sddp_solve(h, P, x, u, &st));                                             /* cold solve */
    const int it0 = st.iters, conv0 = st.converged;
    const double cost0 = st.cost;
    /* warm-started ticks with the parameters and the previous solution resident on the device (sddp_advance shifts them there) */
    CHECK(sddp_set_params(h, P));
    struct timespec t0, t1;
    double ms = 0.0;
    for (int t = 0; t < ticks; ++t) {
        clock_gettime(CLOCK_MONOTONIC, &t0);
        CHECK(sddp_advance(h, P + (size_t)N * np, x + nx));                         /* next tick starts where the plan says: x_1 */
        CHECK(sddp_solve_resident(h, x, u, &st));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (t >= ticks / 4) ms += 1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec);
    }
    printf("{\"cold_iters\": %d, \"cold_converged\": %d, \"cold_cost\": %.17g, \"tick_iters\": %d, \"tick_converged\": %d, \"tick_cost\": %.17g, "
           "\"u0\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"ms_per_tick\": %.4f}\n",
           it0, conv0, cost0, st.iters, st.converged, st.cost, u[0], u[1], u[2], u[3], u[4], u[5], ms / (ticks - ticks / 4));
    sddp_destroy(h);
    free(x0); free(P); free(xs); free(us); free(x); free(u);
    return 0;
}
