/* sddp.h -- C ABI of the MI355X-native SRBD/LIP DDP engine (libsddp_hip.so).
 *
 * This is the drop-in boundary for the per-tick solve of hucebot/srbd_horizon: it replaces the surface of the
 * external native module `pyddp` that the reference's adapter binds (reference python/ddp.py).  Each entry point
 * cites the reference call it replaces.  CasADi `Function` lists (f_list, L_list, L_term: ddp.py:83-94) are
 * replaced by a `model_id` + `sddp_model_consts`, because the models are hand-written analytic HIP code.
 *
 * Conventions: plain C, plain pointers and sizes; every function returns an int status (0 = ok, <0 = error,
 * message via sddp_last_error) and never throws; all arrays are float64, knot-major, C-contiguous:
 *     x      [B][N+1][nx]      u [B][N][nu]      params [B][N+1][np]      x0 [B][nx]
 * (the ctypes shim transposes to the reference's [dim, nodes] numpy layout, ddp.py:139,:146).
 * Host-pointer calls copy through PCIe; the *_device calls take HBM-resident pointers (the bench path).
 * One HIP stream per handle; a handle is not thread-safe (the reference drives its solver from one thread,
 * dsrbd_example.py:82-135).
 */
#ifndef SDDP_H
#define SDDP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SDDP_ABI_VERSION 9

/* model ids */
#define SDDP_MODEL_SRBD13 0 /* nx=13 nu=6  np=19 : BASELINE.json metric model (contacts are per-knot parameters) */
#define SDDP_MODEL_SRBD37 1 /* nx=37 nu=24 np=19 : reference SRBD problem, prb.py:16-246                        */
#define SDDP_MODEL_LIP30  2 /* nx=30 nu=15 np=11 : reference LIP problem,  prb.py:248-441                       */
#define SDDP_MODEL_SRBD61 3 /* nx=61 nu=48 np=27 : reference SRBD problem at its code-default contact_model = 4
                               (nc = 8, prb.py:39-41); no second_order = 2 build, no bound barrier (109 > 64) */

/* status codes */
#define SDDP_OK 0
#define SDDP_ERR_ARG (-1)
#define SDDP_ERR_HIP (-2)
#define SDDP_ERR_MODEL (-3)
#define SDDP_ERR_NOMEM (-4)

/* replaces pyddp.DdpSolverOptions (fields set at ddp.py:14-35) */
typedef struct sddp_options {
    int    max_iters;                   /* ddp.py:17-19  */
    double alpha_0;                     /* ddp.py:20-22  */
    double alpha_converge_threshold;    /* ddp.py:23-25  */
    double line_search_decrease_factor; /* ddp.py:26-28  */
    double beta;                        /* ddp.py:29-31  */
    double cost_reduction_ths;          /* ddp.py:32-33  */
    double mu0;                         /* ddp.py:34-35  */
    int    initial_rollout;             /* 1: single shooting (x warm start ignored); 0: multiple shooting */
    double gap_tol;                     /* defect 1-norm below which the gaps count as closed */
    double mu_min;                      /* regularisation bump floor */
    double mu_max;                      /* give up above this */
    int    second_order;                /* 0: plain Gauss-Newton / iLQR sweep.  1 (default): add the exact bilinear-torque term v'.f_ux
                                           once full steps are accepted.  2: full second-order DDP -- the whole
                                           dynamics tensor v'.f_zz (second derivatives of wdot and of the quaternion kinematics) and the
                                           exact instead of the Gauss-Newton Hessian of the wdot residual, same schedule and fallback;
                                           selects its own kernel build at sddp_create */
    int    waves_per_simd;              /* scheduling hint, no effect on results.  1 (default): the kernel build with the full
                                           register file per instance -- shortest time for ONE batch.  2: the build capped at
                                           half the register file, two instances resident per SIMD (srbd13) or two workgroups
                                           per CU (the 4-wavefront kernel of srbd37 / lip30, whose tiles fit a CU's LDS twice)
                                           -- highest solves/s for queues of many instances.  A build
                                           that gains no resident workgroup from it (srbd37 with a barrier) runs as with 1. */
    int    queue_order;                 /* scheduling hint, no effect on results.  A solve launch runs on the workgroups that are
                                           resident on the device at once ("slots": 256 CUs x 4 SIMDs x waves_per_simd for srbd13,
                                           256 CUs x waves_per_simd for srbd37 / lip30);
                                           a batch with more instances than slots is a work queue the slots pull from.  A launch ends
                                           with its slowest instance, so the slow ones should start first:
                                           1 (default): longest PREVIOUS solve first -- the queue is ordered by the iteration count of
                                              each instance's previous solve on this handle (never-solved first): the predictor of a
                                              fleet whose robots recur tick after tick;
                                           2: largest INITIAL COST first -- a pre-pass of the same launch evaluates the total cost of
                                              every instance's warm start and the queue is sorted by it (descending): needs no history,
                                              the order for cold queues (instances never seen before);
                                           3: longest CLASS HISTORY first -- the caller labels every instance with the class of
                                              problem it is (sddp_set_instance_classes: gait phase, command, ...; no knowledge of
                                              the solution); the key is the mean iteration count the earlier solves of that class
                                              took on this handle, the initial cost of order 2 breaking ties; classes not seen yet
                                              (and unlabelled instances) start first.  New instances of known kinds: what a fleet
                                              server sees.  (v9)
                                           0: index order. */
    int    max_slots;                   /* 0 (default): every workgroup the device can keep resident is a queue slot.  > 0: at most
                                           this many (diagnostics and tests: a queue on few instances); reported by sddp_queue_info */
} sddp_options;

/* replaces what the reference bakes into the CasADi graphs from the URDF and the rosparam server
 * (prb.py:92-99 mass/inertia, prb.py:130-139 feet/com, prb.py:142-150 and :358-362 gains) */
typedef struct sddp_model_consts {
    double m;              /* kg                                   prb.py:92    */
    double I[9];           /* kg m^2, row major                    prb.py:94-95 */
    double com[3];         /*                                      prb.py:138   */
    double feet[12];       /* contact points 0..3 x 3, row major   prb.py:130-131 (all the graphs use: d_initial_1/2 of
                              prb.py:153-154 index feet 0..3 whatever nc is; the other feet only enter x0 / c_ref, host side) */
    double dt;             /* T/ns                                 prb.py:110   */
    double force_scaling;  /* 1000                                 prb.py:98    */
    double r_tracking_gain, rdot_tracking_gain, w_tracking_gain, rel_pos_gain;       /* prb.py:142-147 */
    double force_switch_weight, min_qddot_gain, min_f_gain, zmp_tracking_gain;       /* prb.py:148-150, :360 */
    double lip_height;     /* 0.88                                 prb.py:317   */
    int    inertia_mode;   /* 0: R o I o R^T element-wise (reference-faithful, prb.py:99); 1: R I R^T */
    double lever_sign;     /* +1: (c - r) x f ; -1: (r - c) x f    */
    /* Inequality handling the reference collects and ignores (friction cone prb.py:172-177, exponential barrier ddp.py:197-202,
     * both commented out upstream).  friction_barrier_weight = 0 (default) = the reference's behaviour.  > 0 adds, per contact
     * force and stage node, weight * sum_j exp(sharpness * a_j.f) over the 5 rows of the linearised cone A f <= 0
     * (inner pyramid mu / sqrt(2), unilateral f_z >= 0), with Gauss-Newton Hessians like every other cost. */
    double friction_cone_coefficient;  /* 0.8, rosparam default prb.py:174 */
    double friction_barrier_weight;    /* 0 = off */
    double friction_barrier_sharpness; /* 1 */
    /* Variable bounds: the reference's second commented-out block (ddp.py:203-208) adds, for every state and input variable and
     * stage node, exp(exp_parameter (v - upper)) + exp(exp_parameter (lower - v)) with exp_parameter = 6 (ddp.py:182); prb.py sets
     * no bounds.  bound_barrier_weight = 0 (default) = the reference's behaviour.  > 0 adds weight * that sum over the entries of
     * z = [x u] whose bound is finite (lower[j] / upper[j], j < nx + nu; -inf / +inf = unbounded), Gauss-Newton Hessians like every
     * other cost.  srbd13 and srbd37 only (lower / upper hold 64 entries of z). */
    double bound_barrier_weight;       /* 0 = off */
    double bound_barrier_sharpness;    /* 6 */
    double lower[64];
    double upper[64];
    /* prb.py:166-170 adds the relative-velocity constraints cdot_lead,xy - cdot_i,xy inside a foot only `if contact_model > 1`.
     * 1 (default): they are there -- contact_model = 2 (srbd37, lip30) or 4 (srbd61).  0: the same state layout with POINT feet,
     * number_of_legs = 4 x contact_model = 1 (nc = 4 on the srbd37 / lip30 build): no such rows.  (v9) */
    int    relative_velocity_constraints;
    /* User-declared LINEAR residual rows.  The reference's stage / terminal costs sum WHATEVER residual its function container
     * holds (ddp.py:183-196, :216-226); the analytic models hard-wire prb.py's terms, and take on top up to SDDP_MAX_EXTRA rows
     *     r_j = sqrt(extra_weight[j]) * ( extra_a[j] . z  -  ( p[np + j] + extra_const[j] ) ),      z = [x u],
     * extra_kind[j] 0: a state row, active on nodes 1..N like prb.py's tracking terms (terminal node included; its coefficients on
     * the inputs must be 0), 1: a stage row, nodes 0..N-1 like min_qddot.  n_extra > 0 selects the model's "_x" build (every
     * model; plain build only: no barrier, no second_order = 2), whose parameter vector is SDDP_MAX_EXTRA columns wider
     * than sddp_model_dims says: columns np .. np + 7 of every node are the per-knot references of rows 0..7 (unused ones: 0);
     * sddp_handle_dims reports the handle's width.  (v9) */
    int    n_extra;
    int    extra_kind[8];
    double extra_weight[8];
    double extra_const[8];
    double extra_a[8 * 128];    /* row j: extra_a[128 j + i], i < nx + nu */
} sddp_model_consts;
#define SDDP_MAX_EXTRA 8

/* per-instance solve record (what pyddp exposes only as is_converged(), ddp.py:106, plus the tic/toc of
 * dsrbd_example.py:134-136) */
typedef struct sddp_stats {
    double cost;       /* final total cost                                  */
    double alpha;      /* last accepted (or last tried) step length         */
    double gap;        /* remaining defect 1-norm                           */
    double mu;         /* final regularisation                              */
    double expected;   /* last expected reduction -(dV1+dV2)                */
    double rho;        /* l1 merit weight on the defects at the end (0 while no line search ran with open gaps).  With
                          cost / alpha / gap / mu it is the whole state the iteration carries from one accepted step to the
                          next, so a solve cut at max_iters = k can be continued -- or checked step by step (v9)      */
    int    iters;      /* accepted iterations                               */
    int    converged;  /* 1/0                                               */
    int    status;     /* 0 ok (converged = 1), 1 max_iters, 2 regularisation overflow, 3 non-finite cost, 4 line search
                          exhausted with the optimality conditions not met (converged = 0).  An exhausted line search (no step
                          length >= alpha_converge_threshold decreases the merit function) counts as converged -- status 0 -- only
                          with closed gaps (gap <= gap_tol) and expected <= cost_reduction_ths * max(1, |cost|): a RELATIVE test,
                          where the regular exits test expected / |dJ| < cost_reduction_ths absolutely */
    int    rollouts;   /* forward passes executed                           */
} sddp_stats;

typedef struct sddp_handle sddp_handle;

int  sddp_abi_version(void);
int  sddp_model_dims(int model_id, int* nx, int* nu, int* np);
/* dimensions of THIS handle: np is SDDP_MAX_EXTRA larger than sddp_model_dims' when the handle carries user rows (n_extra > 0) */
int  sddp_handle_dims(sddp_handle* h, int* nx, int* nu, int* np);
void sddp_default_options(sddp_options* opts);
void sddp_default_consts(sddp_model_consts* consts);            /* = sddp_default_consts_for(SDDP_MODEL_SRBD37, ...) */
/* the synthetic robot as model `model_id` needs it: `feet` holds contact points 0..3 of THAT model --
 * the line feet (+-0.08, +-0.1, 0) for srbd13 / srbd37 / lip30, the first four sole corners (+-0.08, 0.13 / 0.07, 0) of the
 * eight-point contact model for srbd61 (prb.py:39-41, :130-131, :153-154).  sddp_create(consts = NULL) uses these.  (v9) */
int  sddp_default_consts_for(int model_id, sddp_model_consts* consts);

/* replaces pyddp.DdpSolver(nx, nu, f_list, L_list, L_term, opts)                        ddp.py:93-94 */
int  sddp_create(sddp_handle** out, int model_id, int N, int batch,
                 const sddp_options* opts, const sddp_model_consts* consts);
void sddp_destroy(sddp_handle* h);
const char* sddp_last_error(const sddp_handle* h); /* h may be NULL: last create() error */
int  sddp_set_options(sddp_handle* h, const sddp_options* opts);

/* replaces DdpSolver.set_initial_state(x0)                                              ddp.py:122-123 */
int  sddp_set_initial_state(sddp_handle* h, const double* x0 /*[B][nx]*/);
/* replaces DdpSolver.set_x_warmstart(x) / set_u_warmstart(u)                            ddp.py:113-117 */
int  sddp_set_x_warmstart(sddp_handle* h, const double* x /*[B][N+1][nx]*/);
int  sddp_set_u_warmstart(sddp_handle* h, const double* u /*[B][N][nu]*/);
/* replaces x, u = DdpSolver.solve(param_values_list)                                    ddp.py:101 */
int  sddp_solve(sddp_handle* h, const double* params /*[B][N+1][np]*/,
                double* x_out /*[B][N+1][nx]*/, double* u_out /*[B][N][nu]*/, sddp_stats* stats /*[B] or NULL*/);
/* replaces DdpSolver.is_converged()                                                     ddp.py:106 */
int  sddp_is_converged(sddp_handle* h, int* flags /*[B]*/);

/* ---- HBM-resident path (no PCIe in the call): pointers are DEVICE pointers ------------------------------- */
int  sddp_set_stream(sddp_handle* h, void* hip_stream);
int  sddp_set_initial_state_device(sddp_handle* h, const double* d_x0);
int  sddp_set_x_warmstart_device(sddp_handle* h, const double* d_x);
int  sddp_set_u_warmstart_device(sddp_handle* h, const double* d_u);
/* asynchronous on the handle's stream; results stay in the handle's buffers (sddp_device_ptr) */
int  sddp_solve_device(sddp_handle* h, const double* d_params);
int  sddp_synchronize(sddp_handle* h);
/* A handle as a queue of instances (a fleet server: batch = every robot it may hold).  Load the instances [first, first+count)
 * (initial state, x / u warm start; device pointers to `count` instances, NULL = leave as is) and solve a contiguous range of
 * the batch in ONE launch; d_params is the whole [B][N+1][np] tensor.  Each is asynchronous on the handle's stream.  The solve
 * of an instance is bit-identical whatever range, order or slot it runs in. */
int  sddp_load_range_device(sddp_handle* h, int first, int count, const double* d_x0, const double* d_x, const double* d_u);
int  sddp_solve_range_device(sddp_handle* h, const double* d_params, int first, int count);
/* The solution record an instance-sharded fleet exchanges after a launch (one all-gather per launch):
 * the instances [first, first + count) packed [count][words] into d_out (a DEVICE pointer, e.g. the collective's send buffer) by
 * one kernel behind the solve on the handle's stream.  mode 0: x [N+1][nx] | u [N][nu] | cost | iterations (the whole plan);
 * mode 1: u_0 [nu] | x_1 [nx] | cost | iterations (what a closed loop applies next; 168 B instead of 4 680 B at (30, 13, 6)).
 * sddp_record_words: doubles per record.  (v9) */
int  sddp_record_words(sddp_handle* h, int mode, int* words);
int  sddp_pack_records_device(sddp_handle* h, int first, int count, int mode, double* d_out);
/* Class labels for queue_order = 3: classes[b] in [0, n_classes) (or -1: unlabelled) says what kind of problem instance b is --
 * anything the caller knows BEFORE the solve that correlates with its length.  The handle keeps, per class, the iterations and the
 * number of solves of every labelled instance it has solved (under any queue order) and orders a queued launch by the class
 * means.  Host pointer [B] (synchronous) / device pointer to `count` labels for the instances [first, first + count) (asynchronous
 * on the handle's stream: the fleet-queue path).  n_classes is fixed by the first call.  sddp_class_history reads one class. (v9) */
int  sddp_set_instance_classes(sddp_handle* h, const int* classes /*[B]*/, int n_classes);
int  sddp_set_instance_classes_range_device(sddp_handle* h, int first, int count, const int* d_classes, int n_classes);
int  sddp_class_history(sddp_handle* h, int cls, double* mean_iters, long long* solves);
/* slots: resident workgroups the work buffers exist for; grid and queue length (0: no queue) of the last solve launch */
int  sddp_queue_info(sddp_handle* h, int* slots, int* last_grid, int* last_queued);
/* which kernel a handle runs: wavefronts per instance (1: solve_kernel, 4: solve_kernel_mw), the build the LAST solve launch used
 * (1: full register file, 2: the `_w2` half-register-file build -- a handle asked for 2 falls back to 1 where that gains no
 * resident workgroup) and the model name the kernels are instantiated for.  Any pointer may be NULL. */
int  sddp_kernel_info(sddp_handle* h, int* wavefronts_per_instance, int* last_waves_per_simd, const char** model_name);
/* what that kernel build takes from a CU, read from its code object (hipFuncGetAttributes): architected vector registers per
 * lane, scratch (spill) bytes per lane, LDS bytes per workgroup (static + dynamic), and how many of its workgroups the device keeps
 * resident per CU (the occupancy query the queue's slot count comes from).  Any pointer may be NULL. */
int  sddp_kernel_resources(sddp_handle* h, int* vgprs, int* scratch_bytes_per_lane, int* lds_bytes, int* workgroups_per_cu);
/* Supported diagnostic: fills the LDS of the device's CUs with NaNs (8 short workgroups per CU, each owning the CU's whole LDS)
 * and waits for it.  A kernel finds in LDS what the previous one left; after this call a word read before it is written shows
 * up as a NaN in the result instead of passing on a lucky leftover.  No solve path calls it (the library reads no environment
 * variable either): a harness that wants every launch on poisoned LDS calls it before each launch (tests/conftest.py does,
 * under SDDP_POISON_LDS=1). */
int  sddp_debug_poison_lds(sddp_handle* h);
/* results of the last device solve (x, u, stats of the whole batch) to host pointers; waits for the stream */
int  sddp_fetch(sddp_handle* h, double* x_out, double* u_out, sddp_stats* stats /*[B] or NULL*/);
/* which: 0 xs [B][N+1][nx], 1 us [B][N][nu], 2 stats [B] (sddp_stats), 3 gains [slots][N][nu*(nx+1)], 4 x0 [B][nx],
 * 5 params [B][N+1][np] (the resident tensor of sddp_set_params), 6 order [last_queued] (int: the instance indices in the
 * order the last queued launch handed them out; queue_order 1 or 2 only), 7 slot times [last_grid][2] (uint64: the 100 MHz
 * constant-rate clock at which each slot of the last solve launch started and found the queue empty).  The feedback gains (3) are work buffers of the queue SLOTS:
 * row b holds instance b's gains (kff then K row-major, of its last backward sweep) only after a launch that covered the
 * instances from 0 without a queue (sddp_solve / sddp_solve_device / sddp_solve_resident with B <= slots, or a range with
 * first = 0 and count <= slots); after any other launch the call returns SDDP_ERR_ARG instead of another robot's gains. */
int  sddp_device_ptr(sddp_handle* h, int which, void** ptr, long long* bytes);
/* average device time (ms) of the last `sddp_solve*` kernel launch measured with HIP events on the handle's stream */
int  sddp_last_kernel_ms(sddp_handle* h, double* ms);
int  sddp_enable_timing(sddp_handle* h, int on);
/* sum and count of the solve-kernel durations (HIP events, one pair per launch) read at the last sddp_synchronize calls */
int  sddp_kernel_time_stats(sddp_handle* h, double* sum_ms, long long* count, int reset);

/* ---- receding horizon with device-resident data ------------------------------------------
 * Per tick the reference shifts every parameter back by one node and writes node N (dsrbd_example.py:102-106, :119-122;
 * wpg.py:74-99), and its solver object keeps the previous solution (dsrbd_example.py:59).  With these three calls only the new
 * last parameter column and the measured state cross PCIe; the shift runs on the device. */
/* upload the whole parameter tensor once (host pointer) */
int  sddp_set_params(sddp_handle* h, const double* params /*[B][N+1][np]*/);
/* one tick: params[k] <- params[k+1] (k < N), params[N] <- p_last;  warm start <- previous solution advanced by one knot
 * (x[k] <- x[k+1], x[N] kept; u[k] <- u[k+1], u[N-1] kept);  initial state <- x0.  Host pointers. */
int  sddp_advance(sddp_handle* h, const double* p_last /*[B][np]*/, const double* x0 /*[B][nx]*/);
/* sddp_solve on the resident parameters */
int  sddp_solve_resident(sddp_handle* h, double* x_out, double* u_out, sddp_stats* stats /*[B] or NULL*/);
/* sddp_solve_resident for a fleet in closed loop: what a tick applies is the first input and the state the plan expects next
 * (dsrbd_example.py:158: u_opt[:, 0]), so only u_0 [B][nu], x_1 [B][nx], the cost and the iteration count / status of every
 * instance leave the device (one small copy); the trajectories stay in HBM as the next tick's warm start (sddp_advance) and can
 * still be read with sddp_fetch.  cost_out / iters_out / status_out may be NULL. */
int  sddp_solve_resident_first(sddp_handle* h, double* u0_out /*[B][nu]*/, double* x1_out /*[B][nx]*/, double* cost_out /*[B]*/,
                               int* iters_out /*[B]*/, int* status_out /*[B]*/);
/* one model step per instance, x_next = f_k(x, u; p) with the handle's model and constants: the closed-loop simulator step
 * of the examples (dsrbd_example.py:158-159: integrator EULER of the same dae, :76) through the solver's own device model
 * code.  k = stage node whose parameters p are (0 <= k < N).  Host pointers; synchronous. */
int  sddp_model_step(sddp_handle* h, const double* x /*[B][nx]*/, const double* u /*[B][nu]*/, const double* p /*[B][np]*/, int k,
                     double* x_next /*[B][nx]*/);

/* ---- building blocks exposed for parity tests (host pointers) ------------------------------------------------
 * The same device code the fused solve kernel uses, one phase at a time.
 * sddp_eval_knots: per-knot model evaluation (reference: CasADi evaluation of f_k / L_k and their derivatives
 * inside pyddp; problem definition prb.py:92-110, :141-204).  nk knots, node index k[i] (k==N: terminal).
 *   f_out [nk][nx], F_out [nk][nx][nx+nu] (= [fx fu]), H_out [nk][nz][nz] (GN Hessian of L), g_out [nk][nz], L_out [nk] */
/* (with consts->n_extra > 0: p is [nk][np + SDDP_MAX_EXTRA]) */
int  sddp_eval_knots(int model_id, const sddp_model_consts* consts, int N, int nk, const int* k,
                     const double* x, const double* u, const double* p,
                     double* f_out, double* F_out, double* H_out, double* g_out, double* L_out);
/* one backward sweep on the handle's current trajectory: gains [B][N][nu*(nx+1)] (kff then K row-major),
 * scal [B][8] = dV1, dV2, G1, G2, ok, mu, qu_inf, Vx0[0] */
int  sddp_backward(sddp_handle* h, const double* params, double mu, double* gains_out, double* scal_out);
/* one forward pass at step `alpha` from the handle's trajectory and the gains of the last sddp_backward */
int  sddp_forward(sddp_handle* h, const double* params, double alpha, double* x_out, double* u_out, double* cost_out);

#ifdef __cplusplus
}
#endif
#endif /* SDDP_H */
