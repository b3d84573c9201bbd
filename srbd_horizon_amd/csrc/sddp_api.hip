// sddp_api.hip -- C ABI (include/sddp.h) over the HIP kernels.  Host side of the drop-in boundary that replaces
// the `pyddp` surface bound by the reference adapter (python/ddp.py:93-94, :101, :106, :113-123).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "sddp.h"
#include "sddp_handle.hpp"
#include "sddp_kernels_host.hpp"
#include "sddp_sort.hpp"

using namespace sddp;

// the model builds of the library (sddp_inst.hip, one translation unit each; srbd_horizon_amd/_lib.py INSTANCES)
namespace sddp {
const ModelOps* ops_srbd13();
const ModelOps* ops_srbd13_b();
const ModelOps* ops_srbd13_s();
const ModelOps* ops_srbd13_bs();
const ModelOps* ops_srbd37();
const ModelOps* ops_srbd37_b();
const ModelOps* ops_srbd37_s();
const ModelOps* ops_srbd37_bs();
const ModelOps* ops_lip30();
const ModelOps* ops_srbd61();
const ModelOps* ops_srbd13_x();
const ModelOps* ops_srbd37_x();
const ModelOps* ops_srbd61_x();
const ModelOps* ops_srbd61_b();
const ModelOps* ops_lip30_x();

std::string& create_error() {
    thread_local std::string e;
    return e;
}

int launch_queue_order(sddp_handle* h, int first, int count) {
    hipLaunchKernelGGL(queue_order_kernel, dim3(1), dim3(1024), 0, h->stream, first, count, h->hist, h->order);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}

int launch_class_keys(sddp_handle* h, int count) {
    hipLaunchKernelGGL(class_key_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, count, h->order_in, h->cls, h->n_cls, h->cls_stat,
                       h->qkey);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}
int launch_class_update(sddp_handle* h, int first, int count) {
    hipLaunchKernelGGL(class_update_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, first, count, h->cls, h->n_cls, h->stats,
                       h->cls_stat);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}

int alloc_cold_queue(sddp_handle* h) {
    if (h->sort_tmp) return SDDP_OK;               // the last pointer of the group: set only when all of it exists
    double *k1 = nullptr, *k2 = nullptr;
    int* oi = nullptr;
    void* tmp = nullptr;
    size_t tb = 0;
    hipError_t e = hipMalloc((void**)&k1, size_t(h->B) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&k2, size_t(h->B) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&oi, size_t(h->B) * sizeof(int));
    if (e == hipSuccess) e = sort_pairs_desc_temp_bytes(h->B, &tb);
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tb, 16));
    if (e != hipSuccess) {
        if (k1) (void)hipFree(k1);
        if (k2) (void)hipFree(k2);
        if (oi) (void)hipFree(oi);
        return fail(h, SDDP_ERR_HIP, std::string("cold-queue buffers: ") + hipGetErrorString(e));
    }
    h->qkey = k1; h->qkey2 = k2; h->order_in = oi; h->sort_tmp_bytes = tb; h->sort_tmp = tmp;
    return SDDP_OK;
}
}  // namespace sddp

namespace {

// bar: the barrier build of the SRBD models (friction cone and / or variable bounds), so2: the full second-order build
const ModelOps* model_ops(int id, bool bar = false, bool so2 = false, bool xr = false) {
    if (xr) {     // user rows: plain builds only
        if (bar || so2) return nullptr;
        switch (id) {
            case SDDP_MODEL_SRBD13: return ops_srbd13_x();
            case SDDP_MODEL_SRBD37: return ops_srbd37_x();
            case SDDP_MODEL_LIP30: return ops_lip30_x();
            case SDDP_MODEL_SRBD61: return ops_srbd61_x();
            default: return nullptr;
        }
    }
    switch (id) {
        case SDDP_MODEL_SRBD13: return so2 ? (bar ? ops_srbd13_bs() : ops_srbd13_s()) : (bar ? ops_srbd13_b() : ops_srbd13());
        case SDDP_MODEL_SRBD37: return so2 ? (bar ? ops_srbd37_bs() : ops_srbd37_s()) : (bar ? ops_srbd37_b() : ops_srbd37());
        case SDDP_MODEL_LIP30: return ops_lip30();
        case SDDP_MODEL_SRBD61: return so2 ? nullptr : (bar ? ops_srbd61_b() : ops_srbd61());   // no second_order = 2 build: LDS is full
        default: return nullptr;
    }
}
// models whose only build is the default one: linear-quadratic (lip30), or no barrier / second_order = 2 build instantiated (srbd61)
bool single_build(int id) { return id == SDDP_MODEL_LIP30; }

// user rows of a constants struct: validation, and the device table of DevConsts::xr
const char* check_extra(const sddp_model_consts& c, int nx, int nu) {
    if (c.n_extra < 0 || c.n_extra > SDDP_MAX_EXTRA) return "n_extra must be 0..8";
    for (int j = 0; j < c.n_extra; ++j) {
        if (c.extra_kind[j] != 0 && c.extra_kind[j] != 1) return "extra_kind must be 0 (state row) or 1 (stage row)";
        if (!(c.extra_weight[j] >= 0.0) || !std::isfinite(c.extra_weight[j]) || !std::isfinite(c.extra_const[j])) return "extra_weight must be finite and >= 0, extra_const finite";
        for (int i = 0; i < 128; ++i) {
            const double v = c.extra_a[128 * j + i];
            if (!std::isfinite(v)) return "extra_a must be finite";
            if (i >= nx + nu && v != 0.0) return "extra_a: entries past nx + nu must be 0";
            if (c.extra_kind[j] == 0 && i >= nx && v != 0.0) return "a state row (extra_kind 0) cannot touch the inputs: it is active at the terminal node";
        }
    }
    return nullptr;
}
void fill_extra_table(const sddp_model_consts& c, double* t /*[kXrWords]*/) {
    std::memset(t, 0, sizeof(double) * kXrWords);
    for (int j = 0; j < c.n_extra; ++j) {
        for (int i = 0; i < kXrStride; ++i) t[j * kXrStride + i] = c.extra_a[128 * j + i];
        t[(c.extra_kind[j] == 0 ? kXrWS : kXrWG) + j] = c.extra_weight[j];
        t[kXrC + j] = c.extra_const[j];
    }
}

// host scratch of a call: plain malloc (no exception can cross the C boundary), freed on every return path
struct host_buf {
    void* p;
    explicit host_buf(size_t bytes) : p(std::malloc(bytes ? bytes : 1)) {}
    ~host_buf() { std::free(p); }
    host_buf(const host_buf&) = delete;
    host_buf& operator=(const host_buf&) = delete;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

SolveArgs make_args(sddp_handle* h, const double* d_params) {
    SolveArgs a;
    a.c = h->dc; a.o = h->opts; a.N = h->N; a.B = h->B;
    a.x0 = h->x0; a.P = d_params; a.xs = h->xs; a.us = h->us; a.xn = h->xn; a.un = h->un; a.xc = h->xc; a.uc = h->uc; a.dft = h->dft;
    a.gains = h->gains; a.rec = h->rec; a.stats = h->stats; a.scal = h->scal; a.alpha = 0.0; a.mu = 0.0;
    a.first = 0; a.count = h->B; a.qhead = nullptr; a.order = nullptr; a.hist = h->hist;
    return a;
}

constexpr size_t kUpRing = size_t(256) << 10;

int check_ready(sddp_handle* h) {
    if (!h) return SDDP_ERR_ARG;
    if (!h->have_x0) return fail(h, SDDP_ERR_ARG, "sddp_set_initial_state has not been called");
    if (!h->have_uws) return fail(h, SDDP_ERR_ARG, "sddp_set_u_warmstart has not been called");
    if (!h->have_xws && !h->opts.initial_rollout)
        return fail(h, SDDP_ERR_ARG, "sddp_set_x_warmstart has not been called (multiple shooting)");
    return SDDP_OK;
}

int validate_options(sddp_handle* h, const sddp_options& o) {
    if (o.max_iters < 0) return fail(h, SDDP_ERR_ARG, "max_iters < 0");
    if (!(o.line_search_decrease_factor > 0.0 && o.line_search_decrease_factor < 1.0))
        return fail(h, SDDP_ERR_ARG, "line_search_decrease_factor must be in (0,1)");
    if (!(o.alpha_0 > 0.0)) return fail(h, SDDP_ERR_ARG, "alpha_0 must be > 0");
    if (!(o.alpha_converge_threshold > 0.0)) return fail(h, SDDP_ERR_ARG, "alpha_converge_threshold must be > 0");
    if (!(o.mu_min > 0.0)) return fail(h, SDDP_ERR_ARG, "mu_min must be > 0");
    if (o.waves_per_simd != 1 && o.waves_per_simd != 2) return fail(h, SDDP_ERR_ARG, "waves_per_simd must be 1 or 2");
    // non-finite values would turn the regularisation loop or the line-search ladder of the persistent kernel into an endless loop
    const double fin[] = {o.alpha_0, o.alpha_converge_threshold, o.line_search_decrease_factor, o.beta, o.cost_reduction_ths, o.mu0,
                          o.gap_tol, o.mu_min, o.mu_max};
    for (double v : fin)
        if (!std::isfinite(v)) return fail(h, SDDP_ERR_ARG, "non-finite value in sddp_options");
    if (!(o.mu_max > o.mu_min)) return fail(h, SDDP_ERR_ARG, "mu_max must be > mu_min");
    // the line-search ladder alpha_0 * factor^j down to alpha_converge_threshold is rolled out inside the kernel: bound its length
    if (o.alpha_converge_threshold < o.alpha_0 &&
        std::log(o.alpha_converge_threshold / o.alpha_0) / std::log(o.line_search_decrease_factor) > 4096.0)
        return fail(h, SDDP_ERR_ARG, "line search ladder longer than 4096 step lengths (line_search_decrease_factor too close to 1 "
                                     "or alpha_converge_threshold too small)");
    if (o.queue_order < 0 || o.queue_order > 3) return fail(h, SDDP_ERR_ARG, "queue_order must be 0, 1, 2 or 3");
    if (o.max_slots < 0) return fail(h, SDDP_ERR_ARG, "max_slots must be >= 0");
    if (o.second_order < 0 || o.second_order > 2) return fail(h, SDDP_ERR_ARG, "second_order must be 0, 1 or 2");
    return SDDP_OK;
}

}  // namespace

extern "C" {

int sddp_abi_version(void) { return SDDP_ABI_VERSION; }

int sddp_model_dims(int model_id, int* nx, int* nu, int* np) {
    const ModelOps* ops = model_ops(model_id);
    if (!ops) return SDDP_ERR_MODEL;
    const Dims d = ops->dims;
    if (nx) *nx = d.nx;
    if (nu) *nu = d.nu;
    if (np) *np = d.np;
    return SDDP_OK;
}

int sddp_handle_dims(sddp_handle* h, int* nx, int* nu, int* np) {
    if (!h) return SDDP_ERR_ARG;
    if (nx) *nx = h->d.nx;
    if (nu) *nu = h->d.nu;
    if (np) *np = h->d.np;
    return SDDP_OK;
}

void sddp_default_options(sddp_options* o) {
    if (!o) return;
    o->max_iters = 100;                      // ddp.py:17
    o->alpha_0 = 1.0;                        // ddp.py:20
    o->alpha_converge_threshold = 1e-1;      // ddp.py:23
    o->line_search_decrease_factor = 0.5;    // ddp.py:26
    o->beta = 1e-4;                          // ddp.py:29
    o->cost_reduction_ths = 1e-6;            // engine default (unpinned upstream)
    o->mu0 = 0.0;                            // engine default (unpinned upstream)
    o->initial_rollout = 0;
    o->gap_tol = 1e-9;
    o->mu_min = 1e-6;
    o->mu_max = 1e12;
    o->second_order = 1;
    o->waves_per_simd = 1;
    o->queue_order = 1;
    o->max_slots = 0;
}

void sddp_default_consts(sddp_model_consts* c) { sddp_default_consts_for(SDDP_MODEL_SRBD37, c); }

int sddp_default_consts_for(int model_id, sddp_model_consts* c) {
    if (!c) return SDDP_ERR_ARG;
    if (!model_ops(model_id)) return SDDP_ERR_MODEL;
    std::memset(c, 0, sizeof(*c));
    c->m = 40.0;
    const double I[9] = {2.0, 0.03, -0.02, 0.03, 1.8, 0.04, -0.02, 0.04, 0.6};
    std::memcpy(c->I, I, sizeof(I));
    c->com[0] = 0.0; c->com[1] = 0.0; c->com[2] = 0.88;
    // contact points 0..3: the line feet (launch:24-25), or the first four sole corners of contact_model = 4 (prb.py:39-41) --
    // the points d_initial_1/2 of prb.py:153-154 name whatever nc is
    const double feet[12] = {0.08, 0.1, 0.0, -0.08, 0.1, 0.0, 0.08, -0.1, 0.0, -0.08, -0.1, 0.0};
    const double feet8[12] = {0.08, 0.13, 0.0, -0.08, 0.13, 0.0, 0.08, 0.07, 0.0, -0.08, 0.07, 0.0};
    std::memcpy(c->feet, model_id == SDDP_MODEL_SRBD61 ? feet8 : feet, sizeof(feet));
    c->dt = 0.05;
    c->force_scaling = 1000.0;
    c->r_tracking_gain = 1e3; c->rdot_tracking_gain = 1e4; c->w_tracking_gain = 1e4; c->rel_pos_gain = 1e4;
    c->force_switch_weight = 1e2; c->min_qddot_gain = 1e0; c->min_f_gain = 1e-2; c->zmp_tracking_gain = 1e3;
    c->lip_height = 0.88;
    c->inertia_mode = 0;
    c->lever_sign = 1.0;
    c->friction_cone_coefficient = 0.8;      // prb.py:174
    c->friction_barrier_weight = 0.0;        // off: the reference ignores its inequality constraints (ddp.py:197-209)
    c->friction_barrier_sharpness = 1.0;
    c->bound_barrier_weight = 0.0;           // off: the reference's bound barriers are commented out (ddp.py:203-208)
    c->bound_barrier_sharpness = 6.0;        // exp_parameter, ddp.py:182
    for (int i = 0; i < 64; ++i) { c->lower[i] = -HUGE_VAL; c->upper[i] = HUGE_VAL; }
    c->relative_velocity_constraints = 1;    // contact_model > 1 (prb.py:166)
    return SDDP_OK;
}

const char* sddp_last_error(const sddp_handle* h) { return h ? h->err.c_str() : create_error().c_str(); }

int sddp_create(sddp_handle** out, int model_id, int N, int batch, const sddp_options* opts, const sddp_model_consts* consts) {
    if (!out) return fail(nullptr, SDDP_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!model_ops(model_id)) return fail(nullptr, SDDP_ERR_MODEL, "unknown model_id");
    if (consts && consts->bound_barrier_weight > 0.0 && (model_id == SDDP_MODEL_LIP30 || model_id == SDDP_MODEL_SRBD61))
        return fail(nullptr, SDDP_ERR_ARG, "bound_barrier_weight > 0: the bound barrier exists for srbd13 and srbd37 only (lower / upper hold 64 "
                                           "entries of z; srbd61 has 109)");
    // barrier builds: the friction-cone barrier and / or the bound barrier
    const bool bar = consts && (consts->friction_barrier_weight > 0.0 || consts->bound_barrier_weight > 0.0) && !single_build(model_id);
    const bool so2 = opts && opts->second_order == 2 && !single_build(model_id);     // (the LIP model is linear-quadratic: nothing to add)
    const bool xr = consts && consts->n_extra != 0;
    const ModelOps* ops = model_ops(model_id, bar, so2, xr);
    if (!ops) return fail(nullptr, SDDP_ERR_ARG, xr ? "user rows (n_extra > 0) exist for the plain builds only (no barrier, no second_order = 2)"
                                                    : "this model has no such build (srbd61: no second_order = 2 build)");
    const Dims d = ops->dims;
    if (N < 1 || batch < 1) return fail(nullptr, SDDP_ERR_ARG, "N and batch must be >= 1");
    if (xr) { const char* msg = check_extra(*consts, d.nx, d.nu); if (msg) return fail(nullptr, SDDP_ERR_ARG, msg); }
    if (consts && (consts->friction_barrier_weight < 0.0 || (consts->friction_barrier_weight > 0.0 && !(consts->friction_cone_coefficient > 0.0))))
        return fail(nullptr, SDDP_ERR_ARG, "friction_barrier_weight must be >= 0 and friction_cone_coefficient > 0");
    if (consts && (consts->bound_barrier_weight < 0.0 || !std::isfinite(consts->bound_barrier_weight) ||
                   (consts->bound_barrier_weight > 0.0 && !(consts->bound_barrier_sharpness > 0.0 && std::isfinite(consts->bound_barrier_sharpness)))))
        return fail(nullptr, SDDP_ERR_ARG, "bound_barrier_weight must be >= 0 and bound_barrier_sharpness > 0");
    if (consts && consts->bound_barrier_weight > 0.0)
        for (int j = 0; j < d.nx + d.nu; ++j)
            if (std::isnan(consts->lower[j]) || std::isnan(consts->upper[j]) || !(consts->lower[j] < consts->upper[j]))
                return fail(nullptr, SDDP_ERR_ARG, "bound barrier: lower[j] < upper[j] is required for every j < nx + nu");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, SDDP_ERR_HIP, "no HIP device visible: the SDDP engine has no CPU fallback");
    sddp_handle* h = new (std::nothrow) sddp_handle();
    if (!h) return fail(nullptr, SDDP_ERR_NOMEM, "out of host memory");
    h->model_id = model_id; h->N = N; h->B = batch; h->d = d; h->ops = ops; h->bar = bar; h->so2 = so2;
    if (opts) h->opts = *opts; else sddp_default_options(&h->opts);
    if (consts) h->consts = *consts; else sddp_default_consts_for(model_id, &h->consts);
    int rc = validate_options(h, h->opts);
    if (rc != SDDP_OK) { create_error() = h->err; delete h; return rc; }
    h->dc = make_dev_consts(h->consts);
    auto alloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes); };
    hipError_t e = hipSuccess;
    if (bar) {   // the bounds of the bound barrier live in device memory (DevConsts::box)
        double hb[128];
        for (int i = 0; i < 64; ++i) { hb[i] = h->consts.lower[i]; hb[64 + i] = h->consts.upper[i]; }
        e = alloc((void**)&h->box_dev, sizeof(hb));
        if (e == hipSuccess) e = hipMemcpy(h->box_dev, hb, sizeof(hb), hipMemcpyHostToDevice);
        h->dc.box = h->box_dev;
    }
    if (xr) {    // the user rows' coefficients, weights and constants live in device memory (DevConsts::xr)
        double t[kXrWords];
        fill_extra_table(h->consts, t);
        if (e == hipSuccess) e = alloc((void**)&h->xr_dev, sizeof(t));
        if (e == hipSuccess) e = hipMemcpy(h->xr_dev, t, sizeof(t), hipMemcpyHostToDevice);
        h->dc.xr = h->xr_dev;
        h->dc.xr_n = h->consts.n_extra;
    }
    const size_t D = sizeof(double);
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) h->cus = prop.multiProcessorCount;
        else h->cus = 1;
    }
    // resident capacity of the solve kernel(s) on this device = number of queue slots; the work buffers exist per slot
    int slots = 0;
    rc = ops->max_slots(h, &slots);
    if (rc != SDDP_OK) { create_error() = h->err; sddp_destroy(h); return rc; }
    h->wslots = std::min(batch, slots);
    if (h->opts.max_slots > 0) h->wslots = std::min(h->wslots, h->opts.max_slots);
    const size_t W = size_t(h->wslots);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    h->own_stream = (e == hipSuccess);
    if (e == hipSuccess) e = alloc((void**)&h->x0, size_t(batch) * d.nx * D);
    if (e == hipSuccess) e = alloc((void**)&h->P, h->n_p() * D);
    // xs | us | stats in ONE allocation: the results of a solve leave the device in one copy (a tick of one robot is three API
    // calls shorter)
    if (e == hipSuccess) e = alloc((void**)&h->xs, (h->n_x() + h->n_u()) * D + size_t(batch) * sizeof(sddp_stats));
    if (e == hipSuccess) { h->us = h->xs + h->n_x(); h->stats = reinterpret_cast<sddp_stats*>(h->us + h->n_u()); }
    if (e == hipSuccess) e = alloc((void**)&h->xn, W * (N + 1) * d.nx * D);
    if (e == hipSuccess) e = alloc((void**)&h->un, W * N * d.nu * D);
    if (!ops->uses_mw) {   // one-wave kernel: two sets of kSlots line-search candidates per slot
        if (e == hipSuccess) e = alloc((void**)&h->xc, W * (N + 1) * d.nx * 2 * kSlots * D);
        if (e == hipSuccess) e = alloc((void**)&h->uc, W * N * d.nu * 2 * kSlots * D);
    }
    if (e == hipSuccess) e = alloc((void**)&h->dft, W * N * d.nx * D);
    if (e == hipSuccess) e = alloc((void**)&h->gains, W * N * d.nu * (d.nx + 1) * D);
    if (e == hipSuccess) e = alloc((void**)&h->rec, W * (N + 1) * d.nrec * D);
    if (e == hipSuccess) e = alloc((void**)&h->scal, size_t(batch) * kScal * D);
    if (e == hipSuccess) e = alloc((void**)&h->qhead, sizeof(int));
    if (e == hipSuccess) e = alloc((void**)&h->order, size_t(batch) * sizeof(int));
    // hist [B] (padded to a multiple of two ints), then the slot clocks [W][2] uint64 (SolveArgs::slot_clock)
    if (e == hipSuccess) e = alloc((void**)&h->hist, size_t((batch + 1) & ~1) * sizeof(int) + std::max<size_t>(W, 1) * 2 * sizeof(unsigned long long));
    // on the handle's own stream, and complete before sddp_create returns: a null-stream hipMemset is asynchronous to the host
    // and is NOT ordered with a non-blocking stream, so it could land in the middle of the first solve (seen once as a
    // different iteration count on a 1-knot problem)
    if (e == hipSuccess) e = hipMemsetAsync(h->stats, 0, size_t(batch) * sizeof(sddp_stats), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->dft, 0, W * N * d.nx * D, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->gains, 0, W * N * d.nu * (d.nx + 1) * D, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->hist, 0xff, size_t(batch) * sizeof(int), h->stream);      // -1: never solved
    if (e == hipSuccess) e = hipMemsetAsync(h->qhead, 0, sizeof(int), h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        create_error() = std::string("sddp_create: ") + hipGetErrorString(e);
        sddp_destroy(h);
        return SDDP_ERR_HIP;
    }
    if (h->opts.queue_order >= 2) {   // cold-queue order: its key / sort buffers exist before the first launch (all or nothing)
        rc = alloc_cold_queue(h);
        if (rc != SDDP_OK) { create_error() = h->err; sddp_destroy(h); return rc; }
    }
    *out = h;
    return SDDP_OK;
}

void sddp_destroy(sddp_handle* h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* bufs[] = {h->x0, h->P, h->xs /* | us | stats */, h->xn, h->un, h->xc, h->uc, h->tick_in, h->step_buf, h->dft, h->gains, h->rec, h->scal,
                    h->qhead, h->order, h->hist, h->qkey, h->qkey2, h->order_in, h->sort_tmp};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->step_pin) (void)hipHostFree(h->step_pin);
    if (h->tick_pin) (void)hipHostFree(h->tick_pin);
    if (h->up_pin) (void)hipHostFree(h->up_pin);
    if (h->first_pin) (void)hipHostFree(h->first_pin);
    if (h->first_dev) (void)hipFree(h->first_dev);
    if (h->box_dev) (void)hipFree(h->box_dev);
    if (h->xr_dev) (void)hipFree(h->xr_dev);
    if (h->cls) (void)hipFree(h->cls);
    if (h->cls_stat) (void)hipFree(h->cls_stat);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int sddp_set_options(sddp_handle* h, const sddp_options* opts) {
    if (!h || !opts) return SDDP_ERR_ARG;
    int rc = validate_options(h, *opts);
    if (rc != SDDP_OK) return rc;
    if ((opts->second_order == 2) != (h->opts.second_order == 2) && h->model_id != SDDP_MODEL_LIP30)
        return fail(h, SDDP_ERR_ARG, "second_order = 2 selects another kernel build and record size: choose it at sddp_create");
    h->opts = *opts;
    return SDDP_OK;
}

int sddp_set_stream(sddp_handle* h, void* s) {
    if (!h) return SDDP_ERR_ARG;
    if (h->stream) (void)hipStreamSynchronize(h->stream);      // uploads enqueued from the pinned ring belong to the old stream
    h->up_off = 0;
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = reinterpret_cast<hipStream_t>(s);
    h->own_stream = false;
    return SDDP_OK;
}

// small host->device upload without a wait: the bytes are copied into a pinned ring and the transfer is enqueued on the handle's
// stream (a copy from pageable memory would have to be waited for before the caller may reuse its buffer).  The ring is reused
// from the start after a stream synchronisation; large uploads go directly and are waited for.
static int upload(sddp_handle* h, void* dst, const void* src, size_t bytes) {
    if (bytes <= kUpRing / 4) {
        if (!h->up_pin && hipHostMalloc((void**)&h->up_pin, kUpRing, hipHostMallocDefault) != hipSuccess) {
            h->up_pin = nullptr;
            (void)hipGetLastError();
        }
        if (h->up_pin) {
            const size_t need = (bytes + 63) & ~size_t(63);
            if (h->up_off + need > kUpRing) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                h->up_off = 0;
            }
            std::memcpy(h->up_pin + h->up_off, src, bytes);
            HIP_TRY(h, hipMemcpyAsync(dst, h->up_pin + h->up_off, bytes, hipMemcpyHostToDevice, h->stream));
            h->up_off += need;
            return SDDP_OK;
        }
    }
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->up_off = 0;
    return SDDP_OK;
}

// ---- host-pointer setters -------------------------------------------------------------------------------------------
int sddp_set_initial_state(sddp_handle* h, const double* x0) {
    if (!h || !x0) return SDDP_ERR_ARG;
    const int rc = upload(h, h->x0, x0, size_t(h->B) * h->d.nx * sizeof(double));
    if (rc == SDDP_OK) h->have_x0 = true;
    return rc;
}
int sddp_set_x_warmstart(sddp_handle* h, const double* x) {
    if (!h || !x) return SDDP_ERR_ARG;
    const int rc = upload(h, h->xs, x, h->n_x() * sizeof(double));
    if (rc == SDDP_OK) h->have_xws = true;
    return rc;
}
int sddp_set_u_warmstart(sddp_handle* h, const double* u) {
    if (!h || !u) return SDDP_ERR_ARG;
    const int rc = upload(h, h->us, u, h->n_u() * sizeof(double));
    if (rc == SDDP_OK) h->have_uws = true;
    return rc;
}
// ---- device-pointer setters -----------------------------------------------------------------------------------------
int sddp_load_range_device(sddp_handle* h, int first, int count, const double* d_x0, const double* d_x, const double* d_u) {
    if (!h) return SDDP_ERR_ARG;
    if (first < 0 || count < 1 || first > h->B - count) return fail(h, SDDP_ERR_ARG, "instance range outside the batch");
    const size_t D = sizeof(double), nx = h->d.nx, nu = h->d.nu, N = h->N;
    if (d_x0) {
        HIP_TRY(h, hipMemcpyAsync(h->x0 + size_t(first) * nx, d_x0, size_t(count) * nx * D, hipMemcpyDeviceToDevice, h->stream));
        h->have_x0 = true;
    }
    if (d_x) {
        HIP_TRY(h, hipMemcpyAsync(h->xs + size_t(first) * (N + 1) * nx, d_x, size_t(count) * (N + 1) * nx * D, hipMemcpyDeviceToDevice, h->stream));
        h->have_xws = true;
    }
    if (d_u) {
        HIP_TRY(h, hipMemcpyAsync(h->us + size_t(first) * N * nu, d_u, size_t(count) * N * nu * D, hipMemcpyDeviceToDevice, h->stream));
        h->have_uws = true;
    }
    return SDDP_OK;
}
int sddp_set_initial_state_device(sddp_handle* h, const double* d) {
    if (!h || !d) return SDDP_ERR_ARG;
    return sddp_load_range_device(h, 0, h->B, d, nullptr, nullptr);
}
int sddp_set_x_warmstart_device(sddp_handle* h, const double* d) {
    if (!h || !d) return SDDP_ERR_ARG;
    return sddp_load_range_device(h, 0, h->B, nullptr, d, nullptr);
}
int sddp_set_u_warmstart_device(sddp_handle* h, const double* d) {
    if (!h || !d) return SDDP_ERR_ARG;
    return sddp_load_range_device(h, 0, h->B, nullptr, nullptr, d);
}

int sddp_solve_range_device(sddp_handle* h, const double* d_params, int first, int count) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!d_params) return fail(h, SDDP_ERR_ARG, "params is NULL");
    if (first < 0 || count < 1 || first > h->B - count) return fail(h, SDDP_ERR_ARG, "instance range outside the batch");
    SolveArgs a = make_args(h, d_params);
    rc = h->ops->launch_solve(h, a, first, count);
    return rc;
}

int sddp_solve_device(sddp_handle* h, const double* d_params) {
    if (!h) return SDDP_ERR_ARG;
    return sddp_solve_range_device(h, d_params, 0, h->B);
}

int sddp_queue_info(sddp_handle* h, int* slots, int* last_grid, int* last_queued) {
    if (!h) return SDDP_ERR_ARG;
    if (slots) *slots = h->wslots;
    if (last_grid) *last_grid = h->last_grid;
    if (last_queued) *last_queued = h->last_queued;
    return SDDP_OK;
}

int sddp_kernel_info(sddp_handle* h, int* wavefronts_per_instance, int* last_waves_per_simd, const char** model_name) {
    if (!h) return SDDP_ERR_ARG;
    if (wavefronts_per_instance) *wavefronts_per_instance = h->ops->uses_mw ? 4 : 1;
    if (last_waves_per_simd) *last_waves_per_simd = h->last_build;
    if (model_name) *model_name = h->ops->name;
    return SDDP_OK;
}

int sddp_kernel_resources(sddp_handle* h, int* vgprs, int* scratch_bytes_per_lane, int* lds_bytes, int* workgroups_per_cu) {
    if (!h) return SDDP_ERR_ARG;
    if (!h->last_kernel) return fail(h, SDDP_ERR_ARG, "no solve launch yet");
    hipFuncAttributes at;
    HIP_TRY(h, hipFuncGetAttributes(&at, h->last_kernel));
    if (vgprs) *vgprs = at.numRegs;
    if (scratch_bytes_per_lane) *scratch_bytes_per_lane = int(at.localSizeBytes);
    if (lds_bytes) *lds_bytes = h->last_lds + int(at.sharedSizeBytes);
    if (workgroups_per_cu) *workgroups_per_cu = h->last_per_cu;
    return SDDP_OK;
}

int sddp_debug_poison_lds(sddp_handle* h) {
    if (!h) return SDDP_ERR_ARG;
    int dev = 0;
    hipDeviceProp_t pr;
    HIP_TRY(h, hipGetDevice(&dev));
    HIP_TRY(h, hipGetDeviceProperties(&pr, dev));
    const int bytes = int(pr.maxSharedMemoryPerMultiProcessor);            // 160 KB on gfx950: one such workgroup per CU at a time
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(8 * pr.multiProcessorCount), dim3(256), bytes, h->stream, bytes / 8, h->hist);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SDDP_OK;
}

int sddp_synchronize(sddp_handle* h) {
    if (!h) return SDDP_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->tick_unsynced = 0;
    h->up_off = 0;
    for (size_t i = 0; i < h->pending; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) == hipSuccess) {
            h->last_ms = ms;
            h->sum_ms += ms;
            ++h->n_ms;
        }
    }
    h->pending = 0;
    return SDDP_OK;
}

// results of the last solve to host pointers.  Small batches (one robot, a handful of robots) go through a pinned staging buffer:
// copies into pageable memory block one by one, copies into pinned memory are enqueued and waited for once.
static int fetch_results(sddp_handle* h, double* x_out, double* u_out, sddp_stats* stats) {
    if (!x_out || !u_out) return fail(h, SDDP_ERR_ARG, "NULL argument");
    const size_t bx = h->n_x() * sizeof(double), bu = h->n_u() * sizeof(double), bs = size_t(h->B) * sizeof(sddp_stats);
    if (!h->pinned && bx + bu + bs <= (size_t(256) << 10)) {
        if (hipHostMalloc(&h->pinned, bx + bu + bs, hipHostMallocDefault) == hipSuccess) h->pinned_bytes = bx + bu + bs;
        else { h->pinned = nullptr; (void)hipGetLastError(); }
    }
    if (h->pinned && h->pinned_bytes >= bx + bu + bs) {
        char* st = static_cast<char*>(h->pinned);
        HIP_TRY(h, hipMemcpyAsync(st, h->xs, bx + bu + (stats ? bs : 0), hipMemcpyDeviceToHost, h->stream));   // xs | us | stats are contiguous
        const int rc = sddp_synchronize(h);
        if (rc != SDDP_OK) return rc;
        std::memcpy(x_out, st, bx);
        std::memcpy(u_out, st + bx, bu);
        if (stats) std::memcpy(stats, st + bx + bu, bs);
        return SDDP_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(x_out, h->xs, bx, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(u_out, h->us, bu, hipMemcpyDeviceToHost, h->stream));
    if (stats) HIP_TRY(h, hipMemcpyAsync(stats, h->stats, bs, hipMemcpyDeviceToHost, h->stream));
    return sddp_synchronize(h);
}

int sddp_fetch(sddp_handle* h, double* x_out, double* u_out, sddp_stats* stats) {
    if (!h) return SDDP_ERR_ARG;
    return fetch_results(h, x_out, u_out, stats);
}

int sddp_solve(sddp_handle* h, const double* params, double* x_out, double* u_out, sddp_stats* stats) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!params || !x_out || !u_out) return fail(h, SDDP_ERR_ARG, "NULL argument");
    rc = upload(h, h->P, params, h->n_p() * sizeof(double));
    if (rc != SDDP_OK) return rc;
    rc = sddp_solve_device(h, h->P);
    if (rc != SDDP_OK) return rc;
    rc = fetch_results(h, x_out, u_out, stats);
    // the solution is the next warm start unless the caller overrides it (solver object persists across ticks,
    // dsrbd_example.py:59)
    h->have_xws = true;
    return rc;
}

int sddp_set_params(sddp_handle* h, const double* params) {
    if (!h || !params) return SDDP_ERR_ARG;
    const int rc = upload(h, h->P, params, h->n_p() * sizeof(double));
    if (rc == SDDP_OK) h->have_params = true;
    return rc;
}

int sddp_advance(sddp_handle* h, const double* p_last, const double* x0) {
    if (!h || !p_last || !x0) return SDDP_ERR_ARG;
    if (!h->have_params) return fail(h, SDDP_ERR_ARG, "sddp_set_params has not been called");
    if (!h->have_xws || !h->have_uws) return fail(h, SDDP_ERR_ARG, "sddp_advance needs a previous solution or warm start");
    if ((h->N + 1) * std::max(h->d.np, h->d.nx) > kAdvanceWords) return fail(h, SDDP_ERR_ARG, "horizon too long for sddp_advance");
    if (!h->tick_in) HIP_TRY(h, hipMalloc((void**)&h->tick_in, size_t(h->B) * (h->d.np + h->d.nx) * sizeof(double)));
    double* d_pl = h->tick_in;
    double* d_x0 = h->tick_in + size_t(h->B) * h->d.np;
    const size_t bp = size_t(h->B) * h->d.np * sizeof(double), bx0 = size_t(h->B) * h->d.nx * sizeof(double);
    if (!h->tick_pin && bp + bx0 <= (size_t(256) << 10) && hipHostMalloc((void**)&h->tick_pin, 2 * (bp + bx0), hipHostMallocDefault) != hipSuccess) {
        h->tick_pin = nullptr;
        (void)hipGetLastError();
    }
    if (h->tick_pin) {   // one enqueued upload from pinned memory; two alternating images, so that the call need not wait for it
        if (h->tick_unsynced >= 2) {   // both images may still be read by earlier uploads: wait (a solve in between does it anyway)
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            h->tick_unsynced = 0;
        }
        ++h->tick_unsynced;
        char* hp = h->tick_pin + (h->tick_flip ^= 1) * (bp + bx0);
        std::memcpy(hp, p_last, bp);
        std::memcpy(hp + bp, x0, bx0);
        HIP_TRY(h, hipMemcpyAsync(d_pl, hp, bp + bx0, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(d_pl, p_last, bp, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(d_x0, x0, bx0, hipMemcpyHostToDevice, h->stream));
    }
    hipLaunchKernelGGL(advance_kernel, dim3(h->B), dim3(256), 0, h->stream, h->N, h->d.nx, h->d.nu, h->d.np, h->P, h->xs, h->us,
                       h->x0, d_pl, d_x0);
    HIP_TRY(h, hipGetLastError());
    h->have_x0 = true;
    return SDDP_OK;
}

int sddp_solve_resident(sddp_handle* h, double* x_out, double* u_out, sddp_stats* stats) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!h->have_params) return fail(h, SDDP_ERR_ARG, "sddp_set_params has not been called");
    if (!x_out || !u_out) return fail(h, SDDP_ERR_ARG, "NULL argument");
    rc = sddp_solve_device(h, h->P);
    if (rc != SDDP_OK) return rc;
    rc = fetch_results(h, x_out, u_out, stats);
    h->have_xws = true;
    return rc;
}

int sddp_solve_resident_first(sddp_handle* h, double* u0_out, double* x1_out, double* cost_out, int* iters_out, int* status_out) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!h->have_params) return fail(h, SDDP_ERR_ARG, "sddp_set_params has not been called");
    if (!u0_out || !x1_out) return fail(h, SDDP_ERR_ARG, "NULL argument");
    rc = sddp_solve_device(h, h->P);
    if (rc != SDDP_OK) return rc;
    const int w = h->d.nu + h->d.nx + 3;
    const size_t bytes = size_t(h->B) * w * sizeof(double);
    if (!h->first_dev) {
        HIP_TRY(h, hipMalloc((void**)&h->first_dev, bytes));
        if (hipHostMalloc((void**)&h->first_pin, bytes, hipHostMallocDefault) != hipSuccess) { h->first_pin = nullptr; (void)hipGetLastError(); }
    }
    const int grid = int(std::min<size_t>((size_t(h->B) * w + 255) / 256, 1024));
    hipLaunchKernelGGL(first_knot_kernel, dim3(grid), dim3(256), 0, h->stream, h->N, h->B, h->d.nx, h->d.nu, h->xs, h->us, h->stats, h->first_dev);
    HIP_TRY(h, hipGetLastError());
    host_buf tmp(h->first_pin ? 0 : bytes);
    if (!tmp.p) return fail(h, SDDP_ERR_NOMEM, "out of host memory");
    double* host = h->first_pin ? h->first_pin : tmp.as<double>();
    HIP_TRY(h, hipMemcpyAsync(host, h->first_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    rc = sddp_synchronize(h);
    if (rc != SDDP_OK) return rc;
    for (int b = 0; b < h->B; ++b) {
        const double* r = host + size_t(b) * w;
        std::memcpy(u0_out + size_t(b) * h->d.nu, r, h->d.nu * sizeof(double));
        std::memcpy(x1_out + size_t(b) * h->d.nx, r + h->d.nu, h->d.nx * sizeof(double));
        if (cost_out) cost_out[b] = r[h->d.nu + h->d.nx];
        if (iters_out) iters_out[b] = int(r[h->d.nu + h->d.nx + 1]);
        if (status_out) status_out[b] = int(r[h->d.nu + h->d.nx + 2]);
    }
    h->have_xws = true;
    return SDDP_OK;
}

// class labels of the instances (queue_order = 3).  The class table is allocated on the first call; n_classes is fixed then.
static int class_buffers(sddp_handle* h, int n_classes) {
    if (n_classes < 1 || n_classes > (1 << 20)) return fail(h, SDDP_ERR_ARG, "n_classes must be 1 .. 2^20");
    if (h->cls) {
        if (n_classes != h->n_cls) return fail(h, SDDP_ERR_ARG, "n_classes differs from the first call's");
        return SDDP_OK;
    }
    int* c = nullptr;
    unsigned long long* st = nullptr;
    hipError_t e = hipMalloc((void**)&c, size_t(h->B) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&st, size_t(n_classes) * 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c, 0xFF, size_t(h->B) * sizeof(int), h->stream);       // -1: unlabelled
    if (e == hipSuccess) e = hipMemsetAsync(st, 0, size_t(n_classes) * 2 * sizeof(unsigned long long), h->stream);
    if (e != hipSuccess) {
        if (c) (void)hipFree(c);
        if (st) (void)hipFree(st);
        return fail(h, SDDP_ERR_HIP, std::string("class buffers: ") + hipGetErrorString(e));
    }
    h->cls = c; h->cls_stat = st; h->n_cls = n_classes;
    return SDDP_OK;
}

int sddp_set_instance_classes(sddp_handle* h, const int* classes, int n_classes) {
    if (!h || !classes) return SDDP_ERR_ARG;
    int rc = class_buffers(h, n_classes);
    if (rc != SDDP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->cls, classes, size_t(h->B) * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SDDP_OK;
}

int sddp_set_instance_classes_range_device(sddp_handle* h, int first, int count, const int* d_classes, int n_classes) {
    if (!h || !d_classes) return SDDP_ERR_ARG;
    if (first < 0 || count < 1 || first > h->B - count) return fail(h, SDDP_ERR_ARG, "instance range outside the batch");
    int rc = class_buffers(h, n_classes);
    if (rc != SDDP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->cls + first, d_classes, size_t(count) * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    return SDDP_OK;
}

int sddp_class_history(sddp_handle* h, int cls, double* mean_iters, long long* solves) {
    if (!h) return SDDP_ERR_ARG;
    if (!h->cls || cls < 0 || cls >= h->n_cls) return fail(h, SDDP_ERR_ARG, "no such class");
    unsigned long long st[2];
    HIP_TRY(h, hipMemcpyAsync(st, h->cls_stat + 2 * cls, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (solves) *solves = (long long)st[1];
    if (mean_iters) *mean_iters = st[1] ? double(st[0]) / double(st[1]) : 0.0;
    return SDDP_OK;
}

int sddp_record_words(sddp_handle* h, int mode, int* words) {
    if (!h || !words) return SDDP_ERR_ARG;
    if (mode != 0 && mode != 1) return fail(h, SDDP_ERR_ARG, "record mode must be 0 (whole plan) or 1 (first knot)");
    *words = mode == 0 ? (h->N + 1) * h->d.nx + h->N * h->d.nu + 2 : h->d.nu + h->d.nx + 2;
    return SDDP_OK;
}

int sddp_pack_records_device(sddp_handle* h, int first, int count, int mode, double* d_out) {
    if (!h || !d_out) return SDDP_ERR_ARG;
    if (mode != 0 && mode != 1) return fail(h, SDDP_ERR_ARG, "record mode must be 0 (whole plan) or 1 (first knot)");
    if (first < 0 || count < 1 || first > h->B - count) return fail(h, SDDP_ERR_ARG, "instance range outside the batch");
    int w = 0;
    sddp_record_words(h, mode, &w);
    const int grid = int(std::min<size_t>((size_t(count) * w + 255) / 256, 2048));
    hipLaunchKernelGGL(pack_records_kernel, dim3(grid), dim3(256), 0, h->stream, h->N, h->d.nx, h->d.nu, first, count, mode, h->xs, h->us,
                       h->stats, d_out);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}

int sddp_model_step(sddp_handle* h, const double* x, const double* u, const double* p, int k, double* x_next) {
    if (!h || !x || !u || !p || !x_next) return SDDP_ERR_ARG;
    if (k < 0 || k >= h->N) return fail(h, SDDP_ERR_ARG, "sddp_model_step: k must be a stage node, 0 <= k < N");
    const size_t B = size_t(h->B), nx = h->d.nx, nu = h->d.nu, np = h->d.np, D = sizeof(double);
    const size_t words = B * (2 * nx + nu + np);
    if (!h->step_buf) HIP_TRY(h, hipMalloc((void**)&h->step_buf, words * D));
    if (!h->step_pin && words * D <= (size_t(256) << 10) && hipHostMalloc((void**)&h->step_pin, words * D, hipHostMallocDefault) != hipSuccess) {
        h->step_pin = nullptr;
        (void)hipGetLastError();
    }
    double *dx = h->step_buf, *du = dx + B * nx, *dp = du + B * nu, *dxn = dp + B * np;
    int rc = SDDP_OK;
    if (h->step_pin) {   // x | u | p packed in pinned memory: one enqueued upload, one enqueued download, one wait
        double* hp = h->step_pin;
        std::memcpy(hp, x, B * nx * D);
        std::memcpy(hp + B * nx, u, B * nu * D);
        std::memcpy(hp + B * (nx + nu), p, B * np * D);
        HIP_TRY(h, hipMemcpyAsync(dx, hp, B * (nx + nu + np) * D, hipMemcpyHostToDevice, h->stream));
        rc = h->ops->launch_model_step(h, k, dx, du, dp, dxn);
        if (rc != SDDP_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(hp + B * (nx + nu + np), dxn, B * nx * D, hipMemcpyDeviceToHost, h->stream));
        rc = sddp_synchronize(h);
        if (rc == SDDP_OK) std::memcpy(x_next, hp + B * (nx + nu + np), B * nx * D);
        return rc;
    }
    HIP_TRY(h, hipMemcpyAsync(dx, x, B * nx * D, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(du, u, B * nu * D, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dp, p, B * np * D, hipMemcpyHostToDevice, h->stream));
    rc = h->ops->launch_model_step(h, k, dx, du, dp, dxn);
    if (rc != SDDP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(x_next, dxn, B * nx * D, hipMemcpyDeviceToHost, h->stream));
    return sddp_synchronize(h);
}

int sddp_is_converged(sddp_handle* h, int* flags) {
    if (!h || !flags) return SDDP_ERR_ARG;
    host_buf buf(size_t(h->B) * sizeof(sddp_stats));
    if (!buf.p) return fail(h, SDDP_ERR_NOMEM, "out of host memory");
    sddp_stats* st = buf.as<sddp_stats>();
    HIP_TRY(h, hipMemcpyAsync(st, h->stats, size_t(h->B) * sizeof(sddp_stats), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->B; ++b) flags[b] = st[b].converged;
    return SDDP_OK;
}

int sddp_device_ptr(sddp_handle* h, int which, void** ptr, long long* bytes) {
    if (!h || !ptr) return SDDP_ERR_ARG;
    long long n = 0;
    switch (which) {
        case 0: *ptr = h->xs; n = (long long)(h->n_x() * sizeof(double)); break;
        case 1: *ptr = h->us; n = (long long)(h->n_u() * sizeof(double)); break;
        case 2: *ptr = h->stats; n = (long long)(size_t(h->B) * sizeof(sddp_stats)); break;
        case 3:   // per slot: row b is instance b's only after a launch without a queue that started at instance 0
            if (!h->gains_by_instance)
                return fail(h, SDDP_ERR_ARG, "the gains are per queue slot: the last solve launch was queued or did not start at instance 0, "
                                             "so row b is not instance b's (solve with B <= slots, sddp_queue_info)");
            *ptr = h->gains; n = (long long)(size_t(h->wslots) * h->N * h->d.nu * (h->d.nx + 1) * sizeof(double)); break;
        case 4: *ptr = h->x0; n = (long long)(size_t(h->B) * h->d.nx * sizeof(double)); break;
        case 5: *ptr = h->P; n = (long long)(h->n_p() * sizeof(double)); break;
        case 6:   // the order the last queued launch handed its instances out in (absolute instance indices)
            if (!h->last_queued || h->opts.queue_order == 0) return fail(h, SDDP_ERR_ARG, "the last solve launch had no ordered queue");
            *ptr = h->order; n = (long long)(size_t(h->last_queued) * sizeof(int)); break;
        case 7:   // per slot of the last solve launch: (start, queue found empty) on the 100 MHz constant-rate clock
            if (h->last_grid < 1) return fail(h, SDDP_ERR_ARG, "no solve launch yet");
            *ptr = h->hist + ((h->B + 1) & ~1); n = (long long)(size_t(h->last_grid) * 2 * sizeof(unsigned long long)); break;
        default: return fail(h, SDDP_ERR_ARG, "unknown buffer id");
    }
    if (bytes) *bytes = n;
    return SDDP_OK;
}

int sddp_enable_timing(sddp_handle* h, int on) {
    if (!h) return SDDP_ERR_ARG;
    h->timing = on != 0;
    return SDDP_OK;
}
int sddp_last_kernel_ms(sddp_handle* h, double* ms) {
    if (!h || !ms) return SDDP_ERR_ARG;
    *ms = h->last_ms;
    return SDDP_OK;
}
int sddp_kernel_time_stats(sddp_handle* h, double* sum_ms, long long* count, int reset) {
    if (!h) return SDDP_ERR_ARG;
    if (sum_ms) *sum_ms = h->sum_ms;
    if (count) *count = h->n_ms;
    if (reset) { h->sum_ms = 0.0; h->n_ms = 0; }
    return SDDP_OK;
}

// ---- test building blocks ----------------------------------------------------------------------------------------------
int sddp_eval_knots(int model_id, const sddp_model_consts* consts, int N, int nk, const int* k, const double* x,
                    const double* u, const double* p, double* f_out, double* F_out, double* H_out, double* g_out, double* L_out) {
    sddp_model_consts cc;
    if (consts) cc = *consts; else if (sddp_default_consts_for(model_id, &cc) != SDDP_OK) return SDDP_ERR_MODEL;
    if (!model_ops(model_id)) return fail(nullptr, SDDP_ERR_MODEL, "unknown model_id");
    const bool bar = (cc.friction_barrier_weight > 0.0 || cc.bound_barrier_weight > 0.0) && !single_build(model_id);
    const bool xr = cc.n_extra != 0;
    const ModelOps* ops = model_ops(model_id, bar, false, xr);
    if (!ops) return fail(nullptr, SDDP_ERR_ARG, "this model has no such build (barrier / user rows)");
    const Dims d = ops->dims;
    if (nk < 1 || !k || !x || !u || !p) return fail(nullptr, SDDP_ERR_ARG, "bad argument");
    if (xr) { const char* msg = check_extra(cc, d.nx, d.nu); if (msg) return fail(nullptr, SDDP_ERR_ARG, msg); }
    DevConsts dc = make_dev_consts(cc);
    const int nz = d.nx + d.nu;
    const size_t D = sizeof(double);
    // every device buffer of the call in one table, freed on every return path
    enum { B_XR, B_BOX, B_K, B_X, B_U, B_P, B_REC, B_F, B_FF, B_H, B_G, B_L, B_N };
    void* buf[B_N] = {};
    const size_t bytes[B_N] = {xr ? kXrWords * D : 0, bar ? 128 * D : 0, nk * sizeof(int), size_t(nk) * d.nx * D, size_t(nk) * d.nu * D, size_t(nk) * d.np * D,
                               size_t(nk) * d.nrec * D, size_t(nk) * d.nx * D, size_t(nk) * d.nx * nz * D, size_t(nk) * nz * nz * D,
                               size_t(nk) * nz * D, size_t(nk) * D};
    auto release = [&]() { for (void* b : buf) if (b) (void)hipFree(b); };
#define TRY0(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { release(); return fail(nullptr, SDDP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    for (int i = 0; i < B_N; ++i)
        if (bytes[i]) TRY0(hipMalloc(&buf[i], bytes[i]));
    if (bar) {
        double hb[128];
        for (int i = 0; i < 64; ++i) { hb[i] = cc.lower[i]; hb[64 + i] = cc.upper[i]; }
        TRY0(hipMemcpy(buf[B_BOX], hb, sizeof(hb), hipMemcpyHostToDevice));
        dc.box = static_cast<double*>(buf[B_BOX]);
    }
    if (xr) {
        double t[kXrWords];
        fill_extra_table(cc, t);
        TRY0(hipMemcpy(buf[B_XR], t, sizeof(t), hipMemcpyHostToDevice));
        dc.xr = static_cast<double*>(buf[B_XR]);
        dc.xr_n = cc.n_extra;
    }
    TRY0(hipMemcpy(buf[B_K], k, bytes[B_K], hipMemcpyHostToDevice));
    TRY0(hipMemcpy(buf[B_X], x, bytes[B_X], hipMemcpyHostToDevice));
    TRY0(hipMemcpy(buf[B_U], u, bytes[B_U], hipMemcpyHostToDevice));
    TRY0(hipMemcpy(buf[B_P], p, bytes[B_P], hipMemcpyHostToDevice));
    TRY0(hipMemset(buf[B_REC], 0, bytes[B_REC]));
    auto dd = [&](int i) { return static_cast<double*>(buf[i]); };
    ops->launch_eval_knots(dc, N, nk, static_cast<const int*>(buf[B_K]), dd(B_X), dd(B_U), dd(B_P), dd(B_REC), dd(B_F), dd(B_FF), dd(B_H),
                           dd(B_G), dd(B_L));
    TRY0(hipGetLastError());
    TRY0(hipDeviceSynchronize());
    if (f_out) TRY0(hipMemcpy(f_out, buf[B_F], bytes[B_F], hipMemcpyDeviceToHost));
    if (F_out) TRY0(hipMemcpy(F_out, buf[B_FF], bytes[B_FF], hipMemcpyDeviceToHost));
    if (H_out) TRY0(hipMemcpy(H_out, buf[B_H], bytes[B_H], hipMemcpyDeviceToHost));
    if (g_out) TRY0(hipMemcpy(g_out, buf[B_G], bytes[B_G], hipMemcpyDeviceToHost));
    if (L_out) TRY0(hipMemcpy(L_out, buf[B_L], bytes[B_L], hipMemcpyDeviceToHost));
#undef TRY0
    release();
    return SDDP_OK;
}

int sddp_backward(sddp_handle* h, const double* params, double mu, double* gains_out, double* scal_out) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!params) return fail(h, SDDP_ERR_ARG, "params is NULL");
    if (h->B > h->wslots) return fail(h, SDDP_ERR_ARG, "phase-level entry points need batch <= resident slots (sddp_queue_info)");
    HIP_TRY(h, hipMemcpyAsync(h->P, params, h->n_p() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    SolveArgs a = make_args(h, h->P);
    a.mu = mu;
    rc = h->ops->launch_backward(h, a);
    if (rc != SDDP_OK) return rc;
    h->gains_by_instance = true;
    if (gains_out) HIP_TRY(h, hipMemcpyAsync(gains_out, h->gains, h->n_g() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (scal_out) {
        host_buf buf(size_t(h->B) * kScal * sizeof(double));
        if (!buf.p) return fail(h, SDDP_ERR_NOMEM, "out of host memory");
        const double* sc = buf.as<double>();
        HIP_TRY(h, hipMemcpy(buf.p, h->scal, size_t(h->B) * kScal * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < h->B; ++b)
            for (int i = 0; i < 8; ++i) scal_out[size_t(b) * 8 + i] = sc[size_t(b) * kScal + i];
    }
    return SDDP_OK;
}

int sddp_forward(sddp_handle* h, const double* params, double alpha, double* x_out, double* u_out, double* cost_out) {
    int rc = check_ready(h);
    if (rc != SDDP_OK) return rc;
    if (!params) return fail(h, SDDP_ERR_ARG, "params is NULL");
    if (h->B > h->wslots) return fail(h, SDDP_ERR_ARG, "phase-level entry points need batch <= resident slots (sddp_queue_info)");
    HIP_TRY(h, hipMemcpyAsync(h->P, params, h->n_p() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    SolveArgs a = make_args(h, h->P);
    a.alpha = alpha;
    rc = h->ops->launch_forward(h, a);
    if (rc != SDDP_OK) return rc;
    if (x_out) HIP_TRY(h, hipMemcpyAsync(x_out, h->xn, h->n_x() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (u_out) HIP_TRY(h, hipMemcpyAsync(u_out, h->un, h->n_u() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (cost_out) {
        host_buf buf(size_t(h->B) * kScal * sizeof(double));
        if (!buf.p) return fail(h, SDDP_ERR_NOMEM, "out of host memory");
        const double* sc = buf.as<double>();
        HIP_TRY(h, hipMemcpy(buf.p, h->scal, size_t(h->B) * kScal * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < h->B; ++b) cost_out[b] = sc[size_t(b) * kScal];
    }
    return SDDP_OK;
}

// diagnostic (not part of include/sddp.h): raw [B][16] scratch record; holds per-phase cycle sums in a -DSDDP_STAMPS build
int sddp_debug_read_scal(sddp_handle* h, double* out) {
    if (!h || !out) return SDDP_ERR_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->scal, size_t(h->B) * kScal * sizeof(double), hipMemcpyDeviceToHost));
    return SDDP_OK;
}

}  // extern "C"
