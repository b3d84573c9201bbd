// sddp_handle.hpp -- the handle behind include/sddp.h and the per-model operation table.
//
// The library is built from one translation unit per model build (sddp_inst.hip, compiled once per entry of
// srbd_horizon_amd/_lib.py INSTANCES, in parallel) plus the model-independent host code (sddp_api.hip).  A model build reaches
// the API through a ModelOps table of plain function pointers: nothing templated crosses a translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "sddp.h"
#include "sddp_kernels.hpp"

struct sddp_handle;

namespace sddp {

struct Dims {
    int nx, nu, np, nrec;
};

// what a model build provides (sddp_launch.hpp: make_ops<M>)
struct ModelOps {
    Dims dims;
    bool uses_mw;          // 4 wavefronts per instance (sddp_kernels_mw.hpp); else one
    bool w2_build;         // a half-register-file build exists (two instances per SIMD / two workgroups per CU)
    const char* name;      // kernel-facing model name (bench / profiles)
    int (*max_slots)(sddp_handle*, int*);
    int (*launch_solve)(sddp_handle*, SolveArgs, int, int);
    int (*launch_backward)(sddp_handle*, const SolveArgs&);
    int (*launch_forward)(sddp_handle*, const SolveArgs&);
    int (*launch_model_step)(sddp_handle*, int, const double*, const double*, const double*, double*);
    void (*launch_eval_knots)(const DevConsts&, int, int, const int*, const double*, const double*, const double*, double*, double*,
                              double*, double*, double*, double*);
};

}  // namespace sddp

struct sddp_handle {
    int model_id = 0, N = 0, B = 0;
    sddp::Dims d{};
    const sddp::ModelOps* ops = nullptr;
    sddp_options opts{};
    sddp_model_consts consts{};
    sddp::DevConsts dc{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device buffers
    double *x0 = nullptr, *P = nullptr, *xs = nullptr, *us = nullptr, *xn = nullptr, *un = nullptr, *xc = nullptr, *uc = nullptr, *dft = nullptr,
           *gains = nullptr, *rec = nullptr, *scal = nullptr;
    sddp_stats* stats = nullptr;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;     // pairs (start, stop), one pair per launch since the last synchronize
    size_t pending = 0;            // launches whose events have not been read yet
    double last_ms = 0.0, sum_ms = 0.0;
    long long n_ms = 0;
    std::string err;
    bool have_x0 = false, have_xws = false, have_uws = false, have_params = false;
    bool bar = false;               // friction-cone barrier build (consts.friction_barrier_weight > 0)
    bool so2 = false;               // full second-order build (opts.second_order == 2 at sddp_create)
    double* tick_in = nullptr;      // [B][np + nx] staging of sddp_advance
    double* step_buf = nullptr;     // [B][2 nx + nu + np] operands and result of sddp_model_step
    char* tick_pin = nullptr;       // two pinned images of tick_in (small batches)
    int tick_flip = 0, tick_unsynced = 0;
    double* step_pin = nullptr;     // pinned host image of step_buf (small batches)
    void* pinned = nullptr;         // small batches: pinned host staging of x | u | stats, so the three result copies are truly asynchronous
    size_t pinned_bytes = 0;
    // work queue: the solve launch runs on `slots` resident workgroups that pull instances from a queue
    int wslots = 0;                 // slots the work buffers (xn un xc uc dft gains rec) are allocated for = min(B, resident capacity)
    int cus = 0;
    int* qhead = nullptr;           // device queue head
    int* order = nullptr;           // [B] queue order of the next launch
    int* hist = nullptr;            // [B] iterations of each instance's previous solve (-1: none), then the slot clocks [wslots][2]
                                    // (uint64, SolveArgs::slot_clock)
    // cold-queue order (queue_order = 2): initial-cost keys of the launch, their sorted copy, the unsorted index list, sort scratch;
    // allocated together at sddp_create when the option asks for it, or on the first launch that needs them (all or nothing)
    double *qkey = nullptr, *qkey2 = nullptr;
    int* order_in = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    // class history (queue_order = 3): the caller's class label per instance, and per class the iterations / solves so far
    int* cls = nullptr;             // [B], -1: unlabelled
    int n_cls = 0;
    unsigned long long* cls_stat = nullptr;   // [n_cls][2]
    bool gains_by_instance = false; // the last solve launch ran instance b on slot b (no queue, first = 0): sddp_device_ptr(3)
    struct KInfo { const void* fn = nullptr; int slots = 0; };
    KInfo kinfo[2];                 // per kernel build: dynamic-LDS attribute set, resident workgroups on this device
    int last_grid = 0, last_queued = 0;
    int last_build = 0;             // waves_per_simd of the kernel build the last solve launch ran (sddp_kernel_info)
    const void* last_kernel = nullptr;   // ... and that kernel, its dynamic LDS bytes and its workgroups per CU (sddp_kernel_resources)
    int last_lds = 0, last_per_cu = 0;
    double* box_dev = nullptr;      // lower[64] | upper[64] of the bound barrier (barrier builds)
    double* xr_dev = nullptr;       // user rows: coefficients | weights | constants (DevConsts::xr, "_x" builds)
    double* first_dev = nullptr;    // [B][nu + nx + 3] packed first knots of sddp_solve_resident_first, and its pinned host image
    double* first_pin = nullptr;
    char* up_pin = nullptr;         // pinned ring for small host->device uploads of the setters (no wait per call)
    size_t up_off = 0;

    size_t n_x() const { return size_t(B) * (N + 1) * d.nx; }
    size_t n_u() const { return size_t(B) * N * d.nu; }
    size_t n_p() const { return size_t(B) * (N + 1) * d.np; }
    size_t n_g() const { return size_t(B) * N * d.nu * (d.nx + 1); }
};

namespace sddp {

// error of a call without a handle (sddp_create, sddp_eval_knots): one per thread, defined in sddp_api.hip
std::string& create_error();

inline int fail(sddp_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else create_error() = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return sddp::fail(h, SDDP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// the cold-queue buffers of a handle, all or nothing (a partial failure leaves every pointer null)
int alloc_cold_queue(sddp_handle* h);
// queue order 1 (longest previous solve first): counting sort of [first, first + count) by h->hist into h->order, on the stream
int launch_queue_order(sddp_handle* h, int first, int count);
// queue order 3: h->qkey (initial costs of the launch's instances, order h->order_in) -> class-history keys; and the update of the
// class statistics behind a solve launch
int launch_class_keys(sddp_handle* h, int count);
int launch_class_update(sddp_handle* h, int first, int count);

}  // namespace sddp
