// sddp_launch.hpp -- host-side launchers of one model build (templates on the device model), collected into a ModelOps table.
// Included by sddp_inst.hip only: one translation unit per model build.
#pragma once
#include <algorithm>

#include "sddp_handle.hpp"
#include "sddp_kernels.hpp"
#include "sddp_kernels_mw.hpp"
#include "sddp_models.hpp"
#include "sddp_sort.hpp"

namespace sddp {

// large models (> 48 KB of LDS per instance: srbd37, srbd61, lip30) run on 4 waves per instance (sddp_kernels_mw.hpp)
template <class M>
constexpr bool use_mw() {
#ifdef SDDP_MW_ALL
    return true;
#else
    return Lds<M>::BYTES > 48 * 1024;
#endif
}
template <class M>
constexpr size_t lds_bytes() {
    if constexpr (use_mw<M>()) return LdsMW<M>::BYTES; else return Lds<M>::BYTES;
}
// a half-register-file build pays where two workgroups fit a CU's 160 KB (4-wave kernel) / always (one-wave kernel)
template <class M>
constexpr bool has_w2() {
    if constexpr (use_mw<M>()) return 2 * LdsMW<M>::BYTES <= size_t(160) * 1024; else return true;
}

// only the kernel a model actually uses is instantiated
using KernelFn = void (*)(SolveArgs);
template <class M> KernelFn pick_solve(int waves_per_simd) {
    if constexpr (use_mw<M>()) {
        if constexpr (has_w2<M>()) return waves_per_simd >= 2 ? solve_kernel_mw_w2<M> : solve_kernel_mw<M>;
        else return solve_kernel_mw<M>;
    } else return waves_per_simd >= 2 ? solve_kernel_w2<M> : solve_kernel<M>;
}
template <class M> KernelFn pick_backward() { if constexpr (use_mw<M>()) return backward_kernel_mw<M>; else return backward_kernel<M>; }
template <class M> KernelFn pick_forward() { if constexpr (use_mw<M>()) return forward_kernel_mw<M>; else return forward_kernel<M>; }

// resident workgroups of `kern` on this device (the queue's slot count) and its dynamic-LDS attribute, once per handle and build
template <class M>
int kernel_slots(sddp_handle* h, KernelFn kern, int wps, int* slots) {
    constexpr bool MW = use_mw<M>();
    constexpr size_t lds = lds_bytes<M>();
    constexpr int threads = MW ? kThreadsMW : kWave;
    for (auto& k : h->kinfo)
        if (k.fn == reinterpret_cast<const void*>(kern)) { *slots = k.slots; return SDDP_OK; }
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
    if (!MW) per_cu = std::min(per_cu, 4 * (wps >= 2 ? 2 : 1));   // the two builds: 1 or 2 wavefronts per SIMD
    per_cu = std::max(1, std::min(per_cu, 32));
    auto& k = h->kinfo[h->kinfo[0].fn ? 1 : 0];
    k.fn = reinterpret_cast<const void*>(kern);
    k.slots = per_cu * std::max(1, h->cus);
    *slots = k.slots;
    return SDDP_OK;
}

// one launch over the instances [first, first + count): grid = resident slots, at most `count` and at most the slots the work
// buffers exist for; more instances than slots -> work queue, in longest-previous-solve-first order when opts.queue_order is set
template <class M>
int launch_solve(sddp_handle* h, SolveArgs a, int first, int count) {
    constexpr bool MW = use_mw<M>();
    int wps = h->opts.waves_per_simd >= 2 && has_w2<M>() ? 2 : 1;
    KernelFn kern = pick_solve<M>(wps);
    constexpr size_t lds = lds_bytes<M>();
    constexpr int threads = MW ? kThreadsMW : kWave;
    int slots = 0;
    int rc = kernel_slots<M>(h, kern, wps, &slots);
    if (rc != SDDP_OK) return rc;
    if constexpr (MW) {   // a half-register-file build that the device still runs one per CU (barrier builds) has nothing to offer
        if (wps >= 2) {
            KernelFn k1 = pick_solve<M>(1);
            int s1 = 0;
            rc = kernel_slots<M>(h, k1, 1, &s1);
            if (rc != SDDP_OK) return rc;
            if (s1 >= slots) { kern = k1; slots = s1; wps = 1; }
        }
    }
    int grid = std::min(count, std::min(slots, h->wslots));
    if (h->opts.max_slots > 0) grid = std::min(grid, h->opts.max_slots);
    a.first = first; a.count = count;
    // the timed interval of a launch covers its queue-ordering pre-pass (key kernel + sort, or the counting sort) as well
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        while (h->ev.size() < 2 * (h->pending + 1)) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreate(&e));
            try { h->ev.push_back(e); } catch (...) { (void)hipEventDestroy(e); return fail(h, SDDP_ERR_NOMEM, "out of host memory"); }
        }
        e0 = h->ev[2 * h->pending];
        e1 = h->ev[2 * h->pending + 1];
        HIP_TRY(h, hipEventRecord(e0, h->stream));
    }
    if (count > grid) {
        HIP_TRY(h, hipMemsetAsync(h->qhead, 0, sizeof(int), h->stream));
        a.qhead = h->qhead;
        if (h->opts.queue_order == 1) {            // longest previous solve first
            rc = launch_queue_order(h, first, count);
            if (rc != SDDP_OK) return rc;
            a.order = h->order;
        } else if (h->opts.queue_order >= 2) {     // largest initial cost first: keys by a pre-pass over the launch's instances
            rc = alloc_cold_queue(h);
            if (rc != SDDP_OK) return rc;
            hipLaunchKernelGGL(queue_cost_key_kernel<M>, dim3(count), dim3(kWave), 0, h->stream, a.c, a.N, first, count, a.x0, a.P, a.xs,
                               a.us, h->qkey, h->order_in);
            HIP_TRY(h, hipGetLastError());
            if (h->opts.queue_order == 3 && h->cls) {   // ... longest class history first, the initial cost breaking ties
                rc = launch_class_keys(h, count);
                if (rc != SDDP_OK) return rc;
            }
            HIP_TRY(h, sort_pairs_desc(h->sort_tmp, h->sort_tmp_bytes, h->qkey, h->qkey2, h->order_in, h->order, count, h->stream));
            a.order = h->order;
        }
    }
    h->last_grid = grid; h->last_queued = count > grid ? count : 0;
    h->last_build = wps;
    h->last_kernel = reinterpret_cast<const void*>(kern);
    h->last_lds = int(lds);
    h->last_per_cu = slots / std::max(1, h->cus);
    h->gains_by_instance = (count <= grid && first == 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    if (h->cls) {                                       // labelled instances: their iteration counts feed the class statistics
        rc = launch_class_update(h, first, count);
        if (rc != SDDP_OK) return rc;
    }
    if (h->timing) {
        HIP_TRY(h, hipEventRecord(e1, h->stream));
        ++h->pending;
    }
    return SDDP_OK;
}
// resident capacity over the builds a handle may switch between (sddp_set_options): sizes the work buffers
template <class M>
int max_slots(sddp_handle* h, int* slots) {
    if constexpr (!has_w2<M>()) return kernel_slots<M>(h, pick_solve<M>(1), 1, slots);
    else {
        int s1 = 0, s2 = 0;
        int rc = kernel_slots<M>(h, pick_solve<M>(1), 1, &s1);
        if (rc == SDDP_OK) rc = kernel_slots<M>(h, pick_solve<M>(2), 2, &s2);
        *slots = std::max(s1, s2);
        return rc;
    }
}
template <class M>
int launch_backward(sddp_handle* h, const SolveArgs& a) {
    constexpr bool MW = use_mw<M>();
    KernelFn kern = pick_backward<M>();
    constexpr size_t lds = lds_bytes<M>();
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(h->B), dim3(MW ? kThreadsMW : kWave), lds, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}
template <class M>
int launch_forward(sddp_handle* h, const SolveArgs& a) {
    constexpr bool MW = use_mw<M>();
    KernelFn kern = pick_forward<M>();
    constexpr size_t lds = lds_bytes<M>();
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(h->B), dim3(MW ? kThreadsMW : kWave), lds, h->stream, a);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}

template <class M>
int launch_model_step(sddp_handle* h, int k, const double* dx, const double* du, const double* dp, double* dxn) {
    hipLaunchKernelGGL(model_step_kernel<M>, dim3((h->B + kWave - 1) / kWave), dim3(kWave), 0, h->stream, h->dc, h->B, k, dx, du, dp, dxn);
    HIP_TRY(h, hipGetLastError());
    return SDDP_OK;
}

template <class M>
void launch_eval_knots(const DevConsts& dc, int N, int nk, const int* dk, const double* dx, const double* du, const double* dp, double* drec,
                       double* df, double* dF, double* dH, double* dg, double* dL) {
    hipLaunchKernelGGL(eval_knots_kernel<M>, dim3(nk), dim3(kWave), 0, 0, dc, N, nk, dk, dx, du, dp, drec, df, dF, dH, dg, dL);
}

template <class M>
ModelOps make_ops(const char* name) {
    ModelOps o;
    o.dims = {M::NX, M::NU, M::NP, M::NREC};
    o.uses_mw = use_mw<M>();
    o.w2_build = has_w2<M>();
    o.name = name;
    o.max_slots = max_slots<M>;
    o.launch_solve = launch_solve<M>;
    o.launch_backward = launch_backward<M>;
    o.launch_forward = launch_forward<M>;
    o.launch_model_step = launch_model_step<M>;
    o.launch_eval_knots = launch_eval_knots<M>;
    return o;
}

}  // namespace sddp
