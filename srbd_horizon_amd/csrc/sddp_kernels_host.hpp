// sddp_kernels_host.hpp -- the model-independent kernels of the library (queue order by history, receding-horizon shift,
// first-knot packing).  Included by sddp_api.hip only: they are not templates, so they must live in ONE translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include "sddp.h"

namespace sddp {

// Queue order for the next launch: instances [first, first + count) sorted by the iteration count of their previous solve,
// longest first, never-solved ones before all others (longest-processing-time-first list scheduling: a launch ends with its
// slowest instance, so the slow ones must start first; a fleet's robots recur tick after tick and the previous count is the
// predictor at hand).  Counting sort in one workgroup; ties in arbitrary order (results do not depend on the order).
constexpr int kOrderBins = 258;   // key = min(hist, 256), -1 -> 257
__global__ __launch_bounds__(1024) void queue_order_kernel(int first, int count, const int* __restrict__ hist, int* __restrict__ order) {
    __shared__ int bins[kOrderBins];
    const int tid = threadIdx.x;
    for (int i = tid; i < kOrderBins; i += 1024) bins[i] = 0;
    __syncthreads();
    for (int i = tid; i < count; i += 1024) {
        const int h = hist[first + i];
        atomicAdd(&bins[h < 0 ? kOrderBins - 1 : (h > 256 ? 256 : h)], 1);
    }
    __syncthreads();
    if (tid == 0) {   // exclusive prefix, largest key first
        int run = 0;
        for (int k = kOrderBins - 1; k >= 0; --k) { const int n = bins[k]; bins[k] = run; run += n; }
    }
    __syncthreads();
    for (int i = tid; i < count; i += 1024) {
        const int h = hist[first + i];
        const int pos = atomicAdd(&bins[h < 0 ? kOrderBins - 1 : (h > 256 ? 256 : h)], 1);
        order[pos] = first + i;
    }
}

// Queue order 3 (class history): the key of an instance is the mean iteration count that earlier solves of its CLASS took on this
// handle (sddp_set_instance_classes: the caller's label of what kind of problem an instance is -- gait phase, command, ...), the
// initial cost (the key of queue order 2, already in `key`) breaking ties inside a class; classes never solved before sort first.
// cls_stat: [n_cls][2] = sum of iterations, solves.  The sum / count are read here and updated by class_update_kernel after the
// solve launch, both on the handle's stream.
__global__ __launch_bounds__(256) void class_key_kernel(int count, const int* __restrict__ idx, const int* __restrict__ cls, int n_cls,
                                                        const unsigned long long* __restrict__ cls_stat, double* __restrict__ key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int c = cls[idx[i]];
    double mean = 1e6;                                         // unlabelled instance / class without history: first
    if (c >= 0 && c < n_cls && cls_stat[2 * c + 1] > 0) mean = double(cls_stat[2 * c]) / double(cls_stat[2 * c + 1]);
    const double j0 = key[i];                                  // initial cost; non-finite: stays first
    key[i] = (j0 == j0 && j0 < 1e300) ? mean + 1e-3 * j0 / (fabs(j0) + 1e9) : j0;      // tie-break in (-1e-3, 1e-3), monotone in j0
}
__global__ __launch_bounds__(256) void class_update_kernel(int first, int count, const int* __restrict__ cls, int n_cls,
                                                           const sddp_stats* __restrict__ st, unsigned long long* __restrict__ cls_stat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int c = cls[first + i];
    if (c < 0 || c >= n_cls) return;
    atomicAdd(&cls_stat[2 * c], (unsigned long long)st[first + i].iters);
    atomicAdd(&cls_stat[2 * c + 1], 1ull);
}

// what an MPC tick applies: the first input u_0 and the state the plan expects next, x_1, of every instance, packed
// [B][nu + nx] (+ cost, iterations, status as three more doubles) for one small copy to the host instead of the whole trajectories
__global__ __launch_bounds__(256) void first_knot_kernel(int N, int B, int nx, int nu, const double* __restrict__ xs,
                                                         const double* __restrict__ us, const sddp_stats* __restrict__ st,
                                                         double* __restrict__ out) {
    const int w = nu + nx + 3;
    for (size_t e = size_t(blockIdx.x) * 256 + threadIdx.x; e < size_t(B) * w; e += size_t(gridDim.x) * 256) {
        const int b = int(e / w), j = int(e % w);
        double v;
        if (j < nu) v = us[size_t(b) * N * nu + j];
        else if (j < nu + nx) v = xs[(size_t(b) * (N + 1) + 1) * nx + (j - nu)];
        else v = j == nu + nx ? st[b].cost : (j == nu + nx + 1 ? double(st[b].iters) : double(st[b].status));
        out[e] = v;
    }
}

// The record an instance-sharded fleet exchanges after a launch: per instance
//   mode 0: x [N+1][nx] | u [N][nu] | cost | iterations            ((N+1) nx + N nu + 2 doubles: the whole plan)
//   mode 1: u_0 [nu] | x_1 [nx] | cost | iterations                (nu + nx + 2 doubles: what a closed loop applies next)
// written [count][words] contiguous into the collective's send buffer: one kernel, whole-wave contiguous stores (the trajectories
// of consecutive instances are consecutive in xs / us, so the loads are contiguous runs too).
__global__ __launch_bounds__(256) void pack_records_kernel(int N, int nx, int nu, int first, int count, int mode,
                                                           const double* __restrict__ xs, const double* __restrict__ us,
                                                           const sddp_stats* __restrict__ st, double* __restrict__ out) {
    const int nxw = mode == 0 ? (N + 1) * nx : nx, nuw = mode == 0 ? N * nu : nu;
    const int w = nxw + nuw + 2;
    for (size_t e = size_t(blockIdx.x) * 256 + threadIdx.x; e < size_t(count) * w; e += size_t(gridDim.x) * 256) {
        const int i = int(e / w), j = int(e % w), b = first + i;
        double v;
        if (mode == 0) {
            if (j < nxw) v = xs[size_t(b) * nxw + j];
            else if (j < nxw + nuw) v = us[size_t(b) * nuw + (j - nxw)];
            else v = j == nxw + nuw ? st[b].cost : double(st[b].iters);
        } else {
            if (j < nu) v = us[size_t(b) * N * nu + j];
            else if (j < nu + nx) v = xs[(size_t(b) * (N + 1) + 1) * nx + (j - nu)];
            else v = j == nu + nx ? st[b].cost : double(st[b].iters);
        }
        out[e] = v;
    }
}

// diagnostic (sddp_debug_poison_lds): a workgroup that owns a CU's whole LDS and fills it with NaNs.  What a kernel finds in LDS
// is whatever the previous one left there; after this one a read of a word the kernel never wrote cannot pass a parity test.
__global__ __launch_bounds__(256) void poison_lds_kernel(int words, int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    for (int e = threadIdx.x; e < words; e += 256) lds_all[e] = __builtin_nan("");
    __syncthreads();
    if (lds_all[threadIdx.x] == 1.0) *sink = 1;          // (never true: keeps the stores)
}

constexpr int kAdvanceWords = 4096;   // longest array advance_kernel shifts in one workgroup: (N+1) * max(nx, np) words

// receding-horizon tick on the device: shift parameters and warm start by one knot (one workgroup per instance; every element
// is read before the barrier and written after it, so the in-place shift is safe)
__global__ __launch_bounds__(256) void advance_kernel(int N, int nx, int nu, int np, double* __restrict__ P, double* __restrict__ xs,
                                                      double* __restrict__ us, double* __restrict__ x0, const double* __restrict__ p_last,
                                                      const double* __restrict__ x0_new) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double* Pb = P + size_t(b) * (N + 1) * np;
    double* xb = xs + size_t(b) * (N + 1) * nx;
    double* ub = us + size_t(b) * N * nu;
    constexpr int R = kAdvanceWords / 256;                 // elements per thread and array
    double rp[R], rx[R], ru[R];
    const int np_all = (N + 1) * np, nx_all = (N + 1) * nx, nu_all = N * nu;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int e = tid + t * 256;
        rp[t] = e < np_all ? (e + np < np_all ? Pb[e + np] : p_last[size_t(b) * np + (e - N * np)]) : 0.0;
        rx[t] = e < nx_all ? xb[e + nx < nx_all ? e + nx : e] : 0.0;
        ru[t] = e < nu_all ? ub[e + nu < nu_all ? e + nu : e] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int e = tid + t * 256;
        if (e < np_all) Pb[e] = rp[t];
        if (e < nx_all) xb[e] = rx[t];
        if (e < nu_all) ub[e] = ru[t];
    }
    if (tid < nx) x0[size_t(b) * nx + tid] = x0_new[size_t(b) * nx + tid];
}

}  // namespace sddp
