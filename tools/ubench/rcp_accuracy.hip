// How accurate is v_rcp_f64, raw and after one / two Newton steps?  (fast_rcp in csrc/sddp_models.hpp uses two.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/rcp_accuracy.hip -o build/ub/rcp_accuracy && build/ub/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = __builtin_amdgcn_rcp(x[i]);
    r0[i] = r;
    r = fma(fma(-x[i], r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-x[i], r, 1.0), r, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + double(s >> 11) / 9007199254740992.0;        // mantissa in [1, 2)
        x[i] = std::ldexp(m, int((s & 255) % 80) - 40) * ((s >> 8) & 1 ? 1 : -1);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error of 1/x over %d values: raw v_rcp_f64 %.3e (%.1f ulp), + 1 Newton step %.3e (%.2f ulp), + 2 steps %.3e (%.2f ulp)\n", n, e0,
           e0 / 2.22e-16, e1, e1 / 2.22e-16, e2, e2 / 2.22e-16);
    return 0;
}
