#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    // A[i][k] = 100*i + k ; B[k][j] = (k == kk) * (j + 1)  -> D[i][j] = (100 i + kk) (j+1): identifies (i, j) per output slot
    for (int kk = 0; kk < 4; ++kk) {
        // hypothesis: a-operand lane l holds A[l%16][l/16], b-operand lane l holds B[l/16][l%16]
        const double a = 100.0 * (lane % 16) + (lane / 16);
        const double b = ((lane / 16) == kk) ? double(lane % 16 + 1) : 0.0;
        double4_t c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(kk * 64 + lane) * 4 + r] = c[r];
    }
}
__global__ void timing(double* out, int reps) {
    const int lane = threadIdx.x;
    double a = lane * 0.001, b = lane * 0.002;
    double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);      // dependent chain
    }
    long long t1 = clock64();
    for (int r = 0; r < reps; ++r) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);      // two independent chains
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
    }
    long long t2 = clock64();
    if (lane == 0 && blockIdx.x == 0) { out[0] = double(t1 - t0) / reps; out[1] = double(t2 - t1) / (2.0 * reps); }
    out[2 + blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
int main() {
    double* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    probe<<<1, 64>>>(d);
    static double h[4 * 64 * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode kk = 0: value = 100 i (j+1) + 0 ; use kk=1 too: 100 i (j+1) + (j+1)
    for (int lane = 0; lane < 64; lane += 5)
        for (int r = 0; r < 4; ++r) {
            const double v0 = h[(0 * 64 + lane) * 4 + r], v1 = h[(1 * 64 + lane) * 4 + r];
            const int jp1 = int(v1 - v0 + 0.5), i = jp1 ? int(v0 / (100.0 * jp1) + 0.5) : -1;
            printf("lane %2d reg %d -> D[i=%d][j=%d]\n", lane, r, i, jp1 - 1);
        }
    for (int waves : {1, 2, 4, 8}) {
        timing<<<1, 64 * waves>>>(d, 2000);
        hipDeviceSynchronize();
        double t[2]; hipMemcpy(t, d, sizeof(t), hipMemcpyDeviceToHost);
        printf("waves/CU %d: cycles per MFMA f64 16x16x4: dependent %.1f, two chains %.1f (clock64 ticks)\n", waves, t[0], t[1]);
    }
    return 0;
}
