// Micro-benchmark (diagnostic, not part of the library): issue cost of the cross-lane / LDS broadcast primitives the
// Gauss-Jordan phases lean on, one wavefront, shader cycles via clock64 (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ __launch_bounds__(1024) void bench(double* out, long long* cyc, int reps) {
    __shared__ double lds_all[16 * 1024];
    double* lds = lds_all + 1024 * (threadIdx.x / 64);      // one private region per wave
    const int lane = threadIdx.x % 64;
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (lane + i);
    for (int i = lane; i < 1024; i += 64) lds[i] = 1.0 + 1e-4 * i;
    __syncthreads();
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));   // opaque: nothing is hoisted out of the repetition
        if (MODE == 0) {          // 64 readlane_d (128 v_readlane_b32) + 64 fma, independent chains of 8
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(readlane_d(a[i], p + 8 * (r & 1)), a[p], acc[i]);
        } else if (MODE == 1) {   // same data movement through LDS broadcast reads: 64 ds_read_b64 (uniform address) + 64 fma
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(lds[(r & 1) * 64 + p * 8 + i], a[p], acc[i]);
        } else if (MODE == 2) {   // 64 fma only, 8 independent chains
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(a[(i + p) & 7], a[p], acc[i]);
        } else if (MODE == 3) {   // 64 fma, ONE dependent chain
#pragma unroll
            for (int p = 0; p < 64; ++p) acc[0] = fma(acc[0], a[p & 7], a[(p + 1) & 7]);
        } else if (MODE == 4) {   // dependent: readlane -> rcp -> mul -> fma -> readlane (the pivot chain), 8 links
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double pv = readlane_d(acc[0] + a[p], p);
                const double t = a[1] * __builtin_amdgcn_rcp(pv);
                acc[0] = fma(-pv, t, acc[0]);
            }
        } else if (MODE == 6) {   // 32 ds_read_b128, distinct 16-byte slots per lane (conflict-free), + 64 fma
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int off = 2 * lane;
                    asm volatile("" : "+v"(off));                    // opaque: the eight rows are re-read, not CSE'd
                    const double2 v = *reinterpret_cast<const double2*>(lds + ((p * 8 + i) & 7) * 128 + off);
                    acc[i] = fma(v.x, a[p], acc[i]);
                    acc[i] = fma(v.y, a[p + 4], acc[i]);
                }
        } else if (MODE == 7) {   // 64 ds_read_b64, distinct slots per lane (conflict-free), + 64 fma
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int off = lane;
                    asm volatile("" : "+v"(off));
                    acc[i] = fma(lds[((p * 8 + i) & 15) * 64 + off], a[p], acc[i]);
                }
        } else if (MODE == 8) {   // 64 ds_read_b64 broadcast forced as b64 (stride 3 doubles: no pairing), + 64 fma
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fma(lds[(r & 1) * 256 + (p * 8 + i) * 3], a[p], acc[i]);
        } else if (MODE == 9) {   // 6x6 solve, every lane factors the (uniform) matrix itself: L D L^T + two substitutions, no lane traffic
            double m[6][6], rd[6], x[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                x[i] = a[i] + acc[i];
#pragma unroll
                for (int j = 0; j <= i; ++j) m[i][j] = lds[(r & 1) * 64 + i * 6 + j] + (i == j ? 50.0 : 0.0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double cc[6];
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    double v = m[i][j];
#pragma unroll
                    for (int k = 0; k < j; ++k) v = fma(-cc[k], m[j][k], v);
                    cc[j] = v;
                }
                double d = m[i][i];
#pragma unroll
                for (int j = 0; j < i; ++j) { m[i][j] = cc[j] * rd[j]; d = fma(-cc[j], m[i][j], d); }
                double q = __builtin_amdgcn_rcp(d);
                q = fma(fma(-d, q, 1.0), q, q);
                rd[i] = fma(fma(-d, q, 1.0), q, q);
            }
#pragma unroll
            for (int i = 1; i < 6; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) x[i] = fma(-m[i][k], x[k], x[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] *= rd[i];
#pragma unroll
            for (int i = 4; i >= 0; --i)
#pragma unroll
                for (int k = i + 1; k < 6; ++k) x[i] = fma(-m[k][i], x[k], x[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = x[i];
        } else if (MODE == 10) {  // the same solve as a shared elimination: lane j owns column j, pivot column broadcast by v_readlane
            double x[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = lds[(r & 1) * 64 + i * 6 + (lane < 6 ? lane : 0)] + ((i == lane) ? 50.0 : 0.0) + (lane >= 6 ? a[i] + acc[i] : 0.0);
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                double pv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) pv[i] = readlane_d(x[i], p);
                double q = __builtin_amdgcn_rcp(pv[p]);
                q = fma(fma(-pv[p], q, 1.0), q, q);
                q = fma(fma(-pv[p], q, 1.0), q, q);
                const double t = x[p] * q;
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = (i == p) ? t : fma(-pv[i], t, x[i]);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = x[i];
        } else if (MODE == 11) {  // 64 dependent ops alternating v_mul_f64 / v_fma_f64 with a negated operand
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const double t = acc[0] * a[p & 7];
                acc[0] = fma(-t, a[(p + 3) & 7], a[(p + 1) & 7]);
            }
        } else if (MODE == 12) {  // 8 dependent v_rcp_f64 + Newton (5 instructions each)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double d = acc[0] + a[p];
                double q = __builtin_amdgcn_rcp(d);
                q = fma(fma(-d, q, 1.0), q, q);
                acc[0] = fma(fma(-d, q, 1.0), q, q);
            }
        } else if (MODE == 5) {   // LDS round trip: write one value per lane, fence, broadcast-read 8 values
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                lds[lane] = acc[0] + a[p];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                acc[0] = fma(lds[p], a[1], acc[0]);
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[threadIdx.x] = s;
    if (lane == 0) { cyc[2 * (threadIdx.x / 64)] = t0; cyc[2 * (threadIdx.x / 64) + 1] = t1; }
}

template <int M> void run(const char* name, double* out, long long* cyc) {
    const int reps = 1000;
    printf("%-50s", name);
    for (int waves : {1, 4, 8, 16}) {
        for (int it = 0; it < 2; ++it) {
            bench<M><<<1, 64 * waves>>>(out, cyc, reps);
            (void)hipDeviceSynchronize();
        }
        long long c[32]; (void)hipMemcpy(c, cyc, 16 * waves, hipMemcpyDeviceToHost);
        long long lo = c[0], hi = c[1];
        for (int w = 0; w < waves; ++w) { lo = c[2 * w] < lo ? c[2 * w] : lo; hi = c[2 * w + 1] > hi ? c[2 * w + 1] : hi; }
        printf(" %8.1f", double(hi - lo) / reps);
    }
    printf("\n");
}

int main() {
    double* out; long long* cyc;
    (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&cyc, 512);
    printf("cycles per repetition (first start to last end over the waves), 1 / 4 / 8 / 16 waves of one workgroup (one CU) running the same loop\n");
    run<2>("64 fma, 8 chains", out, cyc);
    run<3>("64 fma, 1 chain", out, cyc);
    run<0>("64 readlane_d (128 v_readlane_b32) + 64 fma", out, cyc);
    run<1>("32 ds_read_b128 broadcast + 64 fma", out, cyc);
    run<8>("64 ds_read_b64 broadcast + 64 fma", out, cyc);
    run<6>("32 ds_read_b128 per-lane slots + 64 fma", out, cyc);
    run<7>("64 ds_read_b64 per-lane slots + 64 fma", out, cyc);
    run<4>("8 x (readlane_d -> rcp -> mul -> fma)", out, cyc);
    run<5>("8 x (ds_write, fence, ds_read broadcast, fma)", out, cyc);
    run<11>("64 dependent mul/fma alternating", out, cyc);
    run<12>("8 dependent (add, rcp, 2 Newton steps)", out, cyc);
    run<9>("6x6 solve: per-lane LDL^T, no lane traffic", out, cyc);
    run<10>("6x6 solve: shared elimination, readlane pivots", out, cyc);
    return 0;
}
