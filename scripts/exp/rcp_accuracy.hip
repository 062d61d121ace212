// Measures the relative error of v_rcp_f64 followed by 0, 1 or 2 Newton steps against IEEE division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o0, double* o1, double* o2, int n)
    {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    o0[i] = r;
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    o1[i] = r;
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    o2[i] = r;
    }
int main()
    {
    const int n = 1 << 22;
    std::vector<double> h(n);
    for (int i = 0; i < n; i++) h[i] = 0.3 + 9.0 * (double)i / n + 1e-9 * (i % 977);
    double *x, *o0, *o1, *o2;
    hipMalloc(&x, n * 8); hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8); hipMalloc(&o2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, o0, o1, o2, n);
    std::vector<double> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), o0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), o1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++)
        {
        double ex = 1.0 / h[i];
        e0 = fmax(e0, fabs(r0[i] - ex) / ex);
        e1 = fmax(e1, fabs(r1[i] - ex) / ex);
        e2 = fmax(e2, fabs(r2[i] - ex) / ex);
        }
    printf("max rel err: rcp %.3e  +1 Newton %.3e  +2 Newton %.3e\n", e0, e1, e2);
    return 0;
    }
