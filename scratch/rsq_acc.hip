// stand-alone: relative error of v_rsq_f64 / v_rcp_f64 seeds and of the shortened Newton sequences k_fill3 uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    const double g1 = __builtin_fma(g, r, g);                 // one step
    const double d = __builtin_fma(-g1, g1, v);
    const double g15 = __builtin_fma(d, h, g1);               // + correction with the unrefined h
    o[4 * i] = y; o[4 * i + 1] = g1; o[4 * i + 2] = g15; o[4 * i + 3] = __builtin_amdgcn_rcp(v);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(4 * n);
    for (int i = 0; i < n; i++) x[i] = exp(((double)rand() / RAND_MAX) * 40 - 10);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; i++) {
        const long double s = sqrtl((long double)x[i]);
        e0 = fmax(e0, (double)fabsl(o[4 * i] * s - 1));
        e1 = fmax(e1, (double)fabsl(o[4 * i + 1] / s - 1));
        e2 = fmax(e2, (double)fabsl(o[4 * i + 2] / s - 1));
        e3 = fmax(e3, (double)fabsl((long double)o[4 * i + 3] * x[i] - 1));
    }
    printf("rsq seed %.3g, sqrt one step %.3g, one step + correction %.3g, rcp seed %.3g\n", e0, e1, e2, e3);
    return 0;
}
