#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
typedef int32_t i32; typedef int64_t i64;
#define F2_PT 8
struct Fill2Args { const i32 *col_ptr, *col_k; const double* col_val; int normed; double threshold; i64* labels; double* confs; };
struct PredAcc {
    i32 id[F2_PT];
    double acc[F2_PT];
    int used;
    bool over;
    double x2;
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int s = 0; s < F2_PT; s++) { id[s] = -1; acc[s] = 0.0; }
        used = 0; over = false; x2 = 0.0;
    }
    __device__ __forceinline__ void add(const Fill2Args &a, i32 k, double val)
    {
        x2 += val * val;
        const i32 lo = a.col_ptr[k], hi = a.col_ptr[k + 1];
        for (i32 q = lo; q < hi; q++) {
            const i32 cid = a.col_k[q];
            const double term = a.col_val[q] * val;
            bool found = false;
#pragma unroll
            for (int s = 0; s < F2_PT; s++)
                if (id[s] == cid) { acc[s] += term; found = true; }
            if (!found) {
                if (used == F2_PT) { over = true; }
                else {
#pragma unroll
                    for (int s = 0; s < F2_PT; s++)
                        if (s == used) { id[s] = cid; acc[s] = term; }
                    used++;
                }
            }
        }
    }
    __device__ __forceinline__ void finish(const Fill2Args &a, i64 row) const
    {
        const double xn = sqrt(x2);
        double bv = 0.0; i64 bi = -1; bool bnan = false;
#pragma unroll
        for (int s = 0; s < F2_PT; s++) {
            if (id[s] < 0) continue;
            double v = acc[s];
            if (a.normed) v /= xn;
            v = fabs(v);
            const bool vn = isnan(v);
            bool take;
            if (bi < 0) take = true;
            else if (bnan || vn) take = vn && (!bnan || id[s] < bi);
            else take = (v > bv) || (v == bv && id[s] < bi);
            if (take) { bv = v; bi = id[s]; bnan = vn; }
        }
        i64 to; double conf;
        if (bi < 0 || (!bnan && bv == 0.0)) { to = 0; conf = 0.0; }
        else { to = bi; conf = bv; }
        if (conf < a.threshold) { to = -1; conf = 0.0; }
        a.labels[row] = to;
        a.confs[row] = conf;
    }
};
__global__ void k(Fill2Args a, const i32* ks, const double* vals, int n){
  PredAcc p; if (threadIdx.x < 64) p.init();
  for (int e=0;e<n;e++){ double v = vals[e*64+threadIdx.x]; if (v != 0.0) p.add(a, ks[e*64+threadIdx.x], v); }
  p.finish(a, threadIdx.x);
}
int main(){
  i32 col_ptr[]={0,2,4,4}; i32 col_k[]={29,30,29,30}; double col_val[]={0.9998,0.02,0.02,0.9998};
  i32 *dp,*dk; double* dv; i64* dl; double* dc; i32* dks; double* dvals;
  hipMalloc(&dp,sizeof(col_ptr)); hipMalloc(&dk,sizeof(col_k)); hipMalloc(&dv,sizeof(col_val));
  hipMalloc(&dl,64*8); hipMalloc(&dc,64*8); hipMalloc(&dks,2*64*4); hipMalloc(&dvals,2*64*8);
  hipMemcpy(dp,col_ptr,sizeof(col_ptr),hipMemcpyHostToDevice); hipMemcpy(dk,col_k,sizeof(col_k),hipMemcpyHostToDevice); hipMemcpy(dv,col_val,sizeof(col_val),hipMemcpyHostToDevice);
  i32 ks[128]; double vals[128];
  for(int t=0;t<64;t++){ ks[t]=0; ks[64+t]=1; if(t%2){vals[t]=0.018; vals[64+t]=0.99999;} else {vals[t]=0.99999; vals[64+t]=0.018;} if (t%5==0) vals[64+t]=0; }
  hipMemcpy(dks,ks,sizeof(ks),hipMemcpyHostToDevice); hipMemcpy(dvals,vals,sizeof(vals),hipMemcpyHostToDevice);
  Fill2Args a{dp,dk,dv,1,0.8,dl,dc};
  hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,a,dks,dvals,2);
  i64 lab[64]; double conf[64]; hipMemcpy(lab,dl,sizeof(lab),hipMemcpyDeviceToHost); hipMemcpy(conf,dc,sizeof(conf),hipMemcpyDeviceToHost);
  for(int t=0;t<12;t++) printf("t=%d label=%ld conf=%g\n",t,(long)lab[t],conf[t]);
  return 0;
}
