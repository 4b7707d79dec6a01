// development probe: what DS_APPEND addresses, returns and adds (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int *out)
{
    __shared__ int ctr[4];
    if (threadIdx.x == 0) { ctr[0] = 1000; ctr[1] = 0; ctr[2] = 2000; ctr[3] = 3000; }
    __syncthreads();
    int v;
    if (MODE == 0) asm volatile("s_mov_b32 m0, 0x00040000\n\tds_append %0\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : : "memory");
    if (MODE == 1) asm volatile("s_mov_b32 m0, 0\n\tds_append %0 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : : "memory");
    if (MODE == 2) asm volatile("s_mov_b32 m0, 8\n\tds_append %0 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : : "memory");
    __syncthreads();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) { out[512] = ctr[0]; out[513] = ctr[1]; out[514] = ctr[2]; out[515] = ctr[3]; }
}
template <int MODE> void run()
{
    int *d, h[516];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(128), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: v lane0 %d lane63 %d wave1 %d | ctr: %d %d %d %d\n", MODE, h[0], h[63], h[64], h[512], h[513], h[514], h[515]);
    hipFree(d);
}
int main() { run<0>(); run<1>(); run<2>(); return 0; }
