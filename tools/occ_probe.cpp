#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) dummy(int *p) { extern __shared__ int s[]; s[threadIdx.x] = p[0]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x+1)%512]; }
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("name %s sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d regsPerMP %d maxThreadsPerMP %d CUs %d maxBlocksPerMP %d sharedMemPerBlockOptin %zu\n", pr.name, pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock, pr.regsPerMultiprocessor, pr.maxThreadsPerMultiProcessor, pr.multiProcessorCount, pr.maxBlocksPerMultiProcessor, pr.sharedMemPerBlockOptin);
  for (int lds : {16384, 32768, 40960, 46080, 49152, 53248, 65536, 81920}) {
    hipFuncSetAttribute((const void*)dummy, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int nb = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dummy, 512, lds);
    printf("lds %d -> blocks/CU %d (%s)\n", lds, nb, hipGetErrorString(e));
  }
  return 0;
}
