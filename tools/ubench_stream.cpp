#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_read(const u4* __restrict__ buf, uint64_t n16, uint32_t* out, int nt){
  u4 acc={0,0,0,0};
  const uint64_t stride=(uint64_t)gridDim.x*blockDim.x;
  uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  for(; i+7*stride<n16; i+=8*stride){
    u4 v[8];
#pragma unroll
    for(int r=0;r<8;r++) v[r]= nt? __builtin_nontemporal_load(buf+i+r*stride) : buf[i+r*stride];
#pragma unroll
    for(int r=0;r<8;r++) acc^=v[r];
  }
  if((acc.x^acc.y^acc.z^acc.w)==0x12345678u) out[0]=1;
}
int main(){
  uint32_t* out; CHK(hipMalloc(&out,64));
  uint8_t* big; uint64_t bb=60ull<<30; CHK(hipMalloc(&big,bb)); CHK(hipMemset(big,1,bb));
  hipEvent_t a,b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for(int nt=0;nt<2;nt++) for(int blocks: {2048, 8192, 65536}){
    hipLaunchKernelGGL(stream_read,dim3(blocks),dim3(256),0,0,(const u4*)big,bb/16,out,nt); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); hipLaunchKernelGGL(stream_read,dim3(blocks),dim3(256),0,0,(const u4*)big,bb/16,out,nt); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms,a,b));
    printf("sequential read 60 GiB, %5d blocks, nt=%d: %.2f ms  %.0f GB/s\n",blocks,nt,ms,bb/ms/1e6);
  }
  return 0;
}
