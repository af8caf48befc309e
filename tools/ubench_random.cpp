// random row gather microbenchmark: rows of RB bytes at random 64B-aligned (or RB-aligned) positions in a big buffer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix(uint64_t x){x+=0x9e3779b97f4a7c15ULL;x=(x^(x>>30))*0xbf58476d1ce4e5b9ULL;x=(x^(x>>27))*0x94d049bb133111ebULL;return x^(x>>31);}
// LPR lanes per row, each lane 16 B; ROWS rows per lane-group per iteration (unrolled)
template<int LPR,int ROWS>
__global__ void __launch_bounds__(256) gather(const uint8_t* __restrict__ buf, uint64_t nrows, uint32_t stride, int iters, uint32_t* out, uint64_t seed){
  const int lane=threadIdx.x&63; const int g=lane/LPR, li=lane%LPR;
  const uint64_t gid=((uint64_t)blockIdx.x*4+(threadIdx.x>>6))*(64/LPR)+g;
  uint4 acc=make_uint4(0,0,0,0);
  for(int it=0;it<iters;it++){
    uint4 v[ROWS];
#pragma unroll
    for(int r=0;r<ROWS;r++){
      uint64_t row=mix(seed+gid*1000003ULL+(uint64_t)it*ROWS+r)%nrows;
      v[r]=*reinterpret_cast<const uint4*>(buf+row*stride+li*16);
    }
#pragma unroll
    for(int r=0;r<ROWS;r++){acc.x^=v[r].x;acc.y^=v[r].y;acc.z^=v[r].z;acc.w^=v[r].w;}
  }
  if((acc.x^acc.y^acc.z^acc.w)==0x12345678u) out[0]=1;
}
template<int LPR,int ROWS> double run(const uint8_t* buf,uint64_t bytes,uint32_t stride,uint32_t* out,int iters,uint64_t groups){
  uint64_t nrows=bytes/stride; uint64_t waves=groups/(64/LPR); unsigned blocks=(unsigned)(waves/4);
  hipEvent_t a,b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<LPR,ROWS>),dim3(blocks),dim3(256),0,0,buf,nrows,stride,2,out,1ULL); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a)); hipLaunchKernelGGL((gather<LPR,ROWS>),dim3(blocks),dim3(256),0,0,buf,nrows,stride,iters,out,7ULL); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms,a,b));
  double rows=(double)blocks*4*(64/LPR)*iters*ROWS;
  printf("LPR=%2d ROWS=%2d stride=%5u buf=%.1fGB: %.2f ms  %.1f Grows/s  %.0f GB/s (stride bytes)\n",LPR,ROWS,stride,bytes/1e9,ms,rows/ms/1e6,rows*LPR*16/ms/1e6);
  return ms;
}
int main(){
  uint64_t bytes=2300ull<<20; uint8_t* buf; uint32_t* out; CHK(hipMalloc(&buf,bytes)); CHK(hipMalloc(&out,64)); CHK(hipMemset(buf,1,bytes));
  const uint64_t G=1u<<22; // groups
  run<4,8>(buf,bytes,64,out,64,G*4); run<4,16>(buf,bytes,64,out,32,G*4); run<2,8>(buf,bytes,32,out,64,G*8); run<1,8>(buf,bytes,16,out,64,G*16);
  run<8,8>(buf,bytes,128,out,64,G*2); run<16,8>(buf,bytes,256,out,64,G); run<64,8>(buf,bytes,1024,out,64,G/4); run<64,8>(buf,bytes,1920,out,64,G/4);
  // big buffer (beyond MALL effects)
  uint8_t* big; uint64_t bb=40ull<<30; CHK(hipMalloc(&big,bb)); CHK(hipMemset(big,1,bb));
  run<4,8>(big,bb,64,out,64,G*4); run<64,8>(big,bb,1920,out,64,G/4); run<64,8>(big,bb,1024,out,64,G/4);
  return 0;
}
