#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix(uint64_t x){x+=0x9e3779b97f4a7c15ULL;x=(x^(x>>30))*0xbf58476d1ce4e5b9ULL;x=(x^(x>>27))*0x94d049bb133111ebULL;return x^(x>>31);}
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template<int MODE,int ROWS>
__global__ void __launch_bounds__(256) gather(const uint8_t* __restrict__ buf, uint64_t nrows, uint32_t stride, uint32_t rowbytes, int iters, uint32_t* out, uint64_t seed){
  const int lane=threadIdx.x&63;
  const uint64_t gid=((uint64_t)blockIdx.x*4+(threadIdx.x>>6));
  const uint32_t tile=gid&1; const uint64_t unit=gid>>1;
  const uint32_t boff=(tile*64+lane)*16; const bool active=boff<rowbytes;
  u4 acc={0,0,0,0};
  for(int it=0;it<iters;it++){
    u4 v[ROWS];
#pragma unroll
    for(int r=0;r<ROWS;r++){
      uint64_t row=mix(seed+unit*1000003ULL+(uint64_t)it*ROWS+r)%nrows;
      const u4* p=reinterpret_cast<const u4*>(buf+row*stride+boff);
      u4 z={0,0,0,0};
      if(active){ if(MODE==1) z=__builtin_nontemporal_load(p); else z=*p; }
      v[r]=z;
    }
#pragma unroll
    for(int r=0;r<ROWS;r++) acc^=v[r];
  }
  if((acc.x^acc.y^acc.z^acc.w)==0x12345678u) out[0]=1;
}
template<int MODE,int ROWS> void run(const uint8_t* buf,uint64_t bytes,uint32_t stride,uint32_t rowbytes,uint32_t* out,int iters,uint64_t units,const char* tag){
  uint64_t nrows=bytes/stride-1; unsigned blocks=(unsigned)(units*2/4);
  hipEvent_t a,b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<MODE,ROWS>),dim3(blocks),dim3(256),0,0,buf,nrows,stride,rowbytes,2,out,1ULL); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a)); hipLaunchKernelGGL((gather<MODE,ROWS>),dim3(blocks),dim3(256),0,0,buf,nrows,stride,rowbytes,iters,out,7ULL); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms,a,b));
  double rows=(double)units*iters*ROWS;
  printf("%-10s ROWS=%2d stride=%5u rowbytes=%u: %.2f ms  %.0f GB/s useful(rowbytes)  %.0f GB/s stride\n",tag,ROWS,stride,rowbytes,ms,rows*rowbytes/ms/1e6,rows*stride/ms/1e6);
}
int main(){
  uint32_t* out; CHK(hipMalloc(&out,64));
  uint8_t* big; uint64_t bb=60ull<<30; CHK(hipMalloc(&big,bb)); CHK(hipMemset(big,1,bb));
  const uint64_t U=1u<<20;
  run<0,8>(big,bb,1920,1872,out,64,U,"plain"); run<1,8>(big,bb,1920,1872,out,64,U,"nt");
  run<0,8>(big,bb,1872,1872,out,64,U,"plain"); run<1,8>(big,bb,1872,1872,out,64,U,"nt");
  run<0,8>(big,bb,2048,1872,out,64,U,"plain"); run<0,16>(big,bb,1920,1872,out,32,U,"plain16"); run<0,4>(big,bb,1920,1872,out,128,U,"plain4");
  run<0,8>(big,bb,1920,1920,out,64,U,"full1920");
  return 0;
}
