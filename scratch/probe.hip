#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_axpy(const float* x, float* y, float a, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i] = a*x[i]+y[i];
}
// one wave: C(32x32) = A(32xK) * B(Kx32), A row-major [32][K], B row-major [K][32]
__global__ void k_mfma(const float* A, const float* B, float* C, int K){
  int l = threadIdx.x; int i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for(int k=0;k<K;k+=2){
    float a = A[i*K + k + h];
    float b = B[(k+h)*32 + i];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc,0,0,0);
  }
  for(int r=0;r<16;r++){
    int row = (r&3) + 8*(r>>2) + 4*h;
    C[row*32 + i] = acc[r];
  }
}
extern "C" int probe_axpy(const float* x, float* y, float a, int n, void* stream){
  hipLaunchKernelGGL(k_axpy, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, x,y,a,n);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma(const float* A, const float* B, float* C, int K, void* stream){
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, A,B,C,K);
  return (int)hipGetLastError();
}
extern "C" int probe_rtver(){ int v=0; hipRuntimeGetVersion(&v); return v; }
