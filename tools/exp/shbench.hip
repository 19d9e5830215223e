// micro-benchmark: what bounds the 16-lanes-per-Gaussian SH kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/gsraster.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ float row_sum16(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x111, 0xf, 0xf, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x112, 0xf, 0xf, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x114, 0xf, 0xf, true));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x118, 0xf, 0xf, true));
  return v;
}
// V0: pure copy-ish: 12B per lane load, dot with constant, row sum
__global__ __launch_bounds__(256) void v0(unsigned n, const float* __restrict__ dirs, const float* __restrict__ coeffs, float* __restrict__ colors){
  unsigned e = blockIdx.x*blockDim.x+threadIdx.x; unsigned g=e>>4; int k=e&15; if(g>=n) return;
  const float* c = coeffs + (size_t)g*48+3*k;
  float bk = 0.1f*k;
  float r=row_sum16(bk*c[0]), gg=row_sum16(bk*c[1]), b=row_sum16(bk*c[2]);
  if(k==15){colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;}
}
// V1: float4 loads, flat over the coefficient array (16B per lane), no per-gaussian structure: sum -> 1 store per lane
__global__ __launch_bounds__(256) void v1(unsigned n4, const float4* __restrict__ coeffs, float* __restrict__ out){
  unsigned e = blockIdx.x*blockDim.x+threadIdx.x; if(e>=n4) return;
  float4 v = coeffs[e];
  float s = v.x+v.y+v.z+v.w;
  if (s == 123.456f) out[e]=s;
}
// V2: one lane per gaussian, 12 x float4
__global__ __launch_bounds__(256) void v2(unsigned n, const float* __restrict__ dirs, const float4* __restrict__ coeffs, float* __restrict__ colors){
  unsigned g = blockIdx.x*blockDim.x+threadIdx.x; if(g>=n) return;
  const float4* c = coeffs + (size_t)g*12;
  float r=0,gg=0,b=0;
#pragma unroll
  for(int i=0;i<12;i+=3){ float4 a=c[i], bq=c[i+1], cq=c[i+2];
    r += a.x + a.w + bq.z + cq.y; gg += a.y + bq.x + bq.w + cq.z; b += a.z + bq.y + cq.x + cq.w; }
  colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;
}
// V3: like v0 but grid-stride with 2048 blocks
__global__ __launch_bounds__(256) void v3(unsigned n, const float* __restrict__ dirs, const float* __restrict__ coeffs, float* __restrict__ colors){
  unsigned stride = gridDim.x*blockDim.x;
  for (unsigned e = blockIdx.x*blockDim.x+threadIdx.x; e < n*16; e += stride){
    unsigned g=e>>4; int k=e&15;
    const float* c = coeffs + (size_t)g*48+3*k;
    float bk = 0.1f*k;
    float r=row_sum16(bk*c[0]), gg=row_sum16(bk*c[1]), b=row_sum16(bk*c[2]);
    if(k==15){colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;}
  }
}

__device__ __forceinline__ void basis16(float dx,float dy,float dz,float* B){
  const float inv = rsqrtf(dx*dx+dy*dy+dz*dz); const float x=dx*inv,y=dy*inv,z=dz*inv;
  B[0]=0.28209479f; B[1]=-0.48860251f*y; B[2]=0.48860251f*z; B[3]=-0.48860251f*x;
  const float xx=x*x,xy=x*y,xz=x*z,yy=y*y,yz=y*z,zz=z*z;
  B[4]=1.09254843f*xy; B[5]=-1.09254843f*yz; B[6]=0.31539157f*(2.f*zz-xx-yy); B[7]=-1.09254843f*xz; B[8]=0.54627422f*(xx-yy);
  B[9]=-0.59004359f*y*(3.f*xx-yy); B[10]=2.89061144f*xy*z; B[11]=-0.45704580f*y*(4.f*zz-xx-yy); B[12]=0.37317633f*z*(2.f*zz-3.f*xx-3.f*yy);
  B[13]=-0.45704580f*x*(4.f*zz-xx-yy); B[14]=1.44530572f*z*(xx-yy); B[15]=-0.59004359f*x*(xx-3.f*yy);
}
// F1: one lane per gaussian, float4 loads, real basis
__global__ __launch_bounds__(256) void f1(unsigned n, const float* __restrict__ dirs, const float4* __restrict__ coeffs, float* __restrict__ colors){
  unsigned g = blockIdx.x*blockDim.x+threadIdx.x; if(g>=n) return;
  const float4* c = coeffs + (size_t)g*12;
  float4 q[12];
#pragma unroll
  for(int i=0;i<12;i++) q[i]=c[i];
  float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
  const float* f = reinterpret_cast<const float*>(q);
  float r=0,gg=0,b=0;
#pragma unroll
  for(int k=0;k<16;k++){ r+=B[k]*f[3*k]; gg+=B[k]*f[3*k+1]; b+=B[k]*f[3*k+2]; }
  colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;
}
// F2: 16 lanes per gaussian with real basis + select
__global__ __launch_bounds__(256) void f2(unsigned n, const float* __restrict__ dirs, const float* __restrict__ coeffs, float* __restrict__ colors){
  unsigned e = blockIdx.x*blockDim.x+threadIdx.x; unsigned g=e>>4; int k=e&15; if(g>=n) return;
  const float* c = coeffs + (size_t)g*48+3*k;
  float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
  float bk=B[0];
#pragma unroll
  for(int j=1;j<16;j++) bk = (k==j)?B[j]:bk;
  float r=row_sum16(bk*c[0]), gg=row_sum16(bk*c[1]), b=row_sum16(bk*c[2]);
  if(k==15){colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;}
}
// B1: backward, one lane per gaussian, float4 stores
__global__ __launch_bounds__(256) void b1(unsigned n, const float* __restrict__ dirs, const float* __restrict__ vcol, float4* __restrict__ vco){
  unsigned g = blockIdx.x*blockDim.x+threadIdx.x; if(g>=n) return;
  float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
  const float vr=vcol[3*g],vg=vcol[3*g+1],vb=vcol[3*g+2];
  float f[48];
#pragma unroll
  for(int k=0;k<16;k++){ f[3*k]=B[k]*vr; f[3*k+1]=B[k]*vg; f[3*k+2]=B[k]*vb; }
  float4* o = vco + (size_t)g*12;
#pragma unroll
  for(int i=0;i<12;i++) o[i]=make_float4(f[4*i],f[4*i+1],f[4*i+2],f[4*i+3]);
}
// B2: backward 16 lanes per gaussian
__global__ __launch_bounds__(256) void b2(unsigned n, const float* __restrict__ dirs, const float* __restrict__ vcol, float* __restrict__ vco){
  unsigned e = blockIdx.x*blockDim.x+threadIdx.x; unsigned g=e>>4; int k=e&15; if(g>=n) return;
  float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
  float bk=B[0];
#pragma unroll
  for(int j=1;j<16;j++) bk = (k==j)?B[j]:bk;
  float* o = vco + (size_t)g*48+3*k;
  o[0]=bk*vcol[3*g]; o[1]=bk*vcol[3*g+1]; o[2]=bk*vcol[3*g+2];
}
// F3: one lane per gaussian for the math, but the wave moves its 64 x 192 B as 12
// fully coalesced 1-KB dwordx4 rows and transposes through LDS (rows padded to 13 float4)
__global__ __launch_bounds__(256) void f3(unsigned n, const float* __restrict__ dirs, const float4* __restrict__ coeffs, float* __restrict__ colors){
  __shared__ float4 lds[4][64*13];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x*4 + w)*64;        // first gaussian of this wave
  if (g0 >= n) return;
  const float4* src = coeffs + (size_t)g0*12;
  const unsigned navail = (n - g0 < 64 ? n - g0 : 64)*12;
  float4 q[12];
#pragma unroll
  for(int i=0;i<12;i++){ unsigned j=i*64+lane; q[i] = j<navail ? src[j] : make_float4(0,0,0,0); }
#pragma unroll
  for(int i=0;i<12;i++){ unsigned j=i*64+lane; lds[w][(j/12)*13 + j%12] = q[i]; }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const unsigned g = g0+lane;
#pragma unroll
  for(int i=0;i<12;i++) q[i] = lds[w][lane*13+i];
  if (g>=n) return;
  float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
  const float* f = reinterpret_cast<const float*>(q);
  float r=0,gg=0,b=0;
#pragma unroll
  for(int k=0;k<16;k++){ r+=B[k]*f[3*k]; gg+=B[k]*f[3*k+1]; b+=B[k]*f[3*k+2]; }
  colors[3*g]=r;colors[3*g+1]=gg;colors[3*g+2]=b;
}
// B3: backward with the same transposition before the stores
__global__ __launch_bounds__(256) void b3(unsigned n, const float* __restrict__ dirs, const float* __restrict__ vcol, float4* __restrict__ vco){
  __shared__ float4 lds[4][64*13];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x*4 + w)*64;
  if (g0 >= n) return;
  const unsigned g = g0+lane;
  float f[48];
  if (g<n){
    float B[16]; basis16(dirs[3*g],dirs[3*g+1],dirs[3*g+2],B);
    const float vr=vcol[3*g],vg=vcol[3*g+1],vb=vcol[3*g+2];
#pragma unroll
    for(int k=0;k<16;k++){ f[3*k]=B[k]*vr; f[3*k+1]=B[k]*vg; f[3*k+2]=B[k]*vb; }
  } else {
#pragma unroll
    for(int k=0;k<48;k++) f[k]=0.f;
  }
#pragma unroll
  for(int i=0;i<12;i++) lds[w][lane*13+i] = make_float4(f[4*i],f[4*i+1],f[4*i+2],f[4*i+3]);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float4* dst = vco + (size_t)g0*12;
  const unsigned navail = (n - g0 < 64 ? n - g0 : 64)*12;
#pragma unroll
  for(int i=0;i<12;i++){ unsigned j=i*64+lane; if (j<navail) dst[j] = lds[w][(j/12)*13 + j%12]; }
}
int main(){
  unsigned n=1000000; size_t nb=(size_t)n*48*4;
  float *coeffs,*dirs,*colors; CK(hipMalloc(&coeffs,nb)); CK(hipMalloc(&dirs,n*12)); CK(hipMalloc(&colors,(size_t)3000000*48*4));
  CK(hipMemset(coeffs,0,nb)); CK(hipMemset(dirs,0,n*12));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto f){ for(int i=0;i<3;i++) f(); hipEventRecord(a); for(int i=0;i<20;i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); printf("%s: %.1f us  (%.0f GB/s on 192MB)\n", name, ms/20*1000, nb/(ms/20*1e-3)/1e9); };
  run("v0 16lanes x3 loads", [&]{ hipLaunchKernelGGL(v0, dim3((n*16+255)/256), dim3(256),0,0,n,dirs,coeffs,colors); });
  run("v1 flat float4", [&]{ hipLaunchKernelGGL(v1, dim3((n*12+255)/256), dim3(256),0,0,n*12,(const float4*)coeffs,colors); });
  run("v2 lane/gaussian 12xfloat4", [&]{ hipLaunchKernelGGL(v2, dim3((n+255)/256), dim3(256),0,0,n,dirs,(const float4*)coeffs,colors); });
  run("v3 gridstride 2048", [&]{ hipLaunchKernelGGL(v3, dim3(2048), dim3(256),0,0,n,dirs,coeffs,colors); });
  run("v3 gridstride 8192", [&]{ hipLaunchKernelGGL(v3, dim3(8192), dim3(256),0,0,n,dirs,coeffs,colors); });
  // real data, larger than the 256 MiB infinity cache: 3M gaussians = 576 MB
  { unsigned n3=3000000; size_t nb3=(size_t)n3*48*4; float *co3,*d3,*c3; CK(hipMalloc(&co3,nb3)); CK(hipMalloc(&d3,(size_t)n3*12)); CK(hipMalloc(&c3,(size_t)n3*12));
    std::vector<float> h((size_t)n3*3); for(size_t i=0;i<h.size();i++) h[i]=0.1f+((i*2654435761u)%1000)/1000.f; CK(hipMemcpy(d3,h.data(),h.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(c3,h.data(),h.size()*4,hipMemcpyHostToDevice)); CK(hipMemset(co3,0,nb3));
    nb=nb3;
    run("[3M] f1 lane/gaussian float4 + basis", [&]{ hipLaunchKernelGGL(f1, dim3((n3+255)/256), dim3(256),0,0,n3,d3,(const float4*)co3,colors); });
    run("[3M] f2 16 lanes + basis + select", [&]{ hipLaunchKernelGGL(f2, dim3((n3*16+255)/256), dim3(256),0,0,n3,d3,co3,colors); });
    run("[3M] b1 lane/gaussian float4 stores", [&]{ hipLaunchKernelGGL(b1, dim3((n3+255)/256), dim3(256),0,0,n3,d3,c3,(float4*)co3); });
    run("[3M] PROD gsr_sh_forward deg3", [&]{ gsr_sh_forward(n3,3,3,d3,co3,colors,0); });
    run("[3M] PROD gsr_sh_backward deg3", [&]{ gsr_sh_backward(n3,3,3,d3,c3,co3,0); });
    run("[3M] f3 coalesced rows + LDS transpose", [&]{ hipLaunchKernelGGL(f3, dim3((n3+255)/256), dim3(256),0,0,n3,d3,(const float4*)co3,colors); });
    run("[3M] b3 LDS transpose + coalesced rows", [&]{ hipLaunchKernelGGL(b3, dim3((n3+255)/256), dim3(256),0,0,n3,d3,c3,(float4*)co3); });
    { unsigned n1=1000000; nb=(size_t)n1*48*4;
      run("[1M] f1", [&]{ hipLaunchKernelGGL(f1, dim3((n1+255)/256), dim3(256),0,0,n1,d3,(const float4*)co3,colors); });
      run("[1M] f3", [&]{ hipLaunchKernelGGL(f3, dim3((n1+255)/256), dim3(256),0,0,n1,d3,(const float4*)co3,colors); });
      run("[1M] b1", [&]{ hipLaunchKernelGGL(b1, dim3((n1+255)/256), dim3(256),0,0,n1,d3,c3,(float4*)co3); });
      run("[1M] b3", [&]{ hipLaunchKernelGGL(b3, dim3((n1+255)/256), dim3(256),0,0,n1,d3,c3,(float4*)co3); });
      nb=nb3; }
    run("[3M] b2 16 lanes stores", [&]{ hipLaunchKernelGGL(b2, dim3((n3*16+255)/256), dim3(256),0,0,n3,d3,c3,co3); });
  }
  CK(hipDeviceSynchronize());
  return 0;
}
