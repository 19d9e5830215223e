// which rocPRIM configuration sorts 1M (u32 key, i32 value) pairs fastest on MI355X?
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<class Config> void run(const char* name, unsigned n, unsigned* kin, unsigned* kout, int* vout){
  size_t tb=0; CK((rocprim::radix_sort_pairs<Config>(nullptr,tb,(const unsigned*)kin,kout,rocprim::counting_iterator<int>(0),vout,(size_t)n,0u,31u)));
  void* tmp; CK(hipMalloc(&tmp,tb));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<3;i++) CK((rocprim::radix_sort_pairs<Config>(tmp,tb,(const unsigned*)kin,kout,rocprim::counting_iterator<int>(0),vout,(size_t)n,0u,31u)));
  hipEventRecord(a); for(int i=0;i<20;i++) CK((rocprim::radix_sort_pairs<Config>(tmp,tb,(const unsigned*)kin,kout,rocprim::counting_iterator<int>(0),vout,(size_t)n,0u,31u)));
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b);
  printf("%-44s n=%u: %.1f us\n", name, n, ms/20*1000); hipFree(tmp);
}
using namespace rocprim;
template<class M> using RC = radix_sort_config<default_config, M, default_config, (1u<<22)>;
int main(){
  for (unsigned n : {1000000u, 3000000u}) {
    std::vector<unsigned> h(n); std::mt19937 g(1); for(auto& x:h){ float z = 2.f + 8.f*(g()/4294967296.f); memcpy(&x,&z,4);} 
    unsigned *kin,*kout; int* vout; CK(hipMalloc(&kin,n*4)); CK(hipMalloc(&kout,n*4)); CK(hipMalloc(&vout,n*4)); CK(hipMemcpy(kin,h.data(),n*4,hipMemcpyHostToDevice));
    run<default_config>("default", n,kin,kout,vout);
    run<RC<default_config>>("merge path forced (limit 4M), default cfg", n,kin,kout,vout);
    run<RC<merge_sort_config<512,256,8,128,128,4>>>("merge 256x8 block, mp 128x4", n,kin,kout,vout);
    run<RC<merge_sort_config<512,256,16,128,128,4>>>("merge 256x16 block, mp 128x4", n,kin,kout,vout);
    run<RC<merge_sort_config<512,512,8,128,128,4>>>("merge 512x8 block, mp 128x4", n,kin,kout,vout);
    run<RC<merge_sort_config<512,256,16,128,256,8>>>("merge 256x16 block, mp 256x8", n,kin,kout,vout);
    run<RC<merge_sort_config<512,512,16,128,256,8>>>("merge 512x16 block, mp 256x8", n,kin,kout,vout);
    run<RC<merge_sort_config<512,256,16,128,256,4>>>("merge 256x16 block, mp 256x4", n,kin,kout,vout);
    using OS2 = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256,12>, kernel_config<256,16>, 8>, 65536>;
    run<OS2>("onesweep 256x16 8bit", n,kin,kout,vout);
    hipFree(kin); hipFree(kout); hipFree(vout);
  }
  return 0;
}
