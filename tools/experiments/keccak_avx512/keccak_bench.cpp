// Host-side Keccak-f[1600] as the transcript uses it: the append of an 8192-scalar vector (dot_product.rs:196) and the bare permutation, AVX-512 form vs scalar form
// (LASSO_KECCAK_AVX512=0).  Build: g++ -O2 -std=c++17 -march=x86-64-v3 -o keccak_bench tools/experiments/keccak_avx512/keccak_bench.cpp
#include "hashes_with_avx512_keccak.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
int main(){
  std::vector<uint8_t> b(32*8192); for (size_t i=0;i<b.size();i++) b[i]=(uint8_t)(i*131+7);
  lasso::Merlin m("example");
  auto t0=std::chrono::steady_clock::now(); const int R=20;
  for(int r=0;r<R;r++){ m.append_message("a","begin_append_vector",19); for(size_t i=0;i+32<=b.size();i+=32) m.append_message("a",&b[i],32); m.append_message("a","end_append_vector",17);}  
  auto t1=std::chrono::steady_clock::now(); double us=std::chrono::duration<double,std::micro>(t1-t0).count()/R;
  uint8_t out[32]; m.challenge_bytes("x",out,32);
  printf("append 8192 scalars: %.1f us (%.1f ns per scalar) %02x\n", us, us*1e3/8192, out[0]);
  lasso::Keccak1600 k; t0=std::chrono::steady_clock::now(); for(int i=0;i<200000;i++) k.permute(); t1=std::chrono::steady_clock::now();
  printf("permute: %.1f ns %llx\n", std::chrono::duration<double,std::nano>(t1-t0).count()/200000, (unsigned long long)k.A[1]);
}
