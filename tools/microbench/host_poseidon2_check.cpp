// Host-only check and timing of the AVX-512 transcript permutation (csrc/host_poseidon2_simd.h)
// against the scalar template it replaces: random and edge-value states, both fields.
//   g++ -O3 -std=c++17 -I plonky3_recursion_amd/csrc tools/microbench/host_poseidon2_check.cpp -o /tmp/hp2 && /tmp/hp2
// Prints one line per field; exit code = number of mismatching states (0 expected), 77 = no AVX-512.
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "host_poseidon2_simd.h"
using namespace p3r;

template <class PP>
int run(const char* name) {
  using F = Fp<PP>;
  std::mt19937_64 g(7);
  std::vector<uint32_t> rc(p2_num_constants<PP>());
  for (auto& x : rc) x = (uint32_t)(g() % PP::P);
  int bad = 0;
  for (int t = 0; t < 20000; ++t) {
    F a[16], b[16];
    for (int i = 0; i < 16; ++i) {
      uint32_t v = (uint32_t)(g() % PP::P);
      if (t < 16) v = (t == i) ? PP::P - 1 : 0;     // one extreme element, the rest zero
      else if (t < 32) v = PP::P - 1 - (uint32_t)i;  // every element near the modulus
      a[i] = b[i] = F::raw(v);
    }
    p2_permute<PP>(a, rc.data());
    P2Avx512<PP>::permute(b, rc.data());
    for (int i = 0; i < 16; ++i)
      if (a[i].v != b[i].v) { ++bad; break; }
  }
  F s[16];
  for (int i = 0; i < 16; ++i) s[i] = F::raw(i + 1);
  const int N = 200000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) p2_permute<PP>(s, rc.data());
  auto t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) P2Avx512<PP>::permute(s, rc.data());
  auto t2 = std::chrono::steady_clock::now();
  printf("%s: mismatches %d of 20000; scalar %.0f ns, avx512 %.0f ns per permutation (%u)\n", name, bad,
         std::chrono::duration<double, std::nano>(t1 - t0).count() / N,
         std::chrono::duration<double, std::nano>(t2 - t1).count() / N, s[0].v);
  return bad;
}
int main() {
  if (!P2Avx512<KoalaBearParams>::supported()) {
    printf("no AVX-512 on this host: the scalar template is what runs\n");
    return 77;
  }
  return run<KoalaBearParams>("koala-bear") + run<BabyBearParams>("baby-bear");
}
