// np.sum's summation order for f64 (pairwise inside 8192-element pieces of the ufunc buffer,
// the pieces accumulated in order), shared by the four-population sums and DD's means.
#pragma once

#include "common.hpp"

// numpy's *_pairwise_sum for n <= 128 (loops_utils.h.src): plain loop below 8 elements, else eight
// running sums, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail.
template <typename E>
__device__ __forceinline__ double pairwise_leaf(const E& e, int off, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += e(off + i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = e(off + j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += e(off + i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += e(off + i);
  return res;
}

// recursion of numpy's pairwise sum unrolled to the depth an 8192-element piece can reach: SEVEN
// levels -- the right halves are up to 7 elements larger than the left ones (n2 is rounded down to
// a multiple of 8), so pieces of 7689..8191 elements still hold a node of 129..135 elements six
// levels down, which numpy splits once more
template <int DEPTH, typename E>
struct PairwiseNode {
  static __device__ __noinline__ double run(const E& e, int off, int n) {
    if (n <= 128) return pairwise_leaf(e, off, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return PairwiseNode<DEPTH - 1, E>::run(e, off, n2) + PairwiseNode<DEPTH - 1, E>::run(e, off + n2, n - n2);
  }
};
template <typename E>
struct PairwiseNode<0, E> {
  static __device__ __noinline__ double run(const E& e, int off, int n) { return pairwise_leaf(e, off, n); }
};

// np.sum of elements e(off .. off+n): 8192-element pieces (the ufunc buffer), added in order
template <typename E>
__device__ __forceinline__ double numpy_sum(const E& e, int off, int n) {
  double res = 0.0;
  for (int o = 0; o < n; o += 8192) res += PairwiseNode<7, E>::run(e, off + o, min(8192, n - o));
  return res;
}
