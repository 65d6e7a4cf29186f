// Which lane holds what in v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4)?  Random operands, then every assignment of the three
// 2-bit fields of the lane id to (block, row/col, k) is tried against the result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double ha[64], hb[64], hd[64], *a, *b, *d;
  srand(1);
  for (int i = 0; i < 64; ++i) ha[i] = rand() / (double)RAND_MAX, hb[i] = rand() / (double)RAND_MAX;
  hipMalloc(&a, 512), hipMalloc(&b, 512), hipMalloc(&d, 512);
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice), hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  // field f of the lane id = bits 2f, 2f+1; perm[p] = which field carries (block, index, k)
  auto lane_of = [&](const int* pm, int blk, int idx, int k) { return (blk << (2 * pm[0])) | (idx << (2 * pm[1])) | (k << (2 * pm[2])); };
  for (int pa = 0; pa < 6; ++pa)
    for (int pb = 0; pb < 6; ++pb)
      for (int pd = 0; pd < 6; ++pd) {
        double err = 0;
        for (int blk = 0; blk < 4; ++blk)
          for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
              double s = 0;
              for (int k = 0; k < 4; ++k) s += ha[lane_of(perm[pa], blk, i, k)] * hb[lane_of(perm[pb], blk, j, k)];
              // D: fields (block, i, j)
              const int ld = (blk << (2 * perm[pd][0])) | (i << (2 * perm[pd][1])) | (j << (2 * perm[pd][2]));
              err = fmax(err, fabs(s - hd[ld]));
            }
        if (err < 1e-12)
          printf("A: block field %d, row field %d, k field %d | B: block %d, col %d, k %d | D: block %d, row %d, col %d  (field f = lane bits 2f..2f+1)\n",
                 perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2]);
      }
  printf("done\n");
  return 0;
}
