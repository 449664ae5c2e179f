// Driver around the REFERENCE's own host functions (compiled from
// /root/reference/src/acc/libsmm_acc/libsmm_acc_benchmark.cpp where it lies, see
// build_ref.sh): prints the inputs/outputs of its kernel validator so that the oracle's
// restatement (orc_mat_init / orc_stack_init / orc_stack_calc / orc_transpose_d) can be
// checked against the real thing.  TEST INFRASTRUCTURE ONLY.
//   ref_stack_driver m n k n_a n_b n_c n_stack  -> JSON on stdout
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// declarations only (the definitions live in the reference source file)
void matInit(double* mat, int mat_n, int x, int y, int seed);
void stackInit(int* stack, int n_stack, int n_c, int n_a, int n_b, int mat_m, int mat_n, int mat_k);
void stackInitTransp(int* stack, int n_stack, int mat_m, int mat_n);
void stackCalc(int* stack, int n_stack, double* mat_c, double* mat_a, double* mat_b, int mat_m, int mat_n, int mat_k);
void stackTransp(int* stack, int n_stack, double* mat_a, double* mat_atrs, int mat_m, int mat_n);
double checkSum(double* mat_c, int n_c, int mat_m, int mat_n);
double checkSumTransp(double* mat, int n_stack, int mat_m, int mat_n);

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const int m = atoi(argv[1]), n = atoi(argv[2]), k = atoi(argv[3]), na = atoi(argv[4]), nb = atoi(argv[5]), nc = atoi(argv[6]),
            ns = atoi(argv[7]);
  std::vector<double> a((size_t)na * m * k), b((size_t)nb * k * n), c((size_t)nc * m * n, 0.0), at((size_t)na * m * k, 0.0);
  std::vector<int> stack((size_t)3 * ns), trs((size_t)na);
  matInit(a.data(), na, m, k, 42);
  matInit(b.data(), nb, k, n, 24);
  srand(1);  // the reference relies on libc rand() in its default state; pin it so the oracle can replay it
  stackInit(stack.data(), ns, nc, na, nb, m, n, k);
  stackCalc(stack.data(), ns, c.data(), a.data(), b.data(), m, n, k);
  stackInitTransp(trs.data(), na, m, k);
  stackTransp(trs.data(), na, a.data(), at.data(), m, k);
  printf("{\"checksum\": %.17g, \"checksum_transp\": %.17g, \"a0\": %.17g, \"b_last\": %.17g, \"stack\": [", checkSum(c.data(), nc, m, n),
         checkSumTransp(at.data(), na, m, k), a[0], b[b.size() - 1]);
  for (size_t i = 0; i < stack.size(); ++i) printf("%s%d", i ? "," : "", stack[i]);
  printf("], \"c\": [");
  for (size_t i = 0; i < c.size(); ++i) printf("%s%.17g", i ? "," : "", c[i]);
  printf("]}\n");
  return 0;
}
