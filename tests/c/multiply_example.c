/* A host written in plain C on top of the C-ABI alone: device memory and transfers through the acc interface
 * (include/dbcsr_acc.h), the multiply through dbcsr_amd_multiply (include/dbcsr_amd_mm.h).
 * Builds two small block-sparse matrices, computes C = beta*C + alpha*A*B on the GPU and checks it against a dense
 * triple loop.  Exit code 0 = ok.   cc multiply_example.c -I../../include -L../../dbcsr_amd -ldbcsr_acc_amd */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dbcsr_acc.h"
#include "dbcsr_amd_mm.h"

#define NB 6 /* block rows = block columns */
#define CHECK(x)                                              \
  do {                                                        \
    int rc_ = (x);                                            \
    if (rc_ != 0) {                                           \
      fprintf(stderr, "%s failed (%d)\n", #x, rc_);           \
      return 1;                                               \
    }                                                         \
  } while (0)

typedef struct {
  int32_t row_p[NB + 1], col_i[NB * NB];
  int64_t blk_p[NB * NB], nblks, nze;
  double data[64 * NB * NB];
} host_bcsr;

static const int32_t sizes[NB] = {3, 5, 2, 7, 4, 6};

static void make(host_bcsr* m, unsigned seed, int keep_mod) {
  m->nblks = m->nze = 0;
  for (int r = 0; r < NB; ++r) {
    m->row_p[r] = (int32_t)m->nblks;
    for (int c = 0; c < NB; ++c) {
      seed = seed * 1103515245u + 12345u;
      if ((seed >> 16) % keep_mod == 0) continue;
      m->col_i[m->nblks] = c;
      m->blk_p[m->nblks] = m->nze;
      for (int e = 0; e < sizes[r] * sizes[c]; ++e) {
        seed = seed * 1103515245u + 12345u;
        m->data[m->nze++] = (double)((seed >> 16) & 1023) / 512.0 - 1.0;
      }
      ++m->nblks;
    }
  }
  m->row_p[NB] = (int32_t)m->nblks;
}

static void to_dense(const host_bcsr* m, double* d, int n) {
  int off[NB + 1] = {0};
  for (int i = 0; i < NB; ++i) off[i + 1] = off[i] + sizes[i];
  memset(d, 0, sizeof(double) * (size_t)n * n);
  for (int r = 0; r < NB; ++r)
    for (int b = m->row_p[r]; b < m->row_p[r + 1]; ++b) {
      const int c = m->col_i[b];
      for (int j = 0; j < sizes[c]; ++j)
        for (int i = 0; i < sizes[r]; ++i) d[(size_t)(off[r] + i) * n + off[c] + j] = m->data[m->blk_p[b] + i + (int64_t)sizes[r] * j];
    }
}

static int upload(const host_bcsr* h, const int32_t* dsizes, dbcsr_amd_bcsr* d) {
  d->nblkrows = d->nblkcols = NB;
  d->row_blk_size = d->col_blk_size = dsizes;
  d->nblks = h->nblks;
  d->index_stamp = 0; /* unknown generation: plans are compared on the device */
  CHECK(c_dbcsr_acc_dev_mem_allocate((void**)&d->row_p, sizeof(h->row_p)));
  CHECK(c_dbcsr_acc_dev_mem_allocate((void**)&d->col_i, sizeof(int32_t) * (size_t)(h->nblks + 1)));
  CHECK(c_dbcsr_acc_dev_mem_allocate((void**)&d->blk_p, sizeof(int64_t) * (size_t)(h->nblks + 1)));
  CHECK(c_dbcsr_acc_dev_mem_allocate(&d->data, sizeof(double) * (size_t)(h->nze + 1)));
  CHECK(c_dbcsr_acc_memcpy_h2d(h->row_p, d->row_p, sizeof(h->row_p), NULL));
  CHECK(c_dbcsr_acc_memcpy_h2d(h->col_i, d->col_i, sizeof(int32_t) * (size_t)h->nblks, NULL));
  CHECK(c_dbcsr_acc_memcpy_h2d(h->blk_p, d->blk_p, sizeof(int64_t) * (size_t)h->nblks, NULL));
  CHECK(c_dbcsr_acc_memcpy_h2d(h->data, d->data, sizeof(double) * (size_t)h->nze, NULL));
  return 0;
}

int main(void) {
  static host_bcsr A, B, Cm, R;
  const double alpha = -0.75, beta = 1.5;
  int n = 0;
  for (int i = 0; i < NB; ++i) n += sizes[i];
  make(&A, 1u, 2);
  make(&B, 7u, 3);
  make(&Cm, 13u, 4);
  CHECK(c_dbcsr_acc_set_active_device(0));
  CHECK(c_dbcsr_acc_init());
  int32_t* dsizes = NULL;
  CHECK(c_dbcsr_acc_dev_mem_allocate((void**)&dsizes, sizeof(sizes)));
  CHECK(c_dbcsr_acc_memcpy_h2d(sizes, dsizes, sizeof(sizes), NULL));
  dbcsr_amd_bcsr dA, dB, dC, dR;
  if (upload(&A, dsizes, &dA) || upload(&B, dsizes, &dB) || upload(&Cm, dsizes, &dC)) return 1;
  void* h = NULL;
  CHECK(dbcsr_amd_mm_create(&h));
  int64_t flop = 0;
  CHECK(dbcsr_amd_multiply(h, 'N', 'N', dbcsr_type_real_8, alpha, &dA, &dB, beta, &dC, NULL, 0, 0.0, &dR, &flop, NULL));
  /* download the result */
  R.nblks = dR.nblks;
  CHECK(c_dbcsr_acc_memcpy_d2h(dR.row_p, R.row_p, sizeof(R.row_p), NULL));
  CHECK(c_dbcsr_acc_memcpy_d2h(dR.col_i, R.col_i, sizeof(int32_t) * (size_t)R.nblks, NULL));
  CHECK(c_dbcsr_acc_memcpy_d2h(dR.blk_p, R.blk_p, sizeof(int64_t) * (size_t)R.nblks, NULL));
  CHECK(c_dbcsr_acc_device_synchronize());
  R.nze = 0;
  for (int r = 0; r < NB; ++r)
    for (int b = R.row_p[r]; b < R.row_p[r + 1]; ++b) R.nze += (int64_t)sizes[r] * sizes[R.col_i[b]];
  CHECK(c_dbcsr_acc_memcpy_d2h(dR.data, R.data, sizeof(double) * (size_t)R.nze, NULL));
  CHECK(c_dbcsr_acc_device_synchronize());
  /* dense check */
  double *a = malloc(sizeof(double) * n * n), *b = malloc(sizeof(double) * n * n), *c = malloc(sizeof(double) * n * n),
         *r = malloc(sizeof(double) * n * n);
  to_dense(&A, a, n);
  to_dense(&B, b, n);
  to_dense(&Cm, c, n);
  to_dense(&R, r, n);
  double err = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = beta * c[(size_t)i * n + j];
      for (int k = 0; k < n; ++k) s += alpha * a[(size_t)i * n + k] * b[(size_t)k * n + j];
      err = fmax(err, fabs(s - r[(size_t)i * n + j]));
    }
  printf("multiply_example: %lld C blocks, flop %lld, max abs err %.3e\n", (long long)R.nblks, (long long)flop, err);
  CHECK(dbcsr_amd_bcsr_release(&dR));
  CHECK(dbcsr_amd_mm_destroy(h));
  CHECK(c_dbcsr_acc_finalize());
  return err < 1e-12 ? 0 : 2;
}
