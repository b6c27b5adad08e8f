/*
 * dgs_oracle.c -- CPU restatement of the dgSPARSE CSR SpMM / SDDMM / csr2csc hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (dgsparse-lib_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * Every function restates one reference kernel; citations are relative to /root/reference:
 *
 *   orc_spmm_csr_f32        <- include/cuda/spmm_cuda.cuh:10-55 (csrspmm_seqreduce_rowbalance_kernel,
 *                              "algorithm 0", the numerical contract) + include/gspmm.h:13-146
 *                              (REDUCEOP enum values, MIN/MAX macros, init()).
 *                              For reduce==SUM, fma==0 it is also bit-identical to
 *                              example/util/sp_util.hpp:63-84 (spmm_reference_host).
 *   orc_spmm_csr_mask_f32   <- the formula behind include/cuda/spmm_cuda.cuh:400-433
 *                              (max/min backward w.r.t. the dense operand; the reference kernel
 *                              has a stale-variable bug, SURVEY.md 8(a) a9 - the formula is restated).
 *   orc_sddmm_csr_f32       <- example/util/sp_util.hpp:88-112 (sddmm_reference_host, sequential k)
 *                              + MEAN rule of include/cuda/sddmm_cuda.cuh:266-272,304-306.
 *   orc_sddmm_csr_mask_f32  <- include/cuda/sddmm_cuda.cuh:403-507 (sddmmCSR1Scale_with_mask).
 *   orc_csr2csc_i32         <- include/cuda/csr2csc.cuh:8-26 (cusparseCsr2cscEx2, stable) as pinned
 *                              by test/test_csr2csr.py:40-49 against scipy tocsc().
 *
 * Parity pinning (see oracle/README.md, tests/test_oracle_pin.py):
 *   - SUM / SDDMM: bit-exact against the reference's own spmm_reference_host / sddmm_reference_host
 *     compiled from /root/reference into oracle/_ref (oracle/Makefile).
 *   - SUM/MEAN/MAX/MIN + arg index E: against torch.sparse.mm(csr, X, reduce) and
 *     aten::_sparse_mm_reduce_impl on CPU, which is the oracle the reference's own pytest uses
 *     (test/test_spmm.py:25,60,97,134); committed as tests/golden/ (npz files).
 *   - csr2csc: against scipy tocsc() on the reference's fixture example/data/p2p-Gnutella31.mtx.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (no -march flags: products and sums are
 * separately rounded unless fma!=0 is requested, in which case fmaf() gives the single rounding
 * that nvcc/hipcc contraction produces on the device).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* include/gspmm.h:13 */
enum { ORC_SUM = 0, ORC_MAX = 1, ORC_MIN = 2, ORC_MEAN = 3 };

/* include/gspmm.h:133-146: MAX/MIN identities are INT_MIN/INT_MAX converted to float. */
static inline float orc_init(int op) {
  switch (op) {
  case ORC_MAX:
    return (float)INT_MIN;
  case ORC_MIN:
    return (float)INT_MAX;
  default:
    return 0.0f;
  }
}

static void orc_spmm_row(int op, int fma, int64_t N, int64_t r, const int32_t *rowptr,
                         const int32_t *col, const float *val, const float *B, float *C,
                         int32_t *E) {
  const int64_t s = rowptr[r], e = rowptr[r + 1];
  float *c = C + r * N;
  int32_t *er = E ? E + r * N : NULL;
  if (e - s <= 0) { /* spmm_cuda.cuh:49-51: empty row -> 0, E = -1 */
    for (int64_t f = 0; f < N; f++) {
      c[f] = 0.0f;
      if (er) er[f] = -1;
    }
    return;
  }
  for (int64_t f = 0; f < N; f++) {
    float res = orc_init(op);
    int32_t eidx = -1;
    for (int64_t p = s; p < e; p++) {
      const int32_t k = col[p];
      const float w = val ? val[p] : 1.0f; /* cuda_util.cuh:140-146 */
      const float x = B[(int64_t)k * N + f];
      const float t = w * x; /* spmm_cuda.cuh:37, one fp32 rounding */
      /* spmm_cuda.cuh:38-41: strict compare -> first occurrence in CSR order wins */
      if ((op == ORC_MAX && res < t) || (op == ORC_MIN && res > t)) eidx = k;
      switch (op) { /* gspmm.h:16-17 macros, evaluated literally (NaN behaviour included) */
      case ORC_MAX:
        res = (res < t) ? t : res;
        break;
      case ORC_MIN:
        res = (res < t) ? res : t;
        break;
      default:
        res = fma ? fmaf(w, x, res) : res + t;
      }
    }
    if (op == ORC_MEAN) res /= (float)(e - s); /* spmm_cuda.cuh:45-47 */
    c[f] = res;
    if (er) er[f] = eidx;
  }
}

/* C[M,N] = reduce_p val[p] * B[col[p], :]; E (nullable) receives the arg column id. */
int orc_spmm_csr_f32(int reduce_op, int fma, int64_t M, int64_t K, int64_t N, int64_t nnz,
                     const int32_t *rowptr, const int32_t *col, const float *val, const float *B,
                     float *C, int32_t *E, int threads) {
  (void)K;
  (void)nnz;
  if (reduce_op < 0 || reduce_op > 3) return -1;
#ifdef _OPENMP
  if (threads > 1) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t r = 0; r < M; r++) orc_spmm_row(reduce_op, fma, N, r, rowptr, col, val, B, C, E);
    return 0;
  }
#endif
  (void)threads;
  for (int64_t r = 0; r < M; r++) orc_spmm_row(reduce_op, fma, N, r, rowptr, col, val, B, C, E);
  return 0;
}

/*
 * Backward of max/min w.r.t. the dense operand, run on the CSC arrays of A:
 *   out[j,f] = sum_{p in [ptr[j],ptr[j+1])} [E[idx[p], f] == j] * val[p] * G[idx[p], f]
 * (ptr/idx/val = colptr/row/values permuted to CSC order; G = grad of the SpMM output;
 *  E = arg column ids saved by the forward).  Rows of `out` = columns of A.
 */
int orc_spmm_csr_mask_f32(int fma, int64_t Mout, int64_t N, const int32_t *ptr, const int32_t *idx,
                          const float *val, const float *G, const int32_t *E, float *out) {
  for (int64_t j = 0; j < Mout; j++) {
    for (int64_t f = 0; f < N; f++) {
      float res = 0.0f;
      for (int64_t p = ptr[j]; p < ptr[j + 1]; p++) {
        const int64_t i = idx[p];
        if (E[i * N + f] == (int32_t)j) {
          const float w = val ? val[p] : 1.0f;
          const float g = G[i * N + f];
          res = fma ? fmaf(w, g, res) : res + w * g;
        }
      }
      out[j * N + f] = res;
    }
  }
  return 0;
}

/* out[e] = sum_k D1[row(e),k] * D2[col(e),k]  (/ deg(row(e)) when MEAN and deg>0). */
int orc_sddmm_csr_f32(int reduce_op, int fma, int64_t M, int64_t F, int64_t nnz,
                      const int32_t *rowptr, const int32_t *col, const float *D1, const float *D2,
                      float *out, int threads) {
  (void)nnz;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 1 ? threads : 1)
#endif
  for (int64_t i = 0; i < M; i++) {
    const int64_t lb = rowptr[i], hb = rowptr[i + 1];
    for (int64_t p = lb; p < hb; p++) {
      const float *a = D1 + i * F;
      const float *b = D2 + (int64_t)col[p] * F;
      float acc = 0.0f;
      for (int64_t k = 0; k < F; k++) acc = fma ? fmaf(a[k], b[k], acc) : acc + a[k] * b[k];
      if (reduce_op == ORC_MEAN && hb - lb > 0) acc /= (float)(hb - lb);
      out[p] = acc;
    }
  }
  return 0;
}

/* out[e] = sum_k [E[row(e),k] == col(e)] * D1[row(e),k] * D2[col(e),k] */
int orc_sddmm_csr_mask_f32(int fma, int64_t M, int64_t F, int64_t nnz, const int32_t *rowptr,
                           const int32_t *col, const float *D1, const float *D2, const int32_t *E,
                           float *out) {
  (void)nnz;
  for (int64_t i = 0; i < M; i++) {
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
      const int32_t c = col[p];
      const float *a = D1 + i * F;
      const float *b = D2 + (int64_t)c * F;
      const int32_t *e = E + i * F;
      float acc = 0.0f;
      for (int64_t k = 0; k < F; k++)
        if (e[k] == c) acc = fma ? fmaf(a[k], b[k], acc) : acc + a[k] * b[k];
      out[p] = acc;
    }
  }
  return 0;
}

/*
 * Stable CSR -> CSC: colptr[j] = #entries with col < j; inside a column entries keep CSR order
 * (increasing row, then CSR position).  perm[q] = CSR position of CSC slot q.
 */
int orc_csr2csc_i32(int64_t M, int64_t Kcols, int64_t nnz, const int32_t *rowptr,
                    const int32_t *col, const float *val, int32_t *colptr, int32_t *row,
                    float *cscval, int32_t *perm) {
  int64_t *cursor = (int64_t *)calloc((size_t)Kcols + 1, sizeof(int64_t));
  if (!cursor) return -2;
  for (int64_t p = 0; p < nnz; p++) {
    if (col[p] < 0 || col[p] >= Kcols) {
      free(cursor);
      return -3;
    }
    cursor[col[p] + 1]++;
  }
  for (int64_t j = 0; j < Kcols; j++) cursor[j + 1] += cursor[j];
  for (int64_t j = 0; j <= Kcols; j++) colptr[j] = (int32_t)cursor[j];
  for (int64_t i = 0; i < M; i++) {
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
      const int64_t q = cursor[col[p]]++;
      if (row) row[q] = (int32_t)i;
      if (cscval && val) cscval[q] = val[p];
      if (perm) perm[q] = (int32_t)p;
    }
  }
  free(cursor);
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/*
 * Exact-arithmetic yardstick for sum/mean (NOT a restatement of the reference): products and the running
 * sum are carried in double (a float*float product is exact in double), result left in double.  Used by
 * the tests to judge long rows, where the reference's own sequential fp32 chain (spmm_cuda.cuh:27-44) is
 * itself more than 1e-5 away from the exact sum: a split (tree) summation must then be at least as close
 * to the exact value as the sequential chain is.
 */
int orc_spmm_sum_f64(int mean, int absval, int64_t M, int64_t N, const int32_t *rowptr, const int32_t *col,
                     const float *val, const float *B, double *C) {
  for (int64_t r = 0; r < M; r++) {
    const int64_t s = rowptr[r], e = rowptr[r + 1];
    for (int64_t f = 0; f < N; f++) {
      double acc = 0.0;
      for (int64_t p = s; p < e; p++) {
        const double t = (double)(val ? val[p] : 1.0f) * (double)B[(int64_t)col[p] * N + f];
        acc += absval ? fabs(t) : t; /* absval: the condition scale sum|t| of the row */
      }
      if (mean && e > s) acc /= (double)(e - s);
      C[r * N + f] = acc;
    }
  }
  return 0;
}

/*
 * Generalised SpMM of the reference's gspmm-fp demo module: out[r,f] = reduce_p compute(val[p], B[col[p],f]) with
 * compute in {ADD a+b, SUB b-a, MUL a*b, DIV b/a} (src/gspmm-fp/gspmm.h:15-79: enum COMPUTEOP { ADD, SUB, MUL, DIV })
 * and the simple sequential kernel weightedSimpleSPMMKernel (src/gspmm-fp/gspmm.cu:212-245): acc starts at
 * init(reduce) for non-empty rows, 0 for empty ones; MEAN divides by the row length.  No arg output.
 */
int orc_gspmm_csr_f32(int reduce_op, int compute_op, int64_t M, int64_t N, const int32_t *rowptr, const int32_t *col,
                      const float *val, const float *B, float *C) {
  if (reduce_op < 0 || reduce_op > 3 || compute_op < 0 || compute_op > 3) return -1;
  for (int64_t r = 0; r < M; r++) {
    const int64_t s = rowptr[r], e = rowptr[r + 1];
    for (int64_t f = 0; f < N; f++) {
      float acc = (e > s) ? orc_init(reduce_op) : 0.0f;
      for (int64_t p = s; p < e; p++) {
        const float a = val ? val[p] : 1.0f, b = B[(int64_t)col[p] * N + f];
        float t;
        switch (compute_op) {
        case 0: t = a + b; break;
        case 1: t = b - a; break;
        case 2: t = a * b; break;
        default: t = b / a;
        }
        switch (reduce_op) {
        case ORC_MAX: acc = (acc < t) ? t : acc; break;
        case ORC_MIN: acc = (acc < t) ? acc : t; break;
        default: acc = acc + t;
        }
      }
      if (reduce_op == ORC_MEAN && e > s) acc /= (float)(e - s);
      C[r * N + f] = acc;
    }
  }
  return 0;
}

/* float64 yardstick of orc_spmm_csr_mask_f32 (same role as orc_spmm_sum_f64). absval: sum of |val*G| over the
 * entries that pass the mask (condition scale). */
int orc_spmm_mask_f64(int absval, int64_t Mout, int64_t N, const int32_t *ptr, const int32_t *idx, const float *val,
                      const float *G, const int32_t *E, double *out) {
  for (int64_t j = 0; j < Mout; j++)
    for (int64_t f = 0; f < N; f++) {
      double res = 0.0;
      for (int64_t p = ptr[j]; p < ptr[j + 1]; p++) {
        const int64_t i = idx[p];
        if (E[i * N + f] == (int32_t)j) {
          const double t = (double)(val ? val[p] : 1.0f) * (double)G[i * N + f];
          res += absval ? fabs(t) : t;
        }
      }
      out[j * N + f] = res;
    }
  return 0;
}
