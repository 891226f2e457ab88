/*
 * pcx_oracle.c -- CPU restatement of PyChebyshev's batched-evaluation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (pychebyshev_amd/) may link,
 * load or call this file.  Allowed users: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, always as the checker / the thing timed beside the
 * GPU, never as the path being shipped.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle.py
 * against golden vectors produced by importing the reference (v0.21.1) in the build
 * container (tests/golden/generate_golden.py) and against the reference's own
 * .pcb fixtures through oracle/_ref/reader (the reference's C reader compiled from
 * /root/reference/examples/binary_reader/reader.c).
 *
 * Plain C99 + libm (+ OpenMP over the point loop only).  Each function cites the
 * reference file:line whose arithmetic (and order of operations) it follows.
 * Paths are relative to /root/reference/src/pychebyshev/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PCXO_MAX_DIMS 32

/* ------------------------------------------------------------------ */
/* 1-D primitives                                                      */
/* ------------------------------------------------------------------ */

/* numpy.polynomial.chebyshev.chebpts1 (third-party, NumPy >= 2.0):
 *   x = 0.5*pi/n * arange(-n+1, n+1, 2);  return sin(x)
 * followed by the affine map + sort of barycentric.py:440-452 and
 * _extrude_slice.py:66-70.  The sin-form is already ascending; the sort is
 * kept so that lo > hi inputs behave as in the reference. */
static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

void pcxo_nodes(double lo, double hi, int n, double *out) {
    for (int k = 0; k < n; k++) {
        double ang = 0.5 * M_PI / (double)n * (double)(-n + 1 + 2 * k);
        out[k] = 0.5 * (lo + hi) + 0.5 * (hi - lo) * sin(ang);
    }
    qsort(out, (size_t)n, sizeof(double), cmp_double);
}

/* barycentric.py:30-49 -- w_i starts at 1 and is divided by (x_i - x_j) for
 * j ascending, j != i (a division chain, not 1/product). */
void pcxo_bary_weights(const double *x, int n, double *w) {
    for (int i = 0; i < n; i++) {
        double wi = 1.0;
        for (int j = 0; j < n; j++)
            if (j != i) wi /= (x[i] - x[j]);
        w[i] = wi;
    }
}

/* barycentric.py:52-77 -- c_ij = w_j / ((x_i - x_j) * w_i), diagonal = -(row sum
 * of the off-diagonal entries, summed j ascending as numpy's pairwise sum does
 * for n < 128... numpy's add.reduce over a contiguous row of < 8 elements is a
 * plain loop; for 8 <= n < 128 it is an 8-way unrolled pairwise block.  The
 * difference is at the 1e-16 level and the golden test carries a 4-ulp budget. */
void pcxo_diffmat(const double *x, const double *w, int n, double *D) {
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            if (i == j) { D[i * n + j] = 0.0; continue; }
            double c = x[i] - x[j];
            D[i * n + j] = w[j] / (c * w[i]);
        }
    }
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += D[i * n + j];
        D[i * n + i] = -s;
    }
}

/* ------------------------------------------------------------------ */
/* Barycentric tensor path                                             */
/* ------------------------------------------------------------------ */

/* barycentric.py:951-990 (_apply_derivative_passes): for d = D-1..0, repeat
 * order[d] times:  A'[.., i, ..] = sum_j A[.., j, ..] * D_d[i, j]   (arr @ D_d.T
 * along axis d).  `tensor` is C-order; result written to `out` (may not alias). */
static void mode_product(const double *in, double *out, const int *n, int d, int axis,
                         const double *Dm) {
    long outer = 1, inner = 1;
    for (int k = 0; k < axis; k++) outer *= n[k];
    for (int k = axis + 1; k < d; k++) inner *= n[k];
    int na = n[axis];
    for (long o = 0; o < outer; o++)
        for (int i = 0; i < na; i++)
            for (long q = 0; q < inner; q++) {
                double s = 0.0;
                for (int j = 0; j < na; j++) /* k-ordered fma chain, as BLAS dgemm kernels do */
                    s = fma(in[(o * na + j) * inner + q], Dm[i * na + j], s);
                out[(o * na + i) * inner + q] = s;
            }
}

/* Returns a freshly malloc'ed tensor with all passes applied (caller frees). */
double *pcxo_apply_derivative_passes(int d, const int *n, const double *diff_cat,
                                     const double *tensor, const int *order) {
    long total = 1;
    for (int k = 0; k < d; k++) total *= n[k];
    double *cur = (double *)malloc(sizeof(double) * (size_t)total);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)total);
    memcpy(cur, tensor, sizeof(double) * (size_t)total);
    long doff[PCXO_MAX_DIMS];
    long acc = 0;
    for (int k = 0; k < d; k++) { doff[k] = acc; acc += (long)n[k] * n[k]; }
    for (int k = d - 1; k >= 0; k--) {
        int ord = order ? order[k] : 0;
        for (int r = 0; r < ord; r++) {
            mode_product(cur, tmp, n, d, k, diff_cat + doff[k]);
            double *sw = cur; cur = tmp; tmp = sw;
        }
    }
    free(tmp);
    return cur;
}

/* One point of barycentric.py:1035-1046: dims last -> first; exact-node test
 * |x - node| < 1e-14 picks the FIRST matching index; otherwise
 * current = (current @ (w/diff)) / sum(w/diff).  `work` holds >= total/n_last doubles. */
static double bary_point(int d, const int *n, const long *noff, const double *nodes_cat,
                         const double *weights_cat, const double *tensor, const double *x,
                         double *work) {
    long rows = 1;
    for (int k = 0; k < d; k++) rows *= n[k];
    const double *cur = tensor;
    double u[4096];
    for (int k = d - 1; k >= 0; k--) {
        int nk = n[k];
        rows /= nk;
        const double *nd = nodes_cat + noff[k];
        const double *wd = weights_cat + noff[k];
        int exact = -1;
        for (int j = 0; j < nk; j++) {
            double diff = x[k] - nd[j];
            if (fabs(diff) < 1e-14) { exact = j; break; }
        }
        if (exact >= 0) {
            for (long r = 0; r < rows; r++) work[r] = cur[r * nk + exact];
        } else {
            double su = 0.0;
            for (int j = 0; j < nk; j++) { u[j] = wd[j] / (x[k] - nd[j]); }
            for (int j = 0; j < nk; j++) su += u[j];
            for (long r = 0; r < rows; r++) {
                double s = 0.0;
                const double *row = cur + r * nk;
                for (int j = 0; j < nk; j++) s = fma(row[j], u[j], s);
                work[r] = s / su;
            }
        }
        cur = work; /* in-place is safe: work[r] is written after row r (>= r) is read */
    }
    return work[0];
}

/* barycentric.py:992-1047 (vectorized_eval_batch).  pts is (N, d) row-major.
 * order may be NULL (all zeros).  Returns 0, or -1 on bad arguments. */
int pcxo_bary_eval_batch(int d, const int *n, const double *nodes_cat,
                         const double *weights_cat, const double *diff_cat,
                         const double *tensor, const double *pts, long N,
                         const int *order, double *out) {
    if (d < 1 || d > PCXO_MAX_DIMS) return -1;
    long noff[PCXO_MAX_DIMS];
    long acc = 0, total = 1;
    for (int k = 0; k < d; k++) {
        if (n[k] < 1 || n[k] > 4096) return -1;
        noff[k] = acc; acc += n[k]; total *= n[k];
    }
    double *T = pcxo_apply_derivative_passes(d, n, diff_cat, tensor, order);
    long wsize = total / n[d - 1];
    if (wsize < 1) wsize = 1;
#pragma omp parallel
    {
        double *work = (double *)malloc(sizeof(double) * (size_t)wsize);
#pragma omp for schedule(static)
        for (long i = 0; i < N; i++)
            out[i] = bary_point(d, n, noff, nodes_cat, weights_cat, T, pts + i * d, work);
        free(work);
    }
    free(T);
    return 0;
}

/* barycentric.py:1049-1112 (vectorized_eval_multi): normalised weights shared by
 * all m specs; exact test via argmin(|diff|) < 1e-14; per spec the derivative
 * matmuls are interleaved with the contraction on the already-reduced tensor. */
int pcxo_bary_eval_multi(int d, const int *n, const double *nodes_cat,
                         const double *weights_cat, const double *diff_cat,
                         const double *tensor, const double *x, const int *orders, int m,
                         double *out) {
    if (d < 1 || d > PCXO_MAX_DIMS) return -1;
    long noff[PCXO_MAX_DIMS], doff[PCXO_MAX_DIMS];
    long acc = 0, dacc = 0, total = 1;
    for (int k = 0; k < d; k++) {
        noff[k] = acc; acc += n[k];
        doff[k] = dacc; dacc += (long)n[k] * n[k];
        total *= n[k];
    }
    double *wn = (double *)malloc(sizeof(double) * (size_t)acc);
    int exact[PCXO_MAX_DIMS];
    for (int k = 0; k < d; k++) {
        const double *nd = nodes_cat + noff[k];
        const double *wd = weights_cat + noff[k];
        int amin = 0;
        double best = fabs(x[k] - nd[0]);
        for (int j = 1; j < n[k]; j++) {
            double a = fabs(x[k] - nd[j]);
            if (a < best) { best = a; amin = j; }
        }
        if (best < 1e-14) { exact[k] = amin; continue; }
        exact[k] = -1;
        double su = 0.0;
        for (int j = 0; j < n[k]; j++) { wn[noff[k] + j] = wd[j] / (x[k] - nd[j]); }
        for (int j = 0; j < n[k]; j++) su += wn[noff[k] + j];
        for (int j = 0; j < n[k]; j++) wn[noff[k] + j] /= su;
    }
    double *a = (double *)malloc(sizeof(double) * (size_t)total);
    double *b = (double *)malloc(sizeof(double) * (size_t)total);
    for (int s = 0; s < m; s++) {
        const int *ord = orders + (long)s * d;
        memcpy(a, tensor, sizeof(double) * (size_t)total);
        long rows = total;
        for (int k = d - 1; k >= 0; k--) {
            int nk = n[k];
            rows /= nk;
            const double *Dm = diff_cat + doff[k];
            for (int r = 0; r < ord[k]; r++) {
                for (long q = 0; q < rows; q++)
                    for (int i = 0; i < nk; i++) {
                        double acc2 = 0.0;
                        for (int j = 0; j < nk; j++) acc2 = fma(a[q * nk + j], Dm[i * nk + j], acc2);
                        b[q * nk + i] = acc2;
                    }
                double *sw = a; a = b; b = sw;
            }
            if (exact[k] >= 0) {
                for (long q = 0; q < rows; q++) b[q] = a[q * nk + exact[k]];
            } else {
                for (long q = 0; q < rows; q++) {
                    double acc2 = 0.0;
                    for (int j = 0; j < nk; j++) acc2 = fma(a[q * nk + j], wn[noff[k] + j], acc2);
                    b[q] = acc2;
                }
            }
            double *sw = a; a = b; b = sw;
        }
        out[s] = a[0];
    }
    free(a); free(b); free(wn);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Tensor-train path                                                   */
/* ------------------------------------------------------------------ */

/* numpy.polynomial.chebyshev.chebval(x, eye(n)) restated per basis vector e_j
 * (Clenshaw recurrence exactly as NumPy writes it):
 *   n == 1: c0 = c[0], c1 = 0;  n == 2: c0 = c[0], c1 = c[1];
 *   else x2 = 2x; c0 = c[-2]; c1 = c[-1];
 *        for i in 3..n: tmp = c0; c0 = c[-i] - c1; c1 = tmp + c1*x2
 *   return c0 + c1*x                                                  */
static void cheb_basis_clenshaw(double x, int n, double *q) {
    for (int j = 0; j < n; j++) {
        double c0, c1;
#define CJ(idx) ((idx) == j ? 1.0 : 0.0)
        if (n == 1) { c0 = CJ(0); c1 = 0.0; }
        else if (n == 2) { c0 = CJ(0); c1 = CJ(1); }
        else {
            double x2 = 2.0 * x;
            c0 = CJ(n - 2); c1 = CJ(n - 1);
            for (int i = 3; i <= n; i++) {
                double tmp = c0;
                c0 = CJ(n - i) - c1;
                c1 = tmp + c1 * x2;
            }
        }
#undef CJ
        q[j] = c0 + c1 * x;
    }
}

/* tensor_train.py:2217-2265 (eval_batch) and :2199-2214 (_eval_storage_frame):
 * per storage dim k: scaled = 2(x-a)/(b-a) - 1; q = T_j(scaled);
 * V[i,k'] = sum_j q_j G[i,j,k'];  v <- v @ V.   dim_order (may be NULL) maps
 * storage position k to the user's column dim_order[k] (tensor_train.py:2246-2248).
 * cores_cat: cores concatenated, each (r_{k-1}, n_k, r_k) C-order.  */
int pcxo_tt_eval_batch(int d, const int *n, const int *ranks, const double *lo,
                       const double *hi, const double *cores_cat, const int *dim_order,
                       const double *pts, long N, double *out) {
    if (d < 1 || d > PCXO_MAX_DIMS) return -1;
    long coff[PCXO_MAX_DIMS];
    long acc = 0;
    int rmax = 1, nmax = 1;
    for (int k = 0; k < d; k++) {
        coff[k] = acc;
        acc += (long)ranks[k] * n[k] * ranks[k + 1];
        if (ranks[k] > rmax) rmax = ranks[k];
        if (ranks[k + 1] > rmax) rmax = ranks[k + 1];
        if (n[k] > nmax) nmax = n[k];
    }
    if (ranks[0] != 1 || ranks[d] != 1) return -1;
#pragma omp parallel
    {
        double *q = (double *)malloc(sizeof(double) * (size_t)nmax);
        double *V = (double *)malloc(sizeof(double) * (size_t)rmax * rmax);
        double *v = (double *)malloc(sizeof(double) * (size_t)rmax);
        double *v2 = (double *)malloc(sizeof(double) * (size_t)rmax);
#pragma omp for schedule(static)
        for (long p = 0; p < N; p++) {
            v[0] = 1.0;
            for (int k = 0; k < d; k++) {
                int col = dim_order ? dim_order[k] : k;
                double x = pts[p * d + col];
                double scaled = 2.0 * (x - lo[k]) / (hi[k] - lo[k]) - 1.0;
                cheb_basis_clenshaw(scaled, n[k], q);
                int rl = ranks[k], rr = ranks[k + 1];
                const double *G = cores_cat + coff[k];
                for (int i = 0; i < rl; i++)
                    for (int c = 0; c < rr; c++) {
                        double s = 0.0;
                        for (int j = 0; j < n[k]; j++) s += q[j] * G[((long)i * n[k] + j) * rr + c];
                        V[i * rr + c] = s;
                    }
                for (int c = 0; c < rr; c++) {
                    double s = 0.0;
                    for (int i = 0; i < rl; i++) s += v[i] * V[i * rr + c];
                    v2[c] = s;
                }
                double *sw = v; v = v2; v2 = sw;
            }
            out[p] = v[0];
        }
        free(q); free(V); free(v); free(v2);
    }
    return 0;
}

/* tensor_train.py:997-1016 (_value_core_to_coeff_core):
 *   coeff = dct(core[:, ::-1, :], type=2, axis=1) / n ; coeff[:, 0, :] /= 2
 * with SciPy's backward-normalised DCT-II  y_k = 2 sum_j x_j cos(pi k (2j+1)/(2n)). */
void pcxo_value_to_coeff_core(const double *value_core, int rl, int n, int rr,
                              double *coeff_core) {
    for (int i = 0; i < rl; i++)
        for (int c = 0; c < rr; c++)
            for (int k = 0; k < n; k++) {
                double s = 0.0;
                for (int j = 0; j < n; j++) {
                    double xj = value_core[((long)i * n + (n - 1 - j)) * rr + c];
                    s += xj * cos(M_PI * (double)k * (double)(2 * j + 1) / (double)(2 * n));
                }
                s = 2.0 * s / (double)n;
                if (k == 0) s /= 2.0;
                coeff_core[((long)i * n + k) * rr + c] = s;
            }
}

/* tensor_train.py:223-228 (_eval_tt): TT value at integer grid indices through a
 * chain of (1 x r) @ (r x r') products on VALUE cores. */
double pcxo_tt_eval_grid(int d, const int *n, const int *ranks, const double *cores_cat,
                         const int *idx) {
    double v[1024], v2[1024];
    long off = 0;
    v[0] = 1.0;
    for (int k = 0; k < d; k++) {
        int rl = ranks[k], rr = ranks[k + 1];
        const double *G = cores_cat + off;
        for (int c = 0; c < rr; c++) {
            double s = 0.0;
            for (int i = 0; i < rl; i++) s += v[i] * G[((long)i * n[k] + idx[k]) * rr + c];
            v2[c] = s;
        }
        memcpy(v, v2, sizeof(double) * (size_t)rr);
        off += (long)rl * n[k] * rr;
    }
    return v[0];
}

void pcxo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int pcxo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
