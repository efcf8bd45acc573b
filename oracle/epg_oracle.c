/*
 * CPU oracle (plain C99) for the epgpy hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library (through oracle/epg_c.py).  The product library libepgx.so never links it.
 *
 * Restates, voxel by voxel, what the reference (py-baudin/epgpy) computes with NumPy
 * ufuncs over the whole grid inside simulate_simple()  (epgpy/functions.py:173-192):
 *
 *   state of one voxel : rows r = 0..2n (order k = r - n), 3 complex128 columns
 *                        col0 = F_k, col1 = conj(F_-k), col2 = Z_k    (statematrix.py:55,:392)
 *   MAT  (T, Phi...)   : row <- M(3x3) * row for every row            (opmatrix.py:208-221)
 *   SCAL (E, P, R)     : row <- arr(3) .* row ;  row_n(Z) += arr0[2]*density
 *                        (arr0*equilibrium, equilibrium = [0,0,density] at the centre row only)
 *                                                                     (opscalar.py:213-232,
 *                                                                      statematrix.py:379-385)
 *   SHIFT k            : n <- min(n+|k|, nmax) (symmetric zero pad), then
 *                        k>0: col0[r] <- col0[r-k], col1[r] <- col1[r+k], zero fill
 *                        k<0: mirrored                                 (shift.py:82-101,:271-294)
 *   ADC                : F0 = col0[n] (or Z0 = col2[n])                (statematrix.py:148-175)
 *   SPOIL              : col0, col1 <- 0                               (operator.py:281-286)
 *
 * Coefficients (the 3x3 matrices and the arr/arr0 triplets) are NOT computed here: the
 * caller passes them exactly as the reference's operator constructors build them on the
 * host (transition.py:114-151, evolution.py:220-256) -- oracle/epg_numpy.py restates those.
 *
 * The only liberty taken w.r.t. the reference is storage: rows live in a fixed buffer
 * centred at index NCAP so that growing n needs no re-allocation; arithmetic per row is
 * the same, in the same order.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o oracle/libepgoracle.so oracle/epg_oracle.c -lm
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

enum { EPGO_NOP = 0, EPGO_MAT = 1, EPGO_SCAL = 2, EPGO_SHIFT = 3, EPGO_ADC_F0 = 4,
       EPGO_ADC_Z0 = 5, EPGO_SPOIL = 6, EPGO_RESET = 7, EPGO_PD = 8 };

typedef struct {
    int32_t kind;       /* EPGO_*                                             */
    int32_t k;          /* SHIFT: signed integer shift; PD: reset flag        */
    const double *coef; /* MAT: 9 complex (row-major); SCAL: arr[3], arr0[3] complex; PD: 1 real */
    int64_t stride;     /* doubles between consecutive voxels (0 = same for all voxels) */
} epgo_op;

int epgo_abi_version(void) { return 1; }

int epgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* number of rows (2n+1) the state matrix has after the whole sequence */
static int final_nstate(const epgo_op *ops, int n_ops, int n0, int nmax) {
    int n = n0;
    for (int i = 0; i < n_ops; ++i) {
        if (ops[i].kind == EPGO_SHIFT) {
            n += abs(ops[i].k);
            if (nmax > 0 && n > nmax) n = nmax;
        } else if (ops[i].kind == EPGO_RESET) {
            n = 0;
        }
    }
    return n;
}

static int peak_nstate(const epgo_op *ops, int n_ops, int n0, int nmax) {
    int n = n0, peak = n0;
    for (int i = 0; i < n_ops; ++i) {
        if (ops[i].kind == EPGO_SHIFT) {
            n += abs(ops[i].k);
            if (nmax > 0 && n > nmax) n = nmax;
            if (n > peak) peak = n;
        } else if (ops[i].kind == EPGO_RESET) {
            n = 0;
        }
    }
    return peak;
}

int epgo_final_nstate(const epgo_op *ops, int n_ops, int n0, int nmax) {
    return final_nstate(ops, n_ops, n0, nmax);
}

static void run_voxel(const epgo_op *ops, int n_ops, int64_t v, int nmax, int n0, int ncap,
                      const double *init, double density, cplx *buf, double *signal,
                      int64_t nvox, double *states_out, int nfinal) {
    /* buf: (2*ncap+1) rows x 3, order k lives at row ncap+k */
    const int nrow = 2 * ncap + 1;
    memset(buf, 0, sizeof(cplx) * 3 * (size_t)nrow);
    int n = n0;
    if (init) {
        const cplx *src = (const cplx *)init + (size_t)v * 3 * (2 * n0 + 1);
        memcpy(buf + 3 * (ncap - n0), src, sizeof(cplx) * 3 * (2 * n0 + 1));
    } else {
        buf[3 * ncap + 2] = density;
    }
    int64_t iadc = 0;
    for (int i = 0; i < n_ops; ++i) {
        const epgo_op *op = &ops[i];
        const cplx *c = (const cplx *)(op->coef ? op->coef + op->stride * v : NULL);
        switch (op->kind) {
        case EPGO_MAT:
            for (int r = ncap - n; r <= ncap + n; ++r) {
                cplx *row = buf + 3 * r;
                cplx a = row[0], b = row[1], z = row[2];
                row[0] = c[0] * a + c[1] * b + c[2] * z;
                row[1] = c[3] * a + c[4] * b + c[5] * z;
                row[2] = c[6] * a + c[7] * b + c[8] * z;
            }
            break;
        case EPGO_SCAL:
            for (int r = ncap - n; r <= ncap + n; ++r) {
                cplx *row = buf + 3 * r;
                row[0] *= c[0];
                row[1] *= c[1];
                row[2] *= c[2];
            }
            /* + arr0 * equilibrium: equilibrium is [0, 0, density] on the centre row */
            buf[3 * ncap + 2] += c[5] * density;
            break;
        case EPGO_SHIFT: {
            int k = op->k, ak = abs(k);
            int n2 = n + ak;
            if (nmax > 0 && n2 > nmax) n2 = nmax;
            n = n2; /* rows outside the old range are already zero */
            int lo = ncap - n, hi = ncap + n; /* inclusive */
            if (k > 0) {
                for (int r = hi; r >= lo + k; --r) buf[3 * r + 0] = buf[3 * (r - k) + 0];
                for (int r = lo; r <= hi - k; ++r) buf[3 * r + 1] = buf[3 * (r + k) + 1];
                for (int r = lo; r < lo + k && r <= hi; ++r) buf[3 * r + 0] = 0;
                for (int r = hi; r > hi - k && r >= lo; --r) buf[3 * r + 1] = 0;
            } else if (k < 0) {
                for (int r = lo; r <= hi - ak; ++r) buf[3 * r + 0] = buf[3 * (r + ak) + 0];
                for (int r = hi; r >= lo + ak; --r) buf[3 * r + 1] = buf[3 * (r - ak) + 1];
                for (int r = hi; r > hi - ak && r >= lo; --r) buf[3 * r + 0] = 0;
                for (int r = lo; r < lo + ak && r <= hi; ++r) buf[3 * r + 1] = 0;
            }
            break;
        }
        case EPGO_ADC_F0:
        case EPGO_ADC_Z0: {
            cplx val = buf[3 * ncap + (op->kind == EPGO_ADC_F0 ? 0 : 2)];
            signal[2 * (iadc * nvox + v) + 0] = creal(val);
            signal[2 * (iadc * nvox + v) + 1] = cimag(val);
            ++iadc;
            break;
        }
        case EPGO_SPOIL:
            for (int r = ncap - n; r <= ncap + n; ++r) {
                buf[3 * r + 0] = 0;
                buf[3 * r + 1] = 0;
            }
            break;
        case EPGO_RESET:
            memset(buf, 0, sizeof(cplx) * 3 * (size_t)nrow);
            buf[3 * ncap + 2] = density;
            n = 0;
            break;
        case EPGO_PD:
            density = op->coef[op->stride * v];
            if (op->k) { /* reset to the new equilibrium, keep n */
                memset(buf, 0, sizeof(cplx) * 3 * (size_t)nrow);
                buf[3 * ncap + 2] = density;
            }
            break;
        default:
            break;
        }
    }
    if (states_out) {
        cplx *dst = (cplx *)states_out + (size_t)v * 3 * (2 * nfinal + 1);
        memcpy(dst, buf + 3 * (ncap - nfinal), sizeof(cplx) * 3 * (2 * nfinal + 1));
    }
}

/*
 * signal     : [n_adc][nvox] complex128 (interleaved re,im), caller allocated
 * states_out : nullable, [nvox][2*nfinal+1][3] complex128 with nfinal = epgo_final_nstate(...)
 * init       : nullable, [nvox][2*n0+1][3] complex128 initial state (else equilibrium, n0 = 0)
 * density    : nullable, [nvox] equilibrium magnetisation (default 1)
 * returns 0, or -1 on allocation failure
 */
int epgo_simulate(const epgo_op *ops, int n_ops, int64_t nvox, int max_nstate, int n0,
                  const double *init, const double *density, double *signal,
                  double *states_out, int nthreads) {
    const int nmax = max_nstate > 0 ? max_nstate : 0;
    const int ncap = peak_nstate(ops, n_ops, n0, nmax);
    const int nfinal = final_nstate(ops, n_ops, n0, nmax);
    int fail = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        cplx *buf = (cplx *)malloc(sizeof(cplx) * 3 * (size_t)(2 * ncap + 1));
        if (!buf) {
            fail = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
            for (int64_t v = 0; v < nvox; ++v)
                run_voxel(ops, n_ops, v, nmax, n0, ncap, init, density ? density[v] : 1.0, buf,
                          signal, nvox, states_out, nfinal);
            free(buf);
        }
    }
    return fail ? -1 : 0;
}
