/*
 * epgx.h -- C ABI of libepgx.so, the MI355X (gfx950) engine behind epgpy's hot path.
 *
 * The reference (py-baudin/epgpy) has no FFI: its only back-end seam is the array-module
 * switch `set_array_module("cupy")` (epgpy/common.py:21-74).  The boundary this library
 * replaces is therefore the *operator protocol* + `simulate()`:
 *
 *   reference interface (file:line)                     entry point(s) here
 *   --------------------------------------------------  -----------------------------------
 *   functions.simulate / simulate_simple loop           epgx_plan_create + epgx_run
 *     (epgpy/functions.py:50-192)                         (whole sequence or one segment per launch)
 *   DiffOperator.__call__ -> _apply for T/E/P/S         epgx_run on a 1-operator plan
 *     (epgpy/diff.py:119-139, opmatrix.py:199-221,
 *      opscalar.py:213-232, shift.py:82-101,271-294)
 *   StateMatrix storage: states/equilibrium/resize/copy epgx_state_create/upload/download/
 *     (epgpy/statematrix.py:12-80,276-297,654-676)        copy/resize/destroy
 *   Probe.acquire -> asnumpy(F0)  (probe.py:63-66,       signal buffer written by EPGX_OP_ADC,
 *     138-139; statematrix.py:148-175)                    fetched with epgx_memcpy_d2h
 *   cupy device management (common.py:37-47)             epgx_ctx_*, epgx_malloc/free/memcpy
 *
 * Conventions: every function returns EPGX_OK (0) or a negative error code and never throws;
 * epgx_last_error() returns a thread-local message for the last failure on this thread.
 * The caller owns all host buffers; the library owns device buffers it allocates.  All work
 * is enqueued on the context's HIP stream (own stream, or one adopted with
 * epgx_ctx_set_stream); host copies synchronise that stream.  One context per (thread, device).
 * There is NO CPU fallback: with no usable GPU epgx_ctx_create fails with EPGX_ERR_NODEVICE.
 *
 * Data layouts (all complex128 = 2 x float64 interleaved):
 *   state  : [nvox][3][K]  half representation, order k = 0..K-1 contiguous,
 *            comp 0 = F_k, comp 1 = conj(F_-k), comp 2 = Z_k   (reference: [*grid, 2n+1, 3],
 *            statematrix.py:55; the k<0 rows are the mirror image, statematrix.py:416-421)
 *   density: [nvox] float64, equilibrium magnetisation (statematrix.py:379-385)
 *   signal : [n_adc][signal_ld] complex128, slot-major ("(n_adc, *grid)", functions.py:157-165); entry points that hand
 *            records to the HOST take an epgx_signal_dtype: EPGX_SIGNAL_C64 rounds every record once, on the device, to
 *            2 x float32 before it crosses PCIe (the arithmetic and the device-side records stay complex128)
 *
 * The operator stream is given as primitives (one epgx_op per reference operator); the library
 * packs them into fused [T][E][S][ADC] records internally ("S E" is emitted as "E S", which is
 * bit-identical because E multiplies every order by the same factors and S only moves values).
 */
#ifndef EPGX_H
#define EPGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPGX_ABI_VERSION 6
#define EPGX_MAX_DIMS 8    /* grid dimensions                        */
#define EPGX_MAX_SPACES 4  /* distinct operator broadcast patterns   */
#define EPGX_WAVE 64       /* k-states per lane-register (wave64)    */
#define EPGX_MAX_K 2048    /* max k-states per voxel: 1024 for states in HBM (16 per lane), 2048 state-resident */
#define EPGX_MAX_VARS 3    /* derivative variables carried per launch */

enum epgx_status {
    EPGX_OK = 0,
    EPGX_ERR_INVALID = -1,     /* bad argument / inconsistent plan            */
    EPGX_ERR_HIP = -2,         /* a HIP runtime call failed                   */
    EPGX_ERR_NOMEM = -3,       /* device or host allocation failed            */
    EPGX_ERR_UNSUPPORTED = -4, /* valid request the kernels do not cover      */
    EPGX_ERR_NODEVICE = -5     /* no MI355X-class device visible              */
};

/* Operator stream.  Coefficients live in one float64 pool; an operator with coefficients
 * reads `ncoef` doubles at  pool[coef_off + index(space, voxel) * ncoef]. */
enum epgx_opcode {
    EPGX_OP_NOP = 0,
    EPGX_OP_T = 1,     /* RF rotation, 8 coef: m00, Re/Im m01, Re/Im m02, Re/Im m20, m22
                          (m00, m22 real; m10=conj m01, m11=m00, m12=conj m02, m21=conj m20)
                          -- transition.py:114-151                                           */
    EPGX_OP_MAT = 2,   /* general symmetric 3x3, 9 coef: Re/Im m00, Re/Im m01, Re/Im m02,
                          Re/Im m20, m22 (m11 = conj m00) -- opmatrix.py:157-170             */
    EPGX_OP_E = 3,     /* diagonal, 4 coef: Re/Im e0 (multiplies F_k; conj(e0) multiplies
                          col 1), e2 (Z), r0 (Z_0 += r0*density) -- evolution.py:220-256      */
    EPGX_OP_S = 4,     /* integer shift by ia (!= 0) with truncation at K -- shift.py:271-294 */
    EPGX_OP_ADC = 5,   /* signal[ia] <- F0 (ib = 0) or Z0 (ib = 1) -- statematrix.py:148-175  */
    EPGX_OP_SPOIL = 6, /* F <- 0 -- operator.py:281-286                                      */
    EPGX_OP_RESET = 7, /* state <- equilibrium -- operator.py:297-304                        */
    EPGX_OP_PD = 8,    /* density <- coef[0]; ia != 0: state <- new equilibrium
                          -- operator.py:315-341                                             */
    EPGX_OP_D = 9,     /* per-order real diagonal, ncoef = 3*K doubles per entry laid out [3][K]:
                          F_k *= c[0][k], conj(F_-k) *= c[1][k], Z_k *= c[2][k]
                          -- diffusion.py:60-79 (D operator; the table is built on the host from the
                          k-space coordinates and the diffusion coefficient / tensor)             */
    EPGX_OP_GS = 10,   /* host-planned gather shift, ncoef = 3*K/2 doubles = int32 [3][K]: for each
                          new order the old order its F / conj(F-) / Z comes from; -1 = zero,
                          index | 1<<30 = conjugate of the partner array -- shift.py:297-364 (shiftnd) */
    EPGX_OP_MAT0 = 11, /* EPGX_OP_MAT plus a constant term, 14 coef: the 10 of MAT, then Re/Im o0, o2, pad:
                          (o0, conj o0, o2) * density is added to (F_0, conj F_0, Z_0) -- the effect of
                          `mat0 @ equilibrium` (opmatrix.py:199-205); produced by `E @ T` combinations  */
    EPGX_OP_T0 = 12,   /* EPGX_OP_T plus a constant term, 12 coef: the 8 of T, then Re/Im o0, o2, pad.  What a
                          T between two precession-free relaxations collapses to: E2 T E1 keeps T's
                          symmetry and real m00 (the diagonal of E is real), and the recoveries become
                          (o0, conj o0, o2) * density on the k = 0 order.  Emitted by the host's
                          peephole fusion (epgpy_amd/fusion.py); same idea as `E @ T` in the reference */
    EPGX_OP__COUNT
};

typedef struct epgx_op {
    int32_t opcode;   /* enum epgx_opcode                                  */
    int32_t space;    /* index space of the coefficient table, or -1       */
    int32_t ia;       /* S: shift; ADC: output slot; PD: reset flag        */
    int32_t ib;       /* ADC: 0 = F0, 1 = Z0                               */
    int64_t coef_off; /* first double of this operator's table in the pool */
    int32_t ncoef;    /* doubles per table entry                           */
    int32_t reserved;
} epgx_op; /* 32 bytes */

/* First-order partial derivatives of operator i w.r.t. up to EPGX_MAX_VARS variables
 * (DiffOperator order1, epgpy/diff.py:264-288).  coef_off[v] < 0: operator i does not depend on
 * variable v.  Table entry: EPGX_OP_T / MAT / MAT0 -> the 10 coefficients of EPGX_OP_MAT for
 * d(mat)/dv;  EPGX_OP_E -> 4 coefficients (Re/Im d e0, d e2, d r0) for d(arr, arr0)/dv; already
 * combined over the operator's parameters (sum_p coeff[v][p] * dOp/dp).  An EPGX_OP_T0 whose table the library
 * generated (epgx_fuse) takes its partials from tables the library generates as well (epgx_fuse_partial: coef_off in
 * the generated part of the pool, 14 per entry -- the 10 of d(mat)/dv, then Re/Im d o0, d o2, pad).  A partial table may
 * also be one the library ASSEMBLES from per-axis columns (epgx_assemble: d(relaxation)/dT2 over a (T1, T2) grid varies with
 * T2 only): coef_off then names the recipe's destination. */
typedef struct epgx_dop {
    int32_t space[EPGX_MAX_VARS]; /* index space of the partial's table, or -1 */
    int32_t reserved;
    int64_t coef_off[EPGX_MAX_VARS];
} epgx_dop; /* 40 bytes */

/* A table that the LIBRARY generates on the device when the plan is created: the EPGX_OP_T0 table
 * of "T sandwiched with a precession-free E" (see EPGX_OP_T0).  Sources are tables of the pool (host
 * part, or tables generated by earlier entries of the list); the destination lies in the generated
 * part of the pool, [n_coef, n_coef + n_coef_generated).  Every index space that a source varies
 * along must be one the destination varies along.  (The host would need ~0.1 s of NumPy per
 * 1024 x 1024 table; the device writes it in ~0.1 ms.) */
typedef struct epgx_fuse {
    int64_t dst_off;   /* doubles, in the generated part; 12 coefficients per entry (EPGX_OP_T0 layout) */
    int64_t src_off;   /* the rotation: EPGX_OP_T layout (8 per entry) or EPGX_OP_T0 layout (12)          */
    int64_t e_off;     /* the relaxation: EPGX_OP_E layout, Im e0 = 0 in every entry                       */
    int32_t dst_space, src_space, e_space; /* index spaces (-1: one entry for all voxels)                */
    int32_t src_ncoef; /* 8 or 12                                                                          */
    int32_t after;     /* 1: E acts after the rotation (rows scaled), 0: before it (columns scaled)        */
    int32_t reserved;
} epgx_fuse; /* 48 bytes */

/* The partial derivative of an epgx_fuse table with respect to one variable, generated next to it (product rule:
 * d(E T) = dE T + E dT, the constant term likewise).  A differentiated  E . T . E  sandwich then costs the kernels one
 * rotation per state and one accumulation per derivative state instead of three stages each (the 20-echo 1024 x 1024
 * Jacobian with one variable: 66 -> 46 fp64 instructions per order and echo).  Sources: the operands of the epgx_fuse
 * entry that produced the value table, and their partials -- a rotation partial in the host part of the pool (10 per
 * entry, epgx_dop layout), or one generated by an EARLIER entry of this list (14 per entry), or none (-1); a
 * relaxation partial (4 per entry, Im d e0 = 0 in every entry; in the host part of the pool or an assembled table) or none.
 * At least one of the two must be given.
 * Executed after the `fuse` list, in order. */
typedef struct epgx_fuse_partial {
    int64_t dst_off;   /* doubles, in the generated part; 14 per entry: d(mat)/dv (10), Re/Im d o0, d o2, pad      */
    int64_t src_off;   /* the rotation's VALUE table, as in the epgx_fuse entry (8 or 12 per entry)                  */
    int64_t dsrc_off;  /* its partial, or -1                                                                          */
    int64_t e_off;     /* the relaxation's VALUE table (4 per entry)                                                  */
    int64_t de_off;    /* its partial, or -1                                                                          */
    int32_t dst_space, src_space, dsrc_space, e_space, de_space;   /* index spaces (-1: one entry for all voxels)   */
    int32_t src_ncoef; /* 8 or 12                                                                                     */
    int32_t dsrc_ncoef;/* 10 or 14 (ignored when dsrc_off < 0)                                                        */
    int32_t after;     /* as in epgx_fuse                                                                              */
} epgx_fuse_partial; /* 72 bytes */

/* A table that the library ASSEMBLES on the device from per-axis columns when the plan is created.  The
 * reference builds e.g. the relaxation table of E(tau, T1[:, None], T2[None, :]) by evaluating exp() on the
 * un-broadcast parameter arrays and broadcasting the results into [*grid, 3] (evolution.py:220-242): the table is an
 * outer combination of a few small columns.  Shipping the columns (2 x 1024 x 2 doubles) and letting the device write
 * the 1024 x 1024 x 4 table (34 MB in ~10 us) replaces packing and uploading the full table on every simulate()
 * (~5 ms) -- with the SAME bits, because the column values are the reference's own.
 *   destination column c of entry i  =  coef[src[col_src[c]].off + index_i(src[col_src[c]].strides) * ncol + col_idx[c]]
 * Sources lie in the host part of the pool; the destination in the generated part (like epgx_fuse, whose sources
 * may be assembled tables: the list is executed before the fuse list).  Every axis a source varies along must be
 * one the destination's index space varies along. */
#define EPGX_MAX_ASM_SRC 4
#define EPGX_MAX_ASM_COLS 16
typedef struct epgx_asm_src {
    int64_t off;                     /* first double of the source columns: [entries][ncol]        */
    int32_t ncol, reserved;
    int64_t strides[EPGX_MAX_DIMS];  /* entry index = sum_d coord[d] * strides[d]                   */
} epgx_asm_src; /* 80 bytes */
typedef struct epgx_assemble {
    int64_t dst_off;                 /* doubles, in the generated part; ncoef per entry             */
    int32_t dst_space, ncoef;        /* ncoef <= EPGX_MAX_ASM_COLS                                  */
    int32_t n_src, reserved;
    epgx_asm_src src[EPGX_MAX_ASM_SRC];
    uint8_t col_src[EPGX_MAX_ASM_COLS], col_idx[EPGX_MAX_ASM_COLS];
} epgx_assemble; /* 376 bytes */

/* Host-side description of a compiled sequence ("plan").  The parameter grid has `ndim`
 * axes of extent grid_shape[d] (C order, last axis fastest); voxel v has coordinates
 * unravel(v).  Index space s maps a voxel to  sum_d coord[d] * space_strides[s][d]
 * (stride 0 on axes the operator does not depend on: the reference's append-trailing-axes
 * broadcasting, common.py:273-303).  Set struct_size = sizeof(epgx_plan_desc) and zero every member you do not use. */
typedef struct epgx_plan_desc {
    uint32_t struct_size;         /* sizeof(epgx_plan_desc) of the header the CALLER was built against: epgx_plan_create
                                     rejects any other value with EPGX_ERR_INVALID instead of reading past a shorter struct */
    int32_t n_ops;
    const epgx_op *ops;
    int32_t ndim;
    const int64_t *grid_shape;    /* [ndim]                       */
    int32_t n_spaces;
    const int64_t *space_strides; /* [n_spaces][EPGX_MAX_DIMS]    */
    int64_t n_coef;
    const double *coef;           /* [n_coef]                     */
    int32_t n_adc;                /* number of signal slots (rows of the signal buffer)           */
    int32_t n_vars;               /* 0, or 1..EPGX_MAX_VARS: run with first-order derivatives; every
                                     ADC then writes 1 + n_vars consecutive rows starting at its
                                     slot: the probe of the state, then of each derivative state      */
    const epgx_dop *dops;         /* [n_ops] when n_vars > 0, else NULL                               */
    int32_t deriv_flags;          /* EPGX_DERIV_*                                                      */
    int32_t n_fuse;               /* device-generated tables                                           */
    const epgx_fuse *fuse;        /* [n_fuse], executed in order                                        */
    int64_t n_coef_generated;     /* doubles appended to the pool for them (operators may refer to
                                     offsets up to n_coef + n_coef_generated)                          */
    int32_t n_assemble;           /* device-assembled tables (executed before `fuse`)                  */
    int32_t n_fuse_partial;       /* partials of device-generated tables (executed after `fuse`)       */
    const epgx_assemble *assemble;/* [n_assemble]                                                      */
    const epgx_fuse_partial *fuse_partial; /* [n_fuse_partial]                                         */
} epgx_plan_desc;

/* The reference propagates derivative states through its DiffOperators only (T/MAT, E, S);
 * SPOIL, RESET, PD and D are plain Operators there and leave sm.order1 untouched
 * (epgpy/operator.py:95-104 vs epgpy/diff.py:119-139), e.g. the "derivative" of a spoiled signal
 * stays non-zero.  Default (flag clear): the same for SPOIL and D, for identical numbers.  Flag
 * set: these operators, which are linear and independent of the variables, act on the derivative
 * states too (d(Op S)/dv = Op dS/dv), which is the derivative of the signal that is actually
 * simulated.  A reset (RESET, PD with reset) always clears the derivative states: the reference
 * keeps stale ones of the old size there and then NumPy-broadcasts a 1-row partial over them. */
#define EPGX_DERIV_THROUGH_PLAIN_OPS 1
/* Also carried in `deriv_flags`: keep every relaxation a stage of its own.  By default the library folds
 * precession-free relaxations (EPGX_OP_E with Im e0 = 0) into a neighbouring rotation whenever their tables
 * cannot be multiplied ahead of time (different index spaces): the wavefront then computes the coefficients of
 * E_after . T . E_before for its voxels at run time -- rounding-level differences to the operator-by-operator
 * product, as with EPGX_OP_T0 tables.  The fold is a property of the plan (every launch of it, at every capacity,
 * runs the same chains -- the same bits, up to the sum / difference form of rotations about x in the 64-order state-resident
 * kernels, see epgx_run); the host sets this flag for `simulate(fuse=False)` and for plans it runs with 16 orders
 * per voxel (one order per lane: a relaxation stage is then cheaper than the fold's extra loads).
 * Plans WITH derivative states fold as well, for state-resident launches at 64 orders whose records are mostly runs of one
 * repetition shape: the relaxations' partials then enter through their logarithmic form (a real relaxation's partial is a
 * multiple of the relaxation: the library derives (d e / e, d e2 / e2) per table entry on the device when the plan is
 * created and checks d r = -d e2), the rotation's partial is folded like the rotation.  The same route serves the
 * relaxation-only partials of epgx_fuse_partial tables inside runs.  Results equal the unfolded recurrence to rounding;
 * this flag switches all of it off. */
#define EPGX_PLAN_NO_FOLD 2

/* Record type of HOST-side signal arrays (and of narrowed device buffers, epgx_signal_narrow).  The reference returns
 * complex128 (statematrix.py:392); complex64 halves what crosses PCIe / xGMI -- the part of a large simulate() its caller
 * actually waits for -- at one rounding per record (<= 6e-8 relative, inside the 1e-6 parity bar). */
enum epgx_signal_dtype {
    EPGX_SIGNAL_C128 = 0, /* 2 x float64 (default everywhere) */
    EPGX_SIGNAL_C64 = 1   /* 2 x float32                      */
};

typedef struct epgx_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t clock_khz;
    int32_t reserved;
    int64_t hbm_bytes;
} epgx_device_info;

typedef struct epgx_ctx epgx_ctx;
typedef struct epgx_plan epgx_plan;
typedef struct epgx_state epgx_state;

/* ---- library / context -------------------------------------------------------------- */
int epgx_abi_version(void);
const char *epgx_last_error(void);
int epgx_device_count(void);
int epgx_ctx_create(int device, epgx_ctx **out);
int epgx_ctx_destroy(epgx_ctx *ctx);
int epgx_ctx_set_stream(epgx_ctx *ctx, void *hip_stream); /* adopt an external hipStream_t (NULL: own) */
int epgx_ctx_synchronize(epgx_ctx *ctx);
int epgx_ctx_info(epgx_ctx *ctx, epgx_device_info *out);
/* Device blocks freed through this library (epgx_free, plan / state destroy) are cached in the
 * context and recycled (stream-ordered, no device synchronisation); this returns them to HIP. */
int epgx_ctx_release_cache(epgx_ctx *ctx);

/* ---- raw device memory (replaces cupy's allocator for this path) --------------------- */
int epgx_malloc(epgx_ctx *ctx, int64_t bytes, void **dptr);
int epgx_free(epgx_ctx *ctx, void *dptr);
int epgx_memset(epgx_ctx *ctx, void *dptr, int value, int64_t bytes);
int epgx_memcpy_h2d(epgx_ctx *ctx, void *dptr, const void *host, int64_t bytes);
int epgx_memcpy_d2h(epgx_ctx *ctx, void *host, const void *dptr, int64_t bytes);
int epgx_memcpy_d2d(epgx_ctx *ctx, void *dst, const void *src, int64_t bytes);

/* Page-locked host memory, cached in the context like device blocks (pinning 336 MB costs tens of ms; a result
 * array that lives in a recycled pinned block receives its D2H copy at the full PCIe rate, asynchronously). */
int epgx_host_alloc(epgx_ctx *ctx, int64_t bytes, int32_t cached_only, void **hptr); /* cached_only != 0: hand out a recycled
                                                     block or *hptr = NULL (EPGX_OK either way), never pin new memory */
int epgx_host_free(epgx_ctx *ctx, void *hptr);
/* Page-lock memory the CALLER owns (hipHostRegister) -- e.g. a mapping of a shared-memory result that several rank
 * processes fill, each over its own PCIe link (epgpy_amd.distributed.SharedResult): copies into registered memory are
 * direct DMA, like copies into epgx_host_alloc blocks.  Unregister before the memory is unmapped or freed. */
int epgx_host_register(epgx_ctx *ctx, void *hptr, int64_t bytes);
int epgx_host_unregister(epgx_ctx *ctx, void *hptr);

/* ---- timing on the context's stream (HIP events) -------------------------------------- */
int epgx_timer_start(epgx_ctx *ctx);
int epgx_timer_stop(epgx_ctx *ctx, float *elapsed_ms); /* synchronises */

/* ---- plan ---------------------------------------------------------------------------- */
int epgx_plan_create(epgx_ctx *ctx, const epgx_plan_desc *desc, epgx_plan **out);
int epgx_plan_destroy(epgx_plan *plan);

/* ---- state (device-resident StateMatrix storage) -------------------------------------- */
int epgx_state_create(epgx_ctx *ctx, int64_t nvox, int32_t K, epgx_state **out); /* equilibrium, density 1 */
int epgx_state_destroy(epgx_state *st);
int epgx_state_upload(epgx_state *st, const double *half /*[nvox][3][K] c128*/,
                      const double *density /*[nvox] or NULL*/);
int epgx_state_download(const epgx_state *st, double *half, double *density /*nullable*/);
int epgx_state_copy(epgx_state *dst, const epgx_state *src); /* same nvox; K may differ (zero pad / truncate) */
int epgx_state_broadcast(epgx_state *dst, const epgx_state *src, const int32_t *src_index /*[dst nvox], host*/);
int epgx_state_info(const epgx_state *st, int64_t *nvox, int32_t *K, void **data, void **density);
/* dst += alpha * src (same nvox and K); zero_density != 0 also clears dst's density, which makes dst
 * a derivative state (no equilibrium term: DiffOperator.derive1, epgpy/diff.py:103-109). Replaces
 * StateMatrix.__iadd__ in diff.accumulate (epgpy/diff.py:553-563) for operator-by-operator use. */
int epgx_state_axpy(epgx_state *dst, const epgx_state *src, double alpha, int32_t zero_density);

/* ---- run ------------------------------------------------------------------------------ */
/* Apply operators [op_begin, op_end) of `plan` to voxels [vox0, vox0+nvox) of the plan's grid.
 *   in  : state to start from (its voxel j is grid voxel vox0+j), or NULL = equilibrium
 *   out : where the final state goes (may be `in` for in-place), or NULL = discard
 *   signal : device pointer, complex128 [n_adc][signal_ld]; voxel vox0+j writes column
 *            signal_col0 + j; NULL if the range holds no ADC
 *   K   : k-state capacity when both in and out are NULL (else taken from the states):
 *         64 .. 1024; 2048 (four wavefronts per voxel; T / T0 / E operators, shifts by +-1, probes, SPOILER / RESET / PD);
 *         plans with derivative states: up to 1024, at 1024 with ONE variable per plan (EPGX_ERR_UNSUPPORTED otherwise);
 *         or 16 / 32 for short state matrices (state-resident only; shifts by +-1,
 *         T / T0 / E operators and probes, at K = 16 also EPGX_OP_GS / EPGX_OP_D whose tables are then
 *         laid out [3][16] -- EPGX_ERR_UNSUPPORTED otherwise)
 * With in = out = NULL the state never leaves registers (state-resident mode); calling it once
 * per echo with in = out streams the state through HBM once per call (per-timestep mode).
 * State-resident launches from equilibrium at K >= 256 whose records mostly run while the state matrix is still short (every
 * shift adds one order: epgpy/shift.py:86,98) walk them with 1, 2, 4 .. K / 64 orders per lane -- the same bits.
 * One wavefront owns one voxel for the whole range, except in state-resident launches with
 * K <= 128 of ranges made of T / T0 / E / S(+-1) / probe operators only: there one wavefront owns
 * four voxels (16 lanes each, K / 16 orders per lane) -- same results, bit for bit; except rotations about x
 * at K = 64, which that kernel evaluates in a sum / difference form (16 instead of 18 instructions per order
 * slot): the same products in another association order, i.e. the last bits may differ (<= 1e-13 on O(1)
 * signals) from a per-timestep run of the same plan. */
int epgx_run(epgx_ctx *ctx, const epgx_plan *plan, int32_t op_begin, int32_t op_end,
             int64_t vox0, int64_t nvox, const epgx_state *in, epgx_state *out, int32_t K,
             void *signal, int64_t signal_ld, int64_t signal_col0);

/* Which kernel epgx_run would launch for operators [op_begin, op_end) at capacity K with these states (the same checks and the
 * same decision, no launch): its name with template arguments, e.g. "rows_grow_kernel<1>", "run_kernel<1, 2, true>",
 * "drun_kernel<1, 2, 309, 0>".  The decision depends on the plan, the range, K and whether states are given -- never on the
 * number of voxels.  For tests and tools that pin the kernel of a configuration. */
int epgx_kernel_for(epgx_ctx *ctx, const epgx_plan *plan, int32_t op_begin, int32_t op_end, int32_t K, const epgx_state *in,
                    epgx_state *out, char *name_out, int64_t name_bytes);

/* The whole plan, state-resident, over voxels [vox0, vox0 + nvox) in SLABS whose signal columns travel to the host while
 * the next slab computes (second stream + events): what a caller with host buffers waits for is then the PCIe copy
 * alone.
 *   signal_dev : device scratch [n_adc][dev_ld] c128, voxel vox0 + j in column j
 *   signal_host: [n_adc][host_ld] c128, voxel vox0 + j lands in column host_col0 + j.  Page-locked memory
 *                (epgx_host_alloc) receives the copies directly; ORDINARY (pageable) memory -- e.g. a fresh NumPy array
 *                of a caller that keeps every result, or a 16 GB result no pool should pin -- is filled through the
 *                context's ring of page-locked staging blocks by host threads that copy block k into place while
 *                block k + 1 crosses PCIe (HIP's own pageable path reaches 17 GB/s, this one the link rate)
 *   slab       : voxels per launch (0: chosen by the library)
 * Several contexts (GPUs) may run their ranges of one grid into one host array concurrently, each from its own host
 * thread (epgpy_amd `simulate(..., ngpu=N)`).  Synchronises the context before returning.
 *   host_dtype : enum epgx_signal_dtype of `signal_host` (host_ld / host_col0 count records of that type).  EPGX_SIGNAL_C64:
 *                every slab is narrowed on the device (epgx_signal_narrow into scratch of the context) before it leaves */
int epgx_run_to_host(epgx_ctx *ctx, const epgx_plan *plan, int32_t K, int64_t vox0, int64_t nvox, void *signal_dev,
                     int64_t dev_ld, void *signal_host, int64_t host_ld, int64_t host_col0, int64_t slab, int32_t host_dtype);
/* complex128 records src[rows][src_ld] (the first `cols` of every row) -> complex64 dst[rows][dst_ld], on the device, in the
 * context's stream order.  For consumers that move records themselves (epgx_download_2d, epgx_comm_gather) at half the bytes. */
int epgx_signal_narrow(epgx_ctx *ctx, const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols);
/* The copy half alone: rows x width_bytes from device memory (row pitch dev_pitch bytes) into host memory (row pitch
 * host_pitch), through the same pipeline -- direct when `host` is page-locked, staged + host threads otherwise.  Ordered
 * behind the work already enqueued on the context's stream; synchronises before returning. */
int epgx_download_2d(epgx_ctx *ctx, void *host, int64_t host_pitch, const void *dptr, int64_t dev_pitch,
                     int64_t width_bytes, int64_t rows);

/* Weighted reduction of signal rows over grid axes, on the device: what Adc(weights=..., reduce=...)
 * computes on the host in the reference (epgpy/probe.py:141-165: arr * weights, then arr.sum(axis)).
 *   out[r][o] = sum_j  w(o, j) * signal[row0 + r * row_step][voxel(o, j)],     r = 0 .. n_rows-1
 * o runs over the kept axes of the grid, j over the axes with reduce_axis[d] != 0 (both C order).
 * weights: device pointer to complex128 values addressed with weight_strides[d] (in elements, 0 on
 * axes the weights do not depend on), or NULL for a plain sum.  out: device, [n_rows][n_out] c128.
 * The buffer may hold a voxel SLAB of the grid only (multi-GPU runs: every rank reduces what it simulated, the partial
 * sums then meet in epgx_comm_reduce): column j of `signal` is grid voxel vox0 + j, j < nvox; voxels outside the slab
 * count as zero.  vox0 = 0, nvox = the whole grid: the plain case. */
int epgx_signal_reduce(epgx_ctx *ctx, const void *signal, int64_t signal_ld, int32_t row0,
                       int32_t row_step, int32_t n_rows, int32_t ndim, const int64_t *grid_shape,
                       const uint8_t *reduce_axis, const void *weights, const int64_t *weight_strides,
                       void *out, int64_t vox0, int64_t nvox);

/* Convenience for bindings that only have host arrays (what a ctypes/NumPy binding inside
 * the reference would call once per simulate()): builds the plan, runs the whole sequence
 * state-resident over the full grid, copies signal (and optionally the final state) back. */
int epgx_simulate_f64(epgx_ctx *ctx, const epgx_plan_desc *desc, int32_t K,
                      const double *init_half /*nullable [nvox][3][K]*/,
                      const double *density /*nullable [nvox]*/,
                      void *signal_out /*[n_adc][nvox] records of signal_dtype*/,
                      double *state_out /*nullable [nvox][3][K]*/,
                      int32_t signal_dtype /* enum epgx_signal_dtype */);

/* Same, with the voxel range split into contiguous slabs over the first `ngpu` devices of this process (results are
 * identical to the 1-GPU call).  The result is a HOST array, so nothing is gathered on the device side: every GPU runs
 * its slab through the epgx_run_to_host pipeline into its own columns of `signal_out` -- ngpu PCIe links in parallel,
 * no collective, no dependency on librccl.  (Device-resident consumers gather with epgx_comm_* below; with the
 * environment variable EPGX_SHARDED_GATHER=rccl this entry point also takes that route -- slabs to GPU 0 over
 * ncclSend / ncclRecv on a communicator set that is created once per process and `ngpu` -- which is how the
 * single-process form of the gather is exercised on a test box.) */
int epgx_simulate_sharded_f64(const epgx_plan_desc *desc, int32_t K, int32_t ngpu,
                              const double *density /*nullable*/, void *signal_out /*[n_adc][nvox] records of signal_dtype*/,
                              int32_t signal_dtype /* enum epgx_signal_dtype */);

/* ---- multi-GPU: the signal slabs meet ONCE, over RCCL / xGMI ---------------------------- */
/* The reference's only parallel attempt is the commented-out `simulate_parallel`
 * (epgpy/functions.py:195-248): a process pool over chunks of the parameter grid whose results are
 * concatenated at the end.  Here every rank (one process per GPU) simulates a contiguous voxel slab and
 * the slabs meet ONCE, on the device, over RCCL point-to-point transfers (ncclSend / ncclRecv in one
 * group: every peer uses its own xGMI link to the root) -- or, for probes that sum over grid axes
 * (Adc(weights=, reduce=), epgpy/probe.py:141-165), as ONE ncclReduce of the ranks' partial sums, so that
 * only the reduced records travel.  librccl.so.1 is loaded when the first of these functions is called; a
 * process that never calls them never pays for it.
 *   rank 0:      epgx_comm_unique_id(id)  -> hand the 128 bytes to every rank (any side channel)
 *   every rank:  epgx_comm_create(ctx, id, rank, world_size, &comm)      (collective; keep the communicator:
 *                creating one costs 0.1 - 1 s)
 *   every rank:  epgx_comm_gather(comm, slab, gathered, nbytes, root)    (collective)
 * `gathered` (root only; NULL elsewhere) holds world_size blocks of `nbytes`, block r = rank r's slab;
 * the root may have written its own slab straight into its block (send == gathered + rank * nbytes).
 * `nbytes` must be a multiple of 8 and the same on every rank.
 *
 * Stream order.  A communicator owns a side stream.  Every transfer below is enqueued THERE, behind an event that
 * marks what the context's stream holds at the time of the call -- so a rank can keep launching kernels for its
 * next sub-slab while the previous one is on the wire -- and epgx_comm_join makes the context's stream wait for
 * all transfers enqueued so far (no host synchronisation anywhere).  epgx_comm_gather = epgx_comm_gather_part
 * + epgx_comm_join. */
#define EPGX_COMM_ID_BYTES 128
typedef struct epgx_comm epgx_comm;
int epgx_comm_unique_id(void *id_out /* EPGX_COMM_ID_BYTES */);
int epgx_comm_create(epgx_ctx *ctx, const void *id, int32_t rank, int32_t world_size, epgx_comm **out);
int epgx_comm_destroy(epgx_comm *comm);
int epgx_comm_count(const epgx_comm *comm, int32_t *n_ranks); /* ncclCommCount: the ranks RCCL itself sees in the communicator */
int epgx_comm_gather(epgx_comm *comm, const void *send, void *gathered, int64_t nbytes, int32_t root);
/* One PART of a pipelined gather: like epgx_comm_gather, but rank r's `nbytes` land at gathered + r * block_stride
 * (block_stride >= nbytes, bytes), and the context's stream does not wait for the transfer (epgx_comm_join does).
 * A rank that cuts its slab into sub-slabs [n_adc][sub] laid out one after the other sends sub-slab k while it
 * computes sub-slab k + 1:  gathered = base + k * sub_bytes, block_stride = bytes of a whole rank block. */
int epgx_comm_gather_part(epgx_comm *comm, const void *send, void *gathered, int64_t nbytes, int64_t block_stride,
                          int32_t root);
int epgx_comm_join(epgx_comm *comm);
/* Sum of `count` doubles over all ranks, result at the root (recv: root only; may equal send): ncclReduce(ncclDouble,
 * ncclSum).  What Adc(reduce=) needs across GPUs: every rank sums over its own voxels (epgx_signal_reduce with its
 * slab), the partial sums meet here.  Enqueued like a gather part; followed by its own join. */
int epgx_comm_reduce(epgx_comm *comm, const void *send, void *recv, int64_t count, int32_t root);
/* Strided device-to-host copy of a [rows][width_bytes] block into a host array with `host_pitch` bytes
 * per row (assembles gathered slabs [n_adc][slab] into the caller's [n_adc][nvox] array during the
 * download, no host-side pass); synchronises the context's stream. */
int epgx_memcpy_d2h_2d(epgx_ctx *ctx, void *host, int64_t host_pitch, const void *dptr, int64_t dev_pitch,
                       int64_t width_bytes, int64_t rows);

#ifdef __cplusplus
}
#endif
#endif /* EPGX_H */
