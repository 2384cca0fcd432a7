/*
 * chebhip.h -- C ABI of libchebhip.so: the MI355X (gfx950) implementation of
 * the matrix-free Chebyshev spectral operator apply of jedbrown/spectral-petsc.
 *
 * This is the drop-in boundary.  Each entry point names the reference
 * interface it replaces (file:line relative to the reference tree).  The
 * PETSc-level symbols of chebyshev.h:27-34 (MatCreateCheb / ChebMult /
 * ChebDestroy ...) are a thin adapter over these calls: see
 * adapter/chebyshev_petsc.c and INTEGRATION.md.
 *
 * Conventions
 *  - all data IEEE float64; tensors row-major, LAST listed dim fastest
 *    (chebyshev.c:107-120); grid index i along a dim is x_i = cos(i pi/(P-1)).
 *  - *_dev pointers are device (HBM) pointers; `stream` is a hipStream_t passed
 *    as void* (NULL = default stream).  Device-pointer calls are asynchronous
 *    on that stream; *_host calls stage through device memory and return after
 *    the result is in the host buffer.  A stream created with
 *    hipStreamNonBlocking is fine: state the library allocates on first use is
 *    complete before the call that allocates it enqueues anything.  The set_* /
 *    create / update calls that take host data are synchronous copies that do
 *    NOT wait for work still queued on a non-blocking stream: synchronise that
 *    stream before changing the state a queued call reads (as before VecSet on
 *    a vector a MatMult in flight is reading).
 *  - every call returns 0 on success or a CHEBHIP_ERR_* code; nothing throws or
 *    exits across the ABI.  chebhip_last_error() returns a message for the
 *    calling thread's most recent failure.
 *  - a handle may be used from one host thread at a time (the reference's ctx
 *    is likewise non-reentrant: one mutable work buffer, chebyshev.h:23).
 *  - there is NO CPU fallback: without a usable HIP device the create calls
 *    fail with CHEBHIP_ERR_DEVICE.
 */
#ifndef CHEBHIP_H
#define CHEBHIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CHEBHIP_OK = 0,
  CHEBHIP_ERR_SIZE = 1,     /* n < 2              (chebyshev.c:18,98)  -> PETSC_ERR_USER */
  CHEBHIP_ERR_TDIM = 2,     /* tr out of range    (chebyshev.c:106)    -> PETSC_ERR_USER */
  CHEBHIP_ERR_DIMS = 3,     /* bad dims / product (chebyshev.c:122)    -> PETSC_ERR_USER */
  CHEBHIP_ERR_ARG = 4,      /* NULL handle/pointer, unsupported value  -> PETSC_ERR_ARG_WRONG */
  CHEBHIP_ERR_DEVICE = 5,   /* HIP runtime failure / no device         -> PETSC_ERR_LIB */
  CHEBHIP_ERR_MEMORY = 6    /* allocation failure                      -> PETSC_ERR_MEM */
};

const char *chebhip_last_error(void);
/* Library/ABI version (major*100 + minor) and the offload arch it was built for. */
int chebhip_version(void);
const char *chebhip_arch(void);

/* A linear map on device vectors: y = A x.  ell_op_mult, stokes_op_mult and stokes_op_mult_vv have
 * exactly this shape (ctx = the operator handle) and can be passed as is. */
typedef int (*chebhip_apply_fn)(void *ctx, const double *x_dev, double *y_dev, void *stream);
/* Sums `count` device doubles over the ranks in place, ordered on `stream` (ncclAllReduce of a few doubles per
 * Krylov iteration): what a multi-rank host gives the solvers below. */
typedef int (*chebhip_reduce_fn)(void *ctx, double *vals_dev, int count, void *stream);

/* ------------------------------------------------------------------------- */
/* Kernel level: the N-D Chebyshev derivative (chebyshev.h:18-24,31-34).      */
/* ------------------------------------------------------------------------- */
typedef struct cheb_plan cheb_plan;

/* Replaces MatCreateCheb (chebyshev.c:89-138) and, for rank 1, MatCreateChebD1
 * (chebyshev.c:8-33).  dims is copied.  Errors as chebyshev.c:98,106,122. */
int cheb_plan_create(int rank, int tr, const int *dims, cheb_plan **out);

/* Replaces ChebMult (chebyshev.c:142-199) / ChebD1Mult (:37-71):
 * y = d/dx_tr x.  x and y are distinct N-element arrays, x is not modified. */
int cheb_apply(cheb_plan *plan, const double *x_dev, double *y_dev, void *stream);
int cheb_apply_host(cheb_plan *plan, const double *x_host, double *y_host);

/* Replaces ChebDestroy (chebyshev.c:223-235) / ChebD1Destroy (:75-85). */
int cheb_plan_destroy(cheb_plan *plan);

/* Number of elements N = prod(dims) the plan was created for. */
long cheb_plan_size(const cheb_plan *plan);

/* Slab / pencil building block for the multi-GPU path (no counterpart in the serial reference):
 * a plan on a tensor that stores only the INTERIOR points of every line along `tr`
 * (dims[tr] = P-2; the end points are implicit zeros) -- the layout of the reference's global
 * vectors (SetupBC, elliptic.C:372-434) and of any slab cut from them along another dim. */
int cheb_plan_create_trimmed(int rank, int tr, const int *dims, cheb_plan **out);
/* y = acc + alpha * (D_tr D_tr x) at the stored points: one direction of the linear
 * MatMult_Elliptic (elliptic.C:309-334 with eta = 1, deta = 0).  acc may be NULL, or alias y. */
int cheb_apply_lap1d(cheb_plan *plan, const double *x_dev, const double *acc_dev, double alpha,
                     double *y_dev, void *stream);
/* Copies between a slab (m0, M1, R) row-major and the buffer an all-to-all moves: for every peer s the
 * block slab[:, c1[s]:c1[s+1], :] contiguously, blocks in rank order (c1: G+1 host values, 0 .. M1).
 * pack: buf <- slab (before the forward transpose).  unpack_add: out = acc + alpha * slab-ordered(buf)
 * (after the backward transpose; acc may be NULL or alias out). */
int cheb_slab_pack(long m0, long M1, long R, int G, const long *c1_host, const double *slab_dev,
                   double *buf_dev, void *stream);
int cheb_slab_unpack_add(long m0, long M1, long R, int G, const long *c1_host, const double *buf_dev,
                         const double *acc_dev, double alpha, double *out_dev, void *stream);

/* ------------------------------------------------------------------------- */
/* Operator level: the scalar elliptic MatShell (elliptic.C:78-86,250-293).   */
/* Vectors at this boundary are the reference's GLOBAL vectors: interior      */
/* nodes only, row-major (SetupBC, elliptic.C:372-434).  All work vectors     */
/* (w[2+d], gradu[d], eta, deta) live in HBM inside the handle.               */
/* ------------------------------------------------------------------------- */
typedef struct ell_op ell_op;

/* Replaces MatCreate_Elliptic (elliptic.C:250-293) with the homogeneous
 * Dirichlet boundary function DirichletBdy (:468-476); 1 <= d <= 10 as the
 * driver allows (elliptic.C:137).  State after create: eta = 1, deta = 0
 * (elliptic.C:265-266), gradu = 0, dirichlet values = 0. */
int ell_op_create(int d, const int *dims, ell_op **out);
int ell_op_destroy(ell_op *op);                       /* MatDestroy_Elliptic, elliptic.C:343-368 */

long ell_op_local_size(const ell_op *op);             /* N  = prod dims                 */
long ell_op_global_size(const ell_op *op);            /* g  = prod (dims-2): MatShell n */
long ell_op_dirichlet_size(const ell_op *op);         /* N - g                          */

/* Replaces MatMult_Elliptic (elliptic.C:297-339): V = A(eta,deta,gradu) U. */
int ell_op_mult(ell_op *op, const double *U_dev, double *V_dev, void *stream);
int ell_op_mult_host(ell_op *op, const double *U_host, double *V_host);

/* Replaces FormFunction (elliptic.C:481-533): rhs = F(U) - b, and refreshes the
 * operator state eta, deta, gradu used by later ell_op_mult calls.  b may be
 * NULL (treated as 0). */
int ell_op_function(ell_op *op, double gamma, double exponent, const double *U_dev,
                    const double *b_dev, double *rhs_dev, void *stream);
int ell_op_function_host(ell_op *op, double gamma, double exponent, const double *U_host,
                         const double *b_host, double *rhs_host);

/* Sets c->dirichlet (elliptic.C:462,667-668): compact boundary values in
 * BlockIt (row-major) order, ell_op_dirichlet_size() doubles, HOST pointer. */
int ell_op_set_dirichlet(ell_op *op, const double *values_host);

/* Copies operator state to the host for inspection: which = 0 eta, 1 deta,
 * 2+k gradu[k] (each N doubles). */
int ell_op_get_state(ell_op *op, int which, double *dst_host);
/* Overwrites operator state from the host (same `which` codes). */
int ell_op_set_state(ell_op *op, int which, const double *src_host);

/* Slab mode for the multi-GPU path (no counterpart in the serial reference): the handle owns the planes
 * [lo, hi) of grid dimension 0 and its vectors are the serial ones restricted to the slab (contiguous pieces).
 * Sweeps along dimension 0 are delegated: dim0(ctx, 0, 1, in, NULL, alpha, out, stream) must produce
 * out = alpha * D_0 in for one slab field (transpose -> ell_op_pencil_sweep on (dims[0], ncol) -> transpose).
 * ell_op_mult / ell_op_function / set_* / get_state work unchanged on the slab, for any coefficient state. */
typedef int (*ell_dim0_fn)(void *ctx, int kind, int nfields, const double *in_dev, const double *acc_dev,
                           double alpha, double *out_dev, void *stream);
int ell_op_create_slab(int d, const int *dims, int lo, int hi, ell_dim0_fn dim0, void *dim0_ctx, ell_op **out);
int ell_op_pencil_sweep(ell_op *op, long ncol, const double *in_dev, double *out_dev, void *stream);

/* ------------------------------------------------------------------------- */
/* Operator level: the Stokes MatShells (StokesCtx stokes.C:40-65,             */
/* StokesCreate :257-344) with -boundary 0: every boundary node is a velocity  */
/* Dirichlet node, numMixed == 0 (the StokesMixed* hooks are no-ops).          */
/* Vector layouts (StokesSetupDomain, stokes.C:773-938), I = interior nodes:   */
/*   full global  g  = (d+1)*I : [v_0 .. v_{d-1}, p] per interior node         */
/*   velocity     gv = d*I     : node-major        pressure gp = I             */
/*   dirichlet    dv = d*(N-I) : boundary nodes in BlockIt order, node-major   */
/* d = 2 or 3 (StokesPressureReduceOrder, stokes.C:1036).                      */
/* ------------------------------------------------------------------------- */
typedef struct stokes_op stokes_op;

int stokes_op_create(int d, const int *dims, stokes_op **out);      /* StokesCreate, stokes.C:257-344 */
int stokes_op_destroy(stokes_op *op);                               /* StokesDestroy, stokes.C:348-388 */
/* which: 0 local nodes N, 1 interior nodes I, 2 gv, 3 gp, 4 g, 5 dv */
long stokes_op_size(const stokes_op *op, int which);

/* options->rheology (stokes.C:1920-1944): kind 0 = StokesRheologyLinear, 1 = StokesRheologyPower
 * with -hardness, -exponent, -eps, -gamma0 (stokes.C:412-415). */
int stokes_op_set_rheology(stokes_op *op, int kind, double hardness, double exponent,
                           double regularization, double gamma0);
int stokes_op_set_dirichlet(stokes_op *op, const double *values_host);   /* c->dirichlet, dv doubles */
int stokes_op_set_force(stokes_op *op, const double *force_host);        /* c->force, g doubles     */

/* StokesMatMult (stokes.C:499-519) on full global vectors. */
int stokes_op_mult(stokes_op *op, const double *xG_dev, double *yG_dev, void *stream);
/* StokesMatMultVV (:623-676), StokesMatMultPV (:557-566), StokesMatMultVP (:599-619): the inner
 * MatShells MatVV / MatPV / MatVP that the Schur complement (:523-535) and the block
 * preconditioners (:1714-1817) call. */
int stokes_op_mult_vv(stokes_op *op, const double *vG_dev, double *vG_out_dev, void *stream);
int stokes_op_mult_pv(stokes_op *op, const double *vG_dev, double *pG_out_dev, void *stream);
int stokes_op_mult_vp(stokes_op *op, const double *pG_dev, double *vG_out_dev, void *stream);
/* StokesMatMultSchur (:523-535): out = -PV * solve(VV, VP * pG), all on device vectors.
 * `inner_solve(ctx, rhs_dev, sol_dev, stream)` stands for KSPSolve(KSPSchurVelocity) (:531; operators
 * MatVV / MatVVPC, options prefix svel_, :338-341): the PETSc adapter passes a wrapper around that KSP.
 * NULL selects the built-in solver: unpreconditioned restarted GMRES on MatVV with KSP's defaults
 * (restart 30, rtol 1e-5, atol 1e-50, max_it 10000, zero initial guess), see chebhip_fgmres_* below. */
int stokes_op_mult_schur(stokes_op *op, const double *pG_dev, double *pG_out_dev,
                         chebhip_apply_fn inner_solve, void *inner_ctx, void *stream);
/* The same four applies on COMPONENT-MAJOR velocity vectors (component c of interior node n at c * I + n instead of the
 * reference's n * d + c): the layout the block preconditioners keep for their inner Krylov solves (stokes_saddle_*), in which a
 * velocity vector is d stacked scalar fields for MatVVPC's line transforms too -- no (de)interleaving pass per inner iteration.
 * No counterpart in the reference (its vectors are node-major throughout, stokes.C:284-290); pressure vectors are unaffected.
 * stokes_op_mult_schur_cm requires inner_solve, which receives and returns component-major vectors. */
int stokes_op_mult_vv_cm(stokes_op *op, const double *v_cm_dev, double *v_cm_out_dev, void *stream);
int stokes_op_mult_pv_cm(stokes_op *op, const double *v_cm_dev, double *pG_out_dev, void *stream);
int stokes_op_mult_vp_cm(stokes_op *op, const double *pG_dev, double *v_cm_out_dev, void *stream);
int stokes_op_mult_schur_cm(stokes_op *op, const double *pG_dev, double *pG_out_dev,
                            chebhip_apply_fn inner_solve_cm, void *inner_ctx, void *stream);
int stokes_op_set_inner_solver(stokes_op *op, int restart, double rtol, double atol, int max_it);
int stokes_op_inner_iterations(const stokes_op *op);   /* MatVV applies of the last built-in inner solve */
/* Slab mode: the built-in inner solve runs on distributed velocity vectors (see chebhip_fgmres_set_reduce). */
int stokes_op_set_inner_reduce(stokes_op *op, chebhip_reduce_fn reduce, void *ctx);
/* StokesFunction (:680-758): yG = F(xG) - force; refreshes eta, deta, strain. */
int stokes_op_function(stokes_op *op, const double *xG_dev, double *yG_dev, void *stream);
/* Operator state to/from the host: which = 0 eta (N), 1 deta (N), 2+j strain[j] (N*d).  The strain is the symmetrised
 * velocity gradient (stokes.C:718-722): strain[j][k] == strain[k][j]; the Jacobian apply reads the entries with j <= k. */
int stokes_op_get_state(stokes_op *op, int which, double *dst_host);
int stokes_op_set_state(stokes_op *op, int which, const double *src_host);

/* Slab mode for the multi-GPU path (no counterpart in the serial reference).  The handle owns the
 * planes [lo, hi) of grid dimension 0; its vectors are the serial ones restricted to the slab (contiguous
 * pieces, dimension 0 being outermost).  Everything along dimension 0 is delegated to `dim0`:
 *   kind 0: out = (acc ? acc : 0) + alpha * D_0 in   for nfields stacked slab fields of N nodes each
 *   kind 1: out = D_0 (x-line pressure extrapolation of in)              (one field; stokes.C:1064-1074, :611)
 *   kind 2: the first nfields - 1 fields as kind 0 (acc NULL, alpha 1), the LAST field as kind 1: the velocity gradient and the
 *           pressure gradient of a StokesMatMult / StokesFunction along dimension 0 in ONE round trip (the fields are stacked:
 *           the handle keeps the pressure behind the velocity, and the pressure gradient behind the velocity gradient)
 * The driver (spectral-petsc_amd/dist.py) implements it as transpose -> pencil call below -> transpose.
 * All other entry points (mult, mult_vv/pv/vp, function, set_*, get/set_state) work unchanged on the slab. */
typedef int (*stokes_dim0_fn)(void *ctx, int kind, int nfields, const double *in_dev, const double *acc_dev,
                              double alpha, double *out_dev, void *stream);
int stokes_op_create_slab(int d, const int *dims, int lo, int hi, stokes_dim0_fn dim0, void *dim0_ctx, stokes_op **out);
/* Pencil side: arrays (nfields, dims[0], ncol), lines along dimension 0 with stride ncol. */
int stokes_op_pencil_sweep(stokes_op *op, int nfields, long ncol, const double *in_dev, double *out_dev, void *stream);
int stokes_op_pencil_pressure(stokes_op *op, long ncol, double *p_pencil_dev, double *gp0_pencil_dev, void *stream);
/* Kind 2 on the pencils in ONE launch: nvel stacked velocity fields swept with D and, behind them, the pressure field treated as
 * stokes_op_pencil_pressure treats it (two jobs of one launch where the kernels allow it, otherwise the two calls above). */
int stokes_op_pencil_sweep_pressure(stokes_op *op, int nvel, long ncol, double *in_dev, double *out_dev, void *stream);

/* ------------------------------------------------------------------------- */
/* Multi-GPU host for BASELINE config 3: the linear 3-D Poisson    */
/* matvec slab-partitioned along dimension 0, one rank per GPU, two            */
/* all-to-all exchanges per matvec (slab <-> pencil), local sweeps overlapped   */
/* with them on a second stream, accumulation in the serial order k = 0,1,2.   */
/* No counterpart in the serial reference (elliptic.C:262, nk.c:63); any G      */
/* reproduces the G = 1 vector.  Vectors: this rank's contiguous piece of the   */
/* reference's global vector ([chebhip_dist_slab_offset, + local_size)).        */
/* ------------------------------------------------------------------------- */
typedef struct chebhip_dist chebhip_dist;
/* Moves one exchange: send_dev holds, peer-major and contiguous, send_counts[s] doubles for every rank s; the
 * counts received from rank s are recv_counts[s], stored peer-major in recv_dev; ordered on `stream`. */
typedef int (*chebhip_exchange_fn)(void *ctx, const double *send_dev, const long *send_counts, double *recv_dev,
                                   const long *recv_counts, void *stream);
int chebhip_dist_create(int d, const int *dims, int nranks, int rank, chebhip_dist **out);
int chebhip_dist_destroy(chebhip_dist *D);
long chebhip_dist_local_size(const chebhip_dist *D);
long chebhip_dist_slab_offset(const chebhip_dist *D);
/* Transport: grouped ncclSend / ncclRecv on `nccl_comm` (an ncclComm_t of nranks ranks, rank order as at create;
 * rccl.h:700,722,923) -- RCCL is looked up at run time, never linked -- or any other exchange function. */
int chebhip_dist_use_rccl(chebhip_dist *D, void *nccl_comm);
int chebhip_dist_set_exchange(chebhip_dist *D, chebhip_exchange_fn fn, void *ctx);
/* MatMult_Elliptic (elliptic.C:297-339, eta = 1, deta = 0) on the slab: V = -(L_0 + L_1 + ..) U. */
int chebhip_dist_mult(chebhip_dist *D, const double *U_slab_dev, double *V_slab_dev, void *stream);
/* The same matvec on nrhs vectors at once (1..64; U, V: nrhs consecutive slab vectors of chebhip_dist_local_size doubles each): one
 * launch per direction on the stacked slabs / pencils, one pack, ONE grouped exchange each way carrying every vector's blocks, one
 * final sum -- the fixed cost of a launch of 256-point lines and of an RCCL launch is paid per batch, not per vector.  Each vector's
 * result is chebhip_dist_mult's.  For the independent vectors of a block / s-step Krylov method or several right-hand sides; needs a
 * chebhip_comm transport (chebhip_dist_use_comm / _use_rccl).  Collective.  The serial reference has no counterpart. */
int chebhip_dist_mult_batch(chebhip_dist *D, int nrhs, const double *U_slabs_dev, double *V_slabs_dev, void *stream);
/* For hosts without a communicator of their own: rank 0 makes the 128-byte id, the host hands it to every rank. */
int chebhip_rccl_unique_id(void *id128);
int chebhip_rccl_comm_create(int nranks, int rank, const void *id128, void **nccl_comm_out);
int chebhip_rccl_comm_destroy(void *nccl_comm);
/* A chebhip_reduce_fn (ctx = the ncclComm_t): ncclAllReduce of the few doubles a Krylov iteration needs. */
int chebhip_rccl_reduce(void *nccl_comm, double *vals_dev, int count, void *stream);

/* ------------------------------------------------------------------------- */
/* Transports of the slab-partitioned operators.  A chebhip_comm   */
/* carries the exchanges (slab <-> pencil transposes) and the few-double       */
/* reductions of a Krylov iteration among G ranks:                             */
/*   _create_rccl      one process per GPU; each exchange is one grouped       */
/*                     ncclSend/ncclRecv launch over xGMI (rccl.h:700,722,923) */
/*   _create_local     ranks are host threads of ONE process, one stream (and, */
/*                     on a multi-GPU node, one device) each, peer access      */
/*                     between the devices.  A DIRECT transport: the slab      */
/*                     drivers read the peers' arrays in place (no pack, no    */
/*                     messages; two thread rendezvous per round trip), other  */
/*                     exchanges are event-ordered device copies.  Every rank  */
/*                     must make the same sequence of collective calls from    */
/*                     its own thread -- INCLUDING the destroy of a driver     */
/*                     that has run on it (chebhip_dist_destroy,               */
/*                     chebhip_dist_stokes_destroy, chebhip_dist_ell_destroy:  */
/*                     nothing is freed while a peer may still read it).       */
/*   _create_callback  any other transport (the gloo staging of the tests)     */
/*   _create_ipc       ranks are PROCESSES of one node (a launcher's one       */
/*                     process per GPU): the direct route of _create_local     */
/*                     across address spaces -- pointers travel as             */
/*                     (hipIpcMemHandle_t, offset) through a shared-memory     */
/*                     segment, "my arrays are complete" as sequence numbers   */
/*                     in it (hipStreamWriteValue64, one polling launch),      */
/*                     layered over a message transport (_create_rccl, a       */
/*                     callback) that carries the segment exchanges and the    */
/*                     reductions.  The same collective rules as _create_local */
/*                     (destroys included).  Vectors handed to a driver on it  */
/*                     must come from hipMalloc (a caching allocator's blocks  */
/*                     qualify, a virtual-memory allocator's do not).          */
/* ------------------------------------------------------------------------- */
typedef struct chebhip_comm chebhip_comm;
typedef struct chebhip_local_group chebhip_local_group;
typedef struct chebhip_ipc_group chebhip_ipc_group;
/* One exchange: segment i sends send_counts[i] doubles at send_dev[i] to peers[i] and receives recv_counts[i] doubles
 * from it into recv_dev[i]; the k-th segment a rank addresses to peer s meets the k-th segment s addresses to that
 * rank.  The rank's own segments are not passed (device copies inside the library).  Ordered on `stream`. */
typedef int (*chebhip_exchangev_fn)(void *ctx, int nseg, const int *peers, const double *const *send_dev, const long *send_counts,
                                    double *const *recv_dev, const long *recv_counts, void *stream);
int chebhip_comm_create_rccl(void *nccl_comm, int nranks, int rank, chebhip_comm **out);      /* the ncclComm_t is not owned */
int chebhip_local_group_create(int nranks, chebhip_local_group **out);
int chebhip_local_group_destroy(chebhip_local_group *g);   /* after every rank thread has finished; the group outlives its communicators */
int chebhip_local_group_abort(chebhip_local_group *g);   /* a failing rank releases the ranks waiting for it: their calls return an error */
int chebhip_comm_create_local(chebhip_local_group *g, int rank, chebhip_comm **out);          /* call with the rank's device current */
int chebhip_comm_create_callback(int nranks, int rank, chebhip_exchangev_fn xfn, chebhip_reduce_fn rfn, void *ctx, chebhip_comm **out);
/* No wire: every "peer" is the rank itself (the kernels of the direct route with every byte read locally).  For timing the compute
 * side of one rank of an N-rank partition on one GPU; results are meaningless for N > 1. */
int chebhip_comm_create_null(int nranks, int rank, chebhip_comm **out);
/* ... with arrays of their own standing for the peers' k-th posted array (arrays[r], r = 0 .. nranks-1; NULL entries and arrays == NULL:
 * the rank's own): the kernels then read and write nranks distinct arrays, as among real ranks, instead of finding the "peers'" rows
 * in the caches because they are the rank's own.  Each array as large as what the driver posts at index k (chebhip_dist_mult: 0 = the
 * slab vector(s), 1 = the result array of the same size).  Not owned. */
int chebhip_comm_null_set_shadow(chebhip_comm *c, int k, const double *const *arrays);
/* Process ranks.  _open is collective: `name` is a POSIX shared-memory name ("/chebhip-<unique per group>") that rank 0 creates
 * and unlinks again once every rank holds the mapping; call it with the rank's device current.  `inner` (not owned, may be NULL
 * for one rank): a communicator of the same ranks on a message transport.  _close after the communicators and drivers made on
 * the group are gone; _abort releases ranks waiting in a rendezvous (their calls fail). */
int chebhip_ipc_group_open(const char *name, int nranks, int rank, chebhip_ipc_group **out);
int chebhip_ipc_group_close(chebhip_ipc_group *g);
int chebhip_ipc_group_abort(chebhip_ipc_group *g);
int chebhip_comm_create_ipc(chebhip_ipc_group *g, chebhip_comm *inner, chebhip_comm **out);
int chebhip_comm_destroy(chebhip_comm *c);
int chebhip_comm_size(const chebhip_comm *c);
int chebhip_comm_rank(const chebhip_comm *c);
/* A chebhip_reduce_fn (ctx = the chebhip_comm): sums `count` device doubles over the ranks in place; every rank gets the
 * same bits.  For chebhip_fgmres_set_reduce / stokes_op_set_inner_reduce. */
int chebhip_comm_reduce(void *comm, double *vals_dev, int count, void *stream);
/* The linear Poisson host above on any transport (chebhip_dist_use_rccl = _create_rccl + this). */
int chebhip_dist_use_comm(chebhip_dist *D, chebhip_comm *comm);

/* ------------------------------------------------------------------------- */
/* Multi-GPU hosts for BASELINE config 5 and for the elliptic operator in any  */
/* coefficient state, C++ behind this ABI (csrc/slabx.hip): each   */
/* rank owns a slab-mode handle on its planes of grid dimension 0; whatever    */
/* runs along dimension 0 (DV[0], DP[0], D_0, the x-line pressure              */
/* extrapolation stokes.C:1064-1074) is done on pencils: pack -> one grouped   */
/* exchange of all fields of the call -> pencil launch -> exchange -> unpack   */
/* with the AXPY folded in.  The handle returned by *_op() takes every         */
/* stokes_op_* / ell_op_* entry point; its vectors are this rank's contiguous  */
/* pieces of the serial ones (node ranges from *_ranges: [0],[1] interior      */
/* nodes lo/hi, [2],[3] boundary nodes lo/hi, in the serial BlockIt order).    */
/* Every rank must make the same sequence of calls.  No counterpart in the     */
/* serial reference (stokes.C:121 VecCreateSeq).                               */
/* ------------------------------------------------------------------------- */
typedef struct chebhip_dist_stokes chebhip_dist_stokes;
typedef struct chebhip_fdpc chebhip_fdpc;       /* the finite-difference preconditioner, declared below */
int chebhip_dist_stokes_create(int d, const int *dims, chebhip_comm *comm, chebhip_dist_stokes **out);   /* comm NULL: one rank */
int chebhip_dist_stokes_destroy(chebhip_dist_stokes *D);
stokes_op *chebhip_dist_stokes_op(chebhip_dist_stokes *D);        /* owned by D; StokesMatMultSchur's built-in inner solve all-reduces through comm */
/* The preconditioners on slabs (round 4; SURVEY 8f.1 / 8f.3 over ranks): MatVVPC (stokes.C:1160-1241) / FormJacobian's matrix
 * (elliptic.C:537-590) for the slab's unknowns -- chebhip_fdpc handles in slab mode, owned by the driver, line transforms along
 * dimension 0 on pencils through the driver's communicator.  Pass them to chebhip_fdpc_apply / stokes_saddle_create_slab. */
int chebhip_dist_stokes_pc(chebhip_dist_stokes *D, chebhip_fdpc **out);
int chebhip_dist_stokes_ranges(const chebhip_dist_stokes *D, long *ranges4);
typedef struct chebhip_dist_ell chebhip_dist_ell;
int chebhip_dist_ell_create(int d, const int *dims, chebhip_comm *comm, chebhip_dist_ell **out);
int chebhip_dist_ell_destroy(chebhip_dist_ell *D);
ell_op *chebhip_dist_ell_op(chebhip_dist_ell *D);
int chebhip_dist_ell_pc(chebhip_dist_ell *D, chebhip_fdpc **out);
int chebhip_dist_ell_ranges(const chebhip_dist_ell *D, long *ranges4);

/* ------------------------------------------------------------------------- */
/* Krylov driver on device vectors: the caller of the path.     */
/* KSPSolve with KSPFGMRES around MatMult_Elliptic (elliptic.C:181-185) and    */
/* KSPSchurVelocity inside StokesMatMultSchur (stokes.C:531).  Restarted       */
/* flexible GMRES, right preconditioner M (NULL = none; may change between     */
/* applications), classical Gram-Schmidt, convergence on                       */
/* |r| <= max(rtol |b|, atol) as KSPDefaultConverged.  All vectors stay in     */
/* HBM; the host sees one Hessenberg column per iteration.                     */
/* ------------------------------------------------------------------------- */
typedef struct chebhip_fgmres chebhip_fgmres;
int chebhip_fgmres_create(long n, int restart, chebhip_fgmres **out);
/* Vectors distributed over ranks (n = local entries): inner products are completed by `reduce`. NULL = one rank. */
int chebhip_fgmres_set_reduce(chebhip_fgmres *k, chebhip_reduce_fn reduce, void *ctx);
int chebhip_fgmres_destroy(chebhip_fgmres *k);
int chebhip_fgmres_set_tolerances(chebhip_fgmres *k, double rtol, double atol, int max_it);
/* x_nonzero = 0: zero initial guess (x is overwritten); 1: x_dev holds the initial guess. */
int chebhip_fgmres_solve(chebhip_fgmres *k, chebhip_apply_fn A, void *actx, chebhip_apply_fn M, void *mctx,
                         const double *b_dev, double *x_dev, int x_nonzero, void *stream);
int chebhip_fgmres_iterations(const chebhip_fgmres *k);   /* operator applies of the last solve */
double chebhip_fgmres_residual(const chebhip_fgmres *k);  /* last (recurrence) residual norm */
/* KSPConvergedReason of the last solve: 2 rtol, 3 atol, -3 max_it, -9 NaN/breakdown */
int chebhip_fgmres_reason(const chebhip_fgmres *k);

/* ------------------------------------------------------------------------- */
/* The finite-difference preconditioner of the reference:  */
/* FormJacobian's 2d+1-point matrix P on the collocation nodes                  */
/* (elliptic.C:537-590; handed to PCILU, elliptic.C:184-185) and its per-       */
/* component twin MatVVPC of StokesPCSetUp0 (stokes.C:1160-1241).  The matrix   */
/* lives on the device as coefficient arrays; the approximate solve is a fast   */
/* diagonalisation of its constant-coefficient part (dense line transforms on   */
/* the sweep kernel), used alone or inside `sweeps` inner GMRES steps on P.      */
/* Vectors: the operator's global vectors (scalar: g; Stokes: velocity gv).     */
/* ------------------------------------------------------------------------- */
int ell_pc_create(ell_op *op, chebhip_fdpc **out);          /* MatCreateSeqAIJ(.., 1+2d, ..) + PCILU, elliptic.C:163,184 */
int stokes_pc_create(stokes_op *op, chebhip_fdpc **out);    /* MatVVPC, stokes.C:1160-1241 (-pcvel 0)                    */
int chebhip_fdpc_destroy(chebhip_fdpc *pc);
/* Slab mode (the serial reference has no counterpart): the handle preconditions the unknowns of ONE slab of planes of
 * dimension 0; its approximate solve is z = P_1^-1 (r / eta) by fast diagonalisation (sweeps = 0), whose line transforms along
 * dimension 0 run on pencils: `dim0` (collective over the ranks) moves `nfields` stacked interior fields of the slab at in_dev to
 * pencils, applies chebhip_fdpc_pencil_transform and moves the result back to out_dev.  Made by chebhip_dist_stokes_pc /
 * chebhip_dist_ell_pc (csrc/slabx.hip), which supply the callback; i0_offset = interior planes owned by lower ranks. */
typedef int (*chebhip_fdpc_dim0_fn)(void *ctx, int backward, int nfields, const double *in_dev, double *out_dev, void *stream);
int stokes_pc_create_slab(stokes_op *slab_op, long i0_offset, chebhip_fdpc_dim0_fn dim0, void *ctx, chebhip_fdpc **out);
int ell_pc_create_slab(ell_op *slab_op, long i0_offset, chebhip_fdpc_dim0_fn dim0, void *ctx, chebhip_fdpc **out);
int chebhip_fdpc_pencil_transform(chebhip_fdpc *pc, int backward, int nfields, long ncol, const double *in_dev, double *out_dev, void *stream);
/* FormJacobian / StokesPCSetUp0: (re)assemble P from the operator's current eta, deta (and gradu): call after
 * ell_op_function / stokes_op_function or set_state.  Done implicitly on first use. */
int chebhip_fdpc_update(chebhip_fdpc *pc, void *stream);
int chebhip_fdpc_set_sweeps(chebhip_fdpc *pc, int sweeps);  /* inner GMRES steps on P per apply (default 1; 0: P_1^-1 (r/eta) alone) */
/* y = P x: MatMult on the assembled matrix (tests; the defect correction). */
int chebhip_fdpc_mult(chebhip_fdpc *pc, const double *x_dev, double *y_dev, void *stream);
/* z ~= P^-1 r.  Shape of chebhip_apply_fn with ctx = the handle: pass as the M of chebhip_fgmres_solve. */
int chebhip_fdpc_apply(void *pc, const double *r_dev, double *z_dev, void *stream);
/* The same for a Stokes velocity preconditioner (stokes_pc_create*) on component-major vectors (stokes_op_mult_vv_cm): the fast
 * diagonalisation z = P_1^-1 (r / eta) (sweeps = 0) only. */
int chebhip_fdpc_apply_cm(void *pc, const double *r_cm_dev, double *z_cm_dev, void *stream);

/* ------------------------------------------------------------------------- */
/* The block preconditioners of the Stokes saddle-point system:  */
/* StokesPCApply0..3 (stokes.C:1714-1817) with the inner solves KSPVelocity,    */
/* KSPSchur and KSPSchurVelocity (stokes.C:328-341) on device vectors.          */
/* ------------------------------------------------------------------------- */
typedef struct stokes_saddle stokes_saddle;
int stokes_saddle_create(stokes_op *op, stokes_saddle **out);
/* On slabs: slab_op = chebhip_dist_stokes_op(D), slab_pc = chebhip_dist_stokes_pc(D) (borrowed), reduce / reduce_ctx =
 * chebhip_comm_reduce with the driver's communicator.  The inner solves and the removal of the constant pressure mode complete
 * their sums over the ranks; stokes_saddle_apply is then collective.  The serial reference has no counterpart.
 * Destroy order: the saddle BORROWS slab_op, slab_pc and reduce_ctx -- destroy it before chebhip_dist_stokes_destroy(D) (which frees
 * the pc) and before the communicator. */
int stokes_saddle_create_slab(stokes_op *slab_op, chebhip_fdpc *slab_pc, chebhip_reduce_fn reduce, void *reduce_ctx, stokes_saddle **out);
int stokes_saddle_destroy(stokes_saddle *s);
/* -pc_saddle_type (stokes.C:177-187): 0 block LU, 1 upper triangular, 2 block diagonal, 3 lower triangular. */
int stokes_saddle_set_type(stokes_saddle *s, int type);
/* Inner solves: which = 0 KSPVelocity (-vel_), 1 KSPSchur (-schur_), 2 KSPSchurVelocity (-svel_); max_it GMRES
 * iterations at most (restart 30) to relative tolerance rtol.  For the two velocity solves max_it = 0 means
 * -ksp_type preonly: one application of the MatVVPC solve.  Defaults (README:43): 4 / 3 / preonly, rtol 1e-5. */
int stokes_saddle_set_inner(stokes_saddle *s, int which, int max_it, double rtol);
/* KSPSchur's preconditioner: 1 (default) = PCJACOBI with the diagonal of StokesMatGetDiagonalSchur (stokes.C:330-331,
 * :538-553: 1/eta at the interior nodes, so residuals are multiplied by the viscosity; left-preconditioned as PETSc's
 * GMRES is), 0 = none (-schur_pc_type none). */
int stokes_saddle_set_schur_jacobi(stokes_saddle *s, int on);
/* Inner GMRES steps on MatVVPC inside its approximate solve (see chebhip_fdpc_set_sweeps); default 0. */
int stokes_saddle_set_pc_sweeps(stokes_saddle *s, int sweeps);
/* StokesPCSetUp0 (stokes.C:1160-1241): re-assemble MatVVPC from the operator's current eta. */
int stokes_saddle_setup(stokes_saddle *s, void *stream);
/* y = M^-1 x on full global vectors (the pressure part of y has zero mean: KSPSetNullSpace, stokes.C:1017-1019).
 * Shape of chebhip_apply_fn with ctx = the handle. */
int stokes_saddle_apply(void *s, const double *x_dev, double *y_dev, void *stream);
/* MatVV applies (which = 0) / MatSchur applies (1) spent by the inner solves of the last apply. */
int stokes_saddle_iterations(const stokes_saddle *s, int which);

/* ------------------------------------------------------------------------- */
/* Instrumentation (the reference has none: SURVEY 5.1).                      */
/* ------------------------------------------------------------------------- */
/* Run-time options: named integer switches, process-wide, read where they apply (the library never reads the
 * environment).  Unknown names are an error.  Set them before creating the handles they concern.
 *   general_kernels       1: every sweep runs the general 8-byte kernel (A/B against the 16-byte kernels)
 *   separate_launches     1: the d sweeps of a Stokes gradient / divergence are d launches instead of one
 *   vendor_gemm           1: plain sweeps of lines of more than 1024 points go to rocBLAS DGEMM (looked up at run time, never linked)
 *                            instead of the library's own FP64-VALU kernel.  Default 0: no vendor GEMM on any default path
 *   no_raw_transforms     1: the preconditioner's line transforms take two launches instead of one
 *   equal_shares          1: multi-job launches give every job min(tiles, CUs) workgroups instead of proportional shares
 *   force_gemm            1: every extent of 4 .. 256 points takes the library-DGEMM route of the longest lines (read at operator create)
 *   stokes_single_stream  1: StokesMatMult / StokesFunction keep the pressure chain on the caller's stream, also on large grids off the
 *                            fused-z route (read at create)
 *   eta_from_memory       1: FormFunction reads eta instead of forming 1 + gamma u^2 on chip (exponent 2)
 *   gather_pass           1: FormFunction always runs its gather pass, also for homogeneous Dirichlet rows
 *   rccl_self_messages    1: a rank's own block of an exchange goes through ncclSend / ncclRecv too (one-rank smoke runs)
 *   local_timeout_s       seconds a thread rank (LOCAL) or process rank (IPC: barrier and polling launch) waits for its peers before the group is aborted (default 120)
 *   dist_single_stream    chebhip_dist_mult's local sweeps: 0 (default) = by transport -- on a side stream (they overlap both exchanges) when
 *                            data leaves the device (RCCL, a callback transport, thread ranks on several devices), on the caller's stream
 *                            when it does not (one rank, the NULL transport, thread ranks sharing one device: there the side stream only
 *                            adds dependencies and costs 8-27 %); 1 = always the caller's stream; 2 = always the side stream
 *   dist_packed_exchange  1: chebhip_dist_mult on a direct transport (LOCAL thread ranks, NULL) runs pack / segment exchange / final sum
 *                            as on RCCL instead of reading the peers' slabs and pencil results in place; 2: in place, but the pencil is
 *                            filled by a copy launch first instead of the pencil sweep reading the peers' slabs itself; 3: the pencil
 *                            sweep reads the peers' slabs but stays a launch of its own (default on a small slab: one launch of the three
 *                            directions); 4: the pencil results stay where they are computed and the final sum reads the peers' (default:
 *                            the pencil sweep stores every output row into the result array of the rank that owns the plane, the final
 *                            sum reads local memory) (A/B and tests; set it before the first matvec of a handle, on every rank)
 *   long_lines_gemm       1: lines of 257 .. 1024 points go to rocBLAS instead of the library's own matrix-core kernel (A/B)
 *   pressure_passes       1: Stokes handles run the three boundary-extrapolation passes of StokesPressureReduceOrder before the
 *                            pressure gradient instead of folding each direction's extrapolation into its matrix (read at create)
 *   general_viscous       1: StokesMatMult / StokesMatMultVV take the general viscous block also when the viscosity is uniform and
 *                            eta' = 0 (linear rheology), instead of -eta/2 (sum_j D_j D_j v + grad div v) (read at create)
 *   poisson_launches      the constant-coefficient MatMult_Elliptic: 0 = by size (below 6 M unknowns: from 1.5 M on, in 3-D with lines of at most 128 points, two
 *                            jobs in one launch and a last direction that adds both terms as it stores, otherwise one launch of d jobs + a sum;
 *                            above: in 3-D, while the padded field has at most 9 M values, two jobs in one launch + a last direction that adds
 *                            both terms; otherwise a launch per direction), 1 = always the d-job launch (below 6 M unknowns), 2 = always a
 *                            launch per direction, 3 = the two-launch form at every large 3-D size (A/B: it loses from 240^3 on)
 *   dist_exact_order      1: chebhip_dist_mult adds its terms in the serial order V = ((T_0 + A_1) + A_2) (elliptic.C:331-334), which
 *                            reproduces the one-GPU vector to the bit; 0 (default): the local terms are accumulated into one array
 *                            by the sweeps themselves, V = T_0 + (A_1 + A_2) -- equal to rounding, one array less to read
 *   fdm_z_separate        1: the fast-diagonalisation solve runs its last forward line transform, the modal scaling and its first backward
 *                            line transform as separate launches also where the one-launch form exists (last dimension with 66 .. 128
 *                            interior points, an even number) (A/B)
 *   stokes_z_separate     1: StokesMatMult / StokesMatMultVV / StokesFunction run the z direction of the viscous block as separate passes
 *                            (z sweeps of the gradient launch, node loop, z sweeps of the divergence launch) also where the one-launch
 *                            form exists (d = 3 on one GPU, contiguous lines of 68 .. 128 points, at least 14 400 of them); same bits (A/B)
 *   saddle_node_major     1: the block preconditioners (stokes_saddle_*) keep the vectors of their inner velocity solves node-major as
 *                            the reference does, with a (de)interleaving pass around every MatVVPC solve (read at create; A/B)
 *   fdm_passes            1: the fast-diagonalisation solve z = P_1^-1 (r / eta) of the finite-difference preconditioners divides by eta
 *                            and by the modal sums in passes of their own, instead of multiplying by the reciprocals in the load of its
 *                            first and the store of its last forward line transform (A/B; the two differ in the last bits)
 *   full_stress_storage   1: Stokes handles keep all 9 stress / strain components instead of the 6 distinct ones (read at create)
 *   stokes_pressure_stream 1: where the fused-z route runs, the pressure-gradient sweeps go to a second stream between the gather and the
 *                            final scatter (rounds 2-4) instead of being jobs of the route's first launch (read at create; A/B: a tie)
 *   stokes_pressure_sweeps 1: where the fused-z route runs, StokesMatMult / StokesFunction run the three pressure-gradient sweeps and add grad p in
 *                            the final scatter (rounds 1-4), instead of subtracting the pressure -- face values extrapolated along each line, one
 *                            small launch -- from the diagonal stress so that the three divergence sweeps deliver -div tau + grad p at once
 *                            (read per call; A/B: the two agree to rounding, 4e-16 observed)
 *   krylov_exact_norm     1: chebhip_fgmres runs its Gram-Schmidt step as three launches with an explicit norm pass (rounds 1-4) instead of
 *                            two launches with one reduction and the stored vectors' exact norms carried beside the basis (read per solve; A/B)
 *   no_rocblas            deprecated alias (rounds 1-3) of vendor_gemm with the inverted meaning; still accepted */
int chebhip_set_option(const char *name, int value);
int chebhip_get_option(const char *name, int *value);
const char *chebhip_option_name(int index);     /* "" past the last option: enumerate from 0 */

/* Number of sweep-kernel launches issued by this process so far. */
long chebhip_launch_count(void);

/* Per-stage device timers: when enabled, every entry point listed below brackets its work with a hipEvent pair on
 * the caller's stream (inclusive times: stokes_saddle_apply contains the stokes_op_mult_vv calls of its inner
 * solves).  Reading drains the pending events (synchronises with them).  Off by default: no events, no cost. */
enum {
  CHEBHIP_STAGE_CHEB_APPLY = 0, CHEBHIP_STAGE_ELL_MULT, CHEBHIP_STAGE_ELL_FUNCTION, CHEBHIP_STAGE_STOKES_MULT,
  CHEBHIP_STAGE_STOKES_MULT_VV, CHEBHIP_STAGE_STOKES_MULT_PV, CHEBHIP_STAGE_STOKES_MULT_VP, CHEBHIP_STAGE_STOKES_FUNCTION,
  CHEBHIP_STAGE_STOKES_SCHUR, CHEBHIP_STAGE_FDPC_APPLY, CHEBHIP_STAGE_SADDLE_APPLY, CHEBHIP_STAGE_FGMRES_SOLVE,
  CHEBHIP_NSTAGES
};
int chebhip_timers_enable(int on);
int chebhip_timers_reset(void);
int chebhip_timers_read(int stage, double *total_ms, long *calls);
const char *chebhip_stage_name(int stage);

/* min / max of the viscosity left by the last stokes_op_function: the VecMin / VecMax the reference prints inside
 * StokesFunction (stokes.C:731-734).  Synchronises with `stream`; never called implicitly. */
int stokes_op_viscosity_range(stokes_op *op, double *eta_min, double *eta_max, void *stream);
/* StokesStateView (stokes.C:1821-1894, -output_vtk): legacy ASCII VTK file with velocity, pressure, force, eta, deta and
 * strain on the full grid, same sections and number format as the reference.  Host I/O; synchronises the device. */
int stokes_op_write_vtk(stokes_op *op, const double *state_dev, const char *path);

#ifdef __cplusplus
}
#endif
#endif
