/*
 * chebyshev_petsc.c -- the reference-side binding: the six PETSc symbols of chebyshev.h:27-34 and
 * the two operator callbacks (MatMult_Elliptic elliptic.C:297, FormFunction elliptic.C:481)
 * implemented over the C ABI of libchebhip.so (include/chebhip.h).
 *
 * NOT compiled in this repository: no PETSc is installed in the build image.  Written against the
 * current PETSc API (>= 3.19: MatShellSetOperation, VecGetArrayRead, PetscCall); the 2008-era spellings
 * the reference uses (PetscTruth, MatDestroy(Mat) by value, SETERRQ without comm) differ
 * only in the macros below.
 *
 * Build:  mpicc -c chebyshev_petsc.c -I$PETSC_DIR/include -I$PETSC_DIR/$PETSC_ARCH/include -I../include
 * Link :  ... chebyshev_petsc.o -L../spectral-petsc_amd -lchebhip $PETSC_LIB     (instead of -lfftw3)
 */
#include "chebyshev.h"
#include "chebhip.h"

/* Device Vecs (-vec_type hip, build with -DCHEBHIP_USE_DEVICE_VECS): the arrays PETSc hands out belong to work
 * queued on PETSc's own stream, so the chebhip launches go on THAT stream (PetscDeviceContextGetStreamHandle,
 * PETSc >= 3.18); nothing is synchronised with the host and Vec{HIP}RestoreArray* keeps PETSc's ordering. */
#if defined(PETSC_HAVE_HIP) && defined(CHEBHIP_USE_DEVICE_VECS)
#include <petscdevice_hip.h>
static PetscErrorCode cheb_petsc_stream(void **stream) {
  PetscDeviceContext dctx;
  void              *handle;
  PetscFunctionBegin;
  PetscCall(PetscDeviceContextGetCurrentContext(&dctx));
  PetscCall(PetscDeviceContextGetStreamHandle(dctx, &handle));   /* hipStream_t* */
  *stream = (void *)(*(hipStream_t *)handle);
  PetscFunctionReturn(PETSC_SUCCESS);
}
#endif

static PetscErrorCode cheb_err(int rc) {
  /* chebhip.h error codes -> PETSc classes (chebyshev.c:18,98,106,122 use PETSC_ERR_USER) */
  if (!rc) return PETSC_SUCCESS;
  switch (rc) {
    case CHEBHIP_ERR_SIZE: case CHEBHIP_ERR_TDIM: case CHEBHIP_ERR_DIMS:
      SETERRQ(PETSC_COMM_SELF, PETSC_ERR_USER, "%s", chebhip_last_error());
    case CHEBHIP_ERR_ARG:
      SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "%s", chebhip_last_error());
    case CHEBHIP_ERR_MEMORY:
      SETERRQ(PETSC_COMM_SELF, PETSC_ERR_MEM, "%s", chebhip_last_error());
    default:
      SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "%s", chebhip_last_error());
  }
}

/* ---- kernel level: MatCreateCheb / ChebMult / ChebDestroy (chebyshev.c:89-235) ---------------- */
PetscErrorCode MatCreateCheb(MPI_Comm comm, int rank, int tr, int *dims, unsigned flag,
                             Vec vx, Vec vy, Mat *A) {
  cheb_plan *plan;
  PetscInt   n;
  (void)flag; (void)vy;                       /* FFTW planner flag: ignored; Vecs are size prototypes */
  PetscFunctionBegin;
  PetscCall(VecGetSize(vx, &n));
  PetscCall(cheb_err(cheb_plan_create(rank, tr, dims, &plan)));
  {
    const long psize = cheb_plan_size(plan);   /* read before the plan is destroyed */
    if (psize != (long)n) {                    /* chebyshev.c:122 */
      cheb_plan_destroy(plan);
      SETERRQ(comm, PETSC_ERR_USER, "dimensions do not agree: n = %" PetscInt_FMT " but stride = %ld", n, psize);
    }
  }
  PetscCall(MatCreateShell(comm, n, n, n, n, plan, A));
  PetscCall(MatShellSetOperation(*A, MATOP_MULT, (void (*)(void))ChebMult));
  PetscCall(MatShellSetOperation(*A, MATOP_DESTROY, (void (*)(void))ChebDestroy));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode ChebMult(Mat A, Vec vx, Vec vy) {
  cheb_plan         *plan;
  const PetscScalar *x;
  PetscScalar       *y;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(A, &plan));
#if defined(PETSC_HAVE_HIP) && defined(CHEBHIP_USE_DEVICE_VECS)
  /* VECHIP vectors: no staging, asynchronous on PETSc's stream */
  {
    void *stream;
    PetscCall(cheb_petsc_stream(&stream));
    PetscCall(VecHIPGetArrayRead(vx, &x));
    PetscCall(VecHIPGetArrayWrite(vy, &y));
    PetscCall(cheb_err(cheb_apply(plan, x, y, stream)));
    PetscCall(VecHIPRestoreArrayWrite(vy, &y));
    PetscCall(VecHIPRestoreArrayRead(vx, &x));
  }
#else
  /* host Vecs, as the reference's VecGetArray (chebyshev.c:151-152): staged through HBM */
  PetscCall(VecGetArrayRead(vx, &x));
  PetscCall(VecGetArray(vy, &y));
  PetscCall(cheb_err(cheb_apply_host(plan, x, y)));
  PetscCall(VecRestoreArray(vy, &y));
  PetscCall(VecRestoreArrayRead(vx, &x));
#endif
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode ChebDestroy(Mat A) {
  cheb_plan *plan;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(A, &plan));
  cheb_plan_destroy(plan);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* rank-1 twins (chebyshev.c:8-85) */
PetscErrorCode MatCreateChebD1(MPI_Comm comm, Vec vx, Vec vy, unsigned flag, Mat *A) {
  PetscInt n; int dims[1];
  PetscFunctionBegin;
  PetscCall(VecGetSize(vx, &n));
  dims[0] = (int)n;
  PetscCall(MatCreateCheb(comm, 1, 0, dims, flag, vx, vy, A));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode ChebD1Mult(Mat A, Vec vx, Vec vy) { return ChebMult(A, vx, vy); }
PetscErrorCode ChebD1Destroy(Mat A) { return ChebDestroy(A); }

/* ---- operator level: what elliptic.C registers at :179 and :289 ------------------------------- */
/* In elliptic.C the MatElliptic ctx (elliptic.C:78-86) is replaced by one ell_op handle:
 *
 *   MatCreate_Elliptic(comm, d, dim, flag, bf, &vG, &A)   (elliptic.C:250-293)
 *     -> ell_op_create(d, dim, &op); VecCreateSeq(comm, ell_op_global_size(op), &vG);
 *        MatCreateShell(comm, n, n, n, n, op, &A);
 *        MatShellSetOperation(A, MATOP_MULT, MatMult_Elliptic_hip);
 *        MatShellSetOperation(A, MATOP_DESTROY, MatDestroy_Elliptic_hip);
 */
PetscErrorCode MatMult_Elliptic_hip(Mat A, Vec U, Vec V) {
  ell_op            *op;
  const PetscScalar *u;
  PetscScalar       *v;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(A, &op));
#if defined(PETSC_HAVE_HIP) && defined(CHEBHIP_USE_DEVICE_VECS)
  {
    void *stream;
    PetscCall(cheb_petsc_stream(&stream));
    PetscCall(VecHIPGetArrayRead(U, &u));
    PetscCall(VecHIPGetArrayWrite(V, &v));
    PetscCall(cheb_err(ell_op_mult(op, u, v, stream)));   /* the Krylov hot loop: nothing crosses PCIe */
    PetscCall(VecHIPRestoreArrayWrite(V, &v));
    PetscCall(VecHIPRestoreArrayRead(U, &u));
  }
#else
  PetscCall(VecGetArrayRead(U, &u));
  PetscCall(VecGetArray(V, &v));
  PetscCall(cheb_err(ell_op_mult_host(op, u, v)));        /* host Vecs: staged through HBM (plumbing, config 1) */
  PetscCall(VecRestoreArray(V, &v));
  PetscCall(VecRestoreArrayRead(U, &u));
#endif
  PetscFunctionReturn(PETSC_SUCCESS);
}

typedef struct { ell_op *op; Vec b; PetscReal gamma, exponent; } AppCtxHip;   /* AppCtx, elliptic.C:88-94 */

PetscErrorCode FormFunction_hip(SNES snes, Vec U, Vec rhs, void *void_ac) {
  AppCtxHip         *ac = (AppCtxHip *)void_ac;
  const PetscScalar *u, *b;
  PetscScalar       *r;
  (void)snes;
  PetscFunctionBegin;
#if defined(PETSC_HAVE_HIP) && defined(CHEBHIP_USE_DEVICE_VECS)
  {
    void *stream;
    PetscCall(cheb_petsc_stream(&stream));
    PetscCall(VecHIPGetArrayRead(U, &u));
    PetscCall(VecHIPGetArrayRead(ac->b, &b));
    PetscCall(VecHIPGetArrayWrite(rhs, &r));
    PetscCall(cheb_err(ell_op_function(ac->op, ac->gamma, ac->exponent, u, b, r, stream)));
    PetscCall(VecHIPRestoreArrayWrite(rhs, &r));
    PetscCall(VecHIPRestoreArrayRead(ac->b, &b));
    PetscCall(VecHIPRestoreArrayRead(U, &u));
  }
#else
  PetscCall(VecGetArrayRead(U, &u));
  PetscCall(VecGetArrayRead(ac->b, &b));
  PetscCall(VecGetArray(rhs, &r));
  PetscCall(cheb_err(ell_op_function_host(ac->op, ac->gamma, ac->exponent, u, b, r)));
  PetscCall(VecRestoreArray(rhs, &r));
  PetscCall(VecRestoreArrayRead(ac->b, &b));
  PetscCall(VecRestoreArrayRead(U, &u));
#endif
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MatDestroy_Elliptic_hip(Mat A) {
  ell_op *op;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(A, &op));
  ell_op_destroy(op);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- operator level: what stokes.C registers at :153, :309-325 --------------------------------- */
/* StokesCtx (stokes.C:40-65) keeps its PETSc side (KSPs, scatters between the global vector and its velocity /
 * pressure parts, MatVVPC); the Chebyshev work vectors, DP/DV plans, eta, deta and strain are replaced by one
 * stokes_op handle.  The callbacks below take HIP vectors (-vec_type hip): nothing crosses PCIe inside the Krylov
 * loops.  StokesCreate (:257-344) -> stokes_op_create(d, dim, &c->op); stokes_op_set_rheology / _set_dirichlet /
 * _set_force at the places where stokes.C fills options->rheology (:470-480), c->dirichlet (:2003) and c->force. */
#if defined(PETSC_HAVE_HIP) && defined(CHEBHIP_USE_DEVICE_VECS)
typedef struct { stokes_op *op; KSP KSPSchurVelocity; Vec vG0, vG1; } StokesCtxHip;

#define STOKES_SHELL(NAME, CALL)                                                            \
  PetscErrorCode NAME(Mat A, Vec xG, Vec yG) {                                              \
    StokesCtxHip *c; const PetscScalar *x; PetscScalar *y; void *stream;                    \
    PetscFunctionBegin;                                                                     \
    PetscCall(MatShellGetContext(A, &c));                                                   \
    PetscCall(cheb_petsc_stream(&stream));                                                  \
    PetscCall(VecHIPGetArrayRead(xG, &x)); PetscCall(VecHIPGetArrayWrite(yG, &y));          \
    PetscCall(cheb_err(CALL));                                                              \
    PetscCall(VecHIPRestoreArrayWrite(yG, &y)); PetscCall(VecHIPRestoreArrayRead(xG, &x));  \
    PetscFunctionReturn(PETSC_SUCCESS);                                                     \
  }
STOKES_SHELL(StokesMatMult_hip, stokes_op_mult(c->op, x, y, stream))        /* stokes.C:499-519 */
STOKES_SHELL(StokesMatMultVV_hip, stokes_op_mult_vv(c->op, x, y, stream))   /* stokes.C:623-676 */
STOKES_SHELL(StokesMatMultPV_hip, stokes_op_mult_pv(c->op, x, y, stream))   /* stokes.C:557-566 */
STOKES_SHELL(StokesMatMultVP_hip, stokes_op_mult_vp(c->op, x, y, stream))   /* stokes.C:599-619 */

/* the inner solve of the Schur complement stays the user's KSP (options prefix svel_, stokes.C:338-341) */
static int stokes_svel_solve(void *ctx, const double *rhs_dev, double *sol_dev, void *stream) {
  StokesCtxHip *c = (StokesCtxHip *)ctx;
  PetscErrorCode ierr;
  (void)stream;
  ierr = VecHIPPlaceArray(c->vG0, rhs_dev); if (ierr) return (int)ierr;
  ierr = VecHIPPlaceArray(c->vG1, sol_dev); if (ierr) return (int)ierr;
  ierr = KSPSolve(c->KSPSchurVelocity, c->vG0, c->vG1);                                     /* stokes.C:531 */
  VecHIPResetArray(c->vG0); VecHIPResetArray(c->vG1);
  return (int)ierr;
}
STOKES_SHELL(StokesMatMultSchur_hip, stokes_op_mult_schur(c->op, x, y, stokes_svel_solve, c, stream))   /* stokes.C:523-535 */

PetscErrorCode StokesFunction_hip(SNES snes, Vec xG, Vec yG, void *void_ctx) {               /* stokes.C:680-758 */
  StokesCtxHip *c = (StokesCtxHip *)void_ctx; const PetscScalar *x; PetscScalar *y; void *stream;
  (void)snes;
  PetscFunctionBegin;
  PetscCall(cheb_petsc_stream(&stream));
  PetscCall(VecHIPGetArrayRead(xG, &x)); PetscCall(VecHIPGetArrayWrite(yG, &y));
  PetscCall(cheb_err(stokes_op_function(c->op, x, y, stream)));
  PetscCall(VecHIPRestoreArrayWrite(yG, &y)); PetscCall(VecHIPRestoreArrayRead(xG, &x));
  PetscFunctionReturn(PETSC_SUCCESS);
}
/* ---- preconditioning: what elliptic.C:181-185 and stokes.C:159-187 register ---------------------------------- */
/* FormJacobian (elliptic.C:537-590) keeps its signature; instead of filling an AIJ matrix for PCILU it refreshes the
 * device stencil, and the PC of the SNES's KSP becomes a PCSHELL whose apply is the fast-diagonalisation solve:
 *   PCSetType(pc, PCSHELL); PCShellSetContext(pc, fdpc); PCShellSetApply(pc, PCApply_Elliptic_hip);      (for :184-185) */
PetscErrorCode FormJacobian_hip(SNES snes, Vec w, Mat A, Mat P, void *void_pc) {
  void *stream;
  (void)snes; (void)w; (void)A; (void)P;
  PetscFunctionBegin;
  PetscCall(cheb_petsc_stream(&stream));
  PetscCall(cheb_err(chebhip_fdpc_update((chebhip_fdpc *)void_pc, stream)));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode PCApply_Elliptic_hip(PC pc, Vec r, Vec z) {
  chebhip_fdpc *fd; const PetscScalar *rr; PetscScalar *zz; void *stream;
  PetscFunctionBegin;
  PetscCall(PCShellGetContext(pc, &fd));
  PetscCall(cheb_petsc_stream(&stream));
  PetscCall(VecHIPGetArrayRead(r, &rr)); PetscCall(VecHIPGetArrayWrite(z, &zz));
  PetscCall(cheb_err(chebhip_fdpc_apply(fd, rr, zz, stream)));
  PetscCall(VecHIPRestoreArrayWrite(z, &zz)); PetscCall(VecHIPRestoreArrayRead(r, &rr));
  PetscFunctionReturn(PETSC_SUCCESS);
}
/* StokesPCSetUp0 / StokesPCApply0..3 (stokes.C:1160-1241, 1714-1817): PCShellSetSetUp / PCShellSetApply on the outer
 * KSP's PC (stokes.C:159-187), context = a stokes_saddle made with stokes_saddle_create(c->op, ..) and
 * stokes_saddle_set_type(-pc_saddle_type).  The inner KSPs of the reference (vel_, schur_, svel_ options) are
 * replaced by the library's own solves: stokes_saddle_set_inner carries -vel_ksp_max_it / -schur_ksp_max_it. */
PetscErrorCode StokesPCSetUp_hip(PC pc) {
  stokes_saddle *s; void *stream;
  PetscFunctionBegin;
  PetscCall(PCShellGetContext(pc, &s));
  PetscCall(cheb_petsc_stream(&stream));
  PetscCall(cheb_err(stokes_saddle_setup(s, stream)));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode StokesPCApply_hip(PC pc, Vec x, Vec y) {
  stokes_saddle *s; const PetscScalar *xx; PetscScalar *yy; void *stream;
  PetscFunctionBegin;
  PetscCall(PCShellGetContext(pc, &s));
  PetscCall(cheb_petsc_stream(&stream));
  PetscCall(VecHIPGetArrayRead(x, &xx)); PetscCall(VecHIPGetArrayWrite(y, &yy));
  PetscCall(cheb_err(stokes_saddle_apply(s, xx, yy, stream)));
  PetscCall(VecHIPRestoreArrayWrite(y, &yy)); PetscCall(VecHIPRestoreArrayRead(x, &xx));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#endif /* device Vecs: the Stokes callbacks exist only for VECHIP vectors */
