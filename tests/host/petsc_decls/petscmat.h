/*
 * petscmat.h -- NOT PETSc.  Declarations of the handful of PETSc (>= 3.19) types, macros and prototypes that
 * adapter/chebyshev_petsc.c uses, written from the public PETSc manual pages, so that the adapter can be put through a
 * compiler's syntax and type checking in an image that has no PETSc (tests/test_adapter_syntax.py, -fsyntax-only:
 * nothing is linked, nothing runs).  It proves the binding is well-formed C against these signatures; it does not prove
 * it against a real PETSc build (SURVEY 8f.2 stays open).  Test infrastructure only.
 */
#ifndef CHEBHIP_TEST_PETSC_DECLS_H
#define CHEBHIP_TEST_PETSC_DECLS_H
#include <stddef.h>

#ifdef __cplusplus
#define PETSC_EXTERN_CXX_BEGIN extern "C" {
#define PETSC_EXTERN_CXX_END }
#else
#define PETSC_EXTERN_CXX_BEGIN
#define PETSC_EXTERN_CXX_END
#endif

typedef int PetscErrorCode;
typedef int PetscInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef int MPI_Comm;
typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_KSP *KSP;
typedef struct _p_SNES *SNES;
typedef struct _p_PC *PC;
typedef const char *PCType;
typedef enum { MATOP_MULT = 3, MATOP_DESTROY = 60 } MatOperation;

#define PETSC_SUCCESS 0
#define PETSC_ERR_MEM 55
#define PETSC_ERR_LIB 76
#define PETSC_ERR_USER 83
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_COMM_SELF ((MPI_Comm)1)
#define PetscInt_FMT "d"
#define PCSHELL "shell"

PetscErrorCode PetscError(MPI_Comm, int, const char *, const char *, PetscErrorCode, int, const char *, ...);
#define SETERRQ(comm, ierr, ...) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, 0, __VA_ARGS__)
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(v) return (v)
#define PetscCall(...) do { PetscErrorCode ierr_q_ = (__VA_ARGS__); if (ierr_q_) return ierr_q_; } while (0)

PetscErrorCode VecGetSize(Vec, PetscInt *);
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecCreateSeq(MPI_Comm, PetscInt, Vec *);
PetscErrorCode MatCreateShell(MPI_Comm, PetscInt, PetscInt, PetscInt, PetscInt, void *, Mat *);
PetscErrorCode MatShellSetOperation(Mat, MatOperation, void (*)(void));
PetscErrorCode MatShellGetContext(Mat, void *);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode KSPSolve(KSP, Vec, Vec);
PetscErrorCode PCSetType(PC, PCType);
PetscErrorCode PCShellSetApply(PC, PetscErrorCode (*)(PC, Vec, Vec));
PetscErrorCode PCShellSetSetUp(PC, PetscErrorCode (*)(PC));
PetscErrorCode PCShellSetContext(PC, void *);
PetscErrorCode PCShellGetContext(PC, void *);
#endif
