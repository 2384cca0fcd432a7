/*
 * chebyshev.h -- replacement for the reference's chebyshev.h (chebyshev.h:1-38) when linking
 * elliptic.C / stokes.C / cheb.c / poisson.c against libchebhip.so instead of FFTW.
 *
 * Same six symbols and signatures (chebyshev.h:27-34) and the PI macro (:10) the drivers use
 * (cheb.c:69, poisson.c:86, elliptic.C:622).  The context structs of chebyshev.h:12-24 exposed FFTW
 * types, but no caller touches a field (callers only hold the Mat), so they are opaque here.
 * The drivers also reference FFTW_ESTIMATE and fftw_import_system_wisdom() (elliptic.C:135,159;
 * stokes.C:138,259; cheb.c:32,48,58; poisson.c:46,66-67): provided below as a macro and a no-op so the
 * unmodified sources compile without <fftw3.h>.  `flag` arguments are accepted and ignored.
 *
 * Built only where a PETSc exists (none does in the build image of this repo); see INTEGRATION.md.
 */
#ifndef CHEBYSHEV_H
#define CHEBYSHEV_H

#include <petscmat.h>

#ifndef FFTW_ESTIMATE
#define FFTW_ESTIMATE (1U << 6)
#define FFTW_MEASURE (0U)
static inline int fftw_import_system_wisdom(void) { return 0; }
#endif

PETSC_EXTERN_CXX_BEGIN

#define PI 3.14159265358979323846

PetscErrorCode MatCreateChebD1(MPI_Comm comm, Vec vx, Vec vy, unsigned flag, Mat *A);
PetscErrorCode ChebD1Mult(Mat A, Vec vx, Vec vy);
PetscErrorCode ChebD1Destroy(Mat A);

PetscErrorCode MatCreateCheb(MPI_Comm comm, int rank, int tr, int *dims, unsigned flag,
                             Vec vx, Vec vy, Mat *A);
PetscErrorCode ChebMult(Mat A, Vec vx, Vec vy);
PetscErrorCode ChebDestroy(Mat A);

PETSC_EXTERN_CXX_END

#endif
