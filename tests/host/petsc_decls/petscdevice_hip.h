/* petscdevice_hip.h -- NOT PETSc: see petscmat.h in this directory.  The device-Vec entry points the adapter uses
 * under -DPETSC_HAVE_HIP -DCHEBHIP_USE_DEVICE_VECS (PETSc >= 3.18 manual pages). */
#ifndef CHEBHIP_TEST_PETSC_DEVICE_DECLS_H
#define CHEBHIP_TEST_PETSC_DEVICE_DECLS_H
#include "petscmat.h"
typedef struct ihipStream_t *hipStream_t;
typedef struct _n_PetscDeviceContext *PetscDeviceContext;
PetscErrorCode PetscDeviceContextGetCurrentContext(PetscDeviceContext *);
PetscErrorCode PetscDeviceContextGetStreamHandle(PetscDeviceContext, void **);
PetscErrorCode VecHIPGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPGetArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPPlaceArray(Vec, const PetscScalar *);
PetscErrorCode VecHIPResetArray(Vec);
#endif
