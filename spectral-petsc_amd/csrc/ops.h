// ops.h -- what the preconditioner module (precond.hip) needs to see of the operator handles.
#pragma once
#include "../../include/chebhip.h"

namespace chebhip {

// Grid and coefficient state of an operator: everything FormJacobian (elliptic.C:537-590) / StokesPCSetUp0
// (stokes.C:1160-1241) read.  Pointers are device pointers owned by the operator; gradu may be null (Stokes).
struct FdView {
  int d = 0;
  const int *dims = nullptr;      // host, d extents of the local grid (boundary included)
  long N = 0, G = 0;              // local nodes, interior nodes
  const int *ixL = nullptr;       // device [N]: interior index or -1
  const double *eta = nullptr, *deta = nullptr;     // device [N]
  const double *gradu[10] = {nullptr};              // device [N] each, or null: no deta * du0 terms
};

}  // namespace chebhip

int ell_op_sync_coeffs(ell_op *op, void *stream);            // chebhip.hip: makes eta / deta current before the view's arrays are read
int ell_op_fd_view(ell_op *op, chebhip::FdView *v);          // chebhip.hip (allocates the coefficient state if needed)
// the same for slab-mode handles too (precond.hip's slab mode; *gP0: global extent of dimension 0, v->dims[0]: planes of the slab)
int ell_op_fd_view_any(ell_op *op, chebhip::FdView *v, int *gP0);
int stokes_op_fd_view_any(stokes_op *op, chebhip::FdView *v, int *gP0);
int stokes_op_fd_view(stokes_op *op, chebhip::FdView *v);    // stokes.hip
int chebhip_fail(int code, const char *fmt, ...);            // chebhip.hip
