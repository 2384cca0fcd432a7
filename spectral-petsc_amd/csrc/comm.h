// comm.h -- transport among the G ranks of a slab-partitioned operator, shared by dist.hip and slabx.hip.
// One exchange = a list of segments; every rank calls comm_exchange with its own list, collectively.
#pragma once
#include "../../include/chebhip.h"
#include <hip/hip_runtime.h>

namespace chebhip {

// One segment of an exchange: `nsend` doubles at `send` go to `peer`, `nrecv` doubles from `peer` land at `recv`.
// The k-th segment a rank addresses to peer s meets the k-th segment s addresses to that rank (nsend there = nrecv here).
struct XSeg { int peer; const double *send; long nsend; double *recv; long nrecv; };

// All segments of one exchange, ordered on `st`.  Segments whose peer is the calling rank are device copies.
int comm_exchange(chebhip_comm *c, const XSeg *segs, int nseg, hipStream_t st);
// A rank that fails between two exchanges of a collective call releases the thread ranks waiting for it (LOCAL transport:
// their barriers fail at once instead of after local_timeout_s); nothing to do for the other transports.
void comm_abort(chebhip_comm *c);
int comm_size(const chebhip_comm *c);
int comm_rank(const chebhip_comm *c);

}  // namespace chebhip
