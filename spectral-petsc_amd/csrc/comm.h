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

// ---- direct transports: the ranks' buffers are addressable from every rank's kernels --------------------------------------------
// LOCAL (ranks are threads of one process, peer access between their devices) and NULL (timing one rank: every "peer" is the rank
// itself).  There an exchange needs no messages at all: a rank's kernels read the peers' arrays in place, ordered by events.
constexpr int COMM_MAXR = 64, COMM_NPTR = 2, COMM_NEV = 4;
struct PeerView { const double *ptr[COMM_MAXR][COMM_NPTR]; };
bool comm_direct(const chebhip_comm *c);
bool comm_is_null(const chebhip_comm *c);            // the timing transport: every "peer" is the rank itself
bool comm_overlap_pays(const chebhip_comm *c);       // false when nothing travels off the device (one rank, NULL, LOCAL ranks sharing one device)
int comm_group_barrier(chebhip_comm *c);             // LOCAL: host barrier of the rank threads (0 at once for the other kinds)
// Collective rendezvous.  Every rank posts `n` (<= COMM_NPTR) pointers, records its event `slot` (0 .. COMM_NEV-1; the group owns the
// events) at this point of `st`, and when all ranks have done so gets everybody's pointers.  On return the caller's stream has been
// made to wait for the event `slot` of every other rank (their data is complete where they posted it) and, when wait_slot >= 0, for
// their events `wait_slot` of an EARLIER rendezvous (their kernels that read this rank's arrays have finished).
int comm_rendezvous(chebhip_comm *c, const double *const *ptrs, int n, int slot, int wait_slot, hipStream_t st, PeerView *out);
// records the rank's event `slot` on st without a rendezvous (e.g. "my kernels that read the peers' arrays end here"): the peers
// pick it up through wait_slot of a LATER rendezvous
int comm_mark(chebhip_comm *c, int slot, hipStream_t st);

}  // namespace chebhip
