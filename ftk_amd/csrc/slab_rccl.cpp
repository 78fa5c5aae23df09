// The slab pass's messages over RCCL (include/ftkx_slab.h: ftkx_slab_create_rccl): ncclAllGather of the ranks' contributions and grouped
// ncclSend / ncclRecv between neighbouring slabs, queued on the stream the protocol names (the context's stream; the masks of the first
// slice: the slab's side stream) -- nothing here waits on the host.  xGMI is point-to-point: a neighbour message is one send and one receive
// on one link, no ring.  Counterpart in the reference: the MPI calls of its distributed tracker (diy::mpi::gather at
// include/ftk/filters/critical_point_tracker.hh:689; the communicator kept on the filter, include/ftk/filters/filter.hh:47-61).
#include "ctx.hpp"
#include "../../include/ftkx_slab.h"

#include <rccl/rccl.h>

using namespace ftkxh;

namespace {

struct RcclSide { ncclComm_t comm, side_comm; ftkx_ctx *ctx; };

#define NCCL_TRY(call)                                                                                                   \
  do {                                                                                                                   \
    ncclResult_t r_ = (call);                                                                                            \
    if (r_ != ncclSuccess) return fail(nullptr, FTKX_E_DEVICE, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

int rccl_all_gather(void *user, const void *send, void *recv, size_t bytes, void *stream)
{
  RcclSide *R = (RcclSide *)user;
  NCCL_TRY(ncclAllGather(send, recv, bytes, ncclChar, R->comm, (hipStream_t)stream));
  return FTKX_OK;
}

int rccl_exchange(void *user, const void *send, size_t sb, int to, void *recv, size_t rb, int from, void *stream)
{
  RcclSide *R = (RcclSide *)user;
  // (traffic on any stream but the context's own goes over the second communicator where there is one: with a single communicator RCCL
  // runs the operations one after the other in the order they were issued, on whichever streams)
  ncclComm_t comm = (R->side_comm && R->ctx && (hipStream_t)stream != R->ctx->stream) ? R->side_comm : R->comm;
  NCCL_TRY(ncclGroupStart());
  if (to >= 0) NCCL_TRY(ncclSend(send, sb, ncclChar, to, comm, (hipStream_t)stream));
  if (from >= 0) NCCL_TRY(ncclRecv(recv, rb, ncclChar, from, comm, (hipStream_t)stream));
  NCCL_TRY(ncclGroupEnd());
  return FTKX_OK;
}

void rccl_destroy(void *user) { delete (RcclSide *)user; }

}  // namespace

extern "C" {

int ftkx_slab_transport_rccl(void *comm, void *side_comm, ftkx_slab_transport *out)
{
  if (!comm || !out) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_transport_rccl: null argument");
  memset(out, 0, sizeof(*out));
  out->user = new RcclSide{(ncclComm_t)comm, (ncclComm_t)side_comm, nullptr};
  out->all_gather = rccl_all_gather; out->exchange = rccl_exchange; out->queued = 1; out->destroy = rccl_destroy;
  return FTKX_OK;
}

int ftkx_slab_create_rccl(ftkx_ctx *ctx, int nt, int rank, int nranks, void *comm, void *side_comm, ftkx_slab **out)
{
  if (!ctx || !comm) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create_rccl: null argument");
  int crank = -1, csize = -1;
  NCCL_TRY(ncclCommUserRank((ncclComm_t)comm, &crank));
  NCCL_TRY(ncclCommCount((ncclComm_t)comm, &csize));
  if (crank != rank || csize != nranks) return fail(nullptr, FTKX_E_INVALID, "ftkx_slab_create_rccl: rank %d of %d, but the communicator says %d of %d", rank, nranks, crank, csize);
  ftkx_slab_transport tr;
  int rc = ftkx_slab_transport_rccl(comm, side_comm, &tr);
  if (rc) return rc;
  ((RcclSide *)tr.user)->ctx = ctx;
  rc = ftkx_slab_create(ctx, nt, rank, nranks, &tr, out);
  if (rc) rccl_destroy(tr.user);
  return rc;
}

int ftkx_rccl_unique_id(void *id128)
{
  if (!id128) return FTKX_E_INVALID;
  static_assert(sizeof(ncclUniqueId) == 128, "the id is handed around as 128 bytes");
  NCCL_TRY(ncclGetUniqueId((ncclUniqueId *)id128));
  return FTKX_OK;
}

int ftkx_rccl_comm_create(const void *id128, int rank, int nranks, int device, void **comm)
{
  if (!id128 || !comm) return FTKX_E_INVALID;
  HIP_TRY(nullptr, hipSetDevice(device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  NCCL_TRY(ncclCommInitRank(&c, nranks, id, rank));
  *comm = c;
  return FTKX_OK;
}

void ftkx_rccl_comm_destroy(void *comm) { if (comm) (void)ncclCommDestroy((ncclComm_t)comm); }

int ftkx_rccl_version(void)
{
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? v : -1;
}

}  // extern "C"
