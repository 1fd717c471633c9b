// The collectives the sharded paths use, behind one interface with two backends:
//   rccl     : one process per GPU, RCCL over xGMI (production).
//   loopback : the ranks are THREADS of one process, each with its own handle and stream on the same GPU; a collective
//              is device-to-device copies (plus a tiny reduce kernel) between host barriers.  It exists so that the
//              multi-rank code paths -- unequal column slabs, grouped row-block broadcasts, halo send/recv lists, the
//              sharded kNN list all-gather, speculative iteration + collective ordering -- execute at world > 1 on a
//              single MI355X (RCCL refuses two ranks on one device).  Slow by construction; never the default.
// The backend is chosen by the 128-byte id handed to osc_comm_init: osc_comm_unique_id makes an RCCL id,
// osc_comm_loopback_id a loopback one.
#pragma once
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <vector>

#include "common.hpp"
#include "loop_group.hpp"  // CommError, CommXfer, the loopback backend's host barrier

namespace osc {

enum CommDType { COMM_F32 = 0, COMM_F64 = 1, COMM_I32 = 2 };
enum CommOp { COMM_SUM = 0, COMM_MAX = 1 };

class Comm {
 public:
  virtual ~Comm() = default;
  virtual const char* kind() const = 0;
  int rank() const { return rank_; }
  int world() const { return world_; }
  // in place on every rank; n elements of type t; all ranks end with bit-identical results
  virtual void allreduce(void* buf, size_t n, CommDType t, CommOp op, hipStream_t s) = 0;
  // in place: rank r's contribution already sits at buf + r * chunk_bytes
  virtual void allgather(void* buf, size_t chunk_bytes, hipStream_t s) = 0;
  // grouped in-place broadcasts: piece i (same size on every rank) is owned by rank pieces[i].peer
  virtual void broadcast_group(const std::vector<CommXfer>& pieces, hipStream_t s) = 0;
  // grouped point-to-point: the m-th send of rank a to rank b matches the m-th recv of rank b from rank a
  virtual void exchange(const std::vector<CommXfer>& sends, const std::vector<CommXfer>& recvs, hipStream_t s) = 0;

 protected:
  int rank_ = 0, world_ = 1;
};

void comm_rccl_id(char id_out[128]);      // throws CommError
void comm_loopback_id(char id_out[128]);  // process-local group key
int comm_rccl_version();                  // ncclGetVersion's code (e.g. 22105), 0 if the call fails
std::unique_ptr<Comm> comm_create(const char id[128], int rank, int world, int device);

}  // namespace osc
