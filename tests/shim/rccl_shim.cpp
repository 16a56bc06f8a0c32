// tests/shim/rccl_shim.cpp -- TEST INFRASTRUCTURE, never shipped: the nine nccl* entry points liblfx.so binds
// (lidar_feature_extraction_amd/csrc/lfx_gather.hip: rccl()), implemented for several PROCESSES THAT SHARE ONE GPU.
//
// A real RCCL communicator refuses two ranks on one device, and a GPU box of this pool has one device, so the N > 1
// branch of lfx_gather_payload (grouped ncclSend / ncclRecv with per-rank offset arithmetic) could never execute there.
// With LFX_RCCL_LIB pointing at this library it does: every transfer is staged through a file under /dev/shm
// (device -> host -> file, rename to publish; the peer polls for the name, host -> device), messages between a pair of
// ranks are matched in posting order like NCCL's, and a group's operations are carried out at ncclGroupEnd with all
// sends first, so that two ranks exchanging in one group cannot wait for each other.  Semantics that differ from
// RCCL and do not matter to the caller under test: the calls complete synchronously (the stream is drained first).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace
{
struct Comm
{
  std::string dir;
  int rank = 0, world = 1;
  std::vector<uint64_t> sent, received;     // per peer: messages posted so far (the matching order)
  uint64_t collectives = 0;
};

struct Op
{
  bool send;
  const void * src;
  void * dst;
  size_t bytes;
  int peer;
  Comm * comm;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
constexpr int kTimeoutSeconds = 30;

size_t type_bytes(ncclDataType_t t)
{
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

bool publish(const std::string & path, const void * host, size_t bytes)
{
  const std::string tmp = path + ".part";
  FILE * f = std::fopen(tmp.c_str(), "wb");
  if (!f) {return false;}
  const bool ok = bytes == 0 || std::fwrite(host, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

bool take(const std::string & path, void * host, size_t bytes, bool remove_it)
{
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kTimeoutSeconds)) {return false;}
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  FILE * f = std::fopen(path.c_str(), "rb");
  if (!f) {return false;}
  const bool ok = bytes == 0 || std::fread(host, 1, bytes, f) == bytes;
  std::fclose(f);
  if (remove_it) {std::remove(path.c_str());}
  return ok;
}

ncclResult_t run_send(const Op & o)
{
  std::vector<uint8_t> host(o.bytes);
  if (hipStreamSynchronize(o.stream) != hipSuccess) {return ncclUnhandledCudaError;}
  if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) {return ncclUnhandledCudaError;}
  const std::string path = o.comm->dir + "/p2p_" + std::to_string(o.comm->rank) + "_" + std::to_string(o.peer) + "_" +
    std::to_string(o.comm->sent[o.peer]++);
  return publish(path, host.data(), o.bytes) ? ncclSuccess : ncclSystemError;
}

ncclResult_t run_recv(const Op & o)
{
  std::vector<uint8_t> host(o.bytes);
  const std::string path = o.comm->dir + "/p2p_" + std::to_string(o.peer) + "_" + std::to_string(o.comm->rank) + "_" +
    std::to_string(o.comm->received[o.peer]++);
  if (!take(path, host.data(), o.bytes, true)) {return ncclSystemError;}
  if (hipStreamSynchronize(o.stream) != hipSuccess) {return ncclUnhandledCudaError;}
  if (o.bytes && hipMemcpy(o.dst, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) {return ncclUnhandledCudaError;}
  return ncclSuccess;
}

ncclResult_t flush_ops()
{
  std::vector<Op> ops;
  ops.swap(g_ops);
  for (const Op & o : ops) {
    if (o.send) {const ncclResult_t r = run_send(o); if (r != ncclSuccess) {return r;}}
  }
  for (const Op & o : ops) {
    if (!o.send) {const ncclResult_t r = run_recv(o); if (r != ncclSuccess) {return r;}}
  }
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId * id)
{
  if (!id) {return ncclInvalidArgument;}
  std::memset(id, 0, sizeof(*id));
  std::random_device rd;
  std::snprintf(id->internal, sizeof(id->internal), "lfxshim_%08x%08x_%d", rd(), rd(), (int)getpid());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t * comm, int nranks, ncclUniqueId id, int rank)
{
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) {return ncclInvalidArgument;}
  id.internal[sizeof(id.internal) - 1] = 0;
  Comm * c = new Comm();
  c->dir = std::string("/dev/shm/") + id.internal;
  c->rank = rank; c->world = nranks;
  c->sent.assign(nranks, 0); c->received.assign(nranks, 0);
  mkdir(c->dir.c_str(), 0700);
  // rendezvous: every rank says it is here and waits for the others
  if (!publish(c->dir + "/here_" + std::to_string(rank), "", 0)) {delete c; return ncclSystemError;}
  for (int k = 0; k < nranks; k++) {
    if (!take(c->dir + "/here_" + std::to_string(k), nullptr, 0, false)) {delete c; return ncclSystemError;}
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  Comm * c = reinterpret_cast<Comm *>(comm);
  if (!c) {return ncclSuccess;}
  // Nobody removes a file a peer may still be waiting for: a rank says it is leaving and waits (briefly) for the others to
  // say so too -- a rank that finished its last all-gather may be a whole step ahead of one still polling for that file.
  // What is left (the leave markers, the directory) is a few empty files; the tests remove /dev/shm/lfxshim_* afterwards.
  (void)publish(c->dir + "/bye_" + std::to_string(c->rank), "", 0);
  bool all_left = true;
  for (int k = 0; k < c->world && all_left; k++) {all_left = take(c->dir + "/bye_" + std::to_string(k), nullptr, 0, false);}
  if (all_left) {
    std::remove((c->dir + "/here_" + std::to_string(c->rank)).c_str());
    if (c->collectives > 0) {std::remove((c->dir + "/ag_" + std::to_string(c->collectives - 1) + "_" + std::to_string(c->rank)).c_str());}
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void * sendbuff, void * recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
  Comm * c = reinterpret_cast<Comm *>(comm);
  if (!c || !sendbuff || !recvbuff) {return ncclInvalidArgument;}
  const size_t bytes = sendcount * type_bytes(datatype);
  std::vector<uint8_t> host(bytes);
  if (hipStreamSynchronize(stream) != hipSuccess) {return ncclUnhandledCudaError;}
  if (bytes && hipMemcpy(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) {return ncclUnhandledCudaError;}
  const uint64_t seq = c->collectives++;
  const std::string stem = c->dir + "/ag_" + std::to_string(seq) + "_";
  if (!publish(stem + std::to_string(c->rank), host.data(), bytes)) {return ncclSystemError;}
  for (int k = 0; k < c->world; k++) {
    if (!take(stem + std::to_string(k), host.data(), bytes, false)) {return ncclSystemError;}
    if (bytes && hipMemcpy(static_cast<uint8_t *>(recvbuff) + (size_t)k * bytes, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
      return ncclUnhandledCudaError;
    }
  }
  // everyone has read collective seq - 1 by the time anyone publishes seq + 1: its files can go
  if (seq > 0) {std::remove((c->dir + "/ag_" + std::to_string(seq - 1) + "_" + std::to_string(c->rank)).c_str());}
  return ncclSuccess;
}

ncclResult_t ncclSend(const void * sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
  Comm * c = reinterpret_cast<Comm *>(comm);
  if (!c || peer < 0 || peer >= c->world || (count && !sendbuff)) {return ncclInvalidArgument;}
  g_ops.push_back({true, sendbuff, nullptr, count * type_bytes(datatype), peer, c, stream});
  return g_depth > 0 ? ncclSuccess : flush_ops();
}

ncclResult_t ncclRecv(void * recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
  Comm * c = reinterpret_cast<Comm *>(comm);
  if (!c || peer < 0 || peer >= c->world || (count && !recvbuff)) {return ncclInvalidArgument;}
  g_ops.push_back({false, nullptr, recvbuff, count * type_bytes(datatype), peer, c, stream});
  return g_depth > 0 ? ncclSuccess : flush_ops();
}

ncclResult_t ncclGroupStart()
{
  g_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (g_depth > 0) {g_depth--;}
  return g_depth == 0 ? flush_ops() : ncclSuccess;
}

const char * ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "shim: a HIP call failed";
    case ncclSystemError: return "shim: a staging file could not be written or did not arrive in time";
    case ncclInvalidArgument: return "shim: invalid argument";
    default: return "shim: error";
  }
}

}  // extern "C"
