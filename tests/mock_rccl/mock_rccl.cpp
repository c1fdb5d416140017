// mock_rccl.cpp — TEST DOUBLE for librccl (tests only; never part of the product).
//
// ks_reduce (include/ks_hip.h) talks to RCCL through seven entry points loaded with dlopen.  A development box
// has ONE GPU, and RCCL refuses two ranks on the same device, so the multi-rank code path of ks_reduce (count
// exchange, per-peer offsets, grouped send/recv, owner merge in source-rank order, sender reset) could never run
// before the driver's 8-GPU bench.  This library implements the same entry points between PROCESSES THAT SHARE
// ONE GPU: a message is a file under /dev/shm (written under a temporary name, then renamed: a receive polls for
// the name), device buffers are staged through the host after the caller's stream has been drained.  Point
// KS_RCCL_LIB at it and run one process per rank (tests/test_reduce_multiprocess_gpu.py).
//   g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o libmock_rccl.so mock_rccl.cpp -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct MockComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;  // RCCL: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ...
}

struct MockComm {
  std::string tag;
  int rank = 0, world = 1;
  std::vector<uint64_t> sent, received;  // message sequence numbers per peer
};

namespace {
struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; hipStream_t stream; MockComm* comm; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_size(int t) {
  switch (t) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 1; }
}
std::string msg_path(const MockComm* c, int src, int dst, uint64_t seq) {
  char b[256];
  snprintf(b, sizeof(b), "/dev/shm/mockrccl_%s_%d_%d_%llu", c->tag.c_str(), src, dst, (unsigned long long)seq);
  return b;
}
int do_send(const Op& o) {
  std::vector<char> h(o.bytes);
  if (o.bytes && hipMemcpy(h.data(), o.sbuf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const std::string p = msg_path(o.comm, o.comm->rank, o.peer, o.comm->sent[o.peer]++), tmp = p + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return 2;
  const size_t w = o.bytes ? fwrite(h.data(), 1, o.bytes, f) : 0;
  fclose(f);
  if (w != o.bytes || rename(tmp.c_str(), p.c_str()) != 0) return 2;
  return 0;
}
int do_recv(const Op& o) {
  const std::string p = msg_path(o.comm, o.peer, o.comm->rank, o.comm->received[o.peer]++);
  struct stat st;
  const auto t0 = std::chrono::steady_clock::now();
  while (stat(p.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 3;  // a peer died: do not hang the GPU box
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)st.st_size != o.bytes) return 4;  // both sides must agree on the message size, as with RCCL
  std::vector<char> h(o.bytes);
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return 2;
  const size_t r = o.bytes ? fread(h.data(), 1, o.bytes, f) : 0;
  fclose(f);
  unlink(p.c_str());
  if (r != o.bytes) return 2;
  if (o.bytes && hipMemcpy(o.rbuf, h.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}
int run_ops(std::vector<Op>& ops) {
  for (const Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;  // inputs are produced on the caller's stream
  for (const Op& o : ops)  // sends never block (files), so all sends first, then the receives
    if (o.send)
      if (int rc = do_send(o)) return rc;
  for (const Op& o : ops)
    if (!o.send)
      if (int rc = do_recv(o)) return rc;
  return 0;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "%d_%lld", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  MockComm* c = new MockComm();
  c->tag = id.internal;
  c->rank = rank;
  c->world = nranks;
  c->sent.assign(nranks, 0);
  c->received.assign(nranks, 0);
  *comm = c;
  return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return 0; }
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) { case 0: return "success"; case 1: return "mock: hip error"; case 2: return "mock: file error";
               case 3: return "mock: peer timed out"; case 4: return "mock: message size mismatch"; default: return "mock: error"; }
}
ncclResult_t ncclGroupStart() { ++g_depth; return 0; }
ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  g_ops.push_back({true, buf, nullptr, count * type_size(t), peer, stream, comm});
  if (g_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  g_ops.push_back({false, nullptr, buf, count * type_size(t), peer, stream, comm});
  if (g_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
  const size_t bytes = count * type_size(t);
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (hipMemcpy((char*)recvbuf + (size_t)comm->rank * bytes, sendbuf, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
  std::vector<Op> ops;
  for (int p = 0; p < comm->world; ++p) {
    if (p == comm->rank) continue;
    ops.push_back({true, sendbuf, nullptr, bytes, p, stream, comm});
    ops.push_back({false, nullptr, (char*)recvbuf + (size_t)p * bytes, bytes, p, stream, comm});
  }
  return run_ops(ops);
}
}
