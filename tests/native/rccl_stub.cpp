// Stand-in for librccl.so in the single-GPU test of zkwg_calculate_batch_multi's n_dev > 1 branch
// (tests/test_multi.py): the six entry points the result-table gather uses, implemented with hipMemcpyAsync
// on the streams the caller passes.  A send/recv pair of one group is matched at ncclGroupEnd; the receiving
// stream waits for an event recorded on the sending stream, like a point-to-point over xGMI would order them.
// Unlike RCCL it accepts several communicators on one device (devices = [0, 0]).  Test infrastructure only.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

struct ncclComm { int rank, nranks, device; };
struct Op { bool send; int rank, peer; const void* src; void* dst; size_t bytes; hipStream_t st; };
static std::mutex g_mu;
static std::vector<Op> g_ops;
static int g_depth = 0;
static unsigned long g_groups = 0, g_pairs = 0;

extern "C" {
int ncclCommInitAll(ncclComm** comms, int n, const int* devs) {
  for (int i = 0; i < n; ++i) comms[i] = new ncclComm{i, n, devs ? devs[i] : i};
  return 0;
}
int ncclCommDestroy(ncclComm* c) { delete c; return 0; }
int ncclGroupStart() { std::lock_guard<std::mutex> l(g_mu); ++g_depth; return 0; }
static int elem_bytes(int dtype) { return (dtype == 0 || dtype == 1) ? 1 : (dtype == 2 || dtype == 3 || dtype == 7) ? 4 : 8; }
int ncclSend(const void* src, size_t count, int dtype, int peer, ncclComm* c, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return 5;   // ncclInvalidUsage: this stand-in only supports grouped point-to-point
  g_ops.push_back(Op{true, c->rank, peer, src, nullptr, count * elem_bytes(dtype), st});
  return 0;
}
int ncclRecv(void* dst, size_t count, int dtype, int peer, ncclComm* c, hipStream_t st) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return 5;
  g_ops.push_back(Op{false, c->rank, peer, nullptr, dst, count * elem_bytes(dtype), st});
  return 0;
}
int ncclGroupEnd() {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return 5;
  if (--g_depth) return 0;
  ++g_groups;
  int rc = 0;
  std::vector<char> used(g_ops.size(), 0);
  for (size_t i = 0; i < g_ops.size(); ++i) {
    if (g_ops[i].send) continue;
    const Op& r = g_ops[i];
    size_t j = 0;
    for (; j < g_ops.size(); ++j)
      if (!used[j] && g_ops[j].send && g_ops[j].rank == r.peer && g_ops[j].peer == r.rank) break;
    if (j == g_ops.size() || g_ops[j].bytes != r.bytes) { rc = 5; continue; }   // unmatched receive / size mismatch
    used[j] = used[i] = 1;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = 1; continue; }
    if (hipEventRecord(ev, g_ops[j].st) != hipSuccess || hipStreamWaitEvent(r.st, ev, 0) != hipSuccess ||
        hipMemcpyAsync(r.dst, g_ops[j].src, r.bytes, hipMemcpyDeviceToDevice, r.st) != hipSuccess) rc = 1;
    // the sender's stream must not run ahead of the copy that reads its buffer
    hipEvent_t done;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(done, r.st) != hipSuccess || hipStreamWaitEvent(g_ops[j].st, done, 0) != hipSuccess) rc = 1;
      hipEventDestroy(done);
    }
    hipEventDestroy(ev);
    ++g_pairs;
  }
  for (size_t i = 0; i < g_ops.size(); ++i) if (!used[i]) rc = rc ? rc : 5;   // a send nobody receives
  g_ops.clear();
  return rc;
}
// test hooks
unsigned long zk_stub_groups(void) { return g_groups; }
unsigned long zk_stub_pairs(void) { return g_pairs; }
}
