// TEST INFRASTRUCTURE -- a stand-in for librccl.so that lets TWO ranks share ONE GPU.
//
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the builder's boxes have one GPU: the library's native
// exchange (csrc/comm.hip: pseg_comm_init / pseg_allreduce_bucket / pseg_reduce_scatter_bucket / pseg_all_gather_bucket, the
// path PSEG_NATIVE_ALLREDUCE=1 selects in utils/dist.py for the reference's DistributedDataParallel exchange, reference
// train.py:33-35,112-117) had only ever run with one rank, where every collective is a copy.  This file exports the eight nccl*
// symbols comm.hip binds, with N-rank SEMANTICS and stream ORDERING, over a POSIX shared-memory segment:
//
//   collective on stream S = per chunk: hipMemcpyAsync(device -> pinned send buffer, S); hipLaunchHostFunc(S): publish the
//   chunk in this rank's shm slot, wait until every rank has published the same operation, combine (sum in rank order / pick
//   the slices) into the pinned receive buffer; hipMemcpyAsync(pinned receive buffer -> device, S).
//
// Everything is ordered by S exactly as a real collective is (the caller's side stream waits for the bucket's events before,
// the compute stream joins after); a wrong in-place offset, a missing remainder all-reduce, or a collective enqueued before its
// gradients are complete gives wrong sums here just as it would over xGMI.  What it does NOT test: RCCL itself, bandwidth.
// tests/test_dist_gpu.py builds it with hipcc and points PSEG_RCCL_PATH at it.
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kChunkFloats = 1u << 20;            // 4 MiB per chunk and rank
constexpr double kTimeoutS = 120.0;

struct Shared {
  volatile int joined;
  volatile int poisoned;
  volatile uint64_t seq[kMaxRanks];                  // operations (chunks) rank r has published so far
  // then: float slot[2][kMaxRanks][kChunkFloats]
};

struct Comm {
  Shared* sh;
  float* slots;
  size_t map_bytes;
  int nranks, rank;
  float* pin_send;        // pinned staging, one chunk per in-flight operation (ring)
  float* pin_recv;
  uint64_t issued;        // chunks enqueued so far (host side)
  char name[96];
};

constexpr int kRing = 64;     // staging chunks in flight per communicator (a 32 MiB bucket is 8 chunks)

struct Op {
  Comm* c;
  uint64_t n;             // global chunk index of this communicator
  int kind;               // 0 all-reduce, 1 reduce-scatter, 2 all-gather
  size_t count;           // floats this rank contributes in this chunk
  size_t out_count;       // floats this rank receives
};

struct ncclUniqueId {
  char internal[128];
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

float* slot_of(Comm* c, uint64_t n, int r) { return c->slots + ((n & 1) * kMaxRanks + r) * kChunkFloats; }

void host_step(void* user) {
  Op* op = (Op*)user;
  Comm* c = op->c;
  Shared* sh = c->sh;
  const uint64_t n = op->n;
  const float* mine = c->pin_send + (n % kRing) * kChunkFloats;
  float* out = c->pin_recv + (n % kRing) * kChunkFloats;
  memcpy(slot_of(c, n, c->rank), mine, op->count * sizeof(float));
  __atomic_store_n(&sh->seq[c->rank], n + 1, __ATOMIC_RELEASE);
  const double t0 = now_s();
  for (int r = 0; r < c->nranks; ++r)
    while (__atomic_load_n(&sh->seq[r], __ATOMIC_ACQUIRE) < n + 1) {
      if (sh->poisoned || now_s() - t0 > kTimeoutS) {
        if (!sh->poisoned) fprintf(stderr, "[standin-rccl] rank %d: peer %d never reached operation %llu\n", c->rank, r, (unsigned long long)n);
        sh->poisoned = 1;
        delete op;
        return;
      }
      sched_yield();
    }
  if (op->kind == 0) {                       // sum over the ranks, in rank order (every rank adds in the same order)
    for (size_t i = 0; i < op->count; ++i) {
      float s = slot_of(c, n, 0)[i];
      for (int r = 1; r < c->nranks; ++r) s += slot_of(c, n, r)[i];
      out[i] = s;
    }
  } else if (op->kind == 1) {                // every rank published nranks * out_count floats; mine is slice `rank` of the sum
    const size_t o = (size_t)c->rank * op->out_count;
    for (size_t i = 0; i < op->out_count; ++i) {
      float s = slot_of(c, n, 0)[o + i];
      for (int r = 1; r < c->nranks; ++r) s += slot_of(c, n, r)[o + i];
      out[i] = s;
    }
  } else {                                   // all-gather: rank r's `count` floats become slice r
    for (int r = 0; r < c->nranks; ++r) memcpy(out + (size_t)r * op->count, slot_of(c, n, r), op->count * sizeof(float));
  }
  delete op;
}

}  // namespace

extern "C" {

int ncclGetVersion(int* v) {
  *v = 99900;      // (not a real RCCL version: run records show at a glance that the stand-in was bound)
  return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "stand-in rccl error (tests/standin_rccl.cpp)"; }

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/pseg_standin_%d_%lld", (int)getpid(), (long long)(now_s() * 1e6));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
  Comm* c = new Comm();
  memset(c, 0, sizeof(*c));
  c->nranks = nranks;
  c->rank = rank;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = 4096 + sizeof(float) * 2 * kMaxRanks * kChunkFloats;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    fprintf(stderr, "[standin-rccl] shm_open / ftruncate %s: %s\n", c->name, strerror(errno));
    return 2;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  c->sh = (Shared*)p;
  c->slots = (float*)((char*)p + 4096);
  if (hipHostMalloc((void**)&c->pin_send, sizeof(float) * kRing * kChunkFloats, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->pin_recv, sizeof(float) * kRing * kChunkFloats, hipHostMallocDefault) != hipSuccess)
    return 1;
  __atomic_add_fetch(&c->sh->joined, 1, __ATOMIC_ACQ_REL);
  const double t0 = now_s();
  while (__atomic_load_n(&c->sh->joined, __ATOMIC_ACQUIRE) < nranks) {
    if (now_s() - t0 > kTimeoutS) return 6;
    sched_yield();
  }
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (c == nullptr) return 0;
  (void)hipDeviceSynchronize();       // every enqueued host step has run
  const int gone = __atomic_add_fetch(&c->sh->joined, -1, __ATOMIC_ACQ_REL);
  if (gone == 0) shm_unlink(c->name);
  (void)hipHostFree(c->pin_send);
  (void)hipHostFree(c->pin_recv);
  munmap((void*)c->sh, c->map_bytes);
  delete c;
  return 0;
}

// one chunk: `count` floats in from `send`, `out_count` floats out to `recv`
static int enqueue(Comm* c, int kind, const float* send, size_t count, float* recv, size_t out_count, hipStream_t st) {
  if (count > kChunkFloats || out_count > kChunkFloats) return 4;
  // the staging ring: chunk n reuses the buffers of chunk n - kRing, whose copies must have completed -- drain the stream then
  // (never in the tests' sizes: 64 chunks = 256 MiB in flight)
  if (c->issued >= (uint64_t)kRing && (c->issued % kRing) == 0 && hipStreamSynchronize(st) != hipSuccess) return 1;
  const uint64_t n = c->issued++;
  float* ps = c->pin_send + (n % kRing) * kChunkFloats;
  float* pr = c->pin_recv + (n % kRing) * kChunkFloats;
  if (hipMemcpyAsync(ps, send, count * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
  Op* op = new Op{c, n, kind, count, out_count};
  if (hipLaunchHostFunc(st, host_step, op) != hipSuccess) return 1;
  if (hipMemcpyAsync(recv, pr, out_count * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st) {
  if (dtype != 7 || op != 0 || comm == nullptr) return 4;
  Comm* c = (Comm*)comm;
  for (size_t o = 0; o < count; o += kChunkFloats) {
    const size_t n = count - o < kChunkFloats ? count - o : kChunkFloats;
    const int rc = enqueue(c, 0, (const float*)send + o, n, (float*)recv + o, n, st);
    if (rc) return rc;
  }
  return 0;
}

// send: nranks * recvcount floats; recv: this rank's slice of the sum
int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t st) {
  if (dtype != 7 || op != 0 || comm == nullptr) return 4;
  Comm* c = (Comm*)comm;
  // chunked over the slice: chunk j carries [j, j + m) of EVERY rank's slice, packed rank-major in the staging buffer
  const size_t per = kChunkFloats / c->nranks;
  for (size_t o = 0; o < recvcount; o += per) {
    const size_t m = recvcount - o < per ? recvcount - o : per;
    if (c->issued >= (uint64_t)kRing && (c->issued % kRing) == 0 && hipStreamSynchronize(st) != hipSuccess) return 1;
    const uint64_t n = c->issued++;
    float* ps = c->pin_send + (n % kRing) * kChunkFloats;
    float* pr = c->pin_recv + (n % kRing) * kChunkFloats;
    for (int r = 0; r < c->nranks; ++r)
      if (hipMemcpyAsync(ps + (size_t)r * m, (const float*)send + (size_t)r * recvcount + o, m * sizeof(float), hipMemcpyDeviceToHost,
                         st) != hipSuccess)
        return 1;
    Op* opp = new Op{c, n, 1, (size_t)c->nranks * m, m};
    if (hipLaunchHostFunc(st, host_step, opp) != hipSuccess) return 1;
    if (hipMemcpyAsync((float*)recv + o, pr, m * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
  }
  return 0;
}

// send: sendcount floats; recv: nranks * sendcount, rank r's contribution at r * sendcount
int ncclAllGather(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t st) {
  if (dtype != 7 || comm == nullptr) return 4;
  Comm* c = (Comm*)comm;
  const size_t per = kChunkFloats / c->nranks;
  for (size_t o = 0; o < sendcount; o += per) {
    const size_t m = sendcount - o < per ? sendcount - o : per;
    if (c->issued >= (uint64_t)kRing && (c->issued % kRing) == 0 && hipStreamSynchronize(st) != hipSuccess) return 1;
    const uint64_t n = c->issued++;
    float* ps = c->pin_send + (n % kRing) * kChunkFloats;
    float* pr = c->pin_recv + (n % kRing) * kChunkFloats;
    if (hipMemcpyAsync(ps, (const float*)send + o, m * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    Op* opp = new Op{c, n, 2, m, (size_t)c->nranks * m};
    if (hipLaunchHostFunc(st, host_step, opp) != hipSuccess) return 1;
    for (int r = 0; r < c->nranks; ++r)
      if (hipMemcpyAsync((float*)recv + (size_t)r * sendcount + o, pr + (size_t)r * m, m * sizeof(float), hipMemcpyHostToDevice, st) !=
          hipSuccess)
        return 1;
  }
  return 0;
}

}  // extern "C"
