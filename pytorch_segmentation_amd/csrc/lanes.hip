// Lane executor: replays a captured hipGraph as plain launches on a few HIP streams.
//
// Why: the launch-bound configurations of the reference (BASELINE configs[1] UNet 256x256 B=8, configs[4] HRNet-W32
// 512x512 B=8: 500-1000 kernels of 5-40 us per training step) spend their step in the HOST -- ~15 us of Python per
// launch.  hipGraphLaunch removes the host cost only for single-stream graphs: for a captured step that forks its weight
// gradients onto a second stream the graph launch itself costs 12-24 ms on the host (measured, ROCm 7.2), and the
// single-stream graph gives up the overlap (HRNet: 22.6 ms against 19.6 ms eager on two streams).
// What the captured graph DOES hold is everything a replay needs: every kernel with its launch geometry and argument
// block, every memset, and the dependency edges (device-to-device copy nodes cannot be read back: such graphs are refused).  This file walks the graph once (hipGraphGetNodes, node params,
// dependencies), assigns every node to one of a few LANES (a lane = a chain of nodes in stream order; a node continues the
// lane of a parent that is still that lane's tail, preferring the parent that has no other child), turns the edges
// that cross lanes into events, and then replays the step as a tight loop of hipLaunchKernel / hipMemsetAsync /
// hipEventRecord / hipStreamWaitEvent calls: ~2 us of host time per node, real concurrency between
// lanes, no Python, no hipGraphExec.  The graph object (and with it the argument blocks and the private memory pool of
// the capture) must stay alive as long as the executor.
//
// Results are those of the captured launches in a dependency-respecting order: bit-identical to eager execution.
#include "common.h"

#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/pseg_amd.h"

namespace pseg {

constexpr int kMaxLanes = 8;

struct LaneNode {
  hipGraphNodeType type;
  hipKernelNodeParams kp;
  hipMemsetParams ms;
  int lane;
  int record;                 // event recorded after this node (-1: none)
  int marker;                 // index into LaneExec::marks of the marker this memset node stands for (-1: none)
  std::vector<int> waits;     // events this node's lane waits for before it
};

// A MARKER is a one-word memset into a buffer the caller named (pseg_lanes_bind_markers): the point of the step where
// "everything enqueued so far on this stream" matters to somebody outside the executor -- e.g. the last weight gradient of
// a gradient bucket.  The replay records an event there; pseg_lanes_wait_marker makes another stream wait for it.
struct LaneMark {
  int id;
  hipEvent_t ev;
};

struct LaneExec {
  std::vector<LaneNode> nodes;
  std::vector<hipEvent_t> events;
  std::vector<hipStream_t> own_streams;   // lanes >= 1: streams of the process-wide lane pool (never destroyed)
  std::vector<hipEvent_t> lane_done;      // end-of-step marker per lane >= 1
  std::vector<LaneMark> marks;
  hipEvent_t begin;
  int lanes;
  int launches;   // nodes that launch something
  int device;     // the device the executor was built on (its pool streams and events belong to it)
};

#define PSEG_HIP_TRY(expr)                                                                        \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      set_error("lanes: %s failed: %s", #expr, hipGetErrorString(e_));                            \
      return PSEG_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

// The lanes of EVERY executor of a device run on one small pool of streams that lives as long as the process.  Streams are
// not free: the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues in the order of
// their first use, two busy lanes that share a queue serialise (a waiting lane blocks the main chain behind it), and more
// than four busy queues (GPU_MAX_HW_QUEUES=8) fall off a cliff (HRNet -mp 18 ms/step against 8).  A pool that is created --
// and touched -- early (pseg_lanes_reserve, from the Trainer's constructor) gets queues of its own next to the caller's
// stream, whatever streams the process creates later; streams per executor would land wherever the round-robin stood at the
// time of each capture.  (Running the lanes on the streams the capture itself forked onto -- torch's -- was tried and is
// gone: a later hipGraphLaunch of an unrelated forked graph crashed inside the runtime, reproducibly in the full test suite.)
// The pool is process state shared by every Trainer (one per device thread is a legal use): one mutex guards it, it is keyed
// by the device ordinal (no aliasing of large ordinals), and callers take a COPY of the stream handles they use.
static std::mutex& lane_pool_mutex() {
  static std::mutex m;
  return m;
}

static std::map<int, std::vector<hipStream_t>>& lane_pools() {
  static std::map<int, std::vector<hipStream_t>> pools;
  return pools;
}

// -> the first n streams of the current device's pool in `out` (created and touched on demand)
static int lane_pool_reserve(int n, std::vector<hipStream_t>* out, int* device_out) {
  int device = -1, count = 0;
  PSEG_HIP_TRY(hipGetDevice(&device));
  PSEG_HIP_TRY(hipGetDeviceCount(&count));
  PSEG_REQUIRE(device >= 0 && device < count, "lanes: current device %d is outside 0..%d", device, count - 1);
  std::lock_guard<std::mutex> lock(lane_pool_mutex());
  std::vector<hipStream_t>& pool = lane_pools()[device];
  while ((int)pool.size() < n) {
    hipStream_t s;
    PSEG_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // first use binds the hardware queue: an event record is the cheapest command there is
    hipEvent_t ev;
    PSEG_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    PSEG_HIP_TRY(hipEventRecord(ev, s));
    PSEG_HIP_TRY(hipStreamSynchronize(s));
    PSEG_HIP_TRY(hipEventDestroy(ev));
    pool.push_back(s);
  }
  if (out != nullptr) out->assign(pool.begin(), pool.begin() + n);
  if (device_out != nullptr) *device_out = device;
  return PSEG_OK;
}

static int lanes_build(hipGraph_t graph, int max_lanes, LaneExec*& out) {
  size_t n = 0;
  PSEG_HIP_TRY(hipGraphGetNodes(graph, nullptr, &n));
  PSEG_REQUIRE(n > 0 && n < (1u << 24), "lanes: empty or oversized graph (%zu nodes)", n);
  std::vector<hipGraphNode_t> gn(n);
  PSEG_HIP_TRY(hipGraphGetNodes(graph, gn.data(), &n));
  // node handle -> index (graphs of a training step have ~1e3 nodes: sort + bisect)
  std::vector<std::pair<hipGraphNode_t, int>> index(n);
  for (size_t i = 0; i < n; ++i) index[i] = {gn[i], (int)i};
  std::sort(index.begin(), index.end());
  auto find = [&](hipGraphNode_t h) -> int {
    size_t lo = 0, hi = n;
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (index[mid].first < h) lo = mid + 1;
      else hi = mid;
    }
    return (lo < n && index[lo].first == h) ? index[lo].second : -1;
  };
  std::vector<std::vector<int>> parents(n);
  std::vector<int> outdeg(n, 0);
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    PSEG_HIP_TRY(hipGraphNodeGetDependencies(gn[i], nullptr, &nd));
    if (nd == 0) continue;
    std::vector<hipGraphNode_t> deps(nd);
    PSEG_HIP_TRY(hipGraphNodeGetDependencies(gn[i], deps.data(), &nd));
    for (size_t d = 0; d < nd; ++d) {
      const int p = find(deps[d]);
      PSEG_REQUIRE(p >= 0, "lanes: dependency outside the graph");
      parents[i].push_back(p);
      ++outdeg[p];
    }
  }
  // topological order (Kahn, smallest creation index first: stream capture creates nodes in enqueue order)
  std::vector<int> order;
  order.reserve(n);
  {
    std::vector<int> missing(n);
    std::vector<std::vector<int>> children(n);
    for (size_t i = 0; i < n; ++i) {
      missing[i] = (int)parents[i].size();
      for (int p : parents[i]) children[p].push_back((int)i);
    }
    std::vector<int> ready;
    for (size_t i = 0; i < n; ++i)
      if (missing[i] == 0) ready.push_back((int)i);
    std::make_heap(ready.begin(), ready.end(), std::greater<int>());
    while (!ready.empty()) {
      std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
      const int v = ready.back();
      ready.pop_back();
      order.push_back(v);
      for (int c : children[v])
        if (--missing[c] == 0) {
          ready.push_back(c);
          std::push_heap(ready.begin(), ready.end(), std::greater<int>());
        }
    }
    PSEG_REQUIRE(order.size() == n, "lanes: the graph has a cycle");
  }

  LaneExec* ex = new LaneExec();
  ex->nodes.resize(n);
  ex->lanes = 1;
  ex->launches = 0;
  std::vector<int> slot(n, -1);         // graph index -> position in ex->nodes
  std::vector<int> lane_tail;           // graph index of the last node of each lane
  std::vector<std::array<int, kMaxLanes>> clock(n);
  lane_tail.push_back(-1);
  int rc = PSEG_OK;
  for (size_t pos = 0; pos < n && rc == PSEG_OK; ++pos) {
    const int v = order[pos];
    LaneNode& nd = ex->nodes[pos];
    slot[v] = (int)pos;
    nd.record = -1;
    nd.marker = -1;
    hipError_t e = hipGraphNodeGetType(gn[v], &nd.type);
    if (e != hipSuccess) {
      set_error("lanes: hipGraphNodeGetType: %s", hipGetErrorString(e));
      rc = PSEG_ERR_HIP;
      break;
    }
    if (nd.type == hipGraphNodeTypeKernel) {
      e = hipGraphKernelNodeGetParams(gn[v], &nd.kp);
      // (a node launched through `extra` -- hipModuleLaunchKernel style, no kernelParams array -- cannot be re-issued with
      // hipLaunchKernel: refused, the caller replays such a graph with hipGraphLaunch)
      if (e != hipSuccess || nd.kp.func == nullptr || nd.kp.kernelParams == nullptr) {
        set_error("lanes: kernel node %d has no replayable parameter array (%s)", v, hipGetErrorString(e));
        rc = PSEG_ERR_ARG;
        break;
      }
      ++ex->launches;
    } else if (nd.type == hipGraphNodeTypeMemset) {
      e = hipGraphMemsetNodeGetParams(gn[v], &nd.ms);
      if (e != hipSuccess || nd.ms.height > 1) {
        set_error("lanes: memset node %d is not a 1-D memset", v);
        rc = PSEG_ERR_ARG;
        break;
      }
      ++ex->launches;
    } else if (nd.type == hipGraphNodeTypeMemcpy) {
      // A captured hipMemcpyAsync is a 1-D memcpy node, and hipGraphMemcpyNodeGetParams hands back uninitialised 3-D
      // parameters for those (ROCm 7.2: status hipSuccess, garbage extents) -- there is no way to replay it faithfully, and
      // no way to tell it from a genuine 3-D node.  Refuse the graph; the caller falls back to hipGraphLaunch.
      set_error("lanes: node %d is a memcpy node (their parameters cannot be read back reliably); keep device-to-device copies "
                "out of the captured step or replay it with hipGraphLaunch", v);
      rc = PSEG_ERR_ARG;
      break;
    } else if (nd.type != hipGraphNodeTypeEmpty) {
      set_error("lanes: node %d has type %d (only kernel / memset / memcpy / empty nodes can be replayed)", v, (int)nd.type);
      rc = PSEG_ERR_ARG;
      break;
    }
    // lane: continue the lane of a parent that is still its lane's tail; of several, the one with the fewest children
    // (its lane would end otherwise), then the lowest lane.  No such parent: an IDLE lane -- one whose tail is already an
    // ancestor of this node (a branch that has been joined: queueing behind it adds no constraint the graph does not
    // have; without this a model that forks and joins its branches module after module -- HRNet -- would open a lane per
    // fork and serialise everything on lane 0 once they run out) -- else open a lane while there is one, else join the lane
    // of the last parent.  seen[l] = the latest position on lane l that is an ancestor of this node (a vector clock).
    std::array<int, kMaxLanes>& seen = clock[pos];
    seen.fill(-1);
    for (int p : parents[v]) {
      const std::array<int, kMaxLanes>& ps = clock[slot[p]];
      for (int l = 0; l < kMaxLanes; ++l) seen[l] = std::max(seen[l], ps[l]);
    }
    int best = -1;
    for (int p : parents[v]) {
      const int pl = ex->nodes[slot[p]].lane;
      if (lane_tail[pl] != p) continue;
      if (best < 0 || outdeg[p] < outdeg[best] || (outdeg[p] == outdeg[best] && pl < ex->nodes[slot[best]].lane)) best = p;
    }
    int idle = -1;
    if (best < 0 && !parents[v].empty())
      for (int l = 1; l < ex->lanes && idle < 0; ++l)      // (never lane 0: the caller's stream carries the main chain)
        if (lane_tail[l] >= 0 && slot[lane_tail[l]] <= seen[l]) idle = l;
    if (best >= 0) {
      nd.lane = ex->nodes[slot[best]].lane;
    } else if (parents[v].empty()) {
      nd.lane = 0;
    } else if (idle >= 0) {
      nd.lane = idle;
    } else if (ex->lanes < max_lanes) {
      nd.lane = ex->lanes++;
      lane_tail.push_back(-1);
    } else {
      nd.lane = ex->nodes[slot[parents[v].back()]].lane;
    }
    lane_tail[nd.lane] = v;
    seen[nd.lane] = (int)pos;
    for (int p : parents[v]) {
      LaneNode& pn = ex->nodes[slot[p]];
      if (pn.lane == nd.lane) continue;     // stream order
      if (pn.record < 0) {
        pn.record = (int)ex->events.size();
        ex->events.push_back(nullptr);
      }
      nd.waits.push_back(pn.record);
    }
  }
  if (rc != PSEG_OK) {
    delete ex;
    return rc;
  }
  // events and streams of the executor; on a failure everything created so far is released with the executor
  ex->begin = nullptr;
  auto fail = [&](const char* what, hipError_t err) {
    set_error("lanes: %s failed: %s", what, hipGetErrorString(err));
    for (hipEvent_t ev : ex->events)
      if (ev != nullptr) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ex->lane_done) (void)hipEventDestroy(ev);
    if (ex->begin != nullptr) (void)hipEventDestroy(ex->begin);
    delete ex;
    return PSEG_ERR_HIP;
  };
  hipError_t he;
  for (auto& ev : ex->events)
    if ((he = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
      ev = nullptr;
      return fail("hipEventCreateWithFlags", he);
    }
  if ((he = hipEventCreateWithFlags(&ex->begin, hipEventDisableTiming)) != hipSuccess) {
    ex->begin = nullptr;
    return fail("hipEventCreateWithFlags", he);
  }
  if ((he = hipGetDevice(&ex->device)) != hipSuccess) return fail("hipGetDevice", he);
  if (ex->lanes > 1) {
    if (lane_pool_reserve(ex->lanes - 1, &ex->own_streams, &ex->device) != PSEG_OK) {
      // (keep lane_pool_reserve's own message)
      for (hipEvent_t ev : ex->events)
        if (ev != nullptr) (void)hipEventDestroy(ev);
      (void)hipEventDestroy(ex->begin);
      delete ex;
      return PSEG_ERR_HIP;
    }
    for (int l = 1; l < ex->lanes; ++l) {
      hipEvent_t d;
      if ((he = hipEventCreateWithFlags(&d, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreateWithFlags", he);
      ex->lane_done.push_back(d);
    }
  }
  out = ex;
  return PSEG_OK;
}

// An edge to another lane is an event recorded after its source node.  hipEventRecord puts a marker packet of its own
// into the lane's queue, and the lane's NEXT kernel waits for that packet: ~10 us of bubble per record on the recording
// lane (HRNet 512x512 B=8: 97 records on the serial backward chain = 1 ms of a 10 ms step, measured from the kernel trace).
// hipExtLaunchKernel binds the event to the kernel's own completion signal instead: no extra packet, no bubble.
// PSEG_LANES_STOP_EVENT=0 goes back to hipEventRecord (A/B).
static bool stop_event_launch() {
  static const bool on = [] {
    const char* e = getenv("PSEG_LANES_STOP_EVENT");
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

static int lanes_launch(LaneExec* ex, hipStream_t main) {
  const bool bind = stop_event_launch();
  // every lane starts after what the caller has enqueued so far (the input copies), the caller's stream ends after every lane
  if (ex->lanes > 1) {
    PSEG_HIP_TRY(hipEventRecord(ex->begin, main));
    for (hipStream_t s : ex->own_streams) PSEG_HIP_TRY(hipStreamWaitEvent(s, ex->begin, 0));
  }
  for (LaneNode& nd : ex->nodes) {
    hipStream_t s = nd.lane == 0 ? main : ex->own_streams[nd.lane - 1];
    for (int w : nd.waits) PSEG_HIP_TRY(hipStreamWaitEvent(s, ex->events[w], 0));
    bool recorded = false;
    if (nd.type == hipGraphNodeTypeKernel) {
      if (bind && nd.record >= 0) {
        PSEG_HIP_TRY(hipExtLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, s,
                                        nullptr, ex->events[nd.record], 0));
        recorded = true;
      } else {
        PSEG_HIP_TRY(hipLaunchKernel(nd.kp.func, nd.kp.gridDim, nd.kp.blockDim, nd.kp.kernelParams, nd.kp.sharedMemBytes, s));
      }
    } else if (nd.type == hipGraphNodeTypeMemset) {
      const size_t count = nd.ms.width;
      if (nd.ms.elementSize == 4) PSEG_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)nd.ms.dst, (int)nd.ms.value, count, s));
      else if (nd.ms.elementSize == 2) PSEG_HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)nd.ms.dst, (unsigned short)nd.ms.value, count, s));
      else PSEG_HIP_TRY(hipMemsetD8Async((hipDeviceptr_t)nd.ms.dst, (unsigned char)nd.ms.value, count, s));
    }
    if (nd.record >= 0 && !recorded) PSEG_HIP_TRY(hipEventRecord(ex->events[nd.record], s));
    if (nd.marker >= 0) PSEG_HIP_TRY(hipEventRecord(ex->marks[nd.marker].ev, s));
  }
  for (int l = 1; l < ex->lanes; ++l) {
    PSEG_HIP_TRY(hipEventRecord(ex->lane_done[l - 1], ex->own_streams[l - 1]));
    PSEG_HIP_TRY(hipStreamWaitEvent(main, ex->lane_done[l - 1], 0));
  }
  return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_lanes_build(void* hip_graph, int max_lanes, int64_t* handle) {
  PSEG_REQUIRE(hip_graph != nullptr && handle != nullptr && max_lanes >= 1 && max_lanes <= kMaxLanes, "lanes_build: bad argument");
  LaneExec* ex = nullptr;
  const int rc = lanes_build((hipGraph_t)hip_graph, max_lanes, ex);
  if (rc != PSEG_OK) return rc;
  *handle = (int64_t)(intptr_t)ex;
  return PSEG_OK;
}

int pseg_lanes_info(int64_t handle, int* nodes, int* launches, int* lanes, int* events) {
  PSEG_REQUIRE(handle != 0, "lanes_info: null handle");
  LaneExec* ex = (LaneExec*)(intptr_t)handle;
  if (nodes) *nodes = (int)ex->nodes.size();
  if (launches) *launches = ex->launches;
  if (lanes) *lanes = ex->lanes;
  if (events) *events = (int)ex->events.size();
  return PSEG_OK;
}

int pseg_lanes_reserve(int lanes) {
  PSEG_REQUIRE(lanes >= 1 && lanes <= kMaxLanes, "lanes_reserve: 1..%d lanes", kMaxLanes);
  return lane_pool_reserve(lanes - 1, nullptr, nullptr);
}

int pseg_lanes_launch(int64_t handle, void* stream) {
  PSEG_REQUIRE(handle != 0, "lanes_launch: null handle");
  return lanes_launch((LaneExec*)(intptr_t)handle, (hipStream_t)stream);
}

int pseg_mark(void* word, void* stream) {
  PSEG_REQUIRE(word != nullptr && ((uintptr_t)word & 3) == 0, "mark: null or misaligned marker word");
  PSEG_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)word, 0, 1, (hipStream_t)stream));
  return PSEG_OK;
}

int pseg_lanes_bind_markers(int64_t handle, const void* base, int count, int* bound) {
  PSEG_REQUIRE(handle != 0 && base != nullptr && count > 0, "lanes_bind_markers: bad argument");
  LaneExec* ex = (LaneExec*)(intptr_t)handle;
  PSEG_REQUIRE(ex->marks.empty(), "lanes_bind_markers: markers are bound once per executor");
  const uintptr_t lo = (uintptr_t)base, hi = lo + 4 * (uintptr_t)count;
  for (LaneNode& nd : ex->nodes) {
    if (nd.type != hipGraphNodeTypeMemset || nd.ms.elementSize != 4 || nd.ms.width != 1) continue;
    const uintptr_t d = (uintptr_t)nd.ms.dst;
    if (d < lo || d >= hi || ((d - lo) & 3) != 0) continue;
    hipEvent_t ev;
    PSEG_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    nd.marker = (int)ex->marks.size();
    ex->marks.push_back(LaneMark{(int)((d - lo) / 4), ev});
  }
  if (bound) *bound = (int)ex->marks.size();
  return PSEG_OK;
}

int pseg_lanes_wait_marker(int64_t handle, int id, void* stream) {
  PSEG_REQUIRE(handle != 0, "lanes_wait_marker: null handle");
  LaneExec* ex = (LaneExec*)(intptr_t)handle;
  int n = 0;
  for (const LaneMark& m : ex->marks)
    if (m.id == id) {
      PSEG_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, m.ev, 0));
      ++n;
    }
  PSEG_REQUIRE(n > 0, "lanes_wait_marker: the replayed step holds no marker %d", id);
  return PSEG_OK;
}

int pseg_lanes_destroy(int64_t handle) {
  if (handle == 0) return PSEG_OK;
  LaneExec* ex = (LaneExec*)(intptr_t)handle;
  // Every event of the executor is bound to a command of some replay: cross-lane events to the completion signal of a kernel
  // (hipExtLaunchKernel) -- on the pool streams AND on the caller's stream, which carries lane 0 --, markers to whatever
  // stream an exchange waited on.  Nothing of that may be pending when the events go: the whole device is drained first (not
  // only the pool streams, as up to round 4).  Illegal while a stream capture is open anywhere: then nothing is released and
  // the caller tries again later (utils/trainer.py keeps the handle).
  int current = -1;
  (void)hipGetDevice(&current);
  if (current != ex->device) PSEG_HIP_TRY(hipSetDevice(ex->device));
  hipError_t drained = hipDeviceSynchronize();
  if (current != ex->device && current >= 0) (void)hipSetDevice(current);
  if (drained != hipSuccess) {
    set_error("lanes_destroy: hipDeviceSynchronize failed (%s): executor kept", hipGetErrorString(drained));
    return PSEG_ERR_HIP;
  }
  for (const LaneMark& m : ex->marks) (void)hipEventDestroy(m.ev);
  for (hipEvent_t e : ex->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : ex->lane_done) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ex->begin);
  delete ex;
  return PSEG_OK;
}

}  // extern "C"
