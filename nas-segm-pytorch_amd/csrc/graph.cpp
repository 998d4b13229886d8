// Scheduling of a captured training step (include/nasseg.h, "hipGraph scheduling").
//
// engine/graphed.py records forward + loss + backward (+ optimisers) of a candidate from ONE stream, so the hipGraph
// it gets is a line: every kernel waits for the one recorded before it, although the five ops of a ContextualCell
// read the same input, the two cells of a MergeCell share nothing, and no weight gradient is read before the
// optimiser (reference src/nn/micro_decoders.py:54-139).  On the 11x11 ... 81x81 maps of the CVPR cells a dependent
// kernel costs >= 4.6 us whatever it does, ~900 of them per step.  The host side (engine/graph_dag.py) knows what
// every entry point read and wrote; the two functions here are the runtime half: counting the nodes a capture has
// recorded so far (which nodes belong to which call) and replacing the line's edges by the real dependencies.
// Host code only - no kernels.
#include <vector>
#include <unordered_map>
#include <hip/hip_runtime.h>

#include "common.h"

#define GRAPH_CHECK(call, what)                                                              \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError();                                                               \
      return nasseg_fail(NASSEG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_));          \
    }                                                                                        \
  } while (0)

// the nodes of a graph recorded from one stream, in the order they were recorded (= along the line); empty + error
// when the graph is not a line
static int line_order(hipGraph_t graph, std::vector<hipGraphNode_t>& order, std::vector<hipGraphNode_t>& from,
                      std::vector<hipGraphNode_t>& to) {
  size_t nn = 0, ne = 0, nr = 0;
  GRAPH_CHECK(hipGraphGetNodes(graph, nullptr, &nn), "hipGraphGetNodes");
  GRAPH_CHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne), "hipGraphGetEdges");
  from.resize(ne);
  to.resize(ne);
  if (ne) GRAPH_CHECK(hipGraphGetEdges(graph, from.data(), to.data(), &ne), "hipGraphGetEdges");
  GRAPH_CHECK(hipGraphGetRootNodes(graph, nullptr, &nr), "hipGraphGetRootNodes");
  order.clear();
  if (nn == 0) return NASSEG_OK;
  if (nr != 1 || ne + 1 != nn)
    return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "graph: not a line (%zu nodes, %zu edges, %zu roots)", nn, ne, nr);
  hipGraphNode_t root;
  GRAPH_CHECK(hipGraphGetRootNodes(graph, &root, &nr), "hipGraphGetRootNodes");
  std::unordered_map<hipGraphNode_t, hipGraphNode_t> next;
  next.reserve(ne * 2);
  for (size_t e = 0; e < ne; ++e) {
    if (!next.emplace(from[e], to[e]).second)
      return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "graph: not a line (a node with two successors)");
  }
  order.reserve(nn);
  hipGraphNode_t cur = root;
  order.push_back(cur);
  while (order.size() < nn) {
    auto it = next.find(cur);
    if (it == next.end()) return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "graph: not a line (it ends after %zu of %zu nodes)", order.size(), nn);
    cur = it->second;
    order.push_back(cur);
  }
  return NASSEG_OK;
}

extern "C" {

// Nodes recorded so far by the capture `stream` is part of; -1: the stream is not capturing.
int nasseg_graph_capture_nodes(void* stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  GRAPH_CHECK(hipStreamGetCaptureInfo_v2((hipStream_t)stream, &status, &id, &graph, &deps, &ndeps),
              "hipStreamGetCaptureInfo_v2");
  if (status != hipStreamCaptureStatusActive || graph == nullptr) return -1;
  size_t nn = 0;
  GRAPH_CHECK(hipGraphGetNodes(graph, nullptr, &nn), "hipGraphGetNodes");
  return (int)nn;
}

// kinds[i] of the i-th recorded node: 0 kernel, 1 memcpy, 2 memset, 3 anything else.  n_nodes must be the graph's.
int nasseg_graph_node_kinds(void* graph, int n_nodes, int* kinds) {
  std::vector<hipGraphNode_t> order, from, to;
  int rc = line_order((hipGraph_t)graph, order, from, to);
  if (rc != NASSEG_OK) return rc;
  NASSEG_REQUIRE((int)order.size() == n_nodes && kinds, "graph_node_kinds: the graph has %zu nodes, not %d", order.size(), n_nodes);
  for (int i = 0; i < n_nodes; ++i) {
    hipGraphNodeType t;
    GRAPH_CHECK(hipGraphNodeGetType(order[i], &t), "hipGraphNodeGetType");
    kinds[i] = t == hipGraphNodeTypeKernel ? 0 : t == hipGraphNodeTypeMemcpy ? 1 : t == hipGraphNodeTypeMemset ? 2 : 3;
  }
  return NASSEG_OK;
}

// Replace the edges of a graph that was recorded from one stream (a line of n_nodes nodes) by edges[2*e] ->
// edges[2*e+1], node indices in recording order, every edge pointing forward.  The caller guarantees that they
// cover every read-after-write, write-after-read and write-after-write pair of the recorded launches (engine/
// graph_dag.py derives them from the address ranges each entry point was handed); the nodes, their arguments and
// the memory they touch are left as recorded.
int nasseg_graph_rewire(void* graph, int n_nodes, int n_edges, const int* edges) {
  NASSEG_REQUIRE(graph && n_nodes >= 0 && n_edges >= 0 && (edges || n_edges == 0), "graph_rewire: bad arguments");
  hipGraph_t g = (hipGraph_t)graph;
  std::vector<hipGraphNode_t> order, from, to;
  int rc = line_order(g, order, from, to);
  if (rc != NASSEG_OK) return rc;
  NASSEG_REQUIRE((int)order.size() == n_nodes, "graph_rewire: the graph has %zu nodes, the caller counted %d", order.size(), n_nodes);
  std::vector<hipGraphNode_t> nf(n_edges), nt(n_edges);
  for (int e = 0; e < n_edges; ++e) {
    const int a = edges[2 * e], b = edges[2 * e + 1];
    NASSEG_REQUIRE(0 <= a && a < b && b < n_nodes, "graph_rewire: edge %d -> %d of %d nodes", a, b, n_nodes);
    nf[e] = order[a];
    nt[e] = order[b];
  }
  if (!from.empty()) GRAPH_CHECK(hipGraphRemoveDependencies(g, from.data(), to.data(), from.size()), "hipGraphRemoveDependencies");
  if (n_edges) {
    hipError_t e = hipGraphAddDependencies(g, nf.data(), nt.data(), (size_t)n_edges);
    if (e != hipSuccess) {
      // never leave the graph without its order: put the line back (the caller still gets the error, and must
      // drop the graph if THIS fails too - the message says so)
      (void)hipGetLastError();
      size_t ne = 0;
      bool clean = hipGraphGetEdges(g, nullptr, nullptr, &ne) == hipSuccess && ne == 0;
      bool restored = clean && (from.empty() || hipGraphAddDependencies(g, from.data(), to.data(), from.size()) == hipSuccess);
      return nasseg_fail(NASSEG_ERR_LAUNCH, "hipGraphAddDependencies: %s (%s)", hipGetErrorString(e),
                         restored ? "the recorded order was restored" : "GRAPH UNUSABLE: its order could not be restored");
    }
  }
  return NASSEG_OK;
}

// Cut a graph recorded from one stream into n_parts LINE graphs: part p holds the recorded nodes i with part[i] == p,
// in recording order, each waiting for the one before it - kernels-only lines replay from pre-built packets, and
// engine/graph_dag.py launches the parts of a stage side by side on different streams (nasseg_graph_run).  Nodes are
// re-created from their recorded parameters (kernel and memset nodes - a copy node recorded from hipMemcpyAsync
// cannot be re-created from what hipGraphMemcpyNodeGetParams returns on this runtime -; anything else: NASSEG_ERR_UNSUPPORTED and
// nothing is created); the recorded graph is left as it is and keeps owning nothing the parts need.
// execs[p] receives a hipGraphExec_t (0 for an empty part).
int nasseg_graph_split(void* graph, int n_nodes, const int* part, int n_parts, void** execs) {
  NASSEG_REQUIRE(graph && n_nodes > 0 && part && n_parts > 0 && execs, "graph_split: bad arguments");
  std::vector<hipGraphNode_t> order, from, to;
  int rc = line_order((hipGraph_t)graph, order, from, to);
  if (rc != NASSEG_OK) return rc;
  NASSEG_REQUIRE((int)order.size() == n_nodes, "graph_split: the graph has %zu nodes, the caller counted %d", order.size(), n_nodes);
  std::vector<hipGraphNodeType> kinds(n_nodes);
  for (int i = 0; i < n_nodes; ++i) {
    NASSEG_REQUIRE(0 <= part[i] && part[i] < n_parts, "graph_split: node %d in part %d of %d", i, part[i], n_parts);
    GRAPH_CHECK(hipGraphNodeGetType(order[i], &kinds[i]), "hipGraphNodeGetType");
    if (kinds[i] != hipGraphNodeTypeKernel && kinds[i] != hipGraphNodeTypeMemset)
      return nasseg_fail(NASSEG_ERR_UNSUPPORTED, "graph_split: node %d is neither a kernel nor a memset (type %d)", i, (int)kinds[i]);
  }
  std::vector<hipGraph_t> graphs(n_parts, nullptr);
  std::vector<hipGraphNode_t> tail(n_parts, nullptr);
  std::vector<hipGraphExec_t> made(n_parts, nullptr);
  hipError_t e = hipSuccess;
  const char* what = "";
  for (int i = 0; i < n_nodes && e == hipSuccess; ++i) {
    const int p = part[i];
    if (!graphs[p]) {
      e = hipGraphCreate(&graphs[p], 0);
      what = "hipGraphCreate";
      if (e != hipSuccess) break;
    }
    hipGraphNode_t node = nullptr;
    const hipGraphNode_t* deps = tail[p] ? &tail[p] : nullptr;
    const size_t ndeps = tail[p] ? 1 : 0;
    if (kinds[i] == hipGraphNodeTypeKernel) {
      hipKernelNodeParams kp;
      e = hipGraphKernelNodeGetParams(order[i], &kp);
      what = "hipGraphKernelNodeGetParams";
      if (e != hipSuccess) break;
      e = hipGraphAddKernelNode(&node, graphs[p], deps, ndeps, &kp);
      what = "hipGraphAddKernelNode";
    } else {
      hipMemsetParams mp;
      e = hipGraphMemsetNodeGetParams(order[i], &mp);
      what = "hipGraphMemsetNodeGetParams";
      if (e != hipSuccess) break;
      e = hipGraphAddMemsetNode(&node, graphs[p], deps, ndeps, &mp);
      what = "hipGraphAddMemsetNode";
    }
    tail[p] = node;
  }
  for (int p = 0; p < n_parts && e == hipSuccess; ++p) {
    if (!graphs[p]) continue;
    e = hipGraphInstantiate(&made[p], graphs[p], nullptr, nullptr, 0);
    what = "hipGraphInstantiate";
  }
  for (int p = 0; p < n_parts; ++p)
    if (graphs[p]) (void)hipGraphDestroy(graphs[p]);  // (an executable graph does not need its template)
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (int p = 0; p < n_parts; ++p)
      if (made[p]) (void)hipGraphExecDestroy(made[p]);
    return nasseg_fail(NASSEG_ERR_LAUNCH, "graph_split: %s: %s", what, hipGetErrorString(e));
  }
  for (int p = 0; p < n_parts; ++p) execs[p] = (void*)made[p];
  return NASSEG_OK;
}

int nasseg_graph_exec_destroy(void* exec) {
  if (exec) GRAPH_CHECK(hipGraphExecDestroy((hipGraphExec_t)exec), "hipGraphExecDestroy");
  return NASSEG_OK;
}

// a stream / an event for the lanes of nasseg_graph_run (non-blocking stream; event without timing)
int nasseg_lane_stream_create(void** stream) {
  NASSEG_REQUIRE(stream, "lane_stream_create: null");
  hipStream_t s;
  GRAPH_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  *stream = (void*)s;
  return NASSEG_OK;
}
int nasseg_lane_event_create(void** event) {
  NASSEG_REQUIRE(event, "lane_event_create: null");
  hipEvent_t ev;
  GRAPH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreateWithFlags");
  *event = (void*)ev;
  return NASSEG_OK;
}
int nasseg_lane_destroy(void* stream, void* event) {
  if (event) GRAPH_CHECK(hipEventDestroy((hipEvent_t)event), "hipEventDestroy");
  if (stream) GRAPH_CHECK(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy");
  return NASSEG_OK;
}

// One replay of a laid-out step: ops[3*i..] = {kind, a, b}, executed in order from the calling thread -
//   0: launch the executable graph a on stream b      1: record event a on stream b      2: stream b waits for event a
// (b == 0: `stream`, the stream the step belongs to).  Nothing here blocks: the GPU orders the lanes by the events.
int nasseg_graph_run(int n_ops, const int64_t* ops, void* stream) {
  NASSEG_REQUIRE(n_ops >= 0 && (ops || n_ops == 0), "graph_run: bad arguments");
  for (int i = 0; i < n_ops; ++i) {
    const int64_t kind = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
    hipStream_t s = b ? (hipStream_t)(uintptr_t)b : (hipStream_t)stream;
    switch (kind) {
      case 0: GRAPH_CHECK(hipGraphLaunch((hipGraphExec_t)(uintptr_t)a, s), "hipGraphLaunch"); break;
      case 1: GRAPH_CHECK(hipEventRecord((hipEvent_t)(uintptr_t)a, s), "hipEventRecord"); break;
      case 2: GRAPH_CHECK(hipStreamWaitEvent(s, (hipEvent_t)(uintptr_t)a, 0), "hipStreamWaitEvent"); break;
      default: return nasseg_fail(NASSEG_ERR_ARG, "graph_run: op %d has kind %lld", i, (long long)kind);
    }
  }
  return NASSEG_OK;
}

}  // extern "C"
