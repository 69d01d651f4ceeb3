// Launch-plan replay: the kernels (and memset / memcpy operations) of ONE stream's piece of the train step, recorded once
// and re-issued from a single C call.
//
// Recording uses HIP's own stream capture as the recorder: the caller captures the piece into a hipGraph (which also catches
// the launches that do not come from this library -- torch's fill / copy / add kernels), and s2i_plan_create walks that graph
// in dependency order and keeps, per node, what a plain launch needs: function, grid, block, dynamic LDS and the argument
// pointers (owned by the graph, which the caller keeps alive together with the memory pool the capture allocated from).
// s2i_plan_replay then issues hipLaunchKernel / hipMemsetAsync / hipMemcpyAsync in that order on the stream it is given.
//
// Why not hipGraphLaunch: on ROCm 7.2 replaying these graphs costs more than the Python host path it was meant to replace
// (38.6 vs 31.6 ms per fp32 step, 23.4 vs 17.2 ms in the bf16 mode, one graph per stream piece: profiles/r03_graph_replay.md);
// a plain launch from C costs ~3 us of host time, and pieces on different streams stay concurrent because each
// is launched on its own stream with ordinary stream waits between them (the reference's loop body, trainer.py:536-572, has
// no such structure: this is host-side machinery of the MI355X build).
#include "s2i_common.h"
#include <string.h>
#include <vector>
#include <algorithm>
#include <queue>

namespace {

struct PlanNode {
  hipGraphNodeType type;
  hipKernelNodeParams k;
  hipMemsetParams ms;
  hipMemcpy3DParms mc;
};

struct Plan {
  std::vector<PlanNode> nodes;
  int kernels = 0, memsets = 0, memcpys = 0;
};

#define S2I_HIP(call, what)                                                            \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) S2I_FAIL("%s: %s", what, hipGetErrorString(e__));           \
  } while (0)

}  // namespace

extern "C" int s2i_plan_create(void* hip_graph, void** plan_out, int* counts /* [kernels, memsets, memcpys] or null */) {
  S2I_REQUIRE(hip_graph && plan_out, "plan_create: null argument");
  hipGraph_t g = (hipGraph_t)hip_graph;
  size_t n = 0;
  S2I_HIP(hipGraphGetNodes(g, nullptr, &n), "plan_create: hipGraphGetNodes");
  std::vector<hipGraphNode_t> nodes(n);
  if (n) S2I_HIP(hipGraphGetNodes(g, nodes.data(), &n), "plan_create: hipGraphGetNodes");
  size_t ne = 0;
  S2I_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne), "plan_create: hipGraphGetEdges");
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne) S2I_HIP(hipGraphGetEdges(g, from.data(), to.data(), &ne), "plan_create: hipGraphGetEdges");
  // dependency order (Kahn), ties broken by the node's position in the graph's own list: a captured single-stream
  // piece is a chain, so this is simply its launch order
  std::vector<std::pair<hipGraphNode_t, int>> by_handle(n);
  for (size_t i = 0; i < n; ++i) by_handle[i] = {nodes[i], (int)i};
  std::sort(by_handle.begin(), by_handle.end());
  auto index_of = [&](hipGraphNode_t x) -> int {
    auto it = std::lower_bound(by_handle.begin(), by_handle.end(), std::make_pair(x, -1));
    return (it != by_handle.end() && it->first == x) ? it->second : -1;
  };
  std::vector<std::vector<int>> succ(n);
  std::vector<int> indeg(n, 0);
  for (size_t e = 0; e < ne; ++e) {
    const int a = index_of(from[e]), b = index_of(to[e]);
    S2I_REQUIRE(a >= 0 && b >= 0, "plan_create: an edge names a node outside the graph");
    succ[a].push_back(b);
    ++indeg[b];
  }
  std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
  for (size_t i = 0; i < n; ++i)
    if (indeg[i] == 0) ready.push((int)i);
  Plan* pl = new Plan();
  size_t done = 0;
  while (!ready.empty()) {
    const int i = ready.top();
    ready.pop();
    ++done;
    PlanNode pn;
    memset((void*)&pn, 0, sizeof(pn));
    hipError_t e = hipGraphNodeGetType(nodes[i], &pn.type);
    if (e == hipSuccess) {
      if (pn.type == hipGraphNodeTypeKernel) {
        e = hipGraphKernelNodeGetParams(nodes[i], &pn.k);
        if (e == hipSuccess && (pn.k.kernelParams == nullptr || pn.k.func == nullptr)) {
          delete pl;
          S2I_FAIL("plan_create: kernel node %d carries no argument-pointer list (launched through `extra`)", i);
        }
        ++pl->kernels;
      } else if (pn.type == hipGraphNodeTypeMemset) {
        e = hipGraphMemsetNodeGetParams(nodes[i], &pn.ms);
        if (e == hipSuccess && pn.ms.height > 1) { delete pl; S2I_FAIL("plan_create: 2-D memset node"); }
        ++pl->memsets;
      } else if (pn.type == hipGraphNodeTypeMemcpy) {
        e = hipGraphMemcpyNodeGetParams(nodes[i], &pn.mc);
        // a captured hipMemcpyAsync reports only its extent and the two pointers reliably (the pitch / position / array
        // fields of the 3-D descriptor come back uninitialised on ROCm 7.2): accept linear copies only
        if (e == hipSuccess && (pn.mc.extent.height > 1 || pn.mc.extent.depth > 1)) {
          delete pl;
          S2I_FAIL("plan_create: memcpy node %d is not a linear copy (%zu x %zu x %zu)", i, pn.mc.extent.width,
                   pn.mc.extent.height, pn.mc.extent.depth);
        }
        ++pl->memcpys;
      } else if (pn.type != hipGraphNodeTypeEmpty) {
        delete pl;
        S2I_FAIL("plan_create: node %d has type %d (only kernel / memset / memcpy / empty nodes can be re-issued)", i, (int)pn.type);
      }
    }
    if (e != hipSuccess) {
      delete pl;
      S2I_FAIL("plan_create: node %d: %s", i, hipGetErrorString(e));
    }
    if (pn.type != hipGraphNodeTypeEmpty) pl->nodes.push_back(pn);
    for (int b : succ[i])
      if (--indeg[b] == 0) ready.push(b);
  }
  if (done != n) {
    delete pl;
    S2I_FAIL("plan_create: the graph has a cycle (%zu of %zu nodes ordered)", done, n);
  }
  if (counts) { counts[0] = pl->kernels; counts[1] = pl->memsets; counts[2] = pl->memcpys; }
  *plan_out = pl;
  return 0;
}

extern "C" int s2i_plan_replay(void* plan, void* stream) {
  S2I_REQUIRE(plan, "plan_replay: null plan");
  const Plan* pl = (const Plan*)plan;
  hipStream_t st = (hipStream_t)stream;
  for (const PlanNode& pn : pl->nodes) {
    hipError_t e;
    if (pn.type == hipGraphNodeTypeKernel) {
      e = hipLaunchKernel(pn.k.func, pn.k.gridDim, pn.k.blockDim, pn.k.kernelParams, pn.k.sharedMemBytes, st);
    } else if (pn.type == hipGraphNodeTypeMemset) {
      const size_t count = pn.ms.width;
      if (pn.ms.elementSize == 4) e = hipMemsetD32Async((hipDeviceptr_t)pn.ms.dst, (int)pn.ms.value, count, st);
      else if (pn.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)pn.ms.dst, (unsigned short)pn.ms.value, count, st);
      else e = hipMemsetAsync(pn.ms.dst, (int)pn.ms.value, count, st);
    } else {
      e = hipMemcpyAsync(pn.mc.dstPtr.ptr, pn.mc.srcPtr.ptr, pn.mc.extent.width, hipMemcpyDefault, st);
    }
    if (e != hipSuccess) S2I_FAIL("plan_replay: %s", hipGetErrorString(e));
  }
  return 0;
}

extern "C" int s2i_plan_destroy(void* plan) {
  delete (Plan*)plan;
  return 0;
}
