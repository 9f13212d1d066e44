// Launch-graph hygiene for the recorded solves of torchsde_amd/graph.py (the reference has no launch graphs; this is
// the runtime around base_solver.py:114-134's loop once that loop is a HIP graph).
//
// A captured solve consists of kernel nodes -- and, whenever the user's code (or autograd's parameter-gradient sums)
// runs a multi-block torch reduction, of MEMSET nodes: ATen zeroes the reduction's block-counting semaphores with
// hipMemsetAsync before every launch. On this runtime (HIP 7.0 as bundled with torch 2.10, graph packet capture on) a
// recorded memset node stops doing its work once an eager memset has been issued and the host has synchronised between
// two replays: the reductions of every later replay are wrong (tools/probe_graph_reduction4.py,
// profiles/r3q_probe_graph_memset_nodes.txt). Kernel nodes are not affected. So before a recorded graph is
// instantiated, each of its memset nodes is replaced by a kernel node that fills the same bytes, with the same edges.
#include <stdio.h>

#include <vector>

#include "tsde_common.h"
#include "tsde_launch.h"

namespace tsde {

// `width` elements of `elem` bytes (1, 2 or 4) at `dst`, each set to the low bytes of `value`.
__global__ void __launch_bounds__(kBlock) graph_fill_kernel(unsigned char* dst, unsigned int value, unsigned int elem,
                                                            size_t width) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < width; i += (size_t)gridDim.x * kBlock) {
    if (elem == 1) dst[i] = (unsigned char)value;
    else if (elem == 2) ((unsigned short*)dst)[i] = (unsigned short)value;
    else ((unsigned int*)dst)[i] = value;
  }
}

// Returns hipSuccess or the first error; counts what it found and what it replaced (2-D memsets are left alone).
hipError_t memset_nodes_to_kernels(hipGraph_t graph, int* n_memset, int* n_replaced) {
  *n_memset = *n_replaced = 0;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess || n == 0) return e;
  std::vector<hipGraphNode_t> nodes(n);
  if ((e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return e;
  for (hipGraphNode_t node : nodes) {
    hipGraphNodeType type;
    if ((e = hipGraphNodeGetType(node, &type)) != hipSuccess) return e;
    if (type != hipGraphNodeTypeMemset) continue;
    ++*n_memset;
    hipMemsetParams p;
    if ((e = hipGraphMemsetNodeGetParams(node, &p)) != hipSuccess) return e;
    if (p.height > 1 || (p.elementSize != 1 && p.elementSize != 2 && p.elementSize != 4)) continue;
    size_t n_before = 0, n_after = 0;
    if ((e = hipGraphNodeGetDependencies(node, nullptr, &n_before)) != hipSuccess) return e;
    std::vector<hipGraphNode_t> before(n_before);
    if (n_before && (e = hipGraphNodeGetDependencies(node, before.data(), &n_before)) != hipSuccess) return e;
    if ((e = hipGraphNodeGetDependentNodes(node, nullptr, &n_after)) != hipSuccess) return e;
    std::vector<hipGraphNode_t> after(n_after);
    if (n_after && (e = hipGraphNodeGetDependentNodes(node, after.data(), &n_after)) != hipSuccess) return e;

    unsigned char* dst = (unsigned char*)p.dst;
    unsigned int value = p.value, elem = p.elementSize;
    size_t width = p.width;
    void* args[] = {&dst, &value, &elem, &width};
    size_t blocks = (width + kBlock - 1) / kBlock;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipKernelNodeParams k = {};
    k.func = (void*)graph_fill_kernel;
    k.gridDim = dim3((unsigned)blocks);
    k.blockDim = dim3(kBlock);
    k.sharedMemBytes = 0;
    k.kernelParams = args;
    k.extra = nullptr;
    hipGraphNode_t fill;
    if ((e = hipGraphAddKernelNode(&fill, graph, before.data(), n_before, &k)) != hipSuccess) return e;
    for (hipGraphNode_t later : after)
      if ((e = hipGraphAddDependencies(graph, &fill, &later, 1)) != hipSuccess) return e;
    if ((e = hipGraphDestroyNode(node)) != hipSuccess) return e;
    ++*n_replaced;
  }
  return hipSuccess;
}

}  // namespace tsde
