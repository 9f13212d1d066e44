#!/bin/bash
# tools/probe_graph_reduction3.py under the HIP runtime's graph flags, one process each (the flags are read at start-up).
out=gpurun_out/r3q; mkdir -p $out
{
  timeout 300 python tools/probe_graph_reduction3.py 2>&1 | grep -v amdgpu
  echo
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python tools/probe_graph_reduction3.py 2>&1 | grep -v amdgpu
  echo
  DEBUG_HIP_FORCE_GRAPH_QUEUES=1 timeout 300 python tools/probe_graph_reduction3.py 2>&1 | grep -v amdgpu
} > $out/probe_graph_flags.txt
cat $out/probe_graph_flags.txt
