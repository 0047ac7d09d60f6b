#!/bin/bash
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/coop_primitives tools/coop_primitives.hip && timeout -k 5 60 /tmp/coop_primitives
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/fma_latency tools/fma_latency.hip && timeout -k 5 60 /tmp/fma_latency
