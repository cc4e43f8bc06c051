#!/bin/bash
# GPU box: run a command once per library variant in gpurun_scratch/ (TENSOIR_HIP_LIB) and once with the default lib
CMD="$*"
echo "== default"; bash -c "$CMD" 2>&1 | tail -${TAILN:-3}
for so in gpurun_scratch/lib_*.so; do echo "== $so"; TENSOIR_HIP_LIB=$PWD/$so bash -c "$CMD" 2>&1 | tail -${TAILN:-3}; done
