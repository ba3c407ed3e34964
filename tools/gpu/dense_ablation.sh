#!/bin/bash
# timing ablations of the dense product kernel (library built with `make ABLATION=1`): WRONG results by design
export FORMS=1 REPS=3
for d in 0 1 2 3 4 8 16 20 28; do
  echo "EAP_DENSE_DEBUG=$d: $(EAP_DENSE_DEBUG=$d python tools/gpu/dense_time.py 2>&1 | grep 'backward' | sed 's/.*product/product/')"
done
