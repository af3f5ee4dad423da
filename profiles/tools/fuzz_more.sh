#!/bin/bash
# The randomised GPU tests with OTHER seeds than the ones the suite runs (ILUPP_FUZZ_OFFSET shifts every generator):
#   bash profiles/tools/fuzz_more.sh OFFSET [OFFSET ...]      (on the GPU box; a few minutes per offset)
for off in "$@"; do
  echo "== offset $off"
  ILUPP_FUZZ_OFFSET=$off timeout 900 python3 -m pytest -q -x -m gpu \
    tests/test_gpu_ml.py::test_fuzz_against_oracle tests/test_gpu_mlp.py::test_fuzz_against_the_oracle \
    tests/test_gpu_ilucp.py::test_fuzz_against_the_oracle tests/test_gpu_ilutp.py::test_fuzz_against_the_oracle \
    tests/test_gpu_parity.py::test_fuzz_new_kernels tests/test_gpu_level_sweeps.py::test_fuzz_level_order \
    tests/test_gpu_wa.py::test_random_box_shapes_predicted_sizes_and_edge_tiles 2>&1 | grep -v amdgpu.ids | tail -15
done
