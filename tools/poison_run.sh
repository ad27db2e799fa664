#!/bin/bash
# the GPU tests with every new device allocation filled with NaNs (HDM_POISON=1, csrc/alloc.cpp): a kernel that reads memory nobody
# has written turns its results into NaNs and the test that sees them names the path.  Not part of the suite (the at-size tests
# would spend minutes in 100 GB memsets): every device path at small sizes.
#   tools/poison_run.sh            the parity, kernel, ingest and streaming files
#   tools/poison_run.sh all        also the device group, the process-per-GPU rehearsal, the switches and the reference's driver
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
FILES="tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_ingest.py tests/test_gpu_streamed.py"
if [ "$1" = "all" ]; then shift; FILES="$FILES tests/test_gpu_group.py tests/test_gpu_dist.py tests/test_gpu_switches.py tests/test_gpu_reference_driver.py"; fi
HDM_POISON=1 python3 -m pytest $FILES -q -m gpu \
  -k "not full_size and not config5 and not headline and not beyond_int32 and not at_full_dimension" "$@"
