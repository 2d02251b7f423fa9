#!/bin/bash
# build and run the C++ host-mirror GPU test by hand (debugging aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$ROOT/orcvio_amd/lib
ORC=$ROOT/oracle
g++ -std=c++17 -O1 -Wall -o /tmp/test_host_gpu $ROOT/tests/cpp/test_host_gpu.cpp -L $LIB -lorcvio_msckf -Wl,-rpath,$LIB -L $ORC -lorcoracle -Wl,-rpath,$ORC -lm
NCCL_DEBUG=INFO timeout 60 stdbuf -o0 -e0 /tmp/test_host_gpu 2>&1 | tail -60
echo "exit: $?"
