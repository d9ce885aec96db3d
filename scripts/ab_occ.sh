#!/bin/bash
B=cuda-volpath_amd/build
run() { echo -n "$1 blocks=$2 cfg=$3: "; VOLPATH_LIB=$1 VP_BLOCKS_PER_CU=$2 timeout -k 10 120 python scripts/prof_case.py $3 64 | tail -1 || exit 1; }
D=cuda-volpath_amd/libvolpath_hip.so
run $D 5 "0 1 1"; run $D 5 "0 1 1"
run $B/ab_w6/libvolpath_hip.so 6 "0 1 1"; run $B/ab_w6/libvolpath_hip.so 6 "0 1 1"
run $B/ab_w6/libvolpath_hip.so 5 "0 1 1"
run $D 5 "1 1 1"; run $D 4 "1 1 1"
run $B/ab_w6/libvolpath_hip.so 6 "1 1 1"; run $B/ab_w6/libvolpath_hip.so 5 "1 1 1"
run $D 5 "1 8 1"; run $D 5 "1 8 1"
run $B/ab_l640w5/libvolpath_hip.so 5 "1 8 1"; run $B/ab_l640w5/libvolpath_hip.so 5 "1 8 1"
run $B/ab_l768w6/libvolpath_hip.so 5 "1 8 1"; run $B/ab_l768w6/libvolpath_hip.so 5 "1 8 1"
