#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python tools/g16_tables.py 4096 2>&1 | grep "^plain"; }
run ZL_TUNE_G2_TABLE_C=16
run ZL_TUNE_G2_TABLE_C=17
run ZL_TUNE_G2_TABLE_C=18
run ZL_TUNE_G1_TABLE_C=19
run ZL_TUNE_G1_TABLE_C=21
run ZL_TUNE_G2_TABLE_C=16
