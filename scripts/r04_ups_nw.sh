#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for nw in 16 8 4; do echo "== UPS_LAB_NW=$nw"; UPS_LAB_NW=$nw ./scripts/lab/ups_lab 2>&1 | grep -E "^both|tokens after" | head -8; done
