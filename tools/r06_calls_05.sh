#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn_onepass_bench.py --S 2432 --rounds 6 --libs a31,a63,a95,a159,a255,a31slp 2>&1 | tail -1
