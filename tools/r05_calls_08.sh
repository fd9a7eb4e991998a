#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/attn64_check.py --S 2432,3584,4608,8576,1280,640:24:2,4864,2432:24:2 --time 1 2>&1 | cut -c1-80
