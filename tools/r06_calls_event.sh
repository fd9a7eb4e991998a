#!/bin/bash
# round 6: fork / join events of the side-stream launches without the system-scope release (QFX_EVENT_NOFENCE=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_plan_ab.py base,QFX_EVENT_NOFENCE=1 --steps 20 --rounds 3 --out gpurun_out/r06_event_nofence.json 2>&1 | tail -4
QFX_EVENT_NOFENCE=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fulldepth_gpu.py -x -q 2>&1 | tail -3
