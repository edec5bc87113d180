#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LAS_POISON=1 timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -q 2>&1 | tail -3
