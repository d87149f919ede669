#!/bin/bash
# does a stream-role merge (two roles of the step on ONE stream object) reproduce the hipStreamEndCapture crash in a single process?
for m in hw hr wr all; do
  echo "== UNIT_STREAM_MERGE=$m"
  UNIT_STREAM_MERGE=$m PYTHONFAULTHANDLER=1 timeout 300 python -m pytest tests/test_graph_gpu.py -x -q -k "equals_eager" 2>&1 | tail -4
done
