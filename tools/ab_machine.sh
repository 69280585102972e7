#!/bin/bash
# A/B of builds in tools/_ab/lib_*.so by machine time per size class (tools/class_machine_time.py)
# and by the bench, interleaved on one device.
for rep in 1 2; do
  for lib in tools/_ab/lib_*.so; do
    echo "== $lib"
    CTREFINE_LIB=$PWD/$lib timeout -k 10 300 python tools/class_machine_time.py 2>/dev/null | tail -6 | cut -c1-60
    CTREFINE_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
  done
done
