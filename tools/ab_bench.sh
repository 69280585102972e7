#!/bin/bash
# A/B several builds of libctrefine.so on the same device, interleaved.
# Put the builds to compare into tools/_ab/lib_<name>.so (e.g. git stash; make; cp ...), then run this
# on the GPU box.  Perf deltas are only trusted from such interleaved runs on ONE device.
for rep in 1 2 3; do
  for lib in tools/_ab/lib_*.so; do
    r=$(CTREFINE_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")
    echo "$lib $r"
  done
done
