#!/bin/bash
# single-stream conv-family time per step (tools/conv_table.py, head lines only) under the dispatcher's env knobs:
# which tile / split-K thresholds fit the two-term fp16 split (they were tuned on the 3-term bf16 split)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/sweep
run() {  # name, env...
  name=$1; shift
  env "$@" ROWS=0 python mmt-psm_amd/tools/conv_table.py > gpurun_out/sweep/$name.txt 2>&1
  echo "$name: $(head -1 gpurun_out/sweep/$name.txt)"
}
run base MMT_NOP=1
run base2 MMT_NOP=1
run nkt64 MMT_SPLITK_NKT=64
run nkt32 MMT_SPLITK_NKT=32
run nkt256 MMT_SPLITK_NKT=256
run splitk0 MMT_SPLITK=0
run t128_128 MMT_T128=128
run t128_512 MMT_T128=512
run mid0 MMT_MID=0
run lowk128 MMT_LOWK=128
run wslots256 MMT_WGRAD_SLOTS=256
run wslots1024 MMT_WGRAD_SLOTS=1024
run wminpx256 MMT_WGRAD_MINPX=256
run wminpx1024 MMT_WGRAD_MINPX=1024
run rowsmin64 MMT_ROWS_MIN16=64
run rows0 MMT_ROWS=0
run glds4 MMT_GLDS_S=4
run stripsplitk0 MMT_STRIP_SPLITK=0
