#!/bin/bash
bash tools/mb_variants.sh "base:" "p8w4:-DMB_P=8 -DCP_WAVES_PER_SIMD=4" "p8w4ns:-DMB_P=8 -DCP_WAVES_PER_SIMD=4 -DCP_ROW_SCREEN=0" "p8w4bar:-DMB_P=8 -DCP_WAVES_PER_SIMD=4 -DCP_WAVE_LOCAL=0" "p8w3:-DMB_P=8 -DCP_WAVES_PER_SIMD=3 -DMB_WGS_PER_CU=1" 2>&1 | tee gpurun_out/exp_p8.txt
