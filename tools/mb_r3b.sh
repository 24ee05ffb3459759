#!/bin/bash
# round 3, second batch: pass-0 twiddles pinned in the spare registers (with nt row accesses), padded LDS layout
NT="-DCP_ROW_LOAD_AUX=2 -DCP_ROW_STORE_AUX=2"
bash tools/mb_variants.sh "nt:$NT" "pin3:$NT -DCP_PIN_TW0=3" "pin4:$NT -DCP_PIN_TW0=4" "pin6:$NT -DCP_PIN_TW0=6" "pin7:$NT -DCP_PIN_TW0=7" "pin8:$NT -DCP_PIN_TW0=8" "pad:$NT -DCP_PADDED_LDS=1" "padpin6:$NT -DCP_PADDED_LDS=1 -DCP_PIN_TW0=6"
