#!/bin/bash
# round 3, first batch of headline variants: cache policy of the row accesses, sum-based screening
bash tools/mb_variants.sh "base:" "ntl:-DCP_ROW_LOAD_AUX=2" "nts:-DCP_ROW_STORE_AUX=2" "ntls:-DCP_ROW_LOAD_AUX=2 -DCP_ROW_STORE_AUX=2" "sc0s:-DCP_ROW_STORE_AUX=1" "sum:-DCP_SCREEN_SUM=1" "sumnt:-DCP_SCREEN_SUM=1 -DCP_ROW_LOAD_AUX=2 -DCP_ROW_STORE_AUX=2" "nosc:-DCP_ROW_SCREEN=0"
